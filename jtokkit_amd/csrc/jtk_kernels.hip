// jtk_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the batch BPE encode path.
//
// Stage           kernel            replaces (reference, lib/src/main/java/com/knuddels/jtokkit/)
// mark_docs       k_mark_docs       document boundaries of the batch (one Encoding.encode call each)
// validate_utf8   k_validate_utf8   (optional) String.getBytes(UTF_8) well-formedness per document
// pretok_split    k_pretok_split    GptBytePairEncoding.java:77-80 matcher.find()/group() with EncodingFactory.java:63,105;
//                                   for encode(): the special-token check :52-56 (text.contains(specialToken))
// piece_resolve   k_piece_resolve   :81-83 whole-piece shortcut (TokenEncoder lookups) + queueing of the other pieces
// bpe_merge       k_bpe_merge       :84-86 + bytePairMerge :200-275 + getRank :285-300   (last phase: giant pieces > 8 KiB)
// pack            k_tile_scan, k_pack_tokens, k_doc_offsets   out.add / addAll (:82,:117):
//                                   the document-order token stream and per-document offsets
//
// Integer / byte work only; no floating point, no MFMA.  One lane per 64-byte block in pretok_split, one lane per
// piece in piece_resolve and bpe_merge (one wave per piece for long pieces, wave-level leftmost-min), one wave per
// tile in pack.
#include "jtk_kernels.h"

#include <type_traits>

#include "jtk_merge_core.h"
#include "jtk_block_classify.h"
#include "jtk_split_masks.h"
#include "jtk_split_rules.h"

namespace {

constexpr int WAVE = 64;

__device__ __forceinline__ uint64_t lanemask_lt() {
    const unsigned lane = threadIdx.x & 63u;
    return (1ull << lane) - 1ull;
}

// LDS accesses of different lanes of ONE wave, ordered without a workgroup barrier.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    // butterfly over the 64 lanes; every lane ends with the minimum
    v = min(v, (uint32_t)__shfl_xor((int)v, 1));
    v = min(v, (uint32_t)__shfl_xor((int)v, 2));
    v = min(v, (uint32_t)__shfl_xor((int)v, 4));
    v = min(v, (uint32_t)__shfl_xor((int)v, 8));
    v = min(v, (uint32_t)__shfl_xor((int)v, 16));
    v = min(v, (uint32_t)__shfl_xor((int)v, 32));
    return v;
}

// inclusive prefix sum across the wave
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const unsigned lane = threadIdx.x & 63u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d);
        if (lane >= (unsigned)d) v += o;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------
// mark_docs: docmask bit for every doc_off[d], d = 0..n_docs (the last one is the end sentinel)
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mark_docs(JtkWork w) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > w.n_docs) return;
    const int64_t q = w.doc_off[d] - w.text_base;                // position in this chunk
    // the offsets are caller memory: out-of-range or decreasing ones are reported (JTK_ERR_INVALID_ARGUMENT as the
    // batch's worst status), never followed
    const bool bad = q < w.lead || q > w.n_bytes || (d > 0 && w.doc_off[d - 1] - w.text_base > q) || (d == 0 && q != w.lead) ||
                     (d == w.n_docs && q != w.n_bytes);
    if (bad) { atomicMin(&w.result->worst_status, -1 /* JTK_ERR_INVALID_ARGUMENT */); return; }
    atomicOr((unsigned long long*)&w.docmask[q >> 6], 1ull << (q & 63));
}

// index of the document containing byte position p (skipping empty documents)
__device__ int64_t find_doc(const JtkWork& w, int64_t p) {
    const int64_t* doc_off = w.doc_off;
    const int64_t n_docs = w.n_docs;
    p += w.text_base;                            // doc_off holds positions in the whole batch
    int64_t lo = 0, hi = n_docs;                 // first d with doc_off[d] > p
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (doc_off[mid] > p) hi = mid; else lo = mid + 1;
    }
    return lo - 1;
}

// ---------------------------------------------------------------------------------------------------
// special_check: flag documents that contain a special-token literal (GptBytePairEncoding.java:52-56: text.contains).
// The exact test at one position; pretok_split calls it for the bytes it sees that some literal starts with.
// ---------------------------------------------------------------------------------------------------
__device__ void special_check_at(const JtkWork& w, const JtkDeviceTables& t, int64_t p) {
    if (p < 0 || p >= w.n_bytes) return;
    for (int s = 0; s < t.n_specials; s++) {
        const uint32_t o = t.special_off[s];
        const int len = (int)(t.special_off[s + 1] - o);
        if (p + len > w.n_bytes) continue;
        bool eq = true;
        for (int j = 0; j < len && eq; j++) eq = (w.text[p + j] == t.special_blob[o + j]);
        if (!eq) continue;
        const int64_t d = find_doc(w, p);
        if (d >= 0 && p + len <= w.doc_off[d + 1] - w.text_base) atomicMin(&w.status[d], -2 /* JTK_ERR_UNSUPPORTED_SPECIAL */);
    }
}

// ---------------------------------------------------------------------------------------------------
// validate_utf8 (optional): every document must be what String.getBytes(UTF_8) can produce -- shortest
// form, no surrogates, <= U+10FFFF, no sequence cut by the document's end.  One lane per byte: lead bytes
// check their sequence, continuation bytes check that a lead covers them.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_validate_utf8(JtkWork w) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= w.n_bytes) return;
    auto ds = [&](int64_t q) { return q >= w.n_bytes || ((w.docmask[q >> 6] >> (q & 63)) & 1ull) != 0; };
    auto at = [&](int64_t q) -> uint32_t { return (q >= 0 && q < w.n_bytes) ? w.text[q] : 0u; };
    const uint32_t b = w.text[p];
    bool bad = false;
    if (b < 0x80u) return;
    if ((b & 0xC0u) == 0x80u) {
        // find the lead (at most 3 back, same document) and check it is long enough to cover this byte
        bad = true;
        int64_t q = p;
        for (int k = 1; k <= 3; k++) {
            if (ds(q)) break;                       // p..q start a document: no lead before them
            q--;
            const uint32_t c = at(q);
            if ((c & 0xC0u) == 0x80u) continue;
            const int len = c >= 0xF0u ? 4 : (c >= 0xE0u ? 3 : (c >= 0xC2u ? 2 : 0));
            bad = !(c >= 0xC2u && c <= 0xF4u && len > k);
            break;
        }
    } else {
        const int len = b >= 0xF0u ? 4 : (b >= 0xE0u ? 3 : 2);
        if (b < 0xC2u || b > 0xF4u) bad = true;
        for (int k = 1; k < len && !bad; k++) {
            const uint32_t c = at(p + k);
            if (ds(p + k) || (c & 0xC0u) != 0x80u) bad = true;
        }
        if (!bad) {
            const uint32_t c1 = at(p + 1);
            if (b == 0xE0u && c1 < 0xA0u) bad = true;          // overlong
            if (b == 0xEDu && c1 > 0x9Fu) bad = true;          // surrogates
            if (b == 0xF0u && c1 < 0x90u) bad = true;          // overlong
            if (b == 0xF4u && c1 > 0x8Fu) bad = true;          // > U+10FFFF
        }
    }
    if (bad) {
        const int64_t d = find_doc(w, p);
        if (d >= 0) atomicMin(&w.status[d], -6 /* JTK_ERR_BAD_UTF8 */);
    }
}

// ---------------------------------------------------------------------------------------------------
// pretok_split: ONE LANE PER 64-BYTE BLOCK.  A lane loads its block (4 x 16 B), classifies it through a
// 256-entry table of 16 flags per byte value in LDS (shift-or accumulation, jtk_block_classify.h), and evaluates
// the split rules for the whole block as 64-bit mask algebra (jtk_split_masks.h).  Block-to-block carries
// (digit-run phase, swallowed CR/LF chains, ...) are exchanged with __shfl_up and iterated to a fixed
// point: one or two rounds unless a run spans several blocks.  Lanes 0 and 63 of a wave are halo
// blocks, so a wave emits 62 mask words with one coalesced store.
// ---------------------------------------------------------------------------------------------------
struct GlobalText {
    const uint8_t* t; int64_t n;
    __device__ uint32_t byte(int64_t p) const { return (p >= 0 && p < n) ? t[p] : 0u; }
};

constexpr int SPLIT_THREADS = 512;                     // the tables in LDS (24 KB) are shared by 8 waves

// The 4 bytes that start at byte j of the lane's block, from the LDS copy [dword][lane]; row 16 is the next lane's first dword.
struct LdsBlockWords {
    const uint32_t* blk; int tid;
    __device__ uint32_t word(int j) const {
        const uint32_t a = (uint32_t)j >> 2;
        const uint32_t lo = blk[a * SPLIT_THREADS + (uint32_t)tid], hi = blk[(a + 1u) * SPLIT_THREADS + (uint32_t)tid];
        return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)j & 3u);
    }
};
struct GlobalBlockWords {
    const uint8_t* t; int64_t n, p0;
    __device__ uint32_t word(int j) const {
        uint32_t v = 0;
        for (int r = 0; r < 4; r++) { const int64_t p = p0 + j + r; if (p >= 0 && p < n) v |= (uint32_t)t[p] << (8 * r); }
        return v;
    }
};

// Unbounded window over global memory for the rare positions that need a run walk (exact).
struct SlowWin {
    typedef int64_t idx_t;
    static constexpr int kMaxWalk = 0;
    const uint8_t* gtext; int64_t n; const uint64_t* docmask; JtkUcTables uc;
    __device__ uint32_t byte(int64_t p) const { return (p >= 0 && p < n) ? gtext[p] : 0u; }
    __device__ uint32_t cb(int64_t p) const {
        if (p >= n || p < 0) return JTK_CB_DS;
        GlobalText g{gtext, n};
        uint32_t c = jtk_class_byte(g, uc, p);
        if ((docmask[p >> 6] >> (p & 63)) & 1ull) c |= JTK_CB_DS;
        return c;
    }
};

constexpr int SPW = 62;                                // blocks a wave emits
constexpr int SPLIT_BYTES = (SPLIT_THREADS / 64) * SPW * 64;   // bytes per workgroup

__device__ __forceinline__ uint64_t hi_from_prev_lane(uint64_t v) {      // only the top bits are consumed
    return (uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(v >> 32), 1) << 32;
}
__device__ __forceinline__ uint64_t lo_from_next_lane(uint64_t v) {      // only the low bits are consumed
    return (uint64_t)(uint32_t)__shfl_down((int)(uint32_t)v, 1);
}

template <int KIND>
__global__ void __launch_bounds__(SPLIT_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) k_pretok_split(JtkWork w, JtkDeviceTables t) {
    __shared__ __attribute__((aligned(16))) JtkCode4 s_tab[256];   // per-byte flags, laid out for shift-or accumulation (jtk_block_classify.h)
    __shared__ uint32_t s_pin[2048];           // byte pairs that occur inside some table entry
    __shared__ uint32_t s_blk[17 * SPLIT_THREADS];   // the lanes' blocks, [dword][lane] (staged only for text outside ASCII)
    __shared__ __attribute__((aligned(4))) uint8_t s_uc1[JTK_UC_LDS_STAGE1];
    __shared__ uint32_t s_uc2[JTK_UC_LDS_STAGE2];
    __shared__ uint32_t s_uc_ready;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t B = (int64_t)blockIdx.x * SPLIT_BYTES;
    const int64_t n = w.n_bytes;
    if (tid < 256) {
        uint32_t code = jtk_byte_code((uint32_t)tid, KIND == JTK_PAT_CL100K);
        if ((t.special_first[tid >> 5] >> (tid & 31)) & 1u) code |= JTK_F_LT;
        if ((t.lead_letters[tid >> 5] >> (tid & 31)) & 1u) code |= JTK_F_ULL;
        s_tab[tid] = jtk_code4(code);
    }
    if (tid == 0) s_uc_ready = 0;
    for (int i = tid; i < 2048; i += SPLIT_THREADS) s_pin[i] = t.pair_in_token[i];
    __syncthreads();

    // ---- this lane's block
    // (persistent workgroups -- a wave looping over spans -- were tried: the same speed, and 62 spilled registers)
    const int64_t p0 = B - 64 + (int64_t)(wv * SPW + lane) * 64;
    uint32_t d[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int64_t p = p0 + 16 * q;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (p >= 0 && p + 16 <= n) v = *reinterpret_cast<const uint4*>(w.text + p);
        else if (p >= 0 && p < n) {
            uint32_t tmp[4] = {0, 0, 0, 0};
            for (int r = 0; r < 16; r++) if (p + r < n) tmp[r >> 2] |= (uint32_t)w.text[p + r] << (8 * (r & 3));
            v = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
        }
        d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
    }
    JtkBlk cu;
    uint64_t lead, lt, ull;
    jtk_block_masks_ascii(d, s_tab, cu, lead, lt, ull);
    // encode(): the special-token check of GptBytePairEncoding.java:52-56 rides along -- every '<' of the lanes that emit
    if (w.check_special && lane >= 1 && lane <= SPW) {
        for (uint64_t m = lt; m;) {
            const int j = jtk_ctz64(m);
            m &= m - 1;
            special_check_at(w, t, p0 + j);
        }
    }
    {
        // Characters outside ASCII are decoded and classified one by one.  Only workgroups that have any stage what
        // that needs in LDS -- the Unicode class table (12 KB) and each lane's 64 bytes -- so that the per-character
        // chain is LDS reads, not global loads (the kernel runs at 2 waves per SIMD: nothing hides a global latency).
        uint32_t spill = JTK_CLS_O;
        const bool lds_ok = t.uc_stage1_len <= JTK_UC_LDS_STAGE1 && t.uc_stage2_words <= JTK_UC_LDS_STAGE2;
        // leads of characters that are letters whatever follows (CJK ideographs, Hangul, ...) need no decode: `ull`
        if (__ballot((lead & ~ull) != 0)) {                           // per wave: no workgroup barrier on the ASCII path
            if (lds_ok) {
                // the first wave that needs the table copies it; a wave that does not see the flag yet copies it again
                // (same values: a benign race), so no barrier is needed
                if (*(volatile uint32_t*)&s_uc_ready == 0u) {
                    for (uint32_t i = lane; i < t.uc_stage1_len / 4; i += 64) reinterpret_cast<uint32_t*>(s_uc1)[i] = reinterpret_cast<const uint32_t*>(t.uc.stage1)[i];
                    for (uint32_t i = lane; i < t.uc_stage2_words; i += 64) s_uc2[i] = t.uc.stage2[i];
                }
#pragma unroll
                for (int q = 0; q < 16; q++) s_blk[q * SPLIT_THREADS + tid] = d[q];
                s_blk[16 * SPLIT_THREADS + tid] = (uint32_t)__shfl_down((int)d[0], 1);   // lane 63: its own (a halo block: only its low bits are used)
                wave_lds_fence();
                if (lane == 0) *(volatile uint32_t*)&s_uc_ready = 1u;
                const LdsBlockWords bw{s_blk, tid};
                const JtkUcTables ucl{s_uc1, s_uc2};
                jtk_block_fix_nonascii(bw, ucl, lead, ull, cu, spill);
            } else {
                const GlobalBlockWords bw{w.text, n, p0};
                jtk_block_fix_nonascii(bw, t.uc, lead, ull, cu, spill);
            }
        } else if (__ballot(lead != 0)) {
            struct NoWords { __device__ uint32_t word(int) const { return 0u; } };
            jtk_block_fix_nonascii(NoWords{}, t.uc, lead, ull, cu, spill);   // no lane of the wave decodes anything
        }
        uint32_t prev_spill = (uint32_t)__shfl_up((int)spill, 1);
        if (lane == 0) prev_spill = JTK_CLS_O;
        jtk_block_apply_spill(cu, prev_spill);
    }
    {   // document starts; every position >= n counts as one
        const int64_t wd = p0 >> 6;
        uint64_t ds = (p0 >= 0 && wd < w.n_words) ? w.docmask[wd] : 0ull;
        if (p0 + 63 >= n) ds |= (p0 >= n) ? ~0ull : ~((1ull << (n - p0)) - 1ull);
        cu.DS = ds;
    }
    uint32_t prev_byte = (uint32_t)__shfl_up((int)(d[15] >> 24), 1);          // the byte before this block
    if (lane == 0) prev_byte = (p0 > 0 && p0 <= n) ? w.text[p0 - 1] : 0u;

    // ---- the split rules for the whole block
    JtkBlk nx;
    nx.L = nx.N = nx.SP = nx.AP = nx.S1 = nx.RV = nx.C5 = nx.BF = 0;
    nx.W = lo_from_next_lane(cu.W); nx.DS = lo_from_next_lane(cu.DS); nx.CONT = lo_from_next_lane(cu.CONT); nx.NL = lo_from_next_lane(cu.NL);
    nx.E = lo_from_next_lane(cu.E); nx.LL = lo_from_next_lane(cu.LL);
    JtkSplitCarry base;
    base.pL = hi_from_prev_lane(cu.L);   base.pN = hi_from_prev_lane(cu.N);   base.pW = hi_from_prev_lane(cu.W);
    base.pNL = hi_from_prev_lane(cu.NL); base.pSP = hi_from_prev_lane(cu.SP); base.pDS = hi_from_prev_lane(cu.DS);
    base.pCONT = hi_from_prev_lane(cu.CONT);
    base.pS1 = hi_from_prev_lane(cu.S1); base.pRV = hi_from_prev_lane(cu.RV); base.pE = hi_from_prev_lane(cu.E);
    base.pLL = hi_from_prev_lane(cu.LL); base.pC5 = hi_from_prev_lane(cu.C5); base.pBF = hi_from_prev_lane(cu.BF);

    uint64_t oX = 0, oAP = 0, oSW = 0;            // what this lane hands to the next one
    uint32_t oN = 0, oFlags = 0;                  // bit0 n_unknown, bit1 sw_unknown
    uint64_t ms = 0, slow = 0, nlanes = 0;
    uint32_t ncnt_in = 0;
    bool nunk_in = false;
    for (int round = 0; round < 66; round++) {
        JtkSplitCarry cy = base;
        cy.pX = hi_from_prev_lane(oX);
        cy.pMsAP = hi_from_prev_lane(oAP);
        cy.pSW = hi_from_prev_lane(oSW);
        cy.ncnt = (uint32_t)__shfl_up((int)oN, 1);
        const uint32_t fl = (uint32_t)__shfl_up((int)oFlags, 1);
        cy.n_unknown = (fl & 1u) != 0;
        cy.sw_unknown = (fl & 2u) != 0;
        if (lane == 0) { cy.pX = cy.pMsAP = cy.pSW = 0; cy.ncnt = 0; cy.n_unknown = true; cy.sw_unknown = true; }
        ncnt_in = cy.ncnt;
        nunk_in = cy.n_unknown;
        ms = jtk_split_block<KIND>(cu, nx, cy, slow, nlanes);
        const uint32_t nfl = (cy.n_unknown ? 1u : 0u) | (cy.sw_unknown ? 2u : 0u);
        const bool changed = ((cy.pX ^ oX) >> 61) != 0 || ((cy.pMsAP ^ oAP) >> 61) != 0 || ((cy.pSW ^ oSW) >> 63) != 0
                             || cy.ncnt != oN || nfl != oFlags;
        oX = cy.pX; oAP = cy.pMsAP; oSW = cy.pSW; oN = cy.ncnt; oFlags = nfl;
        if (!__ballot(changed)) break;
    }

    // ---- cl100k digit runs (mask algebra unless the block has digits of several bytes), and the rare slow positions
    if (KIND == JTK_PAT_CL100K && __ballot(nlanes != 0)) {
        if ((cu.N & cu.CONT) == 0) {
            uint64_t nslow;
            ms |= jtk_split_n_block(cu, base.pN, ncnt_in, nunk_in, nslow);
            slow |= nslow;
            nlanes = 0;
        }
        for (uint64_t m = nlanes; m;) {
            const int j = jtk_ctz64(m);
            m &= m - 1;
            bool s2 = false;
            const bool v = jtk_split_n_lane(cu, ncnt_in, nunk_in, j, s2);
            if (s2) slow |= 1ull << j;
            else if (v) ms |= 1ull << j;
        }
    }
    for (uint64_t m = slow; m;) {
        const int j = jtk_ctz64(m);
        m &= m - 1;
        const SlowWin sw{w.text, n, w.docmask, t.uc};
        bool dummy = false;
        const bool v = jtk_is_piece_start_t<KIND>(sw, p0 + j, dummy);
        ms = v ? (ms | (1ull << j)) : (ms & ~(1ull << j));
    }

    // Extra cuts inside regex pieces: where two bytes never occur next to each other inside any table entry no
    // merge can cross (every part is a table entry), so bytePairMerge of the piece equals the concatenation of
    // bytePairMerge of the two sides.  Long CJK runs fall apart into a few bytes each.  Any subset of these cuts
    // is exact, so blocks of short ASCII pieces (ordinary text) skip the 64 bitmap lookups.
    uint64_t cut = 0;
    if ((lead | cu.CONT) != 0 || __popcll(ms) < 6) {
        // The block's bytes are read again here (a cache hit) rather than kept in 16 registers through the rules above.
        uint32_t dd[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t p = p0 + 16 * q;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (p >= 0 && p + 16 <= n) v = *reinterpret_cast<const uint4*>(w.text + p);
            else if (p >= 0 && p < n) {
                uint32_t tmp[4] = {0, 0, 0, 0};
                for (int r = 0; r < 16; r++) if (p + r < n) tmp[r >> 2] |= (uint32_t)w.text[p + r] << (8 * (r & 3));
                v = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
            }
            dd[4 * q] = v.x; dd[4 * q + 1] = v.y; dd[4 * q + 2] = v.z; dd[4 * q + 3] = v.w;
        }
        // pair (byte j-1, byte j) as an index: one byte permute; the bit accumulates from the top (alignbit), byte 0 ends at bit 0
        uint32_t acc[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < 64; j++) {
            const int q = j >> 2, r = j & 3;
            const uint32_t hi = dd[q], lo = (j >= 4) ? dd[q - 1] : (prev_byte << 24);
            // index = prev << 8 | cur: byte 0 <- cur = hi byte r (selector 4 + r), byte 1 <- prev = hi byte r-1 or lo byte 3
            const uint32_t sel = 0x0C0C0000u | (uint32_t)(r ? (4 + r - 1) : 3) << 8 | (uint32_t)(4 + r);
            const uint32_t pi = __builtin_amdgcn_perm(hi, lo, sel);
            const uint32_t in = s_pin[pi >> 5] >> (pi & 31u);
            acc[j >> 5] = __builtin_amdgcn_alignbit(in, acc[j >> 5], 1);
        }
        cut = ~(((uint64_t)acc[1] << 32) | acc[0]);
    }

    if (lane >= 1 && lane <= SPW) {
        uint64_t valid = 0;                                           // positions <= n
        if (p0 + 63 <= n) valid = ~0ull;
        else if (p0 <= n) valid = (2ull << (n - p0)) - 1ull;
        const int64_t wd = p0 >> 6;
        if (wd < w.n_words) w.piecemask[wd] = (ms | cut) & valid;
    }
}

// ---------------------------------------------------------------------------------------------------
// piece_resolve: piece list of a tile; pieces of <= 8 bytes that are table entries become that one
// token (GptBytePairEncoding.java:81-83); every other piece is queued for bytePairMerge by length.
// Writes the tile's piece list (plist): one word per piece, in text order.
// ---------------------------------------------------------------------------------------------------
constexpr int T = JTK_TILE;
constexpr int TW = T / 64;                                          // mask words per tile
static_assert(TW <= 64, "tile scans assume at most 64 mask words");

// this tile's queued pieces by bin, in LDS until its slices of the queues are claimed
__host__ __device__ __forceinline__ constexpr int q_off(int bin) {
    return bin == 0 ? 0 : bin == 1 ? JTK_BIN_CAP0 : bin == 2 ? JTK_BIN_CAP0 + JTK_BIN_CAP1 : bin == 3 ? JTK_BIN_CAP0 + JTK_BIN_CAP1 + JTK_BIN_CAP2
         : bin == 4 ? JTK_BIN_CAP0 + JTK_BIN_CAP1 + JTK_BIN_CAP2 + JTK_BIN_CAP3
         : bin == 5 ? JTK_BIN_CAP0 + JTK_BIN_CAP1 + JTK_BIN_CAP2 + JTK_BIN_CAP3 + JTK_BIN_CAP4
         : bin == 6 ? JTK_BIN_CAP0 + JTK_BIN_CAP1 + JTK_BIN_CAP2 + JTK_BIN_CAP3 + JTK_BIN_CAP4 + JTK_BIN_CAP5
         : JTK_BIN_CAP0 + JTK_BIN_CAP1 + JTK_BIN_CAP2 + JTK_BIN_CAP3 + JTK_BIN_CAP4 + JTK_BIN_CAP5 + JTK_BIN_CAP6;   // tiny
}
constexpr int Q_TOTAL = q_off(JTK_BIN_TINY) + JTK_TINY_CAP;

#ifndef JTK_RES_WAVES
#define JTK_RES_WAVES 4
#endif
constexpr int RES_WAVES = JTK_RES_WAVES, RES_THREADS = 64 * RES_WAVES;   // waves per tile
static_assert(RES_THREADS >= (T + 16) / 16 && RES_THREADS > 64, "the prologue's lane roles");

__global__ void __launch_bounds__(RES_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) k_piece_resolve(JtkWork w, JtkDeviceTables t) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tx[T + 16];
    __shared__ uint16_t s_plist[T + 1];
    __shared__ uint64_t s_pm[TW];
    __shared__ uint64_t s_gap[TW];
    __shared__ uint32_t s_q[Q_TOTAL];          // this tile's pieces for the merge kernels, by bin: offset | (len - 1) << 11 | index in this list << 19
    __shared__ uint32_t s_qn[JTK_NBINS + 1], s_qb[JTK_NBINS + 1], s_binc[JTK_NBINS + 1], s_nhard;
    __shared__ int64_t s_next_after;           // first piece start at or after B + T (global position)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t tile = blockIdx.x;
    const int64_t B = tile * T;
    const int64_t n = w.n_bytes;

    if (tid < (T + 16) / 16) {                                       // the tile's text and the 16 bytes after it, 16 bytes per lane
        const int64_t p = B + (int64_t)tid * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (p + 16 <= n) v = *reinterpret_cast<const uint4*>(w.text + p);
        else if (p < n) {
            uint32_t tmp[4] = {0, 0, 0, 0};
            for (int j = 0; j < 16; j++) if (p + j < n) tmp[j >> 2] |= (uint32_t)w.text[p + j] << (8 * (j & 3));
            v = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
        }
        reinterpret_cast<uint4*>(s_tx)[tid] = v;
    }
    if (tid < TW) {
        const int64_t wd = (B >> 6) + tid;
        uint64_t m = (wd < w.n_words) ? w.piecemask[wd] : 0ull;
        // only positions before n start pieces: the end sentinel (bit n) does not, and the padding words after it are
        // not written by pretok_split for every n (they may hold bits of an earlier, longer batch)
        if (wd * 64 + 63 >= n) m &= (wd * 64 >= n) ? 0ull : ((1ull << (n - wd * 64)) - 1ull);
        // a chunk of a larger batch starts at its first document, not at its first (tile-aligned) byte
        if (wd * 64 < w.lead) m &= (wd * 64 + 64 <= w.lead) ? 0ull : ~((1ull << (w.lead - wd * 64)) - 1ull);
        s_pm[tid] = m;
        s_gap[tid] = (w.gapmask && wd < w.n_words) ? w.gapmask[wd] : 0ull;
    }
    if (tid < JTK_NBINS + 1) {
        s_qn[tid] = 0;
        // per bin: where its pieces wait in s_q, how many of its results pack stages, and at which staging slot they start
        s_binc[tid] = (uint32_t)q_off(tid) | (uint32_t)(tid == JTK_BIN_TINY ? JTK_PACK_TINY : JTK_PACK_CAP(tid)) << 12 |
                      (uint32_t)(tid == JTK_BIN_TINY ? 0 : JTK_PACK_OFF(tid)) << 21;
    }
    if (tid == 0) s_nhard = 0;
    if (tid == 64) {
        int64_t pos = -1;                                             // scan ahead for the next piece start
        for (int64_t wd = (B >> 6) + TW; pos < 0 && wd < w.n_words; wd++) {
            const uint64_t m = w.piecemask[wd];
            if (m) pos = wd * 64 + jtk_ctz64(m);
        }
        s_next_after = (pos < 0) ? n : pos;
    }
    __syncthreads();
    // piece list: every wave scans the 32 word counts itself (lane l: word l), then compacts its share of the words
    int np;
    {
        const uint32_t c = lane < TW ? (uint32_t)__popcll(s_pm[lane]) : 0u;
        const uint32_t inc = wave_incl_scan(c);
        np = (int)(uint32_t)__shfl((int)inc, 63);
        const uint32_t pre = inc - c;
        for (int wd = wv; wd < TW; wd += RES_WAVES) {
            const uint64_t m = s_pm[wd];
            const uint32_t base = (uint32_t)__shfl((int)pre, wd);
            if ((m >> lane) & 1ull)
                s_plist[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)(wd * 64 + lane);
        }
        if (tid == 0) {
            // where the last piece ends: the sentinel (bit n) or the next tile's first piece; a piece of more than 64 KB only
            // needs to look longer than every bin (its length is taken again, exactly, where it is queued)
            const int64_t e = (n - B < T) ? (n - B) : (s_next_after - B);
            s_plist[np] = (uint16_t)(e < 0xFFFF ? e : 0xFFFF);
        }
    }
    __syncthreads();
    if (np == 0) {                                                   // (all waves) a tile inside one long piece
        if (tid == 0) { w.tile_np[tile] = 0; w.tile_tot[tile] = 0; }
        if (tid < 16) w.q_meta[tile * 16 + tid] = 0;
        return;
    }

    // One lane per piece, two pieces per lane in flight.  A piece of <= 8 bytes looks itself up in the tok8 table, one
    // of 9..16 bytes in the tok16 table (the reference's whole-piece shortcut, :81-83).  The tables are primary-first
    // (jtk_common.h): ONE scattered fetch per piece (two adjacent words for a 9..16-byte piece) answers hit or miss unless
    // the slot is flagged "overflowed"; only those lanes read their secondary slot in a second round.
    // The way to the answer is branch-free: every lane probes -- a lane beyond the list, or with a longer piece, looks
    // up whatever 16 bytes it has and ignores the answer -- so the wave never splits before the rare cases.
    const int64_t next_after = s_next_after;
    const bool gaps = w.gapmask != nullptr;
    const uint8_t* const tok = reinterpret_cast<const uint8_t*>(t.tok8.slots);   // the tok8 slots, then the tok16 slots: one allocation
    const uint32_t rel16 = (uint32_t)(reinterpret_cast<const uint8_t*>(t.tok16.slots) - tok);
    const uint32_t* const tw = reinterpret_cast<const uint32_t*>(s_tx);
    uint32_t* const plist = w.plist + B;
    struct Probe { uint32_t s, len, k0, k1, k2, k3, mix; uint4 ka; uint2 ma; };   // len 0: no piece
    auto piece_len = [&](int k, int s) -> int64_t {
        int64_t e;
        if (k + 1 < np) e = s_plist[k + 1];
        else e = (n - B < T) ? (n - B) : (next_after - B);            // last piece of the tile: ends at the sentinel
                                                                      // (bit n) or at the next tile's first piece
        return e - s;
    };
    auto slot_off = [&](uint32_t mix, bool small) -> uint32_t {       // byte offset of the slot in `tok`
        const uint32_t h = jtk_reduce32(mix, small ? t.tok8.bits : t.tok16.n);
        return (h << (small ? 4u : 5u)) + (small ? 0u : rel16);
    };
    auto issue = [&](int k, Probe& pr) {
        const bool have = k < np;
        const int kk = have ? k : np - 1;                             // np >= 1 here
        const uint32_t s0 = s_plist[kk], e0 = s_plist[kk + 1];        // s_plist[np] = where the tile's last piece ends (clamped)
        pr.s = have ? s0 : 0u;
        pr.len = have ? e0 - s0 : 0u;
        // up to 16 bytes of the piece, zero beyond its length
        const uint32_t a = pr.s >> 2, sh = pr.s & 3u;
        const uint32_t w0 = tw[a], w1 = tw[a + 1], w2 = tw[a + 2], w3 = tw[a + 3], w4 = tw[a + 4];
        const uint32_t len = pr.len < 16u ? pr.len : 16u;
        const uint64_t run = ~0ull >> ((0u - 8u * len) & 63u);        // 8 len ones (len 8 and 16: all 64)
        const bool big = len > 8u;
        const uint64_t mlo = big ? ~0ull : run, mhi = big ? run : 0ull;
        pr.k0 = __builtin_amdgcn_alignbyte(w1, w0, sh) & (uint32_t)mlo;
        pr.k1 = __builtin_amdgcn_alignbyte(w2, w1, sh) & (uint32_t)(mlo >> 32);
        pr.k2 = __builtin_amdgcn_alignbyte(w3, w2, sh) & (uint32_t)mhi;
        pr.k3 = __builtin_amdgcn_alignbyte(w4, w3, sh) & (uint32_t)(mhi >> 32);
        // one mix for both tables and both choices; only base, slot size and slot count depend on the length
        pr.mix = jtk_tok16_mix(pr.k0, pr.k1, pr.k2, pr.k3, len);
        const uint8_t* sa = tok + slot_off(pr.mix, !big);
        pr.ka = *reinterpret_cast<const uint4*>(sa);                     // tok8: lo, hi, id, len; tok16: the 16 key bytes
        pr.ma = make_uint2(0, 0);
        if (big) pr.ma = *reinterpret_cast<const uint2*>(sa + 16);       // tok16: id, len
    };
    // the answer of a slot: id, or JTK_RANK_NONE; `more`: a miss that the secondary slot has to confirm
    auto check = [&](const Probe& pr, uint32_t& id, bool& more) {
        const bool small = pr.len <= 8u;
        const uint32_t slen = small ? pr.ka.w : pr.ma.y;
        const uint32_t diff = (pr.ka.x ^ pr.k0) | (pr.ka.y ^ pr.k1) | ((slen & JTK_TOK_LEN_MASK) ^ pr.len) |
                              (small ? 0u : ((pr.ka.z ^ pr.k2) | (pr.ka.w ^ pr.k3)));
        id = diff == 0u ? (small ? pr.ka.z : pr.ma.x) : JTK_RANK_NONE;
        // (the slot's filter says whether a key with this mix can be among those it turned away)
        more = diff != 0u && (slen & JTK_TOK_FILTER_BIT(pr.mix)) != 0u && pr.len - 1u < 16u;
    };
    auto issue2 = [&](Probe& pr) {                                       // secondary slot
        const bool small = pr.len <= 8u;
        const uint8_t* sa = tok + slot_off(jtk_pair_mix2(pr.mix), small);
        pr.ka = *reinterpret_cast<const uint4*>(sa);
        if (!small) pr.ma = *reinterpret_cast<const uint2*>(sa + 16);
    };
    auto resolve = [&](int k, const Probe& pr, uint32_t id) {
        if (pr.len == 0u) return;
        const uint32_t s = pr.s, len = pr.len;
        const bool gap = gaps && ((s_gap[s >> 6] >> (s & 63u)) & 1ull);
        uint32_t entry = id | (s << JTK_PL_OFF_SHIFT);                 // (a hit implies len <= 16)
        if (gap || id == JTK_RANK_NONE) {
            entry = JTK_PL_HARD | JTK_PL_NOQUEUE | s;
            if (gap) {
                // text the caller's pattern did not match: no tokens (an htok header with count 0)
                w.htok[B + s] = 0u;
                atomicAdd(&s_nhard, 1u);
            } else if (len <= (uint32_t)JTK_BIN_MAXLEN) {
                // bin by length: 2..3 tiny, 4..8, 9..12, 13..16 (three bits per length from a constant), then 17..32, ..64, ..128, ..256
                // (a 1-byte piece is always a table entry)
                constexpr uint64_t BIN16 = (7ull << 6) | (7ull << 9) | (1ull << 27) | (1ull << 30) | (1ull << 33) | (1ull << 36) |
                                           (2ull << 39) | (2ull << 42) | (2ull << 45) | (2ull << 48);
                const uint32_t bin = len <= 16u ? (uint32_t)(BIN16 >> (3u * len)) & 7u : 30u - (uint32_t)__builtin_clz(len - 1u);
                const uint32_t bc = s_binc[bin];                           // q_off | staging cap << 12 | staging offset << 21
                const uint32_t i = atomicAdd(&s_qn[bin], 1u);
                s_q[(bc & 0xFFFu) + i] = s | ((len - 1u) << 11) | (i << 19);
                // pack finds the result of one of the tile's first few pieces of a bin in its LDS staging area: say where
                const bool st = i < ((bc >> 12) & 0x1FFu);
                const uint32_t idx = st ? (bc >> 21) + i : i;
                entry = JTK_PL_HARD | (st ? JTK_PL_STAGED : 0u) | (bin << JTK_PL_BIN_SHIFT) | (idx << JTK_PL_QI_SHIFT) | s;
            } else {
                const int64_t len64 = piece_len(k, (int)s);
                if (len64 <= JTK_MID_CAP) w.mid_list[atomicAdd(w.mid_count, 1u)] = JtkLongPiece{B + s, len64};
                else if (len64 <= JTK_LONG_CAP) w.long_list[atomicAdd(w.long_count, 1u)] = JtkLongPiece{B + s, len64};
                else {
                    // giant piece (a run of one byte value, mostly): merged by a whole workgroup in the last phase of
                    // k_bpe_merge; its token count goes to docpre[pos + 1] (a position inside the piece: no document starts
                    // there, so pack never writes that word), the htok header only says so
                    w.docpre[B + s + 1] = 0u;
                    if (len64 <= JTK_GIANT_CAP) w.giant_list[atomicAdd(w.n_giant, 1u)] = JtkLongPiece{B + s, len64};
                    else {
                        const int64_t d = find_doc(w, B + s);
                        if (d >= 0) atomicMin(&w.status[d], -10 /* JTK_ERR_PIECE_TOO_LONG */);
                    }
                    w.htok[B + s] = (uint32_t)JTK_HT_ESCAPE << JTK_HT_CNT_SHIFT;         // count: docpre[pos + 1]
                }
                atomicAdd(&s_nhard, 1u);
            }
        }
        plist[k] = entry;
    };
    // Pieces are taken in chunks of 64 (chunk c: pieces 64 c .. 64 c + 63); wave wv takes chunks wv, wv + RES_WAVES, ... two at
    // a time while there are two (so that two probes per lane are in flight), one otherwise: a tile of 530 pieces costs nine
    // chunk passes, not the sixteen of two full rounds of 512.
    for (int c0 = wv; c0 * 64 < np; c0 += 2 * RES_WAVES) {
        const bool two = (c0 + RES_WAVES) * 64 < np;                       // wave-uniform
        const int ka = c0 * 64 + lane, kb = (c0 + RES_WAVES) * 64 + lane;
        Probe p0, p1;
        issue(ka, p0);
        if (two) issue(kb, p1); else { p1.s = 0; p1.len = 0; p1.ka = make_uint4(0, 0, 0, 0); p1.ma = make_uint2(0, 0); p1.k0 = p1.k1 = p1.k2 = p1.k3 = p1.mix = 0; }
        uint32_t id0, id1 = JTK_RANK_NONE;
        bool more0, more1 = false;
        check(p0, id0, more0);
        if (two) check(p1, id1, more1);
        if (__ballot(more0 || more1)) {
            if (more0) issue2(p0);
            if (more1) issue2(p1);
            bool dummy;
            if (more0) check(p0, id0, dummy);
            if (more1) check(p1, id1, dummy);
        }
        resolve(ka, p0, id0);
        if (two) resolve(kb, p1, id1);
    }
    __syncthreads();
    // The tile's slices of its queue shards are claimed with one returning atomic per bin (a device-wide atomic:
    // about 2 us), then the queue entries are written: position and length for every bin, and for bin 0 also the
    // piece's bytes (so that the merge kernel reads 16 dense bytes per piece instead of a 64-byte slab of the text).
    // A tile with few merge pieces (ordinary text) leaves that to wave 0; the other waves are done and leave, so
    // their slots go to the next tile's workgroup while the atomic is in flight.
    const uint32_t n_queued = s_qn[0] + s_qn[1] + s_qn[2] + s_qn[3] + s_qn[4] + s_qn[5] + s_qn[6] + s_qn[JTK_BIN_TINY];
    const bool all_waves = n_queued > 64u;                           // workgroup-uniform
    if (!all_waves && wv != 0) return;
    if (wv == 0) {
        uint32_t nq = 0, qb = 0;
        if (lane < JTK_NBINS + 1) {
            nq = s_qn[lane];
            qb = nq ? atomicAdd(&w.q_count[JTK_QC(lane, tile % JTK_Q_SHARDS)], nq) : 0u;
            w.q_meta[tile * 16 + lane] = qb;
            w.q_meta[tile * 16 + 8 + lane] = nq;
            s_qb[lane] = qb;
        }
        if (lane == 0) {
            w.tile_np[tile] = (uint32_t)np;
            // resolved pieces = one token each; the merge kernels add the merged pieces' tokens
            w.tile_tot[tile] = (uint32_t)np - (s_nhard + n_queued);
        }
    }
    if (all_waves) __syncthreads(); else wave_lds_fence();
    const int nthr = all_waves ? RES_THREADS : WAVE, me = all_waves ? tid : lane;
#pragma unroll
    for (int q = 0; q < JTK_NBINS_BYTES; q++) {   // bins of <= 16 bytes: bytes + meta
        const uint32_t nq0 = s_qn[q];
        const int64_t qbase = (tile % JTK_Q_SHARDS) * w.q_cap[q] + s_qb[q];
        const uint32_t* tw = reinterpret_cast<const uint32_t*>(s_tx);
        for (uint32_t i = (uint32_t)me; i < nq0; i += (uint32_t)nthr) {
            const uint32_t e = s_q[q_off(q) + i];
            const uint32_t off = e & 2047u;
            const int a = (int)(off >> 2);
            const uint32_t sh = off & 3u;
            const uint32_t w0 = tw[a], w1 = tw[a + 1], w2 = tw[a + 2], w3 = tw[a + 3], w4 = tw[a + 4];
            w.qd[q][qbase + i] = make_uint4(__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
                                            __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w4, w3, sh));
            w.qm[q][qbase + i] = (uint64_t)(B + off) | ((uint64_t)((e >> 11) & 255u) << JTK_QE_LEN_SHIFT);
        }
    }
    {   // tiny pieces: position, length and the 2..3 bytes in one word
        const uint32_t nq5 = s_qn[JTK_BIN_TINY];
        uint64_t* dst = w.qt + (tile % JTK_Q_SHARDS) * w.qt_cap + s_qb[JTK_BIN_TINY];
        const uint32_t* tw = reinterpret_cast<const uint32_t*>(s_tx);
        for (uint32_t i = (uint32_t)me; i < nq5; i += (uint32_t)nthr) {
            const uint32_t e = s_q[q_off(JTK_BIN_TINY) + i];
            const uint32_t off = e & 2047u, len = ((e >> 11) & 255u) + 1u;
            const int a = (int)(off >> 2);
            uint32_t by = __builtin_amdgcn_alignbyte(tw[a + 1], tw[a], off & 3u) & 0xFFFFFFu;
            if (len == 2u) by &= 0xFFFFu;
            dst[i] = (uint64_t)(B + off) | ((uint64_t)(len - 2u) << 37) | ((uint64_t)by << 40);
        }
    }
#pragma unroll
    for (int q = JTK_NBINS_BYTES; q < JTK_NBINS; q++) {
        const int qoff = q_off(q);
        const uint32_t nq_q = s_qn[q];
        uint64_t* dst = w.qm[q] + (tile % JTK_Q_SHARDS) * w.q_cap[q] + s_qb[q];
        for (uint32_t i = (uint32_t)me; i < nq_q; i += (uint32_t)nthr) {
            const uint32_t e = s_q[qoff + i];
            dst[i] = (uint64_t)(B + (e & 2047u)) | ((uint64_t)((e >> 11) & 255u) << JTK_QE_LEN_SHIFT);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// lean bins of k_bpe_merge: bytePairMerge (GptBytePairEncoding.java:200-275) of the queued pieces of <= 64 bytes -- all
// but a handful of the pieces that need merging -- ONE LANE PER PIECE, no state machine: a wave takes 64 consecutive
// queue entries, expands them (byte -> id and the 2-byte-token ranks from LDS tables), then all lanes step together:
// leftmost minimum over the pair keys (:234-240), the two neighbour lookups in the (left id, right id) pair table,
// update (:248-259); a lane whose piece is finished idles until the wave's last piece is.  Entries of one wave come
// from the same stretch of text, so their lengths are alike; the slots scanned per step are bounded by the wave's
// longest piece (NS: a compile-time unrolled scan, all LDS reads of a step in flight together).
// What bounds the kernel is the number of scattered cache-line fetches (tools/microbench/gather_rate.hip: a CU
// sustains one per ~2.3 clocks), so a step fetches as few as it can: the pair table is primary-first (jtk_common.h) --
// ONE 16-byte load per lookup, issued for both lookups together; only lanes that miss in a bucket flagged "overflowed"
// read their secondary bucket -- and a lookup whose two parts make up the whole piece is not made at all: the piece
// is not a table entry, or piece_resolve would not have queued it (bin 0).
// Parts live in LDS laid out [slot][lane] (conflict-free for any per-lane slot); key = rank << 6 | slot orders by rank
// first and leftmost among equal ranks (:236).
// Bin 0 (<= 16 bytes): the piece's bytes are in the queue entry (written by piece_resolve); the result replaces them.
// Bins 1, 2 (<= 32, <= 64 bytes; rare): bytes from the text.  Token counts are summed per tile into tile_tot.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t KL_NONE = 0xFFFFFFFFu;
#ifndef JTK_ML_THREADS
#define JTK_ML_THREADS 1024
#endif
constexpr int ML_THREADS = JTK_ML_THREADS;       // lanes (= pieces in flight) per workgroup; 16 part slots x 8 bytes of LDS each
constexpr int ML_WORDS = 16 * ML_THREADS;
constexpr int ML_WGS_PER_SHARD = 4;

struct LeanLds {
    uint32_t* id;            // [16384] parts: token ids, [slot][lane]
    uint32_t* rk;            // [16384] parts: pair keys
    JtkBpLds bp;             // 2-byte tokens
    const uint32_t* brank;   // [256]
};

// merge steps on the first NS slots; returns the live-part mask.  SKIP_WHOLE: the piece itself is known not to be a
// table entry.
template <int NS, int STRIDE, bool SKIP_WHOLE, class M>
__device__ __forceinline__ M lean_steps(uint32_t* id, uint32_t* rk, M alive, const uint8_t* bk, uint32_t nb) {
    for (;;) {
        uint32_t k[NS];
#pragma unroll
        for (int j = 0; j < NS; j++) k[j] = rk[j * STRIDE];
#pragma unroll
        for (int d = 1; d < NS; d <<= 1) {
#pragma unroll
            for (int j = 0; j + d < NS; j += 2 * d) k[j] = min(k[j], k[j + d]);
        }
        const uint32_t m = k[0];
        const bool act = m != KL_NONE;                                                       // :247,:261
        if (!__ballot(act)) break;
        const uint32_t minr = act ? (m >> 6) : 0u, mini = act ? (m & 63u) : 0u;
        const M one = 1;
        const M above = alive & ~(((one << mini) << 1) - one);
        const M above2 = above & (above - one);
        const M below = alive & ((one << mini) - one);
        const bool has_nn = act && above2 != 0, has_pv = act && below != 0;
        uint32_t nxt, nn, pv;
        if (sizeof(M) == 8) {
            nxt = above ? (uint32_t)jtk_ctz64(above) : 0u;
            nn = above2 ? (uint32_t)jtk_ctz64(above2) : 0u;
            pv = below ? 63u - (uint32_t)jtk_clz64(below) : mini;
        } else {
            nxt = above ? (uint32_t)__builtin_ctz((uint32_t)above) : 0u;
            nn = above2 ? (uint32_t)__builtin_ctz((uint32_t)above2) : 0u;
            pv = below ? 31u - (uint32_t)__builtin_clz((uint32_t)below) : mini;
        }
        // (minr, id of the part after next) and (id of the previous part, minr); a pair that would be the whole piece is
        // known to be absent
        bool want1 = has_nn, want2 = has_pv;
        if (SKIP_WHOLE) {
            want1 = want1 && !(mini == 0u && (above2 & (above2 - one)) == 0);
            want2 = want2 && !(pv == 0u && !has_nn);
        }
        const uint32_t idnn = want1 ? id[nn * STRIDE] : 0u, idpv = want2 ? id[pv * STRIDE] : 0u;
        const uint32_t a1 = want1 ? minr : 0u, b2 = want2 ? minr : 0u;
        const uint32_t m1 = jtk_pair_mix(a1, idnn), m2 = jtk_pair_mix(idpv, b2);
        const uint4 v0 = *reinterpret_cast<const uint4*>(bk + ((size_t)jtk_reduce32(m1, nb) << 4));
        const uint4 v2 = *reinterpret_cast<const uint4*>(bk + ((size_t)jtk_reduce32(m2, nb) << 4));
        const uint32_t klo1 = (a1 << JTK_ID_BITS) | idnn, kt1 = (a1 >> (32 - JTK_ID_BITS)) << 30;
        const uint32_t klo2 = (idpv << JTK_ID_BITS) | b2, kt2 = (idpv >> (32 - JTK_ID_BITS)) << 30;
        uint32_t r1 = jtk_pair_match2(v0.x, v0.y, v0.z, v0.w, klo1, kt1);
        uint32_t r2 = jtk_pair_match2(v2.x, v2.y, v2.z, v2.w, klo2, kt2);
        const bool more1 = want1 && r1 == JTK_RANK_NONE && (v0.y & JTK_PAIR_OVERFLOW) != 0u;
        const bool more2 = want2 && r2 == JTK_RANK_NONE && (v2.y & JTK_PAIR_OVERFLOW) != 0u;
        if (more1 || more2) {
            // the secondary buckets, by the lanes that need one only (r03: with every lane taking part -- the others re-reading
            // their primary line -- the kernel was 2-3 % slower; with all four loads of a step issued together 20 % slower).
            // Both loads sit in one branch so that they are in flight together; a lane that needs one re-reads its other primary.
            const uint32_t h1 = more1 ? jtk_reduce32(jtk_pair_mix2(m1), nb) : jtk_reduce32(m1, nb);
            const uint32_t h2 = more2 ? jtk_reduce32(jtk_pair_mix2(m2), nb) : jtk_reduce32(m2, nb);
            const uint4 v1 = *reinterpret_cast<const uint4*>(bk + ((size_t)h1 << 4));
            const uint4 v3 = *reinterpret_cast<const uint4*>(bk + ((size_t)h2 << 4));
            const uint32_t y1 = jtk_pair_match2(v1.x, v1.y, v1.z, v1.w, klo1, kt1), y2 = jtk_pair_match2(v3.x, v3.y, v3.z, v3.w, klo2, kt2);
            r1 = more1 ? y1 : r1;
            r2 = more2 ? y2 : r2;
        }
        r1 = want1 ? r1 : JTK_RANK_NONE;
        r2 = want2 ? r2 : JTK_RANK_NONE;
        if (act) {
            // without a previous part the first store lands on slot mini and is overwritten by the second
            rk[pv * STRIDE] = (r2 == JTK_RANK_NONE) ? KL_NONE : ((r2 << 6) | pv);                           // :255-257
            rk[mini * STRIDE] = (r1 == JTK_RANK_NONE) ? KL_NONE : ((r1 << 6) | mini);                       // :254
            rk[nxt * STRIDE] = KL_NONE;
            id[mini * STRIDE] = minr;
            alive &= ~(one << nxt);                                                                          // :259
        }
    }
    return alive;
}

// expand (:206-221) + merge for pieces whose bytes are in registers: b[j] = byte j of the lane's piece (0 beyond its end)
template <int NS, int STRIDE>
__device__ __forceinline__ uint32_t lean_piece16(const LeanLds& L, uint32_t* id, uint32_t* rk, const uint32_t (&b)[NS + 1], int len,
                                                 const JtkDeviceTables& t) {
#pragma unroll
    for (int j = 0; j < NS; j++) {
        const uint32_t r = (j + 1 < len) ? jtk_bp_lookup(L.bp, (b[j] << 8) | b[j + 1]) : JTK_RANK_NONE;
        id[j * STRIDE] = L.brank[b[j]];
        rk[j * STRIDE] = (r != JTK_RANK_NONE) ? ((r << 6) | (uint32_t)j) : KL_NONE;
    }
    const uint32_t alive0 = (1u << len) - 1u;
    return lean_steps<NS, STRIDE, true, uint32_t>(id, rk, alive0, reinterpret_cast<const uint8_t*>(t.pairs.buckets), t.pairs.bits);
}

// the tiny queue: pieces of 2 or 3 bytes that are not table entries.  bytePairMerge (GptBytePairEncoding.java:200-275) of
// such a piece makes no lookup that can hit beyond the 2-byte-token ranks of its byte pairs: merge the pair of lower rank,
// the left one on a tie (:236), if either is a token; the pair that would follow is the whole piece, which is no entry.
template <int THREADS>
__device__ __forceinline__ void tiny_bin(const JtkWork& w, const LeanLds& L, uint32_t count, uint32_t kq, uint32_t K) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int shard = blockIdx.x % JTK_Q_SHARDS;
    uint64_t* const q = w.qt + (int64_t)shard * w.qt_cap;
    for (uint32_t base = kq * THREADS; base < count; base += K * THREADS) {
        const uint32_t qi = base + (uint32_t)tid;
        const bool have = qi < count;
        const uint64_t e = have ? q[qi] : 0ull;
        const int64_t pos = (int64_t)(e & JTK_QE_POS_MASK);
        const bool three = ((e >> 37) & 1ull) != 0;
        const uint32_t b0 = (uint32_t)(e >> 40) & 255u, b1 = (uint32_t)(e >> 48) & 255u, b2 = (uint32_t)(e >> 56) & 255u;
        const uint32_t r01 = three ? jtk_bp_lookup(L.bp, (b0 << 8) | b1) : JTK_RANK_NONE;
        const uint32_t r12 = three ? jtk_bp_lookup(L.bp, (b1 << 8) | b2) : JTK_RANK_NONE;
        const uint32_t i0 = L.brank[b0], i1 = L.brank[b1], i2 = L.brank[b2];
        uint32_t t0 = i0, t1 = i1, t2 = i2, c = three ? 3u : 2u;
        if (r01 != JTK_RANK_NONE && r01 <= r12) { t0 = r01; t1 = i2; c = 2u; }
        else if (r12 != JTK_RANK_NONE) { t1 = r12; c = 2u; }
        if (c == 2u) t2 = 0u;
        if (have) q[qi] = (uint64_t)t0 | ((uint64_t)t1 << 17) | ((uint64_t)t2 << 34) | ((uint64_t)(c - 1u) << 62);
        // token counts per tile (as in lean_bin)
        const int64_t tile = have ? pos / T : -1;
        const uint32_t cc = have ? c : 0u;
        const uint32_t inc = wave_incl_scan(cc);
        const uint32_t tlo = (uint32_t)tile, thi = (uint32_t)((uint64_t)tile >> 32);
        const uint32_t plo = (uint32_t)__shfl_up((int)tlo, 1), phi = (uint32_t)__shfl_up((int)thi, 1);
        const bool head = lane == 0 || plo != tlo || phi != thi;
        const uint64_t heads = __ballot(head);
        const uint64_t later = heads & ~((2ull << lane) - 1ull);
        const int last = later ? jtk_ctz64(later) - 1 : 63;
        const uint32_t run_end = (uint32_t)__shfl((int)inc, last);
        if (head && have) {
            const uint32_t sum = run_end - (inc - cc);
            if (sum) atomicAdd(&w.tile_tot[tile], sum);
        }
    }
}

template <int SLOTS, int THREADS, int BIN>
__device__ __forceinline__ void lean_bin(const JtkWork& w, const JtkDeviceTables& t, const LeanLds& L, uint32_t count, uint32_t kq, uint32_t K) {
    typedef typename std::conditional<(SLOTS > 32), uint64_t, uint32_t>::type M;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid >= THREADS) return;
    const int shard = blockIdx.x % JTK_Q_SHARDS;
    uint32_t* const id = L.id + tid;
    uint32_t* const rk = L.rk + tid;
    const uint64_t* const qm = w.qm[BIN] + (int64_t)shard * w.q_cap[BIN];
    uint4* const qd = w.qd[BIN] + (int64_t)shard * w.q_cap[BIN];

    for (uint32_t base = kq * THREADS; base < count; base += K * THREADS) {
        const uint32_t qi = base + (uint32_t)tid;
        bool have = qi < count;
        uint64_t meta = 0;
        uint4 by = make_uint4(0, 0, 0, 0);
        if (have) { meta = qm[qi]; if (BIN < JTK_NBINS_BYTES) by = qd[qi]; }
        if (BIN >= JTK_NBINS_BYTES && (meta & JTK_QE_DONE)) have = false;     // a table entry of > 16 bytes: result and count are in place
        const int64_t pos = (int64_t)(meta & JTK_QE_POS_MASK);
        const int len = have ? (int)((meta >> JTK_QE_LEN_SHIFT) & 255u) + 1 : 0;
        M alive;
        if (BIN < JTK_NBINS_BYTES) {
            const uint32_t d4[4] = {by.x, by.y, by.z, by.w};
            uint32_t b[17];
#pragma unroll
            for (int j = 0; j < 16; j++) b[j] = (d4[j >> 2] >> (8 * (j & 3))) & 255u;
            b[16] = 0;
            // the bin's longest piece picks the unrolled variant: 8, 12 or 16 slots
            if (BIN == 0) { uint32_t c[9]; for (int j = 0; j < 9; j++) c[j] = b[j]; alive = lean_piece16<8, THREADS>(L, id, rk, c, len, t); }
            else if (BIN == 1) { uint32_t c[13]; for (int j = 0; j < 13; j++) c[j] = b[j]; alive = lean_piece16<12, THREADS>(L, id, rk, c, len, t); }
            else alive = lean_piece16<16, THREADS>(L, id, rk, b, len, t);
        } else {
            // the piece's bytes from the text: the aligned 16-byte words that cover it are parked in the (idle) key slots,
            // then expanded in two passes (byte pairs into the id slots; ids and keys from those)
            const int64_t tb = pos & ~(int64_t)15;
            const uint32_t off = (uint32_t)(pos & 15);
            constexpr int NQ = SLOTS / 16 + 1;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (have && tb + 16 * q < w.n_bytes && (int)(16 * q) < (int)off + len) v = *reinterpret_cast<const uint4*>(w.text + tb + 16 * q);
                rk[(4 * q + 0) * THREADS] = v.x; rk[(4 * q + 1) * THREADS] = v.y; rk[(4 * q + 2) * THREADS] = v.z; rk[(4 * q + 3) * THREADS] = v.w;
            }
            int maxlen = len;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, d));
            maxlen = __builtin_amdgcn_readfirstlane(maxlen);
            const uint8_t* rkb = reinterpret_cast<const uint8_t*>(rk);
            uint32_t prev = rkb[(off >> 2) * THREADS * 4 + (off & 3)];
            for (int j = 0; j < maxlen; j++) {
                const uint32_t o = off + j + 1;
                const uint32_t cur = rkb[(o >> 2) * THREADS * 4 + (o & 3)];
                id[j * THREADS] = (prev << 8) | cur;                  // byte pair, expanded below
                prev = cur;
            }
            for (int j = 0; j < SLOTS; j++) {
                if (j < maxlen) {
                    const uint32_t bpi = id[j * THREADS];
                    const uint32_t r = (j + 1 < len) ? jtk_bp_lookup(L.bp, bpi & 0xFFFFu) : JTK_RANK_NONE;
                    rk[j * THREADS] = (r != JTK_RANK_NONE) ? ((r << 6) | (uint32_t)j) : KL_NONE;
                    id[j * THREADS] = L.brank[(bpi >> 8) & 255u];
                } else rk[j * THREADS] = KL_NONE;
            }
            const M one = 1;
            const M alive0 = (len >= (int)(8 * sizeof(M))) ? ~(M)0 : ((one << len) - one);
            alive = lean_steps<SLOTS, THREADS, false, M>(id, rk, alive0, reinterpret_cast<const uint8_t*>(t.pairs.buckets), t.pairs.bits);
        }

        // ---- emit (:270-273): one result word per piece; more than seven tokens go to htok
        const uint32_t c = sizeof(M) == 8 ? (uint32_t)__popcll((uint64_t)alive) : (uint32_t)__popc((uint32_t)alive);
        if (have) {
            uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = (c - 1u) << 24;
            if (c <= 7u) {
                M m = alive;
                uint32_t tk[7];
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    const uint32_t j = m ? (sizeof(M) == 8 ? (uint32_t)jtk_ctz64((uint64_t)m) : (uint32_t)__builtin_ctz((uint32_t)m)) : 0u;
                    tk[i] = m ? id[j * THREADS] : 0u;
                    m &= m - (M)1;
                }
                // 17 bits each from bit 0: token i at bit 17 * i
                r0 = tk[0] | (tk[1] << 17);
                r1 = (tk[1] >> 15) | (tk[2] << 2) | (tk[3] << 19);
                r2 = (tk[3] >> 13) | (tk[4] << 4) | (tk[5] << 21);
                r3 |= (tk[5] >> 11) | (tk[6] << 6);
            } else {
                uint32_t* dst = w.htok + pos;
                uint32_t idx = 0;
                for (M m = alive; m;) {
                    const uint32_t j = sizeof(M) == 8 ? (uint32_t)jtk_ctz64((uint64_t)m) : (uint32_t)__builtin_ctz((uint32_t)m);
                    m &= m - (M)1;
                    dst[idx++] = id[j * THREADS];
                }
            }
            qd[qi] = make_uint4(r0, r1, r2, r3);
        }
        // token counts per tile: entries of a tile are consecutive, so a wave sees a few runs of equal tiles; the first
        // lane of each run adds the run's sum
        {
            const int64_t tile = have ? pos / T : -1;
            const uint32_t cc = have ? c : 0u;
            const uint32_t inc = wave_incl_scan(cc);
            const uint32_t tlo = (uint32_t)tile, thi = (uint32_t)((uint64_t)tile >> 32);
            // (the shuffles are evaluated by ALL lanes, outside the condition: a lane that short-circuits an `||` leaves the
            // wave for the rest of the expression, and its neighbour would read a dead lane)
            const uint32_t plo = (uint32_t)__shfl_up((int)tlo, 1), phi = (uint32_t)__shfl_up((int)thi, 1);
            const bool head = lane == 0 || plo != tlo || phi != thi;
            const uint64_t heads = __ballot(head);
            const uint64_t later = heads & ~((2ull << lane) - 1ull);                  // run heads after this lane
            const int last = later ? jtk_ctz64(later) - 1 : 63;                        // last lane of this lane's run
            const uint32_t run_end = (uint32_t)__shfl((int)inc, last);
            if (head && have) {
                const uint32_t sum = run_end - (inc - cc);
                if (sum) atomicAdd(&w.tile_tot[tile], sum);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// bpe_merge: bytePairMerge (GptBytePairEncoding.java:200-275) of the queued pieces, ONE LANE PER PIECE,
// built around what bounds it -- dependent table lookups.  Each lane is a small state machine:
//     NEED -> (queue entry) -> TEXT -> (the piece's bytes) -> EXPAND -> MERGE ... -> EMIT -> NEED
// Every trip of the loop ALL lanes issue the same four 16-byte loads (addresses chosen by state, a hot
// dummy line when idle: no branches, so the loads are in flight together and the wave waits once); a
// piece costs (2..3 + merges) round trips and every wave keeps 64 independent chains in flight.  Lanes
// draw from a dense, sharded queue, so the wave stays full while the queue lasts.
// The parts of a piece (ids, pair ranks) live in LDS laid out [slot][lane] -- conflict-free for any
// per-lane slot index -- and so do the byte -> rank table and the complete 2-byte-token table (bitmap +
// ranks): setting a piece up needs no global lookups; only the (left id, right id) pair table is read
// from L2.  The two expensive divergent steps (EXPAND, EMIT) run when a batch of lanes has gathered.
// Leftmost-minimum (:236): min over key = rank << 9 | slot.
// One instantiation per length bin: <16 slots, 1024 lanes>, <32, 512>, <64, 256>, <128, 128>, <256, 64> --
// 128 KiB of parts each.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t RKP_NONE = 0xFFFFFFFFu;
constexpr int M_CHUNK = 2048;                  // queue entries a workgroup takes at a time

__device__ __forceinline__ uint32_t bsel(uint32_t m, uint32_t x1, uint32_t x0) { return (x1 & m) | (x0 & ~m); }

// live-part bit masks of up to 256 bits, kept in registers (every word index is an unrolled constant)
template <int NW> __device__ __forceinline__ int mask_next_after(const uint64_t (&w)[NW], int p) {   // first set bit > p, or -1
    int res = -1;
#pragma unroll
    for (int k = NW - 1; k >= 0; k--) {
        uint64_t m = w[k];
        const int base = k * 64;
        if (p >= base + 63) m = 0;
        else if (p >= base) m &= ~((2ull << (p - base)) - 1ull);
        if (m) res = base + jtk_ctz64(m);
    }
    return res;
}
template <int NW> __device__ __forceinline__ int mask_prev_before(const uint64_t (&w)[NW], int p) {  // last set bit < p, or -1
    int res = -1;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        uint64_t m = w[k];
        const int base = k * 64;
        if (p <= base) m = 0;
        else if (p < base + 64) m &= (1ull << (p - base)) - 1ull;
        if (m) res = base + 63 - jtk_clz64(m);
    }
    return res;
}
template <int NW> __device__ __forceinline__ void mask_clear(uint64_t (&w)[NW], int j) {
#pragma unroll
    for (int k = 0; k < NW; k++) { const uint64_t hit = 0ull - (uint64_t)((j >> 6) == k); w[k] &= ~((1ull << (j & 63)) & hit); }
}
template <int NW> __device__ __forceinline__ void mask_init(uint64_t (&w)[NW], int len) {
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const int r = len - k * 64;
        w[k] = r >= 64 ? ~0ull : (r > 0 ? ((1ull << r) - 1ull) : 0ull);
    }
}

// LDS of the merge kernel, shared by all its phases
struct MergeLds {
    uint32_t* id;            // [16384] parts: token ids, [slot][lane]
    uint32_t* rk;            // [16384] parts: pair keys
    const uint64_t* bpbits;
    const uint32_t* bpranks;
    const uint16_t* bpcum;
    const uint32_t* brank;
    uint32_t* next;          // [JTK_NBINS] queue positions handed out, one counter per bin
    const uint32_t* count;   // [JTK_NBINS + 3] entries in this workgroup's shard of each bin's queue; mid, long and giant list lengths
};

// One length bin: the first THREADS lanes of the workgroup drain this workgroup's chunks of the bin's queue shard.
template <int SLOTS, int THREADS, int BIN>
__device__ __forceinline__ void merge_bin(const JtkWork& w, const JtkDeviceTables& t, const MergeLds& L) {
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid >= THREADS) return;
    uint32_t* const s_id = L.id;
    uint32_t* const s_rk = L.rk;
    const uint32_t* const s_brank = L.brank;
    uint32_t& s_next = L.next[BIN];

    // dense queue shard `shard`; this workgroup takes chunks kq, kq + K, kq + 2K, ... of it
    const int shard = blockIdx.x % JTK_Q_SHARDS;
    const uint32_t kq = blockIdx.x / JTK_Q_SHARDS, K = gridDim.x / JTK_Q_SHARDS;
    const uint32_t count = L.count[BIN];
    if ((uint64_t)kq * M_CHUNK >= count) return;
    const uint64_t* const queue = w.qm[BIN] + (int64_t)shard * w.q_cap[BIN];     // entries are read in aligned pairs
    uint4* const results = w.qd[BIN] + (int64_t)shard * w.q_cap[BIN];

    const JtkBpLds bp{L.bpbits, L.bpcum, L.bpranks};
    const JtkPairTable pt = t.pairs;
    uint32_t* const id = s_id + tid;
    uint32_t* const rk = s_rk + tid;

    enum { ST_NEED = 0, ST_TEXT = 1, ST_EXPAND = 2, ST_MERGE = 3, ST_EMIT = 4, ST_DONE = 5 };
#ifndef JTK_EXPAND_BATCH
#define JTK_EXPAND_BATCH 48
#endif
#ifndef JTK_EMIT_BATCH
#define JTK_EMIT_BATCH 48
#endif
    constexpr int BATCH = JTK_EXPAND_BATCH, EMIT_BATCH = JTK_EMIT_BATCH;   // lanes that have to wait before the divergent steps run
    int st = ST_NEED;
    uint32_t qi = 0;
    int64_t pos = 0;
    int len = 0, tpart = 0;
    constexpr int NW = (SLOTS + 63) / 64;
    uint64_t alive[NW];
    mask_init<NW>(alive, 0);
    const uint4* const dummy = reinterpret_cast<const uint4*>(pt.buckets);

    for (;;) {
        // (1) merging lanes pick their pair: leftmost minimum of rank << 9 | slot (:234-240)
        uint32_t minr = 0, mini = 0, nxt = 0, nn = 0, pv = 0, idnn = 0, idpv = 0;
        bool has_nn = false, has_pv = false, merging = false;
        if (st == ST_MERGE) {
            uint32_t m = RKP_NONE;
            if (SLOTS <= 16) {
#pragma unroll
                for (int j = 0; j < SLOTS; j++) m = min(m, rk[j * THREADS]);
            } else {
                for (int j = 0; j < len; j++) m = min(m, rk[j * THREADS]);
            }
            if (m != RKP_NONE) {                                                             // :247
                merging = true;
                minr = m >> 9; mini = m & 511u;
                nxt = (uint32_t)mask_next_after<NW>(alive, (int)mini);
                const int nn_i = mask_next_after<NW>(alive, (int)nxt);
                has_nn = nn_i >= 0;
                nn = has_nn ? (uint32_t)nn_i : 0u;
                const int pv_i = mask_prev_before<NW>(alive, (int)mini);
                has_pv = pv_i >= 0;
                pv = has_pv ? (uint32_t)pv_i : 0u;
                idnn = id[nn * THREADS];
                idpv = id[pv * THREADS];
            } else st = ST_EMIT;                                                             // :261
        }
        const uint64_t b_merge = __ballot(merging);
        // (2) idle lanes take the next queue entries of this workgroup's chunks
        const uint64_t want = __ballot(st == ST_NEED);
        if (want) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_next, (uint32_t)__popcll(want));
            base = (uint32_t)__shfl((int)base, 0);
            if (st == ST_NEED) {
                const uint32_t seq = base + (uint32_t)__popcll(want & lanemask_lt());
                const uint64_t idx = (uint64_t)(kq + (seq / M_CHUNK) * K) * M_CHUNK + (seq % M_CHUNK);
                if (idx >= count) st = ST_DONE; else qi = (uint32_t)idx;
            }
        }
        if (!__ballot(st != ST_DONE)) break;

        // (3) the trip's loads: four per lane, unconditional
        const uint4* a0 = dummy; const uint4* a1 = dummy; const uint4* a2 = dummy; const uint4* a3 = dummy;
        const int64_t tbase = (pos & ~(int64_t)15) + 64 * (int64_t)tpart;
        if (st == ST_NEED) a0 = reinterpret_cast<const uint4*>(queue + (qi & ~1u));
        if (st == ST_TEXT) {
            const uint4* tx = reinterpret_cast<const uint4*>(w.text + tbase);
            a0 = tx;                                                  // reads stay inside the text buffer
            a1 = (tbase + 16 < w.n_bytes) ? tx + 1 : tx;
            a2 = (tbase + 32 < w.n_bytes) ? tx + 2 : tx;
            a3 = (tbase + 48 < w.n_bytes) ? tx + 3 : tx;
        }
        if (merging) {
            const uint4* bk = reinterpret_cast<const uint4*>(pt.buckets);
            if (has_nn) { a0 = bk + jtk_pair_hash(minr, idnn, pt.bits); a1 = bk + jtk_pair_hash2(minr, idnn, pt.bits); }
            if (has_pv) { a2 = bk + jtk_pair_hash(idpv, minr, pt.bits); a3 = bk + jtk_pair_hash2(idpv, minr, pt.bits); }
        }
        const uint4 v0 = *a0, v1 = *a1, v2 = *a2, v3 = *a3;

        // (4) consume
        if (st == ST_NEED) {
            const uint64_t entry = (qi & 1u) ? (((uint64_t)v0.w << 32) | v0.z) : (((uint64_t)v0.y << 32) | v0.x);
            pos = (int64_t)(entry & JTK_QE_POS_MASK);
            len = (int)((entry >> JTK_QE_LEN_SHIFT) & 255u) + 1;
            tpart = 0;
            st = (entry & JTK_QE_DONE) ? ST_NEED : ST_TEXT;       // (found by k_long_shortcut: nothing to merge)
        } else if (st == ST_TEXT) {
            // park this 64-byte slab of the window in the (idle) rank slots until the expansion batch runs
            uint32_t* park = rk + 16 * tpart * THREADS;
            if (16 * tpart + 15 < SLOTS || SLOTS >= 32) {
                park[0 * THREADS] = v0.x; park[1 * THREADS] = v0.y; park[2 * THREADS] = v0.z; park[3 * THREADS] = v0.w;
                park[4 * THREADS] = v1.x; park[5 * THREADS] = v1.y; park[6 * THREADS] = v1.z; park[7 * THREADS] = v1.w;
            }
            if (SLOTS >= 32 && 16 * tpart + 8 < SLOTS) {
                park[8 * THREADS] = v2.x; park[9 * THREADS] = v2.y; park[10 * THREADS] = v2.z; park[11 * THREADS] = v2.w;
            }
            if (SLOTS >= 32 && 16 * tpart + 12 < SLOTS) {
                park[12 * THREADS] = v3.x; park[13 * THREADS] = v3.y; park[14 * THREADS] = v3.z; park[15 * THREADS] = v3.w;
            }
            tpart++;
            if ((int64_t)(pos & 15) + len <= 64 * (int64_t)tpart) st = ST_EXPAND;
        } else if (merging) {
            const uint64_t k1 = jtk_pair_key(minr, idnn), k2 = jtk_pair_key(idpv, minr);
            const JtkPairBucket b11{v0.x, v0.y, v0.z, v0.w}, b12{v1.x, v1.y, v1.z, v1.w};
            const JtkPairBucket b21{v2.x, v2.y, v2.z, v2.w}, b22{v3.x, v3.y, v3.z, v3.w};
            uint32_t r1 = JTK_RANK_NONE, r2 = JTK_RANK_NONE;
            if (has_nn) { const uint32_t x = jtk_pair_match(b11, k1), y = jtk_pair_match(b12, k1); r1 = x != JTK_RANK_NONE ? x : y; }
            if (has_pv) { const uint32_t x = jtk_pair_match(b21, k2), y = jtk_pair_match(b22, k2); r2 = x != JTK_RANK_NONE ? x : y; }
            if (has_pv) rk[pv * THREADS] = (r2 == JTK_RANK_NONE) ? RKP_NONE : ((r2 << 9) | pv);     // :255-257
            rk[mini * THREADS] = (r1 == JTK_RANK_NONE) ? RKP_NONE : ((r1 << 9) | mini);             // :254
            rk[nxt * THREADS] = RKP_NONE;
            id[mini * THREADS] = minr;
            mask_clear<NW>(alive, (int)nxt);                                                        // :259
        }
        // (5) expand parked pieces: bytes at `pos` -> single-byte ids and 2-byte-token ranks (:206-221)
        const uint64_t b_exp = __ballot(st == ST_EXPAND);
        if (b_exp && (__popcll(b_exp) >= BATCH || !__ballot(st == ST_MERGE))) {
            if (st == ST_EXPAND) {
                const uint32_t off = (uint32_t)(pos & 15);
                const uint8_t* rkb = reinterpret_cast<const uint8_t*>(rk);
                if (SLOTS <= 16) {
                    // fixed 16 bytes, everything unrolled
                    uint32_t d[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) d[k] = rk[k * THREADS];
                    const uint32_t q = off >> 2, sh = off & 3u;
                    uint32_t e1[7], e2[5], o[4];
#pragma unroll
                    for (int k = 0; k < 7; k++) e1[k] = bsel(0u - (q & 1u), d[k + 1], d[k]);
#pragma unroll
                    for (int k = 0; k < 5; k++) e2[k] = bsel(0u - ((q >> 1) & 1u), e1[k + 2], e1[k]);
#pragma unroll
                    for (int k = 0; k < 4; k++) o[k] = __builtin_amdgcn_alignbyte(e2[k + 1], e2[k], sh);
                    uint32_t by[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) by[j] = (o[j >> 2] >> (8 * (j & 3))) & 255u;
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        id[j * THREADS] = s_brank[by[j]];
                        uint32_t r = JTK_RANK_NONE;
                        if (j + 1 < 16 && j + 1 < len) r = jtk_bp_lookup(bp, (by[j] << 8) | by[(j + 1) & 15]);
                        rk[j * THREADS] = (r == JTK_RANK_NONE) ? RKP_NONE : ((r << 9) | (uint32_t)j);
                    }
                } else {
                    uint32_t prev = rkb[(off >> 2) * THREADS * 4 + (off & 3)];
                    for (int j = 0; j + 1 < len; j++) {
                        const uint32_t o = off + j + 1;
                        const uint32_t cur = rkb[(o >> 2) * THREADS * 4 + (o & 3)];
                        id[j * THREADS] = (prev << 8) | cur;               // byte pair, expanded below
                        prev = cur;
                    }
                    id[(len - 1) * THREADS] = prev << 8;
                    for (int j = 0; j < len; j++) {
                        const uint32_t bpi = id[j * THREADS];
                        const uint32_t r = (j + 1 < len) ? jtk_bp_lookup(bp, bpi) : JTK_RANK_NONE;
                        rk[j * THREADS] = (r == JTK_RANK_NONE) ? RKP_NONE : ((r << 9) | (uint32_t)j);
                        id[j * THREADS] = s_brank[bpi >> 8];
                    }
                }
                mask_init<NW>(alive, len);
                st = ST_MERGE;
            }
        }
        // (6) emit finished pieces (:270-273) last, so the stores drain under the next trip's work.  The result of a
        // piece is ONE 16-byte word at its queue index: count - 1 in the top byte and, if the piece became at most 7
        // tokens, the token ids, 17 bits each.  Longer results leave their tokens in htok, packed from the piece's
        // first byte position.  The count is added to the tile's token total.
        const uint64_t b_emit = __ballot(st == ST_EMIT);
        if (b_emit && (__popcll(b_emit) >= EMIT_BATCH || !b_merge)) {
            if (st == ST_EMIT) {
                uint32_t c = 0;
#pragma unroll
                for (int k = 0; k < NW; k++) c += (uint32_t)__popcll(alive[k]);
                uint64_t lo = 0, hi = (uint64_t)(c - 1) << 56;
                if (c <= 7) {
                    uint32_t sh = 0;
#pragma unroll
                    for (int k = 0; k < NW; k++) {
                        for (uint64_t m = alive[k]; m;) {
                            const int j = k * 64 + jtk_ctz64(m);
                            m &= m - 1;
                            const uint64_t v = id[j * THREADS];
                            if (sh < 64u) lo |= v << sh;
                            if (sh > 47u) hi |= sh < 64u ? v >> (64u - sh) : v << (sh - 64u);
                            sh += 17u;
                        }
                    }
                } else {
                    uint32_t* dst = w.htok + pos;
                    uint32_t idx = 0;
#pragma unroll
                    for (int k = 0; k < NW; k++) {
                        for (uint64_t m = alive[k]; m;) {
                            const int j = k * 64 + jtk_ctz64(m);
                            m &= m - 1;
                            dst[idx++] = id[j * THREADS];
                        }
                    }
                }
                results[qi] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
                atomicAdd(&w.tile_tot[pos / T], c);
                st = ST_NEED;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// bpe_merge_long: one wave per piece of 65..8192 bytes.  Every lane scans a stride of the parts;
// leftmost-minimum selection is a wave reduction on key = rank << 13 | position (positions < 8192),
// which orders by rank first and by position among equal ranks (GptBytePairEncoding.java:236).
// ---------------------------------------------------------------------------------------------------
__device__ void merge_piece_wave(uint32_t* ids, uint32_t* rk, int len, const JtkPairTable pt) {
    const int lane = threadIdx.x & 63;
    for (;;) {
        uint32_t best = 0xFFFFFFFFu;
        for (int j = lane; j < len; j += WAVE) {
            if (ids[j] != JTK_ID_DEAD) {
                const uint32_t r = rk[j];
                if (r != JTK_RANK_NONE) best = min(best, (r << 13) | (uint32_t)j);
            }
        }
        best = wave_min_u32(best);
        if (best == 0xFFFFFFFFu) break;
        const uint32_t minr = best >> 13;
        const int mini = (int)(best & 8191u);
        // next two live parts after mini, previous live part before it (parts are <= 128 bytes long)
        int nxt = -1, nn = -1, pv = -1;
        for (int base = mini + 1; base < len && nn < 0; base += WAVE) {
            const int j = base + lane;
            uint64_t bal = __ballot(j < len && ids[j] != JTK_ID_DEAD);
            if (nxt < 0 && bal) { nxt = base + jtk_ctz64(bal); bal &= bal - 1; }
            if (nxt >= 0 && bal) nn = base + jtk_ctz64(bal);
        }
        for (int base = mini - 1; base >= 0 && pv < 0; base -= WAVE) {
            const int j = base - lane;
            const uint64_t bal = __ballot(j >= 0 && ids[j] != JTK_ID_DEAD);
            if (bal) pv = base - jtk_ctz64(bal);
        }
        uint32_t r = JTK_RANK_NONE;
        if (lane == 0 && nn >= 0) r = jtk_pair_lookup(pt, minr, ids[nn]);
        if (lane == 1 && pv >= 0) r = jtk_pair_lookup(pt, ids[pv], minr);
        wave_lds_fence();
        if (lane == 0) { ids[mini] = minr; rk[mini] = r; ids[nxt] = JTK_ID_DEAD; }
        if (lane == 1 && pv >= 0) rk[pv] = r;
        wave_lds_fence();
    }
}

// wave `wave_id` of `n_waves` takes every n_waves-th piece of the list; parts in this wave's LDS region (CAP words each)
template <int CAP>
__device__ __forceinline__ void merge_long(const JtkWork& w, const JtkDeviceTables& t, uint32_t* s_id, uint32_t* s_rk,
                                           uint32_t wave_id, uint32_t n_waves) {
    const int lane = threadIdx.x & 63;
    const JtkLongPiece* list = (CAP == JTK_MID_CAP) ? w.mid_list : w.long_list;
    const uint32_t cnt = (CAP == JTK_MID_CAP) ? *w.mid_count : *w.long_count;
    for (uint32_t i = wave_id; i < cnt; i += n_waves) {
        const JtkLongPiece lp = list[i];
        if (lp.len <= 0) continue;                                 // found by k_long_shortcut
        const int len = (int)lp.len;
        for (int j = lane; j < len; j += WAVE) {
            const uint32_t b0 = w.text[lp.start + j];
            s_id[j] = t.byte_rank[b0];
            s_rk[j] = (j + 1 < len) ? t.bp_rank[(b0 << 8) | w.text[lp.start + j + 1]] : JTK_RANK_NONE;
        }
        wave_lds_fence();
        merge_piece_wave(s_id, s_rk, len, t.pairs);
        // surviving ids, packed from the piece's first position; the count rides in word 0 (part 0 always survives)
        uint32_t total = 0;
        for (int base = 0; base < len; base += WAVE) {
            const int j = base + lane;
            const bool alive = j < len && s_id[j] != JTK_ID_DEAD;
            const uint64_t bal = __ballot(alive);
            const uint32_t idx = total + (uint32_t)__popcll(bal & lanemask_lt());
            if (alive && idx) w.htok[lp.start + idx] = s_id[j];
            total += (uint32_t)__popcll(bal);
        }
        if (lane == 0) {
            w.htok[lp.start] = s_id[0] | (total << JTK_HT_CNT_SHIFT);
            atomicAdd(&w.tile_tot[lp.start / T], total);
        }
        wave_lds_fence();
    }
}

// ---------------------------------------------------------------------------------------------------
// merge_giant: pieces of 8 KiB .. 1 MiB (a run of one byte value, mostly).  One workgroup per piece, the last
// phase of k_bpe_merge.  Parts live in the scratch words of the piece's own byte positions -- ids in
// htok[start ..], pair ranks in docpre[start ..] (pack writes docpre only later) -- so nothing is sized or
// launched by the host and the whole encode stays asynchronous.  A chunk-minimum cache in LDS (one packed key
// per 256 positions) keeps a merge at O(#chunks / threads + 256) instead of O(len).  Rare; exact; far cheaper
// than the reference's O(n^2) list surgery.  key = rank << 20 | position: rank first, leftmost among ties (:236).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

struct GiantLds {
    uint64_t* cmin;     // [JTK_GIANT_CAP / JTK_GIANT_CHUNK]
    uint64_t* wmin;     // [16]
    int* nb;            // [3] nxt, nn, pv
    uint32_t* r;        // [2]
};

__device__ void merge_giant(const JtkWork& w, const JtkDeviceTables& t, const GiantLds& L, uint32_t gi) {
    constexpr int CH = JTK_GIANT_CHUNK;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, NT = blockDim.x, NWV = NT >> 6;
    const JtkLongPiece lp = w.giant_list[gi];
    if (lp.len <= 0) return;                                      // found by k_long_shortcut (workgroup-uniform)
    const int len = (int)lp.len;
    uint32_t* gid = w.htok + lp.start;
    uint32_t* grk = w.docpre + lp.start;
    const int nch = (len + CH - 1) / CH;
    constexpr uint64_t KNONE = ~0ull;

    for (int j = tid; j < len; j += NT) {
        const uint32_t b0 = w.text[lp.start + j];
        gid[j] = t.byte_rank[b0];
        grk[j] = (j + 1 < len) ? t.bp_rank[(b0 << 8) | w.text[lp.start + j + 1]] : JTK_RANK_NONE;
    }
    __syncthreads();
    auto chunk_min = [&](int c) {                 // one wave: minimum key of chunk c
        uint64_t k = KNONE;
        for (int q = 0; q < CH / 64; q++) {
            const int j = c * CH + q * 64 + lane;
            if (j < len) { const uint32_t r = grk[j]; if (r != JTK_RANK_NONE) { const uint64_t kk = ((uint64_t)r << 20) | (uint32_t)j; k = kk < k ? kk : k; } }
        }
        k = wave_min_u64(k);
        if (lane == 0) L.cmin[c] = k;
    };
    for (int c = wv; c < nch; c += NWV) chunk_min(c);
    __syncthreads();

    for (;;) {
        uint64_t k = KNONE;
        for (int c = tid; c < nch; c += NT) { const uint64_t kk = L.cmin[c]; k = kk < k ? kk : k; }
        k = wave_min_u64(k);
        if (lane == 0) L.wmin[wv] = k;
        __syncthreads();
        k = L.wmin[0];
        for (int q = 1; q < NWV; q++) k = L.wmin[q] < k ? L.wmin[q] : k;
        if (k == KNONE) break;                                                               // :247,:261
        const uint32_t minr = (uint32_t)(k >> 20);
        const int mini = (int)(k & 0xFFFFFu);
        // neighbours (parts are at most 128 bytes long): wave 0 finds nxt and nn, wave 1 finds pv
        if (wv == 0) {
            int nxt = -1, nn = -1;
            for (int base = mini + 1; base < len && nn < 0; base += WAVE) {
                const int j = base + lane;
                uint64_t bal = __ballot(j < len && gid[j] != JTK_ID_DEAD);
                if (nxt < 0 && bal) { nxt = base + jtk_ctz64(bal); bal &= bal - 1; }
                if (nxt >= 0 && bal) nn = base + jtk_ctz64(bal);
            }
            if (lane == 0) { L.nb[0] = nxt; L.nb[1] = nn; L.r[0] = nn >= 0 ? jtk_pair_lookup(t.pairs, minr, gid[nn]) : JTK_RANK_NONE; }
        } else if (wv == 1) {
            int pv = -1;
            for (int base = mini - 1; base >= 0 && pv < 0; base -= WAVE) {
                const int j = base - lane;
                const uint64_t bal = __ballot(j >= 0 && gid[j] != JTK_ID_DEAD);
                if (bal) pv = base - jtk_ctz64(bal);
            }
            if (lane == 0) { L.nb[2] = pv; L.r[1] = pv >= 0 ? jtk_pair_lookup(t.pairs, gid[pv], minr) : JTK_RANK_NONE; }
        }
        __syncthreads();
        const int nxt = L.nb[0], pv = L.nb[2];
        if (tid == 0) {
            gid[mini] = minr; grk[mini] = L.r[0];                                            // :254
            gid[nxt] = JTK_ID_DEAD; grk[nxt] = JTK_RANK_NONE;                                // :259
            if (pv >= 0) grk[pv] = L.r[1];                                                   // :255-257
        }
        __syncthreads();
        // refresh the cached minima of the chunks that changed
        const int c0 = mini / CH, c1 = nxt / CH, c2 = pv >= 0 ? pv / CH : c0;
        if (wv == 0) chunk_min(c0);
        if (wv == 1 && c1 != c0) chunk_min(c1);
        if (wv == 2 && c2 != c0 && c2 != c1) chunk_min(c2);
        __syncthreads();
    }
    // emit (wave 0): surviving ids packed in place from the piece's first position (a survivor never moves up);
    // count in docpre[start + 1] (the pair ranks kept there are no longer needed)
    if (wv == 0) {
        uint32_t total = 0;
        uint32_t first = 0;
        for (int base = 0; base < len; base += WAVE) {
            const int j = base + lane;
            const uint32_t v = j < len ? gid[j] : JTK_ID_DEAD;
            const bool alive = v != JTK_ID_DEAD;
            const uint64_t bal = __ballot(alive);
            const uint32_t idx = total + (uint32_t)__popcll(bal & lanemask_lt());
            if (base == 0) first = (uint32_t)__shfl((int)v, 0);
            if (alive && idx) gid[idx] = v;
            total += (uint32_t)__popcll(bal);
        }
        if (lane == 0) {
            gid[0] = first | ((uint32_t)JTK_HT_ESCAPE << JTK_HT_CNT_SHIFT);
            grk[1] = total;
            atomicAdd(&w.tile_tot[lp.start / T], total);
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------
// k_bpe_merge: ONE persistent launch for all of bytePairMerge: the lean bins (pieces of <= 64 bytes: all but a handful),
// then the state-machine bins for pieces of up to 256 bytes, the wave-per-piece lists (<= 512, <= 8192 bytes) and the
// workgroup-per-piece giants.  All phases share the 128 KiB of LDS parts and the staged tables; a workgroup barrier
// separates them (their LDS layouts differ), but there is no device-wide barrier and no launch gap between them, and
// on ordinary text the later phases find empty queues and cost nothing.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(ML_THREADS) k_bpe_merge(JtkWork w, JtkDeviceTables t) {
    __shared__ uint32_t s_id[ML_WORDS];
    __shared__ uint32_t s_rk[ML_WORDS];
    __shared__ uint64_t s_bpbits[1024];
    __shared__ uint32_t s_bpranks[JTK_BP_MAX];
    __shared__ uint16_t s_bpcum[1024];
    __shared__ uint32_t s_brank[256];
    __shared__ uint32_t s_next[JTK_NBINS];
    __shared__ uint32_t s_count[JTK_NBINS + 3];
    __shared__ uint32_t s_ntiny;
    const int tid = threadIdx.x;
    const int shard = blockIdx.x % JTK_Q_SHARDS;
    const uint32_t kq = blockIdx.x / JTK_Q_SHARDS;
    if (tid < JTK_NBINS) {
        s_next[tid] = 0;
        s_count[tid] = w.q_count[JTK_QC(tid, shard)];
    }
    if (tid == 16) s_ntiny = w.q_count[JTK_QC(JTK_BIN_TINY, shard)];
    if (tid == JTK_NBINS) s_count[JTK_NBINS] = *w.mid_count;
    if (tid == JTK_NBINS + 1) s_count[JTK_NBINS + 1] = *w.long_count;
    if (tid == JTK_NBINS + 2) s_count[JTK_NBINS + 2] = *w.n_giant;
    __syncthreads();
    // (all the counts were read up front: a phase without work costs neither a global load nor a barrier)
    const uint32_t n0 = s_count[0], n1 = s_count[1], n2 = s_count[2], n3 = s_count[3], n4 = s_count[4];
    const uint32_t nt5 = s_ntiny;
    const bool rest = (s_count[5] | s_count[6] | s_count[JTK_NBINS] | s_count[JTK_NBINS + 1] | s_count[JTK_NBINS + 2]) != 0u;
    // Each phase is a chain of dependent lookups (as many as its longest piece has merges).  When every lean bin of the shard
    // fits one workgroup pass -- small batches, where those chains ARE the kernel's time -- the shard's workgroups take one
    // bin each, so the chains run side by side; otherwise every workgroup takes a slice of every bin.
    const bool side_by_side = n0 <= (uint32_t)ML_THREADS && n1 <= (uint32_t)ML_THREADS && n2 <= (uint32_t)ML_THREADS &&
                              nt5 <= (uint32_t)ML_THREADS && n3 <= (uint32_t)(ML_THREADS / 2) && n4 <= (uint32_t)(ML_THREADS / 4) &&
                              gridDim.x / JTK_Q_SHARDS >= 4u;
    const uint32_t K = side_by_side ? 1u : gridDim.x / JTK_Q_SHARDS, k = side_by_side ? 0u : kq;
    bool w0, w1, w2, w3, w4, w5;
    if (side_by_side) {
        w0 = kq == 0u && n0; w1 = kq == 1u && n1; w2 = kq == 2u && n2;
        w5 = kq == 3u && nt5; w3 = kq == 3u && n3; w4 = kq == 3u && n4;
    } else {
        w0 = kq * (uint32_t)ML_THREADS < n0; w1 = kq * (uint32_t)ML_THREADS < n1; w2 = kq * (uint32_t)ML_THREADS < n2;
        w3 = kq * (uint32_t)(ML_THREADS / 2) < n3; w4 = kq * (uint32_t)(ML_THREADS / 4) < n4;
        w5 = kq * (uint32_t)ML_THREADS < nt5;
    }
    if (!(w0 || w1 || w2 || w3 || w4 || w5 || rest)) return;
    for (int i = tid; i < 1024; i += ML_THREADS) { s_bpbits[i] = t.bp.bits[i]; s_bpcum[i] = t.bp.cum[i]; }
    for (int i = tid; i < JTK_BP_MAX; i += ML_THREADS) s_bpranks[i] = t.bp.ranks[i];
    if (tid < 256) s_brank[tid] = t.byte_rank[tid];
    __syncthreads();
    const LeanLds LL{s_id, s_rk, JtkBpLds{s_bpbits, s_bpcum, s_bpranks}, s_brank};
    if (w5) tiny_bin<ML_THREADS>(w, LL, nt5, k, K);              // (no parts in LDS: no barrier needed before the next phase)
    // (the three classes of <= 16 bytes share one LDS layout, [16 slots][1024 lanes], and a lane uses only its own column:
    // no barrier between them)
    if (w0) lean_bin<16, ML_THREADS, 0>(w, t, LL, n0, k, K);
    if (w1) lean_bin<16, ML_THREADS, 1>(w, t, LL, n1, k, K);
    if (w2) lean_bin<16, ML_THREADS, 2>(w, t, LL, n2, k, K);
    if (w3) { __syncthreads(); lean_bin<32, ML_THREADS / 2, 3>(w, t, LL, n3, k, K); }
    if (w4) { __syncthreads(); lean_bin<64, ML_THREADS / 4, 4>(w, t, LL, n4, k, K); }
    if (!rest) return;
    const MergeLds L{s_id, s_rk, s_bpbits, s_bpranks, s_bpcum, s_brank, s_next, s_count};
    if (s_count[5]) { __syncthreads(); merge_bin<128, ML_WORDS / 128, 5>(w, t, L); }
    if (s_count[6]) { __syncthreads(); merge_bin<256, ML_WORDS / 256, 6>(w, t, L); }
    // pieces of 257..512 bytes: every wave of the grid takes pieces, parts in its own 2 x 512 words
    const uint32_t wv = (uint32_t)tid >> 6;
    if (s_count[JTK_NBINS]) {
        __syncthreads();
        merge_long<JTK_MID_CAP>(w, t, s_id + wv * JTK_MID_CAP, s_rk + wv * JTK_MID_CAP, blockIdx.x * (uint32_t)(ML_THREADS / 64) + wv,
                                gridDim.x * (uint32_t)(ML_THREADS / 64));
    }
    // pieces of 513..8192 bytes: one wave per workgroup, parts in 2 x 8192 words
    if (s_count[JTK_NBINS + 1]) {
        __syncthreads();
        if (wv == 0) merge_long<JTK_LONG_CAP>(w, t, s_id, s_rk, blockIdx.x, gridDim.x);
    }
    // giant pieces (listed by piece_resolve): one workgroup per piece
    if (s_count[JTK_NBINS + 2]) {
        __syncthreads();
        const GiantLds G{reinterpret_cast<uint64_t*>(s_id), reinterpret_cast<uint64_t*>(s_rk), reinterpret_cast<int*>(s_rk + 64),
                         s_rk + 72};
        for (uint32_t gi = blockIdx.x; gi < s_count[JTK_NBINS + 2]; gi += gridDim.x) merge_giant(w, t, G, gi);
    }
}

// ---------------------------------------------------------------------------------------------------
// tile_scan: exclusive scan of the tokens per tile (tile_tot: the resolved pieces counted by piece_resolve plus
// what the merge kernels added).  One workgroup per chunk of SCAN_CHUNK tiles; its base is the sum of all earlier
// tiles, which it adds up itself (4 bytes per tile from L2: less than a descriptor hand-off between workgroups on
// different XCDs would cost -- a single-pass look-back scan was measured 4x slower here).
// ---------------------------------------------------------------------------------------------------
constexpr int SCAN_CHUNK = 4096;

__global__ void __launch_bounds__(1024) k_tile_scan(JtkWork w) {
    __shared__ uint64_t s_part[16];
    __shared__ uint32_t s_wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t chunk = blockIdx.x;
    // tokens before this chunk
    uint64_t b = 0;
    {
        const int64_t nb = chunk * SCAN_CHUNK;
        const uint4* t4 = reinterpret_cast<const uint4*>(w.tile_tot);
        for (int64_t i = tid; i < nb / 4; i += 1024) { const uint4 v = t4[i]; b += (uint64_t)v.x + v.y + v.z + v.w; }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)b, d), hi = (uint32_t)__shfl_xor((int)(uint32_t)(b >> 32), d);
        b += ((uint64_t)hi << 32) | lo;
    }
    if (lane == 0) s_part[wv] = b;
    // four consecutive tiles per lane
    const int64_t i0 = chunk * SCAN_CHUNK + (int64_t)tid * 4;
    uint32_t v[4];
    uint32_t sum = 0;
    for (int j = 0; j < 4; j++) { v[j] = (i0 + j < w.n_tiles) ? w.tile_tot[i0 + j] : 0u; sum += v[j]; }
    const uint32_t inc = wave_incl_scan(sum);
    if (lane == 63) s_wsum[wv] = inc;
    __syncthreads();
    const uint64_t job_before = (uint64_t)*w.job_tokens;      // tokens of the batch's earlier chunks (their scans ran before this one)
    uint64_t before = job_before;
    for (int k = 0; k < 16; k++) before += s_part[k];
    for (int k = 0; k < wv; k++) before += s_wsum[k];
    uint64_t run = before + inc - sum;
    for (int j = 0; j < 4; j++) { if (i0 + j < w.n_tiles) w.tile_off[i0 + j] = (int64_t)run; run += v[j]; }
    if (chunk == gridDim.x - 1 && tid == 1023) {           // lanes past the last tile carry the grand total
        w.tile_off[w.n_tiles] = (int64_t)run;
        w.set_info[0] = (int64_t)job_before;               // this chunk's first token
        w.set_info[1] = (int64_t)run;                      // ... and the end of its last
        w.result->n_tokens = (int64_t)run;
        *w.job_tokens_next = (int64_t)run;
    }
}

// ---------------------------------------------------------------------------------------------------
// pack: piece lists -> one packed token stream in text (= document) order.  Per tile: the exclusive
// scan of its pieces' token counts (1 per resolved piece, the htok header count per merged piece), then
// every piece writes its tokens at tile_off + prefix.  Also leaves, at every document's first byte, the
// tokens of its tile before it (docpre) for k_doc_offsets.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hard_count(const JtkWork& w, int64_t pos) {
    const uint32_t c = (w.htok[pos] >> JTK_HT_CNT_SHIFT) & JTK_HT_CNT_MASK;
    if (c != JTK_HT_ESCAPE) return c;
    return w.docpre[pos + 1];                               // giant piece (longer than JTK_GIANT_CAP: 0 tokens, status set)
}

// c token ids from htok to the output, eight loads in flight per round trip (htok is padded by 16 words)
struct __attribute__((packed, aligned(4))) U4Unaligned { uint32_t x, y, z, w; };
__device__ __forceinline__ void pack_copy(uint32_t* dst, const uint32_t* src, uint32_t c) {
    for (uint32_t i = 0; i < c; i += 8) {
        const U4Unaligned a = *reinterpret_cast<const U4Unaligned*>(src + i), b = *reinterpret_cast<const U4Unaligned*>(src + i + 4);
        dst[i] = a.x & JTK_HT_ID_MASK;
        if (i + 1 < c) dst[i + 1] = a.y & JTK_HT_ID_MASK;
        if (i + 2 < c) dst[i + 2] = a.z & JTK_HT_ID_MASK;
        if (i + 3 < c) dst[i + 3] = a.w & JTK_HT_ID_MASK;
        if (i + 4 < c) dst[i + 4] = b.x & JTK_HT_ID_MASK;
        if (i + 5 < c) dst[i + 5] = b.y & JTK_HT_ID_MASK;
        if (i + 6 < c) dst[i + 6] = b.z & JTK_HT_ID_MASK;
        if (i + 7 < c) dst[i + 7] = b.w & JTK_HT_ID_MASK;
    }
}

// token I (0..6) of a 16-byte merge result: 17 bits at bit 17 * I
template <int I> __device__ __forceinline__ uint32_t res_tok(const uint4& r) {
    constexpr int bit = 17 * I, wd = bit / 32, sh = bit % 32;
    const uint32_t w0 = wd == 0 ? r.x : wd == 1 ? r.y : wd == 2 ? r.z : r.w;
    const uint32_t w1 = wd == 0 ? r.y : wd == 1 ? r.z : r.w;
    return (sh + 17 <= 32 ? (w0 >> sh) : __builtin_amdgcn_alignbit(w1, w0, sh)) & JTK_HT_ID_MASK;
}

#ifndef JTK_PACK_STAGE
#define JTK_PACK_STAGE 768
#endif
constexpr int PACK_STAGE = JTK_PACK_STAGE;     // tokens of a tile assembled in LDS (ordinary text: a few hundred)
constexpr int PQT = JTK_PACK_TINY;

__global__ void __launch_bounds__(64) k_pack_tokens(JtkWork w) {
    // ONE WAVE PER TILE, no workgroup barriers.  A wave keeps a whole tile in flight: 8 list entries per lane, the head
    // of the tile's merge results (they are dense: the tile's slice of each bin's queue) and the document mask are all
    // requested before the first wait.  The tile's tokens are assembled in LDS (the few multi-token pieces make sparse
    // writes, cheap there and expensive in memory) and leave in full 256-byte stores.
    __shared__ uint4 s_qe[JTK_PACK_SLOTS];
    __shared__ uint2 s_qt[PQT];                 // staged results of the tile's tiny pieces
    __shared__ uint64_t s_dm[TW];
    __shared__ uint32_t s_out[PACK_STAGE];
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t B = tile * T;
    const int np = (int)w.tile_np[tile];
    const uint32_t total = w.tile_tot[tile];
    const uint32_t meta = lane < 16 ? w.q_meta[tile * 16 + lane] : 0u;    // lanes 0..4: start per bin, lanes 8..12: count per bin
    const bool stage = total <= (uint32_t)PACK_STAGE;
    const bool store = w.count_only == 0;     // countTokens(): offsets only, no token ids
    const uint32_t* plist = w.plist + B;
    int64_t tile_base;
    if (w.inline_scan) {
        // a small job (at most 1024 tiles): the tokens before this tile, added up here -- one launch less
        uint32_t part = 0;
        for (int64_t i = lane; i < tile; i += 64) part += w.tile_tot[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d);
        const int64_t job_before = *w.job_tokens;
        tile_base = job_before + (int64_t)part;
        if (lane == 0) {
            w.tile_off[tile] = tile_base;
            if (tile == w.n_tiles - 1) {
                const int64_t end = tile_base + (int64_t)total;
                w.tile_off[w.n_tiles] = end;
                w.set_info[0] = job_before;
                w.set_info[1] = end;
                w.result->n_tokens = end;
                *w.job_tokens_next = end;
            }
        }
    } else tile_base = w.tile_off[tile];
    uint32_t* const dst = reinterpret_cast<uint32_t*>(w.tokens + tile_base);
    uint32_t e[8];
#pragma unroll
    for (int j = 0; j < 8; j++) e[j] = plist[j * 64 + lane];          // (not waiting for np: entries beyond it are zeroed below)
    if (lane < TW) {
        const int64_t dwd = (B >> 6) + lane;
        s_dm[lane] = (dwd < w.n_words) ? w.docmask[dwd] : 0ull;
    }
    // the tile's merge results: the head of its slice of every bin's queue is staged (one load per lane for the three
    // classes of <= 16 bytes, one more for the longer bins if the tile has any), the rest is read on demand
    const int64_t shard = tile % JTK_Q_SHARDS;
    {
        const int bl = lane < 32 ? 0 : lane < 48 ? 1 : 2, il = lane < 32 ? lane : lane < 48 ? lane - 32 : lane - 48;
        const uint32_t qbv = (uint32_t)__shfl((int)meta, bl), nqv = (uint32_t)__shfl((int)meta, 8 + bl);
        const uint4* src = (bl == 0 ? w.qd[0] + shard * w.q_cap[0] : bl == 1 ? w.qd[1] + shard * w.q_cap[1] : w.qd[2] + shard * w.q_cap[2]) + qbv;
        if ((uint32_t)il < nqv) s_qe[lane] = src[il];
    }
    const uint32_t nq_hi = (uint32_t)__shfl((int)meta, 11) | (uint32_t)__shfl((int)meta, 12) | (uint32_t)__shfl((int)meta, 13) | (uint32_t)__shfl((int)meta, 14);
    if (nq_hi) {                                                          // wave-uniform; rare in ordinary text
        const int bq = 3 + ((lane >> 3) & 3), il = lane & 7;
        const uint32_t qb = (uint32_t)__shfl((int)meta, bq), nq = (uint32_t)__shfl((int)meta, 8 + bq);
        if (lane < 32 && (uint32_t)il < nq) s_qe[JTK_PACK_OFF(3) + lane] = (w.qd[bq] + shard * w.q_cap[bq] + qb)[il];
    }
    const uint32_t qb5 = (uint32_t)__shfl((int)meta, JTK_BIN_TINY), nq5 = (uint32_t)__shfl((int)meta, 8 + JTK_BIN_TINY);
    const uint2* const res5 = reinterpret_cast<const uint2*>(w.qt + shard * w.qt_cap + qb5);
    if (nq5) {                                                            // wave-uniform
#pragma unroll
        for (int r = 0; r < PQT / 64; r++) {
            const uint32_t i = (uint32_t)(r * 64 + lane);
            if (i < nq5) s_qt[i] = res5[i];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) e[j] = (j * 64 + lane < np) ? e[j] : 0u;
    wave_lds_fence();
    uint32_t run = 0;
    // a tiny piece's 8-byte result as a merge result word: the ids are where res_tok<0..2> looks, the count moves up
    auto tiny_word = [](uint2 r) { return make_uint4(r.x, r.y & 0x3FFFFFFFu, 0u, (r.y >> 30) << 24); };
    // one step: 64 consecutive pieces of the list, entry ej in lane order; out = s_out or dst
#define STORE(x) do { if (store) { x; } } while (0)
    auto step = [&](uint32_t* out, uint32_t ej, int k) {
        const bool valid = k < np;
        const bool hard = (ej & JTK_PL_HARD) != 0;
        const uint32_t bin = (ej >> JTK_PL_BIN_SHIFT) & 7u, qi = (ej >> JTK_PL_QI_SHIFT) & 1023u;
        const bool queued = hard && !(ej & JTK_PL_NOQUEUE);
        const bool tinyp = bin == JTK_BIN_TINY;
        const bool staged = queued && (ej & JTK_PL_STAGED) != 0u;       // (piece_resolve knew: qi is the staging slot then)
        const uint32_t sidx = staged && !tinyp ? qi : 0u;
        uint4 qe = s_qe[sidx];
        if (__ballot(queued && tinyp)) { if (staged && tinyp) qe = tiny_word(s_qt[qi]); }
        uint32_t c = valid ? (hard ? (qe.w >> 24) + 1u : 1u) : 0u;
        const uint32_t off = hard ? (ej & 2047u) : ((ej >> JTK_PL_OFF_SHIFT) & 2047u);
        const bool isdoc = valid && ((s_dm[off >> 6] >> (off & 63)) & 1ull);
        uint32_t pre;
        if (!__ballot(valid && hard && (!staged || c > 7u))) {
            // the common case, without divergent branches: exclusive scan of c (1 for most lanes, at most 7) by ballots of
            // the bits of c - 1, first tokens in one full store, the few further ones in sparse stores
            const uint32_t x = c ? c - 1u : 0u;
            const uint64_t bv = __ballot(valid);
            const uint64_t b0 = __ballot((x & 1u) != 0u), b1 = __ballot((x & 2u) != 0u), b2 = __ballot((x & 4u) != 0u);
            // set bits of a mask below this lane, added to acc: v_mbcnt_lo / _hi
            auto below = [](uint64_t m, uint32_t acc) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, acc)); };
            pre = below(b0, below(bv, run));
            run += (uint32_t)__popcll(bv) + (uint32_t)__popcll(b0);
            if (b1 | b2) {                                             // (wave-uniform) some piece became more than two tokens
                pre += 2u * below(b1, 0u) + 4u * below(b2, 0u);
                run += 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
            }
            if (valid) STORE(out[pre] = (hard ? qe.x : ej) & JTK_HT_ID_MASK);
            if (b0 | b1 | b2) {
                if (c > 1u) STORE(out[pre + 1] = res_tok<1>(qe));
                if (c > 2u) STORE(out[pre + 2] = res_tok<2>(qe));
                if (b1 | b2) {
                    if (c > 3u) STORE(out[pre + 3] = res_tok<3>(qe));
                    if (b2) {
                        if (c > 4u) STORE(out[pre + 4] = res_tok<4>(qe));
                        if (c > 5u) STORE(out[pre + 5] = res_tok<5>(qe));
                        if (c > 6u) STORE(out[pre + 6] = res_tok<6>(qe));
                    }
                }
            }
        } else {
            // general case: results beyond the staged head, results of more than 7 tokens (tokens in htok), pieces merged
            // by the wave / workgroup phases (count and tokens in htok)
            const uint32_t qb = (uint32_t)__shfl((int)meta, (int)(bin < JTK_NBINS ? bin : 0u));   // (all lanes take part)
            if (valid && queued && !staged) {
                qe = tinyp ? tiny_word(res5[qi]) : (w.qd[bin] + shard * w.q_cap[bin] + qb)[qi];
                c = (qe.w >> 24) + 1u;
            } else if (valid && hard && !queued) c = hard_count(w, B + off);        // count in the htok header
            const uint32_t inc = wave_incl_scan(c);
            pre = run + inc - c;
            run += (uint32_t)__shfl((int)inc, 63);
            if (valid) {
                if (!hard) STORE(out[pre] = ej & JTK_HT_ID_MASK);
                else if (queued && c <= 7u) {
                    STORE(out[pre] = qe.x & JTK_HT_ID_MASK);
                    if (c > 1u) STORE(out[pre + 1] = res_tok<1>(qe));
                    if (c > 2u) STORE(out[pre + 2] = res_tok<2>(qe));
                    if (c > 3u) STORE(out[pre + 3] = res_tok<3>(qe));
                    if (c > 4u) STORE(out[pre + 4] = res_tok<4>(qe));
                    if (c > 5u) STORE(out[pre + 5] = res_tok<5>(qe));
                    if (c > 6u) STORE(out[pre + 6] = res_tok<6>(qe));
                } else if (store) pack_copy(out + pre, w.htok + B + off, c);
            }
        }
        // document starts among these pieces: tokens of the tile before them
        if (__ballot(isdoc)) { if (isdoc) w.docpre[B + off] = pre; }
    };
#undef STORE
    for (int k0 = 0; k0 < np; k0 += 512) {
        if (k0) {
#pragma unroll
            for (int j = 0; j < 8; j++) { const int k = k0 + j * 64 + lane; e[j] = (k < np) ? plist[k] : 0u; }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (k0 + j * 64 >= np) break;
            if (stage) step(s_out, e[j], k0 + j * 64 + lane);
            else step(dst, e[j], k0 + j * 64 + lane);
        }
    }
    if (stage && store) {
        wave_lds_fence();
        for (uint32_t i = lane; i < total; i += WAVE) dst[i] = s_out[i];
    }
}

__global__ void __launch_bounds__(256) k_doc_offsets(JtkWork w) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > w.n_docs) return;
    const int64_t q = w.doc_off[d] - w.text_base;
    // a document starts a piece; documents at the very end (empty ones, and the end offset) start after the last token
    // (offsets that k_mark_docs rejected are not followed)
    w.tok_off[d] = (q >= w.n_bytes || q < w.lead) ? w.tile_off[w.n_tiles] : w.tile_off[q / T] + w.docpre[q];
    if (d < w.n_docs) {
        const int32_t st = w.status[d];
        if (st < 0) atomicMin(&w.result->worst_status, st);
    }
}

// ---------------------------------------------------------------------------------------------------
// long_shortcut: GptBytePairEncoding.java:81-83 for queued pieces of more than 16 bytes, for rank tables that hold
// entries of that length which bytePairMerge does not reproduce (jtk_common.h, JtkLongTokTable; never launched for the
// shipped tables).  One lane per queued piece: FNV-1a of its bytes, probe, byte-wise verification; a hit becomes the
// piece's one-token result right here and the merge kernels skip it.
// ---------------------------------------------------------------------------------------------------
__device__ uint32_t long_lookup(const JtkWork& w, const JtkDeviceTables& t, int64_t pos, int64_t len) {
    if (len > (int64_t)t.longtok.max_len || len <= 16) return JTK_RANK_NONE;
    uint64_t h = JTK_FNV_BASIS;
    for (int64_t j = 0; j < len; j++) h = jtk_fnv1a_step(h, w.text[pos + j]);
    const uint32_t n = t.longtok.n;
    for (uint32_t i = (uint32_t)(h % n), probes = 0; probes < n; i = (i + 1) % n, probes++) {
        const JtkLongTokSlot sl = t.longtok.slots[i];
        if (sl.len == 0) return JTK_RANK_NONE;
        if (sl.h_lo == (uint32_t)h && sl.h_hi == (uint32_t)(h >> 32) && sl.len == (uint32_t)len) {
            bool eq = true;
            for (int64_t j = 0; j < len && eq; j++) eq = t.longtok.blob[sl.blob_off + j] == w.text[pos + j];
            if (eq) return sl.id;
        }
    }
    return JTK_RANK_NONE;
}

__global__ void __launch_bounds__(256) k_long_shortcut(JtkWork w, JtkDeviceTables t) {
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, gn = gridDim.x * blockDim.x;
    for (int bin = JTK_NBINS_BYTES; bin < JTK_NBINS; bin++) {
        for (int shard = 0; shard < JTK_Q_SHARDS; shard++) {
            const uint32_t count = w.q_count[JTK_QC(bin, shard)];
            uint64_t* qm = w.qm[bin] + (int64_t)shard * w.q_cap[bin];
            uint4* qd = w.qd[bin] + (int64_t)shard * w.q_cap[bin];
            for (uint32_t i = gtid; i < count; i += gn) {
                const uint64_t meta = qm[i];
                const int64_t pos = (int64_t)(meta & JTK_QE_POS_MASK);
                const uint32_t id = long_lookup(w, t, pos, (int64_t)((meta >> JTK_QE_LEN_SHIFT) & 255u) + 1);
                if (id != JTK_RANK_NONE) {
                    qd[i] = make_uint4(id, 0u, 0u, 0u);               // one token (count - 1 = 0 in the top byte)
                    qm[i] = meta | JTK_QE_DONE;
                    atomicAdd(&w.tile_tot[pos / T], 1u);
                }
            }
        }
    }
    for (int which = 0; which < 3; which++) {
        JtkLongPiece* list = which == 0 ? w.mid_list : which == 1 ? w.long_list : w.giant_list;
        const uint32_t count = which == 0 ? *w.mid_count : which == 1 ? *w.long_count : *w.n_giant;
        for (uint32_t i = gtid; i < count; i += gn) {
            const JtkLongPiece lp = list[i];
            const uint32_t id = long_lookup(w, t, lp.start, lp.len);
            if (id != JTK_RANK_NONE) {
                w.htok[lp.start] = id | (1u << JTK_HT_CNT_SHIFT);    // header: one token
                list[i].len = 0;
                atomicAdd(&w.tile_tot[lp.start / T], 1u);
            }
        }
    }
}

}  // namespace

// chunk c of a large batch starts at the first document at or after byte c * chunk_bytes
__global__ void __launch_bounds__(256) k_plan_chunks(const int64_t* doc_off, int64_t n_docs, int64_t chunk_bytes, int n_chunks,
                                                     int64_t* out_doc, int64_t* out_off) {
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (c > n_chunks) return;
    int64_t lo = 0, hi = n_docs;                              // first d with doc_off[d] >= c * chunk_bytes
    if (c == n_chunks) lo = n_docs;
    else {
        const int64_t target = (int64_t)c * chunk_bytes;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (doc_off[mid] >= target) hi = mid; else lo = mid + 1;
        }
    }
    out_doc[c] = lo;
    out_off[c] = doc_off[lo];
}

void jtk_launch_plan_chunks(const int64_t* doc_off, int64_t n_docs, int64_t chunk_bytes, int n_chunks, int64_t* out_doc, int64_t* out_off,
                            hipStream_t s) {
    hipLaunchKernelGGL(k_plan_chunks, dim3((unsigned)((n_chunks + 1 + 255) / 256)), dim3(256), 0, s, doc_off, n_docs, chunk_bytes, n_chunks,
                       out_doc, out_off);
}

// offset stitch of a sharded batch: base = sum of the lower ranks' token totals; global offsets = local offsets + base
__global__ void __launch_bounds__(256) k_stitch(const int64_t* totals, int rank, int64_t* base_out, const int64_t* tok_off, int64_t n_docs,
                                                int64_t* global_off) {
    int64_t base = 0;
    for (int r = 0; r < rank; r++) base += totals[r];
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d == 0) *base_out = base;
    if (global_off && d <= n_docs) global_off[d] = tok_off[d] + base;
}
void jtk_launch_stitch(const int64_t* totals, int rank, int64_t* base_out, const int64_t* tok_off, int64_t n_docs, int64_t* global_off,
                       hipStream_t s) {
    const int64_t n = global_off ? n_docs + 1 : 1;
    hipLaunchKernelGGL(k_stitch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, totals, rank, base_out, tok_off, n_docs, global_off);
}

// ---------------------------------------------------------------------------------------------------
// Caller-supplied pieces (jtk_batch_encode_pieces): the host has run its own java.util.regex.Pattern
// (api/GptBytePairEncodingParams.java:36-46, EncodingFactory.java:117-119) and hands over the matches.  A piece starts
// at begin[i]; where it ends without the next one starting (text the pattern did not match: matcher.find() skips it),
// a dead "gap piece" starts, marked in gapmask; document starts that no match begins at start gap pieces too.
// piecemask and gapmask are zeroed before.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mark_pieces(JtkWork w, const int64_t* begin, const int64_t* end, int64_t n_pieces) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pieces) return;
    const int64_t p = begin[i] - w.text_base, e = end[i] - w.text_base;
    if (p < w.lead || e <= p || e > w.n_bytes || (i + 1 < n_pieces && begin[i + 1] - w.text_base < e)) {
        atomicMin(&w.result->worst_status, -1 /* JTK_ERR_INVALID_ARGUMENT */);
        return;
    }
    atomicOr((unsigned long long*)&w.piecemask[p >> 6], 1ull << (p & 63));
    const bool next_adjacent = (i + 1 < n_pieces) && (begin[i + 1] - w.text_base == e);
    if (!next_adjacent && e < w.n_bytes) {
        atomicOr((unsigned long long*)&w.piecemask[e >> 6], 1ull << (e & 63));
        atomicOr((unsigned long long*)&w.gapmask[e >> 6], 1ull << (e & 63));
    }
}
// after k_mark_pieces: every document start is a piece start (a gap piece unless a match begins there), and so is the
// end sentinel
__global__ void __launch_bounds__(256) k_mark_doc_gaps(JtkWork w) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > w.n_docs) return;
    const int64_t q = w.doc_off[d] - w.text_base;
    if (q < w.lead || q > w.n_bytes) return;
    const uint64_t bit = 1ull << (q & 63);
    const uint64_t old = atomicOr((unsigned long long*)&w.piecemask[q >> 6], bit);
    if (!(old & bit) && q < w.n_bytes) atomicOr((unsigned long long*)&w.gapmask[q >> 6], bit);
}
void jtk_launch_mark_pieces(const JtkWork& w, const int64_t* begin, const int64_t* end, int64_t n_pieces, hipStream_t s) {
    if (n_pieces > 0)
        hipLaunchKernelGGL(k_mark_pieces, dim3((unsigned)((n_pieces + 255) / 256)), dim3(256), 0, s, w, begin, end, n_pieces);
    const int64_t n = w.n_docs + 1;
    hipLaunchKernelGGL(k_mark_doc_gaps, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w);
}

void jtk_launch_mark_docs(const JtkWork& w, hipStream_t s) {
    const int64_t n = w.n_docs + 1;
    hipLaunchKernelGGL(k_mark_docs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w);
}
void jtk_launch_validate_utf8(const JtkWork& w, hipStream_t s) {
    if (w.n_bytes == 0) return;
    hipLaunchKernelGGL(k_validate_utf8, dim3((unsigned)((w.n_bytes + 255) / 256)), dim3(256), 0, s, w);
}
void jtk_launch_pretok_split(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s) {
    const int64_t tiles = (w.n_bytes + 1 + SPLIT_BYTES - 1) / SPLIT_BYTES;
    if (t.kind == JTK_PAT_CL100K) hipLaunchKernelGGL(k_pretok_split<JTK_PAT_CL100K>, dim3((unsigned)tiles), dim3(SPLIT_THREADS), 0, s, w, t);
    else hipLaunchKernelGGL(k_pretok_split<JTK_PAT_R50K>, dim3((unsigned)tiles), dim3(SPLIT_THREADS), 0, s, w, t);
}
void jtk_launch_piece_resolve(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s) {
    hipLaunchKernelGGL(k_piece_resolve, dim3((unsigned)w.n_tiles), dim3(RES_THREADS), 0, s, w, t);
}
void jtk_launch_long_shortcut(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s) {
    if (t.longtok.n) hipLaunchKernelGGL(k_long_shortcut, dim3(256), dim3(256), 0, s, w, t);
}
void jtk_launch_bpe_merge(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s) {
    hipLaunchKernelGGL(k_bpe_merge, dim3(JTK_Q_SHARDS * ML_WGS_PER_SHARD), dim3(ML_THREADS), 0, s, w, t);
}
void jtk_launch_tile_scan(const JtkWork& w, hipStream_t s) {
    hipLaunchKernelGGL(k_tile_scan, dim3((unsigned)((w.n_tiles + SCAN_CHUNK - 1) / SCAN_CHUNK)), dim3(1024), 0, s, w);
}
void jtk_launch_pack(const JtkWork& w, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_tokens, dim3((unsigned)w.n_tiles), dim3(64), 0, s, w);
}
// Rank maps that lack single-byte tokens (rare, hand-made): a token id >= pseudo_base in the output stands for a byte the
// table cannot encode -- the reference throws on such a piece (TokenEncoder.java:66-68).  One lane per token of the chunk:
// the document of an offending token gets JTK_ERR_UNENCODABLE.  Runs behind k_doc_offsets (it searches tok_off).
__global__ void __launch_bounds__(256) k_flag_unencodable(JtkWork w, uint32_t pseudo_base) {
    const int64_t t0 = w.tile_off[0], t1 = w.tile_off[w.n_tiles];
    const int64_t i = t0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t1 || (uint32_t)w.tokens[i] < pseudo_base) return;
    // last document of the chunk whose first token is at or before i (empty documents share an offset: the last one with
    // tokens is the one that ends after i)
    int64_t lo = 0, hi = w.n_docs;                                  // (this chunk's documents) tok_off[lo] <= i < tok_off[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (w.tok_off[mid] <= i) lo = mid; else hi = mid;
    }
    atomicMin(&w.status[lo], -12 /* JTK_ERR_UNENCODABLE */);
    atomicMin(&w.result->worst_status, -12);
}

void jtk_launch_flag_unencodable(const JtkWork& w, uint32_t pseudo_base, hipStream_t s) {
    if (w.n_bytes == 0) return;
    hipLaunchKernelGGL(k_flag_unencodable, dim3((unsigned)((w.n_bytes + 255) / 256)), dim3(256), 0, s, w, pseudo_base);
}

void jtk_launch_doc_offsets(const JtkWork& w, hipStream_t s) {
    const int64_t n = w.n_docs + 1;
    hipLaunchKernelGGL(k_doc_offsets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w);
}
