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
        const int len = t.special_len[s];
        if (p + len > w.n_bytes) continue;
        bool eq = true;
        for (int j = 0; j < len && eq; j++) eq = (w.text[p + j] == t.special[s][j]);
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
        for (int q = 0; q < t.n_specials; q++) if (t.special[q][0] == (uint8_t)tid) code |= JTK_F_LT;
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

constexpr int T = JTK_TILE;

#include "jtk_lean_merge.h"
#include "jtk_strip_common.h"
#include "jtk_strip_encode.h"
#include "jtk_bpe_merge.h"

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
int jtk_strip_encode_grid(int64_t n_tiles) {
    // two workgroups of 12 waves per CU hold it for the whole launch; small jobs spread their strips over as many CUs as they have
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n_cu = pr.multiProcessorCount; if (n_cu <= 0) n_cu = 256; }
    int64_t full = (int64_t)n_cu * ENC_WGS_PER_CU;
    if (full > JTK_MAX_Q_SHARDS) full = JTK_MAX_Q_SHARDS;
    int64_t wgs = n_tiles < full * ENC_WAVES ? (n_tiles < full ? n_tiles : full) : full;   // few strips: one wave each on as many CUs as possible
    return (int)(wgs < 1 ? 1 : wgs);
}
int jtk_strip_encode_waves(void) { return ENC_WAVES; }
void jtk_launch_strip_encode(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s) {
    hipLaunchKernelGGL(k_strip_encode, dim3(w.n_shards), dim3(ENC_THREADS), 0, s, w, t);
}
void jtk_launch_long_shortcut(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s) {
    if (t.longtok.n) hipLaunchKernelGGL(k_long_shortcut, dim3(256), dim3(256), 0, s, w, t);
}
void jtk_launch_bpe_merge(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s) {
    // every queue shard gets at least one workgroup; few shards (a small job) get several each, so that their bins run side by side
    const unsigned per = w.n_shards >= 256u ? 1u : (256u + w.n_shards - 1u) / w.n_shards;
    hipLaunchKernelGGL(k_bpe_merge, dim3(w.n_shards * per), dim3(ML_THREADS), 0, s, w, t);
}
void jtk_launch_tile_scan(const JtkWork& w, hipStream_t s) {
    hipLaunchKernelGGL(k_tile_scan, dim3((unsigned)((w.n_tiles + SCAN_CHUNK - 1) / SCAN_CHUNK)), dim3(1024), 0, s, w);
}
void jtk_launch_strip_expand(const JtkWork& w, hipStream_t s) {
    hipLaunchKernelGGL(k_strip_expand, dim3((unsigned)((w.n_tiles + EXPAND_THREADS / 64 - 1) / (EXPAND_THREADS / 64))), dim3(EXPAND_THREADS), 0, s, w);
}
void jtk_launch_doc_offsets(const JtkWork& w, hipStream_t s) {
    const int64_t n = w.n_docs + 1;
    hipLaunchKernelGGL(k_doc_offsets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w);
}

#ifdef JTK_ENC_STAMP
// diagnostic build (tools/enc_stamps.py): the encode kernel's phase cycles since the last call
extern "C" int jtk_debug_stamps(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_enc_stamp), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_enc_stamp), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
extern "C" int jtk_debug_stamps_expand(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_exp_stamp), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_exp_stamp), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

