// jtk_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the batch BPE encode path.
//
// Stage           kernel            replaces (reference, lib/src/main/java/com/knuddels/jtokkit/)
// mark_docs       k_mark_docs       document boundaries of the batch (one Encoding.encode call each)
// special_check   k_special_check   GptBytePairEncoding.java:52-56 (text.contains(specialToken))
// pretok_split    k_pretok_split    :77-80 matcher.find()/group() with EncodingFactory.java:63,105
// bpe_merge       k_bpe_merge       :81-86 + bytePairMerge :200-275 + getRank :285-300
// bpe_merge_long  k_bpe_merge_long  the same for pieces longer than a tile's LDS window
// tile_scan/pack  k_tile_scan/...   out.add / addAll (:82,:117) -- document-order token stream
//
// Integer / byte work only; no floating point, no MFMA.  One lane per byte in pretok_split, one
// lane per piece in bpe_merge (short pieces) and one wave per piece with a wave-level leftmost-min
// reduction for long pieces.
#include "jtk_kernels.h"

#include "jtk_merge_core.h"
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
    const int64_t q = w.doc_off[d];
    if (q < 0 || q > w.n_bytes) return;
    atomicOr((unsigned long long*)&w.docmask[q >> 6], 1ull << (q & 63));
}

// index of the document containing byte position p (skipping empty documents)
__device__ int64_t find_doc(const int64_t* doc_off, int64_t n_docs, int64_t p) {
    int64_t lo = 0, hi = n_docs;                 // first d with doc_off[d] > p
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (doc_off[mid] > p) hi = mid; else lo = mid + 1;
    }
    return lo - 1;
}

// ---------------------------------------------------------------------------------------------------
// special_check: flag documents that contain a special-token literal (all literals start with "<|")
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_special_check(JtkWork w, JtkDeviceTables t) {
    const int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (base >= w.n_bytes) return;
    uint32_t v[4];
    if (base + 16 <= w.n_bytes) {
        const uint4 q = *reinterpret_cast<const uint4*>(w.text + base);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        for (int i = 0; i < 4; i++) {
            uint32_t x = 0;
            for (int j = 0; j < 4; j++) {
                const int64_t p = base + i * 4 + j;
                if (p < w.n_bytes) x |= (uint32_t)w.text[p] << (8 * j);
            }
            v[i] = x;
        }
    }
    bool any = false;
    for (int i = 0; i < 4; i++) {
        const uint32_t x = v[i] ^ 0x3C3C3C3Cu;                       // '<'
        any |= ((x - 0x01010101u) & ~x & 0x80808080u) != 0;
    }
    if (!any) return;
    for (int k = 0; k < 16; k++) {
        const int64_t p = base + k;
        if (p + 1 >= w.n_bytes || w.text[p] != '<' || w.text[p + 1] != '|') continue;
        for (int s = 0; s < t.n_specials; s++) {
            const int len = t.special_len[s];
            if (p + len > w.n_bytes) continue;
            bool eq = true;
            for (int j = 0; j < len && eq; j++) eq = (w.text[p + j] == t.special[s][j]);
            if (!eq) continue;
            const int64_t d = find_doc(w.doc_off, w.n_docs, p);
            if (d >= 0 && p + len <= w.doc_off[d + 1]) atomicMin(&w.status[d], -2 /* JTK_ERR_UNSUPPORTED_SPECIAL */);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// pretok_split
// ---------------------------------------------------------------------------------------------------
constexpr int ST = JTK_SPLIT_TILE, SH = JTK_SPLIT_HALO;
constexpr int S_CB = ST + 2 * SH;          // class bytes kept in LDS: [B-SH, B+ST+SH)
constexpr int S_TX = S_CB + 8;             // text kept in LDS:        [B-SH-4, B+ST+SH+4)

struct GlobalText {
    const uint8_t* t; int64_t n;
    __device__ uint32_t byte(int64_t p) const { return (p >= 0 && p < n) ? t[p] : 0u; }
};

// Unbounded window: LDS where possible, else class bytes recomputed from global memory (runs longer
// than the halo; slow, exact).
struct SlowWin {
    typedef int64_t idx_t;
    static constexpr int kMaxWalk = 0;
    const uint8_t* cbv;      // LDS, index 0 = position lo
    const uint8_t* txv;      // LDS, index 0 = position lo - 4
    int64_t lo;
    const uint8_t* gtext; int64_t n; const uint64_t* docmask; JtkUcTables uc;
    __device__ uint32_t byte(int64_t p) const {
        const int64_t i = p - (lo - 4);
        if (i >= 0 && i < S_TX) return txv[i];
        return (p >= 0 && p < n) ? gtext[p] : 0u;
    }
    __device__ uint32_t cb(int64_t p) const {
        const int64_t i = p - lo;
        if (i >= 0 && i < S_CB) return cbv[i];
        if (p >= n || p < 0) return JTK_CB_DS;
        GlobalText g{gtext, n};
        uint32_t c = jtk_class_byte(g, uc, p);
        if ((docmask[p >> 6] >> (p & 63)) & 1ull) c |= JTK_CB_DS;
        return c;
    }
};

// Window-relative 32-bit indices straight into LDS; run walks are capped inside the halo.
struct FastWin {
    typedef int idx_t;
    static constexpr int kMaxWalk = SH - 8;
    const uint8_t* cbv;      // index 0 = position lo
    const uint8_t* txv;      // index 0 = position lo - 4
    __device__ uint32_t byte(int i) const { return txv[i + 4]; }
    __device__ uint32_t cb(int i) const { return cbv[i]; }
};

struct LdsText {
    const uint8_t* txv; int64_t tlo;
    __device__ uint32_t byte(int64_t p) const { return txv[p - tlo]; }
};

template <int KIND>
__global__ void __launch_bounds__(256) k_pretok_split(JtkWork w, JtkDeviceTables t) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tx[S_TX];
    __shared__ __attribute__((aligned(16))) uint8_t s_cb[S_CB];
    __shared__ uint64_t s_dm[S_CB / 64 + 1];

    const int tid = threadIdx.x;
    const int64_t B = (int64_t)blockIdx.x * ST;
    const int64_t lo = B - SH, tlo = lo - 4;
    const int64_t n = w.n_bytes;

    // text window, 4 bytes per lane per step (tlo is 4-byte aligned)
    for (int i = tid; i < S_TX / 4; i += 256) {
        const int64_t p = tlo + (int64_t)i * 4;
        uint32_t v = 0;
        if (p >= 0 && p + 4 <= n) v = *reinterpret_cast<const uint32_t*>(w.text + p);
        else for (int j = 0; j < 4; j++) { const int64_t q = p + j; if (q >= 0 && q < n) v |= (uint32_t)w.text[q] << (8 * j); }
        reinterpret_cast<uint32_t*>(s_tx)[i] = v;
    }
    for (int i = tid; i < S_CB / 64 + 1; i += 256) {
        const int64_t wd = (lo >> 6) + i;
        s_dm[i] = (wd >= 0 && wd < w.n_words) ? w.docmask[wd] : 0ull;
    }
    __syncthreads();

    // class byte of every window position
    LdsText lt{s_tx, tlo};
    for (int i = tid; i < S_CB; i += 256) {
        const int64_t p = lo + i;
        uint32_t c;
        if (p < 0) c = 0;
        else if (p >= n) c = JTK_CB_DS;
        else {
            c = jtk_class_byte(lt, t.uc, p);
            if ((s_dm[i >> 6] >> (i & 63)) & 1ull) c |= JTK_CB_DS;
        }
        s_cb[i] = (uint8_t)c;
    }
    __syncthreads();

    const FastWin fw{s_cb, s_tx};
    for (int r = 0; r < ST / 256; r++) {
        const int i = SH + r * 256 + tid;
        const int64_t p = lo + i;
        bool ms = false;
        if (p <= n) {
            bool unresolved = false;
            ms = jtk_is_piece_start_t<KIND>(fw, i, unresolved);
            if (unresolved) {
                const SlowWin sw{s_cb, s_tx, lo, w.text, n, w.docmask, t.uc};
                bool dummy = false;
                ms = jtk_is_piece_start_t<KIND>(sw, p, dummy);
            }
        }
        const uint64_t bal = __ballot(ms);
        if ((tid & 63) == 0) {
            const int64_t wd = p >> 6;
            if (wd < w.n_words) w.piecemask[wd] = bal;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// bpe_merge
// ---------------------------------------------------------------------------------------------------
constexpr int MT = JTK_MERGE_TILE, MO = JTK_MERGE_OVER, MW = MT + MO;
constexpr int M_BLK = MW / 64;                                      // 64-byte blocks in the window
constexpr int M_TW = MT / 64;                                       // piecemask words of the tile proper
static_assert(M_BLK <= 128 && M_TW <= 64, "scan helpers assume at most 128 blocks");

// One wave merges one piece of any length held in LDS (ids/rk indexed by byte position of the piece).
// Leftmost-minimum selection is a wave reduction on key = rank << 13 | position (positions < 8192),
// which orders by rank first and by position among equal ranks (GptBytePairEncoding.java:236).
__device__ void merge_piece_wave(uint32_t* ids, uint32_t* rk, int len, const JtkPairTable pt) {
    const int lane = threadIdx.x & 63;
    for (int j = lane; j < len; j += WAVE)
        rk[j] = (j + 1 < len) ? jtk_pair_lookup(pt, ids[j], ids[j + 1]) : JTK_RANK_NONE;
    wave_lds_fence();
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

// exclusive scan of cnt[0..n) (n <= 128) into pre[0..n], pre[n] = total; called by one whole wave
__device__ __forceinline__ void wave_scan_small(const uint32_t* cnt, uint32_t* pre, int n) {
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    for (int c0 = 0; c0 < n; c0 += WAVE) {
        const uint32_t c = (c0 + lane < n) ? cnt[c0 + lane] : 0u;
        const uint32_t inc = wave_incl_scan(c);
        if (c0 + lane < n) pre[c0 + lane] = base + inc - c;
        base += (uint32_t)__shfl((int)inc, 63);
    }
    if (lane == 0) pre[n] = base;
}

// up to 8 bytes of LDS text starting at (unaligned) offset s, little-endian, zero beyond len
__device__ __forceinline__ void piece_key(const uint8_t* tx, int s, int len, uint32_t& lo, uint32_t& hi) {
    const uint32_t* tw = reinterpret_cast<const uint32_t*>(tx);
    const int a = s >> 2;
    const uint32_t sh = (uint32_t)(s & 3);
    const uint32_t w0 = tw[a], w1 = tw[a + 1], w2 = tw[a + 2];
    lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
    hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
    if (len < 4) { lo &= (1u << (8 * len)) - 1u; hi = 0; }
    else if (len < 8) hi &= (1u << (8 * (len - 4))) - 1u;
}

__global__ void __launch_bounds__(256) k_bpe_merge(JtkWork w, JtkDeviceTables t) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tx[MW + 16];
    __shared__ uint32_t s_id[MW];
    __shared__ uint32_t s_rk[MW];
    __shared__ uint16_t s_plist[MT + 1];
    __shared__ uint16_t s_hard[MT];            // piece indices that need bytePairMerge
    __shared__ uint64_t s_pm[M_BLK + 1];       // piecemask words of the window
    __shared__ uint64_t s_tm[M_BLK];           // token-start masks of the window
    __shared__ uint32_t s_cnt[M_BLK];
    __shared__ uint32_t s_pre[M_BLK + 1];      // scanned counts
    __shared__ uint32_t s_brank[256];
    __shared__ uint16_t s_medium[MT / 64 + 1];
    __shared__ uint32_t s_nmedium, s_nhard;
    __shared__ int64_t s_next_after;           // first piece start at or after B + MT (global position)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t B = (int64_t)blockIdx.x * MT;
    const int64_t n = w.n_bytes;
    const JtkPairTable pt = t.pairs;

    s_brank[tid] = t.byte_rank[tid];
    for (int i = tid; i < (MW + 16) / 4; i += 256) {
        const int64_t p = B + (int64_t)i * 4;
        uint32_t v = 0;
        if (p + 4 <= n) v = *reinterpret_cast<const uint32_t*>(w.text + p);
        else for (int j = 0; j < 4; j++) { if (p + j < n) v |= (uint32_t)w.text[p + j] << (8 * j); }
        reinterpret_cast<uint32_t*>(s_tx)[i] = v;
    }
    for (int i = tid; i < M_BLK + 1; i += 256) {
        const int64_t wd = (B >> 6) + i;
        const uint64_t m = (wd < w.n_words) ? w.piecemask[wd] : 0ull;
        s_pm[i] = m;
        if (i < M_TW) s_cnt[i] = (uint32_t)__popcll(m);
    }
    for (int i = tid; i < MW; i += 256) s_id[i] = JTK_ID_DEAD;
    if (tid == 0) { s_nmedium = 0; s_nhard = 0; }
    __syncthreads();

    // piece list of the tile: starts in [B, B+MT)
    if (wv == 0) wave_scan_small(s_cnt, s_pre, M_TW);
    if (tid == 64) {
        // first piece start at or after B+MT: in the window's overhang words, else scan ahead
        int64_t pos = -1;
        for (int i = M_TW; i < M_BLK + 1 && pos < 0; i++)
            if (s_pm[i]) pos = B + (int64_t)i * 64 + jtk_ctz64(s_pm[i]);
        for (int64_t wd = (B >> 6) + M_BLK + 1; pos < 0 && wd < w.n_words; wd++) {
            const uint64_t m = w.piecemask[wd];
            if (m) pos = wd * 64 + jtk_ctz64(m);
        }
        s_next_after = (pos < 0) ? n : pos;
    }
    __syncthreads();
    const int np = (int)s_pre[M_TW];
    for (int wd = wv; wd < M_TW; wd += 4) {
        const uint64_t m = s_pm[wd];
        if ((m >> lane) & 1ull) s_plist[s_pre[wd] + __popcll(m & lanemask_lt())] = (uint16_t)(wd * 64 + lane);
    }
    __syncthreads();

    // ---- pass 1: one lane per piece.  Pieces of <= 8 bytes that are table entries are that one token
    // (GptBytePairEncoding.java:81-83); everything else is queued for bytePairMerge.  Four rounds of
    // table probes are in flight per lane.
    const int64_t next_after = s_next_after;
    const JtkTok8Slot* t8 = t.tok8.slots;
    const uint32_t t8mask = (1u << t.tok8.bits) - 1u;
    for (int k0 = 0; k0 < np; k0 += 4 * 256) {
        int ps[4], pl[4];
        uint32_t klo[4], khi[4], kh[4];
        uint4 slot[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = k0 + u * 256 + tid;
            ps[u] = -1; pl[u] = 0;
            if (k < np) {
                const int s = s_plist[k];
                if (B + s < n) {                                          // the end sentinel is not a piece
                    const int64_t e = (k + 1 < np) ? (int64_t)s_plist[k + 1] : (next_after - B);
                    ps[u] = s;
                    pl[u] = (e - s > 0x7FFF0000) ? 0x7FFF0000 : (int)(e - s);
                }
            }
            if (ps[u] >= 0 && pl[u] <= 8) {
                piece_key(s_tx, ps[u], pl[u], klo[u], khi[u]);
                kh[u] = jtk_tok8_hash(klo[u], khi[u], (uint32_t)pl[u], t.tok8.bits);
                slot[u] = *reinterpret_cast<const uint4*>(&t8[kh[u]]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            bool hard = false;
            if (ps[u] >= 0) {
                const int s = ps[u], len = pl[u];
                if (len <= 8) {
                    uint4 sl = slot[u];
                    uint32_t h = kh[u];
                    uint32_t id = JTK_RANK_NONE;
                    for (;;) {
                        if (sl.w == (uint32_t)len && sl.x == klo[u] && sl.y == khi[u]) { id = sl.z; break; }
                        if (sl.w == 0) break;
                        h = (h + 1) & t8mask;
                        sl = *reinterpret_cast<const uint4*>(&t8[h]);
                    }
                    if (id != JTK_RANK_NONE) s_id[s] = id; else hard = true;
                } else if (len <= 64) {
                    hard = true;
                } else if ((int64_t)s + len <= MW) {
                    s_medium[atomicAdd(&s_nmedium, 1u)] = (uint16_t)(k0 + u * 256 + tid);
                } else {
                    const uint32_t sl = atomicAdd(w.long_count, 1u);
                    w.long_list[sl] = JtkLongPiece{B + s, (uint32_t)len, blockIdx.x};
                }
            }
            const uint64_t bal = __ballot(hard);
            if (bal) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_nhard, (uint32_t)__popcll(bal));
                base = (uint32_t)__shfl((int)base, 0);
                if (hard) s_hard[base + __popcll(bal & lanemask_lt())] = (uint16_t)(k0 + u * 256 + tid);
            }
        }
    }
    __syncthreads();

    // ---- pass 2: bytePairMerge, one lane per queued piece (<= 64 bytes)
    const int nhard = (int)s_nhard;
    for (int hI = tid; hI < nhard; hI += 256) {
        const int k = s_hard[hI];
        const int s = s_plist[k];
        const int e = (k + 1 < np) ? (int)s_plist[k + 1] : (int)(next_after - B);
        jtk_merge_piece_lane2(&s_id[s], &s_rk[s], &s_tx[s], e - s, pt, t.bp_rank, s_brank);
    }
    // pieces of 65..window bytes: one wave each, cooperative leftmost-min
    const int nmed = (int)s_nmedium;
    for (int m = wv; m < nmed; m += 4) {
        const int k = s_medium[m];
        const int s = s_plist[k];
        const int e = (k + 1 < np) ? (int)s_plist[k + 1] : (int)(next_after - B);
        for (int j = s + lane; j < e; j += WAVE) s_id[j] = s_brank[s_tx[j]];
        wave_lds_fence();
        merge_piece_wave(&s_id[s], &s_rk[s], e - s, pt);
    }
    __syncthreads();

    // ---- pack the tile's tokens in position order
    for (int blk = wv; blk < M_BLK; blk += 4) {
        const uint64_t bal = __ballot(s_id[blk * 64 + lane] != JTK_ID_DEAD);
        if (lane == 0) { s_tm[blk] = bal; s_cnt[blk] = (uint32_t)__popcll(bal); }
    }
    __syncthreads();
    if (wv == 0) wave_scan_small(s_cnt, s_pre, M_BLK);
    __syncthreads();
    const int64_t fs = (np > 0) ? B + s_plist[0] : B;
    for (int blk = wv; blk < M_BLK; blk += 4) {
        const uint64_t m = s_tm[blk];
        if ((m >> lane) & 1ull) w.tmp_tok[fs + s_pre[blk] + __popcll(m & lanemask_lt())] = (int32_t)s_id[blk * 64 + lane];
        if (lane == 0) {
            const int64_t wd = (B >> 6) + blk;
            if (m && wd < w.n_words) atomicOr((unsigned long long*)&w.tokmask[wd], m);
            if (blk < M_TW && wd < w.n_words) w.blk_pre[wd] = (uint16_t)s_pre[blk];
        }
    }
    if (tid == 0) { w.tile_cnt[blockIdx.x] = s_pre[M_BLK]; w.tile_fs[blockIdx.x] = fs; }
}

// ---------------------------------------------------------------------------------------------------
// bpe_merge_long: pieces that leave their tile's window (always the last piece of the tile).
// One 64-lane workgroup per piece, ids/ranks in LDS.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_bpe_merge_long(JtkWork w, JtkDeviceTables t) {
    __shared__ uint32_t s_id[JTK_LONG_CAP];
    __shared__ uint32_t s_rk[JTK_LONG_CAP];
    const int lane = threadIdx.x;
    const uint32_t cnt = *w.long_count;
    for (uint32_t i = blockIdx.x; i < cnt; i += gridDim.x) {
        const JtkLongPiece lp = w.long_list[i];
        if (lp.len > JTK_LONG_CAP) {
            if (lane == 0) {
                const int64_t d = find_doc(w.doc_off, w.n_docs, lp.start);
                if (d >= 0) atomicMin(&w.status[d], -10 /* JTK_ERR_PIECE_TOO_LONG */);
            }
            continue;
        }
        const int len = (int)lp.len;
        for (int j = lane; j < len; j += WAVE) s_id[j] = t.byte_rank[w.text[lp.start + j]];
        wave_lds_fence();
        merge_piece_wave(s_id, s_rk, len, t.pairs);
        // append after the tile's own tokens, in position order
        const int64_t out0 = w.tile_fs[lp.tile] + w.tile_cnt[lp.tile];
        uint32_t total = 0;
        for (int base = 0; base < len; base += WAVE) {
            const int j = base + lane;
            const bool alive = j < len && s_id[j] != JTK_ID_DEAD;
            const uint64_t bal = __ballot(alive);
            if (alive) w.tmp_tok[out0 + total + __popcll(bal & lanemask_lt())] = (int32_t)s_id[j];
            // token-start bits; the piece need not be 64-aligned, so split the ballot over two words
            if (lane == 0 && bal) {
                const int64_t p0 = lp.start + base;
                const int sh = (int)(p0 & 63);
                atomicOr((unsigned long long*)&w.tokmask[p0 >> 6], bal << sh);
                if (sh) atomicOr((unsigned long long*)&w.tokmask[(p0 >> 6) + 1], bal >> (64 - sh));
            }
            total += (uint32_t)__popcll(bal);
        }
        wave_lds_fence();
        if (lane == 0) w.tile_cnt[lp.tile] += total;
    }
}

// ---------------------------------------------------------------------------------------------------
// tile_scan: exclusive scan of per-tile token counts (one workgroup)
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_tile_scan(JtkWork w) {
    __shared__ uint64_t s_wsum[16];
    __shared__ uint64_t s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < w.n_tiles; c0 += 4096) {
        // four consecutive tiles per lane
        const int64_t i0 = c0 + (int64_t)tid * 4;
        uint32_t v[4];
        uint32_t sum = 0;
        for (int j = 0; j < 4; j++) { v[j] = (i0 + j < w.n_tiles) ? w.tile_cnt[i0 + j] : 0u; sum += v[j]; }
        const uint32_t inc = wave_incl_scan(sum);
        if (lane == 63) s_wsum[wv] = inc;
        __syncthreads();
        uint64_t before = s_base;
        for (int k = 0; k < wv; k++) before += s_wsum[k];
        uint64_t run = before + inc - sum;
        for (int j = 0; j < 4; j++) { if (i0 + j < w.n_tiles) w.tile_off[i0 + j] = (int64_t)run; run += v[j]; }
        __syncthreads();
        if (tid == 1023) s_base = run;
        __syncthreads();
    }
    if (tid == 0) {
        w.tile_off[w.n_tiles] = (int64_t)s_base;
        w.result->n_tokens = (int64_t)s_base;
        w.result->n_long = *w.long_count;
    }
}

// ---------------------------------------------------------------------------------------------------
// pack: tiles' tokens -> one packed stream; per-document token offsets
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pack_tokens(JtkWork w) {
    const int64_t tile = blockIdx.x;
    const uint32_t cnt = w.tile_cnt[tile];
    const int32_t* src = w.tmp_tok + w.tile_fs[tile];
    int32_t* dst = w.tokens + w.tile_off[tile];
    for (uint32_t i = threadIdx.x; i < cnt; i += 256) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) k_doc_offsets(JtkWork w) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > w.n_docs) return;
    const int64_t q = w.doc_off[d];
    const int64_t tile = q / MT;
    const int64_t fs = w.tile_fs[tile];
    const int64_t wd = q >> 6;
    uint64_t m = w.tokmask[wd] & ((1ull << (q & 63)) - 1ull);
    uint32_t pre = 0;
    if ((fs >> 6) == wd) m &= ~((1ull << (fs & 63)) - 1ull);       // bits before fs belong to the previous tile
    else pre = w.blk_pre[wd];
    w.tok_off[d] = w.tile_off[tile] + pre + __popcll(m);
    if (d < w.n_docs) {
        const int32_t st = w.status[d];
        if (st < 0) atomicMin(&w.result->worst_status, st);
    }
}

}  // namespace

void jtk_launch_mark_docs(const JtkWork& w, hipStream_t s) {
    const int64_t n = w.n_docs + 1;
    hipLaunchKernelGGL(k_mark_docs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w);
}
void jtk_launch_special_check(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s) {
    if (t.n_specials == 0 || w.n_bytes == 0) return;
    const int64_t threads = (w.n_bytes + 15) / 16;
    hipLaunchKernelGGL(k_special_check, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, w, t);
}
void jtk_launch_pretok_split(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s) {
    const int64_t tiles = (w.n_bytes + 1 + ST - 1) / ST;
    if (t.kind == JTK_PAT_CL100K) hipLaunchKernelGGL(k_pretok_split<JTK_PAT_CL100K>, dim3((unsigned)tiles), dim3(256), 0, s, w, t);
    else hipLaunchKernelGGL(k_pretok_split<JTK_PAT_R50K>, dim3((unsigned)tiles), dim3(256), 0, s, w, t);
}
void jtk_launch_bpe_merge(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s) {
    hipLaunchKernelGGL(k_bpe_merge, dim3((unsigned)w.n_tiles), dim3(256), 0, s, w, t);
}
void jtk_launch_bpe_merge_long(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s) {
    hipLaunchKernelGGL(k_bpe_merge_long, dim3(256), dim3(64), 0, s, w, t);
}
void jtk_launch_tile_scan(const JtkWork& w, hipStream_t s) {
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, w);
}
void jtk_launch_pack(const JtkWork& w, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_tokens, dim3((unsigned)w.n_tiles), dim3(256), 0, s, w);
    const int64_t n = w.n_docs + 1;
    hipLaunchKernelGGL(k_doc_offsets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w);
}
