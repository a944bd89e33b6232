// jtk_strip_encode.h -- k_strip_encode: from the piece mask of a strip of text to its tokens in ONE pass, and
// k_strip_gather, which moves every strip's tokens to their place in the batch's packed output.  Included by
// jtk_kernels.hip inside its anonymous namespace.
//
// This is the loop of encodeOrdinaryInternal (GptBytePairEncoding.java:77-87) -- for every piece: the whole-piece lookup
// (:81-83), else bytePairMerge (:84-86, :200-275), tokens appended in text order -- with nothing written in between: the
// piece list, the pieces that need merging and their results never leave the wave.
//
// ONE WAVE PER STRIP of 4096 bytes = 64 piece-mask words, one per lane.  The wave is a small pipeline over the strip's
// pieces, 64 at a time ("chunks"):
//   resolve  lane j of chunk c takes piece 64 c + j: up to 16 bytes straight from the text (one unaligned 16-byte load: the
//            strip is in L1/L2, neighbouring lanes read overlapping bytes), one probe of the whole-piece tables (tok8 / tok16,
//            primary-first: one scattered fetch; a second round only for lanes that miss in a flagged slot).  The answer is
//            ONE word per piece, kept in a register ring indexed by the chunk (ring[c % 16], lane j): token id, or "hard"
//            (no table entry: its place in the pending list), or "long" (> 16 bytes: merged earlier, tokens in htok), or
//            "gap" (custom patterns: unmatched text).
//   merge    when 64 hard pieces are pending (or nothing is left to resolve) one lane per pending piece runs bytePairMerge
//            (jtk_lean_merge.h) with its parts in the wave's LDS; the results stay there.
//   pack     all pieces before the first one that is still pending: token counts (1, or the merged piece's), a wave prefix
//            sum, tokens stored at the strip's place in `stok` (dense from the strip's first word) and, for a piece that
//            starts a document, the tokens of the strip before it (docpre).
// The waves of a workgroup share only the read-only tables in LDS (2-byte-token ranks, byte -> id): no barrier after the
// prologue.  A workgroup keeps its CU for the whole launch; its waves take strips round robin.
// What leaves the wave per strip: the tokens (4 bytes each, to stok), one count (tile_tot), docpre for document starts.
#ifndef JTK_ENC_WAVES
#define JTK_ENC_WAVES 12
#endif
constexpr int ENC_WAVES = JTK_ENC_WAVES, ENC_THREADS = 64 * ENC_WAVES;
constexpr int ENC_RING = 16;                   // chunks whose answers wait in registers
constexpr int ENC_WIN = 1024;                  // piece starts listed in LDS at a time
#ifndef JTK_ENC_G
#define JTK_ENC_G 2
#endif
constexpr int ENC_G = JTK_ENC_G;               // chunks resolved per step (loads in flight per lane)
constexpr int ENC_PEND = 256;                  // pending hard pieces (a ring: at most 63 + 64 ENC_G wait)
static_assert(63 + 64 * ENC_G < ENC_PEND && ENC_WIN % (64 * ENC_G) == 0, "pending ring / window");
static_assert(T == 4096, "a strip is 64 mask words: one per lane");

// ring word of a piece
constexpr uint32_t RW_S_MASK = 0xFFFu;         // bits 0..11: byte offset of the piece in the strip
constexpr uint32_t RW_DOC = 1u << 12;          // a document starts with this piece
constexpr int RW_KIND_SHIFT = 13;              // bits 13..14
constexpr uint32_t RW_TOKEN = 0u, RW_HARD = 1u, RW_LONG = 2u, RW_GAP = 3u;
constexpr int RW_PAY_SHIFT = 15;               // bits 15..31: token id, or the piece's index in the pending ring

struct __attribute__((aligned(16))) EncWaveLds {
    uint32_t id[16 * 64];                      // parts of the pieces being merged: token ids, [slot][lane] ...
    uint32_t rk[16 * 64];                      // ... and pair keys; after a round rk[lane] = the lane's live-part mask
    uint32_t pend[ENC_PEND];                   // pending hard pieces: piece index (12) | offset << 12 | (len - 1) << 24
    uint16_t starts[ENC_WIN + 8];             // byte offsets of pieces k0 .. k0 + ENC_WIN (one more: where the last ends)
};

struct __attribute__((packed, aligned(1))) U4Bytes { uint32_t x, y, z, w; };

// the 16 bytes at text position p (bytes at or beyond n read as zero); p + 16 <= n is the fast path
__device__ __forceinline__ uint4 load_text16(const uint8_t* text, int64_t p, int64_t n) {
    if (p + 16 <= n) {
        const U4Bytes v = *reinterpret_cast<const U4Bytes*>(text + p);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    uint32_t tmp[4] = {0, 0, 0, 0};
    for (int j = 0; j < 16; j++) if (p + j < n) tmp[j >> 2] |= (uint32_t)text[p + j] << (8 * (j & 3));
    return make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
}

// inclusive prefix sum across the wave with DPP row shifts and row broadcasts (six adds)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);     // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);     // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);     // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);     // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);    // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);    // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) { return mbcnt64_(m); }

#ifdef JTK_ENC_STAMP
// diagnostic build: wave cycles per phase, summed over the launch (never read by the kernels; jtk_debug_stamps() fetches them)
__device__ unsigned long long g_enc_stamp[16];
#define STAMP_BEGIN() const uint64_t st_t0 = __builtin_amdgcn_s_memtime()
#define STAMP_END(i) st_acc[i] += __builtin_amdgcn_s_memtime() - st_t0
#define STAMP_ADD(i, v) st_acc[i] += (uint64_t)(v)
#else
#define STAMP_BEGIN()
#define STAMP_END(i)
#define STAMP_ADD(i, v)
#endif

__global__ void __launch_bounds__(ENC_THREADS) k_strip_encode(JtkWork w, JtkDeviceTables t) {
    __shared__ uint64_t s_bpbits[1024];
    __shared__ uint32_t s_bpranks[JTK_BP_MAX];
    __shared__ uint16_t s_bpcum[1024];
    __shared__ uint32_t s_brank[256];
    __shared__ EncWaveLds s_wave[ENC_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 1024; i += ENC_THREADS) { s_bpbits[i] = t.bp.bits[i]; s_bpcum[i] = t.bp.cum[i]; }
    for (int i = tid; i < JTK_BP_MAX; i += ENC_THREADS) s_bpranks[i] = t.bp.ranks[i];
    if (tid < 256) s_brank[tid] = t.byte_rank[tid];
    __syncthreads();
    // (from here on the waves are independent: no workgroup barrier)
    EncWaveLds& W = s_wave[wv];
    uint32_t* const id = W.id + lane;
    uint32_t* const rk = W.rk + lane;
    const LeanLds LL{W.id, W.rk, JtkBpLds{s_bpbits, s_bpcum, s_bpranks}, s_brank};
    const int64_t n = w.n_bytes;
    const bool gaps = w.gapmask != nullptr;
    const bool store = w.count_only == 0;
    const uint8_t* const tok = reinterpret_cast<const uint8_t*>(t.tok8.slots);   // the tok8 slots, then the tok16 slots: one allocation
    const uint32_t rel16 = (uint32_t)(reinterpret_cast<const uint8_t*>(t.tok16.slots) - tok);
#ifdef JTK_ENC_STAMP
    uint64_t st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t st_wave0 = __builtin_amdgcn_s_memtime();
#endif

    for (int64_t strip = (int64_t)wv * gridDim.x + blockIdx.x; strip < w.n_tiles; strip += (int64_t)gridDim.x * ENC_WAVES) {
        const int64_t B = strip * T;
        const int64_t wd = (B >> 6) + lane;
        // this lane's mask words: piece starts (without the end sentinel), unmatched text, document starts
        uint64_t pm = piece_word(w, wd);
        if (wd * 64 + 63 >= n) pm &= (wd * 64 >= n) ? 0ull : ((1ull << (n - wd * 64)) - 1ull);
        const uint64_t gm = (gaps && wd < w.n_words) ? w.gapmask[wd] : 0ull;
        const uint64_t dm = (wd < w.n_words) ? w.docmask[wd] : 0ull;
        // (touch the strip's text -- one word of every 64-byte block -- so that it is on its way while the masks are scanned)
        const uint32_t touch = (B + lane * 64 < n) ? *reinterpret_cast<const uint32_t*>(w.text + B + lane * 64) : 0u;
        const uint32_t cnt = (uint32_t)__popcll(pm);
        const uint32_t inc = wave_incl_scan_dpp(cnt);
        const int np = (int)(uint32_t)__shfl((int)inc, 63);
        const uint32_t pre = inc - cnt;
        if (np == 0) {                                                   // a strip inside one long piece
            if (lane == 0) w.tile_tot[strip] = 0;
            continue;
        }
        // where the strip's last piece ends, relative to B: the end sentinel or the next strip's first piece (only "more
        // than 16 bytes away" matters beyond that: such a piece was merged by k_bpe_merge, its length is not needed here)
        uint32_t end_rel;
        {
            const uint64_t nw = piece_word(w, (B >> 6) + 64 + lane);
            const uint64_t some = __ballot(nw != 0);
            const int first = some ? jtk_ctz64(some) : 0;
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)nw, first), hi = (uint32_t)__shfl((int)(uint32_t)(nw >> 32), first);
            const uint64_t fw = ((uint64_t)hi << 32) | lo;
            end_rel = some ? (uint32_t)(T + first * 64 + jtk_ctz64(fw)) : (uint32_t)(2 * T);
            if (n - B < T) end_rel = (uint32_t)(n - B);                      // the sentinel is inside this strip
        }

        // ---- the wave's state: all wave-uniform
        int k0 = -ENC_WIN;                        // first piece of the listed window
        int k_res = 0;                            // pieces resolved (a multiple of 64 until the end)
        int k_pack = 0;                           // pieces packed
        uint32_t pn_head = 0, pn_tail = 0;        // pending ring: [head, tail) wait for a merge round
        uint32_t round_base = 0;                  // the pending index that lane 0 of the last merge round took
        uint32_t run = 0;                         // tokens of the strip so far
        uint32_t ring[ENC_RING];
#pragma unroll
        for (int i = 0; i < ENC_RING; i++) ring[i] = 0;
        uint32_t* const out = w.stok + B;

        while (k_pack < np) {
            // ---- resolve chunks while the pending list is short of a full round and the ring has room: ENC_G chunks per
            // step, so that a lane has ENC_G text loads and then ENC_G table probes in flight (what bounds this phase is the
            // latency of those two dependent loads, not their number)
            while (k_res < np && pn_tail - pn_head < 64u && (k_res >> 6) - (k_pack >> 6) < ENC_RING) {
                STAMP_BEGIN();
                if (k_res >= k0 + ENC_WIN) {
                    // list the starts of pieces k0 .. k0 + ENC_WIN (each lane: the set bits of its word)
                    k0 += ENC_WIN;
                    wave_lds_fence();
                    uint32_t i = pre;
                    for (uint64_t m = pm; m; m &= m - 1, i++) {
                        const int rel = (int)i - k0;
                        if (rel >= 0 && rel <= ENC_WIN) W.starts[rel] = (uint16_t)(lane * 64 + jtk_ctz64(m));
                    }
                    if (lane == 0 && np - k0 <= ENC_WIN) W.starts[np - k0] = (uint16_t)end_rel;
                    wave_lds_fence();
                }
                // chunks of this step: as many as the ring, the listed window and the strip allow
                int g = ENC_RING - ((k_res >> 6) - (k_pack >> 6));
                g = min(g, min((k0 + ENC_WIN - k_res) >> 6, (np - k_res + 63) >> 6));
                g = min(g, ENC_G);
                bool have[ENC_G], shortp[ENC_G], big[ENC_G];
                uint32_t s[ENC_G], len[ENC_G], key0[ENC_G], key1[ENC_G], key2[ENC_G], key3[ENC_G], mix[ENC_G];
                uint4 tx[ENC_G], ka[ENC_G];
                uint2 ma[ENC_G];
                // (chunks q >= g are computed too -- on piece 0, results dropped -- so that there is no branch between the loads)
                bool tail = false;
#pragma unroll
                for (int q = 0; q < ENC_G; q++) {
                    const int k = k_res + 64 * q + lane;
                    have[q] = q < g && k < np;
                    const int rel = have[q] ? k - k0 : 0;
                    const uint32_t e = W.starts[rel + 1];
                    s[q] = W.starts[rel];
                    const uint32_t plen = have[q] ? e - s[q] : 0u;                   // (> 16: only "long" matters)
                    shortp[q] = plen <= (uint32_t)JTK_SHORT_MAX;
                    len[q] = shortp[q] ? plen : 0u;
                    tail = tail || B + s[q] + 16 > n;
                }
                // up to 16 bytes of every piece: one unaligned load each, back to back (only the last bytes of the text need care)
                if (!__ballot(tail)) {
#pragma unroll
                    for (int q = 0; q < ENC_G; q++) {
                        const U4Bytes v = *reinterpret_cast<const U4Bytes*>(w.text + B + s[q]);
                        tx[q] = make_uint4(v.x, v.y, v.z, v.w);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < ENC_G; q++) tx[q] = load_text16(w.text, B + s[q], n);
                }
                auto slot_off = [&](int q, uint32_t mx) -> uint32_t {
                    const uint32_t h = jtk_reduce32(mx, big[q] ? t.tok16.n : t.tok8.bits);
                    return big[q] ? (h << 5) + rel16 : (h << 4);
                };
#pragma unroll
                for (int q = 0; q < ENC_G; q++) {
                    const uint64_t runm = ~0ull >> ((0u - 8u * len[q]) & 63u);       // 8 len ones (len 8 and 16: all 64)
                    big[q] = len[q] > 8u;
                    const uint64_t mlo = big[q] ? ~0ull : runm, mhi = big[q] ? runm : 0ull;
                    key0[q] = tx[q].x & (uint32_t)mlo; key1[q] = tx[q].y & (uint32_t)(mlo >> 32);
                    key2[q] = tx[q].z & (uint32_t)mhi; key3[q] = tx[q].w & (uint32_t)(mhi >> 32);
                    // one mix for both tables and both choices; only base, slot size and slot count depend on the length
                    mix[q] = jtk_tok16_mix(key0[q], key1[q], key2[q], key3[q], len[q]);
                    const uint8_t* sa = tok + slot_off(q, mix[q]);
                    ka[q] = *reinterpret_cast<const uint4*>(sa);                     // tok8: lo, hi, id, len; tok16: the 16 key bytes
                    ma[q] = make_uint2(0, 0);
                    if (big[q]) ma[q] = *reinterpret_cast<const uint2*>(sa + 16);    // tok16: id, len
                }
                auto check = [&](int q, uint32_t& idv, bool& more) {
                    const uint32_t slen = big[q] ? ma[q].y : ka[q].w;
                    const uint32_t diff = (ka[q].x ^ key0[q]) | (ka[q].y ^ key1[q]) | ((slen & JTK_TOK_LEN_MASK) ^ len[q]) |
                                          (big[q] ? ((ka[q].z ^ key2[q]) | (ka[q].w ^ key3[q])) : 0u);
                    idv = diff == 0u ? (big[q] ? ma[q].x : ka[q].z) : JTK_RANK_NONE;
                    // (the slot's filter says whether a key with this mix can be among those it turned away)
                    more = diff != 0u && (slen & JTK_TOK_FILTER_BIT(mix[q])) != 0u && len[q] != 0u;
                };
                uint32_t tokid[ENC_G];
                bool more[ENC_G], any_more = false;
#pragma unroll
                for (int q = 0; q < ENC_G; q++) { check(q, tokid[q], more[q]); any_more = any_more || more[q]; }
                if (__ballot(any_more)) {                                            // secondary slots, for the lanes that need them
#pragma unroll
                    for (int q = 0; q < ENC_G; q++) {
                        if (more[q]) {
                            const uint8_t* sa = tok + slot_off(q, jtk_pair_mix2(mix[q]));
                            ka[q] = *reinterpret_cast<const uint4*>(sa);
                            if (big[q]) ma[q] = *reinterpret_cast<const uint2*>(sa + 16);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < ENC_G; q++) { bool dummy; if (more[q]) check(q, tokid[q], dummy); }
                }
#pragma unroll
                for (int q = 0; q < ENC_G; q++) {
                    bool gap = false;
                    if (gaps) {                                                      // (wave-uniform; the shuffles are evaluated by all lanes)
                        const uint32_t glo = (uint32_t)__shfl((int)(uint32_t)gm, (int)(s[q] >> 6)), ghi = (uint32_t)__shfl((int)(uint32_t)(gm >> 32), (int)(s[q] >> 6));
                        gap = have[q] && (((s[q] & 32u) ? ghi : glo) >> (s[q] & 31u)) & 1u;
                    }
                    const uint32_t dlo = (uint32_t)__shfl((int)(uint32_t)dm, (int)(s[q] >> 6)), dhi = (uint32_t)__shfl((int)(uint32_t)(dm >> 32), (int)(s[q] >> 6));
                    const bool isdoc = have[q] && (((s[q] & 32u) ? dhi : dlo) >> (s[q] & 31u)) & 1u;
                    const bool hit = have[q] && shortp[q] && tokid[q] != JTK_RANK_NONE && !gap;
                    const bool hard = have[q] && shortp[q] && !hit && !gap;
                    const uint64_t hb = __ballot(hard);
                    const uint32_t pidx = pn_tail + mbcnt64(hb);
                    if (hard) W.pend[pidx & (ENC_PEND - 1)] = (uint32_t)(k_res + 64 * q + lane) | (s[q] << 12) | ((len[q] - 1u) << 24);
                    pn_tail += (uint32_t)__popcll(hb);
                    const uint32_t kind = gap ? RW_GAP : hit ? RW_TOKEN : hard ? RW_HARD : RW_LONG;
                    const uint32_t pay = hit ? tokid[q] : (pidx & (ENC_PEND - 1));
                    if (q < g)
                        ring[__builtin_amdgcn_readfirstlane(((k_res >> 6) + q) & (ENC_RING - 1))] = s[q] | (isdoc ? RW_DOC : 0u) | (kind << RW_KIND_SHIFT) | (pay << RW_PAY_SHIFT);
                }
                k_res += 64 * g;
                STAMP_END(1);
                STAMP_ADD(4, g);
            }
            // ---- one merge round: a lane per pending piece (GptBytePairEncoding.java:200-275)
            const uint32_t nround = min(64u, pn_tail - pn_head);
            if (nround) {
                STAMP_BEGIN();
                STAMP_ADD(5, 1);
                STAMP_ADD(6, nround);
                wave_lds_fence();                                                // the pending entries; the last round's results are consumed
                const bool mine = (uint32_t)lane < nround;
                const uint32_t pe = mine ? W.pend[(pn_head + (uint32_t)lane) & (ENC_PEND - 1)] : 0u;
                const uint32_t s = (pe >> 12) & 0xFFFu;
                const int len = mine ? (int)(pe >> 24) + 1 : 0;
                const uint4 tx = load_text16(w.text, B + s, n);
                const uint32_t d4[4] = {tx.x, tx.y, tx.z, tx.w};
                uint32_t b[17];
#pragma unroll
                for (int j = 0; j < 16; j++) b[j] = (d4[j >> 2] >> (8 * (j & 3))) & 255u;
                b[16] = 0;
                uint32_t alive;
                // the round's longest piece picks the unrolled variant: 8, 12 or 16 slots
                if (!__ballot(len > 8)) { uint32_t c[9]; for (int j = 0; j < 9; j++) c[j] = b[j]; alive = lean_piece16<8, 64>(LL, id, rk, c, len, t); }
                else if (!__ballot(len > 12)) { uint32_t c[13]; for (int j = 0; j < 13; j++) c[j] = b[j]; alive = lean_piece16<12, 64>(LL, id, rk, c, len, t); }
                else alive = lean_piece16<16, 64>(LL, id, rk, b, len, t);
                rk[0] = alive;                                                   // (slot 0 of the lane's key column is free now)
                wave_lds_fence();
                round_base = pn_head;
                pn_head += nround;
                STAMP_END(2);
            }
            // ---- pack every piece before the first one that still waits for a merge round
            int k_bound = k_res < np ? k_res : np;
            if (pn_head != pn_tail) k_bound = (int)(W.pend[pn_head & (ENC_PEND - 1)] & 0xFFFu);
            k_bound = __builtin_amdgcn_readfirstlane(k_bound);
            STAMP_BEGIN();
            for (int c = k_pack >> 6; c * 64 < k_bound; c++) {
                STAMP_ADD(7, 1);
                const int k = c * 64 + lane;
                const bool act = k >= k_pack && k < k_bound;
                const uint32_t rw = ring[__builtin_amdgcn_readfirstlane(c & (ENC_RING - 1))];
                const uint32_t kind = (rw >> RW_KIND_SHIFT) & 3u, pay = rw >> RW_PAY_SHIFT, s = rw & RW_S_MASK;
                const uint64_t bact = __ballot(act);
                uint32_t pos, total;
                if (!__ballot(act && kind != RW_TOKEN)) {
                    // every piece a table entry: one token each
                    pos = run + mbcnt64(bact);
                    total = (uint32_t)__popcll(bact);
                    if (act && store) out[pos] = pay;
                } else {
                    const bool ishard = act && kind == RW_HARD, islong = act && kind == RW_LONG;
                    const uint32_t m = (pay - round_base) & (ENC_PEND - 1);          // the lane that merged this piece
                    uint32_t alive = ishard ? W.rk[m] : 0u;
                    uint32_t cn = act ? (kind == RW_TOKEN ? 1u : kind == RW_HARD ? (uint32_t)__popc(alive) : 0u) : 0u;
                    uint32_t hd = 0;
                    if (__ballot(islong)) {
                        if (islong) {
                            hd = w.htok[B + s];
                            cn = (hd >> JTK_HT_CNT_SHIFT) & JTK_HT_CNT_MASK;
                            if (cn == JTK_HT_ESCAPE) cn = w.docpre[B + s + 1];       // giant piece
                        }
                    }
                    const uint32_t inc2 = wave_incl_scan_dpp(cn);
                    pos = run + inc2 - cn;
                    total = (uint32_t)__shfl((int)inc2, 63);
                    if (store) {
                        if (act && kind == RW_TOKEN) out[pos] = pay;
                        // merged pieces: the ids of their live parts, in order
                        uint32_t o = pos;
                        while (__ballot(alive != 0u)) {
                            if (alive) {
                                const uint32_t j = (uint32_t)__builtin_ctz(alive);
                                alive &= alive - 1u;
                                out[o++] = W.id[j * 64 + m];
                            }
                        }
                        if (__ballot(islong)) {
                            if (islong && cn) {
                                out[pos] = hd & JTK_HT_ID_MASK;
                                const uint32_t* src = w.htok + B + s;
                                for (uint32_t i = 1; i < cn; i++) out[pos + i] = src[i] & JTK_HT_ID_MASK;
                            }
                        }
                    }
                }
                // document starts among these pieces: tokens of the strip before them
                if (__ballot(act && (rw & RW_DOC))) { if (act && (rw & RW_DOC)) w.docpre[B + s] = pos; }
                run += total;
            }
            STAMP_END(3);
            k_pack = k_bound;
        }
        asm volatile("" ::"v"(touch));
        if (lane == 0) w.tile_tot[strip] = run;
        STAMP_ADD(8, 1);
        STAMP_ADD(9, np);
    }
#ifdef JTK_ENC_STAMP
    st_acc[0] = __builtin_amdgcn_s_memtime() - st_wave0;
    if (lane == 0) for (int i = 0; i < 16; i++) atomicAdd(&g_enc_stamp[i], (unsigned long long)st_acc[i]);
#endif
}

// ---------------------------------------------------------------------------------------------------
// strip_gather: the strips' tokens (stok, dense per strip) to their place in the batch's packed output, once the
// exclusive scan of the strips' counts (k_tile_scan) has said where that is.  One wave per strip; pure data movement.
// ---------------------------------------------------------------------------------------------------
constexpr int GATHER_THREADS = 256;

__global__ void __launch_bounds__(GATHER_THREADS) k_strip_gather(JtkWork w) {
    const int lane = threadIdx.x & 63;
    const int64_t strip = (int64_t)blockIdx.x * (GATHER_THREADS / 64) + (threadIdx.x >> 6);
    if (strip >= w.n_tiles) return;
    const uint32_t total = w.tile_tot[strip];
    int64_t base;
    if (w.inline_scan) {
        // a small job (at most 1024 strips): the tokens before this strip, added up here -- one launch less
        uint32_t part = 0;
        for (int64_t i = lane; i < strip; i += 64) part += w.tile_tot[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d);
        const int64_t job_before = *w.job_tokens;
        base = job_before + (int64_t)part;
        if (lane == 0) {
            w.tile_off[strip] = base;
            if (strip == w.n_tiles - 1) {
                const int64_t end = base + (int64_t)total;
                w.tile_off[w.n_tiles] = end;
                w.set_info[0] = job_before;
                w.set_info[1] = end;
                w.result->n_tokens = end;
                *w.job_tokens_next = end;
            }
        }
    } else base = w.tile_off[strip];
    if (w.count_only) return;
    const uint32_t* src = w.stok + strip * T;
    uint32_t* dst = reinterpret_cast<uint32_t*>(w.tokens) + base;
    // 16 bytes per lane where the destination allows: a head of up to three words, aligned quads, a tail
    const uint32_t head = min(total, (uint32_t)((4u - (uint32_t)(base & 3)) & 3u));
    if ((uint32_t)lane < head) dst[lane] = src[lane];
    const uint32_t nq = (total - head) >> 2;
    struct __attribute__((packed, aligned(4))) U4Words { uint32_t x, y, z, w; };
    for (uint32_t q = (uint32_t)lane; q < nq; q += 64) {
        const U4Words v = *reinterpret_cast<const U4Words*>(src + head + 4 * q);
        *reinterpret_cast<uint4*>(dst + head + 4 * q) = make_uint4(v.x, v.y, v.z, v.w);
    }
    const uint32_t done = head + 4 * nq;
    if (done + (uint32_t)lane < total) dst[done + lane] = src[done + lane];
}
