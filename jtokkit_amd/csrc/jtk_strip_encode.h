// jtk_strip_encode.h -- k_strip_encode: from the piece mask of a strip of text to its tokens without a piece list, a
// queue or a merge result ever leaving the wave, and k_strip_expand, which writes the packed token stream.  Included by
// jtk_kernels.hip inside its anonymous namespace.
//
// This is the loop of encodeOrdinaryInternal (GptBytePairEncoding.java:77-87) -- for every piece: the whole-piece lookup
// (:81-83), else bytePairMerge (:84-86, :200-275), tokens appended in text order.
//
// ONE WAVE PER STRIP of 4096 bytes = 64 piece-mask words, one per lane; a wave takes strips round robin for the whole
// launch.  Work is split by how often it is needed, so that the common case costs few instructions and the rare cases run
// 64 lanes wide:
//   main path   lane j of chunk c takes piece 64 c + j.  A piece of <= 8 bytes probes its PRIMARY slot of the tok8 table
//               (one unaligned 8-byte load of the text, one 16-byte fetch of the slot).  A hit is a "dense" piece: its token
//               goes straight to the strip's dense token block (stok), in order.  Everything else -- a piece of more than
//               8 bytes, an entry displaced to its secondary slot, a piece that is no entry -- is a HOLE: a bit in the
//               strip's hole bitmap and an entry (strip, hole number, offset, length) in the wave's pending ring in LDS.
//   hole batch  when 64 holes are pending, one lane per hole: the complete whole-piece lookup (tok8 / tok16, primary and
//               secondary slot), pieces of more than 16 bytes (merged earlier by k_bpe_merge: count from htok), unmatched
//               text of custom patterns, and pieces of 2..3 bytes that are no entry (at most one merge, from the LDS tables).
//               Each result is an 8-byte hole record (hrec); a piece that needs bytePairMerge moves on to the hard ring.
//   merge round when 64 hard pieces are pending, one lane per piece runs bytePairMerge (jtk_lean_merge.h) with its parts
//               in the wave's LDS; the result is the piece's hole record (more than three tokens: in htok).
// The rings live across strips, so batches and rounds are full except for the last ones of a wave.  Holes do not hold up
// the dense tokens: k_strip_expand merges both streams.
// What leaves the wave per strip: dense tokens (4 B each), the hole bitmap (512 B), hole records (8 B each), the piece
// count; token counts are added to tile_tot.
#ifndef JTK_ENC_WAVES
#define JTK_ENC_WAVES 12
#endif
#ifndef JTK_ENC_WGS_PER_CU
#define JTK_ENC_WGS_PER_CU 2
#endif
constexpr int ENC_WAVES = JTK_ENC_WAVES, ENC_THREADS = 64 * ENC_WAVES, ENC_WGS_PER_CU = JTK_ENC_WGS_PER_CU;
constexpr int ENC_WIN = 512;                   // piece starts listed in LDS at a time
#ifndef JTK_ENC_G
#define JTK_ENC_G 2
#endif
constexpr int ENC_G = JTK_ENC_G;               // chunks per step of the main path (loads in flight per lane)
constexpr int ENC_PEND = 256;                  // ring of pending holes (at most 63 + 64 ENC_G wait)
static_assert(63 + 64 * ENC_G < ENC_PEND && ENC_WIN % (64 * ENC_G) == 0, "pending ring / window");
static_assert(T == 4096, "a strip is 64 mask words: one per lane");

struct __attribute__((aligned(16))) EncWaveLds {
    uint32_t ho_sl[ENC_PEND];                  // pending holes: offset in the strip (12) | (min(len, 17) - 1) << 12,
    uint32_t ho_strip[ENC_PEND];               //                their strip,
    uint32_t ho_idx[ENC_PEND];                 //                their index in the wave's region of stok / hrec
    uint16_t starts[ENC_WIN + 8];              // byte offsets of pieces k0 .. k0 + ENC_WIN (one more: where the last ends)
};

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

constexpr int ENC_WAVES_PER_EU = ENC_WAVES * ENC_WGS_PER_CU / 4;
__global__ void __launch_bounds__(ENC_THREADS) __attribute__((amdgpu_waves_per_eu(ENC_WAVES_PER_EU, ENC_WAVES_PER_EU))) k_strip_encode(JtkWork w, JtkDeviceTables t) {
    __shared__ uint64_t s_bpbits[1024];
    __shared__ uint32_t s_bpranks[JTK_BP_MAX];
    __shared__ uint16_t s_bpcum[1024];
    __shared__ uint32_t s_brank[256];
    __shared__ EncWaveLds s_wave[ENC_WAVES];
    __shared__ uint32_t s_qcount[JTK_NBINS];       // entries this workgroup has put in its queue of each bin

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 1024; i += ENC_THREADS) { s_bpbits[i] = t.bp.bits[i]; s_bpcum[i] = t.bp.cum[i]; }
    for (int i = tid; i < JTK_BP_MAX; i += ENC_THREADS) s_bpranks[i] = t.bp.ranks[i];
    if (tid < 256) s_brank[tid] = t.byte_rank[tid];
    if (tid < JTK_NBINS) s_qcount[tid] = 0;
    __syncthreads();
    // (from here on the waves are independent: no workgroup barrier until the end)
    EncWaveLds& W = s_wave[wv];
    const JtkBpLds bp{s_bpbits, s_bpcum, s_bpranks};
    const int64_t n = w.n_bytes;
    const bool gaps = w.gapmask != nullptr;
    const uint8_t* const tok = reinterpret_cast<const uint8_t*>(t.tok8.slots);   // the tok8 slots, then the tok16 slots: one allocation
    const uint32_t rel16 = (uint32_t)(reinterpret_cast<const uint8_t*>(t.tok16.slots) - tok);
    const uint32_t n8 = t.tok8.bits, n16 = t.tok16.n;
#ifdef JTK_ENC_STAMP
    uint64_t st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t st_wave0 = __builtin_amdgcn_s_memtime();
#endif
    uint4* const memo = w.memo ? w.memo + (size_t)xcc_id() * ((size_t)w.memo_mask + 1u) * 2u : nullptr;   // this XCD's table
    uint32_t ho_head = 0, ho_tail = 0;            // pending holes  [head, tail)   (wave-uniform)
    const int64_t region0 = ((int64_t)blockIdx.x * ENC_WAVES + wv) * (int64_t)w.wave_cap;   // this wave's region of stok / hrec
    uint32_t cursor = 0;                          // ... filled so far

    // ---- one hole batch: a lane per pending hole -- everything the main path does not do
    auto hole_batch = [&](uint32_t nl) {
#ifdef JTK_EXP_NOBATCH                                                        // (timing experiment: results are wrong)
        ho_head += nl;
        return;
#endif
        STAMP_BEGIN();
        STAMP_ADD(10, 1);
        STAMP_ADD(11, nl);
        wave_lds_fence();
        const bool mine = (uint32_t)lane < nl;
        const uint32_t ri = (ho_head + (uint32_t)lane) & (ENC_PEND - 1);
        const uint32_t pe_x = W.ho_sl[ri], strip = W.ho_strip[ri];
        const int64_t h = region0 + W.ho_idx[ri];                               // index of the hole's slot and record
        const uint32_t s = pe_x & 0xFFFu;
        const uint32_t lenc = mine ? ((pe_x >> 12) & 31u) + 1u : 0u;             // 17: more than 16 bytes
        const int64_t pos = mine ? (int64_t)strip * T + s : 0;
        const bool shortp = lenc <= (uint32_t)JTK_SHORT_MAX;
        const uint32_t len = shortp ? lenc : 0u;
        // the whole-piece lookup (:81-83), complete: tok8 or tok16, primary slot, then the secondary one where the primary says so
        const uint4 tx = load_text16(w.text, pos, n);
        const uint64_t runm = ~0ull >> ((0u - 8u * len) & 63u);                  // 8 len ones (len 8 and 16: all 64)
        const bool big = len > 8u;
        const uint64_t mlo = big ? ~0ull : runm, mhi = big ? runm : 0ull;
        const uint32_t key0 = tx.x & (uint32_t)mlo, key1 = tx.y & (uint32_t)(mlo >> 32);
        const uint32_t key2 = tx.z & (uint32_t)mhi, key3 = tx.w & (uint32_t)(mhi >> 32);
        const uint32_t mix = jtk_tok16_mix(key0, key1, key2, key3, len);
        uint4 ka;
        uint2 ma = make_uint2(0, 0);
        auto fetch = [&](uint32_t mx) {
            const uint32_t hh = jtk_reduce32(mx, big ? n16 : n8);
            const uint8_t* sa = tok + (big ? (hh << 5) + rel16 : (hh << 4));
            ka = *reinterpret_cast<const uint4*>(sa);                            // tok8: lo, hi, id, len; tok16: the 16 key bytes
            if (big) ma = *reinterpret_cast<const uint2*>(sa + 16);              // tok16: id, len
        };
        auto check = [&](uint32_t& idv, bool& more) {
            const uint32_t slen = big ? ma.y : ka.w;
            const uint32_t diff = (ka.x ^ key0) | (ka.y ^ key1) | ((slen & JTK_TOK_LEN_MASK) ^ len) | (big ? ((ka.z ^ key2) | (ka.w ^ key3)) : 0u);
            idv = diff == 0u ? (big ? ma.x : ka.z) : JTK_RANK_NONE;
            more = diff != 0u && (slen & JTK_TOK_FILTER_BIT(mix)) != 0u && len != 0u;
        };
        uint32_t tokid;
        bool more;
        fetch(mix);
        check(tokid, more);
        if (__ballot(more)) {
            if (more) { bool dummy; fetch(jtk_pair_mix2(mix)); check(tokid, dummy); }
        }
        bool gap = false;
        if (gaps) gap = mine && ((w.gapmask[pos >> 6] >> (pos & 63)) & 1ull);
        const bool hit = mine && shortp && tokid != JTK_RANK_NONE && !gap;
        const bool islong = mine && !shortp && !gap;
        uint64_t rec = (uint64_t)tokid | (HR_TOKS << HR_KIND_SHIFT);            // one token
        uint32_t cnt = hit ? 1u : 0u;
        if (gap) rec = HR_GAP << HR_KIND_SHIFT;
        // a piece of more than 16 bytes: its exact length from the mask (the next piece start, the end sentinel at the latest)
        int64_t plen = len;
        if (__ballot(islong)) {
            if (islong) {
                int64_t v = pos >> 6;
                uint64_t m = piece_word(w, v);
                m = (pos & 63) == 63 ? 0ull : m & ~((2ull << (pos & 63)) - 1ull);
                while (m == 0 && ++v < w.n_words) m = piece_word(w, v);
                int64_t end = m ? v * 64 + jtk_ctz64(m) : n;
                if (end > n) end = n;
                plen = end - pos;
            }
        }
        bool hard = mine && shortp && !hit && !gap;
        // pieces of 2 or 3 bytes that are no entry: bytePairMerge makes no lookup that can hit beyond the 2-byte-token ranks of
        // their byte pairs: merge the pair of lower rank, the left one on a tie (:236), if either is a token
        const bool tiny = hard && len <= 3u;
        if (__ballot(tiny)) {
            if (tiny) {
                const uint32_t b0 = key0 & 255u, b1 = (key0 >> 8) & 255u, b2 = (key0 >> 16) & 255u;
                const bool three = len == 3u;
                const uint32_t r01 = three ? jtk_bp_lookup(bp, (b0 << 8) | b1) : JTK_RANK_NONE;
                const uint32_t r12 = three ? jtk_bp_lookup(bp, (b1 << 8) | b2) : JTK_RANK_NONE;
                const uint32_t i0 = s_brank[b0], i1 = s_brank[b1], i2 = s_brank[b2];
                uint32_t t0 = i0, t1 = i1, t2 = i2;
                cnt = three ? 3u : 2u;
                if (r01 != JTK_RANK_NONE && r01 <= r12) { t0 = r01; t1 = i2; cnt = 2u; }
                else if (r12 != JTK_RANK_NONE) { t1 = r12; cnt = 2u; }
                if (cnt == 2u) t2 = 0u;
                rec = (uint64_t)t0 | ((uint64_t)t1 << 17) | ((uint64_t)t2 << 34) | ((uint64_t)(cnt - 1u) << 51) | (HR_TOKS << HR_KIND_SHIFT);
            }
            hard = hard && !tiny;
        }
        // a piece this XCD has merged before: its tokens from the memo
        if (memo && __ballot(hard)) {
            const bool cand = hard && key0 != 0u;
            const uint4* e = memo + (size_t)(cand ? memo_slot(mix, w.memo_mask) : 0u) * 2u;
            const uint4 h0 = e[0], h1 = e[1];
            const uint32_t tag = memo_tag(mix);
            const bool ok = cand && ((h0.x ^ key0) | (h0.y ^ key1) | (h0.z ^ key2) | (h0.w ^ key3)) == 0u && (h1.y >> 19) == tag &&
                            (h1.w >> 27) == (tag & 31u) && ((h1.w >> 22) & 31u) == len;
            if (__ballot(ok)) {
                if (ok) {
                    const uint64_t lo = ((uint64_t)h1.y << 32) | h1.x, hi = ((uint64_t)h1.w << 32) | h1.z;
                    cnt = (uint32_t)(hi >> 51) & 7u;
                    if (cnt <= 3u) rec = (lo & ((1ull << 51) - 1ull)) | ((uint64_t)(cnt - 1u) << 51) | (HR_TOKS << HR_KIND_SHIFT);
                    else {
                        uint32_t* dst = w.htok + pos;
                        dst[0] = (uint32_t)lo & JTK_HT_ID_MASK; dst[1] = (uint32_t)(lo >> 17) & JTK_HT_ID_MASK; dst[2] = (uint32_t)(lo >> 34) & JTK_HT_ID_MASK;
                        dst[3] = (uint32_t)hi & JTK_HT_ID_MASK;
                        if (cnt > 4u) dst[4] = (uint32_t)(hi >> 17) & JTK_HT_ID_MASK;
                        if (cnt > 5u) dst[5] = (uint32_t)(hi >> 34) & JTK_HT_ID_MASK;
                        rec = (uint64_t)cnt | ((uint64_t)s << 21) | (HR_REF << HR_KIND_SHIFT);
                    }
                }
                hard = hard && !ok;
                STAMP_ADD(13, __popcll(__ballot(ok)));
            }
        }
        // what is left needs bytePairMerge: queued for k_bpe_merge by length bin, in this workgroup's own queues (the wave claims
        // its entries with one atomic in LDS per bin)
        int cls = -1;                                                            // 0..6 queue bin, 7 mid, 8 long, 9 giant
        if (hard) cls = len <= 8u ? 0 : len <= 12u ? 1 : 2;
        if (islong) {
            cls = plen <= 32 ? 3 : plen <= 64 ? 4 : plen <= 128 ? 5 : plen <= JTK_BIN_MAXLEN ? 6 : plen <= JTK_MID_CAP ? 7 : plen <= JTK_LONG_CAP ? 8
                  : plen <= JTK_GIANT_CAP ? 9 : -1;
            if (cls < 0) {
                // longer than the library accepts: the document gets a status, the piece no tokens
                const int64_t d = find_doc(w, pos);
                if (d >= 0) atomicMin(&w.status[d], -10 /* JTK_ERR_PIECE_TOO_LONG */);
                rec = HR_GAP << HR_KIND_SHIFT;
            }
        }
        const bool queued = cls >= 0;
        if (mine && !queued) put_hole(w, h, cnt, rec);
        add_strip_counts(w, strip, mine && !queued, cnt);
#ifdef JTK_EXP_NOQUEUE                                                        // (timing experiment: results are wrong)
        for (uint64_t todo = 0; todo;) {
#else
        for (uint64_t todo = __ballot(queued); todo;) {
#endif
            const int c = __builtin_amdgcn_readlane(cls, jtk_ctz64(todo));
            const uint64_t mask = __ballot(cls == c);
            todo &= ~mask;
            uint32_t base = 0;
            if (lane == jtk_ctz64(mask)) {
                const uint32_t cn = (uint32_t)__popcll(mask);
                base = c < JTK_NBINS ? atomicAdd(&s_qcount[c], cn) : atomicAdd(c == JTK_NBINS ? w.mid_count : c == JTK_NBINS + 1 ? w.long_count : w.n_giant, cn);
            }
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, jtk_ctz64(mask));
            if (cls == c) {
                const uint32_t i = base + mbcnt64(mask);
                if (c < JTK_NBINS) {
                    const uint64_t meta = (uint64_t)pos | ((uint64_t)(plen - 1) << JTK_QE_LEN_SHIFT);
                    const uint4 ent = make_uint4((uint32_t)meta, (uint32_t)(meta >> 32), (uint32_t)h, (uint32_t)(h >> 32));
                    if (c < JTK_NBINS_SHORT) {
                        // (the short bins' entries carry the piece's bytes: the merge kernel reads dense 32-byte entries, not the text)
                        uint4* e = w.qe[c] + ((int64_t)blockIdx.x * w.q_cap[c] + i) * 2;
                        e[0] = ent;
                        e[1] = make_uint4(key0, key1, key2, key3);
                    } else w.qe[c][(int64_t)blockIdx.x * w.q_cap[c] + i] = ent;
                } else (c == JTK_NBINS ? w.mid_list : c == JTK_NBINS + 1 ? w.long_list : w.giant_list)[i] = JtkLongPiece{pos, (uint64_t)h | ((uint64_t)plen << 40)};
            }
        }
        ho_head += nl;
        wave_lds_fence();
        STAMP_END(3);
    };

    for (int64_t strip = (int64_t)wv * gridDim.x + blockIdx.x; strip < w.n_tiles; strip += (int64_t)gridDim.x * ENC_WAVES) {
        const int64_t B = strip * T;
        const int64_t wd = (B >> 6) + lane;
        // this lane's mask word: piece starts (without the end sentinel)
        uint64_t pm = piece_word(w, wd);
        if (wd * 64 + 63 >= n) pm &= (wd * 64 >= n) ? 0ull : ((1ull << (n - wd * 64)) - 1ull);
        const uint64_t gm = (gaps && wd < w.n_words) ? w.gapmask[wd] : 0ull;    // custom patterns: text no match covers
        // (touch the strip's text -- one word of every 64-byte block -- so that it is on its way while the masks are scanned)
#ifdef JTK_EXP_NOTOUCH
        const uint32_t touch = 0;
#else
        const uint32_t touch = (B + lane * 64 < n) ? *reinterpret_cast<const uint32_t*>(w.text + B + lane * 64) : 0u;
#endif
        const uint32_t cnt = (uint32_t)__popcll(pm);
        const uint32_t inc = wave_incl_scan_dpp(cnt);
        const int np = (int)(uint32_t)__shfl((int)inc, 63);
        const uint32_t pre = inc - cnt;
        const uint32_t sb = cursor;                                      // this strip's slots: region0 + sb ..
        cursor += ((uint32_t)np + 3u) & ~3u;
        if (lane == 0) { w.tile_np[strip] = (uint32_t)np; w.sbase[strip] = sb; }
        if (np == 0) continue;                                           // a strip inside one long piece
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
        uint32_t run = 0;                                                // dense tokens of the strip so far
        const bool tail_strip = B + T + 16 > n;                          // (the last strips of the text: careful loads)
        int k0 = -ENC_WIN;

        for (int kc = 0; kc < np; kc += 64 * ENC_G) {
            STAMP_BEGIN();
            if (kc >= k0 + ENC_WIN) {
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
            // ENC_G chunks per step: their text loads are in flight together, then their table probes (what bounds this path is
            // the latency of those two dependent loads, not their number)
            bool have[ENC_G], small[ENC_G];
            uint32_t s[ENC_G], plen[ENC_G], tx0[ENC_G], tx1[ENC_G], key0[ENC_G], key1[ENC_G], len[ENC_G];
            uint4 ka[ENC_G];
#pragma unroll
            for (int q = 0; q < ENC_G; q++) {
                const int k = kc + 64 * q + lane;
                have[q] = k < np;
                const int rel = have[q] ? k - k0 : 0;
                const uint32_t e = W.starts[rel + 1];
                s[q] = W.starts[rel];
                plen[q] = e - s[q];
            }
            if (!tail_strip) {
#pragma unroll
                for (int q = 0; q < ENC_G; q++) {
                    const U2Bytes v = *reinterpret_cast<const U2Bytes*>(w.text + B + s[q]);
                    tx0[q] = v.x; tx1[q] = v.y;
                }
            } else {
#pragma unroll
                for (int q = 0; q < ENC_G; q++) {
                    const uint4 v = load_text16(w.text, B + s[q], n);
                    tx0[q] = v.x; tx1[q] = v.y;
                }
            }
            // a piece of <= 8 bytes and its primary slot in the tok8 table
#pragma unroll
            for (int q = 0; q < ENC_G; q++) {
                small[q] = have[q] && plen[q] <= 8u;
                len[q] = small[q] ? plen[q] : 0u;
                const uint64_t runm = ~0ull >> ((0u - 8u * len[q]) & 63u);       // 8 len ones
                key0[q] = tx0[q] & (uint32_t)runm; key1[q] = tx1[q] & (uint32_t)(runm >> 32);
                const uint32_t mix = jtk_tok16_mix(key0[q], key1[q], 0u, 0u, len[q]);
                ka[q] = *reinterpret_cast<const uint4*>(tok + ((size_t)jtk_reduce32(mix, n8) << 4));   // lo, hi, id, len
            }
#pragma unroll
            for (int q = 0; q < ENC_G; q++) {
                bool hit = small[q] && ((ka[q].x ^ key0[q]) | (ka[q].y ^ key1[q]) | ((ka[q].w & JTK_TOK_LEN_MASK) ^ len[q])) == 0u;
                if (gaps) {                                                      // (wave-uniform; the shuffles are evaluated by all lanes)
                    const uint32_t glo = (uint32_t)__shfl((int)(uint32_t)gm, (int)(s[q] >> 6)), ghi = (uint32_t)__shfl((int)(uint32_t)(gm >> 32), (int)(s[q] >> 6));
                    if ((((s[q] & 32u) ? ghi : glo) >> (s[q] & 31u)) & 1u) hit = false;   // unmatched text is a hole without tokens
                }
                // one slot per piece, in order: the token of a dense piece, SLOT_HOLE for the others
                if (have[q]) w.stok[region0 + sb + (uint32_t)(kc + 64 * q + lane)] = hit ? ka[q].z : SLOT_HOLE;
                run += (uint32_t)__popcll(__ballot(hit));
                // holes: a bit in the strip's bitmap and an entry in the pending ring
                const uint64_t bo = __ballot(have[q] && !hit);
                if (bo) {
                    if (have[q] && !hit) {
                        const uint32_t ri = (ho_tail + mbcnt64(bo)) & (ENC_PEND - 1);
                        W.ho_sl[ri] = s[q] | ((min(plen[q], 17u) - 1u) << 12);
                        W.ho_strip[ri] = (uint32_t)strip;
                        W.ho_idx[ri] = sb + (uint32_t)(kc + 64 * q + lane);
                    }
                    ho_tail += (uint32_t)__popcll(bo);
                }
            }
            STAMP_END(1);
            STAMP_ADD(4, ENC_G);
            // 64 holes wait: a batch
            while (ho_tail - ho_head >= 64u) hole_batch(64u);
        }
        asm volatile("" ::"v"(touch));
        // the strip's dense pieces' tokens
        if (lane == 0 && run) atomicAdd(&w.tile_tot[strip], run);
        STAMP_ADD(8, 1);
        STAMP_ADD(9, np);
    }
    // ---- the wave's last holes and hard pieces
    while (ho_tail != ho_head) hole_batch(min(64u, ho_tail - ho_head));
    // how many entries this workgroup's queues hold
    __syncthreads();
    if (tid < JTK_NBINS) w.q_count[tid * w.n_shards + blockIdx.x] = s_qcount[tid];
#ifdef JTK_ENC_STAMP
    st_acc[0] = __builtin_amdgcn_s_memtime() - st_wave0;
    if (lane == 0) for (int i = 0; i < 16; i++) atomicAdd(&g_enc_stamp[i], (unsigned long long)st_acc[i]);
#endif
}

// ---------------------------------------------------------------------------------------------------
// strip_expand: the packed token stream.  One wave per strip, once the exclusive scan of the strips' token counts
// (k_tile_scan) has said where its tokens go: pieces in order, 64 at a time -- a dense piece takes the next token of the
// strip's dense block, a hole its record (hrec); a wave prefix sum of the counts gives every token its place.  Also leaves,
// at every document's first byte, the tokens of its strip before it (docpre) for k_doc_offsets.
// ---------------------------------------------------------------------------------------------------
constexpr int EXPAND_THREADS = 256;
constexpr int EXPAND_STAGE = 1024;             // tokens of a step (256 pieces) assembled in LDS; a step with more stores them directly

#ifdef JTK_ENC_STAMP
__device__ unsigned long long g_exp_stamp[16];
#endif
__global__ void __launch_bounds__(EXPAND_THREADS) k_strip_expand(JtkWork w) {
    __shared__ uint32_t s_stage[EXPAND_THREADS / 64][EXPAND_STAGE];
    uint32_t* const stage = s_stage[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
#ifdef JTK_ENC_STAMP
    uint64_t st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t st_wave0 = __builtin_amdgcn_s_memtime();
#endif
    const int64_t strip = (int64_t)blockIdx.x * (EXPAND_THREADS / 64) + (threadIdx.x >> 6);
    if (strip >= w.n_tiles) return;
    const uint32_t total = w.tile_tot[strip];
    const int np = (int)w.tile_np[strip];
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
    if (np == 0) return;
    const int64_t B = strip * T;
    const bool store = w.count_only == 0;
    // document starts in this strip (each is a piece start): lane L keeps those of its mask word, with the pieces before the word
    const int64_t wd = (B >> 6) + lane;
    const uint64_t dm = (wd < w.n_words) ? w.docmask[wd] : 0ull;
    uint64_t pm = 0;
    uint32_t ppre = 0;
    const bool any_doc = __ballot(dm != 0) != 0;
    if (any_doc) {
        pm = piece_word(w, wd);
        if (wd * 64 + 63 >= w.n_bytes) pm &= (wd * 64 >= w.n_bytes) ? 0ull : ((1ull << (w.n_bytes - wd * 64)) - 1ull);
        const uint32_t c = (uint32_t)__popcll(pm);
        ppre = wave_incl_scan_dpp(c) - c;
    }
    const uint64_t dmp = dm & pm;                                        // (a document start that is no piece start here: the end sentinel)
    const uint32_t pcnt = (uint32_t)__popcll(pm);
    uint32_t* const dst = reinterpret_cast<uint32_t*>(w.tokens) + base;
    uint32_t run = 0;                                                    // tokens placed
#ifdef JTK_ENC_STAMP
    st_acc[1] = __builtin_amdgcn_s_memtime() - st_wave0;
#endif
    // FOUR consecutive pieces per lane, 256 per step: one 16-byte load of their slots per lane (the kernel is a stream: what
    // bounds it is the bytes in flight) and one wave scan of the lanes' token counts -- a slot says how many tokens its piece
    // has, so nothing else has to be read before the places are known; only a hole of several tokens reads its record, at
    // the slot's own index.  Everything is requested ahead: the slots two steps, the records (whose lanes the slots tell)
    // one step -- all record loads of a step back to back, without branches (a lane without such a hole reads the strip's
    // first record): with a branch around each load the compiler waited for every one of them in turn.
    const int64_t idx0 = strip_region(w, strip) + w.sbase[strip];        // the strip's slots and hole records: idx0 + piece number
    auto slots_of = [&](int k0) { return k0 + 4 * lane < np ? *reinterpret_cast<const uint4*>(w.stok + idx0 + k0 + 4 * lane) : make_uint4(0, 0, 0, 0); };
    struct Step { uint32_t sl[4], cn[4], tot, nh; bool valid[4], ishole[4]; uint64_t rec[4]; };
    auto prepare = [&](const uint4& v, int k0, Step& st) {
        // the step's pieces from their slots, and its record loads
        const int kb = k0 + 4 * lane;
        st.sl[0] = v.x; st.sl[1] = v.y; st.sl[2] = v.z; st.sl[3] = v.w;
        st.tot = 0; st.nh = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            st.valid[i] = kb + i < np;
            st.ishole[i] = st.valid[i] && (st.sl[i] & SLOT_MULTI) != 0u;
            st.cn[i] = st.valid[i] ? (st.ishole[i] ? (st.sl[i] == SLOT_HOLE ? 0u : st.sl[i] & ~SLOT_MULTI) : 1u) : 0u;
            st.tot += st.cn[i];
            st.nh += st.ishole[i] ? 1u : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) st.rec[i] = w.hrec[(st.ishole[i] && st.cn[i]) ? idx0 + kb + i : idx0];
    };
    uint4 v_nn = slots_of(256);
    Step nx;
    prepare(slots_of(0), 0, nx);
    for (int k0 = 0; k0 < np; k0 += 256) {
        STAMP_BEGIN();
        STAMP_ADD(4, 1);
        const Step st = nx;
        const uint4 v_n = v_nn;
        v_nn = slots_of(k0 + 512);
        if (k0 + 256 < np) prepare(v_n, k0 + 256, nx);
        const uint32_t (&sl)[4] = st.sl;
        const uint32_t (&cn)[4] = st.cn;
        const bool (&valid)[4] = st.valid;
        const bool (&ishole)[4] = st.ishole;
        const uint64_t (&rec)[4] = st.rec;
        const uint32_t tot = st.tot, nh = st.nh;
        const int kb = k0 + 4 * lane;
        (void)kb;
        uint32_t pos0;
        if (!__ballot(nh != 0)) {
            // every piece one token
            pos0 = run + 4u * (uint32_t)lane;
            run += (uint32_t)min(256, np - k0);
        } else {
            const uint32_t inc = wave_incl_scan_dpp(tot);
            pos0 = run + inc - tot;
            run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (store) {
            // The step's tokens are assembled in LDS and leave in full 256-byte stores: written from where they are computed
            // they would be a dozen sparse store instructions per step, and the CU's memory pipeline -- every wave's loads
            // queue behind them -- was what bounded the kernel (13 M write requests for 31 M tokens; 4 x slower).
            const uint32_t step_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos0);      // (lane 0: the step's first token)
            const uint32_t step_tot = run - step_base;
            const bool staged = step_tot <= (uint32_t)EXPAND_STAGE;                               // (wave-uniform)
            // place p of the strip: in the LDS stage (relative to the step's first token) or straight in the output
            auto emit = [&](auto put) {
                if (nh == 0) {
                    if (valid[0]) put(pos0, sl[0]);
                    if (valid[1]) put(pos0 + 1, sl[1]);
                    if (valid[2]) put(pos0 + 2, sl[2]);
                    if (valid[3]) put(pos0 + 3, sl[3]);
                } else {
                    uint32_t p = pos0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (!ishole[i]) { if (valid[i]) put(p, sl[i]); }
                        else if (cn[i]) {
                            const uint32_t kind = (uint32_t)(rec[i] >> HR_KIND_SHIFT) & 3u;
                            if (kind == HR_TOKS) {
                                put(p, (uint32_t)rec[i] & JTK_HT_ID_MASK);
                                if (cn[i] > 1u) put(p + 1, (uint32_t)(rec[i] >> 17) & JTK_HT_ID_MASK);
                                if (cn[i] > 2u) put(p + 2, (uint32_t)(rec[i] >> 34) & JTK_HT_ID_MASK);
                            } else if (kind == HR_REF) {
                                const uint32_t* src = w.htok + B + ((uint32_t)(rec[i] >> 21) & 0xFFFu);
                                for (uint32_t j = 0; j < cn[i]; j++) put(p + j, src[j] & JTK_HT_ID_MASK);
                            }
                        }
                        p += cn[i];
                    }
                }
            };
            if (staged) emit([&](uint32_t p, uint32_t v) { stage[p - step_base] = v; });
            else emit([&](uint32_t p, uint32_t v) { dst[p] = v; });
            if (staged) {
                wave_lds_fence();
                for (uint32_t i = (uint32_t)lane; i < step_tot; i += 64u) dst[step_base + i] = stage[i];
                wave_lds_fence();
            }
        }
#ifdef JTK_ENC_STAMP
        st_acc[3] += __builtin_amdgcn_s_memtime() - st_t0;
#endif
        // document starts among these pieces: tokens of the strip before them (rare: one or two per strip)
        if (any_doc) {
            // lanes whose mask word holds a document start at one of this step's pieces
            for (uint64_t todo = __ballot(dmp != 0 && ppre < (uint32_t)(k0 + 256) && ppre + pcnt > (uint32_t)k0); todo; todo &= todo - 1) {
                const int L = jtk_ctz64(todo);
                const uint64_t dL = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(dmp >> 32), L) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)dmp, L);
                const uint64_t pL = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pm >> 32), L) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pm, L);
                const uint32_t preL = (uint32_t)__builtin_amdgcn_readlane((int)ppre, L);
                for (uint64_t d = dL; d; d &= d - 1) {
                    const int bit = jtk_ctz64(d);
                    const int kk = (int)(preL + (uint32_t)__popcll(pL & ((1ull << bit) - 1ull)));
                    if (kk >= k0 && kk < k0 + 256) {
                        const int ln = (kk - k0) >> 2, sub = (kk - k0) & 3;
                        uint32_t pv = (uint32_t)__builtin_amdgcn_readlane((int)pos0, ln);
                        if (sub > 0) pv += (uint32_t)__builtin_amdgcn_readlane((int)cn[0], ln);
                        if (sub > 1) pv += (uint32_t)__builtin_amdgcn_readlane((int)cn[1], ln);
                        if (sub > 2) pv += (uint32_t)__builtin_amdgcn_readlane((int)cn[2], ln);
                        if (lane == 0) w.docpre[B + L * 64 + bit] = pv;
                    }
                }
            }
        }
    }
#ifdef JTK_ENC_STAMP
    st_acc[0] = __builtin_amdgcn_s_memtime() - st_wave0;
    st_acc[5] = 1;
    if (lane == 0) for (int i = 0; i < 16; i++) atomicAdd(&g_exp_stamp[i], (unsigned long long)st_acc[i]);
#endif
}
