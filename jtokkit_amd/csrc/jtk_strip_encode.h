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
constexpr int ENC_PEND = 128;                  // ring of pending holes / hard pieces (at most 127 wait)
static_assert(T == 4096, "a strip is 64 mask words: one per lane");

struct __attribute__((aligned(16))) EncWaveLds {
    uint2 holes[ENC_PEND];                     // pending holes:  x = offset (12) | (min(len, 17) - 1) << 12 | hole number << 17, y = strip
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
        const uint2 pe = W.holes[(ho_head + (uint32_t)lane) & (ENC_PEND - 1)];
        const uint32_t strip = pe.y, s = pe.x & 0xFFFu, h = pe.x >> 17;
        const uint32_t lenc = mine ? ((pe.x >> 12) & 31u) + 1u : 0u;             // 17: more than 16 bytes
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
        if (mine && !queued) w.hrec[(int64_t)strip * T + h] = rec;
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
                    w.qe[c][(int64_t)blockIdx.x * w.q_cap[c] + i] = make_uint4((uint32_t)meta, (uint32_t)(meta >> 32), h, 0u);
                } else (c == JTK_NBINS ? w.mid_list : c == JTK_NBINS + 1 ? w.long_list : w.giant_list)[i] = JtkLongPiece{pos, (int32_t)plen, h};
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
        const uint32_t touch = (B + lane * 64 < n) ? *reinterpret_cast<const uint32_t*>(w.text + B + lane * 64) : 0u;
        const uint32_t cnt = (uint32_t)__popcll(pm);
        const uint32_t inc = wave_incl_scan_dpp(cnt);
        const int np = (int)(uint32_t)__shfl((int)inc, 63);
        const uint32_t pre = inc - cnt;
        if (lane == 0) w.tile_np[strip] = (uint32_t)np;
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
        uint32_t* const out = w.stok + B;
        uint32_t run = 0;                                                // dense tokens of the strip so far
        uint32_t nholes = 0;
        uint64_t hw = 0;                                                 // lane c: the hole mask of chunk c
        const bool tail_strip = B + T + 16 > n;                          // (the last strips of the text: careful loads)
        int k0 = -ENC_WIN;

        for (int kc = 0; kc < np; kc += 64) {
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
            const int k = kc + lane;
            const bool have = k < np;
            const int rel = have ? k - k0 : 0;
            const uint32_t s = W.starts[rel], e = W.starts[rel + 1];
            const uint32_t plen = e - s;
            // a piece of <= 8 bytes and its primary slot in the tok8 table
            uint32_t tx0, tx1;
            if (!tail_strip) {
                const U2Bytes v = *reinterpret_cast<const U2Bytes*>(w.text + B + s);
                tx0 = v.x; tx1 = v.y;
            } else {
                const uint4 v = load_text16(w.text, B + s, n);
                tx0 = v.x; tx1 = v.y;
            }
            const bool small = have && plen <= 8u;
            const uint32_t len = small ? plen : 0u;
            const uint64_t runm = ~0ull >> ((0u - 8u * len) & 63u);              // 8 len ones
            const uint32_t key0 = tx0 & (uint32_t)runm, key1 = tx1 & (uint32_t)(runm >> 32);
            const uint32_t mix = jtk_tok16_mix(key0, key1, 0u, 0u, len);
            const uint4 ka = *reinterpret_cast<const uint4*>(tok + ((size_t)jtk_reduce32(mix, n8) << 4));   // lo, hi, id, len
            bool hit = small && ((ka.x ^ key0) | (ka.y ^ key1) | ((ka.w & JTK_TOK_LEN_MASK) ^ len)) == 0u;
            if (gaps) {                                                          // (wave-uniform; the shuffles are evaluated by all lanes)
                const uint32_t glo = (uint32_t)__shfl((int)(uint32_t)gm, (int)(s >> 6)), ghi = (uint32_t)__shfl((int)(uint32_t)(gm >> 32), (int)(s >> 6));
                if ((((s & 32u) ? ghi : glo) >> (s & 31u)) & 1u) hit = false;    // unmatched text is a hole without tokens
            }
            // dense pieces: the token, in order
            const uint64_t bh = __ballot(hit);
            if (hit && !w.count_only) out[run + mbcnt64(bh)] = ka.z;
            run += (uint32_t)__popcll(bh);
            // holes: a bit in the strip's bitmap and an entry in the pending ring
            const uint64_t bo = __ballot(have && !hit);
            if (bo) {
                if (have && !hit) {
                    const uint32_t hno = nholes + mbcnt64(bo);
                    W.holes[(ho_tail + mbcnt64(bo)) & (ENC_PEND - 1)] = make_uint2(s | ((min(plen, 17u) - 1u) << 12) | (hno << 17), (uint32_t)strip);
                }
                ho_tail += (uint32_t)__popcll(bo);
                nholes += (uint32_t)__popcll(bo);
            }
            if (lane == (kc >> 6)) hw = bo;
            STAMP_END(1);
            STAMP_ADD(4, 1);
            // 64 holes wait: a batch
            if (ho_tail - ho_head >= 64u) hole_batch(64u);
        }
        asm volatile("" ::"v"(touch));
        // the strip's hole bitmap (one word per chunk) and its dense tokens' count
        if (lane < ((np + 63) >> 6)) w.holebits[strip * 64 + lane] = hw;
        if (lane == 0 && run) atomicAdd(&w.tile_tot[strip], run);
        STAMP_ADD(8, 1);
        STAMP_ADD(9, np);
        STAMP_ADD(12, nholes);
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

__global__ void __launch_bounds__(EXPAND_THREADS) k_strip_expand(JtkWork w) {
    const int lane = threadIdx.x & 63;
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
    const int nchunks = (np + 63) >> 6;
    const bool store = w.count_only == 0;
    // lane c: the hole mask of chunk c and the holes before it
    const uint64_t hw = lane < nchunks ? w.holebits[strip * 64 + lane] : 0ull;
    const uint32_t hc = (uint32_t)__popcll(hw);
    const uint32_t hpre = wave_incl_scan_dpp(hc) - hc;
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
    const uint32_t* const dense = w.stok + B;
    const uint64_t* const hrec = w.hrec + B;
    uint32_t* const dst = reinterpret_cast<uint32_t*>(w.tokens) + base;
    uint32_t run = 0;
    for (int c = 0; c < nchunks; c++) {
        const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)hw, c), whi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(hw >> 32), c);
        const uint64_t word = ((uint64_t)whi << 32) | wlo;
        const uint32_t hb = (uint32_t)__builtin_amdgcn_readlane((int)hpre, c);
        const int k = c * 64 + lane;
        const bool valid = k < np;
        const bool ishole = ((word >> lane) & 1ull) != 0;                 // (bits of a chunk's mask beyond np are clear)
        const uint32_t dbase = (uint32_t)(c * 64) - hb;                     // dense pieces before this chunk
        uint32_t pos;
        if (word == 0) {
            // every piece a dense one: one token each
            pos = run + (uint32_t)lane;
            if (valid && store) dst[pos] = dense[dbase + (uint32_t)lane];
            run += (uint32_t)min(64, np - c * 64);
        } else {
            const uint64_t vmask = (np - c * 64 >= 64) ? ~0ull : ((1ull << (np - c * 64)) - 1ull);
            uint32_t tk = 0;
            uint64_t rec = 0;
            if (valid && !ishole) tk = dense[dbase + mbcnt64(~word & vmask)];
            if (ishole) rec = hrec[hb + mbcnt64(word)];
            const uint32_t kind = (uint32_t)(rec >> HR_KIND_SHIFT) & 3u;
            uint32_t cn = valid ? 1u : 0u;
            if (ishole) cn = kind == HR_TOKS ? (uint32_t)((rec >> 51) & 3u) + 1u : kind == HR_REF ? (uint32_t)(rec & 0x1FFFFFu) : 0u;
            const uint32_t inc = wave_incl_scan_dpp(cn);
            pos = run + inc - cn;
            run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            if (store) {
                if (ishole && kind == HR_TOKS) tk = (uint32_t)rec & JTK_HT_ID_MASK;
                const bool isref = ishole && kind == HR_REF;
                if (cn && !isref) dst[pos] = tk;
                if (ishole && kind == HR_TOKS && cn > 1u) {
                    dst[pos + 1] = (uint32_t)(rec >> 17) & JTK_HT_ID_MASK;
                    if (cn > 2u) dst[pos + 2] = (uint32_t)(rec >> 34) & JTK_HT_ID_MASK;
                }
                if (__ballot(isref)) {
                    if (isref) {
                        const uint32_t* src = w.htok + B + ((uint32_t)(rec >> 21) & 0xFFFu);
                        for (uint32_t i = 0; i < cn; i++) dst[pos + i] = src[i] & JTK_HT_ID_MASK;
                    }
                }
            }
        }
        // document starts among these pieces: tokens of the strip before them (rare: one or two per strip)
        if (any_doc) {
            // lanes whose mask word holds a document start at one of this chunk's pieces
            for (uint64_t todo = __ballot(dmp != 0 && ppre < (uint32_t)(c * 64 + 64) && ppre + pcnt > (uint32_t)(c * 64)); todo; todo &= todo - 1) {
                const int L = jtk_ctz64(todo);
                const uint64_t dL = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(dmp >> 32), L) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)dmp, L);
                const uint64_t pL = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pm >> 32), L) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pm, L);
                const uint32_t preL = (uint32_t)__builtin_amdgcn_readlane((int)ppre, L);
                for (uint64_t d = dL; d; d &= d - 1) {
                    const int bit = jtk_ctz64(d);
                    const int kk = (int)(preL + (uint32_t)__popcll(pL & ((1ull << bit) - 1ull)));
                    if (kk >= c * 64 && kk < c * 64 + 64) {
                        const uint32_t pv = (uint32_t)__builtin_amdgcn_readlane((int)pos, kk - c * 64);
                        if (lane == 0) w.docpre[B + L * 64 + bit] = pv;
                    }
                }
            }
        }
    }
}
