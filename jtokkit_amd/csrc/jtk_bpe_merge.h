// jtk_bpe_merge.h -- k_bpe_merge: bytePairMerge (GptBytePairEncoding.java:200-275) of the pieces k_strip_encode queued, by
// length bin, in one persistent launch; k_long_shortcut for rank tables that need the whole-piece lookup beyond 16 bytes.
// Included by jtk_kernels.hip inside its anonymous namespace.

// ---------------------------------------------------------------------------------------------------
// short bins of k_bpe_merge: the queued pieces of 4..16 bytes that are no table entry -- all but a handful of the pieces
// that need merging -- ONE LANE PER PIECE (jtk_lean_merge.h).  The three bins (4..8, 9..12, 13..16 bytes) run the 8-, 12- and
// 16-slot variants, so that a wave's pieces need about the same number of steps.  The bytes come straight from the text
// (one unaligned 16-byte load); the result is the piece's hole record (more than three tokens: in htok), and -- when it
// is at most six tokens -- an entry in this XCD's memo, so that k_strip_encode answers the piece's next occurrence itself.
// ---------------------------------------------------------------------------------------------------
template <int NS, int THREADS, int BIN>
__device__ __forceinline__ void short_bin(const JtkWork& w, const JtkDeviceTables& t, const LeanLds& L, uint4* memo, uint32_t count, uint32_t kq, uint32_t K) {
    const int tid = threadIdx.x;
    const int shard = blockIdx.x % w.n_shards;
    uint32_t* const id = L.id + tid;
    uint32_t* const rk = L.rk + tid;
    const uint4* const qe = w.qe[BIN] + (int64_t)shard * w.q_cap[BIN] * 2;       // 32-byte entries: position, length, hole number | the bytes
    for (uint32_t base = kq * THREADS; base < count; base += K * THREADS) {
        const uint32_t qi = base + (uint32_t)tid;
        const bool have = qi < count;
        uint4 ent = make_uint4(0, 0, 0, 0), tx = make_uint4(0, 0, 0, 0);
        if (have) { ent = qe[2 * (size_t)qi]; tx = qe[2 * (size_t)qi + 1]; }
        const uint64_t meta = ((uint64_t)ent.y << 32) | ent.x;
        const int64_t pos = (int64_t)(meta & JTK_QE_POS_MASK);
        const int len = have ? (int)((meta >> JTK_QE_LEN_SHIFT) & 31u) + 1 : 0;
        const uint32_t d4[4] = {tx.x, tx.y, tx.z, tx.w};
        uint32_t b[NS + 1];
#pragma unroll
        for (int j = 0; j < NS; j++) b[j] = (d4[j >> 2] >> (8 * (j & 3))) & 255u;
        b[NS] = 0;
        const uint32_t alive = lean_piece16<NS, THREADS>(L, id, rk, b, len, t);
        // emit (:270-273)
        const uint32_t c = (uint32_t)__popc(alive);
        const int64_t strip = pos / T;
        const uint32_t s = (uint32_t)(pos % T);
        if (have) {
            uint64_t rec;
            if (c <= 3u) {
                uint32_t a = alive;
                uint32_t tk[3];
#pragma unroll
                for (int i = 0; i < 3; i++) { const uint32_t j = a ? (uint32_t)__builtin_ctz(a) : 0u; tk[i] = a ? id[j * THREADS] : 0u; a &= a - 1u; }
                rec = (uint64_t)tk[0] | ((uint64_t)tk[1] << 17) | ((uint64_t)tk[2] << 34) | ((uint64_t)(c - 1u) << 51) | (HR_TOKS << HR_KIND_SHIFT);
            } else {
                uint32_t* dst = w.htok + pos;
                uint32_t i = 0;
                for (uint32_t a = alive; a; a &= a - 1u) dst[i++] = id[(uint32_t)__builtin_ctz(a) * THREADS];
                rec = (uint64_t)c | ((uint64_t)s << 21) | (HR_REF << HR_KIND_SHIFT);
            }
            put_hole(w, (int64_t)(((uint64_t)ent.w << 32) | ent.z), c, rec);
        }
        if (memo) {
            // remember the result under the piece's bytes (only the lane that claims an empty slot writes it)
            const uint32_t ulen = (uint32_t)len;
            const uint32_t key0 = tx.x, key1 = tx.y, key2 = tx.z, key3 = tx.w;   // (zero beyond the piece's length: k_strip_encode masked them)
            if (have && c <= MEMO_MAX_TOKENS && key0 != 0u) {
                const uint32_t mix = jtk_tok16_mix(key0, key1, key2, key3, ulen);
                uint4* e = memo + (size_t)memo_slot(mix, w.memo_mask) * 2u;
                unsigned long long* hi64 = reinterpret_cast<unsigned long long*>(e + 1) + 1;
                // (a look first: scattered compare-and-swaps are slow, and most slots are taken after the first pieces)
                if (*hi64 == 0ull && atomicCAS(hi64, 0ull, (unsigned long long)MEMO_BUSY) == 0ull) {
                    uint32_t a = alive;
                    uint64_t tk[6];
#pragma unroll
                    for (int i = 0; i < 6; i++) { const uint32_t j = a ? (uint32_t)__builtin_ctz(a) : 0u; tk[i] = a ? id[j * THREADS] : 0u; a &= a - 1u; }
                    const uint32_t tag = memo_tag(mix);
                    e[0] = make_uint4(key0, key1, key2, key3);
                    reinterpret_cast<unsigned long long*>(e + 1)[0] = tk[0] | (tk[1] << 17) | (tk[2] << 34) | ((uint64_t)tag << 51);
                    *hi64 = tk[3] | (tk[4] << 17) | (tk[5] << 34) | ((uint64_t)c << 51) | ((uint64_t)ulen << 54) | ((uint64_t)(tag & 31u) << 59);
                }
            }
        }
        add_strip_counts(w, (uint32_t)strip, have, c);
    }
}

// ---------------------------------------------------------------------------------------------------
// lean bins of k_bpe_merge: queued pieces of 17..64 bytes, one lane per piece (jtk_lean_merge.h), bytes from the text.
// ---------------------------------------------------------------------------------------------------
template <int SLOTS, int THREADS, int BIN>
__device__ __forceinline__ void lean_bin(const JtkWork& w, const JtkDeviceTables& t, const LeanLds& L, uint32_t count, uint32_t kq, uint32_t K) {
    typedef typename std::conditional<(SLOTS > 32), uint64_t, uint32_t>::type M;
    const int tid = threadIdx.x;
    if (tid >= THREADS) return;
    const int shard = blockIdx.x % w.n_shards;
    uint32_t* const id = L.id + tid;
    uint32_t* const rk = L.rk + tid;
    const uint4* const qe = w.qe[BIN] + (int64_t)shard * w.q_cap[BIN];

    for (uint32_t base = kq * THREADS; base < count; base += K * THREADS) {
        const uint32_t qi = base + (uint32_t)tid;
        bool have = qi < count;
        uint4 ent = make_uint4(0, 0, 0, 0);
        if (have) ent = qe[qi];
        const uint64_t meta = ((uint64_t)ent.y << 32) | ent.x;
        if (meta & JTK_QE_DONE) have = false;                          // a table entry (k_long_shortcut): its record is in place
        const int64_t pos = (int64_t)(meta & JTK_QE_POS_MASK);
        const int len = have ? (int)((meta >> JTK_QE_LEN_SHIFT) & 255u) + 1 : 0;
        M alive;
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
        // ---- emit (:270-273): the piece's tokens in htok, packed from its first byte position; the hole record says how many
        const uint32_t c = sizeof(M) == 8 ? (uint32_t)__popcll((uint64_t)alive) : (uint32_t)__popc((uint32_t)alive);
        if (have) {
            uint32_t* dst = w.htok + pos;
            uint32_t idx = 0;
            for (M m = alive; m;) {
                const uint32_t j = sizeof(M) == 8 ? (uint32_t)jtk_ctz64((uint64_t)m) : (uint32_t)__builtin_ctz((uint32_t)m);
                m &= m - (M)1;
                dst[idx++] = id[j * THREADS];
            }
            put_hole(w, (int64_t)(((uint64_t)ent.w << 32) | ent.z), c, (uint64_t)c | ((uint64_t)(pos % T) << 21) | (HR_REF << HR_KIND_SHIFT));
        }
        add_strip_counts(w, (uint32_t)(pos / T), have, c);
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
    const int shard = blockIdx.x % w.n_shards;
    const uint32_t kq = blockIdx.x / w.n_shards, K = gridDim.x / w.n_shards;
    const uint32_t count = L.count[BIN];
    if ((uint64_t)kq * M_CHUNK >= count) return;
    const uint4* const queue = w.qe[BIN] + (int64_t)shard * w.q_cap[BIN];

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
    uint64_t hole = 0;
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
        if (st == ST_NEED) a0 = queue + qi;
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
            const uint64_t entry = ((uint64_t)v0.y << 32) | v0.x;
            hole = ((uint64_t)v0.w << 32) | v0.z;
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
        // (6) emit finished pieces (:270-273) last, so the stores drain under the next trip's work: the piece's tokens in htok,
        // packed from its first byte position; the hole record says how many
        const uint64_t b_emit = __ballot(st == ST_EMIT);
        if (b_emit && (__popcll(b_emit) >= EMIT_BATCH || !b_merge)) {
            if (st == ST_EMIT) {
                uint32_t c = 0;
#pragma unroll
                for (int k = 0; k < NW; k++) c += (uint32_t)__popcll(alive[k]);
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
                put_hole(w, (int64_t)hole, c, (uint64_t)c | ((uint64_t)(pos % T) << 21) | (HR_REF << HR_KIND_SHIFT));
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
        const int len = (int)(lp.idx_len >> 40);
        if (len <= 0) continue;                                    // found by k_long_shortcut
        for (int j = lane; j < len; j += WAVE) {
            const uint32_t b0 = w.text[lp.start + j];
            s_id[j] = t.byte_rank[b0];
            s_rk[j] = (j + 1 < len) ? t.bp_rank[(b0 << 8) | w.text[lp.start + j + 1]] : JTK_RANK_NONE;
        }
        wave_lds_fence();
        merge_piece_wave(s_id, s_rk, len, t.pairs);
        // surviving ids, packed from the piece's first position
        uint32_t total = 0;
        for (int base = 0; base < len; base += WAVE) {
            const int j = base + lane;
            const bool alive = j < len && s_id[j] != JTK_ID_DEAD;
            const uint64_t bal = __ballot(alive);
            const uint32_t idx = total + (uint32_t)__popcll(bal & lanemask_lt());
            if (alive) w.htok[lp.start + idx] = s_id[j];
            total += (uint32_t)__popcll(bal);
        }
        if (lane == 0) {
            put_hole(w, (int64_t)(lp.idx_len & ((1ull << 40) - 1ull)), total, (uint64_t)total | ((uint64_t)(lp.start % T) << 21) | (HR_REF << HR_KIND_SHIFT));
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
    const int len = (int)(lp.idx_len >> 40);
    if (len <= 0) return;                                         // found by k_long_shortcut (workgroup-uniform)
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
    // emit (wave 0): surviving ids packed in place from the piece's first position (a survivor never moves up)
    if (wv == 0) {
        uint32_t total = 0;
        for (int base = 0; base < len; base += WAVE) {
            const int j = base + lane;
            const uint32_t v = j < len ? gid[j] : JTK_ID_DEAD;
            const bool alive = v != JTK_ID_DEAD;
            const uint64_t bal = __ballot(alive);
            const uint32_t idx = total + (uint32_t)__popcll(bal & lanemask_lt());
            if (alive && idx) gid[idx] = v;
            total += (uint32_t)__popcll(bal);
        }
        if (lane == 0) {
            put_hole(w, (int64_t)(lp.idx_len & ((1ull << 40) - 1ull)), total, (uint64_t)total | ((uint64_t)(lp.start % T) << 21) | (HR_REF << HR_KIND_SHIFT));
            atomicAdd(&w.tile_tot[lp.start / T], total);
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------
// k_bpe_merge: ONE persistent launch for bytePairMerge of every piece that k_strip_encode queued: the short bins (4..16
// bytes, one lane per piece), the lean bins of 17..32 and 33..64 bytes, the state-machine bins for pieces of up to 256
// bytes, the wave-per-piece lists (<= 512, <= 8192 bytes) and the workgroup-per-piece giants.  All phases share the 128 KiB
// of LDS parts and the staged tables; a workgroup barrier separates phases whose LDS layouts differ, but there is no
// device-wide barrier and no launch gap between them, and on ordinary text the later phases find empty queues and cost
// nothing.  Every piece's result is its hole record (and tokens in htok when they are more than three); its token count
// is added to its strip's.
// ---------------------------------------------------------------------------------------------------
#ifndef JTK_ML_THREADS
#define JTK_ML_THREADS 1024
#endif
constexpr int ML_THREADS = JTK_ML_THREADS;       // lanes (= pieces in flight) per workgroup; 16 part slots x 8 bytes of LDS each
constexpr int ML_WORDS = 16 * ML_THREADS;
constexpr int ML_WGS_PER_SHARD = JTK_M_WGS_PER_SHARD;

__global__ void __launch_bounds__(ML_THREADS) k_bpe_merge(JtkWork w, JtkDeviceTables t) {
    __shared__ uint32_t s_id[ML_WORDS];
    __shared__ uint32_t s_rk[ML_WORDS];
    __shared__ uint64_t s_bpbits[1024];
    __shared__ uint32_t s_bpranks[JTK_BP_MAX];
    __shared__ uint16_t s_bpcum[1024];
    __shared__ uint32_t s_brank[256];
    __shared__ uint32_t s_next[JTK_NBINS];
    __shared__ uint32_t s_count[JTK_NBINS + 3];
    const int tid = threadIdx.x;
    const int shard = blockIdx.x % w.n_shards;
    const uint32_t kq = blockIdx.x / w.n_shards, K = gridDim.x / w.n_shards;
    if (tid < JTK_NBINS) {
        s_next[tid] = 0;
        s_count[tid] = w.q_count[tid * w.n_shards + shard];
    }
    if (tid == JTK_NBINS) s_count[JTK_NBINS] = *w.mid_count;
    if (tid == JTK_NBINS + 1) s_count[JTK_NBINS + 1] = *w.long_count;
    if (tid == JTK_NBINS + 2) s_count[JTK_NBINS + 2] = *w.n_giant;
    __syncthreads();
    // (all the counts were read up front: a phase without work costs neither a global load nor a barrier)
    const uint32_t n0 = s_count[0], n1 = s_count[1], n2 = s_count[2], n3 = s_count[3], n4 = s_count[4];
    const bool rest = (s_count[5] | s_count[6] | s_count[JTK_NBINS] | s_count[JTK_NBINS + 1] | s_count[JTK_NBINS + 2]) != 0u;
    // Each phase is a chain of dependent lookups (as many as its longest piece has merges).  When every bin of the shard
    // fits one workgroup pass -- small batches, where those chains ARE the kernel's time -- the shard's workgroups take
    // bins of their own, so the chains run side by side; otherwise every workgroup takes a slice of every bin.
    const bool side_by_side = n0 <= (uint32_t)ML_THREADS && n1 <= (uint32_t)ML_THREADS && n2 <= (uint32_t)ML_THREADS &&
                              n3 <= (uint32_t)(ML_THREADS / 2) && n4 <= (uint32_t)(ML_THREADS / 4) && K >= 4u;
    const uint32_t Kx = side_by_side ? 1u : K, k = side_by_side ? 0u : kq;
    bool w0, w1, w2, w3, w4;
    if (side_by_side) {
        w0 = kq == 0u && n0; w1 = kq == 1u && n1; w2 = kq == 2u && n2; w3 = kq == 3u && n3; w4 = kq == 3u && n4;
    } else {
        w0 = kq * (uint32_t)ML_THREADS < n0; w1 = kq * (uint32_t)ML_THREADS < n1; w2 = kq * (uint32_t)ML_THREADS < n2;
        w3 = kq * (uint32_t)(ML_THREADS / 2) < n3; w4 = kq * (uint32_t)(ML_THREADS / 4) < n4;
    }
    if (!(w0 || w1 || w2 || w3 || w4 || rest)) return;
    for (int i = tid; i < 1024; i += ML_THREADS) { s_bpbits[i] = t.bp.bits[i]; s_bpcum[i] = t.bp.cum[i]; }
    for (int i = tid; i < JTK_BP_MAX; i += ML_THREADS) s_bpranks[i] = t.bp.ranks[i];
    if (tid < 256) s_brank[tid] = t.byte_rank[tid];
    __syncthreads();
    const LeanLds LL{s_id, s_rk, JtkBpLds{s_bpbits, s_bpcum, s_bpranks}, s_brank};
    uint4* const memo = w.memo ? w.memo + (size_t)xcc_id() * ((size_t)w.memo_mask + 1u) * 2u : nullptr;   // this XCD's table
    // (the three short bins share one LDS layout, [16 slots][1024 lanes], and a lane uses only its own column: no barrier
    // between them)
    if (w0) short_bin<8, ML_THREADS, 0>(w, t, LL, memo, n0, k, Kx);
    if (w1) short_bin<12, ML_THREADS, 1>(w, t, LL, memo, n1, k, Kx);
    if (w2) short_bin<16, ML_THREADS, 2>(w, t, LL, memo, n2, k, Kx);
    if (w3) { __syncthreads(); lean_bin<32, ML_THREADS / 2, 3>(w, t, LL, n3, k, Kx); }
    if (w4) { __syncthreads(); lean_bin<64, ML_THREADS / 4, 4>(w, t, LL, n4, k, Kx); }
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
    // giant pieces: one workgroup per piece
    if (s_count[JTK_NBINS + 2]) {
        __syncthreads();
        const GiantLds G{reinterpret_cast<uint64_t*>(s_id), reinterpret_cast<uint64_t*>(s_rk), reinterpret_cast<int*>(s_rk + 64),
                         s_rk + 72};
        for (uint32_t gi = blockIdx.x; gi < s_count[JTK_NBINS + 2]; gi += gridDim.x) merge_giant(w, t, G, gi);
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
    for (int bin = JTK_NBINS_SHORT; bin < JTK_NBINS; bin++) {
        for (int shard = 0; shard < (int)w.n_shards; shard++) {
            const uint32_t count = w.q_count[bin * w.n_shards + shard];
            uint4* qe = w.qe[bin] + (int64_t)shard * w.q_cap[bin];
            for (uint32_t i = gtid; i < count; i += gn) {
                const uint4 ent = qe[i];
                const uint64_t meta = ((uint64_t)ent.y << 32) | ent.x;
                const int64_t pos = (int64_t)(meta & JTK_QE_POS_MASK);
                const uint32_t id = long_lookup(w, t, pos, (int64_t)((meta >> JTK_QE_LEN_SHIFT) & 255u) + 1);
                if (id != JTK_RANK_NONE) {
                    put_hole(w, (int64_t)(((uint64_t)ent.w << 32) | ent.z), 1u, (uint64_t)id | (HR_TOKS << HR_KIND_SHIFT));   // one token
                    atomicAdd(&w.tile_tot[pos / T], 1u);
                    qe[i].y = ent.y | 0x80000000u;                    // JTK_QE_DONE
                }
            }
        }
    }
    for (int which = 0; which < 3; which++) {
        JtkLongPiece* list = which == 0 ? w.mid_list : which == 1 ? w.long_list : w.giant_list;
        const uint32_t count = which == 0 ? *w.mid_count : which == 1 ? *w.long_count : *w.n_giant;
        for (uint32_t i = gtid; i < count; i += gn) {
            const JtkLongPiece lp = list[i];
            const uint32_t id = long_lookup(w, t, lp.start, (int64_t)(lp.idx_len >> 40));
            if (id != JTK_RANK_NONE) {
                put_hole(w, (int64_t)(lp.idx_len & ((1ull << 40) - 1ull)), 1u, (uint64_t)id | (HR_TOKS << HR_KIND_SHIFT));   // one token
                atomicAdd(&w.tile_tot[lp.start / T], 1u);
                list[i].idx_len = lp.idx_len & ((1ull << 40) - 1ull);
            }
        }
    }
}

