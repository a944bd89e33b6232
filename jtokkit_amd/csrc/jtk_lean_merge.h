// jtk_lean_merge.h -- bytePairMerge (GptBytePairEncoding.java:200-275) ONE LANE PER PIECE, all lanes of a wave stepping
// together; included by jtk_kernels.hip (inside its anonymous namespace).  Two users:
//   * k_strip_encode (jtk_strip_encode.h): the pieces of <= 16 bytes that are no table entries, merged by the wave that
//     encodes their strip, 64 at a time;
//   * k_bpe_merge (jtk_long_pieces.h): the queued pieces of 17..64 bytes.
// A wave expands its pieces (byte -> id and the 2-byte-token ranks from LDS tables), then every step is: leftmost minimum
// over the pair keys (:234-240), the two neighbour lookups in the (left id, right id) pair table, update (:248-259); a lane
// whose piece is finished idles until the wave's last piece is.  The slots scanned per step are bounded by the wave's
// longest piece (NS: a compile-time unrolled scan, all LDS reads of a step in flight together).
// What bounds a step is the number of scattered cache-line fetches (tools/microbench/gather_rate.hip: a CU sustains one
// per ~2.3 clocks), so it fetches as few as it can: the pair table is primary-first (jtk_common.h) -- ONE 16-byte load per
// lookup, issued for both lookups together; only lanes that miss in a bucket flagged "overflowed" read their secondary
// bucket -- and a lookup whose two parts make up the whole piece is not made at all when the piece is known to be no table
// entry (SKIP_WHOLE).
// Parts live in LDS laid out [slot][lane] (conflict-free for any per-lane slot); key = rank << 6 | slot orders by rank
// first and leftmost among equal ranks (:236).
constexpr uint32_t KL_NONE = 0xFFFFFFFFu;
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
        if (__ballot(more1 || more2)) {
            // the secondary buckets, for the lanes that need them (the others re-read bucket lines they just had)
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
