// jtk_merge_core.h -- bytePairMerge (GptBytePairEncoding.java:200-275) on token ids, one lane per piece.
//
// parts of the reference  <->  the byte positions of the piece whose ids[] entry is not DEAD;
// parts[i].rank           <->  rk[position of part i] = rank of (part i + next part), NONE if absent.
// The reference recomputes parts[i].rank and parts[i-1].rank with skip=1 BEFORE removing part i+1
// (:248-259); on ids that is pair(merged, id of the part after next) and pair(id of the previous
// part, merged), and the merged part's id is the rank that was just selected.
// The leftmost minimum wins (strict `<` in the scan, :236).
#ifndef JTK_MERGE_CORE_H
#define JTK_MERGE_CORE_H

#include "jtk_common.h"


// ids[0..len): on entry the single-byte token id of every byte; on exit the token id at every
// surviving part start and JTK_ID_DEAD elsewhere.  rk[0..len) is scratch.  1 <= len <= 64.
// Returns the number of tokens.  (Plain form: used by the host table check.)
template <class P>
JTK_HD int jtk_merge_piece_lane(P ids, P rk, int len, const JtkPairTable& pt) {
    uint64_t alive = (len >= 64) ? ~0ull : ((1ull << len) - 1ull);
    for (int j = 0; j + 1 < len; j++) rk[j] = jtk_pair_lookup(pt, ids[j], ids[j + 1]);    // :216-221
    rk[len - 1] = JTK_RANK_NONE;
    int ntok = len;
    while (ntok > 1) {                                                                       // :223
        uint32_t minr = JTK_RANK_NONE;
        int mini = 0;
        for (uint64_t m = alive; m;) {                                                       // :234-240
            const int j = jtk_ctz64(m);
            m &= m - 1;
            const uint32_t r = rk[j];
            if (r < minr) { minr = r; mini = j; }
        }
        if (minr == JTK_RANK_NONE) break;                                                    // :247,:261
        const uint64_t above = alive & ~((2ull << mini) - 1ull);      // parts after mini (mini < 63 here)
        const int nxt = jtk_ctz64(above);                              // exists: rk[mini] != NONE
        const uint64_t above2 = above & (above - 1);
        const uint32_t r1 = above2 ? jtk_pair_lookup(pt, minr, ids[jtk_ctz64(above2)]) : JTK_RANK_NONE;   // :254
        const uint64_t below = alive & ((1ull << mini) - 1ull);
        if (below) {                                                                         // :255-257
            const int pv = 63 - jtk_clz64(below);
            rk[pv] = jtk_pair_lookup(pt, ids[pv], minr);
        }
        ids[mini] = minr;
        rk[mini] = r1;
        ids[nxt] = JTK_ID_DEAD;                                                              // :259
        alive &= ~(1ull << nxt);
        ntok--;
    }
    return ntok;
}

// Same algorithm with the memory-level parallelism the device wants: the initial pair ranks of a piece
// are ranks of 2-byte tokens, read from the direct table bp_rank[b0 << 8 | b1] (no probing, all loads
// independent), and the two lookups after each merge are issued together.
// txt[0..len): the piece's bytes; ids/rk are filled here.
template <class P, class B>
JTK_HD int jtk_merge_piece_lane2(P ids, P rk, B txt, int len, const JtkPairTable& pt, const uint32_t* bp_rank,
                                 const uint32_t* byte_rank) {
    uint64_t alive = (len >= 64) ? ~0ull : ((1ull << len) - 1ull);
    uint32_t prev = txt[0];
    for (int j = 0; j + 1 < len; j++) {                                                      // :206-221
        const uint32_t cur = txt[j + 1];
        ids[j] = byte_rank[prev];
        rk[j] = bp_rank[(prev << 8) | cur];
        prev = cur;
    }
    ids[len - 1] = byte_rank[prev];
    rk[len - 1] = JTK_RANK_NONE;
    int ntok = len;
    while (ntok > 1) {                                                                       // :223
        uint32_t minr = JTK_RANK_NONE;
        int mini = 0;
        for (uint64_t m = alive; m;) {                                                       // :234-240
            const int j = jtk_ctz64(m);
            m &= m - 1;
            const uint32_t r = rk[j];
            if (r < minr) { minr = r; mini = j; }
        }
        if (minr == JTK_RANK_NONE) break;                                                    // :247,:261
        const uint64_t above = alive & ~((2ull << mini) - 1ull);
        const int nxt = jtk_ctz64(above);
        const uint64_t above2 = above & (above - 1);
        const uint64_t below = alive & ((1ull << mini) - 1ull);
        const int nn = above2 ? jtk_ctz64(above2) : 0;
        const int pv = below ? 63 - jtk_clz64(below) : 0;
        uint32_t r1, r2;
        jtk_pair_lookup2(pt, minr, above2 ? (uint32_t)ids[nn] : 0u, above2 != 0, below ? (uint32_t)ids[pv] : 0u, minr,
                         below != 0, r1, r2);                                                // :254-257
        if (below) rk[pv] = r2;
        ids[mini] = minr;
        rk[mini] = r1;
        ids[nxt] = JTK_ID_DEAD;                                                              // :259
        alive &= ~(1ull << nxt);
        ntok--;
    }
    return ntok;
}

#endif
