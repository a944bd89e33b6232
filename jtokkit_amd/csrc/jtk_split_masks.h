// jtk_split_masks.h -- the split rules of jtk_split_rules.h evaluated for 64 bytes at a time on bit masks.
//
// One wave classifies a 64-byte block (one lane per byte) and turns the per-lane predicates into
// 64-bit masks with __ballot; everything below is then scalar bit algebra on those masks (bit j =
// byte j of the block), so the cost per block is independent of what the bytes are.  "prev"/"next"
// byte relations are shifts with the neighbouring blocks' edge bits carried in.  The rules are the
// ones derived in jtk_split_rules.h (same reference lines); the few positions whose answer needs an
// unbounded walk (a non-CR/LF whitespace char right after a CR/LF that was not swallowed by an
// O-piece; runs whose start lies before the wave's first block) are returned in `slow` and are
// evaluated with jtk_is_piece_start_t instead.
#ifndef JTK_SPLIT_MASKS_H
#define JTK_SPLIT_MASKS_H

#include "jtk_common.h"

struct JtkBlk {                 // class masks of one 64-byte block
    uint64_t L, N, W;           // class of the character each byte belongs to (O = none of them)
    uint64_t CONT, NL, SP, DS, AP;
    uint64_t S1, RV, E, LL, C5, BF;   // contraction letters: s|t|m|d, r|v, e, l, 0xC5, 0xBF (case folded for cl100k)
};

struct JtkSplitCarry {          // state handed from block b to block b+1
    uint64_t pL, pN, pW, pNL, pSP, pDS, pCONT;  // the previous block's masks (only the top 3 bits are used)
    uint64_t pS1, pRV, pE, pLL, pC5, pBF;
    uint64_t pMsAP;             // apostrophes of the previous block that start a match
    uint64_t pX;                // bytes of O characters that start a match (previous block)
    uint64_t pSW;               // CR/LF bytes swallowed by an O piece (previous block)
    uint32_t ncnt;              // N characters in the digit run that ends at the previous block's last byte, mod 3
    bool n_unknown;             // ... that run started before the wave's first block
    bool sw_unknown;            // the CR/LF chain ending the previous block started before the wave's first block
};

JTK_HD int jtk_popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll((unsigned long long)x);
#else
    return __builtin_popcountll(x);
#endif
}
JTK_HD int jtk_hibit64(uint64_t x) {   // index of the highest set bit, x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return 63 - __clzll((long long)x);
#else
    return 63 - __builtin_clzll(x);
#endif
}

// bits of `seed` extended upwards through consecutive set bits of `through`
JTK_HD uint64_t jtk_fill_up(uint64_t seed, uint64_t through) {
    uint64_t x = seed, m = through;
    x |= (x << 1) & m;  m &= m << 1;
    x |= (x << 2) & m;  m &= m << 2;
    x |= (x << 4) & m;  m &= m << 4;
    x |= (x << 8) & m;  m &= m << 8;
    x |= (x << 16) & m; m &= m << 16;
    x |= (x << 32) & m;
    return x;
}

// bits of `seed` extended downwards: bit q is set when bit q+1 is and `through` has bit q
JTK_HD uint64_t jtk_fill_down(uint64_t seed, uint64_t through) {
    uint64_t x = seed, m = through;
    x |= (x >> 1) & m;  m &= m >> 1;
    x |= (x >> 2) & m;  m &= m >> 2;
    x |= (x >> 4) & m;  m &= m >> 4;
    x |= (x >> 8) & m;  m &= m >> 8;
    x |= (x >> 16) & m; m &= m >> 16;
    x |= (x >> 32) & m;
    return x;
}

JTK_HD uint32_t jtk_fill_down32(uint32_t seed, uint32_t through) {
    uint32_t x = seed, m = through;
    x |= (x >> 1) & m;  m &= m >> 1;
    x |= (x >> 2) & m;  m &= m >> 2;
    x |= (x >> 4) & m;  m &= m >> 4;
    x |= (x >> 8) & m;  m &= m >> 8;
    x |= (x >> 16) & m;
    return x;
}

JTK_HD void jtk_split_carry_init(JtkSplitCarry& c) {
    c.pL = c.pN = c.pW = c.pNL = c.pSP = c.pDS = c.pCONT = 0;
    c.pS1 = c.pRV = c.pE = c.pLL = c.pC5 = c.pBF = 0;
    c.pMsAP = c.pX = c.pSW = 0;
    c.ncnt = 0;
    c.n_unknown = false;
    c.sw_unknown = false;
}

// cl100k: is the N character whose lead byte is bit j of the block (not a document start) a piece
// start?  "\p{N}{1,3}": every third character of the digit run.  `slow` is set when the run started
// before the wave's first block.
JTK_HD bool jtk_split_n_lane(const JtkBlk& cu, uint32_t ncnt, bool n_unknown, int j, bool& slow) {
    const uint64_t below = (1ull << j) - 1ull;
    const uint64_t lead = ~cu.CONT;
    const uint64_t Z = (~cu.N | cu.DS) & below;          // run breakers below j
    uint32_t cnt;
    if (Z == 0) {
        if (n_unknown) { slow = true; return false; }
        cnt = (uint32_t)jtk_popc64(lead & below) + ncnt;
    } else {
        const int hb = jtk_hibit64(Z);
        const int start = ((cu.N >> hb) & (cu.DS >> hb) & 1ull) ? hb : hb + 1;
        cnt = (uint32_t)jtk_popc64(cu.N & lead & below & ~((1ull << start) - 1ull));
    }
    return cnt % 3u == 0u;
}

// cl100k, all digits of the block at once when each is one byte ("\p{N}{1,3}": every third byte from the run's start; a run
// that comes in from the previous block is `ncnt` digits (mod 3) into its phase).  pN_top: bit 63 = the byte before the
// block is a digit.  Returns the piece starts among the digits; `slow` gets the digits of a run whose start is not visible.
// Blocks with digits of several bytes (cu.N & cu.CONT) go through jtk_split_n_lane per digit instead.
JTK_HD uint64_t jtk_split_n_block(const JtkBlk& cu, uint64_t pN_top, uint32_t ncnt, bool n_unknown, uint64_t& slow) {
    const uint64_t notDS = ~cu.DS;
    const uint64_t pN = (cu.N << 1) | (pN_top >> 63);
    uint64_t S = cu.N & (~pN | cu.DS);
    slow = 0;
    if (cu.N & notDS & pN & 1ull) {                                     // the first byte goes on with a run
        if (n_unknown) slow = jtk_fill_up(1ull, cu.N & notDS);
        else {
            const uint32_t r = ncnt == 0u ? 0u : 3u - ncnt;             // digits to the next piece start
            const uint64_t req = (2ull << r) - 1ull;
            if ((cu.N & req) == req && (cu.DS & req) == 0) S |= 1ull << r;
        }
    }
    const uint64_t nn = cu.N & notDS;
    uint64_t link = cu.N & (nn << 1) & (nn << 2) & (cu.N << 3) & notDS;   // q-3 .. q are digits of one run (q-3 may start a document)
    uint64_t x = S;
    x |= (x << 3) & link;   link &= link << 3;
    x |= (x << 6) & link;   link &= link << 6;
    x |= (x << 12) & link;  link &= link << 12;
    x |= (x << 24) & link;  link &= link << 24;
    x |= (x << 48) & link;
    return x & notDS & ~slow;
}

// Piece-start mask of block `cu` for the positions that are decided by mask algebra; `slow` gets the
// positions the caller must evaluate with jtk_is_piece_start_t, `nlanes` the digits it must pass through jtk_split_n_lane
// (cl100k, only blocks with digits of several bytes; digit runs of ASCII are mask algebra).  `nx` supplies the next block (only
// its low 32 bits are used).  Updates `cy` for the next block.
template <int KIND>
JTK_HD uint64_t jtk_split_block(const JtkBlk& cu, const JtkBlk& nx, JtkSplitCarry& cy, uint64_t& slow, uint64_t& nlanes) {
    constexpr bool cl = (KIND == JTK_PAT_CL100K);
#define JTK_P1(cur, prv) (((cur) << 1) | ((prv) >> 63))
#define JTK_PK(cur, prv, k) (((cur) << (k)) | ((prv) >> (64 - (k))))
#define JTK_N1(cur, nxt) (((cur) >> 1) | ((nxt) << 63))
    const uint64_t lead = ~cu.CONT, notDS = ~cu.DS;
    const uint64_t O = ~(cu.L | cu.N | cu.W);
    const uint64_t pL = JTK_P1(cu.L, cy.pL), pN = JTK_P1(cu.N, cy.pN), pW = JTK_P1(cu.W, cy.pW);
    const uint64_t pO = ~(pL | pN | pW);
    const uint64_t pSP = JTK_P1(cu.SP, cy.pSP), pNL = JTK_P1(cu.NL, cy.pNL), pDS = JTK_P1(cu.DS, cy.pDS);

    // O characters that start a match (incl. document starts), and all bytes of those characters
    const uint64_t mso = O & lead & (cu.DS | (~pO & ~pSP));
    uint64_t X = mso | ((cy.pX >> 63) & cu.CONT & 1ull);
    X |= (X << 1) & cu.CONT;
    X |= (X << 1) & cu.CONT;
    X |= (X << 1) & cu.CONT;
    const uint64_t prevIsMsO = JTK_P1(X, cy.pX);
    const uint64_t msAP = cu.AP & mso;

    // ---- letters
    const uint64_t ctr2 = JTK_PK(msAP, cy.pMsAP, 2) & JTK_P1(cu.S1, cy.pS1) & ~pDS;
    uint64_t tail3 = (JTK_PK(cu.RV, cy.pRV, 2) & JTK_P1(cu.E, cy.pE)) | (JTK_PK(cu.LL, cy.pLL, 2) & JTK_P1(cu.LL, cy.pLL));
    if (cl) tail3 |= JTK_PK(cu.C5, cy.pC5, 2) & JTK_P1(cu.BF, cy.pBF);
    const uint64_t ctr3 = JTK_PK(msAP, cy.pMsAP, 3) & tail3 & ~pDS & ~JTK_PK(cu.DS, cy.pDS, 2);
    uint64_t afterO;
    if (cl) afterO = ~prevIsMsO;
    else {
        // r50k: no one-char prefix; only "the letter of a contraction" is not a start
        const uint64_t c_here = cu.S1 | (((cu.RV & JTK_N1(cu.E, nx.E)) | (cu.LL & JTK_N1(cu.LL, nx.LL))) & ~JTK_N1(cu.DS, nx.DS));
        afterO = ~(JTK_P1(msAP, cy.pMsAP) & c_here);
    }
    const uint64_t afterW = cl ? pNL : ~pSP;
    const uint64_t Lms = cu.L & lead & notDS & (pN | (pW & afterW) | (pO & afterO) | (pL & (ctr2 | ctr3)));

    // ---- numbers
    uint64_t Nms = 0;
    nlanes = 0;
    if (cl) nlanes = cu.N & lead & notDS;                      // jtk_split_n_block / jtk_split_n_lane, once the carry is final
    else Nms = cu.N & lead & notDS & ~pN & ~pSP;

    // ---- whitespace
    const uint64_t nW = JTK_N1(cu.W, nx.W), nDS = JTK_N1(cu.DS, nx.DS);
    uint64_t ylo = cu.W & (~nW | nDS) & ~nDS;                  // last byte of a run that is followed by text
    uint64_t yhi = nx.W & (~(nx.W >> 1) | (nx.DS >> 1)) & ~(nx.DS >> 1) & 7ull;
    for (int it = 0; it < 3; it++) {                           // move the flag to the character's lead byte
        const uint64_t lo_c = ylo & cu.CONT, hi_c = yhi & nx.CONT;
        ylo = (ylo & lead) | (lo_c >> 1) | ((hi_c & 1ull) << 63);
        yhi = (yhi & ~nx.CONT) | (hi_c >> 1);
    }
    const uint64_t wlead = cu.W & lead & notDS;
    uint64_t Wms;
    slow = 0;
    uint64_t SW = 0;
    if (!cl) {
        Wms = wlead & (~pW | ylo);
    } else {
        // CR/LF directly after an O character (and the CR/LFs chained to them) belong to the O piece
        uint64_t seed = cu.NL & pO & notDS;
        const uint64_t chain0 = jtk_fill_up(cu.NL & notDS & 1ull, cu.NL & notDS);   // CR/LF chain from byte 0
        if ((cy.pSW >> 63) & 1ull) seed |= cu.NL & notDS & 1ull;
        SW = jtk_fill_up(seed, cu.NL & notDS);
        const uint64_t pSW = JTK_P1(SW, cy.pSW);
        const uint64_t plain = wlead & ~cu.NL & ~pNL;          // neither a CR/LF nor right after one
        const uint64_t nl = wlead & cu.NL;
        const uint64_t afterNL = wlead & ~cu.NL & pNL;
        // right after a CR/LF that was not swallowed: a start <=> no CR/LF remains in the rest of the whitespace run
        // (jtk_split_rules.h: "\s*[\r\n]+" ends on the LAST CR/LF).  Byte q hears from byte q+1 when q is whitespace and q+1
        // does not start a document; a run that leaves the block through whitespace that is not a CR/LF cannot be decided here.
        const uint64_t cand = afterNL & ~pSW;
        uint64_t undecided = 0;
        Wms = (plain & (~pW | ylo)) | (nl & ~SW & ~pW) | (afterNL & pSW);
        if (cand) {
            const uint64_t thr = cu.W & ~nDS;
            const uint64_t top = thr & ((nx.W & 1ull) << 63);           // the run goes on in the next block,
            // whose first 32 bytes are visible: a CR/LF there that the run reaches, or the run's end
            const uint32_t nxW = (uint32_t)nx.W, nxNL = (uint32_t)nx.NL;
            const uint32_t nthr = nxW & ~(uint32_t)(nx.DS >> 1) & 0x7FFFFFFFu;
            const uint32_t nmore = jtk_fill_down32(nxNL, nthr);
            const uint32_t nund = jtk_fill_down32(nxW & ~nxNL & 0x80000000u, nthr) & ~nmore;
            const uint64_t more = jtk_fill_down(cu.NL | (top & ((uint64_t)(nmore & 1u) << 63)), thr);
            undecided = jtk_fill_down(top & ((uint64_t)(nund & 1u) << 63), thr) & ~more;
            Wms |= cand & ~more & ~undecided;
        }
        slow = cand & undecided;
        if (cy.sw_unknown) {
            // the chain's origin is not visible: its bytes, and the char right after it, go the slow way
            const uint64_t after0 = (chain0 << 1) & wlead & ~cu.NL;
            slow |= (chain0 & wlead) | after0;
        }
        // carry: does the chain at the end of this block still have an invisible origin?
        const bool whole = chain0 == ~0ull;
        cy.sw_unknown = cy.sw_unknown && whole;
    }

    const uint64_t ms = (cu.DS | mso | Lms | Nms | Wms) & ~slow;

    // ---- carry for the next block
    if (cl) {
        if (!((cu.N >> 63) & 1ull)) { cy.ncnt = 0; cy.n_unknown = false; }
        else {
            const uint64_t Z = (~cu.N | cu.DS);
            if (Z == 0) cy.ncnt = (cy.ncnt + (uint32_t)jtk_popc64(lead)) % 3u;   // n_unknown unchanged
            else {
                const int hb = jtk_hibit64(Z);
                const int start = ((cu.N >> hb) & (cu.DS >> hb) & 1ull) ? hb : hb + 1;
                cy.ncnt = (start > 63) ? 0u : (uint32_t)jtk_popc64(cu.N & lead & ~((1ull << start) - 1ull)) % 3u;
                cy.n_unknown = false;
            }
        }
    }
    cy.pL = cu.L; cy.pN = cu.N; cy.pW = cu.W; cy.pNL = cu.NL; cy.pSP = cu.SP; cy.pDS = cu.DS; cy.pCONT = cu.CONT;
    cy.pS1 = cu.S1; cy.pRV = cu.RV; cy.pE = cu.E; cy.pLL = cu.LL; cy.pC5 = cu.C5; cy.pBF = cu.BF;
    cy.pMsAP = msAP; cy.pX = X; cy.pSW = SW;
#undef JTK_P1
#undef JTK_PK
#undef JTK_N1
    return ms;
}

// Carry for a wave that starts at block `first`: derived from the block before it alone.  Runs or
// CR/LF chains that cover that whole block have an origin the wave cannot see (flags *_unknown).
template <int KIND>
JTK_HD void jtk_split_carry_from_halo(const JtkBlk& halo, const JtkBlk& first, JtkSplitCarry& cy) {
    jtk_split_carry_init(cy);
    uint64_t slow, nl;
    (void)jtk_split_block<KIND>(halo, first, cy, slow, nl);
    if (KIND == JTK_PAT_CL100K) {
        cy.n_unknown = ((~halo.N | halo.DS) == 0);
        const uint64_t nlm = halo.NL & ~halo.DS;
        cy.sw_unknown = (nlm == ~0ull);
    }
}

#endif
