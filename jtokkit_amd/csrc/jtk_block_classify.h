// jtk_block_classify.h -- class masks of one 64-byte block computed by ONE lane (block-parallel form).
//
// jtk_split_masks.h consumes 64-bit masks per block.  Producing them with one lane per byte and
// __ballot costs a wave-instruction per mask per block; here each lane classifies its own block:
//   1. every byte is looked up in a 256-entry table of 16-bit flag codes (ASCII bytes are fully
//      classified by it; bytes >= 0x80 only get their UTF-8 role),
//   2. the 64 codes are turned into 16 masks by 8x8 bit-matrix transposes (no per-bit loops),
//   3. characters outside ASCII are decoded and classified one by one (lead bytes only) and their
//      class is copied to all their bytes.
// Host/device shared so that it can be checked against the per-byte construction on the CPU.
#ifndef JTK_BLOCK_CLASSIFY_H
#define JTK_BLOCK_CLASSIFY_H

#include "jtk_common.h"
#include "jtk_split_masks.h"

// flag code of one byte (low byte: classes and roles, high byte: contraction letters)
enum : uint32_t {
    JTK_F_L = 1u << 0, JTK_F_N = 1u << 1, JTK_F_W = 1u << 2, JTK_F_NL = 1u << 3, JTK_F_SP = 1u << 4, JTK_F_AP = 1u << 5,
    JTK_F_CONT = 1u << 6, JTK_F_LEAD = 1u << 7,          // LEAD: first byte of a non-ASCII character
    JTK_F_S1 = 1u << 8, JTK_F_RV = 1u << 9, JTK_F_E = 1u << 10, JTK_F_LL = 1u << 11, JTK_F_C5 = 1u << 12, JTK_F_BF = 1u << 13,
    JTK_F_LT = 1u << 14                                   // a byte a special-token literal starts with (set by the kernel from the encoding's literals)
};

JTK_HD uint32_t jtk_byte_code(uint32_t b, bool case_insensitive) {
    uint32_t c = 0;
    if (b < 0x80u) {
        const uint32_t cls = jtk_class_of_ascii(b);
        if (cls == JTK_CLS_L) c |= JTK_F_L;
        if (cls == JTK_CLS_N) c |= JTK_F_N;
        if (cls == JTK_CLS_W) c |= JTK_F_W;
        if (b == '\r' || b == '\n') c |= JTK_F_NL;
        if (b == 0x20u) c |= JTK_F_SP;
        if (b == '\'') c |= JTK_F_AP;
        const uint32_t f = (case_insensitive && (b - 'A') < 26u) ? (b | 0x20u) : b;
        if (f == 's' || f == 't' || f == 'm' || f == 'd') c |= JTK_F_S1;
        if (f == 'r' || f == 'v') c |= JTK_F_RV;
        if (f == 'e') c |= JTK_F_E;
        if (f == 'l') c |= JTK_F_LL;
    } else {
        if ((b & 0xC0u) == 0x80u) c |= JTK_F_CONT; else c |= JTK_F_LEAD;
        if (b == 0xC5u) c |= JTK_F_C5;
        if (b == 0xBFu) c |= JTK_F_BF;
    }
    return c;
}

// 8x8 bit-matrix transpose: bit (8*r + c) of the result = bit (8*c + r) of x
JTK_HD uint64_t jtk_transpose8x8(uint64_t x) {
    uint64_t t;
    t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;  x ^= t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x ^= t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x ^= t ^ (t << 28);
    return x;
}

// d[0..15]: the block's 64 bytes, little-endian dwords.  codes: 256 x uint16 flag codes.
// Fills the table-derived part of the masks (class bits of non-ASCII characters are still 0).
// lead_out: mask of non-ASCII lead bytes.
template <class CodeTab>
JTK_HD void jtk_block_masks_ascii(const uint32_t (&d)[16], const CodeTab& codes, JtkBlk& k, uint64_t& lead_out, uint64_t& lt_out) {
    uint64_t lo_m[8], hi_m[8];
#pragma unroll
    for (int f = 0; f < 8; f++) { lo_m[f] = 0; hi_m[f] = 0; }
#pragma unroll
    for (int g = 0; g < 8; g++) {
        uint64_t xl = 0, xh = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t b = (d[2 * g + (q >> 2)] >> (8 * (q & 3))) & 255u;
            const uint32_t c = codes[b];
            xl |= (uint64_t)(c & 255u) << (8 * q);
            xh |= (uint64_t)(c >> 8) << (8 * q);
        }
        const uint64_t yl = jtk_transpose8x8(xl), yh = jtk_transpose8x8(xh);
#pragma unroll
        for (int f = 0; f < 8; f++) {
            lo_m[f] |= ((yl >> (8 * f)) & 255ull) << (8 * g);
            if (f < 7) hi_m[f] |= ((yh >> (8 * f)) & 255ull) << (8 * g);
        }
    }
    k.L = lo_m[0]; k.N = lo_m[1]; k.W = lo_m[2]; k.NL = lo_m[3]; k.SP = lo_m[4]; k.AP = lo_m[5]; k.CONT = lo_m[6];
    lead_out = lo_m[7];
    k.S1 = hi_m[0]; k.RV = hi_m[1]; k.E = hi_m[2]; k.LL = hi_m[3]; k.C5 = hi_m[4]; k.BF = hi_m[5];
    lt_out = hi_m[6];
}
template <class CodeTab>
JTK_HD void jtk_block_masks_ascii(const uint32_t (&d)[16], const CodeTab& codes, JtkBlk& k, uint64_t& lead_out) {
    uint64_t lt;
    jtk_block_masks_ascii(d, codes, k, lead_out, lt);
}

// Non-ASCII characters that START in this block: decode (Txt gives byte(p) for any p), classify, and
// give the class to all bytes of the character that lie inside the block.  Returns in spill_cls the
// class of a character that runs over the block's end (or JTK_CLS_O) for the next block.
template <class Txt>
JTK_HD void jtk_block_fix_nonascii(const Txt& txt, const JtkUcTables& uc, int64_t p0, uint64_t lead, JtkBlk& k,
                                   uint32_t& spill_cls) {
    spill_cls = JTK_CLS_O;
    for (uint64_t m = lead; m;) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int j = __ffsll((unsigned long long)m) - 1;
#else
        const int j = __builtin_ctzll(m);
#endif
        m &= m - 1;
        const int64_t p = p0 + j;
        const uint32_t b0 = txt.byte(p);
        uint32_t cp, n;
        if (b0 < 0xE0u) { cp = ((b0 & 0x1Fu) << 6) | (txt.byte(p + 1) & 0x3Fu); n = 2; }
        else if (b0 < 0xF0u) { cp = ((b0 & 0x0Fu) << 12) | ((txt.byte(p + 1) & 0x3Fu) << 6) | (txt.byte(p + 2) & 0x3Fu); n = 3; }
        else { cp = ((b0 & 0x07u) << 18) | ((txt.byte(p + 1) & 0x3Fu) << 12) | ((txt.byte(p + 2) & 0x3Fu) << 6) | (txt.byte(p + 3) & 0x3Fu); n = 4; }
        // CJK ideographs (incl. extension A) and Hangul syllables are letters throughout (Unicode 13: U+3400-4DBF,
        // U+4E00-9FFC, U+AC00-D7A3 are all Lo); everything else goes through the two-stage table
        uint32_t cls;
        if ((cp - 0x3400u) <= (0x4DBFu - 0x3400u) || (cp - 0x4E00u) <= (0x9FFCu - 0x4E00u) || (cp - 0xAC00u) <= (0xD7A3u - 0xAC00u)) cls = JTK_CLS_L;
        else cls = jtk_class_of_cp(uc, cp);
        // the character's bytes: the lead and the continuation bytes that directly follow it (at most n-1)
        uint64_t bytes = 1ull << j;
        for (uint32_t q = 1; q < n; q++) {
            const int jj = j + (int)q;
            if (jj < 64) { if ((k.CONT >> jj) & 1ull) bytes |= 1ull << jj; else break; }
        }
        if (cls == JTK_CLS_L) k.L |= bytes;
        else if (cls == JTK_CLS_N) k.N |= bytes;
        else if (cls == JTK_CLS_W) k.W |= bytes;
        if (j + (int)n > 64) spill_cls = cls;
    }
}

// Continuation bytes at the start of a block belong to a character that started in the previous block.
JTK_HD void jtk_block_apply_spill(JtkBlk& k, uint32_t prev_spill_cls) {
    const uint64_t c = k.CONT;
    uint64_t head = c & 1ull;                        // leading run of continuation bytes, at most 3
    head |= (head << 1) & c;
    head |= (head << 1) & c;
    if (prev_spill_cls == JTK_CLS_L) k.L |= head;
    else if (prev_spill_cls == JTK_CLS_N) k.N |= head;
    else if (prev_spill_cls == JTK_CLS_W) k.W |= head;
}

#endif
