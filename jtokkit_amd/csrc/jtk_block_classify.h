// jtk_block_classify.h -- class masks of one 64-byte block computed by ONE lane (block-parallel form).
//
// jtk_split_masks.h consumes 64-bit masks per block.  Producing them with one lane per byte and
// __ballot costs a wave-instruction per mask per block; here each lane classifies its own block:
//   1. every byte is looked up in a 256-entry table of 16 flags (ASCII bytes are fully classified by
//      it; bytes >= 0x80 get their UTF-8 role and, for lead bytes, "every character is a letter"),
//   2. the table entries are laid out so that a shift-or per text byte accumulates the 16 masks
//      (no per-bit loops, no bit transposes); a 4x4 byte transpose per 32 bytes puts them in place,
//   3. the remaining characters outside ASCII are decoded and classified one by one (lead bytes
//      only) and their class is copied to all their bytes.
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
    JTK_F_LT = 1u << 14,                                  // a byte a special-token literal starts with (set by the kernel from the encoding's literals)
    JTK_F_ULL = 1u << 15                                  // lead byte of characters that are all letters (jtk_lead_all_letters; set by the kernel)
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

// The table the kernel reads: one 16-byte entry per byte value, flag f of the 16-bit code as bit 0 of byte (f & 3) of word
// (f >> 2).  Shifting an accumulator left by one and OR-ing the entry in, for the 8 bytes of a group from the last to the
// first, leaves in byte q of accumulator word w the 8-bit mask of flag 4w + q over the group: no per-bit work, no bit
// transposes; one 16-byte LDS read and four shift-or instructions per text byte.
struct JtkCode4 { uint32_t x, y, z, w; };
JTK_HD JtkCode4 jtk_code4(uint32_t code) {
    JtkCode4 e;
    e.x = (code & 1u) | ((code >> 1) & 1u) << 8 | ((code >> 2) & 1u) << 16 | ((code >> 3) & 1u) << 24;
    e.y = ((code >> 4) & 1u) | ((code >> 5) & 1u) << 8 | ((code >> 6) & 1u) << 16 | ((code >> 7) & 1u) << 24;
    e.z = ((code >> 8) & 1u) | ((code >> 9) & 1u) << 8 | ((code >> 10) & 1u) << 16 | ((code >> 11) & 1u) << 24;
    e.w = ((code >> 12) & 1u) | ((code >> 13) & 1u) << 8 | ((code >> 14) & 1u) << 16 | ((code >> 15) & 1u) << 24;
    return e;
}

// (a << 1) | e as ONE instruction the optimiser may not re-associate (it would otherwise gather eight table entries into
// or-trees per word and spill the other twelve bytes of each entry)
JTK_HD uint32_t jtk_shl1_or(uint32_t a, uint32_t e) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(r) : "v"(a), "v"(e));
    return r;
#else
    return (a << 1) | e;
#endif
}

// bytes of {hi, lo} picked by the four selector bytes of sel (0..3: lo, 4..7: hi) -- v_perm_b32
JTK_HD uint32_t jtk_byteperm(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int q = 0; q < 4; q++) r |= (uint32_t)((v >> (8 * ((sel >> (8 * q)) & 7u))) & 255u) << (8 * q);
    return r;
#endif
}

// 4x4 byte transpose: g[i] byte q  ->  o[q] byte i
JTK_HD void jtk_transpose4x4_bytes(uint32_t g0, uint32_t g1, uint32_t g2, uint32_t g3, uint32_t (&o)[4]) {
    const uint32_t t0 = jtk_byteperm(g1, g0, 0x05010400u), t1 = jtk_byteperm(g3, g2, 0x05010400u);
    const uint32_t t2 = jtk_byteperm(g1, g0, 0x07030602u), t3 = jtk_byteperm(g3, g2, 0x07030602u);
    o[0] = jtk_byteperm(t1, t0, 0x05040100u); o[1] = jtk_byteperm(t1, t0, 0x07060302u);
    o[2] = jtk_byteperm(t3, t2, 0x05040100u); o[3] = jtk_byteperm(t3, t2, 0x07060302u);
}

// d[0..15]: the block's 64 bytes, little-endian dwords.  tab: 256 entries (jtk_code4 of the byte's flag code).
// Fills the table-derived part of the masks (class bits of non-ASCII characters are still 0).
// lead_out: non-ASCII lead bytes; lt_out: bytes a special literal starts with; ull_out: lead bytes of characters that are
// letters whatever their other bytes are.
template <class Tab4>
JTK_HD void jtk_block_masks_ascii(const uint32_t (&d)[16], const Tab4& tab, JtkBlk& k, uint64_t& lead_out, uint64_t& lt_out,
                                  uint64_t& ull_out) {
    uint32_t a[8][4];
#pragma unroll
    for (int g = 0; g < 8; g++) {
        uint32_t a0, a1, a2, a3;
        {
            const JtkCode4 e = tab[d[2 * g + 1] >> 24];
            a0 = e.x; a1 = e.y; a2 = e.z; a3 = e.w;
        }
#pragma unroll
        for (int q = 6; q >= 0; q--) {
            const uint32_t b = (d[2 * g + (q >> 2)] >> (8 * (q & 3))) & 255u;
            const JtkCode4 e = tab[b];
            a0 = jtk_shl1_or(a0, e.x); a1 = jtk_shl1_or(a1, e.y); a2 = jtk_shl1_or(a2, e.z); a3 = jtk_shl1_or(a3, e.w);
        }
        a[g][0] = a0; a[g][1] = a1; a[g][2] = a2; a[g][3] = a3;
    }
    uint64_t m[16];
#pragma unroll
    for (int wd = 0; wd < 4; wd++) {
        uint32_t lo[4], hi[4];
        jtk_transpose4x4_bytes(a[0][wd], a[1][wd], a[2][wd], a[3][wd], lo);
        jtk_transpose4x4_bytes(a[4][wd], a[5][wd], a[6][wd], a[7][wd], hi);
#pragma unroll
        for (int q = 0; q < 4; q++) m[4 * wd + q] = ((uint64_t)hi[q] << 32) | lo[q];
    }
    k.L = m[0]; k.N = m[1]; k.W = m[2]; k.NL = m[3]; k.SP = m[4]; k.AP = m[5]; k.CONT = m[6];
    lead_out = m[7];
    k.S1 = m[8]; k.RV = m[9]; k.E = m[10]; k.LL = m[11]; k.C5 = m[12]; k.BF = m[13];
    lt_out = m[14];
    ull_out = m[15];
}

// Is every character whose UTF-8 form starts with this lead byte a letter?  (CJK ideographs U+5000-8FFF, Hangul
// U+B000-CFFF, basic Cyrillic, ...: computed from the class table, host side, once per encoding.)
inline bool jtk_lead_all_letters(const JtkUcTables& uc, uint32_t b) {
    uint32_t lo, hi;
    if (b >= 0xC2u && b <= 0xDFu) { lo = (b & 0x1Fu) << 6; hi = lo + 63u; }
    else if (b == 0xE0u) { lo = 0x800u; hi = 0xFFFu; }
    else if (b == 0xEDu) { lo = 0xD000u; hi = 0xD7FFu; }              // surrogates have no UTF-8 form
    else if (b >= 0xE1u && b <= 0xEFu) { lo = (b & 0x0Fu) << 12; hi = lo + 0xFFFu; }
    else if (b == 0xF0u) { lo = 0x10000u; hi = 0x3FFFFu; }
    else if (b >= 0xF1u && b <= 0xF3u) { lo = (b & 7u) << 18; hi = lo + 0x3FFFFu; }
    else if (b == 0xF4u) { lo = 0x100000u; hi = 0x10FFFFu; }
    else return false;
    for (uint32_t cp = lo; cp <= hi; cp++) if (jtk_class_of_cp(uc, cp) != JTK_CLS_L) return false;
    return true;
}

// x and the continuation bytes that directly follow each of its bits (at most 3)
JTK_HD uint64_t jtk_with_cont_bytes(uint64_t x, uint64_t cont) {
    const uint64_t c1 = (x << 1) & cont, c2 = (c1 << 1) & cont, c3 = (c2 << 1) & cont;
    return x | c1 | c2 | c3;
}

// Non-ASCII characters that START in this block.  Leads in `ull` are letters without a look at the other bytes; the
// others are decoded (txt.word(j): the 4 bytes that start at byte j of the block, zero beyond the text) and classified
// one by one.  A character's class goes to its lead and the continuation bytes that directly follow it inside the
// block.  spill_cls: the class of a character that runs up to the block's last byte (for the leading continuation
// bytes of the next block; if the character ended there the next block has none), else JTK_CLS_O.
template <class Txt4>
JTK_HD void jtk_block_fix_nonascii(const Txt4& txt, const JtkUcTables& uc, uint64_t lead, uint64_t ull, JtkBlk& k,
                                   uint32_t& spill_cls) {
    uint64_t c0 = ull & lead, c1 = 0;                                  // class bits 0 / 1 at the lead positions
    for (uint64_t m = lead & ~ull; m;) {
        const int j = jtk_ctz64(m);
        m &= m - 1;
        const uint32_t w = txt.word(j);
        const uint32_t b0 = w & 255u, b1 = (w >> 8) & 0x3Fu, b2 = (w >> 16) & 0x3Fu, b3 = (w >> 24) & 0x3Fu;
        uint32_t cp;
        if (b0 < 0xE0u) cp = ((b0 & 0x1Fu) << 6) | b1;
        else if (b0 < 0xF0u) cp = ((b0 & 0x0Fu) << 12) | (b1 << 6) | b2;
        else cp = ((b0 & 0x07u) << 18) | (b1 << 12) | (b2 << 6) | b3;
        const uint32_t cls = jtk_class_of_cp(uc, cp);
        c0 |= (uint64_t)(cls & 1u) << j;
        c1 |= (uint64_t)(cls >> 1) << j;
    }
    spill_cls = JTK_CLS_O;
    if (lead) {
        const int jt = 63 - jtk_clz64(lead);
        if (((~k.CONT >> jt) >> 1) == 0) spill_cls = (uint32_t)((c0 >> jt) & 1ull) | (uint32_t)((c1 >> jt) & 1ull) << 1;
    }
    c0 = jtk_with_cont_bytes(c0, k.CONT);
    c1 = jtk_with_cont_bytes(c1, k.CONT);
    k.L |= c0 & ~c1;
    k.N |= c1 & ~c0;
    k.W |= c0 & c1;
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
