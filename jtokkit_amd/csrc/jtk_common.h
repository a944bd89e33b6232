// jtk_common.h -- definitions shared by the host table builder and the gfx950 kernels.
#ifndef JTK_COMMON_H
#define JTK_COMMON_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#include <hip/hip_runtime.h>
#define JTK_HD __host__ __device__ __forceinline__
#else
#define JTK_HD inline
#endif

// ---- per-byte class codes produced by the classify stage ------------------------------------------
// bits 0-1: class of the character this byte belongs to (continuation bytes inherit their lead's)
enum : uint32_t {
    JTK_CLS_O = 0,   // [^\s\p{L}\p{N}]
    JTK_CLS_L = 1,   // \p{L}
    JTK_CLS_N = 2,   // \p{N}
    JTK_CLS_W = 3,   // \s  (Unicode White_Space as the JDK defines it under UNICODE_CHARACTER_CLASS)
    JTK_CB_CLS = 3,
    JTK_CB_CONT = 4,   // UTF-8 continuation byte (not a character start)
    JTK_CB_NL = 8,     // '\r' or '\n'
    JTK_CB_SP = 16,    // U+0020
    JTK_CB_DS = 32     // a document starts at this byte; also set for every position >= n_bytes
};

enum { JTK_PAT_R50K = 0, JTK_PAT_CL100K = 1 };

// ---- rank-table encoding ----------------------------------------------------------------------------
// Every part of a piece during bytePairMerge is itself a table token (all 256 single bytes are
// tokens and a merge happens only on a table hit, GptBytePairEncoding.java:247-259), and the rank
// of a token is its id.  So getRank (GptBytePairEncoding.java:285-300) on the byte span of two
// adjacent parts equals a lookup of (id_left, id_right) in a table holding every split
// T = A + B with A, B, T all table tokens:  pair(id(A), id(B)) = rank(T).
// Slot layout: key = id_left << 17 | id_right (34 bits) in the high bits, rank in the low 30.
#define JTK_ID_BITS 17
#define JTK_MAX_ID ((1u << JTK_ID_BITS) - 2)
#define JTK_RANK_NONE 0x7FFFFFFFu          // Integer.MAX_VALUE of the reference
#define JTK_ID_DEAD 0xFFFFFFFFu            // byte position that does not start a part
#define JTK_PAIR_EMPTY 0xFFFFFFFFFFFFFFFFull
#define JTK_PAIR_RANK_MASK 0x3FFFFFFFull

JTK_HD uint64_t jtk_pair_key(uint32_t a, uint32_t b) { return ((uint64_t)a << JTK_ID_BITS) | b; }

// Two-choice bucketed cuckoo table: a key lives in one of two buckets of two 8-byte slots each, so a
// lookup is exactly two independent 16-byte loads -- never a dependent probe chain.  (On the device a
// wave advances at the pace of its slowest lane, so "usually one probe, sometimes four" costs four.)
// The bucket count `nb` is sized for the table to sit in one XCD's 4 MiB L2 with room to spare (any
// count, not a power of two: bucket = hash32 * nb >> 32).
JTK_HD uint32_t jtk_reduce32(uint32_t h, uint32_t nb) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(h, nb);
#else
    return (uint32_t)(((uint64_t)h * nb) >> 32);
#endif
}
JTK_HD uint32_t jtk_pair_hash(uint32_t a, uint32_t b, uint32_t nb) {
    uint32_t h = a * 0x9E3779B1u + b * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 16;
    return jtk_reduce32(h, nb);
}
JTK_HD uint32_t jtk_pair_hash2(uint32_t a, uint32_t b, uint32_t nb) {
    uint32_t h = a * 0xC2B2AE3Du + b * 0x27D4EB2Fu + 0x165667B1u;
    h ^= h >> 13;
    h *= 0x9E3779B1u;
    h ^= h >> 16;
    return jtk_reduce32(h, nb);
}

struct JtkPairBucket {          // 16 bytes: two slots
    uint32_t s0lo, s0hi, s1lo, s1hi;
};
struct JtkPairTable {
    const JtkPairBucket* buckets;
    uint32_t bits;   // number of buckets (the field keeps its old name)
};

JTK_HD uint32_t jtk_pair_match(const JtkPairBucket& v, uint64_t key) {
    const uint64_t a = ((uint64_t)v.s0hi << 32) | v.s0lo, b = ((uint64_t)v.s1hi << 32) | v.s1lo;
    if ((a >> 30) == key) return (uint32_t)(a & JTK_PAIR_RANK_MASK);
    if ((b >> 30) == key) return (uint32_t)(b & JTK_PAIR_RANK_MASK);
    return JTK_RANK_NONE;
}

JTK_HD uint32_t jtk_pair_lookup(const JtkPairTable& t, uint32_t a, uint32_t b) {
    const uint64_t key = jtk_pair_key(a, b);
    const JtkPairBucket v1 = t.buckets[jtk_pair_hash(a, b, t.bits)];
    const JtkPairBucket v2 = t.buckets[jtk_pair_hash2(a, b, t.bits)];
    const uint32_t r1 = jtk_pair_match(v1, key), r2 = jtk_pair_match(v2, key);
    return r1 != JTK_RANK_NONE ? r1 : r2;
}

// Two lookups, all four loads issued before any is examined.
JTK_HD void jtk_pair_lookup2(const JtkPairTable& t, uint32_t a1, uint32_t b1, bool want1, uint32_t a2, uint32_t b2,
                             bool want2, uint32_t& r1, uint32_t& r2) {
    const JtkPairBucket none{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    JtkPairBucket v11 = none, v12 = none, v21 = none, v22 = none;
    if (want1) { v11 = t.buckets[jtk_pair_hash(a1, b1, t.bits)]; v12 = t.buckets[jtk_pair_hash2(a1, b1, t.bits)]; }
    if (want2) { v21 = t.buckets[jtk_pair_hash(a2, b2, t.bits)]; v22 = t.buckets[jtk_pair_hash2(a2, b2, t.bits)]; }
    const uint64_t k1 = jtk_pair_key(a1, b1), k2 = jtk_pair_key(a2, b2);
    const uint32_t x1 = jtk_pair_match(v11, k1), y1 = jtk_pair_match(v12, k1);
    const uint32_t x2 = jtk_pair_match(v21, k2), y2 = jtk_pair_match(v22, k2);
    r1 = want1 ? (x1 != JTK_RANK_NONE ? x1 : y1) : JTK_RANK_NONE;
    r2 = want2 ? (x2 != JTK_RANK_NONE ? x2 : y2) : JTK_RANK_NONE;
}

// ---- 2-byte tokens, compressed for LDS: rank of (b0, b1) = ranks[cum[i >> 6] + popcount(bits[i >> 6] below i)]
#define JTK_BP_MAX 4096
struct JtkBpLds {
    const uint64_t* bits;     // [1024]
    const uint16_t* cum;      // [1024]
    const uint32_t* ranks;    // [JTK_BP_MAX]
};
JTK_HD uint32_t jtk_bp_lookup(const JtkBpLds& t, uint32_t idx) {
    const uint64_t wbits = t.bits[idx >> 6];
    const uint32_t j = idx & 63u;
    if (!((wbits >> j) & 1ull)) return JTK_RANK_NONE;
    const uint64_t below = wbits & ((1ull << j) - 1ull);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t k = (uint32_t)__popcll((unsigned long long)below);
#else
    const uint32_t k = (uint32_t)__builtin_popcountll(below);
#endif
    return t.ranks[t.cum[idx >> 6] + k];
}

// ---- whole-piece table for pieces of <= 8 bytes -------------------------------------------------------
// GptBytePairEncoding.java:81-83: a piece that is itself a table entry encodes to that one token.
// Key = the piece's bytes, little-endian in (lo, hi), zero padded, plus its length.  16-byte slots,
// two-choice cuckoo (one slot per choice): a lookup is two independent 16-byte loads.
struct JtkTok8Slot {
    uint32_t lo, hi, id, len;       // len == 0: empty
};
struct JtkTok8Table {
    const JtkTok8Slot* slots;
    uint32_t bits;   // number of slots (the field keeps its old name)
};
// ---- whole-piece table for pieces of 9..16 bytes ------------------------------------------------------
// Same shortcut (GptBytePairEncoding.java:81-83) for the longer words of ordinary text, which would otherwise
// cost the most merge steps.  Key = 16 bytes little-endian in k[4], zero padded, plus the length.  32-byte slots,
// two-choice cuckoo, one slot per choice.
struct JtkTok16Slot {
    uint32_t k[4];
    uint32_t id, len, pad0, pad1;   // len == 0: empty
};
struct JtkTok16Table {
    const JtkTok16Slot* slots;
    uint32_t n;
};
JTK_HD uint32_t jtk_tok16_hash(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t len, uint32_t nslots) {
    uint32_t h = k0 * 0x9E3779B1u + k1 * 0x85EBCA77u + k2 * 0xC2B2AE3Du + (k3 ^ (len << 27)) * 0x27D4EB2Fu;
    h ^= h >> 16;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return jtk_reduce32(h, nslots);
}
JTK_HD uint32_t jtk_tok16_hash2(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t len, uint32_t nslots) {
    uint32_t h = (k0 ^ (len << 29)) * 0xC2B2AE3Du + k1 * 0x27D4EB2Fu + k2 * 0x9E3779B1u + k3 * 0x165667B1u + 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return jtk_reduce32(h, nslots);
}

// the <= 8-byte table hashes the same way (upper key words zero), so the device computes one hash per choice whatever
// the piece's length
JTK_HD uint32_t jtk_tok8_hash(uint32_t lo, uint32_t hi, uint32_t len, uint32_t nslots) { return jtk_tok16_hash(lo, hi, 0u, 0u, len, nslots); }
JTK_HD uint32_t jtk_tok8_hash2(uint32_t lo, uint32_t hi, uint32_t len, uint32_t nslots) { return jtk_tok16_hash2(lo, hi, 0u, 0u, len, nslots); }

// ---- Unicode class lookup ----------------------------------------------------------------------------
struct JtkUcTables {
    const uint8_t* stage1;    // [0x1100]  cp >> 8 -> block
    const uint32_t* stage2;   // [blocks*16] 16 x 2-bit classes per word
};

JTK_HD uint32_t jtk_class_of_cp(const JtkUcTables& u, uint32_t cp) {
    if (cp > 0x10FFFFu) return JTK_CLS_O;
    const uint32_t blk = u.stage1[cp >> 8];
    const uint32_t w = u.stage2[blk * 16 + ((cp & 255u) >> 4)];
    return (w >> (2 * (cp & 15u))) & 3u;
}

JTK_HD uint32_t jtk_class_of_ascii(uint32_t b) {
    if (((b | 0x20u) - 'a') < 26u) return JTK_CLS_L;
    if ((b - '0') < 10u) return JTK_CLS_N;
    if (b == 0x20u || (b - 9u) < 5u) return JTK_CLS_W;
    return JTK_CLS_O;
}

// Class byte of position p.  `T` supplies byte(p) (0 outside the buffer).  Input is assumed to be
// well-formed UTF-8 (String.getBytes(UTF_8)); on malformed input the result is some class, never an
// out-of-range access.
template <class T, class I>
JTK_HD uint32_t jtk_class_byte(const T& txt, const JtkUcTables& u, I p) {
    const uint32_t b = txt.byte(p);
    if (b < 0x80u) {
        uint32_t cb = jtk_class_of_ascii(b);
        if (b == '\r' || b == '\n') cb |= JTK_CB_NL;
        if (b == 0x20u) cb |= JTK_CB_SP;
        return cb;
    }
    I lead = p;
    uint32_t flags = 0;
    if ((b & 0xC0u) == 0x80u) {
        flags = JTK_CB_CONT;
        lead = p - 1;
        if ((txt.byte(lead) & 0xC0u) == 0x80u) {
            lead = p - 2;
            if ((txt.byte(lead) & 0xC0u) == 0x80u) lead = p - 3;
        }
    }
    const uint32_t b0 = txt.byte(lead);
    uint32_t cp;
    if (b0 < 0xE0u) cp = ((b0 & 0x1Fu) << 6) | (txt.byte(lead + 1) & 0x3Fu);
    else if (b0 < 0xF0u) cp = ((b0 & 0x0Fu) << 12) | ((txt.byte(lead + 1) & 0x3Fu) << 6) | (txt.byte(lead + 2) & 0x3Fu);
    else cp = ((b0 & 0x07u) << 18) | ((txt.byte(lead + 1) & 0x3Fu) << 12) | ((txt.byte(lead + 2) & 0x3Fu) << 6)
              | (txt.byte(lead + 3) & 0x3Fu);
    return jtk_class_of_cp(u, cp) | flags;
}

#endif
