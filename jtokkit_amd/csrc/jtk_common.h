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

// bit scans (x != 0)
#if defined(__HIP_DEVICE_COMPILE__)
JTK_HD int jtk_ctz64(uint64_t x) { return __ffsll((unsigned long long)x) - 1; }
JTK_HD int jtk_clz64(uint64_t x) { return __clzll((long long)x); }
#else
JTK_HD int jtk_ctz64(uint64_t x) { return __builtin_ctzll(x); }
JTK_HD int jtk_clz64(uint64_t x) { return __builtin_clzll(x); }
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
#define JTK_PAIR_R_MASK 0x1FFFFFFFu         // rank bits of a slot's r word
#define JTK_PAIR_OVERFLOW 0x20000000u       // in r0: a key whose primary bucket this is lives in its secondary bucket

JTK_HD uint64_t jtk_pair_key(uint32_t a, uint32_t b) { return ((uint64_t)a << JTK_ID_BITS) | b; }

// Two-choice bucketed cuckoo tables, PRIMARY FIRST.  What bounds the lookups on the device is the number of
// scattered cache-line fetches (a CU sustains about one L2 line per 2.3 clocks whatever the width of the load:
// tools/microbench/gather_rate.hip), so the tables are built for ONE 16-byte load per lookup in the common case:
// a key lives in its primary bucket unless that was full when the table was built; a bucket that turned a key
// away carries an OVERFLOW flag.  A lookup reads the primary bucket; a hit, or a miss without the flag, is final
// -- also for keys that are not in the table, which is most lookups of bytePairMerge.  Only a miss in a flagged
// bucket reads the secondary one (a second, dependent fetch for those lanes only).
// Bucket counts are any number, not a power of two: bucket = hash32 * nb >> 32.
JTK_HD uint32_t jtk_reduce32(uint32_t h, uint32_t nb) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(h, nb);
#else
    return (uint32_t)(((uint64_t)h * nb) >> 32);
#endif
}
// The two bucket choices share one 32-bit mix of (a, b): the second hash costs one multiply-add, one xor-shift and
// the reduction on top of the first.
JTK_HD uint32_t jtk_pair_mix(uint32_t a, uint32_t b) {
    uint32_t h = a * 0x9E3779B1u + b * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 16;
    return h;
}
JTK_HD uint32_t jtk_pair_mix2(uint32_t m) {
    uint32_t h = m * 0x27D4EB2Fu + 0x165667B1u;
    h ^= h >> 15;
    return h;
}
JTK_HD uint32_t jtk_pair_hash(uint32_t a, uint32_t b, uint32_t nb) { return jtk_reduce32(jtk_pair_mix(a, b), nb); }
JTK_HD uint32_t jtk_pair_hash2(uint32_t a, uint32_t b, uint32_t nb) { return jtk_reduce32(jtk_pair_mix2(jtk_pair_mix(a, b)), nb); }

// 16 bytes: two slots.  Slot = (k, r): k = low 32 bits of the 34-bit key (id_left << 17 | id_right), r = rank (bits
// 0..28) | the key's top two bits << 30.  Bit 29 of r0 is the bucket's overflow flag.  An empty slot has k = ~0 and
// key bits 11 (no id reaches 2^17 - 1).
struct JtkPairBucket {
    uint32_t k0, r0, k1, r1;
};
struct JtkPairTable {
    const JtkPairBucket* buckets;
    uint32_t bits;   // number of buckets (the field keeps its old name)
};

// branch-free: the rank stored under (klo, ktop) in this bucket, or JTK_RANK_NONE.  klo = (uint32_t)key,
// ktop = (uint32_t)(key >> 32) << 30.
JTK_HD uint32_t jtk_pair_match2(uint32_t k0, uint32_t r0, uint32_t k1, uint32_t r1, uint32_t klo, uint32_t ktop) {
    const uint32_t d0 = (k0 ^ klo) | ((r0 ^ ktop) & 0xC0000000u);
    const uint32_t d1 = (k1 ^ klo) | ((r1 ^ ktop) & 0xC0000000u);
    uint32_t r = JTK_RANK_NONE;
    r = d1 == 0u ? (r1 & JTK_PAIR_R_MASK) : r;
    r = d0 == 0u ? (r0 & JTK_PAIR_R_MASK) : r;
    return r;
}
JTK_HD uint32_t jtk_pair_match(const JtkPairBucket& v, uint64_t key) {
    return jtk_pair_match2(v.k0, v.r0, v.k1, v.r1, (uint32_t)key, (uint32_t)(key >> 32) << 30);
}

JTK_HD uint32_t jtk_pair_lookup(const JtkPairTable& t, uint32_t a, uint32_t b) {
    const uint64_t key = jtk_pair_key(a, b);
    const uint32_t m = jtk_pair_mix(a, b);
    const JtkPairBucket v1 = t.buckets[jtk_reduce32(m, t.bits)];
    const uint32_t r1 = jtk_pair_match(v1, key);
    if (r1 != JTK_RANK_NONE || !(v1.r0 & JTK_PAIR_OVERFLOW)) return r1;
    const JtkPairBucket v2 = t.buckets[jtk_reduce32(jtk_pair_mix2(m), t.bits)];
    return jtk_pair_match(v2, key);
}

// Two lookups (the two neighbours of a merge).
JTK_HD void jtk_pair_lookup2(const JtkPairTable& t, uint32_t a1, uint32_t b1, bool want1, uint32_t a2, uint32_t b2,
                             bool want2, uint32_t& r1, uint32_t& r2) {
    r1 = want1 ? jtk_pair_lookup(t, a1, b1) : JTK_RANK_NONE;
    r2 = want2 ? jtk_pair_lookup(t, a2, b2) : JTK_RANK_NONE;
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
// Key = the piece's bytes, little-endian in (lo, hi), zero padded, plus its length.  16-byte slots, two-choice
// cuckoo, one slot per bucket, primary first (see the pair table): one 16-byte load per lookup unless the primary
// slot carries the overflow flag.
#define JTK_TOK_OVERFLOW 0x80000000u        // in len: a key whose primary slot this is lives in its secondary slot
#define JTK_TOK_LEN_MASK 0xFFu
// ... and which: bits 8..23 of len are a 16-bit filter over the low four bits of those keys' mixes, so that a key that is
// not in the table at all (every piece that has to be merged) seldom has to look at its secondary slot to learn so
#define JTK_TOK_FILTER_BIT(mix) (1u << (8u + ((mix) & 15u)))
struct JtkTok8Slot {
    uint32_t lo, hi, id, len;       // len & JTK_TOK_LEN_MASK == 0: empty
};
struct JtkTok8Table {
    const JtkTok8Slot* slots;
    uint32_t bits;   // number of slots (the field keeps its old name)
};
// ---- whole-piece table for pieces of 9..16 bytes ------------------------------------------------------
// Same shortcut (GptBytePairEncoding.java:81-83) for the longer words of ordinary text, which would otherwise
// cost the most merge steps.  Key = 16 bytes little-endian in k[4], zero padded, plus the length.  32-byte slots,
// two-choice cuckoo, one slot per bucket, primary first.
struct JtkTok16Slot {
    uint32_t k[4];
    uint32_t id, len, pad0, pad1;   // len as in JtkTok8Slot
};
struct JtkTok16Table {
    const JtkTok16Slot* slots;
    uint32_t n;
};
// one 32-bit mix of the key for both bucket choices (the second choice re-mixes it: jtk_pair_mix2)
JTK_HD uint32_t jtk_tok16_mix(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t len) {
    uint32_t h = k0 * 0x9E3779B1u + k1 * 0x85EBCA77u + k2 * 0xC2B2AE3Du + (k3 ^ (len << 27)) * 0x27D4EB2Fu;
    h ^= h >> 16;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return h;
}
JTK_HD uint32_t jtk_tok16_hash(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t len, uint32_t nslots) {
    return jtk_reduce32(jtk_tok16_mix(k0, k1, k2, k3, len), nslots);
}
JTK_HD uint32_t jtk_tok16_hash2(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t len, uint32_t nslots) {
    return jtk_reduce32(jtk_pair_mix2(jtk_tok16_mix(k0, k1, k2, k3, len)), nslots);
}

// the <= 8-byte table hashes the same way (upper key words zero), so the device computes one hash per choice whatever
// the piece's length
JTK_HD uint32_t jtk_tok8_hash(uint32_t lo, uint32_t hi, uint32_t len, uint32_t nslots) { return jtk_tok16_hash(lo, hi, 0u, 0u, len, nslots); }
JTK_HD uint32_t jtk_tok8_hash2(uint32_t lo, uint32_t hi, uint32_t len, uint32_t nslots) { return jtk_tok16_hash2(lo, hi, 0u, 0u, len, nslots); }

// id of the piece (lo, hi, len <= 8) or JTK_RANK_NONE
JTK_HD uint32_t jtk_tok8_find(const JtkTok8Table& t, uint32_t lo, uint32_t hi, uint32_t len) {
    const JtkTok8Slot a = t.slots[jtk_tok8_hash(lo, hi, len, t.bits)];
    if ((a.len & JTK_TOK_LEN_MASK) == len && a.lo == lo && a.hi == hi) return a.id;
    if (!(a.len & JTK_TOK_OVERFLOW)) return JTK_RANK_NONE;
    const JtkTok8Slot b = t.slots[jtk_tok8_hash2(lo, hi, len, t.bits)];
    if ((b.len & JTK_TOK_LEN_MASK) == len && b.lo == lo && b.hi == hi) return b.id;
    return JTK_RANK_NONE;
}
JTK_HD uint32_t jtk_tok16_find(const JtkTok16Table& t, const uint32_t (&k)[4], uint32_t len) {
    const JtkTok16Slot a = t.slots[jtk_tok16_hash(k[0], k[1], k[2], k[3], len, t.n)];
    if ((a.len & JTK_TOK_LEN_MASK) == len && a.k[0] == k[0] && a.k[1] == k[1] && a.k[2] == k[2] && a.k[3] == k[3]) return a.id;
    if (!(a.len & JTK_TOK_OVERFLOW)) return JTK_RANK_NONE;
    const JtkTok16Slot b = t.slots[jtk_tok16_hash2(k[0], k[1], k[2], k[3], len, t.n)];
    if ((b.len & JTK_TOK_LEN_MASK) == len && b.k[0] == k[0] && b.k[1] == k[1] && b.k[2] == k[2] && b.k[3] == k[3]) return b.id;
    return JTK_RANK_NONE;
}

// ---- whole-piece table for table entries of more than 16 bytes that bytePairMerge does not reproduce ---------------
// GptBytePairEncoding.java:81-83 applies the whole-piece lookup to pieces of any length.  For a rank table in which merging
// a token's bytes yields exactly that token (every table trained by byte-pair merging; the three shipped ones) the lookup is
// a pure shortcut and the device applies it to pieces of <= 16 bytes only.  Hand-made tables may hold longer entries that
// merging cannot produce: those (and only those) are listed here and looked up by a small kernel before the merge.
// Open addressing, linear probing, load <= 0.5; key = 64-bit FNV-1a of the bytes, verified against the bytes themselves.
struct JtkLongTokSlot {
    uint32_t h_lo, h_hi, id, len;       // len == 0: empty
    uint32_t blob_off, pad0, pad1, pad2;
};
struct JtkLongTokTable {
    const JtkLongTokSlot* slots;
    const uint8_t* blob;
    uint32_t n;                         // slots (0: the encoding needs no such lookups)
    uint32_t max_len;
};
JTK_HD uint64_t jtk_fnv1a_step(uint64_t h, uint32_t b) { return (h ^ (uint64_t)b) * 0x100000001B3ull; }
#define JTK_FNV_BASIS 0xCBF29CE484222325ull

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
