// jtk_strip_common.h -- what k_strip_encode, k_bpe_merge and k_strip_expand share: the piece mask as the kernels see it, the
// 8-byte hole record, the memo of merged pieces, a few wave primitives.  Included by jtk_kernels.hip inside its anonymous
// namespace.

__device__ __forceinline__ uint64_t piece_word(const JtkWork& w, int64_t wd) {
    // piece starts of mask word wd: positions before the chunk's first document and positions >= n start no piece here, but
    // the end sentinel (bit n) stays: it ends the last piece
    uint64_t m = (wd < w.n_words) ? w.piecemask[wd] : 0ull;
    const int64_t n = w.n_bytes, p0 = wd * 64;
    if (p0 + 63 > n) m &= (p0 > n) ? 0ull : ((2ull << (n - p0)) - 1ull);          // keep positions <= n
    if (p0 < w.lead) m &= (p0 + 64 <= w.lead) ? 0ull : ~((1ull << (w.lead - p0)) - 1ull);
    return m;
}

__device__ __forceinline__ uint32_t mbcnt64_(uint64_t m) {          // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}


// Where the strips' slots and hole records live.  stok (4 B) and hrec (8 B) are indexed alike: one index per piece.  A wave of
// k_strip_encode owns a REGION of that index space (wave_cap indices: enough for the worst case of all its strips) and fills
// it densely, strip after strip, each strip's pieces in order (sbase[strip]: where the strip starts in its wave's region).
// Worst-case regions per STRIP (4096 indices each, a few hundred used) were tried first: the sparse footprint -- 16 and 32 KB
// strides -- made every memory instruction of the expand kernel several times slower than its bytes explain.
__device__ __forceinline__ int64_t strip_region(const JtkWork& w, int64_t strip) {
    const int64_t g = w.n_shards;
    return ((strip % g) * w.enc_waves + (strip / g) % w.enc_waves) * (int64_t)w.wave_cap;
}

// A strip's slot stream (stok) holds one 32-bit slot per piece, in text order:
//   a token id            the piece's one token (a dense piece, or a hole that turned out to be one token)
//   SLOT_HOLE             placeholder of a hole whose result is not in yet (none is left when k_strip_expand runs)
//   SLOT_MULTI | count    a hole with `count` tokens (0: none): they are in its hole record, hrec[the slot's index]
constexpr uint32_t SLOT_HOLE = 0xFFFFFFFFu;
constexpr uint32_t SLOT_MULTI = 0x80000000u;

// hole record (8 bytes)
constexpr int HR_KIND_SHIFT = 53;              // bits 53..54
constexpr uint64_t HR_TOKS = 0;                // bits 0..50: up to three token ids, 17 bits each; bits 51..52: count - 1
constexpr uint64_t HR_REF = 1;                 // bits 0..20: count; bits 21..32: offset of the piece in the strip: tokens in htok
constexpr uint64_t HR_GAP = 2;                 // no tokens


// Memo of merged pieces (per XCD, insert-only, cleared per job): a piece of 4..16 bytes that bytePairMerge turned into at most
// six tokens is remembered under its bytes, so that its next occurrence -- natural text repeats its words -- costs one
// lookup in the hole batch instead of a merge.  An entry is 32 bytes: the piece's 16 key bytes | lo64 | hi64 with
//   lo64 = tokens 0..2 (17 bits each) | tag13 << 51;   hi64 = tokens 3..5 | count << 51 | len << 54 | tag5 << 59.
// Only the wave that claims an empty slot (compare-and-swap of hi64 from 0 to MEMO_BUSY) ever writes it, and never again, so
// each of the entry's words is either still zero or final: a reader that finds its key, its length and both tags (nonzero,
// taken from the key's hash) has read a complete entry of exactly its key; anything else is a miss, and a miss only costs the
// merge.  Each XCD has its own table: its L2 is the point of coherence for all its CUs, and nothing crosses XCDs.
constexpr uint64_t MEMO_BUSY = 1ull << 51;
constexpr uint32_t MEMO_MAX_TOKENS = 6;

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}
__device__ __forceinline__ uint32_t memo_slot(uint32_t mix, uint32_t mask) { return (jtk_pair_mix2(mix) ^ (mix >> 9)) & mask; }
__device__ __forceinline__ uint32_t memo_tag(uint32_t mix) { return ((mix >> 17) & 0x1FFFu) | 1u; }

// the result of the hole at index `idx`: its slot, and its record unless it is a single token
__device__ __forceinline__ void put_hole(const JtkWork& w, int64_t idx, uint32_t cnt, uint64_t rec) {
    const bool one = cnt == 1u && ((rec >> HR_KIND_SHIFT) & 3ull) == HR_TOKS;
    w.stok[idx] = one ? (uint32_t)rec & JTK_HT_ID_MASK : SLOT_MULTI | cnt;
    if (!one && cnt) w.hrec[idx] = rec;
}

struct __attribute__((packed, aligned(1))) U4Bytes { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U2Bytes { uint32_t x, y; };

// the 16 bytes at text position p (bytes at or beyond n read as zero); p + 16 <= n is the fast path
__device__ __forceinline__ uint4 load_text16(const uint8_t* text, int64_t p, int64_t n) {
    if (p + 16 <= n) {
        const U4Bytes v = *reinterpret_cast<const U4Bytes*>(text + p);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    uint32_t tmp[4] = {0, 0, 0, 0};
    for (int j = 0; j < 16; j++) if (p + j < n) tmp[j >> 2] |= (uint32_t)text[p + j] << (8 * (j & 3));
    return make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]);
}

// inclusive prefix sum across the wave with DPP row shifts and row broadcasts (six adds)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);     // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);     // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);     // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);     // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);    // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);    // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) { return mbcnt64_(m); }

// Token counts of a batch of lanes to tile_tot: the lanes' strips come in runs (ring order is text order), so the first
// lane of each run adds the run's sum.  All lanes call this; lanes without a piece pass have = false.
__device__ __forceinline__ void add_strip_counts(const JtkWork& w, uint32_t strip, bool have, uint32_t cc) {
    const int lane = threadIdx.x & 63;
    const uint32_t key = have ? strip : 0xFFFFFFFFu;
    const uint32_t c = have ? cc : 0u;
    const uint32_t inc = wave_incl_scan_dpp(c);
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
    const bool head = lane == 0 || prev != key;
    const uint64_t heads = __ballot(head);
    const uint64_t later = heads & ~((2ull << lane) - 1ull);                  // run heads after this lane
    const int last = later ? jtk_ctz64(later) - 1 : 63;                        // last lane of this lane's run
    const uint32_t run_end = (uint32_t)__shfl((int)inc, last);
    if (head && have) {
        const uint32_t sum = run_end - (inc - c);
        if (sum) atomicAdd(&w.tile_tot[strip], sum);
    }
}

