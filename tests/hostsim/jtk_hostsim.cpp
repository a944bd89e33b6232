// jtk_hostsim.cpp -- TEST INFRASTRUCTURE.  Runs the product's host/device-shared inline logic
// (jtk_split_rules.h, jtk_common.h) on the CPU so that the per-byte split rules can be checked against the oracle in the CPU-only test tier, where no GPU exists.
// Nothing in the product loads this library.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../jtokkit_amd/csrc/jtk_common.h"
#include "../../jtokkit_amd/csrc/jtk_split_rules.h"
#include "../../jtokkit_amd/csrc/jtk_split_masks.h"
#include "../../jtokkit_amd/csrc/jtk_block_classify.h"
#include "../../jtokkit_amd/csrc/jtk_unicode_tables.h"

namespace {
struct Txt {
    const uint8_t* t; int64_t n;
    uint32_t byte(int64_t p) const { return (p >= 0 && p < n) ? t[p] : 0u; }
};
struct Win {
    typedef int64_t idx_t;
    static constexpr int kMaxWalk = 0;
    const uint8_t* t; int64_t n; const uint8_t* cbv;
    uint32_t byte(int64_t p) const { return (p >= 0 && p < n) ? t[p] : 0u; }
    uint32_t cb(int64_t p) const { return (p >= 0 && p < n) ? cbv[p] : (uint32_t)JTK_CB_DS; }
};
}

extern "C" {

// ms_out[n_bytes+1]: 1 where a piece starts (position n_bytes is the sentinel and always 1)
int sim_split(int kind, const uint8_t* text, int64_t n, const int64_t* doc_off, int64_t n_docs, uint8_t* ms_out) {
    JtkUcTables u{jtk_uc_stage1_init, jtk_uc_stage2_init};
    std::vector<uint8_t> cb((size_t)n + 1);
    Txt txt{text, n};
    for (int64_t p = 0; p < n; p++) cb[p] = (uint8_t)jtk_class_byte(txt, u, p);
    for (int64_t d = 0; d <= n_docs; d++) if (doc_off[d] < n) cb[doc_off[d]] |= JTK_CB_DS;
    Win w{text, n, cb.data()};
    for (int64_t p = 0; p <= n; p++) ms_out[p] = jtk_is_piece_start(w, p, kind) ? 1 : 0;
    return 0;
}

}  // extern "C"

// The kernel's form: 64-byte blocks as bit masks (jtk_split_masks.h), waves of `wave_blocks` blocks that
// each start from the block before them, slow lanes through jtk_is_piece_start_t.
static JtkBlk make_blk(const uint8_t* text, int64_t n, const uint8_t* cb, int64_t b, bool ci) {
    JtkBlk k;
    memset(&k, 0, sizeof(k));
    for (int j = 0; j < 64; j++) {
        const int64_t p = b * 64 + j;
        const uint64_t bit = 1ull << j;
        if (p < 0) continue;
        if (p >= n) { k.DS |= bit; continue; }
        const uint32_t c = cb[p], raw = text[p];
        const uint32_t cls = c & JTK_CB_CLS;
        if (cls == JTK_CLS_L) k.L |= bit;
        if (cls == JTK_CLS_N) k.N |= bit;
        if (cls == JTK_CLS_W) k.W |= bit;
        if (c & JTK_CB_CONT) k.CONT |= bit;
        if (c & JTK_CB_NL) k.NL |= bit;
        if (c & JTK_CB_SP) k.SP |= bit;
        if (c & JTK_CB_DS) k.DS |= bit;
        if (raw == '\'') k.AP |= bit;
        const uint32_t f = (ci && (raw - 'A') < 26u) ? (raw | 0x20u) : raw;
        if (f == 's' || f == 't' || f == 'm' || f == 'd') k.S1 |= bit;
        if (f == 'r' || f == 'v') k.RV |= bit;
        if (f == 'e') k.E |= bit;
        if (f == 'l') k.LL |= bit;
        if (raw == 0xC5) k.C5 |= bit;
        if (raw == 0xBF) k.BF |= bit;
    }
    return k;
}

template <int KIND>
static void sim_split_masks_t(const uint8_t* text, int64_t n, const uint8_t* cbv, int wave_blocks, uint8_t* ms_out,
                              int64_t* n_slow, int64_t* stats = nullptr) {
    Win w{text, n, cbv};
    const bool ci = KIND == JTK_PAT_CL100K;
    const int64_t nblk = (n + 1 + 63) / 64;
    for (int64_t f = 0; f < nblk; f += wave_blocks) {
        JtkSplitCarry cy;
        JtkBlk halo = make_blk(text, n, cbv, f - 1, ci), cu = make_blk(text, n, cbv, f, ci);
        jtk_split_carry_from_halo<KIND>(halo, cu, cy);
        for (int64_t b = f; b < f + wave_blocks && b < nblk; b++) {
            JtkBlk nx = make_blk(text, n, cbv, b + 1, ci);
            uint64_t slow = 0, nl = 0;
            const JtkSplitCarry before = cy;
            uint64_t ms = jtk_split_block<KIND>(cu, nx, cy, slow, nl);
            if (KIND == JTK_PAT_CL100K && nl && (cu.N & cu.CONT) == 0) {    // as the kernel does
                uint64_t nslow;
                ms |= jtk_split_n_block(cu, before.pN, before.ncnt, before.n_unknown, nslow);
                slow |= nslow;
                nl = 0;
            }
            if (stats) {
                stats[0]++; stats[1] += __builtin_popcountll(slow); stats[2] += __builtin_popcountll(nl);
                stats[3] += slow != 0; stats[4] += nl != 0;
            }
            for (int j = 0; j < 64; j++) {
                const int64_t p = b * 64 + j;
                if (p > n) break;
                bool v = (ms >> j) & 1ull;
                bool sl = (slow >> j) & 1ull;
                if ((nl >> j) & 1ull) { bool s2 = false; v = jtk_split_n_lane(cu, before.ncnt, before.n_unknown, j, s2); sl = sl || s2; }
                if (sl) { bool dummy = false; v = jtk_is_piece_start_t<KIND>(w, p, dummy); (*n_slow)++; }
                ms_out[p] = v ? 1 : 0;
            }
            cu = nx;
        }
    }
}

// blocks, slow positions, digit-lane positions, blocks with any of either (what the kernel's per-position loops see)
extern "C" int sim_split_stats(int kind, const uint8_t* text, int64_t n, const int64_t* doc_off, int64_t n_docs, int64_t* stats) {
    JtkUcTables u{jtk_uc_stage1_init, jtk_uc_stage2_init};
    std::vector<uint8_t> cb((size_t)n + 1), ms((size_t)n + 1);
    Txt txt{text, n};
    for (int64_t p = 0; p < n; p++) cb[p] = (uint8_t)jtk_class_byte(txt, u, p);
    for (int64_t d = 0; d <= n_docs; d++) if (doc_off[d] < n) cb[doc_off[d]] |= JTK_CB_DS;
    int64_t n_slow = 0;
    for (int i = 0; i < 5; i++) stats[i] = 0;
    if (kind == JTK_PAT_CL100K) sim_split_masks_t<JTK_PAT_CL100K>(text, n, cb.data(), 62, ms.data(), &n_slow, stats);
    else sim_split_masks_t<JTK_PAT_R50K>(text, n, cb.data(), 62, ms.data(), &n_slow, stats);
    return 0;
}

extern "C" int sim_split_masks(int kind, const uint8_t* text, int64_t n, const int64_t* doc_off, int64_t n_docs,
                               int wave_blocks, uint8_t* ms_out, int64_t* n_slow) {
    JtkUcTables u{jtk_uc_stage1_init, jtk_uc_stage2_init};
    std::vector<uint8_t> cb((size_t)n + 1);
    Txt txt{text, n};
    for (int64_t p = 0; p < n; p++) cb[p] = (uint8_t)jtk_class_byte(txt, u, p);
    for (int64_t d = 0; d <= n_docs; d++) if (doc_off[d] < n) cb[doc_off[d]] |= JTK_CB_DS;
    *n_slow = 0;
    if (kind == JTK_PAT_CL100K) sim_split_masks_t<JTK_PAT_CL100K>(text, n, cb.data(), wave_blocks, ms_out, n_slow);
    else sim_split_masks_t<JTK_PAT_R50K>(text, n, cb.data(), wave_blocks, ms_out, n_slow);
    return 0;
}

// block-parallel classification (jtk_block_classify.h) vs the per-byte construction: number of blocks that differ
extern "C" int64_t sim_block_classify_check(int kind, const uint8_t* text, int64_t n) {
    JtkUcTables u{jtk_uc_stage1_init, jtk_uc_stage2_init};
    std::vector<uint8_t> cb((size_t)n + 1);
    Txt txt{text, n};
    for (int64_t p = 0; p < n; p++) cb[p] = (uint8_t)jtk_class_byte(txt, u, p);
    const bool ci = kind == JTK_PAT_CL100K;
    JtkCode4 codes[256];
    for (int b = 0; b < 256; b++)
        codes[b] = jtk_code4(jtk_byte_code((uint32_t)b, ci) | (jtk_lead_all_letters(u, (uint32_t)b) ? (uint32_t)JTK_F_ULL : 0u));
    const int64_t nblk = (n + 63) / 64;
    int64_t bad = 0;
    uint32_t spill = JTK_CLS_O;
    struct Words {                                   // the 4 bytes that start at byte j of block b (zero beyond the text)
        const uint8_t* t; int64_t n, base;
        uint32_t word(int j) const {
            uint32_t v = 0;
            for (int r = 0; r < 4; r++) { const int64_t p = base + j + r; if (p < n) v |= (uint32_t)t[p] << (8 * r); }
            return v;
        }
    };
    for (int64_t b = 0; b < nblk; b++) {
        uint32_t d[16];
        for (int q = 0; q < 16; q++) {
            uint32_t v = 0;
            for (int r = 0; r < 4; r++) { const int64_t p = b * 64 + q * 4 + r; if (p < n) v |= (uint32_t)text[p] << (8 * r); }
            d[q] = v;
        }
        JtkBlk k;
        memset(&k, 0, sizeof(k));
        uint64_t lead = 0, lt = 0, ull = 0;
        jtk_block_masks_ascii(d, codes, k, lead, lt, ull);
        uint32_t my_spill;
        const Words wtxt{text, n, b * 64};
        jtk_block_fix_nonascii(wtxt, u, lead, ull, k, my_spill);
        jtk_block_apply_spill(k, spill);
        spill = my_spill;
        JtkBlk ref = make_blk(text, n, cb.data(), b, ci);
        // bytes beyond n: the reference construction sets DS only; zero bytes classify as O in both
        const uint64_t valid = (b * 64 + 64 <= n) ? ~0ull : ((1ull << (n - b * 64)) - 1ull);
        const bool same = ((k.L ^ ref.L) & valid) == 0 && ((k.N ^ ref.N) & valid) == 0 && ((k.W ^ ref.W) & valid) == 0 &&
                          ((k.CONT ^ ref.CONT) & valid) == 0 && ((k.NL ^ ref.NL) & valid) == 0 && ((k.SP ^ ref.SP) & valid) == 0 &&
                          ((k.AP ^ ref.AP) & valid) == 0 && ((k.S1 ^ ref.S1) & valid) == 0 && ((k.RV ^ ref.RV) & valid) == 0 &&
                          ((k.E ^ ref.E) & valid) == 0 && ((k.LL ^ ref.LL) & valid) == 0 && ((k.C5 ^ ref.C5) & valid) == 0 &&
                          ((k.BF ^ ref.BF) & valid) == 0;
        if (!same) bad++;
    }
    return bad;
}

extern "C" int sim_lead_all_letters(int b) {
    JtkUcTables u{jtk_uc_stage1_init, jtk_uc_stage2_init};
    return jtk_lead_all_letters(u, (uint32_t)b) ? 1 : 0;
}

extern "C" {
uint32_t sim_class_byte(const uint8_t* text, int64_t n, int64_t p) {
    JtkUcTables u{jtk_uc_stage1_init, jtk_uc_stage2_init};
    Txt txt{text, n};
    return jtk_class_byte(txt, u, p);
}
}

// ---- rank tables + lane merge on the host -----------------------------------------------------------
#include <string>
#include "../../jtokkit_amd/csrc/jtk_merge_core.h"
#include "../../jtokkit_amd/csrc/jtk_tables.h"

extern "C" {
void* sim_tables_create(const char* name, int kind, const uint8_t* data, size_t len, int* status) {
    JtkHostTables* t = new JtkHostTables();
    std::string err;
    int rc = jtk_build_tables(name, kind, data, len, nullptr, nullptr, 0, *t, err);
    if (status) *status = rc;
    if (rc != 0) { fprintf(stderr, "sim_tables_create: %s\n", err.c_str()); delete t; return nullptr; }
    return t;
}
void sim_tables_destroy(void* h) { delete (JtkHostTables*)h; }
int64_t sim_tables_pairs(void* h) { return ((JtkHostTables*)h)->n_pairs; }
// first pseudo id of a table that lacks single-byte tokens (0: none are missing)
int64_t sim_tables_pseudo_base(void* h) { JtkHostTables* t = (JtkHostTables*)h; return t->n_missing ? (int64_t)t->pseudo_base : 0; }
int sim_tables_bits(void* h) { return (int)((JtkHostTables*)h)->pair_bits; }
// whole-piece table lookup (pieces of <= 8 bytes): id or -1
int64_t sim_tok8_lookup(void* h, const uint8_t* piece, int len) {
    JtkHostTables* t = (JtkHostTables*)h;
    if (len < 1 || len > 8) return -1;
    uint32_t lo = 0, hi = 0;
    for (int k = 0; k < len; k++) { if (k < 4) lo |= (uint32_t)piece[k] << (8 * k); else hi |= (uint32_t)piece[k] << (8 * (k - 4)); }
    const JtkTok8Table tt{t->tok8.data(), t->tok8_bits};
    const uint32_t id = jtk_tok8_find(tt, lo, hi, (uint32_t)len);
    return id == JTK_RANK_NONE ? -1 : (int64_t)id;
}
// fraction of the pair-table / tok8 entries that live in their secondary bucket (lookups of those cost two fetches)
double sim_pair_displaced(void* h) { JtkHostTables* t = (JtkHostTables*)h; return (double)t->pair_displaced / (double)t->n_pairs; }
double sim_tok8_displaced(void* h) { JtkHostTables* t = (JtkHostTables*)h; return (double)t->tok8_displaced / (double)t->n_tok8; }
int sim_tok8_bits(void* h) { return (int)((JtkHostTables*)h)->tok8_bits; }
int64_t sim_tok8_count(void* h) { return ((JtkHostTables*)h)->n_tok8; }
// fraction of buckets' slots in use
double sim_tables_avg_probe(void* h) {
    JtkHostTables* t = (JtkHostTables*)h;
    double used = 0;
    for (auto& b : t->pair_buckets) {
        if (!(b.k0 == 0xFFFFFFFFu && b.r0 == 0xFFFFFFFFu)) used++;
        if (!(b.k1 == 0xFFFFFFFFu && b.r1 == 0xFFFFFFFFu)) used++;
    }
    return used / (2.0 * (double)t->pair_buckets.size());
}
// bytePairMerge of one piece (len <= 64) with the device's lane algorithm
int sim_merge_piece(void* h, const uint8_t* piece, int len, int32_t* out) {
    JtkHostTables* t = (JtkHostTables*)h;
    if (len < 1 || len > 64) return -1;
    uint32_t ids[64], rk[64];
    for (int i = 0; i < len; i++) ids[i] = t->byte_rank[piece[i]];
    JtkPairTable pt{t->pair_buckets.data(), t->pair_bits};
    jtk_merge_piece_lane(ids, rk, len, pt);
    // the device's form (direct byte-pair table + paired lookups) must agree with the plain one
    uint32_t ids2[64], rk2[64];
    jtk_merge_piece_lane2(ids2, rk2, piece, len, pt, t->bp_rank.data(), t->byte_rank);
    for (int i = 0; i < len; i++) if (ids[i] != ids2[i]) return -2;
    int n = 0;
    for (int i = 0; i < len; i++) if (ids[i] != JTK_ID_DEAD) out[n++] = (int32_t)ids[i];
    return n;
}
}
