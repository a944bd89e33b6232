// jtk_tables.cpp -- rank-file parser and derived-table builder (cold path, host only).
//
// Input format (reference EncodingFactory.java:139-164): one token per line, `base64(bytes) SP rank`.
// Derived tables:
//   byte_rank[256]   id of every single-byte token (the reference's final emit for an unmerged byte
//                    is a table lookup, not the identity: rank('!') = 0, rank(' ') = 220 in cl100k)
//   pair table       every split T = A + B with A, B, T table tokens: (id(A), id(B)) -> rank(T)
//                    (see jtk_common.h for why this equals getRank on byte spans)
#include "jtk_tables.h"

#include <cstdlib>
#include <cstring>

#include "../../include/jtokkit_amd.h"
#include "jtk_merge_core.h"

namespace {

int b64val(int c) {
    if (c >= 'A' && c <= 'Z') return c - 'A';
    if (c >= 'a' && c <= 'z') return c - 'a' + 26;
    if (c >= '0' && c <= '9') return c - '0' + 52;
    if (c == '+') return 62;
    if (c == '/') return 63;
    return -1;
}

bool b64decode(const char* s, size_t n, std::string& out) {
    out.clear();
    if (n == 0 || n % 4 != 0) return false;
    size_t pad = 0;
    if (s[n - 1] == '=') pad++;
    if (s[n - 2] == '=') pad++;
    for (size_t i = 0; i < n; i += 4) {
        uint32_t w = 0;
        for (int j = 0; j < 4; j++) {
            const char c = s[i + j];
            int v;
            if (c == '=' && i + 4 == n && (size_t)j >= 4 - pad) v = 0;
            else if ((v = b64val(c)) < 0) return false;
            w = (w << 6) | (uint32_t)v;
        }
        const bool last = (i + 4 == n);
        out.push_back((char)(w >> 16));
        if (!last || pad < 2) out.push_back((char)(w >> 8));
        if (!last || pad < 1) out.push_back((char)w);
    }
    return true;
}

inline bool is_ws(char c) { return c == ' ' || (c >= 9 && c <= 13); }

}  // namespace

int jtk_build_tables(const char* name, int kind, const uint8_t* data, size_t len,
                     const char* const* special_literals, const int32_t* special_ids, int n_specials,
                     JtkHostTables& t, std::string& err) {
    if (kind != JTK_PAT_R50K && kind != JTK_PAT_CL100K) { err = "unknown pattern kind"; return JTK_ERR_INVALID_ARGUMENT; }
    t.name = name ? name : "";
    t.kind = kind;
    std::vector<std::pair<std::string, uint32_t>> entries;
    size_t i = 0;
    std::string tok;
    while (i < len) {
        size_t e = i;
        while (e < len && data[e] != '\n') e++;
        size_t le = e;
        if (le > i && data[le - 1] == '\r') le--;
        const char* line = (const char*)data + i;
        const size_t n = le - i;
        i = e + 1;
        size_t sp = 0;
        while (sp < n && !is_ws(line[sp])) sp++;
        size_t r = sp;
        while (r < n && is_ws(line[r])) r++;
        if (sp == 0 || r == sp || r >= n) {                               // split("\\s+", 2).length != 2
            err = "Invalid line in rank file: " + std::string(line, n);
            return JTK_ERR_BAD_RANK_FILE;
        }
        if (!b64decode(line, sp, tok)) { err = "Invalid base64 in rank file: " + std::string(line, n); return JTK_ERR_BAD_RANK_FILE; }
        uint64_t rank = 0;
        for (size_t k = r; k < n; k++) {
            if (line[k] < '0' || line[k] > '9' || rank > 0x7FFFFFFFull) { err = "Invalid rank in rank file: " + std::string(line, n); return JTK_ERR_BAD_RANK_FILE; }
            rank = rank * 10 + (uint64_t)(line[k] - '0');
        }
        if (rank > JTK_MAX_ID) { err = "rank exceeds the device table's id range"; return JTK_ERR_UNSUPPORTED_TABLE; }
        entries.emplace_back(tok, (uint32_t)rank);
    }
    if (entries.empty()) { err = "empty rank file"; return JTK_ERR_BAD_RANK_FILE; }

    t.max_id = 0;
    for (auto& en : entries) if (en.second > t.max_id) t.max_id = en.second;
    t.id_to_bytes.assign((size_t)t.max_id + 1, std::string());
    t.id_present.assign((size_t)t.max_id + 1, 0);
    t.bytes_to_id.clear();
    t.bytes_to_id.reserve(entries.size() * 2);
    for (auto& en : entries) {                                            // later lines win, as HashMap.put does
        t.bytes_to_id[en.first] = en.second;
        t.id_to_bytes[en.second] = en.first;
        t.id_present[en.second] = 1;
    }
    t.n_tokens = (int64_t)t.bytes_to_id.size();

    for (int b = 0; b < 256; b++) {
        auto it = t.bytes_to_id.find(std::string(1, (char)b));
        if (it == t.bytes_to_id.end()) { err = "rank table lacks a single-byte token; unsupported on the device path"; return JTK_ERR_UNSUPPORTED_TABLE; }
        t.byte_rank[b] = it->second;
    }

    // pair table: every split of every token
    std::vector<std::pair<uint64_t, uint32_t>> pairs;
    pairs.reserve(entries.size() * 3);
    for (auto& kv : t.bytes_to_id) {
        const std::string& T = kv.first;
        for (size_t k = 1; k < T.size(); k++) {
            auto a = t.bytes_to_id.find(T.substr(0, k));
            if (a == t.bytes_to_id.end()) continue;
            auto b = t.bytes_to_id.find(T.substr(k));
            if (b == t.bytes_to_id.end()) continue;
            pairs.emplace_back(jtk_pair_key(a->second, b->second), kv.second);
        }
    }
    t.n_pairs = (int64_t)pairs.size();
    // two-choice cuckoo, two slots per bucket; random-walk insertion.  Slot load ~0.72: cl100k's 233k
    // pairs take 2.6 MB, leaving room in a 4 MiB L2 for the streams that pass through it.
    {
        double load = 0.72;
        for (;; load *= 0.9) {
            const uint32_t nb = (uint32_t)((double)pairs.size() / (2.0 * load)) + 16;
            std::vector<uint64_t> slots((size_t)2 * nb, JTK_PAIR_EMPTY);
            uint32_t rng = 0x12345u;
            bool ok = true;
            for (auto& p : pairs) {
                uint64_t cur = (p.first << 30) | p.second;
                bool placed = false;
                for (int kick = 0; kick < 5000 && !placed; kick++) {
                    const uint64_t key = cur >> 30;
                    const uint32_t a = (uint32_t)(key >> JTK_ID_BITS), b = (uint32_t)(key & ((1u << JTK_ID_BITS) - 1));
                    const uint32_t bk[2] = {jtk_pair_hash(a, b, nb), jtk_pair_hash2(a, b, nb)};
                    for (int c = 0; c < 2 && !placed; c++)
                        for (int sidx = 0; sidx < 2 && !placed; sidx++)
                            if (slots[(size_t)bk[c] * 2 + sidx] == JTK_PAIR_EMPTY) { slots[(size_t)bk[c] * 2 + sidx] = cur; placed = true; }
                    if (!placed) {
                        rng = rng * 1664525u + 1013904223u;
                        const size_t victim = (size_t)bk[(rng >> 16) & 1] * 2 + ((rng >> 17) & 1);
                        std::swap(cur, slots[victim]);
                    }
                }
                if (!placed) { ok = false; break; }
            }
            if (!ok) continue;
            t.pair_bits = nb;
            t.pair_buckets.resize(nb);
            for (size_t k = 0; k < t.pair_buckets.size(); k++) {
                const uint64_t s0 = slots[2 * k], s1 = slots[2 * k + 1];
                t.pair_buckets[k] = JtkPairBucket{(uint32_t)s0, (uint32_t)(s0 >> 32), (uint32_t)s1, (uint32_t)(s1 >> 32)};
            }
            break;
        }
    }

    // whole-piece table (<= 8 bytes, two-choice cuckoo) and direct byte-pair table
    t.bp_rank.assign(65536, JTK_RANK_NONE);
    std::vector<JtkTok8Slot> shorts;
    for (auto& kv : t.bytes_to_id) {
        const std::string& T = kv.first;
        if (T.size() == 2) t.bp_rank[((uint32_t)(uint8_t)T[0] << 8) | (uint8_t)T[1]] = kv.second;
        if (T.size() > 8) continue;
        uint32_t lo = 0, hi = 0;
        for (size_t k = 0; k < T.size(); k++) {
            if (k < 4) lo |= (uint32_t)(uint8_t)T[k] << (8 * k); else hi |= (uint32_t)(uint8_t)T[k] << (8 * (k - 4));
        }
        shorts.push_back(JtkTok8Slot{lo, hi, kv.second, (uint32_t)T.size()});
    }
    t.n_tok8 = (int64_t)shorts.size();
    // Bytes b0 b1 that never sit next to each other inside any table entry can never end up in one part:
    // every merge result is a table entry.  bytePairMerge therefore never merges across such a position,
    // the sub-pieces on either side merge independently, and the split kernel may cut there.
    t.pair_in_token.assign(2048, 0);
    for (auto& kv : t.bytes_to_id) {
        const std::string& T = kv.first;
        for (size_t k = 0; k + 1 < T.size(); k++) {
            const uint32_t i = ((uint32_t)(uint8_t)T[k] << 8) | (uint8_t)T[k + 1];
            t.pair_in_token[i >> 5] |= 1u << (i & 31u);
        }
    }
    {
        t.bp_bits.assign(1024, 0);
        t.bp_cum.assign(1024, 0);
        t.bp_ranks.assign(JTK_BP_MAX, JTK_RANK_NONE);
        uint32_t k = 0;
        for (uint32_t i = 0; i < 65536; i++) {
            if ((i & 63u) == 0) t.bp_cum[i >> 6] = (uint16_t)k;
            if (t.bp_rank[i] != JTK_RANK_NONE) {
                if (k >= JTK_BP_MAX) { err = "too many 2-byte tokens for the LDS table"; return JTK_ERR_UNSUPPORTED_TABLE; }
                t.bp_bits[i >> 6] |= 1ull << (i & 63u);
                t.bp_ranks[k++] = t.bp_rank[i];
            }
        }
    }
    {
        double load = 0.45;
        for (;; load *= 0.9) {
            const uint32_t ns = (uint32_t)((double)shorts.size() / load) + 16;
            std::vector<JtkTok8Slot> slots(ns, JtkTok8Slot{0, 0, 0, 0});
            uint32_t rng = 0x9876u;
            bool ok = true;
            for (auto cur : shorts) {
                bool placed = false;
                for (int kick = 0; kick < 5000 && !placed; kick++) {
                    const uint32_t h[2] = {jtk_tok8_hash(cur.lo, cur.hi, cur.len, ns), jtk_tok8_hash2(cur.lo, cur.hi, cur.len, ns)};
                    for (int c = 0; c < 2 && !placed; c++)
                        if (slots[h[c]].len == 0) { slots[h[c]] = cur; placed = true; }
                    if (!placed) {
                        rng = rng * 1664525u + 1013904223u;
                        std::swap(cur, slots[h[(rng >> 16) & 1]]);
                    }
                }
                if (!placed) { ok = false; break; }
            }
            if (!ok) continue;
            t.tok8_bits = ns;
            t.tok8 = slots;
            break;
        }
    }

    {   // whole pieces of 9..16 bytes
        std::vector<JtkTok16Slot> mids;
        for (auto& kv : t.bytes_to_id) {
            const std::string& T = kv.first;
            if (T.size() < 9 || T.size() > 16) continue;
            JtkTok16Slot e{{0, 0, 0, 0}, kv.second, (uint32_t)T.size(), 0, 0};
            for (size_t k = 0; k < T.size(); k++) e.k[k >> 2] |= (uint32_t)(uint8_t)T[k] << (8 * (k & 3));
            mids.push_back(e);
        }
        t.n_tok16 = (int64_t)mids.size();
        double load = 0.45;
        for (;; load *= 0.9) {
            const uint32_t ns = (uint32_t)((double)mids.size() / load) + 16;
            std::vector<JtkTok16Slot> slots(ns, JtkTok16Slot{{0, 0, 0, 0}, 0, 0, 0, 0});
            uint32_t rng = 0x1357u;
            bool ok = true;
            for (auto cur : mids) {
                bool placed = false;
                for (int kick = 0; kick < 5000 && !placed; kick++) {
                    const uint32_t h[2] = {jtk_tok16_hash(cur.k[0], cur.k[1], cur.k[2], cur.k[3], cur.len, ns),
                                           jtk_tok16_hash2(cur.k[0], cur.k[1], cur.k[2], cur.k[3], cur.len, ns)};
                    for (int c = 0; c < 2 && !placed; c++)
                        if (slots[h[c]].len == 0) { slots[h[c]] = cur; placed = true; }
                    if (!placed) {
                        rng = rng * 1664525u + 1013904223u;
                        std::swap(cur, slots[h[(rng >> 16) & 1]]);
                    }
                }
                if (!placed) { ok = false; break; }
            }
            if (!ok) continue;
            t.tok16_n = ns;
            t.tok16 = slots;
            break;
        }
    }

    // The shortcut is applied only to pieces of <= 16 bytes on the device; everything else goes through
    // bytePairMerge.  That is only equivalent to GptBytePairEncoding.java:81-86 when merging any table
    // token on its own yields exactly that token.
    JtkPairTable pt{t.pair_buckets.data(), t.pair_bits};
    std::vector<uint32_t> ids, rk;
    for (auto& kv : t.bytes_to_id) {
        const std::string& T = kv.first;
        bool ok;
        if (T.size() <= 64) {
            ids.resize(T.size()); rk.resize(T.size());
            for (size_t k = 0; k < T.size(); k++) ids[k] = t.byte_rank[(uint8_t)T[k]];
            const int nt = jtk_merge_piece_lane(ids.data(), rk.data(), (int)T.size(), pt);
            ok = (nt == 1 && ids[0] == kv.second);
        } else {
            // generic O(n^2) form for the few long entries
            std::vector<uint32_t> v(T.size());
            for (size_t k = 0; k < T.size(); k++) v[k] = t.byte_rank[(uint8_t)T[k]];
            for (;;) {
                uint32_t best = JTK_RANK_NONE; size_t bi = 0;
                for (size_t k = 0; k + 1 < v.size(); k++) {
                    const uint32_t r = jtk_pair_lookup(pt, v[k], v[k + 1]);
                    if (r < best) { best = r; bi = k; }
                }
                if (best == JTK_RANK_NONE) break;
                v[bi] = best;
                v.erase(v.begin() + bi + 1);
            }
            ok = (v.size() == 1 && v[0] == kv.second);
        }
        if (!ok) {
            err = "rank table has an entry that bytePairMerge does not reproduce; the whole-piece shortcut "
                  "would change results (unsupported on the device path)";
            return JTK_ERR_UNSUPPORTED_TABLE;
        }
    }

    t.specials.clear();
    for (int k = 0; k < n_specials; k++) t.specials.emplace_back(std::string(special_literals[k]), special_ids[k]);
    return JTK_OK;
}

int64_t jtk_host_decode(const JtkHostTables& t, const int32_t* ids, int64_t n, uint8_t* out, int64_t cap) {
    int64_t w = 0;
    for (int64_t i = 0; i < n; i++) {
        const std::string* s = nullptr;
        const int32_t id = ids[i];
        if (id >= 0 && (uint32_t)id <= t.max_id && t.id_present[(size_t)id]) s = &t.id_to_bytes[(size_t)id];
        else for (auto& sp : t.specials) if (sp.second == id) { s = &sp.first; break; }   // GptBytePairEncoding.java:308-311
        if (!s) return JTK_ERR_UNKNOWN_TOKEN;
        if (out) {
            if (w + (int64_t)s->size() > cap) return JTK_ERR_CAPACITY;
            memcpy(out + w, s->data(), s->size());
        }
        w += (int64_t)s->size();
    }
    return w;
}
