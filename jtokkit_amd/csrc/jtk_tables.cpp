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
#include <algorithm>
#include <cstring>

#include "../../include/jtokkit_amd.h"
#include "jtk_merge_core.h"

// Slot load of the hash tables.  A probe that misses in a slot which turned a key away costs a second, dependent fetch, and
// what a lookup costs on the device is fetches, not bytes: HBM has 288 GB and the tables do not fit the 4 MB L2 of an XCD at any
// sensible load anyway (the Infinity Cache serves them), so they are built SPARSE -- measured on a 1 GiB chunk of mixed text
// (r03, profiles/r03_experiments/r03b[ijk]_table_loads.txt): loads 0.4 / 0.5 (whole-piece / pair table; 8.5 MB in all): resolve
// 2.53 ms, merge 2.43; 0.2 / 0.2: 2.41 / 2.18; 0.12 / 0.10 (35 MB): 2.42 / 2.06; sparser than that changes nothing.
#ifndef JTK_TOK_TABLE_LOAD
#define JTK_TOK_TABLE_LOAD 0.12
#endif
#ifndef JTK_PAIR_TABLE_LOAD
#define JTK_PAIR_TABLE_LOAD 0.10
#endif

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

// Primary-first two-choice placement (jtk_common.h): item i may live in bucket h1[i] (primary) or h2[i]; a bucket holds
// `cap` items.  Items arrive in the order given -- the callers pass them by ascending rank, i.e. most frequent first, so
// the frequent entries keep their primary buckets -- and stay in the primary bucket if it has room.  The others go to
// their secondary bucket; when that is full too, a resident of either bucket that can move to a free place does so,
// else residents are kicked along a walk (the later, rarer item of a bucket is the one that moves).
// where[i] = the bucket of item i.
static bool place_primary_first(const std::vector<uint32_t>& h1, const std::vector<uint32_t>& h2, uint32_t nb, int cap,
                                std::vector<uint32_t>& where) {
    const size_t n = h1.size();
    where.assign(n, 0xFFFFFFFFu);
    std::vector<std::vector<uint32_t>> res(nb);
    std::vector<uint32_t> pending;
    for (size_t i = 0; i < n; i++) {
        if ((int)res[h1[i]].size() < cap) { res[h1[i]].push_back((uint32_t)i); where[i] = h1[i]; }
        else pending.push_back((uint32_t)i);
    }
    auto other = [&](uint32_t item, uint32_t bucket) { return h1[item] == bucket ? h2[item] : h1[item]; };
    auto move_out = [&](uint32_t bucket, size_t k, uint32_t to) {     // resident k of `bucket` moves to `to` (which has room)
        const uint32_t v = res[bucket][k];
        res[bucket][k] = res[bucket].back(); res[bucket].pop_back();
        res[to].push_back(v); where[v] = to;
    };
    for (uint32_t start : pending) {
        uint32_t cur = start;
        bool placed = false;
        uint32_t target = h2[cur];
        for (int kick = 0; kick < 20000 && !placed; kick++) {
            if ((int)res[target].size() < cap) { res[target].push_back(cur); where[cur] = target; placed = true; break; }
            // one-step moves: a resident of either of cur's buckets whose other bucket has room
            const uint32_t cand[2] = {h1[cur], h2[cur]};
            for (int c = 0; c < 2 && !placed; c++) {
                auto& r = res[cand[c]];
                for (size_t k = 0; k < r.size() && !placed; k++) {
                    const uint32_t to = other(r[k], cand[c]);
                    if (to != cand[c] && (int)res[to].size() < cap) {
                        move_out(cand[c], k, to);
                        res[cand[c]].push_back(cur); where[cur] = cand[c];
                        placed = true;
                    }
                }
            }
            if (placed) break;
            // kick the latest (rarest) resident of the target bucket and continue with it
            size_t v = 0;
            for (size_t k = 1; k < res[target].size(); k++) if (res[target][k] > res[target][v]) v = k;
            const uint32_t victim = res[target][v];
            res[target][v] = cur; where[cur] = target;
            cur = victim;
            target = other(cur, target);
        }
        if (!placed) return false;
    }
    // items sitting in their secondary bucket although their primary has room again go home
    for (size_t i = 0; i < n; i++) {
        if (where[i] != h1[i] && (int)res[h1[i]].size() < cap) {
            auto& r = res[where[i]];
            for (size_t k = 0; k < r.size(); k++) if (r[k] == (uint32_t)i) { move_out(where[i], k, h1[i]); break; }
        }
    }
    return true;
}

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

    t.n_missing = 0;
    t.pseudo_base = 0;
    for (int b = 0; b < 256; b++) {
        auto it = t.bytes_to_id.find(std::string(1, (char)b));
        if (it != t.bytes_to_id.end()) { t.byte_rank[b] = it->second; continue; }
        if (!t.n_missing) t.pseudo_base = t.max_id + 1;
        t.byte_rank[b] = t.pseudo_base + (uint32_t)t.n_missing++;
    }
    if (t.n_missing && t.pseudo_base + (uint32_t)t.n_missing - 1 > JTK_MAX_ID) {
        err = "rank table lacks single-byte tokens and leaves no room for their pseudo ids in the device table's id range";
        return JTK_ERR_UNSUPPORTED_TABLE;
    }
    // a part of a token: the token's id, or the pseudo id of a single byte that is no token
    auto part_id = [&](const std::string& p, uint32_t& id) {
        auto it = t.bytes_to_id.find(p);
        if (it != t.bytes_to_id.end()) { id = it->second; return true; }
        if (p.size() == 1 && t.n_missing) { id = t.byte_rank[(uint8_t)p[0]]; return true; }
        return false;
    };

    // pair table: every split of every token
    std::vector<std::pair<uint64_t, uint32_t>> pairs;
    pairs.reserve(entries.size() * 3);
    for (auto& kv : t.bytes_to_id) {
        const std::string& T = kv.first;
        for (size_t k = 1; k < T.size(); k++) {
            uint32_t a, b;
            if (!part_id(T.substr(0, k), a) || !part_id(T.substr(k), b)) continue;
            pairs.emplace_back(jtk_pair_key(a, b), kv.second);
        }
    }
    t.n_pairs = (int64_t)pairs.size();
    std::sort(pairs.begin(), pairs.end(), [](const std::pair<uint64_t, uint32_t>& x, const std::pair<uint64_t, uint32_t>& y) {
        return x.second != y.second ? x.second < y.second : x.first < y.first; });      // by rank: frequent merges first
    // two-choice cuckoo, two slots per bucket, primary first; sparse (JTK_PAIR_TABLE_LOAD above): hardly any bucket turns a key away
    {
        double load = JTK_PAIR_TABLE_LOAD;
        for (;; load *= 0.9) {
            const uint32_t nb = (uint32_t)((double)pairs.size() / (2.0 * load)) + 16;
            std::vector<uint32_t> h1(pairs.size()), h2(pairs.size()), where;
            for (size_t i = 0; i < pairs.size(); i++) {
                const uint32_t a = (uint32_t)(pairs[i].first >> JTK_ID_BITS), b = (uint32_t)(pairs[i].first & ((1u << JTK_ID_BITS) - 1));
                h1[i] = jtk_pair_hash(a, b, nb);
                h2[i] = jtk_pair_hash2(a, b, nb);
            }
            if (!place_primary_first(h1, h2, nb, 2, where)) continue;
            t.pair_bits = nb;
            t.pair_buckets.assign(nb, JtkPairBucket{0xFFFFFFFFu, 0xDFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu});
            t.pair_displaced = 0;
            for (size_t i = 0; i < pairs.size(); i++) {
                JtkPairBucket& bk = t.pair_buckets[where[i]];
                const uint32_t kw = (uint32_t)pairs[i].first, rw = pairs[i].second | ((uint32_t)(pairs[i].first >> 32) << 30);
                if (bk.k0 == 0xFFFFFFFFu && (bk.r0 | JTK_PAIR_OVERFLOW) == 0xFFFFFFFFu) { bk.k0 = kw; bk.r0 = rw | (bk.r0 & JTK_PAIR_OVERFLOW); }
                else { bk.k1 = kw; bk.r1 = rw; }
            }
            for (size_t i = 0; i < pairs.size(); i++)
                if (where[i] != h1[i]) { t.pair_buckets[h1[i]].r0 |= JTK_PAIR_OVERFLOW; t.pair_displaced++; }
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
    std::sort(shorts.begin(), shorts.end(), [](const JtkTok8Slot& x, const JtkTok8Slot& y) { return x.id < y.id; });   // frequent first
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
        double load = JTK_TOK_TABLE_LOAD;
        for (;; load *= 0.9) {
            const uint32_t ns = (uint32_t)((double)shorts.size() / load) + 16;
            std::vector<uint32_t> h1(shorts.size()), h2(shorts.size()), where;
            for (size_t i = 0; i < shorts.size(); i++) {
                h1[i] = jtk_tok8_hash(shorts[i].lo, shorts[i].hi, shorts[i].len, ns);
                h2[i] = jtk_tok8_hash2(shorts[i].lo, shorts[i].hi, shorts[i].len, ns);
            }
            if (!place_primary_first(h1, h2, ns, 1, where)) continue;
            t.tok8_bits = ns;
            t.tok8.assign(ns, JtkTok8Slot{0, 0, 0, 0});
            t.tok8_displaced = 0;
            for (size_t i = 0; i < shorts.size(); i++) t.tok8[where[i]] = shorts[i];
            for (size_t i = 0; i < shorts.size(); i++)
                if (where[i] != h1[i]) {
                    t.tok8[h1[i]].len |= JTK_TOK_OVERFLOW | JTK_TOK_FILTER_BIT(jtk_tok16_mix(shorts[i].lo, shorts[i].hi, 0u, 0u, shorts[i].len));
                    t.tok8_displaced++;
                }
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
        std::sort(mids.begin(), mids.end(), [](const JtkTok16Slot& x, const JtkTok16Slot& y) { return x.id < y.id; });
        double load = JTK_TOK_TABLE_LOAD;
        for (;; load *= 0.9) {
            const uint32_t ns = (uint32_t)((double)mids.size() / load) + 16;
            std::vector<uint32_t> h1(mids.size()), h2(mids.size()), where;
            for (size_t i = 0; i < mids.size(); i++) {
                h1[i] = jtk_tok16_hash(mids[i].k[0], mids[i].k[1], mids[i].k[2], mids[i].k[3], mids[i].len, ns);
                h2[i] = jtk_tok16_hash2(mids[i].k[0], mids[i].k[1], mids[i].k[2], mids[i].k[3], mids[i].len, ns);
            }
            if (!place_primary_first(h1, h2, ns, 1, where)) continue;
            t.tok16_n = ns;
            t.tok16.assign(ns, JtkTok16Slot{{0, 0, 0, 0}, 0, 0, 0, 0});
            for (size_t i = 0; i < mids.size(); i++) t.tok16[where[i]] = mids[i];
            for (size_t i = 0; i < mids.size(); i++)
                if (where[i] != h1[i])
                    t.tok16[h1[i]].len |= JTK_TOK_OVERFLOW | JTK_TOK_FILTER_BIT(jtk_tok16_mix(mids[i].k[0], mids[i].k[1], mids[i].k[2], mids[i].k[3], mids[i].len));
            break;
        }
    }

    // The shortcut is applied only to pieces of <= 16 bytes on the device; everything else goes through
    // bytePairMerge.  That is only equivalent to GptBytePairEncoding.java:81-86 when merging any table
    // token on its own yields exactly that token.
    JtkPairTable pt{t.pair_buckets.data(), t.pair_bits};
    std::vector<uint32_t> ids, rk;
    std::vector<const std::pair<const std::string, uint32_t>*> unrep_long;
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
            t.n_unreproducible++;
            if (T.size() > 16) unrep_long.push_back(&kv);
        }
    }
    if (t.n_unreproducible) {
        // The intra-piece cuts of pretok_split assume that encoding the two sides of a cut separately equals merging the
        // whole piece; with an entry that merging does not reproduce a side could hit the whole-piece lookup where the
        // reference (looking up the whole regex piece only) merges.  No cuts for such tables.
        t.pair_in_token.assign(2048, 0xFFFFFFFFu);
    }
    if (!unrep_long.empty()) {
        uint32_t ns = 16;
        while (ns < unrep_long.size() * 2 + 16) ns <<= 1;
        t.long_tok.assign(ns, JtkLongTokSlot{0, 0, 0, 0, 0, 0, 0, 0});
        for (auto* kv : unrep_long) {
            const std::string& T = kv->first;
            uint64_t h = JTK_FNV_BASIS;
            for (unsigned char c : T) h = jtk_fnv1a_step(h, c);
            uint32_t i = (uint32_t)(h % ns);
            while (t.long_tok[i].len) i = (i + 1) % ns;
            t.long_tok[i] = JtkLongTokSlot{(uint32_t)h, (uint32_t)(h >> 32), kv->second, (uint32_t)T.size(), (uint32_t)t.long_blob.size(), 0, 0, 0};
            t.long_blob += T;
            if (T.size() > t.long_max_len) t.long_max_len = (uint32_t)T.size();
        }
        t.long_blob.append(16, '\0');
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
