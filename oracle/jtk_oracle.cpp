// jtk_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference's encode hot path, written to be read side by side with
//   /root/reference/lib/src/main/java/com/knuddels/jtokkit/GptBytePairEncoding.java
//   /root/reference/lib/src/main/java/com/knuddels/jtokkit/TokenEncoder.java
//   /root/reference/lib/src/main/java/com/knuddels/jtokkit/ImmutableByteArray.java
//   /root/reference/lib/src/main/java/com/knuddels/jtokkit/EncodingFactory.java
// Same algorithmic class as the reference on purpose: hash map keyed by byte strings, O(n^2)
// leftmost-min merge on a vector with erase(), sequential backtracking matcher for the two split
// patterns.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the
// product (jtokkit_amd/) never links, loads or calls it.
//
// Pinning: checked against the reference's own golden vectors (tests/golden/*_encodings.csv,
// 4 encodings x 423 rows: full encode, encode(.,10) and the truncated flag) and the javadoc/README
// literals by tests/test_oracle_golden.py.  Branches the reference's fixtures do not reach (CR/LF,
// digit runs > 3, upper-case contractions, ...) are additionally cross-checked against the Python
// `regex` engine -- those vectors are "provisional, not JVM-verified" (no JVM exists in the build
// container).
//
// The reference hands the JDK a UTF-16 String; this restatement takes the bytes of
// String.getBytes(UTF_8) (ImmutableByteArray.java:16-19) and matches on code points, which is
// what java.util.regex does for supplementary characters.  Input must be well-formed UTF-8.

#include <atomic>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "uc_ranges.h"

namespace {

enum { ERR_UNSUPPORTED_SPECIAL = -2, ERR_UNKNOWN_TOKEN = -3, ERR_CAPACITY = -4, ERR_BAD_TABLE = -5,
       ERR_BAD_UTF8 = -6 };

// ---- character classes (EncodingFactory.java:129: Pattern.UNICODE_CHARACTER_CLASS) -------------
bool in_ranges(const uint32_t (*r)[2], int n, uint32_t cp) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        if (cp < r[mid][0]) hi = mid - 1;
        else if (cp > r[mid][1]) lo = mid + 1;
        else return true;
    }
    return false;
}
bool isL(uint32_t cp) { return in_ranges(uc_L_ranges, UC_L_NRANGES, cp); }   // \p{L}
bool isN(uint32_t cp) { return in_ranges(uc_N_ranges, UC_N_NRANGES, cp); }   // \p{N}
bool isS(uint32_t cp) { return in_ranges(uc_W_ranges, UC_W_NRANGES, cp); }   // \s
bool isCRLF(uint32_t cp) { return cp == '\r' || cp == '\n'; }                 // [\r\n]
bool isOther(uint32_t cp) { return !isS(cp) && !isL(cp) && !isN(cp); }        // [^\s\p{L}\p{N}]

// java.util.regex case-insensitive + UNICODE_CASE single-char test: pattern char `lower` matches c
// when lower == c or lower == toLowerCase(toUpperCase(c)).  For the letters the contraction
// alternative uses (s t r e v m l d) the only non-ASCII code point that folds onto them is
// U+017F LATIN SMALL LETTER LONG S -> 'S' -> 's'.
uint32_t fold_ci(uint32_t c) {
    if (c >= 'A' && c <= 'Z') return c + 32;
    if (c == 0x17F) return 's';
    return c;
}

struct Text {
    std::vector<uint32_t> cp;    // code points
    std::vector<uint32_t> off;   // byte offset of each code point; off[n] = byte length
    size_t n() const { return cp.size(); }
};

// strict UTF-8 decode; returns false on malformed input
bool decode_utf8(const uint8_t* s, size_t len, Text& t) {
    t.cp.clear(); t.off.clear();
    size_t i = 0;
    while (i < len) {
        uint32_t b0 = s[i], c; int k;
        if (b0 < 0x80) { c = b0; k = 1; }
        else if (b0 >= 0xC2 && b0 <= 0xDF) { c = b0 & 0x1F; k = 2; }
        else if (b0 >= 0xE0 && b0 <= 0xEF) { c = b0 & 0x0F; k = 3; }
        else if (b0 >= 0xF0 && b0 <= 0xF4) { c = b0 & 0x07; k = 4; }
        else return false;
        if (i + k > len) return false;
        for (int j = 1; j < k; j++) {
            uint32_t b = s[i + j];
            if ((b & 0xC0) != 0x80) return false;
            c = (c << 6) | (b & 0x3F);
        }
        if (k == 3 && (c < 0x800 || (c >= 0xD800 && c <= 0xDFFF))) return false;
        if (k == 4 && (c < 0x10000 || c > 0x10FFFF)) return false;
        t.cp.push_back(c); t.off.push_back((uint32_t)i);
        i += k;
    }
    t.off.push_back((uint32_t)len);
    return true;
}

// ---- the two split patterns as literal backtracking matchers ------------------------------------
// Each alt_* returns the end index (in code points) of the match starting at i, or -1.
// Alternatives are tried in pattern order (leftmost-first), quantifiers are greedy and give back
// one character at a time, exactly as java.util.regex does.

// 's|'t|'re|'ve|'m|'ll|'d   (ci = wrapped in (?i: ) for cl100k, EncodingFactory.java:105)
long alt_contraction(const Text& t, size_t i, bool ci) {
    size_t n = t.n();
    if (t.cp[i] != '\'') return -1;
    auto at = [&](size_t k, uint32_t want) {
        if (k >= n) return false;
        uint32_t c = ci ? fold_ci(t.cp[k]) : t.cp[k];
        return c == want;
    };
    if (at(i + 1, 's')) return (long)i + 2;
    if (at(i + 1, 't')) return (long)i + 2;
    if (at(i + 1, 'r') && at(i + 2, 'e')) return (long)i + 3;
    if (at(i + 1, 'v') && at(i + 2, 'e')) return (long)i + 3;
    if (at(i + 1, 'm')) return (long)i + 2;
    if (at(i + 1, 'l') && at(i + 2, 'l')) return (long)i + 3;
    if (at(i + 1, 'd')) return (long)i + 2;
    return -1;
}

template <class Pred> size_t run_end(const Text& t, size_t i, Pred p) {
    while (i < t.n() && p(t.cp[i])) i++;
    return i;
}

// cl100k alt 2: [^\r\n\p{L}\p{N}]?\p{L}+
long cl_alt_letters(const Text& t, size_t i) {
    uint32_t c = t.cp[i];
    if (!isCRLF(c) && !isL(c) && !isN(c)) {           // greedy '?' takes the char first
        size_t e = run_end(t, i + 1, isL);
        if (e > i + 1) return (long)e;
    }
    size_t e = run_end(t, i, isL);                     // backtrack: '?' matches empty
    return e > i ? (long)e : -1;
}
// cl100k alt 3: \p{N}{1,3}
long cl_alt_numbers(const Text& t, size_t i) {
    size_t e = i;
    while (e < t.n() && e < i + 3 && isN(t.cp[e])) e++;
    return e > i ? (long)e : -1;
}
// cl100k alt 4:  ?[^\s\p{L}\p{N}]+[\r\n]*
long cl_alt_other(const Text& t, size_t i) {
    if (t.cp[i] == ' ') {
        size_t e = run_end(t, i + 1, isOther);
        if (e > i + 1) return (long)run_end(t, e, isCRLF);
    }
    size_t e = run_end(t, i, isOther);
    if (e > i) return (long)run_end(t, e, isCRLF);
    return -1;
}
// cl100k alt 5: \s*[\r\n]+
long cl_alt_ws_newline(const Text& t, size_t i) {
    size_t kmax = run_end(t, i, isS);
    for (size_t k = kmax + 1; k-- > i;) {              // \s* gives back one char at a time
        size_t e = run_end(t, k, isCRLF);
        if (e > k) return (long)e;
    }
    return -1;
}
// alt: \s+(?!\S)
long alt_ws_not_before_nonws(const Text& t, size_t i) {
    size_t kmax = run_end(t, i, isS);
    for (size_t k = kmax; k > i; k--) {
        if (k == t.n() || isS(t.cp[k])) return (long)k;
    }
    return -1;
}
// alt: \s+
long alt_ws(const Text& t, size_t i) {
    size_t e = run_end(t, i, isS);
    return e > i ? (long)e : -1;
}
// r50k alts 2-4:  ?X+
template <class Pred> long r_alt_sp_run(const Text& t, size_t i, Pred p) {
    if (t.cp[i] == ' ') {
        size_t e = run_end(t, i + 1, p);
        if (e > i + 1) return (long)e;
    }
    size_t e = run_end(t, i, p);
    return e > i ? (long)e : -1;
}

enum PatternKind { PAT_R50K = 0, PAT_CL100K = 1 };

long match_at(const Text& t, size_t i, int kind) {
    long e;
    if (kind == PAT_CL100K) {                                   // EncodingFactory.java:105
        if ((e = alt_contraction(t, i, true)) >= 0) return e;
        if ((e = cl_alt_letters(t, i)) >= 0) return e;
        if ((e = cl_alt_numbers(t, i)) >= 0) return e;
        if ((e = cl_alt_other(t, i)) >= 0) return e;
        if ((e = cl_alt_ws_newline(t, i)) >= 0) return e;
        if ((e = alt_ws_not_before_nonws(t, i)) >= 0) return e;
        return alt_ws(t, i);
    }
    if ((e = alt_contraction(t, i, false)) >= 0) return e;      // EncodingFactory.java:63,77,91
    if ((e = r_alt_sp_run(t, i, isL)) >= 0) return e;
    if ((e = r_alt_sp_run(t, i, isN)) >= 0) return e;
    if ((e = r_alt_sp_run(t, i, isOther)) >= 0) return e;
    if ((e = alt_ws_not_before_nonws(t, i)) >= 0) return e;
    return alt_ws(t, i);
}

// ---- TokenEncoder (TokenEncoder.java:16-17) -------------------------------------------------------
struct Oracle {
    std::string name;
    int kind;
    std::unordered_map<std::string, int> decodedToEncoded;     // ImmutableByteArray -> rank
    std::unordered_map<int, std::string> encodedToDecoded;
    std::vector<std::pair<std::string, int>> specials;         // specialTokensEncoder
    std::string error;
};

int b64val(int c) {
    if (c >= 'A' && c <= 'Z') return c - 'A';
    if (c >= 'a' && c <= 'z') return c - 'a' + 26;
    if (c >= '0' && c <= '9') return c - '0' + 52;
    if (c == '+') return 62;
    if (c == '/') return 63;
    return -1;
}
bool b64decode(const std::string& s, std::string& out) {
    out.clear();
    size_t n = s.size();
    if (n % 4 != 0) return false;
    for (size_t i = 0; i < n; i += 4) {
        int v[4]; int pad = 0;
        for (int j = 0; j < 4; j++) {
            char c = s[i + j];
            if (c == '=') { if (i + 4 != n || j < 2) return false; v[j] = 0; pad++; }
            else { if (pad) return false; v[j] = b64val(c); if (v[j] < 0) return false; }
        }
        uint32_t w = (v[0] << 18) | (v[1] << 12) | (v[2] << 6) | v[3];
        out.push_back((char)(w >> 16));
        if (pad < 2) out.push_back((char)(w >> 8));
        if (pad < 1) out.push_back((char)w);
    }
    return true;
}

// EncodingFactory.java:139-164 loadMergeableRanks: per line split("\\s+", 2), Base64, parseInt
bool load_ranks(Oracle& o, const uint8_t* data, size_t len) {
    size_t i = 0;
    while (i < len) {
        size_t e = i;
        while (e < len && data[e] != '\n') e++;
        std::string line((const char*)data + i, e - i);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        i = e + 1;
        size_t sp = 0;
        while (sp < line.size() && !(line[sp] == ' ' || (line[sp] >= 9 && line[sp] <= 13))) sp++;
        size_t r = sp;
        while (r < line.size() && (line[r] == ' ' || (line[r] >= 9 && line[r] <= 13))) r++;
        if (sp == line.size() || sp == r) { o.error = "Invalid line: " + line; return false; }
        std::string tok;
        if (!b64decode(line.substr(0, sp), tok)) { o.error = "Invalid base64: " + line; return false; }
        char* endp = nullptr;
        long rank = strtol(line.c_str() + r, &endp, 10);
        if (endp == line.c_str() + r || *endp != 0) { o.error = "Invalid rank: " + line; return false; }
        o.decodedToEncoded[tok] = (int)rank;
        o.encodedToDecoded[(int)rank] = tok;
    }
    return true;
}

// ---- bytePairMerge (GptBytePairEncoding.java:200-275) and getRank (:285-300) ---------------------
struct PieceIndexToRank { int index; int rank; };              // :316-324

bool getRank(const Oracle& o, const std::string& piece, const std::vector<PieceIndexToRank>& parts,
             int startIndex, int skip, int& out) {
    if (startIndex + skip + 2 >= (int)parts.size()) return false;                        // :291
    int pieceStartIndex = parts[startIndex].index;                                         // :295
    int pieceEndIndex = parts[startIndex + skip + 2].index;                                // :296
    std::string encoderIndex = piece.substr(pieceStartIndex, pieceEndIndex - pieceStartIndex);  // slice copy
    auto it = o.decodedToEncoded.find(encoderIndex);                                       // :299
    if (it == o.decodedToEncoded.end()) return false;
    out = it->second;
    return true;
}

long bytePairMerge(const Oracle& o, const std::string& piece, std::vector<int>& out) {
    std::vector<PieceIndexToRank> parts;
    for (int i = 0; i < (int)piece.size() + 1; i++) parts.push_back({i, INT_MAX});        // :206-209
    for (int i = 0; i < (int)parts.size() - 2; i++) {                                      // :216-221
        int rank;
        if (getRank(o, piece, parts, i, 0, rank)) parts[i].rank = rank;
    }
    while (parts.size() > 1) {                                                             // :223
        int minRankIndex = 0, minRank = INT_MAX;
        for (int i = 0; i < (int)parts.size() - 1; i++) {                                  // :234-240
            int rank = parts[i].rank;
            if (rank < minRank) { minRank = rank; minRankIndex = i; }                      // strict <
        }
        if (minRank != INT_MAX) {                                                          // :247
            int r;
            parts[minRankIndex].rank = getRank(o, piece, parts, minRankIndex, 1, r) ? r : INT_MAX;      // :254
            if (minRankIndex > 0)                                                          // :255-257
                parts[minRankIndex - 1].rank = getRank(o, piece, parts, minRankIndex - 1, 1, r) ? r : INT_MAX;
            parts.erase(parts.begin() + minRankIndex + 1);                                 // :259
        } else {
            break;                                                                         // :261
        }
    }
    for (int i = 0; i < (int)parts.size() - 1; i++) {                                      // :271-273
        auto it = o.decodedToEncoded.find(piece.substr(parts[i].index, parts[i + 1].index - parts[i].index));
        if (it == o.decodedToEncoded.end()) return ERR_UNKNOWN_TOKEN;                      // TokenEncoder.java:66-68
        out.push_back(it->second);
    }
    return 0;
}

// new String(bytes, UTF_8).length(): UTF-16 length with one U+FFFD per maximal ill-formed subpart.
// Also reports whether `bytes` (a byte prefix of well-formed text) decodes to a String the text
// starts with -- see encodeOrdinaryInternal below.
size_t utf16_len_wellformed(const uint8_t* s, size_t len) {
    size_t n = 0;
    for (size_t i = 0; i < len; i++) {
        uint8_t b = s[i];
        if ((b & 0xC0) != 0x80) n += (b >= 0xF0) ? 2 : 1;
    }
    return n;
}

// GptBytePairEncoding.java:71-103
long encodeOrdinaryInternal(const Oracle& o, const uint8_t* utf8, size_t len, long maxTokens,
                            std::vector<int>& out, int* truncated) {
    if (truncated) *truncated = 0;
    out.clear();
    Text t;
    if (!decode_utf8(utf8, len, t)) return ERR_BAD_UTF8;
    size_t i = 0, n = t.n();
    long tokenCount = 0;
    while (i < n) {                                                                        // matcher.find()
        if (maxTokens >= 0 && maxTokens <= tokenCount) break;                              // :79, :277-283
        long e = match_at(t, i, o.kind);
        if (e < 0) { i++; continue; }                                                      // find() skips unmatched
        std::string match((const char*)utf8 + t.off[i], t.off[e] - t.off[i]);              // :80
        auto it = o.decodedToEncoded.find(match);
        if (it != o.decodedToEncoded.end()) {                                              // :81-83
            out.push_back(it->second);
            tokenCount++;
        } else {                                                                           // :85-86
            std::vector<int> tokensToAdd;
            long rc = bytePairMerge(o, match, tokensToAdd);
            if (rc < 0) return rc;
            size_t take = tokensToAdd.size();
            if (maxTokens >= 0) {                                                          // addTokens :110-119
                long room = maxTokens - (long)out.size();
                if ((long)take > room) take = (size_t)room;
            }
            out.insert(out.end(), tokensToAdd.begin(), tokensToAdd.begin() + take);
            tokenCount += (long)take;
        }
        i = (size_t)e;
    }
    if (maxTokens >= 0) {                                                                  // :90-100
        // decode(tokens) is a byte prefix of the text (token bytes concatenate to the pieces).
        // text.startsWith(decoded) holds when the prefix ends on a code-point boundary, or when the
        // cut character decodes to one U+FFFD and the text itself has U+FFFD (EF BF BD) there.
        std::vector<size_t> cum(out.size() + 1, 0);
        for (size_t k = 0; k < out.size(); k++) cum[k + 1] = cum[k] + o.encodedToDecoded.at(out[k]).size();
        size_t textLen16 = utf16_len_wellformed(utf8, len);
        for (size_t tokensToRemove = 0; tokensToRemove <= out.size(); tokensToRemove++) {
            size_t keep = out.size() - tokensToRemove;
            size_t nb = cum[keep];
            size_t b = nb;                                   // back up to the enclosing code-point start
            while (b > 0 && b < len && (utf8[b] & 0xC0) == 0x80) b--;
            bool boundary = (nb == len) || ((utf8[nb] & 0xC0) != 0x80);
            size_t decodedLen16;
            bool startsWith;
            if (boundary) {
                decodedLen16 = utf16_len_wellformed(utf8, nb);
                startsWith = true;
            } else {
                decodedLen16 = utf16_len_wellformed(utf8, b) + 1;   // one U+FFFD for the truncated tail
                startsWith = (b + 2 < len) && utf8[b] == 0xEF && utf8[b + 1] == 0xBF && utf8[b + 2] == 0xBD;
            }
            if (startsWith) {
                out.resize(keep);
                if (truncated) *truncated = textLen16 > decodedLen16;                      // :97
                return (long)out.size();
            }
        }
    }
    return (long)out.size();                                                               // :102
}

bool contains(const uint8_t* hay, size_t n, const std::string& needle) {
    if (needle.empty()) return true;
    if (needle.size() > n) return false;
    for (size_t i = 0; i + needle.size() <= n; i++)
        if (hay[i] == (uint8_t)needle[0] && memcmp(hay + i, needle.data(), needle.size()) == 0) return true;
    return false;
}

}  // namespace

extern "C" {

// pattern_kind: 0 = r50k/p50k/p50k_edit pattern, 1 = cl100k pattern.
// specials: `n_specials` NUL-terminated literals back to back, with their ids.
void* jtko_create(const char* name, int pattern_kind, const uint8_t* tiktoken, size_t len,
                  const char* specials, const int* special_ids, int n_specials) {
    Oracle* o = new Oracle();
    o->name = name ? name : "";
    o->kind = pattern_kind;
    if (!load_ranks(*o, tiktoken, len)) { delete o; return nullptr; }
    const char* p = specials;
    for (int i = 0; i < n_specials; i++) {
        std::string s(p);
        o->specials.push_back({s, special_ids[i]});
        p += s.size() + 1;
    }
    return o;
}
void jtko_destroy(void* h) { delete (Oracle*)h; }
long jtko_vocab_size(void* h) { return (long)((Oracle*)h)->decodedToEncoded.size(); }

// ordinary = 0: encode() semantics (special-token literal -> ERR_UNSUPPORTED_SPECIAL,
// GptBytePairEncoding.java:52-56); ordinary = 1: encodeOrdinary().  max_tokens < 0: none.
// Returns the token count, or a negative error.  utf8 == NULL mirrors text == null -> empty.
long jtko_encode(void* h, const uint8_t* utf8, size_t len, int ordinary, long max_tokens,
                 int32_t* out, size_t cap, int* truncated) {
    Oracle* o = (Oracle*)h;
    if (truncated) *truncated = 0;
    if (!utf8) return 0;                                                                   // :48-50
    if (!ordinary)
        for (auto& s : o->specials)
            if (contains(utf8, len, s.first)) return ERR_UNSUPPORTED_SPECIAL;
    std::vector<int> toks;
    long rc = encodeOrdinaryInternal(*o, utf8, len, max_tokens, toks, truncated);
    if (rc < 0) return rc;
    if (out) {
        if (toks.size() > cap) return ERR_CAPACITY;
        for (size_t i = 0; i < toks.size(); i++) out[i] = toks[i];
    }
    return (long)toks.size();
}

// piece end offsets (bytes) of the pre-token split; returns the number of pieces
long jtko_split(void* h, const uint8_t* utf8, size_t len, int64_t* ends, size_t cap) {
    Oracle* o = (Oracle*)h;
    Text t;
    if (!decode_utf8(utf8, len, t)) return ERR_BAD_UTF8;
    size_t i = 0, n = t.n(), k = 0;
    while (i < n) {
        long e = match_at(t, i, o->kind);
        if (e < 0) { i++; continue; }
        if (ends) { if (k >= cap) return ERR_CAPACITY; ends[k] = t.off[e]; }
        k++;
        i = (size_t)e;
    }
    return (long)k;
}

// encodeOrdinaryInternal (GptBytePairEncoding.java:77-87) with the matches of a caller-supplied pattern: match i is
// utf8[begin[i], end[i]); text between matches is skipped as matcher.find() does.  Whole-piece lookup first (:81-83),
// else bytePairMerge (:85-86).  Returns the token count.
long jtko_encode_pieces(void* h, const uint8_t* utf8, const int64_t* begin, const int64_t* end, long n_pieces, int32_t* out, size_t cap) {
    Oracle* o = (Oracle*)h;
    size_t k = 0;
    for (long i = 0; i < n_pieces; i++) {
        std::string match((const char*)utf8 + begin[i], (size_t)(end[i] - begin[i]));      // :80
        auto it = o->decodedToEncoded.find(match);
        if (it != o->decodedToEncoded.end()) {                                              // :81-83
            if (k >= cap) return ERR_CAPACITY;
            out[k++] = it->second;
        } else {                                                                            // :85-86
            std::vector<int> toks;
            long rc = bytePairMerge(*o, match, toks);
            if (rc < 0) return rc;
            if (k + toks.size() > cap) return ERR_CAPACITY;
            for (int t : toks) out[k++] = t;
        }
    }
    return (long)k;
}

// bytePairMerge of one piece (no whole-piece shortcut): GptBytePairEncoding.java:200-275
long jtko_merge_piece(void* h, const uint8_t* piece, size_t len, int32_t* out, size_t cap) {
    Oracle* o = (Oracle*)h;
    std::vector<int> toks;
    long rc = bytePairMerge(*o, std::string((const char*)piece, len), toks);
    if (rc < 0) return rc;
    if (toks.size() > cap) return ERR_CAPACITY;
    for (size_t i = 0; i < toks.size(); i++) out[i] = toks[i];
    return (long)toks.size();
}

// decodeBytes (GptBytePairEncoding.java:137-151, 302-314)
long jtko_decode(void* h, const int32_t* ids, size_t n, uint8_t* out, size_t cap) {
    Oracle* o = (Oracle*)h;
    size_t w = 0;
    for (size_t i = 0; i < n; i++) {
        const std::string* s = nullptr;
        auto it = o->encodedToDecoded.find(ids[i]);
        if (it != o->encodedToDecoded.end()) s = &it->second;
        else for (auto& sp : o->specials) if (sp.second == ids[i]) s = &sp.first;
        if (!s) return ERR_UNKNOWN_TOKEN;                                                  // :313
        if (out) { if (w + s->size() > cap) return ERR_CAPACITY; memcpy(out + w, s->data(), s->size()); }
        w += s->size();
    }
    return (long)w;
}

// The reference's benchmark method (AbstractMultiThreadedBenchmark.java:35-45): every document
// once, one task per document on a fixed pool of `threads`.  Writes per-doc token counts when
// `counts` is non-NULL and the packed tokens when `tokens`/`tok_off` are non-NULL (tok_off must
// already hold the exclusive prefix of counts).  Returns total tokens or a negative error.
long jtko_encode_batch(void* h, const uint8_t* text, const int64_t* doc_off, long n_docs, int ordinary,
                       int threads, int32_t* counts, int32_t* tokens, const int64_t* tok_off) {
    Oracle* o = (Oracle*)h;
    if (threads < 1) threads = 1;
    std::atomic<long> next(0), total(0), err(0);
    auto work = [&]() {
        std::vector<int> toks;
        long local = 0;
        for (;;) {
            long d = next.fetch_add(1);
            if (d >= n_docs) break;
            const uint8_t* p = text + doc_off[d];
            size_t len = (size_t)(doc_off[d + 1] - doc_off[d]);
            long rc = 0;
            if (!ordinary)
                for (auto& s : o->specials)
                    if (contains(p, len, s.first)) rc = ERR_UNSUPPORTED_SPECIAL;
            if (rc == 0) rc = encodeOrdinaryInternal(*o, p, len, -1, toks, nullptr);
            if (rc < 0) { err.store(rc); if (counts) counts[d] = 0; continue; }
            if (counts) counts[d] = (int32_t)rc;
            if (tokens && tok_off) memcpy(tokens + tok_off[d], toks.data(), toks.size() * sizeof(int));
            local += rc;
        }
        total.fetch_add(local);
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < threads; i++) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    long e = err.load();
    return e < 0 ? e : total.load();
}

const char* jtko_unicode_version(void) { return UC_UNICODE_VERSION; }

}  // extern "C"
