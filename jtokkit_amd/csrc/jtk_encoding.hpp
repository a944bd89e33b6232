// jtk_encoding.hpp -- header-only C++ mirror of the reference's `Encoding` interface
// (reference lib/src/main/java/com/knuddels/jtokkit/api/Encoding.java:29-189) over the C ABI.
// The reference is compiled (JVM) code and no JDK exists in the build image, so this class is the
// host-side stand-in for the Java `HipEncoding implements Encoding` shown in INTEGRATION.md:
// same method names, argument meaning and error behaviour (status codes become exceptions).
// Like the Java shim, the per-call methods go through the encoding's jtk_service (thread-safe, concurrent callers
// coalesced into device batches); the batch methods use the object's own jtk_batch (one caller at a time).
// Compiled and run by the CPU test tier (tests/test_abi_and_host.py: no device -> IllegalStateException) and used by
// tools/percall for nothing: it is the C++ face of the boundary, kept honest by that test.
#ifndef JTK_ENCODING_HPP
#define JTK_ENCODING_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/jtokkit_amd.h"

namespace jtokkit {

struct UnsupportedOperationException : std::runtime_error { using std::runtime_error::runtime_error; };
struct IllegalArgumentException : std::invalid_argument { using std::invalid_argument::invalid_argument; };
struct IllegalStateException : std::runtime_error { using std::runtime_error::runtime_error; };

struct EncodingResult {                       // api/EncodingResult.java
    std::vector<int32_t> tokens;
    bool truncated;
    const std::vector<int32_t>& getTokens() const { return tokens; }
    bool isTruncated() const { return truncated; }
};

class Encoding {
public:
    Encoding(const std::string& name, int patternKind, const std::string& tiktokenBytes,
             const std::vector<std::pair<std::string, int32_t>>& specials, int device = 0) {
        std::vector<const char*> lits;
        std::vector<int32_t> ids;
        for (auto& s : specials) { lits.push_back(s.first.c_str()); ids.push_back(s.second); }
        check(jtk_encoding_create(name.c_str(), patternKind, (const uint8_t*)tiktokenBytes.data(), tiktokenBytes.size(),
                                  lits.data(), ids.data(), (int)lits.size(), device, &enc_));
        check(jtk_batch_create(enc_, &batch_));
        check(jtk_service_create(enc_, 2, &svc_));
    }
    ~Encoding() { jtk_service_destroy(svc_); jtk_batch_destroy(batch_); jtk_encoding_destroy(enc_); }
    Encoding(const Encoding&) = delete;
    Encoding& operator=(const Encoding&) = delete;

    std::vector<int32_t> encode(const std::string& text) { return run(text, 0, -1).tokens; }
    EncodingResult encode(const std::string& text, int maxTokens) { return run(text, 0, maxTokens); }
    std::vector<int32_t> encodeOrdinary(const std::string& text) { return run(text, JTK_ENCODE_ORDINARY, -1).tokens; }
    EncodingResult encodeOrdinary(const std::string& text, int maxTokens) { return run(text, JTK_ENCODE_ORDINARY, maxTokens); }
    int countTokens(const std::string& text) { return (int)encode(text).size(); }
    int countTokensOrdinary(const std::string& text) { return (int)encodeOrdinary(text).size(); }
    std::string decodeBytes(const std::vector<int32_t>& tokens) const {
        int64_t len = 0;
        check(jtk_decode(enc_, tokens.data(), (int64_t)tokens.size(), nullptr, 0, &len));
        std::string out((size_t)len, '\0');
        check(jtk_decode(enc_, tokens.data(), (int64_t)tokens.size(), (uint8_t*)&out[0], len, &len));
        return out;
    }
    std::string decode(const std::vector<int32_t>& tokens) const { return decodeBytes(tokens); }
    std::string getName() const { return jtk_encoding_name(enc_); }

    // batch: documents back to back in `utf8`, n+1 offsets -> packed ids + n+1 token offsets + per-doc status
    void encodeBatch(const uint8_t* utf8, const std::vector<int64_t>& docOff, bool ordinary,
                     std::vector<int32_t>& tokens, std::vector<int64_t>& tokOff, std::vector<int32_t>& status) {
        int64_t nt = 0;
        const int64_t n = (int64_t)docOff.size() - 1;
        check(jtk_batch_encode(batch_, utf8, docOff.data(), n, ordinary ? JTK_ENCODE_ORDINARY : 0u, &nt));
        tokens.resize((size_t)nt); tokOff.resize((size_t)n + 1); status.resize((size_t)n);
        check(jtk_batch_fetch(batch_, tokens.data(), nt, tokOff.data(), status.data()));
    }

    // custom split pattern: the caller's matches (byte ranges in the whole batch) instead of the device's split
    // (api/GptBytePairEncodingParams.java:36-46); text between matches is skipped as matcher.find() does
    void encodeBatchPieces(const uint8_t* utf8, const std::vector<int64_t>& docOff, const std::vector<int64_t>& pieceBegin,
                           const std::vector<int64_t>& pieceEnd, bool ordinary, std::vector<int32_t>& tokens,
                           std::vector<int64_t>& tokOff, std::vector<int32_t>& status) {
        int64_t nt = 0;
        const int64_t n = (int64_t)docOff.size() - 1;
        check(jtk_batch_encode_pieces(batch_, utf8, docOff.data(), n, pieceBegin.data(), pieceEnd.data(), (int64_t)pieceBegin.size(),
                                      ordinary ? JTK_ENCODE_ORDINARY : 0u, &nt));
        tokens.resize((size_t)nt); tokOff.resize((size_t)n + 1); status.resize((size_t)n);
        check(jtk_batch_fetch(batch_, tokens.data(), nt, tokOff.data(), status.data()));
    }

    // batch encode(text, maxTokens): after encodeBatch, how many of each document's tokens survive the limit (incl. the
    // reference's back-off to a code-point boundary) and EncodingResult.isTruncated() per document
    void truncateBatch(int64_t maxTokens, std::vector<int64_t>& kept, std::vector<uint8_t>& truncated, size_t nDocs) {
        check(jtk_batch_truncate(batch_, maxTokens));
        kept.resize(nDocs); truncated.resize(nDocs);
        check(jtk_batch_fetch_truncated(batch_, kept.data(), truncated.data()));
    }

    // batch decodeBytes: token lists back to back in `ids`, n+1 offsets -> bytes back to back + n+1 byte offsets + status
    void decodeBatch(const int32_t* ids, const std::vector<int64_t>& seqOff, std::string& bytes,
                     std::vector<int64_t>& byteOff, std::vector<int32_t>& status) {
        int64_t nb = 0;
        const int64_t n = (int64_t)seqOff.size() - 1;
        check(jtk_batch_decode(batch_, ids, seqOff.data(), n, &nb));
        bytes.resize((size_t)nb); byteOff.resize((size_t)n + 1); status.resize((size_t)n);
        check(jtk_batch_decode_fetch(batch_, (uint8_t*)&bytes[0], nb, byteOff.data(), status.data()));
    }

private:
    EncodingResult run(const std::string& text, uint32_t flags, int64_t maxTokens) {
        EncodingResult r{std::vector<int32_t>(text.size() + 1), false};
        int64_t n = 0; int tr = 0;
        check(jtk_service_encode(svc_, (const uint8_t*)text.data(), (int64_t)text.size(), flags, maxTokens, r.tokens.data(),
                                 (int64_t)r.tokens.size(), &n, &tr));
        r.tokens.resize((size_t)n);
        r.truncated = tr != 0;
        return r;
    }
    static void check(int rc) {
        if (rc == JTK_OK) return;
        const std::string msg = jtk_last_error();
        if (rc == JTK_ERR_UNSUPPORTED_SPECIAL) throw UnsupportedOperationException("Encoding special tokens is not supported yet.");
        if (rc == JTK_ERR_UNKNOWN_TOKEN || rc == JTK_ERR_INVALID_ARGUMENT) throw IllegalArgumentException(msg);
        throw IllegalStateException(msg);
    }
    jtk_encoding* enc_ = nullptr;
    jtk_batch* batch_ = nullptr;
    jtk_service* svc_ = nullptr;
};

}  // namespace jtokkit
#endif
