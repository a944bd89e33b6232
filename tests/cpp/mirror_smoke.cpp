// TEST INFRASTRUCTURE: compiles the header-only C++ mirror of the reference's Encoding interface
// (jtokkit_amd/csrc/jtk_encoding.hpp) and drives it.  usage: mirror_smoke <cl100k_base.tiktoken>
// exit 0 = encoded "hello world" to [15339, 1917] on a device; exit 3 = no HIP device (the expected outcome in the CPU
// tier: the mirror must turn JTK_ERR_NO_DEVICE into IllegalStateException, there is no CPU path); anything else = failure.
#include <cstdio>
#include <fstream>
#include <sstream>

#include "../../jtokkit_amd/csrc/jtk_encoding.hpp"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    try {
        jtokkit::Encoding enc("cl100k_base", JTK_PATTERN_CL100K, ss.str(), {{"<|endoftext|>", 100257}}, 0);
        const std::vector<int32_t> ids = enc.encode("hello world");
        if (ids != std::vector<int32_t>{15339, 1917}) return 4;
        if (enc.decode(ids) != "hello world") return 5;
        const jtokkit::EncodingResult r = enc.encode("This is a sample sentence.", 3);
        if (r.getTokens() != std::vector<int32_t>{2028, 374, 264} || !r.isTruncated()) return 6;
        try { enc.encode("a <|endoftext|> b"); return 7; } catch (const jtokkit::UnsupportedOperationException&) {}
        std::printf("mirror ok\n");
        return 0;
    } catch (const jtokkit::IllegalStateException& e) {
        std::printf("IllegalStateException: %s\n", e.what());
        return 3;
    }
}
