// service_spin.cpp -- stress of the per-call service's completion hand-off (jtk_service.cpp): T native threads submit `total`
// documents with `window` in flight each and learn that a ticket is done ONLY by polling jtk_service_done (never sleeping in
// jtk_service_wait), then collect it with jtk_service_wait, which frees the ticket at once -- the interleaving in which a
// worker that touched the ticket after publishing it would touch freed memory.  Every result is compared with the CPU
// oracle's (test infrastructure: the oracle is the checker here, never the product).
// usage: service_spin <libjtokkit_amd.so> <libjtk_oracle.so> <tiktoken file> <threads> <total docs> <window>
#include <dlfcn.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

typedef struct jtk_encoding jtk_encoding;
typedef struct jtk_service jtk_service;
typedef struct jtk_ticket jtk_ticket;

static void* must(void* h, const char* name) {
    void* p = dlsym(h, name);
    if (!p) { fprintf(stderr, "missing symbol %s\n", name); exit(2); }
    return p;
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: see source\n"); return 2; }
    const int T = atoi(argv[4]);
    const long total = atol(argv[5]);
    const int window = atoi(argv[6]);
    void* L = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    void* O = dlopen(argv[2], RTLD_NOW);
    if (!L || !O) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    auto enc_create = (int (*)(const char*, int, const uint8_t*, size_t, const char* const*, const int32_t*, int, int, jtk_encoding**))must(L, "jtk_encoding_create");
    auto enc_destroy = (void (*)(jtk_encoding*))must(L, "jtk_encoding_destroy");
    auto svc_create = (int (*)(const jtk_encoding*, int, jtk_service**))must(L, "jtk_service_create");
    auto svc_destroy = (void (*)(jtk_service*))must(L, "jtk_service_destroy");
    auto svc_submit = (int (*)(jtk_service*, const uint8_t*, int64_t, uint32_t, int64_t, int32_t*, int64_t, jtk_ticket**))must(L, "jtk_service_submit");
    auto svc_wait = (int (*)(jtk_service*, jtk_ticket*, int64_t*, int*))must(L, "jtk_service_wait");
    auto svc_done = (int (*)(const jtk_ticket*))must(L, "jtk_service_done");
    auto o_create = (void* (*)(const char*, int, const uint8_t*, size_t, const char*, const int*, int))must(O, "jtko_create");
    auto o_encode = (long (*)(void*, const uint8_t*, size_t, int, long, int32_t*, size_t, int*))must(O, "jtko_encode");

    FILE* f = fopen(argv[3], "rb");
    if (!f) { perror(argv[3]); return 2; }
    std::vector<uint8_t> tik;
    for (uint8_t buf[65536];;) { const size_t n = fread(buf, 1, sizeof buf, f); if (!n) break; tik.insert(tik.end(), buf, buf + n); }
    fclose(f);
    const char* lits[5] = {"<|endoftext|>", "<|fim_prefix|>", "<|fim_middle|>", "<|fim_suffix|>", "<|endofprompt|>"};
    const int32_t ids[5] = {100257, 100258, 100259, 100260, 100276};
    jtk_encoding* enc = nullptr;
    if (enc_create("cl100k_base", 1, tik.data(), tik.size(), lits, ids, 5, 0, &enc) != 0) { fprintf(stderr, "encoding_create failed\n"); return 3; }
    std::string packed;                                                // the oracle takes the literals back to back, NUL-terminated
    for (const char* l : lits) { packed += l; packed.push_back('\0'); }
    const int oids[5] = {100257, 100258, 100259, 100260, 100276};
    void* ora = o_create("cl100k_base", 1, tik.data(), tik.size(), packed.data(), oids, 5);
    if (!ora) { fprintf(stderr, "oracle create failed\n"); return 3; }

    // documents: short sentences of words and numbers, some non-ASCII; expected tokens from the oracle
    const char* words[] = {"the", " quick", " brown", " fox", " jumps", " over", " lazy", " dog", " tokenization", " 12345", " caf\xc3\xa9",
                           " \xe4\xbd\xa0\xe5\xa5\xbd", "!!", "\n\n", " isn't", " Supercalifragilistic", " x", " \xf0\x9f\x8d\x95", " foo_bar(baz)", "  "};
    const int NW = (int)(sizeof words / sizeof *words), ND = 2048;
    std::vector<std::string> docs(ND);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto& d : docs) { const int k = 1 + (int)(rnd() % 40); for (int i = 0; i < k; i++) d += words[rnd() % NW]; }
    docs[0] = "";
    std::vector<std::vector<int32_t>> want(ND);
    for (int i = 0; i < ND; i++) {
        std::vector<int32_t> buf(docs[i].size() + 1);
        int tr = 0;
        const long n = o_encode(ora, (const uint8_t*)docs[i].data(), docs[i].size(), 1, -1, buf.data(), buf.size(), &tr);
        if (n < 0) { fprintf(stderr, "oracle failed on doc %d\n", i); return 3; }
        want[i].assign(buf.begin(), buf.begin() + n);
    }

    jtk_service* svc = nullptr;
    if (svc_create(enc, 2, &svc) != 0) { fprintf(stderr, "service_create failed\n"); return 3; }
    std::atomic<long> bad{0}, done{0};
    auto run = [&](int t) {
        struct Slot { jtk_ticket* tk = nullptr; int doc = 0; std::vector<int32_t> out; };
        std::vector<Slot> ring((size_t)window);
        long submitted = 0, collected = 0;
        const long mine = total / T + (t < total % T ? 1 : 0);
        uint64_t r = 0x9E3779B97F4A7C15ull * (uint64_t)(t + 1);
        auto collect = [&](Slot& sl) {
            while (!svc_done(sl.tk)) { /* spin: never sleep on the ticket */ }
            int64_t n = 0;
            int tr = 0;
            const int rc = svc_wait(svc, sl.tk, &n, &tr);
            if (rc != 0 || (size_t)n != want[sl.doc].size() || memcmp(sl.out.data(), want[sl.doc].data(), (size_t)n * 4) != 0) bad++;
            done++;
        };
        while (collected < mine) {
            if (submitted < mine && submitted - collected < window) {
                Slot& sl = ring[(size_t)(submitted % window)];
                r ^= r << 13; r ^= r >> 7; r ^= r << 17;
                sl.doc = (int)(r % ND);
                sl.out.assign(docs[sl.doc].size() + 1, -1);
                if (svc_submit(svc, (const uint8_t*)docs[sl.doc].data(), (int64_t)docs[sl.doc].size(), 1u, -1, sl.out.data(), (int64_t)sl.out.size(), &sl.tk) != 0) { bad++; return; }
                submitted++;
            } else {
                collect(ring[(size_t)(collected % window)]);
                collected++;
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(run, t);
    for (auto& x : th) x.join();
    svc_destroy(svc);
    enc_destroy(enc);
    printf("%s: %ld documents from %d threads, %ld wrong\n", bad.load() ? "spin FAILED" : "spin ok", done.load(), T, bad.load());
    return bad.load() ? 1 : 0;
}
