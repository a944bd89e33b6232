// percall_bench.cpp -- the reference's per-call shape (one Encoding.encode call per document from T threads,
// benchmark/.../AbstractMultiThreadedBenchmark.java:35-45) through the C ABI, native threads (no interpreter lock):
//   direct     jtk_encode, one jtk_batch per thread (the literal drop-in: one device round trip per document)
//   service    jtk_service_encode, T blocking callers coalesced into device batches
//   async      jtk_service_submit / jtk_service_wait, T threads keeping K documents in flight each
//   oracle     (bench.py's cpu_baseline leg) the CPU oracle's jtko_encode per call from T threads
// usage: percall_bench <libjtokkit_amd.so> <liboracle.so|-> <tiktoken file> <corpus.bin> <threads> <in_flight> <seconds>
// corpus.bin: int64 n_docs, int64 doc_off[n_docs + 1], bytes.  Prints one JSON object.
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

typedef struct jtk_encoding jtk_encoding;
typedef struct jtk_batch jtk_batch;
typedef struct jtk_service jtk_service;
typedef struct jtk_ticket jtk_ticket;

static void* must(void* h, const char* name) {
    void* p = dlsym(h, name);
    if (!p) { fprintf(stderr, "missing symbol %s\n", name); exit(2); }
    return p;
}

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage: see source\n"); return 2; }
    const char* libp = argv[1]; const char* orap = argv[2]; const char* tikp = argv[3]; const char* corp = argv[4];
    const int T = atoi(argv[5]), K = atoi(argv[6]);
    const double secs = atof(argv[7]);
    const int n_workers = argc > 8 ? atoi(argv[8]) : 2;                 // worker threads of the service (its default: 2)
    void* L = dlopen(libp, RTLD_NOW | RTLD_GLOBAL);
    if (!L) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    auto enc_create = (int (*)(const char*, int, const uint8_t*, size_t, const char* const*, const int32_t*, int, int, jtk_encoding**))must(L, "jtk_encoding_create");
    auto batch_create = (int (*)(const jtk_encoding*, jtk_batch**))must(L, "jtk_batch_create");
    auto batch_destroy = (void (*)(jtk_batch*))must(L, "jtk_batch_destroy");
    auto jencode = (int (*)(jtk_batch*, const uint8_t*, int64_t, uint32_t, int64_t, int32_t*, int64_t, int64_t*, int*))must(L, "jtk_encode");
    auto svc_create = (int (*)(const jtk_encoding*, int, jtk_service**))must(L, "jtk_service_create");
    auto svc_destroy = (void (*)(jtk_service*))must(L, "jtk_service_destroy");
    auto svc_encode = (int (*)(jtk_service*, const uint8_t*, int64_t, uint32_t, int64_t, int32_t*, int64_t, int64_t*, int*))must(L, "jtk_service_encode");
    auto svc_submit = (int (*)(jtk_service*, const uint8_t*, int64_t, uint32_t, int64_t, int32_t*, int64_t, jtk_ticket**))must(L, "jtk_service_submit");
    auto svc_wait = (int (*)(jtk_service*, jtk_ticket*, int64_t*, int*))must(L, "jtk_service_wait");
    auto svc_stats = (int (*)(jtk_service*, int64_t*, int64_t*))must(L, "jtk_service_stats");

    // corpus
    FILE* f = fopen(corp, "rb");
    if (!f) { perror(corp); return 2; }
    int64_t n_docs = 0;
    if (fread(&n_docs, 8, 1, f) != 1) return 2;
    std::vector<int64_t> off((size_t)n_docs + 1);
    if (fread(off.data(), 8, off.size(), f) != off.size()) return 2;
    std::vector<uint8_t> text((size_t)off[n_docs] + 16);
    if (off[n_docs] && fread(text.data(), 1, (size_t)off[n_docs], f) != (size_t)off[n_docs]) return 2;
    fclose(f);
    // rank table
    f = fopen(tikp, "rb");
    if (!f) { perror(tikp); return 2; }
    fseek(f, 0, SEEK_END); long tl = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> tik((size_t)tl);
    if (fread(tik.data(), 1, (size_t)tl, f) != (size_t)tl) return 2;
    fclose(f);
    const char* lits[] = {"<|endoftext|>", "<|fim_prefix|>", "<|fim_middle|>", "<|fim_suffix|>", "<|endofprompt|>"};
    const int32_t ids[] = {100257, 100258, 100259, 100260, 100276};
    jtk_encoding* enc = nullptr;
    if (enc_create("cl100k_base", 1, tik.data(), tik.size(), lits, ids, 5, 0, &enc) != 0) { fprintf(stderr, "encoding_create failed\n"); return 2; }
    int64_t max_len = 0;
    for (int64_t d = 0; d < n_docs; d++) if (off[d + 1] - off[d] > max_len) max_len = off[d + 1] - off[d];

    // every mode: threads take documents round robin (thread t: t, t + T, ...) for `secs` seconds; count docs, bytes, tokens
    struct Res { double docs_s, mb_s; int64_t tokens; double extra; };
    auto run = [&](auto&& body) {
        std::atomic<int64_t> docs{0}, bytes{0}, toks{0};
        std::atomic<bool> stop{false};
        std::vector<std::thread> th;
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < T; t++) th.emplace_back([&, t] { body(t, stop, docs, bytes, toks); });
        std::this_thread::sleep_for(std::chrono::duration<double>(secs));
        stop = true;
        for (auto& x : th) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return Res{docs / dt, bytes / dt / 1e6, toks.load(), 0.0};
    };

    // ---- direct: jtk_encode per document, one batch per thread
    Res direct = run([&](int t, std::atomic<bool>& stop, std::atomic<int64_t>& docs, std::atomic<int64_t>& bytes, std::atomic<int64_t>& toks) {
        jtk_batch* b = nullptr;
        if (batch_create(enc, &b) != 0) return;
        std::vector<int32_t> out((size_t)max_len + 1);
        int64_t nd = 0, nb = 0, nt = 0;
        for (int64_t d = t % n_docs; !stop; d = (d + T) % n_docs) {
            int64_t n = 0;
            if (jencode(b, text.data() + off[d], off[d + 1] - off[d], 0, -1, out.data(), (int64_t)out.size(), &n, nullptr) != 0) break;
            nd++; nb += off[d + 1] - off[d]; nt += n;
        }
        docs += nd; bytes += nb; toks += nt;
        batch_destroy(b);
    });

    // ---- service: blocking callers
    jtk_service* svc = nullptr;
    if (svc_create(enc, n_workers, &svc) != 0) { fprintf(stderr, "service_create failed\n"); return 2; }
    Res service = run([&](int t, std::atomic<bool>& stop, std::atomic<int64_t>& docs, std::atomic<int64_t>& bytes, std::atomic<int64_t>& toks) {
        std::vector<int32_t> out((size_t)max_len + 1);
        int64_t nd = 0, nb = 0, nt = 0;
        for (int64_t d = t % n_docs; !stop; d = (d + T) % n_docs) {
            int64_t n = 0;
            if (svc_encode(svc, text.data() + off[d], off[d + 1] - off[d], 0, -1, out.data(), (int64_t)out.size(), &n, nullptr) != 0) break;
            nd++; nb += off[d + 1] - off[d]; nt += n;
        }
        docs += nd; bytes += nb; toks += nt;
    });
    int64_t nb1 = 0, ndc1 = 0;
    svc_stats(svc, &nb1, &ndc1);
    service.extra = nb1 ? (double)ndc1 / (double)nb1 : 0.0;

    // ---- async: K documents in flight per thread
    Res async = run([&](int t, std::atomic<bool>& stop, std::atomic<int64_t>& docs, std::atomic<int64_t>& bytes, std::atomic<int64_t>& toks) {
        std::vector<std::vector<int32_t>> out((size_t)K, std::vector<int32_t>((size_t)max_len + 1));
        std::vector<jtk_ticket*> tk((size_t)K, nullptr);
        std::vector<int64_t> dd((size_t)K, 0);
        int64_t nd = 0, nb = 0, nt = 0;
        int64_t d = t % n_docs;
        for (int k = 0; k < K; k++) { dd[(size_t)k] = d; svc_submit(svc, text.data() + off[d], off[d + 1] - off[d], 0, -1, out[(size_t)k].data(), (int64_t)out[(size_t)k].size(), &tk[(size_t)k]); d = (d + T) % n_docs; }
        for (int k = 0;; k = (k + 1) % K) {
            int64_t n = 0;
            if (svc_wait(svc, tk[(size_t)k], &n, nullptr) != 0) break;
            const int64_t dk = dd[(size_t)k];
            nd++; nb += off[dk + 1] - off[dk]; nt += n;
            if (stop) { for (int j = (k + 1) % K; j != k; j = (j + 1) % K) svc_wait(svc, tk[(size_t)j], &n, nullptr); break; }
            dd[(size_t)k] = d;
            svc_submit(svc, text.data() + off[d], off[d + 1] - off[d], 0, -1, out[(size_t)k].data(), (int64_t)out[(size_t)k].size(), &tk[(size_t)k]);
            d = (d + T) % n_docs;
        }
        docs += nd; bytes += nb; toks += nt;
    });
    int64_t nb2 = 0, ndc2 = 0;
    svc_stats(svc, &nb2, &ndc2);
    async.extra = (nb2 - nb1) ? (double)(ndc2 - ndc1) / (double)(nb2 - nb1) : 0.0;
    svc_destroy(svc);

    // ---- oracle per call (CPU baseline leg)
    Res oracle{0, 0, 0, 0};
    if (strcmp(orap, "-") != 0) {
        void* O = dlopen(orap, RTLD_NOW);
        if (!O) { fprintf(stderr, "%s\n", dlerror()); return 2; }
        auto ocreate = (void* (*)(const char*, int, const char*, size_t, const char*, const int*, int))must(O, "jtko_create");
        auto oencode = (long (*)(void*, const uint8_t*, size_t, int, long, int32_t*, size_t, int*))must(O, "jtko_encode");
        const char sp[] = "<|endoftext|>\0<|fim_prefix|>\0<|fim_middle|>\0<|fim_suffix|>\0<|endofprompt|>\0";
        const int sid[] = {100257, 100258, 100259, 100260, 100276};
        void* oh = ocreate("cl100k_base", 1, (const char*)tik.data(), tik.size(), sp, sid, 5);
        oracle = run([&](int t, std::atomic<bool>& stop, std::atomic<int64_t>& docs, std::atomic<int64_t>& bytes, std::atomic<int64_t>& toks) {
            std::vector<int32_t> out((size_t)max_len + 1);
            int64_t nd = 0, nb = 0, nt = 0;
            for (int64_t d = t % n_docs; !stop; d = (d + T) % n_docs) {
                const long n = oencode(oh, text.data() + off[d], (size_t)(off[d + 1] - off[d]), 0, -1, out.data(), out.size(), nullptr);
                if (n < 0) break;
                nd++; nb += off[d + 1] - off[d]; nt += n;
            }
            docs += nd; bytes += nb; toks += nt;
        });
    }
    printf("{\"threads\": %d, \"service_workers\": %d, \"in_flight_per_thread\": %d, \"seconds\": %.1f, \"docs\": %lld, \"mean_bytes\": %.1f, "
           "\"direct\": {\"docs_per_s\": %.0f, \"MBps\": %.2f}, "
           "\"service_blocking\": {\"docs_per_s\": %.0f, \"MBps\": %.2f, \"docs_per_device_batch\": %.1f}, "
           "\"service_async\": {\"docs_per_s\": %.0f, \"MBps\": %.2f, \"docs_per_device_batch\": %.1f}, "
           "\"oracle_per_call\": {\"docs_per_s\": %.0f, \"MBps\": %.2f}}\n",
           T, n_workers, K, secs, (long long)n_docs, (double)off[n_docs] / (double)n_docs, direct.docs_s, direct.mb_s, service.docs_s, service.mb_s,
           service.extra, async.docs_s, async.mb_s, async.extra, oracle.docs_s, oracle.mb_s);
    return 0;
}
