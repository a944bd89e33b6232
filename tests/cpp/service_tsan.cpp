// service_tsan.cpp -- the per-call service's host logic (jtokkit_amd/csrc/jtk_service.cpp) under ThreadSanitizer /
// AddressSanitizer on the CPU: the queue sharded by producer thread, the one-word ticket hand-off (a waiter may free its ticket
// the moment it reads "done"), max_docs / max_bytes per take, shutdown with tickets still queued.
// The device is stubbed out: this file defines the few C-ABI entry points the service calls (jtk_batch_*, jtk_host_*,
// hipSetDevice ...) so that "encoding" a batch is one token per input byte -- nothing of the product's encode path is tested
// here, only the service around it.  Built by tests/test_abi_and_host.py as
//   g++ -fsanitize=thread  tests/cpp/service_tsan.cpp jtokkit_amd/csrc/jtk_service.cpp
// usage: service_tsan <threads> <docs per thread> <window>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/jtokkit_amd.h"

// ---- stubs of what jtk_service.cpp links against --------------------------------------------------------------------
struct jtk_encoding { int device; };
struct jtk_batch {
    std::vector<int32_t> tok;
    std::vector<int64_t> off;
    std::vector<int32_t> st;
};
static std::atomic<int64_t> g_batches{0}, g_max_docs_seen{0};

int jtk_fail_msg(int code, const std::string&) { return code; }
int64_t jtk_max_tokens_backoff(const jtk_encoding*, const uint8_t*, int64_t, const int32_t*, int64_t nt, int64_t max_tokens, int* truncated) {
    if (truncated) *truncated = nt > max_tokens;
    return nt < max_tokens ? nt : max_tokens;
}
extern "C" {
int hipSetDevice(int) { return 0; }
int jtk_encoding_device(const jtk_encoding* e) { return e->device; }
int jtk_batch_create(const jtk_encoding*, jtk_batch** out) { *out = new jtk_batch(); return JTK_OK; }
void jtk_batch_destroy(jtk_batch* b) { delete b; }
int jtk_host_alloc(size_t bytes, void** out) { *out = malloc(bytes); return *out ? JTK_OK : JTK_ERR_OUT_OF_MEMORY; }
void jtk_host_free(void* p) { free(p); }
int jtk_batch_encode(jtk_batch* b, const uint8_t* utf8, const int64_t* doc_off, int64_t n_docs, uint32_t, int64_t* n_tokens) {
    // "encode": token i of a document = its byte i (+ 1000); a document starting with '!' has a special token in it
    b->tok.assign((size_t)doc_off[n_docs], 0);
    b->off.assign(doc_off, doc_off + n_docs + 1);
    b->st.assign((size_t)n_docs, JTK_OK);
    for (int64_t i = 0; i < doc_off[n_docs]; i++) b->tok[(size_t)i] = 1000 + utf8[i];
    for (int64_t d = 0; d < n_docs; d++)
        if (doc_off[d + 1] > doc_off[d] && utf8[doc_off[d]] == '!') b->st[(size_t)d] = JTK_ERR_UNSUPPORTED_SPECIAL;
    g_batches++;
    int64_t seen = g_max_docs_seen.load();
    while (n_docs > seen && !g_max_docs_seen.compare_exchange_weak(seen, n_docs)) {}
    std::this_thread::sleep_for(std::chrono::microseconds(30));          // a device round trip: the next batch piles up meanwhile
    if (n_tokens) *n_tokens = doc_off[n_docs];
    return JTK_OK;
}
int jtk_batch_host_result(jtk_batch* b, const int32_t** tokens, const int64_t** tok_off, const int32_t** status) {
    *tokens = b->tok.data(); *tok_off = b->off.data(); *status = b->st.data();
    return JTK_OK;
}
}  // extern "C"

// ---- the driver ---------------------------------------------------------------------------------------------------------
static std::string make_doc(unsigned seed) {
    std::string s;
    unsigned x = seed * 2654435761u + 12345u;
    const int len = (int)(x % 97u);
    for (int i = 0; i < len; i++) { x = x * 1664525u + 1013904223u; s.push_back((char)('a' + (x >> 24) % 26)); }
    if (seed % 53u == 7u && !s.empty()) s[0] = '!';
    return s;
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 8;
    const int per = argc > 2 ? atoi(argv[2]) : 4000;
    const int window = argc > 3 ? atoi(argv[3]) : 32;
    jtk_encoding enc{0};
    std::atomic<long> bad{0}, done{0};
    for (int round = 0; round < 3; round++) {
        jtk_service* svc = nullptr;
        if (jtk_service_create(&enc, 2, &svc) != JTK_OK) return 3;
        if (round == 1) { jtk_service_set_limits(svc, 7, 300); g_max_docs_seen.store(0); }     // tight limits: many small batches
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) {
            th.emplace_back([&, t] {
                struct Slot { jtk_ticket* tk = nullptr; std::string doc; std::vector<int32_t> out; };
                std::vector<Slot> ring((size_t)window);
                auto collect = [&](Slot& sl) {
                    int64_t n = 0; int tr = 0;
                    const int rc = jtk_service_wait(svc, sl.tk, &n, &tr);
                    sl.tk = nullptr;
                    const bool special = !sl.doc.empty() && sl.doc[0] == '!';
                    bool ok = special ? rc == JTK_ERR_UNSUPPORTED_SPECIAL : (rc == JTK_OK && n == (int64_t)sl.doc.size());
                    for (int64_t i = 0; ok && !special && i < n; i++) ok = sl.out[(size_t)i] == 1000 + (uint8_t)sl.doc[(size_t)i];
                    if (!ok) bad++;
                    done++;
                };
                for (int i = 0; i < per; i++) {
                    Slot& sl = ring[(size_t)(i % window)];
                    if (sl.tk) {
                        if ((t + round) % 3 == 0) while (!jtk_service_done(sl.tk)) std::this_thread::yield();   // poll, then collect at once
                        collect(sl);
                    }
                    sl.doc = make_doc((unsigned)(t * 1000003 + i + round * 77));
                    sl.out.assign(sl.doc.size() + 1, -1);
                    if ((t + round) % 3 == 1 && i % 5 == 0) {
                        // the blocking form in between
                        int64_t n = 0; int tr = 0;
                        const int rc = jtk_service_encode(svc, (const uint8_t*)sl.doc.data(), (int64_t)sl.doc.size(), 0, -1, sl.out.data(), (int64_t)sl.out.size(), &n, &tr);
                        const bool special = !sl.doc.empty() && sl.doc[0] == '!';
                        if (special ? rc != JTK_ERR_UNSUPPORTED_SPECIAL : (rc != JTK_OK || n != (int64_t)sl.doc.size())) bad++;
                        done++;
                        continue;
                    }
                    if (jtk_service_submit(svc, (const uint8_t*)sl.doc.data(), (int64_t)sl.doc.size(), 0, -1, sl.out.data(), (int64_t)sl.out.size(), &sl.tk) != JTK_OK) { bad++; sl.tk = nullptr; }
                }
                for (Slot& sl : ring) if (sl.tk) collect(sl);
            });
        }
        for (auto& x : th) x.join();
        // shutdown with work queued: tickets submitted and not yet waited for when jtk_service_destroy is called are still
        // served (the workers leave only when the queues are empty) and can be collected afterwards
        struct Late { jtk_ticket* tk; std::string doc; std::vector<int32_t> out; };
        std::vector<std::vector<Late>> late((size_t)4);
        std::vector<std::thread> lt;
        for (int t = 0; t < 4; t++) {
            lt.emplace_back([&, t] {
                auto& mine = late[(size_t)t];
                mine.resize(1500);
                for (size_t i = 0; i < mine.size(); i++) {
                    mine[i].doc = make_doc((unsigned)(900000 + t * 5000 + i)) + "x";
                    mine[i].doc[0] = 'x';
                    mine[i].out.assign(mine[i].doc.size() + 1, -1);
                    if (jtk_service_submit(svc, (const uint8_t*)mine[i].doc.data(), (int64_t)mine[i].doc.size(), 0, 5, mine[i].out.data(),
                                           (int64_t)mine[i].out.size(), &mine[i].tk) != JTK_OK) { bad++; mine[i].tk = nullptr; }
                }
            });
        }
        for (auto& x : lt) x.join();
        jtk_service_destroy(svc);
        if (round == 1 && g_max_docs_seen.load() > 7) { fprintf(stderr, "a batch of %lld documents with max_docs = 7\n", (long long)g_max_docs_seen.load()); bad++; }
        for (auto& mine : late)
            for (auto& l : mine) {
                if (!l.tk) continue;
                int64_t n = 0; int tr = -1;
                const int64_t want = (int64_t)l.doc.size() < 5 ? (int64_t)l.doc.size() : 5;
                if (!jtk_service_done(l.tk) || jtk_service_wait(svc /* only compared with NULL */, l.tk, &n, &tr) != JTK_OK || n != want ||
                    tr != ((int64_t)l.doc.size() > 5)) bad++;
            }
    }
    const long expect = 3L * T * per;
    if (bad.load() || done.load() != expect) { fprintf(stderr, "FAILED: %ld bad, %ld of %ld done\n", bad.load(), done.load(), expect); return 1; }
    printf("service ok: %ld documents, %lld batches, largest %lld documents\n", done.load(), (long long)g_batches.load(), (long long)g_max_docs_seen.load());
    return 0;
}
