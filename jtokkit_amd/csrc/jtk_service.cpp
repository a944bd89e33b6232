// jtk_service.cpp -- the per-call shape of the reference behind the batch path.
//
// The reference is called one document at a time from many threads (api/Encoding.java; its benchmark,
// benchmark/.../AbstractMultiThreadedBenchmark.java:35-45, is one task per document on a pool of 1..64 threads).  One GPU
// encode per call would pay copies, a dozen launches and a synchronisation per document.  A jtk_service coalesces instead:
// callers hand their document to a queue and block; worker threads (each with its own jtk_batch) take EVERYTHING that is
// queued at that moment -- no timer: while a batch is on the device the next one piles up -- gather it into pinned memory,
// run ONE batch encode with the result streamed back to pinned host memory, and hand every caller its tokens.
// jtk_service_submit / jtk_service_wait are the same without blocking in between, so that one thread can keep thousands of
// documents in flight (a Java shim's CompletableFuture).
#include <hip/hip_runtime.h>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/jtokkit_amd.h"

int jtk_fail_msg(int code, const std::string& msg);
struct jtk_encoding;
int64_t jtk_max_tokens_backoff(const jtk_encoding* enc, const uint8_t* utf8, int64_t len, const int32_t* head, int64_t nt,
                               int64_t max_tokens, int* truncated);

struct jtk_ticket {
    const uint8_t* utf8 = nullptr;
    int64_t len = 0;
    uint32_t flags = 0;
    int64_t max_tokens = -1;
    int32_t* tokens = nullptr;
    int64_t cap = 0;
    // result
    int64_t n_tokens = 0;
    int truncated = 0;
    int status = JTK_OK;
    // ONE word says where the ticket is: 0 pending, 1 done, 2 pending with its waiter asleep on this word (futex).  The worker
    // publishes with an exchange and never touches the ticket again: the waiter may delete it the moment it reads 1.
    std::atomic<int> state{0};
};

struct jtk_service {
    const jtk_encoding* enc = nullptr;
    int device = 0;
    std::atomic<int64_t> max_docs{1 << 16}, max_bytes{(int64_t)64 << 20};     // a device batch takes at most this much; the rest stays queued (jtk_service_set_limits)
    // The queue is sharded by producer thread: a producer holds its shard's lock for one push_back, so producers contend
    // with one another only when they share a shard (one mutex for all of them capped 8 producers at a fifth of what 2 reach).
    static constexpr int N_SHARDS = 16;
    struct alignas(64) Shard {
        std::mutex mu;
        std::vector<jtk_ticket*> queue;
    };
    Shard shards[N_SHARDS];
    std::atomic<unsigned> next_shard{0};
    unsigned id = 0;                  // (a thread remembers its shard per service)
    std::atomic<int> pending{0};      // documents queued; idle workers sleep on it (futex)
    std::atomic<int> idle{0};
    std::atomic<bool> stop{false};
    std::vector<std::thread> workers;
    std::atomic<int64_t> n_batches{0}, n_docs{0};
    // JTK_SERVICE_TRACE=1: where a worker's time goes (ns, summed over workers), printed when the service is destroyed
    bool trace = false;
    std::atomic<int64_t> ns_idle{0}, ns_take{0}, ns_gather{0}, ns_encode{0}, ns_hand{0}, ns_wake{0}, n_wakes{0};
};

namespace {

long futex_wait(std::atomic<int>* a, int expected) {
    return syscall(SYS_futex, reinterpret_cast<int*>(a), FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
}
long futex_wake(std::atomic<int>* a, int n) {
    return syscall(SYS_futex, reinterpret_cast<int*>(a), FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0);
}

void worker_main(jtk_service* s) {
    (void)hipSetDevice(s->device);
    jtk_batch* b = nullptr;
    uint8_t* h_text = nullptr;
    size_t h_text_cap = 0;
    int64_t* doc_off = nullptr;           // pinned too: the offsets go to the device by DMA like the text
    size_t doc_off_cap = 0;
    std::vector<jtk_ticket*> take, group;
    int create_rc = jtk_batch_create(s->enc, &b);
    auto now = [] { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (;;) {
        take.clear();
        const int64_t t_a = s->trace ? now() : 0;
        // everything that is queued right now (no timer: while this batch is on the device the next one piles up)
        while (s->pending.load(std::memory_order_acquire) == 0) {
            if (s->stop.load()) break;
            s->idle.fetch_add(1);
            if (s->pending.load() == 0 && !s->stop.load()) futex_wait(&s->pending, 0);
            s->idle.fetch_sub(1);
        }
        const int64_t t_b = s->trace ? now() : 0;
        {
            // (a document pushed after its shard was visited stays for the next round: `pending` is reduced by what was taken;
            // so does what exceeds max_docs / max_bytes: a device batch and its pinned staging are bounded)
            int taken = 0;
            int64_t bytes = 0;
            const int64_t max_docs = s->max_docs.load(), max_bytes = s->max_bytes.load();
            for (auto& sh : s->shards) {
                std::lock_guard<std::mutex> lk(sh.mu);
                size_t k = 0;
                while (k < sh.queue.size() && (int64_t)take.size() < max_docs && (take.empty() || bytes + sh.queue[k]->len <= max_bytes)) {
                    bytes += sh.queue[k]->len;
                    take.push_back(sh.queue[k++]);
                }
                taken += (int)k;
                sh.queue.erase(sh.queue.begin(), sh.queue.begin() + (long)k);
            }
            s->pending.fetch_sub(taken, std::memory_order_acq_rel);
        }
        if (take.empty()) {
            if (s->stop.load()) {
                bool left = false;                                    // (leave only when nothing is queued: tickets are never stranded)
                for (auto& sh : s->shards) { std::lock_guard<std::mutex> lk(sh.mu); left = left || !sh.queue.empty(); }
                if (!left) break;
            }
            continue;
        }
        const int64_t t_c = s->trace ? now() : 0;
        int64_t ns_g = 0, ns_e = 0;
        // encode() and encodeOrdinary() callers (and count-only ones) form separate device batches
        for (int pass = 0; pass < 4 && !take.empty(); pass++) {
            const uint32_t want = (pass & 1 ? JTK_ENCODE_ORDINARY : 0u) | (pass & 2 ? JTK_ENCODE_COUNT_ONLY : 0u);
            group.clear();
            for (jtk_ticket* t : take)
                if ((t->flags & (JTK_ENCODE_ORDINARY | JTK_ENCODE_COUNT_ONLY)) == want) group.push_back(t);
            if (group.empty()) continue;
            int rc = create_rc;
            if (rc == JTK_OK && group.size() + 1 > doc_off_cap) {
                if (doc_off) jtk_host_free(doc_off);
                doc_off = nullptr;
                doc_off_cap = group.size() * 2 + 1024;
                rc = jtk_host_alloc(doc_off_cap * 8, (void**)&doc_off);
                if (rc != JTK_OK) doc_off_cap = 0;
            }
            int64_t total = 0;
            if (rc == JTK_OK) {
                doc_off[0] = 0;
                for (size_t i = 0; i < group.size(); i++) { total += group[i]->len; doc_off[i + 1] = total; }
            }
            if (rc == JTK_OK && (size_t)total + 64 > h_text_cap) {
                if (h_text) jtk_host_free(h_text);
                h_text = nullptr;
                h_text_cap = (size_t)total * 2 + 4096;
                rc = jtk_host_alloc(h_text_cap, (void**)&h_text);
                if (rc != JTK_OK) h_text_cap = 0;
            }
            const int32_t* r_tok = nullptr;
            const int64_t* r_off = nullptr;
            const int32_t* r_st = nullptr;
            if (rc == JTK_OK) {
                const int64_t t0 = s->trace ? now() : 0;
                for (size_t i = 0; i < group.size(); i++)
                    if (group[i]->len) memcpy(h_text + doc_off[i], group[i]->utf8, (size_t)group[i]->len);
                const int64_t t1 = s->trace ? now() : 0;
                int64_t nt = 0;
                rc = jtk_batch_encode(b, h_text, doc_off, (int64_t)group.size(), want | JTK_ENCODE_TO_HOST, &nt);
                if (rc == JTK_OK) rc = jtk_batch_host_result(b, &r_tok, &r_off, &r_st);
                if (s->trace) { ns_g += t1 - t0; ns_e += now() - t1; }
            }
            for (size_t i = 0; i < group.size(); i++) {
                jtk_ticket* t = group[i];
                t->status = rc;
                if (rc != JTK_OK) continue;
                if (r_st[i] != JTK_OK) { t->status = r_st[i]; continue; }
                int64_t n = r_off[i + 1] - r_off[i];
                if (t->max_tokens >= 0 && !(want & JTK_ENCODE_COUNT_ONLY))
                    n = jtk_max_tokens_backoff(s->enc, t->utf8, t->len, r_tok + r_off[i], n, t->max_tokens, &t->truncated);
                t->n_tokens = n;
                if (t->tokens && !(want & JTK_ENCODE_COUNT_ONLY)) {
                    if (t->cap < n) t->status = JTK_ERR_CAPACITY;
                    else if (n > 0) memcpy(t->tokens, r_tok + r_off[i], (size_t)n * 4);
                }
            }
            s->n_batches++;
            s->n_docs += (int64_t)group.size();
        }
        const int64_t t_d = s->trace ? now() : 0;
        int wakes = 0;
        for (jtk_ticket* t : take) {
            std::atomic<int>* st = &t->state;
            if (st->exchange(1, std::memory_order_seq_cst) == 2) { futex_wake(st, 1); wakes++; }    // (only the address is used after the exchange)
        }
        if (s->trace) {
            const int64_t t_e = now();
            s->ns_idle += t_b - t_a; s->ns_take += t_c - t_b; s->ns_gather += ns_g; s->ns_encode += ns_e;
            s->ns_hand += (t_d - t_c) - ns_g - ns_e; s->ns_wake += t_e - t_d; s->n_wakes += wakes;
        }
    }
    if (h_text) jtk_host_free(h_text);
    if (doc_off) jtk_host_free(doc_off);
    if (b) jtk_batch_destroy(b);
}

}  // namespace

extern "C" {

int jtk_service_create(const jtk_encoding* enc, int n_workers, jtk_service** out) {
    if (!enc || !out) return jtk_fail_msg(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    *out = nullptr;
    if (n_workers <= 0) n_workers = 2;
    if (n_workers > 16) n_workers = 16;
    jtk_service* s = new (std::nothrow) jtk_service();
    if (!s) return jtk_fail_msg(JTK_ERR_OUT_OF_MEMORY, "out of host memory");
    static std::atomic<unsigned> next_id{1};
    s->id = next_id.fetch_add(1);
    s->enc = enc;
    s->trace = getenv("JTK_SERVICE_TRACE") != nullptr;
    s->device = jtk_encoding_device(enc);
    for (int i = 0; i < n_workers; i++) s->workers.emplace_back(worker_main, s);
    *out = s;
    return JTK_OK;
}

void jtk_service_destroy(jtk_service* s) {
    if (!s) return;
    {
        // no submit slips in behind the workers: `stop` is set with every shard locked, and submit checks it under its shard's lock
        for (auto& sh : s->shards) sh.mu.lock();
        s->stop.store(true);
        for (auto& sh : s->shards) sh.mu.unlock();
    }
    s->pending.fetch_add(1);                                     // wakes idle workers; they drain the queues and leave
    futex_wake(&s->pending, INT_MAX);
    for (auto& t : s->workers) t.join();
    if (s->trace && s->n_batches.load() > 0) {
        const double nb = (double)s->n_batches.load();
        fprintf(stderr, "[service] %lld batches, %.1f docs each; per batch (us): idle %.1f take %.1f gather %.1f encode %.1f hand-out %.1f wake %.1f (%.1f wakes)\n",
                (long long)s->n_batches.load(), (double)s->n_docs.load() / nb, s->ns_idle.load() / nb / 1e3, s->ns_take.load() / nb / 1e3,
                s->ns_gather.load() / nb / 1e3, s->ns_encode.load() / nb / 1e3, s->ns_hand.load() / nb / 1e3, s->ns_wake.load() / nb / 1e3,
                (double)s->n_wakes.load() / nb);
    }
    delete s;
}

int jtk_service_submit(jtk_service* s, const uint8_t* utf8, int64_t len, uint32_t flags, int64_t max_tokens,
                       int32_t* tokens, int64_t tokens_cap, jtk_ticket** ticket) {
    if (!s || !ticket || len < 0 || (len > 0 && !utf8)) return jtk_fail_msg(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    jtk_ticket* t = new (std::nothrow) jtk_ticket();
    if (!t) return jtk_fail_msg(JTK_ERR_OUT_OF_MEMORY, "out of host memory");
    t->utf8 = utf8; t->len = len; t->flags = flags & (JTK_ENCODE_ORDINARY | JTK_ENCODE_COUNT_ONLY); t->max_tokens = max_tokens;
    t->tokens = tokens; t->cap = tokens_cap;
    {
        // a thread keeps one shard per service (the last service it used is remembered)
        static thread_local unsigned my_service = 0, my_shard = 0;
        if (my_service != s->id) { my_service = s->id; my_shard = s->next_shard.fetch_add(1) % (unsigned)jtk_service::N_SHARDS; }
        jtk_service::Shard& sh = s->shards[my_shard];
        std::lock_guard<std::mutex> lk(sh.mu);
        if (s->stop.load()) { delete t; return jtk_fail_msg(JTK_ERR_INVALID_ARGUMENT, "service is shutting down"); }
        sh.queue.push_back(t);
        s->pending.fetch_add(1, std::memory_order_release);
    }
    if (s->idle.load(std::memory_order_seq_cst) > 0) futex_wake(&s->pending, 1);
    *ticket = t;
    return JTK_OK;
}

int jtk_service_wait(jtk_service* s, jtk_ticket* t, int64_t* n_tokens, int* truncated) {
    if (!s || !t) return jtk_fail_msg(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    for (int spin = 0; spin < 2000 && t->state.load(std::memory_order_acquire) != 1; spin++) __builtin_ia32_pause();
    if (t->state.load(std::memory_order_acquire) != 1) {
        int expect = 0;
        t->state.compare_exchange_strong(expect, 2, std::memory_order_seq_cst);      // 0 -> 2: asleep (or it just became 1)
        while (t->state.load(std::memory_order_seq_cst) != 1) futex_wait(&t->state, 2);
    }
    const int rc = t->status;
    if (n_tokens) *n_tokens = t->n_tokens;
    if (truncated) *truncated = t->truncated;
    delete t;
    if (rc == JTK_ERR_UNSUPPORTED_SPECIAL) return jtk_fail_msg(rc, "Encoding special tokens is not supported yet.");
    if (rc == JTK_ERR_CAPACITY) return jtk_fail_msg(rc, "tokens buffer too small");
    if (rc == JTK_ERR_UNENCODABLE) return jtk_fail_msg(rc, "Unknown token for encoding: the rank map lacks a single-byte token this text needs");
    if (rc != JTK_OK) return jtk_fail_msg(rc, "document could not be encoded");
    return JTK_OK;
}

int jtk_service_done(const jtk_ticket* t) { return t && t->state.load(std::memory_order_acquire) == 1 ? 1 : 0; }

int jtk_service_encode(jtk_service* s, const uint8_t* utf8, int64_t len, uint32_t flags, int64_t max_tokens,
                       int32_t* tokens, int64_t tokens_cap, int64_t* n_tokens, int* truncated) {
    if (n_tokens) *n_tokens = 0;
    if (truncated) *truncated = 0;
    if (s && !utf8 && len == 0) return JTK_OK;                   // text == null -> empty result (GptBytePairEncoding.java:48-50)
    jtk_ticket* t = nullptr;
    int rc = jtk_service_submit(s, utf8, len, flags, max_tokens, tokens, tokens_cap, &t);
    if (rc != JTK_OK) return rc;
    return jtk_service_wait(s, t, n_tokens, truncated);
}

int jtk_service_set_limits(jtk_service* s, int64_t max_docs, int64_t max_bytes) {
    if (!s) return jtk_fail_msg(JTK_ERR_INVALID_ARGUMENT, "service is NULL");
    if (max_docs >= 1) s->max_docs.store(max_docs);
    if (max_bytes >= 1) s->max_bytes.store(max_bytes);
    return JTK_OK;
}

int jtk_service_stats(jtk_service* s, int64_t* n_batches, int64_t* n_docs) {
    if (!s) return jtk_fail_msg(JTK_ERR_INVALID_ARGUMENT, "service is NULL");
    if (n_batches) *n_batches = s->n_batches.load();
    if (n_docs) *n_docs = s->n_docs.load();
    return JTK_OK;
}

}  // extern "C"
