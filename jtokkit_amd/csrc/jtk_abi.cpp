// jtk_abi.cpp -- the C ABI of include/jtokkit_amd.h: encoding objects, batch scratch, kernel
// orchestration.  Host C++ only; all compute is in jtk_kernels.hip.  There is no CPU fallback:
// without a HIP device every encode entry point fails with JTK_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>
#include <thread>
#include <chrono>

#include "../../include/jtokkit_amd.h"
#include "jtk_kernels.h"
#include "jtk_tables.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? JTK_ERR_OUT_OF_MEMORY : JTK_ERR_HIP,           \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return JTK_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(JTK_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e)); }
        cap = want;
        return JTK_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

constexpr int N_STAGES = 8;
const char* const STAGE_NAMES[N_STAGES] = {"mark_docs", "validate_utf8", "pretok_split", "piece_resolve", "bpe_merge",
                                           "tile_scan", "pack", "doc_offsets"};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

int jtk_fail_msg(int code, const std::string& msg) { return fail(code, msg); }      // for jtk_comm.cpp

struct jtk_encoding {
    JtkHostTables host;
    int device = 0;
    DevBuf uc1, uc2, brank, pairs, tok8, tok16, bprank, bpbits, bpcum, bpranks, pairin, dec_off, dec_blob, longtok, longblob, specials;
    uint32_t n_ids_table = 0;        // ids 0 .. n_ids_table-1 have an entry in the decode table (incl. special tokens)
    JtkDeviceTables dt;
    std::vector<uint32_t> tok_len;   // byte length per id (0 = absent), for the maxTokens back-off
};

int64_t jtk_max_tokens_backoff(const jtk_encoding* enc, const uint8_t* utf8, int64_t len, const int32_t* head, int64_t nt,
                               int64_t max_tokens, int* truncated);

// One chunk in flight: a stream and the scratch of the kernels (sized for the largest chunk it has seen).
struct ChunkSet {
    hipStream_t stream = nullptr;
    hipEvent_t ev_scan = nullptr;    // this set's tile_scan has run (the next chunk's scan waits for it: token order)
    hipEvent_t ev_done = nullptr;    // this set's last kernel has run
    DevBuf zeroed;                   // docmask | list counters | queue counters
    DevBuf piecemask, gapmask, plist, htok, docpre, tile_np, tile_off, queues, q_meta, mid_list, long_list, giant_list;
    JtkWork work{};
    bool used = false;               // by the current job
};

constexpr int MAX_SETS = 4;
constexpr int64_t SMALL_JOB_BYTES = 1 << 20;
constexpr int64_t TINY_JOB_BYTES = 128 << 10;     // host jobs this small are staged so that offsets and text go down in one copy

struct jtk_batch {
    const jtk_encoding* enc = nullptr;
    hipStream_t stream = nullptr;        // the batch's own main stream
    hipStream_t last_stream = nullptr;   // main stream of the last job (the caller's, or `stream`)
    hipStream_t copy_stream = nullptr;   // tokens of finished chunks to pinned host memory (JTK_ENCODE_TO_HOST)
    hipEvent_t ev_fork = nullptr, ev_copy = nullptr;
    int64_t* h_info = nullptr;           // pinned, device-visible: per chunk, its first token and the end of its last
    size_t h_info_cap = 0;
    ChunkSet set[MAX_SETS];
    int n_sets = 2;                      // chunks in flight (JTK_OPT_CHUNKS_IN_FLIGHT)
    bool n_sets_chosen = false;          // by the caller or the environment (else a job that streams ids to the host takes 3)
    int64_t chunk_bytes = (int64_t)1 << 30;    // JTK_OPT_CHUNK_BYTES: device-resident input (large chunks: fewer launches and kernel tails)
    int64_t host_chunk_bytes = (int64_t)32 << 20;   // JTK_OPT_HOST_CHUNK_BYTES: host input (small chunks: copies overlap kernels)
    // the whole batch
    DevBuf in_text, in_off;          // device copy of host input (host-buffer entry point)
    DevBuf in_pieces;                // caller-supplied pieces (jtk_batch_encode_pieces): begin[n] | end[n]
    // One allocation holds everything a job hands back: JtkResult + running token totals per chunk | status per document |
    // token offsets | token ids.  A small job's whole answer is then ONE copy to the host, and its header one memset.
    DevBuf out;
    struct View { void* p = nullptr; };
    View job, status, tok_off, tokens;   // where those parts are in `out` for the last job
    DevBuf plan;                     // chunk plan of a device-resident batch
    int64_t* host_plan = nullptr;    // pinned
    size_t host_plan_cap = 0;
    std::vector<int64_t> chunk_doc, chunk_off;
    // JTK_OPT_REUSE_CHUNK_PLAN: the plan of the last device-resident batch is kept while the same offsets array (same pointer, counts)
    // is encoded again -- a step loop --, which saves the plan kernel and the call's only synchronisation.  The caller promises
    // not to change the offsets in place; if it does, the chunks' first and last documents no longer sit where the plan says and
    // k_mark_docs reports JTK_ERR_INVALID_ARGUMENT -- never a wrong answer.
    const int64_t* plan_doc_off = nullptr;
    int64_t plan_docs = -1, plan_bytes = -1, plan_chunk_bytes = -1;
    bool reuse_plan = false;             // JTK_OPT_REUSE_CHUNK_PLAN
    // results streamed to pinned host memory (JTK_ENCODE_TO_HOST)
    int32_t* h_tokens = nullptr; size_t h_tokens_cap = 0;
    int64_t* h_tok_off = nullptr; size_t h_tok_off_cap = 0;
    int32_t* h_status = nullptr; size_t h_status_cap = 0;
    uint8_t* h_small = nullptr; size_t h_small_cap = 0;   // a small job's whole `out` block (pinned)
    uint8_t* h_in = nullptr; size_t h_in_cap = 0;         // a tiny host job's offsets and text, side by side (pinned)
    uint8_t* h_gather = nullptr; size_t h_gather_cap = 0; // jtk_batch_encode_max_tokens: the documents' leading bytes, gathered (pinned)
    // where the host copy of the last job's result is: the buffers above, or inside h_small
    const int32_t* r_tokens = nullptr; const int64_t* r_tok_off = nullptr; int32_t* r_status = nullptr;
    bool have_host_result = false;
    // batch decode (jtk_batch_decode*)
    DevBuf dec_in_ids, dec_in_off, dec_zero, dec_tile, dec_pre, dec_out, dec_byte_off;
    DevBuf trunc_kept, trunc_flag;   // jtk_batch_truncate
    bool have_trunc = false;
    JtkDecodeWork dwork{};
    bool have_decode = false;
    int64_t dec_total = 0;
    JtkResult* host_result = nullptr;   // pinned: host_result_own, or the head of h_small after a small job
    JtkResult* host_result_own = nullptr;
    // the last job
    const uint8_t* job_text = nullptr;
    const int64_t* job_doc_off = nullptr;
    int64_t job_docs = 0, job_bytes = 0;
    uint32_t job_flags = 0;
    bool have_result = false, synced = false;
    bool profiling = false;
    std::vector<hipEvent_t> prof_ev;     // [chunk][stage][start, end]
    int prof_chunks = 0;                 // chunks of the last job that recorded events
};

#include "jtk_unicode_tables.h"
#include "jtk_block_classify.h"

// Host loops over many documents (the gather of the prefixes and the decisions of the maxTokens early exit): slices of
// [0, n) on up to 8 threads when there is enough to share out.
template <class F> static void host_slices(size_t n, F&& fn) {
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt > 8 ? 8 : (nt < 1 ? 1 : nt);
    if (n < 16384 || nt == 1) { fn(0, (size_t)0, n); return; }
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; k++) th.emplace_back([&, k] { fn((int)k, n * k / nt, n * (k + 1) / nt); });
    for (auto& t : th) t.join();
}

// encode()'s special-token check on the host (GptBytePairEncoding.java:52-56: text.contains(literal), per document) for the entry
// points whose text does not pass through k_pretok_split whole: one memchr pass per distinct FIRST byte of the literals (one
// pass for the shipped encodings: every literal starts with '<'), byte ranges shared out over the host threads.
static void find_special_docs(const jtk_encoding* enc, const uint8_t* utf8, int64_t n_bytes, const int64_t* doc_off, int64_t n_docs,
                              std::vector<uint8_t>& flag) {
    flag.assign((size_t)(n_docs > 0 ? n_docs : 1), 0);
    const auto& sp = enc->host.specials;
    if (sp.empty() || n_bytes <= 0) return;
    bool first[256] = {false};
    for (auto& x : sp) if (!x.first.empty()) first[(uint8_t)x.first[0]] = true;
    const size_t slice = (size_t)1 << 22;
    const size_t n_slices = ((size_t)n_bytes + slice - 1) / slice;
    host_slices(n_slices, [&](int, size_t lo, size_t hi) {
        const int64_t b0 = (int64_t)(lo * slice), b1 = std::min<int64_t>(n_bytes, (int64_t)(hi * slice));   // hits that START in [b0, b1)
        for (int fb = 0; fb < 256; fb++) {
            if (!first[fb]) continue;
            const uint8_t* p = utf8 + b0;
            const uint8_t* const endp = utf8 + b1;
            while (p < endp) {
                p = (const uint8_t*)memchr(p, fb, (size_t)(endp - p));
                if (!p) break;
                const int64_t pos = p - utf8;
                for (auto& x : sp) {
                    const std::string& lit = x.first;
                    if (lit.empty() || (uint8_t)lit[0] != (uint8_t)fb || pos + (int64_t)lit.size() > n_bytes) continue;
                    if (memcmp(p, lit.data(), lit.size()) != 0) continue;
                    const int64_t d = (std::upper_bound(doc_off, doc_off + n_docs + 1, pos) - doc_off) - 1;
                    if (d >= 0 && d < n_docs && pos + (int64_t)lit.size() <= doc_off[d + 1]) flag[(size_t)d] = 1;   // (one value: a benign race)
                }
                p++;
            }
        }
    });
}



extern "C" {

const char* jtk_version(void) { return "jtokkit_amd 0.1 (gfx950; Unicode " JTK_UNICODE_VERSION " class tables)"; }
const char* jtk_last_error(void) { return g_err.c_str(); }

int jtk_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(JTK_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}

int jtk_encoding_create(const char* name, int pattern_kind, const uint8_t* tiktoken, size_t tiktoken_len,
                        const char* const* special_literals, const int32_t* special_ids, int n_specials,
                        int device, jtk_encoding** out) {
    if (!out) return fail(JTK_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!tiktoken || n_specials < 0 || (n_specials > 0 && (!special_literals || !special_ids)))
        return fail(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    if (n_specials > JTK_MAX_SPECIALS) return fail(JTK_ERR_INVALID_ARGUMENT, "too many special tokens");
    for (int i = 0; i < n_specials; i++) {
        const size_t l = strlen(special_literals[i]);
        if (l < 1 || l > JTK_SPECIAL_MAXLEN)
            return fail(JTK_ERR_INVALID_ARGUMENT, "special-token literals must be 1..65535 bytes long");
        if (special_ids[i] < 0 || special_ids[i] > (int32_t)JTK_MAX_ID + (1 << 20))
            return fail(JTK_ERR_INVALID_ARGUMENT, "special-token id out of range");
    }
    jtk_encoding* enc = new (std::nothrow) jtk_encoding();
    if (!enc) return fail(JTK_ERR_OUT_OF_MEMORY, "out of host memory");
    std::string err;
    int rc = jtk_build_tables(name, pattern_kind, tiktoken, tiktoken_len, special_literals, special_ids, n_specials,
                              enc->host, err);
    if (rc != JTK_OK) { delete enc; return fail(rc, err); }
    enc->tok_len.assign((size_t)enc->host.max_id + 1 + (size_t)enc->host.n_missing, 1);     // (pseudo ids: one byte each)
    for (size_t i = 0; i <= (size_t)enc->host.max_id; i++) enc->tok_len[i] = (uint32_t)enc->host.id_to_bytes[i].size();

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        delete enc;
        return fail(JTK_ERR_NO_DEVICE, "no HIP device: jtokkit_amd has no CPU path");
    }
    if (device < 0 || device >= ndev) { delete enc; return fail(JTK_ERR_INVALID_ARGUMENT, "device index out of range"); }
    enc->device = device;
    auto cleanup = [&]() { enc->uc1.release(); enc->uc2.release(); enc->brank.release(); enc->pairs.release(); enc->tok8.release(); enc->tok16.release(); enc->bprank.release(); enc->bpbits.release(); enc->bpcum.release(); enc->bpranks.release(); enc->pairin.release(); enc->dec_off.release(); enc->dec_blob.release(); enc->longtok.release(); enc->longblob.release(); enc->specials.release(); delete enc; };
#define ENC_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(JTK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    ENC_TRY(hipSetDevice(device));
    const size_t tok8_bytes = (enc->host.tok8.size() * sizeof(JtkTok8Slot) + 31) & ~(size_t)31;
    if (enc->uc1.ensure(sizeof(jtk_uc_stage1_init)) || enc->uc2.ensure(sizeof(jtk_uc_stage2_init)) ||
        enc->brank.ensure(256 * 4) || enc->pairs.ensure(enc->host.pair_buckets.size() * sizeof(JtkPairBucket)) ||
        // the two whole-piece tables live in one allocation (tok8 slots, then tok16 slots): piece_resolve addresses both with
        // one base and a 32-bit offset; + 32: a 9..16-byte probe of the last slot may read 16 bytes past a 16-byte slot
        enc->tok8.ensure(tok8_bytes + enc->host.tok16.size() * sizeof(JtkTok16Slot) + 32) ||
        enc->bprank.ensure(65536 * 4) ||
        enc->bpbits.ensure(1024 * 8) || enc->bpcum.ensure(1024 * 2) || enc->bpranks.ensure(JTK_BP_MAX * 4) || enc->pairin.ensure(2048 * 4)) { cleanup(); return JTK_ERR_OUT_OF_MEMORY; }
    ENC_TRY(hipMemcpy(enc->uc1.p, jtk_uc_stage1_init, sizeof(jtk_uc_stage1_init), hipMemcpyHostToDevice));
    ENC_TRY(hipMemcpy(enc->uc2.p, jtk_uc_stage2_init, sizeof(jtk_uc_stage2_init), hipMemcpyHostToDevice));
    ENC_TRY(hipMemcpy(enc->brank.p, enc->host.byte_rank, 256 * 4, hipMemcpyHostToDevice));
    ENC_TRY(hipMemcpy(enc->pairs.p, enc->host.pair_buckets.data(), enc->host.pair_buckets.size() * sizeof(JtkPairBucket), hipMemcpyHostToDevice));
    ENC_TRY(hipMemcpy(enc->tok8.p, enc->host.tok8.data(), enc->host.tok8.size() * sizeof(JtkTok8Slot), hipMemcpyHostToDevice));
    ENC_TRY(hipMemcpy((uint8_t*)enc->tok8.p + tok8_bytes, enc->host.tok16.data(), enc->host.tok16.size() * sizeof(JtkTok16Slot), hipMemcpyHostToDevice));
    ENC_TRY(hipMemcpy(enc->bprank.p, enc->host.bp_rank.data(), 65536 * 4, hipMemcpyHostToDevice));
    ENC_TRY(hipMemcpy(enc->bpbits.p, enc->host.bp_bits.data(), 1024 * 8, hipMemcpyHostToDevice));
    ENC_TRY(hipMemcpy(enc->bpcum.p, enc->host.bp_cum.data(), 1024 * 2, hipMemcpyHostToDevice));
    ENC_TRY(hipMemcpy(enc->bpranks.p, enc->host.bp_ranks.data(), JTK_BP_MAX * 4, hipMemcpyHostToDevice));
    ENC_TRY(hipMemcpy(enc->pairin.p, enc->host.pair_in_token.data(), 2048 * 4, hipMemcpyHostToDevice));
    {   // decode table: byte strings of all ids (rank table + special tokens, GptBytePairEncoding.java:302-314) back to back
        uint32_t n_ids = enc->host.max_id + 1;
        for (auto& sp : enc->host.specials) if (sp.second >= 0 && (uint32_t)sp.second + 1 > n_ids) n_ids = (uint32_t)sp.second + 1;
        std::vector<const std::string*> str(n_ids, nullptr);
        for (uint32_t i = 0; i <= enc->host.max_id; i++) if (enc->host.id_present[i]) str[i] = &enc->host.id_to_bytes[i];
        for (auto& sp : enc->host.specials) if (sp.second >= 0 && !str[(size_t)sp.second]) str[(size_t)sp.second] = &sp.first;
        std::vector<uint32_t> off(n_ids + 1, 0);
        std::string blob;
        for (uint32_t i = 0; i < n_ids; i++) { off[i] = (uint32_t)blob.size(); if (str[i]) blob += *str[i]; }
        off[n_ids] = (uint32_t)blob.size();
        blob.append(16, '\0');
        if (enc->dec_off.ensure(off.size() * 4) || enc->dec_blob.ensure(blob.size())) { cleanup(); return JTK_ERR_OUT_OF_MEMORY; }
        ENC_TRY(hipMemcpy(enc->dec_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice));
        ENC_TRY(hipMemcpy(enc->dec_blob.p, blob.data(), blob.size(), hipMemcpyHostToDevice));
        enc->n_ids_table = n_ids;
    }
    if (!enc->host.long_tok.empty()) {
        if (enc->longtok.ensure(enc->host.long_tok.size() * sizeof(JtkLongTokSlot)) || enc->longblob.ensure(enc->host.long_blob.size())) { cleanup(); return JTK_ERR_OUT_OF_MEMORY; }
        ENC_TRY(hipMemcpy(enc->longtok.p, enc->host.long_tok.data(), enc->host.long_tok.size() * sizeof(JtkLongTokSlot), hipMemcpyHostToDevice));
        ENC_TRY(hipMemcpy(enc->longblob.p, enc->host.long_blob.data(), enc->host.long_blob.size(), hipMemcpyHostToDevice));
    }
    JtkDeviceTables& dt = enc->dt;
    memset(&dt, 0, sizeof(dt));
    dt.longtok.slots = (const JtkLongTokSlot*)enc->longtok.p;
    dt.longtok.blob = (const uint8_t*)enc->longblob.p;
    dt.longtok.n = (uint32_t)enc->host.long_tok.size();
    dt.longtok.max_len = enc->host.long_max_len;
    dt.uc.stage1 = (const uint8_t*)enc->uc1.p;
    dt.uc.stage2 = (const uint32_t*)enc->uc2.p;
    dt.uc_stage1_len = JTK_UC_STAGE1_LEN;
    dt.uc_stage2_words = JTK_UC_STAGE2_WORDS;
    dt.byte_rank = (const uint32_t*)enc->brank.p;
    dt.pairs.buckets = (const JtkPairBucket*)enc->pairs.p;
    dt.pairs.bits = enc->host.pair_bits;
    dt.tok8.slots = (const JtkTok8Slot*)enc->tok8.p;
    dt.tok8.bits = enc->host.tok8_bits;
    dt.tok16.slots = (const JtkTok16Slot*)((const uint8_t*)enc->tok8.p + tok8_bytes);
    dt.tok16.n = enc->host.tok16_n;
    dt.bp_rank = (const uint32_t*)enc->bprank.p;
    dt.bp.bits = (const uint64_t*)enc->bpbits.p;
    dt.bp.cum = (const uint16_t*)enc->bpcum.p;
    dt.bp.ranks = (const uint32_t*)enc->bpranks.p;
    dt.pair_in_token = (const uint32_t*)enc->pairin.p;
    {
        const JtkUcTables host_uc{jtk_uc_stage1_init, jtk_uc_stage2_init};
        for (uint32_t b = 0; b < 256; b++) if (jtk_lead_all_letters(host_uc, b)) dt.lead_letters[b >> 5] |= 1u << (b & 31);
    }
    dt.kind = pattern_kind;
    dt.pseudo_base = enc->host.n_missing ? enc->host.pseudo_base : 0u;
    dt.n_specials = n_specials;
    {   // the literals for the device's text.contains check: offsets [n + 1] (u32), then the bytes
        std::vector<uint32_t> off((size_t)n_specials + 1, 0);
        std::vector<uint8_t> blob;
        for (int i = 0; i < n_specials; i++) {
            const size_t l = strlen(special_literals[i]);
            blob.insert(blob.end(), (const uint8_t*)special_literals[i], (const uint8_t*)special_literals[i] + l);
            off[(size_t)i + 1] = (uint32_t)blob.size();
            const uint8_t f = (uint8_t)special_literals[i][0];
            dt.special_first[f >> 5] |= 1u << (f & 31);
        }
        const size_t off_bytes = align_up(off.size() * 4, 16);
        if (enc->specials.ensure(off_bytes + blob.size() + 16)) { cleanup(); return JTK_ERR_OUT_OF_MEMORY; }
        ENC_TRY(hipMemcpy(enc->specials.p, off.data(), off.size() * 4, hipMemcpyHostToDevice));
        if (!blob.empty()) ENC_TRY(hipMemcpy((uint8_t*)enc->specials.p + off_bytes, blob.data(), blob.size(), hipMemcpyHostToDevice));
        dt.special_off = (const uint32_t*)enc->specials.p;
        dt.special_blob = (const uint8_t*)enc->specials.p + off_bytes;
    }
#undef ENC_TRY
    *out = enc;
    return JTK_OK;
}

void jtk_encoding_destroy(jtk_encoding* enc) {
    if (!enc) return;
    (void)hipSetDevice(enc->device);
    enc->uc1.release(); enc->uc2.release(); enc->brank.release(); enc->pairs.release();
    enc->tok8.release(); enc->tok16.release(); enc->bprank.release(); enc->bpbits.release(); enc->bpcum.release(); enc->bpranks.release(); enc->pairin.release();
    enc->dec_off.release(); enc->dec_blob.release(); enc->longtok.release(); enc->longblob.release(); enc->specials.release();
    delete enc;
}
const char* jtk_encoding_name(const jtk_encoding* enc) { return enc ? enc->host.name.c_str() : ""; }
int jtk_encoding_device(const jtk_encoding* enc) { return enc ? enc->device : -1; }
int64_t jtk_encoding_vocab_size(const jtk_encoding* enc) { return enc ? enc->host.n_tokens : 0; }
int64_t jtk_encoding_pair_count(const jtk_encoding* enc) { return enc ? enc->host.n_pairs : 0; }

int jtk_batch_create(const jtk_encoding* enc, jtk_batch** out) {
    if (!enc || !out) return fail(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    *out = nullptr;
    HIP_TRY(hipSetDevice(enc->device));
    jtk_batch* b = new (std::nothrow) jtk_batch();
    if (!b) return fail(JTK_ERR_OUT_OF_MEMORY, "out of host memory");
    b->enc = enc;
    if (const char* e = getenv("JTK_CHUNK_BYTES")) { const long long v = atoll(e); if (v >= (1 << 20)) b->chunk_bytes = v; }
    if (const char* e = getenv("JTK_HOST_CHUNK_BYTES")) { const long long v = atoll(e); if (v >= (1 << 16)) b->host_chunk_bytes = v; }
    if (const char* e = getenv("JTK_CHUNKS_IN_FLIGHT")) { const int v = atoi(e); if (v >= 1 && v <= MAX_SETS) { b->n_sets = v; b->n_sets_chosen = true; } }
    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void**)&b->host_result_own, sizeof(JtkResult), hipHostMallocDefault);
    b->host_result = b->host_result_own;
    // (the streams and events of the chunk pipeline are created by the first job that forks: a batch that only ever sees
    // small single-chunk jobs -- one per caller thread in the per-call shape -- owns one stream, not six)
    if (e != hipSuccess) {
        jtk_batch_destroy(b);
        return fail(JTK_ERR_HIP, std::string("batch create: ") + hipGetErrorString(e));
    }
    *out = b;
    return JTK_OK;
}

void jtk_batch_destroy(jtk_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->enc->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (ChunkSet& cs : b->set) {
        if (cs.stream) (void)hipStreamSynchronize(cs.stream);
        DevBuf* bufs[] = {&cs.zeroed, &cs.piecemask, &cs.gapmask, &cs.plist, &cs.htok, &cs.docpre, &cs.tile_np, &cs.tile_off, &cs.queues, &cs.q_meta,
                          &cs.mid_list, &cs.long_list, &cs.giant_list};
        for (DevBuf* d : bufs) d->release();
        if (cs.ev_scan) (void)hipEventDestroy(cs.ev_scan);
        if (cs.ev_done) (void)hipEventDestroy(cs.ev_done);
        if (cs.stream) (void)hipStreamDestroy(cs.stream);
    }
    DevBuf* bufs[] = {&b->in_text, &b->in_off, &b->in_pieces, &b->out, &b->plan, &b->dec_in_ids, &b->dec_in_off,
                      &b->dec_zero, &b->dec_tile, &b->dec_pre, &b->dec_out, &b->dec_byte_off, &b->trunc_kept, &b->trunc_flag};
    for (DevBuf* d : bufs) d->release();
    for (hipEvent_t ev : b->prof_ev) (void)hipEventDestroy(ev);
    if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
    if (b->ev_copy) (void)hipEventDestroy(b->ev_copy);
    if (b->copy_stream) { (void)hipStreamSynchronize(b->copy_stream); (void)hipStreamDestroy(b->copy_stream); }
    if (b->h_info) (void)hipHostFree(b->h_info);
    if (b->host_result_own) (void)hipHostFree(b->host_result_own);
    if (b->h_small) (void)hipHostFree(b->h_small);
    if (b->h_in) (void)hipHostFree(b->h_in);
    if (b->h_gather) (void)hipHostFree(b->h_gather);
    if (b->host_plan) (void)hipHostFree(b->host_plan);
    if (b->h_tokens) (void)hipHostFree(b->h_tokens);
    if (b->h_tok_off) (void)hipHostFree(b->h_tok_off);
    if (b->h_status) (void)hipHostFree(b->h_status);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

int jtk_batch_set_option(jtk_batch* b, int option, int64_t value) {
    if (!b) return fail(JTK_ERR_INVALID_ARGUMENT, "batch is NULL");
    switch (option) {
        case JTK_OPT_CHUNK_BYTES:
            if (value < (1 << 16)) return fail(JTK_ERR_INVALID_ARGUMENT, "chunk size must be at least 64 KiB");
            b->chunk_bytes = value;
            return JTK_OK;
        case JTK_OPT_HOST_CHUNK_BYTES:
            if (value < (1 << 16)) return fail(JTK_ERR_INVALID_ARGUMENT, "chunk size must be at least 64 KiB");
            b->host_chunk_bytes = value;
            return JTK_OK;
        case JTK_OPT_CHUNKS_IN_FLIGHT:
            if (value < 1 || value > MAX_SETS) return fail(JTK_ERR_INVALID_ARGUMENT, "chunks in flight: 1..4");
            b->n_sets = (int)value;
            b->n_sets_chosen = true;
            return JTK_OK;
        case JTK_OPT_REUSE_CHUNK_PLAN:
            b->reuse_plan = value != 0;
            b->plan_doc_off = nullptr;
            return JTK_OK;
        default: return fail(JTK_ERR_INVALID_ARGUMENT, "unknown option");
    }
}

int jtk_batch_set_profiling(jtk_batch* b, int enabled) {
    if (!b) return fail(JTK_ERR_INVALID_ARGUMENT, "batch is NULL");
    b->profiling = enabled != 0;
    b->prof_chunks = 0;
    return JTK_OK;
}

int jtk_host_alloc(size_t bytes, void** out) {
    if (!out) return fail(JTK_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(JTK_ERR_OUT_OF_MEMORY, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    return JTK_OK;
}
void jtk_host_free(void* p) { if (p) (void)hipHostFree(p); }

}  // extern "C"

namespace {

int ensure_pinned(void** p, size_t* cap, size_t bytes, size_t keep_bytes) {
    if (bytes <= *cap) return JTK_OK;
    void* q = nullptr;
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(&q, want, hipHostMallocDefault);
    if (e != hipSuccess) return fail(JTK_ERR_OUT_OF_MEMORY, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    if (*p) { if (keep_bytes) memcpy(q, *p, keep_bytes); (void)hipHostFree(*p); }
    *p = q; *cap = want;
    return JTK_OK;
}

// scratch of one chunk of `n_bytes` (from its tile-aligned origin) and `n_docs` documents; fills cs.work
int prepare_set(ChunkSet& cs, int64_t n_bytes, int64_t n_docs, size_t* bytes_to_clear) {
    JtkWork& w = cs.work;
    w.n_bytes = n_bytes;
    w.n_docs = n_docs;
    w.n_words = (n_bytes + 1 + 63) / 64 + 2;
    w.n_tiles = (n_bytes + 1 + JTK_TILE - 1) / JTK_TILE;
    const size_t mask_bytes = (size_t)w.n_words * 8;
    const size_t nt = (size_t)w.n_tiles;
    const size_t qcnt_bytes = (size_t)(JTK_NBINS + 1) * JTK_Q_SHARDS * JTK_QC_STRIDE * 4;
    const size_t zero_bytes = mask_bytes + 32 + qcnt_bytes;
    const size_t n_long_max = (size_t)n_bytes / (JTK_BIN_MAXLEN + 1) + 2;
    const size_t n_giant_max = (size_t)n_bytes / JTK_LONG_CAP + 2;
    const size_t tps = (nt + JTK_Q_SHARDS - 1) / JTK_Q_SHARDS;      // tiles per queue shard
    int rc;
    if ((rc = cs.zeroed.ensure(zero_bytes)) || (rc = cs.piecemask.ensure(mask_bytes)) ||
        (rc = cs.plist.ensure(nt * JTK_TILE * 4)) || (rc = cs.htok.ensure(nt * JTK_TILE * 4 + 64)) ||
        (rc = cs.docpre.ensure(nt * JTK_TILE * 4)) ||
        (rc = cs.tile_np.ensure(align_up(nt * 4, 16) * 2)) || (rc = cs.tile_off.ensure((nt + 1) * 8)) ||
        (rc = cs.q_meta.ensure(nt * 4 * 16)) ||
        (rc = cs.queues.ensure(tps * JTK_Q_SHARDS * ((size_t)(JTK_BIN_CAP0 + JTK_BIN_CAP1 + JTK_BIN_CAP2 + JTK_BIN_CAP3 + JTK_BIN_CAP4 + JTK_BIN_CAP5 + JTK_BIN_CAP6) * 24 + (size_t)JTK_TINY_CAP * 8))) ||
        (rc = cs.mid_list.ensure(n_long_max * sizeof(JtkLongPiece))) ||
        (rc = cs.long_list.ensure(n_long_max * sizeof(JtkLongPiece))) ||
        (rc = cs.giant_list.ensure(n_giant_max * sizeof(JtkLongPiece))))
        return rc;
    uint8_t* z = (uint8_t*)cs.zeroed.p;
    w.docmask = (uint64_t*)z;
    w.mid_count = (uint32_t*)(z + mask_bytes);
    w.long_count = (uint32_t*)(z + mask_bytes + 4);
    w.n_giant = (uint32_t*)(z + mask_bytes + 8);
    w.q_count = (uint32_t*)(z + mask_bytes + 32);
    w.piecemask = (uint64_t*)cs.piecemask.p;
    w.plist = (uint32_t*)cs.plist.p;
    w.htok = (uint32_t*)cs.htok.p;
    w.docpre = (uint32_t*)cs.docpre.p;
    w.tile_np = (uint32_t*)cs.tile_np.p;
    w.tile_tot = (uint32_t*)((uint8_t*)cs.tile_np.p + align_up(nt * 4, 16));
    w.tile_off = (int64_t*)cs.tile_off.p;
    {
        const size_t caps[JTK_NBINS] = {JTK_BIN_CAP0, JTK_BIN_CAP1, JTK_BIN_CAP2, JTK_BIN_CAP3, JTK_BIN_CAP4, JTK_BIN_CAP5, JTK_BIN_CAP6};
        uint8_t* qp = (uint8_t*)cs.queues.p;                      // all the 16-byte arrays first, then the 8-byte ones
        w.q_meta = (uint32_t*)cs.q_meta.p;
        for (int k = 0; k < JTK_NBINS; k++) {
            w.qd[k] = (uint4*)qp;
            w.q_cap[k] = (int64_t)(tps * caps[k]);
            qp += tps * caps[k] * JTK_Q_SHARDS * 16;
        }
        for (int k = 0; k < JTK_NBINS; k++) {
            w.qm[k] = (uint64_t*)qp;
            qp += tps * caps[k] * JTK_Q_SHARDS * 8;
        }
        w.qt = (uint64_t*)qp;
        w.qt_cap = (int64_t)(tps * JTK_TINY_CAP);
    }
    w.mid_list = (JtkLongPiece*)cs.mid_list.p;
    w.long_list = (JtkLongPiece*)cs.long_list.p;
    w.giant_list = (JtkLongPiece*)cs.giant_list.p;
    *bytes_to_clear = zero_bytes;
    return JTK_OK;
}

// The whole job: `chunk_doc` / `chunk_off` (n_chunks + 1 entries, filled by the caller) cut the batch into runs of whole
// documents.  Chunk c runs on scratch set c % n_sets and that set's stream; the sets' streams are forked from the main
// stream `s` and joined into it at the end, so the job as a whole is ordered like one operation on `s`.  The only link
// between chunks is the running token total (chunk c's scan follows chunk c - 1's).
// h_text != NULL: the text comes from host memory, copied chunk by chunk into in_text on the chunk's stream (so the copy of
// chunk c + 1 overlaps the kernels of chunk c).  to_host: every chunk's tokens are copied to the pinned host buffer as
// soon as they are packed.
struct PieceArgs {               // caller-supplied pieces instead of pretok_split
    const int64_t* d_begin; const int64_t* d_end;     // device copies
    const int64_t* h_begin;                           // host: to find each chunk's pieces
    int64_t n;
};

int run_job(jtk_batch* b, const uint8_t* d_text, const uint8_t* h_text, const int64_t* d_doc_off, int64_t n_docs, int64_t n_bytes,
            uint32_t flags, hipStream_t s, bool to_host, const PieceArgs* pieces = nullptr) {
    const jtk_encoding* enc = b->enc;
    const int n_chunks = (int)b->chunk_doc.size() - 1;
    // (ids streamed to the host chunk by chunk: the host waits for a chunk's scan before it can issue that chunk's copy, one
    // chunk behind the enqueueing -- with two sets that wait stalls the pipeline: 404 ms per step on the headline corpus against
    // 163 with three, profiles/r03_experiments/r03bc_e2e_memcpy.txt -- so such a job takes three unless the caller chose)
    const int want_sets = (!b->n_sets_chosen && to_host && h_text && b->n_sets < 3) ? 3 : b->n_sets;   // (host input: small chunks, small sets)
    const int n_sets = n_chunks < want_sets ? (n_chunks < 1 ? 1 : n_chunks) : want_sets;
    int rc;
    // batch-wide buffers
    const size_t status_bytes = align_up((size_t)(n_docs > 0 ? n_docs : 1) * 4, 16);
    const size_t job_bytes = align_up(64 + ((size_t)n_chunks + 2) * 8, 16);
    const size_t off_status = job_bytes, off_tok_off = off_status + status_bytes;
    const size_t off_tokens = align_up(off_tok_off + ((size_t)n_docs + 1) * 8, 256);
    // a small single-chunk job (the per-call service's batches) copies its whole output block -- header, status, offsets and
    // the worst-case token range (one token per byte) -- right behind the kernels in ONE copy, instead of waiting for the
    // token count first: one host synchronisation and three copies less per batch
    const bool small_to_host = to_host && n_chunks == 1 && n_bytes <= SMALL_JOB_BYTES;
    // r03: for the smallest of them (a per-call device batch) the output block IS pinned host memory -- pack, doc_offsets and the
    // status atomics write over the link, and the copy up (a DMA of 14 us for 27 documents) is gone.  JTK_TINY_COPY_OUT=1: the copy.
    static const bool copy_out_env = getenv("JTK_TINY_COPY_OUT") != nullptr;
    const bool zero_copy_out = !copy_out_env && small_to_host && n_bytes <= TINY_JOB_BYTES && !h_text;
    uint8_t* out_base;
    if (zero_copy_out) {
        if ((rc = ensure_pinned((void**)&b->h_small, &b->h_small_cap, off_tokens + ((size_t)n_bytes + 64) * 4 + 4096, 0))) return rc;
        out_base = b->h_small;
    } else {
        if ((rc = b->out.ensure(off_tokens + ((size_t)n_bytes + 64) * 4))) return rc;
        out_base = (uint8_t*)b->out.p;
    }
    b->job.p = out_base;
    b->status.p = out_base + off_status;
    b->tok_off.p = out_base + off_tok_off;
    b->tokens.p = out_base + off_tokens;
    JtkResult* d_result = (JtkResult*)b->job.p;
    int64_t* d_totals = (int64_t*)((uint8_t*)b->job.p + 64);       // [c]: tokens of the chunks before c
    HIP_TRY(hipMemsetAsync(out_base, 0, off_tok_off, s));          // JtkResult, totals and status
    if (to_host && !small_to_host) {
        if ((rc = ensure_pinned((void**)&b->h_tok_off, &b->h_tok_off_cap, ((size_t)n_docs + 1) * 8, 0)) ||
            (rc = ensure_pinned((void**)&b->h_status, &b->h_status_cap, (size_t)(n_docs > 0 ? n_docs : 1) * 4, 0)) ||
            (rc = ensure_pinned((void**)&b->h_tokens, &b->h_tokens_cap, (size_t)n_bytes * (n_bytes <= SMALL_JOB_BYTES ? 4 : 2) + 4096, 0)))     // grown as the chunks report
            return rc;
    }
    const bool prof = b->profiling;
    if (prof) {
        const size_t need = (size_t)n_chunks * N_STAGES * 2;
        while (b->prof_ev.size() < need) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); b->prof_ev.push_back(ev); }
    }
    const bool fork = n_chunks > 1 || (h_text != nullptr && !(n_chunks == 1 && n_bytes <= SMALL_JOB_BYTES));
    if (fork) {
        if (!b->ev_fork) {
            HIP_TRY(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&b->ev_copy, hipEventDisableTiming));
            HIP_TRY(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking));
        }
        for (int k = 0; k < n_sets; k++) {
            ChunkSet& cs = b->set[k];
            if (cs.stream) continue;
            HIP_TRY(hipStreamCreateWithFlags(&cs.stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&cs.ev_scan, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&cs.ev_done, hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(b->ev_fork, s));
    }
    for (ChunkSet& cs : b->set) cs.used = false;

    if ((rc = ensure_pinned((void**)&b->h_info, &b->h_info_cap, ((size_t)n_chunks + 1) * 16, 0))) return rc;
    // tokens of chunk c to the host: on the copy stream, after the chunk's last kernel; the host needs the chunk's token
    // range for that (written to pinned memory by its scan), so this is called one chunk behind the enqueueing
    auto send_chunk_to_host = [&](int c) -> int {
        ChunkSet& cs = b->set[c % n_sets];
        HIP_TRY(hipEventSynchronize(cs.ev_scan));
        const int64_t t0 = b->h_info[2 * c], t1 = b->h_info[2 * c + 1];
        if (t1 > t0 && !(flags & JTK_ENCODE_COUNT_ONLY)) {
            if ((size_t)t1 * 4 > b->h_tokens_cap) {
                // grow: earlier chunks' copies may still be in flight into the old buffer
                HIP_TRY(hipStreamSynchronize(b->copy_stream));
                int rc2 = ensure_pinned((void**)&b->h_tokens, &b->h_tokens_cap, (size_t)t1 * 4 + ((size_t)n_bytes - (size_t)b->chunk_off[c + 1]) * 2, (size_t)t0 * 4);
                if (rc2) return rc2;
            }
            HIP_TRY(hipStreamWaitEvent(b->copy_stream, cs.ev_done, 0));
            HIP_TRY(hipMemcpyAsync(b->h_tokens + t0, (const int32_t*)b->tokens.p + t0, (size_t)(t1 - t0) * 4, hipMemcpyDeviceToHost, b->copy_stream));
        }
        return JTK_OK;
    };

    for (int c = 0; c < n_chunks; c++) {
        ChunkSet& cs = b->set[c % n_sets];
        hipStream_t cst = fork ? cs.stream : s;
        const int64_t d0 = b->chunk_doc[c], d1 = b->chunk_doc[c + 1];
        const int64_t b0 = b->chunk_off[c], b1 = b->chunk_off[c + 1];
        const int64_t origin = b0 & ~(int64_t)(JTK_TILE - 1);
        // (a set is reused by chunk c + n_sets on the same stream: its scratch is free by then)
        size_t zero_bytes = 0;
        if ((rc = prepare_set(cs, b1 - origin, d1 - d0, &zero_bytes)) != JTK_OK) return rc;
        JtkWork& w = cs.work;
        w.set_info = b->h_info + 2 * c;
        w.text = d_text + origin;
        w.text_base = origin;
        w.lead = b0 - origin;
        w.doc_off = d_doc_off + d0;
        w.status = (int32_t*)b->status.p + d0;
        w.tokens = (int32_t*)b->tokens.p;
        w.tok_off = (int64_t*)b->tok_off.p + d0;
        w.result = d_result;
        w.job_tokens = d_totals + c;
        w.job_tokens_next = d_totals + c + 1;
        w.check_special = (!(flags & JTK_ENCODE_ORDINARY) && enc->dt.n_specials > 0) ? 1u : 0u;
        // (a table that lacks single bytes: the ids are needed to tell which documents cannot be encoded, also for a count)
        w.count_only = ((flags & JTK_ENCODE_COUNT_ONLY) && !enc->dt.pseudo_base) ? 1u : 0u;
        w.inline_scan = (!fork && n_chunks == 1 && w.n_tiles >= 1 && w.n_tiles <= 1024) ? 1u : 0u;

        if (fork && !cs.used) { HIP_TRY(hipStreamWaitEvent(cst, b->ev_fork, 0)); cs.used = true; }
        if (h_text && b1 > b0) {
            // this chunk's bytes; the bytes before b0 in the first tile were copied with the previous chunk
            HIP_TRY(hipMemcpyAsync((uint8_t*)b->in_text.p + b0, h_text + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, cst));
        }
        int stage = 0;
        hipEvent_t* pe = prof ? &b->prof_ev[(size_t)c * N_STAGES * 2] : nullptr;
        auto begin = [&]() { if (prof) (void)hipEventRecord(pe[2 * stage], cst); };
        auto end = [&]() { if (prof) (void)hipEventRecord(pe[2 * stage + 1], cst); stage++; };

        begin();                                                    // mark_docs
        HIP_TRY(hipMemsetAsync(cs.zeroed.p, 0, zero_bytes, cst));
        jtk_launch_mark_docs(w, cst);
        end();
        begin();                                                    // optional UTF-8 validation (the special-token check rides in pretok_split)
        if (flags & JTK_ENCODE_VALIDATE_UTF8) jtk_launch_validate_utf8(w, cst);
        end();
        begin();
        if (pieces) {
            const size_t mask_bytes = (size_t)w.n_words * 8;
            if ((rc = cs.gapmask.ensure(mask_bytes))) return rc;
            w.gapmask = (uint64_t*)cs.gapmask.p;
            HIP_TRY(hipMemsetAsync(cs.piecemask.p, 0, mask_bytes, cst));
            HIP_TRY(hipMemsetAsync(cs.gapmask.p, 0, mask_bytes, cst));
            const int64_t* pb = pieces->h_begin;
            const int64_t p0 = std::lower_bound(pb, pb + pieces->n, b0) - pb, p1 = std::lower_bound(pb, pb + pieces->n, b1) - pb;
            jtk_launch_mark_pieces(w, pieces->d_begin + p0, pieces->d_end + p0, p1 - p0, cst);
        } else {
            w.gapmask = nullptr;
            jtk_launch_pretok_split(w, enc->dt, cst);
        }
        end();
        begin();
        jtk_launch_piece_resolve(w, enc->dt, cst);
        end();
        begin();
        jtk_launch_long_shortcut(w, enc->dt, cst);                  // (only for rank tables with entries merging cannot reproduce)
        jtk_launch_bpe_merge(w, enc->dt, cst);
        end();
        begin();
        if (fork && c > 0) HIP_TRY(hipStreamWaitEvent(cst, b->set[(c - 1) % n_sets].ev_scan, 0));
        if (!w.inline_scan) jtk_launch_tile_scan(w, cst);
        if (fork) HIP_TRY(hipEventRecord(cs.ev_scan, cst));
        end();
        begin();
        jtk_launch_pack(w, cst);
        end();
        begin();
        jtk_launch_doc_offsets(w, cst);
        if (enc->dt.pseudo_base) jtk_launch_flag_unencodable(w, enc->dt.pseudo_base, cst);
        end();
        HIP_TRY(hipGetLastError());
        if (small_to_host && !zero_copy_out) {
            // the whole answer in one copy: header, status, offsets and the worst-case token range (one token per byte)
            const size_t total = (flags & JTK_ENCODE_COUNT_ONLY) ? off_tokens : off_tokens + (size_t)n_bytes * 4;
            if ((rc = ensure_pinned((void**)&b->h_small, &b->h_small_cap, total + 4096, 0))) return rc;
            HIP_TRY(hipMemcpyAsync(b->h_small, b->out.p, total, hipMemcpyDeviceToHost, cst));
        }
        if (fork) HIP_TRY(hipEventRecord(cs.ev_done, cst));
        if (to_host && !small_to_host && c > 0) { if ((rc = send_chunk_to_host(c - 1)) != JTK_OK) return rc; }
    }
    if (to_host && !small_to_host && n_chunks > 0) { if ((rc = send_chunk_to_host(n_chunks - 1)) != JTK_OK) return rc; }
    if (fork) {
        for (int k = 0; k < n_sets; k++)
            if (b->set[k].used) HIP_TRY(hipStreamWaitEvent(s, b->set[k].ev_done, 0));
        if (to_host && !small_to_host) {
            HIP_TRY(hipEventRecord(b->ev_copy, b->copy_stream));
            HIP_TRY(hipStreamWaitEvent(s, b->ev_copy, 0));
        }
    }
    if (small_to_host) {
        b->host_result = (JtkResult*)b->h_small;
        b->r_status = (int32_t*)(b->h_small + off_status);
        b->r_tok_off = (const int64_t*)(b->h_small + off_tok_off);
        b->r_tokens = (const int32_t*)(b->h_small + off_tokens);
    } else {
        if (to_host) {
            HIP_TRY(hipMemcpyAsync(b->h_tok_off, b->tok_off.p, ((size_t)n_docs + 1) * 8, hipMemcpyDeviceToHost, s));
            if (n_docs > 0) HIP_TRY(hipMemcpyAsync(b->h_status, b->status.p, (size_t)n_docs * 4, hipMemcpyDeviceToHost, s));
        }
        b->host_result = b->host_result_own;
        HIP_TRY(hipMemcpyAsync(b->host_result, d_result, sizeof(JtkResult), hipMemcpyDeviceToHost, s));
        b->r_status = b->h_status; b->r_tok_off = b->h_tok_off; b->r_tokens = b->h_tokens;
    }
    b->job_text = d_text;
    b->job_doc_off = d_doc_off;
    b->job_docs = n_docs;
    b->job_bytes = n_bytes;
    b->job_flags = flags;
    b->have_result = true;
    b->have_host_result = to_host;
    b->have_trunc = false;
    b->synced = false;
    b->last_stream = s;
    b->prof_chunks = prof ? n_chunks : 0;
    return JTK_OK;
}

}  // namespace

extern "C" {

int jtk_batch_encode_device(jtk_batch* b, const uint8_t* d_utf8, const int64_t* d_doc_off, int64_t n_docs,
                            int64_t n_bytes, uint32_t flags, void* stream_or_null, int64_t* n_tokens) {
    if (!b || n_docs < 0 || n_bytes < 0 || (n_bytes > 0 && !d_utf8) || !d_doc_off)
        return fail(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    if (((uintptr_t)d_utf8 & 15u) != 0) return fail(JTK_ERR_INVALID_ARGUMENT, "device text must be 16-byte aligned");
    if (n_bytes >= (int64_t)1 << 37) return fail(JTK_ERR_INVALID_ARGUMENT, "batch too large (128 GiB of text per call at most)");
    if (flags & JTK_ENCODE_TO_HOST) return fail(JTK_ERR_INVALID_ARGUMENT, "JTK_ENCODE_TO_HOST is for jtk_batch_encode (host buffers)");
    HIP_TRY(hipSetDevice(b->enc->device));
    hipStream_t s = stream_or_null ? (hipStream_t)stream_or_null : b->stream;
    // chunk plan: a batch of up to one chunk needs none; a larger one reads the chunk boundaries from the offsets (the one
    // place where this call waits for the work queued on `s` before it)
    const bool same_plan = b->reuse_plan && b->plan_doc_off == d_doc_off && b->plan_docs == n_docs && b->plan_bytes == n_bytes && b->plan_chunk_bytes == b->chunk_bytes &&
                           b->chunk_doc.size() >= 2 && b->chunk_doc.back() == n_docs && b->chunk_off.back() == n_bytes;
    if (!same_plan) {
    b->chunk_doc.assign(1, 0);
    b->chunk_off.assign(1, 0);
    if (n_bytes > b->chunk_bytes + b->chunk_bytes / 4 && n_docs > 1) {
        const int nc = (int)((n_bytes + b->chunk_bytes - 1) / b->chunk_bytes);
        int rc;
        if ((rc = b->plan.ensure(((size_t)nc + 1) * 16))) return rc;
        if ((rc = ensure_pinned((void**)&b->host_plan, &b->host_plan_cap, ((size_t)nc + 1) * 16, 0))) return rc;
        int64_t* d_doc = (int64_t*)b->plan.p;
        int64_t* d_off = d_doc + nc + 1;
        jtk_launch_plan_chunks(d_doc_off, n_docs, b->chunk_bytes, nc, d_doc, d_off, s);
        HIP_TRY(hipMemcpyAsync(b->host_plan, b->plan.p, ((size_t)nc + 1) * 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const int64_t* h_doc = b->host_plan;
        const int64_t* h_off = b->host_plan + nc + 1;
        for (int c = 1; c < nc; c++) {
            const int64_t d = h_doc[c], o = h_off[c];
            if (d <= b->chunk_doc.back() || d >= n_docs) continue;                 // no new document since the last boundary
            if (o < b->chunk_off.back() || o > n_bytes) return fail(JTK_ERR_INVALID_ARGUMENT, "document offsets are not non-decreasing within [0, n_bytes]");
            b->chunk_doc.push_back(d);
            b->chunk_off.push_back(o);
        }
    }
    b->chunk_doc.push_back(n_docs);
    b->chunk_off.push_back(n_bytes);
    b->plan_doc_off = d_doc_off; b->plan_docs = n_docs; b->plan_bytes = n_bytes; b->plan_chunk_bytes = b->chunk_bytes;
    }
    int rc = run_job(b, d_utf8, nullptr, d_doc_off, n_docs, n_bytes, flags, s, false);
    if (rc != JTK_OK) b->plan_doc_off = nullptr;
    if (rc != JTK_OK) return rc;
    if (n_tokens) {
        HIP_TRY(hipStreamSynchronize(s));
        b->synced = true;
        *n_tokens = b->host_result->n_tokens;
    }
    return JTK_OK;
}

int jtk_batch_encode(jtk_batch* b, const uint8_t* utf8, const int64_t* doc_off, int64_t n_docs,
                     uint32_t flags, int64_t* n_tokens) {
    if (!b || n_docs < 0 || !doc_off) return fail(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    if (doc_off[0] != 0) return fail(JTK_ERR_INVALID_ARGUMENT, "doc_off[0] must be 0");
    const int64_t n_bytes = doc_off[n_docs];
    if (n_bytes > 0 && !utf8) return fail(JTK_ERR_INVALID_ARGUMENT, "utf8 is NULL");
    if (n_bytes >= (int64_t)1 << 37) return fail(JTK_ERR_INVALID_ARGUMENT, "batch too large (128 GiB of text per call at most)");
    // chunk plan (and the check of the offsets) in one pass
    const int64_t cb = b->host_chunk_bytes < b->chunk_bytes ? b->host_chunk_bytes : b->chunk_bytes;
    b->plan_doc_off = nullptr;
    b->chunk_doc.assign(1, 0);
    b->chunk_off.assign(1, 0);
    {
        int64_t next = cb;
        for (int64_t d = 0; d < n_docs; d++) {
            if (doc_off[d + 1] < doc_off[d]) return fail(JTK_ERR_INVALID_ARGUMENT, "doc_off must be non-decreasing");
            if (doc_off[d] >= next && d > b->chunk_doc.back() && n_bytes - doc_off[d] > cb / 4) {
                b->chunk_doc.push_back(d);
                b->chunk_off.push_back(doc_off[d]);
                next = doc_off[d] + cb;
            }
        }
    }
    b->chunk_doc.push_back(n_docs);
    b->chunk_off.push_back(n_bytes);
    HIP_TRY(hipSetDevice(b->enc->device));
    int rc;
    if (b->chunk_doc.size() == 2 && n_bytes <= TINY_JOB_BYTES) {
        // a per-call device batch: offsets and text go down in ONE copy (staged side by side in pinned memory; copying a few KB on
        // the host costs less than a second DMA)
        const size_t off_text = align_up(((size_t)n_docs + 1) * 8, 256), total = off_text + (size_t)n_bytes;
        if ((rc = ensure_pinned((void**)&b->h_in, &b->h_in_cap, total + 64, 0)) || (rc = b->in_text.ensure(total + 64))) return rc;
        memcpy(b->h_in, doc_off, ((size_t)n_docs + 1) * 8);
        if (n_bytes > 0) memcpy(b->h_in + off_text, utf8, (size_t)n_bytes);
        // (the other way round -- the kernels reading text and offsets from the pinned block over the link -- was 5 us slower:
        // four kernels read the text)
        HIP_TRY(hipMemcpyAsync(b->in_text.p, b->h_in, total, hipMemcpyHostToDevice, b->stream));
        rc = run_job(b, (const uint8_t*)b->in_text.p + off_text, nullptr, (const int64_t*)b->in_text.p, n_docs, n_bytes,
                     flags & ~(uint32_t)JTK_ENCODE_TO_HOST, b->stream, (flags & JTK_ENCODE_TO_HOST) != 0);
    } else {
        if ((rc = b->in_text.ensure((size_t)n_bytes + 64)) || (rc = b->in_off.ensure(((size_t)n_docs + 1) * 8))) return rc;
        HIP_TRY(hipMemcpyAsync(b->in_off.p, doc_off, ((size_t)n_docs + 1) * 8, hipMemcpyHostToDevice, b->stream));
        rc = run_job(b, (const uint8_t*)b->in_text.p, utf8, (const int64_t*)b->in_off.p, n_docs, n_bytes, flags & ~(uint32_t)JTK_ENCODE_TO_HOST,
                     b->stream, (flags & JTK_ENCODE_TO_HOST) != 0);
    }
    if (rc != JTK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    b->synced = true;
    if (n_tokens) *n_tokens = b->host_result->n_tokens;
    return JTK_OK;
}

int jtk_batch_encode_pieces(jtk_batch* b, const uint8_t* utf8, const int64_t* doc_off, int64_t n_docs,
                            const int64_t* piece_begin, const int64_t* piece_end, int64_t n_pieces, uint32_t flags, int64_t* n_tokens) {
    if (!b || n_docs < 0 || !doc_off || n_pieces < 0 || (n_pieces > 0 && (!piece_begin || !piece_end)))
        return fail(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    if (doc_off[0] != 0) return fail(JTK_ERR_INVALID_ARGUMENT, "doc_off[0] must be 0");
    const int64_t n_bytes = doc_off[n_docs];
    if (n_bytes > 0 && !utf8) return fail(JTK_ERR_INVALID_ARGUMENT, "utf8 is NULL");
    if (n_bytes >= (int64_t)1 << 37) return fail(JTK_ERR_INVALID_ARGUMENT, "batch too large (128 GiB of text per call at most)");
    // pieces: ascending, non-empty, non-overlapping, each inside one document
    {
        int64_t d = 0, prev_end = 0;
        for (int64_t i = 0; i < n_pieces; i++) {
            const int64_t p = piece_begin[i], e = piece_end[i];
            if (p < prev_end || e <= p || e > n_bytes) return fail(JTK_ERR_INVALID_ARGUMENT, "pieces must be ascending, non-empty and non-overlapping");
            while (d < n_docs && doc_off[d + 1] <= p) d++;
            if (d >= n_docs || e > doc_off[d + 1]) return fail(JTK_ERR_INVALID_ARGUMENT, "a piece crosses a document boundary");
            prev_end = e;
        }
    }
    const int64_t cb = b->host_chunk_bytes < b->chunk_bytes ? b->host_chunk_bytes : b->chunk_bytes;
    b->plan_doc_off = nullptr;
    b->chunk_doc.assign(1, 0);
    b->chunk_off.assign(1, 0);
    {
        int64_t next = cb;
        for (int64_t d = 0; d < n_docs; d++) {
            if (doc_off[d + 1] < doc_off[d]) return fail(JTK_ERR_INVALID_ARGUMENT, "doc_off must be non-decreasing");
            if (doc_off[d] >= next && d > b->chunk_doc.back() && n_bytes - doc_off[d] > cb / 4) {
                b->chunk_doc.push_back(d);
                b->chunk_off.push_back(doc_off[d]);
                next = doc_off[d] + cb;
            }
        }
    }
    b->chunk_doc.push_back(n_docs);
    b->chunk_off.push_back(n_bytes);
    HIP_TRY(hipSetDevice(b->enc->device));
    int rc;
    if ((rc = b->in_text.ensure((size_t)n_bytes + 64)) || (rc = b->in_off.ensure(((size_t)n_docs + 1) * 8)) ||
        (rc = b->in_pieces.ensure((size_t)(n_pieces > 0 ? n_pieces : 1) * 16)))
        return rc;
    HIP_TRY(hipMemcpyAsync(b->in_off.p, doc_off, ((size_t)n_docs + 1) * 8, hipMemcpyHostToDevice, b->stream));
    int64_t* d_begin = (int64_t*)b->in_pieces.p;
    int64_t* d_end = d_begin + (n_pieces > 0 ? n_pieces : 1);
    if (n_pieces > 0) {
        HIP_TRY(hipMemcpyAsync(d_begin, piece_begin, (size_t)n_pieces * 8, hipMemcpyHostToDevice, b->stream));
        HIP_TRY(hipMemcpyAsync(d_end, piece_end, (size_t)n_pieces * 8, hipMemcpyHostToDevice, b->stream));
    }
    const PieceArgs pa{d_begin, d_end, piece_begin, n_pieces};
    // the special-token check of encode() (GptBytePairEncoding.java:52-56, text.contains) rides in pretok_split on the device;
    // with caller-supplied pieces that kernel does not run, so it is done here on the host
    std::vector<int64_t> special_docs;
    if (!(flags & JTK_ENCODE_ORDINARY)) {
        std::vector<uint8_t> flag;
        find_special_docs(b->enc, utf8, n_bytes, doc_off, n_docs, flag);
        for (int64_t d = 0; d < n_docs; d++) if (flag[(size_t)d]) special_docs.push_back(d);
    }
    rc = run_job(b, (const uint8_t*)b->in_text.p, utf8, (const int64_t*)b->in_off.p, n_docs, n_bytes,
                 (flags | JTK_ENCODE_ORDINARY) & ~(uint32_t)JTK_ENCODE_TO_HOST, b->stream, (flags & JTK_ENCODE_TO_HOST) != 0, &pa);
    if (rc != JTK_OK) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (!special_docs.empty()) {
        const int32_t st = JTK_ERR_UNSUPPORTED_SPECIAL;
        for (int64_t d : special_docs) {
            HIP_TRY(hipMemcpy((int32_t*)b->status.p + d, &st, 4, hipMemcpyDefault));
            if (b->have_host_result) b->r_status[d] = st;
        }
        if (b->host_result->worst_status > st) b->host_result->worst_status = st;
    }
    b->synced = true;
    if (n_tokens) *n_tokens = b->host_result->n_tokens;
    return JTK_OK;
}

void* jtk_batch_stream(jtk_batch* b) { return b ? (void*)b->stream : nullptr; }

int jtk_batch_result(jtk_batch* b, int64_t* n_tokens, int64_t* n_docs, int32_t* worst_status) {
    if (!b || !b->have_result) return fail(JTK_ERR_INVALID_ARGUMENT, "no encode has run on this batch");
    HIP_TRY(hipSetDevice(b->enc->device));
    if (!b->synced) { HIP_TRY(hipStreamSynchronize(b->last_stream)); b->synced = true; }
    if (n_tokens) *n_tokens = b->host_result->n_tokens;
    if (n_docs) *n_docs = b->job_docs;
    if (worst_status) *worst_status = b->host_result->worst_status;
    return JTK_OK;
}

int jtk_batch_fetch(jtk_batch* b, int32_t* tokens, int64_t tokens_cap, int64_t* tok_off, int32_t* status) {
    int64_t nt = 0;
    int rc = jtk_batch_result(b, &nt, nullptr, nullptr);
    if (rc != JTK_OK) return rc;
    const bool count_only = (b->job_flags & JTK_ENCODE_COUNT_ONLY) != 0;
    if (tokens) {
        if (count_only) return fail(JTK_ERR_INVALID_ARGUMENT, "the last encode was count-only: there are no token ids");
        if (tokens_cap < nt) return fail(JTK_ERR_CAPACITY, "tokens buffer too small");
        if (nt > 0) {
            if (b->have_host_result) memcpy(tokens, b->r_tokens, (size_t)nt * 4);
            else HIP_TRY(hipMemcpy(tokens, b->tokens.p, (size_t)nt * 4, hipMemcpyDefault));
        }
    }
    if (tok_off) HIP_TRY(hipMemcpy(tok_off, b->tok_off.p, ((size_t)b->job_docs + 1) * 8, hipMemcpyDefault));
    if (status && b->job_docs > 0)
        HIP_TRY(hipMemcpy(status, b->status.p, (size_t)b->job_docs * 4, hipMemcpyDefault));
    return JTK_OK;
}

int jtk_batch_host_result(jtk_batch* b, const int32_t** tokens, const int64_t** tok_off, const int32_t** status) {
    if (!b || !b->have_result || !b->have_host_result)
        return fail(JTK_ERR_INVALID_ARGUMENT, "the last encode on this batch did not run with JTK_ENCODE_TO_HOST");
    HIP_TRY(hipSetDevice(b->enc->device));
    if (!b->synced) { HIP_TRY(hipStreamSynchronize(b->last_stream)); b->synced = true; }
    if (tokens) *tokens = (b->job_flags & JTK_ENCODE_COUNT_ONLY) ? nullptr : b->r_tokens;
    if (tok_off) *tok_off = b->r_tok_off;
    if (status) *status = b->r_status;
    return JTK_OK;
}

int jtk_batch_device_result(jtk_batch* b, const int32_t** d_tokens, const int64_t** d_tok_off,
                            const int32_t** d_status) {
    if (!b || !b->have_result) return fail(JTK_ERR_INVALID_ARGUMENT, "no encode has run on this batch");
    if (d_tokens) *d_tokens = (b->job_flags & JTK_ENCODE_COUNT_ONLY) ? nullptr : (const int32_t*)b->tokens.p;
    if (d_tok_off) *d_tok_off = (const int64_t*)b->tok_off.p;
    if (d_status) *d_status = (const int32_t*)b->status.p;
    return JTK_OK;
}

int jtk_batch_kernel_times(jtk_batch* b, const char** names, float* ms, int cap, int* n) {
    if (!b || !n) return fail(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    *n = 0;
    if (b->prof_chunks <= 0) return fail(JTK_ERR_INVALID_ARGUMENT, "profiling was not enabled for the last encode");
    HIP_TRY(hipSetDevice(b->enc->device));
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    for (int i = 0; i < N_STAGES && i < cap; i++) {
        float sum = 0.f;
        for (int c = 0; c < b->prof_chunks; c++) {
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, b->prof_ev[((size_t)c * N_STAGES + i) * 2], b->prof_ev[((size_t)c * N_STAGES + i) * 2 + 1]));
            sum += t;
        }
        if (names) names[i] = STAGE_NAMES[i];
        if (ms) ms[i] = sum;
        *n = i + 1;
    }
    return JTK_OK;
}

// ---- maxTokens on the device ---------------------------------------------------------------------------
int jtk_batch_truncate(jtk_batch* b, int64_t max_tokens) {
    if (!b || !b->have_result || max_tokens < 0) return fail(JTK_ERR_INVALID_ARGUMENT, "bad arguments (an encode must have run on this batch)");
    HIP_TRY(hipSetDevice(b->enc->device));
    if (!b->synced) { HIP_TRY(hipStreamSynchronize(b->last_stream)); b->synced = true; }
    const int64_t nd = b->job_docs;
    int rc;
    if ((rc = b->trunc_kept.ensure((size_t)(nd > 0 ? nd : 1) * 8)) || (rc = b->trunc_flag.ensure((size_t)(nd > 0 ? nd : 1)))) return rc;
    JtkTruncWork t{};
    t.tokens = (const int32_t*)b->tokens.p; t.tok_off = (const int64_t*)b->tok_off.p; t.text = b->job_text; t.doc_off = b->job_doc_off;
    t.n_docs = nd; t.tab_off = (const uint32_t*)b->enc->dec_off.p; t.max_tokens = max_tokens;
    t.kept = (int64_t*)b->trunc_kept.p; t.truncated = (uint8_t*)b->trunc_flag.p;
    jtk_launch_truncate(t, b->last_stream);
    HIP_TRY(hipGetLastError());
    b->have_trunc = true;
    return JTK_OK;
}

int jtk_batch_fetch_truncated(jtk_batch* b, int64_t* kept, uint8_t* truncated) {
    if (!b || !b->have_trunc) return fail(JTK_ERR_INVALID_ARGUMENT, "jtk_batch_truncate has not run on this batch");
    HIP_TRY(hipSetDevice(b->enc->device));
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    const size_t nd = (size_t)b->job_docs;
    if (kept && nd) HIP_TRY(hipMemcpy(kept, b->trunc_kept.p, nd * 8, hipMemcpyDeviceToHost));
    if (truncated && nd) HIP_TRY(hipMemcpy(truncated, b->trunc_flag.p, nd, hipMemcpyDeviceToHost));
    return JTK_OK;
}

int jtk_batch_device_truncated(jtk_batch* b, const int64_t** d_kept, const uint8_t** d_truncated) {
    if (!b || !b->have_trunc) return fail(JTK_ERR_INVALID_ARGUMENT, "jtk_batch_truncate has not run on this batch");
    if (d_kept) *d_kept = (const int64_t*)b->trunc_kept.p;
    if (d_truncated) *d_truncated = (const uint8_t*)b->trunc_flag.p;
    return JTK_OK;
}

// ---- batch decode ------------------------------------------------------------------------------------
int jtk_batch_decode_device(jtk_batch* b, const int32_t* d_ids, const int64_t* d_seq_off, int64_t n_seqs, int64_t n_ids,
                            void* stream_or_null, int64_t* n_bytes) {
    if (!b || n_seqs < 0 || n_ids < 0 || (n_ids > 0 && !d_ids) || !d_seq_off) return fail(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    const jtk_encoding* enc = b->enc;
    HIP_TRY(hipSetDevice(enc->device));
    hipStream_t s = stream_or_null ? (hipStream_t)stream_or_null : b->stream;
    JtkDecodeWork& w = b->dwork;
    w.ids = d_ids; w.seq_off = d_seq_off; w.n_tok = n_ids; w.n_seqs = n_seqs;
    w.n_tiles = (n_ids + JTK_DEC_TILE - 1) / JTK_DEC_TILE;
    if (w.n_tiles < 1) w.n_tiles = 1;
    w.tab_off = (const uint32_t*)enc->dec_off.p; w.tab_blob = (const uint8_t*)enc->dec_blob.p; w.n_ids_table = enc->n_ids_table;
    const size_t nt = (size_t)w.n_tiles;
    const size_t mask_bytes = (nt * (JTK_DEC_TILE / 64) + 2) * 8;
    const size_t status_bytes = align_up((size_t)(n_seqs > 0 ? n_seqs : 1) * 4, 16);
    const size_t zero_bytes = mask_bytes + status_bytes + 16;
    int rc;
    if ((rc = b->dec_zero.ensure(zero_bytes)) || (rc = b->dec_tile.ensure(nt * 4 + (nt + 1) * 8 + 16)) ||
        (rc = b->dec_pre.ensure(nt * JTK_DEC_TILE * 4)) || (rc = b->dec_byte_off.ensure(((size_t)n_seqs + 1) * 8)))
        return rc;
    uint8_t* z = (uint8_t*)b->dec_zero.p;
    w.seqmask = (uint64_t*)z;
    w.status = (int32_t*)(z + mask_bytes);
    w.total = (int64_t*)(z + mask_bytes + status_bytes);
    w.worst_status = (int32_t*)(z + mask_bytes + status_bytes + 8);
    w.tile_off = (int64_t*)b->dec_tile.p;
    w.tile_bytes = (uint32_t*)((uint8_t*)b->dec_tile.p + (nt + 1) * 8);
    w.seqpre = (uint32_t*)b->dec_pre.p;
    w.byte_off = (int64_t*)b->dec_byte_off.p;
    w.out = nullptr;
    // phase 1: sizes (the output is allocated once they are known)
    HIP_TRY(hipMemsetAsync(z, 0, zero_bytes, s));
    jtk_launch_decode_count(w, s);
    int64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, w.total, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if ((rc = b->dec_out.ensure((size_t)total + 64))) return rc;
    w.out = (uint8_t*)b->dec_out.p;
    // phase 2: bytes and per-sequence offsets
    jtk_launch_decode_scatter(w, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    b->dec_total = total;
    b->have_decode = true;
    if (n_bytes) *n_bytes = total;
    return JTK_OK;
}

int jtk_batch_decode(jtk_batch* b, const int32_t* ids, const int64_t* seq_off, int64_t n_seqs, int64_t* n_bytes) {
    if (!b || n_seqs < 0 || !seq_off) return fail(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    if (seq_off[0] != 0) return fail(JTK_ERR_INVALID_ARGUMENT, "seq_off[0] must be 0");
    for (int64_t q = 0; q < n_seqs; q++)
        if (seq_off[q + 1] < seq_off[q]) return fail(JTK_ERR_INVALID_ARGUMENT, "seq_off must be non-decreasing");
    const int64_t n_ids = seq_off[n_seqs];
    if (n_ids > 0 && !ids) return fail(JTK_ERR_INVALID_ARGUMENT, "ids is NULL");
    HIP_TRY(hipSetDevice(b->enc->device));
    int rc;
    if ((rc = b->dec_in_ids.ensure((size_t)n_ids * 4 + 64)) || (rc = b->dec_in_off.ensure(((size_t)n_seqs + 1) * 8))) return rc;
    if (n_ids > 0) HIP_TRY(hipMemcpyAsync(b->dec_in_ids.p, ids, (size_t)n_ids * 4, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(b->dec_in_off.p, seq_off, ((size_t)n_seqs + 1) * 8, hipMemcpyHostToDevice, b->stream));
    return jtk_batch_decode_device(b, (const int32_t*)b->dec_in_ids.p, (const int64_t*)b->dec_in_off.p, n_seqs, n_ids, nullptr, n_bytes);
}

int jtk_batch_decode_fetch(jtk_batch* b, uint8_t* out, int64_t out_cap, int64_t* byte_off, int32_t* status) {
    if (!b || !b->have_decode) return fail(JTK_ERR_INVALID_ARGUMENT, "no decode has run on this batch");
    HIP_TRY(hipSetDevice(b->enc->device));
    if (out) {
        if (out_cap < b->dec_total) return fail(JTK_ERR_CAPACITY, "output buffer too small");
        if (b->dec_total > 0) HIP_TRY(hipMemcpy(out, b->dwork.out, (size_t)b->dec_total, hipMemcpyDeviceToHost));
    }
    if (byte_off) HIP_TRY(hipMemcpy(byte_off, b->dwork.byte_off, ((size_t)b->dwork.n_seqs + 1) * 8, hipMemcpyDeviceToHost));
    if (status && b->dwork.n_seqs > 0) HIP_TRY(hipMemcpy(status, b->dwork.status, (size_t)b->dwork.n_seqs * 4, hipMemcpyDeviceToHost));
    return JTK_OK;
}

int jtk_batch_decode_device_result(jtk_batch* b, const uint8_t** d_out, const int64_t** d_byte_off, const int32_t** d_status) {
    if (!b || !b->have_decode) return fail(JTK_ERR_INVALID_ARGUMENT, "no decode has run on this batch");
    if (d_out) *d_out = b->dwork.out;
    if (d_byte_off) *d_byte_off = b->dwork.byte_off;
    if (d_status) *d_status = b->dwork.status;
    return JTK_OK;
}

int jtk_decode(const jtk_encoding* enc, const int32_t* ids, int64_t n, uint8_t* out, int64_t cap, int64_t* len) {
    if (!enc || n < 0 || (n > 0 && !ids)) return fail(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    const int64_t r = jtk_host_decode(enc->host, ids, n, out, cap);
    if (r == JTK_ERR_UNKNOWN_TOKEN) return fail(JTK_ERR_UNKNOWN_TOKEN, "Unknown token for decoding");
    if (r < 0) return fail((int)r, "decode buffer too small");
    if (len) *len = r;
    return JTK_OK;
}

}  // extern "C"

// Encoding.encode(text, maxTokens) from the full token list of `text` (GptBytePairEncoding.java:90-100): the first
// min(maxTokens, nt) tokens, backed off until decode(tokens) is a prefix of the text.  head: at least that many leading
// tokens.  Returns the count kept; *truncated = EncodingResult.isTruncated().
int64_t jtk_max_tokens_backoff(const jtk_encoding* enc, const uint8_t* utf8, int64_t len, const int32_t* head, int64_t nt,
                               int64_t max_tokens, int* truncated) {
    int64_t keep = nt < max_tokens ? nt : max_tokens;
    static thread_local std::vector<int64_t> cum;
    cum.assign((size_t)keep + 1, 0);
    for (int64_t k = 0; k < keep; k++) cum[(size_t)k + 1] = cum[(size_t)k] + enc->tok_len[(size_t)head[(size_t)k]];
    if (truncated) *truncated = 0;
    // does text[from:] hold more than `k` UTF-16 units?  (looks at the first few bytes only: the document may be long)
    auto more_units_than = [&](int64_t from, int64_t k) {
        int64_t u = 0;
        for (int64_t i = from; i < len; i++)
            if ((utf8[i] & 0xC0) != 0x80) { u += (utf8[i] >= 0xF0) ? 2 : 1; if (u > k) return true; }
        return false;
    };
    for (;; keep--) {
        // decode(tokens) is the byte prefix [0, nb) of the text.  text.startsWith(decoded) holds when
        // nb is a code-point boundary, or when the cut character decodes to one U+FFFD and the text
        // has U+FFFD there.  truncated = text.length() > decoded.length(), both in UTF-16 units: the units before the
        // cut are common to both, so only the text from the cut on is counted.
        const int64_t nb = cum[(size_t)keep];
        const bool boundary = (nb == len) || ((utf8[nb] & 0xC0) != 0x80);
        bool starts, longer;
        if (boundary) { starts = true; longer = more_units_than(nb, 0); }
        else {
            int64_t c = nb;
            while (c > 0 && (utf8[c] & 0xC0) == 0x80) c--;
            starts = (c + 2 < len) && utf8[c] == 0xEF && utf8[c + 1] == 0xBF && utf8[c + 2] == 0xBD;
            longer = more_units_than(c, 1);                       // the decoded text ends in one U+FFFD for the cut character
        }
        if (starts) {
            if (truncated) *truncated = longer;
            break;
        }
        if (keep == 0) break;
    }
    return keep;
}

extern "C" {

int jtk_encode(jtk_batch* b, const uint8_t* utf8, int64_t len, uint32_t flags, int64_t max_tokens,
               int32_t* tokens, int64_t tokens_cap, int64_t* n_tokens, int* truncated) {
    if (!b || len < 0) return fail(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    if (truncated) *truncated = 0;
    if (n_tokens) *n_tokens = 0;
    if (!utf8) return JTK_OK;                                    // text == null -> empty result
    if (max_tokens >= 0) {
        // encode(text, maxTokens): the leading bytes only (jtk_batch_encode_max_tokens below)
        const int64_t off2[2] = {0, len};
        std::vector<int32_t> head((size_t)((max_tokens < len ? max_tokens : len) + 1));
        int64_t keep = 0; uint8_t tr = 0; int32_t st = 0;
        int rc = jtk_batch_encode_max_tokens(b, utf8, off2, 1, flags & JTK_ENCODE_ORDINARY, max_tokens, head.data(), &keep, &tr, &st);
        if (rc != JTK_OK) return rc;
        if (st == JTK_ERR_UNSUPPORTED_SPECIAL) return fail(st, "Encoding special tokens is not supported yet.");
        if (st == JTK_ERR_UNENCODABLE) return fail(st, "Unknown token for encoding: the rank map lacks a single-byte token this text needs");
        if (st != JTK_OK) return fail(st, "document could not be encoded");
        if (truncated) *truncated = tr;
        if (n_tokens) *n_tokens = keep;
        if (tokens) {
            if (tokens_cap < keep) return fail(JTK_ERR_CAPACITY, "tokens buffer too small");
            if (keep > 0) memcpy(tokens, head.data(), (size_t)keep * 4);
        }
        return JTK_OK;
    }
    const int64_t off[2] = {0, len};
    int64_t nt = 0;
    // (the result comes back with the job -- status and ids in pinned host memory -- instead of two more copies behind it)
    int rc = jtk_batch_encode(b, utf8, off, 1, (flags & ~(uint32_t)JTK_ENCODE_COUNT_ONLY) | JTK_ENCODE_TO_HOST, &nt);
    if (rc != JTK_OK) return rc;
    const int32_t st = b->r_status[0];
    if (st == JTK_ERR_UNSUPPORTED_SPECIAL) return fail(st, "Encoding special tokens is not supported yet.");
    if (st == JTK_ERR_UNENCODABLE) return fail(st, "Unknown token for encoding: the rank map lacks a single-byte token this text needs");
    if (st != JTK_OK) return fail(st, "document could not be encoded");
    if (n_tokens) *n_tokens = nt;
    if (tokens) {
        if (tokens_cap < nt) return fail(JTK_ERR_CAPACITY, "tokens buffer too small");
        if (nt > 0) memcpy(tokens, b->r_tokens, (size_t)nt * 4);
    }
    return JTK_OK;
}

// A byte that may begin a white-space character: the ASCII ones, and the lead bytes of U+0085/U+00A0 (C2), U+1680 (E1),
// U+2000..U+205F (E2) and U+3000 (E3).  Conservative on purpose: it only ever makes the early exit below look further.
static inline bool maybe_space(uint8_t c) { return (c >= 0x09 && c <= 0x0D) || c == 0x20 || c == 0xC2 || c == 0xE1 || c == 0xE2 || c == 0xE3; }

// Encoding.encode(text, maxTokens) for every document, without encoding the documents whole.  The reference stops matching
// once maxTokens tokens exist (GptBytePairEncoding.java:83-88); here each document's leading P bytes are encoded (P = 8 bytes
// per wanted token + 64 to begin with), and the result is taken when it is certain to be the head of the document's full token
// list: that is the tokens before a piece start q that (a) lies at least 16 bytes before the cut -- every look-ahead of the
// patterns (a contraction, the character after a white-space run) is shorter -- and (b) does not sit inside a white-space run
// that might reach the cut (`\s*[\r\n]+` and `\s+(?!\S)` look to the END of the run): text[q] is no white space, or it is one
// ASCII white-space character followed by something else.  Pieces before q are then matched exactly as in the whole text and
// pieces encode independently.  Documents whose prefix holds fewer than maxTokens such tokens go round again with 4x the bytes.
int jtk_batch_encode_max_tokens(jtk_batch* b, const uint8_t* utf8, const int64_t* doc_off, int64_t n_docs, uint32_t flags,
                                int64_t max_tokens, int32_t* tokens, int64_t* kept, uint8_t* truncated, int32_t* status) {
    if (!b || n_docs < 0 || !doc_off || max_tokens < 0 || !kept || (max_tokens > 0 && n_docs > 0 && !tokens))
        return fail(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    if (doc_off[0] != 0) return fail(JTK_ERR_INVALID_ARGUMENT, "doc_off[0] must be 0");
    for (int64_t d = 0; d < n_docs; d++)
        if (doc_off[d + 1] < doc_off[d]) return fail(JTK_ERR_INVALID_ARGUMENT, "doc_off must be non-decreasing");
    const int64_t n_bytes = doc_off[n_docs];
    if (n_bytes > 0 && !utf8) return fail(JTK_ERR_INVALID_ARGUMENT, "utf8 is NULL");
    const jtk_encoding* enc = b->enc;
    std::vector<uint8_t> special((size_t)(n_docs > 0 ? n_docs : 1), 0);
    if (!(flags & JTK_ENCODE_ORDINARY)) find_special_docs(enc, utf8, n_bytes, doc_off, n_docs, special);   // text.contains(special) looks at the whole document (:52-56)
    std::vector<int64_t> active;
    active.reserve((size_t)n_docs);
    for (int64_t d = 0; d < n_docs; d++) {
        kept[d] = 0;
        if (truncated) truncated[d] = 0;
        if (status) status[d] = special[(size_t)d] ? JTK_ERR_UNSUPPORTED_SPECIAL : JTK_OK;
        if (special[(size_t)d]) continue;
        const int64_t len = doc_off[d + 1] - doc_off[d];
        if (len == 0 || max_tokens == 0) {
            int tr = 0;
            jtk_max_tokens_backoff(enc, utf8 + doc_off[d], len, nullptr, 0, 0, &tr);
            if (truncated) truncated[d] = (uint8_t)tr;
            continue;
        }
        active.push_back(d);
    }
    const int64_t margin = 16;
    const int64_t cb = b->host_chunk_bytes < b->chunk_bytes ? b->host_chunk_bytes : b->chunk_bytes;   // a group stays one chunk
    int64_t P = max_tokens > ((int64_t)1 << 40) ? (int64_t)1 << 44 : 8 * max_tokens + 64;
    uint8_t* gtext = nullptr;                                        // (pinned: the prefixes go down by DMA while the kernels start)
    std::vector<int64_t> goff, next_active;
    std::vector<uint64_t> mask;
    const bool trace = getenv("JTK_MAXTOK_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    double t_gather = 0, t_enc = 0, t_dec = 0, t_mask = 0;
    while (!active.empty()) {
        next_active.clear();
        size_t a0 = 0;
        while (a0 < active.size()) {
            // one group: as many prefixes as fit one chunk
            size_t a1 = a0;
            int64_t gbytes = 0;
            goff.assign(1, 0);
            while (a1 < active.size()) {
                const int64_t d = active[a1], len = doc_off[d + 1] - doc_off[d];
                int64_t p = len < P ? len : P;
                if (p > cb) p = len;                                 // past one host chunk the document goes whole, and alone
                if (a1 > a0 && gbytes + p > cb) break;
                gbytes += p;
                goff.push_back(gbytes);
                a1++;
            }
            double t0 = now();
            { const int prc = ensure_pinned((void**)&b->h_gather, &b->h_gather_cap, (size_t)gbytes + 64, 0); if (prc) return prc; }
            gtext = b->h_gather;
            host_slices(a1 - a0, [&](int, size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; i++)
                    memcpy(gtext + goff[i], utf8 + doc_off[active[a0 + i]], (size_t)(goff[i + 1] - goff[i]));
            });
            const int64_t ng = (int64_t)(a1 - a0);
            int64_t nt = 0;
            const int64_t save_host_chunk = b->host_chunk_bytes;
            if (gbytes > cb) b->host_chunk_bytes = gbytes;           // a single document above the host chunk size: still one chunk
            double t1 = now(); t_gather += t1 - t0;
            int rc = jtk_batch_encode(b, gtext, goff.data(), ng, JTK_ENCODE_ORDINARY | JTK_ENCODE_TO_HOST, &nt);
            b->host_chunk_bytes = save_host_chunk;
            double t2 = now(); t_enc += t2 - t1;
            if (rc != JTK_OK) return rc;
            if (b->chunk_doc.size() != 2) {
                // longer than a device chunk: no single piece mask; such a document is encoded whole
                const int32_t* toks = b->r_tokens; const int64_t* toff = b->r_tok_off;
                for (size_t a = a0; a < a1; a++) {
                    const int64_t d = active[a], len = doc_off[d + 1] - doc_off[d], i = (int64_t)(a - a0);
                    if (goff[i + 1] - goff[i] != len) return fail(JTK_ERR_INVALID_ARGUMENT, "max-tokens prefix spans device chunks");
                    int tr = 0;
                    const int64_t k = jtk_max_tokens_backoff(enc, utf8 + doc_off[d], len, toks + toff[i], toff[i + 1] - toff[i], max_tokens, &tr);
                    if (k > 0) memcpy(tokens + d * max_tokens, toks + toff[i], (size_t)k * 4);
                    kept[d] = k;
                    if (truncated) truncated[d] = (uint8_t)tr;
                }
                a0 = a1;
                continue;
            }
            const int32_t* toks = b->r_tokens; const int64_t* toff = b->r_tok_off; const int32_t* st = b->r_status;
            const size_t n_words = (size_t)((gbytes + 1 + 63) / 64);
            mask.resize(n_words);
            {
                double tm = now();
                HIP_TRY(hipMemcpy(mask.data(), b->set[0].piecemask.p, n_words * 8, hipMemcpyDeviceToHost));
                t_mask += now() - tm;
            }
            std::vector<int64_t> again[8];
            host_slices(a1 - a0, [&](int slice, size_t lo, size_t hi) {
                for (size_t ii = lo; ii < hi; ii++) {
                    const int64_t i = (int64_t)ii, d = active[a0 + ii], len = doc_off[d + 1] - doc_off[d];
                    const int64_t p = goff[i + 1] - goff[i], t0 = toff[i], n = toff[i + 1] - t0;
                    if (st[i] != JTK_OK) { if (status) status[d] = st[i]; continue; }
                    int64_t k = -1;
                    if (p == len) k = n;
                    else {
                        // the last safe piece start at or before p - margin
                        const uint8_t* t = gtext + goff[i];
                        int64_t q = 0;
                        for (int64_t pos = goff[i] + p - margin; pos > goff[i]; ) {
                            uint64_t w = mask[(size_t)(pos >> 6)];
                            const int sh = (int)(pos & 63);
                            w = sh == 63 ? w : (w & ((2ull << sh) - 1));                 // bits 0..sh
                            const int64_t wbase = pos & ~(int64_t)63;
                            bool found = false;
                            while (w) {
                                const int bit = 63 - __builtin_clzll(w);
                                const int64_t cand = wbase + bit;
                                if (cand <= goff[i]) { w = 0; break; }
                                const uint8_t c0 = t[cand - goff[i]], c1 = t[cand - goff[i] + 1];
                                if (!maybe_space(c0) || (c0 < 0x80 && !maybe_space(c1))) { q = cand - goff[i]; found = true; break; }
                                w &= ~(1ull << bit);
                            }
                            if (found) break;
                            pos = wbase - 1;
                        }
                        if (q > 0) {
                            int64_t cum = 0, kk = 0;
                            while (kk < n && kk < max_tokens && cum < q) cum += enc->tok_len[(size_t)toks[t0 + kk]], kk++;
                            if (cum <= q && kk >= max_tokens) k = kk;                   // maxTokens tokens, all before q
                        }
                    }
                    if (k < 0) { again[slice].push_back(d); continue; }
                    int tr = 0;
                    // (the back-off looks at the text around the cut only: that is inside the prefix, which is in the cache -- the
                    // document itself is a cold line per look)
                    const int64_t keep = jtk_max_tokens_backoff(enc, gtext + goff[i], len, toks + t0, k, max_tokens, &tr);
                    if (keep > 0) memcpy(tokens + d * max_tokens, toks + t0, (size_t)keep * 4);
                    kept[d] = keep;
                    if (truncated) truncated[d] = (uint8_t)tr;
                }
            });
            for (auto& v : again) next_active.insert(next_active.end(), v.begin(), v.end());
            t_dec += now() - t2;
            a0 = a1;
        }
        if (trace) fprintf(stderr, "[max_tokens] P=%lld docs=%zu -> %zu again; gather %.2f encode %.2f mask %.2f decide(incl. mask) %.2f ms; %.2f ms since the rounds began\n",
                           (long long)P, active.size(), next_active.size(), t_gather, t_enc, t_mask, t_dec, now() - t_begin);
        active.swap(next_active);
        P = P > ((int64_t)1 << 40) ? P : P * 4;
    }
    return JTK_OK;
}

}  // extern "C"
