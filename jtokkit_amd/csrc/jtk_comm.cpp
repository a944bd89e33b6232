// jtk_comm.cpp -- multi-GPU behind the C ABI: byte-balanced document shards and the one exchange step of the path, an
// RCCL all-gather of the per-shard token totals (ncclAllGather, 1 x int64 per rank, over xGMI) for the offset stitch.
// One process per GPU; documents are independent (reference GptBytePairEncoding.java:71-103 keeps no cross-call state), so
// there is no data-path collective.  RCCL is bound at run time (dlopen): single-GPU users never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/jtokkit_amd.h"
#include "jtk_kernels.h"

int jtk_fail_msg(int code, const std::string& msg);          // jtk_abi.cpp: sets jtk_last_error()

namespace {

// the few RCCL entry points used (declarations as in /opt/rocm/include/rccl/rccl.h)
typedef struct { char internal[128]; } RcclUniqueId;
typedef void* RcclComm;
typedef int (*fn_get_unique_id)(RcclUniqueId*);
typedef int (*fn_comm_init_rank)(RcclComm*, int, RcclUniqueId, int);
typedef int (*fn_comm_destroy)(RcclComm);
typedef int (*fn_all_gather)(const void*, void*, size_t, int /*ncclDataType_t*/, RcclComm, hipStream_t);
typedef const char* (*fn_error_string)(int);
constexpr int RCCL_INT64 = 4;                                 // ncclInt64

struct Rccl {
    void* h = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_error_string error_string = nullptr;
    std::string err;
} g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    // (JTK_RCCL_LIB names the library to use: then nothing else is tried)
    const char* forced = getenv("JTK_RCCL_LIB");
    const char* names[] = {forced, forced ? nullptr : "librccl.so", forced ? nullptr : "librccl.so.1"};
    // a copy that is already in the process (PyTorch-ROCm bundles one) first: one RCCL per process
    for (const char* n : names) if (n && !g_rccl.h) g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names) if (n && !g_rccl.h) g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!g_rccl.h) {
        const char* e = dlerror();                                    // (one call: it clears the error it returns)
        g_rccl.err = std::string("RCCL not found: ") + (e ? e : "librccl.so");
        return;
    }
    g_rccl.get_unique_id = (fn_get_unique_id)dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.comm_destroy = (fn_comm_destroy)dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.all_gather = (fn_all_gather)dlsym(g_rccl.h, "ncclAllGather");
    g_rccl.error_string = (fn_error_string)dlsym(g_rccl.h, "ncclGetErrorString");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_gather) {
        g_rccl.err = "RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
        g_rccl.h = nullptr;
    }
}

int rccl_ready() {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.h) return jtk_fail_msg(JTK_ERR_HIP, g_rccl.err);
    return JTK_OK;
}

int rccl_fail(const char* what, int rc) {
    return jtk_fail_msg(JTK_ERR_HIP, std::string(what) + ": " + (g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error"));
}

}  // namespace

struct jtk_comm {
    RcclComm comm = nullptr;
    int world = 1, rank = 0, device = 0;
    int64_t* d_mine = nullptr;       // [1] this rank's token total
    int64_t* d_totals = nullptr;     // [world] every rank's
    int64_t* d_base = nullptr;       // [1] exclusive prefix: this shard's first global token
    int64_t* h_out = nullptr;        // pinned [world + 1]: totals, base
};

extern "C" {

int jtk_shard_plan(const int64_t* doc_off, int64_t n_docs, int world, int64_t* bounds) {
    if (!doc_off || !bounds || n_docs < 0 || world < 1) return jtk_fail_msg(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    // contiguous document ranges balanced by bytes: rank r starts at the first document at or after byte r * total / world
    const int64_t total = doc_off[n_docs] - doc_off[0];
    bounds[0] = 0;
    int64_t d = 0;
    for (int r = 1; r < world; r++) {
        const int64_t target = doc_off[0] + (int64_t)((__int128)total * r / world);
        int64_t lo = d, hi = n_docs;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (doc_off[mid] >= target) hi = mid; else lo = mid + 1; }
        d = lo;
        bounds[r] = d;
    }
    bounds[world] = n_docs;
    return JTK_OK;
}

int jtk_comm_unique_id(uint8_t* id128) {
    if (!id128) return jtk_fail_msg(JTK_ERR_INVALID_ARGUMENT, "id is NULL");
    int rc = rccl_ready();
    if (rc != JTK_OK) return rc;
    RcclUniqueId id;
    const int r = g_rccl.get_unique_id(&id);
    if (r != 0) return rccl_fail("ncclGetUniqueId", r);
    memcpy(id128, id.internal, 128);
    return JTK_OK;
}

int jtk_comm_create(const uint8_t* id128, int world, int rank, int device, jtk_comm** out) {
    if (!out) return jtk_fail_msg(JTK_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!id128 || world < 1 || rank < 0 || rank >= world) return jtk_fail_msg(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    int rc = rccl_ready();
    if (rc != JTK_OK) return rc;
    if (hipSetDevice(device) != hipSuccess) return jtk_fail_msg(JTK_ERR_NO_DEVICE, "hipSetDevice failed");
    jtk_comm* c = new (std::nothrow) jtk_comm();
    if (!c) return jtk_fail_msg(JTK_ERR_OUT_OF_MEMORY, "out of host memory");
    c->world = world; c->rank = rank; c->device = device;
    RcclUniqueId id;
    memcpy(id.internal, id128, 128);
    const int r = g_rccl.comm_init_rank(&c->comm, world, id, rank);
    if (r != 0) { delete c; return rccl_fail("ncclCommInitRank", r); }
    hipError_t e = hipMalloc((void**)&c->d_mine, 8);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_totals, (size_t)world * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_base, 8);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_out, ((size_t)world + 1) * 8, hipHostMallocDefault);
    if (e != hipSuccess) { jtk_comm_destroy(c); return jtk_fail_msg(JTK_ERR_OUT_OF_MEMORY, std::string("comm buffers: ") + hipGetErrorString(e)); }
    *out = c;
    return JTK_OK;
}

void jtk_comm_destroy(jtk_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(c->comm);
    if (c->d_mine) (void)hipFree(c->d_mine);
    if (c->d_totals) (void)hipFree(c->d_totals);
    if (c->d_base) (void)hipFree(c->d_base);
    if (c->h_out) (void)hipHostFree(c->h_out);
    delete c;
}

int jtk_comm_world(const jtk_comm* c) { return c ? c->world : 0; }
int jtk_comm_rank(const jtk_comm* c) { return c ? c->rank : -1; }

int jtk_comm_stitch(jtk_comm* c, const int64_t* d_tok_off, int64_t n_docs, int64_t* d_global_off, void* stream,
                    const int64_t** d_totals, const int64_t** d_base) {
    if (!c || !d_tok_off || n_docs < 0) return jtk_fail_msg(JTK_ERR_INVALID_ARGUMENT, "bad arguments");
    if (hipSetDevice(c->device) != hipSuccess) return jtk_fail_msg(JTK_ERR_NO_DEVICE, "hipSetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    // this shard's token total is the last entry of its token offsets: device to device, no host round trip
    hipError_t e = hipMemcpyAsync(c->d_mine, d_tok_off + n_docs, 8, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return jtk_fail_msg(JTK_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    const int r = g_rccl.all_gather(c->d_mine, c->d_totals, 1, RCCL_INT64, c->comm, s);
    if (r != 0) return rccl_fail("ncclAllGather", r);
    jtk_launch_stitch(c->d_totals, c->rank, c->d_base, d_tok_off, n_docs, d_global_off, s);
    e = hipGetLastError();
    if (e != hipSuccess) return jtk_fail_msg(JTK_ERR_HIP, std::string("stitch: ") + hipGetErrorString(e));
    if (d_totals) *d_totals = c->d_totals;
    if (d_base) *d_base = c->d_base;
    return JTK_OK;
}

int jtk_comm_fetch(jtk_comm* c, void* stream, int64_t* totals, int64_t* base) {
    if (!c) return jtk_fail_msg(JTK_ERR_INVALID_ARGUMENT, "comm is NULL");
    if (hipSetDevice(c->device) != hipSuccess) return jtk_fail_msg(JTK_ERR_NO_DEVICE, "hipSetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(c->h_out, c->d_totals, (size_t)c->world * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_out + c->world, c->d_base, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return jtk_fail_msg(JTK_ERR_HIP, std::string("comm fetch: ") + hipGetErrorString(e));
    if (totals) memcpy(totals, c->h_out, (size_t)c->world * 8);
    if (base) *base = c->h_out[c->world];
    return JTK_OK;
}

}  // extern "C"
