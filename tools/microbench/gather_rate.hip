// gather_rate.hip -- how many scattered (one cache line per lane) loads per clock does a CU sustain?
// Each lane issues UNROLL independent loads per iteration from a table of `mb` MB at hashed indices.
// build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip ; run: ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <class V, int UNROLL>
__global__ void __launch_bounds__(256) k_gather(const V* tab, uint32_t n, int iters, uint32_t* out, int same_line) {
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 0x9E3779B1u + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        V v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            h = h * 0x2C1B3C6Du + 0x165667B1u;
            uint32_t x = h ^ (h >> 15);
            uint32_t idx = __umulhi(x, n);
            if (same_line) idx = (idx & ~63u) + (threadIdx.x & 63u);      // a wave reads 64 consecutive elements
            v[u] = tab[idx];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += *reinterpret_cast<uint32_t*>(&v[u]);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int UNROLL>
__global__ void __launch_bounds__(256) k_lds(int iters, uint32_t* out) {
    __shared__ uint32_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) s[i] = i * 2654435761u;
    __syncthreads();
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 0x9E3779B1u + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            h = h * 0x2C1B3C6Du + 0x165667B1u;
            acc += s[(h >> 19) & 8191u];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class V>
void run(const char* name, size_t mb, int same_line) {
    const size_t n = mb * 1024 * 1024 / sizeof(V);
    V* tab; uint32_t* out;
    hipMalloc(&tab, n * sizeof(V)); hipMalloc(&out, 4);
    hipMemset(tab, 1, n * sizeof(V));
    const int iters = 200, blocks = 256 * 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_gather<V, 4><<<blocks, 256>>>(tab, (uint32_t)n, 10, out, same_line);
    hipEventRecord(a);
    k_gather<V, 4><<<blocks, 256>>>(tab, (uint32_t)n, iters, out, same_line);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double loads = (double)blocks * 256 * iters * 4;
    printf("%-28s table %4zu MB: %.3f ms, %.1f G lane-loads/s, %.2f lane-loads/clk/CU (2.4 GHz, 256 CUs)\n", name, mb, ms,
           loads / ms / 1e6, loads / (ms * 1e-3) / 2.4e9 / 256);
    hipFree(tab); hipFree(out);
}

int main() {
    run<uint4>("16 B scattered", 3, 0);
    run<uint4>("16 B scattered", 64, 0);
    run<uint32_t>("4 B scattered", 1, 0);
    run<uint32_t>("4 B scattered (256 KB)", 0 + 1, 0);
    run<uint4>("16 B, wave-contiguous", 3, 1);
    run<uint32_t>("4 B, wave-contiguous", 3, 1);
    uint32_t* out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_lds<4><<<1024, 256>>>(10, out);
    hipEventRecord(a);
    k_lds<4><<<1024, 256>>>(2000, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double loads = 1024.0 * 256 * 2000 * 4;
    printf("LDS random 4 B reads: %.3f ms, %.2f lane-reads/clk/CU\n", ms, loads / (ms * 1e-3) / 2.4e9 / 256);
    return 0;
}
