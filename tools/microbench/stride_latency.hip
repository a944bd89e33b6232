// Microbenchmark: what does a wave-per-tile gather kernel cost on MI355X as a function of the tile stride
// (dense 2 KB vs one 2 KB head per 8 KB block), the loads in flight per lane and the dependent levels?
// Build: hipcc --offload-arch=gfx950 -O3 -o stride_latency stride_latency.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int LOADS, int LEVELS>
__global__ void __launch_bounds__(64) k_gather(const uint32_t* a, const uint32_t* b, uint32_t* out, size_t stride_words, int tiles) {
    const int lane = threadIdx.x;
    const size_t tile = blockIdx.x;
    const uint32_t* p = a + tile * stride_words;
    uint32_t v[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; j++) v[j] = p[j * 64 + lane];
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < LOADS; j++) s += v[j];
    if (LEVELS >= 2) {
        const uint32_t* q = b + ((tile * 2654435761u + (s & 1u)) % (size_t)tiles) * stride_words;   // depends on level 1
        s += q[lane];
    }
    if (LEVELS >= 3) {
        const uint32_t* q = b + ((tile * 40503u + (s & 1u) + 17u) % (size_t)tiles) * stride_words;
        s += q[64 + lane];
    }
    out[tile * 64 + lane] = s;
}

// STORES store instructions per wave, 4 B per lane, at an unaligned base; SPARSE: only every 13th lane stores
template <int STORES, bool SPARSE>
__global__ void __launch_bounds__(64) k_scatter(uint32_t* out, int tiles) {
    const int lane = threadIdx.x;
    uint32_t* p = out + (size_t)blockIdx.x * (64 * STORES) + 3;
#pragma unroll
    for (int j = 0; j < STORES; j++) {
        if (!SPARSE || (lane % 13) == 0) p[j * 64 + lane] = (uint32_t)j;
    }
}

template <int STORES, bool SPARSE>
float run_st(uint32_t* out, int tiles) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_scatter<STORES, SPARSE>), dim3(tiles), dim3(64), 0, 0, out, tiles);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((k_scatter<STORES, SPARSE>), dim3(tiles), dim3(64), 0, 0, out, tiles);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 10;
}

template <int LOADS, int LEVELS>
float run(const uint32_t* a, const uint32_t* b, uint32_t* out, size_t stride_words, int tiles) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_gather<LOADS, LEVELS>), dim3(tiles), dim3(64), 0, 0, a, b, out, stride_words, tiles);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((k_gather<LOADS, LEVELS>), dim3(tiles), dim3(64), 0, 0, a, b, out, stride_words, tiles);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 10;
}

int main() {
    const int tiles = 50000;
    const size_t max_stride = 4096;     // words (16 KB)
    uint32_t *a, *b, *out;
    CK(hipMalloc(&a, tiles * max_stride * 4)); CK(hipMalloc(&b, tiles * max_stride * 4)); CK(hipMalloc(&out, (size_t)tiles * 64 * 4));
    CK(hipMemset(a, 1, tiles * max_stride * 4)); CK(hipMemset(b, 1, tiles * max_stride * 4));
    const size_t strides[] = {512, 2048, 4096};
    for (size_t st : strides) {
        printf("stride %5zu B: loads/lane=2 L1 %.4f ms | 8 L1 %.4f | 2 L2 %.4f | 8 L2 %.4f | 2 L3 %.4f | 8 L3 %.4f\n", st * 4,
               run<2, 1>(a, b, out, st, tiles), run<8, 1>(a, b, out, st, tiles), run<2, 2>(a, b, out, st, tiles),
               run<8, 2>(a, b, out, st, tiles), run<2, 3>(a, b, out, st, tiles), run<8, 3>(a, b, out, st, tiles));
    }
    uint32_t* big;
    CK(hipMalloc(&big, (size_t)tiles * 64 * 32 * 4 + 64));
    printf("stores/wave dense : 1 %.4f | 8 %.4f | 16 %.4f | 32 %.4f ms\n", run_st<1, false>(big, tiles), run_st<8, false>(big, tiles), run_st<16, false>(big, tiles), run_st<32, false>(big, tiles));
    printf("stores/wave sparse: 1 %.4f | 8 %.4f | 16 %.4f | 32 %.4f ms\n", run_st<1, true>(big, tiles), run_st<8, true>(big, tiles), run_st<16, true>(big, tiles), run_st<32, true>(big, tiles));
    return 0;
}
