// How fast are 64-bit shifts / ands on gfx950 next to 32-bit ones?  (the split kernel's mask algebra is built from them)
//   hipcc --offload-arch=gfx950 -O3 -o shift64_rate shift64_rate.hip && ./shift64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void __launch_bounds__(256) k(uint64_t* out, int iters, uint64_t seed) {
    uint64_t x[8];
    for (int i = 0; i < 8; i++) x[i] = seed + threadIdx.x * 977 + i * 131 + blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) x[i] = (x[i] << 3) ^ seed;                                    // v_lshlrev_b64 + 2 xor
            else if (MODE == 1) {                                                          // the same from 32-bit pieces
                uint32_t lo = (uint32_t)x[i], hi = (uint32_t)(x[i] >> 32);
                hi = __builtin_amdgcn_alignbit(hi, lo, 29);
                lo <<= 3;
                x[i] = (((uint64_t)hi << 32) | lo) ^ seed;
            } else if (MODE == 2) x[i] = (x[i] >> 5) ^ seed;                             // v_lshrrev_b64
            else if (MODE == 3) x[i] = (x[i] & seed) + 0x9E3779B97F4A7C15ull;             // 64-bit add (2 x 32 with carry)
            else if (MODE == 4) { uint32_t lo = (uint32_t)x[i]; lo = (lo << 3) ^ (uint32_t)seed; x[i] = lo; }   // 32-bit only
            else if (MODE == 5) x[i] = (uint64_t)__popcll(x[i]) + (x[i] ^ seed);
        }
    }
    uint64_t r = 0;
    for (int i = 0; i < 8; i++) r ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
static void run(const char* name, int vinstr) {
    uint64_t* d;
    const int blocks = 256 * 8, iters = 4096;
    hipMalloc(&d, blocks * 256 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0x123456789ull);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 0x123456789ull);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * 4 * iters * 8;        // wave-level "statements"
    printf("%-28s %.3f ms  %.1f G statements/s per chip  = %.2f clk per statement per SIMD (%d VALU instr expected)\n", name, ms,
           ops / ms / 1e6, ms * 1e-3 * 2.4e9 * 1024 / ops, vinstr);
    hipFree(d);
}

int main() {
    run<4>("32-bit shl + xor", 2);
    run<0>("64-bit shl + xor", 3);
    run<1>("64-bit shl from 32-bit + xor", 4);
    run<2>("64-bit shr + xor", 3);
    run<3>("64-bit and + add", 4);
    run<5>("popcll + xor + add", 6);
    return 0;
}
