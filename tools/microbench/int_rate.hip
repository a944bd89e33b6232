// Microbenchmark: issue rate of 32-bit integer VALU instructions on gfx950 (wave64), per SIMD.
// Each wave runs ITER x 8 independent chains of one instruction; time -> cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 4096;

#define DEFK(NAME, ASM)                                                                         \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t k) {                      \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    for (int i = 0; i < ITER; i++) {                                                              \
        asm volatile(ASM " %0, %0, %1" : "+v"(a0) : "v"(k)); asm volatile(ASM " %0, %0, %1" : "+v"(a1) : "v"(k)); \
        asm volatile(ASM " %0, %0, %1" : "+v"(a2) : "v"(k)); asm volatile(ASM " %0, %0, %1" : "+v"(a3) : "v"(k)); \
        asm volatile(ASM " %0, %0, %1" : "+v"(a4) : "v"(k)); asm volatile(ASM " %0, %0, %1" : "+v"(a5) : "v"(k)); \
        asm volatile(ASM " %0, %0, %1" : "+v"(a6) : "v"(k)); asm volatile(ASM " %0, %0, %1" : "+v"(a7) : "v"(k)); \
    }                                                                                             \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                 \
}
DEFK(k_add, "v_add_u32")
DEFK(k_xor, "v_xor_b32")
DEFK(k_mul_lo, "v_mul_lo_u32")
DEFK(k_mul_hi, "v_mul_hi_u32")
DEFK(k_mul24, "v_mul_u32_u24")
DEFK(k_lshl, "v_lshlrev_b32")

template <class K> void run(const char* name, K kern, uint32_t* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 8;          // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 3u);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 3u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr_per_simd = (double)blocks * 4 / 1024 * ITER * 8;
    printf("%-14s %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, ms,
           ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
}
int main() {
    uint32_t* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run("v_add_u32", k_add, out); run("v_xor_b32", k_xor, out); run("v_lshlrev_b32", k_lshl, out);
    run("v_mul_u32_u24", k_mul24, out); run("v_mul_lo_u32", k_mul_lo, out); run("v_mul_hi_u32", k_mul_hi, out);
    return 0;
}
