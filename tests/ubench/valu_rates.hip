// Microbenchmark: issue rate of the VALU ops the pair scan is built from (gfx950).
// hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates.bin && ./valu_rates.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 2048
#define R8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define DEF(NAME, TEXT)                                                                         \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)                   \
    {                                                                                           \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, \
                 a7 = a0 * 19, b = a0 ^ 0x5555, c = a0 + 77;                                    \
        for (int it = 0; it < ITERS; it++) {                                                    \
            asm volatile(TEXT "\n" TEXT "\n" TEXT "\n" TEXT                                     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;     \
    }
#define T8(ins) ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8"
#define T8_3(ins) ins " %0, %0, %8, %9\n" ins " %1, %1, %8, %9\n" ins " %2, %2, %8, %9\n" ins " %3, %3, %8, %9\n" ins " %4, %4, %8, %9\n" ins " %5, %5, %8, %9\n" ins " %6, %6, %8, %9\n" ins " %7, %7, %8, %9"
DEF(k_min_u32, T8("v_min_u32"))
DEF(k_sub_u32, T8("v_sub_u32"))
DEF(k_pk_min_u16, T8("v_pk_min_u16"))
DEF(k_pk_max_u16, T8("v_pk_max_u16"))
DEF(k_pk_sub_i16, T8("v_pk_sub_i16"))
DEF(k_min3_u32, T8_3("v_min3_u32"))
DEF(k_max3_u32, T8_3("v_max3_u32"))
DEF(k_min_u16, T8("v_min_u16"))
DEF(k_alignbyte, T8_3("v_alignbyte_b32"))
DEF(k_msad, T8_3("v_msad_u8"))
DEF(k_sad, T8_3("v_sad_u8"))
typedef void (*kfn)(uint32_t *, uint32_t);
static void run(const char *name, kfn f, uint32_t *d)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, 2u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * ITERS * 32;
    printf("%-16s %.3f ms  %.3f wave-instr/clk/SIMD @2.4GHz  (%.1f T lane-op/s)\n", name, ms,
           winstr / (ms * 1e-3) / 2.4e9 / 1024.0, winstr * 64 / (ms * 1e-3) / 1e12);
}
int main()
{
    uint32_t *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run("v_min_u32", k_min_u32, d); run("v_sub_u32", k_sub_u32, d); run("v_pk_min_u16", k_pk_min_u16, d);
    run("v_pk_max_u16", k_pk_max_u16, d); run("v_pk_sub_i16", k_pk_sub_i16, d); run("v_min3_u32", k_min3_u32, d);
    run("v_max3_u32", k_max3_u32, d); run("v_min_u16", k_min_u16, d); run("v_alignbyte_b32", k_alignbyte, d);
    run("v_msad_u8", k_msad, d); run("v_sad_u8", k_sad, d);
    return 0;
}
