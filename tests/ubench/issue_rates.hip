// Microbenchmark (gfx950): what an instruction COSTS a SIMD, the numbers DESIGN.md section 4 prices the PMC counters with.
//   hipcc --offload-arch=gfx950 -O3 issue_rates.hip -o issue_rates.bin && ./issue_rates.bin
// 1. throughput of independent wave64 VALU ops at 1, 2, 4, 8 waves per SIMD (cycles per wave instruction per SIMD);
// 2. issue-to-issue latency of a DEPENDENT chain of the same ops in ONE wave per SIMD (what the walkers and the
//    recurrence's forward sweep are made of);
// 3. ds_read_b32: one address per lane, conflict-free / random / all lanes one bank;
// 4. s_barrier in workgroups of 256 and 1024 threads.
// Clock: s_memtime-free -- wall time by hipEvents over a launch that fills every CU, at the 2.4 GHz the guide quotes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 4096
#define GHZ 2.4

template <int DEP> __global__ __launch_bounds__(1024) void k_valu(uint32_t *out, uint32_t seed)
{
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19,
             b = a0 ^ 0x5555;
    for (int it = 0; it < ITERS; it++) {
        if (DEP)
            asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n"
                         "v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1"
                         : "+v"(a0) : "v"(b));
        else
            asm volatile("v_add_u32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_xor_b32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// v_cmp + v_cndmask pairs (the compare-select idiom of the merge steps), 64-bit shifts, v_alignbyte, v_bfe
template <int KIND> __global__ __launch_bounds__(1024) void k_mix(uint32_t *out, uint32_t seed)
{
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = a0 ^ 0x5555, c = a0 + 9;
    uint64_t q0 = a0, q1 = a1;
    for (int it = 0; it < ITERS; it++) {
        if (KIND == 0)
            asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n"
                         "v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_lt_u32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
        else if (KIND == 1)
            asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshlrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n"
                         "v_lshlrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshlrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1"
                         : "+v"(q0), "+v"(q1));
        else if (KIND == 2)
            asm volatile("v_alignbyte_b32 %0, %0, %4, %5\n v_alignbyte_b32 %1, %1, %4, %5\n v_alignbyte_b32 %2, %2, %4, %5\n v_alignbyte_b32 %3, %3, %4, %5\n"
                         "v_alignbyte_b32 %0, %0, %4, %5\n v_alignbyte_b32 %1, %1, %4, %5\n v_alignbyte_b32 %2, %2, %4, %5\n v_alignbyte_b32 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        else
            asm volatile("v_bfe_u32 %0, %0, %4, %5\n v_bfe_u32 %1, %1, %4, %5\n v_bfe_u32 %2, %2, %4, %5\n v_bfe_u32 %3, %3, %4, %5\n"
                         "v_mbcnt_lo_u32_b32 %0, %4, %0\n v_mbcnt_hi_u32_b32 %1, %4, %1\n v_ffbl_b32 %2, %2\n v_bcnt_u32_b32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)q0 ^ (uint32_t)q1;
}

// MODE 0: lane l reads word l (+ a rotating base); 1: pseudo-random words; 2: every lane a word of ONE bank;
// DEP 1: the address of the next read depends on the value read (one wave's round trip)
template <int MODE, int DEP> __global__ __launch_bounds__(1024) void k_lds(uint32_t *out, uint32_t seed)
{
    __shared__ uint32_t w[8192];
    for (uint32_t i = threadIdx.x; i < 8192; i += blockDim.x) w[i] = (i * 2654435761u + seed) & 8191;
    __syncthreads();
    uint32_t a = MODE == 0 ? threadIdx.x & 8191 : MODE == 1 ? (threadIdx.x * 2654435761u >> 7) & 8191 : (threadIdx.x * 64) & 8191;
    uint32_t acc = 0;
    for (int it = 0; it < ITERS; it++) {
        uint32_t v = w[a];
        acc += v;
        if (DEP) a = MODE == 0 ? (a & ~63u & 8191) ^ (v & 8191 & ~63u) | (threadIdx.x & 63) : MODE == 1 ? v : (v & ~63u) & 8191;
        else a = MODE == 0 ? (a + 64) & 8191 : MODE == 1 ? (a * 5 + 1) & 8191 : (a + 64) & 8191;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ __launch_bounds__(1024) void k_barrier(uint32_t *out, uint32_t seed)
{
    uint32_t a = seed + threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_add_u32 %0, %0, 1" : "+v"(a));
        __builtin_amdgcn_s_barrier();
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

typedef void (*kfn)(uint32_t *, uint32_t);
static float timeit(kfn f, int blocks, int threads, uint32_t *d)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(threads), 0, 0, d, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(threads), 0, 0, d, 2u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
// one workgroup per CU of `threads` threads: threads / 256 waves per SIMD
static void valu(const char *name, kfn f, int threads, int per_iter, uint32_t *d)
{
    float ms = timeit(f, 256, threads, d);
    double cyc = ms * 1e-3 * GHZ * 1e9, wi = (double)(threads / 256) * ITERS * per_iter;   // per SIMD
    printf("%-44s %2d waves/SIMD  %.3f ms  %.2f cycles per wave instruction per SIMD\n", name, threads / 256, ms, cyc / wi);
}
int main()
{
    uint32_t *d; (void)hipMalloc(&d, 256 * 1024 * 4 * 8);
    for (int t = 256; t <= 1024; t *= 2) valu("v_add/v_xor u32, 8 independent", k_valu<0>, t, 8, d);
    valu("v_add/v_xor u32, 8 independent", k_valu<0>, 512 + 256, 8, d);
    for (int t = 256; t <= 1024; t *= 4) valu("v_add/v_xor u32, DEPENDENT chain", k_valu<1>, t, 8, d);
    for (int t = 256; t <= 1024; t *= 4) valu("v_cmp + v_cndmask (4 chains)", k_mix<0>, t, 8, d);
    for (int t = 256; t <= 1024; t *= 4) valu("v_lshl/lshr_b64 (2 chains)", k_mix<1>, t, 8, d);
    for (int t = 256; t <= 1024; t *= 4) valu("v_alignbyte_b32 (4 chains)", k_mix<2>, t, 8, d);
    for (int t = 256; t <= 1024; t *= 4) valu("v_bfe/mbcnt/ffbl/bcnt (4 chains)", k_mix<3>, t, 8, d);
    for (int t = 256; t <= 1024; t *= 4) {
        valu("ds_read_b32 lane-linear, independent", k_lds<0, 0>, t, 1, d);
        valu("ds_read_b32 random words, independent", k_lds<1, 0>, t, 1, d);
        valu("ds_read_b32 one bank, independent", k_lds<2, 0>, t, 1, d);
        valu("ds_read_b32 lane-linear, DEPENDENT", k_lds<0, 1>, t, 1, d);
        valu("ds_read_b32 random words, DEPENDENT", k_lds<1, 1>, t, 1, d);
    }
    for (int t = 256; t <= 1024; t *= 4) {
        float ms = timeit(k_barrier, 256, t, d);
        printf("%-44s %4d threads     %.3f ms  %.0f cycles per barrier\n", "s_barrier + 1 VALU", t, ms, ms * 1e-3 * GHZ * 1e9 / ITERS);
    }
    float ms = timeit(k_barrier, 512, 1024, d);
    printf("%-44s 2 x 1024 per CU    %.3f ms  %.0f cycles per barrier\n", "s_barrier + 1 VALU", ms, ms * 1e-3 * GHZ * 1e9 / ITERS);
    return 0;
}
