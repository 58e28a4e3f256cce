// What does the HIP runtime keep of a queue's SCRATCH (private segment) memory once a kernel that spills has run?
// tools/mem_probe.py sees 19-78 MB of device memory still held after lz77x_shutdown() has freed every buffer and destroyed every
// stream and event the library made; k_c1_chunks, k_big_chunks and k_pw_fwd use scratch (tools/kres.py).  This probe runs a kernel
// with S bytes of scratch per lane and prints hipMemGetInfo before it, after it, and after its stream is destroyed.
//   hipcc --offload-arch=gfx950 -O1 -o tools/scratch_hold_probe tools/scratch_hold_probe.hip && ./tools/scratch_hold_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WORDS> __global__ void k_spill(uint32_t *out, uint32_t seed)
{
    volatile uint32_t a[WORDS];
    for (int i = 0; i < WORDS; i++) a[i] = seed * (uint32_t)i + threadIdx.x;
    uint32_t s = 0;
    for (int i = 0; i < WORDS; i++) s += a[(i * 7 + seed) % WORDS];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static size_t free_now() { size_t f = 0, t = 0; (void)hipMemGetInfo(&f, &t); return f; }
template <int WORDS> static void run(const char *what)
{
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint32_t *d = nullptr;
    (void)hipMalloc(&d, 4096 * 256 * 4);
    const size_t f0 = free_now();
    hipLaunchKernelGGL(k_spill<WORDS>, dim3(4096), dim3(256), 0, s, d, 3u);
    (void)hipStreamSynchronize(s);
    const size_t f1 = free_now();
    (void)hipFree(d);
    (void)hipStreamDestroy(s);
    (void)hipDeviceSynchronize();
    const size_t f2 = free_now();
    printf("%-24s scratch %5d B/lane: held after the kernel %7.1f MB; after hipFree + hipStreamDestroy still %7.1f MB below the free memory before the kernel's buffer\n",
           what, WORDS * 4, ((double)f0 - (double)f1) / 1e6, ((double)f0 + 4096.0 * 256 * 4 - (double)f2) / 1e6);
}
int main()
{
    (void)hipFree(nullptr);
    printf("free at start %.1f MB\n", free_now() / 1e6);
    run<16>("small");
    run<128>("512 B per lane");
    run<512>("2 KB per lane");
    run<128>("512 B per lane again");
    printf("free at end %.1f MB\n", free_now() / 1e6);
    return 0;
}
