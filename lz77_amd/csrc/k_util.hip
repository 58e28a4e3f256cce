/*
 * k_util.hip -- small device utilities: input padding, ring-cell packing for the host recurrence,
 * exclusive scan.
 */
#include "kernels_common.h"

/* ------------------------------------------------------------------ small utilities -- */

__global__ void k_fill_pad(uint8_t *in, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < LZ77X_PAD) in[(size_t)n + i] = 0xFF;
}

__global__ void k_ps_cells(const uint32_t *__restrict__ ps, uint32_t *__restrict__ cells, uint32_t x0, uint32_t x1, uint32_t mask)
{
    for (uint32_t x = x0 + blockIdx.x * blockDim.x + threadIdx.x; x < x1; x += gridDim.x * blockDim.x) {
        const uint32_t v = ps[x];
        cells[x] = ((x + (v & 0xFFFFu)) & mask) | (((x + (v >> 16)) & mask) << 16);
    }
}

hipError_t lz77k_ps_cells(const uint32_t *d_ps, uint32_t *d_cells, uint32_t x0, uint32_t x1, uint32_t mask, hipStream_t s)
{
    if (x1 <= x0) return hipSuccess;
    const uint32_t blocks = min((x1 - x0 + 255u) / 256u, 256u * 16u);
    hipLaunchKernelGGL(k_ps_cells, dim3(blocks), dim3(256), 0, s, d_ps, d_cells, x0, x1, mask);
    return hipGetLastError();
}

hipError_t lz77k_fill_pad(uint8_t *d_in, uint32_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_fill_pad, dim3((LZ77X_PAD + 255) / 256), dim3(256), 0, s, d_in, n);
    return hipGetLastError();
}

/* ---- exclusive scan (uint32), 3 phases, recursive on block sums ---- */

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_CHUNK (SCAN_THREADS * SCAN_ITEMS)

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_local(const uint32_t *in, uint32_t *out,   /* may alias */
                                                             uint32_t m, uint32_t *sums)
{
    __shared__ uint32_t wsum[SCAN_THREADS / 64];
    const uint32_t base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], tot = 0;
    const bool whole = base + SCAN_ITEMS <= m;          /* eight consecutive words of a thread: two 16-byte accesses (the buffers are 256-byte aligned) */
    if (whole) {
        const uint4 a = *reinterpret_cast<const uint4 *>(in + base), b = *reinterpret_cast<const uint4 *>(in + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) v[i] = base + i < m ? in[base + i] : 0;
    }
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) tot += v[i];
    /* inclusive scan of tot across the wave, then across the 4 waves */
    uint32_t incl = tot;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, btot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        if (w < wave) woff += wsum[w];
        btot += wsum[w];
    }
    uint32_t run = woff + incl - tot;
    if (whole) {
        uint32_t o[SCAN_ITEMS];
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) { o[i] = run; run += v[i]; }
        *reinterpret_cast<uint4 *>(out + base) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4 *>(out + base + 4) = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            if (base + i < m) out[base + i] = run;
            run += v[i];
        }
    }
    if (threadIdx.x == 0 && sums) sums[blockIdx.x] = btot;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_add(uint32_t *__restrict__ out, uint32_t m, const uint32_t *__restrict__ offs)
{
    const uint32_t add = offs[blockIdx.x];
    const uint32_t base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_ITEMS;
    if (base + SCAN_ITEMS <= m) {
        uint4 a = *reinterpret_cast<const uint4 *>(out + base), b = *reinterpret_cast<const uint4 *>(out + base + 4);
        a.x += add; a.y += add; a.z += add; a.w += add; b.x += add; b.y += add; b.z += add; b.w += add;
        *reinterpret_cast<uint4 *>(out + base) = a;
        *reinterpret_cast<uint4 *>(out + base + 4) = b;
        return;
    }
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < m) out[base + i] += add;
}

size_t lz77k_scan_tmp_bytes(uint32_t m)
{
    size_t total = 0;
    uint64_t cur = m;
    while (cur > SCAN_CHUNK) {
        cur = (cur + SCAN_CHUNK - 1) / SCAN_CHUNK;
        total += ((cur + 64 + 3) & ~(uint64_t)3) * sizeof(uint32_t);
    }
    return total + 256;
}

hipError_t lz77k_scan_u32(const uint32_t *d_in, uint32_t *d_out, uint32_t m, void *d_tmp, hipStream_t s)
{
    if (m == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)(((uint64_t)m + SCAN_CHUNK - 1) / SCAN_CHUNK);
    if (blocks == 1) {
        hipLaunchKernelGGL(k_scan_local, dim3(1), dim3(SCAN_THREADS), 0, s, d_in, d_out, m, (uint32_t *)nullptr);
        return hipGetLastError();
    }
    uint32_t *sums = reinterpret_cast<uint32_t *>(d_tmp);
    hipLaunchKernelGGL(k_scan_local, dim3(blocks), dim3(SCAN_THREADS), 0, s, d_in, d_out, m, sums);
    hipError_t e = lz77k_scan_u32(sums, sums, blocks, sums + ((blocks + 64u + 3u) & ~3u), s);   /* (16-byte aligned: the kernels read four words at a time) */
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_scan_add, dim3(blocks), dim3(SCAN_THREADS), 0, s, d_out, m, sums);
    return hipGetLastError();
}


/* 64-bit sum of m uint32 (the decoded size of a stream before anything trusts 32-bit offsets) */
__global__ __launch_bounds__(256) void k_sum64(const uint32_t *__restrict__ in, uint32_t m, unsigned long long *__restrict__ out)
{
    unsigned long long acc = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < m; i += gridDim.x * 256u) acc += in[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63u) == 0 && acc) atomicAdd(out, acc);
}

hipError_t lz77k_sum_u32(const uint32_t *d_in, uint32_t m, unsigned long long *d_out, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_out, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess || m == 0) return e;
    const uint32_t blocks = min((m + 255u) / 256u, 2048u);
    hipLaunchKernelGGL(k_sum64, dim3(blocks), dim3(256), 0, s, d_in, m, d_out);
    return hipGetLastError();
}


/* A few words from device memory straight into PINNED host memory (hipHostMalloc: mapped into the device's address space,
 * coherent), by a kernel instead of hipMemcpyAsync: the runtime performs a small device-to-host copy as a blit kernel of its
 * own that takes ~21 us on the stream (profiles/r05_bench_kernel_stats.csv: 70 x __amd_rocclr_copyBuffer in four steps), and the
 * flags, counts and last tokens an encode or decode hands to the host are 4-16 bytes each, a dozen of them on the critical
 * path of every step.  The host reads them after it has synchronised the stream (a kernel's stores to host memory are
 * visible when the kernel has completed). */
__global__ void k_publish(uint32_t *__restrict__ h_dst, const uint32_t *__restrict__ d_src, uint32_t nwords)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nwords) h_dst[i] = d_src[i];
}

hipError_t lz77k_publish(void *h_dst_pinned, const void *d_src, uint32_t nwords, hipStream_t s)
{
    if (nwords == 0) return hipSuccess;
    hipLaunchKernelGGL(k_publish, dim3((nwords + 255u) / 256u), dim3(256), 0, s, reinterpret_cast<uint32_t *>(h_dst_pinned),
                       reinterpret_cast<const uint32_t *>(d_src), nwords);
    return hipGetLastError();
}
