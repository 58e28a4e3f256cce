/*
 * k_chain.hip -- the greedy parse chain on the device.
 *
 * lz77.c:89-98: one token per chain position, p <- p + len + 1 with len = maxlen[p] (tree.c:118-152).
 * Sequential as written, but a token never jumps more than LA positions, so a sub-block of CHAIN_SB
 * positions can only be entered at one of its first LA positions: walk all of them (one lane each,
 * bytes staged in LDS), which gives the sub-block's map  entry offset -> (exit offset, tokens);
 * compose the maps (groups of sub-blocks, then the groups in sequence) to learn every sub-block's true
 * entry and the index of its first token, then walk each sub-block once more from its true entry and
 * write chain[] -- the position of every token, in order.  (chains from different entries
 * merge within a few hundred bytes, but no bound holds, hence the exact composition.)
 */
#include "kernels_common.h"

#define CHAIN_SB 4096u

__device__ __forceinline__ void chain_stage(uint8_t *dst, const uint8_t *__restrict__ maxlen, uint32_t n, uint32_t sub0,
                                            uint32_t nstage, uint32_t nthreads)
{
    /* nstage sub-blocks starting at sub0, 16 bytes per thread and step.  maxlen may start at any byte (a
     * later segment's chain begins wherever its predecessor's last token ended): aligned dwords + funnel
     * shift; the array has >= 64 bytes of slack behind n */
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(maxlen) & 3u);
    const uint32_t *base = reinterpret_cast<const uint32_t *>(maxlen - sh);
    for (uint32_t i = threadIdx.x * 16u; i < nstage * CHAIN_SB; i += nthreads * 16u) {
        const uint64_t pos = (uint64_t)sub0 * CHAIN_SB + i;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (pos < n) {
            const uint32_t *w = base + (pos >> 2);
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
            v = make_uint4(__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
                           __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w4, w3, sh));
        }
        *reinterpret_cast<uint4 *>(dst + i) = v;
    }
}

/* lps = lanes per sub-block (power of two >= la); a workgroup of 256 threads walks 256/lps sub-blocks */
__global__ __launch_bounds__(256) void k_chain_map(const uint8_t *__restrict__ maxlen, uint32_t n, uint32_t la, uint32_t lps,
                                                   uint32_t nsub, uint8_t *__restrict__ exitmap, uint16_t *__restrict__ cnt)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t chain_ml[];
    const uint32_t spw = min(256u / lps, 16u), sub0 = blockIdx.x * spw;      /* <= 64 KiB of staged bytes */
    chain_stage(chain_ml, maxlen, n, sub0, min(spw, nsub - sub0), 256u);
    __syncthreads();
    const uint32_t local = threadIdx.x / lps, e = threadIdx.x % lps, sub = sub0 + local;
    if (local >= spw || sub >= nsub || e >= la) return;
    const uint64_t base = (uint64_t)sub * CHAIN_SB;
    const uint32_t end = n - base < CHAIN_SB ? (uint32_t)(n - base) : CHAIN_SB;
    const uint8_t *ml = chain_ml + local * CHAIN_SB;
    uint32_t p = e, c = 0;
    while (p < end) { p += (uint32_t)ml[p] + 1u; c++; }
    exitmap[(size_t)sub * la + e] = (uint8_t)(p - end);          /* < la: a token spans at most la positions */
    cnt[(size_t)sub * la + e] = (uint16_t)c;
}

/* group g = sub-blocks [g*GC, ...): its composed map, one lane per entry offset, rows staged in LDS */
__global__ __launch_bounds__(256) void k_chain_compose(const uint8_t *__restrict__ exitmap, const uint16_t *__restrict__ cnt,
                                                       uint32_t la, uint32_t nsub, uint32_t GC, uint8_t *__restrict__ gexit,
                                                       uint32_t *__restrict__ gcnt)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t chain_rows[];
    const uint32_t g = blockIdx.x, s0 = g * GC, s1 = min(s0 + GC, nsub), rows = s1 - s0;
    uint16_t *c_l = reinterpret_cast<uint16_t *>(chain_rows);
    uint8_t *e_l = chain_rows + (size_t)GC * la * 2;
    for (uint32_t i = threadIdx.x; i < rows * la; i += 256) {
        c_l[i] = cnt[(size_t)s0 * la + i];
        e_l[i] = exitmap[(size_t)s0 * la + i];
    }
    __syncthreads();
    const uint32_t e = threadIdx.x;
    if (e >= la) return;
    uint32_t cur = e, total = 0;
    for (uint32_t r = 0; r < rows; r++) {
        total += c_l[r * la + cur];
        cur = e_l[r * la + cur];
    }
    gexit[(size_t)g * la + e] = (uint8_t)cur;
    gcnt[(size_t)g * la + e] = total;
}

/* the groups in sequence from the start of the input (entry offset 0): one lane */
__global__ void k_chain_top(const uint8_t *__restrict__ gexit, const uint32_t *__restrict__ gcnt, uint32_t la, uint32_t ng,
                            uint32_t *__restrict__ gentry, uint32_t *__restrict__ gbase, uint32_t *__restrict__ total)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t e = 0, tot = 0;
    for (uint32_t g = 0; g < ng; g++) {
        gentry[g] = e;
        gbase[g] = tot;
        tot += gcnt[(size_t)g * la + e];
        e = gexit[(size_t)g * la + e];
    }
    *total = tot;
}

/* inside every group: each sub-block's true entry offset and the index of its first token */
__global__ __launch_bounds__(256) void k_chain_apply(const uint8_t *__restrict__ exitmap, const uint16_t *__restrict__ cnt,
                                                     uint32_t la, uint32_t nsub, uint32_t GC, const uint32_t *__restrict__ gentry,
                                                     const uint32_t *__restrict__ gbase, uint32_t *__restrict__ entry,
                                                     uint32_t *__restrict__ tbase, uint32_t *__restrict__ exit_out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t chain_rows[];
    const uint32_t g = blockIdx.x, s0 = g * GC, s1 = min(s0 + GC, nsub), rows = s1 - s0;
    uint16_t *c_l = reinterpret_cast<uint16_t *>(chain_rows);
    uint8_t *e_l = chain_rows + (size_t)GC * la * 2;
    for (uint32_t i = threadIdx.x; i < rows * la; i += 256) {
        c_l[i] = cnt[(size_t)s0 * la + i];
        e_l[i] = exitmap[(size_t)s0 * la + i];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    uint32_t cur = gentry[g], tot = gbase[g];
    for (uint32_t r = 0; r < rows; r++) {
        entry[s0 + r] = cur;
        tbase[s0 + r] = tot;
        tot += c_l[r * la + cur];
        cur = e_l[r * la + cur];
    }
    if (s1 == nsub) { tbase[nsub] = tot; *exit_out = cur; }   /* cur: how far past the end the last token reaches */
}

/* 16 sub-blocks per workgroup of 64 (their bytes in LDS): lane l < 16 walks one from its true entry */
#define CHAIN_EMIT_SUBS 16u
__global__ __launch_bounds__(64) void k_chain_emit(const uint8_t *__restrict__ maxlen, uint32_t n, uint32_t nsub,
                                                   const uint32_t *__restrict__ entry, const uint32_t *__restrict__ tbase,
                                                   uint32_t *__restrict__ chain, uint32_t pos_off)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t chain_ml[];
    const uint32_t sub0 = blockIdx.x * CHAIN_EMIT_SUBS;
    chain_stage(chain_ml, maxlen, n, sub0, min(CHAIN_EMIT_SUBS, nsub - sub0), 64u);
    __syncthreads();
    const uint32_t sub = sub0 + threadIdx.x;
    if (threadIdx.x >= CHAIN_EMIT_SUBS || sub >= nsub) return;
    const uint64_t base = (uint64_t)sub * CHAIN_SB;
    const uint32_t end = n - base < CHAIN_SB ? (uint32_t)(n - base) : CHAIN_SB;
    const uint8_t *ml = chain_ml + threadIdx.x * CHAIN_SB;
    uint32_t p = entry[sub], k = tbase[sub];
    while (p < end) {
        chain[k++] = (uint32_t)base + p + pos_off;
        p += (uint32_t)ml[p] + 1u;
    }
}

struct chain_layout {
    uint32_t nsub, GC, ng;
    size_t o_exit, o_cnt, o_gexit, o_gcnt, o_gentry, o_gbase, o_entry, o_tbase, o_total, total;
};

static chain_layout chain_make_layout(uint32_t n, uint32_t la)
{
    chain_layout L;
    L.nsub = (uint32_t)(((uint64_t)n + CHAIN_SB - 1) / CHAIN_SB);
    uint32_t GC = 49152u / (3u * la);
    if (GC > 256u) GC = 256u;
    L.GC = GC;
    L.ng = (L.nsub + GC - 1u) / GC;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    L.o_exit = take((size_t)L.nsub * la);
    L.o_cnt = take((size_t)L.nsub * la * 2);
    L.o_gexit = take((size_t)L.ng * la);
    L.o_gcnt = take((size_t)L.ng * la * 4);
    L.o_gentry = take((size_t)L.ng * 4);
    L.o_gbase = take((size_t)L.ng * 4);
    L.o_entry = take((size_t)L.nsub * 4);
    L.o_tbase = take(((size_t)L.nsub + 1) * 4);
    L.o_total = take(64);
    L.total = o;
    return L;
}

size_t lz77k_chain_tmp_bytes(uint32_t n, int la) { return chain_make_layout(n, (uint32_t)la).total + 256; }
uint32_t lz77k_chain_sub(void) { return CHAIN_SB; }

/* chain[k] = position of token k; *d_tbase -> first-token index of every CHAIN_SB sub-block (nsub + 1
 * words, the last one = ntok), inside d_tmp.  Enqueues only. */
hipError_t lz77k_chain(const uint8_t *d_maxlen_all, uint32_t n_all, int la_i, uint32_t *d_chain, void *d_tmp, hipStream_t s,
                       const uint32_t **d_tbase, uint32_t *nsub_out, uint32_t start, const uint32_t **d_exit)
{
    /* the chain begins at position `start`: sub-blocks are counted from there */
    const uint8_t *d_maxlen = d_maxlen_all + start;
    const uint32_t n = n_all > start ? n_all - start : 0u;
    const uint32_t la = (uint32_t)la_i;
    const chain_layout L = chain_make_layout(n, la);
    uint8_t *base = reinterpret_cast<uint8_t *>(d_tmp);
    uint8_t *exitmap = base + L.o_exit;
    uint16_t *cnt = reinterpret_cast<uint16_t *>(base + L.o_cnt);
    uint8_t *gexit = base + L.o_gexit;
    uint32_t *gcnt = reinterpret_cast<uint32_t *>(base + L.o_gcnt);
    uint32_t *gentry = reinterpret_cast<uint32_t *>(base + L.o_gentry);
    uint32_t *gbase = reinterpret_cast<uint32_t *>(base + L.o_gbase);
    uint32_t *entry = reinterpret_cast<uint32_t *>(base + L.o_entry);
    uint32_t *tbase = reinterpret_cast<uint32_t *>(base + L.o_tbase);
    uint32_t *total = reinterpret_cast<uint32_t *>(base + L.o_total);
    *d_tbase = tbase;
    *nsub_out = L.nsub;
    if (d_exit) *d_exit = total + 1;
    if (n == 0) {
        hipError_t e0 = hipMemsetAsync(tbase, 0, 4, s);
        return e0 == hipSuccess ? hipMemsetAsync(total, 0, 8, s) : e0;
    }
    uint32_t lps = 2;
    while (lps < la) lps <<= 1;
    const uint32_t spw = 256u / lps < 16u ? 256u / lps : 16u;
    hipError_t e;
    {
        const size_t lds = (size_t)spw * CHAIN_SB;
        if (lds > 48 * 1024 &&
            (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain_map), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
            return e;
        hipLaunchKernelGGL(k_chain_map, dim3((L.nsub + spw - 1u) / spw), dim3(256), lds, s, d_maxlen, n, la, lps, L.nsub, exitmap, cnt);
    }
    {
        const size_t lds = (size_t)L.GC * la * 3 + 16;
        if (lds > 48 * 1024) {
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain_compose), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
                return e;
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain_apply), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
                return e;
        }
        hipLaunchKernelGGL(k_chain_compose, dim3(L.ng), dim3(256), lds, s, exitmap, cnt, la, L.nsub, L.GC, gexit, gcnt);
        hipLaunchKernelGGL(k_chain_top, dim3(1), dim3(64), 0, s, gexit, gcnt, la, L.ng, gentry, gbase, total);
        hipLaunchKernelGGL(k_chain_apply, dim3(L.ng), dim3(256), lds, s, exitmap, cnt, la, L.nsub, L.GC, gentry, gbase, entry, tbase, total + 1);
    }
    {
        const size_t lds = (size_t)CHAIN_EMIT_SUBS * CHAIN_SB;
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain_emit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
            return e;
        hipLaunchKernelGGL(k_chain_emit, dim3((L.nsub + CHAIN_EMIT_SUBS - 1u) / CHAIN_EMIT_SUBS), dim3(64), lds, s, d_maxlen, n, L.nsub, entry,
                           tbase, d_chain, start);
    }
    return hipGetLastError();
}
