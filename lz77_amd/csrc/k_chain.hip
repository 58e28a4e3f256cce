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

/* all groups composed, one lane per entry offset: the map of the whole range (what a shard of a stream that
 * is spread over several devices sends to the host, which chains the shards) */
__global__ void k_chain_whole(const uint8_t *__restrict__ gexit, const uint32_t *__restrict__ gcnt, uint32_t la, uint32_t ng,
                              uint8_t *__restrict__ wexit, uint32_t *__restrict__ wcnt)
{
    const uint32_t e = threadIdx.x;
    if (e >= la) return;
    uint32_t cur = e, tot = 0;
    for (uint32_t g = 0; g < ng; g++) {
        tot += gcnt[(size_t)g * la + cur];
        cur = gexit[(size_t)g * la + cur];
    }
    wexit[e] = (uint8_t)cur;
    wcnt[e] = tot;
}

/* the groups in sequence from the entry offset of the first sub-block (0 at the start of a stream): one lane */
__global__ void k_chain_top(const uint8_t *__restrict__ gexit, const uint32_t *__restrict__ gcnt, uint32_t la, uint32_t ng,
                            uint32_t *__restrict__ gentry, uint32_t *__restrict__ gbase, uint32_t *__restrict__ total, uint32_t entry0)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t e = entry0, tot = 0;
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
    size_t o_exit, o_cnt, o_gexit, o_gcnt, o_gentry, o_gbase, o_entry, o_tbase, o_total, o_wexit, o_wcnt, total;
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
    L.o_wexit = take(256);
    L.o_wcnt = take(256 * 4);
    L.total = o;
    return L;
}

size_t lz77k_chain_tmp_bytes(uint32_t n, int la) { return chain_make_layout(n, (uint32_t)la).total + 256; }
uint32_t lz77k_chain_sub(void) { return CHAIN_SB; }

/* Phase 1: the sub-blocks' maps and their composition by groups (and, whole = true, of the whole range: 256
 * bytes of exit offsets + 256 words of token counts, indexed by entry offset).  The chain covers positions
 * [start, n_all): sub-blocks are counted from `start`. */
hipError_t lz77k_chain_maps(const uint8_t *d_maxlen_all, uint32_t n_all, int la_i, void *d_tmp, hipStream_t s, uint32_t start, bool whole,
                            const uint8_t **d_wexit, const uint32_t **d_wcnt)
{
    const uint8_t *d_maxlen = d_maxlen_all + start;
    const uint32_t n = n_all > start ? n_all - start : 0u;
    const uint32_t la = (uint32_t)la_i;
    const chain_layout L = chain_make_layout(n, la);
    uint8_t *base = reinterpret_cast<uint8_t *>(d_tmp);
    uint8_t *exitmap = base + L.o_exit, *gexit = base + L.o_gexit, *wexit = base + L.o_wexit;
    uint16_t *cnt = reinterpret_cast<uint16_t *>(base + L.o_cnt);
    uint32_t *gcnt = reinterpret_cast<uint32_t *>(base + L.o_gcnt), *wcnt = reinterpret_cast<uint32_t *>(base + L.o_wcnt);
    if (d_wexit) *d_wexit = wexit;
    if (d_wcnt) *d_wcnt = wcnt;
    uint32_t lps = 2;
    while (lps < la) lps <<= 1;
    const uint32_t spw = 256u / lps < 16u ? 256u / lps : 16u;
    hipError_t e;
    if (n) {
        const size_t lds = (size_t)spw * CHAIN_SB;
        if (lds > 48 * 1024 &&
            (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain_map), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
            return e;
        hipLaunchKernelGGL(k_chain_map, dim3((L.nsub + spw - 1u) / spw), dim3(256), lds, s, d_maxlen, n, la, lps, L.nsub, exitmap, cnt);
        const size_t lds2 = (size_t)L.GC * la * 3 + 16;
        if (lds2 > 48 * 1024) {
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain_compose), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)) != hipSuccess)
                return e;
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain_apply), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)) != hipSuccess)
                return e;
        }
        hipLaunchKernelGGL(k_chain_compose, dim3(L.ng), dim3(256), lds2, s, exitmap, cnt, la, L.nsub, L.GC, gexit, gcnt);
    }
    if (whole) hipLaunchKernelGGL(k_chain_whole, dim3(1), dim3(256), 0, s, gexit, gcnt, la, n ? L.ng : 0u, wexit, wcnt);
    return hipGetLastError();
}

/* Phase 2: from the entry offset of the first sub-block, every sub-block's true entry and first-token
 * index, then chain[k] = position of token k.  *d_tbase -> first-token index of every CHAIN_SB sub-block
 * (nsub + 1 words, the last one = ntok), *d_exit -> how far past n_all the last token reaches; both inside d_tmp. */
hipError_t lz77k_chain_finish(const uint8_t *d_maxlen_all, uint32_t n_all, int la_i, uint32_t *d_chain, void *d_tmp, hipStream_t s,
                              uint32_t start, uint32_t entry0, const uint32_t **d_tbase, uint32_t *nsub_out, const uint32_t **d_exit)
{
    const uint8_t *d_maxlen = d_maxlen_all + start;
    const uint32_t n = n_all > start ? n_all - start : 0u;
    const uint32_t la = (uint32_t)la_i;
    const chain_layout L = chain_make_layout(n, la);
    uint8_t *base = reinterpret_cast<uint8_t *>(d_tmp);
    uint8_t *exitmap = base + L.o_exit, *gexit = base + L.o_gexit;
    uint16_t *cnt = reinterpret_cast<uint16_t *>(base + L.o_cnt);
    uint32_t *gcnt = reinterpret_cast<uint32_t *>(base + L.o_gcnt);
    uint32_t *gentry = reinterpret_cast<uint32_t *>(base + L.o_gentry);
    uint32_t *gbase = reinterpret_cast<uint32_t *>(base + L.o_gbase);
    uint32_t *entry = reinterpret_cast<uint32_t *>(base + L.o_entry);
    uint32_t *tbase = reinterpret_cast<uint32_t *>(base + L.o_tbase);
    uint32_t *total = reinterpret_cast<uint32_t *>(base + L.o_total);
    *d_tbase = tbase;
    *nsub_out = L.nsub;
    if (d_exit) *d_exit = total + 1;
    hipError_t e;
    if (n == 0) {
        if ((e = hipMemsetAsync(tbase, 0, 4, s)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(total, 0, 4, s)) != hipSuccess) return e;
        return hipMemcpyAsync(total + 1, &entry0, 4, hipMemcpyHostToDevice, s);     /* nothing to walk: the entry passes through */
    }
    const size_t lds2 = (size_t)L.GC * la * 3 + 16;
    hipLaunchKernelGGL(k_chain_top, dim3(1), dim3(64), 0, s, gexit, gcnt, la, L.ng, gentry, gbase, total, entry0);
    hipLaunchKernelGGL(k_chain_apply, dim3(L.ng), dim3(256), lds2, s, exitmap, cnt, la, L.nsub, L.GC, gentry, gbase, entry, tbase, total + 1);
    const size_t lds = (size_t)CHAIN_EMIT_SUBS * CHAIN_SB;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain_emit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
        return e;
    hipLaunchKernelGGL(k_chain_emit, dim3((L.nsub + CHAIN_EMIT_SUBS - 1u) / CHAIN_EMIT_SUBS), dim3(64), lds, s, d_maxlen, n, L.nsub, entry, tbase,
                       d_chain, start);
    return hipGetLastError();
}

/* both phases for a range whose chain begins exactly at `start` */
hipError_t lz77k_chain(const uint8_t *d_maxlen_all, uint32_t n_all, int la_i, uint32_t *d_chain, void *d_tmp, hipStream_t s,
                       const uint32_t **d_tbase, uint32_t *nsub_out, uint32_t start, const uint32_t **d_exit)
{
    hipError_t e = lz77k_chain_maps(d_maxlen_all, n_all, la_i, d_tmp, s, start, false, nullptr, nullptr);
    if (e != hipSuccess) return e;
    return lz77k_chain_finish(d_maxlen_all, n_all, la_i, d_chain, d_tmp, s, start, 0u, d_tbase, nsub_out, d_exit);
}
