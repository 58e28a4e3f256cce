/*
 * k_decode.hip -- lz77.c:148-197 decode, lz77.c:260-283 readcode, bitio.c:256-298 bitIO_read as
 * parse -> scan -> expand -> pointer doubling -> gather.
 */
#include "kernels_common.h"

/* ------------------------------------------------------------------ decode ----------- */

/* lz77.c:260-283 + bitio.c:256-298: fixed-width tokens, so token k is simply bits [32+kT, ..) */
__global__ void k_dec_parse(const uint8_t *__restrict__ z, uint32_t ntok, int ob, int lb, int T,
                            uint32_t *__restrict__ tokval, uint32_t *__restrict__ len1)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ntok) return;
    const uint64_t bit = 32 + (uint64_t)k * (uint64_t)T;
    uint64_t v = ld64u(z + (bit >> 3)) >> (bit & 7);
    v &= T >= 32 ? 0xFFFFFFFFull : ((1ull << T) - 1);
    tokval[k] = (uint32_t)v;
    len1[k] = (((uint32_t)v >> ob) & ((1u << lb) - 1u)) + 1u;
}

/* lz77.c:178-194 as data flow: every copied byte j points at j-off, every literal at itself.
 * Position n is a zero byte that degenerate tokens (off==0 or off>j) point at. */
__global__ void k_dec_expand(const uint32_t *__restrict__ tokval, const uint32_t *__restrict__ dst, uint32_t ntok,
                             int ob, int lb, uint8_t *__restrict__ out, uint32_t *__restrict__ ptr, uint32_t n)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) { out[n] = 0; ptr[n] = n; }
    if (k >= ntok) return;
    const uint32_t v = tokval[k];
    const uint32_t off = ob ? (v & ((1u << ob) - 1u)) : 0;
    const uint32_t len = (v >> ob) & ((1u << lb) - 1u);
    const uint32_t lit = (v >> (ob + lb)) & 0xFFu;
    const uint32_t j0 = dst[k];
    for (uint32_t i = 0; i < len; i++) {
        const uint32_t j = j0 + i;
        ptr[j] = (off > 0 && off <= j) ? j - off : n;
    }
    out[j0 + len] = (uint8_t)lit;
    ptr[j0 + len] = j0 + len;
}

/* pointer jumping, two hops per pass: ptr[j] <- ptr[ptr[ptr[j]]] until every byte points at a literal.
 * A chain of depth d shrinks to ~d/3 per pass (5 passes for the depth-136 chains of text); concurrent
 * updates of other entries only ever move them further along the same chain, so any interleaving is safe. */
__global__ void k_dec_jump(uint32_t *__restrict__ ptr, uint32_t n, uint32_t *__restrict__ changed)
{
    bool any = false;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t p = ptr[j];
        if (p == j) continue;
        const uint32_t q = ptr[p];
        if (q == p) continue;
        const uint32_t r = ptr[q];
        ptr[j] = r;
        any = true;
    }
    if (any) *changed = 1;
}

__global__ void k_dec_gather(uint8_t *__restrict__ out, const uint32_t *__restrict__ ptr, uint32_t n)
{
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t p = ptr[j];
        if (p != j) out[j] = out[p];
    }
}

hipError_t lz77k_dec_parse(const uint8_t *d_z, uint32_t ntok, const lz77x_geom &g, uint32_t *d_tokval, uint32_t *d_len1, hipStream_t s)
{
    if (ntok == 0) return hipSuccess;
    hipLaunchKernelGGL(k_dec_parse, dim3((ntok + 255) / 256), dim3(256), 0, s, d_z, ntok, g.ob, g.lb, g.T, d_tokval, d_len1);
    return hipGetLastError();
}

hipError_t lz77k_dec_expand(const uint32_t *d_tokval, const uint32_t *d_dst, uint32_t ntok, const lz77x_geom &g,
                            uint8_t *d_out, uint32_t *d_ptr, uint32_t n, hipStream_t s)
{
    hipLaunchKernelGGL(k_dec_expand, dim3(ntok ? (ntok + 255) / 256 : 1), dim3(256), 0, s, d_tokval, d_dst, ntok, g.ob, g.lb, d_out, d_ptr, n);
    return hipGetLastError();
}

hipError_t lz77k_dec_jump(uint32_t *d_ptr, uint32_t n, uint32_t *d_changed, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const uint32_t blocks = min((n + 255u) / 256u, 256u * 32u);
    hipLaunchKernelGGL(k_dec_jump, dim3(blocks), dim3(256), 0, s, d_ptr, n, d_changed);
    return hipGetLastError();
}

hipError_t lz77k_dec_gather(uint8_t *d_out, const uint32_t *d_ptr, uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const uint32_t blocks = min((n + 255u) / 256u, 256u * 32u);
    hipLaunchKernelGGL(k_dec_gather, dim3(blocks), dim3(256), 0, s, d_out, d_ptr, n);
    return hipGetLastError();
}

