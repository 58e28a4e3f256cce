/*
 * kernels_common.h -- device helpers shared by the HIP translation units of the hot path
 * (k_match.hip, k_tokens.hip, k_decode.hip, k_util.hip).  gfx950 only, no MFMA: byte/integer work.
 */
#ifndef LZ77X_KERNELS_COMMON_H
#define LZ77X_KERNELS_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "lz77x_internal.h"

/* ------------------------------------------------------------------ helpers ---------- */

__device__ __forceinline__ uint32_t ld32u(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);          /* gfx950: unaligned dword access is native (LDS and global) */
    return v;
}

__device__ __forceinline__ uint64_t ld64u(const uint8_t *p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

/* dword at byte offset `off` of a byte array.  LDS: an unaligned ds_read_b32 is replayed for
 * ~64 cycles per wave on gfx950, so fetch the two aligned dwords and funnel-shift; global
 * memory serves unaligned dwords natively. */
template <bool LDS>
__device__ __forceinline__ uint32_t ld32_at(const uint8_t *by, uint32_t off)
{
    if constexpr (LDS) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(by + (off & ~3u));
        return __builtin_amdgcn_alignbyte(w[1], w[0], off & 3u);
    } else {
        return ld32u(by + off);
    }
}

/* first `la` bytes at a vs b as big-endian words; ties -> lower index first.
 * tree.c:77 orders nodes with memcmp over the lookahead; bytes past the end of input are
 * 0xFF on device, which reproduces the shrinking key length at EOF (DESIGN.md "key order"). */
template <bool LDS>
__device__ __forceinline__ bool key_less(const uint8_t *by, uint32_t a, uint32_t b, int la)
{
    for (int w = 0; w < la; w += 4) {
        uint32_t va = __builtin_bswap32(ld32_at<LDS>(by, a + w));
        uint32_t vb = __builtin_bswap32(ld32_at<LDS>(by, b + w));
        int rem = la - w;
        if (rem < 4) {
            uint32_t m = 0xFFFFFFFFu << (8 * (4 - rem));
            va &= m;
            vb &= m;
        }
        if (va != vb) return va < vb;
    }
    return a < b;
}

template <bool LDS>
__device__ __forceinline__ int lcp_capped(const uint8_t *by, uint32_t a, uint32_t b, int cap)
{
    int i = 0;
    while (i < cap) {
        const uint32_t x = ld32_at<LDS>(by, a + i) ^ ld32_at<LDS>(by, b + i);
        if (x) { i += __builtin_ctz(x) >> 3; break; }
        i += 4;
    }
    return i < cap ? i : cap;
}


#endif
