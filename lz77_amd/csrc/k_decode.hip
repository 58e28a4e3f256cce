/*
 * k_decode.hip -- lz77.c:148-197 decode, lz77.c:260-283 readcode, bitio.c:256-298 bitIO_read as
 * parse -> scan -> expand -> pointer doubling -> gather.
 */
#include "kernels_common.h"

/* ------------------------------------------------------------------ decode ----------- */

/* lz77.c:260-283 + bitio.c:256-298: fixed-width tokens, so token k is simply bits [32+kT, ..) */
__global__ void k_dec_parse(const uint8_t *__restrict__ z, uint32_t ntok, int ob, int lb, int T, uint32_t sb,
                            uint32_t *__restrict__ tokval, uint32_t *__restrict__ len1, uint32_t *__restrict__ stale_flag)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ntok) return;
    const uint64_t bit = 32 + (uint64_t)k * (uint64_t)T;
    uint64_t v = ld64u(z + (bit >> 3)) >> (bit & 7);
    v &= T >= 32 ? 0xFFFFFFFFull : ((1ull << T) - 1);
    tokval[k] = (uint32_t)v;
    const uint32_t off = ob ? ((uint32_t)v & ((1u << ob) - 1u)) : 0u, len = ((uint32_t)v >> ob) & ((1u << lb) - 1u);
    len1[k] = len + 1u;
    /* a copy from distance 0: the reference's encoder emits it when -s is a power of two (the offset sb does
     * not fit its field); its decoder then reads whatever its cyclic buffer holds (lz77.c:178-181) */
    if (off == 0 && len > 0 && stale_flag) stale_flag[0] = 1u;
    /* a distance beyond the window (the field is wider than sb unless sb = 2^k - 1): never written by the reference's
     * encoder (tree.c:140: off <= sb); such a stream takes the general per-byte path, the window-bounded ones
     * (k_dec_seg's ring, the shard hand-off) assume off <= sb */
    if (off > sb && len > 0 && stale_flag) stale_flag[1] = 1u;
}


/* ---- distance-0 copies (power-of-two -s) --------------------------------------------------------
 * The reference's decoder stages its output in W = 3*SB+LA bytes; pass 0 starts at buffer index 0, every later pass at
 * index sb (lz77.c:172-175).  A copy from distance 0 re-reads the byte its buffer still holds at that index
 * (lz77.c:178-181): what an earlier pass left there, or calloc's zero.  Q.cyc[c] = output offset of the first byte of
 * pass c of the list, Q.cyc[ncyc] = the end of the output so far.  All offsets are in the coordinates of the buffer the
 * kernels work on: `pre` bytes of history (a stream decoded range by range: the bytes before the range, and -- Q.img --
 * the image of the reference's buffer at indices [sb, W) as the ranges before left it) followed by the range's output.
 * Pass 0 of the list may have begun before the range (cyc[0] < pre); Q.first0: it is pass 0 of the whole stream. */
__device__ __forceinline__ uint32_t dec_stale_b0(const lz77k_dec_stale &Q, uint32_t c, uint32_t sb) { return (c == 0 && Q.first0) ? 0u : sb; }

__device__ __forceinline__ uint32_t dec_stale_pass(const lz77k_dec_stale &Q, uint32_t d)
{
    uint32_t lo = 0, hi = Q.ncyc;                             /* last pass that starts at or before d */
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (Q.cyc[mid] <= d) lo = mid; else hi = mid; }
    return lo;
}

/* source of the stale read at buffer index idx during pass c: an offset into the working buffer, or `zero` */
__device__ __forceinline__ uint32_t dec_stale_src(const lz77k_dec_stale &Q, uint32_t c, uint32_t idx, uint32_t sb, uint32_t pre, uint32_t zero)
{
    for (uint32_t cc = c; cc-- > 0;) {
        const uint32_t b0 = dec_stale_b0(Q, cc, sb);
        if (idx < b0) break;                                  /* below sb every pass rewrites the index: never read stale */
        const uint32_t cand = Q.cyc[cc] + (idx - b0);
        if (cand < Q.cyc[cc + 1]) {                           /* pass cc got that far */
            if (cand >= pre) return cand;
            break;                                            /* ... before this range began: the image has it */
        }
    }
    return (Q.img != LZ77X_NONE32 && idx >= sb) ? Q.img + (idx - sb) : zero;
}

/* lz77.c:178-194 as data flow: every copied byte j points at j-off, every literal at itself.
 * Position n is a zero byte that degenerate tokens (off==0 or off>j) point at. */
__global__ void k_dec_expand(const uint32_t *__restrict__ tokval, const uint32_t *__restrict__ dst, uint32_t ntok,
                             int ob, int lb, uint8_t *__restrict__ out, uint32_t *__restrict__ ptr, uint32_t n /* pre + the range's bytes */,
                             lz77k_dec_stale Q, uint32_t sb, uint32_t pre /* bytes of history in front of the range's output */)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) { out[n] = 0; ptr[n] = n; }
    for (uint32_t j = k; j < pre; j += gridDim.x * blockDim.x) ptr[j] = j;        /* history: resolved bytes */
    if (k >= ntok) return;
    const uint32_t v = tokval[k];
    const uint32_t off = ob ? (v & ((1u << ob) - 1u)) : 0;
    const uint32_t len = (v >> ob) & ((1u << lb) - 1u);
    const uint32_t lit = (v >> (ob + lb)) & 0xFFu;
    const uint32_t j0 = dst[k] + pre;
    if (off == 0 && len > 0 && Q.cyc) {
        const uint32_t c = dec_stale_pass(Q, j0), idx0 = dec_stale_b0(Q, c, sb) + (j0 - Q.cyc[c]);
        for (uint32_t i = 0; i < len; i++) ptr[j0 + i] = dec_stale_src(Q, c, idx0 + i, sb, pre, n);
        out[j0 + len] = (uint8_t)lit;
        ptr[j0 + len] = j0 + len;
        return;
    }
    for (uint32_t i = 0; i < len; i++) {
        const uint32_t j = j0 + i;
        ptr[j] = (off > 0 && off <= j) ? j - off : n;
    }
    out[j0 + len] = (uint8_t)lit;
    ptr[j0 + len] = j0 + len;
}

/* pointer jumping, up to four hops per pass: ptr[j] <- ptr^4[j], stopping at the first literal, until
 * every byte points at a literal.  A chain of depth d shrinks to ~d/4 per pass; concurrent updates of
 * other entries only ever move them further along the same chain, so any interleaving is safe.
 * The first pass visits every byte and appends the ones that are still short of a literal to a work
 * list; later passes visit only the list of the pass before (most bytes resolve at once: passes 2..
 * touch a small fraction of ptr[]). */
#define JUMP_ITEMS 8
__global__ __launch_bounds__(256) void k_dec_jump(uint32_t *__restrict__ ptr, uint32_t total, const uint32_t *__restrict__ in_list,
                                                  uint32_t *__restrict__ out_list, uint32_t *__restrict__ out_count)
{
    __shared__ uint32_t wtot[4], bbase;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t below = (1ull << lane) - 1ull;
    for (uint32_t base = blockIdx.x * (256u * JUMP_ITEMS); base < total; base += gridDim.x * (256u * JUMP_ITEMS)) {
        uint32_t j[JUMP_ITEMS];
        uint64_t m[JUMP_ITEMS];
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < JUMP_ITEMS; u++) {
            const uint32_t idx = base + 256u * u + threadIdx.x;
            bool keep = false;
            j[u] = 0;
            if (idx < total) {
                j[u] = in_list ? in_list[idx] : idx;
                const uint32_t p = ptr[j[u]];
                if (p != j[u]) {
                    const uint32_t q = ptr[p];
                    if (q != p) {                             /* p is not a literal yet */
                        const uint32_t r = ptr[q];
                        if (r == q) {
                            ptr[j[u]] = q;                    /* q is: done, and it never enters a list */
                        } else {
                            const uint32_t t = ptr[r];
                            ptr[j[u]] = t;                    /* four hops */
                            keep = t != r;                    /* r was a literal <=> t == r <=> done */
                        }
                    }
                }
            }
            m[u] = __ballot(keep);
            mine += (uint32_t)__popcll(m[u]);
        }
        /* one atomic per 2048 entries: a single counter cannot take one per wavefront (1.5 M of them) */
        if (lane == 0) wtot[wave] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t t = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            bbase = t ? atomicAdd(out_count, t) : 0u;
        }
        __syncthreads();
        uint32_t slot = bbase;
        for (uint32_t w = 0; w < wave; w++) slot += wtot[w];
#pragma unroll
        for (int u = 0; u < JUMP_ITEMS; u++) {
            if ((m[u] >> lane) & 1ull) out_list[slot + (uint32_t)__popcll(m[u] & below)] = j[u];
            slot += (uint32_t)__popcll(m[u]);
        }
        __syncthreads();                                      /* wtot / bbase are reused by the next round */
    }
}

__global__ void k_dec_gather(uint8_t *__restrict__ out, const uint32_t *__restrict__ ptr, uint32_t n)
{
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const uint32_t p = ptr[j];
        if (p != j) out[j] = out[p];
    }
}

/* ------------------------------------------------------------------ tile-local decode ---
 *
 * The pointer-jumping decode above keeps a 4-byte pointer for EVERY output byte in HBM and chases it there
 * (67x the algorithmic bytes).  Most of a copy chain is short-range: this path resolves, per tile
 * of DT_TB output bytes, everything that stays inside the tile in LDS -- a byte is a literal, a pointer to an
 * earlier byte of the same tile, or a pointer out of the tile; pointer doubling in LDS leaves only literals
 * (written straight to out[]) and out-of-tile pointers.  Only those go to HBM (ptr32[], one bit per byte in
 * unres[]), and only they take part in the global jumping: an unresolved byte hops along unresolved bytes until
 * it points at a resolved one (whose value the tile pass already wrote and nothing changes any more), then one
 * gather.  unres[] is immutable after the tile pass, ptr32[j] only ever moves along j's own chain: any
 * interleaving is safe. */
#define DT_TB LZ77K_DEC_TILE_BYTES                               /* 5 B of LDS per byte: two workgroups per CU */
#define DT_BLOCK 1024
#define DT_INT 0u
#define DT_LIT 1u
#define DT_EXT 2u

/* tfirst[t] = the token whose bytes contain output offset t * DT_TB */
__global__ void k_dec_bounds(const uint32_t *__restrict__ dst, uint32_t ntok, uint32_t *__restrict__ tfirst, uint32_t pre)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ntok) return;
    const uint32_t a = dst[k] + pre, b = dst[k + 1] + pre;
    const uint32_t t = (a + DT_TB - 1u) / DT_TB;
    if ((uint64_t)t * DT_TB < b) tfirst[t] = k;
}

__global__ __launch_bounds__(DT_BLOCK) void k_dec_tile(const uint32_t *__restrict__ tokval, const uint32_t *__restrict__ dst, uint32_t ntok,
                                                       int ob, int lb, uint8_t *__restrict__ out, uint32_t *__restrict__ ptr32,
                                                       unsigned long long *__restrict__ unres, uint32_t n, uint32_t ntiles,
                                                       const uint32_t *__restrict__ tfirst, lz77k_dec_stale Q, uint32_t sb,
                                                       uint32_t pre /* bytes of history before the range's output: a multiple of DT_TB */)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t dt_smem[];
    uint32_t *val = reinterpret_cast<uint32_t *>(dt_smem);           /* local index | byte value | absolute source */
    uint8_t *tag = dt_smem + (size_t)DT_TB * 4;
    const uint32_t tid = threadIdx.x;
    const uint32_t t = blockIdx.x + pre / DT_TB;
    const uint32_t j0 = t * DT_TB, j1 = n - j0 < DT_TB ? n : j0 + DT_TB, cnt = j1 - j0;
    const uint32_t k0 = tfirst[t], k1 = t + 1 < ntiles ? tfirst[t + 1] + 1u : ntok;

    /* lz77.c:178-194 as data flow: copied byte j <- byte j - off, then the literal */
    for (uint32_t k = k0 + tid; k < k1; k += DT_BLOCK) {
        const uint32_t v = tokval[k];
        const uint32_t off = ob ? (v & ((1u << ob) - 1u)) : 0;
        const uint32_t len = (v >> ob) & ((1u << lb) - 1u);
        const uint32_t lit = (v >> (ob + lb)) & 0xFFu;
        const uint32_t d = dst[k] + pre;
        const bool stale = off == 0 && len > 0 && Q.cyc;
        uint32_t c = 0, idx0 = 0;
        if (stale) {
            /* a copy from distance 0 (power-of-two -s): the byte the reference's 3*SB+LA staging buffer still
             * holds at that index (lz77.c:172-181), see dec_stale_src */
            c = dec_stale_pass(Q, d);
            idx0 = dec_stale_b0(Q, c, sb) + (d - Q.cyc[c]);
        }
        const uint32_t ia = d < j0 ? j0 - d : 0u;                     /* first byte of the token inside the tile */
        for (uint32_t i = ia; i <= len; i++) {
            const uint32_t j = d + i;
            if (j >= j1) break;
            uint32_t tg = DT_LIT, vv = lit;
            if (i < len) {
                uint32_t src = n;                                     /* n = "a zero byte" (degenerate tokens) */
                if (stale) {
                    src = dec_stale_src(Q, c, idx0 + i, sb, pre, n);
                } else if (off > 0 && off <= j) {
                    src = j - off;
                }
                if (src == n) { tg = DT_LIT; vv = 0; }
                else if (src >= j0) { tg = DT_INT; vv = src - j0; }
                else { tg = DT_EXT; vv = src; }
            }
            val[j - j0] = vv;
            tag[j - j0] = (uint8_t)tg;
        }
    }
    __syncthreads();

    /* pointer doubling inside the tile: read, barrier, write -- (val, tag) change together */
    constexpr uint32_t PER = DT_TB / DT_BLOCK;
    for (;;) {
        uint32_t nv[PER], nt[PER];
        int any = 0;
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) {
            const uint32_t i = tid + q * DT_BLOCK;
            nt[q] = 0xFFu;
            nv[q] = 0;
            if (i < cnt && tag[i] == DT_INT) {
                const uint32_t sidx = val[i];
                nt[q] = tag[sidx];
                nv[q] = val[sidx];
                any |= nt[q] == DT_INT;
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) {
            const uint32_t i = tid + q * DT_BLOCK;
            if (nt[q] != 0xFFu) { val[i] = nv[q]; tag[i] = (uint8_t)nt[q]; }
        }
        if (!__syncthreads_or(any)) break;
    }

    /* out: resolved bytes (four per thread), pointers that leave the tile, and the bit that says which */
    for (uint32_t i = tid * 4; i < cnt; i += DT_BLOCK * 4) {
        uint32_t w = 0;
#pragma unroll
        for (uint32_t q = 0; q < 4; q++)
            if (i + q < cnt && tag[i + q] == DT_LIT) w |= (val[i + q] & 0xFFu) << (8 * q);
        if (i + 4 <= cnt) *reinterpret_cast<uint32_t *>(out + j0 + i) = w;
        else for (uint32_t q = 0; i + q < cnt; q++) out[j0 + i + q] = (uint8_t)(w >> (8 * q));
    }
    for (uint32_t i0 = (tid & ~63u); i0 < cnt; i0 += DT_BLOCK) {
        const uint32_t i = i0 + (tid & 63u);
        const bool ext = i < cnt && tag[i] == DT_EXT;
        if (ext) ptr32[j0 + i] = val[i];
        const unsigned long long m = __ballot(ext);
        if ((tid & 63u) == 0) unres[(j0 + i0) >> 6] = m;
    }
}

__device__ __forceinline__ bool dt_unres(const unsigned long long *__restrict__ unres, uint32_t j)
{
    return (unres[j >> 6] >> (j & 63u)) & 1ull;
}

/* one pass of the global jumping: in_list == null visits every unresolved byte (bitmap), later passes the list
 * of the pass before; up to four hops, stopping at the first resolved byte */
__global__ __launch_bounds__(256) void k_dec_jump2(uint32_t *__restrict__ ptr, const unsigned long long *__restrict__ unres, uint32_t total,
                                                   const uint32_t *__restrict__ in_list, uint32_t *__restrict__ out_list,
                                                   uint32_t *__restrict__ out_count)
{
    __shared__ uint32_t wtot[4], bbase;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t below = (1ull << lane) - 1ull;
    for (uint32_t base = blockIdx.x * (256u * JUMP_ITEMS); base < total; base += gridDim.x * (256u * JUMP_ITEMS)) {
        uint32_t j[JUMP_ITEMS];
        uint64_t m[JUMP_ITEMS];
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < JUMP_ITEMS; u++) {
            const uint32_t idx = base + 256u * u + threadIdx.x;
            bool keep = false;
            j[u] = 0;
            if (idx < total) {
                j[u] = in_list ? in_list[idx] : idx;
                if (in_list || dt_unres(unres, j[u])) {
                    const uint32_t p = ptr[j[u]];
                    if (dt_unres(unres, p)) {
                        const uint32_t q = ptr[p];
                        if (!dt_unres(unres, q)) ptr[j[u]] = q;
                        else {
                            const uint32_t r = ptr[q];
                            if (!dt_unres(unres, r)) ptr[j[u]] = r;
                            else {
                                const uint32_t tt = ptr[r];
                                ptr[j[u]] = tt;
                                keep = dt_unres(unres, tt);
                            }
                        }
                    }
                }
            }
            m[u] = __ballot(keep);
            mine += (uint32_t)__popcll(m[u]);
        }
        if (lane == 0) wtot[wave] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tt = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            bbase = tt ? atomicAdd(out_count, tt) : 0u;
        }
        __syncthreads();
        uint32_t slot = bbase;
        for (uint32_t w = 0; w < wave; w++) slot += wtot[w];
#pragma unroll
        for (int u = 0; u < JUMP_ITEMS; u++) {
            if ((m[u] >> lane) & 1ull) out_list[slot + (uint32_t)__popcll(m[u] & below)] = j[u];
            slot += (uint32_t)__popcll(m[u]);
        }
        __syncthreads();
    }
}

__global__ void k_dec_gather2(uint8_t *__restrict__ out, const uint32_t *__restrict__ ptr, const unsigned long long *__restrict__ unres,
                              uint32_t n)
{
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        if (dt_unres(unres, j)) out[j] = out[ptr[j]];
}

size_t lz77k_dec_tile_tmp_bytes(uint32_t n)
{
    const size_t ntiles = ((size_t)n + DT_TB - 1) / DT_TB;
    return (ntiles + 8) * 4 + 256 + ((size_t)n / 64 + 8) * 8;
}

/* tile pass: out[] (resolved bytes), d_ptr[] + bitmap (the others).  d_tmp: lz77k_dec_tile_tmp_bytes(n);
 * *d_unres receives the bitmap's address inside it. */
hipError_t lz77k_dec_tiles(const uint32_t *d_tokval, const uint32_t *d_dst, uint32_t ntok, const lz77x_geom &g, uint8_t *d_out,
                           uint32_t *d_ptr, uint32_t n, void *d_tmp, const unsigned long long **d_unres, hipStream_t s,
                           const lz77k_dec_stale &Q, uint32_t pre)
{
    if (n <= pre || ntok == 0) return hipSuccess;
    const uint32_t ntiles = (uint32_t)(((size_t)n + DT_TB - 1) / DT_TB), t0 = pre / DT_TB;
    uint32_t *tfirst = reinterpret_cast<uint32_t *>(d_tmp);
    unsigned long long *unres = reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(d_tmp) + (((size_t)ntiles + 8) * 4 + 255) / 256 * 256);
    *d_unres = unres;
    const size_t lds = (size_t)DT_TB * 5;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dec_tile), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (pre && (e = hipMemsetAsync(unres, 0, (size_t)pre / 8, s)) != hipSuccess) return e;      /* the history is resolved */
    hipLaunchKernelGGL(k_dec_bounds, dim3((ntok + 255) / 256), dim3(256), 0, s, d_dst, ntok, tfirst, pre);
    hipLaunchKernelGGL(k_dec_tile, dim3(ntiles - t0), dim3(DT_BLOCK), lds, s, d_tokval, d_dst, ntok, g.ob, g.lb, d_out, d_ptr, unres, n, ntiles,
                       tfirst, Q, (uint32_t)g.sb, pre);
    return hipGetLastError();
}

hipError_t lz77k_dec_jump2(uint32_t *d_ptr, const unsigned long long *d_unres, uint32_t total, const uint32_t *d_in_list,
                           uint32_t *d_out_list, uint32_t *d_out_count, hipStream_t s)
{
    if (total == 0) return hipSuccess;
    const uint32_t per = 256u * JUMP_ITEMS;
    const uint32_t blocks = min((total + per - 1) / per, 256u * 16u);
    hipLaunchKernelGGL(k_dec_jump2, dim3(blocks), dim3(256), 0, s, d_ptr, d_unres, total, d_in_list, d_out_list, d_out_count);
    return hipGetLastError();
}

hipError_t lz77k_dec_gather2(uint8_t *d_out, const uint32_t *d_ptr, const unsigned long long *d_unres, uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const uint32_t blocks = min((n + 255u) / 256u, 256u * 32u);
    hipLaunchKernelGGL(k_dec_gather2, dim3(blocks), dim3(256), 0, s, d_out, d_ptr, d_unres, n);
    return hipGetLastError();
}

/* ------------------------------------------------------------------ segment decode -----
 *
 * Copy chains are long (a byte of English text is on average 6 hops from its literal, one in five more than 12),
 * so most chains leave a 12 KB tile and the tile pass above still hands 60 % of the bytes to the global jumping.
 * But a hop reaches at most sb bytes back: a workgroup that walks a SEGMENT of the output front to back, a
 * DS_TS-byte step at a time, only ever needs the ROOTS (literal value, or "unknown") of the sb bytes before the
 * step -- a ring in LDS.  Per step: the step's tokens are expanded into the ring (a copy from before the step
 * takes the root of its source at once, a copy from inside the step becomes a ring pointer), a few rounds of
 * pointer doubling inside the 4 KB step settle the rest, the bytes go out.  Segments run concurrently: what lies
 * before a segment is unknown while it runs, so the sb bytes before it are seeded as symbolic references
 * EXT(i) = "byte i of the previous segment's tail"; the few bytes whose root is such a reference are flagged and
 * patched afterwards: the segments' tails (sb roots each) are resolved front to back by one workgroup, then one
 * pass fills the flagged bytes.  HBM traffic: tokens in, bytes out, 2 B per flagged byte -- no per-byte pointers.
 * For sb <= 8192 and streams without distance-0 copies; others take the tile pass. */
#define DS_TS 4096u
#define DS_R 16384u                                  /* ring slots: >= sb + DS_TS */
#define DS_BLOCK 512
#define DS_TAG 0xC000u
#define DS_LIT 0x0000u                               /* | byte value */
#define DS_INT 0x4000u                               /* | ring slot of the source (inside the current step) */
#define DS_EXT 0x8000u                               /* | index into the previous segment's tail */

__global__ void k_dec_bounds_ts(const uint32_t *__restrict__ dst, uint32_t ntok, uint32_t *__restrict__ tfirst)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ntok) return;
    const uint32_t a = dst[k], b = dst[k + 1];
    const uint32_t t = (a + DS_TS - 1u) / DS_TS;
    if ((uint64_t)t * DS_TS < b) tfirst[t] = k;
}

#define DS_PF 2                                      /* tokens per thread fetched a step ahead (2 * 512 cover text and random bytes) */
#define DS_MAX_STEPS 1024u                           /* steps per segment (segment <= 4 MB) */

/* ---- round 5: the walk reads the STREAM.  Until then k_dec_parse wrote a token word and a
 * length per token, a grid-wide scan turned the lengths into output offsets, and the walk read both back: 16 bytes of
 * HBM traffic per token beside the 3 of the stream, three kernels in front of the walk.  Tokens have a fixed width
 * (lz77.c:260-283: token k is bits [32 + kT, ..)), so the walk extracts its tokens from the stream itself, and the only
 * thing it cannot know alone -- where its first token's bytes go -- is one offset per 4 KB step: k_dec_sums adds up len + 1
 * per block of 2048 tokens (and raises the distance-0 / beyond-the-window flags k_dec_parse raised), a scan of those few
 * sums gives every block its base, k_dec_bounds_fused rescans inside the blocks and records, per step, the token that covers
 * the step's first byte and, per aligned group of 64 tokens, the offset of the group's first one (ntok / 16 bytes).  The walk
 * takes a step's tokens from the aligned group of its first one on, so a wavefront's 64 tokens are exactly one group: a
 * token's offset = its group's + an exclusive scan of len + 1 over the lanes below it (DPP row shifts and broadcasts: no
 * LDS, no barrier; the first version scanned 512 tokens across the workgroup through LDS -- one barrier a chunk and a
 * second array of per-step offsets that took the fourth workgroup off a CU: 0.30 -> 0.46 ms). */
#define DF_TPT 8u                                    /* tokens per thread of the sums / bounds kernels */
#define DF_BLOCK 256u

__device__ __forceinline__ uint32_t dec_token_at(const uint8_t *__restrict__ z, uint32_t k, int T)
{
    const uint64_t bit = 32 + (uint64_t)k * (uint64_t)T;
    const uint64_t v = ld64u(z + (bit >> 3)) >> (bit & 7);
    return (uint32_t)(T >= 32 ? v : v & ((1ull << T) - 1));
}

__global__ __launch_bounds__(DF_BLOCK) void k_dec_sums(const uint8_t *__restrict__ z, uint32_t ntok, int ob, int lb, int T, uint32_t sb,
                                                       uint32_t *__restrict__ bsum, uint32_t *__restrict__ stale_flag)
{
    __shared__ uint32_t ws[DF_BLOCK / 64];
    const uint32_t tid = threadIdx.x, k0 = (blockIdx.x * DF_BLOCK + tid) * DF_TPT;
    const uint32_t omask = ob ? (1u << ob) - 1u : 0u, lmask = (1u << lb) - 1u;
    uint32_t sum = 0, f0 = 0, f1 = 0;
#pragma unroll
    for (uint32_t i = 0; i < DF_TPT; i++) {
        const uint32_t k = k0 + i;
        const uint32_t v = dec_token_at(z, min(k, ntok - 1u), T);
        const uint32_t off = v & omask, len = (v >> ob) & lmask;
        if (k < ntok) {
            sum += len + 1u;
            f0 |= (off == 0 && len > 0) ? 1u : 0u;           /* a copy from distance 0 (power-of-two -s) */
            f1 |= (off > sb && len > 0) ? 1u : 0u;           /* a distance beyond the window: the general path */
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if ((tid & 63u) == 0) ws[tid >> 6] = sum;
    if (__ballot(f0 != 0) && (tid & 63u) == 0) stale_flag[0] = 1u;
    if (__ballot(f1 != 0) && (tid & 63u) == 0) stale_flag[1] = 1u;
    __syncthreads();
    if (tid == 0) bsum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(DF_BLOCK) void k_dec_bounds_fused(const uint8_t *__restrict__ z, uint32_t ntok, int ob, int lb, int T,
                                                               const uint32_t *__restrict__ bofs, uint32_t *__restrict__ tfirst,
                                                               uint32_t *__restrict__ d64 /* [g] = output offset of token 64 g */)
{
    __shared__ uint32_t ws[DF_BLOCK / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, k0 = (blockIdx.x * DF_BLOCK + tid) * DF_TPT;
    const uint32_t lmask = (1u << lb) - 1u;
    uint32_t l1[DF_TPT], sum = 0;
#pragma unroll
    for (uint32_t i = 0; i < DF_TPT; i++) {
        const uint32_t k = k0 + i;
        const uint32_t v = dec_token_at(z, min(k, ntok - 1u), T);
        l1[i] = k < ntok ? ((v >> ob) & lmask) + 1u : 0u;
        sum += l1[i];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += t;
    }
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    uint32_t a = bofs[blockIdx.x] + incl - sum;
    for (uint32_t w = 0; w < wave; w++) a += ws[w];
    if ((k0 & 63u) == 0 && k0 < ntok) d64[k0 >> 6] = a;
#pragma unroll
    for (uint32_t i = 0; i < DF_TPT; i++) {
        const uint32_t b = a + l1[i];
        const uint32_t t = (a + DS_TS - 1u) / DS_TS;
        if (l1[i] && (uint64_t)t * DS_TS < b) tfirst[t] = k0 + i;
        a = b;
    }
}

/* (four workgroups a CU = eight waves per SIMD: at most 64 VGPRs and 80 SGPRs -- the 64-bit token extraction took the fused
 * form to 66 / 106 and with them a workgroup off every CU; tools/kres.py) */
template <bool FUSED>
__global__ __launch_bounds__(DS_BLOCK, 8) __attribute__((amdgpu_num_sgpr(80))) void k_dec_seg(const uint32_t *__restrict__ tokval, const uint32_t *__restrict__ dst, uint32_t ntok,
                                                      int ob, int lb, uint8_t *__restrict__ out, uint16_t *__restrict__ ref16,
                                                      unsigned long long *__restrict__ flags, uint32_t n, uint32_t seg_bytes, uint32_t nseg,
                                                      const uint32_t *__restrict__ tfirst, uint32_t sb, uint16_t *__restrict__ tail,
                                                      uint32_t ext0 /* a shard of a stream cut by token ranges: the sb bytes before output byte 0
                                                                       exist elsewhere (EXT references like any segment's), and the last
                                                                       segment's tail is wanted too */,
                                                      const uint8_t *__restrict__ z /* FUSED: the stream itself */, int T,
                                                      const uint32_t *__restrict__ d64 /* FUSED: output offset of every 64th token */)
{
    __shared__ uint16_t ring[DS_R];
    __shared__ uint32_t s_tf[DS_MAX_STEPS + 2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t sg = blockIdx.x;
    const uint32_t a = sg * seg_bytes;
    const uint32_t b = sg + 1 == nseg ? n : a + seg_bytes;
    const uint32_t nsteps = (b - a + DS_TS - 1u) / DS_TS;
    if (sg > 0 || ext0)
        for (uint32_t i = tid; i < sb; i += DS_BLOCK) ring[(a - sb + i) & (DS_R - 1u)] = (uint16_t)(DS_EXT | i);
    /* first token of every step (and of the step after the last one, when there is one) */
    for (uint32_t i = tid; i <= nsteps; i += DS_BLOCK) {
        const bool in = (uint64_t)a + (uint64_t)i * DS_TS < n;
        s_tf[i] = in ? tfirst[a / DS_TS + i] : ntok;
    }

    __syncthreads();
    auto token = [&](uint32_t k) -> uint32_t {
        if constexpr (FUSED) return dec_token_at(z, k, T);
        else return tokval[k];
    };
    const uint32_t omask = ob ? (1u << ob) - 1u : 0u, lmask = (1u << lb) - 1u;
    auto krange = [&](uint32_t si, uint32_t &k0, uint32_t &k1) {
        const uint32_t ts = a + si * DS_TS, te = b - ts < DS_TS ? b : ts + DS_TS;
        k0 = s_tf[si];
        k1 = te < n ? s_tf[si + 1] + 1u : ntok;
    };
    uint32_t cv[DS_PF], cd[DS_PF], nv[DS_PF], nd[DS_PF];
    {
        uint32_t k0, k1;
        krange(0, k0, k1);
        if constexpr (FUSED) k0 &= ~63u;                     /* from the aligned group of the step's first token on */
#pragma unroll
        for (int q = 0; q < DS_PF; q++) {
            const uint32_t k = min(k0 + tid + q * DS_BLOCK, ntok - 1u);
            cv[q] = token(k);
            cd[q] = FUSED ? d64[k >> 6] : dst[k];
        }
    }
    for (uint32_t si = 0; si < nsteps; si++) {
        const uint32_t ts = a + si * DS_TS;
        const uint32_t te = b - ts < DS_TS ? b : ts + DS_TS;
        uint32_t k0, k1;
        krange(si, k0, k1);
        {
            /* the next step's tokens travel while this step runs (unconditional loads, clamped index) */
            uint32_t f0 = k0, f1 = k1;
            if (si + 1 < nsteps) krange(si + 1, f0, f1);
            if constexpr (FUSED) f0 &= ~63u;
#pragma unroll
            for (int q = 0; q < DS_PF; q++) {
                const uint32_t k = min(f0 + tid + q * DS_BLOCK, ntok - 1u);
                nv[q] = token(k);
                nd[q] = FUSED ? d64[k >> 6] : dst[k];
            }
        }
        /* lz77.c:178-194 as data flow, one thread per token */
        auto expand = [&](uint32_t v, uint32_t d) {
            const uint32_t off = v & omask, len = (v >> ob) & lmask, lit = (v >> (ob + lb)) & 0xFFu;
            const uint32_t ia = d < ts ? ts - d : 0u;
            for (uint32_t i = ia; i <= len; i++) {
                const uint32_t j = d + i;
                if (j >= te) break;
                uint32_t st = DS_LIT | lit;
                if (i < len) {
                    if (off == 0 || (off > j && !(ext0 && off - j <= sb))) st = DS_LIT;   /* degenerate: a zero byte */
                    else if (off > j) st = (uint32_t)ring[(j - off) & (DS_R - 1u)];      /* a shard: from the bytes before it */
                    else {
                        const uint32_t src = j - off;
                        st = src >= ts ? (DS_INT | (src & (DS_R - 1u))) : (uint32_t)ring[src & (DS_R - 1u)];
                    }
                }
                ring[j & (DS_R - 1u)] = (uint16_t)st;
            }
        };
        if constexpr (FUSED) {
            /* a wavefront's 64 tokens are one aligned group: offset = the group's + the lengths of the lanes below */
            const uint32_t kA = k0 & ~63u;
            auto place = [&](uint32_t v, uint32_t g0, uint32_t k) {
                const uint32_t l1 = k < ntok ? ((v >> ob) & lmask) + 1u : 0u;
                uint32_t incl = l1;
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false);      /* row_shr:1 */
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false);      /* row_shr:2 */
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false);      /* row_shr:4 */
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false);      /* row_shr:8 */
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false);      /* row_bcast:15 into rows 1 and 3 */
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false);      /* row_bcast:31 into rows 2 and 3 */
                if (k >= k0 && k < k1) expand(v, g0 + incl - l1);
            };
#pragma unroll
            for (int q = 0; q < DS_PF; q++)
                if (kA + q * DS_BLOCK < k1) place(cv[q], cd[q], kA + tid + q * DS_BLOCK);
            for (uint32_t kc = kA + DS_PF * DS_BLOCK; kc < k1; kc += DS_BLOCK) {
                const uint32_t k = kc + tid, kq = min(k, ntok - 1u);
                place(token(kq), d64[kq >> 6], k);
            }
        } else {
#pragma unroll
            for (int q = 0; q < DS_PF; q++)
                if (k0 + tid + q * DS_BLOCK < k1) expand(cv[q], cd[q]);
            for (uint32_t k = k0 + tid + DS_PF * DS_BLOCK; k < k1; k += DS_BLOCK) expand(tokval[k], dst[k]);
        }
        __syncthreads();
        /* pointer doubling inside the step (16-bit states: a racing reader sees a state further along the chain) */
        uint32_t pending = (1u << (DS_TS / DS_BLOCK)) - 1u;
        for (;;) {
            int any = 0;
#pragma unroll
            for (uint32_t q = 0; q < DS_TS / DS_BLOCK; q++) {
                const uint32_t p = ts + tid + q * DS_BLOCK;
                if ((pending >> q) & 1u) {
                    const uint32_t st = p < te ? (uint32_t)ring[p & (DS_R - 1u)] : 0u;
                    if ((st & DS_TAG) == DS_INT) {
                        const uint32_t s2 = ring[st & (DS_R - 1u)];
                        ring[p & (DS_R - 1u)] = (uint16_t)s2;
                        if ((s2 & DS_TAG) == DS_INT) any = 1; else pending &= ~(1u << q);
                    } else {
                        pending &= ~(1u << q);
                    }
                }
            }
            if (!__syncthreads_or(any)) break;
        }
#pragma unroll
        for (uint32_t q = 0; q < DS_TS / DS_BLOCK; q++) {
            const uint32_t p0 = ts + q * DS_BLOCK + wave * 64u;
            const uint32_t p = p0 + lane;
            const bool valid = p < te;
            const uint32_t st = valid ? ring[p & (DS_R - 1u)] : 0u;
            const bool ext = valid && (st & DS_TAG) == DS_EXT;
            if (valid) out[p] = ext ? (uint8_t)0 : (uint8_t)st;
            if (ext) ref16[p] = (uint16_t)(st & 0x3FFFu);
            const unsigned long long m = __ballot(ext);
            if (lane == 0 && p0 < te) flags[p0 >> 6] = m;
        }
        /* no barrier: the next step writes the slots of [te, te + DS_TS), which alias positions older than ts - sb */
#pragma unroll
        for (int q = 0; q < DS_PF; q++) { cv[q] = nv[q]; cd[q] = nd[q]; }
    }
    if (sg + 1 < nseg || ext0)
        for (uint32_t i = tid; i < sb; i += DS_BLOCK) tail[(size_t)sg * sb + i] = ring[(b - sb + i) & (DS_R - 1u)];
}

/* The segments' tails, front to back: tres[s][i] = value of byte i of the last sb bytes of segment s.
 * A tail is a map on the tail before it ("byte i = constant, or byte ref of the previous tail") and such maps
 * compose, so the chain over S segments runs as ~3*sqrt(S) sequential steps instead of S:
 *   compose: every group of G consecutive tails -> one map           (one workgroup per group, G steps)
 *   top    : the groups' maps in sequence -> values at every group end (one workgroup, S/G steps)
 *   replay : every group again, from the now known values before it   (one workgroup per group, G steps) */
#define DS_TAIL_MAX 8192u
__device__ __forceinline__ void ds_load_tail(uint16_t (&r)[8], const uint16_t *__restrict__ row, uint32_t sb)
{
#pragma unroll
    for (int q = 0; q < 8; q++) { const uint32_t i = threadIdx.x + 1024u * q; r[q] = row[min(i, sb - 1u)]; }
}

__global__ __launch_bounds__(1024) void k_dec_tails_compose(const uint16_t *__restrict__ tail, uint16_t *__restrict__ gmap, uint32_t ntails,
                                                            uint32_t G, uint32_t sb)
{
    __shared__ uint16_t cur[2][DS_TAIL_MAX];
    const uint32_t g = blockIdx.x, s0 = g * G, s1 = min(s0 + G, ntails);
    uint16_t st[8], nx[8];
    ds_load_tail(st, tail + (size_t)s0 * sb, sb);
    int w = 0;
    for (uint32_t s = s0; s < s1; s++) {
        ds_load_tail(nx, tail + (size_t)min(s + 1, s1 - 1u) * sb, sb);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t i = threadIdx.x + 1024u * q;
            if (i < sb) {
                const uint32_t x = st[q];
                cur[w][i] = (s > s0 && (x & DS_TAG) == DS_EXT) ? cur[w ^ 1][x & 0x3FFFu] : (uint16_t)x;
            }
        }
        __syncthreads();
        w ^= 1;
#pragma unroll
        for (int q = 0; q < 8; q++) st[q] = nx[q];
    }
    for (uint32_t i = threadIdx.x; i < sb; i += 1024u) gmap[(size_t)g * sb + i] = cur[w ^ 1][i];
}

/* rows[0..count) applied in sequence to the values `before` (null: nothing lies before -- row 0 holds no
 * reference); vout[r] = the values after row r */
__global__ __launch_bounds__(1024) void k_dec_tails_replay(const uint16_t *__restrict__ rows, uint32_t nrows, uint32_t G,
                                                           const uint8_t *__restrict__ before /* row g-1: the values before group g */,
                                                           uint8_t *__restrict__ vout, uint32_t sb,
                                                           const uint8_t *__restrict__ before0 /* a shard: the values before row 0, else null */)
{
    __shared__ uint8_t cur[2][DS_TAIL_MAX];
    const uint32_t g = blockIdx.x, r0 = g * G, r1 = min(r0 + G, nrows);
    if (r0 >= r1) return;
    if ((before != nullptr && g > 0) || (g == 0 && before0 != nullptr)) {
        const uint8_t *bsrc = g > 0 ? before + (size_t)(g - 1u) * sb : before0;
        for (uint32_t i = threadIdx.x; i < sb; i += 1024u) cur[1][i] = bsrc[i];
    }
    uint16_t st[8], nx[8];
    ds_load_tail(st, rows + (size_t)r0 * sb, sb);
    int w = 0;
    __syncthreads();
    for (uint32_t r = r0; r < r1; r++) {
        ds_load_tail(nx, rows + (size_t)min(r + 1, r1 - 1u) * sb, sb);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t i = threadIdx.x + 1024u * q;
            if (i < sb) {
                const uint32_t x = st[q];
                const uint8_t v = (x & DS_TAG) == DS_EXT ? cur[w ^ 1][x & 0x3FFFu] : (uint8_t)x;
                cur[w][i] = v;
                vout[(size_t)r * sb + i] = v;
            }
        }
        __syncthreads();
        w ^= 1;
#pragma unroll
        for (int q = 0; q < 8; q++) st[q] = nx[q];
    }
}

/* Flagged bytes take their value from the resolved tail of the segment before theirs: a workgroup per segment with the
 * segment's row of resolved bytes in LDS.  On text a quarter of all bytes are flagged (a copy chain is six hops on
 * average, one in five more than twelve: the first 25 KB of a 98 KB segment mostly lead back across its start); until
 * round 3 a grid-wide kernel fetched every one of them with a gather of its own from the row in HBM -- 64 sectors per
 * wave instruction, 0.17 ms, the address path's limit.  Here the row is 4 KB of LDS, the references of eight flag words
 * are in flight together and the bytes leave as whole lines. */
#define DPS_T 1024u                                  /* a wavefront's batches follow one another, two round trips each: many wavefronts per segment */
__global__ __launch_bounds__(DPS_T) void k_dec_patch_seg(uint8_t *__restrict__ out, const uint16_t *__restrict__ ref16,
                                                       const unsigned long long *__restrict__ flags,
                                                       const uint8_t *__restrict__ tres0, uint32_t n, uint32_t seg_bytes, uint32_t sb)
{
    __shared__ uint8_t row[DS_TAIL_MAX];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t sg = blockIdx.x;
    const uint32_t a = sg * seg_bytes, b = n - a < seg_bytes ? n : a + seg_bytes;
    for (uint32_t i = tid * 4u; i < sb; i += DPS_T * 4u)                /* (rows are sb bytes apart: no alignment to count on) */
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) if (i + q < sb) row[i + q] = tres0[(size_t)sg * sb + i + q];
    __syncthreads();
    const uint32_t w_lo = a >> 6, w_hi = (b + 63u) >> 6;               /* seg_bytes is a multiple of 4096 */
    const uint32_t wlast = w_hi - 1u;
    unsigned long long mn = flags[min(w_lo + wave * 8u + (lane & 7u), wlast)];          /* (unconditional, clamped: the next batch's flags travel with this batch's references) */
    for (uint32_t wb = w_lo + wave * 8u; wb < w_hi; wb += DPS_T / 8u) {
        const unsigned long long m = lane < 8u && wb + lane < w_hi ? mn : 0ull;
        mn = flags[min(wb + DPS_T / 8u + (lane & 7u), wlast)];
        if (!__ballot(m != 0ull)) continue;
        uint32_t r[8];
        bool on[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)m, k);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(m >> 32), k);
            const unsigned long long mm = ((unsigned long long)hi << 32) | lo;
            on[k] = (mm >> lane) & 1ull;
            r[k] = ref16[on[k] ? (wb + (uint32_t)k) * 64u + lane : a];      /* unconditional: the eight loads leave together */
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (on[k]) out[(wb + (uint32_t)k) * 64u + lane] = row[r[k]];
    }
}

/* la <= 255 (main.c:103): the step / tile bounds assume a token never spans more than one boundary */
int lz77k_dec_seg_supported(const lz77x_geom &g) { return g.sb <= 8192 && g.la <= 255; }

static void dec_seg_plan(uint32_t n, uint32_t *seg_bytes, uint32_t *nseg)
{
    /* about two segments per CU-slot (four 512-thread workgroups fit a CU), at least 64 KB each */
    const char *e = getenv("LZ77X_DECODE_SEGMENT");
    uint64_t sbytes = e && atoll(e) > 0 ? (uint64_t)atoll(e) : ((uint64_t)n + 1023u) / 1024u;
    if (sbytes < 65536u) sbytes = 65536u;
    if (sbytes > (uint64_t)DS_MAX_STEPS * DS_TS) sbytes = (uint64_t)DS_MAX_STEPS * DS_TS;
    sbytes = (sbytes + DS_TS - 1u) / DS_TS * DS_TS;
    *seg_bytes = (uint32_t)sbytes;
    *nseg = (uint32_t)(((uint64_t)n + sbytes - 1u) / sbytes);
}

size_t lz77k_dec_seg_tmp_bytes(uint32_t n, const lz77x_geom &g)
{
    uint32_t sbytes, nseg;
    dec_seg_plan(n, &sbytes, &nseg);
    return ((size_t)n / DS_TS + 8) * 4 + ((size_t)n / 64 + 8) * 4 /* (d64: a token is a byte at least) */ + ((size_t)n / 64 + 8) * 8 + (size_t)nseg * g.sb * 2 + ((size_t)nseg + 1) * g.sb +
           ((size_t)nseg / 8 + 40) * g.sb * 3 + (size_t)g.sb * 2 + 9 * 256;
}

/* The copy resolution in two phases, so that a stream cut by token ranges over several devices can exchange the sb bytes
 * between its shards in the middle:
 *   front: segment walk (flagged bytes + every segment's tail) and the composition of the tails by groups; ext0: the
 *          bytes before output byte 0 exist (another shard's): segment 0 starts from references too, the last tail is
 *          kept, and *d_smap receives the whole shard as ONE map (sb states: a byte value, or EXT | index into the
 *          incoming sb bytes) -- what the host chains from shard to shard;
 *   back : tails resolved front to back (ext0: from the incoming bytes the caller has put into P.tres0[0..sb)), flagged
 *          bytes patched. */
hipError_t lz77k_dec_segments_front(const uint32_t *d_tokval, const uint32_t *d_dst, uint32_t ntok, const lz77x_geom &g, uint8_t *d_out,
                                    void *d_ref, uint32_t n, void *d_tmp, hipStream_t s, bool ext0, lz77k_dec_seg_state &P,
                                    const uint16_t **d_smap, const uint8_t *d_z, const uint32_t *d_bofs)
{
    P = lz77k_dec_seg_state();
    if (n == 0 || ntok == 0) return hipSuccess;
    dec_seg_plan(n, &P.sbytes, &P.nseg);
    const uint32_t nseg = P.nseg, usb = (uint32_t)g.sb;
    uint8_t *base = reinterpret_cast<uint8_t *>(d_tmp);
    size_t o = 0;
    auto take = [&](size_t bytes) { uint8_t *q = base + o; o += (bytes + 255) & ~(size_t)255; return q; };
    P.tfirst = reinterpret_cast<uint32_t *>(take(((size_t)n / DS_TS + 8) * 4));
    P.flags = reinterpret_cast<unsigned long long *>(take(((size_t)n / 64 + 8) * 8));
    P.tail = reinterpret_cast<uint16_t *>(take((size_t)nseg * g.sb * 2));
    P.tres0 = take(((size_t)nseg + 1) * g.sb);              /* row s: the bytes before segment s */
    P.gmap = reinterpret_cast<uint16_t *>(take(((size_t)nseg / 8 + 40) * g.sb * 2));     /* sqrt(nseg) + 1 group maps */
    P.gres = take(((size_t)nseg / 8 + 40) * g.sb);
    P.smap = reinterpret_cast<uint16_t *>(take((size_t)g.sb * 2));
    P.ext0 = ext0;
    if (d_z) {
        /* the walk reads the stream: d_bofs[b] = output offset of the first token of block b (lz77k_dec_sums + a scan) */
        uint32_t *d64 = reinterpret_cast<uint32_t *>(take(((size_t)ntok / 64 + 8) * 4));
        const uint32_t nblk = (ntok + DF_BLOCK * DF_TPT - 1u) / (DF_BLOCK * DF_TPT);
        hipLaunchKernelGGL(k_dec_bounds_fused, dim3(nblk), dim3(DF_BLOCK), 0, s, d_z, ntok, g.ob, g.lb, g.T, d_bofs, P.tfirst, d64);
        hipLaunchKernelGGL(k_dec_seg<true>, dim3(nseg), dim3(DS_BLOCK), 0, s, (const uint32_t *)nullptr, (const uint32_t *)nullptr, ntok, g.ob, g.lb, d_out,
                           reinterpret_cast<uint16_t *>(d_ref), P.flags, n, P.sbytes, nseg, P.tfirst, usb, P.tail, ext0 ? 1u : 0u, d_z, g.T, d64);
    } else {
        hipLaunchKernelGGL(k_dec_bounds_ts, dim3((ntok + 255) / 256), dim3(256), 0, s, d_dst, ntok, P.tfirst);
        hipLaunchKernelGGL(k_dec_seg<false>, dim3(nseg), dim3(DS_BLOCK), 0, s, d_tokval, d_dst, ntok, g.ob, g.lb, d_out, reinterpret_cast<uint16_t *>(d_ref),
                           P.flags, n, P.sbytes, nseg, P.tfirst, usb, P.tail, ext0 ? 1u : 0u, (const uint8_t *)nullptr, g.T, (const uint32_t *)nullptr);
    }
    P.ntails = ext0 ? nseg : nseg - 1u;
    if (P.ntails) {
        uint32_t G = 1;
        while (G * G < P.ntails) G++;
        P.G = G;
        P.NG = (P.ntails + G - 1u) / G;
        if (P.NG > 1) hipLaunchKernelGGL(k_dec_tails_compose, dim3(P.NG), dim3(1024), 0, s, P.tail, P.gmap, P.ntails, G, usb);
        if (ext0) {
            if (P.NG > 1) hipLaunchKernelGGL(k_dec_tails_compose, dim3(1), dim3(1024), 0, s, P.gmap, P.smap, P.NG, P.NG, usb);
            else hipLaunchKernelGGL(k_dec_tails_compose, dim3(1), dim3(1024), 0, s, P.tail, P.smap, P.ntails, P.ntails, usb);
        }
    }
    if (d_smap) *d_smap = P.smap;
    return hipGetLastError();
}

hipError_t lz77k_dec_segments_back(const lz77x_geom &g, uint8_t *d_out, void *d_ref, uint32_t n, const lz77k_dec_seg_state &P, hipStream_t s)
{
    if (n == 0 || P.nseg == 0 || P.ntails == 0) return hipSuccess;
    const uint32_t usb = (uint32_t)g.sb;
    const uint8_t *before0 = P.ext0 ? P.tres0 : nullptr;
    uint8_t *tres = P.tres0 + usb;                           /* row s + 1: the tail of segment s, resolved */
    if (P.NG > 1) {
        hipLaunchKernelGGL(k_dec_tails_replay, dim3(1), dim3(1024), 0, s, P.gmap, P.NG, P.NG, (const uint8_t *)nullptr, P.gres, usb, before0);
        hipLaunchKernelGGL(k_dec_tails_replay, dim3(P.NG), dim3(1024), 0, s, P.tail, P.ntails, P.G, P.gres, tres, usb, before0);
    } else {
        hipLaunchKernelGGL(k_dec_tails_replay, dim3(1), dim3(1024), 0, s, P.tail, P.ntails, P.ntails, (const uint8_t *)nullptr, tres, usb, before0);
    }
    hipLaunchKernelGGL(k_dec_patch_seg, dim3(P.nseg), dim3(DPS_T), 0, s, d_out, reinterpret_cast<const uint16_t *>(d_ref), P.flags, P.tres0, n, P.sbytes,
                       usb);
    return hipGetLastError();
}

/* the whole copy resolution: d_out[0..n) from the tokens.  d_ref: 2n bytes; d_tmp: lz77k_dec_seg_tmp_bytes */
hipError_t lz77k_dec_segments(const uint32_t *d_tokval, const uint32_t *d_dst, uint32_t ntok, const lz77x_geom &g, uint8_t *d_out,
                              void *d_ref, uint32_t n, void *d_tmp, hipStream_t s)
{
    lz77k_dec_seg_state P;
    hipError_t e = lz77k_dec_segments_front(d_tokval, d_dst, ntok, g, d_out, d_ref, n, d_tmp, s, false, P, nullptr, nullptr, nullptr);
    if (e != hipSuccess) return e;
    return lz77k_dec_segments_back(g, d_out, d_ref, n, P, s);
}

/* d_bsum[b] = the bytes the tokens of block b (lz77k_dec_sum_block() tokens) decode to; d_bsum[nblk] = 0 (the scan's total slot) */
uint32_t lz77k_dec_sum_block(void) { return DF_BLOCK * DF_TPT; }

hipError_t lz77k_dec_sums(const uint8_t *d_z, uint32_t ntok, const lz77x_geom &g, uint32_t *d_bsum, uint32_t *d_stale_flag, hipStream_t s)
{
    if (ntok == 0) return hipSuccess;
    const uint32_t nblk = (ntok + DF_BLOCK * DF_TPT - 1u) / (DF_BLOCK * DF_TPT);
    hipError_t e = hipMemsetAsync(d_bsum + nblk, 0, 4, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_dec_sums, dim3(nblk), dim3(DF_BLOCK), 0, s, d_z, ntok, g.ob, g.lb, g.T, (uint32_t)g.sb, d_bsum, d_stale_flag);
    return hipGetLastError();
}

hipError_t lz77k_dec_parse(const uint8_t *d_z, uint32_t ntok, const lz77x_geom &g, uint32_t *d_tokval, uint32_t *d_len1, hipStream_t s,
                           uint32_t *d_stale_flag)
{
    if (ntok == 0) return hipSuccess;
    hipLaunchKernelGGL(k_dec_parse, dim3((ntok + 255) / 256), dim3(256), 0, s, d_z, ntok, g.ob, g.lb, g.T, (uint32_t)g.sb, d_tokval, d_len1, d_stale_flag);
    return hipGetLastError();
}

hipError_t lz77k_dec_expand(const uint32_t *d_tokval, const uint32_t *d_dst, uint32_t ntok, const lz77x_geom &g,
                            uint8_t *d_out, uint32_t *d_ptr, uint32_t n, hipStream_t s, const lz77k_dec_stale &Q, uint32_t pre)
{
    const uint32_t need = ntok > pre / 64u ? ntok : pre / 64u;
    hipLaunchKernelGGL(k_dec_expand, dim3(need ? (need + 255) / 256 : 1), dim3(256), 0, s, d_tokval, d_dst, ntok, g.ob, g.lb, d_out, d_ptr, n,
                       Q, (uint32_t)g.sb, pre);
    return hipGetLastError();
}

/* ---- a stream decoded range by range (lz77.c:160-195 through bounded memory): what one range leaves to the next ---- */

/* carry_new = the last cb bytes of [carry_old | out[0, n)) */
__global__ void k_dec_carry(const uint8_t *__restrict__ carry_old, const uint8_t *__restrict__ out, uint32_t n, uint32_t cb,
                            uint8_t *__restrict__ carry_new)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cb) return;
    const int64_t sidx = (int64_t)n - (int64_t)cb + (int64_t)i;
    carry_new[i] = sidx >= 0 ? out[sidx] : carry_old[(int64_t)cb + sidx];
}

hipError_t lz77k_dec_carry(const uint8_t *d_carry_old, const uint8_t *d_out, uint32_t n, uint32_t cb, uint8_t *d_carry_new, hipStream_t s)
{
    if (!cb) return hipSuccess;
    hipLaunchKernelGGL(k_dec_carry, dim3((cb + 255) / 256), dim3(256), 0, s, d_carry_old, d_out, n, cb, d_carry_new);
    return hipGetLastError();
}

/* the reference's staging buffer at indices [sb, W) after this range: per index the byte the latest pass that reached
 * it left there (dec_stale_src reads it in the ranges that follow); x = the working buffer, resolved */
__global__ void k_dec_image(const uint8_t *__restrict__ x, lz77k_dec_stale Q, uint32_t sb, uint32_t W, uint32_t pre,
                            const uint8_t *__restrict__ img_old, uint8_t *__restrict__ img_new)
{
    const uint32_t idx = sb + blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= W) return;
    uint8_t v = img_old[idx - sb];
    for (uint32_t cc = Q.ncyc; cc-- > 0;) {
        const uint32_t b0 = dec_stale_b0(Q, cc, sb);
        if (idx < b0) continue;
        const uint32_t cand = Q.cyc[cc] + (idx - b0);
        if (cand < Q.cyc[cc + 1]) {
            if (cand >= pre) v = x[cand];                     /* else: written before this range -- the old image holds it */
            break;
        }
    }
    img_new[idx - sb] = v;
}

hipError_t lz77k_dec_image(const uint8_t *d_x, const lz77k_dec_stale &Q, uint32_t sb, uint32_t W, uint32_t pre, const uint8_t *d_img_old,
                           uint8_t *d_img_new, hipStream_t s)
{
    hipLaunchKernelGGL(k_dec_image, dim3((W - sb + 255) / 256), dim3(256), 0, s, d_x, Q, sb, W, pre, d_img_old, d_img_new);
    return hipGetLastError();
}

/* One stream over several devices, windows the segment walk does not take (decode_sharded): a shard runs the
 * tile pass + jumping on [pre bytes of history | its output] BEFORE the history is known -- afterwards every byte either
 * holds its value or points at a byte that does (inside the shard) or at a byte of the history.  The shard's last cb
 * bytes as a map on the cb bytes before it: map[i] = a byte value, or 0x10000 | index into those bytes; the host chains
 * the shards' maps front to back (lz77x_shard_compose_tail32), every shard then receives its history and gathers. */
__global__ void k_dec_tail_map(const uint8_t *__restrict__ x, const uint32_t *__restrict__ ptr, const unsigned long long *__restrict__ unres,
                               uint32_t pre, uint32_t n /* pre + the shard's bytes */, uint32_t cb, uint32_t *__restrict__ map)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cb) return;
    const uint32_t j = n - cb + i;                           /* (n - pre >= cb: shorter shards decode on one device) */
    uint32_t v;
    if (dt_unres(unres, j)) {
        const uint32_t src = ptr[j];
        v = src < pre ? (src >= pre - cb ? 0x10000u | (src - (pre - cb)) : 0u /* before the first byte of the stream: a zero */) : (uint32_t)x[src];
    } else {
        v = x[j];
    }
    map[i] = v;
}

hipError_t lz77k_dec_tail_map(const uint8_t *d_x, const uint32_t *d_ptr, const unsigned long long *d_unres, uint32_t pre, uint32_t n, uint32_t cb,
                              uint32_t *d_map, hipStream_t s)
{
    hipLaunchKernelGGL(k_dec_tail_map, dim3((cb + 255) / 256), dim3(256), 0, s, d_x, d_ptr, d_unres, pre, n, cb, d_map);
    return hipGetLastError();
}

/* the largest multiple of eight k <= ntok with dst[k] <= cap (a range whose tokens expand past the output budget is cut
 * there: every range starts on a byte of the stream); res[0] = k, res[1] = dst[k] */
__global__ void k_dec_cut(const uint32_t *__restrict__ dst, uint32_t ntok, uint32_t cap, uint32_t *__restrict__ res)
{
    uint32_t lo = 0, hi = ntok / 8u;                          /* dst[8 * lo] <= cap (dst[0] = 0) */
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo + 1u) / 2u;
        if (dst[8u * mid] <= cap) lo = mid; else hi = mid - 1u;
    }
    res[0] = 8u * lo;
    res[1] = dst[8u * lo];
}

hipError_t lz77k_dec_cut(const uint32_t *d_dst, uint32_t ntok, uint32_t cap, uint32_t *d_res, hipStream_t s)
{
    hipLaunchKernelGGL(k_dec_cut, dim3(1), dim3(1), 0, s, d_dst, ntok, cap, d_res);
    return hipGetLastError();
}

hipError_t lz77k_dec_jump(uint32_t *d_ptr, uint32_t total, const uint32_t *d_in_list, uint32_t *d_out_list, uint32_t *d_out_count,
                          hipStream_t s)
{
    if (total == 0) return hipSuccess;
    const uint32_t per = 256u * JUMP_ITEMS;
    const uint32_t blocks = min((total + per - 1) / per, 256u * 16u);
    hipLaunchKernelGGL(k_dec_jump, dim3(blocks), dim3(256), 0, s, d_ptr, total, d_in_list, d_out_list, d_out_count);
    return hipGetLastError();
}

hipError_t lz77k_dec_gather(uint8_t *d_out, const uint32_t *d_ptr, uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const uint32_t blocks = min((n + 255u) / 256u, 256u * 32u);
    hipLaunchKernelGGL(k_dec_gather, dim3(blocks), dim3(256), 0, s, d_out, d_ptr, n);
    return hipGetLastError();
}

