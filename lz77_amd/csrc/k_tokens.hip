/*
 * k_tokens.hip -- token resolution and emission.
 *
 *   k_xfer_*      index of the priority hand-overs produced by the host stage, grouped by destination
 *   k_tokens*     tree.c:139-141 "first strictly longer on the search path wins": among the
 *                 equal-length candidates pick the one nearest the BST root (minimum priority)
 *   k_bidx_*      two-byte candidate index in global memory for large windows
 *   k_pack        lz77.c:246-252 writecode + bitio.c:203-239 bitIO_write as computed bit offsets
 */
#include "kernels_common.h"

/* ------------------------------------------------------------------ transfer index --- */

/* The host stage hands back xval[x] = priority moved from x to its successor S[x] when x is
 * evicted (or NONE).  Group these hand-overs by destination so that k_tokens can ask
 * "what priority did candidate c hold at time p". */
__global__ void k_xfer_count(const uint32_t *__restrict__ ps, const uint32_t *__restrict__ xval, uint32_t xa, uint32_t xb,
                             uint32_t dbase, uint32_t *__restrict__ cnt, uint32_t x_new, unsigned long long *__restrict__ total)
{
    uint32_t mine = 0;
    for (uint32_t x = xa + blockIdx.x * blockDim.x + threadIdx.x; x < xb; x += gridDim.x * blockDim.x) {
        if (xval[x] == LZ77X_NONE32) continue;
        mine += x >= x_new;
        const uint32_t dst = x + (ps[x] >> 16);
        if (dst >= dbase) atomicAdd(&cnt[dst - dbase], 1u);
    }
    if (total) {                                  /* one global atomic per workgroup */
        __shared__ uint32_t bsum;
        if (threadIdx.x == 0) bsum = 0;
        __syncthreads();
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
        if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&bsum, mine);
        __syncthreads();
        if (threadIdx.x == 0 && bsum) atomicAdd(total, (unsigned long long)bsum);
    }
}

__global__ void k_xfer_fill(const uint32_t *__restrict__ ps, const uint32_t *__restrict__ xval, uint32_t xa, uint32_t xb,
                            uint32_t dbase, uint32_t *__restrict__ ofs, uint2 *__restrict__ ent)
{
    for (uint32_t x = xa + blockIdx.x * blockDim.x + threadIdx.x; x < xb; x += gridDim.x * blockDim.x) {
        const uint32_t v = xval[x];
        if (v == LZ77X_NONE32) continue;
        const uint32_t dst = x + (ps[x] >> 16);
        if (dst >= dbase) ent[atomicAdd(&ofs[dst - dbase], 1u)] = make_uint2(x, v);
    }
}

/* The same index built destination block by destination block: a workgroup owns XF_DB consecutive
 * destinations; a hand-over x -> x + S[x] reaches at most sb - 1 positions ahead, so it only has to look at the
 * evictions of its own range and the sb before it (1.5x the range at sb 4095).  Counting and slot assignment
 * then happen in LDS; the only global pass over all destinations is the write of ofs[] itself.  (Round 1 used
 * one global atomic per hand-over, twice, plus a three-kernel scan over every destination: 4.1 ms per 100 MB.) */
#define XF_DB 8192u
#define XF_THREADS 1024

__global__ __launch_bounds__(XF_THREADS) void k_xfer_blocksum(const uint32_t *__restrict__ ps, const uint32_t *__restrict__ xval, uint32_t xa,
                                                             uint32_t xb, uint32_t dbase, uint32_t dend, uint32_t sb,
                                                             uint32_t *__restrict__ blocksum, uint32_t x_new,
                                                             unsigned long long *__restrict__ total)
{
    __shared__ uint32_t wsum[2 * XF_THREADS / 64];
    const uint32_t d0 = dbase + blockIdx.x * XF_DB, d1 = min(d0 + XF_DB, dend);
    const uint32_t xlo = max(xa, d0 >= sb ? d0 - sb + 1u : 0u), xhi = min(xb, d1);
    uint32_t cnt = 0, fresh = 0;
    for (uint32_t x = xlo + threadIdx.x; x < xhi; x += XF_THREADS) {
        if (xval[x] == LZ77X_NONE32) continue;
        const uint32_t dst = x + (ps[x] >> 16);
        if (dst >= d0 && dst < d1) { cnt++; fresh += x >= x_new; }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { cnt += __shfl_xor(cnt, d, 64); fresh += __shfl_xor(fresh, d, 64); }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { wsum[wave] = cnt; wsum[XF_THREADS / 64 + wave] = fresh; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0, f = 0;
        for (int w = 0; w < XF_THREADS / 64; w++) { c += wsum[w]; f += wsum[XF_THREADS / 64 + w]; }
        blocksum[blockIdx.x] = c;
        if (total && f) atomicAdd(total, (unsigned long long)f);
    }
}

__global__ __launch_bounds__(XF_THREADS) void k_xfer_blockfill(const uint32_t *__restrict__ ps, const uint32_t *__restrict__ xval, uint32_t xa,
                                                              uint32_t xb, uint32_t dbase, uint32_t dend, uint32_t sb,
                                                              const uint32_t *__restrict__ blockbase, uint32_t *__restrict__ ofs,
                                                              uint2 *__restrict__ ent)
{
    __shared__ uint32_t cnt[XF_DB];
    __shared__ uint32_t wsum[XF_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t d0 = dbase + blockIdx.x * XF_DB, d1 = min(d0 + XF_DB, dend);
    const uint32_t xlo = max(xa, d0 >= sb ? d0 - sb + 1u : 0u), xhi = min(xb, d1);
    for (uint32_t i = tid; i < XF_DB; i += XF_THREADS) cnt[i] = 0;
    __syncthreads();
    for (uint32_t x = xlo + tid; x < xhi; x += XF_THREADS) {
        if (xval[x] == LZ77X_NONE32) continue;
        const uint32_t dst = x + (ps[x] >> 16);
        if (dst >= d0 && dst < d1) atomicAdd(&cnt[dst - d0], 1u);
    }
    __syncthreads();
    {
        /* exclusive scan of the XF_DB counters (eight per thread), turned into slot cursors; ofs[] gets the END of every list */
        constexpr int PER = XF_DB / XF_THREADS;
        uint32_t c[PER], mine = 0;
#pragma unroll
        for (int q = 0; q < PER; q++) { c[q] = cnt[PER * tid + q]; mine += c[q]; }
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t run = blockbase[blockIdx.x] + incl - mine;
        for (uint32_t w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            cnt[PER * tid + q] = run;                          /* cursor: first slot of this destination's list */
            run += c[q];
            const uint32_t d = d0 + PER * tid + q;
            if (d < d1) ofs[d - dbase] = run;
        }
    }
    __syncthreads();
    for (uint32_t x = xlo + tid; x < xhi; x += XF_THREADS) {
        const uint32_t v = xval[x];
        if (v == LZ77X_NONE32) continue;
        const uint32_t dst = x + (ps[x] >> 16);
        if (dst >= d0 && dst < d1) ent[atomicAdd(&cnt[dst - d0], 1u)] = make_uint2(x, v);
    }
}

/* Index of the hand-overs of evictions x in [xa, xb) into destinations [dbase, dend):
 * afterwards list(c) = ent[ (c > dbase ? ofs[c-dbase-1] : 0) .. ofs[c-dbase] ).  sb: a hand-over reaches less
 * than sb positions ahead (0 = unknown: the round-1 kernels with global atomics). */
hipError_t lz77k_xfer_index(const uint32_t *d_ps, const uint32_t *d_xval, uint32_t xa, uint32_t xb, uint32_t dbase, uint32_t dend,
                            uint32_t *d_ofs, uint2 *d_ent, void *d_scan_tmp, hipStream_t s, uint32_t x_new,
                            unsigned long long *d_total, uint32_t sb)
{
    const uint32_t nd = dend - dbase;
    if (sb && xb > xa && !LZ77X_VENV("LZ77X_XFER_V1")) {
        const uint32_t nblocks = (nd + XF_DB - 1u) / XF_DB;
        uint32_t *sums = reinterpret_cast<uint32_t *>(d_scan_tmp);
        void *tmp2 = sums + ((nblocks + 64u) & ~63u);
        hipLaunchKernelGGL(k_xfer_blocksum, dim3(nblocks), dim3(XF_THREADS), 0, s, d_ps, d_xval, xa, xb, dbase, dend, sb, sums, x_new, d_total);
        hipError_t e = lz77k_scan_u32(sums, sums, nblocks, tmp2, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_xfer_blockfill, dim3(nblocks), dim3(XF_THREADS), 0, s, d_ps, d_xval, xa, xb, dbase, dend, sb, sums, d_ofs, d_ent);
        /* ofs[nd] (one past the last destination) = total, as the tile kernel's LIST_END(t1 - 1) may read up to ofs[nd-1] only: not needed */
        return hipGetLastError();
    }
    hipError_t e = hipMemsetAsync(d_ofs, 0, ((size_t)nd + 1) * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    if (xb <= xa) return hipSuccess;
    const uint32_t blocks = min((xb - xa + 255u) / 256u, 256u * 16u);
    hipLaunchKernelGGL(k_xfer_count, dim3(min(blocks, 1024u)), dim3(256), 0, s, d_ps, d_xval, xa, xb, dbase, d_ofs, x_new, d_total);
    e = lz77k_scan_u32(d_ofs, d_ofs, nd + 1, d_scan_tmp, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_xfer_fill, dim3(blocks), dim3(256), 0, s, d_ps, d_xval, xa, xb, dbase, d_ofs, d_ent);
    return hipGetLastError();
}

/* ------------------------------------------------------------------ k_tokens --------- */

/* One wave per token.  Lanes stride the SB candidates; a candidate qualifies when it shares
 * the first len bytes with p (len is already the maximum, so lcp == len); its priority at
 * time p is the value of the latest hand-over into it that happened before p, else its own
 * position.  The wave min over (priority, position) is the node nearest the BST root. */
__global__ __launch_bounds__(256) void k_tokens(const uint8_t *__restrict__ in, uint32_t n, int sb, int ob, int lb,
                                                const uint32_t *__restrict__ chain, uint32_t ntok,
                                                const uint8_t *__restrict__ maxlen,
                                                const uint32_t *__restrict__ ofs, const uint2 *__restrict__ ent, uint32_t dbase,
                                                uint32_t *__restrict__ tokval, const uint32_t *__restrict__ look, uint32_t nlook,
                                                uint32_t voff)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= ntok) return;
    const uint32_t p = chain[k];
    const uint32_t len = maxlen[p];
    const uint32_t next = in[p + len];
    uint32_t off = 0;
    if (len > 0) {
        const uint8_t *q = in + p;
        const uint32_t head = ld32u(q);
        const uint32_t hmask = len >= 4 ? 0xFFFFFFFFu : (1u << (8 * len)) - 1u;
        const uint32_t c0 = p > (uint32_t)sb ? p - (uint32_t)sb : 0;
        uint64_t best = ~0ull;
        for (uint32_t c = c0 + lane; c < p; c += 64) {
            const uint8_t *r = in + c;
            if ((ld32u(r) ^ head) & hmask) continue;
            bool same = true;
            for (uint32_t i = 4; i < len; i += 4) {
                uint32_t x = ld32u(r + i) ^ ld32u(q + i);
                const uint32_t rem = len - i;
                if (rem < 4) x &= (1u << (8 * rem)) - 1u;
                if (x) { same = false; break; }
            }
            if (!same) continue;
            uint32_t prio = c < nlook ? look[c] : c + voff;         /* a cell's own priority, or what it held when this segment began */
            const uint32_t lo = c > dbase ? ofs[c - dbase - 1] : 0, hi = c >= dbase ? ofs[c - dbase] : 0;
            uint32_t latest = 0;
            bool any = false;
            for (uint32_t e = lo; e < hi; e++) {
                const uint2 t = ent[e];
                /* hand-over at eviction of t.x happens after the match at time t.x+sb */
                if ((uint64_t)t.x + (uint32_t)sb < p && (!any || t.x > latest)) { any = true; latest = t.x; prio = t.y; }
            }
            const uint64_t key = ((uint64_t)prio << 32) | c;
            best = key < best ? key : best;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)best, d, 64);
            best = o < best ? o : best;
        }
        off = p - (uint32_t)(best & 0xFFFFFFFFu);
    }
    if (lane == 0) {
        const uint32_t omask = ob >= 32 ? 0xFFFFFFFFu : (1u << ob) - 1u;
        tokval[k] = (off & omask) | (len << ob) | (next << (ob + lb));      /* lz77.c:249-251 */
    }
}

/* ---- tiled variant: window bytes and the hand-over lists of a tile staged in LDS ---- */

#ifndef TOK_TILE
#define TOK_TILE 2048u
#endif
#ifndef TOK_BLOCK
#define TOK_BLOCK 1024
#endif

__global__ void k_tok_bounds(const uint32_t *__restrict__ chain, uint32_t ntok, uint32_t pos0, uint32_t ntiles,
                             uint32_t *__restrict__ tstart)
{
    /* tstart[t] = first token whose position is >= pos0 + t*TOK_TILE (tiles may be empty) */
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > ntok) return;
    const uint32_t first = k == 0 ? 0u : (chain[k - 1] - pos0) / TOK_TILE + 1;
    const uint32_t last = k == ntok ? ntiles : (chain[k] - pos0) / TOK_TILE;
    for (uint32_t t = first; t <= last; t++) tstart[t] = k;
}

/* Bucket of a candidate = its first byte (major) and three mixed bits of its second byte: a token of
 * length >= 2 visits one bucket, a token of length 1 the eight adjacent buckets of its first byte --
 * on incompressible input 94 % of the tokens have length 1 and used to scan the whole window. */
#define TOK_HASH 2048u
__device__ __forceinline__ uint32_t tok_hash(uint32_t b0, uint32_t b1) { return (b0 << 3) | ((b1 ^ (b1 >> 3)) & 7u); }

/*
 * One workgroup per tile of TOK_TILE positions.  Staged in LDS: the window bytes, the hand-over
 * lists of every candidate position, and (BUCKET) an index of the candidate positions by a hash
 * of their first two bytes, so that a token of length >= 2 only visits the handful of window
 * positions that start with its own two bytes instead of all SB of them.
 */
template <bool BUCKET>
__global__ __launch_bounds__(TOK_BLOCK) void k_tokens_tile(const uint8_t *__restrict__ in, uint32_t n, int sb, int la, int ob, int lb,
                                                           const uint32_t *__restrict__ chain, const uint32_t *__restrict__ tstart,
                                                           const uint8_t *__restrict__ maxlen,
                                                           const uint32_t *__restrict__ ofs, const uint2 *__restrict__ ent,
                                                           uint32_t dbase, uint32_t pos0, uint32_t pos1,
                                                           uint32_t *__restrict__ tokval, uint32_t ent_cap, uint32_t lofs_off,
                                                           uint32_t lent_off, uint32_t bkt_off, const uint32_t *__restrict__ look,
                                                           uint32_t nlook, uint32_t voff)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *by = smem;
    uint16_t *lofs = reinterpret_cast<uint16_t *>(smem + lofs_off);
    uint2 *lent = reinterpret_cast<uint2 *>(smem + lent_off);
    uint32_t *bstart = reinterpret_cast<uint32_t *>(smem + bkt_off);          /* TOK_HASH + 1 (+pad) */
    uint16_t *blist = reinterpret_cast<uint16_t *>(bstart + TOK_HASH + 8);     /* one entry per candidate */
    uint32_t *bcur = reinterpret_cast<uint32_t *>(lent);                       /* TOK_HASH words, dead before lent is staged */
    __shared__ uint32_t wsum[TOK_BLOCK / 64];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t usb = (uint32_t)sb;
    const uint32_t t0 = pos0 + blockIdx.x * TOK_TILE;
    const uint32_t t1 = min(t0 + TOK_TILE, pos1);
    const uint32_t wbase = (t0 > usb ? t0 - usb : 0u) & ~3u;
#define LIST_START(c) ((c) > dbase ? ofs[(c) - dbase - 1] : 0u)
#define LIST_END(c) ((c) >= dbase ? ofs[(c) - dbase] : 0u)
    const uint32_t nb = (t1 + (uint32_t)la + 8 - wbase + 3) & ~3u;
    for (uint32_t i = tid * 4; i < nb; i += TOK_BLOCK * 4)
        *reinterpret_cast<uint32_t *>(by + i) = *reinterpret_cast<const uint32_t *>(in + wbase + i);
    const uint32_t NO = t1 - wbase;
    const uint32_t ebase = LIST_START(wbase);
    const uint32_t ecount = LIST_END(t1 - 1) - ebase;
    const bool staged = ecount <= ent_cap && ecount < 65536u;
    if (BUCKET)
        for (uint32_t i = tid; i < TOK_HASH; i += TOK_BLOCK) bcur[i] = 0;
    __syncthreads();
    if (BUCKET) {
        /* counting sort of the candidate positions by bucket */
        for (uint32_t i = tid; i < NO; i += TOK_BLOCK) atomicAdd(&bcur[tok_hash(by[i], by[i + 1])], 1u);
        __syncthreads();
        {
            constexpr int BPT = TOK_HASH / TOK_BLOCK;         /* buckets per thread */
            static_assert(BPT * TOK_BLOCK == TOK_HASH && BPT >= 1, "whole buckets per thread");
            uint32_t c[BPT], mine = 0;
#pragma unroll
            for (int q = 0; q < BPT; q++) { c[q] = bcur[BPT * tid + q]; mine += c[q]; }
            uint32_t incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if (lane >= (uint32_t)d) incl += t;
            }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            uint32_t woff = 0;
            for (uint32_t w = 0; w < wave; w++) woff += wsum[w];
            uint32_t excl = woff + incl - mine;
#pragma unroll
            for (int q = 0; q < BPT; q++) { bstart[BPT * tid + q] = excl; excl += c[q]; }
            excl -= mine;
            if (tid == TOK_BLOCK - 1) bstart[TOK_HASH] = excl + mine;
        }
        __syncthreads();
        for (uint32_t i = tid; i < TOK_HASH; i += TOK_BLOCK) bcur[i] = bstart[i];
        __syncthreads();
        for (uint32_t i = tid; i < NO; i += TOK_BLOCK) blist[atomicAdd(&bcur[tok_hash(by[i], by[i + 1])], 1u)] = (uint16_t)i;
        __syncthreads();                                      /* bcur is dead: its space becomes the hand-over lists */
    }
    if (staged) {
        for (uint32_t i = tid; i <= NO; i += TOK_BLOCK) {
            const uint32_t c = wbase + i;
            lofs[i] = (uint16_t)(LIST_START(c) - ebase);
        }
        for (uint32_t e = tid; e < ecount; e += TOK_BLOCK) lent[e] = ent[ebase + e];
    }
    __syncthreads();

    const uint32_t omask = ob >= 32 ? 0xFFFFFFFFu : (1u << ob) - 1u;
    const uint32_t k0 = tstart[blockIdx.x], k1 = tstart[blockIdx.x + 1];
    /* each wave takes tokens k0+wave, k0+wave+8, ...; their (position, length) are fetched 64 at a
     * time, one per lane, so that the per-token loop never waits on global memory */
#ifdef TOK_SETUP_ONLY                                         /* timing ablation: staging + index build, no tokens */
    if (k1 != 0xFFFFFFFFu) return;
#endif
    for (uint32_t kb = k0 + wave; kb < k1; kb += 64 * (TOK_BLOCK / 64)) {
        const uint32_t kmine = kb + lane * (TOK_BLOCK / 64);
        uint32_t p_l = 0, len_l = 0;
        if (kmine < k1) { p_l = chain[kmine]; len_l = maxlen[p_l]; }
        for (uint32_t j = 0; j < 64; j++) {
            const uint32_t k = kb + j * (TOK_BLOCK / 64);
            if (k >= k1) break;
            /* j is wave-uniform: v_readlane, not a ds_bpermute round trip */
            const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)p_l, (int)j);
            const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)len_l, (int)j);
            const uint32_t qo = p - wbase;
            const uint32_t next = by[qo + len];
            uint32_t off = 0;
            if (len > 0) {
                const uint32_t cmin = p > usb ? p - usb : 0u;
                uint64_t best = ~0ull;
                /* the token's own first 16 bytes, once per token (every lane compares against them) */
                uint32_t qw[4];
#pragma unroll
                for (int i = 0; i < 4; i++) qw[i] = ld32_at<true>(by, qo + 4 * i);
                /* candidate c shares len bytes with p?  then its priority at time p */
                auto consider = [&](uint32_t c) {
                    const uint32_t co = c - wbase;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if ((uint32_t)(4 * i) < len) {
                            uint32_t x = ld32_at<true>(by, co + 4 * i) ^ qw[i];
                            const uint32_t rem = len - 4 * i;
                            if (rem < 4) x &= (1u << (8 * rem)) - 1u;
                            if (x) return;
                        }
                    }
                    for (uint32_t i = 16; i < len; i += 4) {
                        uint32_t x = ld32_at<true>(by, co + i) ^ ld32_at<true>(by, qo + i);
                        const uint32_t rem = len - i;
                        if (rem < 4) x &= (1u << (8 * rem)) - 1u;
                        if (x) return;
                    }
                    uint32_t prio = c < nlook ? look[c] : c + voff, latest = 0;
                    bool any = false;
                    if (staged) {
                        for (uint32_t e = lofs[co]; e < lofs[co + 1]; e++) {
                            const uint2 t = lent[e];
                            if ((uint64_t)t.x + usb < p && (!any || t.x > latest)) { any = true; latest = t.x; prio = t.y; }
                        }
                    } else {
                        for (uint32_t e = LIST_START(c); e < LIST_END(c); e++) {
                            const uint2 t = ent[e];
                            if ((uint64_t)t.x + usb < p && (!any || t.x > latest)) { any = true; latest = t.x; prio = t.y; }
                        }
                    }
                    const uint64_t key = ((uint64_t)prio << 32) | c;
                    best = key < best ? key : best;
                };
                if (BUCKET) {
                    const uint32_t h = len >= 2 ? tok_hash(qw[0] & 0xFFu, (qw[0] >> 8) & 0xFFu) : (qw[0] & 0xFFu) << 3;
                    const uint32_t e1 = bstart[h + (len >= 2 ? 1u : 8u)];
                    for (uint32_t i = bstart[h] + lane; i < e1; i += 64) {
                        const uint32_t c = wbase + blist[i];
                        if (c >= cmin && c < p) consider(c);
                    }
                } else {
                    const uint32_t head = qw[0];
                    const uint32_t hmask = len >= 4 ? 0xFFFFFFFFu : (1u << (8 * len)) - 1u;
                    for (uint32_t cg = (cmin & ~3u) + lane * 4; cg < p; cg += 256) {
                        const uint8_t *r = by + (cg - wbase);
                        const uint32_t lo = *reinterpret_cast<const uint32_t *>(r);
                        const uint32_t hi = *reinterpret_cast<const uint32_t *>(r + 4);
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) {
                            const uint32_t c = cg + jj;
                            const uint32_t w = jj == 0 ? lo : __builtin_amdgcn_alignbyte(hi, lo, jj);
                            if (((w ^ head) & hmask) == 0 && c >= cmin && c < p) consider(c);
                        }
                    }
                }
                /* wave minimum of (priority, position).  Usually a handful of lanes hold a candidate:
                 * read those lanes directly; the xor butterfly (12 bpermute round trips) only when many do */
                uint64_t have = __ballot(best != ~0ull);
                if (__popcll(have) <= 8) {
                    uint64_t m = ~0ull;
                    for (; have; have &= have - 1) {
                        const int l = __builtin_ctzll(have);
                        const uint64_t v = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(best >> 32), l) << 32) |
                                           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)best, l);
                        m = v < m ? v : m;
                    }
                    best = m;
                } else {
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) {
                        const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)best, d, 64);
                        best = o < best ? o : best;
                    }
                }
                off = p - (uint32_t)(best & 0xFFFFFFFFu);
            }
            if (lane == 0) tokval[k] = (off & omask) | (len << ob) | (next << (ob + lb));
        }
    }
}

#undef LIST_START
#undef LIST_END

/* ---- production for LDS-sized windows: candidates enumerated as RUNS OF THE SORTED ORDER ------------------
 *
 * The candidates whose match with p has the full length len are exactly the window positions that share p's
 * first len bytes, and positions sharing a prefix are contiguous in (key, position) order -- the order the
 * match stage has already computed for every region (k_match keeps it: `order_all`, RP uint16 per region).
 * A tile of TS_TT token positions [a, b) lies inside ONE region together with its whole look-back, so the
 * workgroup filters the region's order down to the cells of [a - sb, b) (a compaction, no sort), and then
 *   A. per token (two lanes: downwards / upwards from p's own slot) gallops to the ends of the run of cells
 *      that share its len bytes -- a handful of probes, every lane busy with its own token;
 *   B. the runs of a batch of tokens are laid end to end and dealt to the 1024 threads in equal contiguous
 *      pieces: per item one window test and one priority look-up (hand-over lists in LDS), minimum per token
 *      through a 64-bit LDS atomic;
 *   C. per token, offset = p - argmin.
 * Compared with the bucket kernel above (one wave per token, 64 lanes scanning a hash bucket of which a
 * handful qualify; VALU-bound at ~130 wave instructions per token) every lane does useful work and only
 * exact candidates are visited (17 per token on text against a bucket of 95).
 *
 * Tile grid: region r (positions [r*TILE, r*TILE + TILE + sb) sorted) owns the backward windows of
 * y in [r*TILE + sb, (r+1)*TILE + sb), cut into ceil(TILE / TS_TT) tiles; region 0 also owns y < sb
 * ("head" tiles). */
#define TS_TT 3072u
#define TS_BLOCK 1024
#define TS_TB 512u                                   /* tokens per batch (two lanes each in phase A) */
/* A token whose run holds this many cells or more is not dealt member by member.  Round 4: all-zero input spent 29 of its
 * 37 ms per 100 MB there -- every token's run is the whole window, 4095 members each.  A member's own priority IS its
 * position (tree.c:102-105: a new node is a leaf; + voff), so among the members' own priorities the winner is simply the
 * OLDEST member in the window: walk the window's positions upwards from p - sb until one's slot lies in the run -- sb / run
 * positions on average instead of `run` members, sixteen lanes a token; what hand-overs lowered comes in through the
 * entries like everybody's (k_tokens_sorted, B2).  Round 4's form (a bitmap of the slots with hand-overs, threshold 1024,
 * a density test) is in k_tokens_sorted_v4.  Measured on text, tie-break ms per 100 MB: 1024 2.81, 128 2.80, 64 2.86,
 * 32 3.21; low-entropy data 5.1 at 1024, 4.3 at 64. */
#define TS_BIG 128u
#define TS_BIG_V4 1024u                                               /* round 4's kernel (variants build) */

struct ts_grid { uint32_t sb, TILE, head, tpr; };

static inline ts_grid ts_make_grid(const lz77x_geom &g)
{
    ts_grid G;
    G.sb = (uint32_t)g.sb;
    G.TILE = g.TILE;
    G.head = (G.sb + TS_TT - 1u) / TS_TT;
    G.tpr = (g.TILE + TS_TT - 1u) / TS_TT;
    return G;
}

__host__ __device__ __forceinline__ uint32_t ts_tile_of(const ts_grid &G, uint32_t pos)
{
    if (pos < G.sb) return pos / TS_TT;
    const uint32_t q = pos - G.sb, r = q / G.TILE;
    return G.head + r * G.tpr + (q - r * G.TILE) / TS_TT;
}

__device__ __forceinline__ void ts_tile_range(const ts_grid &G, uint32_t idx, uint32_t &a, uint32_t &b, uint32_t &region)
{
    if (idx < G.head) {
        a = idx * TS_TT;
        b = min(a + TS_TT, G.sb);
        region = 0;
    } else {
        const uint32_t q = idx - G.head, r = q / G.tpr, i = q - r * G.tpr;
        a = G.sb + r * G.TILE + i * TS_TT;
        b = G.sb + r * G.TILE + min((i + 1u) * TS_TT, G.TILE);
        region = r;
    }
}

__global__ void k_tok_bounds_grid(const uint32_t *__restrict__ chain, uint32_t ntok, ts_grid G, uint32_t tile0, uint32_t ntiles,
                                  uint32_t *__restrict__ tstart)
{
    /* tstart[t] = first token that lies in tile tile0 + t or later */
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > ntok) return;
    const uint32_t first = k == 0 ? 0u : ts_tile_of(G, chain[k - 1]) - tile0 + 1;
    const uint32_t last = k == ntok ? ntiles : ts_tile_of(G, chain[k]) - tile0;
    for (uint32_t t = first; t <= last; t++) tstart[t] = k;
}

/* workgroup barrier that orders LDS traffic only: global loads in flight stay in flight (__syncthreads() drains
 * them: its fence waits for vmcnt(0)) */
__device__ __forceinline__ void ts_barrier_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/* exclusive prefix of one value per thread over the workgroup; *total (LDS) = the sum.  Two barriers. */
__device__ __forceinline__ uint32_t ts_wg_scan(uint32_t v, uint32_t *wsum, uint32_t *total)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += t;
    }
    ts_barrier_lds();                                     /* the previous use of wsum is over */
    if (lane == 63) wsum[wave] = incl;
    ts_barrier_lds();
    uint32_t run = incl - v, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < TS_BLOCK / 64; w++) {
        const uint32_t t = wsum[w];
        run += w < wave ? t : 0u;
        all += t;
    }
    if (threadIdx.x == 0) *total = all;
    return run;
}

/* The hand-over lists of the tile's window cells are built HERE, in LDS, from ps/xval (round 3; until then two kernels
 * -- k_xfer_blocksum/blockfill, 1.2 ms and 4.6 GB of HBM traffic per 100 MB -- grouped every hand-over of the input by
 * destination in HBM and the tile read its slice back).  A hand-over x -> x + S[x] that a token p of the tile [a, b) can
 * see into a cell of its window [a - sb, b) has x + sb < p and x > cell - sb: x in [a - 2 sb, b - sb), at most TS_TT + sb
 * evictions -- as many as the window has cells.  Count per cell (two 16-bit counters per LDS word), prefix sums in
 * place, place through the same counters: list(co) = lent[ lofs[co] .. lofs[co + 1] ). */
#define TS_SRC ((TS_TT + 4096 + TS_BLOCK - 1) / TS_BLOCK)            /* evictions per thread: sb <= 4096 on this path */
#define TS_SLOTS 32u                                                  /* words the tiles' hand-over counts are spread over */

/* total[0] += total[1 .. TS_SLOTS], which are cleared */
__global__ void k_ts_total(unsigned long long *total)
{
    unsigned long long v = threadIdx.x < TS_SLOTS ? total[1u + threadIdx.x] : 0ull;
    if (threadIdx.x < TS_SLOTS) total[1u + threadIdx.x] = 0ull;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    if (threadIdx.x == 0) total[0] += v;
}

/* Exclusive prefixes of TWO values per thread over the workgroup (one set of barriers); *ta, *tb (LDS) = the sums. */
__device__ __forceinline__ void ts_wg_scan2(uint32_t a, uint32_t b, uint32_t *wsa, uint32_t *wsb, uint32_t *ta, uint32_t *tb, uint32_t &ea, uint32_t &eb)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t ia = a, ib = b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(ia, d, 64), u = __shfl_up(ib, d, 64);
        if (lane >= (uint32_t)d) { ia += t; ib += u; }
    }
    ts_barrier_lds();                                     /* the previous use of wsa / wsb is over */
    if (lane == 63) { wsa[wave] = ia; wsb[wave] = ib; }
    ts_barrier_lds();
    uint32_t ra = ia - a, rb = ib - b, alla = 0, allb = 0;
#pragma unroll
    for (uint32_t w = 0; w < TS_BLOCK / 64; w++) {
        const uint32_t t = wsa[w], u = wsb[w];
        ra += w < wave ? t : 0u;
        rb += w < wave ? u : 0u;
        alla += t;
        allb += u;
    }
    if (threadIdx.x == 0) { *ta = alla; *tb = allb; }
    ea = ra;
    eb = rb;
}

/* Round 5: the hand-overs are grouped by the SLOT of their cell in the tile's key order, not by the cell.
 *
 * prio_p(c) = min(what c came with, the hand-overs into c by evictions before p), and the token wants the argmin over the
 * members c of its run inside its window.  A minimum of minima is a minimum over the union: over the members' OWN priorities
 * (c + voff: monotone in c, so the winner among them is simply the oldest member in the window -- a compare and a minimum per
 * member, no list look-up) and over every hand-over (x -> c, v) whose cell is a member -- and with the entries sorted by
 * slot those are ONE contiguous range per token, [first entry with slot >= lo, first entry with slot >= hi), dealt to the
 * threads like the members: per entry a time test, a window test, a 64-bit minimum.  (A stale hand-over never wins: the one
 * that replaced it in the same cell is lower -- tree.c:202-231: the successor only moves up.)  Round 4 walked, per member,
 * the list of its cell: ~170 wave instructions per 64 members with the wavefront's union of list lengths; now ~10 per 64
 * members plus ~15 per 64 entries, and there are four entries to ten members.
 *
 * What it takes: the slot of EVERY window cell (`inv`, not only of the tile's positions: a hand-over's cell -> its slot),
 * 16-bit counters per slot while the lists are built -- they share their 16 KB with the window bytes and the batch arrays,
 * which are staged once the entries are placed --, a coarse index (the first entry of every 16th slot) to find a run's
 * entries, and 8 bytes an entry (slot | eviction << 16, priority).  A long run (TS_BIG cells) needs no bitmap any
 * more: its oldest member is found by walking the window's positions upwards, `inv[c]` in [lo, hi), and its hand-overs are
 * entries like everybody's. */
#ifndef TS_SMALL
#define TS_SMALL 8u                                                   /* a token with at most this many members and entries is settled by its own lane pair (measured: 2 2.54 ms, 4 2.48, 8 2.45; as loops with a run-time bound 8 2.53, 16 2.56, 32 2.71) */
#endif
#define TS_CG 16u                                                     /* slots per coarse list offset */

__global__ __launch_bounds__(TS_BLOCK, 8) __attribute__((amdgpu_num_sgpr(80))) void k_tokens_sorted(const uint8_t *__restrict__ in, uint32_t n, int sb, int la, int ob, int lb,
                                                            const uint32_t *__restrict__ chain, const uint32_t *__restrict__ tstart,
                                                            const uint8_t *__restrict__ maxlen, const uint32_t *__restrict__ ps,
                                                            const uint32_t *__restrict__ xval, uint32_t pos0, uint32_t pos1,
                                                            uint32_t *__restrict__ tokval, uint32_t ent_cap, uint32_t off_tk, uint32_t off_sorted,
                                                            uint32_t off_inv, uint32_t off_cofs, uint32_t off_ent,
                                                            const uint32_t *__restrict__ look, uint32_t nlook, uint32_t voff,
                                                            const uint16_t *__restrict__ order_all, uint32_t RP, ts_grid G, uint32_t tile0, uint32_t ntiles,
                                                            unsigned long long *__restrict__ total /* [1 + a slot] += hand-overs of the evictions [a - sb, b - sb) (a statistic) */
                                                            , uint32_t big_min /* a run of this many cells or more finds its oldest member by walking positions (B') */
#ifdef LZ77X_VARIANTS
                                                            , uint32_t probe /* timing probe (LZ77X_TS_PROBE, wrong output): leave after 1 the set-up, 2 phase A + scans, 3 B1, 4 B2 */
#endif
                                                            )
{
#ifndef LZ77X_VARIANTS
    constexpr uint32_t probe = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    /* [0, off_sorted): while the lists are built, 8 * TS_BLOCK 16-bit counters; afterwards the window bytes and the batch arrays */
    uint16_t *lofs = reinterpret_cast<uint16_t *>(smem);
    uint32_t *lofs32 = reinterpret_cast<uint32_t *>(smem);
    uint8_t *by = smem;
    unsigned long long *tk_best = reinterpret_cast<unsigned long long *>(smem + off_tk);   /* TS_TB */
    uint16_t *tk_nm = reinterpret_cast<uint16_t *>(tk_best);                               /* (before tk_best is set: members / entries per token) */
    uint16_t *tk_ne = tk_nm + TS_TB;
    uint32_t *tk_cum = reinterpret_cast<uint32_t *>(tk_best + TS_TB);                      /* TS_TB + 1: members */
    uint32_t *tk_cume = tk_cum + TS_TB + 4;                                                 /* TS_TB + 1: entries */
    uint32_t *tk_pl = tk_cume + TS_TB + 4;                                                  /* TS_TB: offset | len << 16 */
    uint16_t *tk_lo = reinterpret_cast<uint16_t *>(tk_pl + TS_TB);                         /* TS_TB: first slot of the run */
    uint16_t *tk_elo = tk_lo + TS_TB;                                                       /* TS_TB: first entry of the run */
    uint16_t *sorted = reinterpret_cast<uint16_t *>(smem + off_sorted);        /* window cells (offsets from wbase) in key order */
    uint16_t *inv = reinterpret_cast<uint16_t *>(smem + off_inv);              /* slot of every window cell */
    uint16_t *cofs = reinterpret_cast<uint16_t *>(smem + off_cofs);            /* first entry of slot TS_CG * i */
    uint32_t *ent_sx = reinterpret_cast<uint32_t *>(smem + off_ent);           /* slot | (eviction - xs0) << 16, sorted by slot */
    uint32_t *ent_v = ent_sx + ent_cap;                                        /* staged: the priority handed over */
    __shared__ uint32_t wsum[TS_BLOCK / 64], wsum2[TS_BLOCK / 64], s_own[TS_BLOCK / 64];
    __shared__ uint32_t s_total, s_total2, s_nbig;
    __shared__ uint16_t fb_lo[256], fb_hi[256];
    __shared__ uint16_t big[TS_TB];                                       /* the batch's tokens with a run of TS_BIG cells or more (bit 31 of their tk_pl) */

    const uint32_t tid = threadIdx.x;
    const uint32_t usb = (uint32_t)sb;
    /* consecutive workgroups go to different XCDs (eight L2s): give each XCD a contiguous stretch of tiles, so that
     * the tiles that share a region's order array and overlap in their windows meet in the same L2 */
    const uint32_t per_xcd = gridDim.x >> 3;
    const uint32_t tl = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (tl >= ntiles) return;
    uint32_t a, b, region;
    ts_tile_range(G, tile0 + tl, a, b, region);
    a = max(a, pos0);
    b = min(b, pos1);
    if (a >= b) return;
    const uint32_t t0r = region * G.TILE;
    const uint32_t wlo = a > usb ? a - usb : 0u;
    const uint32_t wbase = wlo & ~3u;
    /* ---- everything the tile needs from HBM is requested at once (one latency, not seven in a row): the region's
     *      order, the token range, the evictions, the window bytes; the barriers in between only order LDS traffic
     *      (ts_barrier_lds does not wait for loads in flight) ---- */
    /* (the slots stay PACKED, two a word: sixteen unpacked registers on top of the fourteen eviction words below were more
     * than the 64 the kernel may hold -- the compiler spilled twelve of the eviction loads one by one, each behind
     * s_waitcnt vmcnt(0): twelve round trips in a row where one was meant, and 3.5 of the kernel's 4.8 GB of HBM traffic) */
    uint32_t mw[8];
    {
        const uint32_t K = RP / TS_BLOCK;                  /* 4, 8 or 16 consecutive slots per thread */
        const uint16_t *ord = order_all + (size_t)region * RP + (size_t)tid * K;
#pragma unroll
        for (int q = 0; q < 8; q++) mw[q] = 0xFFFFFFFFu;    /* (0xFFFF is no slot of a region: RP <= 16384) */
        if (K == 16) {
            const uint4 v0 = *reinterpret_cast<const uint4 *>(ord), v1 = *reinterpret_cast<const uint4 *>(ord + 8);
            mw[0] = v0.x; mw[1] = v0.y; mw[2] = v0.z; mw[3] = v0.w; mw[4] = v1.x; mw[5] = v1.y; mw[6] = v1.z; mw[7] = v1.w;
        } else if (K == 8) {
            const uint4 v0 = *reinterpret_cast<const uint4 *>(ord);
            mw[0] = v0.x; mw[1] = v0.y; mw[2] = v0.z; mw[3] = v0.w;
        } else {
            const uint2 v0 = *reinterpret_cast<const uint2 *>(ord);
            mw[0] = v0.x; mw[1] = v0.y;
        }
    }
    auto mine_at = [&](int q) -> uint32_t { return (q & 1) ? mw[q >> 1] >> 16 : mw[q >> 1] & 0xFFFFu; };
    const uint32_t k0 = tstart[tl], k1 = tstart[tl + 1];
    /* the evictions [xs0, xs1) (every load unconditional, clamped: the compiler keeps them in flight) */
    const uint32_t xs0 = wlo > usb ? wlo - usb : 0u, xs1 = b > usb ? b - usb : 0u;
    uint32_t xv[TS_SRC], xc[TS_SRC];                        /* priority handed over / the cell it goes to, then 1 + its slot */
    {
        const uint32_t xl = xs1 ? xs1 - 1u : 0u;
#pragma unroll
        for (int r = 0; r < TS_SRC; r++) {
            const uint32_t x = xs0 + tid + (uint32_t)r * TS_BLOCK;
            const uint32_t xq = min(x, xl);
            xv[r] = xval[xq];
            xc[r] = ps[xq];
        }
    }
    constexpr int BY_PER = (TS_TT + 4096 + 256 + 16 + 4 * TS_BLOCK - 1) / (4 * TS_BLOCK);
    const uint32_t nb = (b + (uint32_t)la + 8 - wbase + 3) & ~3u;
    *reinterpret_cast<uint4 *>(lofs32 + 4u * tid) = make_uint4(0u, 0u, 0u, 0u);     /* the counters (ordered before the counting by the scan's barriers) */
    /* the region's order, filtered down to the cells of [wlo, b) */
    {
        const uint32_t wl = wlo - t0r, wn = b - wlo;        /* region-local window */
        uint32_t cnt = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) cnt += (mine_at(q) - wl < wn) ? 1u : 0u;
        uint32_t run = ts_wg_scan(cnt, wsum, &s_total);
        const uint32_t shift = t0r - wbase;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const uint32_t m = mine_at(q);
            if (m - wl < wn) {
                const uint32_t co = m + shift;
                sorted[run] = (uint16_t)co;
                inv[co] = (uint16_t)run;
                run++;
            }
        }
    }
    /* the slots' registers are free now: the window bytes and the first batch's token positions (and, once they are here,
     * their lengths) are requested here and travel while the entries are counted and placed -- under the batch loop the
     * positions and lengths were two exposed round trips, one behind the other */
    uint32_t by_raw[BY_PER];
#pragma unroll
    for (int r = 0; r < BY_PER; r++) {
        const uint32_t i = (tid + (uint32_t)r * TS_BLOCK) * 4u;
        by_raw[r] = i < nb ? *reinterpret_cast<const uint32_t *>(in + wbase + i) : 0u;
    }
    const uint32_t p_pre = chain[min(k0 + (tid >> 1), max(k1, 1u) - 1u)];
    ts_barrier_lds();                                       /* every cell has its slot */
    const uint32_t N = s_total;                             /* = b - wlo: every cell of the window is in the region */
    /* hand-overs per slot: counter of slot i = entry i + 1 (entry 0 stays 0), two entries per word */
    {
        uint32_t own = 0;
#pragma unroll
        for (int r = 0; r < TS_SRC; r++) {
            const uint32_t x = xs0 + tid + (uint32_t)r * TS_BLOCK;
            const uint32_t dst = x + (xc[r] >> 16);
            const bool ok = x < xs1 && xv[r] != LZ77X_NONE32 && dst >= wlo;
            own += (ok && x + usb >= a) ? 1u : 0u;
            uint32_t e = 0;
            if (ok) {
                e = (uint32_t)inv[dst - wbase] + 1u;
                atomicAdd(&lofs32[e >> 1], 1u << (16u * (e & 1u)));
            }
            xc[r] = e;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) own += __shfl_xor(own, d, 64);
        if ((tid & 63u) == 0) s_own[tid >> 6] = own;
    }
    const uint32_t len_pre = maxlen[p_pre];
    ts_barrier_lds();
    if (total && tid == 0) {
        /* one atomic per workgroup, spread over TS_SLOTS words: atomics to ONE address queue in the L2 at ~100 cycles each
         * (one per wavefront -- 781 K of them per 100 MB -- took 7 ms) */
        uint32_t own = 0;
        for (uint32_t w = 0; w < TS_BLOCK / 64; w++) own += s_own[w];
        if (own) atomicAdd(&total[1u + (blockIdx.x & (TS_SLOTS - 1u))], (unsigned long long)own);
    }
    {
        /* prefix sums in place: eight entries a thread */
        const uint4 w = *reinterpret_cast<const uint4 *>(lofs32 + 4u * tid);
        uint32_t c[8] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16, w.z & 0xFFFFu, w.z >> 16, w.w & 0xFFFFu, w.w >> 16};
        uint32_t sum = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) sum += c[q];
        uint32_t run = ts_wg_scan(sum, wsum, &s_total);     /* (its first barrier: every thread has read N) */
        /* entry e becomes the START of the list of slot e - 1 ... and, once the placing below has bumped it by the
         * list's length, the start of the list of slot e: list(i) = [lofs[i], lofs[i + 1]) */
#pragma unroll
        for (int q = 0; q < 8; q++) { const uint32_t t = c[q]; c[q] = run; run += t; }
        *reinterpret_cast<uint4 *>(lofs32 + 4u * tid) = make_uint4(c[0] | c[1] << 16, c[2] | c[3] << 16, c[4] | c[5] << 16, c[6] | c[7] << 16);
    }
    ts_barrier_lds();
    /* more hand-overs than the LDS left over holds (never seen on text: four in ten evictions hand over): the entries
     * keep slot and eviction only and a look-up fetches the priority from xval[] */
    const uint32_t etot = s_total;
    const bool staged = etot <= ent_cap;
#pragma unroll
    for (int r = 0; r < TS_SRC; r++) {
        if (xc[r]) {
            const uint32_t x = xs0 + tid + (uint32_t)r * TS_BLOCK;
            const uint32_t sh = 16u * (xc[r] & 1u);
            const uint32_t at = (atomicAdd(&lofs32[xc[r] >> 1], 1u << sh) >> sh) & 0xFFFFu;
            ent_sx[at] = (xc[r] - 1u) | ((x - xs0) << 16);
            if (staged) ent_v[at] = xv[r];
        }
    }
    ts_barrier_lds();
    uint32_t co_first = 0;
    if (tid < 8u * TS_BLOCK / TS_CG) co_first = lofs[TS_CG * tid];
    ts_barrier_lds();                                       /* the counters have served: their place goes to the bytes and the batch arrays */
    if (tid < 8u * TS_BLOCK / TS_CG) cofs[tid] = (uint16_t)co_first;
#pragma unroll
    for (int r = 0; r < BY_PER; r++) {
        const uint32_t i = (tid + (uint32_t)r * TS_BLOCK) * 4u;
        if (i < nb) *reinterpret_cast<uint32_t *>(by + i) = by_raw[r];
    }
    if (tid == 0) s_nbig = 0;
    ts_barrier_lds();
    /* cells by first byte: [fb_lo[c], fb_hi[c]) -- the run of a token of length 1, without a search */
    for (uint32_t i = tid; i < N; i += TS_BLOCK) {
        const uint32_t c = by[sorted[i]];
        const uint32_t cp = i ? (uint32_t)by[sorted[i - 1]] : 256u;
        if (c != cp) {
            fb_lo[c] = (uint16_t)i;
            if (i) fb_hi[cp] = (uint16_t)i;
        }
        if (i + 1 == N) fb_hi[c] = (uint16_t)N;
    }
    __syncthreads();
    /* a cell of a later segment's look-back carries a rank, not its position: the tiles that see one (the first two of a
     * segment) take every member through look[] and no short cut */
    const bool has_look = wlo < nlook;
    if (probe == 1) return;

    const uint32_t omask = ob >= 32 ? 0xFFFFFFFFu : (1u << ob) - 1u;
    for (uint32_t kb = k0; kb < k1; kb += TS_TB) {
        const uint32_t nt = min(TS_TB, k1 - kb);
        /* ---- A: the run of cells sharing the token's len bytes, and the run of entries that go with it; lane pair = (down, up) ---- */
        {
            const uint32_t ti = tid >> 1, up = tid & 1u;
            if (ti < nt) {
                const uint32_t p = kb == k0 ? p_pre : chain[kb + ti];
                const uint32_t len = kb == k0 ? len_pre : (uint32_t)maxlen[p];
                const uint32_t qo = p - wbase;
                int edge = 0;
                uint32_t eedge = 0;
                if (len > 0) {
                    uint32_t qw[4];
#pragma unroll
                    for (int w = 0; w < 4; w++) qw[w] = (uint32_t)(4 * w) < len ? ld32_at<true>(by, qo + 4 * w) : 0u;
                    auto shares = [&](int i) -> bool {
                        const uint32_t co = sorted[i];
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            if ((uint32_t)(4 * w) < len) {
                                uint32_t x = ld32_at<true>(by, co + 4 * w) ^ qw[w];
                                const uint32_t rem = len - 4 * w;
                                if (rem < 4) x &= (1u << (8 * rem)) - 1u;
                                if (x) return false;
                            }
                        }
                        for (uint32_t i2 = 16; i2 < len; i2 += 4) {
                            uint32_t x = ld32_at<true>(by, co + i2) ^ ld32_at<true>(by, qo + i2);
                            const uint32_t rem = len - i2;
                            if (rem < 4) x &= (1u << (8 * rem)) - 1u;
                            if (x) return false;
                        }
                        return true;
                    };
                    const int j = (int)inv[qo];
                    if (len == 1) {
                        edge = up ? (int)fb_hi[qw[0] & 0xFFu] : (int)fb_lo[qw[0] & 0xFFu];
                    } else if (!up) {
                        int lo = j, bad = -1, step = 1;           /* every slot of [lo, j) shares; slot `bad` does not */
                        while (lo > 0) {
                            const int t = lo > step ? lo - step : 0;
                            if (shares(t)) { lo = t; step <<= 1; }
                            else { bad = t; break; }
                        }
                        if (bad >= 0)
                            while (lo - bad > 1) {
                                const int m = (lo + bad) >> 1;
                                if (shares(m)) lo = m; else bad = m;
                            }
                        edge = lo;
                    } else {
                        int hi = j + 1, bad = (int)N, step = 1;   /* every slot of (j, hi) shares; slot `bad` does not */
                        while (hi < (int)N) {
                            const int t = min(hi - 1 + step, (int)N - 1);
                            if (shares(t)) { hi = t + 1; step <<= 1; }
                            else { bad = t; break; }
                        }
                        while (hi < bad) {
                            const int m = (hi + bad) >> 1;
                            if (shares(m)) hi = m + 1; else bad = m;
                        }
                        edge = hi;
                    }
                    /* the first entry whose slot is not below the edge: from the coarse index, a few entries forward */
                    uint32_t e = cofs[(uint32_t)edge / TS_CG];
                    while (e < etot && (ent_sx[e] & 0xFFFFu) < (uint32_t)edge) e++;
                    eedge = e;
                }
                const uint32_t o_edge = (uint32_t)__shfl_xor(edge, 1, 64), o_eedge = __shfl_xor(eedge, 1, 64);
                if (!up) {
                    uint32_t nm = o_edge - (uint32_t)edge, ne = o_eedge - eedge;
                    tk_lo[ti] = (uint16_t)edge;
                    tk_elo[ti] = (uint16_t)eedge;
                    tk_pl[ti] = qo | (len << 16);
                    if (len > 0 && nm <= TS_SMALL && ne <= TS_SMALL) {
                        /* a token with a handful of candidates (most tokens of three bytes and more) is settled right here
                         * by the lane that found its run: dealt with the others it cost every thread whose piece it touched a
                         * per-token prologue -- half of what B1 and B2 executed was such prologues */
                        const uint32_t cmin = p > usb ? p - usb : 0u, wn = p - cmin;
                        unsigned long long best = ~0ull;
#pragma unroll
                        for (uint32_t k = 0; k < TS_SMALL; k++) {
                            if (k < nm) {
                                const uint32_t c = wbase + sorted[(uint32_t)edge + k];
                                if (c - cmin < wn) {
                                    const uint32_t prio = c < nlook ? look[c] : c + voff;
                                    const unsigned long long key = ((unsigned long long)prio << 32) | c;
                                    best = key < best ? key : best;
                                }
                            }
                        }
#pragma unroll
                        for (uint32_t k = 0; k < TS_SMALL; k++) {
                            if (k < ne) {
                                const uint32_t sx = ent_sx[eedge + k];
                                const uint32_t c = wbase + sorted[sx & 0xFFFFu], x = xs0 + (sx >> 16);
                                if (x < cmin && c - cmin < wn) {
                                    const uint32_t v = staged ? ent_v[eedge + k] : xval[x];
                                    const unsigned long long key = ((unsigned long long)v << 32) | c;
                                    best = key < best ? key : best;
                                }
                            }
                        }
                        /* (its winner's cell goes where its counts would have gone: bit 15 set, no members, no entries) */
                        nm = 0x8000u | ((uint32_t)best - wbase);
                        ne = 0;
                    }
                    tk_nm[ti] = (uint16_t)nm;
                    tk_ne[ti] = (uint16_t)ne;
                }
            }
        }
        __syncthreads();
        /* ---- the runs laid end to end, members and entries ---- */
        {
            uint32_t nm = 0, ne = 0;
            unsigned long long settled = ~0ull;
            if (tid < nt) { nm = tk_nm[tid]; ne = tk_ne[tid]; }
            if (nm & 0x8000u) { settled = (unsigned long long)(wbase + (nm & 0x7FFFu)); nm = 0; }     /* settled in phase A */
            /* a big run: its oldest member in the window is found by walking the window's positions (below) */
            /* (its members stay in the count -- hi = lo + cum[ti + 1] - cum[ti] -- and B1 passes over them) */
            if (nm >= big_min && !has_look) { big[atomicAdd(&s_nbig, 1u)] = (uint16_t)tid; tk_pl[tid] |= 0x80000000u; }
            uint32_t em, ee;
            ts_wg_scan2(nm, ne, wsum, wsum2, &s_total, &s_total2, em, ee);
            if (tid < nt) { tk_cum[tid] = em; tk_cume[tid] = ee; tk_best[tid] = settled; }
            __syncthreads();
        }
        const uint32_t Wm = s_total, We = s_total2;
        if (tid == 0) { tk_cum[nt] = Wm; tk_cume[nt] = We; }
        __syncthreads();
        if (probe == 2) continue;
        /* ---- B1: every run member inside the token's window at the priority it came with ---- */
        {
            const uint32_t per = (Wm + TS_BLOCK - 1u) / TS_BLOCK;
            const uint32_t w0 = tid * per, w1 = min(w0 + per, Wm);
            if (w0 < w1) {
                uint32_t lo_i = 0, hi_i = nt;               /* largest ti with cum[ti] <= w0 */
                while (hi_i - lo_i > 1) {
                    const uint32_t m = (lo_i + hi_i) >> 1;
                    if (tk_cum[m] <= w0) lo_i = m; else hi_i = m;
                }
                uint32_t ti = lo_i;
                uint32_t w = w0;
                while (w < w1) {
                    /* skip tokens without members (cum[ti + 1] == cum[ti]) */
                    while (tk_cum[ti + 1] <= w) ti++;
                    const uint32_t pl = tk_pl[ti], cbase = tk_cum[ti], cend = min(tk_cum[ti + 1], w1);
                    const uint32_t p = wbase + (pl & 0xFFFFu);
                    const uint32_t cmin = p > usb ? p - usb : 0u, wn = p - cmin;
                    const uint32_t slot0 = (uint32_t)tk_lo[ti] - cbase;
                    if (pl & 0x80000000u) { w = cend; continue; }        /* a big run: B' */
                    if (!has_look) {
                        /* own priorities are positions (+ voff): the lowest one is the oldest member (tree.c:102-105: a new node is a leaf) */
                        uint32_t bc = 0xFFFFFFFFu;
                        for (; w < cend; w++) {
                            const uint32_t c = wbase + sorted[slot0 + w];
                            bc = c - cmin < wn ? min(bc, c) : bc;
                        }
                        if (bc != 0xFFFFFFFFu) atomicMin(&tk_best[ti], ((unsigned long long)(bc + voff) << 32) | bc);
                    } else {
                        unsigned long long best = ~0ull;
                        for (; w < cend; w++) {
                            const uint32_t c = wbase + sorted[slot0 + w];
                            if (c - cmin >= wn) continue;
                            const uint32_t prio = c < nlook ? look[c] : c + voff;
                            const unsigned long long key = ((unsigned long long)prio << 32) | c;
                            best = key < best ? key : best;
                        }
                        if (best != ~0ull) atomicMin(&tk_best[ti], best);
                    }
                }
            }
        }
        if (probe == 3) { __syncthreads(); continue; }
        /* ---- B2: every hand-over into a member by an eviction before p (x + sb < p) ---- */
        {
            const uint32_t per = (We + TS_BLOCK - 1u) / TS_BLOCK;
            const uint32_t w0 = tid * per, w1 = min(w0 + per, We);
            if (w0 < w1) {
                uint32_t lo_i = 0, hi_i = nt;
                while (hi_i - lo_i > 1) {
                    const uint32_t m = (lo_i + hi_i) >> 1;
                    if (tk_cume[m] <= w0) lo_i = m; else hi_i = m;
                }
                uint32_t ti = lo_i;
                uint32_t w = w0;
                while (w < w1) {
                    while (tk_cume[ti + 1] <= w) ti++;
                    const uint32_t pl = tk_pl[ti], cbase = tk_cume[ti], cend = min(tk_cume[ti + 1], w1);
                    const uint32_t p = wbase + (pl & 0xFFFFu);
                    const uint32_t cmin = p > usb ? p - usb : 0u, wn = p - cmin;
                    const uint32_t xlim = cmin;              /* an eviction x is before p: x + sb < p */
                    const uint32_t e0 = (uint32_t)tk_elo[ti] - cbase;
                    unsigned long long best = ~0ull;
                    if (staged) {
                        for (; w < cend; w++) {
                            const uint32_t sx = ent_sx[e0 + w], v = ent_v[e0 + w];
                            const uint32_t c = wbase + sorted[sx & 0xFFFFu], x = xs0 + (sx >> 16);
                            const unsigned long long key = ((unsigned long long)v << 32) | c;
                            best = (x < xlim && c - cmin < wn && key < best) ? key : best;
                        }
                    } else {
                        for (; w < cend; w++) {
                            const uint32_t sx = ent_sx[e0 + w];
                            const uint32_t c = wbase + sorted[sx & 0xFFFFu], x = xs0 + (sx >> 16);
                            if (x < xlim && c - cmin < wn) {
                                const unsigned long long key = ((unsigned long long)xval[x] << 32) | c;
                                best = key < best ? key : best;
                            }
                        }
                    }
                    if (best != ~0ull) atomicMin(&tk_best[ti], best);
                }
            }
        }
        if (probe == 4) { __syncthreads(); continue; }
        /* ---- B': the big runs, sixteen lanes each: the oldest member of the window is the first position from p - sb
         *      upwards whose slot lies in the run (a run of r cells is met after ~ sb / r positions: one or two steps) ---- */
        {
            const uint32_t nbig = s_nbig, lane = tid & 63u, wave = tid >> 6, grp = lane >> 4, gl = lane & 15u;
            for (uint32_t bi0 = wave * 4u; bi0 < nbig; bi0 += (TS_BLOCK / 64u) * 4u) {
                const uint32_t bi = bi0 + grp;
                bool done = bi >= nbig;
                const uint32_t ti = done ? 0u : (uint32_t)big[bi];
                const uint32_t p = wbase + (tk_pl[ti] & 0xFFFFu), cmin = p > usb ? p - usb : 0u;
                const uint32_t lo = tk_lo[ti], nrun = tk_cum[ti + 1] - tk_cum[ti];
                uint32_t c0 = cmin;
                for (;;) {
                    const uint32_t c = c0 + gl;
                    const bool in = !done && c < p;
                    const bool ok = in && (uint32_t)inv[in ? c - wbase : 0u] - lo < nrun;
                    const uint32_t h16 = (uint32_t)(__ballot(ok) >> (16u * grp)) & 0xFFFFu;
                    if (!done && h16) {
                        const uint32_t c1 = c0 + (uint32_t)__builtin_ctz(h16);
                        if (gl == 0) atomicMin(&tk_best[ti], ((unsigned long long)(c1 + voff) << 32) | c1);
                        done = true;
                    }
                    c0 += 16u;
                    done = done || c0 >= p;
                    if (!__ballot(!done)) break;
                }
            }
        }
        __syncthreads();
        if (tid == 0) s_nbig = 0;                          /* (the next batch counts its own; ordered by the barriers of phase A) */
        /* ---- C: the token ---- */
        if (tid < nt) {
            const uint32_t pl = tk_pl[tid], qo = pl & 0xFFFFu, len = (pl >> 16) & 0x7FFFu;
            const uint32_t next = by[qo + len];
            const uint32_t off = len ? wbase + qo - (uint32_t)(tk_best[tid] & 0xFFFFFFFFull) : 0u;
            tokval[kb + tid] = (off & omask) | (len << ob) | (next << (ob + lb));
        }
        __syncthreads();
    }
}

#ifdef LZ77X_VARIANTS   /* (round 4's tie-break for LDS-sized windows (hand-over lists per cell, a bitmap for long runs)) */
#include "variants/tokens_sorted_v4.inc"
#endif

#ifdef LZ77X_VARIANTS   /* (round 1's large-window tie-break (a two-byte index over the window)) */
#include "variants/tokens_big.inc"
#endif

/* ---- large windows, production: candidates in RANK order -----------------------------------------
 * The candidates whose match with p has the full length `len` are, with every other position of the
 * region that shares those len bytes, one contiguous run around p in the region's sorted order
 *  -- which the match stage has already computed.  A wave walks outward from rank[p], 32
 * lanes down and 32 up: a lane fetches the position at its rank offset, checks the len bytes, and the
 * run ends at the first failure in each direction; run members inside the window [p-sb, p) are the
 * candidates.  No candidate index to build, and a token costs (run length)/32 rounds instead of a scan
 * of every window position that starts with its two bytes (tens of thousands for "\0\0" at sb 65535). */
/* ---- large windows: the tokens of LENGTH ONE without visiting their candidates one by one (round 3) --------------------
 * The candidates of a length-1 token are the window cells with its first byte: 256 of the 65535 on random bytes, where
 * 37 % of the tokens have length 1, and the rank walk below pays a dependent `ofs` -> `ent` look-up for each: 4.3 G
 * requests to the L2 per launch, all but a tenth of what the kernel does.  With the positions cut into blocks of sb, a
 * token p of block b sees cells of b (every one still alive) and the cells >= p - sb of b - 1, and its answer is the
 * minimum of four scans over CONTIGUOUS arrays of the (block, first byte) buckets:
 *   cells of (b, v) below p at their own priority; hand-overs into cells of (b, v) by evictions before p;
 *   cells of (b-1, v) from p - sb on at `base` = their priority at the block boundary T = b * sb;
 *   hand-overs into cells of (b-1, v) by evictions in [T, p) whose cell is still in the window.
 * A cell only ever decreases, so the minimum over "own priority or any hand-over seen so far" is the minimum over the
 * cells' CURRENT priorities: the lowest value cannot have been replaced by a lower one in its own cell.  The buckets
 * are built once per token launch from the hand-over lists (k_sx_count, a scan, k_sx_scatter). */
#define SX_CHUNK 8192u                               /* cells per workgroup of the index kernels: at most two blocks (sb > 8192 here) */

struct sx_index {
    const uint32_t *off_c, *off_h;                   /* bucket -> first cell / hand-over record (exclusive prefix sums, one past the end) */
    const uint2 *cells;                              /* (position, base) */
    const uint2 *hx;                                 /* (eviction, priority handed over) */
    const uint32_t *hd;                              /* its destination cell */
    uint32_t bl0, nbl;                               /* first block of the index, number of blocks */
};

__device__ __forceinline__ uint32_t sx_own(uint32_t c, const uint32_t *__restrict__ look, uint32_t nlook, uint32_t voff)
{
    return c < nlook ? look[c] : c + voff;
}

/* phase 0 / 1 of both index kernels: the cells [c0, c1) of this workgroup by bucket, in LDS */
__global__ __launch_bounds__(256) void k_sx_count(const uint8_t *__restrict__ in, uint32_t c0, uint32_t c1, uint32_t sb, uint32_t bl0,
                                                  const uint32_t *__restrict__ ofs, uint32_t dbase, uint32_t *__restrict__ cnt_c,
                                                  uint32_t *__restrict__ cnt_h)
{
    __shared__ uint32_t h_c[2][256], h_h[2][256];
    const uint32_t tid = threadIdx.x;
    const uint32_t a = c0 + blockIdx.x * SX_CHUNK, b = min(a + SX_CHUNK, c1);
    const uint32_t blk_a = a / sb;
    for (uint32_t i = tid; i < 512; i += 256) { (&h_c[0][0])[i] = 0; (&h_h[0][0])[i] = 0; }
    __syncthreads();
    for (uint32_t c = a + tid; c < b; c += 256) {
        const uint32_t j = c / sb - blk_a, v = in[c];
        const uint32_t lo = c > dbase ? ofs[c - dbase - 1] : 0u, hi = ofs[c - dbase];
        atomicAdd(&h_c[j][v], 1u);
        if (hi > lo) atomicAdd(&h_h[j][v], hi - lo);
    }
    __syncthreads();
    for (uint32_t i = tid; i < 512; i += 256) {
        const uint32_t j = i >> 8, v = i & 255u;
        const uint32_t bucket = (blk_a + j - bl0) * 256u + v;
        if (h_c[j][v]) atomicAdd(&cnt_c[bucket], h_c[j][v]);
        if (h_h[j][v]) atomicAdd(&cnt_h[bucket], h_h[j][v]);
    }
}

__global__ __launch_bounds__(256) void k_sx_scatter(const uint8_t *__restrict__ in, uint32_t c0, uint32_t c1, uint32_t sb, uint32_t bl0,
                                                    const uint32_t *__restrict__ ofs, const uint2 *__restrict__ ent, uint32_t dbase,
                                                    const uint32_t *__restrict__ off_c, const uint32_t *__restrict__ off_h,
                                                    uint32_t *__restrict__ cur_c, uint32_t *__restrict__ cur_h, uint2 *__restrict__ cells,
                                                    uint2 *__restrict__ hx, uint32_t *__restrict__ hd, const uint32_t *__restrict__ look,
                                                    uint32_t nlook, uint32_t voff)
{
    __shared__ uint32_t h_c[2][256], h_h[2][256];
    const uint32_t tid = threadIdx.x;
    const uint32_t a = c0 + blockIdx.x * SX_CHUNK, b = min(a + SX_CHUNK, c1);
    const uint32_t blk_a = a / sb;
    for (uint32_t i = tid; i < 512; i += 256) { (&h_c[0][0])[i] = 0; (&h_h[0][0])[i] = 0; }
    __syncthreads();
    for (uint32_t c = a + tid; c < b; c += 256) {
        const uint32_t j = c / sb - blk_a, v = in[c];
        const uint32_t lo = c > dbase ? ofs[c - dbase - 1] : 0u, hi = ofs[c - dbase];
        atomicAdd(&h_c[j][v], 1u);
        if (hi > lo) atomicAdd(&h_h[j][v], hi - lo);
    }
    __syncthreads();
    /* my share of every bucket: a range behind what the workgroups before me (in time, not in position: the records of a
     * bucket are in no particular order) have taken */
    for (uint32_t i = tid; i < 512; i += 256) {
        const uint32_t j = i >> 8, v = i & 255u;
        const uint32_t bucket = (blk_a + j - bl0) * 256u + v;
        const uint32_t nc = h_c[j][v], nh = h_h[j][v];
        h_c[j][v] = nc ? off_c[bucket] + atomicAdd(&cur_c[bucket], nc) : 0u;
        h_h[j][v] = nh ? off_h[bucket] + atomicAdd(&cur_h[bucket], nh) : 0u;
    }
    __syncthreads();
    for (uint32_t c = a + tid; c < b; c += 256) {
        const uint32_t blk = c / sb, j = blk - blk_a, v = in[c];
        const uint32_t lo = c > dbase ? ofs[c - dbase - 1] : 0u, hi = ofs[c - dbase];
        /* base: the cell's priority when the next block begins (every token of that block sees these hand-overs) */
        const uint64_t T = ((uint64_t)blk + 1u) * sb;
        uint32_t base = sx_own(c, look, nlook, voff), latest = 0;
        bool any = false;
        uint32_t slot_h = hi > lo ? atomicAdd(&h_h[j][v], hi - lo) : 0u;
        for (uint32_t e = lo; e < hi; e++) {
            const uint2 t = ent[e];
            if ((uint64_t)t.x + sb < T && (!any || t.x > latest)) { any = true; latest = t.x; base = t.y; }
            hx[slot_h] = t;
            hd[slot_h] = c;
            slot_h++;
        }
        cells[atomicAdd(&h_c[j][v], 1u)] = make_uint2(c, base);
    }
}

/* the token at p (its first byte v) of length one: offset of the candidate nearest the root (see above); all 64 lanes */
__device__ __forceinline__ uint32_t sx_query(const sx_index &X, uint32_t p, uint32_t v, uint32_t lane, uint32_t usb,
                                             const uint32_t *__restrict__ look, uint32_t nlook, uint32_t voff)
{
    const uint32_t b = p / usb;
    const uint32_t cmin = p > usb ? p - usb : 0u;
    uint64_t best = ~0ull;
    /* The four scans, SXU steps of 64 records each with their loads in flight together (a bucket holds ~256 records: one
     * or two round trips a scan where a load per step made four or five; round 5) */
    constexpr uint32_t SXU = 4;
    const uint32_t bk0 = (b - X.bl0) * 256u + v;
    const bool prev = b > X.bl0;
    const uint32_t bk1 = prev ? bk0 - 256u : bk0;
    const uint32_t c0a = X.off_c[bk0], c0e = X.off_c[bk0 + 1], h0a = X.off_h[bk0], h0e = X.off_h[bk0 + 1];
    const uint32_t c1a = prev ? X.off_c[bk1] : 0u, c1e = prev ? X.off_c[bk1 + 1] : 0u;
    const uint32_t h1a = prev ? X.off_h[bk1] : 0u, h1e = prev ? X.off_h[bk1 + 1] : 0u;
    const uint64_t T = (uint64_t)b * usb;
    for (uint32_t i0 = c0a + lane; i0 < c0e; i0 += 64 * SXU) {
        uint32_t cc[SXU];
#pragma unroll
        for (uint32_t u = 0; u < SXU; u++) cc[u] = X.cells[min(i0 + 64u * u, c0e - 1u)].x;
#pragma unroll
        for (uint32_t u = 0; u < SXU; u++) {
            const uint32_t c = cc[u];
            if (i0 + 64u * u < c0e && c < p) {                                 /* (a cell of block b is never older than p - sb) */
                const uint64_t key = ((uint64_t)sx_own(c, look, nlook, voff) << 32) | c;
                best = key < best ? key : best;
            }
        }
    }
    for (uint32_t i0 = h0a + lane; i0 < h0e; i0 += 64 * SXU) {
        uint2 t[SXU];
        uint32_t d[SXU];
#pragma unroll
        for (uint32_t u = 0; u < SXU; u++) { const uint32_t i = min(i0 + 64u * u, h0e - 1u); t[u] = X.hx[i]; d[u] = X.hd[i]; }
#pragma unroll
        for (uint32_t u = 0; u < SXU; u++) {
            if (i0 + 64u * u < h0e && (uint64_t)t[u].x + usb < p) {
                const uint64_t key = ((uint64_t)t[u].y << 32) | d[u];
                best = key < best ? key : best;
            }
        }
    }
    for (uint32_t i0 = c1a + lane; i0 < c1e; i0 += 64 * SXU) {
        uint2 t[SXU];
#pragma unroll
        for (uint32_t u = 0; u < SXU; u++) t[u] = X.cells[min(i0 + 64u * u, c1e - 1u)];
#pragma unroll
        for (uint32_t u = 0; u < SXU; u++) {
            if (i0 + 64u * u < c1e && t[u].x >= cmin) {
                const uint64_t key = ((uint64_t)t[u].y << 32) | t[u].x;
                best = key < best ? key : best;
            }
        }
    }
    for (uint32_t i0 = h1a + lane; i0 < h1e; i0 += 64 * SXU) {
        uint2 t[SXU];
        uint32_t d[SXU];
#pragma unroll
        for (uint32_t u = 0; u < SXU; u++) { const uint32_t i = min(i0 + 64u * u, h1e - 1u); t[u] = X.hx[i]; d[u] = X.hd[i]; }
#pragma unroll
        for (uint32_t u = 0; u < SXU; u++) {
            if (i0 + 64u * u < h1e && (uint64_t)t[u].x + usb >= T && (uint64_t)t[u].x + usb < p && d[u] >= cmin) {
                const uint64_t key = ((uint64_t)t[u].y << 32) | d[u];
                best = key < best ? key : best;
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)best, d, 64);
        best = o < best ? o : best;
    }
    return p - (uint32_t)(best & 0xFFFFFFFFu);
}

__device__ __forceinline__ void rank_token(uint32_t k, const uint32_t lane, const uint8_t *__restrict__ in, uint32_t n, int sb, int ob, int lb,
                                           uint32_t RP, uint32_t TILE, const uint32_t *__restrict__ ranks_all,
                                           const uint32_t *__restrict__ chain, const uint8_t *__restrict__ maxlen,
                                           const uint32_t *__restrict__ ofs, const uint2 *__restrict__ ent, uint32_t dbase,
                                           uint32_t *__restrict__ tokval, const uint32_t *__restrict__ look, uint32_t nlook, uint32_t voff,
                                           uint32_t whole_order /* the order holds every position < n of the region's RP slots
                                                                   (lz77k_big_sort_shared), not only its first R */,
                                           const sx_index &X /* off_c == null: no index, length-1 tokens walk like the others */,
                                           bool known = false, uint32_t kp = 0, uint32_t klen = 0, uint32_t knext = 0, uint32_t kry = 0
                                           /* known: the caller has the token's position, length, next byte and rank (k_tokens_rank_group:
                                              four dependent round trips a deferred token need not repeat) */)
{
    const uint32_t p = known ? kp : chain[k];
    const uint32_t len = known ? klen : maxlen[p];
    const uint32_t next = known ? knext : in[p + len];
    const uint32_t usb = (uint32_t)sb;
    uint32_t off = 0;
    if (len == 1 && X.off_c) {
        off = sx_query(X, p, in[p], lane, usb, look, nlook, voff);
    } else if (len > 0) {
        const uint32_t reg = p >= usb ? (p - usb) / TILE : 0u;          /* the region whose walk answered p */
        const uint32_t t0 = reg * TILE, ly = p - t0;
        const uint64_t rend = (uint64_t)t0 + (whole_order ? RP : TILE + usb);
        const uint32_t R = (rend < n ? (uint32_t)rend : n) - t0;            /* ranks [0, R) are sorted positions */
        const uint32_t *rk = ranks_all + (size_t)reg * (2 * (size_t)RP + 8), *ix = rk + RP + 8;
        const uint8_t *by = in + t0, *q = in + p;
        const uint32_t ry = known ? kry : rk[ly];
        const bool up = lane >= 32;
        const uint32_t sub = lane & 31;
        bool open_dn = true, open_up = true;                             /* wave-uniform */
        uint64_t best = ~0ull;
        auto shares = [&](uint32_t pos) -> bool {                        /* len bytes at pos == len bytes at p ? */
            const uint8_t *r = by + pos;
            for (uint32_t j = 0; j < len; j += 8) {
                uint64_t x = ld64u(r + j) ^ ld64u(q + j);
                const uint32_t rem = len - j;
                if (rem < 8) x &= (1ull << (8 * rem)) - 1ull;
                if (x) return false;
            }
            return true;
        };
        auto consider = [&](uint32_t e) {                                /* candidate at local index e: its priority at time p */
            const uint32_t c = t0 + e;
            uint32_t prio = c < nlook ? look[c] : c + voff, latest = 0;
            bool any = false;
            const uint32_t lo = c > dbase ? ofs[c - dbase - 1] : 0, hi = c >= dbase ? ofs[c - dbase] : 0;
            for (uint32_t i = lo; i < hi; i++) {
                const uint2 t = ent[i];
                if ((uint64_t)t.x + usb < p && (!any || t.x > latest)) { any = true; latest = t.x; prio = t.y; }
            }
            const uint64_t key = ((uint64_t)prio << 32) | c;
            best = key < best ? key : best;
        };
        bool long_run = false;
        for (uint32_t base = 1; open_dn || open_up; base += 32) {
            if (base > 4 * 32) { long_run = true; break; }
            const uint32_t d = base + sub;
            const bool live = up ? (open_up && ry + d < R) : (open_dn && d <= ry);
            uint32_t e = 0;
            bool same = false;
            if (live) e = ix[up ? ry + d : ry - d];
            /* The order is sorted: if the FARTHEST element of a round still shares the len bytes, so do the
             * 31 before it; only the round that contains the end of the run compares on every lane. */
            const bool probe = live && sub == 31;
            if (probe) same = shares(e);
            const uint64_t far = __ballot(same);
            const bool whole = (far >> (up ? 63 : 31)) & 1ull;
            if (live && !probe) same = whole || shares(e);
            const uint64_t okm = __ballot(same);
            const uint32_t ok_dn = (uint32_t)okm, ok_up = (uint32_t)(okm >> 32);
            const uint32_t n_dn = ok_dn == 0xFFFFFFFFu ? 32u : (uint32_t)__builtin_ctz(~ok_dn);   /* leading successes */
            const uint32_t n_up = ok_up == 0xFFFFFFFFu ? 32u : (uint32_t)__builtin_ctz(~ok_up);
            if (same && sub < (up ? n_up : n_dn) && e < ly && ly - e <= usb) consider(e);
            open_dn = open_dn && n_dn == 32u;
            open_up = open_up && n_up == 32u;
        }
        if (long_run) {
            /* The run goes on (short matches share their bytes with thousands of positions; a stretch of
             * equal bytes makes the WHOLE region one run, four times the window).  Find where it ends in
             * each direction -- "shares" is monotone along the order, a 64-ary search needs three or four
             * rounds -- then either finish the walk without comparing bytes, or, if the run is longer than
             * half the window, sweep the window itself (sb coalesced rank loads) and keep the positions
             * whose rank falls inside the run. */
            auto extent = [&](bool updir) -> uint32_t {                  /* largest rank offset still in the run */
                uint32_t lo = 0, hi = (updir ? R - 1 - ry : ry) + 1;     /* true at lo, false (out of range) at hi */
                while (hi - lo > 1) {
                    const uint64_t span = hi - lo;
                    const uint32_t d = lo + (uint32_t)(span * (lane + 1) / 65);          /* lo <= d < hi, increasing in lane */
                    const bool ok = d == lo || shares(ix[updir ? ry + d : ry - d]);
                    const uint64_t m = __ballot(ok);
                    const uint32_t top = m == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m);   /* lanes 0..top-1 are in */
                    const uint32_t nlo = top ? lo + (uint32_t)(span * top / 65) : lo;
                    const uint32_t nhi = top < 64 ? lo + (uint32_t)(span * (top + 1) / 65) : hi;
                    lo = nlo;
                    hi = nhi;
                }
                return lo;
            };
            const uint32_t d_dn = open_dn ? extent(false) : 0u, d_up = open_up ? extent(true) : 0u;
            if (max(d_dn, d_up) > usb / 2) {
                /* sweeping the window (sb/64 rounds) is cheaper than walking the run (max(d)/32 rounds) */
                const uint32_t r_lo = ry - d_dn, r_hi = ry + d_up;
                for (uint32_t c0 = ly > usb ? ly - usb : 0u; c0 < ly; c0 += 64) {
                    const uint32_t e = c0 + lane;
                    if (e < ly) {
                        const uint32_t rc = rk[e];
                        if (rc >= r_lo && rc <= r_hi) consider(e);
                    }
                }
            } else {
                /* the rest of the run, now without comparing bytes.  WB steps of 32 cells a side at a time, the loads of a
                 * stage together: a step was three dependent round trips (the order, the cell's priority and list bounds,
                 * the list), a token with 4 K cells on a side 128 steps -- the eight wavefronts of a SIMD cannot hide that
                 * (round 5: 2.8 G run cells on S3, 60 % of them in runs of a thousand cells and more) */
                constexpr uint32_t WB = 4;
                const uint32_t lim = up ? d_up : d_dn;
                for (uint32_t d0 = 4 * 32 + 1 + sub; d0 <= lim; d0 += 32 * WB) {
                    uint32_t e[WB], pr[WB], l0[WB], h0[WB];
                    bool ok[WB];
#pragma unroll
                    for (uint32_t u = 0; u < WB; u++) { const uint32_t d = min(d0 + 32u * u, lim); e[u] = ix[up ? ry + d : ry - d]; }
#pragma unroll
                    for (uint32_t u = 0; u < WB; u++) {
                        ok[u] = d0 + 32u * u <= lim && e[u] < ly && ly - e[u] <= usb;
                        const uint32_t c = ok[u] ? t0 + e[u] : p;                  /* (p: a cell every lane may read) */
                        pr[u] = c < nlook ? look[c] : c + voff;
                        l0[u] = c > dbase ? ofs[c - dbase - 1] : 0;
                        h0[u] = c >= dbase ? ofs[c - dbase] : 0;
                    }
#pragma unroll
                    for (uint32_t u = 0; u < WB; u++) {
                        if (ok[u]) {
                            uint32_t prio = pr[u], latest = 0;
                            bool any = false;
                            for (uint32_t i = l0[u]; i < h0[u]; i++) {
                                const uint2 t = ent[i];
                                if ((uint64_t)t.x + usb < p && (!any || t.x > latest)) { any = true; latest = t.x; prio = t.y; }
                            }
                            const uint64_t key = ((uint64_t)prio << 32) | (t0 + e[u]);
                            best = key < best ? key : best;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)best, d, 64);
            best = o < best ? o : best;
        }
        off = p - (uint32_t)(best & 0xFFFFFFFFu);
    }
    if (lane == 0) {
        const uint32_t omask = ob >= 32 ? 0xFFFFFFFFu : (1u << ob) - 1u;
        tokval[k] = (off & omask) | (len << ob) | (next << (ob + lb));
    }
}


/* ---- hand-overs by RANK of their cell (round 6) ----------------------------------------------------------------------
 * The walk above pays three requests at unrelated addresses for every cell of a token's run that lies in its window (the
 * cell's own priority or look[], the bounds of its hand-over list, the list): 158 GB of HBM traffic for S3's 327 MB of
 * algorithmic bytes, a 128-byte line for every 4-byte word (profiles/r05_c2_pmc_summary.csv).  But (tree.c:202-231)
 *     prio_p(c) = min( what c came with,  the hand-overs into c by evictions before p )
 * and the token wants the argmin over the members of its run inside its window, a minimum over a UNION:
 *  (a) a member's own priority is its position (+ voff: tree.c:102-105, a new node is a leaf), monotone in c -- among the own
 *      priorities the winner is the OLDEST member in the window: no look-up per cell, only positions (the region's order
 *      ix[] read along the run, coalesced; or, for a run of more than 512 cells, the window's ranks rk[] read upwards from
 *      p - sb until one falls inside the run);
 *  (b) a hand-over (x -> c, v) competes with its v where c is a member, c >= p - sb and x + sb < p; a stale one never wins
 *      (the one that replaced it in the same cell is lower).  With a region's hand-overs sorted by the RANK of their cell in
 *      the region's key order, the hand-overs of a run [r_lo, r_hi] are ONE contiguous range of 10-byte records
 *      (k_hr_count / scan / k_hr_scatter, per token launch: a counting sort whose key is a permutation).
 * Cells of a later segment's look-back carry a rank, not their position (look[]): a token whose window reaches into them
 * keeps the walk above. */
struct hr_index {
    const uint32_t *hofs;                            /* per region of the launch RP + 8 words: [r] = first record of the cell at rank r, [r + 1] its end; null: no index */
    const uint4 *rec;                                /* a record: x = the priority handed over, y = its destination cell, z = cell - eviction (1 .. sb); one 16-byte store / load */
    unsigned long long *rcache;                      /* per region HR_CACHE words: the rank bounds of a run, by a hash of its bytes (below) */
    uint32_t reg0;                                   /* first region of the index */
};
#define HR_CACHE 2048u

__device__ __forceinline__ uint32_t hr_region_cells(uint32_t reg, uint32_t RP, uint32_t TILE, uint32_t usb, uint32_t n, uint32_t whole_order)
{
    const uint64_t t0 = (uint64_t)reg * TILE, rend = t0 + (whole_order ? RP : TILE + usb);
    return (uint32_t)((rend < n ? rend : (uint64_t)n) - t0);
}

__global__ __launch_bounds__(256) void k_hr_count(const uint32_t *__restrict__ ranks_all, uint32_t RP, uint32_t TILE, uint32_t usb, uint32_t n,
                                                  uint32_t whole_order, uint32_t reg0, const uint32_t *__restrict__ ofs, uint32_t dbase, uint32_t c1,
                                                  uint32_t *__restrict__ hcnt)
{
    const uint32_t reg = reg0 + blockIdx.y, e = blockIdx.x * 256u + threadIdx.x;
    if ((uint64_t)reg * TILE >= n) return;
    const uint32_t R = hr_region_cells(reg, RP, TILE, usb, n, whole_order);
    if (e >= R) return;
    const uint32_t c = reg * TILE + e;
    if (c < dbase || c >= c1) return;
    const uint32_t lo = c > dbase ? ofs[c - dbase - 1] : 0u, hi = ofs[c - dbase];
    if (hi > lo) hcnt[(size_t)blockIdx.y * (RP + 8) + (ranks_all + (size_t)reg * (2 * (size_t)RP + 8))[e]] = hi - lo;
}

__global__ __launch_bounds__(256) void k_hr_scatter(const uint32_t *__restrict__ ranks_all, uint32_t RP, uint32_t TILE, uint32_t usb, uint32_t n,
                                                    uint32_t whole_order, uint32_t reg0, const uint32_t *__restrict__ ofs, const uint2 *__restrict__ ent,
                                                    uint32_t dbase, uint32_t c1, const uint32_t *__restrict__ hofs, uint4 *__restrict__ rec)
{
    const uint32_t reg = reg0 + blockIdx.y, e = blockIdx.x * 256u + threadIdx.x;
    if ((uint64_t)reg * TILE >= n) return;
    const uint32_t R = hr_region_cells(reg, RP, TILE, usb, n, whole_order);
    if (e >= R) return;
    const uint32_t c = reg * TILE + e;
    if (c < dbase || c >= c1) return;
    const uint32_t lo = c > dbase ? ofs[c - dbase - 1] : 0u, hi = ofs[c - dbase];
    if (hi <= lo) return;
    uint32_t at = hofs[(size_t)blockIdx.y * (RP + 8) + (ranks_all + (size_t)reg * (2 * (size_t)RP + 8))[e]];
    for (uint32_t i = lo; i < hi; i++, at++) {
        const uint2 t = ent[i];                      /* (eviction, priority handed over) */
        rec[at] = make_uint4(t.y, c, c - t.x, 0u);
    }
}

#define HR_ENUM 512u                                 /* runs up to this many cells: the oldest member in the window by reading the order along the run */

/* One token, the whole wavefront, from the index above.  dn / up: cells of the run already known below / above the token's
 * own rank (the group phase of k_tokens_rank_group found them), open_*: that direction may go on. */
__device__ __forceinline__ void rank_token_hr(uint32_t k, const uint32_t lane, const uint8_t *__restrict__ in, uint32_t n, uint32_t usb, int ob, int lb,
                                              uint32_t RP, uint32_t TILE, const uint32_t *__restrict__ ranks_all, uint32_t *__restrict__ tokval,
                                              uint32_t voff, uint32_t whole_order, const hr_index &H, uint32_t p, uint32_t len, uint32_t next, uint32_t ry,
                                              uint32_t dn, bool open_dn, uint32_t up, bool open_up)
{
    const uint32_t reg = p >= usb ? (p - usb) / TILE : 0u;
    const uint32_t t0 = reg * TILE, ly = p - t0;
    const uint32_t R = hr_region_cells(reg, RP, TILE, usb, n, whole_order);
    const uint32_t *rk = ranks_all + (size_t)reg * (2 * (size_t)RP + 8), *ix = rk + RP + 8;
    const uint32_t *hofs = H.hofs + (size_t)(reg - H.reg0) * (RP + 8);
    const uint8_t *by = in + t0, *q = in + p;
    auto shares = [&](uint32_t pos) -> bool {                        /* len bytes at pos == len bytes at p ? */
        const uint8_t *r = by + pos;
        for (uint32_t j = 0; j < len; j += 8) {
            uint64_t x = ld64u(r + j) ^ ld64u(q + j);
            const uint32_t rem = len - j;
            if (rem < 8) x &= (1ull << (8 * rem)) - 1ull;
            if (x) return false;
        }
        return true;
    };
    /* Where the run ends.  Tokens that share their len bytes share their run, and the tokens that come here are the ones with
     * the common prefixes: the bounds of a run are kept per region under a hash of (len, bytes) -- ONE 64-bit word, so a torn
     * or colliding entry is impossible / harmless: whatever the word says is VALIDATED with four probes in one round (both
     * ends share, their outer neighbours do not, the token's own rank lies between) before it is believed.  A miss searches
     * -- "shares" is monotone along the order: a 32-ary search per direction, both directions at once, from what the group
     * phase already knows -- and leaves its result behind. */
    uint32_t d_dn = dn, d_up = up, sp0 = 0, sp1 = 0;
    bool have_sp = false;
    if (open_dn || open_up) {
        unsigned long long *slot;
        {
            uint64_t h = ld64u(q) & (len >= 8 ? ~0ull : (1ull << (8 * len)) - 1ull);
            if (len > 8) h ^= ld64u(q + len - 8) * 0x9E3779B97F4A7C15ull;
            h = (h ^ len) * 0xD6E8FEB86659FD93ull;
            slot = H.rcache + (size_t)(reg - H.reg0) * HR_CACHE + (uint32_t)(h >> 53);           /* HR_CACHE = 2^11 */
        }
        const unsigned long long w = *slot;
        const uint32_t ca = (uint32_t)w, cb = (uint32_t)(w >> 32);                               /* r_lo + 1, r_hi + 1; 0: empty */
        bool hit = false;
        const bool cand = ca != 0u && ca - 1u <= ry && cb - 1u >= ry && cb <= R;
        /* (the record range of the cached run travels with the validation's probes: one round trip less on a hit) */
        sp0 = hofs[cand ? ca - 1u : ry];
        sp1 = hofs[(cand ? cb - 1u : ry) + 1u];
        if (cand) {
            const uint32_t a = ca - 1u, b = cb - 1u;
            /* lanes 0..3: rank a, rank b (must share), a - 1, b + 1 (must not, where they exist) */
            const uint32_t rr = lane == 0 ? a : lane == 1 ? b : lane == 2 ? (a ? a - 1u : a) : (b + 1u < R ? b + 1u : b);
            const bool exists = lane < 2 || (lane == 2 ? a > 0u : b + 1u < R);
            bool ok = true;
            if (lane < 4) { const bool sh = shares(ix[rr]); ok = lane < 2 ? sh : (!exists || !sh); }
            hit = __ballot(!ok) == 0ull;
            if (hit) { d_dn = ry - a; d_up = b - ry; have_sp = true; }
        }
        if (!hit) {
            const bool updir = lane >= 32;
            const uint32_t sl = lane & 31u;
            uint32_t lo = updir ? up : dn, hi = (updir ? R - 1u - ry : ry) + 1u;             /* true at lo, false (out of range) at hi */
            if (!(updir ? open_up : open_dn)) hi = lo + 1u;                                   /* (this direction is settled) */
            while (__ballot(hi - lo > 1u)) {
                const uint64_t span = hi - lo;
                const uint32_t d = lo + (uint32_t)(span * (sl + 1u) / 33u);                   /* lo <= d < hi, increasing in the lane */
                const bool ok = hi - lo <= 1u || d == lo || shares(ix[updir ? ry + d : ry - d]);
                const uint64_t m = __ballot(ok);
                const uint32_t mine = (uint32_t)(m >> (updir ? 32 : 0));
                const uint32_t top = mine == 0xFFFFFFFFu ? 32u : (uint32_t)__builtin_ctz(~mine); /* lanes 0 .. top-1 of my half are in */
                if (hi - lo > 1u) {
                    const uint32_t nlo = top ? lo + (uint32_t)(span * top / 33u) : lo;
                    const uint32_t nhi = top < 32u ? lo + (uint32_t)(span * (top + 1u) / 33u) : hi;
                    lo = nlo;
                    hi = nhi;
                }
            }
            d_dn = (uint32_t)__builtin_amdgcn_readlane((int)lo, 0);
            d_up = (uint32_t)__builtin_amdgcn_readlane((int)lo, 32);
            if (lane == 0) *slot = (unsigned long long)(ry - d_dn + 1u) | ((unsigned long long)(ry + d_up + 1u) << 32);
        }
    }
    const uint32_t r_lo = ry - d_dn, r_hi = ry + d_up, run_len = d_dn + d_up + 1u;
    const uint32_t w_lo = ly > usb ? ly - usb : 0u;                  /* the window's first cell (local) */
    /* (a) the oldest member of the run inside the window */
    uint32_t oldest = ~0u;
    bool window_seen = false;
    if (run_len > HR_ENUM) {
        /* a member every 4 RP / run_len positions on average: the window's ranks upwards from its first cell, at most as many
         * loads as reading the run itself would take */
        const uint32_t cap = run_len / 64u;
        uint32_t e0 = w_lo;
        for (uint32_t st = 0; e0 < ly && st < cap && oldest == ~0u; e0 += 256u, st += 4u) {
            uint32_t rc[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) rc[u] = rk[min(e0 + 64u * u + lane, ly - 1u)];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t e = e0 + 64u * u + lane;
                const uint64_t m = __ballot(e < ly && rc[u] >= r_lo && rc[u] <= r_hi);
                if (m && oldest == ~0u) oldest = e0 + 64u * u + (uint32_t)__builtin_ctzll(m);      /* (wave-uniform) */
            }
        }
        window_seen = oldest == ~0u && e0 >= ly;            /* (no member in the window: cannot happen for len > 0) */
    }
    if (oldest == ~0u && !window_seen) {
        for (uint32_t r0 = r_lo; r0 <= r_hi; r0 += 256u) {
            uint32_t ee[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) ee[u] = ix[min(r0 + 64u * u + lane, r_hi)];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++)
                if (r0 + 64u * u + lane <= r_hi && ee[u] >= w_lo && ee[u] < ly) oldest = min(oldest, ee[u]);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) oldest = min(oldest, (uint32_t)__shfl_xor((int)oldest, d, 64));
    }
    uint64_t best = ~0ull;
    if (oldest != ~0u) best = ((uint64_t)(t0 + oldest + voff) << 32) | (t0 + oldest);
    /* (b) the hand-overs into the run's cells: one contiguous range of records */
    const uint32_t i0 = have_sp ? sp0 : hofs[r_lo], i1 = have_sp ? sp1 : hofs[r_hi + 1u];
    for (uint32_t ib = i0; ib < i1; ib += 256u) {
        uint4 rr[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) rr[u] = H.rec[min(ib + 64u * u + lane, i1 - 1u)];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            /* the eviction lies before the window, its cell inside it (a cell is below p: c = x + S[x] < x + sb < p) */
            if (ib + 64u * u + lane < i1 && (uint64_t)(rr[u].y - rr[u].z) + usb < p && (uint64_t)rr[u].y + usb >= p) {
                const uint64_t key = ((uint64_t)rr[u].x << 32) | rr[u].y;
                best = key < best ? key : best;
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)best, d, 64);
        best = o < best ? o : best;
    }
    if (lane == 0) {
        const uint32_t off = p - (uint32_t)(best & 0xFFFFFFFFu);
        const uint32_t omask = ob >= 32 ? 0xFFFFFFFFu : (1u << ob) - 1u;
        tokval[k] = (off & omask) | (len << ob) | (next << (ob + lb));
    }
}

__global__ __launch_bounds__(256) void k_tokens_rank(const uint8_t *__restrict__ in, uint32_t n, int sb, int ob, int lb,
                                                     uint32_t RP, uint32_t TILE, const uint32_t *__restrict__ ranks_all,
                                                     const uint32_t *__restrict__ chain, uint32_t ntok,
                                                     const uint8_t *__restrict__ maxlen, const uint32_t *__restrict__ ofs,
                                                     const uint2 *__restrict__ ent, uint32_t dbase, uint32_t *__restrict__ tokval,
                                                     const uint32_t *__restrict__ look, uint32_t nlook, uint32_t voff, uint32_t whole_order,
                                                     sx_index X)
{
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w < ntok) rank_token(w, threadIdx.x & 63, in, n, sb, ob, lb, RP, TILE, ranks_all, chain, maxlen, ofs, ent, dbase, tokval, look, nlook, voff, whole_order, X);
}

/* The same walk with LPT lanes per token (LPT / 2 down, LPT / 2 up) and 64 / LPT tokens per wavefront: once the tokens
 * of length one come from the buckets and most runs are a handful of cells, a wavefront per token is five dependent round
 * trips to HBM with one token's worth of loads in flight, 28.7 M wavefronts on S3.  A token whose run is still open after
 * RANKG_ROUNDS rounds in one direction, and every token of length one, is then resolved by the whole wavefront
 * (rank_token): the 64-ary search for the ends of a long run and the bucket scans want all its lanes. */
#define RANKG_ROUNDS 4u

template <int LPT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_tokens_rank_group(const uint8_t *__restrict__ in, uint32_t n, int sb, int ob, int lb,
                                                           uint32_t RP, uint32_t TILE, const uint32_t *__restrict__ ranks_all,
                                                           const uint32_t *__restrict__ chain, uint32_t ntok,
                                                           const uint8_t *__restrict__ maxlen, const uint32_t *__restrict__ ofs,
                                                           const uint2 *__restrict__ ent, uint32_t dbase, uint32_t *__restrict__ tokval,
                                                           const uint32_t *__restrict__ look, uint32_t nlook, uint32_t voff,
                                                           uint32_t whole_order, sx_index X, hr_index H,
                                                           uint32_t probe /* variants build, timing only (wrong output): 1 no deferred tokens, 2 no bucket tokens, 4 no long runs */)
{
    constexpr uint32_t HL = LPT / 2, TPW = 64 / LPT;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t grp = lane / LPT, gl = lane % LPT, sub = gl % HL;
    const bool up = gl >= HL;
    const uint32_t kk = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * TPW + grp;
    const bool valid = kk < ntok;
    const uint32_t k = valid ? kk : ntok - 1u;
    const uint32_t p = chain[k];
    const uint32_t len = maxlen[p];
    const uint32_t next = in[p + len];
    const uint32_t usb = (uint32_t)sb;
    const uint32_t reg = p >= usb ? (p - usb) / TILE : 0u;              /* the region whose walk answered p */
    const uint32_t t0 = reg * TILE, ly = p - t0;
    const uint64_t rend = (uint64_t)t0 + (whole_order ? RP : TILE + usb);
    const uint32_t R = (rend < n ? (uint32_t)rend : n) - t0;                /* ranks [0, R) are sorted positions */
    const uint32_t *rk = ranks_all + (size_t)reg * (2 * (size_t)RP + 8), *ix = rk + RP + 8;
    const uint8_t *by = in + t0, *q = in + p;
    const uint32_t ry = rk[ly];
    const uint64_t q0 = ld64u(q);                                          /* (the input is padded past n) */
    const uint64_t m0 = len >= 8 ? ~0ull : (1ull << (8 * len)) - 1ull;
    const bool whole_wave = len == 1 && X.off_c;                           /* the buckets: all 64 lanes, below */
    /* the hand-overs by rank serve a token whose whole window lies past the look-back cells (their priorities are carried
     * ranks, not positions) */
    const bool by_rank = H.hofs && (nlook == 0u || p >= nlook + usb);
    const uint32_t *hofs = by_rank ? H.hofs + (size_t)(reg - H.reg0) * (RP + 8) : nullptr;
    bool open = valid && len > 0 && !whole_wave;                           /* my direction of my token */
    uint32_t found = 0;                                                    /* cells of the run seen in my direction */
    uint64_t best = ~0ull;
    for (uint32_t r = 0; r < RANKG_ROUNDS && __ballot(open); r++) {
        const uint32_t d = 1u + r * HL + sub;
        const bool live = open && (up ? ry + d < R : d <= ry);
        const uint32_t rr = up ? ry + d : ry - d;
        uint32_t e = 0;
        bool same = false;
        if (live) {
            e = ix[rr];
            const uint8_t *c = by + e;
            same = ((ld64u(c) ^ q0) & m0) == 0ull;
            for (uint32_t j = 8; same && j < len; j += 8) {
                uint64_t x = ld64u(c + j) ^ ld64u(q + j);
                const uint32_t rem = len - j;
                if (rem < 8) x &= (1ull << (8 * rem)) - 1ull;
                same = x == 0ull;
            }
        }
        /* my direction's lanes of this round: how many from the nearest on share? */
        const uint64_t okm = __ballot(same);
        const uint32_t mine = (uint32_t)(okm >> (grp * LPT + (up ? HL : 0u))) & ((1u << HL) - 1u);
        const uint32_t lead = mine == (1u << HL) - 1u ? HL : (uint32_t)__builtin_ctz(~mine);
        if (same && sub < lead && e < ly && ly - e <= usb) {
            const uint32_t c = t0 + e;
            uint32_t prio;
            if (by_rank) {
                /* the cell's records sit at its RANK: the bounds of neighbouring lanes are neighbouring words */
                prio = c + voff;
                const uint32_t lo = hofs[rr], hi = hofs[rr + 1u];
                for (uint32_t i = lo; i < hi; i++) {
                    const uint4 t = H.rec[i];
                    if ((uint64_t)(c - t.z) + usb < p) prio = min(prio, t.x);              /* (a hand-over only lowers its cell: the latest is the smallest) */
                }
            } else {
                uint32_t latest = 0;
                bool any = false;
                prio = c < nlook ? look[c] : c + voff;
                const uint32_t lo = c > dbase ? ofs[c - dbase - 1] : 0, hi = c >= dbase ? ofs[c - dbase] : 0;
                for (uint32_t i = lo; i < hi; i++) {
                    const uint2 t = ent[i];
                    if ((uint64_t)t.x + usb < p && (!any || t.x > latest)) { any = true; latest = t.x; prio = t.y; }
                }
            }
            const uint64_t key = ((uint64_t)prio << 32) | c;
            best = key < best ? key : best;
        }
        if (open) found += lead;
        open = open && lead == HL;
    }
    /* a direction still open, or a token for the buckets: the whole token goes to the whole wavefront */
    const uint64_t openm = __ballot(open);
    const uint64_t om = openm | __ballot(valid && whole_wave);
    const bool defer = ((om >> (grp * LPT)) & ((1ull << LPT) - 1ull)) != 0ull;
#pragma unroll
    for (int d = LPT / 2; d > 0; d >>= 1) {
        const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)best, d, 64);
        best = o < best ? o : best;
    }
    if (gl == 0 && valid && !defer) {
        const uint32_t off = len ? p - (uint32_t)(best & 0xFFFFFFFFu) : 0u;
        const uint32_t omask = ob >= 32 ? 0xFFFFFFFFu : (1u << ob) - 1u;
        tokval[k] = (off & omask) | (len << ob) | (next << (ob + lb));
    }
    uint64_t dm = __ballot(defer && gl == 0 && valid);
    const uint64_t rankm = __ballot(by_rank && !whole_wave);
#ifdef LZ77X_VARIANTS
    if (probe & 1u) dm = 0;
    if (probe & 2u) dm &= ~__ballot(whole_wave);
    if (probe & 4u) dm &= __ballot(whole_wave);
#endif
    for (; dm; dm &= dm - 1) {
        const int src = __builtin_ctzll(dm);
        const uint32_t kd = (uint32_t)__builtin_amdgcn_readlane((int)k, src);
        const uint32_t pd = (uint32_t)__builtin_amdgcn_readlane((int)p, src), ld = (uint32_t)__builtin_amdgcn_readlane((int)len, src);
        const uint32_t nd = (uint32_t)__builtin_amdgcn_readlane((int)next, src), rd = (uint32_t)__builtin_amdgcn_readlane((int)ry, src);
        if ((rankm >> src) & 1ull)
            rank_token_hr(kd, lane, in, n, usb, ob, lb, RP, TILE, ranks_all, tokval, voff, whole_order, H, pd, ld, nd, rd,
                          (uint32_t)__builtin_amdgcn_readlane((int)found, src), ((openm >> src) & 1ull) != 0ull,
                          (uint32_t)__builtin_amdgcn_readlane((int)found, src + (int)HL), ((openm >> (src + (int)HL)) & 1ull) != 0ull);
        else
            rank_token(kd, lane, in, n, sb, ob, lb, RP, TILE, ranks_all, chain, maxlen, ofs, ent, dbase, tokval, look, nlook, voff, whole_order, X, true, pd, ld, nd, rd);
    }
}

/* the hand-overs-by-rank index of a token launch over npos positions: regions, records (a cell lies in at most two regions) */
static void hr_plan(const lz77x_geom &g, size_t npos, size_t *nreg, size_t *cap)
{
    *nreg = npos / g.TILE + 3;
    *cap = 2 * (npos + 3 * (size_t)g.sb) + 1024;
}

/* bytes of the global candidate index for token positions spanning npos (0 when the LDS tile kernel applies) */
size_t lz77k_tokens_index_bytes(const lz77x_geom &g, size_t npos)
{
    if (g.fast) return 0;
    size_t need = 0;
    if (g.sb > 8192) {
        /* the (block, first byte) buckets of the length-1 tokens: npos token positions look back over npos + sb cells */
        const size_t nc = npos + (size_t)g.sb + 16, nbk = (nc / (size_t)g.sb + 3) * 256;
        need = 4 * ((nbk + 1) * 4 + 256) + lz77k_scan_tmp_bytes((uint32_t)nbk + 1) + 256 + nc * 8 + (nc + g.sb + 16) * 12 + 3 * 256;
    }
    {
        /* the hand-overs by rank (hr_index): RP + 8 offsets per region, sixteen bytes a record, the run cache */
        size_t nreg, cap;
        hr_plan(g, npos, &nreg, &cap);
        const size_t nslots = nreg * ((size_t)g.RP + 8);
        need += (nslots + 1) * 4 + lz77k_scan_tmp_bytes((uint32_t)nslots + 1) + cap * 16 + nreg * HR_CACHE * 8 + 6 * 256;
    }
#ifdef LZ77X_VARIANTS
    const size_t ntiles = (npos + BIG_TT - 1) / BIG_TT;
    const size_t v3 = ntiles * ((size_t)BIG_KEYS + BIG_TT + (size_t)g.sb + 8) * sizeof(uint32_t) + 256;
    if (v3 > need) need = v3;
#endif
    return need;
}

/* one word per tile + 1: the sorted-order kernel cuts every region into ceil(TILE / TS_TT) tiles (TILE >= 3064) */
size_t lz77k_tokens_tmp_bytes(uint32_t n, const lz77x_geom &) { return ((size_t)(n / 1024u) + 64) * sizeof(uint32_t); }

bool lz77k_tokens_builds_lists(const lz77x_geom &g, int variant, const void *d_ranks_all) { return variant == 0 && g.fast && d_ranks_all; }

/* tokens d_chain[0..ntok) all lie in [pos0, pos1); the hand-over index covers dst >= dbase */
hipError_t lz77k_tokens(const uint8_t *d_in, uint32_t n, const lz77x_geom &g, const uint32_t *d_chain, uint32_t ntok,
                        const uint8_t *d_maxlen, const uint32_t *d_ofs, const uint2 *d_ent, uint32_t dbase,
                        uint32_t pos0, uint32_t pos1, uint32_t *d_tokval, uint32_t *d_tstart, void *d_index, int variant,
                        hipStream_t s, hipEvent_t *ev_tie, const uint32_t *d_ranks_all, const uint32_t *d_look, uint32_t nlook, uint32_t voff,
                        const uint32_t *d_ps, const uint32_t *d_xval, unsigned long long *d_total)
{
    if (ntok == 0) return hipSuccess;
#define TIE_EV(i) do { if (ev_tie) { hipError_t ee_ = hipEventRecord(ev_tie[i], s); if (ee_ != hipSuccess) return ee_; } } while (0)
    if (variant == 0 && !g.fast && d_ranks_all) {
        TIE_EV(0);
        sx_index X = {};
        hr_index Hx = {};
        uint8_t *base = reinterpret_cast<uint8_t *>(d_index);
        size_t o = 0;
        auto take = [&](size_t bytes) { uint8_t *q = base + o; o += (bytes + 255) & ~(size_t)255; return q; };
        if (d_index && g.sb > 8192 && !LZ77X_VENV("LZ77X_NO_SHORT_INDEX")) {
            /* the (block, first byte) buckets of the cells [dbase, pos1) and of the hand-overs into them: what the tokens of
             * length one are resolved from (sx_query) */
            const uint32_t usb = (uint32_t)g.sb, c0 = dbase, c1 = pos1, nc = c1 - c0;
            const uint32_t bl0 = c0 / usb, nbl = (c1 - 1u) / usb - bl0 + 1u, nbk = nbl * 256u;
            uint32_t *off_c = reinterpret_cast<uint32_t *>(take(((size_t)nbk + 1) * 4)), *off_h = reinterpret_cast<uint32_t *>(take(((size_t)nbk + 1) * 4));
            uint32_t *cur_c = reinterpret_cast<uint32_t *>(take((size_t)nbk * 4)), *cur_h = reinterpret_cast<uint32_t *>(take((size_t)nbk * 4));
            void *stmp = take(lz77k_scan_tmp_bytes(nbk + 1));
            uint2 *cells = reinterpret_cast<uint2 *>(take((size_t)nc * 8));
            uint2 *hx = reinterpret_cast<uint2 *>(take(((size_t)nc + usb + 16) * 8));
            uint32_t *hd = reinterpret_cast<uint32_t *>(take(((size_t)nc + usb + 16) * 4));
            hipError_t e = hipMemsetAsync(base, 0, (size_t)(reinterpret_cast<uint8_t *>(stmp) - base), s);
            if (e != hipSuccess) return e;
            const uint32_t wgs = (nc + SX_CHUNK - 1u) / SX_CHUNK;
            hipLaunchKernelGGL(k_sx_count, dim3(wgs), dim3(256), 0, s, d_in, c0, c1, usb, bl0, d_ofs, dbase, off_c, off_h);
            if ((e = lz77k_scan_u32(off_c, off_c, nbk + 1, stmp, s)) != hipSuccess) return e;
            if ((e = lz77k_scan_u32(off_h, off_h, nbk + 1, stmp, s)) != hipSuccess) return e;
            hipLaunchKernelGGL(k_sx_scatter, dim3(wgs), dim3(256), 0, s, d_in, c0, c1, usb, bl0, d_ofs, d_ent, dbase, off_c, off_h, cur_c, cur_h, cells, hx,
                               hd, d_look, nlook, voff);
            X.off_c = off_c; X.off_h = off_h; X.cells = cells; X.hx = hx; X.hd = hd; X.bl0 = bl0; X.nbl = nbl;
        }
        if (d_index && !LZ77X_VENV("LZ77X_NO_RANK_INDEX")) {
            /* the hand-overs into the cells [dbase, pos1) by (region, rank of the cell): a counting sort whose key is a
             * permutation -- counts at the ranks, one scan over all regions' slots, the records behind their offsets */
            const uint32_t usb = (uint32_t)g.sb, whole = (uint32_t)lz77k_big_sort_shared(g);
            const uint32_t reg0 = pos0 >= usb ? (pos0 - usb) / g.TILE : 0u, reg1 = pos1 - 1u >= usb ? (pos1 - 1u - usb) / g.TILE : 0u;
            const uint32_t nreg = reg1 - reg0 + 1u;
            size_t nreg_max, cap;
            hr_plan(g, (size_t)pos1 - pos0, &nreg_max, &cap);
            if (nreg > nreg_max) return hipErrorInvalidValue;
            const size_t nslots = (size_t)nreg * ((size_t)g.RP + 8);
            uint32_t *hofs = reinterpret_cast<uint32_t *>(take((nslots + 1) * 4));
            void *stmp = take(lz77k_scan_tmp_bytes((uint32_t)nslots + 1));
            uint4 *rec = reinterpret_cast<uint4 *>(take(cap * 16));
            unsigned long long *rcache = reinterpret_cast<unsigned long long *>(take((size_t)nreg * HR_CACHE * 8));
            hipError_t e = hipMemsetAsync(hofs, 0, (nslots + 1) * 4, s);
            if (e != hipSuccess) return e;
            if ((e = hipMemsetAsync(rcache, 0, (size_t)nreg * HR_CACHE * 8, s)) != hipSuccess) return e;
            const dim3 grid((g.RP + 255u) / 256u, nreg);
            hipLaunchKernelGGL(k_hr_count, grid, dim3(256), 0, s, d_ranks_all, g.RP, g.TILE, usb, n, whole, reg0, d_ofs, dbase, pos1, hofs);
            if ((e = lz77k_scan_u32(hofs, hofs, (uint32_t)nslots + 1u, stmp, s)) != hipSuccess) return e;
            hipLaunchKernelGGL(k_hr_scatter, grid, dim3(256), 0, s, d_ranks_all, g.RP, g.TILE, usb, n, whole, reg0, d_ofs, d_ent, dbase, pos1, hofs, rec);
            Hx.hofs = hofs; Hx.rec = rec; Hx.rcache = rcache; Hx.reg0 = reg0;
        }
        {
            /* several tokens per wavefront (LZ77X_RANK_LPT=64 in the variants build: one) */
            const char *le = LZ77X_VENV("LZ77X_RANK_LPT");
            const int lpt = le ? atoi(le) : 8;
            if (lpt >= 64)
                hipLaunchKernelGGL(k_tokens_rank, dim3((ntok + 3) / 4), dim3(256), 0, s, d_in, n, g.sb, g.ob, g.lb, g.RP, g.TILE, d_ranks_all,
                                   d_chain, ntok, d_maxlen, d_ofs, d_ent, dbase, d_tokval, d_look, nlook, voff, (uint32_t)lz77k_big_sort_shared(g), X);
            else {
                const uint32_t tpw = 64u / (lpt == 8 ? 8u : 16u), waves = (ntok + tpw - 1u) / tpw;
                auto fn = lpt == 8 ? k_tokens_rank_group<8> : k_tokens_rank_group<16>;
                hipLaunchKernelGGL(fn, dim3((waves + 3u) / 4u), dim3(256), 0, s, d_in, n, g.sb, g.ob, g.lb, g.RP, g.TILE, d_ranks_all, d_chain, ntok,
                                   d_maxlen, d_ofs, d_ent, dbase, d_tokval, d_look, nlook, voff, (uint32_t)lz77k_big_sort_shared(g), X, Hx,
                                   LZ77X_VENV("LZ77X_RANK_PROBE") ? (uint32_t)atoi(LZ77X_VENV("LZ77X_RANK_PROBE")) : 0u);
            }
        }
        TIE_EV(1);
        return hipGetLastError();
    }
#ifdef LZ77X_VARIANTS
    if ((variant == 0 || variant == 3) && g.sb > 8192 && d_index) {
        const uint32_t ntiles = (pos1 - pos0 + BIG_TT - 1) / BIG_TT;
        const uint32_t span = BIG_TT + (uint32_t)g.sb + 8;
        uint32_t *bs = reinterpret_cast<uint32_t *>(d_index);
        uint32_t *blist = bs + (size_t)ntiles * BIG_KEYS;
        hipError_t e = hipMemsetAsync(bs, 0, (size_t)ntiles * BIG_KEYS * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
        const dim3 grid((span + 255) / 256, ntiles);
        hipLaunchKernelGGL(k_bidx_count, grid, dim3(256), 0, s, d_in, g.sb, pos0, pos1, bs);
        hipLaunchKernelGGL(k_bidx_scan, dim3(ntiles), dim3(1024), 0, s, bs);
        hipLaunchKernelGGL(k_bidx_fill, grid, dim3(256), 0, s, d_in, g.sb, pos0, pos1, bs, blist, span);
        TIE_EV(0);
        hipLaunchKernelGGL(k_tokens_big, dim3((ntok + 3) / 4), dim3(256), 0, s, d_in, n, g.sb, g.ob, g.lb, d_chain, ntok, d_maxlen,
                           d_ofs, d_ent, dbase, pos0, bs, blist, span, d_tokval, d_look, nlook, voff);
        TIE_EV(1);
        return hipGetLastError();
    }
#else
    (void)d_index;
#endif
    if (d_ps && d_xval && d_tstart && lz77k_tokens_builds_lists(g, variant, d_ranks_all)) {
        /* production, LDS-sized windows: runs of the regions' sorted order (d_ranks_all = RP uint16 per region) */
        const ts_grid G = ts_make_grid(g);
        const uint32_t tile0 = ts_tile_of(G, pos0), ntiles = ts_tile_of(G, pos1 - 1u) - tile0 + 1u;
        const uint32_t span = TS_TT + (uint32_t)g.sb + 16;
        hipLaunchKernelGGL(k_tok_bounds_grid, dim3((ntok + 256) / 256), dim3(256), 0, s, d_chain, ntok, G, tile0, ntiles, d_tstart);
        const uint32_t budget = 77u * 1024u - 256u;                       /* two workgroups per CU: 160 KB less 2 x 3 KB of static LDS */
#ifdef LZ77X_VARIANTS
        if (LZ77X_VENV("LZ77X_TS_V4")) {
            /* round 4's kernel: lists per cell (the cross-check) */
            const uint32_t off_sorted = (span + (uint32_t)g.la + 16 + 15) & ~15u;
            const uint32_t off_inv = (off_sorted + 2 * span + 15) & ~15u;
            const uint32_t off_lofs = (off_inv + 2 * TS_TT + 15) & ~15u;
            const uint32_t off_tk = off_lofs + 2 * 8 * TS_BLOCK;                /* eight list offsets per thread: span + 2 <= 8192 */
            const uint32_t off_lent = (off_tk + TS_TB * 8 + (TS_TB + 4) * 4 + TS_TB * 4 + TS_TB * 2 * 2 + 15) & ~15u;
            /* a list entry is 6 bytes (eviction: uint16 from the tile's first, priority: uint32, two arrays); the area also
             * holds a uint16 per eviction of a tile, what the lists fall back to when the priorities do not fit */
            uint32_t ent_cap = ((budget - off_lent) / 6u) & ~1u;
            if (const char *ec = LZ77X_VENV("LZ77X_TS_ENTCAP")) ent_cap = min(ent_cap, (uint32_t)atoi(ec) & ~1u);     /* test hook: the fallback lists */
            const size_t lds = (size_t)off_lent + max((size_t)ent_cap * 6, (size_t)span * 2);
            if (lds > 48 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_tokens_sorted_v4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
            }
            TIE_EV(0);
            hipLaunchKernelGGL(k_tokens_sorted_v4, dim3((ntiles + 7u) / 8u * 8u), dim3(TS_BLOCK), lds, s, d_in, n, g.sb, g.la, g.ob, g.lb, d_chain, d_tstart, d_maxlen,
                               d_ps, d_xval, pos0, pos1, d_tokval, ent_cap, off_sorted, off_inv, off_lofs, off_tk, off_lent, d_look, nlook, voff,
                               reinterpret_cast<const uint16_t *>(d_ranks_all), g.RP, G, tile0, ntiles, d_total);
            TIE_EV(1);
            if (d_total) hipLaunchKernelGGL(k_ts_total, dim3(1), dim3(64), 0, s, d_total);
            return hipGetLastError();
        }
#endif
        /* [0, off_sorted): the 16-bit counters per slot while the lists are built, then the window bytes and the batch arrays */
        const uint32_t off_tk = (span + (uint32_t)g.la + 16 + 15) & ~15u;
        const uint32_t tk_bytes = TS_TB * 8 + 2 * (TS_TB + 4) * 4 + TS_TB * 4 + 2 * TS_TB * 2;
        const uint32_t off_sorted = (max(off_tk + tk_bytes, 2u * 8u * TS_BLOCK) + 15) & ~15u;
        const uint32_t off_inv = (off_sorted + 2 * span + 15) & ~15u;
        const uint32_t off_cofs = (off_inv + 2 * span + 15) & ~15u;
        const uint32_t off_ent = (off_cofs + 2 * (8 * TS_BLOCK / TS_CG) + 15) & ~15u;
        /* an entry is 8 bytes (slot | eviction << 16, priority: two arrays); the area also holds 4 bytes per eviction of a
         * tile, what the entries fall back to when the priorities do not fit */
        const uint32_t min_ent = (TS_TT + (uint32_t)g.sb) * 4u;
        if (off_ent + min_ent > budget) return hipErrorInvalidValue;
        uint32_t ent_cap = ((budget - off_ent) / 8u) & ~1u;
        if (const char *ec = LZ77X_VENV("LZ77X_TS_ENTCAP")) ent_cap = min(ent_cap, (uint32_t)atoi(ec) & ~1u);     /* test hook: the fallback entries */
        const size_t lds = (size_t)off_ent + max((size_t)ent_cap * 8, (size_t)min_ent);
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_tokens_sorted), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        TIE_EV(0);
        hipLaunchKernelGGL(k_tokens_sorted, dim3((ntiles + 7u) / 8u * 8u), dim3(TS_BLOCK), lds, s, d_in, n, g.sb, g.la, g.ob, g.lb, d_chain, d_tstart, d_maxlen,
                           d_ps, d_xval, pos0, pos1, d_tokval, ent_cap, off_tk, off_sorted, off_inv, off_cofs, off_ent, d_look, nlook, voff,
                           reinterpret_cast<const uint16_t *>(d_ranks_all), g.RP, G, tile0, ntiles, d_total,
                           (uint32_t)(LZ77X_VENV("LZ77X_TS_BIG") ? atoi(LZ77X_VENV("LZ77X_TS_BIG")) : TS_BIG)
#ifdef LZ77X_VARIANTS
                           , (uint32_t)(LZ77X_VENV("LZ77X_TS_PROBE") ? atoi(LZ77X_VENV("LZ77X_TS_PROBE")) : 0)
#endif
                           );
        TIE_EV(1);
        if (d_total) hipLaunchKernelGGL(k_ts_total, dim3(1), dim3(64), 0, s, d_total);
        return hipGetLastError();
    }
    if ((variant == 0 || variant == 2) && g.sb <= 8192 && d_tstart) {
        const bool bucket = variant == 0;
        const uint32_t ntiles = (pos1 - pos0 + TOK_TILE - 1) / TOK_TILE;
        const uint32_t span = TOK_TILE + (uint32_t)g.sb + 16;
        const uint32_t lofs_off = (span + (uint32_t)g.la + 16 + 15) & ~15u;
        const uint32_t bkt_off = (lofs_off + 2 * (span + 2) + 15) & ~15u;
        const uint32_t bkt_bytes = bucket ? ((TOK_HASH + 8) * 4 + 2 * span + 15) & ~15u : 0u;
        const uint32_t lent_off = bkt_off + bkt_bytes;
#ifndef TOK_LDS_KB
#define TOK_LDS_KB 78u                                               /* two workgroups per CU */
#endif
        const uint32_t budget = TOK_LDS_KB * 1024u;
        uint32_t ent_cap = lent_off + 5 * span < budget ? (budget - lent_off) / 8 : span;
        if (bucket && ent_cap < TOK_HASH / 2) ent_cap = TOK_HASH / 2; /* the lent area doubles as bcur (TOK_HASH words) */
        const size_t lds = (size_t)lent_off + (size_t)ent_cap * 8;
#ifdef LZ77X_VARIANTS
        auto fn = bucket ? k_tokens_tile<true> : k_tokens_tile<false>;
#else
        auto fn = k_tokens_tile<true>;
        if (!bucket) return hipErrorNotSupported;
#endif
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k_tok_bounds, dim3((ntok + 256) / 256), dim3(256), 0, s, d_chain, ntok, pos0, ntiles, d_tstart);
        TIE_EV(0);
        hipLaunchKernelGGL(fn, dim3(ntiles), dim3(TOK_BLOCK), lds, s, d_in, n, g.sb, g.la, g.ob, g.lb, d_chain,
                           d_tstart, d_maxlen, d_ofs, d_ent, dbase, pos0, pos1, d_tokval, ent_cap, lofs_off, lent_off, bkt_off, d_look, nlook, voff);
        TIE_EV(1);
        return hipGetLastError();
    }
    const uint32_t blocks = (ntok + 3) / 4;
    TIE_EV(0);
    hipLaunchKernelGGL(k_tokens, dim3(blocks), dim3(256), 0, s, d_in, n, g.sb, g.ob, g.lb, d_chain, ntok, d_maxlen, d_ofs, d_ent,
                       dbase, d_tokval, d_look, nlook, voff);
    TIE_EV(1);
    return hipGetLastError();
#undef TIE_EV
}

/* ------------------------------------------------------------------ k_pack ----------- */

/* bitio.c:203-239 without the per-bit loop: token k occupies stream bits [32+kT, 32+(k+1)T);
 * stream bit b is bit (b & 31) of little-endian word b >> 5.  One thread assembles one word.
 * General form: the words [w0, w0+nw) of the stream from the tokens [k_first, k_end) (tokval[0] is token
 * k_first): a segment of a long input packs the words its own tokens START in, and carries up to four
 * tokens of its predecessor in front of its own so that its first word is complete. */
__global__ void k_pack(const uint32_t *__restrict__ tokval, uint64_t k_first, uint64_t k_end, int sb, int la, int T,
                       uint32_t *__restrict__ out, uint64_t w0, uint64_t nw)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw) return;
    const uint64_t w = w0 + i;
    if (w == 0) { out[0] = (uint32_t)sb | ((uint32_t)la << 16); return; }      /* lz77.c:74-75 */
    const uint64_t b0 = 32 * (w - 1);                     /* first token-area bit of this word */
    uint64_t k = b0 / (uint64_t)T;
    if (k < k_first) k = k_first;                         /* (never needed when the carried tokens are there) */
    uint32_t word = 0;
    for (; k < k_end; k++) {
        const int64_t sh = (int64_t)(k * (uint64_t)T) - (int64_t)b0;
        if (sh >= 32) break;
        const uint32_t v = tokval[k - k_first];
        word |= sh >= 0 ? (v << sh) : (v >> (-sh));
    }
    out[i] = word;
}

hipError_t lz77k_pack_range(const uint32_t *d_tokval, uint64_t k_first, uint64_t k_end, const lz77x_geom &g, uint32_t *d_out_words,
                            uint64_t w0, uint64_t nw, hipStream_t s)
{
    if (nw == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((nw + 255) / 256);
    hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, s, d_tokval, k_first, k_end, g.sb, g.la, g.T, d_out_words, w0, nw);
    return hipGetLastError();
}

hipError_t lz77k_pack(const uint32_t *d_tokval, uint64_t ntok, const lz77x_geom &g, uint32_t *d_out_words, uint64_t nwords, hipStream_t s)
{
    return lz77k_pack_range(d_tokval, 0, ntok, g, d_out_words, 0, nwords, s);
}
