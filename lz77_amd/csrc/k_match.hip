/*
 * k_match.hip -- the match stage: what replaces tree.c:62-243 (insert / delete / find on the BST).
 *
 *   k_match      per-region key sort -> rank[] and its inverse (+ the exhaustive rank-difference pair
 *                scan kept as a cross-check, LZ77X_MATCH_VARIANT=1/3)
 *   k_walk       O(1)-per-position sliding-window neighbour search on per-lane bitmaps
 *   k_walk_final ranks -> ps[x] (in-order neighbours at eviction time) and maxlen[x] (longest match)
 */
#include "kernels_common.h"

#define MATCH_BLOCK 1024

/* workgroup barrier that orders LDS traffic only: global stores and loads in flight stay in flight (__syncthreads()
 * drains them: its fence waits for vmcnt(0)) */
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

/* Key bytes staged BACKWARDS (REV): the buffer holds by[top + 3 - i] at offset i, so the plain little-endian dword at
 * offset top - a IS the big-endian dword of the bytes a .. a+3 -- the form keys are compared in (tree.c:77 is a memcmp).
 * The production sorts stage their window this way: a key load is five LDS reads and four funnel shifts, no byte swaps
 * (four v_perm per 16-byte key, on the one instruction stream the sort is bound by). */
template <bool BYTES_LDS, bool REV>
__device__ __forceinline__ uint32_t be32_at(const uint8_t *by, uint32_t top, uint32_t a)
{
    if constexpr (REV) return ld32_at<true>(by, top - a);
    else return __builtin_bswap32(ld32_at<BYTES_LDS>(by, a));
}

/* key_less of kernels_common.h on either staging */
template <bool BYTES_LDS, bool REV>
__device__ __forceinline__ bool key_less_v(const uint8_t *by, uint32_t top, uint32_t a, uint32_t b, int la)
{
    if constexpr (!REV) return key_less<BYTES_LDS>(by, a, b, la);
    for (int w = 0; w < la; w += 4) {
        uint32_t va = be32_at<true, true>(by, top, a + (uint32_t)w), vb = be32_at<true, true>(by, top, b + (uint32_t)w);
        const int rem = la - w;
        if (rem < 4) {
            const uint32_t m = 0xFFFFFFFFu << (8 * (4 - rem));
            va &= m;
            vb &= m;
        }
        if (va != vb) return va < vb;
    }
    return a < b;
}

/*
 * Bitonic sort of RP = 16*1024 local indices held in LDS, organised by how far apart the two
 * elements of a compare-exchange are:
 *   stride <  16   : both live in ONE thread (a thread owns 16 consecutive slots): registers only,
 *                    with the first four key bytes cached so most compares never touch LDS bytes
 *   stride < 1024  : both live in the 1024-slot segment ONE wave owns: LDS, no workgroup barrier
 *   stride >= 1024 : across waves: LDS + __syncthreads (10 of the 105 steps)
 */
template <bool BYTES_LDS, bool REV = false> __device__ __forceinline__ bool key_tail_less(const uint8_t *by, uint32_t a, uint32_t b, int la, int from, uint32_t top = 0);

template <bool BYTES_LDS, bool REV = false>
__device__ __forceinline__ bool sort_less(const uint8_t *by, uint32_t a, uint32_t pa, uint32_t b, uint32_t pb, uint32_t R, int la, uint32_t top = 0)
{
    if (pa != pb) return pa < pb;
    if (a >= R || b >= R) return a < b;
    return la > 16 ? key_tail_less<BYTES_LDS, REV>(by, a, b, la, 0, top) : key_less_v<BYTES_LDS, REV>(by, top, a, b, la);
}

/* The merge sort's register stage on 16-bit slot indices (round 4): an element is ONE 64-bit word, the key's first six
 * bytes above the index, and a compare-exchange one 64-bit compare and four selects; the byte loop is behind "the six
 * bytes tie".  (With four cached bytes beside the index, one compare-exchange in four had some lane of the wavefront tie
 * and take all 64 through that loop.) */
template <bool BYTES_LDS, bool REV>
__device__ __forceinline__ bool packed_less(const uint8_t *by, uint64_t xa, uint64_t xb, uint32_t R, int la, uint32_t top)
{
    if ((xa ^ xb) >> 16) return xa < xb;
    const uint32_t a = (uint32_t)xa & 0xFFFFu, b = (uint32_t)xb & 0xFFFFu;
    if (a >= R || b >= R) return a < b;
    return la > 6 ? key_tail_less<BYTES_LDS, REV>(by, a, b, la, 6, top) : a < b;
}

template <int J, bool STATIC_DIR, bool BYTES_LDS, bool REV>
__device__ __forceinline__ void sort_packed_pass(uint64_t (&x)[16], const uint8_t *by, uint32_t R, int la, int k_static, bool up_uniform, uint32_t top)
{
#pragma unroll
    for (int r = 0; r < 16; r++) {
        if (r & J) continue;
        const int s = r | J;
        const bool up = STATIC_DIR ? ((r & k_static) == 0) : up_uniform;
        const bool s_lt_r = packed_less<BYTES_LDS, REV>(by, x[s], x[r], R, la, top);
        if (up ? s_lt_r : !s_lt_r) {
            const uint64_t t = x[r];
            x[r] = x[s];
            x[s] = t;
        }
    }
}

template <int J, bool STATIC_DIR, bool BYTES_LDS, bool REV = false>
__device__ __forceinline__ void sort_local_pass(uint32_t (&v)[16], uint32_t (&pf)[16], const uint8_t *by, uint32_t R, int la,
                                                int k_static, bool up_uniform, uint32_t top = 0)
{
#pragma unroll
    for (int r = 0; r < 16; r++) {
        if (r & J) continue;
        const int s = r | J;
        const bool up = STATIC_DIR ? ((r & k_static) == 0) : up_uniform;
        const bool s_lt_r = sort_less<BYTES_LDS, REV>(by, v[s], pf[s], v[r], pf[r], R, la, top);
        if (up ? s_lt_r : !s_lt_r) {
            const uint32_t tv = v[r], tp = pf[r];
            v[r] = v[s]; pf[r] = pf[s];
            v[s] = tv; pf[s] = tp;
        }
    }
}

/* Sorts (tail_only = false) or finishes the top phase of (tail_only = true) CH = 16*1024 slots in
 * LDS.  The top phase k == CH runs ascending iff top_up (a chunk of a larger bitonic network gets
 * its direction from its position in that network). */
template <class IdxT, bool BYTES_LDS>
__device__ __forceinline__ void region_sort_blocked(IdxT *ix, const uint8_t *by, uint32_t R, int la, uint32_t tid, bool tail_only,
                                                    bool top_up)
{
    constexpr uint32_t CH = 16 * MATCH_BLOCK;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t pmask = la >= 4 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * (4 - la));
    uint32_t v[16], pf[16];
    auto load_mine = [&]() {
        if constexpr (sizeof(IdxT) == 2) {
            const uint4 a = *reinterpret_cast<const uint4 *>(ix + 16 * tid), b = *reinterpret_cast<const uint4 *>(ix + 16 * tid + 8);
            const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = (w[r >> 1] >> (16 * (r & 1))) & 0xFFFFu;
        } else {
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const uint4 a = *reinterpret_cast<const uint4 *>(ix + 16 * tid + r);
                v[r] = a.x; v[r + 1] = a.y; v[r + 2] = a.z; v[r + 3] = a.w;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r++)
            pf[r] = v[r] < R ? (__builtin_bswap32(ld32_at<BYTES_LDS>(by, v[r])) & pmask) : 0xFFFFFFFFu;
    };
    auto store_mine = [&]() {
        if constexpr (sizeof(IdxT) == 2) {
            uint32_t w[8];
#pragma unroll
            for (int r = 0; r < 8; r++) w[r] = v[2 * r] | (v[2 * r + 1] << 16);
            *reinterpret_cast<uint4 *>(ix + 16 * tid) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4 *>(ix + 16 * tid + 8) = make_uint4(w[4], w[5], w[6], w[7]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; r += 4)
                *reinterpret_cast<uint4 *>(ix + 16 * tid + r) = make_uint4(v[r], v[r + 1], v[r + 2], v[r + 3]);
        }
    };
    /* four disjoint compare-exchanges of stride j at pairs t0, t0+tstep, ...: all index loads first,
     * then all key loads, then the stores: ~3 LDS round trips per four instead of 4x3, while staying
     * within 64 VGPRs (two workgroups per CU) */
    auto lds_step4 = [&](uint32_t k, bool top, uint32_t j, uint32_t t0, uint32_t tstep) {
        uint32_t ii[4], a[4], b[4], ka[4], kb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t t = t0 + (uint32_t)u * tstep;
            ii[u] = 2 * t - (t & (j - 1));
            a[u] = ix[ii[u]];
            b[u] = ix[ii[u] + j];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            ka[u] = a[u] < R ? (__builtin_bswap32(ld32_at<BYTES_LDS>(by, a[u])) & pmask) : 0xFFFFFFFFu;
            kb[u] = b[u] < R ? (__builtin_bswap32(ld32_at<BYTES_LDS>(by, b[u])) & pmask) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool up = top ? top_up : (ii[u] & k) == 0;
            const bool b_lt_a = sort_less<BYTES_LDS>(by, b[u], kb[u], a[u], ka[u], R, la);
            if (up ? b_lt_a : !b_lt_a) { ix[ii[u]] = (IdxT)b[u]; ix[ii[u] + j] = (IdxT)a[u]; }
        }
    };
    auto local_tail = [&](bool up) {
        sort_local_pass<8, false, BYTES_LDS>(v, pf, by, R, la, 0, up);
        sort_local_pass<4, false, BYTES_LDS>(v, pf, by, R, la, 0, up);
        sort_local_pass<2, false, BYTES_LDS>(v, pf, by, R, la, 0, up);
        sort_local_pass<1, false, BYTES_LDS>(v, pf, by, R, la, 0, up);
    };

    if (!tail_only) {
        /* phases k = 2..16: entirely inside a thread */
        load_mine();
        sort_local_pass<1, true, BYTES_LDS>(v, pf, by, R, la, 2, false);
        sort_local_pass<2, true, BYTES_LDS>(v, pf, by, R, la, 4, false);
        sort_local_pass<1, true, BYTES_LDS>(v, pf, by, R, la, 4, false);
        sort_local_pass<4, true, BYTES_LDS>(v, pf, by, R, la, 8, false);
        sort_local_pass<2, true, BYTES_LDS>(v, pf, by, R, la, 8, false);
        sort_local_pass<1, true, BYTES_LDS>(v, pf, by, R, la, 8, false);
        local_tail(((16 * tid) & 16) == 0);
        store_mine();
    }
    for (uint32_t k = tail_only ? CH : 32; k <= CH; k <<= 1) {
        const bool top = k == CH;
        uint32_t j = k >> 1;
        if (j >= 1024) {
            __syncthreads();                                          /* other waves' segments are read next */
            for (; j >= 1024; j >>= 1) {
                lds_step4(k, top, j, tid, MATCH_BLOCK);                  /* pairs tid, tid+1024, ... */
                lds_step4(k, top, j, tid + 4 * MATCH_BLOCK, MATCH_BLOCK);
                __syncthreads();
            }
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        for (; j >= 16; j >>= 1) {                                    /* inside this wave's 1024 slots */
            lds_step4(k, top, j, 512 * wave + lane, 64);
            lds_step4(k, top, j, 512 * wave + lane + 256, 64);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        load_mine();
        local_tail(top ? top_up : ((16 * tid) & k) == 0);
        store_mine();
    }
    __syncthreads();
}

/* the REV staging of N bytes (a multiple of four; four more must be addressable behind them) of the input from `base`:
 * positions past the input's 0xFF pad read 0xFF too */
template <uint32_t NT>
__device__ __forceinline__ void stage_rev(uint8_t *lby, uint32_t N, const uint8_t *in, uint64_t base, uint32_t n, uint32_t tid)
{
    const uint64_t lim = (uint64_t)n + LZ77X_PAD;
    for (uint32_t i = tid * 4; i < N; i += NT * 4) {
        const uint32_t v = base + i + 4 <= lim ? ld32u(in + base + i) : 0xFFFFFFFFu;
        *reinterpret_cast<uint32_t *>(lby + (N - 4u - i)) = __builtin_bswap32(v);
    }
}

/* copy s of load_key16's four (s = 1..3): byte o of it = byte o + s of the REV staging = input byte base + N - 1 - o - s; its
 * top s bytes would be the s bytes BEFORE base: no key reads them (0xFF) */
template <uint32_t NT>
__device__ __forceinline__ void stage_rev_shifted(uint8_t *copy, uint32_t s, uint32_t N, const uint8_t *in, uint64_t base, uint32_t n, uint32_t tid)
{
    const uint64_t lim = (uint64_t)n + LZ77X_PAD;
    for (uint32_t i = tid * 4; i < N; i += NT * 4) {
        uint32_t v;
        if (base + i >= s) v = base + i - s + 4 <= lim ? ld32u(in + (base + i - s)) : 0xFFFFFFFFu;
        else {
            v = 0;
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) v |= (base + i + q >= s ? (uint32_t)in[base + i + q - s] : 0xFFu) << (8u * q);
        }
        *reinterpret_cast<uint32_t *>(copy + (N - 4u - i)) = __builtin_bswap32(v);
    }
}

/* The merge levels compare whole 16-byte key heads held in registers (big-endian dwords, bytes past
 * `la` masked off): with a 4-byte prefix nearly every step had SOME lane of the wave tie and drag all
 * 64 through the byte-loop fallback -- neighbours in key order share long prefixes.  For la <= 16
 * (C1) a compare never touches memory again. */
struct key16 { uint64_t hi, lo; };

/* CS > 0 (REV only): the staged bytes exist FOUR times, copy s (at by + s * CS) shifted down by s bytes, so that the sixteen
 * bytes of any key are four ALIGNED dwords of copy (offset & 3): two ds_read2_b32 and no funnel shifts, where one copy takes
 * five reads and four v_alignbyte.  The chunk sort is bound by its LDS index pipe as much as by its VALU (SQ_LDS_IDX_ACTIVE
 * 0.86 G of the launch's 1.0 G CU-cycles, 63 % of them bank conflicts of exactly these reads): a fifth fewer dwords. */
template <bool BYTES_LDS, bool REV = false, uint32_t CS = 0>
__device__ __forceinline__ key16 load_key16(const uint8_t *by, uint32_t a, bool valid, const uint32_t (&m)[4], uint32_t top = 0)
{
    uint32_t k[4];
    if constexpr (REV && CS > 0) {
        (void)valid;
        const uint32_t lo = top - 12u - a;
        const uint32_t *w = reinterpret_cast<const uint32_t *>(by + (lo & 3u) * CS + (lo & ~3u));
        k[3] = w[0]; k[2] = w[1]; k[1] = w[2]; k[0] = w[3];
        key16 r;
        r.hi = (((uint64_t)k[0] << 32) | k[1]) & (((uint64_t)m[0] << 32) | m[1]);
        r.lo = (((uint64_t)k[2] << 32) | k[3]) & (((uint64_t)m[2] << 32) | m[3]);
        return r;
    } else if constexpr (REV) {
        /* REV staging: every slot's bytes exist in LDS (padding slots and everything past the input read 0xFF, which sorts
         * them last like the `valid` selects of the forward form did); `a` is clamped by the caller.  k[j] = the dword at
         * offset top - a - 4j */
        (void)valid;
        const uint32_t lo = top - 12u - a;
        const uint32_t *w = reinterpret_cast<const uint32_t *>(by + (lo & ~3u));
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4], sh = lo & 3u;
        k[3] = __builtin_amdgcn_alignbyte(w1, w0, sh);
        k[2] = __builtin_amdgcn_alignbyte(w2, w1, sh);
        k[1] = __builtin_amdgcn_alignbyte(w3, w2, sh);
        k[0] = __builtin_amdgcn_alignbyte(w4, w3, sh);
        key16 r;
        r.hi = (((uint64_t)k[0] << 32) | k[1]) & (((uint64_t)m[0] << 32) | m[1]);
        r.lo = (((uint64_t)k[2] << 32) | k[3]) & (((uint64_t)m[2] << 32) | m[3]);
        return r;
    } else {
    const uint32_t at = valid ? a : 0u;
    if constexpr (BYTES_LDS) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(by + (at & ~3u));
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4], sh = at & 3u;
        k[0] = __builtin_amdgcn_alignbyte(w1, w0, sh);
        k[1] = __builtin_amdgcn_alignbyte(w2, w1, sh);
        k[2] = __builtin_amdgcn_alignbyte(w3, w2, sh);
        k[3] = __builtin_amdgcn_alignbyte(w4, w3, sh);
    } else {
        const uint64_t x = ld64u(by + at), y = ld64u(by + at + 8);
        k[0] = (uint32_t)x; k[1] = (uint32_t)(x >> 32); k[2] = (uint32_t)y; k[3] = (uint32_t)(y >> 32);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) k[i] = valid ? (__builtin_bswap32(k[i]) & m[i]) : 0xFFFFFFFFu;
    key16 r;
    r.hi = ((uint64_t)k[0] << 32) | k[1];
    r.lo = ((uint64_t)k[2] << 32) | k[3];
    return r;
    }
}

/* bytes from .. la-1 of two keys that agree before `from`, 16 at a time (la = 255 and repetitive
 * data: most late compares get here, and every round trip is paid by the whole wavefront) */
template <bool BYTES_LDS, bool REV>
__device__ __forceinline__ bool key_tail_less(const uint8_t *by, uint32_t a, uint32_t b, int la, int from, uint32_t top)
{
    for (int w = from; w < la; w += 16) {
        uint32_t m[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int rem = la - w - 4 * i;
            m[i] = rem >= 4 ? 0xFFFFFFFFu : rem <= 0 ? 0u : 0xFFFFFFFFu << (8 * (4 - rem));
        }
        const key16 ka = load_key16<BYTES_LDS, REV>(by, a + (uint32_t)w, true, m, top), kb = load_key16<BYTES_LDS, REV>(by, b + (uint32_t)w, true, m, top);
        if (ka.hi != kb.hi) return ka.hi < kb.hi;
        if (ka.lo != kb.lo) return ka.lo < kb.lo;
    }
    return a < b;
}

template <bool BYTES_LDS, bool REV = false>
__device__ __forceinline__ bool sort_less16(const uint8_t *by, uint32_t a, const key16 &ka, uint32_t b, const key16 &kb, uint32_t R, int la, uint32_t top = 0)
{
    if (ka.hi != kb.hi) return ka.hi < kb.hi;
    if (ka.lo != kb.lo) return ka.lo < kb.lo;
    if (a >= R || b >= R) return a < b;
    if (la > 16) return key_tail_less<BYTES_LDS, REV>(by, a, b, la, 16, top);
    return a < b;
}

/* v[0..16) = outputs d .. d+15 of the merge of the sorted runs A[0..L) and A[L..2L) */
template <class IdxT, bool BYTES_LDS, bool REV = false, uint32_t CS = 0>
__device__ __forceinline__ void merge16(const IdxT *A, uint32_t L, uint32_t d, const uint8_t *by, uint32_t R, int la,
                                        uint32_t (&v)[16], uint32_t top = 0, uint32_t slots = 0 /* REV: slots with staged bytes (an exhausted run's index is clamped to it) */)
{
    uint32_t m[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int rem = la - 4 * i;
        m[i] = rem >= 4 ? 0xFFFFFFFFu : rem <= 0 ? 0u : 0xFFFFFFFFu << (8 * (4 - rem));
    }
    const IdxT *B = A + L;
    uint32_t lo = d > L ? d - L : 0, hi = d < L ? d : L;
    while (lo < hi) {                                        /* merge path: how many of the first d outputs come from A */
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t a = A[mid], b = B[d - 1 - mid];
        const key16 ka = load_key16<BYTES_LDS, REV, CS>(by, a, a < R, m, top), kb = load_key16<BYTES_LDS, REV, CS>(by, b, b < R, m, top);
        if (sort_less16<BYTES_LDS, REV>(by, a, ka, b, kb, R, la, top)) lo = mid + 1; else hi = mid;
    }
    uint32_t ia = lo, ib = d - lo;
    bool va = ia < L, vb = ib < L;
    uint32_t a = va ? (uint32_t)A[ia] : 0xFFFFFFFFu, b = vb ? (uint32_t)B[ib] : 0xFFFFFFFFu;
    key16 ka = load_key16<BYTES_LDS, REV, CS>(by, REV ? min(a, slots) : a, a < R, m, top), kb = load_key16<BYTES_LDS, REV, CS>(by, REV ? min(b, slots) : b, b < R, m, top);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const bool take_a = !vb || (va && sort_less16<BYTES_LDS, REV>(by, a, ka, b, kb, R, la, top));
        v[r] = take_a ? a : b;
        if (r == 15) break;
        ia += take_a ? 1u : 0u;
        ib += take_a ? 0u : 1u;
        const uint32_t pos = take_a ? ia : L + ib;
        const bool valid = (take_a ? ia : ib) < L;
        const uint32_t nv = valid ? (uint32_t)A[pos] : 0xFFFFFFFFu;
        const key16 nk = load_key16<BYTES_LDS, REV, CS>(by, REV ? min(nv, slots) : nv, nv < R, m, top);
        if (take_a) { a = nv; ka = nk; va = valid; } else { b = nv; kb = nk; vb = valid; }
    }
}

/* Global-memory merge levels of the large regions: out[0..nout) = outputs d .. d+nout-1 of the merge of
 * A[0..L) and A[L..2L).  Every access is an L2 / HBM round trip and a workgroup has a CU to itself, so
 * the merge-path search (log2 L dependent round trips) is amortised over 64 outputs instead of 16. */
__device__ __forceinline__ void merge_run_global(const uint32_t *A, uint32_t L, uint32_t d, const uint8_t *by, uint32_t R, int la,
                                                 uint32_t *out, uint32_t nout)
{
    uint32_t m[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int rem = la - 4 * i;
        m[i] = rem >= 4 ? 0xFFFFFFFFu : rem <= 0 ? 0u : 0xFFFFFFFFu << (8 * (4 - rem));
    }
    const uint32_t *B = A + L;
    uint32_t lo = d > L ? d - L : 0, hi = d < L ? d : L;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t a = A[mid], b = B[d - 1 - mid];
        const key16 ka = load_key16<false>(by, a, a < R, m), kb = load_key16<false>(by, b, b < R, m);
        if (sort_less16<false>(by, a, ka, b, kb, R, la)) lo = mid + 1; else hi = mid;
    }
    uint32_t ia = lo, ib = d - lo;
    bool va = ia < L, vb = ib < L;
    uint32_t a = va ? A[ia] : 0xFFFFFFFFu, b = vb ? B[ib] : 0xFFFFFFFFu;
    key16 ka = load_key16<false>(by, a, a < R, m), kb = load_key16<false>(by, b, b < R, m);
    for (uint32_t g = 0; g < nout; g += 4) {
        uint32_t o[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const bool take_a = !vb || (va && sort_less16<false>(by, a, ka, b, kb, R, la));
            o[r] = take_a ? a : b;
            ia += take_a ? 1u : 0u;
            ib += take_a ? 0u : 1u;
            const uint32_t pos = take_a ? ia : L + ib;
            const bool valid = (take_a ? ia : ib) < L;
            const uint32_t nv = valid ? A[pos] : 0xFFFFFFFFu;
            const key16 nk = load_key16<false>(by, nv, nv < R, m);
            if (take_a) { a = nv; ka = nk; va = valid; } else { b = nv; kb = nk; vb = valid; }
        }
        *reinterpret_cast<uint4 *>(out + g) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

/*
 * Merge sort of CH = 16*1024 slots in LDS (the production sort): a thread sorts its 16 consecutive
 * slots in registers, then log2(CH/16) = 10 merge levels.  In a level every thread produces 16
 * consecutive outputs of the merge of two sorted runs A|B of L slots each: a merge-path binary search
 * on its diagonal (<= log2(L)+1 key compares) finds where its outputs start in A and in B, then 16
 * serial steps take the smaller head.  Outputs stay in registers until every reader of the pair is
 * done (a wavefront barrier while the pair lies inside the 1024 slots one wave owns, __syncthreads
 * above), then overwrite the thread's own 16 slots: in place, no second index buffer in LDS.
 * (key, index) is a strict total order, so the merge path is unique.  O(CH log CH) key compares
 * instead of the bitonic network's O(CH log^2 CH): ~3.5x fewer random LDS reads per region.
 */
template <class IdxT, bool BYTES_LDS, uint32_t NT = MATCH_BLOCK, bool REV = false, uint32_t CS = 0>
__device__ __forceinline__ void region_sort_merge(IdxT *ix, const uint8_t *by, uint32_t R, int la, uint32_t tid,
                                                  uint32_t L_first = 0 /* > 0: ix[] already holds sorted runs of L_first slots */,
                                                  uint32_t top = 0 /* REV: offset of the dword that holds the bytes 0..3 (be32_at) */)
{
    constexpr uint32_t CH = 16 * NT;
    uint32_t v[16];
    if (!L_first) {
        if constexpr (sizeof(IdxT) == 2) {
            const uint64_t pmask = la >= 6 ? ~0xFFFFull : ~0ull << (8 * (8 - la));
            auto prefix = [&](uint32_t a) -> uint64_t {                           /* the key's first six bytes, big-endian, << 16 */
                if constexpr (REV) {
                    /* (padding slots read 0xFF bytes); the dwords at the offsets top - a - 4 and top - a */
                    const uint32_t lo = top - 4u - a;
                    const uint32_t *w = reinterpret_cast<const uint32_t *>(by + (lo & ~3u));
                    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], sh = lo & 3u;
                    return (((uint64_t)__builtin_amdgcn_alignbyte(w2, w1, sh) << 32) | __builtin_amdgcn_alignbyte(w1, w0, sh)) & pmask;
                } else {
                    if (a >= R) return ~0xFFFFull;
                    return (((uint64_t)__builtin_bswap32(ld32_at<BYTES_LDS>(by, a)) << 32) | __builtin_bswap32(ld32_at<BYTES_LDS>(by, a + 4u))) & pmask;
                }
            };
            uint64_t x[16];
            {
                const uint4 a = *reinterpret_cast<const uint4 *>(ix + 16 * tid), b = *reinterpret_cast<const uint4 *>(ix + 16 * tid + 8);
                const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t i = (w[r >> 1] >> (16 * (r & 1))) & 0xFFFFu;
                    x[r] = prefix(i) | i;
                }
            }
            sort_packed_pass<1, true, BYTES_LDS, REV>(x, by, R, la, 2, false, top);
            sort_packed_pass<2, true, BYTES_LDS, REV>(x, by, R, la, 4, false, top);
            sort_packed_pass<1, true, BYTES_LDS, REV>(x, by, R, la, 4, false, top);
            sort_packed_pass<4, true, BYTES_LDS, REV>(x, by, R, la, 8, false, top);
            sort_packed_pass<2, true, BYTES_LDS, REV>(x, by, R, la, 8, false, top);
            sort_packed_pass<1, true, BYTES_LDS, REV>(x, by, R, la, 8, false, top);
            sort_packed_pass<8, false, BYTES_LDS, REV>(x, by, R, la, 0, true, top);
            sort_packed_pass<4, false, BYTES_LDS, REV>(x, by, R, la, 0, true, top);
            sort_packed_pass<2, false, BYTES_LDS, REV>(x, by, R, la, 0, true, top);
            sort_packed_pass<1, false, BYTES_LDS, REV>(x, by, R, la, 0, true, top);
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = (uint32_t)x[r] & 0xFFFFu;
        } else {
            const uint32_t pmask = la >= 4 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * (4 - la));
            auto prefix = [&](uint32_t a) -> uint32_t {
                if constexpr (REV) return be32_at<true, true>(by, top, a) & pmask;       /* (padding slots read 0xFF bytes) */
                else return a < R ? (__builtin_bswap32(ld32_at<BYTES_LDS>(by, a)) & pmask) : 0xFFFFFFFFu;
            };
            uint32_t pf[16];
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const uint4 a = *reinterpret_cast<const uint4 *>(ix + 16 * tid + r);
                v[r] = a.x; v[r + 1] = a.y; v[r + 2] = a.z; v[r + 3] = a.w;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) pf[r] = prefix(v[r]);
            sort_local_pass<1, true, BYTES_LDS, REV>(v, pf, by, R, la, 2, false, top);
            sort_local_pass<2, true, BYTES_LDS, REV>(v, pf, by, R, la, 4, false, top);
            sort_local_pass<1, true, BYTES_LDS, REV>(v, pf, by, R, la, 4, false, top);
            sort_local_pass<4, true, BYTES_LDS, REV>(v, pf, by, R, la, 8, false, top);
            sort_local_pass<2, true, BYTES_LDS, REV>(v, pf, by, R, la, 8, false, top);
            sort_local_pass<1, true, BYTES_LDS, REV>(v, pf, by, R, la, 8, false, top);
            sort_local_pass<8, false, BYTES_LDS, REV>(v, pf, by, R, la, 0, true, top);
            sort_local_pass<4, false, BYTES_LDS, REV>(v, pf, by, R, la, 0, true, top);
            sort_local_pass<2, false, BYTES_LDS, REV>(v, pf, by, R, la, 0, true, top);
            sort_local_pass<1, false, BYTES_LDS, REV>(v, pf, by, R, la, 0, true, top);
        }
    }
    auto store_mine = [&]() {
        if constexpr (sizeof(IdxT) == 2) {
            uint32_t w[8];
#pragma unroll
            for (int r = 0; r < 8; r++) w[r] = v[2 * r] | (v[2 * r + 1] << 16);
            *reinterpret_cast<uint4 *>(ix + 16 * tid) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4 *>(ix + 16 * tid + 8) = make_uint4(w[4], w[5], w[6], w[7]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; r += 4)
                *reinterpret_cast<uint4 *>(ix + 16 * tid + r) = make_uint4(v[r], v[r + 1], v[r + 2], v[r + 3]);
        }
    };
    auto barrier = [&](bool wide) {
        if (wide) __syncthreads();
        else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    };
    if (!L_first) store_mine();
    for (uint32_t L = L_first ? L_first : 16; L < CH; L <<= 1) {
        const bool wide = 2 * L > 1024;
        barrier(wide);
        const uint32_t o0 = 16 * tid, base = o0 & ~(2 * L - 1);
        merge16<IdxT, BYTES_LDS, REV, CS>(ix + base, L, o0 - base, by, R, la, v, top, CH - 1u);
        barrier(wide);
        store_mine();
    }
    __syncthreads();
}

/* ------------------------------------------------------------------ k_match ---------- */

/* Small windows whose tile is three quarters of the region (sb 4089..4096: RP 16384 = 4 x 4096, TILE = 3 x 4096): the
 * regions are unions of globally aligned 4 K chunks and overlap by one, and (key, position) is one total order -- so
 * every chunk is sorted ONCE (a 256-thread workgroup: the same register sort + merge levels as a region, eight levels
 * instead of ten) and a region only runs the last two merge levels over its four chunks: 12 of the 14 levels on 1x
 * instead of 4/3x the data. */
#define C1_CH 4096u
#define C1_BLOCK 256
#define C1_CS (C1_CH + 256 + 32)                     /* bytes between the four copies of a chunk's staged bytes */

__global__ __launch_bounds__(C1_BLOCK) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(6, 6))) void k_c1_chunks(const uint8_t *__restrict__ in, uint32_t n, int la, uint64_t pos0,
                                                       uint16_t *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint16_t ix[C1_CH];
    __shared__ __attribute__((aligned(16))) uint8_t lby[4 * C1_CS];    /* the REV staging and its three shifted copies (load_key16) */
    const uint32_t tid = threadIdx.x;
    const uint64_t base = pos0 + (uint64_t)blockIdx.x * C1_CH;
    const uint32_t Rl = base >= n ? 0u : (n - (uint32_t)base < C1_CH ? n - (uint32_t)base : C1_CH);
    constexpr uint32_t NB = C1_CH + 256 + 24;             /* every slot's key: 4096 positions + la <= 255 + the key loads' slack */
    if (Rl) {
        stage_rev<C1_BLOCK>(lby, NB, in, base, n, tid);
#pragma unroll
        for (uint32_t sft = 1; sft < 4; sft++) stage_rev_shifted<C1_BLOCK>(lby + sft * C1_CS, sft, NB, in, base, n, tid);
    }
    for (uint32_t i = tid; i < C1_CH; i += C1_BLOCK) ix[i] = (uint16_t)i;
    __syncthreads();
    if (Rl) region_sort_merge<uint16_t, true, C1_BLOCK, true, C1_CS>(ix, lby, Rl, la, tid, 0, NB - 4u);
    uint16_t *o = out + (size_t)blockIdx.x * C1_CH;
    for (uint32_t e = tid * 8; e < C1_CH; e += C1_BLOCK * 8) *reinterpret_cast<uint4 *>(o + e) = *reinterpret_cast<const uint4 *>(ix + e);
}

static bool c1_shared_sort(const lz77x_geom &g)
{
    return g.fast && g.shifted && g.RP == 16u * MATCH_BLOCK && g.SBu * 4u == g.RP && !LZ77X_VENV("LZ77X_C1_SORT_V1") &&
           !(LZ77X_VENV("LZ77X_SORT_VARIANT") && atoi(LZ77X_VENV("LZ77X_SORT_VARIANT")));
}

template <bool FAST> struct rank_traits;
template <> struct rank_traits<true>  { typedef uint16_t rank_t; static constexpr uint32_t HALF = 0x8000u; static constexpr uint32_t MASK = 0xFFFFu; };
template <> struct rank_traits<false> { typedef uint32_t rank_t; static constexpr uint32_t HALF = 0x80000000u; static constexpr uint32_t MASK = 0xFFFFFFFFu; };

/* Running min / max of rank differences u = rank[y]-rank[x] (mod 2^16 or 2^32) for the eight
 * consecutive positions ("octet") a thread owns.  With all ranks < HALF, positive differences
 * are < HALF and negative ones wrap to >= HALF, so
 *     min u  = distance to the in-order successor   (valid iff < HALF)
 *     max u  = -distance to the in-order predecessor (valid iff >= HALF)
 * which turns the BST neighbour search into sub/min/max: 1.5 VALU ops per pair on packed
 * 16-bit lanes (v_pk_sub_i16 / v_pk_min_u16 / v_pk_max_u16), no cross-lane traffic. */
template <bool FAST> struct acc8;

template <> struct acc8<true> {
    us2 mn[8], mx[8], xr[8];
    __device__ __forceinline__ void init(const uint16_t *rk, uint32_t lx0)
    {
        const uint4 v = *reinterpret_cast<const uint4 *>(rk + lx0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint16_t r = (uint16_t)(w[i >> 1] >> (16 * (i & 1)));
            xr[i] = (us2){r, r};
        }
        reset();
    }
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int i = 0; i < 8; i++) { mn[i] = (us2){0xFFFF, 0xFFFF}; mx[i] = (us2){0, 0}; }
    }
    /* all 64 pairs of the candidate octet v with the eight owned positions */
    __device__ __forceinline__ void octet(const uint4 v)
    {
        const us2 A = __builtin_bit_cast(us2, v.x), B = __builtin_bit_cast(us2, v.y);
        const us2 C = __builtin_bit_cast(us2, v.z), D = __builtin_bit_cast(us2, v.w);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const us2 dA = A - xr[i], dB = B - xr[i], dC = C - xr[i], dD = D - xr[i];
            mn[i] = __builtin_elementwise_min(__builtin_elementwise_min(mn[i], dA), __builtin_elementwise_min(dB, __builtin_elementwise_min(dC, dD)));
            mx[i] = __builtin_elementwise_max(__builtin_elementwise_max(mx[i], dA), __builtin_elementwise_max(dB, __builtin_elementwise_max(dC, dD)));
        }
    }
    __device__ __forceinline__ static uint4 load(const uint16_t *rk, uint32_t M) { return *reinterpret_cast<const uint4 *>(rk + 8 * M); }
    __device__ __forceinline__ void one(int i, uint32_t ry)
    {
        const uint16_t u = (uint16_t)(ry - xr[i].x);
        mn[i].x = u < mn[i].x ? u : mn[i].x;
        mx[i].x = u > mx[i].x ? u : mx[i].x;
    }
    __device__ __forceinline__ uint32_t minu(int i) const { return mn[i].x < mn[i].y ? mn[i].x : mn[i].y; }
    __device__ __forceinline__ uint32_t maxu(int i) const { return mx[i].x > mx[i].y ? mx[i].x : mx[i].y; }
    __device__ __forceinline__ uint32_t xrank(int i) const { return xr[i].x; }
};

template <> struct acc8<false> {
    uint32_t mn[8], mx[8], xr[8];
    struct oct { uint4 a, b; };
    __device__ __forceinline__ void init(const uint32_t *rk, uint32_t lx0)
    {
        const uint4 a = *reinterpret_cast<const uint4 *>(rk + lx0), b = *reinterpret_cast<const uint4 *>(rk + lx0 + 4);
        xr[0] = a.x; xr[1] = a.y; xr[2] = a.z; xr[3] = a.w; xr[4] = b.x; xr[5] = b.y; xr[6] = b.z; xr[7] = b.w;
        reset();
    }
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int i = 0; i < 8; i++) { mn[i] = 0xFFFFFFFFu; mx[i] = 0; }
    }
    __device__ __forceinline__ void octet(const oct v)
    {
        const uint32_t y[8] = {v.a.x, v.a.y, v.a.z, v.a.w, v.b.x, v.b.y, v.b.z, v.b.w};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t lo = mn[i], hi = mx[i];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const uint32_t d0 = y[j] - xr[i], d1 = y[j + 1] - xr[i];
                lo = min(lo, min(d0, d1));      /* v_min3_u32 */
                hi = max(hi, max(d0, d1));      /* v_max3_u32 */
            }
            mn[i] = lo;
            mx[i] = hi;
        }
    }
    __device__ __forceinline__ static oct load(const uint32_t *rk, uint32_t M)
    {
        oct o;
        o.a = *reinterpret_cast<const uint4 *>(rk + 8 * M);
        o.b = *reinterpret_cast<const uint4 *>(rk + 8 * M + 4);
        return o;
    }
    __device__ __forceinline__ void one(int i, uint32_t ry)
    {
        const uint32_t u = ry - xr[i];
        mn[i] = min(mn[i], u);
        mx[i] = max(mx[i], u);
    }
    __device__ __forceinline__ uint32_t minu(int i) const { return mn[i]; }
    __device__ __forceinline__ uint32_t maxu(int i) const { return mx[i]; }
    __device__ __forceinline__ uint32_t xrank(int i) const { return xr[i]; }
};

/* masked path for the few octets that straddle a window edge: y = 8M+j against x = lx0+i,
 * pair kept iff lo <= (fwd ? y-x : x-y) <= hi */
template <bool FAST, bool FWD, class RankT>
__device__ __forceinline__ void edge_octet(acc8<FAST> &acc, const RankT *rk, uint32_t M, uint32_t lx0, uint32_t R, int lo, int hi)
{
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t y = 8 * M + j;
        if (y >= R) continue;
        const uint32_t ry = rk[y];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t d = FWD ? (int32_t)y - (int32_t)(lx0 + i) : (int32_t)(lx0 + i) - (int32_t)y;
            if (d >= lo && d <= hi) acc.one(i, ry);
        }
    }
}

/*
 * One workgroup per region.
 *   FAST : ranks/index are uint16 in LDS and the window bytes are staged in LDS (RP <= 16384)
 *          else they are uint32 in a per-region global scratch and bytes come from L1/L2.
 *   MODE : 0 = production (unmasked octets inside the window, masked octets on its edges)
 *          1 = every octet through the masked path (slow; on-device self check)
 *          2 = sort + ranks only (timing probe)
 */
template <bool FAST, int MODE>
__global__ __launch_bounds__(MATCH_BLOCK, (FAST && MODE == 3) ? 8 : 1) void k_match(const uint8_t *__restrict__ in, uint32_t n, int sb, int la,
                                                       uint32_t SBu, uint32_t RP, uint32_t TILE, uint32_t region0,
                                                       uint32_t *__restrict__ ps, uint8_t *__restrict__ maxlen,
                                                       uint32_t *__restrict__ scratch, int sort_variant, uint32_t walk_run,
                                                       uint16_t *__restrict__ order_all /* FAST production: RP uint16 per region of the input
                                                                                           (sorted order -> position - t0), kept for the tie-break */,
                                                       const uint16_t *__restrict__ chunks /* FAST production, tile = 3/4 region: the sorted orders of
                                                                                              the launch's 4 K chunks (k_c1_chunks), else null */)
{
#ifndef LZ77X_VARIANTS
    sort_variant = 0;                                    /* the product build has the merge sorts only (the bitonic networks and the
                                                            timing ablations below are dead code there) */
#endif
    typedef typename rank_traits<FAST>::rank_t rank_t;
    constexpr uint32_t HALF = rank_traits<FAST>::HALF;
    constexpr uint32_t RMASK = rank_traits<FAST>::MASK;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const uint32_t tid = threadIdx.x;
    const uint32_t region = region0 + blockIdx.x;
    const uint64_t t0_64 = (uint64_t)region * TILE;
    if (t0_64 >= n) return;
    const uint32_t t0 = (uint32_t)t0_64;
    /* MODE >= 2 (production sort, probe): the region is [t0, t0+TILE+sb); pair scans: halos either side
     * of [t0, t0+TILE) (the two layouts of lz77x_geom) */
    constexpr bool SHIFTED = MODE >= 2;
    const uint32_t rstart = SHIFTED ? t0 : (t0 >= SBu ? t0 - SBu : 0);
    const uint64_t rend64 = (uint64_t)t0 + TILE + (uint32_t)sb - (SHIFTED ? 0u : 1u);
    const uint32_t rend = rend64 < n ? (uint32_t)rend64 : n;
    const uint32_t R = rend - rstart;                      /* valid local indices [0,R) */
    const uint32_t lt0 = t0 - rstart;                      /* multiple of 8 */
    const uint32_t lt1 = ((uint64_t)t0 + TILE < n ? t0 + TILE : n) - rstart;

    rank_t *rk, *ix;
    const uint8_t *by;
    if constexpr (FAST) {
        /* LDS: ix[RP] | union { window bytes (sort phase), rk[RP+8] (scan phase) }.  The bytes are
         * only needed by the key compares of the sort; sharing their space with the ranks keeps
         * a C1 region at 64 KiB so two workgroups fit one CU. */
        ix = reinterpret_cast<rank_t *>(smem);             /* RP */
        rk = ix + RP;                                      /* RP + 8, written after the sort */
        uint8_t *stage = reinterpret_cast<uint8_t *>(rk);  /* R + la + 11 <= 2*RP + 16 bytes */
        /* shared chunks: the order holds every position < n of the RP slots (the one past TILE + sb is sorted in its
         * chunk like any other), so the merge levels must see its key too */
        const uint32_t Rs = chunks ? (n - rstart < RP ? n - rstart : RP) : R;
        if (chunks) {
            /* the merge levels over the shared chunks read their keys from the REV staging (be32_at): every slot's bytes,
             * RP + la + slack.  (Only here: a slot beyond the region's positions must sort last, and in this layout such a
             * slot lies beyond the input, where REV reads 0xFF; a region that sorts all its slots itself has padding slots
             * INSIDE the input and keeps the forward staging with its validity selects.) */
            stage_rev<MATCH_BLOCK>(stage, RP + 288u, in, rstart, n, tid);
        } else {
            const uint32_t nb = (Rs + (uint32_t)la + 8 + 3) & ~3u;
            for (uint32_t i = tid * 4; i < nb; i += MATCH_BLOCK * 4)
                *reinterpret_cast<uint32_t *>(stage + i) = *reinterpret_cast<const uint32_t *>(in + rstart + i);
        }
        by = stage;
    } else {
        rk = reinterpret_cast<rank_t *>(scratch + (size_t)blockIdx.x * (2 * (size_t)RP + 8));
        ix = rk + RP + 8;
        by = in + rstart;
    }
    if (FAST && chunks) {
        if constexpr (FAST) {
            /* the region = four consecutive 4 K chunks of the launch, each already in key order (k_c1_chunks): the last
             * two merge levels are all that is left */
            const uint16_t *src = chunks + (size_t)blockIdx.x * 3u * C1_CH;
            for (uint32_t e = tid * 8; e < RP; e += MATCH_BLOCK * 8) {
                const uint4 v = *reinterpret_cast<const uint4 *>(src + e);
                const uint32_t add = (e / C1_CH) * C1_CH * 0x10001u;
                *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(ix) + e) = make_uint4(v.x + add, v.y + add, v.z + add, v.w + add);
            }
        }
    } else
    for (uint32_t i = tid; i < RP; i += MATCH_BLOCK) ix[i] = (rank_t)i;
    __syncthreads();

    /* ---- bitonic sort of local indices by (key, index); indices >= R sort last ---- */
    if (FAST && chunks) {
        if constexpr (FAST) {
            const uint32_t Rs = n - rstart < RP ? n - rstart : RP;
            region_sort_merge<uint16_t, true, MATCH_BLOCK, true>(reinterpret_cast<uint16_t *>(ix), by, Rs, la, tid, C1_CH, RP + 284u);
        }
    } else if (FAST && RP == 16 * MATCH_BLOCK && sort_variant == 0) {
        if constexpr (FAST) region_sort_merge<uint16_t, true>(reinterpret_cast<uint16_t *>(ix), by, R, la, tid);
    } else if (FAST && RP == 16 * MATCH_BLOCK && sort_variant == 2) {
        if constexpr (FAST) region_sort_blocked<uint16_t, true>(reinterpret_cast<uint16_t *>(ix), by, R, la, tid, false, true);
    } else if (!FAST && RP > 16 * MATCH_BLOCK && (sort_variant & 15) == 0) {
        if constexpr (!FAST) {
            /* large region: merge-sort RP/CH chunks in LDS, then merge levels L = CH, 2CH, ... between
             * the two global index arrays of the region (the rank array is free until the sort ends) */
            constexpr uint32_t CH = 16 * MATCH_BLOCK;
            uint32_t *lix = reinterpret_cast<uint32_t *>(smem);          /* CH words */
            uint32_t *src = reinterpret_cast<uint32_t *>(ix), *dst = reinterpret_cast<uint32_t *>(rk);
            /* a chunk is CH CONSECUTIVE positions: its CH + la key bytes fit LDS even though the region's
             * do not, so the chunk sorts compare out of LDS exactly like the small-window path */
            uint8_t *lby = smem + CH * sizeof(uint32_t);
            for (uint32_t c = 0; c < RP / CH; c++) {
                const uint32_t cb = c * CH;
                const uint32_t Rl = R > cb ? (R - cb < CH ? R - cb : CH) : 0u;       /* valid slots of this chunk */
                const uint32_t nb = Rl ? (Rl + (uint32_t)la + 24 + 3) & ~3u : 0u;
                for (uint32_t i = tid * 4; i < nb; i += MATCH_BLOCK * 4)
                    *reinterpret_cast<uint32_t *>(lby + i) = ld32u(by + cb + i);
                for (uint32_t i = tid; i < CH; i += MATCH_BLOCK) lix[i] = i;
                __syncthreads();
                if (Rl && !(sort_variant & 32)) region_sort_merge<uint32_t, true>(lix, lby, Rl, la, tid);   /* bit 5: timing ablation */
                for (uint32_t i = tid * 4; i < CH; i += MATCH_BLOCK * 4) {                /* slots >= Rl stay >= R */
                    const uint4 v = *reinterpret_cast<const uint4 *>(lix + i);
                    *reinterpret_cast<uint4 *>(src + cb + i) = make_uint4(v.x + cb, v.y + cb, v.z + cb, v.w + cb);
                }
                __syncthreads();
            }
            for (uint32_t L = CH; L < RP && !(sort_variant & 16); L <<= 1) {                 /* bit 4: timing ablation */
                for (uint32_t o0 = 64 * tid; o0 < RP; o0 += 64 * MATCH_BLOCK) {
                    const uint32_t base = o0 & ~(2 * L - 1);
                    merge_run_global(src + base, L, o0 - base, by, R, la, dst + o0, 64);
                }
                __syncthreads();
                uint32_t *t = src; src = dst; dst = t;
            }
            if (src != reinterpret_cast<uint32_t *>(ix)) {
                for (uint32_t i = tid * 4; i < RP; i += MATCH_BLOCK * 4)
                    *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(src + i);
                __syncthreads();
            }
        }
    } else if (!FAST && RP > 16 * MATCH_BLOCK && sort_variant == 2) {
        if constexpr (!FAST) {
            /* large region: bitonic network over RP/CH chunks; strides below CH run in LDS chunk by
             * chunk (blocked routine above), strides >= CH directly on the global index array */
            constexpr uint32_t CH = 16 * MATCH_BLOCK;
            uint32_t *lix = reinterpret_cast<uint32_t *>(smem);          /* CH words */
            uint32_t *gix = reinterpret_cast<uint32_t *>(ix);
            const uint32_t NC = RP / CH;
            for (uint32_t c = 0; c < NC; c++) {
                for (uint32_t i = tid; i < CH; i += MATCH_BLOCK) lix[i] = c * CH + i;
                __syncthreads();
                region_sort_blocked<uint32_t, false>(lix, by, R, la, tid, false, (c & 1) == 0);
                for (uint32_t i = tid * 4; i < CH; i += MATCH_BLOCK * 4)
                    *reinterpret_cast<uint4 *>(gix + c * CH + i) = *reinterpret_cast<const uint4 *>(lix + i);
                __syncthreads();
            }
            for (uint32_t k = 2 * CH; k <= RP; k <<= 1) {
                for (uint32_t j = k >> 1; j >= CH; j >>= 1) {
                    for (uint32_t t = tid; t < (RP >> 1); t += MATCH_BLOCK) {
                        const uint32_t i = 2 * t - (t & (j - 1)), l = i + j;
                        const uint32_t a = gix[i], b = gix[l];
                        const bool up = (i & k) == 0;
                        bool b_lt_a;
                        if (a >= R || b >= R) b_lt_a = b < a;
                        else b_lt_a = key_less<false>(by, b, a, la);
                        if (up ? b_lt_a : !b_lt_a) { gix[i] = b; gix[l] = a; }
                    }
                    __syncthreads();
                }
                for (uint32_t c = 0; c < NC; c++) {
                    for (uint32_t i = tid * 4; i < CH; i += MATCH_BLOCK * 4)
                        *reinterpret_cast<uint4 *>(lix + i) = *reinterpret_cast<const uint4 *>(gix + c * CH + i);
                    __syncthreads();
                    region_sort_blocked<uint32_t, false>(lix, by, R, la, tid, true, ((c * CH) & k) == 0);
                    for (uint32_t i = tid * 4; i < CH; i += MATCH_BLOCK * 4)
                        *reinterpret_cast<uint4 *>(gix + c * CH + i) = *reinterpret_cast<const uint4 *>(lix + i);
                    __syncthreads();
                }
            }
        }
    } else {
        for (uint32_t k = 2; k <= RP; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = tid; t < (RP >> 1); t += MATCH_BLOCK) {
                    const uint32_t i = 2 * t - (t & (j - 1));
                    const uint32_t l = i + j;
                    const uint32_t a = ix[i], b = ix[l];
                    const bool up = (i & k) == 0;
                    bool b_lt_a, a_lt_b;
                    if (a >= R || b >= R) { b_lt_a = b < a; a_lt_b = a < b; }
                    else { b_lt_a = key_less<FAST>(by, b, a, la); a_lt_b = !b_lt_a; }
                    if (up ? b_lt_a : a_lt_b) { ix[i] = (rank_t)b; ix[l] = (rank_t)a; }
                }
                __syncthreads();
            }
        }
    }
    if constexpr (FAST && MODE == 3) {
        /* Hand the order to the window walkers, run by run.  A walker only ever touches the
         * walk_run + sb consecutive positions [lo, lo+span) of its run, so it gets their ranks AMONG
         * THEMSELVES ("sub-ranks": a prefix count over the sorted order) and the inverse: its bitmap
         * shrinks from RP bits to walk_run+SBu bits (three wavefronts of walkers per CU instead of
         * one) and is two-thirds full, so it needs no summary level.  Staged in the LDS the key bytes
         * occupied, copied out 16 bytes at a time.  The barriers order LDS traffic only (lds_barrier): a
         * __syncthreads() would also wait for the previous run's 24 KB of global stores to complete, 18 times
         * per region. */
        __shared__ uint32_t wsum[MATCH_BLOCK / 64];
        const uint32_t K = RP / MATCH_BLOCK;                 /* sorted slots per thread: 4, 8 or 16 */
        const uint32_t lane = tid & 63, wave = tid >> 6;
        const uint32_t steps = n - t0 < TILE ? n - t0 : TILE;
        const uint32_t NR = (TILE + walk_run - 1) / walk_run, SUB = walk_run + SBu;
        uint16_t *st_rk = reinterpret_cast<uint16_t *>(rk);                              /* SUB <= RP */
        uint16_t *gout = reinterpret_cast<uint16_t *>(scratch) + (size_t)blockIdx.x * NR * SUB;
        uint32_t mine[16];
#pragma unroll
        for (int q = 0; q < 16; q++) mine[q] = (uint32_t)q < K ? (uint32_t)ix[tid * K + q] : 0xFFFFFFFFu;
        if (order_all) {
            /* the region's sorted order itself stays resident: the tie-break enumerates equal-length candidates
             * as runs of it (k_tokens_sorted) */
            uint16_t *go = order_all + (size_t)region * RP;
            for (uint32_t e = tid * 8; e < RP; e += MATCH_BLOCK * 8)
                *reinterpret_cast<uint4 *>(go + e) = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(ix) + e);
        }
        lds_barrier();                                       /* the sort's last readers of the byte area are done */
        for (uint32_t j = 0; j < NR; j++) {
            const uint32_t lo = j * walk_run;
            if (lo >= steps) break;
            const uint32_t span = R - lo < SUB ? R - lo : SUB;
            uint32_t cnt = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) cnt += (mine[q] - lo < span) ? 1u : 0u;
            uint32_t incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if (lane >= (uint32_t)d) incl += t;
            }
            if (lane == 63) wsum[wave] = incl;
            lds_barrier();
            uint32_t run = incl - cnt;
            for (uint32_t w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const uint32_t i = mine[q] - lo;
                if (i < span) { st_rk[i] = (uint16_t)run; run++; }     /* (the inverse is a scatter of this array: k_walk_final_lds) */
            }
            lds_barrier();
            uint16_t *g = gout + (size_t)j * SUB;
            for (uint32_t e = tid * 8; e < SUB; e += MATCH_BLOCK * 8)
                *reinterpret_cast<uint4 *>(g + e) = *reinterpret_cast<const uint4 *>(st_rk + e);
            lds_barrier();
        }
        return;
    }
    /* FAST: the staged bytes are dead from here on, their space becomes the ranks (the few remaining
     * byte reads -- LCPs -- go to L1/L2); generic: the rank array was the sort's second buffer */
    for (uint32_t i = tid; i < RP + 8; i += MATCH_BLOCK) rk[i] = 0;
    __syncthreads();
    by = in + rstart;
    for (uint32_t r = tid; r < RP; r += MATCH_BLOCK) {
        const uint32_t a = ix[r];
        if (a < R) rk[a] = (rank_t)r;
    }
    __syncthreads();
    if (MODE == 2 || (!FAST && MODE == 3)) return;           /* sort + ranks only (generic path: they live in scratch) */

    /* ---- pair scan: one octet of positions per thread, window streamed in octets ---- */
    const uint32_t usb = (uint32_t)sb;
    const uint32_t Mlast = (R - 1) >> 3;                     /* last octet holding a valid index */
    for (uint32_t lx0 = lt0 + 8 * tid; lx0 < lt1; lx0 += 8 * MATCH_BLOCK) {
        const uint32_t G = lx0 >> 3;
        acc8<FAST> acc;
        acc.init(rk, lx0);

        /* forward window y in [x+1, x+sb-1], y < R : in-order neighbours at eviction time.
         * Octets G+1 .. G+(sb-8)/8 lie inside every owned position's window. */
        {
            const uint32_t mend = min((lx0 + 7 + usb - 1) >> 3, Mlast);
            uint32_t ilo = G + 1, ihi = G;                   /* empty */
            if (MODE == 0 && usb >= 16 && R >= 8) ihi = min(G + (usb - 8) / 8, (R - 8) >> 3);
            edge_octet<FAST, true>(acc, rk, G, lx0, R, 1, sb - 1);
            uint32_t M = ilo;
            for (; M + 1 <= ihi; M += 2) {
                const auto v0 = acc8<FAST>::load(rk, M), v1 = acc8<FAST>::load(rk, M + 1);
                acc.octet(v0);
                acc.octet(v1);
            }
            if (M <= ihi) { acc.octet(acc8<FAST>::load(rk, M)); M++; }
            for (; M <= mend; M++) edge_octet<FAST, true>(acc, rk, M, lx0, R, 1, sb - 1);
        }
        uint32_t psv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t x = lx0 + i;
            uint32_t P = 0, S = 0;
            if (x < lt1 && (uint64_t)rstart + x + usb < n) {       /* only evicted positions matter */
                const uint32_t mnu = acc.minu(i), mxu = acc.maxu(i), xr = acc.xrank(i);
                if (mnu < HALF) S = (uint32_t)ix[(xr + mnu) & RMASK] - x;
                if (mxu >= HALF) P = (uint32_t)ix[(xr + mxu) & RMASK] - x;
            }
            psv[i] = P | (S << 16);
        }

        /* backward window c in [x-sb, x-1], c >= 0 : longest match (tree.c:118-152 length).
         * Octets G-(sb-7)/8 .. G-1 lie inside every owned position's window. */
        acc.reset();
        {
            const uint32_t mfirst = lx0 >= usb ? (lx0 - usb) >> 3 : 0;
            uint32_t ilo = G, ihi = G;                       /* ihi exclusive here; empty */
            if (MODE == 0 && usb >= 15 && G >= 1) {
                const uint32_t span = (usb - 7) / 8;
                ilo = G > span ? G - span : 0;
            }
            uint32_t M = mfirst;
            for (; M < ilo; M++) edge_octet<FAST, false>(acc, rk, M, lx0, R, 1, sb);
            for (; M + 1 < ihi; M += 2) {
                const auto v0 = acc8<FAST>::load(rk, M), v1 = acc8<FAST>::load(rk, M + 1);
                acc.octet(v0);
                acc.octet(v1);
            }
            if (M < ihi) { acc.octet(acc8<FAST>::load(rk, M)); M++; }
            edge_octet<FAST, false>(acc, rk, G, lx0, R, 1, sb);
        }
        uint32_t mlv[2] = {0, 0};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t x = lx0 + i;
            uint32_t best = 0;
            if (x < lt1) {
                const uint32_t left = n - (rstart + x);
                const int cap = (int)(left < (uint32_t)la ? left : (uint32_t)la) - 1;
                const uint32_t mnu = acc.minu(i), mxu = acc.maxu(i), xr = acc.xrank(i);
                if (mnu < HALF) best = (uint32_t)lcp_capped<false>(by, (uint32_t)ix[(xr + mnu) & RMASK], x, cap);
                if (mxu >= HALF) {
                    const uint32_t l2 = (uint32_t)lcp_capped<false>(by, (uint32_t)ix[(xr + mxu) & RMASK], x, cap);
                    best = l2 > best ? l2 : best;
                }
            }
            mlv[i >> 2] |= best << (8 * (i & 3));
        }

        const uint32_t xa = rstart + lx0;
        if (lx0 + 8 <= lt1) {
            *reinterpret_cast<uint4 *>(ps + xa) = make_uint4(psv[0], psv[1], psv[2], psv[3]);
            *reinterpret_cast<uint4 *>(ps + xa + 4) = make_uint4(psv[4], psv[5], psv[6], psv[7]);
            *reinterpret_cast<uint2 *>(maxlen + xa) = make_uint2(mlv[0], mlv[1]);
        } else {
            for (int i = 0; i < 8 && lx0 + i < lt1; i++) {
                ps[xa + i] = psv[i];
                maxlen[xa + i] = (uint8_t)(mlv[i >> 2] >> (8 * (i & 3)));
            }
        }
    }
}

/* ------------------------------------------------------------------ window walkers --- */

/*
 * Sliding-window neighbour search in O(1) per position (replaces the O(SB) pair scan).
 *
 * A walker is ONE LANE.  It owns a bitmap over the region's rank space (bit r set <=> the
 * position with rank r is inside the current window) plus a one-bit-per-word summary, and slides
 * the window W_t = [t, t+sb) one position at a time.  Step t:
 *     set   rank[t+sb-1]                       -> bitmap = W_t
 *     query rank[t+sb]  (not in the set)       -> backward result of y = t+sb: its in-order
 *                                                 neighbours in [y-sb, y-1], the two candidates of the
 *                                                 longest match (tree.c:118-152)
 *     clear rank[t]
 *     query rank[t]                            -> forward result of x = t: its in-order neighbours among
 *                                                 the sb-1 positions after it, i.e. what the BST holds
 *                                                 when x is evicted (tree.c:182)
 * A query is "first set bit above / below a rank": the in-order successor / predecessor.  64 walkers
 * share a wavefront; their bitmaps are word-interleaved in LDS (word w of lane l at (w*64+l)*4) so
 * every access of the wave hits 64 distinct banks.  A walker covers run_len steps of one region after
 * filling its first window (sb-1 sets); the very first walker of the input answers the backward
 * queries of y < sb while it fills.
 */
#define WALK_NONE 0xFFFFu

typedef __attribute__((address_space(1))) uint32_t g_u32;
typedef __attribute__((address_space(1))) uint64_t g_u64;
struct u32x2 { uint32_t x, y; };
typedef __attribute__((address_space(1))) u32x2 g_uint2;

__global__ __launch_bounds__(64) void k_walk(const uint16_t *__restrict__ subs, uint32_t n, int sb, uint32_t SBu,
                                             uint32_t TILE, uint32_t region0, uint32_t nregions, uint32_t run_len,
                                             uint32_t runs_per_tile, uint32_t *__restrict__ wf, uint32_t *__restrict__ wb,
                                             uint32_t *__restrict__ wb0, int head_block)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t bm[];
    const uint32_t lane = threadIdx.x;
    const uint32_t SUB = run_len + SBu, NW = (SUB + 31) >> 5;
#define BM_WORD(w) bm[(w) * 64u + lane]
    /* head block (the last one of a launch that starts at region 0): its 64 lanes share the backward queries of the
     * first sb positions of the input, which look back at [0, y) only.  The first walker used to answer them while it
     * filled its first window -- 4095 query + store steps on ONE lane that every other wavefront of the launch
     * finished long before (1.5 of the kernel's 2.1 ms per 100 MB) */
    const bool head = head_block && blockIdx.x == gridDim.x - 1;
    const uint32_t id = head ? 0u : blockIdx.x * 64u + lane;
    const uint32_t run = id % runs_per_tile;
    const uint32_t reg = id / runs_per_tile;
    if (reg >= nregions) return;
    const uint64_t t0_64 = (uint64_t)(region0 + reg) * TILE;
    if (t0_64 >= n) return;
    const uint32_t t0 = (uint32_t)t0_64;
    const uint32_t usb = (uint32_t)sb;
    const uint64_t rend64 = t0_64 + TILE + usb;
    const uint32_t Rreg = (rend64 < n ? (uint32_t)rend64 : n) - t0;      /* sorted local indices of the region */
    const uint32_t lt1 = n - t0 < TILE ? n - t0 : TILE;                  /* steps of the region */
    const uint32_t lo = run * run_len;
    if (lo >= lt1) return;
    const uint32_t tb = min(run_len, lt1 - lo);                          /* steps t in [0, tb), relative to lo */
    const uint32_t R = min(SUB, Rreg - lo);                              /* sub-ranked positions [0, R) of this run */
    const uint16_t *rk = subs + ((size_t)reg * runs_per_tile + run) * SUB;
    uint32_t *of = wf + (size_t)reg * TILE + lo, *ob = wb + (size_t)reg * TILE + lo;   /* indexed by t */

    /* a summary level -- bit w of it: word w of the bitmap is not empty -- behind the bitmap, word-interleaved like it
     * (round 4).  The window's ranks are dense on text (two thirds of the bitmap: a neighbour lies in the three words at
     * hand), but on low-entropy data they come in long stretches -- equal keys order by position, so what has left the
     * window is a prefix of every key's stretch -- and the word-by-word scans for a far neighbour made the walkers 4.6x
     * slower there (3.3 ms against 0.7 per 100 MB).  With the summary a far neighbour is two or three dependent reads
     * wherever it lies, and a query that HAS no neighbour on one side (every step of a run of equal bytes) finds that out
     * from at most NS words. */
    const uint32_t NS = (NW + 31u) >> 5;
#define SUM_WORD(k) bm[(NW + (k)) * 64u + lane]
    for (uint32_t w = 0; w < NW + NS; w++) bm[w * 64u + lane] = 0;
    /* no set bit lies in a word below lo_w or above hi_w: a query that HAS no neighbour on one side (every step of a run of
     * equal bytes: the window is the top or the bottom of the order) answers from the bounds.  Sets widen them, a scan
     * that finds nothing tightens them. */
    uint32_t lo_w = NW, hi_w = 0;
    auto set_bit = [&](uint32_t r) {
        atomicOr(&BM_WORD(r >> 5), 1u << (r & 31));
        atomicOr(&SUM_WORD(r >> 10), 1u << ((r >> 5) & 31));
        lo_w = min(lo_w, r >> 5);
        hi_w = max(hi_w, r >> 5);
    };
    /* sub-ranks i..i+7, any alignment.  ONE unconditional 16-byte load: entries past SUB are never used (every use is
     * guarded by R <= SUB) and the run's inverse array follows its sub-ranks, so the bytes exist.  A load under a
     * branch makes the compiler drain vmcnt right behind it -- the "prefetch" then waits for its own round trip AND
     * for the scattered result stores of the group before (that was half of this kernel's time). */
    auto load8 = [&](uint32_t i, uint32_t (&v)[4]) {
        uint4 t;
        __builtin_memcpy(&t, rk + i, 16);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    };
    /* first set bit strictly above / below bit b0 of word w0, given that word and its two neighbours.  The bitmap is
     * dense (two thirds full) on text, so the answer nearly always lies in those three words: that case is branch-free
     * 64-bit arithmetic (the walk is bound by the instructions and branches a lone wavefront per SIMD can issue: 2.4 ms ->
     * see DESIGN.md); further words through the summary */
    auto succ_far = [&](uint32_t w0) -> uint32_t {                       /* first non-empty word >= w0 + 2 */
        const uint32_t f = w0 + 2u;
        if (f > hi_w) return WALK_NONE;
        uint32_t k = f >> 5, m = SUM_WORD(k) & (~0u << (f & 31u));
        while (!m && ++k <= (hi_w >> 5)) m = SUM_WORD(k);
        if (!m) { hi_w = min(hi_w, w0 + 1u); return WALK_NONE; }
        const uint32_t w = (k << 5) + (uint32_t)__builtin_ctz(m);
        return (w << 5) + (uint32_t)__builtin_ctz(BM_WORD(w));
    };
    auto pred_far = [&](uint32_t w0) -> uint32_t {                       /* last non-empty word <= w0 - 2 */
        if (w0 < 2u || w0 - 2u < lo_w) return WALK_NONE;
        const uint32_t l = w0 - 2u;
        uint32_t k = l >> 5, m = SUM_WORD(k) & (~0u >> (31u - (l & 31u)));
        while (!m && k > (lo_w >> 5)) m = SUM_WORD(--k);
        if (!m) { lo_w = max(lo_w, w0 - 1u); return WALK_NONE; }
        const uint32_t w = (k << 5) + 31u - (uint32_t)__builtin_clz(m);
        return (w << 5) + 31u - (uint32_t)__builtin_clz(BM_WORD(w));
    };
    /* the three-word case; returns bit 0: nothing above in them, bit 1: nothing below (the far scans are then due) */
    auto near3 = [&](uint32_t w0, uint32_t b0, uint32_t here, uint32_t next, uint32_t prev, uint32_t &su, uint32_t &pr) -> uint32_t {
        const uint32_t nx = w0 + 1 < NW ? next : 0u, pv = w0 ? prev : 0u;
        const uint64_t up = ((((uint64_t)nx << 32) | here) >> 1) >> b0;                       /* bits above b0, then word w0+1 */
        const uint64_t dn = (((uint64_t)here << 32) | pv) & ((1ull << (32u + b0)) - 1ull);     /* word w0-1, then bits below b0 */
        su = (w0 << 5) + b0 + 1u + (uint32_t)__builtin_ctzll(up | (1ull << 63));
        pr = (w0 << 5) + 31u - (uint32_t)__builtin_clzll(dn | 1ull);                           /* ((w0-1) << 5) + 63 - clz */
        return (up ? 0u : 1u) | (dn ? 0u : 2u);
    };
    auto neighbours = [&](uint32_t w0, uint32_t b0, uint32_t here, uint32_t next, uint32_t prev) -> uint32_t {   /* successor | predecessor << 16 */
        uint32_t su, pr;
        const uint32_t miss = near3(w0, b0, here, next, prev, su, pr);
        if (miss & 1u) su = succ_far(w0);
        if (miss & 2u) pr = pred_far(w0);
        return su | (pr << 16);
    };
    auto query = [&](uint32_t q) -> uint32_t {                /* successor | predecessor << 16 */
        const uint32_t w0 = q >> 5, b0 = q & 31;
        const uint32_t here = BM_WORD(w0), next = BM_WORD(min(w0 + 1, NW - 1)), prev = BM_WORD(w0 ? w0 - 1 : 0);
        return neighbours(w0, b0, here, next, prev);
    };
    /* (the fill leaves the summary alone -- 4095 more LDS operations per walker -- and builds it from the bitmap afterwards) */
    auto fill_bit = [&](uint32_t r) {
        atomicOr(&BM_WORD(r >> 5), 1u << (r & 31));
        lo_w = min(lo_w, r >> 5);
        hi_w = max(hi_w, r >> 5);
    };
    auto summarize = [&]() {
        for (uint32_t k = 0; k < NS; k++) {
            uint32_t m = 0;
            for (uint32_t j = 0; j < 32u && (k << 5) + j < NW; j++) m |= BM_WORD((k << 5) + j) ? 1u << j : 0u;
            SUM_WORD(k) = m;
        }
    };
    auto fill = [&](uint32_t b) {                            /* set the bits of positions [0, b) */
        uint32_t i = 0;
        for (; i + 32 <= b; i += 32) {                       /* four loads in flight: the fill is bound by their latency */
            uint32_t v[4][4];
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) load8(i + 8 * g4, v[g4]);
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++)
#pragma unroll
                for (int j = 0; j < 8; j++) fill_bit((v[g4][j >> 1] >> (16 * (j & 1))) & 0xFFFFu);
        }
        for (; i + 8 <= b; i += 8) {
            uint32_t v[4];
            load8(i, v);
#pragma unroll
            for (int j = 0; j < 8; j++) fill_bit((v[j >> 1] >> (16 * (j & 1))) & 0xFFFFu);
        }
        for (; i < b; i++) fill_bit(rk[i]);
        summarize();
    };
    if (head) {
        /* start of the input: y < sb looks back at [0, y) only; lane l takes y in [l*C, (l+1)*C) after setting [0, l*C) */
        const uint32_t ymax = min(usb, R), C = (usb + 63u) / 64u;
        const uint32_t y0 = min(lane * C, ymax), y1 = min(y0 + C, ymax);
        fill(y0);
        for (uint32_t y = y0; y < y1; y++) {
            const uint32_t r = rk[y];
            wb0[y] = query(r);
            set_bit(r);
        }
        return;
    }
    fill(min(usb - 1, R));                                   /* first window minus its last position: [0, sb-1) */
    uint32_t r_add = usb - 1 < R ? (uint32_t)rk[usb - 1] : WALK_NONE;
    /* One step.  The forward query never looks at q's own bit (strict masks), so it may read the
     * bitmap BEFORE q is cleared, together with the backward query; and since the lane owns its bitmap
     * the clear is a plain store of the word it has just read. */
    auto step = [&](uint32_t q, uint32_t ry, uint32_t &resf, uint32_t &resb) {
        if (r_add != WALK_NONE) set_bit(r_add);
        const bool hasy = ry != WALK_NONE;
        const uint32_t wq = q >> 5, bq = q & 31, wy = hasy ? ry >> 5 : wq, by_ = ry & 31;
        const uint32_t hq = BM_WORD(wq), nq = BM_WORD(min(wq + 1, NW - 1)), pq = BM_WORD(wq ? wq - 1 : 0);
        const uint32_t hy = BM_WORD(wy), ny = BM_WORD(min(wy + 1, NW - 1)), py = BM_WORD(wy ? wy - 1 : 0);
        /* both queries out of the words at hand; ONE wave-uniform branch for the far scans of either (rare) */
        uint32_t suq, prq, suy, pry;
        const uint32_t mq = near3(wq, bq, hq, nq, pq, suq, prq);
        const uint32_t my = hasy ? near3(wy, by_, hy, ny, py, suy, pry) : 0u;
        if (__ballot((mq | my) != 0u)) {
            if (mq & 1u) suq = succ_far(wq);
            if (mq & 2u) prq = pred_far(wq);
            if (my & 1u) suy = succ_far(wy);
            if (my & 2u) pry = pred_far(wy);
        }
        resf = suq | (prq << 16);
        resb = hasy ? suy | (pry << 16) : (WALK_NONE | (WALK_NONE << 16));
        const uint32_t left = hq & ~(1u << bq);
        BM_WORD(wq) = left;
        atomicAnd(&SUM_WORD(wq >> 5), left ? ~0u : ~(1u << (wq & 31u)));     /* (unconditional: cheaper than a branch around it) */
        r_add = ry;                                          /* position t+sb enters at the next step */
    };
    uint32_t t = 0;
    /* groups of 16 steps: four 16-byte sub-rank loads, eight 16-byte result stores = one whole 64-byte line per lane
     * and result array (a lane's results are its own stream: 32-byte half lines made the L2 fetch the rest of every
     * line: 4.1 GB of HBM traffic per 100 MB for 0.8 GB of results).  The loads run TWO groups ahead and are
     * unconditional (always inside the run's two arrays): the wait for a group's sub-ranks then leaves the result
     * stores of the two groups before it in flight; with the loads one group ahead (or under a branch) every group
     * waited for its predecessor's scattered stores to complete */
    uint32_t vq[8], vy[8], aq[8], ay[8], bq[8], by8[8], rf[16], rb[16];
    auto load16 = [&](uint32_t i, uint32_t (&v)[8]) {
        uint4 t0, t1;
        __builtin_memcpy(&t0, rk + i, 16);
        __builtin_memcpy(&t1, rk + i + 8, 16);
        v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w;
        v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
    };
    load16(t, vq);
    load16(t + usb, vy);
    load16(t + 16, aq);
    load16(t + 16 + usb, ay);
    /* the four loads land here, once (an empty asm that consumes them): otherwise the compiler's wait at the top of
     * the loop body must also cover the first pass, where vq / vy are still load destinations, and ends up draining
     * the previous group's stores on every pass */
    asm volatile("" ::"v"(vq[0] ^ vq[7]), "v"(vy[0] ^ vy[7]), "v"(aq[0] ^ aq[7]), "v"(ay[0] ^ ay[7]));
    for (; t + 16 <= tb; t += 16) {
        load16(t + 32, bq);                                  /* t + 32 + usb + 16 <= SUB + 32: at most into the next run's array / the results behind the last one */
        load16(t + 32 + usb, by8);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t q = (vq[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
            const uint32_t ry = t + j + usb < R ? (vy[j >> 1] >> (16 * (j & 1))) & 0xFFFFu : WALK_NONE;
            step(q, ry, rf[j], rb[j]);
        }
        uint4 *o = reinterpret_cast<uint4 *>(of + t);
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = make_uint4(rf[4 * j], rf[4 * j + 1], rf[4 * j + 2], rf[4 * j + 3]);
        o = reinterpret_cast<uint4 *>(ob + t);
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = make_uint4(rb[4 * j], rb[4 * j + 1], rb[4 * j + 2], rb[4 * j + 3]);
#pragma unroll
        for (int j = 0; j < 8; j++) { vq[j] = aq[j]; vy[j] = ay[j]; aq[j] = bq[j]; ay[j] = by8[j]; }
    }
    for (; t < tb; t++) {
        uint32_t f1, b1;
        step(rk[t], t + usb < R ? (uint32_t)rk[t + usb] : WALK_NONE, f1, b1);
        of[t] = f1;
        ob[t] = b1;
    }
#undef BM_WORD
#undef SUM_WORD
}

/* sub-ranks -> positions -> the two per-position results of the match stage, per RUN with the run's inverse array
 * (scattered from its sub-ranks) and bytes in LDS: the four inverse look-ups and the two LCPs of a position are
 * gathers (64 lanes, 64 different lines) -- one thread per position straight out of L1 they took 1.8 ms per 100 MB,
 * bound by the texture path (round 1); out of LDS they are bank accesses.  One workgroup per walker run; the
 * workgroup of the input's first run also answers y < sb (wb0). */
#define WFIN_BLOCK 256
#define WFIN_PER 8                                   /* positions per thread whose walker results are fetched ahead (runs of 2048) */
__global__ __launch_bounds__(WFIN_BLOCK) void k_walk_final_lds(const uint8_t *__restrict__ in, uint32_t n, int sb, int la, uint32_t SBu,
                                                               uint32_t TILE, uint32_t region0, uint32_t nregions, uint32_t run_len,
                                                               uint32_t runs_per_tile, const uint16_t *__restrict__ subs,
                                                               const uint32_t *__restrict__ wf, const uint32_t *__restrict__ wb,
                                                               const uint32_t *__restrict__ wb0, uint32_t *__restrict__ ps,
                                                               uint8_t *__restrict__ maxlen)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t wfin_smem[];
    const uint32_t SUB = run_len + SBu;
    uint16_t *s_ix = reinterpret_cast<uint16_t *>(wfin_smem);                       /* SUB entries (SUB is a multiple of 8) */
    uint8_t *s_by = wfin_smem + (size_t)SUB * 2;                                    /* SUB + la + 16 bytes */
    const uint32_t tid = threadIdx.x;
    const uint32_t reg = blockIdx.x / runs_per_tile, run = blockIdx.x - reg * runs_per_tile;
    const uint64_t t0_64 = (uint64_t)(region0 + reg) * TILE;
    if (t0_64 >= n) return;
    const uint32_t t0 = (uint32_t)t0_64;
    const uint32_t lt1 = n - t0 < TILE ? n - t0 : TILE;
    const uint32_t lo = run * run_len;
    if (lo >= lt1) return;
    const uint32_t tb = min(run_len, lt1 - lo);
    {
        /* the inverse (sub-rank -> position - lo) is a scatter of the run's sub-ranks: the sort kernel used to export it
         * as a second array (1.2 GB written and read again per 100 MB) */
        const uint16_t *rkg = subs + ((size_t)reg * runs_per_tile + run) * SUB;
        const uint64_t rend64 = t0_64 + TILE + (uint32_t)sb;
        const uint32_t Rreg = (rend64 < n ? (uint32_t)rend64 : n) - t0;
        const uint32_t R = min(SUB, Rreg - lo);                                      /* sub-ranked positions [0, R) of this run */
        for (uint32_t e = tid * 8; e < R; e += WFIN_BLOCK * 8) {
            const uint4 v = *reinterpret_cast<const uint4 *>(rkg + e);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (e + q < R) s_ix[(w[q >> 1] >> (16 * (q & 1))) & 0xFFFFu] = (uint16_t)(e + q);
        }
    }
    {
        const uint64_t avail = (uint64_t)n + LZ77X_PAD - ((uint64_t)t0 + lo);       /* bytes that exist from t0 + lo on (0xFF tail included) */
        const uint32_t want = SUB + (uint32_t)la + 16;
        const uint32_t nb = (uint32_t)(avail < want ? avail : want) & ~3u;
        const uint8_t *src = in + t0 + lo;                                          /* t0, lo: multiples of 8 */
        for (uint32_t i = tid * 4; i < nb; i += WFIN_BLOCK * 4)
            *reinterpret_cast<uint32_t *>(s_by + i) = *reinterpret_cast<const uint32_t *>(src + i);
    }
    /* the walkers' answers for this thread's positions travel with the staging loads (round 5: under the loop below every
     * iteration began with its own round trip; run_len <= WFIN_BLOCK * WFIN_PER) */
    uint32_t fv[WFIN_PER], bv[WFIN_PER];
    const bool pre = run_len <= WFIN_BLOCK * WFIN_PER;
    if (pre) {
#pragma unroll
        for (int q = 0; q < WFIN_PER; q++) {
            const uint32_t t = min(tid + (uint32_t)q * WFIN_BLOCK, tb - 1u);
            const size_t rel = (size_t)reg * TILE + lo + t;
            fv[q] = wf[rel];
            bv[q] = wb[rel];
        }
    }
    __syncthreads();
    const uint32_t usb = (uint32_t)sb;
    auto longest = [&](uint32_t b, uint32_t ly /* relative to lo */) -> uint32_t {
        const uint32_t left = n - (t0 + lo + ly);
        const int cap = (int)(left < (uint32_t)la ? left : (uint32_t)la) - 1;
        uint32_t best = 0;
        if ((b & 0xFFFFu) != WALK_NONE) best = (uint32_t)lcp_capped<true>(s_by, (uint32_t)s_ix[b & 0xFFFFu], ly, cap);
        if ((b >> 16) != WALK_NONE) {
            const uint32_t l2 = (uint32_t)lcp_capped<true>(s_by, (uint32_t)s_ix[b >> 16], ly, cap);
            best = l2 > best ? l2 : best;
        }
        return best;
    };
    auto one = [&](uint32_t t, uint32_t f, uint32_t bw) {
        const uint32_t lx = lo + t, x = t0 + lx;
        const bool evicted = (uint64_t)x + usb < n;                                /* only evicted positions matter */
        uint32_t P = 0, S = 0;
        if (evicted) {
            if ((f & 0xFFFFu) != WALK_NONE) S = (uint32_t)s_ix[f & 0xFFFFu] - t;
            if ((f >> 16) != WALK_NONE) P = (uint32_t)s_ix[f >> 16] - t;
        }
        ps[x] = P | (S << 16);
        if (evicted) maxlen[x + usb] = (uint8_t)longest(bw, t + usb);
    };
    if (pre) {
#pragma unroll
        for (int q = 0; q < WFIN_PER; q++) {
            const uint32_t t = tid + (uint32_t)q * WFIN_BLOCK;
            if (t < tb) one(t, fv[q], bv[q]);
        }
    } else {
        for (uint32_t t = tid; t < tb; t += WFIN_BLOCK) {
            const size_t rel = (size_t)reg * TILE + lo + t;
            one(t, wf[rel], wb[rel]);
        }
    }
    if (region0 + reg == 0 && run == 0) {
        const uint32_t lim = min(usb, n);
        for (uint32_t lx = tid; lx < lim; lx += WFIN_BLOCK) maxlen[lx] = (uint8_t)longest(wb0[lx], lx);
    }
}

#ifdef LZ77X_VARIANTS   /* (round 1's large-window walkers (a bitmap per lane in HBM)) */
#include "variants/walk_big.inc"
#endif

/* ---- large windows, production: ONE WAVEFRONT per run, the bitmap in LDS, 64 steps at a time -------------
 *
 * k_walk_big above gives every lane a 34 KB bitmap in HBM: each access of a step is a random 64-byte line and the
 * stage is bound by how many of those the memory system moves (114 ms on S3).  Here a wavefront owns one bitmap
 * over the region's rank space in LDS (RP bits + a one-bit-per-word summary: 33 KB at RP 262144, four wavefronts
 * per CU) and its 64 lanes take 64 CONSECUTIVE steps t = tg + i together.  Lane i's window is [t, t+sb); all 64
 * windows share the core [tg+64, tg+sb), differ in the fringe: the old positions O_j = tg+j (in lane i's window
 * for j >= i; for its forward query, j > i) and the new ones N_j = tg+sb+j (for j < i).  So per group:
 *   1. every lane clears its own O_i: the bitmap is the core;
 *   2. every lane queries the core for the neighbours of rank[t+sb] (backward) and rank[t] (forward);
 *   3. all-to-all over the 2 x 64 fringe ranks (broadcast with v_readlane): a fringe rank that is admissible
 *      for the lane and closer than the core's answer replaces it;
 *   4. every lane sets its N_i: the bitmap is the window of the next group.
 * The first run of the input starts sb steps early with an empty bitmap: its steps t < 0 have no forward query
 * and answer the backward queries of y < sb (wb0).  Results leave as ranks, like k_walk_big's. */
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}

#define WW_WAVES 4u                                 /* wavefronts that share a run's bitmap (below) */

/* inclusive OR over the lanes up to mine (DPP: rows of 16, then the rows' last lanes) */
__device__ __forceinline__ uint32_t wave_incl_or(uint32_t x)
{
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);      /* row_shr:1 */
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);      /* row_shr:2 */
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);      /* row_shr:4 */
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);      /* row_shr:8 */
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);      /* row_bcast:15 into rows 1 and 3 */
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);      /* row_bcast:31 into rows 2 and 3 */
    return x;
}

__global__ __launch_bounds__(64 * WW_WAVES) void k_walk_wave(const uint32_t *__restrict__ ranks, uint32_t n, int sb, uint32_t RP,
                                                             uint32_t TILE, uint32_t region0, uint32_t nregions, uint32_t run_len,
                                                             uint32_t runs_per_tile, uint2 *__restrict__ wf, uint2 *__restrict__ wb,
                                                             uint2 *__restrict__ wb0, uint32_t fringe_v4 /* variants build: rounds 3-4's fringe */)
{
    /* FOUR wavefronts share a run and its bitmap (round 3).  A group of 64 steps is (1) clear, (2) core queries, (4) set --
     * which must follow one another group after group -- and (3) the fringe all-to-all, 64 x 8 compare/select/min-max:
     * 4200 of a group's 5100 instructions, and it reads nothing but the group's own 128 ranks.  With one wavefront per
     * run (and the 33 KB bitmap allowing four per CU, one per SIMD, a dependent instruction every ~7 cycles: 36 K
     * cycles a group) everything was serial.  Now wavefront w takes the groups w, w+4, ...: it computes its group's
     * fringe whenever it likes and does (1)(2)(4) when the turn counter in LDS reaches its group. */
    extern __shared__ uint32_t wv_bm[];
    __shared__ uint32_t s_turn, s_lo, s_hi, s_mn[WW_WAVES], s_mx[WW_WAVES];
    __shared__ uint32_t s_ftab[128 * WW_WAVES];                 /* per wavefront: local rank of a fringe value -> the rank */
    const uint32_t NW = RP >> 5, NS = (NW + 31) >> 5;
    uint32_t *word = wv_bm, *summ = wv_bm + NW;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t run = blockIdx.x % runs_per_tile, reg = blockIdx.x / runs_per_tile;
    const uint32_t usb = (uint32_t)sb;
    const uint32_t NONE = 0xFFFFFFFFu;
    if (reg >= nregions) return;
    const uint64_t t0_64 = (uint64_t)(region0 + reg) * TILE;
    if (t0_64 >= n) return;
    const uint32_t t0 = (uint32_t)t0_64;
    const uint64_t rend64 = (uint64_t)t0 + TILE + usb;
    const uint32_t R = (rend64 < n ? (uint32_t)rend64 : n) - t0;
    const uint32_t lt1 = n - t0 < TILE ? n - t0 : TILE;
    const uint32_t ta = run * run_len;
    if (ta >= lt1) return;
    const uint32_t tb = min(ta + run_len, lt1);
    const uint32_t *rk = ranks + (size_t)reg * (2 * (size_t)RP + 8);
    const bool first = region0 + reg == 0 && run == 0;
    auto wsync = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };

    for (uint32_t w = threadIdx.x; w < NW + NS; w += 64 * WW_WAVES) wv_bm[w] = 0;
    if (threadIdx.x == 0) s_turn = 0;
    __syncthreads();
    /* no set bit lies in a summary word below lo_s or above hi_s: what keeps a query that has no neighbour on one side
     * -- every step of a stretch of equal bytes -- from scanning the whole summary.  Shared through LDS: read at the
     * start of a turn, written back at its end */
    {
        uint32_t mn = NONE, mx = 0;
        if (!first) {
            const uint32_t e = min(ta + usb, R);
            for (uint32_t i0 = ta; i0 < e; i0 += 64 * WW_WAVES * 4) {
                uint32_t r[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const uint32_t i = i0 + 64 * WW_WAVES * u + threadIdx.x; r[u] = i < e ? rk[i] : NONE; }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (r[u] != NONE) {
                        atomicOr(&word[r[u] >> 5], 1u << (r[u] & 31));
                        atomicOr(&summ[r[u] >> 10], 1u << ((r[u] >> 5) & 31));
                        mn = min(mn, r[u] >> 10);
                        mx = max(mx, r[u] >> 10);
                    }
            }
        }
        mn = wave_min_u32(mn);
        mx = wave_max_u32(mx);
        if (lane == 0) { s_mn[wave] = mn; s_mx[wave] = mx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t a = NONE, b = 0;
            for (uint32_t w = 0; w < WW_WAVES; w++) { a = min(a, s_mn[w]); b = max(b, s_mx[w]); }
            s_lo = first ? NS : min(a, NS);
            s_hi = first ? 0u : b;
        }
        __syncthreads();
    }
    uint32_t lo_s = NS, hi_s = 0;
    /* first set rank in words > w / last set rank in words < w, through the summary (a summary bit whose word is
     * empty is skipped); *dry: the scan ran out -- nothing lies beyond w on that side */
    auto up_slow = [&](uint32_t w, bool &dry) -> uint32_t {
        uint32_t sw = w >> 5;
        if (sw > hi_s) { dry = true; return NONE; }
        uint32_t sm = summ[sw] & ~((2u << (w & 31)) - 1u);
        for (;;) {
            while (!sm && ++sw <= hi_s) sm = summ[sw];
            if (!sm) { dry = true; return NONE; }
            const uint32_t w2 = (sw << 5) + (uint32_t)__builtin_ctz(sm);
            const uint32_t m = word[w2];
            if (m) return (w2 << 5) + (uint32_t)__builtin_ctz(m);
            sm &= sm - 1;
        }
    };
    auto down_slow = [&](uint32_t w, bool &dry) -> uint32_t {
        int32_t sw = (int32_t)(w >> 5);
        if (sw < (int32_t)lo_s) { dry = true; return NONE; }
        uint32_t sm = summ[sw] & ((1u << (w & 31)) - 1u);
        for (;;) {
            while (!sm && --sw >= (int32_t)lo_s) sm = summ[sw];
            if (!sm) { dry = true; return NONE; }
            const uint32_t top = 31u - (uint32_t)__builtin_clz(sm);
            const uint32_t w2 = ((uint32_t)sw << 5) + top;
            const uint32_t m = word[w2];
            if (m) return (w2 << 5) + 31u - (uint32_t)__builtin_clz(m);
            sm &= ~(1u << top);
        }
    };
    uint32_t dry_up = NONE, dry_dn = 0;                      /* lowest word an up-scan / highest word a down-scan ran dry from */
    auto core = [&](uint32_t q, uint32_t &su, uint32_t &pr) {     /* q's own bit is not set */
        const uint32_t w0 = q >> 5, b0 = q & 31;
        const uint32_t here = word[w0], next = word[min(w0 + 1, NW - 1)], prev = word[w0 ? w0 - 1 : 0];
        const uint32_t m1 = here & ~((2u << b0) - 1u), m2 = here & ((1u << b0) - 1u);
        bool dry = false;
        if (m1) su = (w0 << 5) + (uint32_t)__builtin_ctz(m1);
        else if (w0 + 1 < NW && next) su = ((w0 + 1) << 5) + (uint32_t)__builtin_ctz(next);
        else {
            su = w0 + 1 < NW ? up_slow(w0 + 1, dry) : NONE;
            if (dry) dry_up = min(dry_up, w0 >> 5);
        }
        dry = false;
        if (m2) pr = (w0 << 5) + 31u - (uint32_t)__builtin_clz(m2);
        else if (w0 > 0 && prev) pr = ((w0 - 1) << 5) + 31u - (uint32_t)__builtin_clz(prev);
        else {
            pr = w0 > 0 ? down_slow(w0 - 1, dry) : NONE;
            if (dry) dry_dn = max(dry_dn, w0 >> 5);
        }
    };

    uint2 *of = wf + (size_t)reg * TILE, *ob = wb + (size_t)reg * TILE;
    const int32_t tstart = first ? -(int32_t)usb : (int32_t)ta;
    auto fetch = [&](int32_t tg, uint32_t &rx, uint32_t &ry) {
        const int32_t t = tg + (int32_t)lane;
        const bool act = t < (int32_t)tb;
        rx = (act && t >= 0) ? rk[t] : NONE;
        const uint32_t y = (uint32_t)(t + (int32_t)usb);
        ry = (act && y < R) ? rk[y] : NONE;
    };
    const int32_t stride = 64 * (int32_t)WW_WAVES;
    uint32_t rx, ry, rxn = NONE, ryn = NONE;
    int32_t tg0 = tstart + 64 * (int32_t)wave;
    if (tg0 < (int32_t)tb) fetch(tg0, rx, ry);
    uint32_t gi = wave;                                          /* my group's number: its turn */
    for (int32_t tg = tg0; tg < (int32_t)tb; tg += stride, gi += WW_WAVES) {
        if (tg + stride < (int32_t)tb) fetch(tg + stride, rxn, ryn);      /* my next group's ranks travel while this one runs */
        const int32_t t = tg + (int32_t)lane;
        /* 3. the fringe (needs nothing but this group's ranks): O_j counts for the backward query of lanes i <= j and the
         *    forward query of lanes i < j, N_j for both queries of lanes i > j.  Ranks are distinct; "closer than the
         *    current answer" in unsigned arithmetic with NONE = 0xFFFFFFFF as "no successor" and pred stored + 1 (0 = none). */
        uint32_t fs = NONE, bs = NONE, fp1 = 0u, bp1 = 0u;
#ifdef LZ77X_VARIANTS
        if (fringe_v4) {
            /* rounds 3-4: all-to-all, a candidate costs compare + mask + select + min/max (the cross-check of the form below) */
            const uint64_t hasx = __ballot(rx != NONE), hasy = __ballot(ry != NONE);
            for (int j = 0; j < 64; j++) {
                const uint32_t co = (uint32_t)__builtin_amdgcn_readlane((int)rx, j);
                const uint32_t cn = (uint32_t)__builtin_amdgcn_readlane((int)ry, j);
                const uint64_t upto = (2ull << j) - 1ull;          /* lanes i <= j */
                const uint64_t below = upto >> 1;                  /* lanes i < j */
                const uint64_t ob_ok = ((hasx >> j) & 1ull) ? (upto & hasy) : 0ull;
                const uint64_t of_ok = ((hasx >> j) & 1ull) ? (below & hasx) : 0ull;
                const uint64_t nb_ok = ((hasy >> j) & 1ull) ? (~upto & hasy) : 0ull;
                const uint64_t nf_ok = ((hasy >> j) & 1ull) ? (~upto & hasx) : 0ull;
                const bool me_ob = (ob_ok >> lane) & 1ull, me_of = (of_ok >> lane) & 1ull;
                const bool me_nb = (nb_ok >> lane) & 1ull, me_nf = (nf_ok >> lane) & 1ull;
                bs = min(bs, (me_ob && co > ry) ? co : NONE);
                bp1 = max(bp1, (me_ob && co < ry) ? co + 1u : 0u);
                fs = min(fs, (me_of && co > rx) ? co : NONE);
                fp1 = max(fp1, (me_of && co < rx) ? co + 1u : 0u);
                bs = min(bs, (me_nb && cn > ry) ? cn : NONE);
                bp1 = max(bp1, (me_nb && cn < ry) ? cn + 1u : 0u);
                fs = min(fs, (me_nf && cn > rx) ? cn : NONE);
                fp1 = max(fp1, (me_nf && cn < rx) ? cn + 1u : 0u);
            }
        } else
#endif
        {
            /* Round 5.  The fringe is the walk itself in small: S = O_0 .. O_63 N_0 .. N_63 is a sequence of 128 ranks, lane
             * i's backward query asks for the neighbours of S[64+i] in the window S[i .. i+63] and its forward query for
             * those of S[i] in S[i+1 .. i+63].  So: rank the 128 values among themselves (the only all-to-all left: two
             * broadcasts, four compare-and-adds a step), make the windows 128-bit sets of LOCAL ranks -- the O bits of the
             * lanes from i on are all O bits minus a prefix OR, the N bits below i a prefix OR (DPP scans) -- and a query is
             * a find-first-bit on either side of the value's own bit; a 128-entry table per wavefront turns the local rank
             * back into the rank.  800 instructions a group where the eight select/min-max per pair took 4200. */
            uint32_t Lx = 0, Ly = 0;
            for (int j = 0; j < 64; j++) {
                const uint32_t co = (uint32_t)__builtin_amdgcn_readlane((int)rx, j);
                const uint32_t cn = (uint32_t)__builtin_amdgcn_readlane((int)ry, j);
                Lx += (co < rx ? 1u : 0u) + (cn < rx ? 1u : 0u);     /* (NONE is below nothing) */
                Ly += (co < ry ? 1u : 0u) + (cn < ry ? 1u : 0u);
            }
            const bool vx = rx != NONE, vy = ry != NONE;
            uint32_t *tab = s_ftab + 128u * wave;
            if (vx) tab[Lx] = rx;
            if (vy) tab[Ly] = ry;
            uint32_t ox[4], io[4], in_[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                ox[k] = vx && (Lx >> 5) == k ? 1u << (Lx & 31u) : 0u;
                const uint32_t nyk = vy && (Ly >> 5) == k ? 1u << (Ly & 31u) : 0u;
                io[k] = wave_incl_or(ox[k]);
                in_[k] = wave_incl_or(nyk) & ~nyk;                   /* N bits of the lanes below mine */
            }
            uint32_t wbk[4], wfk[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t to = (uint32_t)__builtin_amdgcn_readlane((int)io[k], 63);   /* every O bit */
                wbk[k] = (to & ~(io[k] & ~ox[k])) | in_[k];         /* O bits of the lanes from mine on, N bits below */
                wfk[k] = wbk[k] & ~ox[k];
            }
            const uint64_t wb_lo = ((uint64_t)wbk[1] << 32) | wbk[0], wb_hi = ((uint64_t)wbk[3] << 32) | wbk[2];
            const uint64_t wf_lo = ((uint64_t)wfk[1] << 32) | wfk[0], wf_hi = ((uint64_t)wfk[3] << 32) | wfk[2];
            auto succ128 = [](uint64_t lo, uint64_t hi, uint32_t L) -> uint32_t {      /* first member above L, 128: none */
                const uint64_t a_lo = L < 64u ? lo & ~((2ull << (L & 63u)) - 1ull) : 0ull;
                const uint64_t a_hi = L < 64u ? hi : hi & ~((2ull << (L & 63u)) - 1ull);
                return a_lo ? (uint32_t)__builtin_ctzll(a_lo) : a_hi ? 64u + (uint32_t)__builtin_ctzll(a_hi) : 128u;
            };
            auto pred128 = [](uint64_t lo, uint64_t hi, uint32_t L) -> uint32_t {      /* last member below L, 128: none */
                const uint64_t b_hi = L >= 64u ? hi & ((1ull << (L & 63u)) - 1ull) : 0ull;
                const uint64_t b_lo = L >= 64u ? lo : lo & ((1ull << (L & 63u)) - 1ull);
                return b_hi ? 127u - (uint32_t)__builtin_clzll(b_hi) : b_lo ? 63u - (uint32_t)__builtin_clzll(b_lo) : 128u;
            };
            const uint32_t sb_ = succ128(wb_lo, wb_hi, Ly), pb_ = pred128(wb_lo, wb_hi, Ly);
            const uint32_t sf_ = succ128(wf_lo, wf_hi, Lx), pf_ = pred128(wf_lo, wf_hi, Lx);
            wsync();                                                 /* the table is written */
            if (vy) {
                if (sb_ < 128u) bs = tab[sb_];
                if (pb_ < 128u) bp1 = tab[pb_] + 1u;
            }
            if (vx) {
                if (sf_ < 128u) fs = tab[sf_];
                if (pf_ < 128u) fp1 = tab[pf_] + 1u;
            }
            wsync();                                                 /* (read before the next group of this wavefront writes) */
        }
        /* my turn: the groups before mine have left the bitmap as my window's */
        while (__hip_atomic_load(&s_turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != gi) __builtin_amdgcn_s_sleep(2);
        lo_s = s_lo;
        hi_s = s_hi;
        /* 1. the old positions leave */
        if (rx != NONE) {
            const uint32_t bit = 1u << (rx & 31);
            const uint32_t old = atomicAnd(&word[rx >> 5], ~bit);
            if ((old & ~bit) == 0u) atomicAnd(&summ[rx >> 10], ~(1u << ((rx >> 5) & 31)));
        }
        wsync();
        /* 2. the core */
        uint32_t cfs = NONE, cfp = NONE, cbs = NONE, cbp = NONE;
        dry_up = NONE;
        dry_dn = 0;
        if (rx != NONE) core(rx, cfs, cfp);
        if (ry != NONE) core(ry, cbs, cbp);
        fs = min(fs, cfs);
        bs = min(bs, cbs);
        const uint32_t fp = max(fp1, cfp + 1u) - 1u, bp = max(bp1, cbp + 1u) - 1u;      /* NONE + 1 = 0 */
        wsync();
        /* 4. the new positions enter */
        uint32_t mn = NONE, mx = 0;
        if (ry != NONE) {
            atomicOr(&word[ry >> 5], 1u << (ry & 31));
            atomicOr(&summ[ry >> 10], 1u << ((ry >> 5) & 31));
            mn = mx = ry >> 10;
        }
        /* bounds: sets widen them, a scan that ran dry tightens them (conservative either way) */
        {
            const uint32_t du = wave_min_u32(dry_up), dd = wave_max_u32(dry_dn);
            if (du != NONE) hi_s = min(hi_s, du);
            if (dd != 0u) lo_s = max(lo_s, dd);
            const uint32_t smn = wave_min_u32(mn), smx = wave_max_u32(mx);
            if (smn != NONE) { lo_s = min(lo_s, smn); hi_s = max(hi_s, smx); }
        }
        if (lane == 0) { s_lo = lo_s; s_hi = hi_s; }
        wsync();                                                 /* (every LDS operation of the turn has been performed) */
        if (lane == 0) __hip_atomic_store(&s_turn, gi + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (t < (int32_t)tb) {
            if (t >= 0) {
                of[t] = make_uint2(fs, fp);
                ob[t] = make_uint2(bs, bp);
            } else {
                wb0[t + (int32_t)usb] = make_uint2(bs, bp);
            }
        }
        rx = rxn;
        ry = ryn;
    }
}

__global__ __launch_bounds__(256) void k_walk_final_big(const uint8_t *__restrict__ in, uint32_t n, int sb, int la, uint32_t RP,
                                                        uint32_t TILE, uint32_t region0, uint32_t nregions,
                                                        const uint32_t *__restrict__ ranks, const uint2 *__restrict__ wf,
                                                        const uint2 *__restrict__ wb, const uint2 *__restrict__ wb0,
                                                        uint32_t *__restrict__ ps, uint8_t *__restrict__ maxlen)
{
    const uint64_t rel = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (rel >= (uint64_t)nregions * TILE) return;
    const uint64_t x64 = (uint64_t)region0 * TILE + rel;
    if (x64 >= n) return;
    const uint32_t reg = (uint32_t)(rel / TILE);
    const uint32_t t0 = (region0 + reg) * TILE, lx = (uint32_t)x64 - t0;
    const uint32_t *ix = ranks + (size_t)reg * (2 * (size_t)RP + 8) + RP + 8;
    const uint32_t NONE = 0xFFFFFFFFu;
    const uint8_t *by = in + t0;
    {
        const uint2 f = wf[rel];
        uint32_t P = 0, S = 0;
        if (x64 + (uint32_t)sb < n) {                                    /* only evicted positions matter */
            if (f.x != NONE) S = ix[f.x] - lx;
            if (f.y != NONE) P = ix[f.y] - lx;
        }
        ps[x64] = P | (S << 16);
    }
    auto longest = [&](uint2 b, uint32_t ly) -> uint32_t {               /* max LCP with the two candidates */
        const uint32_t left = n - (t0 + ly);
        const int cap = (int)(left < (uint32_t)la ? left : (uint32_t)la) - 1;
        uint32_t best = 0;
        if (b.x != NONE) best = (uint32_t)lcp_capped<false>(by, ix[b.x], ly, cap);
        if (b.y != NONE) {
            const uint32_t l2 = (uint32_t)lcp_capped<false>(by, ix[b.y], ly, cap);
            best = l2 > best ? l2 : best;
        }
        return best;
    };
    const uint64_t y64 = x64 + (uint32_t)sb;
    if (y64 < n) maxlen[y64] = (uint8_t)longest(wb[rel], lx + (uint32_t)sb);
    if (region0 + reg == 0 && lx < (uint32_t)sb) maxlen[x64] = (uint8_t)longest(wb0[lx], lx);
}

/* ------------------------------------------------------------------ large regions: one sort for all of them ----
 *
 * k_match<false,3> gives every region of 262144 slots to ONE workgroup: 16 chunk sorts in LDS one after the other,
 * then merge levels straight out of global memory, sixteen wavefronts on a CU to hide their round trips (S3: 115 ms).
 * But regions overlap (region r = positions [r*TILE, r*TILE+RP), TILE = 3/4 RP) and (key, position) is ONE total
 * order: a sorted run of positions is the same run in every region that contains it.  When TILE is a multiple of
 * 65536 the regions are unions of globally aligned 64 K blocks, so
 *   k_big_chunks     sorts every 16 K chunk of the input once, in LDS, exactly like a small-window region
 *                    (uint16 indices + staged key bytes: two workgroups per CU);
 *   k_big_partition  + k_big_merge: merge levels over the WHOLE launch, a workgroup per 2048 outputs: the merge
 *                    path of its two diagonals comes from the partition kernel, its slice of both runs is loaded
 *                    coalesced, the 16-byte key heads of its 2048 elements are fetched in parallel (the only
 *                    random global reads) and the serial merge steps run on LDS.  16 K -> 32 K -> 64 K once for
 *                    everybody (indices relative to the block: uint16), 64 K -> 128 K -> 256 K per region (uint32,
 *                    relative to the region's first position; the rank array doubles as the second buffer);
 *   k_big_ranks      inverts the order into the rank array.
 * A region's order then also holds the positions [R, RP) the old kernel pushed to the end as "invalid" (they belong
 * to the next region's tile); no window ever contains them, the walkers never ask for their ranks, and the rank-order
 * tie-break bounds its walk by the positions that exist (lz77k_big_sort_shared). */
#define BIG_CH  (16u * MATCH_BLOCK)
#define BIG_BLK 65536u
#define BIG_T   2048u
#define BIG_MB  256

__global__ __launch_bounds__(MATCH_BLOCK, 8) void k_big_chunks(const uint8_t *__restrict__ in, uint32_t n, int la, uint64_t pos0,
                                                              uint16_t *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *ix = reinterpret_cast<uint16_t *>(smem);
    uint8_t *lby = smem + BIG_CH * sizeof(uint16_t);
    const uint32_t tid = threadIdx.x;
    const uint64_t base = pos0 + (uint64_t)blockIdx.x * BIG_CH;
    const uint32_t Rl = base >= n ? 0u : (n - (uint32_t)base < BIG_CH ? n - (uint32_t)base : BIG_CH);
    constexpr uint32_t NB = BIG_CH + 256 + 24;            /* (lds: BIG_CH + 256 + 32 bytes behind the indices) */
    if (Rl) stage_rev<MATCH_BLOCK>(lby, NB, in, base, n, tid);
    for (uint32_t i = tid; i < BIG_CH; i += MATCH_BLOCK) ix[i] = (uint16_t)i;
    __syncthreads();
    if (Rl) region_sort_merge<uint16_t, true, MATCH_BLOCK, true>(ix, lby, Rl, la, tid, 0, NB - 4u);
    uint16_t *o = out + (size_t)blockIdx.x * BIG_CH;
    for (uint32_t e = tid * 8; e < BIG_CH; e += MATCH_BLOCK * 8) *reinterpret_cast<uint4 *>(o + e) = *reinterpret_cast<const uint4 *>(ix + e);
}

/* One merge level: pair p = (group g = p / ppg, j = p % ppg) merges the runs A | B of L elements each that start at
 * src + g*src_gstride + j*2L into dst + g*dst_gstride + j*2L; keys are read at in + pos0 + g*kb_gstride + index.
 * rel: the stored indices are relative to their own run (shared uint16 levels): + j*2L (+ L for B) makes them
 * relative to the group's key base. */
struct big_level {
    const void *src;
    void *dst;
    uint64_t src_gstride, dst_gstride, pos0;
    uint32_t L, ppg, kb_gstride, rel, npairs;
};

template <class InT> struct big_pair {
    const InT *A, *B;
    const uint8_t *by;
    uint32_t a_add, b_add, R;
    size_t out_off;
};

template <class InT>
__device__ __forceinline__ big_pair<InT> big_pair_of(const big_level &lv, const uint8_t *in, uint32_t n, uint32_t p)
{
    big_pair<InT> bp;
    const uint32_t g = p / lv.ppg, j = p % lv.ppg;
    bp.A = reinterpret_cast<const InT *>(lv.src) + (size_t)g * lv.src_gstride + (size_t)j * 2 * lv.L;
    bp.B = bp.A + lv.L;
    bp.a_add = lv.rel ? j * 2 * lv.L : 0u;
    bp.b_add = lv.rel ? j * 2 * lv.L + lv.L : 0u;
    const uint64_t kb = lv.pos0 + (uint64_t)g * lv.kb_gstride;
    bp.by = kb < n ? in + kb : in;                           /* nothing valid: R = 0 and no key is ever looked at */
    bp.R = kb < n ? n - (uint32_t)kb : 0u;
    bp.out_off = (size_t)g * lv.dst_gstride + (size_t)j * 2 * lv.L;
    return bp;
}

__device__ __forceinline__ void big_masks(int la, uint32_t (&m)[4])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int rem = la - 4 * i;
        m[i] = rem >= 4 ? 0xFFFFFFFFu : rem <= 0 ? 0u : 0xFFFFFFFFu << (8 * (4 - rem));
    }
}

/* split[p*(ntile+1) + k] = how many of the first k*BIG_T outputs of pair p come from A */
template <class InT>
__global__ __launch_bounds__(256) void k_big_partition(const uint8_t *__restrict__ in, uint32_t n, int la, big_level lv,
                                                       uint32_t *__restrict__ split)
{
    const uint32_t ntile = 2 * lv.L / BIG_T;
    const uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t p = (uint32_t)(gid / (ntile + 1)), k = (uint32_t)(gid % (ntile + 1));
    if (p >= lv.npairs) return;
    const big_pair<InT> bp = big_pair_of<InT>(lv, in, n, p);
    uint32_t m[4];
    big_masks(la, m);
    const uint32_t L = lv.L, d = k * BIG_T;
    uint32_t lo = d > L ? d - L : 0, hi = d < L ? d : L;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t a = (uint32_t)bp.A[mid] + bp.a_add, b = (uint32_t)bp.B[d - 1 - mid] + bp.b_add;
        const key16 ka = load_key16<false>(bp.by, a, a < bp.R, m), kb = load_key16<false>(bp.by, b, b < bp.R, m);
        if (sort_less16<false>(bp.by, a, ka, b, kb, bp.R, la)) lo = mid + 1; else hi = mid;
    }
    split[gid] = lo;
}

template <class InT, class OutT>
__global__ __launch_bounds__(BIG_MB) void k_big_merge(const uint8_t *__restrict__ in, uint32_t n, int la, big_level lv,
                                                     const uint32_t *__restrict__ split)
{
    constexpr uint32_t VT = BIG_T / BIG_MB;
    __shared__ uint32_t lidx[BIG_T];
    __shared__ __attribute__((aligned(16))) uint64_t lkey[2 * BIG_T];
    const uint32_t ntile = 2 * lv.L / BIG_T, tid = threadIdx.x;
    const uint32_t p = blockIdx.x / ntile, k = blockIdx.x % ntile;
    const big_pair<InT> bp = big_pair_of<InT>(lv, in, n, p);
    uint32_t m[4];
    big_masks(la, m);
    const uint32_t a0 = split[(size_t)p * (ntile + 1) + k], a1 = split[(size_t)p * (ntile + 1) + k + 1];
    const uint32_t b0 = k * BIG_T - a0, na = a1 - a0, nb = BIG_T - na;
    const uint32_t R = bp.R;
    const uint8_t *by = bp.by;
#pragma unroll
    for (uint32_t q = 0; q < VT; q++) {
        const uint32_t i = tid + BIG_MB * q;
        const uint32_t v = i < na ? (uint32_t)bp.A[a0 + i] + bp.a_add : (uint32_t)bp.B[b0 + (i - na)] + bp.b_add;
        lidx[i] = v;
        const key16 kk = load_key16<false>(by, v, v < R, m);
        lkey[2 * i] = kk.hi;
        lkey[2 * i + 1] = kk.lo;
    }
    __syncthreads();
    auto less = [&](uint32_t ia, uint32_t ib) -> bool {      /* LDS slots */
        key16 ka, kb;
        ka.hi = lkey[2 * ia]; ka.lo = lkey[2 * ia + 1];
        kb.hi = lkey[2 * ib]; kb.lo = lkey[2 * ib + 1];
        return sort_less16<false>(by, lidx[ia], ka, lidx[ib], kb, R, la);
    };
    const uint32_t d = tid * VT;
    uint32_t lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (less(mid, na + (d - 1 - mid))) lo = mid + 1; else hi = mid;
    }
    uint32_t ia = lo, ib = d - lo, o[VT];
#pragma unroll
    for (uint32_t r = 0; r < VT; r++) {
        const bool va = ia < na, vb = ib < nb;
        const bool take_a = !vb || (va && less(ia, na + ib));
        o[r] = lidx[take_a ? ia : na + ib];
        ia += take_a ? 1u : 0u;
        ib += take_a ? 0u : 1u;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < VT; r++) lidx[d + r] = o[r];
    __syncthreads();
    OutT *out = reinterpret_cast<OutT *>(lv.dst) + bp.out_off + (size_t)k * BIG_T;
#pragma unroll
    for (uint32_t q = 0; q < VT; q++) out[tid + BIG_MB * q] = (OutT)lidx[tid + BIG_MB * q];
}

/* rk[ix[r]] = r: the rank array from the order */
__global__ __launch_bounds__(256) void k_big_ranks(uint32_t *__restrict__ ranks, uint32_t RP)
{
    uint32_t *rk = ranks + (size_t)blockIdx.y * (2 * (size_t)RP + 8);
    const uint32_t *ix = rk + RP + 8;
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r < RP) rk[ix[r]] = r;
    if (r < 8) rk[RP + r] = 0;
}

int lz77k_big_sort_shared(const lz77x_geom &g)
{
    return !g.fast && g.shifted && g.RP >= 2 * BIG_BLK && g.TILE % BIG_BLK == 0 && !LZ77X_VENV("LZ77X_BIG_SORT_V1") &&
           !(LZ77X_VENV("LZ77X_SORT_VARIANT") && atoi(LZ77X_VENV("LZ77X_SORT_VARIANT")));
}

/* bytes behind the walkers' part of the scratch: two uint16 arrays over the launch's positions + the splits */
static size_t big_sort_span(const lz77x_geom &g, uint32_t nregions) { return (size_t)(nregions - 1) * g.TILE + g.RP; }
static size_t big_sort_split_words(const lz77x_geom &g, uint32_t nregions)
{
    /* per level: npairs * (2L / T + 1); elements never exceed nregions * RP */
    return ((size_t)nregions * g.RP) / BIG_T + ((size_t)nregions * g.RP) / (2 * BIG_CH) + 64;
}
static size_t big_sort_extra_bytes(const lz77x_geom &g, uint32_t nregions)
{
    if (!nregions) return 0;
    return 2 * ((big_sort_span(g, nregions) * 2 + 255) & ~(size_t)255) + big_sort_split_words(g, nregions) * 4 + 256;
}

template <class InT, class OutT>
static hipError_t big_level_run(const uint8_t *d_in, uint32_t n, int la, const big_level &lv, uint32_t *d_split, hipStream_t s)
{
    const uint32_t ntile = 2 * lv.L / BIG_T;
    const uint64_t nsplit = (uint64_t)lv.npairs * (ntile + 1);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_big_partition<InT>), dim3((uint32_t)((nsplit + 255) / 256)), dim3(256), 0, s, d_in, n, la, lv, d_split);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_big_merge<InT, OutT>), dim3(lv.npairs * ntile), dim3(BIG_MB), 0, s, d_in, n, la, lv, d_split);
    return hipGetLastError();
}

/* order + ranks of regions [region0, region0 + nregions) into ranks[] ((2RP+8) words per region); d_extra:
 * big_sort_extra_bytes() */
static hipError_t big_sort(const uint8_t *d_in, uint32_t n, const lz77x_geom &g, uint32_t region0, uint32_t nregions, uint32_t *ranks,
                           void *d_extra, hipStream_t s)
{
    const size_t span = big_sort_span(g, nregions), sbytes = (span * 2 + 255) & ~(size_t)255;
    uint16_t *S[2] = {reinterpret_cast<uint16_t *>(d_extra), reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(d_extra) + sbytes)};
    uint32_t *split = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(d_extra) + 2 * sbytes);
    const uint64_t pos0 = (uint64_t)region0 * g.TILE;
    const size_t lds = (size_t)BIG_CH * 2 + BIG_CH + 256 + 32;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_big_chunks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_big_chunks, dim3((uint32_t)(span / BIG_CH)), dim3(MATCH_BLOCK), lds, s, d_in, n, g.la, pos0, S[0]);
    int cur = 0;
    for (uint32_t L = BIG_CH; L < BIG_BLK; L <<= 1) {       /* shared levels */
        big_level lv;
        lv.src = S[cur]; lv.dst = S[cur ^ 1];
        lv.src_gstride = lv.dst_gstride = 2 * (size_t)L;
        lv.pos0 = pos0; lv.L = L; lv.ppg = 1; lv.kb_gstride = 2 * L; lv.rel = 1;
        lv.npairs = (uint32_t)(span / (2 * (size_t)L));
        if ((e = big_level_run<uint16_t, uint16_t>(d_in, n, g.la, lv, split, s)) != hipSuccess) return e;
        cur ^= 1;
    }
    /* region levels: the last one writes the order (ix = ranks + RP + 8), the one before it the rank array's space */
    const size_t stride = 2 * (size_t)g.RP + 8;
    int nlev = 0;
    for (uint32_t L = BIG_BLK; L < g.RP; L <<= 1) nlev++;
    int lev = 0;
    for (uint32_t L = BIG_BLK; L < g.RP; L <<= 1, lev++) {
        const bool to_ix = ((nlev - 1 - lev) & 1) == 0;
        big_level lv;
        lv.dst = ranks + (to_ix ? g.RP + 8 : 0);
        lv.dst_gstride = stride;
        lv.pos0 = pos0; lv.L = L; lv.ppg = g.RP / (2 * L); lv.kb_gstride = g.TILE;
        lv.npairs = nregions * lv.ppg;
        if (lev == 0) {
            lv.src = S[cur]; lv.src_gstride = g.TILE; lv.rel = 1;
            e = big_level_run<uint16_t, uint32_t>(d_in, n, g.la, lv, split, s);
        } else {
            lv.src = ranks + (to_ix ? 0 : g.RP + 8); lv.src_gstride = stride; lv.rel = 0;
            e = big_level_run<uint32_t, uint32_t>(d_in, n, g.la, lv, split, s);
        }
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_big_ranks, dim3(g.RP / 256, nregions), dim3(256), 0, s, ranks, g.RP);
    return hipGetLastError();
}

#define WALK_RUN_BIG_DEFAULT 1024u

/* C1 (TILE 12288): 2048 steps per walker = six runs per region: half the sub-rank exports of the sort kernel and
 * half the window fills of the walkers against 1024 (match stage 8.1 -> 7.2 ms per 100 MB); 2560+ no longer
 * divide the tile evenly and lose walker occupancy (48 -> 64 KB of bitmaps per wavefront) */
#define WALK_RUN_DEFAULT 2048u

/* steps per walker on the LDS path (LZ77X_WALK_RUN overrides): a multiple of 8 in [256, RP/2 - SBu] --
 * the two sub-rank arrays of a run (2 * (run + SBu) uint16) are staged in the RP*2 bytes of LDS the key
 * bytes occupied, and a short run means many arrays per region */
static uint32_t walk_run_lds(const lz77x_geom &g)
{
    const char *rl = getenv("LZ77X_WALK_RUN");
    uint32_t run_len = rl && atoi(rl) > 0 ? (uint32_t)atoi(rl) : WALK_RUN_DEFAULT;
    const uint32_t hi = (g.RP / 2 - g.SBu) & ~7u;
    if (run_len > hi) run_len = hi;
    if (run_len < 256) run_len = 256 < hi ? 256 : hi;
    if (run_len > g.TILE) run_len = g.TILE;
    return (run_len + 7u) & ~7u;
}

/* steps per walker on the large-window path (LZ77X_WALK_RUN_BIG overrides) */
static uint32_t walk_run_big(const lz77x_geom &g)
{
    const char *rl = getenv("LZ77X_WALK_RUN_BIG");
    uint32_t run_len = rl && atoi(rl) > 0 ? (uint32_t)atoi(rl) : WALK_RUN_BIG_DEFAULT;
    return run_len < g.TILE ? run_len : g.TILE;
}

/* steps per wavefront of the LDS large-window walker (LZ77X_WALK_RUN_WAVE overrides): its fill costs sb bit sets,
 * so runs are long */
static uint32_t walk_run_wave(const lz77x_geom &g)
{
    const char *rl = getenv("LZ77X_WALK_RUN_WAVE");
    uint32_t run_len = rl && atoi(rl) > 0 ? (uint32_t)atoi(rl) : 8192u;
    run_len = (run_len + 63u) & ~63u;
    return run_len < g.TILE ? run_len : g.TILE;
}

size_t lz77k_match_lds_bytes(const lz77x_geom &g)
{
    /* large windows: one chunk of the index (uint32) + that chunk's key bytes (CH + la + slack) */
    if (!g.fast) return g.RP > 16u * MATCH_BLOCK ? (size_t)16 * MATCH_BLOCK * (sizeof(uint32_t) + 1) + 512 : 0;
    return (size_t)g.RP * 2 + (size_t)(g.RP + 8) * 2;      /* ix + union{bytes, rk}: RP >= 4096 > la + 11 */
}

/* large windows: rank + inverse (uint32), one global bitmap per walker (v1 walkers), backward results per position */
static size_t big_walk_scratch_bytes(const lz77x_geom &g, uint32_t nregions)
{
    const size_t run_len = walk_run_big(g);
    const size_t runs = (g.TILE + run_len - 1) / run_len;
    const size_t nws = (g.RP >> 5) + (((g.RP >> 5) + 31) >> 5);
    return (((size_t)nregions * ((2 * (size_t)g.RP + 8) * 4 + runs * nws * 4 + (size_t)g.TILE * 16) + (size_t)g.SBu * 8 + 256) + 255) & ~(size_t)255;
}

/* small windows: sub-rank + inverse (uint16 each) per run, then the walkers' fwd/bwd results per position */
static size_t c1_walk_scratch_bytes(const lz77x_geom &g, uint32_t nregions)
{
    const size_t rl = walk_run_lds(g), runs = (g.TILE + rl - 1) / rl;
    return (((size_t)nregions * (runs * (rl + g.SBu) * 2 + (size_t)g.TILE * 8) + (size_t)g.SBu * 4 + 256) + 255) & ~(size_t)255;
}

size_t lz77k_match_scratch_bytes(const lz77x_geom &g, uint32_t nregions)
{
    if (g.fast) {
        return c1_walk_scratch_bytes(g, nregions) + (c1_shared_sort(g) ? ((size_t)nregions * 3 + 1) * C1_CH * 2 + 256 : 0);
    }
    return big_walk_scratch_bytes(g, nregions) + (lz77k_big_sort_shared(g) ? big_sort_extra_bytes(g, nregions) : 0);
}

template <bool FAST, int MODE>
static hipError_t launch_match(const uint8_t *d_in, uint32_t n, const lz77x_geom &g, uint32_t region0, uint32_t nregions,
                               uint32_t *d_ps, uint8_t *d_maxlen, void *d_scratch, hipStream_t s, uint16_t *d_order = nullptr,
                               const uint16_t *d_chunks = nullptr)
{
    const size_t lds = lz77k_match_lds_bytes(g);
    auto fn = k_match<FAST, MODE>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const char *sv = LZ77X_VENV("LZ77X_SORT_VARIANT");
    hipLaunchKernelGGL(fn, dim3(nregions), dim3(MATCH_BLOCK), lds, s, d_in, n, g.sb, g.la, g.SBu, g.RP, g.TILE, region0,
                       d_ps, d_maxlen, reinterpret_cast<uint32_t *>(d_scratch), sv ? atoi(sv) : 0, g.fast ? walk_run_lds(g) : 0u, d_order, d_chunks);
    return hipGetLastError();
}

hipError_t lz77k_match(const uint8_t *d_in, uint32_t n, const lz77x_geom &g, uint32_t region0, uint32_t nregions,
                       uint32_t *d_ps, uint8_t *d_maxlen, void *d_scratch, int variant, hipStream_t s, hipEvent_t *ev_sort,
                       uint32_t *d_ranks_all)
{
    if (nregions == 0) return hipSuccess;
    if (g.shifted != !(variant == 1 || variant == 3)) return hipErrorInvalidValue;     /* layout of g must match the variant */
    if (ev_sort && !(variant == 0 || variant > 3)) ev_sort = nullptr;
#define LZ77K_MATCH_ARGS d_in, n, g, region0, nregions, d_ps, d_maxlen, d_scratch, s
    if (g.fast) {
#ifdef LZ77X_VARIANTS
        if (variant == 1) return launch_match<true, 1>(LZ77K_MATCH_ARGS);
        if (variant == 2) return launch_match<true, 2>(LZ77K_MATCH_ARGS);
        if (variant == 3) return launch_match<true, 0>(LZ77K_MATCH_ARGS);      /* exhaustive packed pair scan */
#else
        if (variant != 0) return hipErrorNotSupported;
#endif
        /* production: sort -> per-lane bitmap walkers -> finalize */
        hipError_t e = ev_sort ? hipEventRecord(ev_sort[0], s) : hipSuccess;
        if (e != hipSuccess) return e;
        const uint16_t *d_chunks = nullptr;
        if (c1_shared_sort(g)) {
            uint16_t *ch = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(d_scratch) + c1_walk_scratch_bytes(g, nregions));
            hipLaunchKernelGGL(k_c1_chunks, dim3(nregions * 3u + 1u), dim3(C1_BLOCK), 0, s, d_in, n, g.la, (uint64_t)region0 * g.TILE, ch);
            d_chunks = ch;
        }
        if (ev_sort && (e = hipEventRecord(ev_sort[3], s)) != hipSuccess) return e;        /* [0] .. [3]: the chunk sort alone */
        e = launch_match<true, 3>(LZ77K_MATCH_ARGS, reinterpret_cast<uint16_t *>(d_ranks_all), d_chunks);
        if (e != hipSuccess) return e;
        if (ev_sort && (e = hipEventRecord(ev_sort[1], s)) != hipSuccess) return e;
        const uint32_t run_len = walk_run_lds(g);
        const uint32_t runs = (g.TILE + run_len - 1) / run_len, SUB = run_len + g.SBu;
        uint16_t *subs = reinterpret_cast<uint16_t *>(d_scratch);
        uint32_t *wf = reinterpret_cast<uint32_t *>(subs + (size_t)nregions * runs * SUB);
        uint32_t *wb = wf + (size_t)nregions * g.TILE;
        uint32_t *wb0 = wb + (size_t)nregions * g.TILE;
        const uint64_t walkers = (uint64_t)nregions * runs;
        const size_t nwords = (SUB + 31) >> 5;
        const size_t lds = (nwords + ((nwords + 31) >> 5)) * 64 * sizeof(uint32_t);      /* the lanes' bitmaps + their summaries */
        if (lds > 48 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_walk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k_walk, dim3((uint32_t)((walkers + 63) / 64) + (region0 == 0 ? 1u : 0u)), dim3(64), lds, s, subs, n, g.sb, g.SBu, g.TILE,
                           region0, nregions, run_len, runs, wf, wb, wb0, region0 == 0 ? 1 : 0);
        if (ev_sort && (e = hipEventRecord(ev_sort[2], s)) != hipSuccess) return e;
        const size_t flds = (size_t)SUB * 3 + (size_t)g.la + 32;
        hipLaunchKernelGGL(k_walk_final_lds, dim3((uint32_t)walkers), dim3(WFIN_BLOCK), flds, s, d_in, n, g.sb, g.la, g.SBu,
                           g.TILE, region0, nregions, run_len, runs, subs, wf, wb, wb0, d_ps, d_maxlen);
        return hipGetLastError();
    }
#ifdef LZ77X_VARIANTS
    if (variant == 1) return launch_match<false, 1>(LZ77K_MATCH_ARGS);
    if (variant == 2) return launch_match<false, 2>(LZ77K_MATCH_ARGS);
    if (variant == 3) return launch_match<false, 0>(LZ77K_MATCH_ARGS);         /* exhaustive pair scan */
#else
    if (variant != 0) return hipErrorNotSupported;
#endif
    {
        /* production for large windows: sort (ranks stay in scratch) -> global-bitmap walkers -> finalize */
        hipError_t e = ev_sort ? hipEventRecord(ev_sort[0], s) : hipSuccess;
        if (e != hipSuccess) return e;
        if (ev_sort && (e = hipEventRecord(ev_sort[3], s)) != hipSuccess) return e;        /* (no chunk kernel of its own to time) */
        /* rank + inverse of every region: in the caller's persistent array (the rank-order tie-break
         * reads them again once the host stage is through the chunk) or at the head of the scratch */
        const size_t stride = 2 * (size_t)g.RP + 8;
        uint32_t *ranks = d_ranks_all ? d_ranks_all + (size_t)region0 * stride : reinterpret_cast<uint32_t *>(d_scratch);
        uint32_t *bitmaps = d_ranks_all ? reinterpret_cast<uint32_t *>(d_scratch) : ranks + (size_t)nregions * stride;
        if (lz77k_big_sort_shared(g))
            e = big_sort(d_in, n, g, region0, nregions, ranks, reinterpret_cast<uint8_t *>(d_scratch) + big_walk_scratch_bytes(g, nregions), s);
        else
            e = launch_match<false, 3>(d_in, n, g, region0, nregions, d_ps, d_maxlen, ranks, s);
        if (e != hipSuccess) return e;
        if (ev_sort && (e = hipEventRecord(ev_sort[1], s)) != hipSuccess) return e;
        const size_t nws = (g.RP >> 5) + (((g.RP >> 5) + 31) >> 5);
        const bool wave_walk = !LZ77X_VENV("LZ77X_WALK_BIG_V1") && nws * 4 <= 64 * 1024;
        const uint32_t run_len = wave_walk ? walk_run_wave(g) : walk_run_big(g);
        const uint32_t runs = (g.TILE + run_len - 1) / run_len;
        const uint64_t walkers = (uint64_t)nregions * runs;
        uint2 *wf = reinterpret_cast<uint2 *>(bitmaps + (wave_walk ? 0 : walkers * nws));      /* (2RP+8)*4 and nws*4 are multiples of 8 */
        uint2 *wb = wf + (size_t)nregions * g.TILE;
        uint2 *wb0 = wb + (size_t)nregions * g.TILE;
        if (wave_walk) {
            /* one wavefront per run, bitmap in LDS, 64 steps at a time */
            const size_t wlds = nws * 4;
            if (wlds > 48 * 1024) {
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_walk_wave), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds);
                if (e != hipSuccess) return e;
            }
            hipLaunchKernelGGL(k_walk_wave, dim3((uint32_t)walkers), dim3(64 * WW_WAVES), wlds, s, ranks, n, g.sb, g.RP, g.TILE, region0, nregions, run_len,
                               runs, wf, wb, wb0, LZ77X_VENV("LZ77X_WALK_FRINGE_V4") ? 1u : 0u);
        } else {
#ifdef LZ77X_VARIANTS
            hipLaunchKernelGGL(k_walk_big, dim3((uint32_t)((walkers + 63) / 64)), dim3(64), 0, s, ranks, n, g.sb, g.RP, g.TILE,
                               region0, nregions, run_len, runs, bitmaps, wf, wb, wb0, LZ77X_VENV("LZ77X_WALK_DEBUG") ? atoi(LZ77X_VENV("LZ77X_WALK_DEBUG")) : 0);
#else
            return hipErrorNotSupported;                    /* (RP <= 2^18: the bitmap of every legal window fits the LDS) */
#endif
        }
        if (ev_sort && (e = hipEventRecord(ev_sort[2], s)) != hipSuccess) return e;
        const uint64_t npos = (uint64_t)nregions * g.TILE;
        hipLaunchKernelGGL(k_walk_final_big, dim3((uint32_t)((npos + 255) / 256)), dim3(256), 0, s, d_in, n, g.sb, g.la, g.RP, g.TILE,
                           region0, nregions, ranks, wf, wb, wb0, d_ps, d_maxlen);
        return hipGetLastError();
    }
#undef LZ77K_MATCH_ARGS
}

