/*
 * k_priow.hip -- the priority recurrence of k_prio.hip with a WORKGROUP per block of steps.
 *
 * k_prio.hip gives a block of the recurrence (tree.c:202-231: which node a delete() promotes) to ONE wavefront whose
 * 64 lanes take 64 consecutive steps, split into rounds wherever a step reads a cell an earlier one of the 64 writes;
 * the sb live cells are a ring of 32-bit priorities in LDS.  Two things stop that design at sb = 4096:
 *   - the ring: 65535 cells x 4 B do not fit the 160 KB of a CU;
 *   - the latency: one wavefront per CU (the ring takes the whole LDS) at ~20 cycles per step is slower than a host core.
 * Here W = 256 .. 1024 threads take W consecutive steps.  The conflicts that split a group into rounds are between
 * steps whose neighbours lie close together in the window; with windows of 64 K they are rare -- measured on the S3
 * stream: 5.5 rounds per 1024 steps against 17 (one per 64-step group) -- so a block costs a third of the rounds and a
 * round is one workgroup barrier.  And the ring holds 18-bit CODES instead of priorities: every live priority is either
 * OLD (< the block's first position: then it sits in one of the sb cells the block starts from, and only its order
 * among those matters: its rank) or the own position of a cell of this block:
 *       code = rank among the old entry values            (0 .. sb-1)
 *            | sb + (position - x0)                        (sb .. sb + B + ring_n)
 * an order-preserving map into 2^18 values: 16 bits in a uint16 ring + a 2-bit plane = 147 KB.  The ranks come from a
 * sort of the old entry values in LDS before the sweep (the ring's space, not yet in use); what is handed over leaves as
 * a true priority again through a per-block table rank -> value.
 *
 * The backward sweep (the blocks' maps) keeps a uint16 ring of exit cells, stores every step's exit cell in a scratch row and
 * takes the per-exit-cell minima from that row with LDS atomics once the ring is no longer needed (round 5: until then it
 * sent them to HBM atomically, 80-110 K atomics a block); the boundary scan (k_pw_maps, k_pw_cdest) applies the maps of a
 * group in sequence, half the cells at a time through LDS, because two vectors of 65535 priorities do not fit LDS either.
 * Runs of equal bytes (a step whose own cell the step before it writes) are chains inside a round, resolved with a scan of
 * threshold maps across the workgroup's wavefronts like in k_prio.hip (round 5).
 */
#include "kernels_common.h"
#include <stdio.h>
#include <type_traits>

typedef uint32_t pw_u32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t pw_u16x4 __attribute__((ext_vector_type(4)));
typedef uint16_t pw_u16x8 __attribute__((ext_vector_type(8)));

#define PW_NONE 0xFFFFFFFFu
#define PW_DEAD 0xFFFFu
#define PW_RANK0 0xFFFFFFFEu             /* k_pw_fwd: xval marker of old rank r = PW_RANK0 - r, until the block's last pass */
#define PW_SORT_CAP 32768u              /* keys sorted per pass of the rank prologue (128 KB of LDS) */

/* LZ77X_PW_DEBUG=1: cycle stamps of workgroup 0 of the last back / fwd / prep launch (development aid) */
#ifdef LZ77X_VARIANTS
__device__ unsigned long long pw_dbg[64];
#define PW_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) pw_dbg[k] = __builtin_readcyclecounter(); } while (0)
#define PW_NOTE(k, v) do { pw_dbg[k] = (v); } while (0)
#else
#define PW_STAMP(k) do { } while (0)
#define PW_NOTE(k, v) do { (void)(v); } while (0)
#endif

/* orders LDS traffic only: __syncthreads() also waits for every global load and store in flight (s_waitcnt vmcnt(0)),
 * which put a round trip to HBM -- the next group's prefetched operands, the last group's result stores -- into every
 * barrier of the sweeps' inner loops */
__device__ __forceinline__ void pw_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/* Rows that ONE workgroup writes and reads back within one launch (the scratch row of k_pw_back) stay at WORKGROUP scope: its
 * own loads and stores meet in the L2 of its XCD; other launches see the rows after the launch boundary.  Agent scope is the
 * wrong tool on this part (eight XCDs, an L2 each): agent-scope atomics and sc1 accesses are performed beyond the L2 -- a
 * round trip of several microseconds that every s_waitcnt vmcnt behind them pays -- and an agent-scope fence
 * (__threadfence()) is `buffer_wbl2 sc1` + `buffer_inv sc1`, a write-back and invalidate of the whole L2.  (Until round 5 the
 * blocks' loc rows were lowered with workgroup-scope atomics in HBM: ~50 G a second for the whole chip, section 2.2b of
 * DESIGN.md.)  Ordering inside the workgroup: pw_fence_wg (the operations have been performed: s_waitcnt vmcnt(0)) + a barrier. */
__device__ __forceinline__ void pw_fence_wg()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

/* ------------------------------------------------------------------ round masks ------ */

__device__ __forceinline__ void pw_cas_min16(uint32_t *tab, uint32_t idx, uint32_t tag)
{
    uint32_t *w = tab + (idx >> 1);
    const uint32_t sh = (idx & 1u) * 16u;
    uint32_t old = *w;                                      /* (no volatile: a volatile access through a cast pointer is a FLAT load, sc0 sc1,
                                                               behind s_waitcnt vmcnt(0) -- every outstanding global operation) */
    while (((old >> sh) & 0xFFFFu) > tag) {
        const uint32_t nw = (old & ~(0xFFFFu << sh)) | (tag << sh);
        const uint32_t got = atomicCAS(w, old, nw);
        if (got == old) break;
        old = got;
    }
}

__device__ __forceinline__ uint32_t pw_rd16(const uint32_t *tab, uint32_t idx)
{
    return (reinterpret_cast<const uint16_t *>(tab))[idx];  /* (re-read after every barrier: pw_lds_barrier clobbers memory) */
}

#ifdef LZ77X_VARIANTS   /* (round 3's round masks of the large windows (a tag per cell)) */
#include "variants/pw_prep.inc"
#endif

/* The same round masks with the cells RANKED first (round 4).  k_pw_prep keeps a 16-bit tag for each of the W + sb cells a
 * group can touch: 133 KB at sb 65535 -- ONE workgroup per CU -- and a compare-and-swap loop per tag.  But a group only
 * touches the <= 3 W cells its steps read or write: mark them in a bitmap (one bit per cell: 8 KB), take word popcounts and
 * their prefix sums, and a cell's rank among the touched cells (prefix of its word + the set bits below its own) indexes a
 * table of 3 W 32-bit entries -- native ds_min_u32, versions that never run out, 29 KB per workgroup.  256 threads take the
 * W steps (item q of thread t = step q * 256 + t, so a wavefront's lanes hold 64 consecutive steps and a ballot is a word
 * of the mask): eight workgroups per CU. */
#define PWP_T 256
template <int W>
__global__ __launch_bounds__(PWP_T, 8) void k_pw_prep_ranked(const uint32_t *__restrict__ ps, uint32_t nx, uint32_t tagn,
                                                           uint64_t *__restrict__ rmask, uint64_t *__restrict__ gate0,
                                                           uint64_t *__restrict__ cmask /* steps whose own cell the step before them writes in their round (k_prio.hip, "chains inside a round") */)
{
    constexpr int ITEMS = W / PWP_T;
    extern __shared__ uint32_t pwp_lds[];
    const uint32_t nw = (tagn + 31u) / 32u;                  /* words of the bitmap */
    uint32_t *bits = pwp_lds;                                /* nw */
    uint32_t *pref = bits + nw;                              /* nw: touched cells before the word */
    uint32_t *tab = pref + nw;                               /* 3 W: (version, lowest step of the round that writes the cell) */
    __shared__ uint32_t s_first[2], wsum[PWP_T / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t ngroups = (nx + W - 1u) / W;
    for (uint32_t i = tid; i < 3u * W; i += PWP_T) tab[i] = 0xFFFFFFFFu;
    if (tid < 2) s_first[tid] = W;
    uint32_t ver = 0;                                        /* counts up; code = 0x3FFFFE - ver: newer entries are smaller */
    int par = 0;
    const uint32_t wpt = (nw + PWP_T - 1u) / PWP_T;          /* bitmap words per thread (9 at sb 65535) */
    for (uint32_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        uint32_t p[ITEMS], sd[ITEMS];
        bool has[ITEMS];
#pragma unroll
        for (int q = 0; q < ITEMS; q++) {
            const uint32_t x = g * W + (uint32_t)q * PWP_T + tid;
            const uint32_t v = x < nx ? ps[x] : 0u;
            p[q] = v & 0xFFFFu;
            sd[q] = v >> 16;
            has[q] = p[q] && sd[q];
        }
        for (uint32_t i = tid; i < nw; i += PWP_T) bits[i] = 0u;
        pw_lds_barrier();
#pragma unroll
        for (int q = 0; q < ITEMS; q++) {
            if (has[q]) {
                const uint32_t c0 = (uint32_t)q * PWP_T + tid, c1 = c0 + p[q], c2 = c0 + sd[q];
                atomicOr(&bits[c0 >> 5], 1u << (c0 & 31u));
                atomicOr(&bits[c1 >> 5], 1u << (c1 & 31u));
                atomicOr(&bits[c2 >> 5], 1u << (c2 & 31u));
            }
        }
        pw_lds_barrier();
        {
            /* prefix sums of the word popcounts: a thread owns wpt consecutive words */
            const uint32_t w0 = tid * wpt;
            uint32_t cnt = 0;
            for (uint32_t k = 0; k < wpt; k++) cnt += w0 + k < nw ? (uint32_t)__builtin_popcount(bits[w0 + k]) : 0u;
            uint32_t incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, 64);
                if (lane >= (uint32_t)d) incl += t;
            }
            if (lane == 63) wsum[wave] = incl;
            pw_lds_barrier();
            uint32_t run = incl - cnt;
            for (uint32_t w = 0; w < wave; w++) run += wsum[w];
            for (uint32_t k = 0; k < wpt; k++)
                if (w0 + k < nw) { pref[w0 + k] = run; run += (uint32_t)__builtin_popcount(bits[w0 + k]); }
        }
        pw_lds_barrier();
        uint32_t r0[ITEMS], r1[ITEMS], r2[ITEMS];
#pragma unroll
        for (int q = 0; q < ITEMS; q++) {
            const uint32_t c0 = (uint32_t)q * PWP_T + tid, c1 = c0 + p[q], c2 = c0 + sd[q];
            auto rank = [&](uint32_t c) { return pref[c >> 5] + (uint32_t)__builtin_popcount(bits[c >> 5] & ((1u << (c & 31u)) - 1u)); };
            r0[q] = has[q] ? rank(c0) : 0u;
            r1[q] = has[q] ? rank(c1) : 0u;
            r2[q] = has[q] ? rank(c2) : 0u;
        }
        uint32_t isstart = tid == 0 ? 1u : 0u;               /* bit q: item q opens a round */
        uint32_t islink = 0;                                 /* bit q: item q is a link of a chain inside its round */
        uint32_t start = 0;
        for (;;) {
            const uint32_t code = 0x3FFFFEu - ver;
#pragma unroll
            for (int q = 0; q < ITEMS; q++) {
                const uint32_t i = (uint32_t)q * PWP_T + tid;
                if (has[q] && i >= start) atomicMin(&tab[r2[q]], (code << 10) | i);
            }
            pw_lds_barrier();
            if (tid == 0) s_first[par ^ 1] = W;
            uint32_t first_mine = W, linked_now = 0;
#pragma unroll
            for (int q = 0; q < ITEMS; q++) {
                const uint32_t i = (uint32_t)q * PWP_T + tid;
                bool blocked = false;
                if (has[q] && i > start) {
                    const uint32_t t0 = tab[r0[q]], t1 = tab[r1[q]], t2 = tab[r2[q]];
                    /* the step before this one hands its priority to THIS cell (S[x-1] = 1: runs of equal bytes): not a new
                     * round -- the sweep resolves such chains with a scan.  One writer per cell and round: a second one would
                     * read the first one's successor cell and start a round of its own (b2). */
                    const bool own = (t0 >> 10) == code && (t0 & 1023u) < i;
                    const bool linked = own && (t0 & 1023u) + 1u == i;
                    const bool b0 = own && !linked;
                    const bool b1 = (t1 >> 10) == code && (t1 & 1023u) < i;
                    const bool b2 = (t2 >> 10) == code && (t2 & 1023u) < i;
                    blocked = b0 | b1 | b2;
                    linked_now |= linked ? 1u << q : 0u;
                }
                const uint64_t bm = __ballot(blocked);
                if (bm) first_mine = min(first_mine, (uint32_t)q * PWP_T + wave * 64u + (uint32_t)__builtin_ctzll(bm));
            }
            if (first_mine < (uint32_t)W && lane == 0) atomicMin(&s_first[par], first_mine);
            pw_lds_barrier();
            const uint32_t first = s_first[par];
            ver++;
            par ^= 1;
            /* the steps [start, first) are a round; what they saw of each other is final */
#pragma unroll
            for (int q = 0; q < ITEMS; q++)
                if ((uint32_t)q * PWP_T + tid < first) islink |= linked_now & (1u << q);
            if (first >= (uint32_t)W) break;
            start = first;
#pragma unroll
            for (int q = 0; q < ITEMS; q++)
                if ((uint32_t)q * PWP_T + tid == first) isstart |= 1u << q;
        }
        /* (bit 0 of a group's first round-mask word says nothing -- step 0 always opens a round: cleared, it flags a group
         * with chains, as in k_prio_prep) */
        const bool any_link = __syncthreads_or(islink != 0u);
#pragma unroll
        for (int q = 0; q < ITEMS; q++) {
            uint64_t rm = __ballot((isstart >> q) & 1u);
            const uint64_t hm = __ballot(has[q]), cm = __ballot((islink >> q) & 1u);
            const uint32_t x0 = g * W + (uint32_t)q * PWP_T + wave * 64u;
            if (any_link && q == 0 && wave == 0) rm &= ~1ull;
            if (lane == 0 && x0 < nx) { rmask[x0 >> 6] = rm; gate0[x0 >> 6] = hm; cmask[x0 >> 6] = cm; }
        }
    }
}

/* ------------------------------------------------------------------ forward sweep ---- */

/* the live cells of a sweep: 32-bit priorities, or 18-bit codes as a uint16 array + a plane of 2 bits per cell */
template <bool PACK> struct pw_ring;

template <> struct pw_ring<false> {
    uint32_t *r;
    __device__ __forceinline__ uint32_t rd(uint32_t i) const { return r[i]; }
    __device__ __forceinline__ void wr(uint32_t i, uint32_t nv, uint32_t) const { r[i] = nv; }
    __device__ __forceinline__ void put(uint32_t i, uint32_t nv) const { r[i] = nv; }        /* old content unknown */
};

template <> struct pw_ring<true> {
    uint16_t *lo;
    uint32_t *hi;
    __device__ __forceinline__ uint32_t rd(uint32_t i) const
    {
        return (uint32_t)lo[i] | (((hi[i >> 4] >> ((i & 15u) * 2u)) & 3u) << 16);
    }
    /* one writer per cell and round, and it has read the cell in this round: the plane changes by an exclusive or of
     * old and new (other cells of the word may change under other lanes at the same time) */
    __device__ __forceinline__ void wr(uint32_t i, uint32_t nv, uint32_t ov) const
    {
        lo[i] = (uint16_t)nv;
        const uint32_t d = ((nv ^ ov) >> 16) & 3u;
        if (d) atomicXor(&hi[i >> 4], d << ((i & 15u) * 2u));
    }
    __device__ __forceinline__ void put(uint32_t i, uint32_t nv) const
    {
        const uint32_t oh = (hi[i >> 4] >> ((i & 15u) * 2u)) & 3u;
        lo[i] = (uint16_t)nv;
        const uint32_t d = ((nv >> 16) ^ oh) & 3u;
        if (d) atomicXor(&hi[i >> 4], d << ((i & 15u) * 2u));
    }
};

/* chains inside a round (k_prio.hip has the derivation): threshold maps f(a) = a < t ? a : c, closed under composition */
struct pw_tc { uint32_t t, c; };

__device__ __forceinline__ pw_tc pw_tc_then(pw_tc first, pw_tc second)
{
    pw_tc r;
    r.t = min(first.t, second.t);
    r.c = second.t < first.t ? second.c : (first.c < second.t ? first.c : second.c);
    return r;
}

template <int CTRL, int ROWS>
__device__ __forceinline__ pw_tc pw_tc_dpp(pw_tc v)
{
    /* lanes without a source (or outside the row mask) get the identity (t = 2^32 - 1 passes every value) */
    pw_tc e;
    e.t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v.t, CTRL, ROWS, 0xF, false);
    e.c = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.c, CTRL, ROWS, 0xF, false);
    return e;
}

/* inclusive scan over the 64 lanes: lane i <- v[i] o v[i-1] o ... o v[0] */
__device__ __forceinline__ pw_tc pw_tc_scan(pw_tc v)
{
    v = pw_tc_then(pw_tc_dpp<0x111, 0xF>(v), v);          /* row_shr:1 */
    v = pw_tc_then(pw_tc_dpp<0x112, 0xF>(v), v);          /* row_shr:2 */
    v = pw_tc_then(pw_tc_dpp<0x114, 0xF>(v), v);          /* row_shr:4 */
    v = pw_tc_then(pw_tc_dpp<0x118, 0xF>(v), v);          /* row_shr:8 */
    v = pw_tc_then(pw_tc_dpp<0x142, 0xA>(v), v);          /* row_bcast:15 into rows 1 and 3 */
    v = pw_tc_then(pw_tc_dpp<0x143, 0xC>(v), v);          /* row_bcast:31 into rows 2 and 3 */
    return v;
}

/* the plane word of sixteen consecutive codes c0 .. c0 + 15: their bits 16-17 change at most once, at the multiple of 2^16 */
__device__ __forceinline__ uint32_t pw_plane_word(uint32_t c0)
{
    const uint32_t h0 = (c0 >> 16) & 3u, t16 = 0x10000u - (c0 & 0xFFFFu);
    uint32_t hw = h0 * 0x55555555u;
    if (t16 < 16u) hw ^= (((h0 ^ (h0 + 1u)) & 3u) * 0x55555555u) & ~((1u << (2u * t16)) - 1u);
    return hw;
}

/* exclusive prefix of one value per thread over the workgroup (W threads); *total = the sum.  Two barriers. */
template <int W>
__device__ __forceinline__ uint32_t pw_block_excl(uint32_t v, uint32_t *s_w /* W/64 + 1 words */, uint32_t *total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= (uint32_t)d) inc += t;
    }
    __syncthreads();
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < W / 64; w++) {
        const uint32_t t = s_w[w];
        if ((uint32_t)w < wave) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

/* Block b = steps [b*B, min((b+1)*B, nx)).  in[b][i] = value of cell b*B+i before the block's first step.  The sweep is
 * the recurrence itself; gold[] (the gates the boundary values came from) is only compared against.
 * PACK: codes[b][i] / gval[b][r] are the block's scratch rows (code of entry cell i; value of old rank r). */
template <int W, bool PACK>
__global__ __launch_bounds__(W) void k_pw_fwd(const uint32_t *__restrict__ ps, uint32_t nx, uint32_t sb, uint32_t rs /* row stride of in / in_prev / codes / gval */, uint32_t B, uint32_t ring_n,
                                              uint32_t b_first, const uint64_t *__restrict__ rmask,
                                              const uint64_t *__restrict__ cmask /* steps that are links of a chain inside their round (k_pw_prep_ranked) */,
                                              const uint64_t *__restrict__ gold,
                                              uint64_t *__restrict__ gnew, const uint32_t *__restrict__ in, uint32_t *__restrict__ xval,
                                              uint32_t *__restrict__ summary, uint32_t voff, uint32_t *__restrict__ out_state,
                                              uint32_t ncarried, uint32_t *__restrict__ codes, uint32_t *__restrict__ gval,
                                              uint32_t sort_cap /* keys per pass of the rank prologue (<= PW_SORT_CAP) */,
                                              uint32_t probe /* timing probes (results wrong): 1 no main loop, 2 no sort, 4 no rank search */,
                                              uint32_t *__restrict__ in_prev /* rows: the cells each block's last sweep started from */,
                                              uint32_t have_prev /* 0: first sweep, in_prev holds nothing yet */,
                                              uint32_t *__restrict__ gates_changed /* [b] = this sweep flipped a gate of block b */)
{
    extern __shared__ uint32_t pw_lds[];
    __shared__ uint32_t s_w[W / 64 + 1];
    __shared__ uint32_t s_flip;
    __shared__ uint32_t s_tc[2 * (W / 64)];                  /* chains inside a round: a wavefront's composed map (t, c) */
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t b = b_first + blockIdx.x;
    const uint32_t x0 = b * B;
    const uint32_t x1 = nx - x0 < B ? nx : x0 + B;
    pw_ring<PACK> ring;
    if constexpr (PACK) {
        ring.lo = reinterpret_cast<uint16_t *>(pw_lds);
        ring.hi = pw_lds + ring_n / 2u;
    } else {
        ring.r = pw_lds;
    }
    if (tid == 0) s_flip = 0;
    const uint32_t *inb = in + (size_t)b * rs;
    PW_STAMP(8);
    /* PACK: entry i = q*W + tid lives in a register from the one coalesced read of the row to the ring */
    constexpr uint32_t EPT = PACK ? 65536u / W : 1u;        /* entries per thread, at most (sb <= 65535) */
    uint32_t val[EPT];
    if constexpr (PACK) {
#pragma unroll
        for (uint32_t q = 0; q < EPT; q++) { uint32_t tq = tid; asm volatile("" : "+v"(tq)); val[q] = inb[min(q * W + tq, sb - 1u)]; }
    }
    {
        /* the sweep is a function of the entry cells alone: the same cells as last time give the same xval[] and the gates
         * the maps of this iteration were built from -- nothing to do, no flip.  (The tail iterations change the cells
         * of a fraction of the blocks.)  Rows of loads in flight at a time: one load, one compare, one store at a time
         * was 64 round trips to HBM in a row, 130 K of a block's 1.2 M cycles -- and all there is to a skipped block. */
        uint32_t *pb = in_prev + (size_t)b * rs;
        bool diff = !have_prev;
        if constexpr (PACK) {
            constexpr uint32_t CB = 16;
#pragma unroll
            for (uint32_t q0 = 0; q0 < EPT; q0 += CB) {
                uint32_t pv[CB];
#pragma unroll
                for (uint32_t k = 0; k < CB; k++) { uint32_t tq = tid; asm volatile("" : "+v"(tq)); pv[k] = pb[min((q0 + k) * W + tq, sb - 1u)]; }
#pragma unroll
                for (uint32_t k = 0; k < CB; k++) {
                    uint32_t tq = tid; asm volatile("" : "+v"(tq)); const uint32_t i = (q0 + k) * W + tq;
                    if (i < sb && val[q0 + k] != pv[k]) { diff = true; pb[i] = val[q0 + k]; }
                }
            }
        } else {
            for (uint32_t i0 = tid; i0 < sb; i0 += 8u * W) {
                uint32_t a[8], pv[8];
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) { const uint32_t i = min(i0 + j * W, sb - 1u); a[j] = inb[i]; pv[j] = pb[i]; }
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) {
                    const uint32_t i = i0 + j * W;
                    if (i < sb && a[j] != pv[j]) { diff = true; pb[i] = a[j]; }
                }
            }
        }
        if (!__syncthreads_or(diff) && !(out_state && x1 == nx)) {
            for (uint32_t i = tid; i < (x1 - x0 + 63u) / 64u; i += W) gnew[(x0 >> 6) + i] = gold[(x0 >> 6) + i];
            if (tid == 0) gates_changed[b] = 0;
            return;
        }
    }

    if constexpr (PACK) {
        /* ---- ranks of the old entry values (see the header).  Entry i = q*W + tid lives in a register from the one
         *      coalesced read of the row to the ring: val[q] holds the value until its rank is known, then its code.
         *      Old values are positions below hi; most lie within a few windows of it, so they are ranked by COUNTING:
         *      pass r marks the values at distance [r*D, (r+1)*D) below hi in a bitmap of D bits and a value's rank in
         *      the pass is the number of set bits below its own (word popcounts, two levels of prefix sums).  The few
         *      that lie further back (priorities are handed down for ever) are sorted.  A bitonic sort of all old values
         *      (19 K of the 65535 on the S3 stream) took 300 us per block, twice the sweep proper. ---- */
        constexpr uint32_t BM_WORDS = 24576u, BM_BITS = BM_WORDS * 32u;     /* D = 786432 = 3 << 18 positions: 96 KB */
        constexpr uint32_t NPASS = 6u;                      /* 4.7 M positions back; beyond: the sorted tail */
        constexpr uint32_t NCO = BM_BITS / 1024u;           /* coarse prefix entries (1024 bits each) */
        uint32_t *bm = pw_lds;
        uint32_t *coarse = pw_lds + BM_WORDS;
        uint16_t *fine = reinterpret_cast<uint16_t *>(pw_lds + BM_WORDS + NCO);   /* per 256 bits, relative to coarse */
        uint32_t *skey = pw_lds;                            /* the tail sort comes first: the bitmap's space */
        uint32_t *cb = codes + (size_t)b * rs, *gv = gval + (size_t)b * rs;
        const bool all_old = b == 0 && ncarried != 0;       /* carried cells hold ranks of their own: none is "its own position" */
        const uint32_t hi = all_old ? sb : x0 + voff;       /* every old value is below hi */
        __shared__ uint32_t s_cls[NPASS + 2];               /* old values per pass; [NPASS] the tail */
        if (tid < NPASS + 2) s_cls[tid] = 0;
        __syncthreads();
        /* class of an entry: its bitmap pass, NPASS = tail, 7 = not old / beyond the row.  Counted in eight 8-bit fields
         * of one 64-bit word per thread (at most 64 entries a class) -- six compare-and-add pairs and a ballot with a
         * branch per entry made this loop 125 instructions an entry, 160 K of a block's 950 K cycles. */
        uint32_t c03 = 0, c47 = 0, olo = 0, ohi = 0;
#pragma unroll
        for (uint32_t q = 0; q < EPT; q++) {
            uint32_t tq = tid; asm volatile("" : "+v"(tq)); const uint32_t i = q * W + tq;
            const uint32_t v = val[q];
            const bool old = i < sb && (all_old || v != x0 + i + voff);
            const uint32_t u = hi - 1u - v;
            const uint32_t r = old ? min((u >> 18) / 3u, NPASS) : 7u;
            const uint32_t inc = 1u << (8u * (r & 3u));
            c03 += r < 4u ? inc : 0u;
            c47 += r < 4u ? 0u : inc;
            if (q < 32) olo |= old ? 1u << (q & 31u) : 0u;
            else ohi |= old ? 1u << (q & 31u) : 0u;
            asm volatile("" : "+v"(c03), "+v"(c47), "+v"(olo), "+v"(ohi));     /* (a chain, not a tree of 64 live terms) */
        }
        uint64_t oldm = ((uint64_t)ohi << 32) | olo;        /* bit q: entry q is old and has no rank yet */
        const uint64_t cnt = ((uint64_t)c47 << 32) | c03;
        const uint64_t old0 = oldm;                         /* bit q: entry q is old */
        {
            /* per wavefront: 16-bit fields (64 lanes x 64 entries), classes 0 2 4 6 and 1 3 5 7 */
            uint64_t ce = cnt & 0x00FF00FF00FF00FFull, co = (cnt >> 8) & 0x00FF00FF00FF00FFull;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                ce += (uint64_t)__shfl_xor((unsigned long long)ce, d, 64);
                co += (uint64_t)__shfl_xor((unsigned long long)co, d, 64);
            }
            if (lane <= NPASS) {
                const uint32_t t = (uint32_t)(((lane & 1u) ? co : ce) >> (16u * (lane >> 1))) & 0xFFFFu;
                if (t) atomicAdd(&s_cls[lane], t);
            }
        }
        __syncthreads();
        if (s_cls[NPASS]) {
            /* tail values into the sort buffer (any order does): rare -- priorities handed down for more than 4.7 M positions */
            uint32_t tot;
            uint32_t slot = pw_block_excl<W>((uint32_t)(cnt >> (8u * NPASS)) & 0xFFu, s_w, &tot);
#pragma unroll
            for (uint32_t q = 0; q < EPT; q++) {
                uint32_t tq = tid; asm volatile("" : "+v"(tq)); const uint32_t i = q * W + tq;
                if (((oldm >> q) & 1ull) && (hi - 1u - val[q]) / BM_BITS >= NPASS) {
                    if (slot < sort_cap) skey[slot] = val[q];
                    cb[i] = slot;                           /* only read back when the tail needs several passes */
                    slot++;
                }
            }
            __syncthreads();
        }
        const uint32_t Kt = s_cls[NPASS];
        PW_STAMP(11);
        /* ---- the tail: sorted, sort_cap keys a pass (one pass unless most of the window holds ancient priorities) ---- */
        if (Kt) {
            const bool one = Kt <= sort_cap;
            if (!one) {
#pragma unroll
                for (uint32_t q = 0; q < EPT; q++) {
                    uint32_t tq = tid; asm volatile("" : "+v"(tq)); const uint32_t i = q * W + tq;
                    if (((oldm >> q) & 1ull) && (hi - 1u - val[q]) / BM_BITS >= NPASS) gv[i] = 0;     /* rank accumulators */
                }
            }
            for (uint32_t c0 = 0; c0 < Kt; c0 += sort_cap) {
                const uint32_t m = min(Kt - c0, sort_cap);
                uint32_t P = 64;
                while (P < m) P <<= 1;
                if (c0) {
                    __syncthreads();
#pragma unroll
                    for (uint32_t q = 0; q < EPT; q++) {
                        uint32_t tq = tid; asm volatile("" : "+v"(tq)); const uint32_t i = q * W + tq;
                        if (((oldm >> q) & 1ull) && (hi - 1u - val[q]) / BM_BITS >= NPASS) {
                            const uint32_t sl = cb[i];
                            if (sl >= c0 && sl < c0 + m) skey[sl - c0] = val[q];
                        }
                    }
                }
                for (uint32_t i = m + tid; i < P; i += W) skey[i] = 0xFFFFFFFFu;
                __syncthreads();
                for (uint32_t k = 2; k <= P && !(probe & 2u); k <<= 1) {
                    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                        for (uint32_t q = tid; q < P / 2u; q += W) {
                            const uint32_t a = ((q & ~(j - 1u)) << 1) | (q & (j - 1u)), c = a | j;
                            const uint32_t ka = skey[a], kc = skey[c];
                            const bool up = (a & k) == 0;
                            if ((ka > kc) == up) { skey[a] = kc; skey[c] = ka; }
                        }
                        __syncthreads();
                    }
                }
#pragma unroll
                for (uint32_t q = 0; q < EPT; q++) {
                    uint32_t tq = tid; asm volatile("" : "+v"(tq)); const uint32_t i = q * W + tq;
                    const uint32_t v = val[q];
                    if (((oldm >> q) & 1ull) && (hi - 1u - v) / BM_BITS >= NPASS) {
                        uint32_t pos = 0;
                        for (uint32_t st = P >> 1; st; st >>= 1)
                            if (skey[pos + st - 1u] < v) pos += st;
                        pos += skey[pos] < v ? 1u : 0u;         /* a full pass of keys that are all below v (v sits in another pass) */
                        if (one) { gv[pos] = v; val[q] = pos; oldm &= ~(1ull << q); }
                        else gv[i] += pos;
                    }
                }
            }
            if (!one) {
                __syncthreads();
                uint32_t rk[EPT / 8];                           /* (register pressure: eight entries at a time) */
                for (uint32_t q0 = 0; q0 < EPT; q0 += EPT / 8) {
#pragma unroll
                    for (uint32_t k = 0; k < EPT / 8; k++) {
                        uint32_t tq = tid; asm volatile("" : "+v"(tq)); const uint32_t q = q0 + k, i = q * W + tq;
                        rk[k] = ((oldm >> q) & 1ull) && (hi - 1u - val[q]) / BM_BITS >= NPASS ? gv[i] : PW_NONE;
                    }
                    __syncthreads();                            /* (every accumulator of the slice is read before a table entry lands on one) */
#pragma unroll
                    for (uint32_t k = 0; k < EPT / 8; k++) {
                        const uint32_t q = q0 + k;
                        if (rk[k] != PW_NONE) { cb[q * W + tid] = rk[k]; }
                    }
                }
                __syncthreads();
#pragma unroll
                for (uint32_t q = 0; q < EPT; q++) {
                    uint32_t tq = tid; asm volatile("" : "+v"(tq)); const uint32_t i = q * W + tq;
                    if (((oldm >> q) & 1ull) && (hi - 1u - val[q]) / BM_BITS >= NPASS) {
                        const uint32_t rkq = cb[i];
                        gv[rkq] = val[q];
                        val[q] = rkq;
                        oldm &= ~(1ull << q);
                    }
                }
            }
            __syncthreads();
        }
        PW_STAMP(12);
        /* ---- the bitmap passes, oldest first: a pass's ranks follow everything older ---- */
        uint32_t below = Kt;
        for (int r = (int)NPASS - 1; r >= 0; r--) {
            const uint32_t Kr = s_cls[r];
            if (!Kr) continue;
            for (uint32_t i = tid; i < BM_WORDS; i += W) bm[i] = 0;
            __syncthreads();
#pragma unroll
            for (uint32_t q = 0; q < EPT; q++) {
                const uint32_t u = hi - 1u - val[q];
                if (((oldm >> q) & 1ull) && u / BM_BITS == (uint32_t)r) {
                    const uint32_t bit = BM_BITS - 1u - (u - (uint32_t)r * BM_BITS);
                    atomicOr(&bm[bit >> 5], 1u << (bit & 31u));
                }
            }
            __syncthreads();
            uint32_t tot = 0;
            if (tid < NCO) {
#pragma unroll
                for (uint32_t w = 0; w < 32; w++) {
                    if ((w & 7u) == 0) fine[tid * 4u + (w >> 3)] = (uint16_t)tot;
                    tot += (uint32_t)__popc(bm[tid * 32u + w]);
                }
            }
            uint32_t all;
            const uint32_t ex = pw_block_excl<W>(tot, s_w, &all);
            if (tid < NCO) coarse[tid] = ex;
            __syncthreads();
#pragma unroll
            for (uint32_t q = 0; q < EPT; q++) {
                const uint32_t v = val[q];
                const uint32_t u = hi - 1u - v;
                if (((oldm >> q) & 1ull) && u / BM_BITS == (uint32_t)r) {
                    const uint32_t bit = BM_BITS - 1u - (u - (uint32_t)r * BM_BITS);
                    uint32_t rk = below + coarse[bit >> 10] + (uint32_t)fine[bit >> 8];
                    for (uint32_t w = (bit >> 8) * 8u; w < (bit >> 5); w++) rk += (uint32_t)__popc(bm[w]);
                    rk += (uint32_t)__popc(bm[bit >> 5] & ((1u << (bit & 31u)) - 1u));
                    gv[rk] = v;
                    val[q] = rk;
                    oldm &= ~(1ull << q);
                }
            }
            below += Kr;
            __syncthreads();
        }
        PW_STAMP(13);
        if (blockIdx.x == 0 && tid == 0) PW_NOTE(14, Kt);
        /* ---- codes into the ring: own positions are sb + i (the plane is filled as if every cell held its own, then
         *      the old entries clear their bits: ranks are below 2^16) ---- */
        for (uint32_t wd = tid; wd < ring_n / 16u; wd += W) ring.hi[wd] = pw_plane_word(sb + wd * 16u);
        __syncthreads();
        /* (a plane word is sixteen consecutive cells = a DPP row of one entry index: the row's last lane writes it -- an
         * atomic per old entry made sixteen lanes queue on every word) */
#pragma unroll
        for (uint32_t q = 0; q < EPT; q++) {
            uint32_t tq = tid; asm volatile("" : "+v"(tq)); const uint32_t i = q * W + tq;
            const bool own = !((old0 >> q) & 1ull);             /* (beyond the row: own) */
            const uint32_t code = own ? sb + i : val[q];
            if (i < sb) ring.lo[i] = (uint16_t)code;
            uint32_t hv = ((code >> 16) & 3u) << ((tid & 15u) * 2u);
            hv |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hv, 0x111, 0xF, 0xF, false);      /* row_shr:1 */
            hv |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hv, 0x112, 0xF, 0xF, false);      /* row_shr:2 */
            hv |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hv, 0x114, 0xF, 0xF, false);      /* row_shr:4 */
            hv |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hv, 0x118, 0xF, 0xF, false);      /* row_shr:8 */
            if ((tid & 15u) == 15u && i < ring_n) ring.hi[i >> 4] = hv;
        }
        for (uint32_t r = sb + tid; r < ring_n; r += W) ring.lo[r] = (uint16_t)(sb + r);
    } else {
        for (uint32_t r = tid; r < ring_n; r += W) ring.r[r] = r < sb ? inb[r] : x0 + r + voff;
    }
    __syncthreads();

    PW_STAMP(9);
    constexpr uint32_t NW = W / 64;
    static_assert(NW == 4 || NW == 16, "the round counts' prefix runs along a DPP row");
    const uint32_t wave_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    const uint32_t xlast = x1 - 1u;
    uint32_t off = 0;                                       /* ring slot of cell xg */
    uint32_t nflip = 0;
    /* the operands of SG groups are fetched together, one super-group ahead (unconditional, clamped loads: see
     * k_prio_fwd): a group is about as long as a round trip to HBM under load, one group ahead was not enough */
    constexpr int SG = 4;
    uint32_t v[SG], vn[SG];
    uint64_t rmw[SG], rmn[SG], gow[SG], gon[SG], cmw[SG], cmn[SG];
    auto fetch = [&](uint32_t xs, uint32_t (&vv)[SG], uint64_t (&rr)[SG], uint64_t (&gg)[SG], uint64_t (&cc)[SG]) {
#pragma unroll
        for (int k = 0; k < SG; k++) {
            const uint32_t xk = xs + (uint32_t)k * W;
            vv[k] = ps[min(xk + tid, xlast)];
            rr[k] = rmask[min(xk + 64u * (lane & (NW - 1u)), xlast) >> 6];
            gg[k] = gold[min(xk + 64u * wave, xlast) >> 6];
            cc[k] = cmask[min(xk + 64u * wave_s, xlast) >> 6];          /* (a wave-uniform address: a scalar load) */
        }
    };
    fetch(x0, v, rmw, gow, cmw);
    /* One group of W steps.  FULL: every lane has a step -- and every vector-memory operation of the group is
     * unconditional: `s_waitcnt vmcnt` counts loads and stores in ONE order, so a wait for a prefetched operand behind a
     * store whose issue the compiler cannot count (a branch around it) becomes vmcnt(0) and pays the store's round trip
     * (microseconds) in every group.  Until round 5 the look-up of a handed-over old rank in gval[] sat in this loop,
     * one group deferred, behind exactly such a wait: 7400 cycles per group, of which the rounds are 2000.  Now an old
     * rank r leaves as the marker PW_RANK0 - r (no priority is that high: positions end at LZ77X_MAX_N, voff <= sb) and
     * every thread translates the markers it wrote itself when the block is through. */
    auto group = [&](auto full_tag, uint32_t xg, uint32_t vk, uint64_t rmk, uint64_t gok, uint64_t cmk /* my wavefront's chain links */) {
        constexpr bool FULL = decltype(full_tag)::value;
        const uint32_t x = xg + tid;
        const uint32_t vv = FULL || x < x1 ? vk : 0u;
        const uint32_t p = vv & 0xFFFFu, s = vv >> 16;
        const bool has = p && s;
        /* my round: round starts at or before me, minus one; rounds of the group.  Lane l holds the mask word of
         * wavefront l mod NW: the counts' prefix sums run along a row of 16 lanes (DPP), this wavefront's word and the
         * rounds before it are two scalar reads -- the loop over NW readlanes with a vector compare against the
         * wavefront number was 100 of a group's 320 instructions, sixteen wavefronts over. */
        /* (bit 0 of the group's first word: cleared = some round of the group holds a chain) */
        const bool chains = ((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rmk, 0) & 1u) == 0u;
        rmk |= (lane & (NW - 1u)) == 0u ? 1ull : 0ull;
        const uint32_t pc = (uint32_t)__popcll(rmk);
        uint32_t incl = pc;
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false);      /* row_shr:1 */
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false);      /* row_shr:2 */
        if constexpr (NW > 4) {
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false);  /* row_shr:4 */
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false);  /* row_shr:8 */
        }
        const uint32_t nr = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)NW - 1);
        const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)(incl - pc), (int)wave_s);
        const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rmk, (int)wave_s);
        const uint32_t mhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rmk >> 32), (int)wave_s);
        /* bits 0 .. lane of the word = bit 0 + the bits below the lane of the word shifted down by one */
        const uint32_t myr = before + (mlo & 1u) - 1u +
                             __builtin_amdgcn_mbcnt_hi(mhi >> 1, __builtin_amdgcn_mbcnt_lo((mlo >> 1) | (mhi << 31), 0u));
        uint32_t ix = off + tid;
        ix -= ix >= ring_n ? ring_n : 0u;
        uint32_t ip = ix + p;
        ip -= ip >= ring_n ? ring_n : 0u;
        uint32_t is = ix + s;
        is -= is >= ring_n ? ring_n : 0u;
        uint32_t ng = 0, out = PW_NONE;
        if (__builtin_expect(chains, 0)) {
            /* Round 5 (the run cliff of the large windows: runs of 1000-2000 equal bytes swept 7-13x slower than text).  A step
             * whose own cell the step before it writes in the same round is a link: along such a chain the step is the
             * threshold map f(a) = a < min(w, c) ? a : c on ONE value, so the round resolves its chains with a scan of the
             * maps -- along the wavefront by DPP, across the sixteen wavefronts through 16 pairs in LDS, between the two
             * barriers a round has anyway -- instead of a round (two barriers) per link.  A step that is no link enters as a
             * constant map, which cuts off whatever lies to its left. */
            const bool clink = (cmk >> lane) & 1ull;
            for (uint32_t r = 0; r < nr; r++) {
                const bool mine = has && myr == r;
                const bool linked = mine && clink;
                const bool wl = __ballot(linked) != 0ull;       /* links among my wavefront's steps of this round? */
                uint32_t a = 0, w = 0, sv = 0;
                if (mine) { a = ring.rd(ix); w = ring.rd(ip); sv = ring.rd(is); }
                pw_tc f;
                f.t = min(w, sv);
                f.c = sv;
                if (!linked) { f.c = a < f.t ? a : f.c; f.t = 0u; }     /* its own cell is what the ring holds: a constant */
                if (!mine) { f.t = 0u; f.c = 0u; }
                if (wl) f = pw_tc_scan(f);                      /* (without a link every map is a constant: the scan would change nothing) */
                if (lane == 63u) { s_tc[2u * wave] = f.t; s_tc[2u * wave + 1u] = f.c; }
                pw_lds_barrier();                               /* every read of the round is done; the wavefronts' maps are out */
                if (wl) {
                    uint32_t cin = 0;                           /* what the last step of the wavefront before mine leaves behind */
                    for (uint32_t ww = 0; ww < wave_s; ww++) {
                        const uint32_t tt = s_tc[2u * ww], tcv = s_tc[2u * ww + 1u];
                        cin = cin < tt ? cin : tcv;
                    }
                    const uint32_t mine_out = cin < f.t ? cin : f.c;   /* the cell my step leaves behind */
                    uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine_out, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
                    if (lane == 0u) left = cin;
                    if (linked) a = left;
                }
                bool hand = false;
                if (mine) {
                    const bool gate = a < w;
                    ng = gate ? 1u : 0u;
                    hand = gate && a < sv;
                }
                if (hand) { ring.wr(is, a, sv); out = a; }
                pw_lds_barrier();
            }
        } else
        for (uint32_t r = 0; r < nr; r++) {
            /* a round in two halves: every step of the round reads, then every step writes.  The round masks only
             * split where a step READS what an earlier step of the round writes; a later step of the round may well
             * write what an earlier one reads (its successor cell = the other's predecessor cell), and across
             * wavefronts nothing but a barrier orders that write behind the read (inside one wavefront the lanes run
             * in lockstep: k_prio_fwd) */
            const bool mine = has && myr == r;
            uint32_t a = 0, sv = 0;
            bool hand = false;
            if (mine) {
                a = ring.rd(ix);
                const uint32_t w = ring.rd(ip);
                sv = ring.rd(is);
                const bool gate = a < w;
                ng = gate ? 1u : 0u;                        /* the gate: x's predecessor hangs below x */
                hand = gate && a < sv;
            }
            pw_lds_barrier();
            if (hand) { ring.wr(is, a, sv); out = a; }      /* tree.c:202-231: S takes x's place */
            pw_lds_barrier();
        }
        const uint64_t gnb = __ballot(ng != 0u);
        const bool wave_in = FULL || xg + 64u * wave < x1;
        nflip += wave_in ? (uint32_t)__popcll(gnb ^ gok) : 0u;
        if (lane == 0 && wave_in) gnew[(xg >> 6) + wave] = gnb;
        uint32_t o = out;
        if constexpr (PACK) o = out == PW_NONE ? PW_NONE : (out < sb ? PW_RANK0 - out : x0 + (out - sb) + voff);
        if (FULL || x < x1) xval[x] = o;
        /* cell xg + ring_n + tid becomes live with the next group; its slot held cell xg + tid (the lanes past the
         * last step keep their cells: out_state) */
        if constexpr (PACK) {
            const uint32_t nc = sb + (xg + ring_n + tid - x0);
            if (FULL) {
                /* whole group: the plane a word (16 cells, 16 consecutive codes) at a time -- an exclusive or per cell
                 * made 16 lanes queue on every word */
                ring.lo[ix] = (uint16_t)nc;
                if ((tid & 15u) == 0) ring.hi[ix >> 4] = pw_plane_word(nc);
            } else if (x < x1) ring.put(ix, nc);
        } else {
            if (FULL || x < x1) ring.put(ix, xg + ring_n + tid + voff);
        }
        off += W;
        off -= off >= ring_n ? ring_n : 0u;
        pw_lds_barrier();
    };
    uint32_t xs = x0;
    if (!(probe & 1u)) {
        for (; x1 - xs >= SG * W; xs += SG * W) {
            fetch(xs + SG * W, vn, rmn, gon, cmn);
#pragma unroll
            for (int k = 0; k < SG; k++) group(std::true_type{}, xs + (uint32_t)k * W, v[k], rmw[k], gow[k], cmw[k]);
#pragma unroll
            for (int k = 0; k < SG; k++) { v[k] = vn[k]; rmw[k] = rmn[k]; gow[k] = gon[k]; cmw[k] = cmn[k]; }
        }
#pragma unroll
        for (int k = 0; k < SG; k++)
            if (xs + (uint32_t)k * W < x1) group(std::false_type{}, xs + (uint32_t)k * W, v[k], rmw[k], gow[k], cmw[k]);   /* (workgroup-uniform) */
    }
    if constexpr (PACK) {
        /* the markers of this thread's own stores (x = x0 + tid mod W): rank -> value */
        const uint32_t *gvb = gval + (size_t)b * rs;
        constexpr int TB = 16;
        for (uint32_t xb = x0 + tid; xb < x1; xb += TB * W) {
            uint32_t c[TB];
#pragma unroll
            for (int j = 0; j < TB; j++) c[j] = xval[min(xb + (uint32_t)j * W, xlast)];
#pragma unroll
            for (int j = 0; j < TB; j++) {
                const bool mk = xb + (uint32_t)j * W < x1 && c[j] != PW_NONE && c[j] > PW_RANK0 - 65536u;
                c[j] = mk ? gvb[PW_RANK0 - c[j]] : PW_NONE;
            }
#pragma unroll
            for (int j = 0; j < TB; j++)
                if (c[j] != PW_NONE) xval[xb + (uint32_t)j * W] = c[j];
        }
    }
    PW_STAMP(10);
    if (out_state && x1 == nx) {
        /* cells nx .. nx+sb-1, what the next segment of a long input starts from: the ring holds the cells [x1, x1 + ring_n) */
        for (uint32_t i = tid; i < sb; i += W) {
            const uint32_t c = ring.rd((x1 - x0 + i) % ring_n);
            if constexpr (PACK) out_state[i] = c < sb ? gval[(size_t)b * rs + c] : x0 + (c - sb) + voff;
            else out_state[i] = c;
        }
    }
    if (lane == 0 && nflip) atomicAdd(&s_flip, nflip);
    __syncthreads();
    if (tid == 0) gates_changed[b] = s_flip ? 1u : 0u;
    if (tid == 0 && s_flip) {
        atomicAdd(&summary[0], s_flip);
        atomicMin(&summary[1], b);
        atomicMax(&summary[2], b);
    }
}

/* ------------------------------------------------------------------ backward sweep --- */

/* dest[b][i]: the exit cell (relative to the block's end x1) that the chain of open gates starting at entry cell b*B+i
 * reaches, or DEAD when it ends inside the block; loc[b][d]: the lowest position of the block whose chain reaches exit
 * cell d.
 *
 * Round 5: two phases a block.  Until then every step whose chain left the block lowered loc[d] in HBM with an atomic
 * (loc, 32 bits x sb, does not fit beside the ring), deduplicated per group through a table in LDS -- but a group of 4096
 * steps reaches ~1400 different exit cells, 80-110 K atomics a block went out (counted: LZ77X_PW_DEBUG), and the L2 does
 * about 50 G of them a second: 0.7 of a block's 1.0-1.5 M cycles.  Now phase 1 (the backward sweep) only stores every
 * step's exit cell, 16 bits, in a scratch row of the workgroup (`destx`), and phase 2 -- the ring is no longer needed --
 * has the LDS for loc itself, half the exit cells at a time: initialise, one pass of LDS atomics over the row, write out.
 * The grid is persistent (a workgroup takes the blocks blockIdx.x, + gridDim.x, ...): a scratch row per workgroup. */
template <int W>
__global__ __launch_bounds__(W) void k_pw_back(const uint32_t *__restrict__ ps, uint32_t nx, uint32_t sb, uint32_t rs /* row stride of dest / loc */, uint32_t B, uint32_t ring_n,
                                               uint32_t b_first, uint32_t nb, const uint64_t *__restrict__ gates, uint16_t *__restrict__ dest,
                                               uint32_t *__restrict__ loc, uint32_t voff, uint32_t ncarried,
                                               const uint32_t *__restrict__ gates_changed, uint16_t *__restrict__ destx_all)
{
    constexpr int SPT = 4;                                             /* steps a thread: a group is GW = SPT * W steps */
    constexpr uint32_t GW = (uint32_t)SPT * W;
    extern __shared__ uint32_t pw_lds[];
    __shared__ uint16_t s_d[GW];
    __shared__ int16_t s_p[GW];
    __shared__ uint32_t s_any[2];
    uint16_t *dr = reinterpret_cast<uint16_t *>(pw_lds);               /* ring_n entries */
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    /* (a group must fit the ring: small windows under LZ77X_PRIO_WIDE take fewer steps a thread) */
    const uint32_t spt = min((uint32_t)SPT, ring_n / (uint32_t)W), gwr = spt * (uint32_t)W;
    uint16_t *destx = destx_all + (size_t)blockIdx.x * ((size_t)B + 16u);
    uint32_t anyk = 0;                                                 /* parity of the "anything pending" flag in use */
    if (tid < 2) s_any[tid] = 0;
    for (uint32_t bi = blockIdx.x; bi < nb; bi += gridDim.x) {
        const uint32_t b = b_first + bi;
        if (gates_changed && !gates_changed[b]) continue;              /* the block's map of the last iteration still holds (uniform) */
        const uint32_t x0 = b * B;
        const uint32_t x1 = nx - x0 < B ? nx : x0 + B;
        uint32_t *lb = loc + (size_t)b * rs;
        PW_STAMP(0);
        __syncthreads();                                               /* (the block before is through with the LDS) */
        for (uint32_t i = tid; i < ring_n; i += W) dr[i] = (uint16_t)PW_DEAD;
        __syncthreads();
        PW_STAMP(1);
        /* ---- phase 1: groups of GW steps from the last to the first, a thread SPT of them (before: W steps a group, one
         *      hop a round).  The chains inside a group are followed by pointer jumping through LDS, two hops a round: a
         *      pointer triples its reach. ---- */
        const uint32_t ngr = (x1 - x0 + gwr - 1u) / gwr;
        const uint32_t xlast = x1 - 1u;
        uint32_t off_run = (uint32_t)(((uint64_t)ngr * gwr) % ring_n);
        uint32_t v[SPT], vn[SPT];
        uint64_t gw[SPT], gwn[SPT];
        auto fetch = [&](int32_t gi, uint32_t (&vv)[SPT], uint64_t (&gg)[SPT]) {       /* unconditional, clamped loads (see k_prio_fwd) */
            const uint32_t xs = x0 + (uint32_t)(gi < 0 ? 0 : gi) * gwr;
#pragma unroll
            for (int k = 0; k < SPT; k++) {
                vv[k] = ps[min(xs + (uint32_t)k * W + tid, xlast)];
                gg[k] = gates[min(xs + (uint32_t)k * W + 64u * wave, xlast) >> 6];
            }
        };
        fetch((int32_t)ngr - 1, v, gw);
        for (int32_t gi = (int32_t)ngr - 1; gi >= 0; gi--) {
            fetch(gi - 1, vn, gwn);
            const uint32_t xg = x0 + (uint32_t)gi * gwr;
            off_run = off_run >= gwr ? off_run - gwr : off_run + ring_n - gwr;                /* = (xg - x0) % ring_n */
            uint32_t d[SPT];
            int ptr[SPT];
#pragma unroll
            for (int k = 0; k < SPT; k++) {
                const uint32_t li = (uint32_t)k * W + tid, x = xg + li;
                const bool valid = x < x1 && (uint32_t)k < spt;
                const uint32_t sk = (valid ? v[k] : 0u) >> 16;
                const bool gate = valid && xg + (uint32_t)k * W + 64u * wave < x1 && ((gw[k] >> lane) & 1ull);
                const uint32_t t = x + sk;
                uint32_t it = valid ? off_run + li + (gate ? sk : 0u) : 0u;
                it -= it >= ring_n ? ring_n : 0u;
                it -= it >= ring_n ? ring_n : 0u;
                const uint32_t dring = dr[it];
                const bool past = t >= x1, near = t < xg + gwr;
                d[k] = !gate ? (uint32_t)PW_DEAD : past ? t - x1 : near ? (uint32_t)PW_DEAD : dring;
                ptr[k] = gate && !past && near ? (int)(t - xg) : -1;
            }
            /* chains inside the group (pointers only lead to later steps).  s_any[k]: some step still has a pointer; the
             * flag of the other parity is cleared while this one is in use (LDS-only barriers: the prefetched operands
             * stay in flight) */
            for (;;) {
                bool pend = false;
#pragma unroll
                for (int k = 0; k < SPT; k++) {
                    s_d[(uint32_t)k * W + tid] = (uint16_t)d[k];
                    s_p[(uint32_t)k * W + tid] = (int16_t)ptr[k];
                    pend = pend || ptr[k] >= 0;
                }
                if (__ballot(pend) && lane == 0) s_any[anyk] = 1u;
                pw_lds_barrier();
                const bool any = s_any[anyk] != 0u;
                if (tid == 0) s_any[anyk ^ 1u] = 0u;
                anyk ^= 1u;
                if (!any) break;
#pragma unroll
                for (int k = 0; k < SPT; k++) {
                    if (ptr[k] >= 0) {
                        const uint32_t d1 = s_d[ptr[k]];
                        const int p1 = s_p[ptr[k]];
                        if (p1 < 0) { d[k] = d1; ptr[k] = -1; }
                        else {
                            const uint32_t d2 = s_d[p1];
                            const int p2 = s_p[p1];
                            if (p2 < 0) { d[k] = d2; ptr[k] = -1; } else ptr[k] = p2;
                        }
                    }
                }
                pw_lds_barrier();
            }
            /* (every read of the ring by this group came before the loop's first barrier) */
#pragma unroll
            for (int k = 0; k < SPT; k++) {
                const uint32_t li = (uint32_t)k * W + tid;
                uint32_t ix = off_run + li;
                ix -= ix >= ring_n ? ring_n : 0u;
                if (xg + li < x1 && (uint32_t)k < spt) {
                    dr[ix] = (uint16_t)d[k];
                    destx[xg - x0 + li] = (uint16_t)d[k];
                }
            }
            pw_lds_barrier();
#pragma unroll
            for (int k = 0; k < SPT; k++) { v[k] = vn[k]; gw[k] = gwn[k]; }
        }
        PW_STAMP(2);
        for (uint32_t i = tid; i < sb; i += W) {
            /* an entry cell that is not evicted inside a (short, last) block is still live at its end */
            dest[(size_t)b * rs + i] = x0 + i < x1 ? dr[i] : (uint16_t)(i - (x1 - x0));
        }
        /* ---- phase 2: loc.  What reaches exit cell x1+d when nothing older comes in is its own priority -- unless it is a
         *      carried cell (a short first block of a later segment: its value comes in through in[0]); x's own priority
         *      reaches its exit cell -- unless x is a carried cell. ---- */
        pw_fence_wg();                                                 /* the scratch row is written */
        __syncthreads();
        {
            const uint32_t nst = x1 - x0;
            const uint32_t CH = (ring_n / 2u) & ~3u;                   /* exit cells a pass: the ring's space, 32 bits a cell */
            const pw_u16x8 *dx8 = reinterpret_cast<const pw_u16x8 *>(destx);
            for (uint32_t c0 = 0; c0 < sb; c0 += CH) {
                const uint32_t c1 = min(c0 + CH, sb);
                for (uint32_t i = tid; i < c1 - c0; i += W) pw_lds[i] = x1 + c0 + i >= ncarried ? x1 + c0 + i + voff : PW_NONE;
                pw_lds_barrier();
                for (uint32_t q0 = 0; q0 * 8u < nst; q0 += 4u * W) {
                    pw_u16x8 dq[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; u++) { uint32_t tq = tid; asm volatile("" : "+v"(tq)); dq[u] = dx8[min(q0 + u * W + tq, (nst - 1u) / 8u)]; }
#pragma unroll
                    for (uint32_t u = 0; u < 4; u++) {
                        const uint32_t l0 = (q0 + u * W + tid) * 8u;
#pragma unroll
                        for (uint32_t j = 0; j < 8; j++) {
                            const uint32_t dj = dq[u][j], li = l0 + j;
                            if (li < nst && dj - c0 < c1 - c0 && dj != PW_DEAD && x0 + li >= ncarried) atomicMin(&pw_lds[dj - c0], x0 + li + voff);
                        }
                    }
                }
                pw_lds_barrier();
                for (uint32_t i = tid; i < c1 - c0; i += W) lb[c0 + i] = pw_lds[i];
                pw_lds_barrier();
            }
        }
        PW_STAMP(3);
        if (blockIdx.x == 0 && threadIdx.x == 0) PW_NOTE(4, ngr);
    }
}

/* ------------------------------------------------------------------ boundary scan ---- */

/* in[j+1] = F_j(in[j]),  F_j(v)[d] = min(loc_j[d], min{ v[c] : dest_j[c] = d }),  groups of G maps as in k_prio_scan_*:
 * compose each group's maps, run the group maps in sequence, replay every group.  A vector of sb priorities does not fit
 * LDS twice, so it lives in REGISTERS: one workgroup of 512 threads per group, thread t keeps the cells c = q*512 + t
 * (128 values), and a map is applied through LDS in two halves of the destination range -- cells with a live value lower
 * acc[dest] (LDS atomics), every thread then takes its own cells back.  Nothing of the running vector goes through HBM
 * (the rows are only written out, for the sweeps), and no global atomic is involved: the first versions applied a map with
 * global atomicMin -- many cells share a destination (the chains of a block merge), and same-address atomics queue in
 * the L2 at ~100 cycles each: 2-5 ms per launch. */
#define PW_SCAN_T 512u                  /* 512 threads: 256 VGPRs each -- the 128 cells of a thread plus a batch of operands
                                           (1024 threads x 64 cells spilled 800 registers) */
#define PW_SCAN_EPT 128u
#define PW_SCAN_HALF 32768u

/* One map applied by a workgroup: out[d] = min(lj[d], min{ cur[c] : dj[c] = d }) for d < sb.  The vectors are rows in
 * HBM (the one just written sits in the L2 of the workgroup's XCD); LDS holds acc[] for half of the destination range
 * at a time: it starts as the map's loc, the cells with a live value lower their destination (LDS atomics), the new
 * row is written from it.  `cur` null: nothing comes in (the first map of a composition).
 * An earlier version kept the vector in registers (128 cells a thread, the destinations re-read or packed beside them):
 * fully unrolled over a register array it spilled, waited on vmcnt 494 times a step and took 65 us a map; streaming
 * the rows costs 1.3 MB of traffic a map instead of 0.8 and is several times faster. */
#define PW_MAP_T 1024u
#define PW_MAP_HALF 32768u

/* (rows are 16-byte aligned: four cells a lane per load -- a workgroup's rate is its bytes in flight over the latency,
 * and with one cell a lane a map took 54 us) */



/* group g = blockIdx.x applies its maps m0 .. m1-1 (map number m is row `first + m` of dest / loc) in sequence.
 * REPLAY: from row g of vin; row `first + m + 1` of `v` receives the vector after map m.
 * else (compose, loc half): from "nothing" (the identity map); row g of `v` receives what the group's maps send to each
 * exit cell from inside the group; the running vector alternates between two rows of `tmp` (2 rows a group).
 *
 * One map: out[d] = min(loc[d], min{cur[c] : dest[c] = d}), the destinations in two halves of 32 K cells through LDS
 * (`acc`, 128 KB).  The maps of a group are a chain, and until round 5 every link was six round trips to memory in a row
 * (loc half -> LDS, dest + cur, the result out, a fence; twice) -- 30 us a map, 12 maps a launch, 5 launches an iteration.
 * Now only the chain's own data waits: a thread reads back the quads of `cur` it stored itself (program order, no fence),
 * loc and dest do not depend on the chain and are in registers one phase / one map ahead, and the thread that reads a
 * slot of `acc` out puts the next phase's loc value in its place (no barrier between the two). */
template <bool REPLAY>
__global__ __launch_bounds__(PW_MAP_T) void k_pw_maps(const uint16_t *__restrict__ dest, const uint32_t *__restrict__ loc, uint32_t sb, uint32_t rs, uint32_t first,
                                                      uint32_t nmaps, uint32_t G, const uint32_t *vin, uint32_t *v, uint32_t *tmp)
{
    extern __shared__ uint32_t pw_lds[];
    const uint32_t g = blockIdx.x, tid = threadIdx.x;
    const uint32_t m0 = g * G, m1 = min(m0 + G, nmaps);
    if (m0 >= m1) return;
    constexpr uint32_t HQ = PW_MAP_HALF / 4u, NK = HQ / PW_MAP_T;       /* quads a half, quads a thread and half */
    const uint32_t sb4 = (sb + 3u) / 4u;                    /* quads of cells; the rows are padded to a multiple of 8 cells */
    uint32_t *acc = pw_lds;
    pw_u32x4 *acc4 = reinterpret_cast<pw_u32x4 *>(pw_lds);
    const uint32_t *cur = REPLAY ? vin + (size_t)g * rs : nullptr;
    pw_u32x4 lq[NK];                                        /* loc of the next phase (a phase = one half of one map) */
    pw_u16x4 dq[2 * NK];                                    /* dest of the map at hand */
    auto load_loc = [&](size_t j, uint32_t h) {
        const pw_u32x4 *lj4 = reinterpret_cast<const pw_u32x4 *>(loc + j * rs);
#pragma unroll
        for (uint32_t k = 0; k < NK; k++) { uint32_t tq = tid; asm volatile("" : "+v"(tq)); lq[k] = lj4[min(h * HQ + tq + k * PW_MAP_T, sb4 - 1u)]; }
    };
    auto load_dest = [&](size_t j) {
        const pw_u16x4 *dj4 = reinterpret_cast<const pw_u16x4 *>(dest + j * rs);
#pragma unroll
        for (uint32_t k = 0; k < 2 * NK; k++) { uint32_t tq = tid; asm volatile("" : "+v"(tq)); dq[k] = dj4[min(tq + k * PW_MAP_T, sb4 - 1u)]; }
    };
    load_loc((size_t)first + m0, 0);
    if (cur) load_dest((size_t)first + m0);
#pragma unroll
    for (uint32_t k = 0; k < NK; k++) acc4[tid + k * PW_MAP_T] = lq[k];
    for (uint32_t m = m0; m < m1; m++) {
        const size_t j = (size_t)first + m;
        uint32_t *out;
        if constexpr (REPLAY) out = v + (j + 1) * rs;
        else out = m + 1u == m1 ? v + (size_t)g * rs : tmp + ((size_t)2 * g + ((m - m0) & 1u)) * rs;
        pw_u32x4 *out4 = reinterpret_cast<pw_u32x4 *>(out);
        const pw_u32x4 *cur4 = reinterpret_cast<const pw_u32x4 *>(cur);
        for (uint32_t h = 0; h < 2; h++) {
            const bool more = h == 0 || m + 1u < m1;
            if (more) load_loc(h == 0 ? j : j + 1, h ^ 1u);
            pw_lds_barrier();                               /* acc holds the loc half */
            if (cur) {
                const uint32_t base = h * PW_MAP_HALF;
#pragma unroll
                for (uint32_t bt = 0; bt < 2; bt++) {
                    pw_u32x4 cv[NK];
#pragma unroll
                    for (uint32_t k = 0; k < NK; k++) { uint32_t tq = tid; asm volatile("" : "+v"(tq)); cv[k] = cur4[min(tq + (bt * NK + k) * PW_MAP_T, sb4 - 1u)]; }
#pragma unroll
                    for (uint32_t k = 0; k < NK; k++) {
                        const uint32_t q = tid + (bt * NK + k) * PW_MAP_T;
                        const pw_u16x4 d = dq[bt * NK + k];
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const uint32_t dk = d[c], vk = cv[k][c];
                            if (4u * q + (uint32_t)c < sb && dk != PW_DEAD && (dk >> 15) == h && vk != PW_NONE) atomicMin(&acc[dk - base], vk);
                        }
                    }
                }
            }
            if (h == 1 && m + 1u < m1) load_dest(j + 1);   /* (dq is free: the map's last atomics are out) */
            pw_lds_barrier();                               /* the half is complete */
            const uint32_t q1 = min((h + 1u) * HQ, sb4);
#pragma unroll
            for (uint32_t k = 0; k < NK; k++) {
                uint32_t tq = tid; asm volatile("" : "+v"(tq));
                const uint32_t sl = tq + k * PW_MAP_T, q = h * HQ + sl;
                const pw_u32x4 x = acc4[sl];
                if (more) acc4[sl] = lq[k];
                if (q < q1) out4[q] = x;
            }
        }
        cur = out;
    }
}

/* the dest half of a group's composed map: gdest[g][c] = the exit cell entry cell c reaches through all the group's
 * maps (or DEAD): 16 bits a cell in registers, one gather per map */
__global__ __launch_bounds__(PW_SCAN_T) void k_pw_cdest(const uint16_t *__restrict__ dest, uint32_t sb, uint32_t rs, uint32_t first, uint32_t nmaps, uint32_t G,
                                                        uint16_t *__restrict__ gdest)
{
    /* (round 5: the map's row goes through LDS -- it is in registers one map ahead, a gather is 128 LDS reads a thread; out
     * of the L2 a map was 15 us of dependent 2-byte reads) */
    extern __shared__ uint32_t pw_lds[];
    const uint16_t *row = reinterpret_cast<const uint16_t *>(pw_lds);
    pw_u32x4 *row4 = reinterpret_cast<pw_u32x4 *>(pw_lds);
    const uint32_t g = blockIdx.x, tid = threadIdx.x;
    const uint32_t m0 = g * G, m1 = min(m0 + G, nmaps);
    if (m0 >= m1) return;
    constexpr uint32_t NQ = 65536u * 2u / 16u / PW_SCAN_T;   /* 16-byte pieces of a row a thread (rs <= 65536) */
    const uint32_t nq4 = rs / 8u;                            /* (rs is a multiple of 8) */
    uint32_t cd[PW_SCAN_EPT];
#pragma unroll
    for (uint32_t q = 0; q < PW_SCAN_EPT; q++) cd[q] = q * PW_SCAN_T + tid < sb ? q * PW_SCAN_T + tid : (uint32_t)PW_DEAD;
    pw_u32x4 pre[NQ];
    auto load_row = [&](uint32_t m) {
        const pw_u32x4 *dj4 = reinterpret_cast<const pw_u32x4 *>(dest + ((size_t)first + m) * rs);
#pragma unroll
        for (uint32_t k = 0; k < NQ; k++) { uint32_t tq = tid; asm volatile("" : "+v"(tq)); pre[k] = dj4[min(tq + k * PW_SCAN_T, nq4 - 1u)]; }
    };
    load_row(m0);
    for (uint32_t m = m0; m < m1; m++) {
#pragma unroll
        for (uint32_t k = 0; k < NQ; k++)
            if (tid + k * PW_SCAN_T < nq4) row4[tid + k * PW_SCAN_T] = pre[k];
        if (m + 1u < m1) load_row(m + 1u);
        pw_lds_barrier();
#pragma unroll
        for (uint32_t q = 0; q < PW_SCAN_EPT; q++) cd[q] = cd[q] == PW_DEAD ? (uint32_t)PW_DEAD : (uint32_t)row[min(cd[q], sb - 1u)];
        pw_lds_barrier();
    }
#pragma unroll
    for (uint32_t q = 0; q < PW_SCAN_EPT; q++)
        if (q * PW_SCAN_T + tid < sb) gdest[(size_t)g * rs + q * PW_SCAN_T + tid] = (uint16_t)cd[q];
}

/* ------------------------------------------------------------------ launchers -------- */

void lz77kw_debug_dump(void)
{
#ifdef LZ77X_VARIANTS
    if (!LZ77X_VENV("LZ77X_PW_DEBUG")) return;
    unsigned long long h[64];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(pw_dbg), sizeof h) != hipSuccess) return;
    fprintf(stderr, "[pw] back wg0: init %llu loop %llu (%llu groups: %llu each) out %llu cycles | fwd wg0: prologue %llu (classify %llu, tail of %llu keys %llu, bitmaps %llu, ring %llu) loop %llu | prep wg0: %llu cycles, %llu rounds, %llu groups\n",
            h[1] - h[0], h[2] - h[1], h[4], h[4] ? (h[2] - h[1]) / h[4] : 0ull, h[3] - h[2], h[9] - h[8], h[11] - h[8], h[14], h[12] - h[11], h[13] - h[12],
            h[9] - h[13], h[10] - h[9], h[17] - h[16], h[18], h[19]);
#endif
}

uint32_t lz77kw_width(int sb)
{
    const char *e = LZ77X_VENV("LZ77X_PRIO_WIDE");
    if (e) { const int w = atoi(e); if (w == 256 || w == 1024) return (uint32_t)w; if (w == 64) return 64u; }
    return sb > 4096 ? 1024u : 64u;
}

/* the ring of a W-wide sweep: 32-bit priorities while they fit the LDS of a CU, else 18-bit codes */
bool lz77kw_pack18(uint32_t ring_n) { return (size_t)ring_n * 4 > (size_t)150 * 1024; }

template <class K> static hipError_t pw_lds_attr(K kern, size_t lds)
{
    if (lds <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

hipError_t lz77kw_prep(const uint32_t *d_ps, uint32_t nx, uint32_t sb_r, uint32_t W, uint64_t *d_rmask, uint64_t *d_gate0, uint64_t *d_cmask, hipStream_t s)
{
    const uint32_t tagn = sb_r + W;                        /* cells a group can touch, relative to its first step */
    const uint32_t ngroups = (nx + W - 1u) / W;
    hipError_t e;
#ifdef LZ77X_VARIANTS
    if (LZ77X_VENV("LZ77X_PW_PREP_V1")) {                  /* round 3's tag table (one 16-bit tag per cell): the cross-check */
        const size_t lds = (size_t)tagn * 2;
        if (W == 1024u) {
            if ((e = pw_lds_attr(k_pw_prep<1024>, lds)) != hipSuccess) return e;
            const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(2, ((size_t)160 * 1024) / (lds + 64)));
            hipLaunchKernelGGL(k_pw_prep<1024>, dim3(std::min(ngroups, 256u * per_cu)), dim3(1024), lds, s, d_ps, nx, tagn, d_rmask, d_gate0, d_cmask);
        } else {
            if ((e = pw_lds_attr(k_pw_prep<256>, lds)) != hipSuccess) return e;
            hipLaunchKernelGGL(k_pw_prep<256>, dim3(std::min(ngroups, 256u * 8u)), dim3(256), lds, s, d_ps, nx, tagn, d_rmask, d_gate0, d_cmask);
        }
        return hipGetLastError();
    }
#endif
    const size_t lds = ((size_t)(tagn + 31u) / 32u * 2 + 3 * (size_t)W) * 4;
    const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(8, ((size_t)160 * 1024) / (lds + 128)));
    if (W == 1024u) {
        if ((e = pw_lds_attr(k_pw_prep_ranked<1024>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_pw_prep_ranked<1024>, dim3(std::min(ngroups, 256u * per_cu)), dim3(PWP_T), lds, s, d_ps, nx, tagn, d_rmask, d_gate0, d_cmask);
    } else {
        if ((e = pw_lds_attr(k_pw_prep_ranked<256>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_pw_prep_ranked<256>, dim3(std::min(ngroups, 256u * per_cu)), dim3(PWP_T), lds, s, d_ps, nx, tagn, d_rmask, d_gate0, d_cmask);
    }
    return hipGetLastError();
}

hipError_t lz77kw_fwd(const uint32_t *d_ps, uint32_t nx, uint32_t sb, uint32_t rs, uint32_t B, uint32_t ring_n, uint32_t W, uint32_t b_first, uint32_t nb,
                      const uint64_t *d_rmask, const uint64_t *d_cmask, const uint64_t *d_gold, uint64_t *d_gnew, const uint32_t *d_in, uint32_t *d_xval,
                      uint32_t *d_summary, uint32_t voff, uint32_t *d_out_state, uint32_t ncarried, uint32_t *d_codes, uint32_t *d_gval,
                      uint32_t *d_in_prev, uint32_t have_prev, uint32_t *d_gates_changed, hipStream_t s)
{
    const bool pack = lz77kw_pack18(ring_n);
    hipError_t e;
    uint32_t sort_cap = PW_SORT_CAP;
    {
        const char *ce = LZ77X_VENV("LZ77X_PRIO_SORTCAP");      /* test hook: several passes of the rank prologue on small inputs */
        if (ce && atoi(ce) >= 64 && (uint32_t)atoi(ce) < PW_SORT_CAP) sort_cap = (uint32_t)atoi(ce);
    }
    const uint32_t probe = LZ77X_VENV("LZ77X_PW_PROBE") ? (uint32_t)atoi(LZ77X_VENV("LZ77X_PW_PROBE")) : 0u;
#define PW_FWD(WW, PK, LDS)                                                                                                        \
    do {                                                                                                                           \
        if ((e = pw_lds_attr(k_pw_fwd<WW, PK>, (LDS))) != hipSuccess) return e;                                                    \
        hipLaunchKernelGGL((k_pw_fwd<WW, PK>), dim3(nb), dim3(WW), (LDS), s, d_ps, nx, sb, rs, B, ring_n, b_first, d_rmask, d_cmask, d_gold, d_gnew, \
                           d_in, d_xval, d_summary, voff, d_out_state, ncarried, d_codes, d_gval, sort_cap, probe, d_in_prev, have_prev, d_gates_changed); \
    } while (0)
    if (pack) {
        /* the uint16 ring + the 2-bit plane; the rank prologue sorts up to PW_SORT_CAP keys in the same space */
        const size_t lds = std::max((size_t)ring_n * 2 + (size_t)ring_n / 4, (size_t)PW_SORT_CAP * 4);
        if (W != 1024u) return hipErrorInvalidValue;
        PW_FWD(1024, true, lds);
    } else {
        const size_t lds = (size_t)ring_n * 4;
        if (W == 1024u) PW_FWD(1024, false, lds); else PW_FWD(256, false, lds);
    }
#undef PW_FWD
    return hipGetLastError();
}

uint32_t lz77kw_back_slots(uint32_t ring_n, uint32_t W)
{
    /* workgroups that can be resident at once (256 CUs; the ring + the static arrays against 160 KB of LDS), at most four a CU */
    const size_t per = (size_t)ring_n * 2 + (size_t)W * 4 * 4 + 512;
    size_t k = (size_t)160 * 1024 / per;
    k = k < 1 ? 1 : k > 4 ? 4 : k;
    return (uint32_t)(256u * k);
}

size_t lz77kw_back_scratch_bytes(uint32_t NB, uint32_t B, uint32_t ring_n, uint32_t W)
{
    const uint32_t slots = lz77kw_back_slots(ring_n, W);
    return (size_t)(NB < slots ? NB : slots) * ((size_t)B + 16u) * 2 + 256;
}

hipError_t lz77kw_back(const uint32_t *d_ps, uint32_t nx, uint32_t sb, uint32_t rs, uint32_t B, uint32_t ring_n, uint32_t W, uint32_t b_first, uint32_t nb,
                       const uint64_t *d_gates, uint16_t *d_dest, uint32_t *d_loc, uint32_t voff, uint32_t ncarried, const uint32_t *d_gates_changed,
                       uint16_t *d_destx /* lz77kw_back_scratch_bytes */, hipStream_t s)
{
    const size_t lds = (size_t)ring_n * 2;
    const uint32_t slots = lz77kw_back_slots(ring_n, W), grid = nb < slots ? nb : slots;
    hipError_t e;
    if (W == 1024u) {
        if ((e = pw_lds_attr(k_pw_back<1024>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_pw_back<1024>, dim3(grid), dim3(1024), lds, s, d_ps, nx, sb, rs, B, ring_n, b_first, nb, d_gates, d_dest, d_loc, voff, ncarried, d_gates_changed, d_destx);
    } else {
        if ((e = pw_lds_attr(k_pw_back<256>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_pw_back<256>, dim3(grid), dim3(256), lds, s, d_ps, nx, sb, rs, B, ring_n, b_first, nb, d_gates, d_dest, d_loc, voff, ncarried, d_gates_changed, d_destx);
    }
    return hipGetLastError();
}

/* the scan's own workspace: the composed maps and inputs of the groups of groups (second level) */
size_t lz77kw_scan_tmp_bytes(uint32_t NG, uint32_t rs)
{
    uint32_t G2 = 1;
    while ((uint64_t)G2 * G2 < NG) G2++;
    const size_t NG2 = (NG + G2 - 1) / G2 + 6;              /* (+ slack: the scan sizes its groups from the maps it is given) */
    /* + two rows a group for the running vector of a composition, at both levels */
    return NG2 * rs * (2 + 4 + 4) + ((size_t)NG + NG2 + 4) * 2 * rs * 4 + 1024;
}

/* ALL the maps 0 .. nmaps-1 of a plan composed into ONE (what a shard of a stream cut over several devices sends to the
 * host): groups of G maps composed side by side, then the group maps by one workgroup.  The result is row
 * NG + 1 of gdest / gloc (NG = ceil(nmaps / G)); d_tmp: lz77kw_scan_tmp_bytes(NG, rs), the boundary scan's workspace. */
hipError_t lz77kw_compose_all(const uint16_t *d_dest, const uint32_t *d_loc, uint32_t sb, uint32_t rs, uint32_t nmaps, uint32_t G,
                              uint16_t *d_gdest, uint32_t *d_gloc, void *d_tmp, hipStream_t s)
{
    if (!nmaps) return hipSuccess;
    const uint32_t NG = (nmaps + G - 1u) / G;
    const size_t lds = (size_t)PW_MAP_HALF * 4;
    hipError_t e;
    if ((e = pw_lds_attr(k_pw_maps<false>, lds)) != hipSuccess) return e;
    if ((e = pw_lds_attr(k_pw_cdest, (size_t)rs * 2)) != hipSuccess) return e;
    uint32_t G2 = 1;
    while ((uint64_t)G2 * G2 < NG) G2++;
    const size_t NG2 = (NG + G2 - 1) / G2 + 6;              /* (the layout of lz77kw_scan_tmp_bytes) */
    uint8_t *t = reinterpret_cast<uint8_t *>(d_tmp);
    uint32_t *rows1 = reinterpret_cast<uint32_t *>(t + ((NG2 * rs * 10 + 255) & ~(size_t)255));
    uint32_t *rows2 = rows1 + (size_t)2 * NG * rs;
    uint16_t *out_dest = d_gdest + (size_t)(NG + 1u) * rs;
    uint32_t *out_loc = d_gloc + (size_t)(NG + 1u) * rs;
    hipLaunchKernelGGL(k_pw_cdest, dim3(NG), dim3(PW_SCAN_T), (size_t)rs * 2, s, d_dest, sb, rs, 0u, nmaps, G, d_gdest);
    hipLaunchKernelGGL(k_pw_maps<false>, dim3(NG), dim3(PW_MAP_T), lds, s, d_dest, d_loc, sb, rs, 0u, nmaps, G, (const uint32_t *)nullptr, d_gloc, rows1);
    /* the NG group maps (rows 0 .. NG-1 of gdest / gloc) in sequence */
    hipLaunchKernelGGL(k_pw_cdest, dim3(1), dim3(PW_SCAN_T), (size_t)rs * 2, s, d_gdest, sb, rs, 0u, NG, NG, out_dest);
    hipLaunchKernelGGL(k_pw_maps<false>, dim3(1), dim3(PW_MAP_T), lds, s, d_gdest, d_gloc, sb, rs, 0u, NG, NG, (const uint32_t *)nullptr, out_loc, rows2);
    return hipGetLastError();
}

/* in rows first+1 .. first+nmaps from in[first] through the maps first .. first+nmaps-1, groups of G; rows are rs
 * cells apart (a multiple of 8: every row 16-byte aligned).  gdest/gloc/gin: NG+2 rows each.  One workgroup applies a
 * map in ~20 us whatever else runs, so what counts is how few maps each applies in sequence: the group maps are scanned
 * the same way (1630 blocks: 136 groups of 12, 12 groups of 12 of those -- 59 maps in sequence instead of 121) */
hipError_t lz77kw_scan(const uint16_t *d_dest, const uint32_t *d_loc, uint32_t *d_in, uint32_t sb, uint32_t rs, uint32_t first, uint32_t nmaps, uint32_t G,
                       uint16_t *d_gdest, uint32_t *d_gloc, uint32_t *d_gin, void *d_tmp, hipStream_t s)
{
    if (!nmaps) return hipSuccess;
    const uint32_t NG = (nmaps + G - 1u) / G;
    const size_t lds = (size_t)PW_MAP_HALF * 4;
    hipError_t e;
    if ((e = pw_lds_attr(k_pw_maps<true>, lds)) != hipSuccess) return e;
    if ((e = pw_lds_attr(k_pw_maps<false>, lds)) != hipSuccess) return e;
    if ((e = pw_lds_attr(k_pw_cdest, (size_t)rs * 2)) != hipSuccess) return e;
    const uint32_t *vin = d_in + (size_t)first * rs;
    if (NG > 1) {
        /* workspace: [second level: gloc2, gin2, gdest2] [compose rows of this level] [compose rows of the second level] */
        uint32_t G2 = 1;
        while ((uint64_t)G2 * G2 < NG - 1u) G2++;
        const size_t NG2 = (NG - 1u + G2 - 1) / G2 + 2;
        uint8_t *t = reinterpret_cast<uint8_t *>(d_tmp);
        uint32_t *gloc2 = reinterpret_cast<uint32_t *>(t);
        uint32_t *gin2 = gloc2 + NG2 * rs;
        uint16_t *gdest2 = reinterpret_cast<uint16_t *>(gin2 + NG2 * rs);
        uint32_t *rows1 = reinterpret_cast<uint32_t *>(t + ((NG2 * rs * 10 + 255) & ~(size_t)255));
        uint32_t *rows2 = rows1 + (size_t)2 * NG * rs;
        /* the groups' composed maps (the last group's is nobody's input) */
        hipLaunchKernelGGL(k_pw_cdest, dim3(NG - 1u), dim3(PW_SCAN_T), (size_t)rs * 2, s, d_dest, sb, rs, first, nmaps, G, d_gdest);
        hipLaunchKernelGGL(k_pw_maps<false>, dim3(NG - 1u), dim3(PW_MAP_T), lds, s, d_dest, d_loc, sb, rs, first, nmaps, G, (const uint32_t *)nullptr, d_gloc, rows1);
        /* gin[0] = in[first]; gin[g+1] = group map g applied to gin[g] */
        if ((e = hipMemcpyAsync(d_gin, d_in + (size_t)first * rs, (size_t)sb * 4, hipMemcpyDeviceToDevice, s)) != hipSuccess) return e;
        if (NG - 1u > 16u) {
            /* the group maps scanned the same way, one level up */
            const uint32_t n2 = NG - 1u, NGG = (n2 + G2 - 1u) / G2;
            if (NGG > 1) {
                hipLaunchKernelGGL(k_pw_cdest, dim3(NGG - 1u), dim3(PW_SCAN_T), (size_t)rs * 2, s, d_gdest, sb, rs, 0u, n2, G2, gdest2);
                hipLaunchKernelGGL(k_pw_maps<false>, dim3(NGG - 1u), dim3(PW_MAP_T), lds, s, d_gdest, d_gloc, sb, rs, 0u, n2, G2, (const uint32_t *)nullptr, gloc2, rows2);
                if ((e = hipMemcpyAsync(gin2, d_gin, (size_t)sb * 4, hipMemcpyDeviceToDevice, s)) != hipSuccess) return e;
                hipLaunchKernelGGL(k_pw_maps<true>, dim3(1), dim3(PW_MAP_T), lds, s, gdest2, gloc2, sb, rs, 0u, NGG - 1u, NGG - 1u, gin2, gin2, (uint32_t *)nullptr);
                hipLaunchKernelGGL(k_pw_maps<true>, dim3(NGG), dim3(PW_MAP_T), lds, s, d_gdest, d_gloc, sb, rs, 0u, n2, G2, gin2, d_gin, (uint32_t *)nullptr);
            } else {
                hipLaunchKernelGGL(k_pw_maps<true>, dim3(1), dim3(PW_MAP_T), lds, s, d_gdest, d_gloc, sb, rs, 0u, n2, n2, d_gin, d_gin, (uint32_t *)nullptr);
            }
        } else {
            hipLaunchKernelGGL(k_pw_maps<true>, dim3(1), dim3(PW_MAP_T), lds, s, d_gdest, d_gloc, sb, rs, 0u, NG - 1u, NG - 1u, d_gin, d_gin, (uint32_t *)nullptr);
        }
        vin = d_gin;
    }
    /* every group from its input (row g of gin; a single group: in[first] itself) */
    hipLaunchKernelGGL(k_pw_maps<true>, dim3(NG), dim3(PW_MAP_T), lds, s, d_dest, d_loc, sb, rs, first, nmaps, G, vin, d_in, (uint32_t *)nullptr);
    return hipGetLastError();
}
