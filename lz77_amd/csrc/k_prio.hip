/*
 * k_prio.hip -- the priority recurrence of the offset tie-break, on the device.
 *
 * What it replaces: which node a tree.c:182-243 delete() promotes.  Evicting the oldest position x
 * (lz77.c:101-103) with two children puts its in-order successor S in x's place (tree.c:202-231); in
 * the treap restatement (SURVEY A.5 stage B) that is, for x = 0, 1, 2, ... in order,
 *
 *     if P[x], S[x] exist and pi[x] < pi[P[x]] and pi[x] < pi[S[x]]:  pi[S[x]] <- pi[x]       (1)
 *
 * with pi[c] = c when c is inserted.  Round 1 ran (1) on one host core (0.76 ns per position, 95 % of
 * an encode).  Cut into blocks of B consecutive steps it parallelises:
 *
 *   - a block only needs the values of the sb cells that are live at its first step ("in"); given them,
 *     one wavefront runs (1) over the block exactly (k_prio_fwd) -- every block at once;
 *   - a cell only ever DEcreases, and a write that fails its S-side test would not have lowered the
 *     cell, so (1) is   pi[S[x]] <- min(pi[S[x]], pi[x])   guarded by the P-side test alone: the GATE
 *     g[x] = "P, S exist and pi_x[x] < pi_x[P[x]]"  (pi_x = cells just before step x).  With the gates
 *     fixed the cell values are a min over a forest (x -> S[x] for open gates), and a block is a map on
 *     boundary values,  out[d] = min(loc[d], min{ in[c] : dest[c] = d })  (k_prio_back), which
 *     composes: the values at every block boundary are a scan over the blocks' maps (k_prio_scan_*);
 *   - so iterate: gates -> maps -> boundary values -> exact block sweeps -> gates.  Block 0 is exact
 *     after the first sweep, and by induction every block before the first block whose gates changed in
 *     an iteration is final (later iterations start there); a sweep that reproduces the gates its
 *     boundary values were computed from is the fixed point = the sequential result.  Measured: 5-7
 *     iterations on text, random bytes, low-entropy runs and record-structured data alike at B = 16K-64K
 *     (wrong gates shrink 30-100x per iteration: a gate mostly depends on recent history, which the
 *     exact sweep inside a block resolves at once).
 *
 * One WAVEFRONT owns a block; its 64 lanes take 64 consecutive steps, split into "rounds" wherever a
 * step reads a cell an earlier step of the same 64 writes (a static property of P/S: k_prio_prep
 * computes the round masks once).  The sb + 64 live cells of a sweep are a ring in LDS (18.7 KB at
 * sb 4095: eight wavefronts per CU).  xval[x] -- the priority handed over at the eviction of x, what
 * k_tokens consumes -- is written by every sweep; the last sweep of a block is the exact one.
 */
#include "kernels_common.h"
#include <stdio.h>

#define PRIO_NONE 0xFFFFFFFFu
#define PRIO_DEAD 0xFFFFu
#define PRIO_SG 8u                       /* groups of 64 steps fetched per round trip to global memory */
/* 1024 threads x 4 cells per map step; 256 x 16 is twice as slow (measured): a step is bound by the LDS work per
 * thread, not by its two barriers */
#define PRIO_SCAN_BLOCK 1024

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

/* ------------------------------------------------------------------ prep ------------- */

/* Per group of 64 consecutive steps: the initial gates (every step that has both neighbours) and the
 * round mask -- bit i set <=> lane i must start a new round because some lane j of the current round
 * (j < i) writes a cell lane i reads: its own (S[j] = x_i), its predecessor's (gate test) or its
 * successor's (the "did the hand-over happen" test).  tag[c] = (version, 63 - lowest lane of the current round that
 * writes cell c), kept with atomicMax: every round of every group has a version of its own, so stale tags are simply
 * older and nothing is ever reset (a round is two wavefront barriers instead of three and no clearing stores). */
__global__ __launch_bounds__(256) void k_prio_prep(const uint32_t *__restrict__ ps, uint32_t nx, uint32_t tagn,
                                                   uint64_t *__restrict__ rmask, uint64_t *__restrict__ gate0, uint64_t *__restrict__ cmask)
{
    extern __shared__ uint32_t prep_tags[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t *tag = prep_tags + wave * tagn;
    const uint32_t ngroups = (nx + 63u) / 64u;
    for (uint32_t i = lane; i < tagn; i += 64) tag[i] = 0u;
    wave_sync();
    uint32_t ver = 1;                                     /* < 2^26: a wavefront sees at most ngroups / 8192 groups of <= 64 rounds */
    /* the next group's neighbours travel while this one's rounds run (an unconditional, clamped load: under a branch the
     * compiler would wait for it at once).  Until round 5 every group began with its own round trip to HBM -- two waves per
     * SIMD hide none of it: ~2100 cycles a group where the rounds themselves take ~500, 0.66 ms per 100 MB */
    const uint32_t gstep = gridDim.x * 4u, xlast = nx - 1u;
    uint32_t g = blockIdx.x * 4u + wave;
    uint32_t vnext = g < ngroups ? ps[min(g * 64u + lane, xlast)] : 0u;
    for (; g < ngroups; g += gstep) {
        const uint32_t x = g * 64u + lane;
        const uint32_t v = x < nx ? vnext : 0u;
        {
            const uint32_t gn = g + gstep;
            vnext = ps[min((gn < ngroups ? gn : g) * 64u + lane, xlast)];
        }
        const uint32_t p = v & 0xFFFFu, s = v >> 16;
        const bool has = p && s;
        const uint32_t cp = lane + p, cs = lane + s;
        uint64_t mask = 1, chain = 0;
        uint32_t start = 0;
        for (;;) {
            if (has && lane >= start) atomicMax(&tag[cs], (ver << 6) | (63u - lane));
            wave_sync();
            uint32_t blocked = 0, linked = 0;
            if (has && lane > start) {
                const uint32_t t0 = tag[lane], t1 = tag[cp], t2 = tag[cs];
                const uint32_t w0 = 63u - (t0 & 63u);
                /* the step before this one hands its priority to THIS cell (S[x-1] = x: runs of equal bytes, periods):
                 * not a new round -- the sweep resolves such chains with a scan (k_prio_fwd).  One writer per cell and round:
                 * a second one would read the first one's successor cell and start a round of its own (b2). */
                const uint32_t own = (t0 >> 6) == ver && w0 < lane;
                linked = own && w0 + 1u == lane;
                const uint32_t b0 = own && !linked;
                const uint32_t b1 = (t1 >> 6) == ver && 63u - (t1 & 63u) < lane;
                const uint32_t b2 = (t2 >> 6) == ver && 63u - (t2 & 63u) < lane;
                blocked = b0 | b1 | b2;
            }
            wave_sync();
            ver++;
            const uint64_t bm = __ballot(blocked != 0u), lm = __ballot(linked != 0u);
            const uint32_t end = bm ? (uint32_t)__builtin_ctzll(bm) : 64u;
            /* the lanes [start, end) are a round; what they saw of each other is final */
            chain |= lm & ((end < 64u ? (1ull << end) : 0ull) - (1ull << start));
            if (!bm) break;
            start = end;
            mask |= 1ull << start;
        }
        const uint64_t hm = __ballot(has);
        /* (bit 0 of a round mask says nothing -- lane 0 always opens a round: cleared, it flags a group with chains) */
        if (lane == 0) { rmask[g] = chain ? mask & ~1ull : mask; gate0[g] = hm; cmask[g] = chain; }
    }
}

/* in[0]: at the start of the input every cell holds its own position (+ voff, the value numbering of a
 * shard); a later segment of a long input starts from the cells its predecessor left behind (carried) */
__global__ void k_prio_in0(uint32_t *__restrict__ in0, uint32_t sb, uint32_t voff, const uint32_t *__restrict__ carried)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < sb) in0[i] = carried ? carried[i] : i + voff;
}

/* ---- chains inside a round --------------------------------------------------------------------------------
 * Where the eviction of x hands its priority to x + 1 (S[x] = x + 1: runs of equal bytes at least a lookahead long,
 * periodic data), step x + 1 reads the cell step x has just written, and a group of 64 such steps took 64 rounds (the
 * "run cliff": low-entropy data swept 8x slower than text).  But along such a chain the step is a map on ONE value:
 * with w = the predecessor cell and c = the successor cell as they were before the round (nobody else in the round
 * writes them), the cell x + 1 holds after step x
 *     f(a) = a < min(w, c) ? a : c          (tree.c:202-231: the successor takes x's place only below both children)
 * and threshold maps (t, c), c >= t, are closed under composition:
 *     (t2, c2) o (t1, c1) = (min(t1, t2),  t2 < t1 ? c2 : c1 < t2 ? c1 : c2)
 * so a round resolves its chains with an inclusive scan over the lanes (DPP row shifts + two row broadcasts) instead
 * of one round per link.  A lane whose own cell nobody of the round writes enters the scan as a CONSTANT map
 * (threshold 0): whatever lies to its left is cut off, no segment flags needed. */
struct prio_tc { uint32_t t, c; };

__device__ __forceinline__ prio_tc prio_tc_then(prio_tc first, prio_tc second)
{
    prio_tc r;
    r.t = min(first.t, second.t);
    r.c = second.t < first.t ? second.c : (first.c < second.t ? first.c : second.c);
    return r;
}

template <int CTRL, int ROWS>
__device__ __forceinline__ prio_tc prio_tc_dpp(prio_tc v)
{
    /* lanes without a source (or outside the row mask) get the identity (t = 2^32 - 1 passes every priority) */
    prio_tc e;
    e.t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v.t, CTRL, ROWS, 0xF, false);
    e.c = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.c, CTRL, ROWS, 0xF, false);
    return e;
}

/* inclusive scan over the 64 lanes: lane i <- v[i] o v[i-1] o ... o v[0] */
__device__ __forceinline__ prio_tc prio_tc_scan(prio_tc v)
{
    v = prio_tc_then(prio_tc_dpp<0x111, 0xF>(v), v);          /* row_shr:1 */
    v = prio_tc_then(prio_tc_dpp<0x112, 0xF>(v), v);          /* row_shr:2 */
    v = prio_tc_then(prio_tc_dpp<0x114, 0xF>(v), v);          /* row_shr:4 */
    v = prio_tc_then(prio_tc_dpp<0x118, 0xF>(v), v);          /* row_shr:8 */
    v = prio_tc_then(prio_tc_dpp<0x142, 0xA>(v), v);          /* row_bcast:15 into rows 1 and 3 */
    v = prio_tc_then(prio_tc_dpp<0x143, 0xC>(v), v);          /* row_bcast:31 into rows 2 and 3 */
    return v;
}

/* ------------------------------------------------------------------ forward sweep ---- */

/* Block b = steps [b*B, min((b+1)*B, nx)).  in[b][i] = value of cell b*B+i before the block's first
 * step (cells beyond b*B+sb have not been written yet: they hold their own position).  The sweep is
 * recurrence (1) itself; gold[] (the gates the boundary values came from) is only compared against. */
/* STORE: write xval[] (every production sweep does; false is a timing probe) */
template <bool STORE>
__global__ __launch_bounds__(64) void k_prio_fwd(const uint32_t *__restrict__ ps, uint32_t nx, uint32_t sb, uint32_t B,
                                                 uint32_t ring_n, uint32_t b_first, const uint64_t *__restrict__ rmask,
                                                 const uint64_t *__restrict__ cmask /* lanes whose own cell the lane before them writes in their round */,
                                                 const uint64_t *__restrict__ gold, uint64_t *__restrict__ gnew,
                                                 const uint32_t *__restrict__ in, uint32_t *__restrict__ xval,
                                                 uint32_t *__restrict__ summary /* [0] += flips, [1] = min block with a flip */,
                                                 uint32_t voff /* a cell's own priority is its position + voff */,
                                                 uint32_t *__restrict__ out_state /* last block: the sb cells left live after the last step */,
                                                 uint32_t *__restrict__ gates_changed /* [b] = this sweep flipped a gate of block b */,
                                                 const uint32_t *__restrict__ in_changed /* [b] = 0: the block's entry cells are those of its last sweep */)
{
    extern __shared__ uint32_t ring[];
    __builtin_amdgcn_s_setprio(3);      /* a chain of dependent instructions: issue ahead of any co-resident throughput kernel */
    const uint32_t lane = threadIdx.x;
    const uint32_t b = b_first + blockIdx.x;
    const uint32_t x0 = b * B;
    const uint32_t x1 = nx - x0 < B ? nx : x0 + B;
    if (in_changed && !in_changed[b] && !(out_state && x1 == nx)) {
        /* the sweep is a function of the entry cells alone: the same cells as last time give the same xval[] and the
         * gates the maps of this iteration were built from -- no flip.  (On inputs of many rounds of blocks the tail
         * iterations touch a few per cent of them.) */
        for (uint32_t i = lane; i < (x1 - x0 + 63u) / 64u; i += 64) gnew[(x0 >> 6) + i] = gold[(x0 >> 6) + i];
        if (lane == 0) gates_changed[b] = 0;
        return;
    }
    for (uint32_t r = lane; r < ring_n; r += 64) ring[r] = r < sb ? in[(size_t)b * sb + r] : x0 + r + voff;
    wave_sync();

    uint32_t off = 0;                                  /* ring slot of cell xg */
    uint32_t nflip = 0;
    uint32_t v[PRIO_SG], vn[PRIO_SG];
    uint64_t rm_l = 0, go_l = 0, rm_n = 0, go_n = 0, cm_l = 0, cm_n = 0;
    /* every load is unconditional (clamped address, value masked afterwards): a load under a branch makes
     * the compiler wait for ALL outstanding loads at the first use, and the prefetch would overlap nothing */
    const uint32_t xlast = x1 - 1u;
    auto fetch = [&](uint32_t xs, uint32_t (&vv)[PRIO_SG], uint64_t &rml, uint64_t &gol, uint64_t &cml) {
#pragma unroll
        for (uint32_t k = 0; k < PRIO_SG; k++) {
            const uint32_t x = xs + 64u * k + lane;
            const uint32_t t = ps[min(x, xlast)];
            vv[k] = x < x1 ? t : 0u;
        }
        const uint32_t xq = min(xs + 64u * (lane & (PRIO_SG - 1u)), xlast);
        rml = rmask[xq >> 6];
        gol = gold[xq >> 6];
        cml = cmask[xq >> 6];
    };
    fetch(x0, v, rm_l, go_l, cm_l);
    for (uint32_t xs = x0; xs < x1; xs += 64u * PRIO_SG) {
        fetch(xs + 64u * PRIO_SG, vn, rm_n, go_n, cm_n);     /* next super-group in flight while this one runs */
        uint64_t gn_l = 0;
#pragma unroll
        for (uint32_t k = 0; k < PRIO_SG; k++) {
            const uint32_t xg = xs + 64u * k;
            if (xg < x1) {
                const uint64_t rm = readlane64(rm_l, (int)k), go = readlane64(go_l, (int)k);
                const uint32_t p = v[k] & 0xFFFFu, s = v[k] >> 16;
                const bool has = p && s;
                const uint32_t x = xg + lane;
                uint32_t ix = off + lane;
                ix -= ix >= ring_n ? ring_n : 0u;
                uint32_t ip = ix + p;
                ip -= ip >= ring_n ? ring_n : 0u;
                uint32_t is = ix + s;
                is -= is >= ring_n ? ring_n : 0u;
                uint32_t ng = 0;                                      /* a VGPR flag, not a bool: as a lane mask the compiler merges it
                                                                         through three levels of exec masks, 20 scalar instructions a round */
                uint32_t out = PRIO_NONE;
                uint64_t r = rm | 1ull;
                if (__builtin_expect((rm & 1ull) != 0ull, 1)) {
                    /* no chain in the group (wave-uniform): rounds of independent steps */
                    do {
                        const uint32_t start = (uint32_t)__builtin_ctzll(r);
                        r &= r - 1;
                        const uint32_t end = r ? (uint32_t)__builtin_ctzll(r) : 64u;
                        if (has && lane >= start && lane < end) {
                            /* the three reads leave together (one LDS round trip per round, not two: no short circuit) */
                            const uint32_t a = ring[ix], w = ring[ip], sv = ring[is];
                            const bool gate = a < w, lower = a < sv;
                            ng = gate ? 1u : 0u;                          /* the gate: x's predecessor hangs below x */
                            if (gate & lower) { ring[is] = a; out = a; }  /* tree.c:202-231: S takes x's place */
                        }
                        wave_sync();
                    } while (r);
                } else {
                    /* some rounds hold chains: every lane takes part in their scans */
                    const uint64_t cm = readlane64(cm_l, (int)k);
                    do {
                        const uint32_t start = (uint32_t)__builtin_ctzll(r);
                        r &= r - 1;
                        const uint32_t end = r ? (uint32_t)__builtin_ctzll(r) : 64u;
                        const bool on = has && lane >= start && lane < end;
                        const bool linked = on && ((cm >> lane) & 1ull);
                        uint32_t a = 0, w = 0, sv = 0;
                        if (on) { a = ring[ix]; w = ring[ip]; sv = ring[is]; }
                        if (cm & ((end < 64u ? (1ull << end) : 0ull) - (1ull << start))) {
                            prio_tc f;
                            f.t = min(w, sv);
                            f.c = sv;
                            if (!linked) { f.c = a < f.t ? a : f.c; f.t = 0u; }     /* its own cell is what the ring holds: a constant */
                            if (!on) { f.t = 0u; f.c = 0u; }
                            f = prio_tc_scan(f);
                            /* lane i now holds the cell its step leaves behind; a linked lane starts from its left neighbour's */
                            const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f.c, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
                            if (linked) a = left;
                        }
                        if (on) {
                            const bool gate = a < w, lower = a < sv;
                            ng = gate ? 1u : 0u;
                            if (gate & lower) { ring[is] = a; out = a; }
                        }
                        wave_sync();
                    } while (r);
                }
                const uint64_t gnb = __ballot(ng != 0u);
                nflip += (uint32_t)__popcll(gnb ^ go);
                if (lane == k) gn_l = gnb;
                if (STORE && x < x1) xval[x] = out;
                /* cell xg+64+sb+lane becomes live with the next group; its slot held cell xg+lane */
                uint32_t fi = off + lane;                             /* (off + 64 + sb_r + lane) mod ring_n, ring_n = sb_r + 64 */
                fi -= fi >= ring_n ? ring_n : 0u;
                if (x < x1) ring[fi] = xg + ring_n + lane + voff;  /* (the lanes past the last step keep their cells: out_state) */
                off += 64u;
                off -= off >= ring_n ? ring_n : 0u;
                wave_sync();
            }
        }
        /* the super-group's new gates straight to HBM (eight consecutive words): kept in LDS until the end of the block
         * they cost 8 KB of a wavefront's 25 KB and with them a third of the sweeps in flight on large inputs */
        if (lane < PRIO_SG && xs + 64u * lane < x1) gnew[(xs >> 6) + lane] = gn_l;
#pragma unroll
        for (uint32_t k = 0; k < PRIO_SG; k++) v[k] = vn[k];
        rm_l = rm_n;
        go_l = go_n;
        cm_l = cm_n;
    }
    wave_sync();
    if (out_state && x1 == nx) {
        /* cells nx .. nx+sb-1, what the next segment of a long input starts from: the ring now holds exactly
         * the cells [x1, x1 + ring_n) */
        for (uint32_t i = lane; i < sb; i += 64) out_state[i] = ring[(x1 - x0 + i) % ring_n];
    }
    if (lane == 0 && nflip) {
        atomicAdd(&summary[0], nflip);
        atomicMin(&summary[1], b);
        atomicMax(&summary[2], b);
    }
    if (lane == 0 && gates_changed) gates_changed[b] = nflip ? 1u : 0u;
}

#ifdef LZ77X_VARIANTS   /* (the sequential form of the boundary maps) */
#include "variants/prio_back.inc"
#endif

/* The same maps WITHOUT a sequential sweep (round 3).  With the gates fixed a block is a forest of pointers
 * x -> x + S[x] (open gates), and dest is "follow them until they leave the block": pointer doubling, not a
 * recurrence.  A workgroup takes the block in sub-blocks of BK2_SUB cells from the last to the first: the cells' pointers
 * (relative, uint16) are doubled in LDS until every one has left the sub-block or died (a pointer advances at least one
 * cell a hop: log2 rounds at worst, three to five on text), then one look-up in the resolved first sb cells of the
 * sub-block behind it (dn[]) finishes them.  k_prio_back walks 64 K steps on one wavefront (0.28 ms whatever the number
 * of blocks: 1.4 of an encode's 13.4 ms); this is ~10 us of a workgroup per block. */
#define BK2_T 1024u
#define BK2_SUB 8192u
#define BK2_CPT (BK2_SUB / BK2_T)

__device__ __forceinline__ void bk2_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   /* LDS traffic only: the prefetched rows stay in flight */
}

__global__ __launch_bounds__(BK2_T, 8) __attribute__((amdgpu_num_sgpr(80))) void k_prio_back2(const uint32_t *__restrict__ ps, uint32_t nx, uint32_t sb, uint32_t B, uint32_t b_first,
                                                     const uint64_t *__restrict__ gates, uint16_t *__restrict__ dest,
                                                     uint32_t *__restrict__ loc, uint32_t voff, uint32_t ncarried,
                                                     const uint32_t *__restrict__ gates_changed)
{
    __shared__ uint16_t ptr[BK2_SUB];                    /* pointer of cell c of the sub-block: < L inside, >= L an exit (cell y1 + ptr - L), DEAD */
    __shared__ uint16_t dn[2][4096];                     /* dest of the cells y1 + i behind the current sub-block */
    __shared__ uint32_t lloc[4096];
    __shared__ uint32_t s_more[2];                       /* "a pointer is still inside": one flag per parity of the doubling round */
    const uint32_t b = b_first + blockIdx.x;
    if (gates_changed && !gates_changed[b]) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t x0 = b * B;
    const uint32_t x1 = nx - x0 < B ? nx : x0 + B;
    const uint32_t nsub = (x1 - x0 + BK2_SUB - 1u) / BK2_SUB;
    for (uint32_t i = tid; i < sb; i += BK2_T) {
        lloc[i] = x1 + i >= ncarried ? x1 + i + voff : PRIO_NONE;     /* what reaches exit cell i when nothing older comes in (see k_prio_back) */
        dn[0][i] = (uint16_t)i;                                        /* cell x1 + i IS exit cell i */
    }
    if (tid < 2) s_more[tid] = 0;
    uint32_t v[BK2_CPT], vn[BK2_CPT];
    uint32_t g[BK2_CPT], gn[BK2_CPT];
    const uint32_t xlast = x1 - 1u;
    const uint32_t *g32 = reinterpret_cast<const uint32_t *>(gates);                        /* bit x of the gates = bit x & 31 of word x >> 5 */
    auto fetch = [&](int32_t j, uint32_t (&vv)[BK2_CPT], uint32_t (&gg)[BK2_CPT]) {        /* unconditional, clamped loads */
        const uint32_t y0 = x0 + BK2_SUB * (uint32_t)(j < 0 ? 0 : j);
#pragma unroll
        for (uint32_t k = 0; k < BK2_CPT; k++) {
            const uint32_t x = min(y0 + tid + BK2_T * k, xlast);
            vv[k] = ps[x];
            gg[k] = g32[x >> 5];
        }
    };
    fetch((int32_t)nsub - 1, v, g);
    int w = 0;
    for (int32_t j = (int32_t)nsub - 1; j >= 0; j--) {
        fetch(j - 1, vn, gn);
        const uint32_t y0 = x0 + BK2_SUB * (uint32_t)j, y1 = min(y0 + BK2_SUB, x1), L = y1 - y0;
        uint32_t p[BK2_CPT];
#pragma unroll
        for (uint32_t k = 0; k < BK2_CPT; k++) {
            const uint32_t c = tid + BK2_T * k, x = y0 + c;
            const bool gate = c < L && ((g[k] >> (x & 31u)) & 1u);
            p[k] = gate ? c + (v[k] >> 16) : (uint32_t)PRIO_DEAD;
            ptr[c] = (uint16_t)p[k];
        }
        bk2_barrier();                                    /* (also: the flags, dn[w] of the sub-block before; __syncthreads() would wait for the prefetch) */
        for (uint32_t r = 0;; r++) {
            bool more = false, mine = false;
            uint32_t q[BK2_CPT];
#pragma unroll
            for (uint32_t k = 0; k < BK2_CPT; k++) mine |= p[k] < L;
            /* a wavefront whose pointers have all left the sub-block only keeps the barriers company (after two or three
             * rounds that is most of them) */
            const bool active = __ballot(mine) != 0ull;
            if (active) {
                /* two hops a round (half the rounds, and a round is mostly its two barriers); the reads of a hop leave
                 * together: one LDS round trip for the eight cells, not eight */
                uint32_t h[BK2_CPT];
#pragma unroll
                for (uint32_t k = 0; k < BK2_CPT; k++) h[k] = ptr[min(p[k], L - 1u)];
#pragma unroll
                for (uint32_t k = 0; k < BK2_CPT; k++) h[k] = p[k] < L ? h[k] : p[k];          /* (DEAD >= L) */
#pragma unroll
                for (uint32_t k = 0; k < BK2_CPT; k++) q[k] = ptr[min(h[k], L - 1u)];
#pragma unroll
                for (uint32_t k = 0; k < BK2_CPT; k++) {
                    q[k] = h[k] < L ? q[k] : h[k];
                    more |= q[k] < L;
                }
                if (__ballot(more) && (tid & 63u) == 0) s_more[r & 1u] = 1;
            }
            bk2_barrier();                                 /* every read of the round before any write */
            const bool again = s_more[r & 1u] != 0;
            if (tid == 0) s_more[(r + 1u) & 1u] = 0;       /* (last read before this round's first barrier, next written behind its second) */
            if (active) {
#pragma unroll
                for (uint32_t k = 0; k < BK2_CPT; k++) {
                    if (q[k] != p[k]) ptr[tid + BK2_T * k] = (uint16_t)q[k];
                    p[k] = q[k];
                }
            }
            bk2_barrier();
            /* a round replaces ptr by ptr o ptr o ptr (both hops read the array as the round found it): the reach triples, a
             * sub-block of 8192 cells is through after nine rounds; pointers only ever point forward (S >= 1), so the bound is
             * never met on any ps[] the match stage can write */
            if (!again || r >= 15u) break;
        }
        /* (should corrupt input ever stop the loop at its bound: a pointer still inside is dropped, not turned into an exit
         * cell below zero) */
#pragma unroll
        for (uint32_t k = 0; k < BK2_CPT; k++) p[k] = p[k] < L ? (uint32_t)PRIO_DEAD : p[k];
        if (tid < 2) s_more[tid] = 0;                      /* (ordered before the next sub-block's rounds by its barrier) */
        /* the exits: dest of cell y1 + e, e = pointer - L */
        const uint16_t *dcur = dn[w];
        uint16_t *dnew = dn[w ^ 1];
        uint32_t d[BK2_CPT], dk[BK2_CPT];
        /* (unconditional reads, selected afterwards: they leave together) */
#pragma unroll
        for (uint32_t k = 0; k < BK2_CPT; k++) d[k] = dcur[p[k] == PRIO_DEAD ? 0u : p[k] - L];
#pragma unroll
        for (uint32_t k = 0; k < BK2_CPT; k++) {
            const uint32_t c = tid + BK2_T * k;
            dk[k] = c < sb && c >= L ? (uint32_t)dcur[c - L] : 0u;       /* (a short last sub-block: the cells behind it) */
        }
#pragma unroll
        for (uint32_t k = 0; k < BK2_CPT; k++) d[k] = p[k] == PRIO_DEAD ? (uint32_t)PRIO_DEAD : d[k];
#pragma unroll
        for (uint32_t k = 0; k < BK2_CPT; k++) {
            const uint32_t x = y0 + tid + BK2_T * k;
            if (d[k] != PRIO_DEAD && x >= ncarried) atomicMin(&lloc[d[k]], x + voff);    /* (d != DEAD: an open gate, a cell of the sub-block) */
        }
        /* dn for the sub-block before this one: dest of the cells y0 + i, i < sb -- mine, then the cells behind me */
#pragma unroll
        for (uint32_t k = 0; k < BK2_CPT; k++) {
            const uint32_t c = tid + BK2_T * k;
            if (c < sb) dnew[c] = (uint16_t)(c < L ? d[k] : dk[k]);
        }
        w ^= 1;
#pragma unroll
        for (uint32_t k = 0; k < BK2_CPT; k++) { v[k] = vn[k]; g[k] = gn[k]; }
        bk2_barrier();
    }
    for (uint32_t i = tid; i < sb; i += BK2_T) {
        dest[(size_t)b * sb + i] = dn[w][i];             /* (an entry cell a short last block does not evict: still live at its end) */
        loc[(size_t)b * sb + i] = lloc[i];
    }
}

/* ------------------------------------------------------------------ scan over blocks -- */

/* in[j+1] = F_j(in[j]),  F_j(v)[d] = min(loc_j[d], min{ v[c] : dest_j[c] = d }),  for the maps
 * j = b_first .. NB-2 in groups of G: compose each group's maps, run the group maps in sequence,
 * then replay every group from its now known input.  The per-step cost is two workgroup barriers and a
 * few LDS operations per cell; the next map's rows are fetched while the current one is applied. */
#define SCAN_AHEAD 4u                                 /* maps whose rows are in flight */
#define SCAN_CPT (4096 / PRIO_SCAN_BLOCK)             /* cells per thread: sb <= 4096 on this path */

struct scan_regs { uint32_t d[SCAN_CPT], l[SCAN_CPT]; };

__device__ __forceinline__ void scan_fetch(scan_regs &r, const uint16_t *dest, const uint32_t *loc, size_t j, uint32_t sb)
{
    /* unconditional loads (clamped index), so that the compiler can keep them in flight across a step */
#pragma unroll
    for (int q = 0; q < SCAN_CPT; q++) {
        const uint32_t i = min(threadIdx.x + PRIO_SCAN_BLOCK * q, sb - 1u);
        r.d[q] = (uint32_t)dest[j * sb + i];
        r.l[q] = loc[j * sb + i];
    }
}

/* one step on LDS vectors: cur (values at the map's entry cells) -> nxt.  The two vectors are addressed as offsets into
 * the kernel's LDS array: selected through an array of pointers they were GENERIC pointers, every access a flat
 * operation behind s_waitcnt vmcnt(0) lgkmcnt(0) -- each step drained the prefetched rows (2 us per step) */
__device__ __forceinline__ void scan_apply(const scan_regs &r, uint32_t *lds, uint32_t cur_off, uint32_t nxt_off, uint32_t sb)
{
    const uint32_t *cur = lds + cur_off;
    uint32_t *nxt = lds + nxt_off;
#pragma unroll
    for (int q = 0; q < SCAN_CPT; q++) {
        const uint32_t i = threadIdx.x + PRIO_SCAN_BLOCK * q;
        if (i < sb) nxt[i] = r.l[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SCAN_CPT; q++) {
        const uint32_t i = threadIdx.x + PRIO_SCAN_BLOCK * q;
        if (i < sb && r.d[q] != PRIO_DEAD) {
            const uint32_t val = cur[i];
            if (val != PRIO_NONE) atomicMin(&nxt[r.d[q]], val);
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(PRIO_SCAN_BLOCK) void k_prio_scan_compose(const uint16_t *__restrict__ dest, const uint32_t *__restrict__ loc,
                                                                       uint32_t sb, uint32_t sb_r, uint32_t b_first, uint32_t nmaps, uint32_t G,
                                                                       uint16_t *__restrict__ gdest, uint32_t *__restrict__ gloc)
{
    extern __shared__ uint32_t scan_lds[];
    __builtin_amdgcn_s_setprio(3);
    uint16_t *cd = reinterpret_cast<uint16_t *>(scan_lds + 2 * sb_r);       /* composed dest */
    uint16_t *dj = cd + sb_r;                                                /* the current map's dest row */
    const uint32_t gi = blockIdx.x;
    const uint32_t m0 = gi * G, m1 = min(m0 + G, nmaps);
    for (uint32_t i = threadIdx.x; i < sb; i += PRIO_SCAN_BLOCK) { scan_lds[i] = PRIO_NONE; cd[i] = (uint16_t)i; }
    /* the rows of the next SCAN_AHEAD maps are in flight while a map is applied: a step is two barriers and a few
     * LDS operations, far shorter than a round trip to the rows k_prio_back has just written */
    scan_regs rr[SCAN_AHEAD];
#pragma unroll
    for (uint32_t u = 0; u < SCAN_AHEAD; u++) scan_fetch(rr[u], dest, loc, (size_t)b_first + min(m0 + u, m1 - 1u), sb);
    int w = 0;
    __syncthreads();
    for (uint32_t mb = m0; mb < m1; mb += SCAN_AHEAD) {
#pragma unroll
        for (uint32_t u = 0; u < SCAN_AHEAD; u++) {
            const uint32_t m = mb + u;
            if (m < m1) {
                const scan_regs cur = rr[u];
                scan_fetch(rr[u], dest, loc, (size_t)b_first + min(m + SCAN_AHEAD, m1 - 1u), sb);
#pragma unroll
                for (int q = 0; q < SCAN_CPT; q++) {
                    const uint32_t i = threadIdx.x + PRIO_SCAN_BLOCK * q;
                    if (i < sb) dj[i] = (uint16_t)cur.d[q];
                }
                scan_apply(cur, scan_lds, w ? sb_r : 0u, w ? 0u : sb_r, sb);          /* its first barrier also publishes dj */
#pragma unroll
                for (int q = 0; q < SCAN_CPT; q++) {
                    const uint32_t i = threadIdx.x + PRIO_SCAN_BLOCK * q;
                    if (i < sb) { const uint32_t c = cd[i]; cd[i] = c == PRIO_DEAD ? (uint16_t)PRIO_DEAD : dj[c]; }
                }
                __syncthreads();
                w ^= 1;
            }
        }
    }
    for (uint32_t i = threadIdx.x; i < sb; i += PRIO_SCAN_BLOCK) {
        gdest[(size_t)gi * sb + i] = cd[i];
        gloc[(size_t)gi * sb + i] = scan_lds[(w ? sb_r : 0u) + i];
    }
}

/* replay `count` maps starting at map index m0 (rows of dest/loc start at row0) from the input vector
 * vin; out[m+1-th row] receives the vector after map m.  Used twice: over the group maps (one
 * workgroup) and inside every group (one workgroup per group). */
template <bool TRACK>      /* TRACK: compare what is written to vout with what was there (a load per value: not for free) */
__global__ __launch_bounds__(PRIO_SCAN_BLOCK) void k_prio_scan_replay(const uint16_t *__restrict__ dest, const uint32_t *__restrict__ loc,
                                                                      uint32_t sb, uint32_t sb_r, size_t row0, uint32_t nmaps, uint32_t G,
                                                                      const uint32_t *__restrict__ vin, size_t vin_stride,
                                                                      uint32_t *__restrict__ vout, size_t vout_row0, uint32_t store_first,
                                                                      uint32_t *__restrict__ changed /* [row] = 1 when a value written to that row of vout
                                                                                                        differs from what was there (null: not tracked) */)
{
    extern __shared__ uint32_t scan_lds[];
    __builtin_amdgcn_s_setprio(3);
    const uint32_t gi = blockIdx.x;
    const uint32_t m0 = gi * G, m1 = min(m0 + G, nmaps);
    if (m0 > nmaps) return;
    const uint32_t *src = vin + (size_t)gi * vin_stride;
    for (uint32_t i = threadIdx.x; i < sb; i += PRIO_SCAN_BLOCK) {
        const uint32_t val = src[i];
        scan_lds[i] = val;
        if (store_first) vout[(vout_row0 + m0) * sb + i] = val;
    }
    if (m0 >= m1) return;                               /* (a group that only passes its input on: the row after the last map) */
    scan_regs rr[SCAN_AHEAD];
#pragma unroll
    for (uint32_t u = 0; u < SCAN_AHEAD; u++) scan_fetch(rr[u], dest, loc, row0 + min(m0 + u, m1 - 1u), sb);
    int w = 0;
    __syncthreads();
    for (uint32_t mb = m0; mb < m1; mb += SCAN_AHEAD) {
#pragma unroll
        for (uint32_t u = 0; u < SCAN_AHEAD; u++) {
            const uint32_t m = mb + u;
            if (m < m1) {
                const scan_regs cur = rr[u];
                scan_fetch(rr[u], dest, loc, row0 + min(m + SCAN_AHEAD, m1 - 1u), sb);
                scan_apply(cur, scan_lds, w ? sb_r : 0u, w ? 0u : sb_r, sb);
                w ^= 1;
                bool diff = false;
                for (uint32_t i = threadIdx.x; i < sb; i += PRIO_SCAN_BLOCK) {
                    const uint32_t val = scan_lds[(w ? sb_r : 0u) + i];
                    uint32_t *q = vout + (vout_row0 + m + 1) * sb + i;
                    if constexpr (TRACK) diff |= *q != val;
                    *q = val;
                }
                if constexpr (TRACK) { if (diff) changed[vout_row0 + m + 1] = 1u; }
            }
        }
    }
}

__global__ void k_prio_reset(uint32_t *summary)
{
    summary[0] = 0;
    summary[1] = PRIO_NONE;
    summary[2] = 0;                      /* the last block with a flip (with [1]: how wide the stretch of wrong gates is) */
}

/* ------------------------------------------------------------------ host driver ------- */

/* Steps per block.  Large blocks converge in fewer iterations (fewer boundaries whose values lag one
 * iteration behind: 8 iterations at 16K, 5 at 64K on 100 MB of text) and make the boundary scan short;
 * but a sweep wants >= ~1500 blocks in flight (one wavefront each): nx/1536 clamped to [16K, 64K].
 * LZ77X_PRIO_BLOCK overrides. */
static uint32_t prio_block_steps(uint32_t nx, uint32_t sb, uint32_t W, bool pack18, uint32_t ring_n)
{
    const char *e = getenv("LZ77X_PRIO_BLOCK");
    uint32_t B = e && atoi(e) > 0 ? (uint32_t)atoi(e) : nx / 1536u;
    const uint32_t unit = W > 64u ? W : 64u * PRIO_SG;
    if (!(e && atoi(e) > 0)) {
        /* a workgroup per block is through a block 10x sooner than a wavefront, and the maps of a block are sb entries
         * whatever its length: longer blocks, fewer maps to build, scan and keep */
        const uint32_t hi = sb > 4096u ? 131072u : 65536u;
        if (B > hi) B = hi;
        if (B < 16384u) B = 16384u;
    }
    if (B < sb) B = sb;                                   /* every entry cell must be evicted inside its block */
    B = (B + unit - 1u) / unit * unit;
    if (pack18) {
        /* codes: sb ranks + the positions x0 .. x0 + B + ring_n must fit 18 bits */
        const uint32_t cap = ((1u << 18) - sb - ring_n) / unit * unit;
        if (B > cap) B = cap;
    }
    return B;
}

/* ---- the iteration, as phases (one device drives them in a loop: lz77k_prio; several devices, each with a
 *      shard of one stream, interleave them with a host exchange of the shards' boundary maps) ---- */

static void prio_layout(lz77k_prio_plan &P)
{
    const uint32_t sb = P.sb, nx = P.nx;
    P.W = lz77kw_width((int)sb);
    P.rs = sb > 4096u ? (sb + 7u) & ~7u : sb;             /* (the HBM scan reads its rows 16 bytes a lane) */
    const size_t rs = P.rs;
    P.sb_r = (sb + 63u) & ~63u;
    P.ring_n = P.sb_r + P.W;
    P.pack18 = P.W > 64u && lz77kw_pack18(P.ring_n);
    P.B = prio_block_steps(nx, sb, P.W, P.pack18, P.ring_n);
    P.NB = nx ? (nx + P.B - 1u) / P.B : 0u;
    P.ngroups = (nx + 63u) / 64u;
    uint32_t G = 1;
    while ((uint64_t)G * G * G < P.NB) G++;                      /* two levels of groups: the cube root */
    const char *e = getenv("LZ77X_PRIO_SCAN_GROUP");
    if (e && atoi(e) > 0) G = (uint32_t)atoi(e);
    P.G = G;
    P.NG = P.NB ? (P.NB + G - 1u) / G : 0u;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    P.o_gate[0] = take((size_t)P.ngroups * 8 + 64);
    P.o_gate[1] = take((size_t)P.ngroups * 8 + 64);
    P.o_rmask = take((size_t)P.ngroups * 8 + 64);
    P.o_cmask = take((size_t)P.ngroups * 8 + 64);
    P.o_dest = take(((size_t)P.NB + 1) * rs * 2);
    P.o_loc = take(((size_t)P.NB + 1) * rs * 4);
    P.o_in = take(((size_t)P.NB + 2) * rs * 4);
    P.o_gdest = take(((size_t)P.NG + 2) * rs * 2);
    P.o_gloc = take(((size_t)P.NG + 2) * rs * 4);
    P.o_gin = take(((size_t)P.NG + 2) * rs * 4);
    P.NG2 = P.NG ? (P.NG + G - 1u) / G : 0u;
    if (sb <= 4096u) {                                           /* the groups of groups (LDS scans) */
        P.o_g2dest = take(((size_t)P.NG2 + 2) * rs * 2);
        P.o_g2loc = take(((size_t)P.NG2 + 2) * rs * 4);
        P.o_g2in = take(((size_t)P.NG2 + 2) * rs * 4);
    }
    P.o_sum = take(256);
    P.o_dirty = take(((size_t)P.NB + 2) * 2 * 4);            /* per block: [0, NB+2) its gates changed in the last sweep, then its entry cells changed in the last scan */
    if (P.pack18) {
        P.o_codes = take(((size_t)P.NB + 1) * rs * 4);
        P.o_gval = take(((size_t)P.NB + 1) * rs * 4);
    }
    if (sb > 4096u) P.o_scan = take(lz77kw_scan_tmp_bytes(P.NG, P.rs));
    if (P.W > 64u) P.o_inprev = take(((size_t)P.NB + 2) * rs * 4);        /* the cells every block's last sweep started from */
    if (P.W > 64u) P.o_destx = take(lz77kw_back_scratch_bytes(P.NB, P.B, P.ring_n, P.W));
    P.total = o;
}

size_t lz77k_prio_tmp_bytes(uint32_t nx, int sb)
{
    lz77k_prio_plan P;
    P.nx = nx;
    P.sb = (uint32_t)sb;
    prio_layout(P);
    return P.total + 256;
}

int lz77k_prio_supported(int sb) { return sb >= 1 && sb <= 65535; }

#define PRIO_PTR(T, off) reinterpret_cast<T *>(reinterpret_cast<uint8_t *>(P.tmp) + (off))

/* round masks + initial gates (every step with both neighbours), and the cells block 0 starts from */
hipError_t lz77k_prio_begin(lz77k_prio_plan &P, const uint32_t *d_ps, uint32_t nx, int sb, uint32_t *d_xval, void *d_tmp, uint32_t voff,
                            const uint32_t *d_carried, hipStream_t s)
{
    P.ps = d_ps;
    P.nx = nx;
    P.sb = (uint32_t)sb;
    P.xval = d_xval;
    P.tmp = d_tmp;
    P.voff = voff;
    P.ncarried = d_carried ? (uint32_t)sb : 0u;
    P.cur = 0;
    P.first = 0;
    prio_layout(P);
    if (nx == 0) return hipSuccess;
    hipError_t e;
    if (P.W > 64u) {
        if ((e = lz77kw_prep(d_ps, nx, P.sb_r, P.W, PRIO_PTR(uint64_t, P.o_rmask), PRIO_PTR(uint64_t, P.o_gate[0]), PRIO_PTR(uint64_t, P.o_cmask), s)) != hipSuccess) return e;
    } else {
        const uint32_t tagn = P.ring_n + 64u;
        const size_t lds = (size_t)4 * tagn * sizeof(uint32_t);
        if (lds > 48 * 1024 &&
            (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_prio_prep), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
            return e;
        const uint32_t blocks = min((P.ngroups + 3u) / 4u, 256u * 8u);
        hipLaunchKernelGGL(k_prio_prep, dim3(blocks), dim3(256), lds, s, d_ps, nx, tagn, PRIO_PTR(uint64_t, P.o_rmask), PRIO_PTR(uint64_t, P.o_gate[0]),
                           PRIO_PTR(uint64_t, P.o_cmask));
    }
    hipLaunchKernelGGL(k_prio_in0, dim3((P.sb + 255u) / 256u), dim3(256), 0, s, PRIO_PTR(uint32_t, P.o_in), P.sb, voff, d_carried);
    /* every block's map has to be built and every block swept once */
    P.in0_dirty = false;
    P.sweeps = 0;
    return hipMemsetAsync(PRIO_PTR(uint8_t, P.o_dirty), 1, ((size_t)P.NB + 2) * 2 * 4, s);
}

/* the cells block 0 starts from, when they only become known later (a shard: they follow from the maps of
 * the shards before it) */
hipError_t lz77k_prio_set_in0(lz77k_prio_plan &P, const uint32_t *h_or_d_in0, hipMemcpyKind kind, hipStream_t s)
{
    if (P.nx == 0) return hipSuccess;
    P.in0_dirty = true;
    return hipMemcpyAsync(PRIO_PTR(uint32_t, P.o_in), h_or_d_in0, (size_t)P.sb * 4, kind, s);
}

/* the blocks' boundary maps from the current gates (blocks >= P.first); whole = also the composition of
 * ALL blocks of this plan into one map (row NG+1 of gdest/gloc: what a shard sends to the host) */
hipError_t lz77k_prio_maps(lz77k_prio_plan &P, hipStream_t s, bool whole, const uint16_t **d_sdest, const uint32_t **d_sloc)
{
    if (P.nx == 0) return hipSuccess;
    [[maybe_unused]] const size_t lds_back = (size_t)P.sb_r * 4 + (size_t)P.ring_n * 2;
    const size_t lds_scan = (size_t)P.sb_r * (4 + 4 + 2 + 2);
    uint16_t *dest = PRIO_PTR(uint16_t, P.o_dest), *gdest = PRIO_PTR(uint16_t, P.o_gdest);
    uint32_t *loc = PRIO_PTR(uint32_t, P.o_loc), *gloc = PRIO_PTR(uint32_t, P.o_gloc);
    const uint32_t nb = P.NB - P.first;
    if (P.W > 64u) {
        hipError_t e = lz77kw_back(P.ps, P.nx, P.sb, P.rs, P.B, P.ring_n, P.W, P.first, nb, PRIO_PTR(uint64_t, P.o_gate[P.cur]), dest, loc, P.voff, P.ncarried,
                                   PRIO_PTR(uint32_t, P.o_dirty), PRIO_PTR(uint16_t, P.o_destx), s);
        if (e != hipSuccess) return e;
        if (whole && P.sb > 4096u) {
            /* the whole-plan map of a shard, through HBM like the boundary scan (a vector of sb priorities does not fit LDS
             * twice): every block 0 .. NB-1 -- the maps of the blocks before P.first are still there and final */
            if ((e = lz77kw_compose_all(dest, loc, P.sb, P.rs, P.NB, P.G, gdest, gloc, PRIO_PTR(uint8_t, P.o_scan), s)) != hipSuccess) return e;
            if (d_sdest) *d_sdest = gdest + (size_t)(P.NG + 1) * P.rs;
            if (d_sloc) *d_sloc = gloc + (size_t)(P.NG + 1) * P.rs;
            return hipGetLastError();
        }
    } else {
#ifdef LZ77X_VARIANTS
        if (LZ77X_VENV("LZ77X_PRIO_BACK_SWEEP"))                                           /* the sequential form (cross-check) */
            hipLaunchKernelGGL(k_prio_back, dim3(nb), dim3(64), lds_back, s, P.ps, P.nx, P.sb, P.B, P.ring_n, P.first, PRIO_PTR(uint64_t, P.o_gate[P.cur]),
                               dest, loc, P.voff, P.ncarried, PRIO_PTR(uint32_t, P.o_dirty));
        else
#endif
            hipLaunchKernelGGL(k_prio_back2, dim3(nb), dim3(BK2_T), 0, s, P.ps, P.nx, P.sb, P.B, P.first, PRIO_PTR(uint64_t, P.o_gate[P.cur]),
                               dest, loc, P.voff, P.ncarried, PRIO_PTR(uint32_t, P.o_dirty));
    }
    if (whole) {
        /* every block 0 .. NB-1 (the maps of the blocks before P.first are still there and final), in groups,
         * then the groups */
        hipLaunchKernelGGL(k_prio_scan_compose, dim3(P.NG), dim3(PRIO_SCAN_BLOCK), lds_scan, s, dest, loc, P.sb, P.sb_r, 0u, P.NB, P.G, gdest, gloc);
        hipLaunchKernelGGL(k_prio_scan_compose, dim3(1), dim3(PRIO_SCAN_BLOCK), lds_scan, s, gdest, gloc, P.sb, P.sb_r, 0u, P.NG, P.NG,
                           gdest + (size_t)(P.NG + 1) * P.sb, gloc + (size_t)(P.NG + 1) * P.sb);
        if (d_sdest) *d_sdest = gdest + (size_t)(P.NG + 1) * P.sb;
        if (d_sloc) *d_sloc = gloc + (size_t)(P.NG + 1) * P.sb;
    }
    return hipGetLastError();
}

/* boundary values from the maps (in[first] is final), then the exact sweep of every block >= first;
 * the flip summary lands in h_flag (8 bytes, pinned) once the stream has drained */
hipError_t lz77k_prio_sweep(lz77k_prio_plan &P, hipStream_t s, uint32_t *h_flag, uint32_t *d_out_state, hipEvent_t *ev3)
{
    if (P.nx == 0) { h_flag[0] = 0; h_flag[1] = PRIO_NONE; h_flag[2] = 0; return hipSuccess; }
    const size_t lds_fwd = (size_t)P.ring_n * 4;
    const size_t lds_scan = (size_t)P.sb_r * (4 + 4 + 2 + 2);
    uint16_t *dest = PRIO_PTR(uint16_t, P.o_dest), *gdest = PRIO_PTR(uint16_t, P.o_gdest);
    uint32_t *loc = PRIO_PTR(uint32_t, P.o_loc), *gloc = PRIO_PTR(uint32_t, P.o_gloc);
    uint32_t *in = PRIO_PTR(uint32_t, P.o_in), *gin = PRIO_PTR(uint32_t, P.o_gin), *summary = PRIO_PTR(uint32_t, P.o_sum);
    const uint32_t first = P.first, nb = P.NB - first, sb = P.sb;
    hipError_t e;
    hipLaunchKernelGGL(k_prio_reset, dim3(1), dim3(1), 0, s, summary);
    if (ev3 && (e = hipEventRecord(ev3[0], s)) != hipSuccess) return e;
    /* which blocks' entry cells does this scan change?  A block whose cells stay what its last sweep started from is not
     * swept again (k_prio_fwd), one whose gates that sweep did not flip keeps its map (k_prio_back).  Nothing on S1 (one
     * round of blocks: a sweep is one wavefront's latency whatever their number); on 1 GiB the tail iterations touch a
     * few per cent of the 16 K blocks. */
    uint32_t *gates_changed = PRIO_PTR(uint32_t, P.o_dirty), *in_changed = gates_changed + P.NB + 2;
    /* only where it pays: with at most one round of blocks in flight (9 wavefronts per CU) a sweep is one wavefront's
     * latency whatever the number of blocks, and comparing the rows costs the scans 0.4 ms per 100 MB */
    const char *sk = getenv("LZ77X_PRIO_SKIP");
    const bool track = P.W == 64u && P.sweeps > 0 && (sk ? atoi(sk) != 0 : P.NB > 2304u);
    if (track) {
        if ((e = hipMemsetAsync(in_changed + first, 0, (size_t)nb * 4, s)) != hipSuccess) return e;
        if (P.in0_dirty && first == 0 && (e = hipMemsetAsync(in_changed, 1, 4, s)) != hipSuccess) return e;
    } else if (P.sweeps > 0 && P.W == 64u) {
        /* every block is swept; which blocks' gates the last sweep flipped stays as it is: a block without a flip keeps
         * its map (k_prio_back2 is bound by what it reads, not by one block's latency: the tail iterations flip gates in
         * a few dozen blocks) */
        if ((e = hipMemsetAsync(in_changed, 1, ((size_t)P.NB + 2) * 4, s)) != hipSuccess) return e;
    }
    const uint32_t have_prev = P.sweeps > 0 ? 1u : 0u;                   /* (a workgroup sweep keeps its own record of what changed) */
    P.in0_dirty = false;
    P.sweeps++;
    uint32_t *changed = track ? in_changed : nullptr;
    if (nb > 1 && sb > 4096u) {
        if ((e = lz77kw_scan(dest, loc, in, sb, P.rs, first, nb - 1u, P.G, gdest, gloc, gin, PRIO_PTR(uint8_t, P.o_scan), s)) != hipSuccess) return e;
    } else if (nb > 1) {
        /* maps first .. NB-2 */
        const uint32_t nmaps = nb - 1, G = P.G;
        const uint32_t NG = (nmaps + G - 1u) / G;
        const uint32_t NG2 = (NG + G - 1u) / G;
        if (NG2 > 1) {
            /* three levels: groups of G maps, groups of G group maps, and those in sequence -- 5 G map applications one after
             * the other instead of the 3 sqrt(NB) of two levels (1526 blocks: 60 instead of 117, each ~1.4 us) */
            uint16_t *g2dest = PRIO_PTR(uint16_t, P.o_g2dest);
            uint32_t *g2loc = PRIO_PTR(uint32_t, P.o_g2loc), *g2in = PRIO_PTR(uint32_t, P.o_g2in);
            hipLaunchKernelGGL(k_prio_scan_compose, dim3(NG), dim3(PRIO_SCAN_BLOCK), lds_scan, s, dest, loc, sb, P.sb_r, first, nmaps, G, gdest, gloc);
            hipLaunchKernelGGL(k_prio_scan_compose, dim3(NG2), dim3(PRIO_SCAN_BLOCK), lds_scan, s, gdest, gloc, sb, P.sb_r, 0u, NG, G, g2dest, g2loc);
            /* g2in[j] = input of super-group j; gin[g] = input of group g; in[first + m + 1] = the cells after map m */
            hipLaunchKernelGGL(k_prio_scan_replay<false>, dim3(1), dim3(PRIO_SCAN_BLOCK), lds_scan, s, g2dest, g2loc, sb, P.sb_r, (size_t)0, NG2 - 1u, NG2,
                               in + (size_t)first * sb, (size_t)0, g2in, (size_t)0, 1u, (uint32_t *)nullptr);
            hipLaunchKernelGGL(k_prio_scan_replay<false>, dim3(NG2), dim3(PRIO_SCAN_BLOCK), lds_scan, s, gdest, gloc, sb, P.sb_r, (size_t)0, NG - 1u, G,
                               g2in, (size_t)sb, gin, (size_t)0, 1u, (uint32_t *)nullptr);
            if (changed)
                hipLaunchKernelGGL(k_prio_scan_replay<true>, dim3(NG), dim3(PRIO_SCAN_BLOCK), lds_scan, s, dest, loc, sb, P.sb_r, (size_t)first, nmaps, G, gin,
                                   (size_t)sb, in, (size_t)first, 0u, changed);
            else
                hipLaunchKernelGGL(k_prio_scan_replay<false>, dim3(NG), dim3(PRIO_SCAN_BLOCK), lds_scan, s, dest, loc, sb, P.sb_r, (size_t)first, nmaps, G, gin,
                                   (size_t)sb, in, (size_t)first, 0u, changed);
        } else if (NG > 1) {
            hipLaunchKernelGGL(k_prio_scan_compose, dim3(NG), dim3(PRIO_SCAN_BLOCK), lds_scan, s, dest, loc, sb, P.sb_r, first, nmaps, G, gdest, gloc);
            /* gin[g] = input of group g: replay the group maps from in[first] */
            hipLaunchKernelGGL(k_prio_scan_replay<false>, dim3(1), dim3(PRIO_SCAN_BLOCK), lds_scan, s, gdest, gloc, sb, P.sb_r, (size_t)0, NG - 1u, NG,
                               in + (size_t)first * sb, (size_t)0, gin, (size_t)0, 1u, (uint32_t *)nullptr);
            if (changed)
                hipLaunchKernelGGL(k_prio_scan_replay<true>, dim3(NG), dim3(PRIO_SCAN_BLOCK), lds_scan, s, dest, loc, sb, P.sb_r, (size_t)first, nmaps, G, gin,
                                   (size_t)sb, in, (size_t)first, 0u, changed);
            else
                hipLaunchKernelGGL(k_prio_scan_replay<false>, dim3(NG), dim3(PRIO_SCAN_BLOCK), lds_scan, s, dest, loc, sb, P.sb_r, (size_t)first, nmaps, G, gin,
                                   (size_t)sb, in, (size_t)first, 0u, changed);
        } else {
            if (changed)
                hipLaunchKernelGGL(k_prio_scan_replay<true>, dim3(1), dim3(PRIO_SCAN_BLOCK), lds_scan, s, dest, loc, sb, P.sb_r, (size_t)first, nmaps, nmaps,
                                   in + (size_t)first * sb, (size_t)0, in, (size_t)first, 0u, changed);
            else
                hipLaunchKernelGGL(k_prio_scan_replay<false>, dim3(1), dim3(PRIO_SCAN_BLOCK), lds_scan, s, dest, loc, sb, P.sb_r, (size_t)first, nmaps, nmaps,
                                   in + (size_t)first * sb, (size_t)0, in, (size_t)first, 0u, changed);
        }
    }
    if (ev3 && (e = hipEventRecord(ev3[1], s)) != hipSuccess) return e;
    if (P.W > 64u) {
        if ((e = lz77kw_fwd(P.ps, P.nx, sb, P.rs, P.B, P.ring_n, P.W, first, nb, PRIO_PTR(uint64_t, P.o_rmask), PRIO_PTR(uint64_t, P.o_cmask), PRIO_PTR(uint64_t, P.o_gate[P.cur]),
                            PRIO_PTR(uint64_t, P.o_gate[P.cur ^ 1]), in, P.xval, summary, P.voff, d_out_state, P.ncarried,
                            P.pack18 ? PRIO_PTR(uint32_t, P.o_codes) : nullptr, P.pack18 ? PRIO_PTR(uint32_t, P.o_gval) : nullptr,
                            PRIO_PTR(uint32_t, P.o_inprev), have_prev, gates_changed, s)) != hipSuccess)
            return e;
    } else
    hipLaunchKernelGGL(k_prio_fwd<true>, dim3(nb), dim3(64), lds_fwd, s, P.ps, P.nx, sb, P.B, P.ring_n, first, PRIO_PTR(uint64_t, P.o_rmask),
                       PRIO_PTR(uint64_t, P.o_cmask), PRIO_PTR(uint64_t, P.o_gate[P.cur]), PRIO_PTR(uint64_t, P.o_gate[P.cur ^ 1]), in, P.xval, summary, P.voff, d_out_state,
                       gates_changed, (const uint32_t *)in_changed);
    if (ev3 && (e = hipEventRecord(ev3[2], s)) != hipSuccess) return e;
    return lz77k_publish(h_flag, summary, 3, s);              /* (h_flag: pinned by hipHostMalloc -- every caller passes its context's h_small) */
}

/* after the stream has drained: take the sweep's gates as current.  Gates before the first flip are final
 * (a block's sweep is exact once the blocks before it are): those blocks keep their xval and in[] and are
 * not visited again; both gate buffers agree on that prefix (a block without a flip wrote back what it read).
 * restart_at0: the cells this plan starts from may still change (a shard whose predecessors have not
 * converged): nothing of it is final yet. */
void lz77k_prio_advance(lz77k_prio_plan &P, const uint32_t *h_flag, bool restart_at0)
{
    if (P.nx == 0) return;
    if (h_flag[0] == 0) { if (restart_at0) P.first = 0; return; }   /* the gate buffers are equal from `first` on: keep cur */
    P.first = restart_at0 ? 0u : h_flag[1];
    P.cur ^= 1;
}

/* xval[x] for x < nx from ps[] (distances P | S << 16).  h_flag: 8 bytes of pinned host memory.  Returns
 * hipSuccess with *converged = 0 when max_iters did not suffice (the caller then takes the host path). */
hipError_t lz77k_prio(const uint32_t *d_ps, uint32_t nx, int sb_i, uint32_t *d_xval, void *d_tmp, hipStream_t s,
                      uint32_t *h_flag, int max_iters, int *iters, int *converged, hipEvent_t *ev4, float *ms3,
                      uint32_t voff, const uint32_t *d_carried, uint32_t *d_out_state)
{
    *iters = 0;
    *converged = 1;
    const uint32_t sb = (uint32_t)sb_i;
    if (nx == 0) {
        if (d_out_state) hipLaunchKernelGGL(k_prio_in0, dim3((sb + 255u) / 256u), dim3(256), 0, s, d_out_state, sb, voff, d_carried);
        return hipGetLastError();
    }
    lz77k_prio_plan P;
    hipError_t e;
    if ((e = lz77k_prio_begin(P, d_ps, nx, sb_i, d_xval, d_tmp, voff, d_carried, s)) != hipSuccess) return e;
    /* the last six iterations' flips and first flipped blocks (lz77x_prio_hopeless) */
    uint64_t hist_f[6] = {0, 0, 0, 0, 0, 0}, hist_b[6] = {0, 0, 0, 0, 0, 0};
    for (int it = 0;; it++) {
        if (it >= max_iters) { *converged = 0; break; }
        if (ev4 && (e = hipEventRecord(ev4[3], s)) != hipSuccess) return e;
        if ((e = lz77k_prio_maps(P, s, false, nullptr, nullptr)) != hipSuccess) return e;
        if ((e = lz77k_prio_sweep(P, s, h_flag, d_out_state, ev4)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        *iters = it + 1;
        if (ev4 && ms3) {
            float t = 0;
            if ((e = hipEventElapsedTime(&t, ev4[1], ev4[2])) != hipSuccess) return e;
            ms3[0] += t;
            if ((e = hipEventElapsedTime(&t, ev4[3], ev4[0])) != hipSuccess) return e;
            ms3[1] += t;
            if ((e = hipEventElapsedTime(&t, ev4[0], ev4[1])) != hipSuccess) return e;
            ms3[2] += t;
        }
        if (P.W > 64u) lz77kw_debug_dump();
        if (getenv("LZ77X_PRIO_TRACE"))
            fprintf(stderr, "prio it %d: B %u NB %u first %u flips %u min flipped block %u max %u\n", it, P.B, P.NB, P.first, h_flag[0], h_flag[1], h_flag[2]);
        if (h_flag[0] == 0) break;
        /* On input that repeats with a period of about a window (a repeated random block of 4096 bytes; rows of an image) the
         * iteration repairs a block or a few per iteration -- an error front -- or, with noise in the repeats, decays by a
         * tenth per iteration everywhere: NB iterations, or a hundred.  When the last six iterations say that
         * the budget will not do (lz77x_prio_hopeless), give up NOW and leave the recurrence to the sequential form (the
         * caller's fallback) instead of after max_iters sweeps of the whole input.  Only where the caller has a budget. */
        for (int q = 0; q < 5; q++) { hist_f[q] = hist_f[q + 1]; hist_b[q] = hist_b[q + 1]; }
        hist_f[5] = h_flag[0];
        hist_b[5] = h_flag[1];
        if (lz77x_prio_hopeless(hist_f, hist_b, P.NB > h_flag[1] ? P.NB - h_flag[1] : 0u, it + 1, max_iters)) {
            if (getenv("LZ77X_PRIO_TRACE")) fprintf(stderr, "prio: %u flips, first open block %u of %u: the budget of %d iterations will not do; giving up after %d\n", h_flag[0], h_flag[1], P.NB, max_iters, it + 1);
            *converged = 0;
            break;
        }
        lz77k_prio_advance(P, h_flag, false);
    }
    return hipGetLastError();
}
