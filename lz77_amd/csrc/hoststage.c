/*
 * hoststage.c -- host-side helpers of the hot path (plain C): the token geometry (bitof, regions) and the
 * SEQUENTIAL form of the two recurrences of the reference,
 *   - the greedy parse chain  p <- p + len + 1                      (lz77.c:89-98)
 *   - which node a tree.c:182-243 delete() promotes, restated as a priority
 *     hand-over between a position and its in-order successor       (SURVEY A.5 stage B).
 * Since round 3 both run on the DEVICE for every window size (k_chain.hip, k_prio.hip, k_priow.hip); the loops
 * here are what the host-assisted pipeline (encode_core_host in encode_host.cpp) runs -- the fallback when the
 * device's gate iteration does not converge -- and the lz77x_stage_priorities cross-check of the parity tests.  O(1) per byte on per-position results of the match
 * kernels.
 */
#include "lz77x_internal.h"
#include <stdlib.h>
#include <string.h>
#if defined(__SSE2__)
#include <emmintrin.h>
/* xval is written once, front to back, and next read by the DMA engine: a streaming store skips the
 * read-for-ownership of every fresh cache line */
#define LZ77X_STREAM_STORE(p, v) _mm_stream_si32((int *)(p), (int)(v))
#define LZ77X_STREAM_FENCE() _mm_sfence()
#else
#define LZ77X_STREAM_STORE(p, v) (*(p) = (v))
#define LZ77X_STREAM_FENCE() ((void)0)
#endif

/* bitio.c:41-43: (int)ceil(log(n)/log(2)) for n >= 1, in integers */
int lz77x_bitof(int n)
{
    int b = 0;
    while (b < 31 && (1 << b) < n) b++;
    return b;
}

void lz77x_make_geom(lz77x_geom *g, int sb, int la)
{
    g->sb = sb;
    g->la = la;
    g->ob = lz77x_bitof(sb);
    g->lb = lz77x_bitof(la);
    g->T = g->ob + g->lb + 8;                       /* lz77.c:249-251 */
    g->SBu = ((uint32_t)sb + 7u) & ~7u;
    uint32_t rp = 4096;
    while (rp < 4u * g->SBu) rp <<= 1;
    g->RP = rp;
    g->TILE = rp - g->SBu;
    /* large windows (sb > 32768): tiles of whole 64 K blocks (three of the region's four), so that the regions are
     * unions of globally aligned blocks and share one hierarchical sort (k_big_*): up to 13 % more regions against
     * half the sort.  (At RP 131072 the tile would shrink from ~100 K to 64 K: measured at sb 32768, the walkers
     * lose more, 24 -> 35 ms per 100 MB, than the sort gains, 33 -> 28.) */
    if (rp >= 262144u) g->TILE = (rp - g->SBu) / 65536u * 65536u;
    g->shifted = 1;
    g->fast = rp <= 16384u;
}

void lz77x_geom_legacy(lz77x_geom *g)
{
    g->TILE = (g->RP - g->SBu - (uint32_t)g->sb) & ~7u;
    g->shifted = 0;
}

size_t lz77x_host_chain(const uint8_t *maxlen, size_t limit, size_t p, uint32_t *chain, size_t *ntok)
{
    size_t k = *ntok;
    while (p < limit) {
        chain[k++] = (uint32_t)p;
        p += (size_t)maxlen[p] + 1;
    }
    *ntok = k;
    return p;
}

uint32_t lz77x_prio_mask(int sb)
{
    uint32_t size = 1;
    while (size < (uint32_t)sb + 1u) size <<= 1;
    return size - 1;
}

int lz77x_prio_init(lz77x_prio_state *st, int sb)
{
    const uint32_t size = lz77x_prio_mask(sb) + 1u;
    st->ring = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)size);
    st->mask = size - 1;
    st->next = 0;
    st->transfers = 0;
    return st->ring != NULL;
}

void lz77x_prio_free(lz77x_prio_state *st)
{
    free(st->ring);
    st->ring = NULL;
}

/* Inserting position t (priority t) is preceded, once the window is full, by evicting
 * x = t - sb.  tree.c:202-231: a node with two children is replaced by its in-order
 * successor, which thereby inherits the node's place (priority); a node with fewer
 * children just splices out.  "x has two children" <=> both in-order neighbours lie in
 * x's subtrees <=> both have larger priority than x. */
void lz77x_prio_run(lz77x_prio_state *st, const uint32_t *restrict ps, int sb, size_t upto, uint32_t *restrict xval)
{
    uint32_t *restrict ring = st->ring;
    const uint32_t mask = st->mask;
    const size_t usb = (size_t)sb;
    size_t t = st->next;
    for (; t < upto && t < usb; t++) ring[t & mask] = (uint32_t)t;
    /* branch-free on purpose: `go` is taken ~2/3 of the time with no pattern, a compiled branch here
     * mispredicts every third position.  ps[x] holds the ring cells ((x+P)&mask, (x+S)&mask) of x's
     * neighbours; a missing neighbour has distance 0, i.e. x's own cell, and mine < mine is false. */
    /* go <=> mine < min(pp, sp).  On x86-64 the two selects are written as cmp/cmov by hand: the
     * compiler's own if-conversion either branches (mispredicting every third step) or spends
     * twice the instructions on mask arithmetic, and this loop runs at the core's issue limit. */
#if defined(__x86_64__) && defined(__GNUC__)
#define LZ77X_SELECT(mine, pp, sp, ns, xv)                                              \
    do {                                                                                \
        uint32_t mn_ = (pp);                                                            \
        (ns) = (sp);                                                                    \
        (xv) = LZ77X_NONE32;                                                            \
        __asm__("cmpl %[s], %[m]\n\tcmoval %[s], %[m]" : [m] "+r"(mn_) : [s] "r"(sp) : "cc"); \
        __asm__("cmpl %[m], %[i]\n\tcmovbl %[i], %[n]\n\tcmovbl %[i], %[x]"             \
                : [n] "+r"(ns), [x] "+r"(xv) : [m] "r"(mn_), [i] "r"(mine) : "cc");     \
    } while (0)
#else
#define LZ77X_SELECT(mine, pp, sp, ns, xv)                                              \
    do {                                                                                \
        const uint64_t lt_ = ((uint64_t)(mine) - (uint64_t)(pp)) & ((uint64_t)(mine) - (uint64_t)(sp)); \
        const uint32_t m_ = (uint32_t)((int64_t)lt_ >> 63);      /* all ones iff mine < pp && mine < sp */ \
        (ns) = (sp) ^ (((sp) ^ (mine)) & m_);                                           \
        (xv) = (mine) | ~m_;                                                            \
    } while (0)
#endif
#define LZ77X_PRIO_STEP(T)                                                              \
    do {                                                                                \
        const uint32_t v = ps[(T) - usb];      /* ring cells of P and S, packed by k_ps_cells */ \
        const uint32_t sidx = v >> 16;                                                  \
        const uint32_t mine = ring[(uint32_t)((T) - usb) & mask];                       \
        const uint32_t pp = ring[v & 0xFFFFu];                                          \
        const uint32_t sp = ring[sidx];                                                 \
        uint32_t ns, xv;                                                                \
        LZ77X_SELECT(mine, pp, sp, ns, xv);                                             \
        ring[sidx] = ns;                                                                \
        LZ77X_STREAM_STORE(&xval[(T) - usb], xv);                  /* LZ77X_NONE32 when nothing moves */ \
        ring[(uint32_t)(T) & mask] = (uint32_t)(T);                                     \
    } while (0)
    /* tests/ubench/prio_variants.c, on the GPU box's EPYC 9575F: a software prefetch of the cell stream
     * is worth 5 % standalone and nothing inside the pipeline; a double-size ring with vector-filled
     * natural priorities (no insert store) is worth nothing either -- the loop is neither store- nor
     * stream-bound */
    for (; t + 4 <= upto; t += 4) {
        LZ77X_PRIO_STEP(t);
        LZ77X_PRIO_STEP(t + 1);
        LZ77X_PRIO_STEP(t + 2);
        LZ77X_PRIO_STEP(t + 3);
    }
    for (; t < upto; t++) LZ77X_PRIO_STEP(t);
#undef LZ77X_PRIO_STEP
#undef LZ77X_SELECT
    LZ77X_STREAM_FENCE();
    st->next = t;
}

/* The same recurrence on what the DEVICE pipeline holds: ps[x] = P | S << 16 as distances (0 = no such neighbour), the sb
 * cells a segment starts from (cells_in: the carried ranks of a later segment of a long input, or NULL: every cell its own
 * position + voff) and the sb cells it leaves behind (cells_out, may be NULL) -- k_prio_fwd's exact sweep as one plain loop.
 * encode_pipe.cpp comes here when the device's gate iteration meets an error front (k_prio.hip, lz77k_prio) in a segment of
 * a multi-segment input, where starting the whole encode over on the host-assisted pipeline is not an option; about 2 ns a
 * step, against an iteration per block of the segment. */
int lz77x_prio_run_cells(const uint32_t *restrict ps, size_t nx, int sb, const uint32_t *cells_in, uint32_t voff, uint32_t *restrict xval,
                         uint32_t *cells_out)
{
    const uint32_t mask = lz77x_prio_mask(sb);
    const size_t usb = (size_t)sb;
    uint32_t *ring = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)mask + 1));
    if (!ring) return 0;
    for (size_t i = 0; i < usb; i++) ring[i & mask] = cells_in ? cells_in[i] : (uint32_t)i + voff;
    for (size_t x = 0; x < nx; x++) {
        const uint32_t v = ps[x], p = v & 0xFFFFu, s = v >> 16;
        const uint32_t mine = ring[x & mask];
        uint32_t out = LZ77X_NONE32;
        if (p && s) {
            const uint32_t pp = ring[(x + p) & mask], sp = ring[(x + s) & mask];
            if (mine < pp && mine < sp) {                 /* tree.c:202-231: both children hang below x -- the successor takes its place */
                ring[(x + s) & mask] = mine;
                out = mine;
            }
        }
        xval[x] = out;
        ring[(x + usb) & mask] = (uint32_t)(x + usb) + voff;    /* tree.c:102-105: the new node is a leaf */
    }
    if (cells_out)
        for (size_t i = 0; i < usb; i++) cells_out[i] = ring[(nx + i) & mask];
    free(ring);
    return 1;
}
