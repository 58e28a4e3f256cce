// Variants of the host priority recurrence (hoststage.c lz77x_prio_run), same results, different
// instruction mixes.  gcc -O3 -march=native prio_variants.c -o prio_variants.bin ; ./prio_variants.bin ps.bin N
// ps.bin: P | S<<16 distances per position (tests/ubench/mkps.py).  sb = 4095.
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <time.h>
#include <immintrin.h>
#define SB 4095u
#define NONE 0xFFFFFFFFu
static double now(void){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+t.tv_nsec*1e-9;}

#define SELECT(mine, pp, sp, ns, xv) do { uint32_t mn_ = (pp); (ns) = (sp); (xv) = NONE; \
    __asm__("cmpl %[s], %[m]\n\tcmoval %[s], %[m]" : [m] "+r"(mn_) : [s] "r"(sp) : "cc"); \
    __asm__("cmpl %[m], %[i]\n\tcmovbl %[i], %[n]\n\tcmovbl %[i], %[x]" : [n] "+r"(ns), [x] "+r"(xv) : [m] "r"(mn_), [i] "r"(mine) : "cc"); } while (0)

// A: as shipped: ring of 4096 cells, insert store every step
static void run_a(const uint32_t *restrict cells, size_t n, uint32_t *restrict xval, uint32_t *restrict ring)
{
    const uint32_t mask = 4095; size_t t = 0;
    for (; t < n && t < SB; t++) ring[t & mask] = (uint32_t)t;
#define STEP(T) do { const uint32_t v = cells[(T) - SB]; const uint32_t sidx = v >> 16; \
        const uint32_t mine = ring[(uint32_t)((T) - SB) & mask]; const uint32_t pp = ring[v & 0xFFFFu]; const uint32_t sp = ring[sidx]; \
        uint32_t ns, xv; SELECT(mine, pp, sp, ns, xv); ring[sidx] = ns; _mm_stream_si32((int *)&xval[(T) - SB], (int)xv); \
        ring[(uint32_t)(T) & mask] = (uint32_t)(T); } while (0)
    for (; t + 2 <= n; t += 2) { STEP(t); STEP(t + 1); }
    for (; t < n; t++) STEP(t);
#undef STEP
    _mm_sfence();
}
// B: ring of 8192 cells: the cell of position T+j (j < 4096) was last used by T+j-8192, long evicted, so the
// natural priorities of the next 2048 positions are written ahead of time with vector stores
static void run_b(const uint32_t *restrict cells, size_t n, uint32_t *restrict xval, uint32_t *restrict ring, int prefetch)
{
    const uint32_t mask = 8191; size_t t = 0, filled = 0;
#define FILL(upto) do { for (; filled < (upto); filled += 8) { \
        const __m256i v = _mm256_add_epi32(_mm256_set1_epi32((int)filled), _mm256_setr_epi32(0,1,2,3,4,5,6,7)); \
        _mm256_store_si256((__m256i *)&ring[filled & mask], v); } } while (0)
    FILL(SB + 2048 < n + 8 ? SB + 2048 : ((n + 7) & ~(size_t)7));
    t = n < SB ? n : SB;
#define STEP(T) do { const uint32_t v = cells[(T) - SB]; const uint32_t sidx = v >> 16; \
        const uint32_t mine = ring[(uint32_t)((T) - SB) & mask]; const uint32_t pp = ring[v & 0xFFFFu]; const uint32_t sp = ring[sidx]; \
        uint32_t ns, xv; SELECT(mine, pp, sp, ns, xv); ring[sidx] = ns; _mm_stream_si32((int *)&xval[(T) - SB], (int)xv); } while (0)
    while (t < n) {
        size_t blk = t + 2048 < n ? t + 2048 : n;
        /* positions < blk + 2048 must be in the ring before any step < blk can name them (x+sb-1 < T+sb) */
        FILL(blk + 2048 < ((n + 7) & ~(size_t)7) ? blk + 2048 : ((n + 7) & ~(size_t)7));
        for (; t + 2 <= blk; t += 2) {
            if (prefetch) _mm_prefetch((const char *)&cells[t - SB + 256], _MM_HINT_NTA);
            STEP(t); STEP(t + 1);
        }
        for (; t < blk; t++) STEP(t);
    }
#undef STEP
#undef FILL
    _mm_sfence();
}
// D: A with the loop unrolled four times and the cell stream prefetched
static void run_d(const uint32_t *restrict cells, size_t n, uint32_t *restrict xval, uint32_t *restrict ring)
{
    const uint32_t mask = 4095; size_t t = 0;
    for (; t < n && t < SB; t++) ring[t & mask] = (uint32_t)t;
#define STEP(T) do { const uint32_t v = cells[(T) - SB]; const uint32_t sidx = v >> 16; \
        const uint32_t mine = ring[(uint32_t)((T) - SB) & mask]; const uint32_t pp = ring[v & 0xFFFFu]; const uint32_t sp = ring[sidx]; \
        uint32_t ns, xv; SELECT(mine, pp, sp, ns, xv); ring[sidx] = ns; _mm_stream_si32((int *)&xval[(T) - SB], (int)xv); \
        ring[(uint32_t)(T) & mask] = (uint32_t)(T); } while (0)
    for (; t + 4 <= n; t += 4) { _mm_prefetch((const char *)&cells[t - SB + 256], _MM_HINT_NTA); STEP(t); STEP(t + 1); STEP(t + 2); STEP(t + 3); }
    for (; t < n; t++) STEP(t);
#undef STEP
    _mm_sfence();
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    size_t n = strtoull(argv[2], 0, 10);
    uint32_t *ps = aligned_alloc(64, n * 4 + 64), *c4 = aligned_alloc(64, n * 4 + 64), *c8 = aligned_alloc(64, n * 4 + 64);
    uint32_t *xa = aligned_alloc(64, n * 4 + 64), *xb = aligned_alloc(64, n * 4 + 64), *ring = aligned_alloc(64, 8192 * 4);
    FILE *f = fopen(argv[1], "rb"); if (!f || fread(ps, 4, n, f) != n) return 1; fclose(f);
    for (size_t i = 0; i < n; i++) { uint32_t v = ps[i], x = (uint32_t)i;
        c4[i] = ((x + (v & 0xFFFF)) & 4095) | (((x + (v >> 16)) & 4095) << 16);
        c8[i] = ((x + (v & 0xFFFF)) & 8191) | (((x + (v >> 16)) & 8191) << 16); }
    memset(xa, 0, n * 4); memset(xb, 0, n * 4);
    const size_t nx = n - SB;
    for (int it = 0; it < 3; it++) {
        double t = now(); run_a(c4, n, xa, ring); double da = now() - t;
        t = now(); run_b(c8, n, xb, ring, 0); double db = now() - t;
        int okb = memcmp(xa, xb, nx * 4) == 0;
        t = now(); run_b(c8, n, xb, ring, 1); double dc = now() - t;
        int okc = memcmp(xa, xb, nx * 4) == 0;
        t = now(); run_d(c4, n, xb, ring); double dd = now() - t;
        int okd = memcmp(xa, xb, nx * 4) == 0;
        printf("A shipped %.3f | B ring8192+bulk insert %.3f (%s) | C =B+prefetch %.3f (%s) | D unroll4+prefetch %.3f (%s)  ns/pos\n",
               da / n * 1e9, db / n * 1e9, okb ? "same" : "DIFF", dc / n * 1e9, okc ? "same" : "DIFF", dd / n * 1e9, okd ? "same" : "DIFF");
    }
    return 0;
}
