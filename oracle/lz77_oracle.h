/*
 * lz77_oracle.h -- CPU restatement of the cstdvd/lz77 hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lz77_amd/ (the product) may include,
 * link, dlopen or execute this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / reported CPU
 * baseline -- never as the thing measured or shipped.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_*.py)
 * against (a) the known-answer vectors of SURVEY.md Appendix C, (b) golden
 * fixtures under tests/golden/ produced by the compiled reference
 * (oracle/_ref/lz77_ref, recipe in oracle/Makefile, generator in
 * tests/golden/make_golden.py), and (c) when oracle/_ref is present, the
 * reference itself run live on seeded random inputs.
 *
 * All functions work on flat in-memory buffers (the reference streams through a
 * 3*SB+LA window, lz77.c:67,113-129; the flat model is observably identical,
 * SURVEY.md A.2).
 */
#ifndef LZ77_ORACLE_H
#define LZ77_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LZ77O_NONE32 0xFFFFFFFFu

/* bitio.c:41-43  bitof(n) = (int)ceil(log(n)/log(2)); integer restatement, n>=1 */
int lz77o_bitof(int n);

/* 4 + ceil(n*T/8): the most bytes encode can emit (every token a bare literal) */
size_t lz77o_bound(size_t n, int sb, int la);

/* lz77.c:51-140 + tree.c:62-243 + bitio.c:203-239, array-backed BST with
 * delete-by-successor: same algorithmic class as the reference (CPU baseline).
 * Returns bytes written, or (size_t)-1 if cap is too small / bad arguments. */
size_t lz77o_encode_bst(const uint8_t *in, size_t n, int sb, int la,
                        uint8_t *out, size_t cap);

/* Same stream, derived a second, independent way (SURVEY.md A.3 + A.5):
 * brute-force longest match + treap-priority tie-break.  O(n*SB*LA): small n only. */
size_t lz77o_encode_model(const uint8_t *in, size_t n, int sb, int la,
                          uint8_t *out, size_t cap);

/* lz77.c:148-197,260-283 + bitio.c:256-298.  Returns decoded size.  out==NULL
 * (cap ignored) just counts.  (size_t)-1 if zn<4 or cap too small. */
size_t lz77o_decode(const uint8_t *z, size_t zn, uint8_t *out, size_t cap);

/* Token-level view of a stream: fills off/len/next (each may be NULL) for up to
 * cap tokens, returns ntok = floor((8*zn-32)/T); *sb,*la from the header. */
size_t lz77o_tokens(const uint8_t *z, size_t zn, int *sb, int *la,
                    int32_t *off, int32_t *len, uint8_t *next, size_t cap);

/* ---- intermediates of the parallel formulation (for kernel-level parity) ---- */

/* A.3: maxlen[p] = max_c lcp(c,p) over c in [max(0,p-SB),p-1], capped at min(LA,n-p)-1 */
void lz77o_maxlen(const uint8_t *in, size_t n, int sb, int la, uint8_t *maxlen);

/* A.5 stage A by exhaustive pair scan: in-order predecessor / successor of x among
 * y in [x+1, x+SB-1], as distances y-x (0 = none), for x < n-SB (others 0). */
void lz77o_stage_a(const uint8_t *in, size_t n, int sb, int la,
                   uint16_t *P, uint16_t *S);

/* Same quantities read off the live BST at each eviction (tree.c:182-243):
 * independent of lz77o_stage_a.  two[x]=1 iff the evicted node had two children. */
void lz77o_stage_a_tree(const uint8_t *in, size_t n, int sb, int la,
                        uint16_t *P, uint16_t *S, uint8_t *two);

/* A.5 stage B: sequential priority recurrence.  xval[x] = priority handed to
 * S[x] when x is evicted, or LZ77O_NONE32.  Returns number of transfers. */
size_t lz77o_stage_b(const uint16_t *P, const uint16_t *S, size_t n, int sb,
                     uint32_t *xval);

/* splitmix64 byte stream, for fixtures shared by C and Python */
void lz77o_splitmix_fill(uint64_t seed, uint8_t *dst, size_t n);

#ifdef __cplusplus
}
#endif
#endif
