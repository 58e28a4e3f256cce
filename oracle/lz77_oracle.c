/*
 * lz77_oracle.c -- CPU restatement of the cstdvd/lz77 hot path (see lz77_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: checker + reported CPU baseline, never the product.
 * Parity status: PINNED against the compiled reference (tests/golden/, oracle/_ref).
 *
 * Written from the behavioural spec in SURVEY.md Appendix A; each function cites
 * the reference lines whose observable behaviour it restates.  Everything works
 * on flat buffers and absolute positions; there is no sliding window here.
 */
#include "lz77_oracle.h"
#include <stdlib.h>
#include <string.h>

#define BAD ((size_t)-1)

/* ------------------------------------------------------------------ bits ---- */

/* bitio.c:41-43: (int)ceil(log(n)/log(2)) == smallest b with 2^b >= n, for n>=1
 * (identical to libm's answer on [1,65535]). */
int lz77o_bitof(int n)
{
    int b = 0;
    while (b < 31 && (1 << b) < n) b++;
    return b;
}

static int token_bits(int sb, int la) { return lz77o_bitof(sb) + lz77o_bitof(la) + 8; }

size_t lz77o_bound(size_t n, int sb, int la)
{
    return 4 + (n * (size_t)token_bits(sb, la) + 7) / 8;
}

/* bitio.c:203-239: stream bit k is bit (k&7) of byte (k>>3); a field's bit i goes
 * to stream bit base+i.  bitio.c:180-182: final partial byte is zero padded. */
typedef struct {
    uint8_t *dst;
    size_t cap, used;
    uint64_t acc;
    int fill;
    int overflow;
} bitsink;

static void sink_put(bitsink *s, uint32_t v, int nbits)
{
    if (nbits <= 0) return;
    if (nbits < 32) v &= (1u << nbits) - 1u;
    s->acc |= (uint64_t)v << s->fill;
    s->fill += nbits;
    while (s->fill >= 8) {
        if (s->used < s->cap) s->dst[s->used] = (uint8_t)s->acc; else s->overflow = 1;
        s->used++;
        s->acc >>= 8;
        s->fill -= 8;
    }
}

static size_t sink_finish(bitsink *s)
{
    if (s->fill > 0) sink_put(s, 0, 8 - s->fill);
    return s->overflow ? BAD : s->used;
}

/* lz77.c:74-75 header (two 16-bit fields), lz77.c:246-252 token */
static void sink_header(bitsink *s, int sb, int la)
{
    sink_put(s, (uint32_t)sb, 16);
    sink_put(s, (uint32_t)la, 16);
}

static void sink_token(bitsink *s, int ob, int lb, int off, int len, uint8_t next)
{
    sink_put(s, (uint32_t)off, ob);
    sink_put(s, (uint32_t)len, lb);
    sink_put(s, next, 8);
}

/* bitio.c:256-298 for a field of w<=32 bits starting at stream bit `at` */
static uint32_t peek_bits(const uint8_t *z, size_t zn, uint64_t at, int w)
{
    uint64_t acc = 0;
    size_t b0 = (size_t)(at >> 3);
    for (int i = 0; i < 6 && b0 + (size_t)i < zn; i++) acc |= (uint64_t)z[b0 + i] << (8 * i);
    acc >>= (at & 7);
    return w >= 32 ? (uint32_t)acc : (uint32_t)(acc & ((1ull << w) - 1));
}

/* ------------------------------------------------------- match finder (BST) -- */

/* tree.c:23-27 node, as parallel arrays.  slot = position % SB (tree.c:66,187). */
typedef struct {
    const uint8_t *in;
    size_t n;
    int sb, la;
    int64_t *where;     /* absolute position stored in slot */
    int32_t *lo, *hi, *up;
    int32_t top;        /* root slot or -1 */
} finder;

static int finder_init(finder *f, const uint8_t *in, size_t n, int sb, int la)
{
    f->in = in; f->n = n; f->sb = sb; f->la = la; f->top = -1;
    f->where = malloc(sizeof(int64_t) * (size_t)sb);
    f->lo = malloc(sizeof(int32_t) * (size_t)sb);
    f->hi = malloc(sizeof(int32_t) * (size_t)sb);
    f->up = malloc(sizeof(int32_t) * (size_t)sb);
    return f->where && f->lo && f->hi && f->up;
}

static void finder_free(finder *f) { free(f->where); free(f->lo); free(f->hi); free(f->up); }

static inline int key_len(const finder *f, size_t p)
{
    size_t left = f->n - p;
    return left < (size_t)f->la ? (int)left : f->la;      /* lz77.c:87,134 */
}

/* tree.c:62-106: new position becomes a leaf; memcmp<0 goes left, otherwise right */
static void finder_add(finder *f, size_t p)
{
    int32_t s = (int32_t)(p % (size_t)f->sb);
    int klen = key_len(f, p);
    f->where[s] = (int64_t)p;
    f->lo[s] = f->hi[s] = -1;
    if (f->top < 0) { f->top = s; f->up[s] = -1; return; }
    int32_t at = f->top;
    for (;;) {
        int32_t *link = memcmp(f->in + p, f->in + f->where[at], (size_t)klen) < 0 ? &f->lo[at] : &f->hi[at];
        if (*link < 0) { *link = s; f->up[s] = at; return; }
        at = *link;
    }
}

/* tree.c:182-243: 0/1 child -> splice; 2 children -> in-order successor takes the place */
static void finder_drop(finder *f, size_t p)
{
    int32_t s = (int32_t)(p % (size_t)f->sb);
    int32_t l = f->lo[s], r = f->hi[s], u = f->up[s], heir;
    if (l < 0) {
        heir = r;
        if (heir >= 0) f->up[heir] = u;
    } else if (r < 0) {
        heir = l;
        f->up[heir] = u;
    } else {
        heir = r;
        while (f->lo[heir] >= 0) heir = f->lo[heir];          /* tree.c:162-170 */
        if (heir != r) {
            int32_t hp = f->up[heir], hr = f->hi[heir];
            f->lo[hp] = hr;
            if (hr >= 0) f->up[hr] = hp;
            f->hi[heir] = r;
            f->up[r] = heir;
        }
        f->lo[heir] = l;
        f->up[l] = heir;
        f->up[heir] = u;
    }
    if (u < 0) f->top = heir;
    else if (f->hi[u] == s) f->hi[u] = heir;
    else f->lo[u] = heir;
}

/* tree.c:118-152: walk down, lcp capped at size-1, strictly-longer replaces,
 * branch on the first differing byte, stop on equality or a missing child. */
static void finder_best(const finder *f, size_t p, int size, int *off, int *len)
{
    *off = 0; *len = 0;
    int32_t at = f->top;
    const uint8_t *q = f->in + p;
    while (at >= 0) {
        const uint8_t *c = f->in + f->where[at];
        int i = 0;
        while (i < size - 1 && q[i] == c[i]) i++;
        if (i > *len) { *len = i; *off = (int)((int64_t)p - f->where[at]); }
        if (q[i] < c[i]) at = f->lo[at];
        else if (q[i] > c[i]) at = f->hi[at];
        else break;
    }
}

/* in-order neighbours of slot s in the live tree, as absolute positions or -1 */
static int64_t finder_prev(const finder *f, int32_t s)
{
    if (f->lo[s] >= 0) { int32_t t = f->lo[s]; while (f->hi[t] >= 0) t = f->hi[t]; return f->where[t]; }
    int32_t c = s, u = f->up[s];
    while (u >= 0 && f->lo[u] == c) { c = u; u = f->up[u]; }
    return u >= 0 ? f->where[u] : -1;
}

static int64_t finder_next(const finder *f, int32_t s)
{
    if (f->hi[s] >= 0) { int32_t t = f->hi[s]; while (f->lo[t] >= 0) t = f->lo[t]; return f->where[t]; }
    int32_t c = s, u = f->up[s];
    while (u >= 0 && f->hi[u] == c) { c = u; u = f->up[u]; }
    return u >= 0 ? f->where[u] : -1;
}

static int args_ok(int sb, int la) { return sb >= 1 && sb <= 65535 && la >= 2 && la <= 255; }

/* lz77.c:89-136 greedy loop on the flat buffer */
size_t lz77o_encode_bst(const uint8_t *in, size_t n, int sb, int la, uint8_t *out, size_t cap)
{
    if (!args_ok(sb, la)) return BAD;
    finder f;
    if (!finder_init(&f, in, n, sb, la)) { finder_free(&f); return BAD; }
    bitsink s = { out, cap, 0, 0, 0, 0 };
    int ob = lz77o_bitof(sb), lb = lz77o_bitof(la);
    sink_header(&s, sb, la);
    size_t p = 0, oldest = 0, live = 0;
    while (p < n) {
        int off, len;
        finder_best(&f, p, key_len(&f, p), &off, &len);
        sink_token(&s, ob, lb, off, len, in[p + (size_t)len]);
        for (int i = 0; i <= len; i++) {
            if (live == (size_t)sb) finder_drop(&f, oldest++); else live++;   /* lz77.c:101-105 */
            finder_add(&f, p + (size_t)i);                                    /* lz77.c:108 */
        }
        p += (size_t)len + 1;
    }
    finder_free(&f);
    return sink_finish(&s);
}

void lz77o_stage_a_tree(const uint8_t *in, size_t n, int sb, int la,
                        uint16_t *P, uint16_t *S, uint8_t *two)
{
    memset(P, 0, n * sizeof *P);
    memset(S, 0, n * sizeof *S);
    if (two) memset(two, 0, n);
    if (!args_ok(sb, la)) return;
    finder f;
    if (!finder_init(&f, in, n, sb, la)) { finder_free(&f); return; }
    for (size_t t = 0; t < n; t++) {
        if (t >= (size_t)sb) {
            size_t x = t - (size_t)sb;
            int32_t s = (int32_t)(x % (size_t)sb);
            int64_t a = finder_prev(&f, s), b = finder_next(&f, s);
            P[x] = a < 0 ? 0 : (uint16_t)((size_t)a - x);
            S[x] = b < 0 ? 0 : (uint16_t)((size_t)b - x);
            if (two) two[x] = (uint8_t)(f.lo[s] >= 0 && f.hi[s] >= 0);
            finder_drop(&f, x);
        }
        finder_add(&f, t);
    }
    finder_free(&f);
}

/* ------------------------------------------------ parallel formulation ------- */

static inline int lcp_cap(const uint8_t *a, const uint8_t *b, int cap)
{
    int i = 0;
    while (i < cap && a[i] == b[i]) i++;
    return i;
}

/* SURVEY A.3 (follows from tree.c:136,139): exhaustive, history free */
void lz77o_maxlen(const uint8_t *in, size_t n, int sb, int la, uint8_t *maxlen)
{
    for (size_t p = 0; p < n; p++) {
        size_t left = n - p;
        int cap = (int)(left < (size_t)la ? left : (size_t)la) - 1;
        size_t c0 = p > (size_t)sb ? p - (size_t)sb : 0;
        int best = 0;
        for (size_t c = c0; c < p && best < cap; c++) {
            int l = lcp_cap(in + c, in + p, cap);
            if (l > best) best = l;
        }
        maxlen[p] = (uint8_t)best;
    }
}

/* in-order relation of two live positions a<b (tree.c:77: key length is the later
 * position's lookahead size; ties keep insertion order) */
static inline int goes_before(const uint8_t *in, size_t n, int la, size_t a, size_t b)
{
    size_t left = n - b;
    size_t klen = left < (size_t)la ? left : (size_t)la;
    return memcmp(in + a, in + b, klen) <= 0;
}

static inline int in_order(const uint8_t *in, size_t n, int la, size_t u, size_t v)
{
    return u < v ? goes_before(in, n, la, u, v) : !goes_before(in, n, la, v, u);
}

void lz77o_stage_a(const uint8_t *in, size_t n, int sb, int la, uint16_t *P, uint16_t *S)
{
    memset(P, 0, n * sizeof *P);
    memset(S, 0, n * sizeof *S);
    if (n <= (size_t)sb) return;
    for (size_t x = 0; x + (size_t)sb < n; x++) {
        size_t pr = 0, su = 0;      /* 0 = none (y > x >= 0 so 0 is free) */
        for (size_t y = x + 1; y < x + (size_t)sb; y++) {
            if (goes_before(in, n, la, x, y)) {
                if (!su || in_order(in, n, la, y, su)) su = y;
            } else {
                if (!pr || in_order(in, n, la, pr, y)) pr = y;
            }
        }
        P[x] = pr ? (uint16_t)(pr - x) : 0;
        S[x] = su ? (uint16_t)(su - x) : 0;
    }
}

/* SURVEY A.5 stage B.  prio[] is indexed by absolute position (n entries: oracle
 * favours clarity over the product's ring buffer). */
size_t lz77o_stage_b(const uint16_t *P, const uint16_t *S, size_t n, int sb, uint32_t *xval)
{
    size_t moved = 0;
    for (size_t i = 0; i < n; i++) xval[i] = LZ77O_NONE32;
    if (n <= (size_t)sb) return 0;
    uint32_t *prio = malloc(sizeof(uint32_t) * n);
    if (!prio) return BAD;
    for (size_t t = 0; t < n; t++) {
        if (t >= (size_t)sb) {
            size_t x = t - (size_t)sb;
            if (P[x] && S[x]) {
                uint32_t mine = prio[x];
                if (prio[x + P[x]] > mine && prio[x + S[x]] > mine) {
                    prio[x + S[x]] = mine;
                    xval[x] = mine;
                    moved++;
                }
            }
        }
        prio[t] = (uint32_t)t;
    }
    free(prio);
    return moved;
}

size_t lz77o_encode_model(const uint8_t *in, size_t n, int sb, int la, uint8_t *out, size_t cap)
{
    if (!args_ok(sb, la)) return BAD;
    uint16_t *P = malloc(sizeof(uint16_t) * (n + 1));
    uint16_t *S = malloc(sizeof(uint16_t) * (n + 1));
    uint32_t *prio = malloc(sizeof(uint32_t) * (n + 1));
    if (!P || !S || !prio) { free(P); free(S); free(prio); return BAD; }
    lz77o_stage_a(in, n, sb, la, P, S);
    bitsink s = { out, cap, 0, 0, 0, 0 };
    int ob = lz77o_bitof(sb), lb = lz77o_bitof(la);
    sink_header(&s, sb, la);
    size_t p = 0;
    for (size_t t = 0; t < n; t++) {
        if (t == p) {
            size_t left = n - t;
            int capl = (int)(left < (size_t)la ? left : (size_t)la) - 1;
            size_t c0 = t > (size_t)sb ? t - (size_t)sb : 0;
            int best = 0;
            size_t arg = 0;
            for (size_t c = c0; c < t; c++) {
                int l = lcp_cap(in + c, in + t, capl);
                if (l > best || (l == best && best > 0 && prio[c] < prio[arg])) { best = l; arg = c; }
            }
            sink_token(&s, ob, lb, best ? (int)(t - arg) : 0, best, in[t + (size_t)best]);
            p += (size_t)best + 1;
        }
        if (t >= (size_t)sb) {
            size_t x = t - (size_t)sb;
            if (P[x] && S[x] && prio[x + P[x]] > prio[x] && prio[x + S[x]] > prio[x])
                prio[x + S[x]] = prio[x];
        }
        prio[t] = (uint32_t)t;
    }
    free(P); free(S); free(prio);
    return sink_finish(&s);
}

/* ----------------------------------------------------------------- decode ---- */

size_t lz77o_tokens(const uint8_t *z, size_t zn, int *sb_out, int *la_out,
                    int32_t *off, int32_t *len, uint8_t *next, size_t cap)
{
    if (zn < 4) return BAD;
    int sb = z[0] | (z[1] << 8), la = z[2] | (z[3] << 8);           /* lz77.c:157-158 */
    if (sb_out) *sb_out = sb;
    if (la_out) *la_out = la;
    if (sb < 1 || la < 1) return BAD;
    int ob = lz77o_bitof(sb), lb = lz77o_bitof(la), T = ob + lb + 8;
    size_t ntok = (size_t)(((uint64_t)zn * 8 - 32) / (uint64_t)T);  /* lz77.c:271: short read = EOF */
    for (size_t k = 0; k < ntok && k < cap; k++) {
        uint64_t at = 32 + (uint64_t)k * (uint64_t)T;
        if (off) off[k] = (int32_t)peek_bits(z, zn, at, ob);
        if (len) len[k] = (int32_t)peek_bits(z, zn, at + (uint64_t)ob, lb);
        if (next) next[k] = (uint8_t)peek_bits(z, zn, at + (uint64_t)(ob + lb), 8);
    }
    return ntok;
}

/* lz77.c:148-197.  The reference's 3*SB+LA staging buffer is reproduced so that
 * even degenerate tokens (off==0 with len>0, emitted when -s is a power of two,
 * SURVEY A.7) decode to the same bytes the reference produces. */
size_t lz77o_decode(const uint8_t *z, size_t zn, uint8_t *out, size_t cap)
{
    int sb, la;
    size_t ntok = lz77o_tokens(z, zn, &sb, &la, NULL, NULL, NULL, 0);
    if (ntok == BAD) return BAD;
    int ob = lz77o_bitof(sb), lb = lz77o_bitof(la), T = ob + lb + 8;
    size_t W = (size_t)sb * 3 + (size_t)la;
    size_t slack = (size_t)1 << lb;                       /* malformed len may exceed la-1 */
    uint8_t *buf = calloc(W + slack + 1, 1);
    if (!buf) return BAD;
    size_t back = 0, j = 0;
    int overflow = 0;
    for (size_t k = 0; k < ntok; k++) {
        uint64_t at = 32 + (uint64_t)k * (uint64_t)T;
        size_t off = peek_bits(z, zn, at, ob);
        size_t len = peek_bits(z, zn, at + (uint64_t)ob, lb);
        uint8_t lit = (uint8_t)peek_bits(z, zn, at + (uint64_t)(ob + lb), 8);
        if (back + len > W - 1) {                         /* lz77.c:172-175 */
            if (back >= (size_t)sb) memmove(buf, buf + back - (size_t)sb, (size_t)sb);
            back = (size_t)sb;
        }
        for (size_t i = 0; i <= len; i++) {
            uint8_t b = i < len ? (off <= back ? buf[back - off] : 0) : lit;
            buf[back++] = b;
            if (out) { if (j < cap) out[j] = b; else overflow = 1; }
            j++;
        }
    }
    free(buf);
    return overflow ? BAD : j;
}

/* --------------------------------------------------------------- fixtures ---- */

void lz77o_splitmix_fill(uint64_t seed, uint8_t *dst, size_t n)
{
    uint64_t s = seed;
    size_t i = 0;
    while (i < n) {
        uint64_t v = (s += 0x9E3779B97F4A7C15ull);
        v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
        v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
        v ^= v >> 31;
        for (int k = 0; k < 8 && i < n; k++, i++) dst[i] = (uint8_t)(v >> (8 * k));
    }
}
