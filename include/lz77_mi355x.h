/*
 * lz77_mi355x.h -- C ABI of the MI355X-native LZ77 hot path (liblz77_mi355x.so).
 *
 * Drop-in boundary for cstdvd/lz77.  The reference's hot path is entered through
 * exactly two functions, called only from main.c:150,161:
 *
 *     void encode(FILE *file, struct bitFILE *out, int la, int sb);   lz77.h:14 / lz77.c:51
 *     void decode(struct bitFILE *file, FILE *out);                   lz77.h:15 / lz77.c:148
 *
 * Everything below them (tree.c's BST match finder, bitio.c's bit-at-a-time file
 * I/O) is replaced outright by HIP kernels; the wire format and the
 * token choice -- including the history-dependent choice among equal-length
 * matches (A.5) -- are reproduced bit for bit.
 *
 * Plain C, plain pointers and sizes.  No global state is required between calls;
 * lazily created per-process contexts cache device scratch buffers.  Thread safe: concurrent
 * callers each lease a context of their own (up to LZ77X_MAX_CONTEXTS, default 4; others wait).
 * All functions return 0 on success or a negative LZ77X_E_* code and never print.  The library is C++ inside;
 * no C++ exception ever crosses this boundary (a failed host allocation or a host thread that cannot start comes back
 * as LZ77X_E_NOMEM / LZ77X_E_HIP with lz77x_last_error() set).
 */
#ifndef LZ77_MI355X_H
#define LZ77_MI355X_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LZ77X_OK          0
#define LZ77X_E_ARG      (-1)   /* bad -s/-l or NULL pointer (main.c:35-38 limits; -s 0 is rejected) */
#define LZ77X_E_NOMEM    (-2)   /* host allocation failed */
#define LZ77X_E_HIP      (-3)   /* a HIP runtime call failed (see lz77x_last_error) */
#define LZ77X_E_NODEV    (-4)   /* no gfx950 device visible: there is NO CPU fallback */
#define LZ77X_E_FORMAT   (-5)   /* stream shorter than its 4-byte header / zero sb or la / a header whose token
                                   would be wider than 32 bits (la > 255 with a wide sb: main.c:103 never emits it) */
#define LZ77X_E_CAP      (-6)   /* caller-provided output buffer too small (*out_n holds the need) */
#define LZ77X_E_IO       (-7)   /* fread/fwrite failed */
#define LZ77X_E_TOOBIG   (-8)   /* only the stage-level entry points (lz77x_stage_*: the parity tests' view of one stage over a
                                   whole input), at >= 4 GiB: positions are 32-bit on a device.  lz77x_encode*, lz77x_decode*
                                   take any length on one device or several: segments / stretches of position shards /
                                   token ranges through bounded device memory */

#define LZ77X_DEFAULT_LA 15     /* lz77.c:21 */
#define LZ77X_DEFAULT_SB 4095   /* lz77.c:22 */

/* ---- buffer level (host memory in, host memory out) ---------------------------------- */

/* Replaces encode() (lz77.c:51-140) on an in-memory file.  sb in [1,65535], la in
 * [2,255]; -1 selects the default like lz77.c:65-66.  *out is malloc'ed by the
 * library (release with lz77x_free); *out_n = 4 + ceil(ntok*T/8) bytes. */
int lz77x_encode(const uint8_t *in, size_t n, int sb, int la, uint8_t **out, size_t *out_n);

/* Replaces decode() (lz77.c:148-197).  The header inside the stream supplies sb/la
 * (lz77.c:157-158); a trailing partial token is dropped (lz77.c:271-280).  Like the reference (lz77.c:160-195: any
 * length through a 3*SB+LA buffer) the decoder takes a stream of any length: it runs range by range -- at most
 * LZ77X_DECODE_RANGE tokens (default 2^26) and LZ77X_DECODE_RANGE_BYTES output bytes (default 2^30) at a time, device
 * memory independent of the stream's length; a stream that fits one range is decoded in one piece. */
int lz77x_decode(const uint8_t *z, size_t zn, uint8_t **out, size_t *out_n);

void lz77x_free(void *p);

/* Upper bound of lz77x_encode's output for n input bytes (every token a bare literal). */
size_t lz77x_encode_bound(size_t n, int sb, int la);

/* ---- device level (buffers already resident in HBM; what bench.py times) -------------- */

/* d_in/d_out are device pointers on the current HIP device.  stream is a hipStream_t
 * (NULL = default stream) on which all kernels of the call are enqueued; the call
 * returns after the result is complete.  out_cap >= lz77x_encode_bound(). */
int lz77x_encode_device(const void *d_in, size_t n, int sb, int la,
                        void *d_out, size_t out_cap, size_t *out_n, void *stream);

/* Decode needs the decoded size before the caller can size d_out: call with
 * d_out == NULL to get *out_n, then again with a buffer of at least that many bytes. */
int lz77x_decode_device(const void *d_z, size_t zn,
                        void *d_out, size_t out_cap, size_t *out_n, void *stream);

/* ---- FILE* level: what main() of the reference calls -------------------------------- */

/* Same argument meaning and ORDER as encode(file,out,la,sb) (lz77.h:14): la then sb,
 * -1 = default (main.c:67).  Reads `in` to EOF, writes the complete stream to `out`. */
int lz77x_encode_file(FILE *in, FILE *out, int la, int sb);
int lz77x_decode_file(FILE *in, FILE *out);

/* ---- configuration / introspection --------------------------------------------------- */

/* Number of logical shards the positions of one input are split into (default 1, or
 * env LZ77X_SHARDS).  Shards are spread round-robin over the visible devices; output
 * bytes are identical for every shard count.  lz77x_encode / lz77x_encode_file cut the input by
 * positions -- an input of any length in stretches of at most 4 GiB, every stretch over all the devices, the file
 * entry point holding one stretch in host memory --, lz77x_decode cuts the stream by token ranges (the sb bytes before a range reach it
 * as a map chained on the host; lz77.c:172-192 across the cuts), a long stream in stretches of tokens, every stretch
 * over all the devices (a tail too short for a window of output per device joins the stretch before it); streams it cannot
 * cut that way (distance-0 copies: a power-of-two -s, whose staging-buffer image at a cut is a function of everything
 * before it, lz77.c:172-188) decode on one device, range by range.  The shards of one call are driven by one host thread each. */
int lz77x_set_shards(int shards);
int lz77x_device_count(void);
/* Release every cached device/pinned buffer, stream and event (they are otherwise kept for the
 * life of the process and re-created on the next call). */
void lz77x_shutdown(void);
const char *lz77x_strerror(int code);
const char *lz77x_last_error(void);       /* detail of the last LZ77X_E_HIP, thread local */
const char *lz77x_version(void);

/* ---- one stream on several devices: the host-side plan and exchange (no device needed) ------------
 * With lz77x_set_shards(D) > 1 lz77x_encode cuts the positions of one stream into D contiguous shards, one per
 * device; each device only ever holds its shard.  The two sequential loops of lz77.c cross the
 * cuts as small maps chained on the host; these three functions ARE that host side (the library calls them
 * itself; exported so that the multi-rank plan can be checked without a GPU). */
typedef struct lz77x_shard {
    uint64_t first_token_pos, end_token_pos;   /* the shard emits the tokens whose position lies in [first, end) */
    uint64_t local0;                            /* global position of the shard's first byte (= first - lookback) */
    uint64_t local_bytes;                       /* bytes the device holds: look-back + shard + look-ahead */
    uint64_t steps;                             /* evictions (lz77.c:101-103) it simulates: [local0, end - sb) */
    uint32_t lookback, reserved;                /* sb, or 0 for the first shard */
} lz77x_shard;
int lz77x_shard_plan(size_t n, int sb, int la, int shards, lz77x_shard *out);       /* -> shards used, or < 0 */
void lz77x_shard_compose_cells(const uint16_t *dest, const uint32_t *loc, int sb, uint32_t *cells);
void lz77x_shard_compose_chain(const uint8_t *exit_of, const uint32_t *tokens_of, uint32_t *entry, uint64_t *tokens);
/* decode side (lz77.c:172-192 across the cuts): first token of shard d of `shards` over a stream of ntok tokens (a
 * multiple of eight: every shard starts on a byte of the stream), and one shard's map applied to the sb bytes before
 * it: map[i] = a byte value, or 0x8000 | index into `incoming`; outgoing = the last sb bytes of the shard's output */
uint64_t lz77x_shard_token_cut(uint64_t ntok, int shards, int d);
void lz77x_shard_compose_tail(const uint16_t *map, int sb, const uint8_t *incoming, uint8_t *outgoing);
/* the same for windows above 8192 (the tile pass, 32-bit states): map[i] = a byte value, or 0x10000 | index into `incoming` */
void lz77x_shard_compose_tail32(const uint32_t *map, int sb, const uint8_t *incoming, uint8_t *outgoing);

/* ---- several files at once (each on a context of its own, up to LZ77X_MAX_CONTEXTS at a time) -------
 * rc[i] receives the status of file i; returns 0 when all succeeded, else the first failure. */
int lz77x_encode_files(int n_files, FILE **in, FILE **out, int la, int sb, int *rc);
int lz77x_decode_files(int n_files, FILE **in, FILE **out, int *rc);

/* Per-call timing of the last encode/decode on this thread, milliseconds.  Kernel
 * times are hipEvent pairs on the call's stream; host_* are wall clock. */
typedef struct lz77x_stats {
    double total_ms;          /* whole call */
    double k_match_ms;        /* match stage (replaces tree.c): region sort + window walkers + finalize */
    double k_sort_ms;         /* of which: the region-sort kernel alone (0 if that path was not taken) */
    double k_token_ms;        /* transfer index + tie-break + token pack kernels */
    double k_decode_ms;       /* parse + scan + copy-resolution kernels */
    double host_chain_ms;     /* greedy parse chain walk (host) */
    double host_stageb_ms;    /* sequential priority recurrence (host) */
    double copy_ms;           /* host wall time blocked waiting for the device (not overlapped) */
    uint64_t n;               /* uncompressed bytes */
    uint64_t zn;              /* compressed bytes */
    uint64_t ntok;            /* tokens */
    uint64_t transfers;       /* stage-B priority hand-overs */
    uint32_t match_launches;  /* launches of the region kernel in the call */
    uint32_t decode_rounds;   /* pointer-jumping rounds */
    double k_walk_ms;         /* of k_match_ms: the window-walker kernel alone (0 if that path was not taken) */
    double k_tiebreak_ms;     /* of k_token_ms: the tie-break kernel alone (k_tokens_sorted; large windows: k_tokens_rank_group) */
    uint32_t token_launches;  /* launches of the tie-break kernel in the call */
    uint32_t prio_iters;      /* gate iterations of the device priority recurrence (0: it ran on the host) */
    double k_prio_ms;         /* device priority recurrence (replaces host_stageb_ms when it runs) */
    double k_chain_ms;        /* device parse chain (replaces host_chain_ms when it runs) */
    double k_prio_fwd_ms;     /* of k_prio_ms: the forward sweeps (k_prio_fwd), summed over the iterations */
    double k_prio_back_ms;    /* of k_prio_ms: the backward sweeps (k_prio_back) */
    double k_prio_scan_ms;    /* of k_prio_ms: the block-boundary scans (k_prio_scan_*) */
    double k_sort_chunks_ms;  /* of k_sort_ms: the chunk-sort kernel alone (k_c1_chunks; 0 where the region kernel sorts everything) */
} lz77x_stats;
int lz77x_last_stats(lz77x_stats *st);

/* ---- stage-level entry points (kernel parity tests; not needed by a drop-in user) ----- */

/* maxlen[p] for every position, host buffers */
int lz77x_stage_maxlen(const uint8_t *in, size_t n, int sb, int la, uint8_t *maxlen);
/* in-order predecessor/successor distances of every evicted position (SURVEY A.5 stage A) */
int lz77x_stage_neighbours(const uint8_t *in, size_t n, int sb, int la, uint16_t *P, uint16_t *S);
/* host sequential stage (A.5 stage B) on caller-provided P/S: xval[x] or 0xFFFFFFFF */
int lz77x_stage_priorities(const uint16_t *P, const uint16_t *S, size_t n, int sb, uint32_t *xval);
/* the same recurrence on the device (k_prio: gate iteration over block sweeps), sb <= 4096; *iters
 * receives the number of iterations it took (may be NULL) */
int lz77x_stage_priorities_device(const uint16_t *P, const uint16_t *S, size_t n, int sb, uint32_t *xval, int *iters);
/* the greedy parse chain (lz77.c:89-98) on the device from a caller-provided maxlen[]: chain[] must hold
 * n entries; *ntok receives the number of tokens */
int lz77x_stage_chain_device(const uint8_t *maxlen, size_t n, int la, uint32_t *chain, size_t *ntok);

#ifdef __cplusplus
}
#endif
#endif
