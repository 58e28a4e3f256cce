/*
 * lz77x_internal.h -- shared between the HIP translation units (k_*.hip), the
 * host side (host.h: ctx, hostio, encode_pipe, encode_host, decode_pipe, shard, api) and the sequential host stage (hoststage.c).
 * Not part of the public ABI (that is include/lz77_mi355x.h).
 */
#ifndef LZ77X_INTERNAL_H
#define LZ77X_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#define LZ77X_PAD      1024u          /* 0xFF bytes kept after the input on device (>= LA + 16) */
#define LZ77X_NONE32   0xFFFFFFFFu
#define LZ77X_MAX_N    0xFFF00000u    /* positions are 32-bit on device */

#ifdef __cplusplus
extern "C" {
#endif

/* Geometry of one call: token widths and the region decomposition of the match stage.
 *
 * A region is the unit one workgroup sorts: the positions [t0, t0+TILE+SB) of the input, TILE =
 * RP - SBu, RP a power of two >= 4*SBu (so TILE+SB <= RP slots are sorted per TILE positions: 4/3
 * redundancy).  One window W_t = [t, t+SB) slides over it for t in [t0, t0+TILE) and answers BOTH
 * neighbour queries of the match stage at every step:
 *     forward  (SURVEY A.5 stage A): in-order neighbours of x = t among W_t \ {t}    -> ps[x]
 *     backward (A.3, longest match): in-order neighbours of y = t+SB inside W_t      -> maxlen[y]
 * so region r writes ps[] for [t0, t0+TILE) and maxlen[] for [t0+SB, t0+TILE+SB) -- the backward
 * results are shifted by SB against the forward ones, which is what lets one sorted set and one
 * window walk serve both (region 0 also answers y < SB while its first window fills).  A launch
 * of regions [r0, r1) therefore needs maxlen[r0*TILE .. r0*TILE+SB) from the launch before it;
 * whoever starts at r0 > 0 without one (a shard) launches region r0-1 as well.
 *
 * `shifted == 0` is the layout of the exhaustive pair-scan cross-checks (LZ77X_MATCH_VARIANT 1/3):
 * TILE = RP - SBu - SB positions [t0, t0+TILE) with halos of SBu to the left and SB-1 to the right,
 * both results on the same TILE.
 */
typedef struct lz77x_geom {
    int sb, la;            /* search buffer, lookahead */
    int ob, lb, T;         /* bitof(sb), bitof(la), token bits (lz77.c:249-251) */
    uint32_t SBu;          /* sb rounded up to a multiple of 8 */
    uint32_t RP;           /* padded region size (power of two) */
    uint32_t TILE;         /* positions produced per region (multiple of 8) */
    int fast;              /* 1: ranks are 16-bit and live in LDS (RP <= 16384) */
    int shifted;           /* 1: production layout (above); 0: pair-scan layout */
} lz77x_geom;

int  lz77x_bitof(int n);                                   /* bitio.c:41-43, integer form */
void lz77x_make_geom(lz77x_geom *g, int sb, int la);
void lz77x_geom_legacy(lz77x_geom *g);                     /* switch g to the pair-scan layout */

/* ---- sequential host stage (hoststage.c) -------------------------------------------- */

/* Greedy parse chain (lz77.c:98: p += len+1) from maxlen[]; appends chain positions to
 * chain[] starting at *ntok; resumes at *p for positions < limit.  Returns new *p. */
size_t lz77x_host_chain(const uint8_t *maxlen, size_t limit, size_t p, uint32_t *chain, size_t *ntok);

typedef struct lz77x_prio_state {
    uint32_t *ring;        /* priority of live position q at ring[q & mask] */
    uint32_t mask;
    size_t next;           /* next position to insert (== positions processed) */
    uint64_t transfers;    /* unused by the hot loop (the device counts hand-overs while indexing them) */
} lz77x_prio_state;

int  lz77x_prio_init(lz77x_prio_state *st, int sb);
void lz77x_prio_free(lz77x_prio_state *st);
/* Advance the recurrence (SURVEY A.5 stage B) over insert times [st->next, upto);
 * ps[x] = ((x+P) & mask) | ((x+S) & mask) << 16 (ring cells of x's neighbours, P/S = 0 when
 * missing) must be available for x < upto - sb; writes xval[x] for those x. */
uint32_t lz77x_prio_mask(int sb);
void lz77x_prio_run(lz77x_prio_state *st, const uint32_t *ps, int sb, size_t upto, uint32_t *xval);
int lz77x_prio_run_cells(const uint32_t *ps, size_t nx, int sb, const uint32_t *cells_in, uint32_t voff, uint32_t *xval, uint32_t *cells_out);

#include <stdio.h>

#ifdef __cplusplus
}
#endif

#ifdef __cplusplus
#include <hip/hip_runtime_api.h>

/* Cross-check variants, timing probes and forced fallbacks (the kernels of earlier rounds, the pair-scan matcher, the
 * bitonic sorts, the host recurrence on demand, ...) exist in the build the tests load -- -DLZ77X_VARIANTS,
 * liblz77_mi355x_variants.so -- and nowhere else: the product library has one path per stage and does not read their
 * environment knobs. */
#ifdef LZ77X_VARIANTS
#define LZ77X_VENV(name) getenv(name)
#else
#define LZ77X_VENV(name) (static_cast<const char *>(nullptr))
#endif

/* ---- kernel launchers (k_match.hip, k_tokens.hip, k_decode.hip, k_util.hip).  All enqueue on `s` and return immediately. ----- */

size_t lz77k_match_scratch_bytes(const lz77x_geom &g, uint32_t nregions);
size_t lz77k_match_lds_bytes(const lz77x_geom &g);
/* regions [region0, region0+nregions) of an n-byte padded input; variant 0 = production
 * (fast geometry: sort + bitmap window walkers; else exhaustive pair scan), 1 = all-masked pair scan
 * (self-check), 2 = sort only (timing probe), 3 = exhaustive packed pair scan */
hipError_t lz77k_match(const uint8_t *d_in, uint32_t n, const lz77x_geom &g,
                       uint32_t region0, uint32_t nregions,
                       uint32_t *d_ps, uint8_t *d_maxlen, void *d_scratch, int variant, hipStream_t s,
                       hipEvent_t *ev_sort = nullptr /* [3]: before the region sort, between sort and walkers, after the walkers */,
                       uint32_t *d_ranks_all = nullptr /* large windows: (2RP+8) words per region of the WHOLE input, kept for lz77k_tokens */);

/* large windows: 1 when the regions' orders come from the sort shared between overlapping regions (a region's order
 * then holds every position < n of its RP slots, not only the TILE + sb it owns) */
int lz77k_big_sort_shared(const lz77x_geom &g);

hipError_t lz77k_fill_pad(uint8_t *d_in, uint32_t n, hipStream_t s);

/* cells[x] = ((x+P)&mask) | ((x+S)&mask)<<16 for x in [x0, x1): what the host recurrence indexes its ring with */
hipError_t lz77k_ps_cells(const uint32_t *d_ps, uint32_t *d_cells, uint32_t x0, uint32_t x1, uint32_t mask, hipStream_t s);

/* exclusive scan of m uint32 (in place allowed); d_tmp needs lz77k_scan_tmp_bytes(m) */
size_t lz77k_scan_tmp_bytes(uint32_t m);
hipError_t lz77k_scan_u32(const uint32_t *d_in, uint32_t *d_out, uint32_t m, void *d_tmp, hipStream_t s);
/* nwords 32-bit words from device memory into pinned host memory (hipHostMalloc, not hipHostRegister) by a kernel: a tenth of what
 * the runtime's blit of a small device-to-host copy costs the stream */
hipError_t lz77k_publish(void *h_dst_pinned, const void *d_src, uint32_t nwords, hipStream_t s);

/* ---- device-resident sequential stages (k_prio.hip, k_chain.hip) ---- */

/* The priority recurrence (SURVEY A.5 stage B, tree.c:202-231) on the device: xval[x] for x < nx from
 * ps[] (distances P | S << 16).  sb <= 4096: a wavefront per block, the live cells a ring of priorities in LDS
 * (k_prio.hip); above: a workgroup per block, 18-bit codes in the ring, boundary scan through HBM (k_priow.hip).
 * h_flag: 8 bytes of pinned host memory.  Synchronises the stream once per iteration; *converged = 0
 * when max_iters did not suffice (the caller then runs the host recurrence instead). */
size_t lz77k_prio_tmp_bytes(uint32_t nx, int sb);
int lz77k_prio_supported(int sb);
hipError_t lz77k_prio(const uint32_t *d_ps, uint32_t nx, int sb, uint32_t *d_xval, void *d_tmp, hipStream_t s,
                      uint32_t *h_flag, int max_iters, int *iters, int *converged,
                      hipEvent_t *ev4 = nullptr /* four events: per-kernel times are added to ms3 */,
                      float *ms3 = nullptr /* += forward sweeps, backward sweeps, boundary scans */,
                      uint32_t voff = 0 /* a cell's own priority = its position + voff */,
                      const uint32_t *d_carried = nullptr /* sb values of cells 0..sb-1 before step 0 (else: their own) */,
                      uint32_t *d_out_state = nullptr /* sb values of cells nx..nx+sb-1 after the last step */);

/* The same iteration as phases, for several devices that each hold a shard of ONE stream: every iteration
 * the shards' whole-shard maps go to the host, which chains them into the cells each shard starts from.
 *     begin -> { maps(whole) -> [host: compose, set_in0] -> sweep -> [sync] -> advance } until no shard flips */
struct lz77k_prio_plan {
    const uint32_t *ps = nullptr;
    uint32_t *xval = nullptr;
    void *tmp = nullptr;
    uint32_t nx = 0, sb = 0, voff = 0, ncarried = 0;
    uint32_t B = 0, NB = 0, ngroups = 0, sb_r = 0, ring_n = 0, G = 0, NG = 0, NG2 = 0;
    uint32_t rs = 0;          /* cells between the rows of the per-block arrays (dest, loc, in, ...): sb, or sb rounded up to 8 for windows above 4096 */
    uint32_t W = 64;          /* steps a sweep takes together: 64 = one wavefront per block (k_prio.hip), 256 / 1024 = a workgroup (k_priow.hip) */
    bool pack18 = false;      /* W > 64: the ring holds 18-bit codes (rank of an old value | block-local position), not priorities */
    size_t o_gate[2] = {0, 0}, o_rmask = 0, o_cmask = 0, o_dest = 0, o_loc = 0, o_in = 0, o_gdest = 0, o_gloc = 0, o_gin = 0, o_sum = 0, o_dirty = 0, total = 0;
    size_t o_g2dest = 0, o_g2loc = 0, o_g2in = 0;                  /* sb <= 4096: the groups of groups of the boundary scan */
    size_t o_destx = 0;                                            /* W > 64: the backward sweep's scratch rows (every step's exit cell, a row per resident workgroup) */
    size_t o_codes = 0, o_gval = 0, o_scan = 0, o_inprev = 0;      /* pack18: per-block rows entry cell -> code, rank -> value; sb > 4096: the HBM scan's running pairs */
    int cur = 0;              /* gate buffer the next maps/sweep read */
    uint32_t sweeps = 0;      /* sweeps so far (the first one visits every block) */
    bool in0_dirty = false;   /* the cells block 0 starts from were replaced (lz77k_prio_set_in0) since its last sweep */
    uint32_t first = 0;       /* blocks before it are final */
};
/* k_priow.hip: the sweeps with a workgroup of W threads per block, the boundary scan through HBM (windows whose
 * vectors of sb priorities do not fit LDS) */
uint32_t lz77kw_width(int sb);
void lz77kw_debug_dump(void);
bool lz77kw_pack18(uint32_t ring_n);
hipError_t lz77kw_prep(const uint32_t *d_ps, uint32_t nx, uint32_t sb_r, uint32_t W, uint64_t *d_rmask, uint64_t *d_gate0, uint64_t *d_cmask, hipStream_t s);
hipError_t lz77kw_fwd(const uint32_t *d_ps, uint32_t nx, uint32_t sb, uint32_t rs, uint32_t B, uint32_t ring_n, uint32_t W, uint32_t b_first, uint32_t nb,
                      const uint64_t *d_rmask, const uint64_t *d_cmask, const uint64_t *d_gold, uint64_t *d_gnew, const uint32_t *d_in, uint32_t *d_xval,
                      uint32_t *d_summary, uint32_t voff, uint32_t *d_out_state, uint32_t ncarried, uint32_t *d_codes, uint32_t *d_gval,
                      uint32_t *d_in_prev, uint32_t have_prev, uint32_t *d_gates_changed, hipStream_t s);
hipError_t lz77kw_back(const uint32_t *d_ps, uint32_t nx, uint32_t sb, uint32_t rs, uint32_t B, uint32_t ring_n, uint32_t W, uint32_t b_first, uint32_t nb,
                       const uint64_t *d_gates, uint16_t *d_dest, uint32_t *d_loc, uint32_t voff, uint32_t ncarried, const uint32_t *d_gates_changed,
                       uint16_t *d_destx, hipStream_t s);
size_t lz77kw_back_scratch_bytes(uint32_t NB, uint32_t B, uint32_t ring_n, uint32_t W);
size_t lz77kw_scan_tmp_bytes(uint32_t NG, uint32_t rs);
hipError_t lz77kw_compose_all(const uint16_t *d_dest, const uint32_t *d_loc, uint32_t sb, uint32_t rs, uint32_t nmaps, uint32_t G,
                              uint16_t *d_gdest, uint32_t *d_gloc, void *d_tmp, hipStream_t s);
hipError_t lz77kw_scan(const uint16_t *d_dest, const uint32_t *d_loc, uint32_t *d_in, uint32_t sb, uint32_t rs, uint32_t first, uint32_t nmaps, uint32_t G,
                       uint16_t *d_gdest, uint32_t *d_gloc, uint32_t *d_gin, void *d_tmp, hipStream_t s);

/* Will the gate iteration end within its budget?  f[0..5] / b[0..5]: the flips and the first block with a flip of the last
 * six iterations (oldest first), open: blocks from there to the end, it: iterations done.  Two ways it does not (DESIGN
 * 2.2d, tools/worst_cases.py): an error FRONT -- the flips stop decaying (a few dozen, in a band behind the first open
 * block) and the band moves a few blocks an iteration: open / rate iterations to go --, and a slow global decay (a
 * repeated block with noise: flips everywhere, x 0.85-0.9 an iteration): ln(flips) / -ln(rho) to go.  Classes that converge
 * decay by 0.2-0.8 an iteration while their flips are many (text is through before six samples exist; record-structured data
 * at C2 takes 25-31 iterations, most of them below a thousand flips). */
static inline bool lz77x_prio_hopeless(const uint64_t f[6], const uint64_t b[6], uint64_t open, int it, int max_iters)
{
    if (it < 6 || max_iters >= (1 << 29)) return false;
    const double early = (double)(f[0] + f[1] + f[2]), late = (double)(f[3] + f[4] + f[5]);
    if (early <= 0 || late <= 0) return false;
    const double rho = __builtin_cbrt(late / early);
    double to_go;
    if (rho >= 0.97) {
        /* (a plateau of a handful of flips is also how a slow class ENDS -- record-structured data at C2: 6, 8, 9, 10, 6, 8
         * flips, then three more iterations -- so a front only counts when it is three budgets away, not one) */
        const double rate = b[5] > b[0] ? (double)(b[5] - b[0]) / 5.0 : 0.0;
        to_go = (double)open / (rate > 0.2 ? rate : 0.2) / 3.0;
    } else {
        /* (below a thousand flips the decay is no longer geometric -- the last few hundred collapse within a handful of
         * iterations: record-structured data at C2 went 159, 141, 98, ... to 0 in nine -- so only a decay that is slow while
         * the flips are still many counts as hopeless) */
        if (f[5] < 1024) return false;
        to_go = __builtin_log((double)f[5]) / -__builtin_log(rho);
    }
    return (double)it + to_go > (double)max_iters;
}

hipError_t lz77k_prio_begin(lz77k_prio_plan &P, const uint32_t *d_ps, uint32_t nx, int sb, uint32_t *d_xval, void *d_tmp, uint32_t voff,
                            const uint32_t *d_carried, hipStream_t s);
hipError_t lz77k_prio_set_in0(lz77k_prio_plan &P, const uint32_t *h_or_d_in0, hipMemcpyKind kind, hipStream_t s);
hipError_t lz77k_prio_maps(lz77k_prio_plan &P, hipStream_t s, bool whole, const uint16_t **d_sdest, const uint32_t **d_sloc);
hipError_t lz77k_prio_sweep(lz77k_prio_plan &P, hipStream_t s, uint32_t *h_flag, uint32_t *d_out_state, hipEvent_t *ev3 = nullptr);
void lz77k_prio_advance(lz77k_prio_plan &P, const uint32_t *h_flag, bool restart_at0);

/* The greedy parse chain (lz77.c:89-98) on the device: chain[k] = position of token k.  *d_tbase points
 * (inside d_tmp) at the index of the first token of every lz77k_chain_sub()-position sub-block, nsub + 1
 * words, the last one = ntok.  Enqueues only. */
size_t lz77k_chain_tmp_bytes(uint32_t n, int la);
uint32_t lz77k_chain_sub(void);
/* the two phases of lz77k_chain: (1) the maps of the sub-blocks of [start, n), composed by groups and -- whole --
 * over the whole range (d_wexit[e], d_wcnt[e]: exit offset and token count when the range is entered at start + e);
 * (2) from the true entry offset, chain[] and the sub-blocks' first-token indices */
hipError_t lz77k_chain_maps(const uint8_t *d_maxlen, uint32_t n, int la, void *d_tmp, hipStream_t s, uint32_t start, bool whole,
                            const uint8_t **d_wexit, const uint32_t **d_wcnt);
hipError_t lz77k_chain_finish(const uint8_t *d_maxlen, uint32_t n, int la, uint32_t *d_chain, void *d_tmp, hipStream_t s,
                              uint32_t start, uint32_t entry0, const uint32_t **d_tbase, uint32_t *nsub, const uint32_t **d_exit);
hipError_t lz77k_chain(const uint8_t *d_maxlen, uint32_t n, int la, uint32_t *d_chain, void *d_tmp, hipStream_t s,
                       const uint32_t **d_tbase, uint32_t *nsub,
                       uint32_t start = 0 /* the chain begins here; sub-blocks are counted from it */,
                       const uint32_t **d_exit = nullptr /* -> how far past n the last token reaches (device word) */);

/* *d_out = 64-bit sum of m uint32 */
hipError_t lz77k_sum_u32(const uint32_t *d_in, uint32_t m, unsigned long long *d_out, hipStream_t s);

/* hand-over index of evictions x in [xa, xb) into destinations [dbase, dend):
 * list(c) = ent[ (c > dbase ? ofs[c-dbase-1] : 0) .. ofs[c-dbase] ).  d_ofs: dend-dbase+1 words */
hipError_t lz77k_xfer_index(const uint32_t *d_ps, const uint32_t *d_xval, uint32_t xa, uint32_t xb,
                            uint32_t dbase, uint32_t dend, uint32_t *d_ofs, uint2 *d_ent,
                            void *d_scan_tmp, hipStream_t s,
                            uint32_t x_new = 0, unsigned long long *d_total = nullptr /* += hand-overs with x >= x_new */,
                            uint32_t sb = 0 /* a hand-over reaches < sb positions ahead: lets destination blocks build their lists in LDS */);

/* tokens d_chain[0..ntok) lie in [pos0, pos1).  variant 0: tiled kernel (window, hand-over lists and a
 * two-byte candidate index in LDS) when sb <= 8192, rank-order enumeration (d_ranks_all) above; variant 2:
 * tiled without the index (every candidate visited); variant 3: large windows through the global two-byte
 * index; else / variant 1: one wave per token straight from global memory.
 * d_tstart: lz77k_tokens_tmp_bytes(pos1-pos0). */
size_t lz77k_tokens_tmp_bytes(uint32_t n, const lz77x_geom &g);
/* large windows (sb > 8192): bytes of the global two-byte candidate index for npos token positions */
size_t lz77k_tokens_index_bytes(const lz77x_geom &g, size_t npos);
hipError_t lz77k_tokens(const uint8_t *d_in, uint32_t n, const lz77x_geom &g,
                        const uint32_t *d_chain, uint32_t ntok, const uint8_t *d_maxlen,
                        const uint32_t *d_ofs, const uint2 *d_ent, uint32_t dbase,
                        uint32_t pos0, uint32_t pos1, uint32_t *d_tokval,
                        uint32_t *d_tstart, void *d_index, int variant, hipStream_t s,
                        hipEvent_t *ev_tie = nullptr /* [2]: recorded around the tie-break kernel */,
                        const uint32_t *d_ranks_all = nullptr /* large windows: the regions' rank + inverse arrays */,
                        const uint32_t *d_look = nullptr, uint32_t nlook = 0 /* positions < nlook start with priority d_look[c] (a later
                                                                                  segment's look-back) */,
                        uint32_t voff = 0 /* a position's own priority = position + voff */,
                        const uint32_t *d_ps = nullptr, const uint32_t *d_xval = nullptr /* lz77k_tokens_builds_lists: the tie-break
                                                                                            builds its tiles' hand-over lists itself */,
                        unsigned long long *d_total = nullptr /* [0] += hand-overs of the evictions [pos0 - sb, pos1 - sb); [1..32]: zeroed scratch */);
/* the production tie-break of LDS-sized windows reads ps/xval itself (no lz77k_xfer_index, d_ofs/d_ent unused) */
bool lz77k_tokens_builds_lists(const lz77x_geom &g, int variant, const void *d_ranks_all);

/* words [w0, w0+nw) of the output stream (word 0 = header) from tokval[0..ntok) */
hipError_t lz77k_pack(const uint32_t *d_tokval, uint64_t ntok, const lz77x_geom &g,
                      uint32_t *d_out_words, uint64_t nwords, hipStream_t s);
/* words [w0, w0+nw) from the tokens [k_first, k_end), d_tokval[0] = token k_first, d_out_words[0] = word w0 */
hipError_t lz77k_pack_range(const uint32_t *d_tokval, uint64_t k_first, uint64_t k_end, const lz77x_geom &g, uint32_t *d_out_words,
                            uint64_t w0, uint64_t nw, hipStream_t s);

/* d_stale_flag (may be null; two words): [0] is set when a token copies from distance 0 (power-of-two -s),
 * [1] when one copies from beyond the window (off > sb: no stream of the reference's encoder) */
hipError_t lz77k_dec_parse(const uint8_t *d_z, uint32_t ntok, const lz77x_geom &g,
                           uint32_t *d_tokval, uint32_t *d_len1, hipStream_t s, uint32_t *d_stale_flag = nullptr);
/* Distance-0 copies (power-of-two -s): cyc[0..ncyc] = offsets at which the reference's staging buffer starts a new pass
 * (+ a final end), in the coordinates of the working buffer = `pre` bytes of history + the output; cyc == null when the
 * stream has none.  A stream decoded range by range: pass 0 of the list may have begun before the range (first0: it is the
 * first pass of the whole stream), img = offset in the working buffer of the image of the reference's buffer at index sb
 * (LZ77X_NONE32: there is none, a never-written index reads as zero).  See dec_stale_src in k_decode.hip. */
struct lz77k_dec_stale {
    const uint32_t *cyc = nullptr;
    uint32_t ncyc = 0, first0 = 1, img = LZ77X_NONE32;
};
/* n = pre + output bytes: the working buffer d_out / d_ptr holds `pre` bytes of resolved history in front of the output */
hipError_t lz77k_dec_expand(const uint32_t *d_tokval, const uint32_t *d_dst, uint32_t ntok,
                            const lz77x_geom &g, uint8_t *d_out, uint32_t *d_ptr, uint32_t n, hipStream_t s,
                            const lz77k_dec_stale &Q = lz77k_dec_stale(), uint32_t pre = 0);
/* what a range of a stream leaves to the next one (k_decode.hip) */
hipError_t lz77k_dec_carry(const uint8_t *d_carry_old, const uint8_t *d_out, uint32_t n, uint32_t cb, uint8_t *d_carry_new, hipStream_t s);
hipError_t lz77k_dec_image(const uint8_t *d_x, const lz77k_dec_stale &Q, uint32_t sb, uint32_t W, uint32_t pre, const uint8_t *d_img_old,
                           uint8_t *d_img_new, hipStream_t s);
hipError_t lz77k_dec_cut(const uint32_t *d_dst, uint32_t ntok, uint32_t cap, uint32_t *d_res, hipStream_t s);
/* a shard's last cb bytes as a map on the cb bytes before it (after the tile pass and the jumping on [pre | output]) */
hipError_t lz77k_dec_tail_map(const uint8_t *d_x, const uint32_t *d_ptr, const unsigned long long *d_unres, uint32_t pre, uint32_t n, uint32_t cb,
                              uint32_t *d_map, hipStream_t s);
#define LZ77K_DEC_TILE_BYTES 12288u      /* the tile pass works on tiles of this many output bytes: `pre` is a multiple of it */
/* one pass over in_list[0..total) (or over every j < total when in_list is null); entries that moved are
 * appended to out_list, *out_count += their number */
hipError_t lz77k_dec_jump(uint32_t *d_ptr, uint32_t total, const uint32_t *d_in_list, uint32_t *d_out_list, uint32_t *d_out_count,
                          hipStream_t s);
hipError_t lz77k_dec_gather(uint8_t *d_out, const uint32_t *d_ptr, uint32_t n, hipStream_t s);
/* tile-local decode (production): per tile of the output everything that stays inside the tile is resolved in
 * LDS; only pointers that leave a tile reach HBM (d_ptr[] + one bit per byte) and take part in the jumping */
size_t lz77k_dec_tile_tmp_bytes(uint32_t n);
hipError_t lz77k_dec_tiles(const uint32_t *d_tokval, const uint32_t *d_dst, uint32_t ntok, const lz77x_geom &g, uint8_t *d_out,
                           uint32_t *d_ptr, uint32_t n, void *d_tmp, const unsigned long long **d_unres, hipStream_t s,
                           const lz77k_dec_stale &Q = lz77k_dec_stale(), uint32_t pre = 0);
hipError_t lz77k_dec_jump2(uint32_t *d_ptr, const unsigned long long *d_unres, uint32_t total, const uint32_t *d_in_list,
                           uint32_t *d_out_list, uint32_t *d_out_count, hipStream_t s);
/* segment decode (production for sb <= 8192, no distance-0 copies): a workgroup walks a segment of the output
 * front to back with the roots of the last sb bytes in an LDS ring; no per-byte pointers in HBM */
int lz77k_dec_seg_supported(const lz77x_geom &g);
size_t lz77k_dec_seg_tmp_bytes(uint32_t n, const lz77x_geom &g);
hipError_t lz77k_dec_segments(const uint32_t *d_tokval, const uint32_t *d_dst, uint32_t ntok, const lz77x_geom &g, uint8_t *d_out,
                              void *d_ref, uint32_t n, void *d_tmp, hipStream_t s);
/* the two phases of lz77k_dec_segments (k_decode.hip), for a stream whose token ranges are decoded on several devices */
struct lz77k_dec_seg_state {
    uint32_t sbytes = 0, nseg = 0, ntails = 0, G = 0, NG = 0;
    bool ext0 = false;
    uint32_t *tfirst = nullptr;
    unsigned long long *flags = nullptr;
    uint16_t *tail = nullptr, *gmap = nullptr, *smap = nullptr;
    uint8_t *tres0 = nullptr, *gres = nullptr;      /* tres0[0..sb): the bytes before output byte 0 (ext0: filled by the caller before _back) */
};
/* d_z != nullptr: the walk reads the stream itself (d_tokval / d_dst unused), d_bofs = the scanned block sums of lz77k_dec_sums */
hipError_t lz77k_dec_segments_front(const uint32_t *d_tokval, const uint32_t *d_dst, uint32_t ntok, const lz77x_geom &g, uint8_t *d_out,
                                    void *d_ref, uint32_t n, void *d_tmp, hipStream_t s, bool ext0, lz77k_dec_seg_state &P,
                                    const uint16_t **d_smap, const uint8_t *d_z = nullptr, const uint32_t *d_bofs = nullptr);
uint32_t lz77k_dec_sum_block(void);
hipError_t lz77k_dec_sums(const uint8_t *d_z, uint32_t ntok, const lz77x_geom &g, uint32_t *d_bsum, uint32_t *d_stale_flag, hipStream_t s);
hipError_t lz77k_dec_segments_back(const lz77x_geom &g, uint8_t *d_out, void *d_ref, uint32_t n, const lz77k_dec_seg_state &P, hipStream_t s);
hipError_t lz77k_dec_gather2(uint8_t *d_out, const uint32_t *d_ptr, const unsigned long long *d_unres, uint32_t n, hipStream_t s);
#endif

#endif
