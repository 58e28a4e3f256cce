/*
 * shard.cpp -- ONE stream on SEVERAL devices, both directions, and the host-only shard arithmetic the C ABI exports (what the
 * world-size-2 gloo test drives).
 */
#include "host.h"

using namespace lz77x_host;

extern "C" {

/* host-only: how one stream of n bytes is cut for `shards` devices.  Returns the number of shards actually
 * used (a shard is never smaller than 4*sb + 12 KiB) */
int lz77x_shard_plan(size_t n, int sb, int la, int shards, lz77x_shard *out)
{
    if (check_geom(sb, la) != LZ77X_OK || shards < 1) return LZ77X_E_ARG;
    return shard_plan_range(n, 0, n, sb, la, shards, out);
}

/* host-only: one shard's whole map of boundary cells applied to the cells it starts from (in place):
 * cells[d] <- min(loc[d], min{ cells[c] : dest[c] = d }) */
void lz77x_shard_compose_cells(const uint16_t *dest, const uint32_t *loc, int sb, uint32_t *cells)
{
    std::vector<uint32_t> vo(loc, loc + sb);
    for (int i = 0; i < sb; i++)
        if (dest[i] != 0xFFFFu && cells[i] < vo[dest[i]]) vo[dest[i]] = cells[i];
    memcpy(cells, vo.data(), (size_t)sb * 4);
}

/* host-only: one shard's parse-chain map applied to the running (entry offset, token count) */
void lz77x_shard_compose_chain(const uint8_t *exit_of, const uint32_t *tokens_of, uint32_t *entry, uint64_t *tokens)
{
    *tokens += tokens_of[*entry];
    *entry = exit_of[*entry];
}

/* host-only, decode: where shard d's tokens begin */
uint64_t lz77x_shard_token_cut(uint64_t ntok, int shards, int d)
{
    if (shards < 1 || d <= 0) return 0;
    if (d >= shards) return ntok;
    return (ntok * (uint64_t)d / (uint64_t)shards) & ~(uint64_t)7;
}

/* host-only, decode: one shard's map (the composition of its segments' tails) applied to the sb bytes before it */
void lz77x_shard_compose_tail(const uint16_t *map, int sb, const uint8_t *incoming, uint8_t *outgoing)
{
    for (int i = 0; i < sb; i++) {
        const uint16_t x = map[i];
        outgoing[i] = (x & 0xC000u) == 0x8000u ? incoming[x & 0x3FFFu] : (uint8_t)x;
    }
}

/* the same for windows above 8192 (the tile pass leaves 32-bit states: k_dec_tail_map) */
void lz77x_shard_compose_tail32(const uint32_t *map, int sb, const uint8_t *incoming, uint8_t *outgoing)
{
    for (int i = 0; i < sb; i++) {
        const uint32_t x = map[i];
        outgoing[i] = (x & 0x10000u) ? incoming[x & 0xFFFFu] : (uint8_t)x;
    }
}

}  // extern "C"

LZ77X_HOST_NS {

/* The token positions [t0, t1) of a local buffer of nbytes bytes cut into at most `shards` contiguous shards; the positions
 * before t0 are look-back (t0 = 0: the start of the input; t0 = sb: a later stretch of a long stream, whose first shard
 * starts from carried cells like every other shard starts from its predecessor's), the bytes from t1 on look-ahead.
 * -> shards used (a shard is never smaller than 4*sb + 12 KiB) */
int shard_plan_range(size_t nbytes, size_t t0, size_t t1, int sb, int la, int shards, lz77x_shard *out)
{
    const size_t usb = (size_t)sb, halo = (size_t)la + 64;
    size_t D = (size_t)shards;
    const size_t min_shard = 4 * usb + 3 * (size_t)4096, span = t1 - t0;
    while (D > 1 && span / D < min_shard) D--;
    for (size_t d = 0; d < D && out; d++) {
        lz77x_shard &j = out[d];
        j.first_token_pos = t0 + (uint64_t)span * d / D;
        j.end_token_pos = t0 + (uint64_t)span * (d + 1) / D;
        j.lookback = (d || t0) ? (uint32_t)usb : 0u;
        j.local0 = j.first_token_pos - j.lookback;
        const uint64_t end = j.end_token_pos + halo < nbytes ? j.end_token_pos + halo : (uint64_t)nbytes;
        j.local_bytes = end - j.local0;
        j.steps = j.end_token_pos - j.local0 > usb ? j.end_token_pos - j.local0 - usb : 0;
        j.reserved = 0;
    }
    return (int)D;
}

/* ONE stream decoded on SEVERAL devices: the tokens are cut into D contiguous ranges at multiples of
 * eight tokens (every range then starts on a byte of the stream); device d parses and scans its range and walks
 * its segments with the sb bytes before its first output byte as symbolic references, like any segment's
 * (k_dec_seg ext0) -- nothing it does depends on another shard.  What crosses the cuts is, per shard, ONE map of sb
 * states (a byte value, or "byte i of the bytes before me": the composition of all its segments' tails), chained on
 * the host front to back (D steps of sb table look-ups), after which every shard is handed its sb incoming bytes,
 * resolves its tails and patches its flagged bytes.  No device-to-device traffic, no collective; device memory per
 * shard ~ its share of the tokens and of the output.  *handled = 0: not a case for this path (distance-0 copies,
 * windows above 8192, a shard shorter than the window, too few tokens) -- the caller decodes on one device. */
int decode_sharded(std::vector<Ctx *> &cs, const uint8_t *z, size_t zn, uint8_t **out, size_t *out_n, int *handled)
{
    const double t_begin = now_ms();
    *handled = 0;
    int rc;
    DeviceRestore restore(cs[0]->device);
    if (zn < 4) return LZ77X_E_FORMAT;
    const int sb = z[0] | (z[1] << 8), la = z[2] | (z[3] << 8);               /* lz77.c:157-158 */
    if (sb < 1 || la < 1) return LZ77X_E_FORMAT;
    lz77x_geom g;
    lz77x_make_geom(&g, sb, la);
    if (g.T > 32) return LZ77X_E_FORMAT;
    const uint64_t ntok64 = ((uint64_t)zn * 8 - 32) / (uint64_t)g.T;
    const size_t D = cs.size();
    if (la > 255 || LZ77X_VENV("LZ77X_DECODE_V1") || ntok64 < 64 * D) return LZ77X_OK;
    /* Round 5: a stream of any length, in STRETCHES of tokens (a multiple of eight: a stretch starts on a byte of the stream),
     * every stretch cut over all the devices.  What a stretch needs from everything before it is the last sb bytes of the
     * output -- the incoming bytes of its first shard, where the first stretch has zeros (lz77.c:172-192: a copy reaches at
     * most sb back).  Offsets inside a shard are 32-bit, so a shard takes at most 0xF0000000 / (la + 1) tokens. */
    uint64_t stretch_tok = (uint64_t)D * (((uint64_t)0xF0000000u / (uint64_t)(la + 1)) & ~(uint64_t)7);
    if (const char *e = getenv("LZ77X_DECODE_SHARD_STRETCH")) if (atoll(e) > 0) stretch_tok = ((uint64_t)atoll(e) + 7) & ~(uint64_t)7;
    if (stretch_tok < 64 * D) stretch_tok = (64 * D + 7) & ~(size_t)7;
    /* a stretch's token count and its cuts are 32-bit (k0[], Sh::ntok): with a short lookahead D * 0xF0000000 / (la + 1) passes 2^32.
     * The tail merged into the last stretch (below) is at most (64 + sb) * D tokens more. */
    {
        const uint64_t cap32 = ((uint64_t)0xFFFFFFF8u - (uint64_t)(64 + sb) * D) & ~(uint64_t)7;
        if (stretch_tok > cap32) stretch_tok = cap32;
    }
    uint8_t *buf = nullptr;                                  /* the whole output, grown stretch by stretch */
    size_t buf_cap = 0;
    uint64_t n_done = 0;
    std::vector<uint8_t> carry_in((size_t)sb, 0);            /* the sb bytes before the stretch (zeros before the first) */
    struct BufGuard { uint8_t **p; bool keep = false; ~BufGuard() { if (!keep) { free(*p); *p = nullptr; } } } guard{&buf};
    for (uint64_t T0 = 0; T0 < ntok64;) {
    uint64_t T1 = T0 + stretch_tok < ntok64 ? T0 + stretch_tok : ntok64;
    /* no tail too short to cut: a shard must produce at least a window of bytes (its map is over its last sb bytes), and a
     * token at least one -- a tail of fewer than (64 + sb) tokens per device joins the stretch before it instead of failing
     * the `fits` test below, which would drop everything the earlier stretches decoded */
    if (ntok64 - T1 < (64 + (uint64_t)sb) * D) T1 = ntok64;
    if (T1 - T0 > (uint64_t)0xFFFFFFF8u) return LZ77X_OK;     /* (cannot happen with the cap above; one device rather than a wrapped count) */
    const uint32_t ntok = (uint32_t)(T1 - T0);
    const uint8_t *zs = z + (size_t)(T0 / 8) * (size_t)g.T;  /* the stretch's tokens begin at zs + 4 */
    /* windows the segment walk takes (sb <= 8192): symbolic tails per segment; above: the tile pass on [history | output]
     * with the history still unknown (lz77k_dec_tail_map) */
    const bool tiles = !lz77k_dec_seg_supported(g) || LZ77X_VENV("LZ77X_DECODE_VARIANT");
    const size_t usb = (size_t)sb;
    std::vector<uint32_t> k0(D + 1);
    for (size_t d = 0; d <= D; d++) k0[d] = (uint32_t)lz77x_shard_token_cut(ntok, (int)D, (int)d);
    struct Sh { uint32_t ntok = 0, n = 0; lz77k_dec_seg_state P; const uint16_t *d_smap = nullptr; };
    std::vector<Sh> sh(D);
    /* 1. every shard: its bytes of the stream behind a header of its own, parse, scan (a host thread per shard: the
     *    copies out of the caller's pageable stream block their thread) */
    rc = for_each_shard(D, [&](size_t d) -> int {
        int rc;
        Ctx &c = *cs[d];
        HIPCHK(hipSetDevice(c.device));
        hipStream_t s = c.stream;
        Sh &S = sh[d];
        S.ntok = k0[d + 1] - k0[d];
        const size_t b0 = 4 + (size_t)k0[d] * g.T / 8, b1 = 4 + ((size_t)k0[d + 1] * g.T + 7) / 8;      /* k0 is a multiple of 8: b0 exact */

        const size_t zb = 4 + (b1 - b0);
        if ((rc = c.z.need(zb + 32))) return rc;
        if ((rc = c.h_small.need(128))) return rc;
        HIPCHK(hipMemsetAsync(c.z.as<uint8_t>() + zb, 0, 32, s));
        HIPCHK(hipMemcpyAsync(c.z.p, z, 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        if ((rc = upload_pageable(c, c.z.as<uint8_t>() + 4, zs + b0, b1 - b0))) return rc;       /* (zs: this stretch's part of the stream) */
        if ((rc = c.tokval.need(((size_t)S.ntok + 8) * 4))) return rc;
        if ((rc = c.len1.need(((size_t)S.ntok + 8) * 4))) return rc;
        if ((rc = c.dst.need(((size_t)S.ntok + 8) * 4))) return rc;
        if ((rc = c.scantmp.need(lz77k_scan_tmp_bytes(S.ntok + 1)))) return rc;
        if ((rc = c.flag.need(64))) return rc;
        HIPCHK(hipMemsetAsync(c.flag.as<uint32_t>() + 8, 0, 8, s));
        HIPCHK(lz77k_dec_parse(c.z.as<uint8_t>(), S.ntok, g, c.tokval.as<uint32_t>(), c.len1.as<uint32_t>(), s, c.flag.as<uint32_t>() + 8));
        HIPCHK(hipMemsetAsync(c.len1.as<uint32_t>() + S.ntok, 0, 4, s));
        HIPCHK(lz77k_sum_u32(c.len1.as<uint32_t>(), S.ntok, c.flag.as<unsigned long long>() + 2, s));
        HIPCHK(lz77k_scan_u32(c.len1.as<uint32_t>(), c.dst.as<uint32_t>(), S.ntok + 1, c.scantmp.p, s));
        uint32_t *h = c.h_small.as<uint32_t>();
        HIPCHK(hipMemcpyAsync(h + 4, c.flag.as<unsigned long long>() + 2, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h + 8, c.flag.as<uint32_t>() + 8, 8, hipMemcpyDeviceToHost, s));
        return LZ77X_OK;
    });
    if (rc) return rc;
    uint64_t n = 0;
    bool fits = true;
    std::vector<uint64_t> o0(D + 1, 0);
    for (size_t d = 0; d < D; d++) {
        Ctx &c = *cs[d];
        HIPCHK(hipSetDevice(c.device));
        HIPCHK(hipStreamSynchronize(c.stream));
        const uint32_t *h = c.h_small.as<uint32_t>();
        const uint64_t nd = *reinterpret_cast<const unsigned long long *>(h + 4);
        if (h[8] != 0 || h[9] != 0 || nd < usb || nd > LZ77X_MAX_N) fits = false;   /* distance-0 copies, distances beyond the window / a shard inside one window */
        sh[d].n = (uint32_t)nd;
        o0[d + 1] = o0[d] + nd;
    }
    n = o0[D];
    if (!fits) { HIPCHK(hipSetDevice(cs[0]->device)); return LZ77X_OK; }        /* (what the stretches before produced is dropped: one device) */
    if (n_done + n > buf_cap) {
        size_t ncap = buf_cap ? buf_cap : (size_t)1 << 20;
        while (ncap < n_done + n) ncap += ncap / 2 + ((size_t)64 << 20);
        uint8_t *nb = (uint8_t *)realloc(buf, ncap);
        if (!nb) return LZ77X_E_NOMEM;
        buf = nb;
        buf_cap = ncap;
    }
    const uint32_t pre = tiles ? (uint32_t)((usb + LZ77K_DEC_TILE_BYTES - 1) / LZ77K_DEC_TILE_BYTES * LZ77K_DEC_TILE_BYTES) : 0u;
    std::vector<const unsigned long long *> d_unres(D, nullptr);
    /* (pageable, and the source of asynchronous copies in both branches: it lives until the shards' streams have been
     * synchronised by the gather below) */
    std::vector<std::vector<uint8_t>> incoming(D + 1, std::vector<uint8_t>(usb, 0));
    if (tiles) {
        /* 2t. every shard, on a host thread of its own (the jumping rounds look at a counter between passes): tile pass and
         *     jumping on [pre bytes of history | output]; the history counts as resolved, so afterwards every byte holds its
         *     value or points at one that does -- inside the shard or in the history; then the shard's last sb bytes as a map */
        std::vector<std::vector<uint32_t>> tmap(D, std::vector<uint32_t>(usb));
        rc = for_each_shard(D, [&](size_t d) -> int {
            int rc;
            Ctx &c = *cs[d];
            HIPCHK(hipSetDevice(c.device));
            hipStream_t st = c.stream;
            Sh &S = sh[d];
            const uint32_t N = pre + S.n;
            if ((rc = c.out.need((size_t)N + 16))) return rc;
            if ((rc = c.ptr.need(((size_t)N + 8) * 4))) return rc;
            if ((rc = c.ps.need(((size_t)N + 8) * 4))) return rc;
            if ((rc = c.cells.need(((size_t)N + 8) * 4))) return rc;
            if ((rc = c.tstart.need(lz77k_dec_tile_tmp_bytes(N) + usb * 4 + 256))) return rc;
            if ((rc = c.flag.need(64))) return rc;
            uint8_t *X = c.out.as<uint8_t>();
            HIPCHK(hipMemsetAsync(X, 0, pre, st));
            HIPCHK(lz77k_dec_tiles(c.tokval.as<uint32_t>(), c.dst.as<uint32_t>(), S.ntok, g, X, c.ptr.as<uint32_t>(), N, c.tstart.p, &d_unres[d], st,
                                   lz77k_dec_stale(), pre));
            uint32_t *lists[2] = {c.ps.as<uint32_t>(), c.cells.as<uint32_t>()};
            uint32_t *hcount = c.h_small.as<uint32_t>() + 16;
            uint32_t total = N, rounds = 0;
            const uint32_t *in_list = nullptr;
            for (;;) {
                HIPCHK(hipMemsetAsync(c.flag.p, 0, 4, st));
                HIPCHK(lz77k_dec_jump2(c.ptr.as<uint32_t>(), d_unres[d], total, in_list, lists[rounds & 1], c.flag.as<uint32_t>(), st));
                HIPCHK(hipMemcpyAsync(hcount, c.flag.p, 4, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                in_list = lists[rounds & 1];
                total = *hcount;
                rounds += 1;
                if (!total || rounds > 80) break;
            }
            uint32_t *d_map = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(c.tstart.p) + ((lz77k_dec_tile_tmp_bytes(N) + 255) & ~(size_t)255));
            HIPCHK(lz77k_dec_tail_map(X, c.ptr.as<uint32_t>(), d_unres[d], pre, N, (uint32_t)usb, d_map, st));
            HIPCHK(hipMemcpyAsync(tmap[d].data(), d_map, usb * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            return LZ77X_OK;
        });
        if (rc) return rc;
        /* 3t. the host chains the maps (nothing lies before the first shard: zeros) */
        incoming[0] = carry_in;
        for (size_t d = 0; d < D; d++) lz77x_shard_compose_tail32(tmap[d].data(), sb, incoming[d].data(), incoming[d + 1].data());
        carry_in = incoming[D];                              /* the next stretch's history */
        /* 4t. every shard: its history in, the bytes that point somewhere gathered */
        for (size_t d = 0; d < D; d++) {
            Ctx &c = *cs[d];
            hipError_t e = hipSetDevice(c.device);
            uint8_t *X = c.out.as<uint8_t>();
            if (e == hipSuccess) e = hipMemcpyAsync(X + pre - usb, incoming[d].data(), usb, hipMemcpyHostToDevice, c.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c.stream);                 /* (incoming[] is pageable) */
            if (e == hipSuccess) e = lz77k_dec_gather2(X, c.ptr.as<uint32_t>(), d_unres[d], pre + sh[d].n, c.stream);
            if (e != hipSuccess) { snprintf(g_err, sizeof g_err, "HIP: %s", hipGetErrorString(e)); return LZ77X_E_HIP; }
        }
    } else {
    /* 2. every shard: segment walk, tails composed into the shard's map */
    std::vector<std::vector<uint16_t>> smap(D, std::vector<uint16_t>(usb));
    for (size_t d = 0; d < D; d++) {
        Ctx &c = *cs[d];
        HIPCHK(hipSetDevice(c.device));
        Sh &S = sh[d];
        if ((rc = c.out.need((size_t)S.n + 16))) return rc;
        if ((rc = c.ptr.need(((size_t)S.n + 8) * 4))) return rc;
        if ((rc = c.tstart.need(lz77k_dec_seg_tmp_bytes(S.n, g)))) return rc;
        HIPCHK(lz77k_dec_segments_front(c.tokval.as<uint32_t>(), c.dst.as<uint32_t>(), S.ntok, g, c.out.as<uint8_t>(), c.ptr.p, S.n, c.tstart.p,
                                        c.stream, true, S.P, &S.d_smap));
        HIPCHK(hipMemcpyAsync(smap[d].data(), S.d_smap, usb * 2, hipMemcpyDeviceToHost, c.stream));
    }
    /* 3. the host chains the maps: incoming bytes of every shard (nothing lies before the first: zero bytes, what a
     *    copy from before the start of the output reads in the single-device decoder too) */
    incoming[0] = carry_in;
    for (size_t d = 0; d < D; d++) {
        Ctx &c = *cs[d];
        HIPCHK(hipSetDevice(c.device));
        HIPCHK(hipStreamSynchronize(c.stream));
        lz77x_shard_compose_tail(smap[d].data(), sb, incoming[d].data(), incoming[d + 1].data());
    }
    carry_in = incoming[D];                                  /* the next stretch's history */
    /* 4. every shard: incoming bytes in, tails resolved, flagged bytes patched, output to the host */
    for (size_t d = 0; d < D; d++) {
        Ctx &c = *cs[d];
        hipError_t e = hipSetDevice(c.device);
        if (e == hipSuccess) e = hipMemcpyAsync(sh[d].P.tres0, incoming[d].data(), usb, hipMemcpyHostToDevice, c.stream);
        if (e == hipSuccess) e = lz77k_dec_segments_back(g, c.out.as<uint8_t>(), c.ptr.p, sh[d].n, sh[d].P, c.stream);
        if (e != hipSuccess) { snprintf(g_err, sizeof g_err, "HIP: %s", hipGetErrorString(e)); return LZ77X_E_HIP; }
    }
    }
    rc = for_each_shard(D, [&](size_t d) -> int {           /* the gather: every device fetches its bytes at once */
        Ctx &c = *cs[d];
        HIPCHK(hipSetDevice(c.device));
        HIPCHK(hipStreamSynchronize(c.stream));
        return fetch_result(c, buf + n_done + o0[d], c.out.as<uint8_t>() + pre, sh[d].n);
    });
    if (rc) return rc;
    n_done += n;
    T0 = T1;
    }   /* stretches */
    HIPCHK(hipSetDevice(cs[0]->device));
    memset(&g_stats, 0, sizeof g_stats);
    g_stats.n = n_done;
    g_stats.zn = zn;
    g_stats.ntok = ntok64;
    g_stats.total_ms = now_ms() - t_begin;
    if (!buf) buf = (uint8_t *)malloc(1);
    if (!buf) return LZ77X_E_NOMEM;
    guard.keep = true;
    *out = buf;
    *out_n = (size_t)n_done;
    *handled = 1;
    return LZ77X_OK;
}

struct ShardJob {
    Ctx *c = nullptr;
    uint64_t a = 0, b = 0, gpos0 = 0;      /* tokens [a, b) (global), local 0 = gpos0 */
    uint32_t look = 0, nloc = 0, E = 0, nx = 0, entry = 0, start = 0, ntok = 0, nsub = 0;
    uint64_t K0 = 0;
    lz77k_prio_plan P;
    const uint32_t *d_tbase = nullptr;
    uint32_t *h = nullptr;                  /* pinned scratch of this shard (c->h_tbase) */
};

/* What one stretch of a long stream hands to the next (the SegCarry of encode_pipe.cpp, across ALL its shards): the state of
 * lz77.c's two sequential loops at the cut -- where the next token starts, how many there were, the last four token words (a
 * stream word can straddle the cut) and the sb live priorities renumbered by rank (the tie-break only compares them). */
struct StretchCarry {
    bool first = true;
    uint64_t chain_pos = 0;        /* global position of the next token */
    uint64_t ntok = 0;
    uint32_t tail[4] = {0, 0, 0, 0};
    uint32_t ntail = 0;
    std::vector<uint32_t> cells;   /* sb ranks (after the first stretch) */
};

/* One stretch: the local buffer src[0, nbytes) starts at global position origin; its tokens are the chain positions in
 * [t0, t1) (local; t0 = 0 for the first stretch, sb afterwards: the cells before t0 are look-back and hold the carried
 * priorities), cut into position shards.  last: the input ends with this stretch. */
static int encode_sharded_stretch(std::vector<Ctx *> &cs, const uint8_t *src, size_t nbytes, size_t t0, size_t t1, uint64_t origin, bool last_stretch,
                                  StretchCarry &carry, const lz77x_geom &g, Sink &sink, double *host_serial_out)
{
    const size_t usb = (size_t)g.sb;
    const uint32_t csub = lz77k_chain_sub();
    std::vector<lz77x_shard> plan(cs.size());
    const int planned = shard_plan_range(nbytes, t0, t1, g.sb, g.la, (int)cs.size(), plan.data());
    if (planned < 1) return LZ77X_E_ARG;
    const size_t D = (size_t)planned;
    std::vector<ShardJob> J(D);
    int rc;
    const char *tv = LZ77X_VENV("LZ77X_TOKEN_VARIANT");
    const int tvariant = tv ? atoi(tv) : 0;
    auto dev = [&](ShardJob &j) -> int { HIPCHK(hipSetDevice(j.c->device)); return LZ77X_OK; };
    const size_t hwords = 4 * usb + 1024;                 /* pinned words per shard beyond the tbase copy */

    /* -- phase A: every shard on its own: input, match stage, the parse chain's maps, round masks.  One host thread
     *    per shard: the copy out of the caller's pageable buffer blocks its thread (hipMemcpyAsync stages it), and D of
     *    them in sequence on one thread were D x n/D bytes of serial PCIe time before the last device saw a byte -- */
    DeviceRestore restore(cs[0]->device);
    double host_serial_ms = 0;                              /* host time between the phases that no device overlaps */
    std::vector<uint32_t> launches_of(D, 0);
    rc = for_each_shard(D, [&](size_t d) -> int {
        int rc;
        ShardJob &j = J[d];
        j.c = cs[d];
        Ctx &c = *j.c;
        j.a = plan[d].first_token_pos;
        j.b = plan[d].end_token_pos;
        j.look = plan[d].lookback;
        j.gpos0 = plan[d].local0;
        j.nloc = (uint32_t)plan[d].local_bytes;
        j.E = (uint32_t)(j.b - j.gpos0);
        j.nx = (uint32_t)plan[d].steps;
        if ((rc = dev(j))) return rc;
        hipStream_t s = c.stream;
        const size_t np = j.nloc;
        const uint32_t nregions = (uint32_t)(((size_t)j.E + g.TILE - 1) / g.TILE) < (uint32_t)((np + g.TILE - 1) / g.TILE)
                                      ? (uint32_t)(((size_t)j.E + g.TILE - 1) / g.TILE) : (uint32_t)((np + g.TILE - 1) / g.TILE);
        uint32_t batch = nregions;
        {
            const size_t per = lz77k_match_scratch_bytes(g, 1);
            const uint32_t fit = (uint32_t)(((size_t)2 << 30) / per);
            if (batch > fit) batch = fit ? fit : 1;
        }
        const size_t span = (size_t)j.E - j.look;
        const size_t idx_span = span + 3 * usb + 64;
        if ((rc = c.in.need(np + LZ77X_PAD + 64))) return rc;
        if ((rc = c.scratch.need(lz77k_match_scratch_bytes(g, batch)))) return rc;
        if ((rc = c.ps.need((np + 8) * 4))) return rc;
        if ((rc = c.maxlen.need(np + 64))) return rc;
        if ((rc = c.xval.need((np + 8) * 4))) return rc;
        if ((rc = c.chain.need((np + 8) * 4))) return rc;
        if ((rc = c.tokval.need((np + 16) * 4))) return rc;
        if ((rc = c.ofs.need((idx_span + 8) * 4))) return rc;
        if ((rc = c.ent.need((idx_span + 8) * 8))) return rc;
        if ((rc = c.scantmp.need(lz77k_scan_tmp_bytes((uint32_t)idx_span + 1)))) return rc;
        if ((rc = c.tstart.need(lz77k_tokens_tmp_bytes((uint32_t)idx_span, g)))) return rc;
        if ((rc = c.flag.need(64))) return rc;
        if ((rc = c.prio_tmp.need(lz77k_prio_tmp_bytes(j.nx, g.sb)))) return rc;
        if ((rc = c.chain_tmp.need(lz77k_chain_tmp_bytes((uint32_t)span, g.la)))) return rc;
        if ((rc = c.h_small.need(128))) return rc;
        /* large windows: the regions' rank + inverse arrays stay resident for the rank-order tie-break, and the (block,
         * first byte) buckets of the tokens of length one (as in a segment of the single-device pipeline) */
        if (!g.fast) {
            if ((rc = c.ranks_all.need((size_t)nregions * (2 * (size_t)g.RP + 8) * sizeof(uint32_t)))) return rc;
            if ((rc = c.bidx.need(lz77k_tokens_index_bytes(g, idx_span)))) return rc;
        }
        const uint32_t nsub_max = (uint32_t)((span + csub - 1) / csub);
        if ((rc = c.h_tbase.need(((size_t)nsub_max + 2 + hwords) * 4))) return rc;
        j.h = c.h_tbase.as<uint32_t>() + nsub_max + 2;
        HIPCHK(hipMemsetAsync(c.flag.p, 0, 64, s));
        if ((rc = upload_pageable(c, c.in.as<uint8_t>(), src + j.gpos0, np))) return rc;
        HIPCHK(lz77k_fill_pad(c.in.as<uint8_t>(), j.nloc, s));
        for (uint32_t r0 = 0; r0 < nregions; r0 += batch) {
            const uint32_t nr = nregions - r0 < batch ? nregions - r0 : batch;
            HIPCHK(lz77k_match(c.in.as<uint8_t>(), j.nloc, g, r0, nr, c.ps.as<uint32_t>(), c.maxlen.as<uint8_t>(), c.scratch.p, 0, s, nullptr,
                               g.fast ? nullptr : c.ranks_all.as<uint32_t>()));
            launches_of[d]++;
        }
        const uint8_t *d_wexit = nullptr;
        const uint32_t *d_wcnt = nullptr;
        HIPCHK(lz77k_chain_maps(c.maxlen.as<uint8_t>(), j.E, g.la, c.chain_tmp.p, s, j.look, true, &d_wexit, &d_wcnt));
        HIPCHK(hipMemcpyAsync(j.h, d_wcnt, 256 * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(j.h + 256, d_wexit, 256, hipMemcpyDeviceToHost, s));
        /* c.look: [0, sb) the cells the first shard of a later stretch starts from, [sb + 8, ..) the cells the last shard leaves */
        if ((rc = c.look.need((size_t)2 * (usb + 8) * 4))) return rc;
        const uint32_t *d_carried = nullptr;
        if (d == 0 && !carry.first) {
            HIPCHK(hipMemcpyAsync(c.look.p, carry.cells.data(), usb * 4, hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));                 /* (pageable source) */
            d_carried = c.look.as<uint32_t>();
        }
        HIPCHK(lz77k_prio_begin(j.P, c.ps.as<uint32_t>(), j.nx, g.sb, c.xval.as<uint32_t>(), c.prio_tmp.p, (uint32_t)j.gpos0, d_carried, s));
        HIPCHK(hipStreamSynchronize(s));
        return LZ77X_OK;
    });
    if (rc) return rc;
    for (uint32_t l : launches_of) g_stats.match_launches += l;
    double t_serial = now_ms();

    /* -- the parse chain across the cuts (lz77.c:98): entry offset and first-token index of every shard -- */
    {
        /* (a later stretch: the token its predecessor's last one led to; at most la past the cut) */
        uint32_t e = carry.first ? 0u : (uint32_t)(carry.chain_pos - (origin + t0));
        uint64_t K = carry.ntok;
        for (ShardJob &j : J) {
            j.entry = e;
            j.K0 = K;
            j.ntok = j.h[e];
            lz77x_shard_compose_chain(reinterpret_cast<const uint8_t *>(j.h + 256), j.h, &e, &K);
            j.start = j.look + j.entry;
        }
        carry.chain_pos = origin + t1 + e;
        carry.ntok = K;
    }
    /* (the second half of the chain -- every shard's tokens from its entry on -- is enqueued by the shard's own thread below) */

    /* -- the priority recurrence across the cuts (tree.c:202-231): all shards iterate together.  Round 6: every shard has a
     *    host thread of its own for the whole iteration (as phase A has): it enqueues its maps, waits for ITS device, and
     *    after the exchange enqueues its sweep and waits again -- D devices work side by side and no thread waits for a
     *    device that is not its own.  What is serial is what thread 0 does alone between two barriers: chaining the D maps
     *    (D x sb look-ups) and reading the D flip summaries.  Round 5 drove every shard from one thread -- per iteration
     *    2 D hipSetDevice + enqueue rounds and two sync_all loops over all devices. -- */
    {
        std::vector<uint32_t> v(usb);
        std::vector<char> flipped(D, 0);
        uint32_t iters = 0;
        /* the error front of lz77k_prio (k_prio.hip), across the shards: flips and the first open block of the last six iterations */
        uint64_t hist_f[6] = {0, 0, 0, 0, 0, 0}, hist_b[6] = {0, 0, 0, 0, 0, 0};
        int front_budget = 64;
        if (const char *me = getenv("LZ77X_PRIO_MAX_ITERS")) if (atoi(me) > 0) front_budget = atoi(me);
        bool on_host = false;
        struct Team {
            size_t D; std::mutex m; std::condition_variable cv; size_t waiting = 0; uint64_t gen = 0;
            std::atomic<int> failed{0};
            bool done = false, hopeless = false;
            void barrier()
            {
                std::unique_lock<std::mutex> lk(m);
                const uint64_t g0 = gen;
                if (++waiting == D) { waiting = 0; gen++; cv.notify_all(); }
                else cv.wait(lk, [&] { return gen != g0; });
            }
        } team;
        team.D = D;
        host_serial_ms += now_ms() - t_serial;               /* (the chain's exchange and the enqueue of its second half) */
        /* What is serial in the joint iteration, honestly: from the moment the LAST shard's device has finished a phase to the
         * moment the FIRST shard's thread starts enqueuing the next one -- thread 0's work between the barriers AND the
         * wake-ups of the barriers themselves (all stamps are one steady clock). */
        double serial_it = 0;
        std::vector<double> t_done(D, 0.0), t_begin(D, 0.0);
        auto last_done = [&]() { double m = 0; for (double t : t_done) m = t > m ? t : m; return m; };
        auto first_begin = [&]() { double m = t_begin[0]; for (double t : t_begin) m = t < m ? t : m; return m; };
        bool have_done = false;                               /* t_done holds the stamps of a finished phase */
        rc = run_team(D, [&](size_t d) -> int {
            ShardJob &j = J[d];
            int my_rc = LZ77X_OK;
            /* a failing shard keeps meeting the barriers; everybody leaves together at the end of the iteration */
            auto step = [&](auto fn) { if (my_rc == LZ77X_OK && !team.failed.load()) { my_rc = fn(); if (my_rc) team.failed.store(1); } };
            step([&]() -> int {
                Ctx &c = *j.c;
                HIPCHK(hipSetDevice(c.device));
                HIPCHK(lz77k_chain_finish(c.maxlen.as<uint8_t>(), j.E, g.la, c.chain.as<uint32_t>(), c.chain_tmp.p, c.stream, j.look, j.entry, &j.d_tbase,
                                          &j.nsub, nullptr));
                HIPCHK(hipMemcpyAsync(c.h_tbase.p, j.d_tbase, ((size_t)j.nsub + 1) * 4, hipMemcpyDeviceToHost, c.stream));
                return LZ77X_OK;
            });
            for (;;) {
                /* 1. this shard's maps from its current gates; all but the last shard: the whole plan as ONE map, to the host */
                t_begin[d] = now_ms();
                step([&]() -> int {
                    const uint16_t *d_sdest = nullptr;
                    const uint32_t *d_sloc = nullptr;
                    HIPCHK(hipSetDevice(j.c->device));
                    if (d + 1 < D) {
                        HIPCHK(lz77k_prio_maps(j.P, j.c->stream, true, &d_sdest, &d_sloc));
                        if (j.nx) {
                            HIPCHK(hipMemcpyAsync(j.h + 512, d_sloc, usb * 4, hipMemcpyDeviceToHost, j.c->stream));
                            HIPCHK(hipMemcpyAsync(j.h + 512 + usb, d_sdest, usb * 2, hipMemcpyDeviceToHost, j.c->stream));
                        }
                    } else
                        HIPCHK(lz77k_prio_maps(j.P, j.c->stream, false, nullptr, nullptr));   /* the last shard's whole map is nobody's input */
                    if (d + 1 < D) HIPCHK(hipStreamSynchronize(j.c->stream));                 /* (the last shard's maps only precede its own sweep) */
                    return LZ77X_OK;
                });
                const double my_done1 = now_ms();
                team.barrier();
                /* 2. thread 0: the cells every shard starts from -- the maps chained front to back (lz77x_shard_compose_cells) */
                if (d == 0 && !team.failed.load()) {
                    /* (the gap behind the sweeps of the iteration before: its last device done -> this iteration's first enqueue) */
                    if (have_done) serial_it += first_begin() - last_done();
                    if (carry.first) for (size_t i = 0; i < usb; i++) v[i] = (uint32_t)i;      /* the start of the input: every cell its own position */
                    else memcpy(v.data(), carry.cells.data(), usb * 4);                         /* a later stretch: the carried ranks */
                    for (size_t q = 0; q < D; q++) {
                        ShardJob &jq = J[q];
                        if (q > 0) memcpy(jq.h + 512 + 2 * usb, v.data(), usb * 4);            /* pinned copy of the cells shard q starts from */
                        if (q + 1 < D && jq.nx)                                               /* v <- shard q's whole map applied to v (no step: the cells pass through) */
                            lz77x_shard_compose_cells(reinterpret_cast<const uint16_t *>(jq.h + 512 + usb), jq.h + 512, g.sb, v.data());
                    }
                }
                team.barrier();
                t_done[d] = my_done1;                          /* (thread 0 has read the stamps of the phase before: two barriers ago) */
                t_begin[d] = now_ms();
                /* 3. the cells in, the exact sweep of every block that is not final, the flip summary out */
                step([&]() -> int {
                    HIPCHK(hipSetDevice(j.c->device));
                    if (d > 0) HIPCHK(lz77k_prio_set_in0(j.P, j.h + 512 + 2 * usb, hipMemcpyHostToDevice, j.c->stream));
                    /* (the last shard of a stretch that is not the last: the sb cells left live after its last step) */
                    uint32_t *d_state = (d + 1 == D && !last_stretch) ? j.c->look.as<uint32_t>() + usb + 8 : nullptr;
                    HIPCHK(lz77k_prio_sweep(j.P, j.c->stream, j.c->h_small.as<uint32_t>() + 8, d_state));
                    HIPCHK(hipStreamSynchronize(j.c->stream));
                    return LZ77X_OK;
                });
                const double my_done3 = now_ms();
                team.barrier();
                /* 4. thread 0: who flipped, which blocks are final, is the budget enough */
                if (d == 0) {
                    /* (the gap behind the maps: their last device done -> the first sweep enqueued) */
                    serial_it += first_begin() - last_done();
                    if (team.failed.load()) team.done = true;
                    else {
                        iters++;
                        bool any = false, earlier = false;
                        uint64_t flips = 0, blocks_before = 0, first_open = 0;
                        for (size_t q = 0; q < D; q++) {
                            const uint32_t *hf = J[q].c->h_small.as<uint32_t>() + 8;
                            flipped[q] = hf[0] != 0;
                            flips += hf[0];
                            if (flipped[q] && !any) first_open = blocks_before + hf[1];
                            blocks_before += J[q].P.NB;
                            /* a shard after one that still changes may be handed different cells next time: nothing of it is final */
                            lz77k_prio_advance(J[q].P, hf, earlier);
                            earlier = earlier || flipped[q];
                            any = any || flipped[q];
                        }
                        if (!any) team.done = true;
                        else {
                            for (int q = 0; q < 5; q++) { hist_f[q] = hist_f[q + 1]; hist_b[q] = hist_b[q + 1]; }
                            hist_f[5] = flips;
                            hist_b[5] = first_open;
                            if ((int)iters >= front_budget || lz77x_prio_hopeless(hist_f, hist_b, blocks_before - first_open, (int)iters, front_budget))
                                team.done = team.hopeless = true;
                        }
                    }
                }
                team.barrier();
                t_done[d] = my_done3;
                if (d == 0) have_done = true;                  /* (read by thread 0 only) */
                if (team.done) break;
            }
            return my_rc;
        });
        if (rc) return rc;
        host_serial_ms += serial_it;
        t_serial = last_done();                                /* (the tail: from the last sweep's end on) */
        if (team.hopeless) {
            /* one block an iteration (input that repeats with a period of about a window): the recurrence of the whole
             * stretch on a host core instead, shard after shard -- each shard's evictions follow its predecessor's, the
             * cells one leaves behind are the cells the next starts from (hoststage.c lz77x_prio_run_cells) */
            if (carry.first) for (size_t i = 0; i < usb; i++) v[i] = (uint32_t)i;
            else memcpy(v.data(), carry.cells.data(), usb * 4);
            const double th = now_ms();
            for (size_t d = 0; d < D; d++) {
                ShardJob &j = J[d];
                if ((rc = dev(j))) return rc;
                std::vector<uint32_t> h_ps, h_xv, out(usb);
                h_ps.resize((size_t)j.nx + 16);
                h_xv.resize((size_t)j.nx + 16);
                HIPCHK(hipMemcpy(h_ps.data(), j.c->ps.p, (size_t)j.nx * 4, hipMemcpyDeviceToHost));
                if (d > 0) {
                    uint32_t *pin = j.h + 512 + 2 * usb;            /* the cells this shard REALLY starts from: the tie-break's look-back */
                    memcpy(pin, v.data(), usb * 4);
                    HIPCHK(lz77k_prio_set_in0(j.P, pin, hipMemcpyHostToDevice, j.c->stream));
                    HIPCHK(hipStreamSynchronize(j.c->stream));
                }
                if (!lz77x_prio_run_cells(h_ps.data(), j.nx, g.sb, v.data(), (uint32_t)j.gpos0, h_xv.data(), out.data())) return LZ77X_E_NOMEM;
                HIPCHK(hipMemcpy(j.c->xval.p, h_xv.data(), (size_t)j.nx * 4, hipMemcpyHostToDevice));
                v = out;
            }
            g_stats.host_stageb_ms += now_ms() - th;
            on_host = true;
            t_serial = now_ms();                               /* (the host loop is reported as host_stageb_ms, not as exchange time) */
        }
        g_stats.prio_iters += iters;
        if (!last_stretch) {
            /* the cells left live, renumbered by rank (sb values): what the next stretch starts from */
            ShardJob &j = J[D - 1];
            if ((rc = dev(j))) return rc;
            std::vector<uint32_t> st(usb);
            if (on_host) memcpy(st.data(), v.data(), usb * 4);  /* (the host loop's last cells) */
            else if (j.nx) HIPCHK(hipMemcpy(st.data(), j.c->look.as<uint32_t>() + usb + 8, usb * 4, hipMemcpyDeviceToHost));
            else memcpy(st.data(), v.data(), usb * 4);      /* (no step in the last shard: what it was handed) */
            std::vector<std::pair<uint32_t, uint32_t>> order(usb);
            for (size_t i = 0; i < usb; i++) order[i] = {st[i], (uint32_t)i};
            std::sort(order.begin(), order.end());
            carry.cells.resize(usb);
            for (size_t r = 0; r < usb; r++) carry.cells[order[r].second] = (uint32_t)r;
        }
    }

    /* -- tokens: every shard resolves its own (look-back priorities = the cells it started from), a host thread per shard:
     *    each waits for its own device only, and the output piece's buffer is allocated here, beside the kernels -- */
    const uint64_t T = (uint64_t)g.T;
    const uint64_t K_all = D ? J[D - 1].K0 + J[D - 1].ntok : 0;
    const uint64_t zn_total = stream_bytes(K_all, g.T);                /* (of the whole stream so far: it ends here when last_stretch) */
    auto words_of = [&](size_t d, uint64_t *wlo_out) -> uint64_t {
        const ShardJob &j = J[d];
        const bool last = d + 1 == D && last_stretch;
        const uint64_t K0 = j.K0, K1 = K0 + j.ntok;
        const uint64_t wlo = K0 == 0 ? 0 : (32 + K0 * T) / 32;
        const uint64_t whi = last ? (zn_total + 3) / 4 : (32 + K1 * T) / 32;
        *wlo_out = wlo;
        return whi > wlo ? whi - wlo : 0;
    };
    host_serial_ms += now_ms() - t_serial;
    rc = for_each_shard(D, [&](size_t d) -> int {
        int rc;
        ShardJob &j = J[d];
        Ctx &c = *j.c;
        HIPCHK(hipSetDevice(c.device));
        hipStream_t s = c.stream;
        HIPCHK(hipStreamSynchronize(s));                                   /* (h_tbase has landed) */
        const uint32_t *h_tbase = c.h_tbase.as<uint32_t>();
        if (h_tbase[j.nsub] != j.ntok) { snprintf(g_err, sizeof g_err, "shard chain mismatch: %u vs %u", h_tbase[j.nsub], j.ntok); return LZ77X_E_HIP; }
        const uint32_t *look = j.look ? reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(j.P.tmp) + j.P.o_in) : nullptr;
        const size_t b = j.start, e = j.E;
        if (e > b) {
            const size_t x_done = e > usb ? e - usb : 0;
            const uint32_t dbase = b > usb ? (uint32_t)(b - usb) : 0u;
            const uint32_t xa = dbase > (uint32_t)g.sb ? dbase - (uint32_t)g.sb : 0u;
            HIPCHK(lz77k_xfer_index(c.ps.as<uint32_t>(), c.xval.as<uint32_t>(), xa, (uint32_t)x_done, dbase, (uint32_t)e, c.ofs.as<uint32_t>(),
                                    c.ent.as<uint2>(), c.scantmp.p, s, 0u, c.flag.as<unsigned long long>() + 1, g.fast ? (uint32_t)g.sb : 0u));
            HIPCHK(lz77k_tokens(c.in.as<uint8_t>(), j.nloc, g, c.chain.as<uint32_t>(), j.ntok, c.maxlen.as<uint8_t>(), c.ofs.as<uint32_t>(),
                                c.ent.as<uint2>(), dbase, (uint32_t)b, (uint32_t)e, c.tokval.as<uint32_t>() + 4, c.tstart.as<uint32_t>(),
                                g.fast ? nullptr : c.bidx.p, tvariant, s, nullptr, g.fast ? nullptr : c.ranks_all.as<uint32_t>(), look, j.look,
                                (uint32_t)j.gpos0));
        }
        const uint32_t have = j.ntok < 4 ? j.ntok : 4;
        if (have) HIPCHK(hipMemcpyAsync(j.h, c.tokval.as<uint32_t>() + 4 + j.ntok - have, have * 4, hipMemcpyDeviceToHost, s));
        uint64_t wlo = 0;
        if ((rc = c.out.need(words_of(d, &wlo) * 4 + 16))) return rc;     /* (beside the tie-break: hipMalloc of a new piece takes milliseconds) */
        HIPCHK(hipStreamSynchronize(s));
        j.h[8] = have;
        return LZ77X_OK;
    });
    if (rc) return rc;
    t_serial = now_ms();

    /* -- pack (lz77.c:246-252): each shard the stream words its tokens start in, with its predecessors' last
     *    tokens in front; then the pieces leave in order -- */
    uint32_t tail[4] = {carry.tail[0], carry.tail[1], carry.tail[2], carry.tail[3]};
    uint32_t ntail = carry.ntail;
    std::vector<uint64_t> piece(D, 0);
    for (size_t d = 0; d < D; d++) {
        ShardJob &j = J[d];
        Ctx &c = *j.c;
        if ((rc = dev(j))) return rc;
        const bool last = d + 1 == D && last_stretch;
        const uint64_t K0 = j.K0, K1 = K0 + j.ntok;
        uint64_t wlo = 0;
        const uint64_t nw = words_of(d, &wlo);
        uint32_t *pin = j.h + 16;
        memcpy(pin, tail, sizeof tail);
        if (ntail) HIPCHK(hipMemcpyAsync(c.tokval.as<uint32_t>() + 4 - ntail, pin + 4 - ntail, ntail * 4, hipMemcpyHostToDevice, c.stream));
        HIPCHK(lz77k_pack_range(c.tokval.as<uint32_t>() + 4 - ntail, K0 - ntail, K1, g, c.out.as<uint32_t>(), wlo, nw, c.stream));
        piece[d] = last ? zn_total - 4 * wlo : 4 * nw;
        /* the last four tokens so far */
        const uint32_t have = j.h[8];
        uint32_t merged[8], m = 0;
        for (uint32_t i = 0; i < ntail; i++) merged[m++] = tail[4 - ntail + i];
        for (uint32_t i = 0; i < have; i++) merged[m++] = j.h[i];
        ntail = m < 4 ? m : 4;
        for (uint32_t i = 0; i < ntail; i++) tail[4 - ntail + i] = merged[m - ntail + i];
    }
    memcpy(carry.tail, tail, sizeof tail);
    carry.ntail = ntail;
    carry.first = false;
    host_serial_ms += now_ms() - t_serial;
    {
        /* the pieces leave: every device fetches its own into the sink's memory at once when the sink is host memory
         * (the gather north_star describes), else one after the other in stream order */
        uint64_t all = 0;
        std::vector<uint64_t> at(D, 0);
        for (size_t d = 0; d < D; d++) { at[d] = all; all += piece[d]; }
        std::vector<unsigned long long> cnt(D, 0);
        uint8_t *base = sink.direct((size_t)all);
        if (base) {
            rc = for_each_shard(D, [&](size_t d) -> int {
                ShardJob &j = J[d];
                HIPCHK(hipSetDevice(j.c->device));
                HIPCHK(hipStreamSynchronize(j.c->stream));
                HIPCHK(hipMemcpy(&cnt[d], j.c->flag.as<unsigned long long>() + 1, 8, hipMemcpyDeviceToHost));
                return fetch_result(*j.c, base + at[d], j.c->out.p, (size_t)piece[d]);
            });
            if (rc) return rc;
        } else {
            for (size_t d = 0; d < D; d++) {
                ShardJob &j = J[d];
                if ((rc = dev(j))) return rc;
                if ((rc = sink.write(*j.c, j.c->out.as<uint8_t>(), (size_t)piece[d], j.c->stream))) return rc;
                HIPCHK(hipMemcpy(&cnt[d], j.c->flag.as<unsigned long long>() + 1, 8, hipMemcpyDeviceToHost));
            }
        }
        for (unsigned long long c : cnt) g_stats.transfers += c;
    }
    HIPCHK(hipSetDevice(cs[0]->device));
    *host_serial_out += host_serial_ms;
    return LZ77X_OK;
}

/* ONE stream on SEVERAL devices, any length (round 5: shards and stretches compose).  The input is taken in stretches of at
 * most `stretch` token positions (LZ77X_SHARD_STRETCH; default one GiB per device, 256 MB per device when the bytes come out
 * of a FILE*: the host then holds one stretch, not the file); a stretch is cut into position shards, one per device, and
 * hands the next one the state of lz77.c's two loops (StretchCarry) exactly as a segment of the single-device pipeline does.
 * Positions inside a stretch are 32-bit, token counts and stream offsets 64-bit: no LZ77X_E_TOOBIG. */
int encode_sharded(std::vector<Ctx *> &cs, HostWindow &in, const lz77x_geom &g, Sink &sink)
{
    const double t_begin = now_ms();
    memset(&g_stats, 0, sizeof g_stats);
    const size_t usb = (size_t)g.sb, halo = (size_t)g.la + 64, D = cs.size();
    size_t stretch = D * (in.from_file() ? (size_t)256 << 20 : (size_t)1 << 30);
    if (const char *e = getenv("LZ77X_SHARD_STRETCH")) if (atoll(e) > 0) stretch = (size_t)atoll(e);
    const size_t most = ((size_t)0xF0000000u) - usb - halo;            /* positions of a stretch are 32-bit (+ voff) */
    if (stretch > most) stretch = most;
    if (stretch < 4 * usb + 3 * (size_t)4096) stretch = 4 * usb + 3 * (size_t)4096;
    StretchCarry carry;
    uint64_t origin = 0, n_total = 0;
    double host_serial_ms = 0;
    for (;;) {
        const size_t t0 = carry.first ? 0 : usb;
        const size_t want = t0 + stretch + halo;
        const uint8_t *p = nullptr;
        size_t got = 0;
        int rc = in.get(origin, want, &p, &got);
        if (rc) return rc;
        const bool last = got < want;
        const size_t t1 = last ? got : t0 + stretch;
        if ((rc = encode_sharded_stretch(cs, p, got, t0, t1, origin, last, carry, g, sink, &host_serial_ms))) return rc;
        n_total = origin + got;
        if (last) break;
        origin += t1 - usb;
    }
    g_stats.host_chain_ms = 0;
    g_stats.copy_ms = host_serial_ms;                      /* sharded: host time no device overlaps (exchange + enqueue) */
    g_stats.n = n_total;
    g_stats.zn = sink.total;
    g_stats.ntok = carry.ntok;
    g_stats.total_ms = now_ms() - t_begin;
    TRACE("encode_sharded total", t_begin);
    return LZ77X_OK;
}

/* Which pipeline an encode takes: everything on the device when the geometry allows it (one device,
 * sb <= 4096, production kernels), the round-1 pipeline with the two recurrences on host cores
 * otherwise (LZ77X_HOST_STAGEB=1 forces it) or when the gate iteration gives up. */
bool device_pipeline_ok(size_t ndev, const lz77x_geom &g)
{
    const char *hs = LZ77X_VENV("LZ77X_HOST_STAGEB"), *vs = LZ77X_VENV("LZ77X_MATCH_VARIANT");
    return ndev == 1 && g.shifted && lz77k_prio_supported(g.sb) && !(hs && atoi(hs)) && !(vs && atoi(vs)) && !LZ77X_VENV("LZ77X_SERIAL");
}

/* memory -> sink.  src: host or device pointer of n bytes */
int encode_any(std::vector<Ctx *> &cs, const void *src, bool src_on_device, size_t n, const lz77x_geom &g, hipStream_t s, Sink &sink)
{
    Ctx &c0 = *cs[0];
    const void *host_src = src;
    bool host_on_device = src_on_device;
    uint32_t iters = 0;                                    /* gate iterations spent before giving up */
    /* one stream over several devices: every window size on the device pipeline (large windows compose their shards'
     * whole-plan maps through HBM, lz77kw_compose_all) */
    if (cs.size() > 1 && !src_on_device && device_pipeline_ok(1, g)) {
        MemWindow mw(reinterpret_cast<const uint8_t *>(src), n);
        return encode_sharded(cs, mw, g, sink);
    }
    if (device_pipeline_ok(cs.size(), g)) {
        MemSource ms(src, n, src_on_device);
        bool fallback = false;
        size_t nfb = 0;
        const int rc = encode_stream_device(c0, ms, sink, g, s, &fallback, &nfb);
        if (rc != LZ77X_OK || !fallback) return rc;
        /* (the other pipeline takes the context over: nothing of the attempt may still be in flight on any of its streams) */
        HIPCHK(hipDeviceSynchronize());
        host_src = c0.in.p;                                /* single segment: the whole input is in c.in */
        host_on_device = true;
        iters = g_stats.prio_iters;
    }
    size_t zn = 0;
    int rc = encode_core_host(cs, host_src, host_on_device, n, g, s, &zn);
    g_stats.prio_iters = iters;
    if (rc) return rc;
    return sink.write(c0, c0.out.as<uint8_t>(), zn, s);
}

}  // namespace lz77x_host
