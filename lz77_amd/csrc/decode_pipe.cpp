/*
 * decode_pipe.cpp -- the decoder (replaces lz77.c:148-197): a stream of any length in token ranges through bounded device memory, two ranges in flight.
 */
#include "host.h"

LZ77X_HOST_NS {

/* lz77.c:160-195 decodes a stream of any length through a buffer of 3*SB+LA bytes.  Here a stream is decoded RANGE by
 * range: a range is a run of consecutive tokens that starts on a multiple of eight tokens -- tokens have a fixed width
 * T, so it starts on a byte of the stream -- of at most `range_tokens` tokens and at most `range_bytes` bytes of output
 * (a range whose tokens expand further is cut at the last multiple of eight that fits; what was read beyond the cut
 * opens the next range).  Device memory is a function of those two numbers and not of the stream's length, host memory
 * is two staging slots; the stream may be a pipe and may decode to more than 4 GiB (offsets inside a range are 32-bit,
 * counts across ranges 64-bit).  What a range needs from everything before it is what the reference's buffer holds
 * (DecCarry): the last cb bytes of the output, and for streams with distance-0 copies (a power-of-two -s)
 * the image of the staging buffer's upper 2*SB+LA bytes and where its current pass began. */
struct DecCarry {
    uint32_t cb = 0;             /* bytes of history a copy can reach: max(sb, 2^ob - 1) (the offset field is wider than sb unless sb = 2^k - 1) */
    uint32_t W = 0;              /* 3*sb + la: the reference's buffer (lz77.c:160) */
    uint32_t pre = 0;            /* the paths that keep pointers work on [pre bytes of history | the range's output]: a multiple of the tile size */
    bool track = false;          /* distance-0 copies are followed: pass structure + image */
    int cur = 0;                 /* which half of the double-buffered device state is current */
    uint8_t *d_carry[2] = {nullptr, nullptr};
    uint8_t *d_img[2] = {nullptr, nullptr};
    uint64_t produced = 0;       /* output bytes before the range */
    uint64_t pass_start = 0;     /* output offset at which the staging buffer's current pass began */
    bool first_pass = true;      /* ... and it is the first pass of the stream (it starts at buffer index 0, the others at sb) */
};

/* Where the reference's staging buffer starts a new pass (lz77.c:172-175: when the next token's copy would not fit) is a
 * sequential function of the token lengths -- one step per ~2*SB bytes of output, walked here on the host over dst[] (only
 * streams from a power-of-two -s ever come this way).  Offsets are in working-buffer coordinates (pre + dst[k]); the pass
 * in progress when the range begins started at `start` (<= pre). */
void dec_pass_walk(const uint32_t *hdst, uint32_t ntok, const lz77x_geom &g, uint32_t pre, uint32_t start, bool first_pass,
                   std::vector<uint32_t> &cyc)
{
    const uint64_t W = 3 * (uint64_t)g.sb + (uint64_t)g.la;
    const uint32_t lmax = (1u << g.lb) - 1u;
    const uint64_t safe = W - 1 > lmax ? W - 1 - lmax : 0;
    cyc.clear();
    cyc.push_back(start);
    bool firstp = first_pass;
    uint32_t ks = 0;
    for (;;) {
        const uint64_t back0 = firstp ? 0 : (uint64_t)g.sb, J = cyc.back();
        /* first token k >= ks with back0 + (pre + dst[k] - J) + len_k > W - 1 */
        uint32_t lo = ks, hi = ntok;                              /* tokens below lo certainly fit */
        while (lo < hi) {                                         /* first k whose start is past the always-safe zone */
            const uint32_t mid = lo + (hi - lo) / 2;
            if (back0 + ((uint64_t)pre + hdst[mid] - J) <= safe) lo = mid + 1; else hi = mid;
        }
        uint32_t k = lo > ks ? lo - 1 : ks;
        for (; k < ntok; k++) {
            const uint64_t len = (uint64_t)hdst[k + 1] - hdst[k] - 1;
            if (back0 + ((uint64_t)pre + hdst[k] - J) + len > W - 1) break;
        }
        if (k >= ntok) break;
        if ((uint64_t)pre + hdst[k] == J && !firstp) break;      /* a token longer than the buffer opens the pass: malformed */
        cyc.push_back(pre + hdst[k]);
        ks = k;
        firstp = false;
    }
    cyc.push_back(pre + hdst[ntok]);
}

/* The copy resolution of one range: tokens c.tokval / c.dst [0, ntok) -> n bytes at *d_bytes (inside c.out), complete in
 * stream order on s.  K == null: the range is the whole stream. */
int decode_resolve(Ctx &c, const lz77x_geom &g, uint32_t ntok, uint32_t n, hipStream_t s, bool stale, bool general, DecCarry *K,
                   uint8_t **d_bytes, uint32_t *rounds_out, DevBuf &outb, const uint8_t *d_z_fused = nullptr, const uint32_t *d_bofs = nullptr)
{
    int rc;
    uint32_t rounds = 0;
    const char *dv = LZ77X_VENV("LZ77X_DECODE_VARIANT");            /* 0 production, 1 tile pass + jumping, (LZ77X_DECODE_V1: round 1) */
    const bool track = K && K->track;
    const bool use_seg = !stale && !general && !track && lz77k_dec_seg_supported(g) && !(dv && atoi(dv)) && !LZ77X_VENV("LZ77X_DECODE_V1");
    const uint32_t pre = K && !use_seg ? K->pre : 0u;
    const uint32_t N = pre + n;
    if ((rc = outb.need((size_t)N + 16))) return rc;
    if ((rc = c.ptr.need(use_seg ? ((size_t)N + 8) * 2 : ((size_t)N + 8) * 4))) return rc;
    if ((rc = c.flag.need(64))) return rc;
    if ((rc = c.h_small.need(128))) return rc;
    uint8_t *hdr = c.h_small.as<uint8_t>();
    uint8_t *X = outb.as<uint8_t>();
    *d_bytes = X + pre;
    lz77k_dec_stale Q;
    std::vector<uint32_t> cyc;
    if (pre) {
        /* history in front of the output: zeros (what a copy from before the first byte reads), the image, the carry */
        HIPCHK(hipMemsetAsync(X, 0, pre, s));
        HIPCHK(hipMemcpyAsync(X + pre - K->cb, K->d_carry[K->cur], K->cb, hipMemcpyDeviceToDevice, s));
        if (track) {
            Q.img = pre - K->cb - (K->W - (uint32_t)g.sb);
            HIPCHK(hipMemcpyAsync(X + Q.img, K->d_img[K->cur], K->W - (uint32_t)g.sb, hipMemcpyDeviceToDevice, s));
        }
    }
    if (stale || track) {
        std::vector<uint32_t> hdst((size_t)ntok + 1);
        HIPCHK(hipMemcpyAsync(hdst.data(), c.dst.p, ((size_t)ntok + 1) * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const uint32_t start = K ? (uint32_t)((uint64_t)pre - (K->produced - K->pass_start)) : 0u;
        Q.first0 = K ? (K->first_pass ? 1u : 0u) : 1u;
        dec_pass_walk(hdst.data(), ntok, g, pre, start, Q.first0 != 0, cyc);
        Q.ncyc = (uint32_t)cyc.size() - 1;
        if ((rc = c.scratch.need(cyc.size() * 4 + 64))) return rc;
        HIPCHK(hipMemcpyAsync(c.scratch.p, cyc.data(), cyc.size() * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));                                     /* cyc is pageable: the copy has left it */
        Q.cyc = c.scratch.as<uint32_t>();
    }
    if (use_seg) {
        /* a workgroup per segment of the output, the roots of the last sb bytes in an LDS ring (k_dec_seg); the sb bytes
         * before the range are symbolic references like those before any segment, resolved from the carry */
        const bool ext0 = K && K->produced > 0;
        if ((rc = c.tstart.need(lz77k_dec_seg_tmp_bytes(n, g)))) return rc;
        lz77k_dec_seg_state P;
        /* (d_z_fused: the walk reads the stream itself; c.tokval / c.dst were not filled) */
        HIPCHK(lz77k_dec_segments_front(c.tokval.as<uint32_t>(), c.dst.as<uint32_t>(), ntok, g, X, c.ptr.p, n, c.tstart.p, s, ext0, P, nullptr,
                                        d_z_fused, d_bofs));
        if (ext0 && P.tres0)
            HIPCHK(hipMemcpyAsync(P.tres0, K->d_carry[K->cur] + (K->cb - (uint32_t)g.sb), (size_t)g.sb, hipMemcpyDeviceToDevice, s));
        HIPCHK(lz77k_dec_segments_back(g, X, c.ptr.p, n, P, s));
    } else {
        /* work lists of the pointer-jumping passes (encode's ps/cells buffers are idle during a decode) */
        if ((rc = c.ps.need(((size_t)N + 8) * 4))) return rc;
        if ((rc = c.cells.need(((size_t)N + 8) * 4))) return rc;
        uint32_t *lists[2] = {c.ps.as<uint32_t>(), c.cells.as<uint32_t>()};
        uint32_t *hcount = reinterpret_cast<uint32_t *>(hdr + 32);
        uint32_t total = N;
        const uint32_t *in_list = nullptr;
        if (general || LZ77X_VENV("LZ77X_DECODE_V1")) {
            /* a pointer per output byte in HBM, jumped there: streams that no run of the reference's encoder produces
             * (distances beyond the window, la > 255) -- it assumes nothing about either -- and round 1's cross-check */
            HIPCHK(lz77k_dec_expand(c.tokval.as<uint32_t>(), c.dst.as<uint32_t>(), ntok, g, X, c.ptr.as<uint32_t>(), N, s, Q, pre));
            for (;;) {
                HIPCHK(hipMemsetAsync(c.flag.p, 0, 4, s));
                HIPCHK(lz77k_dec_jump(c.ptr.as<uint32_t>(), total, in_list, lists[rounds & 1], c.flag.as<uint32_t>(), s));
                HIPCHK(hipMemcpyAsync(hcount, c.flag.p, 4, hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                in_list = lists[rounds & 1];
                total = *hcount;
                rounds += 1;
                if (!total || rounds > 80) break;
            }
            HIPCHK(lz77k_dec_gather(X, c.ptr.as<uint32_t>(), N, s));
        } else {
            /* tiles resolve in LDS what stays inside them; only the pointers that leave a tile are jumped in HBM */
            if ((rc = c.tstart.need(lz77k_dec_tile_tmp_bytes(N)))) return rc;
            const unsigned long long *d_unres = nullptr;
            HIPCHK(lz77k_dec_tiles(c.tokval.as<uint32_t>(), c.dst.as<uint32_t>(), ntok, g, X, c.ptr.as<uint32_t>(), N, c.tstart.p, &d_unres, s, Q, pre));
            for (;;) {
                HIPCHK(hipMemsetAsync(c.flag.p, 0, 4, s));
                HIPCHK(lz77k_dec_jump2(c.ptr.as<uint32_t>(), d_unres, total, in_list, lists[rounds & 1], c.flag.as<uint32_t>(), s));
                HIPCHK(hipMemcpyAsync(hcount, c.flag.p, 4, hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                in_list = lists[rounds & 1];
                total = *hcount;
                rounds += 1;
                if (!total || rounds > 80) break;
            }
            HIPCHK(lz77k_dec_gather2(X, c.ptr.as<uint32_t>(), d_unres, N, s));
        }
    }
    if (K) {
        /* what the next range starts from */
        HIPCHK(lz77k_dec_carry(K->d_carry[K->cur], X + pre, n, K->cb, K->d_carry[K->cur ^ 1], s));
        if (track) {
            HIPCHK(lz77k_dec_image(X, Q, (uint32_t)g.sb, K->W, pre, K->d_img[K->cur], K->d_img[K->cur ^ 1], s));
            if (Q.ncyc > 1) K->first_pass = false;
            K->pass_start = K->produced + (uint64_t)cyc[Q.ncyc - 1] - pre;      /* (cyc[0] < pre: wraps back to the carried start) */
        }
        K->cur ^= 1;
        K->produced += n;
    }
    *rounds_out += rounds;
    return LZ77X_OK;
}

/* knobs of the range decoder: tokens per range (a multiple of eight) and bytes of output per range */
void dec_range_plan(const lz77x_geom &g, size_t avail, uint32_t *range_tokens, uint32_t *range_bytes, size_t *planned = nullptr)
{
    const char *e = getenv("LZ77X_DECODE_RANGE");
    uint64_t R = e && atoll(e) > 0 ? (uint64_t)atoll(e) : (uint64_t)1 << 26;
    e = getenv("LZ77X_DECODE_RANGE_BYTES");
    uint64_t cap = e && atoll(e) > 0 ? (uint64_t)atoll(e) : (uint64_t)1 << 30;
    const uint64_t rmax = (uint64_t)0xFF000000u >> g.lb;              /* 32-bit offsets inside a range, whatever its tokens hold */
    if (R > rmax) R = rmax;
    R &= ~(uint64_t)7;
    if (R < 8) R = 8;
    if (cap < ((uint64_t)8 << g.lb)) cap = (uint64_t)8 << g.lb;       /* eight tokens always fit */
    if (cap > 0xFF000000u) cap = 0xFF000000u;
    /* a device with less to spare (device_budget) gets smaller ranges: per token two stream buffers + token words, lengths
     * and offsets; per output byte the byte itself + a 16-bit reference (segment walk) or a pointer and two work-list
     * entries (tile pass / per-byte pointers) */
    /* (a power-of-two window means distance-0 copies: decode_resolve follows the reference's staging buffer and takes the
     * pointer paths, whatever the segment walk supports) */
    const bool seg_walk = lz77k_dec_seg_supported(g) && (g.sb & (g.sb - 1)) != 0;
    const double per_tok = 2.0 * g.T / 8.0 + 12.5, per_byte = seg_walk ? 4.3 : 14.3;     /* (two output buffers: RangeDrain) */
    const double need = 1.125 * (per_tok * (double)R + per_byte * (double)cap) + 64e6;
    if (avail && need > 0.9 * (double)avail) {
        const double f = 0.9 * (double)avail / need;
        R = (uint64_t)((double)R * f) & ~(uint64_t)7;
        cap = (uint64_t)((double)cap * f);
        if (R < 8) R = 8;
        if (cap < ((uint64_t)8 << g.lb)) cap = (uint64_t)8 << g.lb;
    }
    *range_tokens = (uint32_t)R;
    *range_bytes = (uint32_t)cap;
    if (planned) *planned = (size_t)(1.125 * (per_tok * (double)R + per_byte * (double)cap) + 64e6);
}

/* The decoder: stream from `src` (its first four bytes are the header, lz77.c:157-158), bytes to `sink` (null: only the
 * decoded size is wanted).  s: the stream every kernel is enqueued on. */
int decode_stream(Ctx &c, Source &src, Sink *sink, hipStream_t s, uint64_t *n_out)
{
    const double t_begin = now_ms();
    memset(&g_stats, 0, sizeof g_stats);
    int rc;
    if ((rc = c.h_small.need(128))) return rc;
    if ((rc = c.z.need(64))) return rc;
    if ((rc = c.flag.need(64))) return rc;
    uint8_t *hdr = c.h_small.as<uint8_t>();
    size_t got = 0;
    if ((rc = src.read(c, c.z.as<uint8_t>(), 4, s, &got))) return rc;
    if (got < 4) return LZ77X_E_FORMAT;
    HIPCHK(small_d2h(c.h_small, hdr, c.z.p, 4, s));
    HIPCHK(hipStreamSynchronize(s));
    const int sb = hdr[0] | (hdr[1] << 8), la = hdr[2] | (hdr[3] << 8);       /* lz77.c:157-158 */
    if (sb < 1 || la < 1) return LZ77X_E_FORMAT;
    lz77x_geom g;
    lz77x_make_geom(&g, sb, la);
    /* the header is 16 bits of la, but main.c:103 never lets la past 255: a token wider than 32 bits
     * cannot come from the reference's encoder, and the kernels carry tokens in 32-bit words */
    if (g.T > 32) return LZ77X_E_FORMAT;
    uint32_t R = 0, cap = 0;
    size_t avail = 0;
    if ((rc = device_budget(c, &avail))) return rc;
    size_t planned = 0;
    dec_range_plan(g, avail, &R, &cap, &planned);
    /* a stream whose size is known and lies inside one range is sized by what it holds: the range shrinks to the stream plus
     * one token (reading then meets the end of the stream inside it) */
    {
        const size_t hint = src.size_hint();
        if (hint && hint / (size_t)g.T + 2 < (size_t)R / 8) {
            const uint32_t R0 = R;
            R = (uint32_t)((hint / (size_t)g.T + 2) * 8);
            planned = (size_t)((double)planned * (double)R / (double)R0) + ((size_t)64 << 20);      /* (tokens and bytes shrink together) */
        }
    }
    budget_commit(c, planned);
    const size_t rbytes = (size_t)R / 8 * (size_t)g.T;                        /* R tokens are exactly this many bytes */
    DevBuf *zb[2] = {&c.z, &c.z2};
    DevBuf *outb[2] = {&c.out, &c.out2};
    RangeDrain drain;                                      /* (joined on every way out of this function) */
    bool use_drain = false;
    const bool pipelined = !(getenv("LZ77X_PIPELINE") && atoi(getenv("LZ77X_PIPELINE")) == 0);
    DecCarry K;
    bool have_k = false;
    /* (the conditions of decode_resolve's segment walk that are known before the range is looked at) */
    const bool fused_ok = lz77k_dec_seg_supported(g) && (sb & (sb - 1)) != 0 && la <= 255 && !LZ77X_VENV("LZ77X_DECODE_VARIANT") &&
                          !LZ77X_VENV("LZ77X_DECODE_V1") && !LZ77X_VENV("LZ77X_DECODE_UNFUSED");
    uint64_t total_out = 0, total_tok = 0, zn = 4;
    size_t L = 0;                                                             /* bytes of zb[cur] already there (read past the last cut) */
    int cur = 0;
    bool eof = false;
    uint32_t rounds = 0, range_idx = 0;
    float k_ms = 0;
    for (;;) {
        DevBuf &zc = *zb[cur];
        if ((rc = zc.need(4 + rbytes + 32))) return rc;
        size_t avail = L;
        if (!eof) {
            const size_t want = rbytes - L;
            got = 0;
            if ((rc = src.read(c, zc.as<uint8_t>() + 4 + L, want, s, &got))) return rc;
            if (got < want) eof = true;
            avail += got;
            zn += got;
        }
        DevBuf &zr = *zb[cur];
        const uint64_t ntok64 = eof ? (uint64_t)avail * 8 / (uint64_t)g.T : (uint64_t)R;   /* lz77.c:271: short read = EOF */
        if (ntok64 == 0) break;
        uint32_t ntok = (uint32_t)ntok64;
        HIPCHK(hipMemsetAsync(zr.as<uint8_t>() + 4 + avail, 0, 32, s));
        HIPCHK(hipEventRecord(c.ev[0], s));
        uint32_t *tot = reinterpret_cast<uint32_t *>(hdr + 16);
        /* The segment walk reads the stream itself (k_dec_seg<true>): all it needs in front of it are the sums of len + 1 per
         * block of tokens and their scan.  A range that turns out to need anything else -- a distance-0 copy, a distance
         * beyond the window, more bytes than the range's cap -- goes through the token words, lengths and offsets of the
         * full parse + scan below (those streams pay the sums twice). */
        bool fused = fused_ok && !(have_k && K.track);
        const uint32_t *d_bofs = nullptr;
        uint32_t n = 0, use = ntok;
        bool stale = false, general = la > 255;
        if (fused) {
            const uint32_t nblk = (ntok + lz77k_dec_sum_block() - 1u) / lz77k_dec_sum_block();
            if ((rc = c.len1.need(((size_t)nblk + 8) * 4))) return rc;
            if ((rc = c.dst.need(((size_t)nblk + 8) * 4))) return rc;
            if ((rc = c.scantmp.need(lz77k_scan_tmp_bytes(nblk + 1)))) return rc;
            HIPCHK(hipMemsetAsync(c.flag.as<uint32_t>() + 8, 0, 8, s));
            HIPCHK(lz77k_dec_sums(zr.as<uint8_t>(), ntok, g, c.len1.as<uint32_t>(), c.flag.as<uint32_t>() + 8, s));
            HIPCHK(lz77k_scan_u32(c.len1.as<uint32_t>(), c.dst.as<uint32_t>(), nblk + 1, c.scantmp.p, s));
            HIPCHK(small_d2h(c.h_small, tot, c.dst.as<uint32_t>() + nblk, 4, s));
            HIPCHK(small_d2h(c.h_small, tot + 1, c.flag.as<uint32_t>() + 8, 8, s));
            HIPCHK(hipStreamSynchronize(s));
            n = tot[0];
            stale = tot[1] != 0;
            general = tot[2] != 0 || la > 255;
            if (stale || general || n > cap) fused = false;
            else d_bofs = c.dst.as<uint32_t>();
        }
        if (!fused) {
            if ((rc = c.tokval.need(((size_t)ntok + 8) * 4))) return rc;
            if ((rc = c.len1.need(((size_t)ntok + 8) * 4))) return rc;
            if ((rc = c.dst.need(((size_t)ntok + 8) * 4))) return rc;
            if ((rc = c.scantmp.need(lz77k_scan_tmp_bytes(ntok + 1)))) return rc;
            HIPCHK(hipMemsetAsync(c.flag.as<uint32_t>() + 8, 0, 8, s));
            HIPCHK(lz77k_dec_parse(zr.as<uint8_t>(), ntok, g, c.tokval.as<uint32_t>(), c.len1.as<uint32_t>(), s, c.flag.as<uint32_t>() + 8));
            HIPCHK(hipMemsetAsync(c.len1.as<uint32_t>() + ntok, 0, 4, s));
            HIPCHK(lz77k_scan_u32(c.len1.as<uint32_t>(), c.dst.as<uint32_t>(), ntok + 1, c.scantmp.p, s));     /* ntok << lb fits 32 bits (dec_range_plan) */
            HIPCHK(small_d2h(c.h_small, tot, c.dst.as<uint32_t>() + ntok, 4, s));
            HIPCHK(small_d2h(c.h_small, tot + 1, c.flag.as<uint32_t>() + 8, 8, s));
            HIPCHK(hipStreamSynchronize(s));
            n = tot[0];
            stale = tot[1] != 0;             /* the range copies from distance 0 somewhere (power-of-two -s) */
            /* distances beyond the window, or a lookahead field no CLI run can produce (main.c:103 caps -l at 255; a token
             * may then span several tiles): the per-byte pointer path, which assumes nothing about either */
            general = tot[2] != 0 || la > 255;
        }
        if (n > cap) {
            HIPCHK(lz77k_dec_cut(c.dst.as<uint32_t>(), ntok, cap, c.flag.as<uint32_t>() + 12, s));
            HIPCHK(small_d2h(c.h_small, tot + 4, c.flag.as<uint32_t>() + 12, 8, s));
            HIPCHK(hipStreamSynchronize(s));
            use = tot[4];
            n = tot[5];
        }
        const bool last = eof && use == ntok;
        if (range_idx == 0 && !last) {
            /* more than one range: the state they hand on.  Distance-0 copies are followed from the first range on when
             * the window is a power of two (the only streams of the reference's encoder that hold them) or the first
             * range holds one. */
            K.cb = (uint32_t)sb;
            if (g.ob && (1u << g.ob) - 1u > K.cb) K.cb = (1u << g.ob) - 1u;
            K.W = 3u * (uint32_t)sb + (uint32_t)la;
            K.track = (sb & (sb - 1)) == 0 || stale;
            const uint32_t hist = K.cb + (K.track ? K.W - (uint32_t)sb : 0u);
            K.pre = (hist + LZ77K_DEC_TILE_BYTES - 1u) / LZ77K_DEC_TILE_BYTES * LZ77K_DEC_TILE_BYTES;
            const size_t img = (size_t)K.W - (size_t)sb;
            if ((rc = c.dcarry.need(2 * ((size_t)K.cb + 256) + 2 * (img + 256)))) return rc;
            uint8_t *b = c.dcarry.as<uint8_t>();
            K.d_carry[0] = b;
            K.d_carry[1] = b + K.cb + 256;
            K.d_img[0] = b + 2 * ((size_t)K.cb + 256);
            K.d_img[1] = K.d_img[0] + img + 256;
            HIPCHK(hipMemsetAsync(b, 0, 2 * ((size_t)K.cb + 256) + 2 * (img + 256), s));
            have_k = true;
        }
        if (have_k && stale && !K.track) {
            /* the reference reads a byte of its staging buffer whose history this decoder did not follow */
            snprintf(g_err, sizeof g_err, "a distance-0 copy appears %llu tokens into a stream whose window is not a power of two",
                     (unsigned long long)total_tok);
            return LZ77X_E_FORMAT;
        }
        if (sink && !sink->overflowed() && n) {
            uint8_t *d_bytes = nullptr;
            /* several ranges into a sink that blocks on the host: two in flight -- this range resolves into the buffer the
             * range before last has left, while the last one's bytes are still on their way out (RangeDrain) */
            const bool async = have_k && pipelined && sink->blocks_on_host();
            if (use_drain && (rc = drain.wait(1))) return rc;
            DevBuf &ob = *outb[async ? (range_idx & 1u) : 0u];
            if ((rc = decode_resolve(c, g, use, n, s, stale, general, have_k ? &K : nullptr, &d_bytes, &rounds, ob,
                                     fused ? zr.as<uint8_t>() : nullptr, d_bofs))) return rc;
            HIPCHK(hipEventRecord(c.ev[1], s));
            if (async) {
                if (!use_drain) {
                    Ctx *dc = nullptr;
                    if ((rc = ctx_sibling(c, &dc))) return rc;
                    if ((rc = drain.start(sink, dc))) return rc;
                    use_drain = true;
                }
                hipEvent_t ready = c.pipe_ev[range_idx & 1u];
                HIPCHK(hipEventRecord(ready, s));
                if ((rc = drain.submit(d_bytes, n, ready))) return rc;
            } else if ((rc = sink->write(c, d_bytes, n, s))) return rc;
        } else {
            HIPCHK(hipEventRecord(c.ev[1], s));
            if (sink) sink->total += n;
        }
        HIPCHK(hipStreamSynchronize(s));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
        k_ms += ms;
        total_out += n;
        total_tok += use;
        range_idx += 1;
        if (last) break;
        const size_t used = (size_t)use / 8 * (size_t)g.T;                     /* use is a multiple of eight here */
        L = avail - used;
        if (L) {
            DevBuf &zo = *zb[cur ^ 1];
            if ((rc = zo.need(4 + rbytes + 32))) return rc;
            HIPCHK(hipMemcpyAsync(zo.as<uint8_t>() + 4, zr.as<uint8_t>() + 4 + used, L, hipMemcpyDeviceToDevice, s));
        }
        cur ^= L ? 1 : 0;
    }
    if (use_drain && (rc = drain.wait(0))) return rc;
    *n_out = total_out;
    g_stats.k_decode_ms = k_ms;
    g_stats.n = total_out;
    g_stats.zn = zn;
    g_stats.ntok = total_tok;
    g_stats.decode_rounds = rounds;
    g_stats.match_launches = range_idx;          /* ranges the stream was decoded in */
    g_stats.total_ms = now_ms() - t_begin;
    g_stats.copy_ms = g_stats.total_ms - g_stats.k_decode_ms;
    return LZ77X_OK;
}

}  // namespace lz77x_host
