/*
 * api.cpp -- the C ABI of include/lz77_mi355x.h: the drop-in boundary (lz77.h:14-15 of the reference).  Every entry point leases a
 * context set, checks the geometry and hands over to one of the pipelines; without a HIP device it returns LZ77X_E_NODEV.
 */
#include "host.h"

using namespace lz77x_host;

extern "C" {

size_t lz77x_encode_bound(size_t n, int sb, int la)
{
    if (check_geom(sb, la) != LZ77X_OK) return 0;
    lz77x_geom g;
    lz77x_make_geom(&g, sb, la);
    return stream_bytes(n, g.T);
}

int lz77x_encode(const uint8_t *in, size_t n, int sb, int la, uint8_t **out, size_t *out_n)
try {
    if (!out || !out_n || (!in && n)) return LZ77X_E_ARG;
    int rc = check_geom(sb, la);
    if (rc) return rc;
    Lease lease;
    Ctx &g_ctx = lease.set->primary;
    (void)g_ctx;
    const double t0 = now_ms();
    std::vector<Ctx *> cs;
    int shards = g_shards;
    if (shards <= 0) { const char *e = getenv("LZ77X_SHARDS"); shards = e ? atoi(e) : 1; }
    if ((rc = shard_contexts(*lease.set, shards < 1 ? 1 : shards, cs))) return rc;
    TRACE("runtime + context init", t0);
    lz77x_geom g;
    make_encode_geom(&g, sb, la);
    const double t1 = now_ms();
    HostSink sink;
    if ((rc = encode_any(cs, in, false, n, g, g_ctx.stream, sink))) return rc;
    TRACE("encode (incl. allocs, result fetch)", t1);
    *out_n = sink.total;
    *out = sink.release();
    return *out ? LZ77X_OK : LZ77X_E_NOMEM;
} LZ77X_API_CATCH

int lz77x_encode_device(const void *d_in, size_t n, int sb, int la, void *d_out, size_t out_cap, size_t *out_n, void *stream)
try {
    if (!out_n || (!d_in && n) || !d_out) return LZ77X_E_ARG;
    int rc = check_geom(sb, la);
    if (rc) return rc;
    Lease lease;
    Ctx &g_ctx = lease.set->primary;
    (void)g_ctx;
    std::vector<Ctx *> cs;
    if ((rc = shard_contexts(*lease.set, 1, cs))) return rc;      /* device-resident buffers: the caller's device only */
    lz77x_geom g;
    make_encode_geom(&g, sb, la);
    hipStream_t s = (hipStream_t)stream;
    DeviceSink sink(d_out, out_cap);
    if ((rc = encode_any(cs, d_in, true, n, g, s, sink))) return rc;
    *out_n = sink.total;
    HIPCHK(hipStreamSynchronize(s));
    return sink.total > out_cap ? LZ77X_E_CAP : LZ77X_OK;
} LZ77X_API_CATCH

int lz77x_decode(const uint8_t *z, size_t zn, uint8_t **out, size_t *out_n)
try {
    if (!out || !out_n || (!z && zn)) return LZ77X_E_ARG;
    Lease lease;
    Ctx &g_ctx = lease.set->primary;
    (void)g_ctx;
    int rc;
    int shards = g_shards;
    if (shards <= 0) { const char *e = getenv("LZ77X_SHARDS"); shards = e ? atoi(e) : 1; }
    if (shards > 1) {
        /* one stream on several devices: token ranges with a chained sb-byte hand-off (decode_sharded) */
        std::vector<Ctx *> cs;
        if ((rc = shard_contexts(*lease.set, shards, cs))) return rc;
        if (cs.size() > 1) {
            int handled = 0;
            if ((rc = decode_sharded(cs, z, zn, out, out_n, &handled))) return rc;
            if (handled) return LZ77X_OK;
        }
    }
    if ((rc = primary_context(*lease.set))) return rc;
    if (zn < 4) return LZ77X_E_FORMAT;
    MemSource src(z, zn, false);
    HostSink sink;
    uint64_t n = 0;
    if ((rc = decode_stream(g_ctx, src, &sink, g_ctx.stream, &n))) return rc;
    *out_n = sink.total;
    *out = sink.release();
    return *out ? LZ77X_OK : LZ77X_E_NOMEM;
} LZ77X_API_CATCH

int lz77x_decode_device(const void *d_z, size_t zn, void *d_out, size_t out_cap, size_t *out_n, void *stream)
try {
    if (!out_n || (!d_z && zn)) return LZ77X_E_ARG;
    Lease lease;
    Ctx &g_ctx = lease.set->primary;
    (void)g_ctx;
    int rc;
    if ((rc = primary_context(*lease.set))) return rc;
    if (zn < 4) return LZ77X_E_FORMAT;
    hipStream_t s = (hipStream_t)stream;
    MemSource src(d_z, zn, true);
    uint64_t n = 0;
    if (!d_out) {
        if ((rc = decode_stream(g_ctx, src, nullptr, s, &n))) return rc;
        *out_n = (size_t)n;
        return LZ77X_OK;
    }
    DeviceSink sink(d_out, out_cap);
    if ((rc = decode_stream(g_ctx, src, &sink, s, &n))) return rc;
    *out_n = (size_t)n;
    HIPCHK(hipStreamSynchronize(s));
    return n > out_cap ? LZ77X_E_CAP : LZ77X_OK;
} LZ77X_API_CATCH

/* lz77.h:14 encode(file, out, la, sb) as called at main.c:150 */
int lz77x_encode_file(FILE *in, FILE *out, int la, int sb)
try {
    if (!in || !out) return LZ77X_E_ARG;
    int rc = check_geom(sb, la);
    if (rc) return rc;
    int shards = g_shards;
    if (shards <= 0) { const char *e = getenv("LZ77X_SHARDS"); shards = e ? atoi(e) : 1; }
    Lease lease;
    Ctx &g_ctx = lease.set->primary;
    (void)g_ctx;
    const double t0 = now_ms();
    std::vector<Ctx *> cs;
    if ((rc = shard_contexts(*lease.set, shards < 1 ? 1 : shards, cs))) return rc;
    TRACE("runtime + context init", t0);
    Ctx &c = g_ctx;
    lz77x_geom g;
    make_encode_geom(&g, sb == -1 ? LZ77X_DEFAULT_SB : sb, la == -1 ? LZ77X_DEFAULT_LA : la);
    size_t n = 0;
    if (cs.size() > 1 && device_pipeline_ok(1, g)) {
        /* one stream over several devices: stretch by stretch out of the file (the host holds one stretch), every stretch cut
         * into position shards */
        FileWindow win(in);
        FileSink sink(out);
        const double t1 = now_ms();
        rc = encode_sharded(cs, win, g, sink);
        TRACE("file -> devices -> file", t1);
        return rc;
    }
    if (cs.size() > 1) cs.resize(1);                            /* (a geometry without the device pipeline: one device) */
    if (device_pipeline_ok(1, g)) {
        /* any size, any kind of file: segment by segment through bounded device memory */
        FileSource src(in);
        FileSink sink(out);
        bool fallback = false;
        const double t1 = now_ms();
        rc = encode_stream_device(c, src, sink, g, c.stream, &fallback, &n);
        TRACE("file -> device -> file", t1);
        trace_allocs("  of which allocations:");
        if (rc || !fallback) return rc;                         /* fallback: the whole (single-segment) input sits in c.in */
        HIPCHK(hipDeviceSynchronize());                         /* (nothing of the attempt in flight when the other pipeline takes the context over) */
    } else {
        const double t1 = now_ms();
        if ((rc = stream_in(c, in, c.in, LZ77X_PAD + 16, &n))) return rc;
        TRACE("file -> device", t1);
    }
    size_t zn = 0;
    const double t2 = now_ms();
    if ((rc = encode_core_host(cs, c.in.p, true, n, g, c.stream, &zn))) return rc;
    TRACE("encode_core_host (incl. allocs)", t2);
    const double t3 = now_ms();
    rc = stream_out(c, out, c.out.p, zn);
    TRACE("device -> file", t3);
    return rc;
} LZ77X_API_CATCH

/* lz77.h:15 decode(file, out) as called at main.c:161 */
int lz77x_decode_file(FILE *in, FILE *out)
try {
    if (!in || !out) return LZ77X_E_ARG;
    Lease lease;
    Ctx &g_ctx = lease.set->primary;
    (void)g_ctx;
    int rc;
    const double t0 = now_ms();
    if ((rc = primary_context(*lease.set))) return rc;
    TRACE("runtime + context init", t0);
    /* any size, any kind of file: range by range through bounded device memory (lz77.c:160-195) */
    FileSource src(in);
    FileSink sink(out);
    uint64_t n = 0;
    const double t1 = now_ms();
    rc = decode_stream(g_ctx, src, &sink, g_ctx.stream, &n);
    TRACE("file -> device -> file (decode)", t1);
    trace_allocs("  of which allocations:");
    return rc;
} LZ77X_API_CATCH

void lz77x_free(void *p) { free(p); }

int lz77x_set_shards(int shards)
{
    if (shards < 1 || shards > 64) return LZ77X_E_ARG;
    g_shards = shards;
    return LZ77X_OK;
}

int lz77x_device_count(void)
{
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess) return 0;
    return nd;
}

const char *lz77x_strerror(int code)
{
    switch (code) {
    case LZ77X_OK: return "ok";
    case LZ77X_E_ARG: return "bad argument";
    case LZ77X_E_NOMEM: return "out of host memory";
    case LZ77X_E_HIP: return "HIP runtime error";
    case LZ77X_E_NODEV: return "no MI355X/HIP device available (there is no CPU fallback)";
    case LZ77X_E_FORMAT: return "not an lz77 stream";
    case LZ77X_E_CAP: return "output buffer too small";
    case LZ77X_E_IO: return "I/O error";
    case LZ77X_E_TOOBIG: return "input too large for one call";
    default: return "unknown error";
    }
}

void lz77x_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_pool.size();) {
        CtxSet *s = g_pool[i];
        if (s->busy) { i++; continue; }                      /* another thread is inside the library with it */
        for (Ctx *c : s->more) { ctx_release(*c); delete c; }
        ctx_release(s->primary);
        delete s;
        g_pool.erase(g_pool.begin() + (long)i);
    }
}

const char *lz77x_last_error(void) { return g_err; }
const char *lz77x_version(void) { return "lz77-mi355x 0.1 (gfx950)"; }

int lz77x_last_stats(lz77x_stats *st)
{
    if (!st) return LZ77X_E_ARG;
    *st = g_stats;
    return LZ77X_OK;
}

/* ---- stage-level entry points ---- */

static int run_match_only(CtxSet &S, const uint8_t *in, size_t n, int sb, int la, lz77x_geom *g)
{
    int rc = check_geom(sb, la);
    if (rc) return rc;
    if ((rc = primary_context(S))) return rc;
    if (n > LZ77X_MAX_N) return LZ77X_E_TOOBIG;
    make_encode_geom(g, sb, la);
    Ctx &c = S.primary;
    hipStream_t s = c.stream;
    if ((rc = c.in.need(n + LZ77X_PAD + 16))) return rc;
    if ((rc = c.ps.need((n + 8) * 4))) return rc;
    if ((rc = c.maxlen.need(n + 8))) return rc;
    if (n) HIPCHK(hipMemcpyAsync(c.in.p, in, n, hipMemcpyHostToDevice, s));
    HIPCHK(lz77k_fill_pad(c.in.as<uint8_t>(), (uint32_t)n, s));
    HIPCHK(hipMemsetAsync(c.ps.p, 0, (n + 8) * 4, s));
    const uint32_t nregions = (uint32_t)((n + g->TILE - 1) / g->TILE);
    uint32_t batch = nregions;
    if (nregions) {
        const size_t per = lz77k_match_scratch_bytes(*g, 1);
        batch = (uint32_t)(((size_t)2 << 30) / per);
        if (batch < 1) batch = 1;
        if (batch > nregions) batch = nregions;
        if ((rc = c.scratch.need(lz77k_match_scratch_bytes(*g, batch)))) return rc;
    }
    const char *vs = LZ77X_VENV("LZ77X_MATCH_VARIANT");
    const int variant = vs ? atoi(vs) : 0;
    for (uint32_t r0 = 0; r0 < nregions; r0 += batch) {
        const uint32_t nr = nregions - r0 < batch ? nregions - r0 : batch;
        HIPCHK(lz77k_match(c.in.as<uint8_t>(), (uint32_t)n, *g, r0, nr, c.ps.as<uint32_t>(), c.maxlen.as<uint8_t>(),
                           c.scratch.p, variant, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    return LZ77X_OK;
}

int lz77x_stage_maxlen(const uint8_t *in, size_t n, int sb, int la, uint8_t *maxlen)
try {
    if ((!in || !maxlen) && n) return LZ77X_E_ARG;
    Lease lease;
    Ctx &g_ctx = lease.set->primary;
    (void)g_ctx;
    lz77x_geom g;
    int rc = run_match_only(*lease.set, in, n, sb, la, &g);
    if (rc) return rc;
    if (n) HIPCHK(hipMemcpy(maxlen, g_ctx.maxlen.p, n, hipMemcpyDeviceToHost));
    return LZ77X_OK;
} LZ77X_API_CATCH

int lz77x_stage_neighbours(const uint8_t *in, size_t n, int sb, int la, uint16_t *P, uint16_t *S)
try {
    if ((!in || !P || !S) && n) return LZ77X_E_ARG;
    Lease lease;
    Ctx &g_ctx = lease.set->primary;
    (void)g_ctx;
    lz77x_geom g;
    int rc = run_match_only(*lease.set, in, n, sb, la, &g);
    if (rc) return rc;
    if (!n) return LZ77X_OK;
    uint32_t *tmp = (uint32_t *)malloc(n * 4);
    if (!tmp) return LZ77X_E_NOMEM;
    hipError_t e = hipMemcpy(tmp, g_ctx.ps.p, n * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(tmp); HIPCHK(e); }
    for (size_t i = 0; i < n; i++) { P[i] = (uint16_t)(tmp[i] & 0xFFFF); S[i] = (uint16_t)(tmp[i] >> 16); }
    free(tmp);
    return LZ77X_OK;
} LZ77X_API_CATCH

int lz77x_stage_priorities(const uint16_t *P, const uint16_t *S, size_t n, int sb, uint32_t *xval)
try {
    if ((!P || !S || !xval) && n) return LZ77X_E_ARG;
    if (sb < 1 || sb > 65535) return LZ77X_E_ARG;
    uint32_t *ps = (uint32_t *)malloc((n + 1) * 4);
    if (!ps) return LZ77X_E_NOMEM;
    const uint32_t rmask = lz77x_prio_mask(sb);
    for (size_t i = 0; i < n; i++) {
        ps[i] = (((uint32_t)i + P[i]) & rmask) | ((((uint32_t)i + S[i]) & rmask) << 16);
        xval[i] = LZ77X_NONE32;
    }
    lz77x_prio_state st;
    if (!lz77x_prio_init(&st, sb)) { free(ps); return LZ77X_E_NOMEM; }
    lz77x_prio_run(&st, ps, sb, n, xval);
    lz77x_prio_free(&st);
    free(ps);
    return LZ77X_OK;
} LZ77X_API_CATCH

int lz77x_stage_priorities_device(const uint16_t *P, const uint16_t *S, size_t n, int sb, uint32_t *xval, int *iters_out)
try {
    if ((!P || !S || !xval) && n) return LZ77X_E_ARG;
    if (!lz77k_prio_supported(sb) || n > LZ77X_MAX_N) return LZ77X_E_ARG;
    Lease lease;
    int rc;
    if ((rc = primary_context(*lease.set))) return rc;
    Ctx &c = lease.set->primary;
    if (iters_out) *iters_out = 0;
    if (!n) return LZ77X_OK;
    uint32_t *ps = (uint32_t *)malloc(n * 4);
    if (!ps) return LZ77X_E_NOMEM;
    for (size_t i = 0; i < n; i++) ps[i] = (uint32_t)P[i] | ((uint32_t)S[i] << 16);
    rc = LZ77X_OK;
    int iters = 0, converged = 1;
    do {
        if ((rc = c.ps.need((n + 8) * 4))) break;
        if ((rc = c.xval.need((n + 8) * 4))) break;
        if ((rc = c.prio_tmp.need(lz77k_prio_tmp_bytes((uint32_t)n, sb)))) break;
        if ((rc = c.h_small.need(64))) break;
        hipError_t e = hipMemcpyAsync(c.ps.p, ps, n * 4, hipMemcpyHostToDevice, c.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c.stream);              /* pageable source */
        const char *me = getenv("LZ77X_PRIO_MAX_ITERS");
        /* like lz77x_stage_priorities: only x < n - sb is ever evicted (lz77.c:101-103), the rest stays NONE */
        const size_t nx = n > (size_t)sb ? n - (size_t)sb : 0;
        for (size_t i = nx; i < n; i++) xval[i] = LZ77X_NONE32;
        if (e == hipSuccess)
            e = lz77k_prio(c.ps.as<uint32_t>(), (uint32_t)nx, sb, c.xval.as<uint32_t>(), c.prio_tmp.p, c.stream, c.h_small.as<uint32_t>() + 8,
                           me && atoi(me) > 0 ? atoi(me) : 1 << 20, &iters, &converged);
        if (e == hipSuccess) e = hipStreamSynchronize(c.stream);              /* the closing sweep is only enqueued */
        if (e == hipSuccess && nx) e = hipMemcpy(xval, c.xval.p, nx * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            snprintf(g_err, sizeof g_err, "stage_priorities_device: %s", hipGetErrorString(e));
            rc = LZ77X_E_HIP;
        }
    } while (0);
    free(ps);
    if (iters_out) *iters_out = converged ? iters : -iters;
    return rc;
} LZ77X_API_CATCH

int lz77x_stage_chain_device(const uint8_t *maxlen, size_t n, int la, uint32_t *chain, size_t *ntok)
try {
    if ((!maxlen || !chain) && n) return LZ77X_E_ARG;
    if (!ntok || la < 2 || la > 255 || n > LZ77X_MAX_N) return LZ77X_E_ARG;
    Lease lease;
    int rc;
    if ((rc = primary_context(*lease.set))) return rc;
    Ctx &c = lease.set->primary;
    *ntok = 0;
    if (!n) return LZ77X_OK;
    if ((rc = c.maxlen.need(n + 64))) return rc;
    if ((rc = c.chain.need((n + 8) * 4))) return rc;
    if ((rc = c.chain_tmp.need(lz77k_chain_tmp_bytes((uint32_t)n, la)))) return rc;
    HIPCHK(hipMemcpy(c.maxlen.p, maxlen, n, hipMemcpyHostToDevice));
    const uint32_t *d_tbase = nullptr;
    uint32_t nsub = 0, total = 0;
    HIPCHK(lz77k_chain(c.maxlen.as<uint8_t>(), (uint32_t)n, la, c.chain.as<uint32_t>(), c.chain_tmp.p, c.stream, &d_tbase, &nsub));
    HIPCHK(hipStreamSynchronize(c.stream));
    HIPCHK(hipMemcpy(&total, d_tbase + nsub, 4, hipMemcpyDeviceToHost));
    if (total) HIPCHK(hipMemcpy(chain, c.chain.p, (size_t)total * 4, hipMemcpyDeviceToHost));
    *ntok = total;
    return LZ77X_OK;
} LZ77X_API_CATCH

/* several files at once: a thread per file in flight, each leasing its own device context (the lease blocks
 * further threads until a context is free), which is what overlaps the kernels of one file with the host
 * work and the transfers of the others */
static int run_files(int n_files, FILE **in, FILE **out, int la, int sb, int *rcs, bool enc)
{
    if (n_files < 0 || (n_files && (!in || !out))) return LZ77X_E_ARG;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) cur = -1;
    std::vector<int> rc((size_t)n_files, LZ77X_OK);
    std::atomic<int> next{0};
    auto work = [&]() {
        if (cur >= 0) { hipError_t e = hipSetDevice(cur); (void)e; }
        for (int i = next.fetch_add(1); i < n_files; i = next.fetch_add(1))
            rc[(size_t)i] = enc ? lz77x_encode_file(in[i], out[i], la, sb) : lz77x_decode_file(in[i], out[i]);
    };
    int lanes = 4;
    { const char *e = getenv("LZ77X_MAX_CONTEXTS"); if (e && atoi(e) > 0) lanes = atoi(e); }
    if (lanes > n_files) lanes = n_files;
    std::vector<std::thread> th;
    try {
        th.reserve((size_t)(lanes > 1 ? lanes - 1 : 0));
        for (int t = 1; t < lanes; t++) th.emplace_back(work);
    } catch (...) { /* (a lane that could not start: the others take its files; a joinable std::thread must not be destroyed) */ }
    if (n_files) work();
    for (auto &t : th) t.join();
    int first = LZ77X_OK;
    for (int i = 0; i < n_files; i++) {
        if (rcs) rcs[i] = rc[(size_t)i];
        if (first == LZ77X_OK && rc[(size_t)i] != LZ77X_OK) first = rc[(size_t)i];
    }
    return first;
}

int lz77x_encode_files(int n_files, FILE **in, FILE **out, int la, int sb, int *rc)
try { return run_files(n_files, in, out, la, sb, rc, true); } LZ77X_API_CATCH
int lz77x_decode_files(int n_files, FILE **in, FILE **out, int *rc)
try { return run_files(n_files, in, out, 0, 0, rc, false); } LZ77X_API_CATCH

}  // extern "C"
