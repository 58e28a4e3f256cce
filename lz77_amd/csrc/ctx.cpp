/*
 * ctx.cpp -- device contexts, the pool of context sets concurrent callers lease, the device-memory budget; the one definition of what host.h
 * declares extern.
 */
#include "host.h"

LZ77X_HOST_NS {

__thread char g_err[256] = "";
__thread lz77x_stats g_stats;
int g_shards = 0;
__thread double g_alloc_ms = 0, g_pin_ms = 0, g_fread_ms = 0, g_fwrite_ms = 0;
__thread size_t g_alloc_bytes = 0, g_pin_bytes = 0;
CopyPool g_copy;                                           /* towards the device: copies out of pageable memory, file reads */
CopyPool g_copy_out;                                       /* away from it: copies into pageable memory, file writes */
std::vector<CtxSet *> g_pool;
std::mutex g_mu;
std::condition_variable g_cv;
__thread CtxSet *tl_set = nullptr;

/* LZ77X_TRACE=1: phase timestamps on stderr (opt-in; the default run prints nothing) */
bool trace_on()
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("LZ77X_TRACE"); on = e && atoi(e) ? 1 : 0; }
    return on == 1;
}

bool poison_on()
{
    static const bool on = [] { const char *e = getenv("LZ77X_POISON"); return e && atoi(e); }();      /* (read once, by whichever thread comes first) */
    return on;
}

/* LZ77X_POISON=1: what the set's contexts cached from earlier calls holds 0xA5 when the next call starts */
static void poison_ctx(Ctx &c)
{
    if (c.pipe) poison_ctx(*c.pipe);
    if (c.drain) poison_ctx(*c.drain);
    if (!c.ready) return;
    hipError_t e = hipSetDevice(c.device);
    /* LZ77X_POISON_MASK: bit i = the i-th buffer of dev_bufs(), bit 32 + i = the i-th of pin_bufs() (to find WHICH stale
     * buffer a failure depends on); default: all */
    static const unsigned long long mask = [] { const char *m = getenv("LZ77X_POISON_MASK"); return m ? strtoull(m, nullptr, 0) : ~0ull; }();
    int i = 0;
    for (DevBuf *b : c.dev_bufs()) {
        if (b->p && ((mask >> i) & 1ull)) e = hipMemset(b->p, 0xA5, b->cap);
        i++;
    }
    e = hipDeviceSynchronize();
    (void)e;
    i = 32;
    for (PinBuf *b : c.pin_bufs()) {
        if (b->p && ((mask >> i) & 1ull)) memset(b->p, 0xA5, b->cap);
        i++;
    }
}

bool poison_fresh(const DevBuf *b)
{
    static const unsigned long long mask = [] { const char *m = getenv("LZ77X_POISON_FRESH_MASK"); return m ? strtoull(m, nullptr, 0) : ~0ull; }();
    if (mask == ~0ull || !tl_set) return mask != 0;
    std::vector<Ctx *> all = {&tl_set->primary};
    for (Ctx *c : tl_set->more) all.push_back(c);
    for (size_t k = 0; k < all.size(); k++) {
        Ctx *c = all[k];
        if (c->pipe) all.push_back(c->pipe);
        if (c->drain) all.push_back(c->drain);
        int i = 0;
        for (DevBuf *q : c->dev_bufs()) { if (q == b) return (mask >> i) & 1ull; i++; }
    }
    return true;
}

void poison_set(CtxSet &S)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) cur = -1;
    poison_ctx(S.primary);
    for (Ctx *c : S.more) poison_ctx(*c);
    if (cur >= 0) { hipError_t e = hipSetDevice(cur); (void)e; }
}

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

void trace_allocs(const char *what)
{
    if (!trace_on()) return;
    fprintf(stderr, "[lz77x] %-28s hipMalloc %.2f ms (%.1f MB), pinned host %.2f ms (%.1f MB); in file reads %.2f ms, in file writes %.2f ms\n", what,
            g_alloc_ms, g_alloc_bytes / 1e6, g_pin_ms, g_pin_bytes / 1e6, g_fread_ms, g_fwrite_ms);
    g_alloc_ms = g_pin_ms = g_fread_ms = g_fwrite_ms = 0;
    g_alloc_bytes = g_pin_bytes = 0;
}

/* Device memory one call may plan with: what is free now, shared with the other callers inside the library at this
 * moment (each leases a context set of its own), plus what this call's context already holds in its cached buffers.
 * LZ77X_DEVICE_MEM_LIMIT (bytes) caps it -- a test knob, and a way to keep the library's footprint below a share of the
 * device.  The plans below (segment size of an encode, range size of a decode) size themselves to fit; they never
 * change the output bytes. */
int device_budget(Ctx &c, size_t *avail)
{
    size_t fr = 0, total = 0;
    HIPCHK(hipMemGetInfo(&fr, &total));
    size_t held = 0;
    for (DevBuf *b : c.dev_bufs()) held += b->cap;
    if (c.pipe) for (DevBuf *b : c.pipe->dev_bufs()) held += b->cap;
    /* what the other callers inside the library were promised and have not allocated yet is not free: a first caller that
     * sees busy == 1 plans with nearly everything, and without the reservation a second one would plan with the same bytes
     * (a 288 GB device hides it, a shared or smaller one ends in hipErrorOutOfMemory).  A caller that has not planned yet counts
     * for an equal share. */
    size_t busy = 0, waiting = 0, reserved = 0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (CtxSet *q : g_pool) {
            if (!q->busy) continue;
            busy++;
            if (q == tl_set) continue;
            if (q->promised) reserved += q->promised;
            else waiting++;                            /* inside the library, not planned yet: an equal share */
        }
    }
    (void)busy;
    const size_t fr_eff = fr > reserved ? fr - reserved : 0;
    size_t a = fr_eff / (waiting + 1) + held;
    const char *e = getenv("LZ77X_DEVICE_MEM_LIMIT");
    if (e && atoll(e) > 0 && (size_t)atoll(e) < a) a = (size_t)atoll(e);
    *avail = a;
    return LZ77X_OK;
}

/* the call's plan needs `planned` bytes of device memory in all: what it does not hold yet is reserved against the other
 * callers' budgets until the call returns (Lease) */
void budget_commit(Ctx &c, size_t planned)
{
    size_t held = 0;
    for (DevBuf *b : c.dev_bufs()) held += b->cap;
    if (c.pipe) for (DevBuf *b : c.pipe->dev_bufs()) held += b->cap;
    if (!tl_set) return;
    std::lock_guard<std::mutex> lk(g_mu);
    tl_set->promised = planned > held ? planned - held : 0;
}

int ctx_init(Ctx &c, int device)
{
    if (c.ready) return LZ77X_OK;
    int nd = 0;
    const double t_rt = now_ms();
    hipError_t e = hipGetDeviceCount(&nd);
    TRACE("  hipGetDeviceCount (runtime init)", t_rt);
    if (e != hipSuccess || nd <= 0) {
        snprintf(g_err, sizeof g_err, "no HIP device (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
        return LZ77X_E_NODEV;
    }
    c.ndev = nd;
    if (device < 0) HIPCHK(hipGetDevice(&device));
    c.device = device;
    const double t_dev = now_ms();
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    TRACE("  hipSetDevice + first stream", t_dev);
    /* (copy / up / tok: created by the paths that use them, need_stream -- a stream costs ~8 ms of a short-lived process) */
    for (auto &ev : c.ev) HIPCHK(hipEventCreate(&ev));
    for (auto &ev : c.pipe_ev) HIPCHK(hipEventCreate(&ev));
    c.ready = true;
    return LZ77X_OK;
}

/* the context's auxiliary streams exist from their first use on */
int need_stream(Ctx &c, hipStream_t Ctx::*m)
{
    if (c.*m) return LZ77X_OK;
    int cur = -1;
    HIPCHK(hipGetDevice(&cur));
    if (cur != c.device) HIPCHK(hipSetDevice(c.device));
    HIPCHK(hipStreamCreateWithFlags(&(c.*m), hipStreamNonBlocking));
    if (cur != c.device) HIPCHK(hipSetDevice(cur));
    return LZ77X_OK;
}

int check_geom(int &sb, int &la)
{
    if (sb == -1) sb = LZ77X_DEFAULT_SB;     /* lz77.c:65-66 */
    if (la == -1) la = LZ77X_DEFAULT_LA;
    if (sb < 1 || sb > 65535 || la < 2 || la > 255) return LZ77X_E_ARG;   /* main.c:35-38; -s 0 crashes the reference */
    return LZ77X_OK;
}

/* The primary context lives on whatever device is current when the library is entered; if the
 * caller has switched devices since the last call, the cached contexts are rebuilt there. */
int primary_context(CtxSet &S)
{
    Ctx &g_ctx = S.primary;
    int cur = -1;
    if (g_ctx.ready && hipGetDevice(&cur) == hipSuccess && cur != g_ctx.device) {
        for (Ctx *c : S.more) { ctx_release(*c); delete c; }
        S.more.clear();
        ctx_release(g_ctx);
        HIPCHK(hipSetDevice(cur));
    }
    return ctx_init(g_ctx);
}

int shard_contexts(CtxSet &S, int want, std::vector<Ctx *> &cs)
{
    int rc = primary_context(S);
    if (rc) return rc;
    Ctx &g_ctx = S.primary;
    std::vector<Ctx *> &g_more = S.more;
    cs.clear();
    cs.push_back(&g_ctx);
    int logical = g_ctx.ndev;
    const char *fk = getenv("LZ77X_FAKE_DEVICES");
    if (fk && atoi(fk) > logical) logical = atoi(fk);
    if (want > logical) want = logical;
    for (int i = 1; i < want; i++) {
        if ((int)g_more.size() < i) g_more.push_back(new Ctx());
        Ctx *c = g_more[i - 1];
        if ((rc = ctx_init(*c, (g_ctx.device + i) % g_ctx.ndev))) return rc;
        cs.push_back(c);
    }
    HIPCHK(hipSetDevice(g_ctx.device));
    return LZ77X_OK;
}

/* geometry of an encode: the production layout unless a pair-scan cross-check is selected */
void make_encode_geom(lz77x_geom *g, int sb, int la)
{
    lz77x_make_geom(g, sb, la);
    const char *vs = LZ77X_VENV("LZ77X_MATCH_VARIANT");
    const int variant = vs ? atoi(vs) : 0;
    if (variant == 1 || variant == 3) lz77x_geom_legacy(g);
}

void ctx_release(Ctx &c)
{
    if (c.pipe) { ctx_release(*c.pipe); delete c.pipe; c.pipe = nullptr; }
    if (c.drain) { ctx_release(*c.drain); delete c.drain; c.drain = nullptr; }
    if (!c.ready) return;
    hipError_t e = hipSetDevice(c.device);
    e = hipDeviceSynchronize();
    for (DevBuf *b : c.dev_bufs()) {
        if (b->p) e = hipFree(b->p);
        b->p = nullptr;
        b->cap = 0;
    }
    for (PinBuf *b : c.pin_bufs()) b->release();
    for (auto *v : {&c.chunk_ev, &c.tok_ev, &c.sort_ev, &c.match_ev, &c.tie_ev}) {
        for (hipEvent_t ev : *v) e = hipEventDestroy(ev);
        v->clear();
    }
    for (auto &ev : c.ev) e = hipEventDestroy(ev);
    for (auto &ev : c.pipe_ev) e = hipEventDestroy(ev);
    e = hipStreamDestroy(c.stream);
    for (hipStream_t *q : {&c.copy, &c.up, &c.tok})
        if (*q) { e = hipStreamDestroy(*q); *q = nullptr; }
    (void)e;
    c.ready = false;
}

}  // namespace lz77x_host
