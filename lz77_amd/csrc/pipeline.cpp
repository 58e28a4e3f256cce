/*
 * pipeline.cpp -- host side of the hot path and the C ABI (include/lz77_mi355x.h).
 *
 * encode (replaces lz77.c:51-140):
 *     input -> HBM -> k_match (longest match + in-order neighbours, all positions)
 *           -> D2H {ps, maxlen} -> host: parse chain + priority recurrence (hoststage.c)
 *           -> H2D {chain, xval} -> k_xfer_* -> k_tokens -> k_pack -> stream in HBM
 * decode (replaces lz77.c:148-197):
 *     stream -> HBM -> k_dec_parse -> scan -> k_dec_expand -> k_dec_jump* -> k_dec_gather
 *
 * There is NO CPU fallback: without a HIP device every entry point returns LZ77X_E_NODEV.
 */
#include "lz77x_internal.h"
#include "../../include/lz77_mi355x.h"

#include <hip/hip_runtime_api.h>
#include <errno.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#ifndef MADV_HUGEPAGE
#define MADV_HUGEPAGE 14          /* Linux; not exposed in every compilation pass of hipcc */
#endif

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[256] = "";
thread_local lz77x_stats g_stats;
int g_shards = 0;

#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            snprintf(g_err, sizeof g_err, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return LZ77X_E_HIP;                                                                \
        }                                                                                      \
    } while (0)

/* LZ77X_TRACE=1: phase timestamps on stderr (opt-in; the default run prints nothing) */
bool trace_on()
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("LZ77X_TRACE"); on = e && atoi(e) ? 1 : 0; }
    return on == 1;
}
#define TRACE(label, t0)                                                                  \
    do { if (trace_on()) fprintf(stderr, "[lz77x] %-28s %8.2f ms\n", label, now_ms() - (t0)); } while (0)

double now_ms();
void trace_allocs(const char *what);
double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

/* LZ77X_TRACE: where a call's wall time goes besides kernels and copies (per thread, reset by the entry points) */
thread_local double g_alloc_ms = 0, g_pin_ms = 0, g_fread_ms = 0, g_fwrite_ms = 0;
thread_local size_t g_alloc_bytes = 0, g_pin_bytes = 0;

void trace_allocs(const char *what)
{
    if (!trace_on()) return;
    fprintf(stderr, "[lz77x] %-28s hipMalloc %.2f ms (%.1f MB), pinned host %.2f ms (%.1f MB); in file reads %.2f ms, in file writes %.2f ms\n", what,
            g_alloc_ms, g_alloc_bytes / 1e6, g_pin_ms, g_pin_bytes / 1e6, g_fread_ms, g_fwrite_ms);
    g_alloc_ms = g_pin_ms = g_fread_ms = g_fwrite_ms = 0;
    g_alloc_bytes = g_pin_bytes = 0;
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int need(size_t bytes)
    {
        if (bytes <= cap) return LZ77X_OK;
        const double t0 = trace_on() ? now_ms() : 0;
        if (p) { hipError_t e0 = hipFree(p); (void)e0; p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 4096;
        HIPCHK(hipMalloc(&p, want));
        cap = want;
        if (trace_on()) { g_alloc_ms += now_ms() - t0; g_alloc_bytes += want; }
        return LZ77X_OK;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

/* Pinned host memory.  Large buffers are 2 MiB-aligned anonymous memory advised to transparent huge
 * pages and then registered with HIP (the host recurrence streams through them: fewer TLB misses,
 * measured -6..8 % on it); small ones, or if anything in that path fails, plain hipHostMalloc. */
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool registered = false;
    void release()
    {
        if (!p) return;
        if (registered) { hipError_t e0 = hipHostUnregister(p); (void)e0; free(p); }
        else { hipError_t e0 = hipHostFree(p); (void)e0; }
        p = nullptr;
        cap = 0;
        registered = false;
    }
    int need(size_t bytes)
    {
        if (bytes <= cap) return LZ77X_OK;
        struct Timer {
            double t0 = trace_on() ? now_ms() : 0;
            size_t bytes;
            explicit Timer(size_t b) : bytes(b) {}
            ~Timer() { if (trace_on()) { g_pin_ms += now_ms() - t0; g_pin_bytes += bytes; } }
        } timer(bytes);
        release();
        size_t want = bytes + bytes / 8 + 4096;
        const char *hp = getenv("LZ77X_HUGEPAGES");
        if (want >= ((size_t)8 << 20) && !(hp && !atoi(hp))) {
            const size_t two_mb = (size_t)2 << 20;
            want = (want + two_mb - 1) & ~(two_mb - 1);
            void *q = nullptr;
            if (posix_memalign(&q, two_mb, want) == 0) {
                madvise(q, want, MADV_HUGEPAGE);
                if (hipHostRegister(q, want, hipHostRegisterPortable) == hipSuccess) {
                    p = q;
                    cap = want;
                    registered = true;
                    return LZ77X_OK;
                }
                (void)hipGetLastError();
                free(q);
            }
        }
        HIPCHK(hipHostMalloc(&p, want, hipHostMallocPortable));   /* every shard's device copies to/from it */
        cap = want;
        return LZ77X_OK;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

/* ---- host copies between the caller's pageable memory and the pinned staging slots --------------------------------
 * hipMemcpy on pageable memory is staged by the runtime on ONE thread (~5 GB/s: 100 MB in and 47 MB out cost an encode
 * through lz77x_encode three times its kernels).  The buffer-level entry points stage through the context's two pinned
 * slots themselves and cut every piece over a few host threads (memory bandwidth, not a core, is then the limit); the DMA
 * of one slot runs while the other is filled or drained. */
class CopyPool {
    static constexpr int NW = 3;                          /* helpers beside the calling thread */
    std::mutex job_mu;                                     /* one parallel copy at a time */
    std::mutex mu;
    std::condition_variable cv, done_cv;
    std::thread th[NW];
    /* kind 0: memcpy(d, s, n); 1: pread(fd, d, n, off); 2: pwrite(fd, s, n, off) -- the last two until done, EOF or error */
    struct Task { int kind = 0, fd = -1; uint8_t *d = nullptr; const uint8_t *s = nullptr; size_t n = 0; off_t off = 0; ssize_t done = 0; } task[NW + 1];
    int pending = 0;
    bool started = false, failed = false, stop = false;
    static void run(Task &t)
    {
        if (t.kind == 0) { memcpy(t.d, t.s, t.n); t.done = (ssize_t)t.n; return; }
        size_t at = 0;
        while (at < t.n) {
            const ssize_t r = t.kind == 1 ? pread(t.fd, t.d + at, t.n - at, t.off + (off_t)at) : pwrite(t.fd, t.s + at, t.n - at, t.off + (off_t)at);
            if (r < 0) { if (errno == EINTR) continue; t.done = -1; return; }
            if (r == 0) break;                             /* end of the file */
            at += (size_t)r;
        }
        t.done = (ssize_t)at;
    }
    void worker(int i)
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return stop || task[i].n; });
            if (stop) return;
            lk.unlock();
            run(task[i]);
            lk.lock();
            task[i].n = 0;
            if (--pending == 0) done_cv.notify_all();
        }
    }
    /* the job cut into NW + 1 parts; -> bytes done in order (a short part ends the count), or -1 */
    ssize_t parallel(int kind, int fd, uint8_t *d, const uint8_t *sp, size_t n, off_t off)
    {
        std::lock_guard<std::mutex> job(job_mu);
        if (!started && !failed) {
            try { for (int i = 0; i < NW; i++) th[i] = std::thread(&CopyPool::worker, this, i); started = true; }
            catch (...) { failed = true; }                 /* (no helpers: the caller works alone; what did start is stopped by the destructor) */
        }
        const int parts = started && n >= ((size_t)2 << 20) ? NW + 1 : 1;
        const size_t part = parts == 1 ? n : (n / (size_t)parts + 4095) & ~(size_t)4095;
        size_t want[NW + 1];
        {
            std::lock_guard<std::mutex> lk(mu);
            for (int i = 0; i < parts; i++) {
                Task &t = task[i == 0 ? NW : i - 1];       /* part 0 is the caller's */
                const size_t b = (size_t)i * part, e = i + 1 < parts ? b + part : n;
                t.kind = kind; t.fd = fd; t.d = d ? d + b : nullptr; t.s = sp ? sp + b : nullptr; t.off = off + (off_t)b; t.done = 0;
                want[i] = e > b ? e - b : 0;
                t.n = i == 0 ? 0 : want[i];                /* (workers wake on n != 0) */
            }
            pending = 0;
            for (int i = 1; i < parts; i++) pending += want[i] ? 1 : 0;
        }
        if (parts > 1) cv.notify_all();
        Task mine = task[NW];
        mine.n = want[0];
        run(mine);
        if (parts > 1) {
            std::unique_lock<std::mutex> lk(mu);
            done_cv.wait(lk, [&] { return pending == 0; });
        }
        ssize_t total = 0;
        for (int i = 0; i < parts; i++) {
            const ssize_t dn = i == 0 ? mine.done : task[i - 1].done;
            if (dn < 0) return -1;
            total += dn;
            if ((size_t)dn < want[i]) break;               /* a short part: what lies behind it does not count */
        }
        return total;
    }
public:
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto &t : th) if (t.joinable()) t.join();
    }
    void copy(void *dst, const void *src, size_t n)
    {
        if (n < ((size_t)2 << 20)) { memcpy(dst, src, n); return; }
        (void)parallel(0, -1, reinterpret_cast<uint8_t *>(dst), reinterpret_cast<const uint8_t *>(src), n, 0);
    }
    /* a regular file's bytes [off, off + n) into dst / from src, cut over the threads: -> bytes moved (short at the end of
     * the file), or -1 */
    ssize_t read_at(int fd, void *dst, size_t n, off_t off) { return parallel(1, fd, reinterpret_cast<uint8_t *>(dst), nullptr, n, off); }
    ssize_t write_at(int fd, const void *src, size_t n, off_t off) { return parallel(2, fd, nullptr, reinterpret_cast<const uint8_t *>(src), n, off); }
};
CopyPool g_copy;                                           /* towards the device: copies out of pageable memory, file reads */
CopyPool g_copy_out;                                       /* away from it: copies into pageable memory, file writes -- an encode of several
                                                              segments reads its next one while the last one's words are written */

/* whatever way a multi-device function is left, the thread's current device is the one it came in with (a
 * HIPCHK return in the middle of a per-shard loop would otherwise leave another shard's device current, and the
 * next library call would rebuild every cached context there) */
struct DeviceRestore {
    int dev;
    explicit DeviceRestore(int d) : dev(d) {}
    ~DeviceRestore() { hipError_t e = hipSetDevice(dev); (void)e; }
    DeviceRestore(const DeviceRestore &) = delete;
    DeviceRestore &operator=(const DeviceRestore &) = delete;
};

/* fn(d) for every shard d, each on a host thread of its own (the first on the calling thread): per-device
 * allocations, copies from the caller's pageable buffer (which block their thread) and result fetches of D devices
 * overlap instead of queueing behind one another.  fn makes its device current itself; the first error wins, its
 * text ends up in the caller's g_err */
template <class F> int for_each_shard(size_t D, F fn)
{
    std::vector<int> rcs(D, LZ77X_OK);
    std::vector<std::string> msgs(D);
    auto run = [&](size_t d) {
        g_err[0] = 0;
        rcs[d] = fn(d);
        if (rcs[d]) msgs[d] = g_err;
    };
    std::vector<std::thread> th;
    bool spawn_failed = false;
    try {
        th.reserve(D);
        for (size_t d = 1; d < D; d++) th.emplace_back(run, d);
    } catch (...) {
        /* no exception crosses the C ABI, and a joinable std::thread must not be destroyed: what started is joined below,
         * the shards without a thread run on this one */
        spawn_failed = true;
    }
    if (D) run(0);
    if (spawn_failed)
        for (size_t d = th.size() + 1; d < D; d++) run(d);
    for (auto &t : th) t.join();
    for (size_t d = 0; d < D; d++)
        if (rcs[d]) { snprintf(g_err, sizeof g_err, "%s", msgs[d].c_str()); return rcs[d]; }
    return LZ77X_OK;
}

struct Ctx {
    bool ready = false;
    int ndev = 0;
    int device = 0;                          /* physical HIP device this context lives on */
    hipStream_t stream = nullptr;            /* used when the caller passes none (host-level API) */
    hipStream_t copy = nullptr;              /* device->host copies of intermediates, overlapped with kernels */
    hipStream_t up = nullptr;                /* host->device copies (own stream: never queued behind a D2H that
                                                still waits for a later match launch) */
    hipStream_t tok = nullptr;               /* per-chunk index + tie-break + pack kernels */
    hipEvent_t ev[6] = {};
    hipEvent_t pipe_ev[3] = {};              /* [0] this context's input has arrived, [1] its last segment's result is out,
                                                [2] the parse chain (runs beside the recurrence on `tok`) is done */
    Ctx *pipe = nullptr;                     /* second context set on the same device (two segments of one stream in flight) */
    Ctx *drain = nullptr;                    /* a stream and two pinned slots for the thread that hands an encode's segments to a
                                                file or host memory while the next ones are computed (nothing else is used) */
    std::vector<hipEvent_t> chunk_ev, tok_ev, sort_ev, match_ev, tie_ev;
    DevBuf in, ps, maxlen, scratch, xval, chain, ofs, ent, tokval, out, scantmp;
    DevBuf z, z2, out2, dcarry, len1, dst, ptr, flag, tstart, bidx, cells, ranks_all, prio_tmp, chain_tmp, look;
    PinBuf h_ps, h_maxlen, h_xval, h_chain, h_small, h_tok, h_stage, h_tbase;
    /* every cached buffer, so that no release path can forget one */
    std::vector<DevBuf *> dev_bufs()
    {
        return {&in, &ps, &maxlen, &scratch, &xval, &chain, &ofs, &ent, &tokval, &out, &scantmp, &z, &z2, &out2, &dcarry, &len1, &dst, &ptr,
                &flag, &tstart, &bidx, &cells, &ranks_all, &prio_tmp, &chain_tmp, &look};
    }
    std::vector<PinBuf *> pin_bufs() { return {&h_ps, &h_maxlen, &h_xval, &h_chain, &h_small, &h_tok, &h_stage, &h_tbase}; }
};

/* One CtxSet serves one call at a time: `primary` lives on the caller's current device (pinned host
 * buffers, final stream), `more` are the contexts of the other shards.  Concurrent callers (threads
 * compressing different files) each lease their own set, up to LZ77X_MAX_CONTEXTS (default 4), so that
 * the host recurrence of one stream overlaps the GPU work and the recurrences of the others -- one
 * stream keeps the GPU busy for only a third of its own wall time. */
struct CtxSet {
    Ctx primary;
    std::vector<Ctx *> more;
    bool busy = false;
    size_t promised = 0;                  /* device memory this call's plan needs and does not hold yet (budget_commit) */
};
thread_local CtxSet *tl_set = nullptr;    /* the set leased by this thread's call */
std::vector<CtxSet *> g_pool;
std::mutex g_mu;
std::condition_variable g_cv;

struct Lease {
    CtxSet *set = nullptr;
    Lease()
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        static int cap = 0;
        std::unique_lock<std::mutex> lk(g_mu);
        if (!cap) { const char *e = getenv("LZ77X_MAX_CONTEXTS"); cap = e && atoi(e) > 0 ? atoi(e) : 4; }
        for (;;) {
            CtxSet *elsewhere = nullptr;
            for (CtxSet *s : g_pool) {
                if (s->busy) continue;
                if (!s->primary.ready || s->primary.device == cur) { set = s; break; }
                elsewhere = s;
            }
            if (set) break;
            if ((int)g_pool.size() < cap) { set = new CtxSet(); g_pool.push_back(set); break; }
            if (elsewhere) { set = elsewhere; break; }       /* primary_context(*lease.set) moves it to this device */
            g_cv.wait(lk);
        }
        set->busy = true;
        set->promised = 0;
        tl_set = set;
    }
    ~Lease()
    {
        { std::lock_guard<std::mutex> lk(g_mu); set->busy = false; set->promised = 0; }
        tl_set = nullptr;
        g_cv.notify_one();
    }
    Lease(const Lease &) = delete;
    Lease &operator=(const Lease &) = delete;
};

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

int ctx_init(Ctx &c, int device = -1)
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

size_t stream_bytes(uint64_t ntok, int T) { return 4 + (size_t)((ntok * (uint64_t)T + 7) / 8); }

/* ---------------------------------------------------------------- encode ------------ */

/* Contexts taking part in one encode: cs[0] is the caller's device (holds the pinned host buffers
 * and the final stream), cs[1..] the other shards.  LZ77X_FAKE_DEVICES=k lets k contexts share one
 * physical GPU so that the multi-device path can be exercised on a single-GPU box. */
void ctx_release(Ctx &c);

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
static void make_encode_geom(lz77x_geom *g, int sb, int la)
{
    lz77x_make_geom(g, sb, la);
    const char *vs = LZ77X_VENV("LZ77X_MATCH_VARIANT");
    const int variant = vs ? atoi(vs) : 0;
    if (variant == 1 || variant == 3) lz77x_geom_legacy(g);
}

/* src is a device pointer on cs[0]'s device (src_on_device, single shard only) or a host pointer.
 * On success the stream is in cs[0]->out (device) and *zn holds its size.
 *
 * Positions are cut into host chunks of per_chunk regions; contiguous runs of chunks form the
 * shards, one per context/device (read-only halos, no device-to-device traffic).
 * Per device the match kernels are launched for groups of chunks; each chunk's {ps, maxlen} is
 * copied to the host as soon as its launch retires, and the host's sequential stage consumes chunk
 * i while the GPUs are already working on later chunks.  The host's products (xval, chain) go back
 * to the chunk's owner, whose token stream resolves and emits that chunk's tokens at once. */
int encode_core_host(std::vector<Ctx *> &cs, const void *src, bool src_on_device, size_t n, const lz77x_geom &g, hipStream_t s, size_t *zn)
{
    const double t_begin = now_ms();
    memset(&g_stats, 0, sizeof g_stats);
    if (n > LZ77X_MAX_N) return LZ77X_E_TOOBIG;
    for (Ctx *cc : cs)
        for (hipStream_t Ctx::*m : {&Ctx::copy, &Ctx::up, &Ctx::tok})
            if (int r = need_stream(*cc, m)) return r;
    Ctx &c0 = *cs[0];
    const uint32_t D = (uint32_t)cs.size();
    if (D > 1 && src_on_device) return LZ77X_E_ARG;
    const uint32_t n32 = (uint32_t)n;
    int rc;
    double waited = 0;
    auto kstream = [&](uint32_t d) { return d == 0 ? s : cs[d]->stream; };

    TRACE("encode_core entry", t_begin);
    for (uint32_t d = 0; d < D; d++) {
        Ctx &c = *cs[d];
        HIPCHK(hipSetDevice(c.device));
        if ((rc = c.in.need(n + LZ77X_PAD + 16))) return rc;
        if (n && src != c.in.p)                                /* the file path streams straight into c.in */
            HIPCHK(hipMemcpyAsync(c.in.p, src, n, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, kstream(d)));
        HIPCHK(lz77k_fill_pad(c.in.as<uint8_t>(), n32, kstream(d)));
    }
    HIPCHK(hipSetDevice(c0.device));

    uint32_t ntok = 0, nchunks_done = 0, launches0_total = 0;
    uint64_t transfers = 0;
    bool sort_timed = false;
    std::vector<uint32_t> owner;
    std::vector<char> tie_timed;
    if (n) {
        const size_t nx = n > (size_t)g.sb ? n - (size_t)g.sb : 0;     /* evicted positions */
        const uint32_t ring_mask = lz77x_prio_mask(g.sb);
        const uint32_t nregions = (uint32_t)((n + g.TILE - 1) / g.TILE);
        /* host chunk: ~4M positions on the LDS path, >= 256 regions for large windows; match launch: a
         * group of chunks (8 on the LDS path so that the walkers fill the chip; 256 large-window regions
         * are one full round of resident workgroups already, and finer chunks pipeline better with the
         * host stage: 512 -> 256 regions took 40 ms off a 350 ms encode of S3) */
        uint32_t per_chunk = (uint32_t)((((size_t)4 << 20) + g.TILE - 1) / g.TILE);
        /* large windows.  Per-region sort kernel: one workgroup per CU, a full round of 256 regions.  Shared
         * hierarchical sort (grid-wide launches): 128 -- the host recurrence is the critical path (S3: 183 of a
         * 210 ms encode) and smaller chunks shorten what runs before its first and after its last position
         * (228 -> 210 ms), while 64 starts to cost the walkers their occupancy */
        if (!g.fast && per_chunk < 256) per_chunk = lz77k_big_sort_shared(g) ? 128 : 256;
        const char *cs_env = getenv("LZ77X_CHUNK_REGIONS");
        if (cs_env && atoi(cs_env) > 0) per_chunk = (uint32_t)atoi(cs_env);
        uint32_t group = g.fast ? 8u : 1u;
        const char *gs = getenv("LZ77X_MATCH_GROUP");
        if (gs && atoi(gs) > 0) group = (uint32_t)atoi(gs);
        {
            const size_t per = lz77k_match_scratch_bytes(g, 1);
            /* scratch budget of the match stage: small for the LDS path, generous for large windows
             * (their walkers are latency bound and want every region of the input in one launch) */
            const uint32_t fit = (uint32_t)(((size_t)(g.fast ? 2 : 12) << 30) / per);
            if (per_chunk > fit) per_chunk = fit ? fit : 1;
            if ((uint64_t)per_chunk * group > fit) group = fit / per_chunk ? fit / per_chunk : 1;
        }
        const uint32_t nchunks = (nregions + per_chunk - 1) / per_chunk;
        const size_t chunk_pos = (size_t)per_chunk * g.TILE;
        const size_t idx_span = (chunk_pos < n ? chunk_pos : n) + 2 * (size_t)g.sb + 16;
        owner.resize(nchunks);
        std::vector<uint32_t> first_chunk(D + 1, nchunks);
        for (uint32_t ci = 0; ci < nchunks; ci++) {
            owner[ci] = (uint32_t)((uint64_t)ci * D / nchunks);
            if (first_chunk[owner[ci]] == nchunks) first_chunk[owner[ci]] = ci;
        }
        for (int d = (int)D - 1; d >= 0; d--)
            if (first_chunk[d] == nchunks) first_chunk[d] = first_chunk[d + 1];      /* shard without chunks */

        const char *vs = LZ77X_VENV("LZ77X_MATCH_VARIANT");
        const int variant = vs ? atoi(vs) : 0;
        const char *tv = LZ77X_VENV("LZ77X_TOKEN_VARIANT");
        const int tvariant = tv ? atoi(tv) : 0;
        const bool keep_ranks = !g.fast && tvariant == 0 && (variant == 0 || variant > 3);
        const char *sv = LZ77X_VENV("LZ77X_SERIAL");               /* profiling aid: token kernels queue behind */
        const bool serial = sv && atoi(sv);                    /* the match launches, no overlap */
        auto tstream = [&](uint32_t d) { return serial ? kstream(d) : cs[d]->tok; };

        /* Pinned host memory is a set of rings of chunk-sized slots, not n-sized arrays (SURVEY 8f-2:
         * host RAM must not scale with the input several times over): K slots receive {cells, maxlen}
         * from the devices, K2 slots carry {xval, chain} back.  A slot is recycled once the chunk
         * AFTER it has been consumed (both recurrences look sb positions back into the previous chunk;
         * chunk_pos >= 2*sb by construction of TILE). */
        uint32_t K = 16, K2 = 4;               /* 12..32 slots measure the same: the host paces the pipeline */
        const char *rs = getenv("LZ77X_RING_SLOTS");
        if (rs && atoi(rs) > 0) K = (uint32_t)atoi(rs);
        if (K < group + 2) K = group + 2;                      /* the group being filled + the two chunks in use */
        if (K > nchunks) K = nchunks;
        if (K2 > nchunks) K2 = nchunks;
        const bool ring_d2h = K < nchunks, ring_h2d = K2 < nchunks;
        {
            const double t_pin = now_ms();
            if ((rc = c0.h_ps.need(((size_t)K * chunk_pos + 8) * 4))) return rc;
            if ((rc = c0.h_maxlen.need((size_t)K * chunk_pos + 8))) return rc;
            if ((rc = c0.h_xval.need(((size_t)K2 * chunk_pos + 8) * 4))) return rc;
            if ((rc = c0.h_chain.need(((size_t)K2 * chunk_pos + 8) * 4))) return rc;
            TRACE("pinned host buffers", t_pin);
        }
        auto chunk_b = [&](uint32_t ci) { return (size_t)ci * chunk_pos; };
        auto chunk_e = [&](uint32_t ci) { const size_t e = (size_t)(ci + 1) * chunk_pos; return e < n ? e : n; };
        /* slot of chunk ci, and the same pointer rebased so that it can be indexed by absolute position */
        auto ps_slot = [&](uint32_t ci) { return c0.h_ps.as<uint32_t>() + (size_t)(ci % K) * chunk_pos; };
        auto ml_slot = [&](uint32_t ci) { return c0.h_maxlen.as<uint8_t>() + (size_t)(ci % K) * chunk_pos; };
        auto xv_slot = [&](uint32_t ci) { return c0.h_xval.as<uint32_t>() + (size_t)(ci % K2) * chunk_pos; };
        auto ch_slot = [&](uint32_t ci) { return c0.h_chain.as<uint32_t>() + (size_t)(ci % K2) * chunk_pos; };
        auto rebase32 = [&](uint32_t *slot, uint32_t ci) {
            return reinterpret_cast<uint32_t *>(reinterpret_cast<uintptr_t>(slot) - chunk_b(ci) * sizeof(uint32_t));
        };
        auto rebase8 = [&](uint8_t *slot, uint32_t ci) {
            return reinterpret_cast<uint8_t *>(reinterpret_cast<uintptr_t>(slot) - chunk_b(ci));
        };

        for (uint32_t d = 0; d < D; d++) {
            Ctx &c = *cs[d];
            HIPCHK(hipSetDevice(c.device));
            const uint64_t most = (uint64_t)per_chunk * group < nregions ? (uint64_t)per_chunk * group : nregions;
            if ((rc = c.scratch.need(lz77k_match_scratch_bytes(g, (uint32_t)most + (d > 0 ? 1u : 0u))))) return rc;
            if ((rc = c.ps.need((n + 8) * 4))) return rc;
            if ((rc = c.cells.need((n + 8) * 4))) return rc;
            if ((rc = c.maxlen.need(n + 8))) return rc;
            if ((rc = c.xval.need((n + 8) * 4))) return rc;
            if ((rc = c.chain.need((n + 8) * 4))) return rc;
            if ((rc = c.ofs.need((idx_span + 8) * 4))) return rc;
            if ((rc = c.ent.need((idx_span + 8) * 8))) return rc;
            if ((rc = c.tokval.need((n + 8) * 4))) return rc;
            if ((rc = c.scantmp.need(lz77k_scan_tmp_bytes((uint32_t)idx_span + 1)))) return rc;
            if ((rc = c.tstart.need(lz77k_tokens_tmp_bytes((uint32_t)idx_span, g)))) return rc;
            if ((rc = c.flag.need(64))) return rc;
            HIPCHK(hipMemsetAsync(c.flag.p, 0, 64, kstream(d)));
            if (keep_ranks) {
                /* large windows: the regions' rank + inverse arrays stay resident for the rank-order tie-break */
                if ((rc = c.ranks_all.need((size_t)nregions * (2 * (size_t)g.RP + 8) * sizeof(uint32_t)))) return rc;
            }
            /* (the rank-order tie-break builds its short-token buckets here too; the variants' two-byte index shares the buffer) */
            if ((rc = c.bidx.need(lz77k_tokens_index_bytes(g, (chunk_pos < n ? chunk_pos : n) + 2 * (size_t)g.sb)))) return rc;
            while (c.chunk_ev.size() < 3 * (size_t)nchunks) {
                hipEvent_t e;
                /* ordering only; host waiters sleep instead of spinning next to the recurrence thread */
                HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));
                c.chunk_ev.push_back(e);
            }
            while (c.sort_ev.size() < 4 * (size_t)nchunks) {
                hipEvent_t e;
                HIPCHK(hipEventCreate(&e));                                    /* region sort | walkers */
                c.sort_ev.push_back(e);
            }
            while (c.tok_ev.size() < 2 * (size_t)nchunks) {
                hipEvent_t e;
                HIPCHK(hipEventCreate(&e));                                    /* token-stream kernel time */
                c.tok_ev.push_back(e);
                HIPCHK(hipEventCreate(&e));                                    /* tie-break kernel time */
                c.tie_ev.push_back(e);
                HIPCHK(hipEventCreate(&e));                                    /* whole match group time */
                c.match_ev.push_back(e);
            }
        }
        HIPCHK(hipSetDevice(c0.device));

        /* -- match launches cover groups of host chunks (the window walkers want >= 256 wavefronts per
         *    launch); the first groups are small (1, 2, 4 chunks) so that the host stage can start early.
         *    Groups are enqueued in stream order as ring slots become free. -- */
        struct Group { uint32_t d, ci, nchunks; };
        std::vector<Group> groups;
        for (uint32_t d = 0; d < D; d++) {
            uint32_t ramp = d == 0 ? 1u : group;
            for (uint32_t ci = first_chunk[d]; ci < first_chunk[d + 1];) {
                uint32_t gchunks = ramp < group ? ramp : group;
                ramp = ramp < group ? ramp * 2 : group;
                if (ci + gchunks > first_chunk[d + 1]) gchunks = first_chunk[d + 1] - ci;
                groups.push_back({d, ci, gchunks});
                ci += gchunks;
            }
        }
        uint32_t launches0 = 0;                                /* match launches on the first device (timed) */
        auto enqueue_group = [&](const Group &G) -> int {
            Ctx &c = *cs[G.d];
            HIPCHK(hipSetDevice(c.device));
            uint32_t r0 = G.ci * per_chunk;
            uint32_t nr = G.nchunks * per_chunk;
            if (nr > nregions - r0) nr = nregions - r0;
            if (g.shifted && G.d > 0 && G.ci == first_chunk[G.d] && r0 > 0) {
                /* first launch of a shard: maxlen[] of its first sb positions comes from the region before */
                r0--;
                nr++;
            }
            if (G.d == 0) HIPCHK(hipEventRecord(c.match_ev[2 * launches0], kstream(G.d)));
            HIPCHK(lz77k_match(c.in.as<uint8_t>(), n32, g, r0, nr, c.ps.as<uint32_t>(), c.maxlen.as<uint8_t>(),
                               c.scratch.p, variant, kstream(G.d), G.d == 0 ? &c.sort_ev[4 * launches0] : nullptr,
                               keep_ranks ? c.ranks_all.as<uint32_t>() : nullptr));
            g_stats.match_launches++;
            {
                const size_t gb = (size_t)r0 * g.TILE;
                size_t ge = (size_t)(r0 + nr) * g.TILE;
                if (ge > nx) ge = nx;
                if (ge > gb) HIPCHK(lz77k_ps_cells(c.ps.as<uint32_t>(), c.cells.as<uint32_t>(), (uint32_t)gb, (uint32_t)ge, ring_mask, kstream(G.d)));
            }
            if (G.d == 0) {
                HIPCHK(hipEventRecord(c.match_ev[2 * launches0 + 1], kstream(G.d)));
                sort_timed = variant == 0 || variant > 3;
                launches0++;
            }
            HIPCHK(hipEventRecord(c.chunk_ev[3 * G.ci], kstream(G.d)));
            HIPCHK(hipStreamWaitEvent(c.copy, c.chunk_ev[3 * G.ci], 0));
            for (uint32_t cj = G.ci; cj < G.ci + G.nchunks; cj++) {
                const size_t b = chunk_b(cj), e = chunk_e(cj);
                HIPCHK(hipMemcpyAsync(ml_slot(cj), c.maxlen.as<uint8_t>() + b, e - b, hipMemcpyDeviceToHost, c.copy));
                const size_t pe = e < nx ? e : nx;
                if (pe > b) HIPCHK(hipMemcpyAsync(ps_slot(cj), c.cells.as<uint32_t>() + b, (pe - b) * 4, hipMemcpyDeviceToHost, c.copy));
                HIPCHK(hipEventRecord(c.chunk_ev[3 * cj + 1], c.copy));
            }
            return hipSetDevice(c0.device) == hipSuccess ? LZ77X_OK : LZ77X_E_HIP;
        };

        lz77x_prio_state st;
        if (!lz77x_prio_init(&st, g.sb)) return LZ77X_E_NOMEM;
        /* whatever way this scope is left: the recurrence's ring is freed, and on an error exit every
         * device is drained first -- copies into the pinned rings and kernels on the side streams may
         * still be in flight, and the cached buffers go back to the pool with the lease */
        struct StageGuard {
            lz77x_prio_state &st; std::vector<Ctx *> &cs; bool ok = false;
            ~StageGuard()
            {
                if (!ok) {
                    for (Ctx *c : cs) { hipError_t q = hipSetDevice(c->device); q = hipDeviceSynchronize(); (void)q; }
                    hipError_t q = hipSetDevice(cs[0]->device); (void)q;
                }
                lz77x_prio_free(&st);
            }
        } stage_guard{st, cs};

        /* The two host recurrences are independent of each other (SURVEY A.2 vs A.5).  The priority
         * recurrence -- the critical path of the whole encode -- gets a thread of its own that does
         * nothing else; the calling thread walks the parse chain (8x cheaper) and does all the HIP
         * enqueueing as ring slots, device results and the recurrence allow. */
        std::mutex pm;
        std::condition_variable pcv;
        uint32_t prio_chunks = 0;                      /* chunks the recurrence is through           (guarded by pm) */
        uint32_t enq_chunks = 0;                       /* chunks whose device->host copy is enqueued (guarded by pm) */
        uint32_t h2d_done = 0;                         /* chunks whose host->device copy has landed  (guarded by pm) */
        bool abort_all = false;
        std::atomic<int> prio_err{0};
        double t_prio = 0, t_prio_first = 0, t_prio_last = 0;     /* recurrence: start of chunk 0, end of the last chunk */
        std::thread prio_thread([&]() {
            for (uint32_t ci = 0; ci < nchunks; ci++) {
                const size_t b = chunk_b(ci), e = chunk_e(ci);
                {
                    std::unique_lock<std::mutex> lk(pm);
                    /* xval slot of this chunk: free once the copy of the chunk K2-1 before it has landed */
                    pcv.wait(lk, [&] { return abort_all || (enq_chunks > ci && (!ring_h2d || h2d_done + K2 >= ci + 2)); });
                    if (abort_all) return;
                }
                if (hipEventSynchronize(cs[owner[ci]]->chunk_ev[3 * ci + 1]) != hipSuccess) prio_err.store(1);
                const double t0 = now_ms();
                if (ci > 0) {                          /* evictions x < b: their cells/xval live in the previous chunk's slots */
                    const size_t upto = e < b + (size_t)g.sb ? e : b + (size_t)g.sb;
                    lz77x_prio_run(&st, rebase32(ps_slot(ci - 1), ci - 1), g.sb, upto, rebase32(xv_slot(ci - 1), ci - 1));
                }
                lz77x_prio_run(&st, rebase32(ps_slot(ci), ci), g.sb, e, rebase32(xv_slot(ci), ci));
                t_prio += now_ms() - t0;
                if (ci == 0) t_prio_first = t0;
                t_prio_last = now_ms();
                { std::lock_guard<std::mutex> lk(pm); prio_chunks = ci + 1; }
                pcv.notify_all();
            }
        });
        struct Joiner {
            std::thread &t; std::mutex &m; std::condition_variable &cv; bool &flag;
            ~Joiner() { { std::lock_guard<std::mutex> lk(m); flag = true; } cv.notify_all(); if (t.joinable()) t.join(); }
        } joiner{prio_thread, pm, pcv, abort_all};

        size_t ntok_sz = 0, chain_p = 0, gi = 0;
        std::vector<size_t> tok_sent(D, 0), x_sent(D, 0), toks_at(nchunks + 1, 0);   /* tokens before chunk ci */
        std::vector<uint32_t> lookback;
        tie_timed.assign(nchunks, 0);
        double t_chain = 0;
        int err = LZ77X_OK;
        HIPCHK(hipEventRecord(c0.ev[0], kstream(0)));
        for (uint32_t ci = 0; ci < nchunks && err == LZ77X_OK; ci++) {
            const uint32_t d = owner[ci];
            Ctx &c = *cs[d];
            const size_t b = chunk_b(ci), e = chunk_e(ci);
            /* top up the device queue: a group may go out once every slot it lands in is free, i.e. the
             * chunk K before each of its chunks AND that chunk's successor have been consumed by both
             * host recurrences.  Chunk ci itself must be out before we can wait for it. */
            for (;;) {
                uint32_t snapshot;                              /* chunks the recurrence was through when we looked */
                { std::lock_guard<std::mutex> lk(pm); snapshot = prio_chunks; }
                const uint32_t through = snapshot > ci ? ci : snapshot;
                while (gi < groups.size() &&
                       (!ring_d2h || groups[gi].ci + groups[gi].nchunks + 1 <= (uint64_t)through + K)) {
                    if ((rc = enqueue_group(groups[gi]))) { err = rc; break; }
                    const uint32_t upto = groups[gi].ci + groups[gi].nchunks;
                    gi++;
                    { std::lock_guard<std::mutex> lk(pm); enq_chunks = upto; }
                    pcv.notify_all();
                }
                if (err != LZ77X_OK) break;
                {
                    std::unique_lock<std::mutex> lk(pm);
                    if (enq_chunks > ci) break;
                    /* ring full: wait until the recurrence has moved past the state the decision above was
                     * taken on (waiting for a change relative to a LATER reading could sleep through the
                     * very advance that frees the slot while the recurrence waits for this thread) */
                    pcv.wait(lk, [&] { return prio_chunks != snapshot; });
                }
            }
            if (err != LZ77X_OK) break;
            const double tw = now_ms();
            hipError_t he = hipEventSynchronize(c.chunk_ev[3 * ci + 1]);
            if (he == hipSuccess && ring_h2d && ci >= K2)      /* chain slot: the copy that last read it */
                he = hipEventSynchronize(cs[owner[ci - K2]]->chunk_ev[3 * (ci - K2) + 2]);
            const double t1 = now_ms();
            waited += t1 - tw;
            if (he != hipSuccess) { err = LZ77X_E_HIP; snprintf(g_err, sizeof g_err, "chunk sync: %s", hipGetErrorString(he)); break; }
            const size_t tok_before = ntok_sz;
            chain_p = lz77x_host_chain(rebase8(ml_slot(ci), ci), e, chain_p,
                                       reinterpret_cast<uint32_t *>(reinterpret_cast<uintptr_t>(ch_slot(ci)) - tok_before * sizeof(uint32_t)),
                                       &ntok_sz);
            const double t2 = now_ms();
            t_chain += t2 - t1;
            { std::unique_lock<std::mutex> lk(pm); pcv.wait(lk, [&] { return prio_chunks > ci; }); }
            if (prio_err.load()) { err = LZ77X_E_HIP; snprintf(g_err, sizeof g_err, "recurrence thread: event wait failed"); break; }
            const size_t x_done = e > (size_t)g.sb ? e - (size_t)g.sb : 0;
            /* hand-overs that can matter to tokens in [b, e): evictions before e-sb into dst >= b-sb */
            const uint32_t dbase = b > (size_t)g.sb ? (uint32_t)(b - (size_t)g.sb) : 0u;
            const uint32_t xa = dbase > (uint32_t)g.sb ? dbase - (uint32_t)g.sb : 0u;
            const size_t x_new = b > (size_t)g.sb ? b - (size_t)g.sb : 0;     /* evictions first seen with this chunk */
            auto enqueue = [&]() -> hipError_t {
                hipError_t q;
                if ((q = hipSetDevice(c.device)) != hipSuccess) return q;
                if (ci == first_chunk[d]) {
                    /* first chunk of a shard: its look-back window belongs to the previous shard */
                    tok_sent[d] = tok_before;
                    x_sent[d] = xa;
                    if (d > 0 && b > xa) {
                        /* the host holds ring cells; the index kernels want distances again */
                        lookback.resize(b - xa);
                        const uint32_t *hc = rebase32(ps_slot(ci - 1), ci - 1);      /* [xa, b) lies in the previous chunk */
                        for (size_t x = xa; x < b; x++) {
                            const uint32_t v = hc[x], x32 = (uint32_t)x;
                            lookback[x - xa] = (((v & 0xFFFFu) - x32) & ring_mask) | ((((v >> 16) - x32) & ring_mask) << 16);
                        }
                        if ((q = hipMemcpyAsync(c.ps.as<uint32_t>() + xa, lookback.data(), (b - xa) * 4,
                                                hipMemcpyHostToDevice, c.up)) != hipSuccess) return q;
                        if ((q = hipStreamSynchronize(c.up)) != hipSuccess) return q;      /* pageable source */
                    }
                }
                /* xval [x_sent, x_done): the part below b sits in the previous chunk's slot */
                if (x_sent[d] < b && x_done > x_sent[d]) {
                    const size_t hi = x_done < b ? x_done : b;
                    if ((q = hipMemcpyAsync(c.xval.as<uint32_t>() + x_sent[d], rebase32(xv_slot(ci - 1), ci - 1) + x_sent[d],
                                            (hi - x_sent[d]) * 4, hipMemcpyHostToDevice, c.up)) != hipSuccess) return q;
                }
                if (x_done > b) {
                    const size_t lo = x_sent[d] > b ? x_sent[d] : b;
                    if ((q = hipMemcpyAsync(c.xval.as<uint32_t>() + lo, rebase32(xv_slot(ci), ci) + lo, (x_done - lo) * 4,
                                            hipMemcpyHostToDevice, c.up)) != hipSuccess) return q;
                }
                if (ntok_sz > tok_before &&
                    (q = hipMemcpyAsync(c.chain.as<uint32_t>() + tok_before, ch_slot(ci), (ntok_sz - tok_before) * 4,
                                        hipMemcpyHostToDevice, c.up)) != hipSuccess) return q;
                if ((q = hipEventRecord(c.chunk_ev[3 * ci + 2], c.up)) != hipSuccess) return q;
                if ((q = hipStreamWaitEvent(tstream(d), c.chunk_ev[3 * ci + 2], 0)) != hipSuccess) return q;
                if ((q = hipEventRecord(c.tok_ev[2 * ci], tstream(d))) != hipSuccess) return q;
                if ((q = lz77k_xfer_index(c.ps.as<uint32_t>(), c.xval.as<uint32_t>(), xa, (uint32_t)x_done, dbase, (uint32_t)e,
                                          c.ofs.as<uint32_t>(), c.ent.as<uint2>(), c.scantmp.p, tstream(d),
                                          (uint32_t)x_new, c.flag.as<unsigned long long>() + 1)) != hipSuccess) return q;
                if ((q = lz77k_tokens(c.in.as<uint8_t>(), n32, g, c.chain.as<uint32_t>() + tok_sent[d], (uint32_t)(ntok_sz - tok_sent[d]),
                                      c.maxlen.as<uint8_t>(), c.ofs.as<uint32_t>(), c.ent.as<uint2>(), dbase, (uint32_t)b, (uint32_t)e,
                                      c.tokval.as<uint32_t>() + tok_sent[d], c.tstart.as<uint32_t>(), c.bidx.p, tvariant, tstream(d),
                                      &c.tie_ev[2 * ci], keep_ranks ? c.ranks_all.as<uint32_t>() : nullptr)) != hipSuccess) return q;
                tie_timed[ci] = ntok_sz > tok_sent[d];
                if ((q = hipEventRecord(c.tok_ev[2 * ci + 1], tstream(d))) != hipSuccess) return q;
                return tstream(d) == c.tok ? hipSuccess : hipStreamWaitEvent(c.tok, c.tok_ev[2 * ci + 1], 0);
            };
            he = enqueue();
            if (he != hipSuccess) { err = LZ77X_E_HIP; snprintf(g_err, sizeof g_err, "chunk enqueue: %s", hipGetErrorString(he)); break; }
            x_sent[d] = x_done;
            tok_sent[d] = ntok_sz;
            toks_at[ci + 1] = ntok_sz;
            if (ring_h2d && ci >= 1) {
                /* chunk ci-1's copy was queued one chunk ago: by now it has landed; tell the recurrence */
                he = hipEventSynchronize(cs[owner[ci - 1]]->chunk_ev[3 * (ci - 1) + 2]);
                if (he != hipSuccess) { err = LZ77X_E_HIP; snprintf(g_err, sizeof g_err, "h2d sync: %s", hipGetErrorString(he)); break; }
                { std::lock_guard<std::mutex> lk(pm); h2d_done = ci; }
                pcv.notify_all();
            }
        }
        if (err == LZ77X_OK && launches0) HIPCHK(hipEventRecord(c0.ev[1], kstream(0)));
        { std::lock_guard<std::mutex> lk(pm); abort_all = err != LZ77X_OK; }
        pcv.notify_all();
        prio_thread.join();
        if (prio_err.load()) err = LZ77X_E_HIP;
        ntok = (uint32_t)ntok_sz;
        nchunks_done = nchunks;
        launches0_total = launches0;
        if (err != LZ77X_OK) return err;                       /* stage_guard drains the devices */
        g_stats.host_chain_ms = t_chain;
        g_stats.host_stageb_ms = t_prio;
        if (trace_on())
            fprintf(stderr, "[lz77x] recurrence starts %.2f ms into the call, ends at %.2f; chunk loop done at %.2f\n",
                    t_prio_first - t_begin, t_prio_last - t_begin, now_ms() - t_begin);

        /* hand-overs were counted by the index kernels (each eviction once, by the shard that first saw it) */
        for (uint32_t d = 0; d < D; d++) {
            Ctx &c = *cs[d];
            unsigned long long cnt = 0;
            HIPCHK(hipSetDevice(c.device));
            HIPCHK(hipMemcpyAsync(&cnt, c.flag.as<unsigned long long>() + 1, 8, hipMemcpyDeviceToHost, tstream(d)));
            HIPCHK(hipStreamSynchronize(tstream(d)));
            transfers += cnt;
        }
        HIPCHK(hipSetDevice(c0.device));

        /* -- other shards hand their token values to the first device through the host -- */
        if (D > 1) {
            if ((rc = c0.h_tok.need(((size_t)ntok + 8) * 4))) return rc;
            for (uint32_t d = 1; d < D; d++) {
                Ctx &c = *cs[d];
                if (first_chunk[d] >= first_chunk[d + 1]) continue;
                const size_t ta = toks_at[first_chunk[d]], tb = toks_at[first_chunk[d + 1]];
                HIPCHK(hipSetDevice(c.device));
                HIPCHK(hipMemcpyAsync(c0.h_tok.as<uint32_t>() + ta, c.tokval.as<uint32_t>() + ta, (tb - ta) * 4, hipMemcpyDeviceToHost, c.tok));
                const double tw = now_ms();
                HIPCHK(hipStreamSynchronize(c.tok));
                waited += now_ms() - tw;
                HIPCHK(hipSetDevice(c0.device));
                HIPCHK(hipMemcpyAsync(c0.tokval.as<uint32_t>() + ta, c0.h_tok.as<uint32_t>() + ta, (tb - ta) * 4, hipMemcpyHostToDevice, c0.tok));
            }
            HIPCHK(hipSetDevice(c0.device));
        }
        stage_guard.ok = true;
    } else {
        if ((rc = c0.tokval.need(64))) return rc;
        HIPCHK(hipEventRecord(c0.ev[0], s));
        HIPCHK(hipEventRecord(c0.ev[1], s));
    }
    *zn = stream_bytes(ntok, g.T);
    const uint64_t nwords = (*zn + 3) / 4;
    if ((rc = c0.out.need(nwords * 4 + 16))) return rc;
    HIPCHK(lz77k_pack(c0.tokval.as<uint32_t>(), ntok, g, c0.out.as<uint32_t>(), nwords, c0.tok));
    HIPCHK(hipEventRecord(c0.ev[3], c0.tok));
    HIPCHK(hipStreamWaitEvent(s, c0.ev[3], 0));       /* later work on the caller's stream sees the result */
    const double tw = now_ms();
    HIPCHK(hipStreamSynchronize(s));
    waited += now_ms() - tw;

    float ms = 0;
    {
        double match_ms = 0;
        for (uint32_t i = 0; i < launches0_total; i++) {
            HIPCHK(hipEventElapsedTime(&ms, c0.match_ev[2 * i], c0.match_ev[2 * i + 1]));
            match_ms += ms;
        }
        g_stats.k_match_ms = match_ms;
    }
    double tok_ms = 0;
    for (uint32_t ci = 0; ci < nchunks_done; ci++) {
        Ctx &c = *cs[owner[ci]];
        HIPCHK(hipEventElapsedTime(&ms, c.tok_ev[2 * ci], c.tok_ev[2 * ci + 1]));
        tok_ms += ms;
    }
    g_stats.k_token_ms = tok_ms;
    {
        double tie_ms = 0;
        for (uint32_t ci = 0; ci < nchunks_done && ci < tie_timed.size(); ci++) {
            if (!tie_timed[ci]) continue;
            Ctx &c = *cs[owner[ci]];
            HIPCHK(hipEventElapsedTime(&ms, c.tie_ev[2 * ci], c.tie_ev[2 * ci + 1]));
            tie_ms += ms;
            g_stats.token_launches++;
        }
        g_stats.k_tiebreak_ms = tie_ms;
    }
    if (sort_timed && D == 1) {
        double sort_ms = 0, walk_ms = 0;
        for (uint32_t i = 0; i < launches0_total; i++) {
            HIPCHK(hipEventElapsedTime(&ms, c0.sort_ev[4 * i], c0.sort_ev[4 * i + 3]));
            g_stats.k_sort_chunks_ms += ms;
            HIPCHK(hipEventElapsedTime(&ms, c0.sort_ev[4 * i], c0.sort_ev[4 * i + 1]));
            sort_ms += ms;
            HIPCHK(hipEventElapsedTime(&ms, c0.sort_ev[4 * i + 1], c0.sort_ev[4 * i + 2]));
            walk_ms += ms;
        }
        g_stats.k_sort_ms = sort_ms;
        g_stats.k_walk_ms = walk_ms;
    }
    g_stats.n = n;
    g_stats.zn = *zn;
    g_stats.ntok = ntok;
    g_stats.transfers = transfers;
    g_stats.total_ms = now_ms() - t_begin;
    TRACE("encode_core total", t_begin);
    g_stats.copy_ms = waited;               /* host time blocked on the device (not overlapped) */
    return LZ77X_OK;
}

/* ---------------------------------------------------------------- decode ------------ */

/* (the decoder proper -- decode_stream -- follows the sources and sinks it reads from and writes to) */

/* (st: an IDLE stream of the context's device the copies may use -- the caller has synchronised it -- or null: the context's
 * staging stream, created on first use; a stream costs a short-lived process 8 ms) */
int fetch_result(Ctx &c, uint8_t *dst, const void *d_src, size_t bytes, hipStream_t st = nullptr);
int upload_pageable(Ctx &c, uint8_t *d_dst, const uint8_t *h_src, size_t bytes, hipStream_t st = nullptr);
extern "C" uint64_t lz77x_shard_token_cut(uint64_t ntok, int shards, int d);
extern "C" void lz77x_shard_compose_tail(const uint16_t *map, int sb, const uint8_t *incoming, uint8_t *outgoing);
extern "C" void lz77x_shard_compose_tail32(const uint32_t *map, int sb, const uint8_t *incoming, uint8_t *outgoing);

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
    if (ntok64 > LZ77X_MAX_N) return LZ77X_OK;                                  /* (the range decoder takes any length on one device) */
    const uint32_t ntok = (uint32_t)ntok64;
    const size_t D = cs.size();
    if (la > 255 || LZ77X_VENV("LZ77X_DECODE_V1") || ntok < 64 * D) return LZ77X_OK;
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
        if ((rc = upload_pageable(c, c.z.as<uint8_t>() + 4, z + b0, b1 - b0))) return rc;
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
    if (!fits || n > LZ77X_MAX_N) { HIPCHK(hipSetDevice(cs[0]->device)); return LZ77X_OK; }
    uint8_t *buf = nullptr;
    const uint32_t pre = tiles ? (uint32_t)((usb + LZ77K_DEC_TILE_BYTES - 1) / LZ77K_DEC_TILE_BYTES * LZ77K_DEC_TILE_BYTES) : 0u;
    std::vector<const unsigned long long *> d_unres(D, nullptr);
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
        std::vector<std::vector<uint8_t>> incoming(D, std::vector<uint8_t>(usb, 0));
        for (size_t d = 0; d + 1 < D; d++) lz77x_shard_compose_tail32(tmap[d].data(), sb, incoming[d].data(), incoming[d + 1].data());
        /* 4t. every shard: its history in, the bytes that point somewhere gathered */
        buf = (uint8_t *)malloc(n ? (size_t)n : 1);
        if (!buf) return LZ77X_E_NOMEM;
        for (size_t d = 0; d < D; d++) {
            Ctx &c = *cs[d];
            hipError_t e = hipSetDevice(c.device);
            uint8_t *X = c.out.as<uint8_t>();
            if (e == hipSuccess) e = hipMemcpyAsync(X + pre - usb, incoming[d].data(), usb, hipMemcpyHostToDevice, c.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c.stream);                 /* (incoming[] is pageable) */
            if (e == hipSuccess) e = lz77k_dec_gather2(X, c.ptr.as<uint32_t>(), d_unres[d], pre + sh[d].n, c.stream);
            if (e != hipSuccess) { free(buf); snprintf(g_err, sizeof g_err, "HIP: %s", hipGetErrorString(e)); return LZ77X_E_HIP; }
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
    std::vector<std::vector<uint8_t>> incoming(D, std::vector<uint8_t>(usb, 0));
    for (size_t d = 0; d + 1 < D; d++) {
        Ctx &c = *cs[d];
        HIPCHK(hipSetDevice(c.device));
        HIPCHK(hipStreamSynchronize(c.stream));
        lz77x_shard_compose_tail(smap[d].data(), sb, incoming[d].data(), incoming[d + 1].data());
    }
    /* 4. every shard: incoming bytes in, tails resolved, flagged bytes patched, output to the host */
    buf = (uint8_t *)malloc(n ? (size_t)n : 1);
    if (!buf) return LZ77X_E_NOMEM;
    for (size_t d = 0; d < D; d++) {
        Ctx &c = *cs[d];
        hipError_t e = hipSetDevice(c.device);
        if (e == hipSuccess) e = hipMemcpyAsync(sh[d].P.tres0, incoming[d].data(), usb, hipMemcpyHostToDevice, c.stream);
        if (e == hipSuccess) e = lz77k_dec_segments_back(g, c.out.as<uint8_t>(), c.ptr.p, sh[d].n, sh[d].P, c.stream);
        if (e != hipSuccess) { free(buf); snprintf(g_err, sizeof g_err, "HIP: %s", hipGetErrorString(e)); return LZ77X_E_HIP; }
    }
    }
    rc = for_each_shard(D, [&](size_t d) -> int {           /* the gather: every device fetches its bytes at once */
        Ctx &c = *cs[d];
        HIPCHK(hipSetDevice(c.device));
        HIPCHK(hipStreamSynchronize(c.stream));
        return fetch_result(c, buf + o0[d], c.out.as<uint8_t>() + pre, sh[d].n);
    });
    if (rc) { free(buf); return rc; }
    HIPCHK(hipSetDevice(cs[0]->device));
    memset(&g_stats, 0, sizeof g_stats);
    g_stats.n = n;
    g_stats.zn = zn;
    g_stats.ntok = ntok;
    g_stats.total_ms = now_ms() - t_begin;
    *out = buf;
    *out_n = (size_t)n;
    *handled = 1;
    return LZ77X_OK;
}

/* caller's pageable buffer -> device through the two pinned staging slots (the counterpart of fetch_result): the copy of
 * piece k+1 into its slot runs while the DMA of piece k drains.  Returns when the last DMA has been waited for. */
int upload_pageable(Ctx &c, uint8_t *d_dst, const uint8_t *h_src, size_t bytes, hipStream_t st)
{
    const size_t piece = (size_t)16 << 20;
    int rc;
    if (!bytes) return LZ77X_OK;
    if (bytes <= 65536) { HIPCHK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice)); return LZ77X_OK; }
    if (!st) { if ((rc = need_stream(c, &Ctx::up))) return rc; st = c.up; }
    if ((rc = c.h_stage.need(2 * piece))) return rc;
    uint8_t *slot[2] = {c.h_stage.as<uint8_t>(), c.h_stage.as<uint8_t>() + piece};
    bool used[2] = {false, false};
    size_t at = 0;
    for (int k = 0; at < bytes; k++) {
        const int sl = k & 1;
        const size_t m = bytes - at < piece ? bytes - at : piece;
        if (used[sl]) HIPCHK(hipEventSynchronize(c.ev[4 + sl]));
        g_copy.copy(slot[sl], h_src + at, m);
        HIPCHK(hipMemcpyAsync(d_dst + at, slot[sl], m, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(c.ev[4 + sl], st));
        used[sl] = true;
        at += m;
    }
    HIPCHK(hipStreamSynchronize(st));
    return LZ77X_OK;
}

/* device -> caller's pageable buffer through two pinned staging slots: the DMA of piece k+1 runs
 * while the host copies piece k out (a direct hipMemcpy into pageable memory is ~2 GB/s) */
int fetch_result(Ctx &c, uint8_t *dst, const void *d_src, size_t bytes, hipStream_t st)
{
    const size_t piece = (size_t)16 << 20;
    int rc;
    if (!st) { if ((rc = need_stream(c, &Ctx::copy))) return rc; st = c.copy; }
    if ((rc = c.h_stage.need(2 * piece))) return rc;
    uint8_t *slot[2] = {c.h_stage.as<uint8_t>(), c.h_stage.as<uint8_t>() + piece};
    const uint8_t *src = reinterpret_cast<const uint8_t *>(d_src);
    size_t issued = 0, done = 0;
    int k = 0;
    if (bytes) {
        const size_t m = bytes < piece ? bytes : piece;
        HIPCHK(hipMemcpyAsync(slot[0], src, m, hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(c.ev[4], st));
        issued = m;
    }
    while (done < bytes) {
        const size_t cur = (issued - done);
        HIPCHK(hipEventSynchronize(c.ev[4 + (k & 1)]));
        if (issued < bytes) {
            const size_t m = bytes - issued < piece ? bytes - issued : piece;
            HIPCHK(hipMemcpyAsync(slot[(k + 1) & 1], src + issued, m, hipMemcpyDeviceToHost, st));
            HIPCHK(hipEventRecord(c.ev[4 + ((k + 1) & 1)], st));
            issued += m;
        }
        g_copy_out.copy(dst + done, slot[k & 1], cur);
        done += cur;
        k++;
    }
    return LZ77X_OK;
}

/* A regular file behind a FILE* can be read and written at offsets by several threads at once (CopyPool::read_at /
 * write_at: the page cache hands out 4-5 GB/s to one thread): its descriptor and position, or fd = -1 for anything else
 * (a pipe, a cookie stream such as the shim's bitFILE, a file opened for appending), which keeps fread / fwrite.  The
 * FILE's own position is put where the descriptor's work ended (raw_done). */
struct RawFile { int fd = -1; off_t off = 0; };
RawFile raw_file(FILE *f, bool writing)
{
    RawFile r;
    const int fd = fileno(f);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return r;
    if (writing) {
        if (fflush(f) != 0) return r;
        const int fl = fcntl(fd, F_GETFL);
        if (fl < 0 || (fl & O_APPEND)) return r;
    }
    const off_t at = ftello(f);
    if (at < 0) return r;
    r.fd = fd;
    r.off = at;
    return r;
}
bool raw_done(FILE *f, const RawFile &r) { return r.fd < 0 || fseeko(f, r.off, SEEK_SET) == 0; }

/* FILE* -> device buffer `dst` (grown as needed, `slack` spare bytes kept behind the data), streamed
 * through the two pinned staging slots: the fread of piece k+1 overlaps the DMA of piece k.  Host
 * memory stays at two pieces whatever the file size (the reference streams through a
 * 3*SB+LA window, lz77.c:113-129). */
int stream_in(Ctx &c, FILE *f, DevBuf &dst, size_t slack, size_t *n_out)
{
    const size_t piece = (size_t)16 << 20;
    int rc;
    if ((rc = need_stream(c, &Ctx::up))) return rc;
    if ((rc = c.h_stage.need(2 * piece))) return rc;
    uint8_t *slot[2] = {c.h_stage.as<uint8_t>(), c.h_stage.as<uint8_t>() + piece};
    size_t hint = 0;
    {
        struct stat st;
        const long at = ftell(f);
        if (at >= 0 && fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) && (size_t)st.st_size > (size_t)at)
            hint = (size_t)st.st_size - (size_t)at;
    }
    if (hint > LZ77X_MAX_N) return LZ77X_E_TOOBIG;                       /* before allocating or reading anything */
    if ((rc = dst.need((hint ? hint : piece) + slack))) return rc;
    size_t len = 0;
    bool used[2] = {false, false};
    for (int k = 0;; k++) {
        const int sl = k & 1;
        if (used[sl]) HIPCHK(hipEventSynchronize(c.ev[4 + sl]));          /* its previous DMA has drained */
        const size_t got = fread(slot[sl], 1, piece, f);
        if (got == 0) {
            if (ferror(f)) return LZ77X_E_IO;
            break;
        }
        if (len + got > LZ77X_MAX_N) return LZ77X_E_TOOBIG;
        if (len + got + slack > dst.cap) {                               /* pipe or growing file: double, keep the data */
            DevBuf bigger;
            if ((rc = bigger.need(2 * (len + got) + slack))) return rc;
            HIPCHK(hipStreamSynchronize(c.up));
            if (len) HIPCHK(hipMemcpyAsync(bigger.p, dst.p, len, hipMemcpyDeviceToDevice, c.up));
            HIPCHK(hipStreamSynchronize(c.up));
            hipError_t e0 = hipFree(dst.p); (void)e0;
            dst = bigger;
        }
        HIPCHK(hipMemcpyAsync(dst.as<uint8_t>() + len, slot[sl], got, hipMemcpyHostToDevice, c.up));
        HIPCHK(hipEventRecord(c.ev[4 + sl], c.up));
        used[sl] = true;
        len += got;
    }
    HIPCHK(hipStreamSynchronize(c.up));
    *n_out = len;
    return LZ77X_OK;
}

/* device -> FILE*, the DMA of piece k+1 overlapping the fwrite of piece k */
int stream_out(Ctx &c, FILE *f, const void *d_src, size_t bytes, hipStream_t st = nullptr /* an idle stream to copy on, or null: the staging stream */)
{
    const size_t piece = (size_t)16 << 20;
    int rc;
    if (!st) { if ((rc = need_stream(c, &Ctx::copy))) return rc; st = c.copy; }
    if ((rc = c.h_stage.need(2 * piece))) return rc;
    uint8_t *slot[2] = {c.h_stage.as<uint8_t>(), c.h_stage.as<uint8_t>() + piece};
    const uint8_t *src = reinterpret_cast<const uint8_t *>(d_src);
    size_t issued = 0, done = 0;
    int k = 0;
    RawFile raw = raw_file(f, true);
    if (bytes) {
        const size_t m = bytes < piece ? bytes : piece;
        HIPCHK(hipMemcpyAsync(slot[0], src, m, hipMemcpyDeviceToHost, st));
        HIPCHK(hipEventRecord(c.ev[4], st));
        issued = m;
    }
    while (done < bytes) {
        const size_t cur = issued - done;
        HIPCHK(hipEventSynchronize(c.ev[4 + (k & 1)]));
        if (issued < bytes) {
            const size_t m = bytes - issued < piece ? bytes - issued : piece;
            HIPCHK(hipMemcpyAsync(slot[(k + 1) & 1], src + issued, m, hipMemcpyDeviceToHost, st));
            HIPCHK(hipEventRecord(c.ev[4 + ((k + 1) & 1)], st));
            issued += m;
        }
        bool ok;
        const double tw = trace_on() ? now_ms() : 0;
        if (raw.fd >= 0) {
            ok = g_copy_out.write_at(raw.fd, slot[k & 1], cur, raw.off) == (ssize_t)cur;
            raw.off += (off_t)cur;
        } else {
            ok = fwrite(slot[k & 1], 1, cur, f) == cur;
        }
        if (trace_on()) g_fwrite_ms += now_ms() - tw;
        if (!ok) { hipError_t e0 = hipStreamSynchronize(st); (void)e0; return LZ77X_E_IO; }
        done += cur;
        k++;
    }
    if (!raw_done(f, raw)) return LZ77X_E_IO;
    return fflush(f) == 0 ? LZ77X_OK : LZ77X_E_IO;
}


/* ---------------------------------------------------------------- device-resident encode ------------ */

/* Where the input comes from and where the stream goes: device memory, host memory or a FILE*.  Both are
 * strictly sequential (a pipe works), which is what lets an input of any size run through a bounded
 * amount of device memory (the reference streams through 3*SB+LA bytes, lz77.c:113-129). */
struct Source {
    virtual ~Source() {}
    /* up to `want` bytes to device address d_dst, enqueued on / ordered with stream s; fewer only at the end */
    virtual int read(Ctx &c, uint8_t *d_dst, size_t want, hipStream_t s, size_t *got) = 0;
    virtual size_t size_hint() const { return 0; }            /* bytes still to come, when known */
    /* the bytes come out of host memory or a file: reading them keeps a host thread busy (an encode then loads its next
     * segment from a thread of its own, beside the recurrence of the current one) */
    virtual bool host_backed() const { return false; }
};
struct Sink {
    virtual ~Sink() {}
    /* the next `bytes` of the stream, resident at d_src and complete in stream order on s */
    virtual int write(Ctx &c, const uint8_t *d_src, size_t bytes, hipStream_t s) = 0;
    /* host memory for the next `bytes` of the stream, to be filled by the caller in any order (several devices fetch
     * their pieces at once); null when the sink only takes bytes in sequence */
    virtual uint8_t *direct(size_t bytes) { (void)bytes; return nullptr; }
    /* a sink of fixed capacity that has been offered more than it holds: only the count matters from here on */
    virtual bool overflowed() const { return false; }
    /* the sink copies what it is handed out of device memory itself and blocks on the host while it does (a file, host
     * memory): a decode of several ranges hands such a sink its ranges from a thread of its own (RangeDrain) */
    virtual bool blocks_on_host() const { return false; }
    size_t total = 0;
};

struct MemSource : Source {
    const uint8_t *p; size_t n, at = 0; bool on_device;
    MemSource(const void *src, size_t bytes, bool dev) : p(reinterpret_cast<const uint8_t *>(src)), n(bytes), on_device(dev) {}
    int read(Ctx &c, uint8_t *d_dst, size_t want, hipStream_t s, size_t *got) override
    {
        const size_t m = n - at < want ? n - at : want;
        if (m && on_device) HIPCHK(hipMemcpyAsync(d_dst, p + at, m, hipMemcpyDeviceToDevice, s));
        else if (m) {
            /* pageable memory: through the pinned slots, the copies cut over a few threads (upload_pageable) */
            HIPCHK(hipStreamSynchronize(s));               /* d_dst may still be read by the kernels of the segment before */
            int rc = upload_pageable(c, d_dst, p + at, m, s);
            if (rc) return rc;
        }
        at += m;
        *got = m;
        return LZ77X_OK;
    }
    size_t size_hint() const override { return n - at; }
    bool host_backed() const override { return !on_device; }
};

struct FileSource : Source {
    FILE *f;
    explicit FileSource(FILE *file) : f(file) {}
    bool host_backed() const override { return true; }
    size_t size_hint() const override
    {
        /* regular files only (a pipe has no size): what lies between the read position and the end */
        struct stat st;
        const int fd = fileno(f);
        if (fd < 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return 0;
        const off_t at = ftello(f);
        return at >= 0 && st.st_size > at ? (size_t)(st.st_size - at) : 0;
    }
    int read(Ctx &c, uint8_t *d_dst, size_t want, hipStream_t s, size_t *got) override
    {
        /* fread of piece k+1 overlaps the DMA of piece k (two pinned staging slots) */
        const size_t piece = (size_t)16 << 20;
        int rc;
        if ((rc = c.h_stage.need(2 * piece))) return rc;
        uint8_t *slot[2] = {c.h_stage.as<uint8_t>(), c.h_stage.as<uint8_t>() + piece};
        HIPCHK(hipStreamSynchronize(s));                       /* d_dst may still be read by the previous segment's kernels; s is idle from
                                                                  here on and carries the copies itself (no stream of their own) */
        size_t len = 0;
        bool used[2] = {false, false};
        RawFile raw = raw_file(f, false);                      /* a regular file: its pieces are read by several threads at once */
        for (int k = 0; len < want; k++) {
            const int sl = k & 1;
            if (used[sl]) HIPCHK(hipEventSynchronize(c.ev[4 + sl]));
            const size_t ask = want - len < piece ? want - len : piece;
            size_t m;
            const double tr = trace_on() ? now_ms() : 0;
            if (raw.fd >= 0) {
                const ssize_t r = g_copy.read_at(raw.fd, slot[sl], ask, raw.off);
                if (r < 0) return LZ77X_E_IO;
                m = (size_t)r;
                raw.off += (off_t)m;
                if (!raw_done(f, raw)) return LZ77X_E_IO;      /* (the FILE follows: ftello / a later fread see what was consumed) */
            } else {
                m = fread(slot[sl], 1, ask, f);
            }
            if (trace_on()) g_fread_ms += now_ms() - tr;
            if (m == 0) {
                if (raw.fd < 0 && ferror(f)) return LZ77X_E_IO;
                break;
            }
            HIPCHK(hipMemcpyAsync(d_dst + len, slot[sl], m, hipMemcpyHostToDevice, s));
            HIPCHK(hipEventRecord(c.ev[4 + sl], s));
            used[sl] = true;
            len += m;
            if (m < ask) break;
        }
        HIPCHK(hipStreamSynchronize(s));
        *got = len;
        return LZ77X_OK;
    }
};

struct DeviceSink : Sink {
    uint8_t *d_out; size_t cap;
    DeviceSink(void *out, size_t capacity) : d_out(reinterpret_cast<uint8_t *>(out)), cap(capacity) {}
    int write(Ctx &, const uint8_t *d_src, size_t bytes, hipStream_t s) override
    {
        if (total + bytes <= cap && bytes) HIPCHK(hipMemcpyAsync(d_out + total, d_src, bytes, hipMemcpyDeviceToDevice, s));
        total += bytes;                                        /* past cap: keep counting, the caller reports the need */
        return LZ77X_OK;
    }
    bool overflowed() const override { return total > cap; }
};

struct HostSink : Sink {
    uint8_t *buf = nullptr; size_t cap = 0;
    ~HostSink() override { free(buf); }
    int write(Ctx &c, const uint8_t *d_src, size_t bytes, hipStream_t s) override
    {
        if (total + bytes > cap) {
            size_t ncap = cap ? cap : (size_t)1 << 20;
            while (ncap < total + bytes) ncap *= 2;
            uint8_t *nb = (uint8_t *)realloc(buf, ncap);
            if (!nb) return LZ77X_E_NOMEM;
            buf = nb;
            cap = ncap;
        }
        HIPCHK(hipStreamSynchronize(s));
        const int rc = fetch_result(c, buf + total, d_src, bytes, s);
        total += bytes;
        return rc;
    }
    uint8_t *direct(size_t bytes) override
    {
        if (total + bytes > cap) {
            size_t ncap = cap ? cap : (size_t)1 << 20;
            while (ncap < total + bytes) ncap *= 2;
            uint8_t *nb = (uint8_t *)realloc(buf, ncap);
            if (!nb) return nullptr;
            buf = nb;
            cap = ncap;
        }
        uint8_t *at = buf + total;
        total += bytes;
        return at;
    }
    uint8_t *release() { uint8_t *b = buf; buf = nullptr; return b ? b : (uint8_t *)malloc(1); }
    bool blocks_on_host() const override { return true; }
};

struct FileSink : Sink {
    FILE *f;
    explicit FileSink(FILE *file) : f(file) {}
    int write(Ctx &c, const uint8_t *d_src, size_t bytes, hipStream_t s) override
    {
        HIPCHK(hipStreamSynchronize(s));
        total += bytes;
        return stream_out(c, f, d_src, bytes, s);
    }
    bool blocks_on_host() const override { return true; }
};

/* ---------------------------------------------------------------- decode ------------ */

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
                   uint8_t **d_bytes, uint32_t *rounds_out, DevBuf &outb)
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
        HIPCHK(lz77k_dec_segments_front(c.tokval.as<uint32_t>(), c.dst.as<uint32_t>(), ntok, g, X, c.ptr.p, n, c.tstart.p, s, ext0, P, nullptr));
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

int ctx_sibling(Ctx &c, Ctx **out);

/* A decode of several ranges: the bytes of range r leave -- D2H through pinned slots, then fwrite / pwrite or a copy into the
 * caller's buffer: host work, a quarter of a second per gigabyte -- while range r + 1 is read, parsed and resolved (two
 * ranges in flight: two output buffers, the sibling context's stream, slots and events for the drain).  One thread, first
 * in first out, so a sink sees its bytes in order. */
struct RangeDrain {
    struct Job { const uint8_t *d = nullptr; size_t n = 0; hipEvent_t ready = nullptr; };
    Sink *sink = nullptr;
    Ctx *dc = nullptr;                    /* the context whose stream / staging slots the drain uses */
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q;
    uint64_t submitted = 0, done = 0;
    int rc = LZ77X_OK;
    char err[256] = "";
    bool stop = false, started = false;
    bool abort = false;                   /* set by the destructor unless everything submitted was waited for: queued jobs are dropped */
    void run()
    {
        hipError_t e = hipSetDevice(dc->device);
        (void)e;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                j = q.front();
                q.pop_front();
            }
            int r = LZ77X_OK;
            g_err[0] = 0;
            bool drop;
            { std::lock_guard<std::mutex> lk(mu); drop = abort; }
            if (rc == LZ77X_OK && !drop) {
                if (hipStreamWaitEvent(dc->stream, j.ready, 0) != hipSuccess) r = LZ77X_E_HIP;
                else r = sink->write(*dc, j.d, j.n, dc->stream);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (r != LZ77X_OK && rc == LZ77X_OK) { rc = r; snprintf(err, sizeof err, "%s", g_err); }
                done++;
            }
            cv.notify_all();
        }
    }
    int start(Sink *sk, Ctx *drain_ctx)
    {
        sink = sk;
        dc = drain_ctx;
        try { th = std::thread(&RangeDrain::run, this); started = true; }
        catch (...) { return LZ77X_E_NOMEM; }
        return LZ77X_OK;
    }
    /* the bytes [d, d + n) are complete once `ready` has passed */
    int submit(const uint8_t *d, size_t n, hipEvent_t ready)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (rc != LZ77X_OK) { snprintf(g_err, sizeof g_err, "%s", err); return rc; }
        q.push_back(Job{d, n, ready});
        submitted++;
        cv.notify_all();
        return LZ77X_OK;
    }
    /* until at most `in_flight` submitted jobs are unfinished */
    int wait(uint64_t in_flight)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return submitted - done <= in_flight; });
        if (rc != LZ77X_OK) snprintf(g_err, sizeof g_err, "%s", err);
        return rc;
    }
    ~RangeDrain()
    {
        if (!started) return;
        /* a caller that leaves on an error has not waited for its jobs: no more output after a failed call (and no
         * long pwrite before the error returns).  On the success path wait(0) has emptied the queue. */
        { std::lock_guard<std::mutex> lk(mu); stop = true; if (submitted != done) abort = true; }
        cv.notify_all();
        th.join();
    }
};

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
    HIPCHK(hipMemcpyAsync(hdr, c.z.p, 4, hipMemcpyDeviceToHost, s));
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
        if ((rc = c.tokval.need(((size_t)ntok + 8) * 4))) return rc;
        if ((rc = c.len1.need(((size_t)ntok + 8) * 4))) return rc;
        if ((rc = c.dst.need(((size_t)ntok + 8) * 4))) return rc;
        if ((rc = c.scantmp.need(lz77k_scan_tmp_bytes(ntok + 1)))) return rc;
        HIPCHK(hipMemsetAsync(c.flag.as<uint32_t>() + 8, 0, 8, s));
        HIPCHK(lz77k_dec_parse(zr.as<uint8_t>(), ntok, g, c.tokval.as<uint32_t>(), c.len1.as<uint32_t>(), s, c.flag.as<uint32_t>() + 8));
        HIPCHK(hipMemsetAsync(c.len1.as<uint32_t>() + ntok, 0, 4, s));
        HIPCHK(lz77k_scan_u32(c.len1.as<uint32_t>(), c.dst.as<uint32_t>(), ntok + 1, c.scantmp.p, s));     /* ntok << lb fits 32 bits (dec_range_plan) */
        uint32_t *tot = reinterpret_cast<uint32_t *>(hdr + 16);
        HIPCHK(hipMemcpyAsync(tot, c.dst.as<uint32_t>() + ntok, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(tot + 1, c.flag.as<uint32_t>() + 8, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        uint32_t n = tot[0], use = ntok;
        const bool stale = tot[1] != 0;             /* the range copies from distance 0 somewhere (power-of-two -s) */
        /* distances beyond the window, or a lookahead field no CLI run can produce (main.c:103 caps -l at 255; a token
         * may then span several tiles): the per-byte pointer path, which assumes nothing about either */
        const bool general = tot[2] != 0 || la > 255;
        if (n > cap) {
            HIPCHK(lz77k_dec_cut(c.dst.as<uint32_t>(), ntok, cap, c.flag.as<uint32_t>() + 12, s));
            HIPCHK(hipMemcpyAsync(tot + 4, c.flag.as<uint32_t>() + 12, 8, hipMemcpyDeviceToHost, s));
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
            if ((rc = decode_resolve(c, g, use, n, s, stale, general, have_k ? &K : nullptr, &d_bytes, &rounds, ob))) return rc;
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

/* What one segment hands to the next (host side): where the parse chain continues, how many tokens are
 * out, the last tokens (a stream word can straddle the boundary), and the priorities of the sb cells that are
 * live at the boundary, renumbered 0..sb-1 in order (the tie-break only ever compares priorities; every
 * position of the next segment is newer than all of them). */
struct SegCarry {
    bool first = true;
    uint64_t chain_pos = 0;        /* global position of the next token */
    uint64_t ntok = 0;
    uint32_t tail[4] = {0, 0, 0, 0};
    uint32_t ntail = 0;
    std::vector<uint32_t> cells;   /* sb ranks (after the first segment) */
};

/* One segment in flight.  The bytes c->in[0, nloc) are the input from global position gpos0 on; its tokens are
 * the chain positions in [start, E) (local).  Everything is computed in local 32-bit coordinates, in four
 * phases so that two segments can be in flight on two context sets of the same device:
 *     seg_front   match stage over [0, cover): needs nothing from the segment before
 *     seg_mid     parse chain from `start` | priority recurrence over steps [0, E-sb) from the carried cells
 *                 (the host drives the gate iteration) -> the carry of the two sequential loops
 *     seg_tokens  hand-over index + tie-break + the stream words this segment's tokens start in (enqueue only)
 *     seg_finish  wait, last tokens to the carry, the words to the sink, timings */
struct SegJob {
    Ctx *c = nullptr;
    hipStream_t s = nullptr;
    uint64_t gpos0 = 0;
    uint32_t nloc = 0, cover = 0;
    bool last = false, first = false;
    uint32_t start = 0, E = 0;
    uint32_t nx = 0, nlook = 0, ntok = 0, exit_off = 0;
    uint32_t nregions = 0, launches = 0, nchunks = 0, nsub = 0;
    uint32_t *d_order = nullptr;
    int tvariant = 0;
    uint64_t K0 = 0;
    uint32_t have_tail = 0, ntail_in = 0;
    uint64_t out_bytes = 0;
    size_t scratch_cap = 0;          /* match-stage scratch per launch (0: the default of the geometry) */
    size_t token_chunk = (size_t)128 << 20;   /* positions per token launch (LZ77X_TOKEN_CHUNK, read once per call) */
    std::vector<char> tie_timed;
};

int seg_front(SegJob &J, const lz77x_geom &g)
{
    Ctx &c = *J.c;
    hipStream_t s = J.s;
    int rc;
    HIPCHK(lz77k_fill_pad(c.in.as<uint8_t>(), J.nloc, s));
    J.launches = 0;
    uint32_t nregions = (uint32_t)(((size_t)J.cover + g.TILE - 1) / g.TILE);
    {
        const uint32_t all = (uint32_t)(((size_t)J.nloc + g.TILE - 1) / g.TILE);
        if (nregions > all) nregions = all;
    }
    J.nregions = nregions;
    const char *tv = LZ77X_VENV("LZ77X_TOKEN_VARIANT");
    J.tvariant = tv ? atoi(tv) : 0;
    J.d_order = nullptr;
    if (!nregions) {
        HIPCHK(hipEventRecord(c.ev[0], s));
        HIPCHK(hipEventRecord(c.ev[1], s));
        return LZ77X_OK;
    }
    uint32_t batch = nregions;
    {
        /* 3 GB of scratch: 100 MB at C1 (8138 regions of 287 KB) in ONE launch -- the walkers are latency bound (a
         * launch takes its fill + 2048 steps whatever its size), a second launch is a second 0.75 ms */
        const size_t per = lz77k_match_scratch_bytes(g, 1);
        /* (large windows: 16 GB -- their walkers are latency bound too and a region's scratch is 16x a small window's) */
        const size_t cap = J.scratch_cap ? J.scratch_cap : (size_t)(g.fast ? 3 : 16) << 30;      /* (encode_mem_plan lowers it on a tight device) */
        const uint32_t fit = (uint32_t)(cap / per);
        if (batch > fit) batch = fit ? fit : 1;
        const char *gs = getenv("LZ77X_MATCH_BATCH");
        if (gs && atoi(gs) > 0 && (uint32_t)atoi(gs) < batch) batch = (uint32_t)atoi(gs);
    }
    const size_t np = (size_t)J.nloc;
    if ((rc = c.scratch.need(lz77k_match_scratch_bytes(g, batch)))) return rc;
    if ((rc = c.ps.need((np + 8) * 4))) return rc;
    if ((rc = c.maxlen.need(np + 64))) return rc;
    /* the regions' sorted order stays resident for the tie-break (RP uint16 per region: 2.7 B per input byte) */
    const bool keep_order = J.tvariant == 0 && !(g.fast && LZ77X_VENV("LZ77X_TOKENS_BUCKET"));
    /* large windows: rank + inverse arrays, (2RP + 8) words per region, for the rank-order tie-break */
    if (keep_order && (rc = c.ranks_all.need(g.fast ? (size_t)nregions * g.RP * 2 + 64 : (size_t)nregions * (2 * (size_t)g.RP + 8) * sizeof(uint32_t)))) return rc;
    J.d_order = keep_order ? c.ranks_all.as<uint32_t>() : nullptr;
    /* large windows: the (block, first byte) buckets the tokens of length one are resolved from (built per token chunk) */
    if (!g.fast) {
        size_t chunk_pos = J.token_chunk;
        chunk_pos += lz77k_chain_sub();
        if ((rc = c.bidx.need(lz77k_tokens_index_bytes(g, chunk_pos < (size_t)J.nloc ? chunk_pos : (size_t)J.nloc)))) return rc;
    }
    const uint32_t nlaunch = (nregions + batch - 1) / batch;
    while (c.sort_ev.size() < 4 * (size_t)nlaunch) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c.sort_ev.push_back(e); }
    while (c.match_ev.size() < 8) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c.match_ev.push_back(e); }
    /* -- match stage (replaces tree.c insert/delete/find): ps[], maxlen[] -- */
    HIPCHK(hipEventRecord(c.ev[0], s));
    for (uint32_t r0 = 0; r0 < nregions; r0 += batch) {
        const uint32_t nr = nregions - r0 < batch ? nregions - r0 : batch;
        HIPCHK(lz77k_match(c.in.as<uint8_t>(), J.nloc, g, r0, nr, c.ps.as<uint32_t>(), c.maxlen.as<uint8_t>(), c.scratch.p, 0, s,
                           &c.sort_ev[4 * J.launches], J.d_order));
        J.launches++;
    }
    HIPCHK(hipEventRecord(c.ev[1], s));
    g_stats.match_launches += J.launches;
    return LZ77X_OK;
}

/* *fallback: the gate iteration gave up (only possible when allow_fallback), nothing was emitted. */
int seg_mid(SegJob &J, const lz77x_geom &g, SegCarry &carry, bool allow_fallback, bool *fallback, double *waited)
{
    Ctx &c = *J.c;
    hipStream_t s = J.s;
    int rc;
    *fallback = false;
    const size_t usb = (size_t)g.sb;
    const uint32_t start = J.start, E = J.E;
    J.first = carry.first;
    J.nlook = J.first ? 0u : (uint32_t)g.sb;
    J.nx = E > (uint32_t)g.sb ? E - (uint32_t)g.sb : 0u;
    J.ntok = 0;
    J.exit_off = 0;
    J.K0 = carry.ntok;
    if ((rc = c.h_small.need(128))) return rc;
    if (E > start) {
        const uint32_t csub = lz77k_chain_sub();
        const size_t np = (size_t)J.nloc, span = (size_t)E - start;
        if ((rc = c.xval.need((np + 8) * 4))) return rc;
        if ((rc = c.chain.need((np + 8) * 4))) return rc;
        if ((rc = c.flag.need(1024))) return rc;          /* [64, 64 + 8 * 33): the hand-over count and the slots the tiles spread it over */
        if ((rc = c.prio_tmp.need(lz77k_prio_tmp_bytes(J.nx, g.sb)))) return rc;
        if ((rc = c.chain_tmp.need(lz77k_chain_tmp_bytes(E - start, g.la)))) return rc;
        if ((rc = c.look.need((size_t)2 * (usb + 8) * 4))) return rc;
        const uint32_t nsub_max = (uint32_t)((span + csub - 1) / csub);
        if ((rc = c.h_tbase.need(((size_t)nsub_max + 2) * 4 + (usb + 8) * 4))) return rc;
        HIPCHK(hipMemsetAsync(c.flag.p, 0, 1024, s));
        /* c.look: [0, sb) the cells this segment starts from, [sb+8, ..) the cells it leaves behind */
        uint32_t *look_cur = c.look.as<uint32_t>(), *look_next = look_cur + usb + 8;
        uint32_t *h_tbase = c.h_tbase.as<uint32_t>(), *h_state = h_tbase + nsub_max + 2;
        if (!J.first) {
            memcpy(h_state, carry.cells.data(), usb * 4);
            HIPCHK(hipMemcpyAsync(look_cur, h_state, usb * 4, hipMemcpyHostToDevice, s));
        }

        /* -- parse chain (lz77.c:98) over [start, E): needs maxlen[] only and nothing needs it before the tie-break;
         *    LZ77X_CHAIN_STREAM=1 runs it on a stream of its own beside the recurrence -- */
        const uint32_t *d_tbase = nullptr, *d_exit = nullptr;
        uint32_t nsub = 0;
        if (LZ77X_VENV("LZ77X_CHAIN_STREAM") && (rc = need_stream(c, &Ctx::tok))) return rc;
        hipStream_t sc = LZ77X_VENV("LZ77X_CHAIN_STREAM") ? c.tok : s;   /* measured: beside the recurrence it costs the recurrence more (6.0 -> 6.5 ms) than it hides (0.4) */
        if (sc != s) HIPCHK(hipStreamWaitEvent(sc, c.ev[1], 0));               /* the match stage is through */
        HIPCHK(hipEventRecord(c.match_ev[0], sc));
        HIPCHK(lz77k_chain(c.maxlen.as<uint8_t>(), E, g.la, c.chain.as<uint32_t>(), c.chain_tmp.p, sc, &d_tbase, &nsub, start, &d_exit));
        HIPCHK(hipEventRecord(c.match_ev[1], sc));
        HIPCHK(hipMemcpyAsync(h_tbase, d_tbase, ((size_t)nsub + 1) * 4, hipMemcpyDeviceToHost, sc));
        HIPCHK(hipMemcpyAsync(c.h_small.as<uint32_t>() + 16, d_exit, 4, hipMemcpyDeviceToHost, sc));
        HIPCHK(hipEventRecord(c.pipe_ev[2], sc));
        J.nsub = nsub;

        /* -- priority recurrence (tree.c:202-231) over steps [0, nx) from the carried cells -- */
        int iters = 0, converged = 1;
        /* a sweep finalises at least one more block: it always ends.  The guard before the host loop takes over comes from
         * what was measured (profiles/r03_prio_classes.json, tools/time_c2.py per data class): 4-7 iterations on every class at
         * C1, 9-13 at C2 except record-structured data (31: the flips decay slowly but steadily, and the seven iterations
         * past 24 are cheaper than starting over on the host).  Twice the largest count seen; an iteration costs 1/20
         * (C2) to 1/100 (C1, text) of the host loop, so a pathological input is bounded at about three times its cost */
        int max_iters = allow_fallback ? 64 : 1 << 30;
        {
            const char *me = getenv("LZ77X_PRIO_MAX_ITERS");
            if (me && atoi(me) > 0 && allow_fallback) max_iters = atoi(me);
        }
        HIPCHK(hipEventRecord(c.match_ev[2], s));
        const double tw0 = now_ms();
        float prio_ms3[3] = {0, 0, 0};
        HIPCHK(lz77k_prio(c.ps.as<uint32_t>(), J.nx, g.sb, c.xval.as<uint32_t>(), c.prio_tmp.p, s, c.h_small.as<uint32_t>() + 8, max_iters,
                          &iters, &converged, &c.match_ev[4], prio_ms3, 0u, J.first ? nullptr : look_cur, J.last ? nullptr : look_next));
        HIPCHK(hipEventRecord(c.match_ev[3], s));
        if (!J.last) HIPCHK(hipMemcpyAsync(h_state, look_next, usb * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));                       /* (nx == 0: the recurrence did not sync) */
        HIPCHK(hipEventSynchronize(c.pipe_ev[2]));             /* tbase has landed */
        HIPCHK(hipStreamWaitEvent(s, c.pipe_ev[2], 0));        /* chain[] is there for the tie-break */
        *waited += now_ms() - tw0;
        g_stats.k_prio_fwd_ms += prio_ms3[0];
        g_stats.k_prio_back_ms += prio_ms3[1];
        g_stats.k_prio_scan_ms += prio_ms3[2];
        g_stats.prio_iters += (uint32_t)iters;
        if (!converged) {
            *fallback = true;
            return LZ77X_OK;
        }
        J.ntok = h_tbase[nsub];
        J.exit_off = c.h_small.as<uint32_t>()[16];
        if (!J.last) {
            /* the cells left live, renumbered by rank (sb values): what the next segment starts from */
            std::vector<std::pair<uint32_t, uint32_t>> order(usb);
            for (size_t i = 0; i < usb; i++) order[i] = {h_state[i], (uint32_t)i};
            std::sort(order.begin(), order.end());
            carry.cells.resize(usb);
            for (size_t r = 0; r < usb; r++) carry.cells[order[r].second] = (uint32_t)r;
        }
    }
    carry.first = false;
    carry.ntok = J.K0 + J.ntok;
    carry.chain_pos = J.gpos0 + E + J.exit_off;
    return LZ77X_OK;
}

/* enqueue only; carry.tail is the predecessor's (its seg_finish has run) */
int seg_tokens(SegJob &J, const lz77x_geom &g, const SegCarry &carry)
{
    Ctx &c = *J.c;
    hipStream_t s = J.s;
    int rc;
    const size_t usb = (size_t)g.sb;
    const uint32_t start = J.start, E = J.E, ntok = J.ntok;
    J.ntail_in = carry.ntail;
    J.nchunks = 0;
    if (E > start) {
        const uint32_t csub = lz77k_chain_sub();
        /* token chunks: up to 128M positions (one hand-over index and one tie-break launch each; the index
         * costs 12 bytes of scratch per position), a multiple of the chain sub-block, counted from `start` */
        size_t chunk_pos = (J.token_chunk + csub - 1) / csub * csub;
        const size_t span = (size_t)E - start, np = (size_t)J.nloc;
        const uint32_t nchunks = (uint32_t)((span + chunk_pos - 1) / chunk_pos);
        J.nchunks = nchunks;
        const size_t idx_span = (chunk_pos < span ? chunk_pos : span) + 2 * usb + 16;
        if ((rc = c.tokval.need((np + 16) * 4))) return rc;
        if ((rc = c.ofs.need((idx_span + 8) * 4))) return rc;
        if ((rc = c.ent.need((idx_span + 8) * 8))) return rc;
        if ((rc = c.scantmp.need(lz77k_scan_tmp_bytes((uint32_t)idx_span + 1)))) return rc;
        if ((rc = c.tstart.need(lz77k_tokens_tmp_bytes((uint32_t)idx_span, g)))) return rc;
        while (c.tie_ev.size() < 2 * (size_t)nchunks + 8) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c.tie_ev.push_back(e); }
        uint32_t *look_cur = c.look.as<uint32_t>();
        const uint32_t *h_tbase = c.h_tbase.as<uint32_t>();
        /* -- tokens: per chunk, the hand-over index of the evictions that can matter and the tie-break.  Tokens
         *    land behind four slots that hold the predecessor's last tokens (for the first stream word) -- */
        uint32_t *tokbuf = c.tokval.as<uint32_t>();
        if (carry.ntail) HIPCHK(hipMemcpyAsync(tokbuf + 4 - carry.ntail, carry.tail + 4 - carry.ntail, carry.ntail * 4, hipMemcpyHostToDevice, s));
        J.tie_timed.assign(nchunks, 0);
        HIPCHK(hipEventRecord(c.ev[2], s));
        for (uint32_t ci = 0; ci < nchunks; ci++) {
            const size_t b = start + (size_t)ci * chunk_pos, e = b + chunk_pos < E ? b + chunk_pos : E;
            const uint32_t ta = h_tbase[(b - start) / csub], tb = e == E ? ntok : h_tbase[(e - start) / csub];
            const size_t x_done = e > usb ? e - usb : 0;
            const uint32_t dbase = b > usb ? (uint32_t)(b - usb) : 0u;
            const uint32_t xa = dbase > (uint32_t)g.sb ? dbase - (uint32_t)g.sb : 0u;
            const size_t x_new = ci == 0 ? 0 : (b > usb ? b - usb : 0);
            /* (destination blocks build their lists in LDS from the evictions of the sb positions before them: a 9-fold
             * re-read at sb = 65535; large windows count and place through HBM instead) */
            /* (LDS-sized windows: the tie-break builds the lists of a tile's window in LDS, straight from ps/xval) */
            const bool fused = lz77k_tokens_builds_lists(g, J.tvariant, J.d_order);
            if (!fused)
                HIPCHK(lz77k_xfer_index(c.ps.as<uint32_t>(), c.xval.as<uint32_t>(), xa, (uint32_t)x_done, dbase, (uint32_t)e, c.ofs.as<uint32_t>(),
                                        c.ent.as<uint2>(), c.scantmp.p, s, (uint32_t)x_new, c.flag.as<unsigned long long>() + 8, g.fast ? (uint32_t)g.sb : 0u));
            HIPCHK(lz77k_tokens(c.in.as<uint8_t>(), J.nloc, g, c.chain.as<uint32_t>() + ta, tb - ta, c.maxlen.as<uint8_t>(), c.ofs.as<uint32_t>(),
                                c.ent.as<uint2>(), dbase, (uint32_t)b, (uint32_t)e, tokbuf + 4 + ta, c.tstart.as<uint32_t>(),
                                g.fast ? nullptr : c.bidx.p,
                                J.tvariant, s, &c.tie_ev[2 * ci], J.d_order, J.first ? nullptr : look_cur, J.nlook, 0u,
                                fused ? c.ps.as<uint32_t>() : nullptr, fused ? c.xval.as<uint32_t>() : nullptr, c.flag.as<unsigned long long>() + 8));
            J.tie_timed[ci] = tb > ta;
        }
        HIPCHK(hipMemcpyAsync(c.h_small.as<unsigned long long>() + 2, c.flag.as<unsigned long long>() + 8, 8, hipMemcpyDeviceToHost, s));
    } else {
        if ((rc = c.tokval.need(64))) return rc;
        uint32_t *tokbuf = c.tokval.as<uint32_t>();
        if (carry.ntail) HIPCHK(hipMemcpyAsync(tokbuf + 4 - carry.ntail, carry.tail + 4 - carry.ntail, carry.ntail * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipEventRecord(c.ev[2], s));
    }

    /* -- pack (lz77.c:246-252): the words this segment's tokens start in; the last segment also the rest -- */
    const uint64_t K0 = J.K0, K1 = K0 + ntok;
    const uint64_t T = (uint64_t)g.T;
    const uint64_t zn_total = stream_bytes(K1, g.T);
    const uint64_t wlo = K0 == 0 ? 0 : (32 + K0 * T) / 32;
    const uint64_t whi = J.last ? (zn_total + 3) / 4 : (32 + K1 * T) / 32;
    const uint64_t nw = whi > wlo ? whi - wlo : 0;
    if ((rc = c.out.need(nw * 4 + 16))) return rc;
    HIPCHK(lz77k_pack_range(c.tokval.as<uint32_t>() + 4 - carry.ntail, K0 - carry.ntail, K1, g, c.out.as<uint32_t>(), wlo, nw, s));
    HIPCHK(hipEventRecord(c.ev[3], s));
    /* carry: the last four tokens seen so far */
    J.have_tail = ntok + carry.ntail < 4 ? ntok + carry.ntail : 4;
    if (J.have_tail)
        HIPCHK(hipMemcpyAsync(c.h_small.as<uint32_t>() + 20, c.tokval.as<uint32_t>() + 4 + ntok - J.have_tail, J.have_tail * 4, hipMemcpyDeviceToHost, s));
    J.out_bytes = J.last ? zn_total - 4 * wlo : 4 * nw;
    return LZ77X_OK;
}

int seg_finish(SegJob &J, SegCarry &carry, Sink &sink, double *waited, hipStream_t caller, RangeDrain *dr = nullptr /* a sink that
                   blocks on the host, behind a thread of its own: the segment's words are handed over, not written here */)
{
    Ctx &c = *J.c;
    hipStream_t s = J.s;
    int rc;
    {
        const double tw = now_ms();
        HIPCHK(hipStreamSynchronize(s));
        *waited += now_ms() - tw;
        for (uint32_t i = 0; i < J.have_tail; i++) carry.tail[4 - J.have_tail + i] = c.h_small.as<uint32_t>()[20 + i];
        carry.ntail = J.have_tail;
    }
    if (dr) {
        HIPCHK(hipEventRecord(c.pipe_ev[1], s));
        if ((rc = dr->submit(c.out.as<uint8_t>(), (size_t)J.out_bytes, c.pipe_ev[1]))) return rc;
    } else if ((rc = sink.write(c, c.out.as<uint8_t>(), (size_t)J.out_bytes, s))) return rc;
    if (!dr && caller && caller != s) {
        /* a device sink copies on this segment's stream: the caller's stream must see every segment's words, not
         * only those of the last one (an odd segment runs on the sibling stream) */
        HIPCHK(hipEventRecord(c.pipe_ev[1], s));
        HIPCHK(hipStreamWaitEvent(caller, c.pipe_ev[1], 0));
    }
    if (J.E > J.start) g_stats.transfers += c.h_small.as<unsigned long long>()[2];

    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
    g_stats.k_match_ms += ms;
    HIPCHK(hipEventElapsedTime(&ms, c.ev[2], c.ev[3]));
    g_stats.k_token_ms += ms;
    for (uint32_t i = 0; i < J.launches; i++) {
        HIPCHK(hipEventElapsedTime(&ms, c.sort_ev[4 * i], c.sort_ev[4 * i + 3]));
        g_stats.k_sort_chunks_ms += ms;
        HIPCHK(hipEventElapsedTime(&ms, c.sort_ev[4 * i], c.sort_ev[4 * i + 1]));
        g_stats.k_sort_ms += ms;
        HIPCHK(hipEventElapsedTime(&ms, c.sort_ev[4 * i + 1], c.sort_ev[4 * i + 2]));
        g_stats.k_walk_ms += ms;
    }
    if (J.E > J.start) {
        HIPCHK(hipEventElapsedTime(&ms, c.match_ev[0], c.match_ev[1]));
        g_stats.k_chain_ms += ms;
        HIPCHK(hipEventElapsedTime(&ms, c.match_ev[2], c.match_ev[3]));
        g_stats.k_prio_ms += ms;
        for (uint32_t ci = 0; ci < J.nchunks; ci++) {
            if (!J.tie_timed[ci]) continue;
            HIPCHK(hipEventElapsedTime(&ms, c.tie_ev[2 * ci], c.tie_ev[2 * ci + 1]));
            g_stats.k_tiebreak_ms += ms;
            g_stats.token_launches++;
        }
    }
    return LZ77X_OK;
}

/* the second context set of a device (same device as c): lets two segments of one stream be in flight */
int ctx_sibling(Ctx &c, Ctx **out)
{
    if (!c.pipe) c.pipe = new Ctx();
    int rc = ctx_init(*c.pipe, c.device);
    if (rc) return rc;
    *out = c.pipe;
    return LZ77X_OK;
}

int ctx_drain(Ctx &c, Ctx **out)
{
    if (!c.drain) c.drain = new Ctx();
    int rc = ctx_init(*c.drain, c.device);
    if (rc) return rc;
    *out = c.drain;
    return LZ77X_OK;
}

/* The device-resident encode of an input of any size: the source is cut into segments of up to
 * LZ77X_SEGMENT positions (default 2^30); a segment
 * starts sb bytes before its first token (the look-back window), so consecutive segments overlap by sb + the
 * look-ahead, and hands the state of lz77.c's two sequential loops to the next one (SegCarry).
 *
 * Two segments are in flight, on two context sets and two streams of the device: the gate iteration of the
 * priority recurrence is a chain of latency-bound launches (one wavefront per block, a host round trip per
 * iteration) that leaves the CUs' issue slots idle, and the only thing segment k+1's match stage or segment
 * k-1's tie-break need from it is nothing -- so while the host drives the recurrence of segment k on one
 * stream, the other stream runs the tie-break of k-1 and then the match stage of k+1:
 *     stream A:  match 0 | chain, recurrence 0 | tie-break, pack 0 |  match 2 (after match 1)  | ...
 *     stream B:           (after match 0) match 1 | chain, recurrence 1 | tie-break, pack 1 | ...
 * LZ77X_PIPELINE=0: one context, one segment at a time.  One device, sb <= 4096.  Nothing but the stream
 * (and a few words per segment) leaves the GPU. */
int encode_stream_device(Ctx &c, Source &src, Sink &sink, const lz77x_geom &g, hipStream_t s, bool *fallback, size_t *n_fallback)
{
    const double t_begin = now_ms();
    memset(&g_stats, 0, sizeof g_stats);
    *fallback = false;
    int rc;
    double waited = 0;
    HIPCHK(hipSetDevice(c.device));
    const size_t usb = (size_t)g.sb, halo = (size_t)g.la + 64;
    const uint32_t csub = lz77k_chain_sub();
    size_t seg = (size_t)1 << 30, scratch_cap = 0, token_chunk = (size_t)128 << 20;
    {
        const char *ce = getenv("LZ77X_TOKEN_CHUNK");
        if (ce && atoll(ce) > 0) token_chunk = (size_t)atoll(ce);
        if (token_chunk > ((size_t)1 << 31)) token_chunk = (size_t)1 << 31;
    }
    bool pipelined = !(getenv("LZ77X_PIPELINE") && atoi(getenv("LZ77X_PIPELINE")) == 0);
    {
        const size_t lo = 4 * usb + 3 * (size_t)csub;
        const char *se = getenv("LZ77X_SEGMENT");
        if (se && atoll(se) > 0) seg = (size_t)atoll(se);
        else {
            /* LZ77X_SPLIT=1: cut an input of known size above 32 MB in two so that the halves overlap.  Off by
             * default: the recurrence of a half takes as long as that of the whole (it is latency bound: 6.8 ms
             * per 50 MB half against 6.0 for 100 MB, co-running kernels included), so S1 ends at 23.1 ms against
             * 20.8 in one segment */
            const size_t hint = src.size_hint();
            const char *sp = LZ77X_VENV("LZ77X_SPLIT");
            if (sp && atoi(sp) && pipelined && hint >= ((size_t)32 << 20) && hint / 2 + csub < seg) seg = (hint / 2 + csub) / csub * csub;
            /* a long input out of host memory or a file (or one of unknown length: a pipe): segments of 128 MB, so that
             * the bytes of the next one travel -- from a thread of its own -- while the recurrence of this one runs, and
             * the stream of the one before leaves while this one's kernels run.  In one segment of 2^30 the whole input
             * crosses PCIe before the first kernel starts (1 GB of text from host memory: 162 ms against 108 resident;
             * eight segments cost the resident case 117) */
            const size_t host_seg = ((size_t)128 << 20) / csub * csub;
            if (pipelined && g.fast && src.host_backed() && (hint == 0 || hint > 3 * host_seg) && seg > host_seg) seg = host_seg;
        }
        /* a source of known size below a segment: buffers sized for it, not for 2^30 positions (the whole input
         * is then one segment: want = seg + halo > what is left, so the first load sees the end) */
        const size_t known = src.size_hint();
        if (known && known < seg) seg = (known + csub - 1) / csub * csub;
        if (seg < lo) seg = lo;
        if (seg > ((size_t)3 << 30)) seg = (size_t)3 << 30;      /* local coordinates are 32-bit */
        /* ... and the device must hold it: per position of a segment ~27 B (windows in LDS: input, ps, maxlen, the regions'
         * order, xval, chain, token words, gates) or ~62 B (large windows: rank + inverse arrays, hand-over index, bucket
         * records), plus the match stage's scratch per launch; twice when a second segment is in flight (measured:
         * tools/mem_probe.py).  A device with less to spare gets smaller launches, then smaller segments. */
        size_t avail = 0;
        if ((rc = device_budget(c, &avail))) return rc;
        const size_t per_pos = g.fast ? 34 : 70, slack = (size_t)384 << 20;    /* (the cached buffers carry an eighth of headroom each) */
        const size_t one_region = lz77k_match_scratch_bytes(g, 1);
        size_t planned = 0;
        for (int pass = 0; pass < 2; pass++) {
            /* one context set while the input is one segment; two as soon as it is not (the second pass) */
            const bool two = pipelined && (pass == 1 || !(known && known <= seg));
            const size_t share = avail / 10 * 9 / (two ? 2 : 1);
            scratch_cap = (size_t)(g.fast ? 3 : 16) << 30;
            if (scratch_cap > share / 4) scratch_cap = share / 4;
            if (scratch_cap < one_region) scratch_cap = one_region;
            const size_t fixed = scratch_cap + slack + (g.fast ? 0 : (size_t)1 << 30);
            if (share < fixed + per_pos * lo) {
                snprintf(g_err, sizeof g_err, "device memory: %.1f MB to plan with, a segment of %zu positions needs %.1f MB", avail / 1e6, lo,
                         (fixed + per_pos * lo) / 1e6);
                return LZ77X_E_HIP;
            }
            const size_t seg_fit = (share - fixed) / per_pos / csub * csub;
            const size_t seg_new = seg > seg_fit ? (seg_fit < lo ? lo : seg_fit) : seg;
            const bool multi = !(known && known <= seg_new);
            seg = seg_new;
            planned = (fixed + per_pos * seg) * (two ? 2 : 1);
            if (two || !multi || !pipelined) break;              /* (else: it became several segments -- plan again for two in flight) */
        }
        budget_commit(c, planned);
        if (trace_on())
            fprintf(stderr, "[lz77x] memory plan: %.1f MB to plan with, segments of %zu positions, %.1f MB of match scratch per launch\n", avail / 1e6,
                    seg, scratch_cap / 1e6);
    }
    Ctx *cx[2] = {&c, &c};
    hipStream_t sx[2] = {s, s};            /* (the second context set: created when a second segment turns up) */
    double t_finish = 0, t_tokens = 0, t_join = 0, t_front = 0;
    SegCarry carry;
    SegJob J[2];
    uint64_t n_total = 0;
    bool eof = false;

    /* input of segment k into context k & 1: the tail of its predecessor's buffer, then the source */
    auto load = [&](int k, const SegJob *prev) -> int {
        SegJob &N = J[k & 1];
        N = SegJob();
        N.c = cx[k & 1];
        N.s = sx[k & 1];
        N.scratch_cap = scratch_cap;
        N.token_chunk = token_chunk;
        Ctx &cn = *N.c;
        const size_t want_local = (k ? usb : 0) + seg + halo;
        int r;
        size_t have = 0;
        /* sized for every segment at once: the buffer must not move once a predecessor's tail sits in it */
        if ((r = cn.in.need(usb + seg + halo + LZ77X_PAD + 64))) return r;
        if (prev) {
            const size_t keep0 = (size_t)prev->E - usb, keep = (size_t)prev->nloc - keep0;
            N.gpos0 = prev->gpos0 + keep0;
            if (prev->c != N.c) {
                HIPCHK(hipStreamWaitEvent(N.s, prev->c->pipe_ev[0], 0));        /* its input has arrived */
                HIPCHK(hipMemcpyAsync(cn.in.p, prev->c->in.as<uint8_t>() + keep0, keep, hipMemcpyDeviceToDevice, N.s));
            } else {
                /* same buffer: move [E - sb, have) to the front (through a spare buffer: the ranges overlap) */
                if ((r = cn.bidx.need(keep + 64))) return r;
                HIPCHK(hipMemcpyAsync(cn.bidx.p, cn.in.as<uint8_t>() + keep0, keep, hipMemcpyDeviceToDevice, N.s));
                HIPCHK(hipMemcpyAsync(cn.in.p, cn.bidx.p, keep, hipMemcpyDeviceToDevice, N.s));
            }
            have = keep;
        }
        if (!eof && have < want_local) {
            size_t got = 0;
            if ((r = src.read(cn, cn.in.as<uint8_t>() + have, want_local - have, N.s, &got))) return r;
            if (got < want_local - have) eof = true;
            have += got;
            n_total += got;
        }
        HIPCHK(hipEventRecord(cn.pipe_ev[0], N.s));
        N.last = eof;
        N.nloc = (uint32_t)have;
        N.cover = N.last ? N.nloc : (uint32_t)((k ? usb : 0) + seg);
        return LZ77X_OK;
    };
    /* where segment k's tokens start and end (needs the carry of k - 1) */
    auto place = [&](SegJob &K) {
        K.start = (uint32_t)(carry.chain_pos - K.gpos0);
        if (K.last) K.E = K.nloc;
        else K.E = K.start + (K.cover - K.start) / csub * csub;
    };

    /* a file or host memory as the sink of several segments: a thread of its own copies a segment's words out of the device
     * and writes them (RangeDrain, as in the decoder) while this one drives the next segment's recurrence -- on 1 GB of
     * text from host memory to host memory the segments' words were 95 of 165 ms on this thread */
    RangeDrain drain;
    RangeDrain *dr = nullptr;
    /* the thread that loads the next segment: ONE per call, started when a second segment turns up, handed a segment at a
     * time (a thread per segment was hundreds of short-lived threads, each with its own state inside the runtime, on the
     * small segments the tests force) and joined on every way out */
    struct Loader {
        std::thread t;
        std::mutex mu;
        std::condition_variable cv;
        std::function<int(int, SegJob *)> fn;
        int device = 0;
        int req = -1;                         /* the segment to load, -1: none */
        SegJob *prev = nullptr;
        bool busy = false, stop = false, started = false, on = false;   /* on: a load has been handed over and not yet waited for */
        int rc = LZ77X_OK;
        char err[256] = "";                   /* g_err is thread_local: what the loader thread wrote there comes back through here */
        void run()
        {
            const bool dev_ok = hipSetDevice(device) == hipSuccess;
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [&] { return stop || req >= 0; });
                if (req < 0) return;
                const int k = req;
                SegJob *p = prev;
                req = -1;
                lk.unlock();
                g_err[0] = 0;
                const int r = dev_ok ? fn(k, p) : LZ77X_E_HIP;
                lk.lock();
                rc = r;
                if (r != LZ77X_OK) snprintf(err, sizeof err, "%s", dev_ok ? g_err : "hipSetDevice failed on the loader thread");
                busy = false;
                cv.notify_all();
            }
        }
        bool kick(int k, SegJob *p)
        {
            if (!started) {
                try { t = std::thread(&Loader::run, this); started = true; }
                catch (...) { return false; }
            }
            { std::lock_guard<std::mutex> lk(mu); req = k; prev = p; busy = true; }
            cv.notify_all();
            on = true;
            return true;
        }
        int join()
        {
            if (!on) return LZ77X_OK;
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !busy; });
            on = false;
            if (rc != LZ77X_OK) snprintf(g_err, sizeof g_err, "%s", err);
            return rc;
        }
        ~Loader()
        {
            if (!started) return;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !busy; }); stop = true; }
            cv.notify_all();
            t.join();
        }
    } loader;
    loader.device = c.device;
    loader.fn = [&](int k, SegJob *p) { return load(k, p); };
    {
        const double tl = now_ms();
        if ((rc = load(0, nullptr))) return rc;
        TRACE("  first segment loaded", tl);
    }
    if ((rc = seg_front(J[0], g))) return rc;
    int prev_unfinished = -1;
    for (int k = 0;; k++) {
        SegJob &K = J[k & 1];
        place(K);
        const bool single = carry.first && K.last;
        if (pipelined && !K.last) {
            /* the next segment's input and match stage, behind this one's match stage on the other stream */
            if (prev_unfinished >= 0) {
                /* its context is the one segment k-1 still occupies: its last tokens and its words are taken first */
                const double tf = now_ms();
                if ((rc = seg_finish(J[prev_unfinished & 1], carry, sink, &waited, s, dr))) return rc;
                t_finish += now_ms() - tf;
                prev_unfinished = -1;
            }
            if (cx[1] == &c) {
                if ((rc = ctx_sibling(c, &cx[1]))) return rc;
                sx[1] = cx[1]->stream;
                if (sink.blocks_on_host()) {
                    Ctx *dctx = nullptr;
                    if ((rc = ctx_drain(c, &dctx))) return rc;
                    if (drain.start(&sink, dctx) == LZ77X_OK) dr = &drain;
                }
            }
            if (src.host_backed()) {
                /* reading the source blocks a host thread (preads or copies out of pageable memory into the pinned slots):
                 * a thread of its own does it while this one drives the recurrence of segment k; the match stage of k + 1
                 * then runs beside the tie-break of k instead of beside its recurrence */
                (void)loader.kick(k + 1, &K);                   /* (no thread to be had: loaded right here, below) */
            }
            if (!loader.on) {
                if ((rc = load(k + 1, &K))) return rc;
                HIPCHK(hipStreamWaitEvent(J[(k + 1) & 1].s, K.c->ev[1], 0));
                if ((rc = seg_front(J[(k + 1) & 1], g))) return rc;
            }
        }
        bool fb = false;
        const double tm = now_ms();
        if ((rc = seg_mid(K, g, carry, single, &fb, &waited))) return rc;
        TRACE("  match stage waited for, chain + recurrence", tm);
        if (fb) { *fallback = true; *n_fallback = K.nloc; return LZ77X_OK; }
        if (loader.on) {
            const double tj = now_ms();
            if ((rc = loader.join())) return rc;
            t_join += now_ms() - tj;
            HIPCHK(hipStreamWaitEvent(J[(k + 1) & 1].s, K.c->ev[1], 0));
            const double tf = now_ms();
            if ((rc = seg_front(J[(k + 1) & 1], g))) return rc;
            t_front += now_ms() - tf;
        }
        if (prev_unfinished >= 0) {
            const double tf = now_ms();
            if ((rc = seg_finish(J[prev_unfinished & 1], carry, sink, &waited, s, dr))) return rc;
            t_finish += now_ms() - tf;
            prev_unfinished = -1;
        }
        const double tt = now_ms();
        if (dr && (rc = dr->wait(1))) return rc;                /* this context's words of two segments ago have left its buffer */
        if ((rc = seg_tokens(K, g, carry))) return rc;
        t_tokens += now_ms() - tt;
        if (K.last) {
            if ((rc = seg_finish(K, carry, sink, &waited, s, dr))) return rc;     /* (joins the caller's stream) */
            if (dr && (rc = dr->wait(0))) return rc;
            TRACE("  tokens + the stream to the sink", tt);
            break;
        }
        if (pipelined) prev_unfinished = k;
        else {
            if ((rc = seg_finish(K, carry, sink, &waited, s, dr))) return rc;
            if ((rc = load(k + 1, &K))) return rc;
            if ((rc = seg_front(J[(k + 1) & 1], g))) return rc;
        }
    }
    g_stats.n = n_total;
    g_stats.zn = sink.total;
    g_stats.ntok = carry.ntok;
    g_stats.total_ms = now_ms() - t_begin;
    g_stats.copy_ms = waited;
    if (trace_on())
        fprintf(stderr, "[lz77x]   in all: finishing segments %.2f ms, token stages %.2f ms, waiting for the loader %.2f ms, match stages enqueued in %.2f ms\n",
                t_finish, t_tokens, t_join, t_front);
    TRACE("encode_stream_device total", t_begin);
    return LZ77X_OK;
}

/* ONE stream on SEVERAL devices (BASELINE config 5): the positions are cut into D contiguous
 * shards, device d holds only its shard's bytes (plus sb of look-back and the look-ahead) and only its share
 * of every intermediate array -- memory per device ~ n/D.  Every stage is local to a shard except the two
 * sequential loops of lz77.c, which cross the cuts as a few KB through the host:
 *   - parse chain: each shard's map  entry offset -> (exit offset, tokens)  (la entries), chained on the host;
 *   - priority recurrence: each gate iteration, each shard's whole map of boundary cells (sb entries), chained
 *     on the host into the cells every shard starts from;
 *   - packing: the last four tokens of a shard go to its successor (a stream word can straddle the cut).
 * No device-to-device traffic, no collective.  Priorities are GLOBAL positions here (local + voff): the shards
 * iterate together, so nothing can be renumbered; one call therefore handles < 4 GiB (a longer stream goes
 * through segments on one device, encode_stream_device). */
}  // namespace

extern "C" {

/* host-only: how one stream of n bytes is cut for `shards` devices.  Returns the number of shards actually
 * used (a shard is never smaller than 4*sb + 12 KiB) */
int lz77x_shard_plan(size_t n, int sb, int la, int shards, lz77x_shard *out)
{
    if (check_geom(sb, la) != LZ77X_OK || shards < 1) return LZ77X_E_ARG;
    const size_t usb = (size_t)sb, halo = (size_t)la + 64;
    size_t D = (size_t)shards;
    const size_t min_shard = 4 * usb + 3 * (size_t)4096;
    while (D > 1 && n / D < min_shard) D--;
    for (size_t d = 0; d < D && out; d++) {
        lz77x_shard &j = out[d];
        j.first_token_pos = (uint64_t)n * d / D;
        j.end_token_pos = (uint64_t)n * (d + 1) / D;
        j.lookback = d ? (uint32_t)usb : 0u;
        j.local0 = j.first_token_pos - j.lookback;
        const uint64_t end = d + 1 == D ? (uint64_t)n : (j.end_token_pos + halo < n ? j.end_token_pos + halo : (uint64_t)n);
        j.local_bytes = end - j.local0;
        j.steps = j.end_token_pos - j.local0 > usb ? j.end_token_pos - j.local0 - usb : 0;
    }
    return (int)D;
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

namespace {

struct ShardJob {
    Ctx *c = nullptr;
    uint64_t a = 0, b = 0, gpos0 = 0;      /* tokens [a, b) (global), local 0 = gpos0 */
    uint32_t look = 0, nloc = 0, E = 0, nx = 0, entry = 0, start = 0, ntok = 0, nsub = 0;
    uint64_t K0 = 0;
    lz77k_prio_plan P;
    const uint32_t *d_tbase = nullptr;
    uint32_t *h = nullptr;                  /* pinned scratch of this shard (c->h_tbase) */
};

int encode_sharded(std::vector<Ctx *> &cs, const uint8_t *src, size_t n, const lz77x_geom &g, Sink &sink)
{
    const double t_begin = now_ms();
    memset(&g_stats, 0, sizeof g_stats);
    if (n > LZ77X_MAX_N) return LZ77X_E_TOOBIG;
    const size_t usb = (size_t)g.sb;
    const uint32_t csub = lz77k_chain_sub();
    std::vector<lz77x_shard> plan(cs.size());
    const int planned = lz77x_shard_plan(n, g.sb, g.la, (int)cs.size(), plan.data());
    if (planned < 1) return LZ77X_E_ARG;
    const size_t D = (size_t)planned;
    std::vector<ShardJob> J(D);
    int rc;
    const char *tv = LZ77X_VENV("LZ77X_TOKEN_VARIANT");
    const int tvariant = tv ? atoi(tv) : 0;
    auto dev = [&](ShardJob &j) -> int { HIPCHK(hipSetDevice(j.c->device)); return LZ77X_OK; };
    auto sync_all = [&]() -> int {
        for (ShardJob &j : J) { HIPCHK(hipSetDevice(j.c->device)); HIPCHK(hipStreamSynchronize(j.c->stream)); }
        return LZ77X_OK;
    };
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
        HIPCHK(lz77k_prio_begin(j.P, c.ps.as<uint32_t>(), j.nx, g.sb, c.xval.as<uint32_t>(), c.prio_tmp.p, (uint32_t)j.gpos0, nullptr, s));
        HIPCHK(hipStreamSynchronize(s));
        return LZ77X_OK;
    });
    if (rc) return rc;
    for (uint32_t l : launches_of) g_stats.match_launches += l;
    double t_serial = now_ms();

    /* -- the parse chain across the cuts (lz77.c:98): entry offset and first-token index of every shard -- */
    {
        uint32_t e = 0;
        uint64_t K = 0;
        for (ShardJob &j : J) {
            j.entry = e;
            j.K0 = K;
            j.ntok = j.h[e];
            lz77x_shard_compose_chain(reinterpret_cast<const uint8_t *>(j.h + 256), j.h, &e, &K);
            j.start = j.look + j.entry;
        }
    }
    for (ShardJob &j : J) {
        Ctx &c = *j.c;
        if ((rc = dev(j))) return rc;
        HIPCHK(lz77k_chain_finish(c.maxlen.as<uint8_t>(), j.E, g.la, c.chain.as<uint32_t>(), c.chain_tmp.p, c.stream, j.look, j.entry, &j.d_tbase,
                                  &j.nsub, nullptr));
        HIPCHK(hipMemcpyAsync(c.h_tbase.p, j.d_tbase, ((size_t)j.nsub + 1) * 4, hipMemcpyDeviceToHost, c.stream));
    }

    /* -- the priority recurrence across the cuts (tree.c:202-231): all shards iterate together -- */
    {
        std::vector<uint32_t> v(usb);
        std::vector<char> flipped(D, 0);
        const uint16_t *d_sdest = nullptr;
        const uint32_t *d_sloc = nullptr;
        int max_iters = 1 << 30;
        uint32_t iters = 0;
        for (int it = 0; it < max_iters; it++) {
            for (size_t d = 0; d + 1 < D; d++) {            /* the last shard's whole map is nobody's input */
                ShardJob &j = J[d];
                if ((rc = dev(j))) return rc;
                HIPCHK(lz77k_prio_maps(j.P, j.c->stream, true, &d_sdest, &d_sloc));
                if (j.nx) {
                    HIPCHK(hipMemcpyAsync(j.h + 512, d_sloc, usb * 4, hipMemcpyDeviceToHost, j.c->stream));
                    HIPCHK(hipMemcpyAsync(j.h + 512 + usb, d_sdest, usb * 2, hipMemcpyDeviceToHost, j.c->stream));
                }
            }
            if (D > 0) {
                ShardJob &j = J[D - 1];
                if ((rc = dev(j))) return rc;
                HIPCHK(lz77k_prio_maps(j.P, j.c->stream, false, nullptr, nullptr));
            }
            host_serial_ms += now_ms() - t_serial;
            if ((rc = sync_all())) return rc;
            t_serial = now_ms();
            for (size_t i = 0; i < usb; i++) v[i] = (uint32_t)i;          /* the start of the input: every cell its own position */
            for (size_t d = 0; d < D; d++) {
                ShardJob &j = J[d];
                if ((rc = dev(j))) return rc;
                if (d > 0) {
                    uint32_t *pin = j.h + 512 + 2 * usb;                /* pinned copy of the cells this shard starts from */
                    memcpy(pin, v.data(), usb * 4);
                    HIPCHK(lz77k_prio_set_in0(j.P, pin, hipMemcpyHostToDevice, j.c->stream));
                }
                if (d + 1 < D) {                                       /* v <- this shard's whole map applied to v */
                    if (j.nx == 0) continue;                           /* no step: the cells pass through */
                    lz77x_shard_compose_cells(reinterpret_cast<const uint16_t *>(j.h + 512 + usb), j.h + 512, g.sb, v.data());
                }
            }
            for (ShardJob &j : J) {
                if ((rc = dev(j))) return rc;
                HIPCHK(lz77k_prio_sweep(j.P, j.c->stream, j.c->h_small.as<uint32_t>() + 8, nullptr));
            }
            host_serial_ms += now_ms() - t_serial;
            if ((rc = sync_all())) return rc;
            t_serial = now_ms();
            iters++;
            bool any = false, earlier = false;
            for (size_t d = 0; d < D; d++) {
                const uint32_t *hf = J[d].c->h_small.as<uint32_t>() + 8;
                flipped[d] = hf[0] != 0;
                /* a shard after one that still changes may be handed different cells next time: nothing of it is final */
                lz77k_prio_advance(J[d].P, hf, earlier);
                earlier = earlier || flipped[d];
                any = any || flipped[d];
            }
            if (!any) break;
        }
        g_stats.prio_iters = iters;
    }

    /* -- tokens: every shard resolves its own (look-back priorities = the cells it started from) -- */
    for (ShardJob &j : J) {
        Ctx &c = *j.c;
        if ((rc = dev(j))) return rc;
        hipStream_t s = c.stream;
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
        j.h[8] = have;
    }
    if ((rc = sync_all())) return rc;

    /* -- pack (lz77.c:246-252): each shard the stream words its tokens start in, with its predecessors' last
     *    tokens in front; then the pieces leave in order -- */
    const uint64_t T = (uint64_t)g.T;
    const uint64_t K_all = D ? J[D - 1].K0 + J[D - 1].ntok : 0;
    const uint64_t zn_total = stream_bytes(K_all, g.T);
    uint32_t tail[4] = {0, 0, 0, 0};
    uint32_t ntail = 0;
    std::vector<uint64_t> piece(D, 0);
    for (size_t d = 0; d < D; d++) {
        ShardJob &j = J[d];
        Ctx &c = *j.c;
        if ((rc = dev(j))) return rc;
        const bool last = d + 1 == D;
        const uint64_t K0 = j.K0, K1 = K0 + j.ntok;
        const uint64_t wlo = K0 == 0 ? 0 : (32 + K0 * T) / 32;
        const uint64_t whi = last ? (zn_total + 3) / 4 : (32 + K1 * T) / 32;
        const uint64_t nw = whi > wlo ? whi - wlo : 0;
        if ((rc = c.out.need(nw * 4 + 16))) return rc;
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
    g_stats.host_chain_ms = 0;
    g_stats.copy_ms = host_serial_ms;                      /* sharded: host time no device overlaps (exchange + enqueue) */
    g_stats.n = n;
    g_stats.zn = sink.total;
    g_stats.ntok = K_all;
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
    if (cs.size() > 1 && !src_on_device && device_pipeline_ok(1, g))
        return encode_sharded(cs, reinterpret_cast<const uint8_t *>(src), n, g, sink);
    if (device_pipeline_ok(cs.size(), g)) {
        MemSource ms(src, n, src_on_device);
        bool fallback = false;
        size_t nfb = 0;
        const int rc = encode_stream_device(c0, ms, sink, g, s, &fallback, &nfb);
        if (rc != LZ77X_OK || !fallback) return rc;
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

}  // namespace

/* ==================================================================== C ABI ========= */

extern "C" {

size_t lz77x_encode_bound(size_t n, int sb, int la)
{
    if (check_geom(sb, la) != LZ77X_OK) return 0;
    lz77x_geom g;
    lz77x_make_geom(&g, sb, la);
    return stream_bytes(n, g.T);
}

int lz77x_encode(const uint8_t *in, size_t n, int sb, int la, uint8_t **out, size_t *out_n)
{
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
}

int lz77x_encode_device(const void *d_in, size_t n, int sb, int la, void *d_out, size_t out_cap, size_t *out_n, void *stream)
{
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
}

int lz77x_decode(const uint8_t *z, size_t zn, uint8_t **out, size_t *out_n)
{
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
}

int lz77x_decode_device(const void *d_z, size_t zn, void *d_out, size_t out_cap, size_t *out_n, void *stream)
{
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
}

/* lz77.h:14 encode(file, out, la, sb) as called at main.c:150 */
int lz77x_encode_file(FILE *in, FILE *out, int la, int sb)
{
    if (!in || !out) return LZ77X_E_ARG;
    int rc = check_geom(sb, la);
    if (rc) return rc;
    int shards = g_shards;
    if (shards <= 0) { const char *e = getenv("LZ77X_SHARDS"); shards = e ? atoi(e) : 1; }
    if (shards > 1) return lz77x_encode_file_buffered(in, out, la, sb);
    Lease lease;
    Ctx &g_ctx = lease.set->primary;
    (void)g_ctx;
    const double t0 = now_ms();
    std::vector<Ctx *> cs;
    if ((rc = shard_contexts(*lease.set, 1, cs))) return rc;
    TRACE("runtime + context init", t0);
    Ctx &c = g_ctx;
    lz77x_geom g;
    make_encode_geom(&g, sb == -1 ? LZ77X_DEFAULT_SB : sb, la == -1 ? LZ77X_DEFAULT_LA : la);
    size_t n = 0;
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
}

/* lz77.h:15 decode(file, out) as called at main.c:161 */
int lz77x_decode_file(FILE *in, FILE *out)
{
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
}

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

}  // extern "C" (reopened below)

namespace {
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
}  // namespace

extern "C" {

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
{
    if ((!in || !maxlen) && n) return LZ77X_E_ARG;
    Lease lease;
    Ctx &g_ctx = lease.set->primary;
    (void)g_ctx;
    lz77x_geom g;
    int rc = run_match_only(*lease.set, in, n, sb, la, &g);
    if (rc) return rc;
    if (n) HIPCHK(hipMemcpy(maxlen, g_ctx.maxlen.p, n, hipMemcpyDeviceToHost));
    return LZ77X_OK;
}

int lz77x_stage_neighbours(const uint8_t *in, size_t n, int sb, int la, uint16_t *P, uint16_t *S)
{
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
}

int lz77x_stage_priorities(const uint16_t *P, const uint16_t *S, size_t n, int sb, uint32_t *xval)
{
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
}

int lz77x_stage_priorities_device(const uint16_t *P, const uint16_t *S, size_t n, int sb, uint32_t *xval, int *iters_out)
{
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
}

int lz77x_stage_chain_device(const uint8_t *maxlen, size_t n, int la, uint32_t *chain, size_t *ntok)
{
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
}

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
    for (int t = 1; t < lanes; t++) th.emplace_back(work);
    if (n_files) work();
    for (auto &t : th) t.join();
    int first = LZ77X_OK;
    for (int i = 0; i < n_files; i++) {
        if (rcs) rcs[i] = rc[(size_t)i];
        if (first == LZ77X_OK && rc[(size_t)i] != LZ77X_OK) first = rc[(size_t)i];
    }
    return first;
}

int lz77x_encode_files(int n_files, FILE **in, FILE **out, int la, int sb, int *rc) { return run_files(n_files, in, out, la, sb, rc, true); }
int lz77x_decode_files(int n_files, FILE **in, FILE **out, int *rc) { return run_files(n_files, in, out, 0, 0, rc, false); }

}  // extern "C"
