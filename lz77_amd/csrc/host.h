/*
 * host.h -- what the host-side translation units of the library share (round 5: pipeline.cpp, 3,500 lines, cut along its
 * seams without a change of behaviour):
 *
 *   ctx.cpp          device contexts, the pool concurrent callers lease them from, the memory budget   (this header's types)
 *   hostio.cpp       pageable memory / FILE* <-> device through the pinned staging slots
 *   encode_pipe.cpp  the device pipeline of an encode: segments, two in flight (replaces lz77.c:51-140)
 *   encode_host.cpp  the host-assisted pipeline of round 1: the fallback when the gate iteration gives up
 *   decode_pipe.cpp  the decoder: token ranges, two in flight (replaces lz77.c:148-197)
 *   shard.cpp        one stream on several devices, both directions, and the host-only shard arithmetic of the C ABI
 *   api.cpp          the C ABI of include/lz77_mi355x.h
 *
 * There is NO CPU fallback: without a HIP device every entry point returns LZ77X_E_NODEV.
 */
#ifndef LZ77X_HOST_H
#define LZ77X_HOST_H

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

/* nothing of the host side leaves the library but the C ABI: every block of the namespace is hidden */
#define LZ77X_HOST_NS namespace lz77x_host __attribute__((visibility("hidden")))

LZ77X_HOST_NS {

struct Ctx;
struct CtxSet;
struct DevBuf;
class CopyPool;

/* ctx.cpp */
bool trace_on();
double now_ms();
void trace_allocs(const char *what);
int device_budget(Ctx &c, size_t *avail);
void budget_commit(Ctx &c, size_t planned);
int ctx_init(Ctx &c, int device = -1);
int need_stream(Ctx &c, hipStream_t Ctx::*m);
int check_geom(int &sb, int &la);
int primary_context(CtxSet &S);
int shard_contexts(CtxSet &S, int want, std::vector<Ctx *> &cs);
void make_encode_geom(lz77x_geom *g, int sb, int la);
void ctx_release(Ctx &c);
int ctx_sibling(Ctx &c, Ctx **out);
int ctx_drain(Ctx &c, Ctx **out);

/* (__thread, not thread_local: an `extern thread_local` is reached through a weak init wrapper, and inside a hidden
 * namespace of a shared library the undefined weak resolves to the library's base instead of null -- a call to it) */
extern __thread char g_err[256];
extern __thread lz77x_stats g_stats;
extern int g_shards;

#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            snprintf(g_err, sizeof g_err, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return LZ77X_E_HIP;                                                                \
        }                                                                                      \
    } while (0)

/* No C++ exception crosses the C ABI: the entry points are function-try-blocks that end in this (a std::bad_alloc of a
 * vector, the std::system_error of a thread that could not start -- std::terminate inside a C caller otherwise) */
#define LZ77X_API_CATCH                                                                                            \
    catch (const std::bad_alloc &) { snprintf(g_err, sizeof g_err, "out of host memory"); return LZ77X_E_NOMEM; }  \
    catch (const std::exception &ex_) { snprintf(g_err, sizeof g_err, "host exception: %s", ex_.what()); return LZ77X_E_HIP; } \
    catch (...) { snprintf(g_err, sizeof g_err, "host exception"); return LZ77X_E_HIP; }

#define TRACE(label, t0)                                                                  \
    do { if (trace_on()) fprintf(stderr, "[lz77x] %-28s %8.2f ms\n", label, now_ms() - (t0)); } while (0)


/* LZ77X_TRACE: where a call's wall time goes besides kernels and copies (per thread, reset by the entry points) */
extern __thread double g_alloc_ms, g_pin_ms, g_fread_ms, g_fwrite_ms;
extern __thread size_t g_alloc_bytes, g_pin_bytes;


/* match-stage scratch per launch at the large windows, GB (16 until round 6: S3's 1079 regions in one launch; the stage's kernels
 * are throughput-bound there -- grid-wide sorts, 25 rounds of walker workgroups -- so a second launch costs little and the
 * footprint of a 212 MB encode drops by 4 GB) */
#ifndef LZ77X_BIG_SCRATCH_GB
#define LZ77X_BIG_SCRATCH_GB 12
#endif

/* LZ77X_POISON=1 (debug aid, never changes the output of correct code): every cached device and pinned buffer is filled with
 * 0xA5 when a call leases its context set and when a buffer grows -- a kernel or host loop that reads what this call has not
 * written then sees garbage instead of the zeroes of fresh memory or the plausible values of the call before (ctx.cpp) */
bool poison_on();
struct DevBuf;
bool poison_fresh(const DevBuf *b);   /* LZ77X_POISON_FRESH_MASK: which buffers (bit i = the i-th of Ctx::dev_bufs()) are poisoned when they are allocated */

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    /* limit: the headroom (an eighth, against re-allocation when the next input is a little larger) stops there -- a buffer
     * that is sized by a cap, like the match stage's scratch per launch, does not grow past it */
    int need(size_t bytes, size_t limit = 0)
    {
        if (bytes <= cap) return LZ77X_OK;
        const double t0 = trace_on() ? now_ms() : 0;
        if (p) { hipError_t e0 = hipFree(p); (void)e0; p = nullptr; cap = 0; }
        /* headroom against re-allocation when the next input is a little larger: an eighth, at most 64 MB (round 5 gave
         * every buffer an eighth whatever its size: 2 GB of a 30 GB large-window encode were headroom) */
        size_t want = bytes + (bytes / 8 < ((size_t)64 << 20) ? bytes / 8 : ((size_t)64 << 20)) + 4096;
        if (limit && want > limit) want = bytes > limit ? bytes : limit;
        HIPCHK(hipMalloc(&p, want));
        cap = want;
        /* (hipMemset may return before the fill has run, and the null stream does not order it against the non-blocking
         * streams the kernels use: without the synchronize the poison lands on top of what the call has written since) */
        if (poison_on() && poison_fresh(this)) { HIPCHK(hipMemset(p, 0xA5, want)); HIPCHK(hipDeviceSynchronize()); }
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
                    if (poison_on()) memset(p, 0xA5, cap);
                    return LZ77X_OK;
                }
                (void)hipGetLastError();
                free(q);
            }
        }
        HIPCHK(hipHostMalloc(&p, want, hipHostMallocPortable));   /* every shard's device copies to/from it */
        cap = want;
        if (poison_on()) memset(p, 0xA5, cap);
        return LZ77X_OK;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

/* a small result (whole words) from device memory into a pinned buffer of the context: by kernel where the buffer came from
 * hipHostMalloc (lz77k_publish), by the runtime's copy where it is registered memory */
inline hipError_t small_d2h(const PinBuf &pb, void *h_dst, const void *d_src, size_t bytes, hipStream_t s)
{
    if (!pb.registered && (bytes & 3) == 0 && bytes <= ((size_t)1 << 20)) return lz77k_publish(h_dst, d_src, (uint32_t)(bytes / 4), s);
    return hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, s);
}

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
extern CopyPool g_copy;                                           /* towards the device: copies out of pageable memory, file reads */
extern CopyPool g_copy_out;                                       /* away from it: copies into pageable memory, file writes -- an encode of several
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
        /* (a worker thread has no function-try-block of the C ABI above it: nothing may leave fn) */
        try { rcs[d] = fn(d); }
        catch (const std::bad_alloc &) { rcs[d] = LZ77X_E_NOMEM; snprintf(g_err, sizeof g_err, "out of host memory"); }
        catch (...) { rcs[d] = LZ77X_E_HIP; snprintf(g_err, sizeof g_err, "host exception on a shard's thread"); }
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

/* fn(d) for every shard d on D host threads that ALL exist before any of them starts: the shards of a joint iteration meet
 * at barriers (shard.cpp), so a thread that could not start must not leave the others waiting -- then none runs and the
 * call fails.  The first shard runs on the calling thread; the first error wins, its text ends up in the caller's g_err. */
template <class F> int run_team(size_t D, F fn)
{
    std::vector<int> rcs(D, LZ77X_OK);
    std::vector<std::string> msgs(D);
    std::mutex m;
    std::condition_variable cv;
    int state = 0;                                          /* 0: wait, 1: go, 2: cancelled */
    auto run = [&](size_t d, bool gated) {
        if (gated) {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return state != 0; });
            if (state == 2) return;
        }
        g_err[0] = 0;
        /* (fn meets the other shards at barriers: it must not throw between them -- the joint iteration allocates nothing;
         * what could still throw is turned into an error code here, on the thread it happened on) */
        try { rcs[d] = fn(d); }
        catch (const std::bad_alloc &) { rcs[d] = LZ77X_E_NOMEM; snprintf(g_err, sizeof g_err, "out of host memory"); }
        catch (...) { rcs[d] = LZ77X_E_HIP; snprintf(g_err, sizeof g_err, "host exception on a shard's thread"); }
        if (rcs[d]) msgs[d] = g_err;
    };
    std::vector<std::thread> th;
    bool ok = true;
    try {
        th.reserve(D);
        for (size_t d = 1; d < D; d++) th.emplace_back(run, d, true);
    } catch (...) { ok = false; }
    { std::lock_guard<std::mutex> lk(m); state = ok ? 1 : 2; }
    cv.notify_all();
    if (ok && D) run(0, false);
    for (auto &t : th) t.join();
    if (!ok) { snprintf(g_err, sizeof g_err, "could not start a host thread per shard"); return LZ77X_E_NOMEM; }
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
extern __thread CtxSet *tl_set;    /* the set leased by this thread's call */
extern std::vector<CtxSet *> g_pool;
extern std::mutex g_mu;
extern std::condition_variable g_cv;

void poison_set(CtxSet &S);

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
        lk.unlock();
        if (poison_on()) poison_set(*set);
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






inline size_t stream_bytes(uint64_t ntok, int T) { return 4 + (size_t)((ntok * (uint64_t)T + 7) / 8); }

/* ---------------------------------------------------------------- encode ------------ */

/* Contexts taking part in one encode: cs[0] is the caller's device (holds the pinned host buffers
 * and the final stream), cs[1..] the other shards.  LZ77X_FAKE_DEVICES=k lets k contexts share one
 * physical GPU so that the multi-device path can be exercised on a single-GPU box. */

/* ---- hostio.cpp ------------------------------------------------------------------------------------------------------ */
/* (st: an IDLE stream of the context's device the copies may use -- the caller has synchronised it -- or null: the context's
 * staging stream, created on first use; a stream costs a short-lived process 8 ms) */
int fetch_result(Ctx &c, uint8_t *dst, const void *d_src, size_t bytes, hipStream_t st = nullptr);
int upload_pageable(Ctx &c, uint8_t *d_dst, const uint8_t *h_src, size_t bytes, hipStream_t st = nullptr);
extern "C" uint64_t lz77x_shard_token_cut(uint64_t ntok, int shards, int d);
extern "C" void lz77x_shard_compose_tail(const uint16_t *map, int sb, const uint8_t *incoming, uint8_t *outgoing);
extern "C" void lz77x_shard_compose_tail32(const uint32_t *map, int sb, const uint8_t *incoming, uint8_t *outgoing);

/* A regular file behind a FILE* can be read and written at offsets by several threads at once (CopyPool::read_at /
 * write_at: the page cache hands out 4-5 GB/s to one thread): its descriptor and position, or fd = -1 for anything else
 * (a pipe, a cookie stream such as the shim's bitFILE, a file opened for appending), which keeps fread / fwrite.  The
 * FILE's own position is put where the descriptor's work ended (raw_done). */
struct RawFile { int fd = -1; off_t off = 0; };
RawFile raw_file(FILE *f, bool writing);
bool raw_done(FILE *f, const RawFile &r);
int stream_in(Ctx &c, FILE *f, DevBuf &dst, size_t slack, size_t *n_out);
int stream_out(Ctx &c, FILE *f, const void *d_src, size_t bytes, hipStream_t st = nullptr /* an idle stream to copy on, or null: the staging stream */);

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

/* Host memory in front of the sharded encode: the bytes [off, off + got) of the input for increasing off (a stretch of a long
 * stream and the look-back of the next one overlap); the pointer stays valid until the next get().  Out of the caller's
 * buffer it is a pointer into it; out of a FILE* one stretch is held and what the next one needs again is moved to the front
 * (the reference streams through a 3*SB+LA window, lz77.c:113-129: a file of any length, bounded host memory). */
struct HostWindow {
    virtual ~HostWindow() {}
    virtual int get(uint64_t off, size_t want, const uint8_t **p, size_t *got) = 0;
    virtual bool from_file() const { return false; }
};

struct MemWindow : HostWindow {
    const uint8_t *base; size_t n;
    MemWindow(const uint8_t *b, size_t bytes) : base(b), n(bytes) {}
    int get(uint64_t off, size_t want, const uint8_t **p, size_t *got) override
    {
        const size_t left = off < n ? n - (size_t)off : 0;
        *p = base + (off < n ? off : n);
        *got = want < left ? want : left;
        return LZ77X_OK;
    }
};

struct FileWindow : HostWindow {
    FILE *f;
    uint8_t *buf = nullptr; size_t cap = 0, have = 0;      /* buf[0, have) = the file's bytes [at, at + have) */
    uint64_t at = 0;
    bool eof = false;
    explicit FileWindow(FILE *file) : f(file) {}
    ~FileWindow() override { free(buf); }
    bool from_file() const override { return true; }
    int get(uint64_t off, size_t want, const uint8_t **p, size_t *got) override
    {
        if (off < at) { snprintf(g_err, sizeof g_err, "FileWindow: offsets must not decrease"); return LZ77X_E_ARG; }
        const size_t drop = (size_t)(off - at) < have ? (size_t)(off - at) : have;
        if (drop) { memmove(buf, buf + drop, have - drop); have -= drop; at += drop; }
        if (at < off) {                                       /* (never: consecutive stretches overlap) */
            snprintf(g_err, sizeof g_err, "FileWindow: a gap between stretches");
            return LZ77X_E_ARG;
        }
        if (want > cap) {
            uint8_t *nb = (uint8_t *)realloc(buf, want);
            if (!nb) return LZ77X_E_NOMEM;
            buf = nb;
            cap = want;
        }
        while (have < want && !eof) {
            const size_t r = fread(buf + have, 1, want - have, f);
            have += r;
            if (r == 0) {
                if (ferror(f)) return LZ77X_E_IO;
                eof = true;
            }
        }
        *p = buf;
        *got = have < want ? have : want;
        return LZ77X_OK;
    }
};

/* ---- the pipelines ----------------------------------------------------------------------------------------------------- */
/* encode_host.cpp */
int encode_core_host(std::vector<Ctx *> &cs, const void *src, bool src_on_device, size_t n, const lz77x_geom &g, hipStream_t s, size_t *zn);
/* encode_pipe.cpp */
int encode_stream_device(Ctx &c, Source &src, Sink &sink, const lz77x_geom &g, hipStream_t s, bool *fallback, size_t *n_fallback);
/* decode_pipe.cpp */
int decode_stream(Ctx &c, Source &src, Sink *sink, hipStream_t s, uint64_t *n_out);
/* shard.cpp */
int decode_sharded(std::vector<Ctx *> &cs, const uint8_t *z, size_t zn, uint8_t **out, size_t *out_n, int *handled);
int encode_sharded(std::vector<Ctx *> &cs, HostWindow &in, const lz77x_geom &g, Sink &sink);
int shard_plan_range(size_t nbytes, size_t t0, size_t t1, int sb, int la, int shards, lz77x_shard *out);
bool device_pipeline_ok(size_t ndev, const lz77x_geom &g);
int encode_any(std::vector<Ctx *> &cs, const void *src, bool src_on_device, size_t n, const lz77x_geom &g, hipStream_t s, Sink &sink);

}  // namespace lz77x_host

#endif
