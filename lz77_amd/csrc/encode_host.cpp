/*
 * encode_host.cpp -- the host-assisted encode of round 1 (lz77.c:51-140 with its two sequential loops on a host core, hoststage.c): taken only when
 * the device's gate iteration gives up, or when the variants build forces it (LZ77X_HOST_STAGEB=1: the cross-check).
 */
#include "host.h"

LZ77X_HOST_NS {

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

}  // namespace lz77x_host
