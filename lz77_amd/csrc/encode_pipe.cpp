/*
 * encode_pipe.cpp -- the device pipeline of an encode (replaces lz77.c:51-140): match -> parse chain -> priority recurrence -> tie-break -> pack, an input
 * of any length in segments, two in flight.
 */
#include "host.h"

LZ77X_HOST_NS {

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
    /* Round 6: the match stage's scratch (16 B per position at C1, the largest buffer of an encode) is dead once its last
     * launch is enqueued, and everything the later stages of THIS segment allocate -- xval, the parse chain, the token words,
     * the recurrence's gates and maps, the hand-over index -- is only touched by kernels behind it on the same stream: those
     * arrays are carved out of the scratch buffer (stream order is the only synchronisation needed; a piece that does not
     * fit keeps its own cached buffer).  `out` is NOT among them: its words leave through the drain thread while this
     * context's next match stage already runs. */
    uint8_t *a_base = nullptr;
    size_t a_cap = 0, a_at = 0;
    uint32_t *d_xval = nullptr, *d_chain = nullptr, *d_tokval = nullptr, *d_ofs = nullptr, *d_tstart = nullptr;
    uint2 *d_ent = nullptr;
    void *d_prio_tmp = nullptr, *d_chain_tmp = nullptr, *d_scantmp = nullptr, *d_index = nullptr;
    /* bytes for a later stage: from the dead scratch, else from the stage's own cached buffer; null: out of memory (g_err set) */
    void *place(DevBuf &own, size_t bytes)
    {
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (a_base && a_at + need <= a_cap) { void *q = a_base + a_at; a_at += need; return q; }
        if (own.need(bytes) != LZ77X_OK) return nullptr;
        return own.p;
    }
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
    size_t scratch_limit = 0;
    {
        /* 3 GB of scratch: 100 MB at C1 (8138 regions of 287 KB) in ONE launch -- the walkers are latency bound (a
         * launch takes its fill + 2048 steps whatever its size), a second launch is a second 0.75 ms */
        const size_t per = lz77k_match_scratch_bytes(g, 1);
        /* (large windows: LZ77X_BIG_SCRATCH_GB -- a region's scratch is 16x a small window's; host.h has why it is 12 and not 16) */
        const size_t cap = J.scratch_cap ? J.scratch_cap : (size_t)(g.fast ? 3 : LZ77X_BIG_SCRATCH_GB) << 30;      /* (encode_mem_plan lowers it on a tight device) */
        const uint32_t fit = (uint32_t)(cap / per);
        if (batch > fit) batch = fit ? fit : 1;
        scratch_limit = cap;
        const char *gs = getenv("LZ77X_MATCH_BATCH");
        if (gs && atoi(gs) > 0 && (uint32_t)atoi(gs) < batch) batch = (uint32_t)atoi(gs);
    }
    const size_t np = (size_t)J.nloc;
    if ((rc = c.scratch.need(lz77k_match_scratch_bytes(g, batch), scratch_limit))) return rc;
    if ((rc = c.ps.need((np + 8) * 4))) return rc;
    if ((rc = c.maxlen.need(np + 64))) return rc;
    /* the regions' sorted order stays resident for the tie-break (RP uint16 per region: 2.7 B per input byte) */
    const bool keep_order = J.tvariant == 0 && !(g.fast && LZ77X_VENV("LZ77X_TOKENS_BUCKET"));
    /* large windows: rank + inverse arrays, (2RP + 8) words per region, for the rank-order tie-break */
    if (keep_order && (rc = c.ranks_all.need(g.fast ? (size_t)nregions * g.RP * 2 + 64 : (size_t)nregions * (2 * (size_t)g.RP + 8) * sizeof(uint32_t)))) return rc;
    J.d_order = keep_order ? c.ranks_all.as<uint32_t>() : nullptr;
    {
        const char *ae = getenv("LZ77X_NO_ALIAS");           /* (memory probe: every stage in its own cached buffer, as until round 5) */
        J.a_base = ae && atoi(ae) ? nullptr : c.scratch.as<uint8_t>();
        J.a_cap = c.scratch.cap;
        J.a_at = 0;
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
        if (!(J.d_xval = reinterpret_cast<uint32_t *>(J.place(c.xval, (np + 8) * 4)))) return LZ77X_E_HIP;
        if (!(J.d_chain = reinterpret_cast<uint32_t *>(J.place(c.chain, (np + 8) * 4)))) return LZ77X_E_HIP;
        if ((rc = c.flag.need(1024))) return rc;          /* [64, 64 + 8 * 33): the hand-over count and the slots the tiles spread it over */
        if (!(J.d_prio_tmp = J.place(c.prio_tmp, lz77k_prio_tmp_bytes(J.nx, g.sb)))) return LZ77X_E_HIP;
        if (!(J.d_chain_tmp = J.place(c.chain_tmp, lz77k_chain_tmp_bytes(E - start, g.la)))) return LZ77X_E_HIP;
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
        HIPCHK(lz77k_chain(c.maxlen.as<uint8_t>(), E, g.la, J.d_chain, J.d_chain_tmp, sc, &d_tbase, &nsub, start, &d_exit));
        HIPCHK(hipEventRecord(c.match_ev[1], sc));
        HIPCHK(small_d2h(c.h_tbase, h_tbase, d_tbase, ((size_t)nsub + 1) * 4, sc));
        HIPCHK(small_d2h(c.h_small, c.h_small.as<uint32_t>() + 16, d_exit, 4, sc));
        HIPCHK(hipEventRecord(c.pipe_ev[2], sc));
        J.nsub = nsub;

#ifdef LZ77X_VARIANTS
        static hipEvent_t probe_ev = nullptr;
        bool probe_on = false;
        if (LZ77X_VENV("LZ77X_TS_OVERLAP_PROBE") && lz77k_tokens_builds_lists(g, 0, J.d_order)) {
            /* TIMING PROBE (wrong priorities, its token words are overwritten by the real launch later): what does the
             * tie-break cost the recurrence, and the recurrence the tie-break, when they share the chip?   */
            static hipStream_t probe_stream = nullptr;
            if (!probe_ev) HIPCHK(hipEventCreateWithFlags(&probe_ev, hipEventDisableTiming));
            if (!probe_stream) {
                int lo = 0, hi = 0;
                HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
                HIPCHK(hipStreamCreateWithPriority(&probe_stream, hipStreamNonBlocking, atoi(LZ77X_VENV("LZ77X_TS_OVERLAP_PROBE")) == 2 ? hi : lo));
            }
            HIPCHK(hipEventSynchronize(c.pipe_ev[2]));
            const uint32_t ntok_p = h_tbase[nsub];
            if ((rc = c.tokval.need((np + 16) * 4))) return rc;
            if ((rc = c.tstart.need(lz77k_tokens_tmp_bytes((uint32_t)np + 2 * (uint32_t)usb + 16, g)))) return rc;
            HIPCHK(hipStreamWaitEvent(probe_stream, c.pipe_ev[2], 0));
            HIPCHK(lz77k_tokens(c.in.as<uint8_t>(), J.nloc, g, J.d_chain, ntok_p, c.maxlen.as<uint8_t>(), nullptr, nullptr, 0u, start, E,
                                c.tokval.as<uint32_t>() + 4, c.tstart.as<uint32_t>(), nullptr, 0, probe_stream, nullptr, J.d_order, J.first ? nullptr : look_cur,
                                J.nlook, 0u, c.ps.as<uint32_t>(), J.d_xval, nullptr));
            HIPCHK(hipEventRecord(probe_ev, probe_stream));
            probe_on = true;
        }
#endif
        /* -- priority recurrence (tree.c:202-231) over steps [0, nx) from the carried cells -- */
        int iters = 0, converged = 1;
        /* a sweep finalises at least one more block: it always ends.  The guard before the host loop takes over comes from
         * what was measured (profiles/r03_prio_classes.json, tools/time_c2.py per data class): 4-7 iterations on every class at
         * C1, 9-13 at C2 except record-structured data (31: the flips decay slowly but steadily, and the seven iterations
         * past 24 are cheaper than starting over on the host).  Twice the largest count seen; an iteration costs 1/20
         * (C2) to 1/100 (C1, text) of the host loop, so a pathological input is bounded at about three times its cost */
        int max_iters = 64;
        {
            const char *me = getenv("LZ77X_PRIO_MAX_ITERS");
            if (me && atoi(me) > 0) max_iters = atoi(me);
        }
        HIPCHK(hipEventRecord(c.match_ev[2], s));
        const double tw0 = now_ms();
        float prio_ms3[3] = {0, 0, 0};
        HIPCHK(lz77k_prio(c.ps.as<uint32_t>(), J.nx, g.sb, J.d_xval, J.d_prio_tmp, s, c.h_small.as<uint32_t>() + 8, max_iters,
                          &iters, &converged, &c.match_ev[4], prio_ms3, 0u, J.first ? nullptr : look_cur, J.last ? nullptr : look_next));
        HIPCHK(hipEventRecord(c.match_ev[3], s));
#ifdef LZ77X_VARIANTS
        if (probe_on) HIPCHK(hipStreamWaitEvent(s, probe_ev, 0));     /* the real tie-break starts when the probe's is through */
#endif
        bool host_cells = false;
        if (!converged && !allow_fallback) {
            /* A segment of a multi-segment input whose gate iteration gave up (an error front: input that repeats with a
             * period of about a window, lz77k_prio): the whole encode cannot start over on the host-assisted pipeline -- its
             * predecessors' words have left -- so THIS segment's recurrence alone goes to a host core: ps out, the exact loop
             * (hoststage.c lz77x_prio_run_cells, from the carried cells), xval and the cells it leaves behind back.  Round 4
             * iterated without a bound here: a block per iteration, 16 K iterations for a segment of 2^30 positions. */
            std::vector<uint32_t> h_ps, h_xv, cells_out(usb);
            try { h_ps.resize((size_t)J.nx + 16); h_xv.resize((size_t)J.nx + 16); } catch (...) { return LZ77X_E_NOMEM; }
            HIPCHK(hipMemcpyAsync(h_ps.data(), c.ps.p, (size_t)J.nx * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            const double th = now_ms();
            if (!lz77x_prio_run_cells(h_ps.data(), J.nx, g.sb, J.first ? nullptr : carry.cells.data(), 0u, h_xv.data(), cells_out.data())) return LZ77X_E_NOMEM;
            g_stats.host_stageb_ms += now_ms() - th;
            HIPCHK(hipMemcpyAsync(J.d_xval, h_xv.data(), (size_t)J.nx * 4, hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));                    /* (pageable source) */
            memcpy(h_state, cells_out.data(), usb * 4);
            host_cells = true;
            converged = 1;
        }
        if (!J.last && !host_cells) HIPCHK(small_d2h(c.h_tbase, h_state, look_next, usb * 4, s));
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
        const bool fused_lists = lz77k_tokens_builds_lists(g, J.tvariant, J.d_order);
        if (!(J.d_tokval = reinterpret_cast<uint32_t *>(J.place(c.tokval, (np + 16) * 4)))) return LZ77X_E_HIP;
        if (!(J.d_tstart = reinterpret_cast<uint32_t *>(J.place(c.tstart, lz77k_tokens_tmp_bytes((uint32_t)idx_span, g))))) return LZ77X_E_HIP;
        J.d_ofs = nullptr; J.d_ent = nullptr; J.d_scantmp = nullptr; J.d_index = nullptr;
        if (!fused_lists) {
            /* (LDS-sized windows build their hand-over lists per tile: no index in HBM) */
            if (!(J.d_ofs = reinterpret_cast<uint32_t *>(J.place(c.ofs, (idx_span + 8) * 4)))) return LZ77X_E_HIP;
            if (!(J.d_ent = reinterpret_cast<uint2 *>(J.place(c.ent, (idx_span + 8) * 8)))) return LZ77X_E_HIP;
            if (!(J.d_scantmp = J.place(c.scantmp, lz77k_scan_tmp_bytes((uint32_t)idx_span + 1)))) return LZ77X_E_HIP;
        }
        if (!g.fast) {
            /* large windows: the (block, first byte) buckets of the tokens of length one and the hand-overs by rank (built per token chunk) */
            const size_t ib = lz77k_tokens_index_bytes(g, (chunk_pos < span ? chunk_pos : span) + csub);
            if (ib && !(J.d_index = J.place(c.bidx, ib))) return LZ77X_E_HIP;
        }
        while (c.tie_ev.size() < 2 * (size_t)nchunks + 8) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c.tie_ev.push_back(e); }
        uint32_t *look_cur = c.look.as<uint32_t>();
        const uint32_t *h_tbase = c.h_tbase.as<uint32_t>();
        /* -- tokens: per chunk, the hand-over index of the evictions that can matter and the tie-break.  Tokens
         *    land behind four slots that hold the predecessor's last tokens (for the first stream word) -- */
        uint32_t *tokbuf = J.d_tokval;
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
            const bool fused = fused_lists;
            if (!fused)
                HIPCHK(lz77k_xfer_index(c.ps.as<uint32_t>(), J.d_xval, xa, (uint32_t)x_done, dbase, (uint32_t)e, J.d_ofs,
                                        J.d_ent, J.d_scantmp, s, (uint32_t)x_new, c.flag.as<unsigned long long>() + 8, g.fast ? (uint32_t)g.sb : 0u));
            HIPCHK(lz77k_tokens(c.in.as<uint8_t>(), J.nloc, g, J.d_chain + ta, tb - ta, c.maxlen.as<uint8_t>(), J.d_ofs,
                                J.d_ent, dbase, (uint32_t)b, (uint32_t)e, tokbuf + 4 + ta, J.d_tstart,
                                g.fast ? nullptr : J.d_index,
                                J.tvariant, s, &c.tie_ev[2 * ci], J.d_order, J.first ? nullptr : look_cur, J.nlook, 0u,
                                fused ? c.ps.as<uint32_t>() : nullptr, fused ? J.d_xval : nullptr, c.flag.as<unsigned long long>() + 8));
            J.tie_timed[ci] = tb > ta;
        }
        HIPCHK(small_d2h(c.h_small, c.h_small.as<unsigned long long>() + 2, c.flag.as<unsigned long long>() + 8, 8, s));
    } else {
        if (!(J.d_tokval = reinterpret_cast<uint32_t *>(J.place(c.tokval, 64)))) return LZ77X_E_HIP;
        uint32_t *tokbuf = J.d_tokval;
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
    HIPCHK(lz77k_pack_range(J.d_tokval + 4 - carry.ntail, K0 - carry.ntail, K1, g, c.out.as<uint32_t>(), wlo, nw, s));
    HIPCHK(hipEventRecord(c.ev[3], s));
    /* carry: the last four tokens seen so far */
    J.have_tail = ntok + carry.ntail < 4 ? ntok + carry.ntail : 4;
    if (J.have_tail)
        HIPCHK(small_d2h(c.h_small, c.h_small.as<uint32_t>() + 20, J.d_tokval + 4 + ntok - J.have_tail, J.have_tail * 4, s));
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
        /* (large windows: 64 M -- a chunk's hand-over index, first-byte buckets and hand-overs by rank are ~70 bytes of
         * reservation per position, and half the chunk lets them all live in the dead match scratch, SegJob::place) */
        if (!g.fast) token_chunk = (size_t)64 << 20;
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
            scratch_cap = (size_t)(g.fast ? 3 : LZ77X_BIG_SCRATCH_GB) << 30;
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

}  // namespace lz77x_host
