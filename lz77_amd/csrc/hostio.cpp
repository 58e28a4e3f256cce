/*
 * hostio.cpp -- host buffers and FILE* to and from the device through the context's two pinned staging slots (the reference
 * streams through a 3*SB+LA window, lz77.c:113-129).
 */
#include "host.h"

LZ77X_HOST_NS {

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
int stream_out(Ctx &c, FILE *f, const void *d_src, size_t bytes, hipStream_t st /* an idle stream to copy on, or null: the staging stream */)
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

}  // namespace lz77x_host
