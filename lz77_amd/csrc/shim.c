/*
 * shim.c -- the reference's two hot-path SYMBOLS on top of liblz77_mi355x.so, so that cstdvd/lz77's
 * own main.c links and runs unmodified (INTEGRATION.md option B):
 *
 *     void encode(FILE *file, struct bitFILE *out, int la, int sb);     lz77.h:14, called at main.c:150
 *     void decode(struct bitFILE *file, FILE *out);                     lz77.h:15, called at main.c:161
 *
 * main() opens the compressed side with bitIO_open() (bitio.c:124) and closes it with bitIO_close()
 * (bitio.c:171), so `struct bitFILE` stays the reference's own opaque type and the stream crosses it
 * through the reference's own accessors, bitIO_write() / bitIO_read() (bitio.h:30-31; declared here by
 * prototype, resolved from the reference's bitio.o at link time).  Nothing of tree.c or lz77.c is linked.
 *
 * Link:  gcc main.c bitio.c lz77_shim.o -Llz77_amd -llz77_mi355x -lm        (oracle/Makefile: ref-shim)
 *
 * Error behaviour follows lz77.c: void returns; a failed read of the input prints the reference's
 * message on stdout and returns (lz77.c:79-82); device failures are reported on stderr and end the
 * process with EXIT_FAILURE like the reference's own fatal path (lz77.c:273-277).
 */
#define _GNU_SOURCE                                                     /* fopencookie */
#include "../../include/lz77_mi355x.h"
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>

struct bitFILE;                                                         /* bitio.h:18, opaque */
int bitIO_write(struct bitFILE *bitF, void *info, int nbit);           /* bitio.h:30 */
int bitIO_read(struct bitFILE *bitF, void *info, int info_s, int nbit); /* bitio.h:31 */

void encode(FILE *file, struct bitFILE *out, int la, int sb);
void decode(struct bitFILE *file, FILE *out);

#define SHIM_PIECE (1 << 20)                                            /* bytes per bitIO call */

static void die(const char *what, int rc)
{
    fprintf(stderr, "lz77 (MI355X): %s: %s: %s\n", what, lz77x_strerror(rc), lz77x_last_error());
    exit(EXIT_FAILURE);
}

/* The compressed side as a stdio stream: a FILE* whose writes / reads go through the reference's own bitIO_write /
 * bitIO_read, so that lz77x_encode_file / lz77x_decode_file STREAM both sides (segments through bounded device
 * memory, two pinned staging slots) exactly as they do for the CLI -- the first version read the whole input and
 * held the whole stream in host memory. */
static ssize_t bit_sink(void *cookie, const char *buf, size_t size)
{
    struct bitFILE *out = (struct bitFILE *)cookie;
    for (size_t at = 0; at < size;) {                                   /* byte-aligned appends; bitIO_close pads nothing more */
        const size_t m = size - at < SHIM_PIECE ? size - at : SHIM_PIECE;
        if (bitIO_write(out, (void *)(buf + at), (int)(8 * m)) != (int)(8 * m)) return 0;   /* short write: bitio.c:87-88 stays silent too */
        at += m;
    }
    return (ssize_t)size;
}

static ssize_t bit_source(void *cookie, char *buf, size_t size)
{
    struct bitFILE *in = (struct bitFILE *)cookie;
    size_t got = 0;
    while (got < size) {
        const size_t m = size - got < SHIM_PIECE ? size - got : SHIM_PIECE;
        const int bits = bitIO_read(in, buf + got, (int)m, (int)(8 * m));
        if (bits <= 0) break;
        got += (size_t)bits / 8;                                        /* a partial trailing byte cannot hold a token */
        if (bits < (int)(8 * m)) break;
    }
    return (ssize_t)got;
}

void encode(FILE *file, struct bitFILE *out, int la, int sb)
{
    cookie_io_functions_t io = {NULL, bit_sink, NULL, NULL};
    FILE *sink = fopencookie(out, "w", io);
    if (!sink) { fprintf(stderr, "lz77 (MI355X): out of memory\n"); exit(EXIT_FAILURE); }
    const int rc = lz77x_encode_file(file, sink, la, sb);               /* -1 = default like lz77.c:65-66 */
    const int in_failed = ferror(file);                                 /* which side LZ77X_E_IO came from */
    fclose(sink);                                                       /* (flushes stdio's buffer into bitIO_write; `out` stays main()'s) */
    if (rc == LZ77X_E_IO) {
        /* a failed read of the input is the reference's one message (lz77.c:79-82, on stdout); a short write of the stream
         * passes in silence there (bitio.c:87-88, 231-232) and does here */
        if (in_failed) printf("Error loading the data in the window.\n");
        return;
    }
    if (rc != LZ77X_OK) die("encode", rc);
}

void decode(struct bitFILE *file, FILE *out)
{
    cookie_io_functions_t io = {bit_source, NULL, NULL, NULL};
    FILE *src = fopencookie(file, "r", io);
    if (!src) { fprintf(stderr, "lz77 (MI355X): out of memory\n"); exit(EXIT_FAILURE); }
    const int rc = lz77x_decode_file(src, out);
    const int in_failed = ferror(src);
    fclose(src);
    if (rc == LZ77X_E_FORMAT) return;                                   /* shorter than its header: nothing to write */
    if (rc == LZ77X_E_IO) {
        /* lz77.c:273-277: a read error of the compressed side is perror("Error reading bits.\\n") + exit(EXIT_FAILURE); a failed putc
         * of the output is not looked at by the reference at all */
        if (in_failed) { perror("Error reading bits.\n"); exit(EXIT_FAILURE); }
        return;
    }
    if (rc != LZ77X_OK) die("decode", rc);
}
