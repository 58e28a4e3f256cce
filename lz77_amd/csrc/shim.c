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
#include "../../include/lz77_mi355x.h"
#include <stdlib.h>
#include <string.h>

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

static uint8_t *grow(uint8_t *buf, size_t *cap, size_t need)
{
    if (need <= *cap) return buf;
    size_t ncap = *cap ? *cap : (size_t)1 << 20;
    while (ncap < need) ncap *= 2;
    uint8_t *nb = (uint8_t *)realloc(buf, ncap);
    if (!nb) { free(buf); fprintf(stderr, "lz77 (MI355X): out of memory\n"); exit(EXIT_FAILURE); }
    *cap = ncap;
    return nb;
}

void encode(FILE *file, struct bitFILE *out, int la, int sb)
{
    uint8_t *data = NULL, *z = NULL;
    size_t cap = 0, n = 0, zn = 0;
    for (;;) {                                                          /* the whole input (lz77.c:78,121 read it window by window) */
        data = grow(data, &cap, n + SHIM_PIECE);
        const size_t got = fread(data + n, 1, SHIM_PIECE, file);
        n += got;
        if (got < SHIM_PIECE) {
            if (ferror(file)) { printf("Error loading the data in the window.\n"); free(data); return; }   /* lz77.c:79-82 */
            break;
        }
    }
    const int rc = lz77x_encode(data, n, sb, la, &z, &zn);              /* -1 = default like lz77.c:65-66 */
    free(data);
    if (rc != LZ77X_OK) die("encode", rc);
    for (size_t at = 0; at < zn;) {                                     /* byte-aligned appends; bitIO_close pads nothing more */
        const size_t m = zn - at < SHIM_PIECE ? zn - at : SHIM_PIECE;
        if (bitIO_write(out, z + at, (int)(8 * m)) != (int)(8 * m)) break;   /* short write: bitio.c:87-88 stays silent too */
        at += m;
    }
    lz77x_free(z);
}

void decode(struct bitFILE *file, FILE *out)
{
    uint8_t *z = NULL, *data = NULL;
    size_t cap = 0, zn = 0, n = 0;
    for (;;) {
        z = grow(z, &cap, zn + SHIM_PIECE);
        const int bits = bitIO_read(file, z + zn, SHIM_PIECE, 8 * SHIM_PIECE);
        if (bits <= 0) break;
        zn += (size_t)bits / 8;                                         /* a partial trailing byte cannot hold a token */
        if (bits < 8 * SHIM_PIECE) break;
    }
    const int rc = lz77x_decode(z, zn, &data, &n);
    free(z);
    if (rc == LZ77X_E_FORMAT) return;                                   /* shorter than its header: nothing to write */
    if (rc != LZ77X_OK) die("decode", rc);
    if (n && fwrite(data, 1, n, out) != n) perror("Writing output file");
    lz77x_free(data);
}
