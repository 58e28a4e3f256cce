/*
 * fileio.c -- whole-file-in-host-memory form of lz77x_encode_file (plain C).  The public entry point
 * (api.cpp, encode_pipe.cpp) streams the file through two pinned staging slots instead and only comes here when one
 * stream is cut into several shards (LZ77X_SHARDS > 1: the other devices need a host copy of the input).
 */
#include "../../include/lz77_mi355x.h"
#include "lz77x_internal.h"
#include <stdlib.h>
#include <string.h>

static int slurp(FILE *f, uint8_t **data, size_t *n)
{
    size_t cap = 1 << 20, len = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    if (!buf) return LZ77X_E_NOMEM;
    for (;;) {
        if (len == cap) {
            size_t ncap = cap * 2;
            uint8_t *nb = (uint8_t *)realloc(buf, ncap);
            if (!nb) { free(buf); return LZ77X_E_NOMEM; }
            buf = nb;
            cap = ncap;
        }
        size_t got = fread(buf + len, 1, cap - len, f);
        len += got;
        if (got == 0) {
            if (ferror(f)) { free(buf); return LZ77X_E_IO; }
            break;
        }
    }
    *data = buf;
    *n = len;
    return LZ77X_OK;
}

static int spill(FILE *f, const uint8_t *data, size_t n)
{
    if (n && fwrite(data, 1, n, f) != n) return LZ77X_E_IO;
    if (fflush(f) != 0) return LZ77X_E_IO;
    return LZ77X_OK;
}

int lz77x_encode_file_buffered(FILE *in, FILE *out, int la, int sb)
{
    if (!in || !out) return LZ77X_E_ARG;
    uint8_t *data = NULL, *z = NULL;
    size_t n = 0, zn = 0;
    int rc = slurp(in, &data, &n);
    if (rc) return rc;
    rc = lz77x_encode(data, n, sb, la, &z, &zn);
    free(data);
    if (rc) return rc;
    rc = spill(out, z, zn);
    lz77x_free(z);
    return rc;
}
