/*
 * main.c -- command line of the MI355X LZ77 codec; same contract as the reference's
 * main.c:59-180: flags -c -d -i -o -l -s -h, same diagnostics on stderr,
 * exit status 1 on error, silent success.  Host code is plain C and reaches the GPU only
 * through the C ABI in include/lz77_mi355x.h.
 *
 * I/O errors keep the reference's contract too (lz77.c:79-82, 273-277; bitio.c:87-88): see the end of main().
 *
 * Deliberate differences: "-s 0" (which makes the reference divide by zero, tree.c:66) is
 * refused with the search-buffer diagnostic; device/runtime failures and streams the reference would
 * misread (shorter than a header: it decodes uninitialised geometry, lz77.c:157-158) are reported on stderr.
 */
#define _POSIX_C_SOURCE 200809L
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "../../include/lz77_mi355x.h"

enum action { ACT_NONE, ACT_PACK, ACT_UNPACK };

static void usage(void)
{
    fputs("Usage: lz77 <options>\n"
          "  -c : Encode input file to output file.\n"
          "  -d : Decode input file to output file.\n"
          "  -i <filename> : Name of input file.\n"
          "  -o <filename> : Name of output file.\n"
          "  -l <value> : Lookahead size (default 15)\n"
          "  -s <value> : Search-buffer size (default 4095)\n"
          "  -h : Command line options.\n\n", stdout);
}

/* LZ77X_TRACE=1 with LZ77X_T0=<ns since the epoch at which the caller started us> (tools/cli_trace.sh): where the wall
 * time of a run goes outside the library -- loading the HIP runtime before main(), tearing it down after */
static void stamp(const char *what)
{
    const char *tr = getenv("LZ77X_TRACE"), *t0 = getenv("LZ77X_T0");
    if (!tr || !atoi(tr) || !t0) return;
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    const double ms = ((double)ts.tv_sec * 1e9 + (double)ts.tv_nsec - strtod(t0, NULL)) / 1e6;
    fprintf(stderr, "[lz77 ] %-28s %8.2f ms since the caller's clock\n", what, ms);
}

static int fail(const char *msg)
{
    fprintf(stderr, "%s\n", msg);
    return EXIT_FAILURE;
}

int main(int argc, char **argv)
{
    enum action act = ACT_NONE;
    const char *src = NULL, *dst = NULL;
    int la = -1, sb = -1, ch;
    stamp("main() entered");

    while ((ch = getopt(argc, argv, "cdi:o:l:s:h")) != -1) {
        switch (ch) {
        case 'c': act = ACT_PACK; break;
        case 'd': act = ACT_UNPACK; break;
        case 'i':
            if (src) return fail("Multiple input files not allowed.");
            src = optarg;
            break;
        case 'o':
            if (dst) return fail("Multiple output files not allowed.");
            dst = optarg;
            break;
        case 'l':
            la = atoi(optarg);
            if (la < 2 || la > 255) return fail("Bad lookahead size value.");
            break;
        case 's':
            sb = atoi(optarg);
            if (sb < 1 || sb > 65535) return fail("Bad search-buffer size value.");
            break;
        case 'h': usage(); break;
        default: break;                     /* getopt already complained; the reference carries on */
        }
    }
    if (!src) return fail("Input file must be provided");
    if (!dst) return fail("Output file must be provided");
    if (act == ACT_NONE) return fail("Select ENCODE or DECODE mode");

    FILE *fin = fopen(src, "rb");
    if (!fin) { perror("Opening input file"); return EXIT_FAILURE; }
    FILE *fout = fopen(dst, "wb");
    if (!fout) { perror("Opening output file"); fclose(fin); return EXIT_FAILURE; }

    int rc = act == ACT_PACK ? lz77x_encode_file(fin, fout, la, sb) : lz77x_decode_file(fin, fout);
    stamp("library call returned");
    const int in_failed = ferror(fin);                  /* which side LZ77X_E_IO came from */
    fclose(fin);
    const int close_failed = fclose(fout) != 0;
    stamp("files closed");
    if (rc == LZ77X_E_IO) {
        /* the reference's own contract for I/O errors.  Encode (lz77.c:79-82): a failed read of the input is one line on
         * STDOUT, encode() returns and main() leaves with 0; a short write of the stream passes in silence (bitio.c:87-88,
         * 231-232).  Decode (lz77.c:273-277): a failed read of the stream is perror("Error reading bits.\n") and
         * exit(EXIT_FAILURE); a failed putc of the output is not looked at. */
        if (act == ACT_PACK) {
            if (in_failed) fputs("Error loading the data in the window.\n", stdout);
            return EXIT_SUCCESS;
        }
        if (in_failed) { perror("Error reading bits.\n"); return EXIT_FAILURE; }
        return EXIT_SUCCESS;
    }
    if (rc != LZ77X_OK) {
        const char *detail = lz77x_last_error();
        fprintf(stderr, "lz77: %s%s%s\n", lz77x_strerror(rc), detail[0] ? ": " : "", detail);
        return EXIT_FAILURE;
    }
    (void)close_failed;                                 /* main.c:167-168 of the reference ignores fclose() as well */
    /* both files are closed and flushed: leave without the HIP runtime's atexit teardown (60-100 ms of a run whose kernels
     * take 12; the driver reclaims the device memory with the process either way).  LZ77X_FAST_EXIT=0 -- and any run under
     * LZ77X_TRACE -- returns normally instead, so that atexit handlers run (rocprofv3 / roctracer flushes, gcov, LSan,
     * LD_PRELOAD tools); INTEGRATION.md lists the knob. */
    fflush(NULL);
    const char *fe = getenv("LZ77X_FAST_EXIT"), *tr = getenv("LZ77X_TRACE");
    if ((fe && !atoi(fe)) || (tr && atoi(tr))) return EXIT_SUCCESS;
    _exit(0);
}
