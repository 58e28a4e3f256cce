/*
 * main.c -- command line of the MI355X LZ77 codec; same contract as the reference's
 * main.c:59-180: flags -c -d -i -o -l -s -h, same diagnostics on stderr,
 * exit status 1 on error, silent success.  Host code is plain C and reaches the GPU only
 * through the C ABI in include/lz77_mi355x.h.
 *
 * Deliberate differences: "-s 0" (which makes the reference divide by zero, tree.c:66) is
 * refused with the search-buffer diagnostic; device/runtime failures are reported on stderr.
 */
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
    fclose(fin);
    if (fclose(fout) != 0 && rc == LZ77X_OK) rc = LZ77X_E_IO;
    if (rc != LZ77X_OK) {
        const char *detail = lz77x_last_error();
        fprintf(stderr, "lz77: %s%s%s\n", lz77x_strerror(rc), detail[0] ? ": " : "", detail);
        return EXIT_FAILURE;
    }
    return 0;
}
