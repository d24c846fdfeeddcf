/*
 * ref_harness.c -- TEST INFRASTRUCTURE (not product code, never shipped).
 *
 * Thin driver linked against the *unmodified* reference objects (everything in
 * /root/reference/Makefile:69-89 except main.o).  It calls the reference's own library
 * entry point memagrep() (agrep.c:3282) in memory mode and then reads the reference's
 * global query state, so that tests can pin the CPU restatement in agrep_oracle.c against
 * what the reference itself computes:
 *
 *   ref_harness tables [agrep options] PATTERN
 *       -> one JSON object with Mask[256], Init[0], Init1, NO_ERR_MASK, endposition,
 *          D_endpos, D_length, M-as-seen-by-bitap, AND, SGREP   (maskgen.c:218-266)
 *   ref_harness count  FILE [agrep options] PATTERN
 *       -> "<num_of_matched>\n"   (memory mode: free of quirk Q1)
 *   ref_harness lines  FILE [agrep options] PATTERN
 *       -> the matched records exactly as the reference prints them
 *   ref_harness membuf:N FILE [agrep options] PATTERN
 *       -> memory mode on BOTH sides: output into a caller buffer of N bytes (agrep_outbuffer,
 *          agrep.h:130 OUTPUT_OVERFLOW); prints "ret=<r> matched=<n> outlen=<bytes>\n" and the bytes
 *   ref_harness fileapi FILE [agrep options] PATTERN
 *       -> FILE mode as a host application (glimpse) uses it: registers an atexit() handler of its own, calls
 *          fileagrep() (agrep.c:3300) on FILE with output to stdout and leaves through exit(): the handler's line
 *          "host-atexit-ran" must follow the records -- a drop-in engine may not end the caller's process its own way
 *
 * The same source linked with agrep_amd/host/ref_shim.c instead of the reference's engine objects
 * (oracle/Makefile, _ref/ref_harness_gpu) exercises the shim's memory mode.
 *
 * Memory-mode contract (docs/README:104-115): the buffer begins with '\n' and has
 * writable slack behind it; a dummy existing file name must be the last argv
 * (agrep.c:2922-2935).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern unsigned Mask[256];
extern unsigned Init1, NO_ERR_MASK, Init[];
extern unsigned wildmask, endposition, D_endpos;
extern int num_of_matched, D_length, AND, SGREP, DELIMITER, NOUPPER;
extern int memagrep(int argc, char *argv[], int input_len, char *input_buffer,
                    int output_len, void *output);
extern int fileagrep(int argc, char *argv[], int output_len, void *output);

static void host_atexit(void) { fputs("host-atexit-ran\n", stdout); fflush(stdout); }

static char *slurp(const char *path, long *len_out)
{
    FILE *f = fopen(path, "rb");
    long n;
    char *buf;
    if (!f) { perror(path); exit(3); }
    fseek(f, 0, SEEK_END);
    n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf = (char *)calloc((size_t)n + 1 + 4096, 1);
    buf[0] = '\n';
    if (fread(buf + 1, 1, (size_t)n, f) != (size_t)n) { perror("read"); exit(3); }
    fclose(f);
    *len_out = n + 1;
    return buf;
}

int main(int argc, char **argv)
{
    char *av[64];
    int ac = 0, i, first_opt, ret;
    long len = 0;
    char *buf;
    const char *mode;

    if (argc < 3) {
        fprintf(stderr, "usage: %s tables|count|lines [FILE] [options] PATTERN\n", argv[0]);
        return 2;
    }
    mode = argv[1];
    av[ac++] = "agrep";
    if (strcmp(mode, "fileapi") == 0) {
        atexit(host_atexit);
        av[ac++] = "-V0";
        for (i = 3; i < argc && ac < 62; i++) av[ac++] = argv[i];
        av[ac++] = argv[2];
        av[ac] = NULL;
        ret = fileagrep(ac, av, 0, stdout);
        fflush(stdout);
        exit(ret < 0 ? 1 : 0);
    }
    if (strcmp(mode, "tables") == 0) {
        first_opt = 2;
        buf = (char *)calloc(8192, 1);
        buf[0] = '\n';
        strcpy(buf + 1, "zzzz\n");
        len = 6;
    } else {
        first_opt = 3;
        buf = slurp(argv[2], &len);
    }
    av[ac++] = "-V0";
    for (i = first_opt; i < argc && ac < 62; i++) av[ac++] = argv[i];
    av[ac++] = "/dev/null";           /* dummy existing target file */
    av[ac] = NULL;

    if (strcmp(mode, "lines") == 0) {
        ret = memagrep(ac, av, (int)len, buf, 0, stdout);
        fflush(stdout);
        return ret < 0 ? 1 : 0;
    }
    if (strncmp(mode, "membuf:", 7) == 0) {
        int cap = atoi(mode + 7), i2;
        char *out = (char *)calloc((size_t)cap + 16, 1);
        extern int agrep_outpointer;
        ret = memagrep(ac, av, (int)len, buf, cap, out);
        printf("ret=%d matched=%d outlen=%d\n", ret < 0 ? -1 : 0, num_of_matched, agrep_outpointer);
        for (i2 = 0; i2 < agrep_outpointer && i2 < cap; i2++) putchar(out[i2]);
        fflush(stdout);
        return 0;
    }
    {
        /* silence record output for tables/count: route it to /dev/null */
        FILE *sink = fopen("/dev/null", "w");
        ret = memagrep(ac, av, (int)len, buf, 0, sink);
        fclose(sink);
    }
    if (strcmp(mode, "count") == 0) {
        printf("%d\n", ret < 0 ? -1 : num_of_matched);
        return ret < 0 ? 1 : 0;
    }
    printf("{\"ret\": %d, \"SGREP\": %d, \"AND\": %d, \"D_length\": %d, \"NOUPPER\": %d,\n",
           ret, SGREP, AND, D_length, NOUPPER);
    printf(" \"Init0\": %u, \"Init1\": %u, \"NO_ERR_MASK\": %u, \"endposition\": %u,"
           " \"D_endpos\": %u, \"wildmask\": %u,\n",
           Init[0], Init1, NO_ERR_MASK, endposition, D_endpos, wildmask);
    printf(" \"Mask\": [");
    for (i = 0; i < 256; i++) printf("%u%s", Mask[i], i == 255 ? "" : ",");
    printf("]}\n");
    return 0;
}
