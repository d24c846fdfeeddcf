/*
 * agrep_hip.c -- C host side of the MI355X agrep scanner: the command-line surface of the
 * reference for the k-error hot path ( -# -c -l -i -w -x -d -B -y -n -h -s -e -k -f -I -S -D -V0 ), driving the
 * HIP C-ABI of include/agrep_hip.h.  It mirrors, for literal patterns and for the non-regex pattern language
 * (classes, '.', '#', <exact>, ^ $, ';' / ',' lists: agh_query_pattern restates preprocess() + maskgen()),
 *
 *   option parsing      agrep.c:2121-2739  (grouped flags, a digit run ends its group)
 *   engine dispatch     agrep.c:3357-3361 / 3428-3432  -> agh_query_literal + agh_scan_fd
 *   -c / -l / prefixes  agrep.c:3444-3558, asearch.c:130-161, agrep.c:3845-3875
 *   -B best match       agrep.c:3582-3728
 *   Grand Total, exit   agrep.c:3229-3231, main.c:78-96
 *
 * Everything outside the hot path (regular expressions, -v, -r ...)
 * is rejected with exit status 2: this binary is the hot-path driver, not a re-implementation of
 * agrep's control plane (the reference's own front end linked onto the same engines: ref_shim.c).  There is no CPU scan
 * engine here; without a HIP device the library calls fail and so does this program.
 */
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../include/agrep_hip.h"

#define MAX_FILES 4096

static const char *Progname = "agrep-hip";

static struct {
    int D;                 /* -#  */
    int COUNT;             /* -c  */
    int FILENAMEONLY;      /* -l  */
    int NOUPPER;           /* -i  */
    int LINENUM;           /* -n  */
    int NOFILENAME;        /* -h  */
    int SILENT;            /* -s  */
    int INVERSE;           /* -v  */
    int BESTMATCH;         /* -B  */
    int WORDBOUND;         /* -w  */
    int WHOLELINE;         /* -x  */
    int NOPROMPT;          /* -y  */
    int VERBOSE;           /* -V# (default 1: print the Grand Total line) */
    int APPROX;            /* a -# was given */
    int I, S, DD;          /* -I# -S# -D#: edit costs (asearch1.c), 0 = not given */
    unsigned char delim[AGH_MAX_DELIM + 1];
    int dlen;
    const char *pattern;
    const char *pattern_file;  /* -f  PAT_FILE */
    int gpus;              /* --gpus N: shard every file over N GPUs (0: the plain one-GPU path) */
    int approx_f;          /* --approx-f: -# applies to -f patterns too (BASELINE config 5; the reference ignores it) */
    int fancy;             /* the pattern uses the pattern language (classes, . # <> ^ $ ; ,): agh_query_pattern */
} opt;

static void die_usage(const char *msg)
{
    fprintf(stderr, "%s: %s\n", Progname, msg);
    fprintf(stderr,
            "usage: %s [--gpus N] [--approx-f] [-#cilwxnhsvyB] [-V0] [-d delim] [-e pattern | -f patternfile | pattern] [file ...]\n",
            Progname);
    exit(2);
}

/* checksg.c:45-102: any of these makes the pattern non-simple (regex / boolean / class ...).
 * -k (agrep.c, "pattern is a constant") switches the test off. */
static int pattern_is_literal(const char *p)
{
    return strpbrk(p, ";,.*-[]()<>|#{}~^$\\") == NULL;
}

/* agrep.c:2265-2316 + preproce.c:181-213: '^' and '$' in a delimiter mean newline,
 * a backslash quotes the next byte. */
static void set_delimiter(const char *arg)
{
    int n = 0;
    size_t i, len = strlen(arg);
    if (len == 0) die_usage("the -d option must have a delimiter argument");
    for (i = 0; i < len; i++) {
        unsigned char c = (unsigned char)arg[i];
        if (c == '\\' && i + 1 < len) c = (unsigned char)arg[++i];
        else if (c == '^' || c == '$') c = '\n';
        if (n >= AGH_MAX_DELIM) {
            fprintf(stderr, "%s: delimiter pattern too long (has > %d chars)\n", Progname,
                    AGH_MAX_DELIM);
            exit(2);
        }
        opt.delim[n++] = c;
    }
    opt.dlen = n;
}

static int parse_options(int argc, char **argv, char **files)
{
    int nfiles = 0, i, literal_only = 0, have_d = 0;
    opt.VERBOSE = 1;
    opt.delim[0] = '\n';
    opt.dlen = 1;
    for (i = 1; i < argc; i++) {
        char *a = argv[i];
        if (opt.pattern == NULL && strcmp(a, "--gpus") == 0) {   /* not a reference option: SURVEY 8e */
            if (i + 1 >= argc || (opt.gpus = atoi(argv[++i])) < 1 || opt.gpus > 64)
                die_usage("--gpus needs a device count in 1..64");
            continue;
        }
        if (opt.pattern == NULL && strcmp(a, "--approx-f") == 0) {   /* not a reference option either */
            opt.approx_f = 1;
            continue;
        }
        if (a[0] == '-' && a[1] != '\0') {
            char *p = a + 1;
            if (opt.pattern != NULL) {           /* after the pattern everything is a file */
                if (nfiles >= MAX_FILES) die_usage("too many files");
                files[nfiles++] = a;
                continue;
            }
            while (*p) {                        /* grouped single-letter options */
                char c = *p++;
                if (c >= '0' && c <= '9') {     /* agrep.c:2716-2728: a number ends the group */
                    opt.D = atoi(p - 1);
                    opt.APPROX = 1;
                    if (opt.D > AGH_MAX_ERRORS) {
                        fprintf(stderr, "%s: the maximum number of errors is %d\n", Progname,
                                AGH_MAX_ERRORS);
                        exit(2);
                    }
                    break;
                }
                switch (c) {
                case 'c': opt.COUNT = 1; break;
                case 'l': opt.FILENAMEONLY = 1; break;
                case 'i': opt.NOUPPER = 1; break;
                case 'n': opt.LINENUM = 1; break;
                case 'h': opt.NOFILENAME = 1; break;
                case 's': opt.SILENT = 1; break;
                case 'v': opt.INVERSE = 1; break;
                case 'y': opt.NOPROMPT = 1; break;
                case 'B': opt.BESTMATCH = 1; break;
                case 'w': opt.WORDBOUND = 1; break;
                case 'x': opt.WHOLELINE = 1; break;
                case 'k': literal_only = 1; break;
                case 'V':
                    opt.VERBOSE = (*p >= '0' && *p <= '9') ? atoi(p) : 1;
                    while (*p >= '0' && *p <= '9') p++;
                    break;
                case 'I': opt.I = atoi(p); p = (char *)""; break;    /* agrep.c:2680-2696 */
                case 'S': opt.S = atoi(p); p = (char *)""; break;
                case 'D': opt.DD = atoi(p); p = (char *)""; break;
                case 'd':
                    have_d = 1;
                    if (*p) set_delimiter(p);
                    else if (i + 1 < argc) set_delimiter(argv[++i]);
                    else die_usage("the -d option must have a delimiter argument");
                    p = (char *)"";
                    break;
                case 'e':
                    if (i + 1 >= argc) die_usage("the -e option must have a pattern argument");
                    opt.pattern = argv[++i];
                    p = (char *)"";
                    break;
                case 'f':                       /* agrep.c:2410-2460: patterns, one per line */
                    if (*p) opt.pattern_file = p;
                    else if (i + 1 < argc) opt.pattern_file = argv[++i];
                    else die_usage("the -f option must have a file name argument");
                    opt.pattern = "";           /* no pattern argument follows */
                    p = (char *)"";
                    break;
                default:
                    fprintf(stderr, "%s: option -%c is outside the GPU hot path of this build\n",
                            Progname, c);
                    exit(2);
                }
            }
        } else if (opt.pattern == NULL) {
            opt.pattern = a;
        } else {
            if (nfiles >= MAX_FILES) die_usage("too many files");
            files[nfiles++] = a;
        }
    }
    if (opt.pattern == NULL) die_usage("no pattern");
    /* agrep.c:2188-2196 (and :2661-2664): "illegal option combination (-x and -w)" */
    if (opt.WORDBOUND && opt.WHOLELINE) die_usage("illegal option combination (-x and -w)");
    if (opt.WHOLELINE && have_d) die_usage("-d and -x are not compatible");     /* compat.c:89-96 */
    /* -f with errors checks an occurrence by edit distance, the -w / -x tests need a verbatim one */
    if (opt.pattern_file && opt.approx_f && opt.D > 0 && (opt.WORDBOUND || opt.WHOLELINE))
        die_usage("-w / -x with a pattern file need exact matching (drop --approx-f or -#)");
    /* a delimiter of several bytes with letters under -i: where a record ends is decided on folded bytes,
     * which agh_shard_cuts_fd (raw bytes) does not see -- such files are not cut into shards */
    if (opt.gpus && opt.NOUPPER && opt.dlen > 1) {
        int j, letters = 0;
        for (j = 0; j < opt.dlen; j++)
            letters |= (opt.delim[j] >= 'a' && opt.delim[j] <= 'z') || (opt.delim[j] >= 'A' && opt.delim[j] <= 'Z');
        if (letters) die_usage("--gpus with -i and a multi-byte delimiter that holds letters is not supported");
    }
    if (opt.pattern_file) {
        /* compat.c:26-37: -B is ignored with -f; -# is not supported with -f (warning only) */
        if (opt.BESTMATCH) opt.BESTMATCH = 0;
        if (opt.APPROX && opt.D > 0 && !opt.approx_f)
            fprintf(stderr, "%s: approximate matching is not supported with -f option\n", Progname);
        if (!opt.approx_f) opt.D = 0;
        if (opt.COUNT && opt.FILENAMEONLY) opt.FILENAMEONLY = 0;
        return nfiles;
    }
    /* classes, '.', '#', <exact>, ^ $ anchors and ';' / ',' lists are compiled by the library
     * (agh_query_pattern = preprocess() + maskgen() restated); it refuses regular expressions */
    opt.fancy = !literal_only && !pattern_is_literal(opt.pattern);
    if (opt.fancy && opt.BESTMATCH) die_usage("-B needs a literal pattern in this build");
    if (opt.fancy) {                            /* host-only: a pattern the library cannot compile is reported
                                                 * before any device is touched (maskgen.c / preproce.c messages) */
        agh_pattern_tables tb;
        const unsigned qf = (opt.NOUPPER ? AGH_Q_NOCASE : 0u) | (opt.WORDBOUND ? AGH_Q_WORD : 0u) |
                            (opt.WHOLELINE ? AGH_Q_WHOLELINE : 0u);
        if (agh_compile_pattern((const unsigned char *)opt.pattern, (int)strlen(opt.pattern), qf, opt.delim,
                                opt.dlen, &tb)) {
            fprintf(stderr, "%s: %s\n", Progname, agh_last_error());
            exit(2);
        }
    }
    /* compat.c:26-29: -B is ignored together with -c, -l or -# */
    if (opt.BESTMATCH && (opt.COUNT || opt.FILENAMEONLY || opt.APPROX)) opt.BESTMATCH = 0;
    if (opt.COUNT && opt.FILENAMEONLY) opt.FILENAMEONLY = 0;     /* agrep.c:2896-2899 */
    return nfiles;
}

struct filehit {
    agh_result res;
    agh_match *matches;
    unsigned char *bytes;  /* matched records back to back (agh_fetch_records) */
};

static unsigned char *slurp(int fd, size_t *len)
{
    size_t cap = 1 << 20, used = 0;
    unsigned char *buf = (unsigned char *)malloc(cap);
    if (!buf) return NULL;
    for (;;) {
        ssize_t r;
        if (used == cap) {
            unsigned char *nb = (unsigned char *)realloc(buf, cap * 2);
            if (!nb) { free(buf); return NULL; }
            buf = nb;
            cap *= 2;
        }
        r = read(fd, buf + used, cap - used);
        if (r < 0) {
            if (errno == EINTR) continue;
            free(buf);
            return NULL;
        }
        if (r == 0) break;
        used += (size_t)r;
    }
    *len = used;
    return buf;
}

/* One file through the device engines; count-only unless records have to be printed.  The file is
 * streamed to HBM by the library (pinned ring, two bounded device segments); matched records are
 * printed from emit_records() after every segment, while the rest of the file is still being read --
 * as output() is called from inside the reference's block loop (asearch.c:66-324). */
struct emit_ctx {
    const char *name;
    int with_name;
};
static void print_records(const struct filehit *h, const char *name, int with_name);

static int emit_records(void *vctx, const agh_match *m, size_t n, const unsigned char *bytes, size_t n_bytes)
{
    const struct emit_ctx *c = (const struct emit_ctx *)vctx;
    struct filehit h;
    (void)n_bytes;
    memset(&h, 0, sizeof(h));
    h.res.n_stored = n;
    h.matches = (agh_match *)m;
    h.bytes = (unsigned char *)bytes;
    print_records(&h, c->name, c->with_name);
    return 0;
}

static int scan_one(agh_query *q, int fd, int want_records, struct filehit *out, const char *name, int with_name)
{
    struct emit_ctx c;
    memset(out, 0, sizeof(*out));
    if (!want_records) {
        unsigned flags = (opt.FILENAMEONLY ? AGH_FILENAMEONLY : AGH_COUNT) | (opt.INVERSE ? AGH_INVERT : 0u);
        return agh_scan_fd(q, fd, flags, &out->res, NULL, 0);
    }
    c.name = name;
    c.with_name = with_name;
    return agh_scan_fd_emit(q, fd, opt.INVERSE ? AGH_INVERT : 0u, &out->res, emit_records, &c);
}

/* agrep.c:3805-3956 output(): [file: ][N: ]record\n for newline-delimited records.  With a
 * user delimiter (-d, OUTTAIL off, agrep.c:2265-2316) output() prints buffer[lasti .. i-D_length-1]:
 * the delimiter IN FRONT of the record and the record, nothing behind it -- except for the first
 * record of a file, which has no delimiter in front (asearch.c:162-170, lasti = Max_record).  The
 * delimiter printed here is the one given with -d (under -i the text's own bytes may differ in
 * case; the reference front end linked onto these engines prints those: INTEGRATION.md). */
static int first_output = 1, eat_first = 0;     /* output()'s FIRSTOUTPUT / EATFIRST (agrep.c:3820-3826, 3731-3741) */

static void print_records(const struct filehit *h, const char *name, int with_name)
{
    uint64_t i;
    size_t o = 0;
    const int user_delim = !(opt.dlen == 1 && opt.delim[0] == '\n');
    for (i = 0; i < h->res.n_stored; i++) {
        const agh_match *m = &h->matches[i];
        const size_t len = (size_t)(m->end - m->start);
        const unsigned char *body = h->bytes + o;
        /* what output() walks over: [delimiter] record, or record + newline */
        const size_t dl = (user_delim && m->index > 0) ? (size_t)opt.dlen : 0;
        size_t p = 0;                           /* position in (delimiter, record) */
#define SEQ(x) ((x) < dl ? opt.delim[(x)] : body[(x) - dl])
        if (first_output) {                     /* the very first output eats one leading newline ... */
            if (dl + len > 0 && SEQ(0) == '\n') { p = 1; eat_first = 1; }
            first_output = 0;
        }
        while (p < dl + len && SEQ(p) == '\n') { fputc('\n', stdout); p++; }   /* agrep.c:3832-3843 */
        if (with_name) printf("%s: ", name);
        if (opt.LINENUM) printf("%llu: ", (unsigned long long)(m->index + 1));
        for (; p < dl; p++) fputc(opt.delim[p], stdout);
        if (p - dl < len) fwrite(body + (p - dl), 1, len - (p - dl), stdout);
#undef SEQ
        if (!user_delim) fputc('\n', stdout);
        o += len;
    }
}

static long run_pass(agh_query *q, char **files, int nfiles, int print, int count_only,
                     long *files_matched)
{
    long total = 0;
    int i;
    for (i = 0; i < (nfiles ? nfiles : 1); i++) {
        const char *name = nfiles ? files[i] : "stdin";
        int fd = nfiles ? open(files[i], O_RDONLY) : 0;
        struct filehit h;
        int want_records = print && !count_only && !opt.COUNT && !opt.FILENAMEONLY && !opt.SILENT;
        if (fd < 0) {                           /* agrep.c:2952-2958 */
            fprintf(stderr, "%s: '%s' no such file or directory\n", Progname, name);
            continue;
        }
        if (scan_one(q, fd, want_records, &h, name, nfiles > 1 && !opt.NOFILENAME)) {
            fprintf(stderr, "%s: %s: %s\n", Progname, name, agh_last_error());
            exit(2);
        }
        if (nfiles) close(fd);
        if (print && !opt.SILENT) {
            if (opt.COUNT) {                    /* agrep.c:3501-3556 */
                if (nfiles > 1 && !opt.NOFILENAME)
                    printf("%s: %llu\n", name, (unsigned long long)h.res.n_matched);
                else
                    printf("%llu\n", (unsigned long long)h.res.n_matched);
            } else if (opt.FILENAMEONLY) {      /* asearch.c:130-161 */
                if (h.res.n_matched) printf("%s\n", name);
            }                                   /* (records: printed by emit_records() during the scan) */
        }
        if (h.res.n_matched) (*files_matched)++;
        /* -l counts files, everything else counts records (sgrep.c:1188, Appendix A) */
        total += opt.FILENAMEONLY ? (h.res.n_matched ? 1 : 0) : (long)h.res.n_matched;
        free(h.matches);
        free(h.bytes);
    }
    return total;
}

/* ---------------------------------------------------------------------------------------
 * --gpus N: one process, N devices, one host thread per device.
 *   -c / record output: every file is cut into N record-aligned shards (agh_shard_cuts_fd), GPU r
 *       scans shard r (-c: agh_scan_fd_range; records: agh_scan_fd_range_emit, streamed); the per-file
 *       count exec() prints (agrep.c:3444-3558) is the sum of the shard counts; matched records are
 *       printed shard after shard, i.e. in file order, while the later shards are still being scanned,
 *       record numbers offset by the shards in front (struct shard_order).
 *   -l: the files are dealt out to the GPUs (file f -> GPU f mod N), each scanned whole with the
 *       early exit; the file list is the OR of the per-GPU hit vectors, printed in argument order.
 * The N threads share this process's memory, so the sums are plain host additions: there is no
 * exchange step and no communicator (ncclCommInitAll costs ~2-3 s per process, more than most jobs).
 * RCCL belongs to the one-process-per-GPU form (agh_scan_device_reduce, agh_reduce_counts: bench.py);
 * AGH_CLI_RCCL=1 runs the same sums through agh_reduce_counts_all / agh_reduce_file_hits_all as
 * well and compares them (tests).
 * The queries are built per device by the same code as the one-GPU path.
 * --------------------------------------------------------------------------------------- */
typedef agh_query *(*query_builder)(void);

struct gpu_task {
    int rank, ngpus, device;
    query_builder build;
    char **files;
    int nfiles;
    int want_records;
    struct shard_order *order;  /* record output: whose records go to stdout now */
    int with_name;
    /* results */
    struct filehit *hits;       /* [nfiles]: this rank's shard of every file (-c / records) */
    unsigned char *file_hit;    /* [nfiles]: -l */
    int failed;
    char err[512];
};

/* Record output of the shards, in file order, while the shards are still being scanned (asearch.c:162-170 prints
 * from inside its block loop): every (file, shard) pair has a number in the order its records belong on stdout --
 * file after file, shard after shard -- and `turn` says whose records may go out.  The shard whose turn it is prints
 * straight from its emit() calls; a shard further back keeps what its scan hands over in host memory until the
 * shards in front are done (and knows only then how many records lie in front of it: `rec_off`, for -n and for
 * "the first record of a file has no delimiter in front"). */
struct shard_order {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    long turn;                  /* file * ngpus + rank of the shard that prints */
    uint64_t rec_off;           /* records of this file's shards in front of it */
};

struct shard_emit {
    struct shard_order *ord;
    long seq;
    const char *name;
    int with_name;
    /* held back while it is another shard's turn */
    agh_match *m;
    size_t n, cap_m;
    unsigned char *bytes;
    size_t nb, cap_b;
    int oom;
};

static void shard_print(const struct shard_emit *c, const agh_match *m, size_t n, const unsigned char *bytes,
                        uint64_t rec_off)
{
    struct filehit h;
    agh_match *mm;
    size_t i;
    if (!n) return;
    mm = (agh_match *)malloc(n * sizeof(*mm));
    if (!mm) { fprintf(stderr, "%s: out of memory\n", Progname); exit(2); }
    for (i = 0; i < n; i++) { mm[i] = m[i]; mm[i].index += rec_off; }
    memset(&h, 0, sizeof(h));
    h.res.n_stored = n;
    h.matches = mm;
    h.bytes = (unsigned char *)bytes;
    print_records(&h, c->name, c->with_name);
    free(mm);
}

static int shard_my_turn(struct shard_emit *c, int wait, uint64_t *rec_off)
{
    int mine;
    pthread_mutex_lock(&c->ord->mu);
    while (wait && c->ord->turn != c->seq) pthread_cond_wait(&c->ord->cv, &c->ord->mu);
    mine = c->ord->turn == c->seq;
    *rec_off = c->ord->rec_off;
    pthread_mutex_unlock(&c->ord->mu);
    return mine;
}

static void shard_flush_held(struct shard_emit *c, uint64_t rec_off)
{
    shard_print(c, c->m, c->n, c->bytes, rec_off);
    c->n = c->nb = 0;
}

static int shard_emit_records(void *vctx, const agh_match *m, size_t n, const unsigned char *bytes, size_t n_bytes)
{
    struct shard_emit *c = (struct shard_emit *)vctx;
    uint64_t rec_off;
    if (shard_my_turn(c, 0, &rec_off)) {         /* (nobody else prints: turn moves on only when this shard is done) */
        shard_flush_held(c, rec_off);
        shard_print(c, m, n, bytes, rec_off);
        return 0;
    }
    if (c->n + n > c->cap_m) {
        size_t cap = (c->n + n) * 2;
        agh_match *nm = (agh_match *)realloc(c->m, cap * sizeof(*nm));
        if (!nm) { c->oom = 1; return 1; }
        c->m = nm;
        c->cap_m = cap;
    }
    if (c->nb + n_bytes > c->cap_b) {
        size_t cap = (c->nb + n_bytes) * 2;
        unsigned char *nbuf = (unsigned char *)realloc(c->bytes, cap);
        if (!nbuf) { c->oom = 1; return 1; }
        c->bytes = nbuf;
        c->cap_b = cap;
    }
    memcpy(c->m + c->n, m, n * sizeof(*m));
    if (n_bytes) memcpy(c->bytes + c->nb, bytes, n_bytes);
    c->n += n;
    c->nb += n_bytes;
    return 0;
}

/* this shard is done (or was never scanned: its file could not be opened): print what is still held back when the
 * shards in front have finished, then hand the turn on */
static void shard_done(struct shard_emit *c, uint64_t n_records, int last_of_file)
{
    uint64_t rec_off;
    (void)shard_my_turn(c, 1, &rec_off);
    shard_flush_held(c, rec_off);
    pthread_mutex_lock(&c->ord->mu);
    c->ord->rec_off = last_of_file ? 0 : c->ord->rec_off + n_records;
    c->ord->turn = c->seq + 1;
    pthread_cond_broadcast(&c->ord->cv);
    pthread_mutex_unlock(&c->ord->mu);
    free(c->m);
    free(c->bytes);
    c->m = NULL;
    c->bytes = NULL;
    c->cap_m = c->cap_b = 0;
}

static int scan_range(agh_query *q, int fd, uint64_t b, uint64_t e, struct shard_emit *emit, struct filehit *out)
{
    unsigned inv = opt.INVERSE ? AGH_INVERT : 0u;
    memset(out, 0, sizeof(*out));
    if (!emit)
        return agh_scan_fd_range(q, fd, b, e, AGH_COUNT | inv, &out->res, NULL, 0);
    /* the shard streams through two bounded device segments like a file of its own (agh_scan_fd_range_emit): HBM
     * and host memory stay bounded whatever the size of the file (round 5 staged the whole shard and fetched its
     * records after the scan) */
    if (agh_scan_fd_range_emit(q, fd, b, e, inv, &out->res, shard_emit_records, emit)) return -1;
    if (emit->oom) { fprintf(stderr, "%s: out of memory\n", Progname); exit(2); }
    return 0;
}

static void *gpu_worker(void *arg)
{
    struct gpu_task *t = (struct gpu_task *)arg;
    agh_query *q;
    int f;
    q = NULL;
    if (agh_set_device(t->device) || !(q = t->build())) {
        t->failed = 1;
        snprintf(t->err, sizeof(t->err), "GPU %d: %s", t->rank, agh_last_error());
    }
    for (f = 0; f < t->nfiles && !t->failed; f++) {
        int fd;
        struct shard_emit em;
        if (opt.FILENAMEONLY && f % t->ngpus != t->rank) continue;     /* dealt to another GPU */
        memset(&em, 0, sizeof(em));
        em.ord = t->order;
        em.seq = (long)f * t->ngpus + t->rank;
        em.name = t->files[f];
        em.with_name = t->with_name;
        fd = open(t->files[f], O_RDONLY);
        if (fd < 0) {                            /* reported once by the main thread */
            if (t->want_records) shard_done(&em, 0, t->rank == t->ngpus - 1);
            continue;
        }
        if (opt.FILENAMEONLY) {
            agh_result r;
            /* the same flags as the one-GPU path (scan_one): -l -v lists the files with a record
             * that does NOT match */
            if (agh_scan_fd(q, fd, AGH_FILENAMEONLY | (opt.INVERSE ? AGH_INVERT : 0u), &r, NULL, 0)) t->failed = 1;
            else t->file_hit[f] = r.n_matched ? 1 : 0;
        } else {
            uint64_t cuts[65];
            if (agh_shard_cuts_fd(fd, opt.delim, opt.dlen, t->ngpus, cuts) ||
                scan_range(q, fd, cuts[t->rank], cuts[t->rank + 1], t->want_records ? &em : NULL, &t->hits[f]))
                t->failed = 1;
            /* (a failed shard still hands the turn on: the other threads must come to their join) */
            if (t->want_records) shard_done(&em, t->failed ? 0 : t->hits[f].res.n_records, t->rank == t->ngpus - 1);
        }
        if (t->failed) snprintf(t->err, sizeof(t->err), "%s: %s", t->files[f], agh_last_error());
        close(fd);
    }
    /* a thread that gave up hands on the turns of the shards it will not scan */
    if (t->failed && t->want_records)
        for (; f < t->nfiles; f++) {             /* (f: the first file this thread has not handed on) */
            struct shard_emit em;
            memset(&em, 0, sizeof(em));
            em.ord = t->order;
            em.seq = (long)f * t->ngpus + t->rank;
            shard_done(&em, 0, t->rank == t->ngpus - 1);
        }
    if (q) agh_query_free(q);
    return NULL;
}

struct comm_init {
    agh_comm **comms;
    int n, rc;
    char err[256];
};

static void *comm_init_thread(void *arg)
{
    struct comm_init *c = (struct comm_init *)arg;
    c->rc = agh_comm_init_all(c->comms, c->n, NULL);
    if (c->rc) snprintf(c->err, sizeof(c->err), "%s", agh_last_error());
    return NULL;
}

static long run_multi_gpu(query_builder build, char **files, int nfiles, long *files_matched)
{
    const int G = opt.gpus;
    struct comm_init ci;
    pthread_t ci_th;
    const int want_records = !opt.COUNT && !opt.FILENAMEONLY && !opt.SILENT;
    const char *rccl_env = getenv("AGH_CLI_RCCL");
    const int use_rccl = rccl_env && rccl_env[0] == '1';
    struct gpu_task *tasks = (struct gpu_task *)calloc((size_t)G, sizeof(*tasks));
    pthread_t *th = (pthread_t *)calloc((size_t)G, sizeof(*th));
    agh_comm *comms[64];
    long total = 0;
    int r, f, ndev, ci_joined = 0;
    const char *share;
    struct shard_order order;
    if (nfiles == 0) die_usage("--gpus needs file arguments (stdin cannot be cut into shards)");
    pthread_mutex_init(&order.mu, NULL);
    pthread_cond_init(&order.cv, NULL);
    order.turn = 0;
    order.rec_off = 0;
    /* AGH_CLI_SHARE_DEVICES=1 (test hook for one-GPU boxes): shard r runs on device r mod the visible devices, so
     * the N-thread control flow -- cuts, ordered printing, sums -- runs with N > 1 on one GPU */
    ndev = agh_device_count();
    share = getenv("AGH_CLI_SHARE_DEVICES");
    if (ndev < G && !(share && share[0] == '1' && ndev >= 1)) {
        fprintf(stderr, "%s: --gpus %d but only %d HIP device(s) are visible\n", Progname, G, ndev);
        exit(2);
    }
    /* AGH_CLI_RCCL=1: the cross-check through RCCL; ncclCommInitAll runs on a thread of its own while
     * the workers read and scan their shards */
    ci.comms = comms;
    ci.n = G;
    ci.rc = 0;
    ci.err[0] = 0;
    if (use_rccl) pthread_create(&ci_th, NULL, comm_init_thread, &ci);
    /* (RCCL prints its banner on stdout when a communicator is made, and the library points fd 1 at /dev/null for
     * that moment: workers that print records from their emit() calls must not start before it is back) */
    if (use_rccl && want_records) { pthread_join(ci_th, NULL); ci_joined = 1; }
    for (r = 0; r < G; r++) {
        tasks[r].rank = r;
        tasks[r].device = r % ndev;
        tasks[r].ngpus = G;
        tasks[r].build = build;
        tasks[r].files = files;
        tasks[r].nfiles = nfiles;
        tasks[r].want_records = want_records;
        tasks[r].order = &order;
        tasks[r].with_name = nfiles > 1 && !opt.NOFILENAME;
        tasks[r].hits = (struct filehit *)calloc((size_t)nfiles, sizeof(struct filehit));
        tasks[r].file_hit = (unsigned char *)calloc((size_t)nfiles, 1);
        pthread_create(&th[r], NULL, gpu_worker, &tasks[r]);
    }
    for (r = 0; r < G; r++) pthread_join(th[r], NULL);
    if (use_rccl) {
        if (!ci_joined) pthread_join(ci_th, NULL);
        if (ci.rc) {
            fprintf(stderr, "%s: RCCL: %s\n", Progname, ci.err);
            exit(2);
        }
    }
    for (r = 0; r < G; r++)
        if (tasks[r].failed) { fprintf(stderr, "%s: %s\n", Progname, tasks[r].err); exit(2); }

    if (opt.FILENAMEONLY) {                      /* the -l hit vector: OR over the GPUs, into rank 0's */
        unsigned char *host_or = (unsigned char *)calloc((size_t)nfiles + 1, 1);
        for (r = 0; r < G; r++)
            for (f = 0; f < nfiles; f++) host_or[f] |= tasks[r].file_hit[f];
        if (use_rccl) {
            unsigned char *vec[64];
            for (r = 0; r < G; r++) vec[r] = tasks[r].file_hit;
            if (agh_reduce_file_hits_all(comms, G, vec, (size_t)nfiles)) {
                fprintf(stderr, "%s: RCCL: %s\n", Progname, agh_last_error());
                exit(2);
            }
            if (memcmp(tasks[0].file_hit, host_or, (size_t)nfiles)) {
                fprintf(stderr, "%s: internal error: RCCL max differs from the host OR\n", Progname);
                exit(2);
            }
        }
        memcpy(tasks[0].file_hit, host_or, (size_t)nfiles);
        free(host_or);
    }
    for (f = 0; f < nfiles; f++) {
        uint64_t counts[64][2], host_sum = 0;
        int fd = open(files[f], O_RDONLY);
        if (fd < 0) {                            /* agrep.c:2952-2958 */
            fprintf(stderr, "%s: '%s' no such file or directory\n", Progname, files[f]);
            continue;
        }
        close(fd);
        if (opt.FILENAMEONLY) {                  /* asearch.c:130-161 */
            if (tasks[0].file_hit[f]) {
                if (!opt.SILENT) printf("%s\n", files[f]);
                (*files_matched)++;
                total++;
            }
            continue;
        }
        for (r = 0; r < G; r++) {
            counts[r][0] = tasks[r].hits[f].res.n_matched;
            counts[r][1] = tasks[r].hits[f].res.n_records;
            host_sum += counts[r][0];
        }
        if (use_rccl) {
            if (agh_reduce_counts_all(comms, G, counts)) {   /* ncclAllReduce(sum) over the shard counts */
                fprintf(stderr, "%s: RCCL: %s\n", Progname, agh_last_error());
                exit(2);
            }
            if (counts[0][0] != host_sum) {
                fprintf(stderr, "%s: internal error: RCCL sum %llu != host sum %llu\n", Progname,
                        (unsigned long long)counts[0][0], (unsigned long long)host_sum);
                exit(2);
            }
        }
        counts[0][0] = host_sum;
        if (!opt.SILENT) {
            if (opt.COUNT) {                     /* agrep.c:3501-3556 */
                if (nfiles > 1 && !opt.NOFILENAME)
                    printf("%s: %llu\n", files[f], (unsigned long long)counts[0][0]);
                else
                    printf("%llu\n", (unsigned long long)counts[0][0]);
            }                                    /* (records: printed by the workers, shard after shard) */
        }
        if (counts[0][0]) (*files_matched)++;
        total += (long)counts[0][0];
    }
    for (r = 0; r < G; r++) {
        for (f = 0; f < nfiles; f++) { free(tasks[r].hits[f].matches); free(tasks[r].hits[f].bytes); }
        free(tasks[r].hits);
        free(tasks[r].file_hit);
        if (use_rccl) agh_comm_free(comms[r]);
    }
    free(tasks);
    free(th);
    return total;
}

/* query of the command line for the calling thread's device */
static const unsigned char **g_multi_pats;
static int *g_multi_lens;
static int g_multi_n;

static agh_query *build_cli_query(void)
{
    agh_query *q;
    const unsigned qf = (opt.NOUPPER ? AGH_Q_NOCASE : 0u) | (opt.WORDBOUND ? AGH_Q_WORD : 0u) |
                        (opt.WHOLELINE ? AGH_Q_WHOLELINE : 0u);
    if (opt.pattern_file && opt.approx_f && opt.D > 0)      /* union of the k-error predicate over the patterns */
        return agh_query_multi_approx(g_multi_pats, g_multi_lens, g_multi_n, opt.D, opt.NOUPPER, opt.delim, opt.dlen);
    if (opt.pattern_file)
        return agh_query_multi_ex(g_multi_pats, g_multi_lens, g_multi_n, qf, opt.delim, opt.dlen);
    if (opt.fancy)
        q = agh_query_pattern((const unsigned char *)opt.pattern, (int)strlen(opt.pattern), opt.D, qf, opt.delim,
                              opt.dlen);
    else
        q = agh_query_literal_ex((const unsigned char *)opt.pattern, (int)strlen(opt.pattern), opt.D, qf,
                                 opt.delim, opt.dlen);
    if (q && (opt.I || opt.S || opt.DD) &&
        agh_query_set_costs(q, opt.I ? opt.I : 1, opt.S ? opt.S : 1, opt.DD ? opt.DD : 1)) {
        agh_query_free(q);
        return NULL;
    }
    return q;
}

/* The process ends here: nothing of the query needs to be given back one by one (2 GiB of device segments, the
 * pinned ring, streams, the runtime's own teardown: 30-40 ms of a 0.3 s run on 4 GiB, profiles/r05_startup.log) --
 * the output is flushed and the kernel reclaims the rest.  AGH_CLI_TEARDOWN=1 frees everything in order (leak
 * checkers). */
static int fast_exit(void)
{
    const char *e = getenv("AGH_CLI_TEARDOWN");
    return !(e && e[0] == '1');
}

static void finish(agh_query *q, long total)
{
    if (!fast_exit()) {
        if (q) agh_query_free(q);
        return;
    }
    fflush(stdout);
    fflush(stderr);
    _exit((int)(total & 0xff));                 /* main.c:79,96: exit status = matches (mod 256) */
}

int main(int argc, char **argv)
{
    static char *files[MAX_FILES];
    int nfiles, m;
    long total = 0, files_matched = 0;
    agh_query *q;

    nfiles = parse_options(argc, argv, files);
    if (opt.pattern_file) {
        /* prepf() (newmgrep.c:192-375): one literal pattern per line, empty lines skipped */
        size_t len = 0, i, start = 0, np = 0, cap = 64;
        int fd = open(opt.pattern_file, O_RDONLY);
        unsigned char *buf;
        const unsigned char **pp;
        int *ll;
        if (fd < 0) {
            fprintf(stderr, "%s: can't open pattern file for reading: %s\n", Progname, opt.pattern_file);
            exit(2);
        }
        buf = slurp(fd, &len);
        close(fd);
        if (!buf) exit(2);
        pp = (const unsigned char **)malloc(cap * sizeof(*pp));
        ll = (int *)malloc(cap * sizeof(*ll));
        for (i = 0; i <= len; i++)
            if (i == len || buf[i] == '\n') {
                if (i > start) {
                    if (np == cap) {
                        cap *= 2;
                        pp = (const unsigned char **)realloc(pp, cap * sizeof(*pp));
                        ll = (int *)realloc(ll, cap * sizeof(*ll));
                    }
                    pp[np] = buf + start;
                    ll[np] = (int)(i - start);
                    np++;
                }
                start = i + 1;
            }
        if (np == 0) die_usage("the pattern file holds no pattern");
        if (agh_device_count() <= 0) {
            fprintf(stderr, "%s: no usable HIP device (this build has no CPU scan engine)\n", Progname);
            exit(2);
        }
        g_multi_pats = pp;
        g_multi_lens = ll;
        g_multi_n = (int)np;
        if (opt.gpus) {
            total = run_multi_gpu(build_cli_query, files, nfiles, &files_matched);
        } else {
            q = build_cli_query();
            if (!q) { fprintf(stderr, "%s: %s\n", Progname, agh_last_error()); exit(2); }
            total = run_pass(q, files, nfiles, 1, 0, &files_matched);
            if (eat_first) fputc('\n', stdout);
            if (opt.VERBOSE > 0 && !opt.SILENT) printf("Grand Total: %ld match(es) found.\n", total);
            finish(q, total);
            return (int)total;
        }
        if (eat_first) fputc('\n', stdout);
        if (opt.VERBOSE > 0 && !opt.SILENT) printf("Grand Total: %ld match(es) found.\n", total);
        return (int)total;
    }
    m = (int)strlen(opt.pattern);
    if (opt.D >= m) {                           /* checksg.c:34-41 */
        fprintf(stderr, "%s: size of pattern must be greater than number of errors\n", Progname);
        exit(2);
    }
    if (agh_device_count() <= 0) {
        fprintf(stderr, "%s: no usable HIP device (this build has no CPU scan engine)\n", Progname);
        exit(2);
    }

    if (opt.gpus && opt.BESTMATCH) die_usage("--gpus does not combine with -B");
    if (opt.gpus) {
        total = run_multi_gpu(build_cli_query, files, nfiles, &files_matched);
    } else if (!opt.BESTMATCH) {
        q = build_cli_query();
        if (!q) { fprintf(stderr, "%s: %s\n", Progname, agh_last_error()); exit(2); }
        total = run_pass(q, files, nfiles, 1, 0, &files_matched);
        if (eat_first) fputc('\n', stdout);    /* ... and gives it back at the end (agrep.c:3731-3741) */
        if (opt.VERBOSE > 0 && !opt.SILENT)     /* agrep.c:3229-3231 */
            printf("Grand Total: %ld match(es) found.\n", total);
        finish(q, total);
        return (int)total;
    } else {
        /* agrep.c:3582-3728: exact first; then the smallest D in 1..min(M-1, 8) with a match,
         * where M = m + 2 counts the delimiter slots (maskgen.c), so D may reach m: at that
         * point every non-empty record is within D errors and the whole file is printed. */
        int D = 0, maxD = m + 1 < AGH_MAX_ERRORS ? m + 1 : AGH_MAX_ERRORS;
        long found = 0;
        if (nfiles == 0) die_usage("-B needs file arguments (stdin cannot be re-read)");
        q = NULL;
        for (D = 0; D <= maxD && D < m; D++) {
            long fm = 0;
            q = agh_query_literal_ex((const unsigned char *)opt.pattern, m, D,
                                     (opt.NOUPPER ? AGH_Q_NOCASE : 0u) | (opt.WORDBOUND ? AGH_Q_WORD : 0u) |
                                         (opt.WHOLELINE ? AGH_Q_WHOLELINE : 0u),
                                     opt.delim, opt.dlen);
            if (!q) { fprintf(stderr, "%s: %s\n", Progname, agh_last_error()); exit(2); }
            found = run_pass(q, files, nfiles, 0, 1, &fm);
            if (found > 0) break;
            agh_query_free(q);
            q = NULL;
        }
        if (found == 0 && D <= maxD) {
            /* D == m: count the non-empty records with a plain delimiter census */
            int i;
            for (i = 0; i < nfiles; i++) {
                size_t len = 0, j, start = 0;
                int fd = open(files[i], O_RDONLY);
                unsigned char *t;
                if (fd < 0) continue;
                t = slurp(fd, &len);
                close(fd);
                if (!t) continue;
                for (j = 0; j <= len; j++)
                    if (j == len || t[j] == opt.delim[0]) {
                        if (j > start) found++;
                        start = j + 1;
                    }
                free(t);
            }
        }
        if (found > 0) {
            int go = 1;
            if (D > 0) {
                char c[8] = "y";
                if (found == 1) fprintf(stderr, "%s: 1 word matches within ", Progname);
                else fprintf(stderr, "%s: %ld words match within ", Progname, found);
                if (D == 1) fprintf(stderr, "1 error");
                else fprintf(stderr, "%d errors", D);
                fflush(stderr);
                if (opt.NOPROMPT) fprintf(stderr, "\n");
                else {
                    fprintf(stderr, found == 1 ? "; search for it? (y/n)" : "; search for them? (y/n)");
                    if (fgets(c, 4, stdin) == NULL || c[0] != 'y') go = 0;
                }
            }
            if (go && q) total = run_pass(q, files, nfiles, 1, 0, &files_matched);
            else if (go) {                      /* D == m: every non-empty record */
                int i;
                for (i = 0; i < nfiles; i++) {
                    size_t len = 0, j, start = 0;
                    int fd = open(files[i], O_RDONLY);
                    unsigned char *t;
                    if (fd < 0) continue;
                    t = slurp(fd, &len);
                    close(fd);
                    if (!t) continue;
                    for (j = 0; j <= len; j++)
                        if (j == len || t[j] == opt.delim[0]) {
                            if (j > start) {
                                if (!opt.SILENT) {
                                    if (nfiles > 1 && !opt.NOFILENAME) printf("%s: ", files[i]);
                                    fwrite(t + start, 1, j - start, stdout);
                                    fputc('\n', stdout);
                                }
                                total++;
                            }
                            start = j + 1;
                        }
                    free(t);
                }
            }
            errno = D;
        }
        if (q) agh_query_free(q);
    }

    if (eat_first) fputc('\n', stdout);         /* ... and gives it back at the end (agrep.c:3731-3741) */
    if (opt.VERBOSE > 0 && !opt.SILENT)          /* agrep.c:3229-3231 */
        printf("Grand Total: %ld match(es) found.\n", total);
    return (int)total;                          /* main.c:79,96: exit status = matches (mod 256) */
}
