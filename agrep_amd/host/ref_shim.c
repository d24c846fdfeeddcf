/*
 * ref_shim.c -- link-level drop-in: the eight symbols the reference FRONT END needs from its
 * engine objects (nm on the reference objects: bitap.o sgrep.o newmgrep.o asearch.o asearch1.o
 * define, and agrep.o & co. use, exactly these)
 *
 *     bitap  sgrep  mgrep  prepf  pat_spool  fill_buf  alloc_buf  free_buf
 *
 * implemented on the MI355X engines of libagrep_hip.so.  Linked with the UNMODIFIED reference
 * objects agrep.o maskgen.o preproce.o checksg.o parse.o ... in place of bitap.o sgrep.o
 * newmgrep.o asearch.o asearch1.o (oracle/Makefile, target ref_gpu) this gives the reference's
 * own command line -- option parsing, preprocess(), maskgen(), checksg(), exec(), output(),
 * Grand Total, exit status -- driving the GPU for the scan itself:
 *
 *   bitap(old_D_pat, Pattern, fd, M, D)   bitap.c:78-306 (+ asearch.c, asearch1.c through it):
 *       the query is maskgen()'s globals handed over unchanged (agh_query_from_maskgen);
 *   sgrep(pat, m, fd, D, samepattern)     sgrep.c:262-682: literal pattern, agh_query_literal;
 *   prepf(fd, buf, len) / mgrep(fd)       newmgrep.c:192-375 / 463-691: agh_query_multi.
 *
 * Matched records are printed by the reference's own output() (agrep.c:3805-3956), one call per
 * record with the same (buffer, i1, i2, j) convention asearch.c:162-170 uses, so prefixes, -n,
 * -d head/tail placement and the memory-mode output buffer (agrep_outbuffer, OUTPUT_OVERFLOW)
 * are the reference's code, not a re-implementation.
 *
 * Deliberate differences (stated in DESIGN.md): no quirk Q1..Q10 is reproduced -- in particular
 * the sgrep path is case-sensitive without -i at k = 0 (Q6) and counts a record once (Q4); regular
 * expressions stay on the reference's CPU engines re()/re1() (agrep.c), which is why fill_buf /
 * alloc_buf / free_buf are provided for them.
 */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/agrep_hip.h"

/* ---- the reference's globals (definitions: agrep.c:113-215) ---------------------------- */
extern unsigned Mask[], Init[], Init1, NO_ERR_MASK, endposition, D_endpos;
extern int AND, REGEX, JUMP, I, S, DD, INVERSE, NOUPPER, COUNT, FILENAMEONLY, SILENT, DELIMITER,
    OUTTAIL, LINENUM, WORDBOUND, WHOLELINE, LIMITOUTPUT, LIMITPERFILE, NEW_FILE, POST_FILTER,
    EXITONERROR, CurrentByteOffset, TRUNCATE, D_length, CONSTANT, FIRSTOUTPUT;
extern int num_of_matched, prev_num_of_matched;
extern unsigned char D_pattern[], CurrentFileName[], Progname[];
extern FILE *agrep_finalfp;
extern unsigned char *agrep_inbuffer, *agrep_outbuffer;
extern int agrep_inlen, agrep_outlen, agrep_outpointer;
extern int glimpse_clientdied;
int output();                   /* agrep.c:3805 */
int re();                       /* agrep.c:1267 */
int re1();                      /* agrep.c:468 */

char *pat_spool = NULL;         /* newmgrep.c: freed by agrep.c:333 */

#define AGREP_ERROR 123         /* agrep.h:173 */
#define SHORTREG 15             /* agrep.h: M <= SHORTREG -> re(), else re1() (bitap.c:108-113) */

/* ---- block I/O helpers the front end (re / re1 / file_out) still calls: bitap.c:450-505 --- */
int fill_buf(int fd, unsigned char *buf, int record_size)
{
    int total = 0;
    if (fd < 0) return 0;       /* memory mode never comes here (AGREP_POINTER) */
    while (total < record_size) {
        ssize_t r;
        if (glimpse_clientdied) return 0;
        r = read(fd, buf + total, (size_t)(record_size - total));
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) break;
        total += (int)r;
    }
    return glimpse_clientdied ? 0 : total;
}

void alloc_buf(int fd, unsigned char **buf, int size)
{
    if (fd != -1) *buf = (unsigned char *)malloc((size_t)size);
}

void free_buf(int fd, unsigned char *buf)
{
    if (fd != -1) free(buf);
}

/* ---- errors: agrep.h:173 convention --------------------------------------------------- */
static int shim_fail(const char *what)
{
    fprintf(stderr, "%s: %s\n", Progname, what);
    if (!EXITONERROR) {
        errno = AGREP_ERROR;
        return -1;
    }
    exit(2);
}

/* asearch.c:130-161: -l prints the name once and the engine returns */
static int print_filename(void)
{
    num_of_matched++;
    if (agrep_finalfp != NULL) {
        fprintf(agrep_finalfp, "%s\n", CurrentFileName);
    } else {
        int i;
        for (i = 0; i + agrep_outpointer < agrep_outlen && CurrentFileName[i] != '\0'; i++)
            agrep_outbuffer[agrep_outpointer + i] = CurrentFileName[i];
        if (CurrentFileName[i] != '\0' || i + agrep_outpointer + 1 >= agrep_outlen) {
            fprintf(stderr, "Output buffer overflow after %d bytes @ %s:%d !!\n", agrep_outpointer,
                    __FILE__, __LINE__);
            return -1;
        }
        agrep_outbuffer[agrep_outpointer + i++] = '\n';
        agrep_outpointer += i;
    }
    NEW_FILE = 0;
    return 0;
}

/* What asearch.c hands to output() (asearch.c:162: output(buffer, lasti, i - D_length - 1, j)): the
 * buffer holds the delimiter in front of the record (lasti points at it), the record, and the
 * delimiter behind it; j = delimiters seen when the record closes.  File mode feeds a virtual
 * '\n' first (buffer[Max_record-1], asearch.c:69-78): when the delimiter IS "\n" that byte closes
 * an empty record, so j runs one ahead and even the first record has a delimiter in front of it;
 * with any other delimiter the first record starts at the first text byte and, if the text opens
 * with the delimiter, counting starts at -1 (asearch.c:79-84).
 * wide = text[start - pre, end + post): pre is 0 or dlen, post <= dlen.  Returns 0, -1 (error) or
 * 1 (an output limit of -L was reached: stop, asearch.c:171-175). */
static int emit_one(const unsigned char *wide, size_t wlen, size_t pre, size_t body, uint64_t index,
                    uint64_t end_off, const unsigned char *delim, int dlen, int lead_delim)
{
    const int virt = dlen == 1 && delim[0] == '\n';
    const size_t post = wlen - pre - body;
    const size_t vpre = (pre == 0 && virt) ? 1 : 0;     /* the virtual '\n' */
    static unsigned char *rec;                          /* (one buffer for all records: grown, never shrunk) */
    static size_t rec_cap;
    int i2, j, rc = 0;
    if (wlen + (size_t)dlen + 4 > rec_cap) {
        free(rec);
        rec_cap = (wlen + (size_t)dlen + 4) * 2 + 256;
        rec = (unsigned char *)malloc(rec_cap);
        if (!rec) { rec_cap = 0; return shim_fail("out of memory"); }
    }
    if (vpre) rec[0] = '\n';
    memcpy(rec + vpre, wide, wlen);
    /* the delimiter the reference appends at end of input (asearch.c:87-91) */
    if (post < (size_t)dlen) memcpy(rec + vpre + pre + body, delim, (size_t)dlen);
    rec[vpre + pre + body + (size_t)dlen] = '\0';
    i2 = (int)(vpre + pre + body) - 1;
    j = (int)index + 1 + virt - ((DELIMITER && lead_delim) ? 1 : 0);
    CurrentByteOffset = (int)(end_off + 1);
    TRUNCATE = 0;
    if (-1 == output(rec, 0, i2, j)) rc = -1;
    if (rc == 0 && ((LIMITOUTPUT > 0 && LIMITOUTPUT <= num_of_matched) ||
                    (LIMITPERFILE > 0 && LIMITPERFILE <= num_of_matched - prev_num_of_matched)))
        rc = 1;
    return rc;
}

/* ---- one scan: count / -l / records through output() ------------------------------------ */
struct text_src {
    int fd;                     /* >= 0: file / pipe; -1: memory */
    const unsigned char *mem;
    size_t mem_len;
};

/* File mode prints while the file is still being read, as asearch.c:66-324 does from inside its block loop:
 * agh_scan_fd_emit streams the input through two bounded device segments and hands every segment's matched
 * records to this callback in file order -- each with the delimiter in front of it and behind it
 * (AGH_EMIT_HEAD_DELIM | AGH_EMIT_TAIL_DELIM: the buffer shape of asearch.c:162-170), so output() is called
 * exactly as before, only earlier and without the whole file in HBM. */
struct stream_ctx {
    const agh_query *q;
    const unsigned char *delim;
    int dlen, lead_delim, lead_known, rc;
};

static int stream_emit(void *vctx, const agh_match *m, size_t n, const unsigned char *bytes, size_t n_bytes)
{
    struct stream_ctx *c = (struct stream_ctx *)vctx;
    size_t off = 0, i;
    if (!c->lead_known) {
        /* -d: does the input open with the delimiter (asearch.c:79-84 starts counting at -1)?  The library has
         * kept the first bytes of what it read: no second look at the input, so a pipe streams like a file */
        unsigned char first[AGH_MAX_DELIM];
        c->lead_delim = agh_input_head(c->q, first, (size_t)c->dlen) == (size_t)c->dlen &&
                        memcmp(first, c->delim, (size_t)c->dlen) == 0;
        c->lead_known = 1;
    }
    for (i = 0; i < n; i++) {
        const size_t pre = m[i].start < (uint64_t)c->dlen ? (size_t)m[i].start : (size_t)c->dlen;
        const size_t body = (size_t)(m[i].end - m[i].start);
        const size_t wlen = pre + body + (size_t)c->dlen;
        int rc;
        if (off + wlen > n_bytes) { c->rc = shim_fail("internal error: short record buffer"); return 1; }
        rc = emit_one(bytes + off, wlen, pre, body, m[i].index, m[i].end, c->delim, c->dlen, c->lead_delim);
        off += wlen;
        if (rc == 1) return 1;                          /* -L limits reached: stop reading */
        if (rc) { c->rc = rc; return 1; }
    }
    return 0;
}

/* The reference's main() ends in exit(ret) (main.c:79,96).  A process that has scanned on the GPU would then tear
 * down two device segments, the pinned ring and the HIP runtime itself: 50-80 ms of a 0.3 s run on a 4 GiB file
 * (profiles/r05_startup.log).  Registered at the first scan, this handler runs before the runtime's own (handlers
 * run in reverse order of registration): it flushes what the front end has printed and leaves with the same
 * status -- the kernel reclaims the rest.  AGH_CLI_TEARDOWN=1 keeps the orderly teardown.
 *
 * Only where the shim OWNS THE PROCESS: -DAGH_SHIM_OWNS_PROCESS=1 is given by the link recipe of the command-line
 * binary (the reference's main.o is the program: oracle/Makefile agrep_gpu).  A host application
 * that links the front end as a library and calls fileagrep() / fileagrep_search() (glimpse; SURVEY 8b "higher-level
 * API") is file mode too, and its own atexit handlers registered earlier must run: without the macro nothing is
 * armed.  (EXITONERROR cannot tell the two apart at run time: main.c:78 sets it to 1 and the initial_value() of every
 * agrep_init() puts it back to 0, agrep.c:347, before any engine runs.) */
#if defined(AGH_SHIM_OWNS_PROCESS) && AGH_SHIM_OWNS_PROCESS
static void fast_exit(int status, void *unused)
{
    (void)unused;
    fflush(NULL);
    _exit(status);
}

static void arm_fast_exit(void)
{
    static int armed;
    const char *e = getenv("AGH_CLI_TEARDOWN");
    if (armed || (e && e[0] == '1')) return;
    armed = 1;
    on_exit(fast_exit, NULL);
}
#else
static void arm_fast_exit(void) {}
#endif

static int run_scan(agh_query *q, const struct text_src *src, const unsigned char *delim, int dlen)
{
    agh_result res;
    agh_match *ms = NULL;
    unsigned char *bytes = NULL;
    unsigned flags = INVERSE ? AGH_INVERT : 0u;
    size_t cap = 65536, total = 0, off = 0;
    uint64_t i, text_len;
    int rc = 0, lead_delim = 0;

    if (src->fd >= 0) arm_fast_exit();          /* (command-line build only; memory mode: never) */
    if (COUNT || (FILENAMEONLY && (NEW_FILE || !POST_FILTER))) {
        flags |= FILENAMEONLY && !COUNT ? AGH_FILENAMEONLY : AGH_COUNT;
        rc = src->fd >= 0 ? agh_scan_fd(q, src->fd, flags, &res, NULL, 0)
                          : agh_scan_buffer(q, src->mem, src->mem_len, flags, &res, NULL, 0);
        if (rc) return shim_fail(agh_last_error());
        if (COUNT) {
            num_of_matched += (int)res.n_matched;   /* output() would count one by one */
            return 0;
        }
        return res.n_matched ? print_filename() : 0;
    }

    if (src->fd >= 0) {                         /* files and pipes alike, with or without -d */
        struct stream_ctx c;
        c.q = q;
        c.delim = delim;
        c.dlen = dlen;
        c.rc = 0;
        c.lead_delim = 0;
        c.lead_known = !DELIMITER;
        rc = agh_scan_fd_emit(q, src->fd, flags | AGH_EMIT_HEAD_DELIM | AGH_EMIT_TAIL_DELIM, &res, stream_emit, &c);
        if (rc) return shim_fail(agh_last_error());
        return c.rc;
    }

    ms = (agh_match *)malloc(cap * sizeof(*ms));
    if (!ms) return shim_fail("out of memory");
    /* memory mode (fd == -1): the caller's buffer is staged whole */
    rc = agh_scan_buffer(q, src->mem, src->mem_len, flags, &res, ms, cap);
    if (!rc && res.truncated) {                 /* more matches than guessed: the text is staged */
        free(ms);
        cap = (size_t)res.n_matched + 16;
        ms = (agh_match *)malloc(cap * sizeof(*ms));
        if (!ms) return shim_fail("out of memory");
        rc = agh_rescan_staged(q, flags, &res, ms, cap);
    }
    if (rc) { free(ms); return shim_fail(agh_last_error()); }
    text_len = res.n_bytes;
    if (res.n_stored == 0) { free(ms); return 0; }

    /* output() wants the record with the delimiter in front of it and behind it in one buffer
     * (asearch.c:162: output(buffer, lasti, i - D_length - 1, j), lasti = start of the delimiter
     * before the record): fetch [start - dlen, end + dlen) of every match in one go */
    {
        agh_match *wide = (agh_match *)malloc((size_t)res.n_stored * sizeof(*wide));
        agh_match head;
        unsigned char first[AGH_MAX_DELIM];
        if (!wide) { free(ms); return shim_fail("out of memory"); }
        for (i = 0; i < res.n_stored; i++) {
            wide[i].start = ms[i].start >= (uint64_t)dlen ? ms[i].start - (uint64_t)dlen : 0;
            wide[i].end = ms[i].end + (uint64_t)dlen <= text_len ? ms[i].end + (uint64_t)dlen : text_len;
            wide[i].index = ms[i].index;
            total += (size_t)(wide[i].end - wide[i].start);
        }
        bytes = (unsigned char *)malloc(total + (size_t)dlen + 16);
        if (!bytes) { free(ms); free(wide); return shim_fail("out of memory"); }
        if (agh_fetch_records(q, wide, (size_t)res.n_stored, bytes, total, &total)) {
            free(ms); free(wide); free(bytes);
            return shim_fail(agh_last_error());
        }
        /* does the text open with the delimiter?  (-d: asearch.c:79-84 starts counting at -1) */
        if (DELIMITER && text_len >= (uint64_t)dlen && ms[0].start > 0) {
            size_t got = 0;
            head.start = 0;
            head.end = (uint64_t)dlen;
            head.index = 0;
            if (agh_fetch_records(q, &head, 1, first, sizeof(first), &got) == 0 && got == (size_t)dlen)
                lead_delim = memcmp(first, delim, (size_t)dlen) == 0;
        }
        for (i = 0; i < res.n_stored && rc == 0; i++) {
            const size_t wlen = (size_t)(wide[i].end - wide[i].start);
            rc = emit_one(bytes + off, wlen, (size_t)(ms[i].start - wide[i].start),
                          (size_t)(ms[i].end - ms[i].start), ms[i].index, ms[i].end, delim, dlen, lead_delim);
            off += wlen;
            if (rc == 1) { rc = 0; break; }             /* -L limits reached */
        }
        free(wide);
    }
    free(ms);
    free(bytes);
    return rc;
}

static void text_of(int fd, struct text_src *src)
{
    src->fd = fd;
    src->mem = NULL;
    src->mem_len = 0;
    if (fd == -1) {
        /* memory mode (asearch.c:326-572): the caller's buffer opens with the newline that the
         * file mode supplies itself (buffer[Max_record-1] = '\n'); the device engines supply it
         * too, so the scan starts behind it */
        src->mem = agrep_inbuffer;
        src->mem_len = agrep_inlen > 0 ? (size_t)agrep_inlen : 0;
        if (src->mem_len && src->mem[0] == '\n') {
            src->mem++;
            src->mem_len--;
        }
    }
}

/* The simple-pattern engines print through s_output() (sgrep.c:1274-1483) in the reference: the
 * record with its delimiter where -d puts it, no "eat the first newline" step.  output() does the
 * same once it sees the state bitap() would have left: D_length = the delimiter's real length
 * (the sgrep path keeps 2 for the default "\n; ", agrep.c:378) and FIRSTOUTPUT already spent. */
static int run_simple(agh_query *q, const struct text_src *src, const unsigned char *delim, int dlen)
{
    const int saved = D_length;
    int rc;
    FIRSTOUTPUT = 0;
    if (!DELIMITER) D_length = 1;
    rc = run_scan(q, src, delim, dlen);
    D_length = saved;
    return rc;
}

/* ---- bitap(): maskgen()'s tables, unchanged -------------------------------------------- */
static agh_query *g_bq;
static unsigned g_bq_sum;
static int g_bq_valid;

static unsigned tables_sum(const char *old_D_pat, int M, int D)
{
    unsigned h = 2166136261u, c;
    const unsigned char *p;
    size_t n;
#define MIX(ptr, len) for (p = (const unsigned char *)(ptr), n = (len); n; --n) h = (h ^ *p++) * 16777619u
    MIX(Mask, 256 * sizeof(unsigned));
    MIX(&Init[0], sizeof(unsigned));
    MIX(&Init1, sizeof(unsigned));
    MIX(&NO_ERR_MASK, sizeof(unsigned));
    MIX(&endposition, sizeof(unsigned));
    MIX(&D_endpos, sizeof(unsigned));
    MIX(old_D_pat, strlen(old_D_pat));
    c = (unsigned)M * 31u + (unsigned)D * 7u + (unsigned)AND + (unsigned)(JUMP ? (I * 64 + S * 8 + DD) * 1024 : 0);
    MIX(&c, sizeof(c));
#undef MIX
    return h;
}

int bitap(char old_D_pat[], char *Pattern, int fd, int M, int D)
{
    struct text_src src;
    unsigned i, sum;
    (void)Pattern;
    D_length = (int)strlen(old_D_pat);
    for (i = 0; i < (unsigned)D_length; i++)            /* bitap.c:92-94 */
        if (old_D_pat[i] == '^' || old_D_pat[i] == '$') old_D_pat[i] = '\n';

    if (REGEX) {                                        /* bitap.c:96-113: the CPU regex engines */
        if (D > 4)
            return shim_fail("the maximum number of erorrs allowed for full regular expressions is 4");
        return M <= SHORTREG ? re(fd, M, D) : re1(fd, M, D);
    }
    if (I == 0) Init1 = 037777777777u;                  /* -p: bitap.c:123, asearch.c:49 */

    sum = tables_sum(old_D_pat, M, D);
    if (!g_bq_valid || sum != g_bq_sum) {
        if (g_bq) agh_query_free(g_bq);
        g_bq = agh_query_from_maskgen((const uint32_t *)Mask, Init[0], Init1, NO_ERR_MASK, endposition,
                                      D_endpos, M, (const unsigned char *)old_D_pat, D_length, D, AND);
        if (g_bq && D > 0 && JUMP && agh_query_set_costs(g_bq, I, S, DD)) {    /* asearch1.c */
            agh_query_free(g_bq);
            g_bq = NULL;
        }
        g_bq_valid = g_bq != NULL;
        g_bq_sum = sum;
        if (!g_bq) return shim_fail(agh_last_error());
    }
    text_of(fd, &src);
    return run_scan(g_bq, &src, (const unsigned char *)old_D_pat, D_length);
}

/* AGH_REF_QUIRKS=q6: reproduce quirk Q6 of the reference on request -- its simple-pattern engines
 * fold case whether or not -i was given (char_tr() fills TR[] unconditionally, sgrep.c:226-236;
 * bm() compares through TR[]), so `agrep word file` also prints "Word".  Off by default: the GPU
 * engines are case-sensitive without -i, like the reference's own bitap path. */
static int quirk_q6(void)
{
    const char *e = getenv("AGH_REF_QUIRKS");
    return e && (strstr(e, "q6") || strstr(e, "Q6"));
}

/* the guards the simple-pattern engines read from globals: -i, -w (bm()'s isalnum test,
 * sgrep.c:750-756; monkey1()'s, newmgrep.c:869-872), -x (char_tr()'s "\n pat \n", sgrep.c:252-259) */
static unsigned simple_qflags(int D)
{
    return ((NOUPPER || (D == 0 && quirk_q6())) ? AGH_Q_NOCASE : 0u) | (WORDBOUND ? AGH_Q_WORD : 0u) |
           (WHOLELINE ? AGH_Q_WHOLELINE : 0u);
}

/* ---- sgrep(): the simple-pattern engines ------------------------------------------------ */
static agh_query *g_sq;
static unsigned char g_sq_pat[300];
static int g_sq_m = -1, g_sq_D, g_sq_dlen;
static unsigned g_sq_i;
static unsigned char g_sq_delim[AGH_MAX_DELIM + 1];

int sgrep(unsigned char *in_pat, int in_m, int fd, int D, int samepattern)
{
    unsigned char pat[300];
    const unsigned char *delim = (const unsigned char *)"\n";
    int m = in_m, k, j, dlen = 1;
    unsigned qf;
    struct text_src src;
    (void)samepattern;
    if (m <= 0 || m >= 256) return shim_fail("pattern too long");
    memcpy(pat, in_pat, (size_t)m);
    pat[m] = '\0';
    if (!CONSTANT) {                                    /* sgrep.c:289-293 */
        if (pat[0] == '^' || pat[0] == '$') pat[0] = '\n';
        if (m > 1 && pat[m - 2] != '\\' && (pat[m - 1] == '^' || pat[m - 1] == '$')) pat[m - 1] = '\n';
    }
    for (k = 0; k < m; k++)                             /* sgrep.c:294-300: backslash quotes */
        if (pat[k] == '\\') {
            for (j = k; j < m; j++) pat[j] = pat[j + 1];
            m--;
        }
    if (DELIMITER) {                                    /* agrep.c:3182-3185: the exact bytes */
        delim = D_pattern;
        dlen = D_length;
    }
    if (dlen < 1 || dlen > AGH_MAX_DELIM) return shim_fail("delimiter pattern too long");
    qf = simple_qflags(D);
    if (!g_sq || m != g_sq_m || D != g_sq_D || qf != g_sq_i || dlen != g_sq_dlen ||
        memcmp(pat, g_sq_pat, (size_t)m) || memcmp(delim, g_sq_delim, (size_t)dlen)) {
        if (g_sq) agh_query_free(g_sq);
        g_sq = agh_query_literal_ex(pat, m, D, qf, delim, dlen);
        if (!g_sq) { g_sq_m = -1; return shim_fail(agh_last_error()); }
        memcpy(g_sq_pat, pat, (size_t)m);
        memcpy(g_sq_delim, delim, (size_t)dlen);
        g_sq_m = m; g_sq_D = D; g_sq_i = qf; g_sq_dlen = dlen;
    }
    text_of(fd, &src);
    return run_simple(g_sq, &src, delim, dlen);
}

/* ---- prepf() / mgrep(): -f pattern files ------------------------------------------------ */
static agh_query *g_mq;
static const unsigned char **g_mp;
static int *g_ml, g_mn, g_mq_dlen;
static unsigned g_mq_i = ~0u;
static unsigned char g_mq_delim[AGH_MAX_DELIM + 1];

int prepf(int mfp, unsigned char *mbuf, int mlen)
{
    unsigned char *buf;
    size_t len = 0, i, start, w;
    int cap = 64;
    if (mfp == -1 && (mbuf == NULL || mlen <= 0)) return -1;    /* newmgrep.c:203 */
    if (mfp != -1) {
        struct stat sb;
        size_t have = 0;
        if (fstat(mfp, &sb) == -1 || !S_ISREG(sb.st_mode)) {
            fprintf(stderr, "%s: pattern file not regular file\n", Progname);
            return -1;
        }
        buf = (unsigned char *)malloc((size_t)sb.st_size + 2);
        if (!buf) return -1;
        while (have < (size_t)sb.st_size) {
            ssize_t r = read(mfp, buf + have, (size_t)sb.st_size - have);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) break;
            have += (size_t)r;
        }
        len = have;
    } else {
        buf = (unsigned char *)malloc((size_t)mlen + 2);
        if (!buf) return -1;
        memcpy(buf, mbuf, (size_t)mlen);
        len = (size_t)mlen;
    }
    buf[len] = '\n';
    if (pat_spool) free(pat_spool);
    pat_spool = (char *)buf;                            /* owned here, freed by agrep.c:333 */
    free(g_mp);
    free(g_ml);
    g_mp = (const unsigned char **)malloc((size_t)cap * sizeof(*g_mp));
    g_ml = (int *)malloc((size_t)cap * sizeof(*g_ml));
    g_mn = 0;
    /* one pattern per line, a backslash quotes the next byte (newmgrep.c:262-300), in place */
    for (i = 0, start = 0; i <= len; i++) {
        if (buf[i] != '\n' && i < len) continue;
        {
            size_t r = start;
            w = start;
            while (r < i) {
                if (buf[r] == '\\' && r + 1 < i) r++;
                buf[w++] = buf[r++];
            }
        }
        if (w > start) {
            if (g_mn == cap) {
                cap *= 2;
                g_mp = (const unsigned char **)realloc(g_mp, (size_t)cap * sizeof(*g_mp));
                g_ml = (int *)realloc(g_ml, (size_t)cap * sizeof(*g_ml));
            }
            g_mp[g_mn] = buf + start;
            g_ml[g_mn] = (int)(w - start);
            g_mn++;
        }
        start = i + 1;
    }
    if (g_mq) { agh_query_free(g_mq); g_mq = NULL; }
    return 0;
}

/* Boolean patterns: agrep_search turns "a;b" / "a,b" into a multi-pattern search whose terminals
 * come through prepf() and whose operator arrives as mgrep()'s second argument (asplit.c, the call
 * at agrep.c:3357: mgrep(fd, AParse)).  OR = any terminal = the multi-pattern query.  AND = every
 * terminal somewhere in the record (newmgrep.c:903-905): one exact scan per terminal, the record
 * lists intersected here.  Parse TREES (parentheses, mixed operators) stay with the CPU. */
#define AND_EXP 0x1             /* agrep.h:144 */
extern int AComplexBoolean;

static int mgrep_all_terminals(int fd, const unsigned char *delim, int dlen)
{
    struct text_src src;
    unsigned char *own = NULL;
    const unsigned char *text;
    size_t len = 0, cap = 0;
    agh_match *cur = NULL, *ms = NULL;
    size_t ncur = 0;
    int t, rc = 0, lead_delim;
    text_of(fd, &src);
    if (fd >= 0) {                                      /* every terminal scans the same bytes */
        for (;;) {
            ssize_t r;
            if (len == cap) {
                cap = cap ? cap * 2 : (1u << 20);
                own = (unsigned char *)realloc(own, cap);
                if (!own) return shim_fail("out of memory");
            }
            r = read(fd, own + len, cap - len);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) break;
            len += (size_t)r;
        }
        text = own;
    } else {
        text = src.mem;
        len = src.mem_len;
    }
    for (t = 0; t < g_mn && rc == 0; t++) {
        agh_query *q = agh_query_literal_ex(g_mp[t], g_ml[t], 0, simple_qflags(0), delim, dlen);
        agh_result res;
        size_t mcap = 65536, a, b, w;
        if (!q) { rc = shim_fail(agh_last_error()); break; }
        for (;;) {
            free(ms);
            ms = (agh_match *)malloc(mcap * sizeof(*ms));
            if (!ms || agh_scan_buffer(q, text, len, 0, &res, ms, mcap)) { rc = shim_fail(ms ? agh_last_error() : "out of memory"); break; }
            if (!res.truncated) break;
            mcap = (size_t)res.n_matched + 16;
        }
        agh_query_free(q);
        if (rc) break;
        if (t == 0) {
            cur = ms;
            ncur = (size_t)res.n_stored;
            ms = NULL;
        } else {                                        /* both lists are in file order */
            for (a = b = w = 0; a < ncur && b < (size_t)res.n_stored;) {
                if (cur[a].start == ms[b].start) { cur[w++] = cur[a]; a++; b++; }
                else if (cur[a].start < ms[b].start) a++;
                else b++;
            }
            ncur = w;
        }
    }
    free(ms);
    if (rc == 0) {
        if (COUNT) {
            num_of_matched += (int)ncur;
        } else if (FILENAMEONLY && (NEW_FILE || !POST_FILTER)) {
            if (ncur) rc = print_filename();
        } else {
            size_t i;
            lead_delim = DELIMITER && len >= (size_t)dlen && memcmp(text, delim, (size_t)dlen) == 0;
            for (i = 0; i < ncur && rc == 0; i++) {
                const uint64_t ws = cur[i].start >= (uint64_t)dlen ? cur[i].start - (uint64_t)dlen : 0;
                const uint64_t we = cur[i].end + (uint64_t)dlen <= len ? cur[i].end + (uint64_t)dlen : len;
                rc = emit_one(text + ws, (size_t)(we - ws), (size_t)(cur[i].start - ws),
                              (size_t)(cur[i].end - cur[i].start), cur[i].index, cur[i].end, delim, dlen,
                              lead_delim);
                if (rc == 1) { rc = 0; break; }
            }
        }
    }
    free(cur);
    free(own);
    return rc;
}

int mgrep(int fd, void *AParse)
{
    const unsigned char *delim = (const unsigned char *)"\n";
    int dlen = 1;
    struct text_src src;
    if (g_mn <= 0) return 0;
    if (DELIMITER) {
        delim = D_pattern;
        dlen = D_length;
    }
    if (AParse != NULL) {
        if (AComplexBoolean)
            return shim_fail("boolean patterns with parentheses / mixed operators are not served by the GPU engines");
        if (INVERSE) return shim_fail("-v with a boolean pattern is not served by the GPU engines");
        if ((long)AParse & AND_EXP) {
            const int saved = D_length;
            int rc;
            FIRSTOUTPUT = 0;
            if (!DELIMITER) D_length = 1;
            rc = mgrep_all_terminals(fd, delim, dlen);
            D_length = saved;
            return rc;
        }
    }
    if (!g_mq || g_mq_i != simple_qflags(0) || g_mq_dlen != dlen || memcmp(delim, g_mq_delim, (size_t)dlen)) {
        if (g_mq) agh_query_free(g_mq);
        g_mq = agh_query_multi_ex(g_mp, g_ml, g_mn, simple_qflags(0), delim, dlen);
        if (!g_mq) return shim_fail(agh_last_error());
        g_mq_i = simple_qflags(0);
        g_mq_dlen = dlen;
        memcpy(g_mq_delim, delim, (size_t)dlen);
    }
    text_of(fd, &src);
    return run_simple(g_mq, &src, delim, dlen);
}
