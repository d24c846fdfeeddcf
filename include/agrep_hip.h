/*
 * agrep_hip.h -- C-ABI of libagrep_hip.so: the MI355X (gfx950) replacement for the scan
 * engines of Wikinaut/agrep (layer L4 of SURVEY.md: bitap.c / asearch.c / sgrep.c).
 *
 * The reference has no plugin API; the seam is the three-way engine call in exec()
 * (agrep.c:3357-3361 and :3428-3432):
 *
 *     if (PAT_FILE || PAT_BUFFER) mgrep(fd, AParse);
 *     else if (SGREP)             ret = sgrep(OldPattern, strlen(OldPattern), fd, D, i);
 *     else                        ret = bitap(old_D_pat, Pattern, fd, M, D);
 *
 * A maintainer replaces the second and third arm by  agh_query_* + agh_scan_*  (the cgo-
 * style stub is in INTEGRATION.md).  Plain C types only: no C++, no torch, no HIP types.
 *
 * Conventions kept from the reference:
 *   - errors: return -1 / NULL and set errno = AGH_ERRNO (= AGREP_ERROR 123, agrep.h:173);
 *     never abort, never print (agh_last_error() holds the message).
 *   - fd >= 0: read sequentially to EOF, caller opens/closes (agrep.c:3411-3424);
 *     memory mode (fd == -1 in the reference, AGREP_POINTER): agh_scan_buffer().
 *   - the library never frees or keeps caller memory.
 *   - match semantics: a record is the text between delimiters; a record matches iff the
 *     k-error automaton of asearch.c:94-199 reports it (== some substring within
 *     Levenshtein distance k of the pattern).  Records are counted once
 *     (no Q4 double count), in file order.  None of the reference quirks Q1..Q10 is
 *     reproduced.
 *   - the product path has no CPU fallback: without a usable HIP device every entry point
 *     fails with -1.
 */
#ifndef AGREP_HIP_H
#define AGREP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGH_ERRNO        123   /* agrep.h:173 AGREP_ERROR */
#define AGH_MAX_PATTERN  64    /* one 64-bit state word per error level (reference: 29) */
#define AGH_MAX_ERRORS   8     /* agrep.h:44 MaxError */
#define AGH_MAX_DELIM    8     /* agrep.h:34 MAXDELIM */

/* scan flags (the reference passes these through globals, asearch.c:4-24) */
#define AGH_COUNT          0x01u  /* -c  COUNT: only count matched records */
#define AGH_FILENAMEONLY   0x02u  /* -l  FILENAMEONLY: caller only needs n_matched > 0 */
#define AGH_INVERT         0x04u  /* -v  INVERSE (asearch.c:128): the records that do NOT match */
#define AGH_FORCE_FULLSCAN 0x10u  /* diagnostics: skip the q-gram filter, run the automaton
                                     over every byte (the asearch.c shape) */
#define AGH_FORCE_FILTER   0x20u  /* diagnostics: fail instead of falling back to full scan */
#define AGH_TIME_SWEEP     0x80u  /* also time the sweep kernel alone (agh_result.sweep_ms) with two extra
                                     HIP events around it on the scan's stream: ~11 us per scan */
#define AGH_TIME_SCAN      0x100u /* time the whole kernel sequence of the scan (agh_result.device_ms): two HIP
                                     events as well */
#define AGH_FORCE_NUMBERED 0x40u  /* diagnostics: compute record numbers even for -c / -l scans */
#define AGH_NO_BYTES       0x200u /* agh_scan_*_emit: offsets and record numbers only, no record bytes */
#define AGH_EMIT_HEAD_DELIM 0x400u /* agh_scan_*_emit: bytes of record i = text[start - pre, end) with pre = min(dlen, start):
                                      the delimiter in front of the record, as far as the input has one there */
#define AGH_EMIT_TAIL_DELIM 0x800u /* agh_scan_*_emit: ... followed by the dlen bytes behind the record: the delimiter from
                                      the text, or -- behind a last, unterminated record -- the one the reference appends
                                      (asearch.c:87-91).  Both together: the buffer asearch.c:162-170 hands to output();
                                      tail alone with the default delimiter: byte for byte the lines the reference prints */

/* engine that produced a result */
#define AGH_ENGINE_FULLSCAN 1u    /* k-error automaton over every byte (asearch.c:94-116) */
#define AGH_ENGINE_FILTER   2u    /* lossless q-gram sample filter + automaton on candidate
                                     windows (the role of sgrep.c:1130-1239) */

typedef struct agh_query agh_query;   /* compiled pattern + device tables, opaque */

/* One matched record, delimiters excluded: text[start, end). */
typedef struct {
    uint64_t start;
    uint64_t end;
    uint64_t index;       /* 0-based record number (the reference's -n prints index+1) */
} agh_match;

typedef struct {
    uint64_t n_matched;     /* what the engines add to num_of_matched (agrep.c:148) */
    uint64_t n_records;     /* records in the scanned text; 0 when a count-only scan (AGH_COUNT /
                               AGH_FILENAMEONLY) took the lean path that never numbers records */
    uint64_t n_bytes;       /* bytes scanned */
    uint64_t n_candidates;  /* filter engine: candidate windows verified */
    uint64_t n_stored;      /* matches written to the caller's agh_match array */
    uint32_t engine;        /* AGH_ENGINE_* */
    uint32_t truncated;     /* 1: more matches than the agh_match array could hold */
    double   device_ms;     /* AGH_TIME_SCAN: GPU time of the whole scan (hipEvent), excluding staging; else 0 */
    double   sweep_ms;      /* AGH_TIME_SWEEP (else 0): of which the k_sweep / k_sweep_fused kernel launches (the kernel that
                               reads every byte), hipEvents recorded right around them on the scan stream */
    uint32_t sweep_launches;/* AGH_TIME_SWEEP: number of k_sweep launches sweep_ms is the sum of */
    uint32_t lean_reruns;   /* segments whose count-only (lean) scan gave up (a record start more than 1 MiB
                               in front of a match, hash set full, candidate slices full) and were scanned
                               again on the numbered pipeline: the result is exact, the time doubled */
    uint32_t n_segments;    /* kernel sequences the text was cut into (<= 8 GiB each, at record boundaries) */
    uint32_t fused_segments;/* of which count-only segments ran as ONE kernel (k_sweep_fused: sweep + verify;
                               sweep_ms then covers that kernel) */
    uint32_t copied_segments;/* of which segments were scanned from an aligned copy: no record ended on a
                               16-byte boundary near the cut (fixed-width records behind an odd header) */
    uint32_t reserved;
} agh_result;

/* ---- query construction ------------------------------------------------------------- */

/* Replaces sgrep()'s own pattern processing (sgrep.c:289-320: char_tr/prep/initmask) and,
 * for literal patterns, preprocess()+maskgen() (preproce.c:181-228, maskgen.c:26-269).
 * pat[0..m): literal bytes, 1 <= m <= AGH_MAX_PATTERN, D = number of errors (-#),
 * 0 <= D <= AGH_MAX_ERRORS and D < m (checksg.c:34-41).  nocase = -i (ASCII folding,
 * maskgen.c:259-266).  delim[0..dlen): record delimiter, "\n" by default (-d). */
agh_query *agh_query_literal(const unsigned char *pat, int m, int D, int nocase,
                             const unsigned char *delim, int dlen);

/* The same with the guards of the simple-pattern engines (the globals sgrep() and mgrep() consult):
 *   AGH_Q_NOCASE     -i   NOUPPER
 *   AGH_Q_WORD       -w   WORDBOUND: the occurrence must have a non-alphanumeric byte (isalnum() of the
 *                         C locale; the ends of the text count) on both sides -- bm()'s test,
 *                         sgrep.c:750-756, monkey1()'s for -f, newmgrep.c:869-872
 *   AGH_Q_WHOLELINE  -x   WHOLELINE: char_tr() wraps the pattern into "\n pat \n" (sgrep.c:252-259):
 *                         the occurrence is a whole line
 * With D > 0 no error may touch the guard positions (maskgen.c:171-187 sets their NO_ERR_MASK bits).
 * m <= AGH_MAX_PATTERN - 2 with a guard.  AGH_Q_WORD together with AGH_Q_WHOLELINE is refused, as the
 * reference refuses -w with -x (agrep.c:2188-2196). */
#define AGH_Q_NOCASE    0x1u
#define AGH_Q_WORD      0x2u
#define AGH_Q_WHOLELINE 0x4u
agh_query *agh_query_literal_ex(const unsigned char *pat, int m, int D, unsigned qflags,
                                const unsigned char *delim, int dlen);

/* Replaces the consumption of maskgen()'s globals by bitap()/asearch()/asearch0()
 * (externs at bitap.c:43-64, asearch.c:4-24): pass the reference's tables unchanged.
 * Mask[256], Init0 = Init[0], Init1, NO_ERR_MASK, endposition, D_endpos as maskgen() left
 * them (maskgen.c:218-266); M = maskgen's return value; old_D_pat / D_length = raw
 * delimiter (asearch.c:54); D = errors; AND = the AND flag (maskgen.c:150-163).
 * Literals, [classes], -w / -x guards and <exact> segments (NO_ERR_MASK) run on the byte-
 * parallel engines; tables with '#' wildcards (sticky Init1 bits, maskgen.c:231-232) or
 * ';' AND / ',' OR (several endposition bits, maskgen.c:136-163) are kept unchanged and run
 * record-parallel by the table engine (delimiters of 1..8 bytes, also letters under -i; edit costs
 * through agh_query_set_costs).  NULL for a pattern that matches the empty record and for malformed
 * tables. */
agh_query *agh_query_from_maskgen(const uint32_t Mask[256], uint32_t Init0, uint32_t Init1,
                                  uint32_t NO_ERR_MASK, uint32_t endposition,
                                  uint32_t D_endpos, int M, const unsigned char *old_D_pat,
                                  int D_length, int D, int AND);

/* Replaces preprocess() + maskgen() (preproce.c:137-332, maskgen.c:26-269) for callers that do not link the
 * reference's front end: agrep's non-regex pattern language -- c  \c  [a-fxyz]  [^...]  .  #  <...>  ^  $  a;b
 * a,b  and the -w / -x / -i options (qflags: AGH_Q_*) -- compiled by the library itself.
 * agh_compile_pattern is host-only (no device needed): it fills the tables in maskgen's own layout -- the ones
 * agh_query_from_maskgen takes, bit for bit what the reference's maskgen() leaves in its globals
 * (tests/test_pattern_compiler.py compares them with the reference's).  M <= 31 positions: the delimiter, one
 * separator and the pattern (maskgen.c:201-208: 29 - |delimiter| pattern positions).  Regular expressions ( * | ( ) ), the boolean-pattern
 * syntax ( { } ~ ), unescaped meta
 * characters inside [] and a '-' inside [] that is not between two bytes of its own ([a-c-e], [-a], [a-]:
 * the reference reads these in ways of its own) are refused (-1, errno 123): what compiles, compiles like
 * the reference.  simple = 1: a plain literal (nothing but bytes, \c and
 * the option guards).
 * agh_query_pattern = compile + the query: a plain literal goes to agh_query_literal_ex (sample filter), the
 * rest to agh_query_from_maskgen (classes / <> on the byte-parallel engines, # ; , on the table engine). */
typedef struct {
    uint32_t Mask[256];
    uint32_t Init0, Init1, NO_ERR_MASK, endposition, D_endpos, wildmask;
    int M, AND, simple;
} agh_pattern_tables;
int agh_compile_pattern(const unsigned char *pat, int len, unsigned qflags, const unsigned char *delim, int dlen,
                        agh_pattern_tables *out);
agh_query *agh_query_pattern(const unsigned char *pat, int len, int D, unsigned qflags,
                             const unsigned char *delim, int dlen);

/* Replaces prepf() (newmgrep.c:192-375, called from agrep_init for -f / -m): npat literal
 * patterns pats[i][0..lens[i]).  A record matches iff it contains any pattern verbatim --
 * exact matching only, like mgrep() (compat.c:34-37 ignores -# with -f).  nocase = -i. */
agh_query *agh_query_multi(const unsigned char *const *pats, const int *lens, int npat,
                           int nocase, const unsigned char *delim, int dlen);
/* ... with -w / -x (AGH_Q_* above; newmgrep.c:835-872): exact patterns only. */
agh_query *agh_query_multi_ex(const unsigned char *const *pats, const int *lens, int npat,
                              unsigned qflags, const unsigned char *delim, int dlen);

/* -f with errors (BASELINE config 5; beyond the reference, which ignores -# together with -f,
 * compat.c:34-37): a record matches iff it holds a substring within edit distance D of ANY
 * pattern -- the union over the patterns of agh_query_literal(pat, D)'s predicate.  Pattern
 * lengths D+1..32, no delimiter / newline bytes inside patterns; D = 0 is agh_query_multi. */
agh_query *agh_query_multi_approx(const unsigned char *const *pats, const int *lens, int npat,
                                  int D, int nocase, const unsigned char *delim, int dlen);

/* Replaces the cost globals I, S, DD of asearch1() (asearch1.c:28-44, options -I# -S# -D#,
 * agrep.c:2680-2696): cost of an insertion / substitution / deletion, each >= 1; a record
 * matches iff some substring is within total cost D.  Costs above D behave as D + 1. */
int agh_query_set_costs(agh_query *q, int I, int S, int DD);

void agh_query_free(agh_query *q);

/* Introspection used by tests/bench: pattern length, errors, filter sample shape
 * (q bytes every h bytes; h == 0: the filter does not apply, full scan only). */
int agh_query_info(const agh_query *q, int *m, int *D, int *filter_q, int *filter_h);

/* ---- scanning ----------------------------------------------------------------------- */

/* Memory mode (the reference's fd == -1 path, asearch.c:326-572): text[0..len) is host
 * memory holding the raw file contents (no leading delimiter needed, no slack needed).
 * matches/cap may be NULL/0 (then only counts are produced). */
int agh_scan_buffer(agh_query *q, const unsigned char *text, size_t len, unsigned flags,
                    agh_result *res, agh_match *matches, size_t cap);

/* File mode (fd >= 0, asearch.c:66-324 / sgrep.c:334-547): reads fd to EOF through a ring of pinned
 * buffers (read() of the next chunks overlaps the H2D copy of chunk i); works for pipes too.  Count-only
 * scans (matches == NULL) stream through two device segments of bounded size; with a match ARRAY the whole
 * input is staged in HBM (agh_fetch_records / agh_rescan_staged work on that copy) -- for record output of
 * inputs of any size use agh_scan_fd_emit. */
int agh_scan_fd(agh_query *q, int fd, unsigned flags, agh_result *res, agh_match *matches,
                size_t cap);

/* Record output while the input is still being read -- what asearch.c:66-324 / bitap.c:169-284 do when they
 * call output() from inside their block loop.  The input streams through two device segments of bounded
 * size (the scan of one overlaps the read + H2D copy of the next); after every segment emit() receives that
 * segment's matched records in file order: m[0..n) (offsets relative to the start of the input, index = the
 * record number), and bytes[0..n_bytes) = the records themselves back to back (record i has m[i].end -
 * m[i].start bytes, plus its delimiters with AGH_EMIT_HEAD_DELIM / AGH_EMIT_TAIL_DELIM; NULL / 0 with
 * AGH_NO_BYTES).  A segment with many or long matched records comes in several emit() calls (at most 2^20
 * records / 64 MiB of record bytes each, file order kept).  emit() is called from a worker thread of the
 * library, never concurrently; a non-zero return stops the scan (the call returns 0 with res->truncated = 1).
 * HBM use does not grow with the input, pipes work, records of any number come out.  The list is produced in
 * file order on the device (ordered compaction of the record bitmap: no sort) and comes back in one copy per
 * emit().  flags: AGH_INVERT, AGH_NO_BYTES, AGH_EMIT_HEAD_DELIM, AGH_EMIT_TAIL_DELIM. */
typedef int (*agh_emit_fn)(void *ctx, const agh_match *m, size_t n, const unsigned char *bytes, size_t n_bytes);
int agh_scan_fd_emit(agh_query *q, int fd, unsigned flags, agh_result *res, agh_emit_fn emit, void *ctx);
/* ... restricted to the byte range [begin, end) of a seekable file (one rank's shard): offsets and record numbers
 * are relative to `begin` (the shard is scanned as an input of its own: add begin / the record count of the
 * shards in front); AGH_EMIT_HEAD_DELIM gives the first record of the shard no delimiter in front. */
int agh_scan_fd_range_emit(agh_query *q, int fd, uint64_t begin, uint64_t end, unsigned flags,
                           agh_result *res, agh_emit_fn emit, void *ctx);
/* The first bytes (at most 64) of the input of the streamed scan that is running or ran last on this query
 * (agh_scan_fd / agh_scan_fd_emit and their ranged forms): known from the first chunk on, i.e. inside every emit()
 * call.  What asearch.c:79-84 looks at -- "does the input open with the delimiter?" decides where its record
 * numbers start -- without a second read of the input: a pipe can be streamed too.  Returns the bytes copied. */
size_t agh_input_head(const agh_query *q, unsigned char *out, size_t cap);
/* The same for text already resident in HBM (dev_text as for agh_scan_device): numbered scan, record bounds
 * and the gather of the record bytes on the device. */
int agh_scan_device_emit(agh_query *q, const void *dev_text, size_t len, unsigned flags, agh_result *res,
                         agh_emit_fn emit, void *ctx);

/* Scan again the text the last agh_scan_fd / agh_scan_buffer staged in HBM (a pipe cannot be
 * read twice): used after `truncated` with a larger match array. */
int agh_rescan_staged(agh_query *q, unsigned flags, agh_result *res, agh_match *matches,
                      size_t cap);

/* Bytes of matched records of the most recent agh_scan_fd / agh_scan_buffer on this query,
 * concatenated in the order of m[] (no delimiters in between) -- what output() prints
 * (agrep.c:3930-3945) -- gathered on the device and copied back in one piece.  out_len
 * receives the number of bytes (also when out_cap is too small, which fails). */
int agh_fetch_records(agh_query *q, const agh_match *m, size_t n_matches, unsigned char *out,
                      size_t out_cap, size_t *out_len);

/* Text already resident in HBM (the measured configuration): dev_text is a device pointer,
 * 16-byte aligned, readable up to the next multiple of 16 bytes after len.  stream is a
 * hipStream_t (NULL = default stream).  dev_match_pos (optional device pointer to
 * match_cap uint64) receives, in file order, one byte offset inside each matched record. */
int agh_scan_device(agh_query *q, const void *dev_text, size_t len, void *stream,
                    unsigned flags, agh_result *res, void *dev_match_pos, size_t match_cap);

/* ---- multi-GPU: records shard, the only exchange is the aggregate --------------------- */

/* Records are independent (every engine resets at a delimiter, asearch.c:175-196), so G GPUs
 * scan G record-aligned byte ranges with no data-path collective; what exec() prints per file
 * (agrep.c:3444-3558: the -c count, the -l name) needs one tiny all-reduce, done here directly
 * on RCCL (ncclAllReduce over xGMI).  RCCL is opened at the first call (dlopen), never before.
 * Two ways to form the communicator:
 *   one process per GPU:  rank 0 calls agh_comm_unique_id() and hands the 128 bytes to the other
 *       ranks by any means (MPI, a file, torch.distributed); every rank then calls
 *       agh_comm_init_rank() with its own device selected (agh_set_device);
 *   one process, N GPUs:  agh_comm_init_all() (ncclCommInitAll), one communicator per device; the
 *       *_all reductions run them as one RCCL group. */
#define AGH_UNIQUE_ID_BYTES 128
typedef struct agh_comm agh_comm;
int agh_comm_unique_id(unsigned char id[AGH_UNIQUE_ID_BYTES]);
agh_comm *agh_comm_init_rank(const unsigned char id[AGH_UNIQUE_ID_BYTES], int nranks, int rank);
int agh_comm_init_all(agh_comm **comms, int ndev, const int *devices /* NULL: 0..ndev-1 */);
/* ... or over the caller's own transport (MPI, gloo, a socket -- ranks that RCCL does not connect): every reduction
 * calls fn(ctx, buf, count, elem_bytes) on host memory (elem_bytes 8: sum of uint64 values; 1: maximum of bytes) and
 * expects the reduced values in buf; non-zero = failure.  agh_scan_device_reduce then brings its counts to the host
 * for the exchange (one more synchronisation per step than over RCCL). */
typedef int (*agh_allreduce_fn)(void *ctx, void *buf, size_t count, int elem_bytes);
agh_comm *agh_comm_init_custom(agh_allreduce_fn fn, void *ctx, int nranks, int rank);
int agh_comm_info(const agh_comm *c, int *rank, int *nranks, int *device);
void agh_comm_free(agh_comm *c);

/* counts[0] = n_matched, counts[1] = n_records of this rank's shard -> the sums over all ranks
 * (ncclSum over uint64), identical on every rank: what -c prints for a sharded file. */
int agh_reduce_counts(agh_comm *c, uint64_t counts[2]);
int agh_reduce_counts_all(agh_comm *const *comms, int n, uint64_t (*counts)[2]);

/* One step of a sharded count-only scan: agh_scan_device(flags with AGH_COUNT / AGH_FILENAMEONLY) on this
 * rank's shard + the sum of (n_matched, n_records) over all ranks in totals[0..1].  The counts stay in device
 * memory and the all-reduce is enqueued on the scan's stream behind its kernels: one host synchronisation per
 * step.  Collective: every rank of the communicator calls it (agrep.c:3444-3558 prints one count per file). */
int agh_scan_device_reduce(agh_query *q, agh_comm *c, const void *dev_text, size_t len, void *stream,
                           unsigned flags, agh_result *res, uint64_t totals[2]);

/* hits[f] != 0 iff this rank found a match in file f -> the OR over all ranks (ncclMax over
 * bytes): the -l file list of files that were sharded or dealt out across ranks. */
int agh_reduce_file_hits(agh_comm *c, unsigned char *hits, size_t n_files);
int agh_reduce_file_hits_all(agh_comm *const *comms, int n, unsigned char *const *hits,
                             size_t n_files);

/* Record-aligned shards of a file (SURVEY 8e ownership rule: a record belongs to the range that
 * holds its first byte): cuts[0] = 0 <= cuts[1] <= ... <= cuts[nranks] = file size, inner cut r
 * is the nominal offset r * size / nranks itself if a record starts there (the byte in front of it is
 * the delimiter), else the offset just after the next delimiter at or after it -- the rule of
 * agrep_amd/shard.py:record_cuts, so callers that derive the cuts themselves get the same shards as
 * `agrep-hip --gpus`.  fd must be seekable (pread).  delim/dlen as in agh_query_literal; a delimiter of
 * several bytes must not overlap itself (no proper prefix that is also a suffix: "\r\n", "; ", "$$$" is
 * refused with errno 123) -- only then does the leftmost non-overlapping reading of the delimiters not
 * depend on where the search starts.  The search is on raw bytes: under -i a delimiter of several bytes that
 * holds letters ends records on its case variants too, which this function does not see -- do not shard such
 * inputs (agrep-hip --gpus refuses them). */
int agh_shard_cuts_fd(int fd, const unsigned char *delim, int dlen, int nranks, uint64_t *cuts);

/* agh_scan_fd restricted to the byte range [begin, end) of a seekable file -- one rank's shard. */
int agh_scan_fd_range(agh_query *q, int fd, uint64_t begin, uint64_t end, unsigned flags,
                      agh_result *res, agh_match *matches, size_t cap);

/* ---- device / bench support --------------------------------------------------------- */

/* Number of usable HIP devices (0: none -> every scan fails). */
int agh_device_count(void);
/* Select the device for this thread's subsequent calls (one process per GPU normally). */
int agh_set_device(int ordinal);

/* Synthetic corpus of SURVEY.md 8d, generated directly in HBM (bench/test support; the
 * CPU twin is oracle/corpus_gen.c).  Fills n_pages*4096 bytes at dev_out with pages
 * first_page.. ; variants: n_variants strings of vlen[i] <= 80 bytes laid out at
 * variants + 80*i; planted[8] (host, optional) receives planted-record counts. */
int agh_corpus_fill_device(void *dev_out, uint64_t first_page, uint64_t n_pages,
                           uint64_t seed, const unsigned char *variants,
                           const uint32_t *vlen, uint32_t n_variants, uint32_t plant_period,
                           uint32_t upper_permille, uint64_t *planted, void *stream);

/* Streaming-read ceiling probe: reads len bytes with the same 16 B/lane access pattern as
 * the sweep kernel and no arithmetic beyond a checksum; returns the kernel time in ms. */
int agh_probe_read_ms(const void *dev_text, size_t len, void *stream, double *ms);

const char *agh_last_error(void);
const char *agh_version(void);

#ifdef __cplusplus
}
#endif
#endif
