/*
 * agrep_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's bit-parallel k-error record scan
 * (Wikinaut/agrep: maskgen.c, asearch.c, sgrep.c:agrep(), preproce.c literal path) plus a
 * Sellers dynamic-programming ground truth.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (agrep_amd/) never does.
 *
 * Parity status: PINNED for m <= 29 / k <= 8 / literal patterns against (a) the golden
 * vectors in tests/golden/ that were produced by running the reference itself
 * (oracle/_ref/ref_harness, see oracle/gen_golden.py) and (b) the live reference binary
 * when oracle/_ref exists.  The multi-word extension (m > 29, orc_wm_count) is
 * "parity unpinned" against the reference (the reference rejects such patterns,
 * maskgen.c:201-208); it is pinned against the DP and against the one-word automaton by
 * running the same code with artificially narrow words.
 */
#ifndef AGREP_ORACLE_H
#define AGREP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_WORD 32        /* agrep.h:43  WORD */
#define ORC_MAXERR 8       /* agrep.h:44  MaxError */
#define ORC_MAXDELIM 8     /* agrep.h:34  MAXDELIM */

/* Reference-layout query tables: exactly the globals maskgen() leaves behind
 * (agrep.c:135-140).  Bit[i] = 1u << (32 - i) (agrep.c:281-282). */
typedef struct {
    uint32_t Mask[256];
    uint32_t Init0;        /* Init[0] */
    uint32_t Init1;
    uint32_t NO_ERR_MASK;
    uint32_t endposition;
    uint32_t D_endpos;
    uint32_t wildmask;
    int32_t  M;            /* positions incl. delimiter + separator (maskgen return value) */
    int32_t  D_length;     /* strlen(old_D_pat), i.e. the raw delimiter length */
    int32_t  AND;
} orc_tables;

/* One matched record: [start, end) excludes the delimiters on both sides. */
typedef struct {
    uint64_t start;
    uint64_t end;
} orc_record;

/* preproce.c:181-228 (literal subset) + maskgen.c:26-269.
 * pat: literal pattern bytes (no meta characters are interpreted), delim: raw delimiter
 * bytes ('\n' for the default).  nocase = -i.  Returns M (= m + dlen + 1) or -1 when the
 * reference would say "pattern too long" (maskgen.c:201-208). */
int orc_maskgen_literal(const uint8_t *pat, int m, const uint8_t *delim, int dlen,
                        int nocase, orc_tables *out);

/* asearch.c:94-199 / 620-707 recurrence, record boundary and reset, k = 0..8
 * (k = 0 is bitap.c:169-284 without the LUT indirection).  The text is scanned the way
 * file mode does it: a virtual '\n' in front (asearch.c:69-78) and the delimiter appended
 * at EOF when the text does not end with it (asearch.c:87-91).  Returns the number of
 * matched records; up to cap of them are stored in recs (may be NULL). */
int64_t orc_asearch(const orc_tables *t, int k, const uint8_t *text, size_t n,
                    const uint8_t *delim, int dlen, orc_record *recs, size_t cap);

/* asearch1.c:28-435: non-unit costs (-I# -S# -D#).  State words are indexed by accumulated
 * cost 0..D (the reference keeps them at A[D..2D] above D dummy zero words, asearch1.c:55-56);
 * insertion reads level cost-I, substitution cost-S, deletion the NEW level cost-DD
 * (asearch1.c:90-97); costs above D are clamped to D+1 (asearch1.c:42-44).  Same text
 * framing as orc_asearch.  I = S = DD = 1 is orc_asearch. */
int64_t orc_asearch_costs(const orc_tables *t, int k, int ci, int cs, int cd,
                          const uint8_t *text, size_t n, const uint8_t *delim, int dlen,
                          orc_record *recs, size_t cap);

/* sgrep.c:1023-1051 (initmask) + the verify loop sgrep.c:1166-1239 run over the whole text
 * (the BM/hash candidate filter sgrep.c:1130-1154 is lossless, so its windows are replaced
 * by "everything").  Newline is the hard-wired reset character (sgrep.c:1179-1181).
 * clean = 0 follows the reference literally (quirk Q2: no leading-deletion state after a
 * matched record); clean = 1 re-arms the initial condition.  m <= 32. */
int64_t orc_sgrep_verify(const uint8_t *pat, int m, int k, const uint8_t *text, size_t n,
                         int clean, orc_record *recs, size_t cap);

/* Ground truth: a record matches iff some substring of it is within
 * Levenshtein distance <= k of the pattern (unit costs).  Records are split at the
 * leftmost non-overlapping occurrences of delim; an unterminated last record counts. */
int64_t orc_dp_count(const uint8_t *pat, int m, int k, int nocase, const uint8_t *text,
                     size_t n, const uint8_t *delim, int dlen, orc_record *recs, size_t cap);

/* min over substrings s of rec of Levenshtein(s, pat) (Sellers). */
int orc_dp_best(const uint8_t *pat, int m, int nocase, const uint8_t *rec, size_t len);

/* Multi-word Wu-Manber extension: delimiter handled out of band, pattern
 * of m <= 256 positions held in ceil(m / word_bits) words of word_bits in {8,16,32,64}
 * bits with explicit carry between words.  Same verdict contract as orc_dp_count. */
int64_t orc_wm_count(const uint8_t *pat, int m, int k, int nocase, int word_bits,
                     const uint8_t *text, size_t n, const uint8_t *delim, int dlen,
                     orc_record *recs, size_t cap);

/* Exact multi-pattern ground truth for the -f path (newmgrep.c semantics, k = 0):
 * a record matches iff it contains any of the npat patterns verbatim. */
int64_t orc_multi_exact_count(const uint8_t *const *pats, const int *lens, int npat,
                              int nocase, const uint8_t *text, size_t n,
                              const uint8_t *delim, int dlen, orc_record *recs, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
