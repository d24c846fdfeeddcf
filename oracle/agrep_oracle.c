/*
 * agrep_oracle.c -- TEST INFRASTRUCTURE ONLY (see agrep_oracle.h).
 *
 * Scalar CPU restatement of the reference hot path.  Every function cites the reference
 * file:line whose behaviour it follows.  Nothing here is copied from the reference: the
 * algorithms are re-expressed (single-step loops instead of the 2x-unrolled A/B register
 * ping-pong, explicit record bookkeeping instead of output()), and they are pinned against
 * the reference by tests/test_oracle_golden.py (vectors the compiled reference produced:
 * oracle/gen_golden.py) and tests/test_oracle_fuzz.py (restatement vs DP vs the live reference).
 */
#include "agrep_oracle.h"

#include <stdlib.h>
#include <string.h>

/* agrep.h:66-83 -- internal code of the AND separator that preprocess() puts between the
 * delimiter and the pattern (preproce.c:213). */
#define ORC_ANDPAT 144

static inline uint32_t orc_bit(int i) { return 1u << (ORC_WORD - i); } /* agrep.c:281-282 */

static inline int orc_isupper(int c) { return c >= 'A' && c <= 'Z'; }
static inline int orc_tolower(int c) { return orc_isupper(c) ? c + 32 : c; }

static void put_record(orc_record *recs, size_t cap, int64_t idx, uint64_t s, uint64_t e)
{
    if (recs && (uint64_t)idx < (uint64_t)cap) {
        recs[idx].start = s;
        recs[idx].end = e;
    }
}

/* ------------------------------------------------------------------------------------ */
/* preproce.c:181-228 + maskgen.c:26-269, literal subset                                  */
/* ------------------------------------------------------------------------------------ */
int orc_maskgen_literal(const uint8_t *pat, int m, const uint8_t *delim, int dlen,
                        int nocase, orc_tables *out)
{
    uint8_t pp[ORC_WORD + 64];
    int n = 0, i, j, k, M, base;
    int cls[ORC_WORD + 2];          /* the single byte of each position's class, -1 = empty */
    /* maskgen.c declares Mask/Bit/Init/NO_ERR_MASK with implicit int, so the NO_ERR_MASK
     * arithmetic is signed (arithmetic right shifts) -- maskgen.c:19, 222-223. */
    int32_t no_err = 0;
    uint32_t endposition = 0, init0 = 0, wildmask = 0;
    int D_length_global;            /* preproce.c:221-224: strlen(delim) + 1 */

    if (m <= 0 || dlen <= 0 || dlen > ORC_MAXDELIM || m > ORC_WORD) return -1;
    memset(out, 0, sizeof(*out));

    /* preproce.c:181-213: Pattern := delimiter bytes, ANDPAT, literal pattern */
    for (i = 0; i < dlen; i++) pp[n++] = delim[i];
    pp[n++] = ORC_ANDPAT;
    for (i = 0; i < m; i++) pp[n++] = pat[i];
    D_length_global = dlen + 1;

    /* maskgen.c:52-59: -i lowercases the whole preprocessed pattern (delimiter included) */
    if (nocase)
        for (i = 0; i < n; i++) pp[i] = (uint8_t)orc_tolower(pp[i]);

    /* maskgen.c:68-209: one position per byte; '\n' and ANDPAT are special */
    for (i = 0, j = 1; i < n; i++) {
        uint8_t c = pp[i];
        if (c == ORC_ANDPAT) {                        /* maskgen.c:150-163 */
            cls[j] = -1;
            if (j > D_length_global) out->AND = 1;
            endposition |= orc_bit(j);
            j++;
        } else if (c == '\n') {                       /* maskgen.c:171-175 */
            no_err |= (int32_t)orc_bit(j);
            cls[j] = '\n';
            j++;
        } else {                                      /* maskgen.c:194-199 */
            /* a user delimiter arrives wrapped as "<delim>; " (agrep.c:2287-2309), i.e.
             * between LANGLE/RANGLE, so its positions are no-error positions too */
            if (i < dlen) no_err |= (int32_t)orc_bit(j);
            cls[j] = c;
            j++;
        }
        if (j > ORC_WORD) return -1;                  /* maskgen.c:201-208 "pattern too long" */
    }
    M = j - 1;
    base = ORC_WORD - M;

    /* maskgen.c:218-234 */
    wildmask >>= base;
    endposition >>= base;
    no_err = (no_err >> 1) & (int32_t)~orc_bit(1);
    no_err = (~no_err) >> (base - 1);
    for (i = 1; i <= ORC_WORD - M; i++) init0 |= orc_bit(i);
    init0 |= endposition;
    endposition = (endposition << 1) + 1;
    out->Init0 = init0;
    out->Init1 = init0 | wildmask | endposition;
    out->D_endpos = (endposition >> (M - D_length_global)) << (M - D_length_global);
    out->endposition = endposition ^ out->D_endpos;
    out->NO_ERR_MASK = (uint32_t)no_err;
    out->wildmask = wildmask;

    /* maskgen.c:239-257: Mask[c] |= Bit[base + k] iff c is in the class of position k */
    for (k = 1; k <= M; k++)
        if (cls[k] >= 0) out->Mask[cls[k]] |= orc_bit(base + k);

    /* maskgen.c:259-266: -i aliases the upper-case rows onto the lower-case ones */
    if (nocase)
        for (i = 0; i < 256; i++)
            if (orc_isupper(i)) out->Mask[i] = out->Mask[orc_tolower(i)];

    out->M = M;
    out->D_length = dlen;
    return M;
}

/* ------------------------------------------------------------------------------------ */
/* asearch.c:94-199 (k <= 4 unrolled) == asearch.c:620-707 (generic) ; k = 0: bitap.c     */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    const orc_tables *t;
    int k;
    int ci, cs, cd;        /* costs of insertion / substitution / deletion (asearch1.c) */
    uint32_t B[ORC_MAXERR + 1];
    uint32_t D_Mask;
    uint64_t rec_start;
    int64_t count;
    orc_record *recs;
    size_t cap;
    uint64_t n;
} as_state;

/* one text byte c at (possibly virtual) offset i; returns nothing, updates st */
static void as_feed(as_state *st, uint32_t c, int64_t i)
{
    const orc_tables *t = st->t;
    const int k = st->k;
    uint32_t A[ORC_MAXERR + 1];
    uint32_t CM = t->Mask[c & 255];
    int e;

    /* asearch.c:94-116; with costs asearch1.c:88-97 (levels below cost 0 are zero words) */
#define LV(arr, idx) ((idx) >= 0 ? (arr)[idx] : 0u)
    A[0] = ((st->B[0] >> 1) & CM) | (t->Init1 & st->B[0]);
    for (e = 1; e <= k; e++)
        A[e] = ((st->B[e] >> 1) & CM) | (t->Init1 & st->B[e]) | LV(st->B, e - st->ci) |
               (((LV(A, e - st->cd) | LV(st->B, e - st->cs)) >> 1) & t->NO_ERR_MASK);

    if (A[0] & t->D_endpos) {                          /* asearch.c:119 record boundary */
        uint32_t r1 = A[k];
        int hit = t->AND ? ((r1 & t->endposition) == t->endposition)
                         : ((r1 & t->endposition) != 0);      /* asearch.c:128 */
        int64_t rec_end = i + 1 - t->D_length;         /* first byte of the delimiter */
        if (hit) {
            uint64_t s = st->rec_start;
            uint64_t en = rec_end < 0 ? 0 : (uint64_t)rec_end;
            if (en > st->n) en = st->n;
            if (en < s) en = s;
            put_record(st->recs, st->cap, st->count, s, en);
            st->count++;
        }
        st->rec_start = (uint64_t)(i + 1);
        if (st->rec_start > st->n) st->rec_start = st->n;
        /* asearch.c:175-186: reset every level to Init[0], re-consume the same byte,
         * level 0 additionally masked so the delimiter is not re-detected */
        for (e = 0; e <= k; e++) st->B[e] = t->Init0;
        A[0] = (((st->B[0] >> 1) & CM) | (st->B[0] & t->Init1)) & st->D_Mask;
        for (e = 1; e <= k; e++)
            A[e] = ((st->B[e] >> 1) & CM) | (t->Init1 & st->B[e]) | LV(st->B, e - st->ci) |
                   (((LV(A, e - st->cd) | LV(st->B, e - st->cs)) >> 1) & t->NO_ERR_MASK);
    }
#undef LV
    for (e = 0; e <= k; e++) st->B[e] = A[e];
}

int64_t orc_asearch(const orc_tables *t, int k, const uint8_t *text, size_t n,
                    const uint8_t *delim, int dlen, orc_record *recs, size_t cap)
{
    return orc_asearch_costs(t, k, 1, 1, 1, text, n, delim, dlen, recs, cap);
}

int64_t orc_asearch_costs(const orc_tables *t, int k, int ci, int cs, int cd,
                          const uint8_t *text, size_t n, const uint8_t *delim, int dlen,
                          orc_record *recs, size_t cap)
{
    as_state st;
    size_t i;
    int e;

    if (k < 0 || k > ORC_MAXERR || dlen != t->D_length) return -1;
    if (ci < 1 || cs < 1 || cd < 1) return -1;
    memset(&st, 0, sizeof(st));
    st.t = t;
    st.k = k;
    st.ci = ci > k ? k + 1 : ci;                       /* asearch1.c:42-44 */
    st.cs = cs > k ? k + 1 : cs;
    st.cd = cd > k ? k + 1 : cd;
    st.recs = recs;
    st.cap = cap;
    st.n = n;
    /* asearch.c:54-57 */
    st.D_Mask = t->D_endpos;
    for (e = 1; e < dlen; e++) st.D_Mask = (st.D_Mask << 1) | st.D_Mask;
    st.D_Mask = ~st.D_Mask;
    for (e = 0; e <= k; e++) st.B[e] = t->Init0;       /* asearch.c:63-64 */

    as_feed(&st, '\n', -1);                            /* asearch.c:69-78: buffer[Max_record-1]='\n' */
    for (i = 0; i < n; i++) as_feed(&st, text[i], (int64_t)i);
    for (e = 0; e < dlen; e++)                         /* asearch.c:87-91: delimiter appended at EOF */
        as_feed(&st, delim[e], (int64_t)n + e);
    return st.count;
}

/* ------------------------------------------------------------------------------------ */
/* sgrep.c:1023-1051 initmask + sgrep.c:1166-1239 verify loop                             */
/* ------------------------------------------------------------------------------------ */
int64_t orc_sgrep_verify(const uint8_t *pat, int m, int k, const uint8_t *text, size_t n,
                         int clean, orc_record *recs, size_t cap)
{
    uint32_t Mask[256], R[ORC_MAXERR + 1], N[ORC_MAXERR + 1];
    const uint32_t Bit1 = 0x80000000u;
    uint32_t endpos;
    int64_t count = 0;
    size_t i;
    int e, j;

    if (m <= 0 || m > 32 || k < 0 || k > ORC_MAXERR) return -1;
    /* sgrep.c:1034-1050 (called with D = 0: one end bit, inverted masks) */
    endpos = Bit1 >> (m - 1);
    for (j = 0; j < 256; j++) Mask[j] = ~0u;
    for (j = 0; j < m; j++) Mask[pat[j]] &= ~(Bit1 >> j);

    /* sgrep.c:1172-1174: level e starts with e leading deletions granted */
    R[0] = ~0u;
    for (e = 1; e <= k; e++) R[e] = (R[e - 1] >> 1) & R[e - 1];

    i = 0;
    while (i < n) {
        uint32_t c = text[i++];
        uint32_t r1;
        if (c == '\n')                                 /* sgrep.c:1179-1181 */
            for (e = 0; e <= k; e++) R[e] = ~0u;
        r1 = Mask[c];
        N[0] = (R[0] >> 1) | r1;                       /* sgrep.c:1183-1185 */
        for (e = 1; e <= k; e++)
            N[e] = ((R[e] >> 1) | r1) & R[e - 1] & ((N[e - 1] & R[e - 1]) >> 1);
        for (e = 0; e <= k; e++) R[e] = N[e];
        if ((R[k] & endpos) == 0) {                    /* sgrep.c:1186 */
            /* s_output (sgrep.c:1303-1306, 1327-1329): record bounds around the hit,
             * scan index jumps just past the record's newline */
            size_t b = i, en = i;
            while (b > 0 && text[b - 1] != '\n') b--;
            while (en < n && text[en] != '\n') en++;
            put_record(recs, cap, count, b, en);
            count++;
            i = en < n ? en + 1 : en;
            for (e = 0; e <= k; e++) R[e] = ~0u;       /* sgrep.c:1201 */
            if (clean) {
                /* what a quirk-free restart looks like (Q2): the consumed newline would have
                 * left level e with e leading deletions */
                for (e = 1; e <= k; e++) R[e] = (R[e - 1] >> 1) & R[e - 1];
            }
        }
    }
    return count;
}

/* ------------------------------------------------------------------------------------ */
/* Sellers DP ground truth                                                */
/* ------------------------------------------------------------------------------------ */
int orc_dp_best(const uint8_t *pat, int m, int nocase, const uint8_t *rec, size_t len)
{
    int col[260], best, i;
    size_t j;
    if (m > 256) return -1;
    for (i = 0; i <= m; i++) col[i] = i;               /* empty text prefix: i deletions */
    best = col[m];
    for (j = 0; j < len; j++) {
        int c = nocase ? orc_tolower(rec[j]) : rec[j];
        int diag = col[0];                             /* col[0] stays 0: free start */
        for (i = 1; i <= m; i++) {
            int p = nocase ? orc_tolower(pat[i - 1]) : pat[i - 1];
            int sub = diag + (p != c);
            int ins = col[i] + 1;                      /* extra text byte */
            int del = col[i - 1] + 1;                  /* pattern byte skipped */
            int v = sub < ins ? sub : ins;
            diag = col[i];
            col[i] = v < del ? v : del;
        }
        if (col[m] < best) best = col[m];
    }
    return best;
}

/* Delimiter test on the text as file mode sees it: with the delimiter appended at EOF
 * (asearch.c:87-91), so a record may be closed by a delimiter that is only partly real. */
static int delim_at(const uint8_t *text, size_t n, size_t i, const uint8_t *delim, int dlen,
                    int nocase)
{
    int e;
    for (e = 0; e < dlen; e++) {
        size_t at = i + (size_t)e;
        int a = at < n ? text[at] : delim[at - n], b = delim[e];
        if (at >= n + (size_t)dlen) return 0;
        if (nocase) { a = orc_tolower(a); b = orc_tolower(b); } /* maskgen.c:52-59 folds it too */
        if (a != b) return 0;
    }
    return 1;
}

typedef int (*rec_pred)(void *ctx, const uint8_t *rec, size_t len);

static int64_t for_each_record(const uint8_t *text, size_t n, const uint8_t *delim, int dlen,
                               int nocase, rec_pred pred, void *ctx, orc_record *recs,
                               size_t cap)
{
    int64_t count = 0;
    size_t start = 0, i = 0;
    while (i < n) {
        if (delim_at(text, n, i, delim, dlen, nocase)) {
            if (pred(ctx, text + start, i - start)) {
                put_record(recs, cap, count, start, i);
                count++;
            }
            i += (size_t)dlen;
            start = i;
        } else {
            i++;
        }
    }
    if (start < n && pred(ctx, text + start, n - start)) { /* unterminated last record */
        put_record(recs, cap, count, start, n);
        count++;
    }
    return count;
}

typedef struct { const uint8_t *pat; int m, k, nocase; } dp_ctx;

static int dp_pred(void *vctx, const uint8_t *rec, size_t len)
{
    dp_ctx *c = (dp_ctx *)vctx;
    return orc_dp_best(c->pat, c->m, c->nocase, rec, len) <= c->k;
}

int64_t orc_dp_count(const uint8_t *pat, int m, int k, int nocase, const uint8_t *text,
                     size_t n, const uint8_t *delim, int dlen, orc_record *recs, size_t cap)
{
    dp_ctx c;
    if (m <= 0 || m > 256 || dlen <= 0) return -1;
    c.pat = pat; c.m = m; c.k = k; c.nocase = nocase;
    return for_each_record(text, n, delim, dlen, nocase, dp_pred, &c, recs, cap);
}

/* ------------------------------------------------------------------------------------ */
/* Multi-word Wu-Manber extension, shift-AND, 1 = active.                 */
/* Position p (1..m) lives at bit (p-1) of a little-endian multi-word integer; a step is  */
/*   R0' = ((R0 << 1) | 1) & M[c]                                                         */
/*   Re' = (((Re << 1) | 1) & M[c]) | R(e-1) | (((R(e-1) | R(e-1)') << 1) | 1)            */
/* which is asearch.c:94-116 mirrored (left shifts, no in-word delimiter/sticky bits).    */
/* ------------------------------------------------------------------------------------ */
#define WM_MAXW 32  /* 256 positions / 8-bit words */

typedef struct {
    int nw, wbits, m, k;
    uint64_t wmask;
    uint64_t M[256][WM_MAXW];
} wm_query;

static void wm_shl1_or1(const wm_query *q, const uint64_t *x, uint64_t *out)
{
    uint64_t carry = 1;                                /* the always-active start state */
    int w;
    for (w = 0; w < q->nw; w++) {
        uint64_t v = x[w];
        out[w] = ((v << 1) | carry) & q->wmask;
        carry = (v >> (q->wbits - 1)) & 1;             /* explicit inter-word carry */
    }
}

static int wm_pred(void *vctx, const uint8_t *rec, size_t len)
{
    const wm_query *q = (const wm_query *)vctx;
    uint64_t R[ORC_MAXERR + 1][WM_MAXW], N[ORC_MAXERR + 1][WM_MAXW];
    uint64_t t1[WM_MAXW], t2[WM_MAXW];
    const int fw = (q->m - 1) / q->wbits, fb = (q->m - 1) % q->wbits;
    int e, w;
    size_t j;

    /* record start: level e has its first e positions active (e leading deletions) */
    for (e = 0; e <= q->k; e++)
        for (w = 0; w < q->nw; w++) {
            int lo = w * q->wbits;
            int bits = e - lo;
            if (bits <= 0) R[e][w] = 0;
            else if (bits >= q->wbits) R[e][w] = q->wmask;
            else R[e][w] = ((uint64_t)1 << bits) - 1;
        }
    if (q->k >= q->m) return 1;
    for (j = 0; j < len; j++) {
        const uint64_t *CM = q->M[rec[j]];
        wm_shl1_or1(q, R[0], t1);
        for (w = 0; w < q->nw; w++) N[0][w] = t1[w] & CM[w];
        for (e = 1; e <= q->k; e++) {
            wm_shl1_or1(q, R[e], t1);
            for (w = 0; w < q->nw; w++) t2[w] = R[e - 1][w] | N[e - 1][w];
            wm_shl1_or1(q, t2, t2);
            for (w = 0; w < q->nw; w++) N[e][w] = (t1[w] & CM[w]) | R[e - 1][w] | t2[w];
        }
        memcpy(R, N, sizeof(R));
        if ((R[q->k][fw] >> fb) & 1) return 1;
    }
    return 0;
}

int64_t orc_wm_count(const uint8_t *pat, int m, int k, int nocase, int word_bits,
                     const uint8_t *text, size_t n, const uint8_t *delim, int dlen,
                     orc_record *recs, size_t cap)
{
    wm_query *q;
    int64_t r;
    int p, c;
    if (m <= 0 || m > 256 || k < 0 || k > ORC_MAXERR) return -1;
    if (word_bits != 8 && word_bits != 16 && word_bits != 32 && word_bits != 64) return -1;
    q = (wm_query *)calloc(1, sizeof(*q));
    if (!q) return -1;
    q->wbits = word_bits;
    q->nw = (m + word_bits - 1) / word_bits;
    q->m = m;
    q->k = k;
    q->wmask = word_bits == 64 ? ~(uint64_t)0 : (((uint64_t)1 << word_bits) - 1);
    for (p = 0; p < m; p++) {
        int pc = nocase ? orc_tolower(pat[p]) : pat[p];
        for (c = 0; c < 256; c++) {
            int tc = nocase ? orc_tolower(c) : c;
            if (tc == pc) q->M[c][p / word_bits] |= (uint64_t)1 << (p % word_bits);
        }
    }
    r = for_each_record(text, n, delim, dlen, nocase, wm_pred, q, recs, cap);
    free(q);
    return r;
}

/* ------------------------------------------------------------------------------------ */
/* -f multi-pattern exact ground truth (newmgrep.c:839-1012 semantics, k = 0)             */
/* ------------------------------------------------------------------------------------ */
typedef struct { const uint8_t *const *pats; const int *lens; int npat, nocase; } mp_ctx;

static int mp_pred(void *vctx, const uint8_t *rec, size_t len)
{
    mp_ctx *c = (mp_ctx *)vctx;
    int p;
    for (p = 0; p < c->npat; p++) {
        size_t L = (size_t)c->lens[p], i, j;
        if (L == 0 || L > len) continue;
        for (i = 0; i + L <= len; i++) {
            for (j = 0; j < L; j++) {
                int a = rec[i + j], b = c->pats[p][j];
                if (c->nocase) { a = orc_tolower(a); b = orc_tolower(b); }
                if (a != b) break;
            }
            if (j == L) return 1;
        }
    }
    return 0;
}

int64_t orc_multi_exact_count(const uint8_t *const *pats, const int *lens, int npat,
                              int nocase, const uint8_t *text, size_t n,
                              const uint8_t *delim, int dlen, orc_record *recs, size_t cap)
{
    mp_ctx c;
    c.pats = pats; c.lens = lens; c.npat = npat; c.nocase = nocase;
    return for_each_record(text, n, delim, dlen, nocase, mp_pred, &c, recs, cap);
}
