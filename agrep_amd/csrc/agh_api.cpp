// agh_api.cpp -- the C-ABI of libagrep_hip.so (include/agrep_hip.h): query compilation, workspace
// management and kernel orchestration for text resident in HBM (staging, files and record output:
// agh_stage.cpp).  No CPU scan path exists here: if HIP is unusable every entry point fails with
// -1 / errno = 123.
#include "agh_internal.h"

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int agh_fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    errno = AGH_ERRNO;
    return -1;
}

extern "C" const char *agh_last_error(void) { return g_err; }
// (internal: agh_comm.cpp reports through the same text; not part of the ABI, hence hidden)
extern "C" __attribute__((visibility("hidden"))) void agh_set_error(const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
}
extern "C" const char *agh_version(void) { return "agrep-hip 0.1 (gfx950)"; }

extern "C" int agh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int agh_set_device(int ordinal)
{
    HIP_TRY(hipSetDevice(ordinal));
    return 0;
}


static bool is_upper(int c) { return c >= 'A' && c <= 'Z'; }
static bool is_lower(int c) { return c >= 'a' && c <= 'z'; }

// Bytes a pattern position accepts (its class, maskgen.c:86-135; one byte for a literal, a
// case pair under -i).
static std::vector<uint8_t> position_members(const agh_query *q, int p)
{
    std::vector<uint8_t> v;
    for (int c = 0; c < 256; ++c)
        if ((q->mask[c] >> p) & 1) v.push_back((uint8_t)c);
    return v;
}

static bool is_case_pair(const std::vector<uint8_t> &v)
{
    return v.size() == 2 && is_upper(v[0]) && v[1] == v[0] + 32;
}

// Distinct bytes of a position as the sample filter sees them: with folding (OR 0x20 on both
// sides) several members collapse into one.
static std::vector<uint8_t> folded_members(const std::vector<uint8_t> &v, bool fold)
{
    bool seen[256] = {false};
    std::vector<uint8_t> out;
    for (uint8_t c : v) {
        const uint8_t f = fold ? (uint8_t)(c | 0x20u) : c;
        if (!seen[f]) { seen[f] = true; out.push_back(f); }
    }
    return out;
}

#ifndef AGH_SHAPE_H2_DEFAULT
// 64 GiB, k = 2, fused: 10.61 -> 10.30 ms, candidates 17.5 M -> 9.9 M; 8 GiB 1.413 -> 1.392 ms
// (profiles/r03_ab_headline.log)
#define AGH_SHAPE_H2_DEFAULT 1
#endif
#define AGH_CLASS_MAX 40        // largest class a sampled position may have ([a-z], [0-9a-z] ...)
#define AGH_GRAMS_MAX 2048      // expanded q-grams of one query (32 Ki table slots: <= 6 % full)

// Every concrete q-gram of the pattern window [i, i + fq): the cartesian product of the
// positions' (folded) members.  fn(sample) is called once per gram.
template <typename F>
static void for_each_gram(const agh_query *q, int i, bool fold, F fn)
{
    std::vector<uint8_t> mem[4];
    for (int t = 0; t < q->fq; ++t) mem[t] = folded_members(position_members(q, i + t), fold);
    size_t idx[4] = {0, 0, 0, 0};
    for (;;) {
        uint32_t s = 0;
        for (int t = 0; t < q->fq; ++t) s |= (uint32_t)mem[t][idx[t]] << (8 * t);
        fn((s & q->qmask) | q->fold);
        int t = 0;
        while (t < q->fq && ++idx[t] == mem[t].size()) idx[t++] = 0;
        if (t == q->fq) break;
    }
}

// Choose the q-gram sample shape: samples of q bytes at every multiple of h bytes.  Lossless
// iff an occurrence (>= m-k text bytes) always contains >= k+1 whole samples, because k errors
// can spoil at most k disjoint samples:  floor((m - k - q + 1) / h) >= k + 1.
static void choose_filter(agh_query *q)
{
    q->fq = q->fh = 0;
    q->qmask = q->fold = 0;
    q->run_a = 0;
    q->run_len = 0;
    // The samples come from a run of positions with SMALL classes: a single byte, an ASCII case
    // pair, or a class of at most AGH_CLASS_MAX bytes ([xyz], [0-9], [a-z]) whose q-grams are
    // enumerated (a sample that lies inside an error-free stretch of an occurrence equals one
    // concrete choice of the classes' members).  For -w / -x / <exact> / wide-class patterns the
    // run is a core of the pattern: an occurrence of the pattern with <= k errors contains an
    // occurrence of the core with <= k errors, so the lemma applies with the core's length.
    // Among all runs the one that allows the cheapest sample shape wins, ties by fewer grams.
    std::vector<size_t> width((size_t)q->m);
    bool any_pair = false;
    for (int p = 0; p < q->m; ++p) {
        const std::vector<uint8_t> v = position_members(q, p);
        if (is_case_pair(v)) any_pair = true;
    }
    for (int p = 0; p < q->m; ++p) {
        const std::vector<uint8_t> raw = position_members(q, p);
        const std::vector<uint8_t> v = folded_members(raw, any_pair);
        width[(size_t)p] = v.size();
        // A position that accepts '\n' or a delimiter byte (-x / -w guards, patterns that hold the
        // record separator) can be matched by a byte that is NOT in the scanned text: the virtual
        // '\n' in front of a file, the delimiter appended at its end (asearch.c:69-91), the last byte
        // of the previous segment.  Samples are taken from real text only, so such positions end a
        // run: the sampled core of an occurrence then lies inside its record.
        for (uint8_t c : raw) {
            bool virt = c == '\n';
            for (int j = 0; j < q->dlen; ++j) virt = virt || c == q->delim[j] || (q->delim_fold && (c | 0x20) == q->delim[j]);
            if (virt) { width[(size_t)p] = 0; break; }
        }
    }
    static const int hs[3] = {16, 8, 4};
    int best_h = 0, best_q = 0, best_a = 0, best_len = 0;
    double best_grams = 0;
    for (int a = 0; a < q->m; ++a) {
        for (int len = 3; a + len <= q->m; ++len) {
            if (width[(size_t)(a + len - 1)] == 0 || width[(size_t)(a + len - 1)] > AGH_CLASS_MAX) break;
            bool ok = true;
            for (int p = a; p < a + len && ok; ++p) ok = width[(size_t)p] >= 1 && width[(size_t)p] <= AGH_CLASS_MAX;
            if (!ok) break;
            for (int i = 0; i < 3; ++i) {
                const int h = hs[i];
                int qmax = len - q->k + 1 - h * (q->k + 1);
                if (qmax > 4) qmax = 4;
                if (qmax > h) qmax = h;
                if (qmax < 3) continue;
                double grams = 0;
                for (int g = a; g + qmax <= a + len; ++g) {
                    double prod = 1;
                    for (int t = 0; t < qmax; ++t) prod *= (double)width[(size_t)(g + t)];
                    grams += prod;
                }
                if (grams > AGH_GRAMS_MAX) continue;
                // larger stride first (fewer probes per 16 bytes), then longer samples, then fewer grams
                const bool better = h > best_h || (h == best_h && qmax > best_q) ||
                                    (h == best_h && qmax == best_q && grams < best_grams);
                if (better) { best_h = h; best_q = qmax; best_a = a; best_len = len; best_grams = grams; }
                break;                          // smaller strides of the same run are never better
            }
        }
    }
    // H = 2: 4-byte samples at every even offset.  They overlap, so one error can spoil two of
    // them: lossless iff floor((len - k - 4 + 1) / 2) >= 2k + 1.  (m, k) = (16, 2) gets q = 4 this way
    // instead of q = 3 every 4 bytes: twice the probes, but the chance occurrences of the pattern's
    // grams in the text -- two thirds of all candidates on the bench corpus -- become 27 x rarer.
    // AGH_SHAPE_H2: 0 never, 1 (default) where the best other shape samples 3 bytes or none applies.
    {
        const char *e = getenv("AGH_SHAPE_H2");
        const int h2_mode = e && *e ? atoi(e) : AGH_SHAPE_H2_DEFAULT;
        if (h2_mode > 0 && best_q < 4) {
            int h2_a = -1, h2_len = 0;
            double h2_grams = 0;
            for (int a = 0; a < q->m; ++a)
                for (int len = 4; a + len <= q->m; ++len) {
                    bool ok = true;
                    for (int p = a; p < a + len && ok; ++p) ok = width[(size_t)p] >= 1 && width[(size_t)p] <= AGH_CLASS_MAX;
                    if (!ok) break;
                    if ((len - q->k - 3) / 2 < 2 * q->k + 1) continue;
                    double grams = 0;
                    for (int g = a; g + 4 <= a + len; ++g) {
                        double prod = 1;
                        for (int t = 0; t < 4; ++t) prod *= (double)width[(size_t)(g + t)];
                        grams += prod;
                    }
                    if (grams > AGH_GRAMS_MAX) continue;
                    if (h2_a < 0 || grams < h2_grams) { h2_a = a; h2_len = len; h2_grams = grams; }
                    break;                      // longer runs from the same start only add grams
                }
            if (h2_a >= 0) { best_h = 2; best_q = 4; best_a = h2_a; best_len = h2_len; }
        }
    }
    if (!best_h) return;
    q->fq = best_q;
    q->fh = best_h;
    q->run_a = best_a;
    q->run_len = best_len;
    q->qmask = q->fq == 4 ? 0xffffffffu : ((1u << (8 * q->fq)) - 1u);
    q->fold = any_pair ? (0x20202020u & q->qmask) : 0u;
}

static int upload_common(agh_query *q)
{
    HIP_TRY(hipMalloc((void **)&q->d_counters, (AGH_LEAN_SLOTS + 1) * AGH_C_COUNT * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&q->d_chunk_totals, 128 * sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc((void **)&q->h_counters, (AGH_MAX_SEGS + 1) * AGH_C_COUNT * sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc((void **)&q->h_cuts, 3 * AGH_MAX_SEGS * sizeof(uint64_t)));
    HIP_TRY(hipEventCreate(&q->ev0));
    HIP_TRY(hipEventCreate(&q->ev1));
    HIP_TRY(hipEventCreate(&q->ev2));
    HIP_TRY(hipEventCreate(&q->ev3));
    return 0;
}

static int upload_tables(agh_query *q)
{
    if (q->wide) {
        HIP_TRY(hipMalloc(&q->d_mask, 256 * sizeof(uint64_t)));
        HIP_TRY(hipMemcpy(q->d_mask, q->mask, 256 * sizeof(uint64_t), hipMemcpyHostToDevice));
    } else {
        uint32_t m32[256];
        for (int c = 0; c < 256; ++c) m32[c] = (uint32_t)q->mask[c];
        HIP_TRY(hipMalloc(&q->d_mask, sizeof(m32)));
        HIP_TRY(hipMemcpy(q->d_mask, m32, sizeof(m32), hipMemcpyHostToDevice));
    }
    if (q->fq) {
        std::vector<uint8_t> tab(AGH_FT_SIZE, 0);
        const bool fold = q->fold != 0;
        const int run_end = q->run_a + q->run_len;          // grams of the sampled run only
        // Per hash slot: which gram sits there and where in the pattern it occurs -- the lean
        // verifier drops hash false positives before running the automaton and, knowing the
        // gram's offset o, walks [j-o-k, j-o+m+k) instead of the offset-blind window.
        //   bits 0..31 gram, 32..39 first offset, 40..47 last offset, bit 63: ambiguous
        //   (two different grams share the slot, or offsets too far apart) -> full window
        std::vector<uint64_t> gt(AGH_FT_SIZE, AGH_GT_AMBIGUOUS);
        std::vector<char> used(AGH_FT_SIZE, 0);
        for (int i = q->run_a; i + q->fq <= run_end; ++i)
            for_each_gram(q, i, fold, [&](uint32_t s) {
                const uint32_t pr = q->fq == 4 ? agh_sample_prod_q4(s) : agh_sample_prod_q3(s);
                const uint32_t h = q->fq == 4 ? AGH_Q4_SLOT(pr) : AGH_Q3_SLOT(pr);
                tab[h] |= (uint8_t)(1u << (q->fq == 4 ? AGH_Q4_BIT(pr) : AGH_Q3_BIT(pr)));   // 8 bits per slot
                if (!used[h]) {
                    used[h] = 1;
                    gt[h] = (uint64_t)s | ((uint64_t)i << 32) | ((uint64_t)i << 40);
                } else if (!(gt[h] & AGH_GT_AMBIGUOUS) && (uint32_t)gt[h] == s) {
                    if ((uint32_t)i > (uint32_t)((gt[h] >> 40) & 0xff))
                        gt[h] = (gt[h] & ~((uint64_t)0xff << 40)) | ((uint64_t)i << 40);   // last offset
                } else {
                    gt[h] = AGH_GT_AMBIGUOUS;
                }
            });
        HIP_TRY(hipMalloc((void **)&q->d_ftab, AGH_FT_SIZE));
        HIP_TRY(hipMemcpy(q->d_ftab, tab.data(), AGH_FT_SIZE, hipMemcpyHostToDevice));
        q->gram_spread = 0;
        for (uint32_t h = 0; h < AGH_FT_SIZE; ++h) {
            if (!used[h] || (gt[h] & AGH_GT_AMBIGUOUS)) continue;
            const uint32_t sp = (uint32_t)((gt[h] >> 40) & 0xff) - (uint32_t)((gt[h] >> 32) & 0xff);
            if (sp > 8) gt[h] = AGH_GT_AMBIGUOUS;
            else if (sp > q->gram_spread) q->gram_spread = sp;
        }
        HIP_TRY(hipMalloc((void **)&q->d_gtab, AGH_FT_SIZE * sizeof(uint64_t)));
        HIP_TRY(hipMemcpy(q->d_gtab, gt.data(), AGH_FT_SIZE * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    return upload_common(q);
}

static int fill_multi_tables(agh_query *q, const unsigned char *const *pats, const int *lens,
                             int npat, int D, int nocase, unsigned char delim0, int *fq_out,
                             uint32_t *qmask_out, uint32_t *fold_out, int *minlen_out);

// A literal pattern without a usable sample filter (m < 5k+6: short words, many errors) would
// read every byte through the automaton (~1 TB/s).  The partition lemma still applies: give
// it the multi-pattern engine with its k+1 pieces as entries (agh_multi.hip).
static int attach_piece_engine(agh_query *q)
{
    if (q->fq || q->general || q->table || q_mb(q) || q->m > 32 || q->m <= q->k) return 0;
    // Pieces of 1-2 bytes select next to nothing (a 2-byte piece hits every ~500th position of
    // English-like text: 10-25 M candidates per 4 GiB): the census-free full scan is faster then
    // (scripts/perf_short.py: 'approxim' k=2 1.06 vs 1.54 TB/s, 'match' k=1 1.33 vs 1.98).
    if (q->m / (q->k + 1) < 3) return 0;
    unsigned char pat[32];
    bool any_pair = false, any_single_letter = false;
    for (int p = 0; p < q->m; ++p) {
        int members = 0, lo = -1;
        for (int c = 0; c < 256; ++c)
            if ((q->mask[c] >> p) & 1) { ++members; if (lo < 0) lo = c; }
        if (members == 2 && is_upper(lo) && ((q->mask[lo + 32] >> p) & 1)) {
            any_pair = true;
            pat[p] = (unsigned char)(lo + 32);
        } else if (members == 1) {
            if (is_upper(lo) || is_lower(lo)) any_single_letter = true;
            pat[p] = (unsigned char)lo;
        } else {
            return 0;                               // a class: not a literal
        }
        if (pat[p] == q->delim[0] || pat[p] == '\n') return 0;
    }
    if (any_pair && any_single_letter) return 0;    // neither plain nor -i
    const unsigned char *pp = pat;
    const int len = q->m;
    if (fill_multi_tables(q, &pp, &len, 1, q->k, any_pair ? 1 : 0, q->delim[0], &q->pe_fq,
                          &q->pe_qmask, &q->pe_fold, &q->pe_minlen))
        return -1;
    q->piece_single = true;
    return 0;
}

static agh_query *finish_query(agh_query *q)
{
    q->wide = q->m > 32;
    choose_filter(q);
    if (agh_device_count() <= 0) {
        fail("no usable HIP device: libagrep_hip has no CPU path");
        delete q;
        return nullptr;
    }
    if (upload_tables(q) != 0 || attach_piece_engine(q) != 0) {
        agh_query_free(q);
        return nullptr;
    }
    return q;
}

static bool c_isalnum(int c)       // isalnum() of the C locale, the one bm() and monkey1() see
{
    return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z');
}

extern "C" agh_query *agh_query_literal_ex(const unsigned char *pat, int m, int D, unsigned qflags,
                                           const unsigned char *delim, int dlen)
{
    const int nocase = (qflags & AGH_Q_NOCASE) ? 1 : 0;
    if ((qflags & AGH_Q_WORD) && (qflags & AGH_Q_WHOLELINE)) {          // agrep.c:2188-2196
        fail("illegal option combination (-x and -w)");
        return nullptr;
    }
    // guard positions around the pattern: -x wraps it into '\n' (sgrep.c:252-259), -w into the class
    // of non-alphanumeric bytes (the test bm() makes on the bytes next to an occurrence, sgrep.c:750-756)
    const int guard = (qflags & AGH_Q_WHOLELINE) ? 2 : ((qflags & AGH_Q_WORD) ? 1 : 0);
    const int M = m + (guard ? 2 : 0);
    if (!pat || m < 1 || M > AGH_MAX_PATTERN) {
        fail("pattern length %d outside 1..%d", m, AGH_MAX_PATTERN - (guard ? 2 : 0));
        return nullptr;
    }
    if (D < 0 || D > AGH_MAX_ERRORS || D >= m) {   // checksg.c:34-41
        fail("number of errors %d must be in 0..%d and smaller than the pattern length %d", D,
             AGH_MAX_ERRORS, m);
        return nullptr;
    }
    if (!delim || dlen < 1 || dlen > AGH_MAX_DELIM) {
        fail("delimiter length %d outside 1..%d", dlen, AGH_MAX_DELIM);
        return nullptr;
    }
    bool delim_letters = false;
    for (int i = 0; i < dlen; ++i) delim_letters = delim_letters || is_upper(delim[i]) || is_lower(delim[i]);
    // -i with letters in the delimiter: maskgen.c:259-266 aliases the upper-case rows of Mask[] for
    // the delimiter positions too ("X" ends a record of -d x); such delimiters go through the
    // delimiter bitmap (delim_class folds), whatever their length
    agh_query *q = new agh_query();
    q->m = M;
    q->k = D;
    q->dlen = dlen;
    memcpy(q->delim, delim, (size_t)dlen);
    if (nocase && delim_letters) {
        q->delim_fold = true;
        for (int i = 0; i < dlen; ++i)
            if (is_upper(q->delim[i])) q->delim[i] = (unsigned char)(q->delim[i] + 32);
    }
    memset(q->mask, 0, sizeof(q->mask));
    const int off = guard ? 1 : 0;
    for (int p = 0; p < m; ++p) {
        int c = pat[p];
        if (nocase && is_upper(c)) c += 32;             // maskgen.c:52-59
        q->mask[c] |= (uint64_t)1 << (p + off);
        if (nocase && is_lower(c)) q->mask[c - 32] |= (uint64_t)1 << (p + off);   // maskgen.c:259-266
    }
    if (guard) {
        const uint64_t ends = (uint64_t)1 | ((uint64_t)1 << (M - 1));
        for (int c = 0; c < 256; ++c)
            if (guard == 2 ? c == '\n' : !c_isalnum(c)) q->mask[c] |= ends;
        if (D > 0) {                                    // no error may touch a guard (maskgen.c:171-187)
            q->general = true;
            q->no_err = (M == 64 ? ~0ull : (((uint64_t)1 << M) - 1)) & ~ends;
        }
    }
    return finish_query(q);
}

extern "C" agh_query *agh_query_literal(const unsigned char *pat, int m, int D, int nocase,
                                        const unsigned char *delim, int dlen)
{
    return agh_query_literal_ex(pat, m, D, nocase ? AGH_Q_NOCASE : 0u, delim, dlen);
}

// One byte through asearch.c's recurrence on reference-layout tables (host copy of what
// agh_table.hip runs; used only to reject degenerate queries when the query is built).
static uint32_t table_feed(const agh_dev_tables &T, const uint32_t *Mask, int k, uint32_t *B,
                           unsigned c)
{
    uint32_t A[AGH_MAX_ERRORS + 1], CM = Mask[c & 255u], ret = 0;
    A[0] = ((B[0] >> 1) & CM) | (T.Init1 & B[0]);
    for (int e = 1; e <= k; ++e)
        A[e] = ((B[e] >> 1) & CM) | (T.Init1 & B[e]) | B[e - 1] |
               (((A[e - 1] | B[e - 1]) >> 1) & T.NO_ERR);
    if (A[0] & T.D_endpos) {
        const uint32_t r1 = A[k] & T.endposition;
        ret = 1u | ((T.AND ? r1 == T.endposition : r1 != 0u) ? 2u : 0u);
        A[0] = (((T.Init0 >> 1) & CM) | (T.Init0 & T.Init1)) & T.D_Mask;
        for (int e = 1; e <= k; ++e)
            A[e] = ((T.Init0 >> 1) & CM) | (T.Init1 & T.Init0) | T.Init0 |
                   (((A[e - 1] | T.Init0) >> 1) & T.NO_ERR);
    }
    for (int e = 0; e <= k; ++e) B[e] = A[e];
    return ret;
}

// Queries whose state is not bounded by the last m+k+1 bytes ('#' wildcards, -p) or whose
// verdict needs several end bits (';' AND, ',' OR): the reference tables are kept as they are
// and run by the table engine (agh_table.hip).
static agh_query *table_query(const uint32_t Mask[256], uint32_t Init0, uint32_t Init1,
                              uint32_t NO_ERR_MASK, uint32_t endposition, uint32_t D_endpos,
                              int M, const unsigned char *old_D_pat, int D_length, int D, int AND,
                              bool delim_fold)
{
    // delimiter position p lives at bit M - p and accepts the byte old_D_pat[p - 1] (its case pair
    // under -i: the caller checked the classes); D_endpos is the last of them
    unsigned char dl[AGH_MAX_DELIM];
    for (int i = 0; i < D_length; ++i) {
        unsigned char c = old_D_pat[i];
        if (c == '^' || c == '$') c = '\n';                         // bitap.c:92-94
        if (delim_fold && is_upper(c)) c = (unsigned char)(c + 32);
        dl[i] = c;
        if (!((Mask[c] >> (M - 1 - i)) & 1u)) {
            fail("delimiter position %d of the tables does not accept the byte 0x%02x", i + 1, c);
            return nullptr;
        }
    }
    if (D_endpos != (1u << (M - D_length))) {
        fail("D_endpos is not the last delimiter position");
        return nullptr;
    }
    const unsigned char dc = dl[D_length - 1];
    agh_query *q = new agh_query();
    q->table = true;
    q->tab.Init0 = Init0;
    q->tab.Init1 = Init1;
    q->tab.NO_ERR = NO_ERR_MASK;
    q->tab.endposition = endposition;
    q->tab.D_endpos = D_endpos;
    {
        uint32_t dm = D_endpos;                                     // asearch.c:54-57
        for (int i = 1; i < D_length; ++i) dm = (dm << 1) | dm;
        q->tab.D_Mask = ~dm;
    }
    q->tab.AND = AND ? 1u : 0u;
    q->m = M - D_length - 1;
    q->k = D;
    q->dlen = D_length;
    memcpy(q->delim, dl, (size_t)D_length);
    q->delim_fold = delim_fold;
    memset(q->mask, 0, sizeof(q->mask));
    for (int c = 0; c < 256; ++c) q->mask[c] = Mask[c];             // uploaded unchanged
    // An empty record must not match (the virtual '\n' in front of the text and the delimiter
    // appended at EOF would otherwise produce records that do not exist, asearch.c:69-91).
    {
        uint32_t B[AGH_MAX_ERRORS + 1];
        for (int e = 0; e <= D; ++e) B[e] = Init0;
        uint32_t r = table_feed(q->tab, Mask, D, B, '\n');
        for (int rep = 0; rep < 2; ++rep)
            for (int i = 0; i < D_length; ++i) r |= table_feed(q->tab, Mask, D, B, dl[i]);
        (void)dc;
        if (r & 2u) {
            delete q;
            fail("the pattern matches the empty record with %d errors", D);
            return nullptr;
        }
    }
    if (agh_device_count() <= 0) {
        fail("no usable HIP device: libagrep_hip has no CPU path");
        delete q;
        return nullptr;
    }
    if (upload_tables(q) != 0) {
        agh_query_free(q);
        return nullptr;
    }
    return q;
}

extern "C" agh_query *agh_query_from_maskgen(const uint32_t Mask[256], uint32_t Init0,
                                             uint32_t Init1, uint32_t NO_ERR_MASK,
                                             uint32_t endposition, uint32_t D_endpos, int M,
                                             const unsigned char *old_D_pat, int D_length,
                                             int D, int AND)
{
    // reference layout (maskgen.c:218-257): positions 1..D_length = delimiter, D_length+1 =
    // the AND separator, D_length+2..M = pattern; position p lives at bit (M - p).
    if (!Mask || !old_D_pat || D_length < 1 || D_length > AGH_MAX_DELIM || M < D_length + 2 ||
        M > 31) {
        fail("malformed maskgen tables (M=%d, D_length=%d)", M, D_length);
        return nullptr;
    }
    const int m = M - D_length - 1;
    if (D < 0 || D > AGH_MAX_ERRORS || D >= m) {
        fail("number of errors %d must be smaller than the pattern length %d", D, m);
        return nullptr;
    }
    bool delim_fold = false;
    for (int p = 1; p <= D_length; ++p) {           // delimiter position p lives at bit M - p
        int members = 0, lo = -1;
        for (int c = 0; c < 256; ++c)
            if ((Mask[c] >> (M - p)) & 1u) { ++members; if (lo < 0) lo = c; }
        const bool pair = members == 2 && is_upper(lo) && ((Mask[lo + 32] >> (M - p)) & 1u);
        if (pair) { delim_fold = true; continue; }
        if (members != 1) {
            fail("delimiter position %d matches %d different bytes", p, members);
            return nullptr;
        }
    }
    const uint32_t sep = 1u << (M - D_length - 1);
    const uint32_t pad = M == 32 ? 0u : ~((1u << M) - 1u);
    if (AND || endposition != 1u || Init0 != (pad | sep) || Init1 != (Init0 | 1u | D_endpos) ||
        D_endpos != (1u << (M - D_length)))
        return table_query(Mask, Init0, Init1, NO_ERR_MASK, endposition, D_endpos, M, old_D_pat,
                           D_length, D, AND, delim_fold);
    // NO_ERR_MASK: 0-bits forbid error transitions into a position (<exact> segments,
    // maskgen.c:80-95, 222-223); pattern position p is reference bit (m - p) -> device bit p-1
    uint64_t no_err = 0;
    for (int p = 1; p <= m; ++p)
        if ((NO_ERR_MASK >> (m - p)) & 1u) no_err |= (uint64_t)1 << (p - 1);
    const bool exact_parts = no_err != (m == 64 ? ~0ull : (((uint64_t)1 << m) - 1));
    agh_query *q = new agh_query();
    q->m = m;
    q->k = D;
    q->dlen = D_length;
    for (int i = 0; i < D_length; ++i) {
        unsigned char c = old_D_pat[i];
        q->delim[i] = (c == '^' || c == '$') ? '\n' : c;    // bitap.c:92-94
        if (delim_fold && is_upper(q->delim[i])) q->delim[i] = (unsigned char)(q->delim[i] + 32);
    }
    q->delim_fold = delim_fold;
    for (int c = 0; c < 256; ++c) {
        uint64_t v = 0;
        for (int p = 1; p <= m; ++p)
            if ((Mask[c] >> (m - p)) & 1u) v |= (uint64_t)1 << (p - 1);
        q->mask[c] = v;
    }
    if (exact_parts) {
        q->general = true;
        q->no_err = no_err;
    }
    return finish_query(q);
}

// Device tables of the multi-pattern engine for `q` (prefix bit table, buckets, piece pool,
// per-pattern masks).  Entries: whole patterns (D = 0) or D+1 disjoint pieces of every pattern.
static int fill_multi_tables(agh_query *q, const unsigned char *const *pats, const int *lens,
                             int npat, int D, int nocase, unsigned char delim0, int *fq_out,
                             uint32_t *qmask_out, uint32_t *fold_out, int *minlen_out)
{
    // table entries: whole patterns (D = 0) or D+1 disjoint pieces of every pattern
    struct piece { int owner, po, len; };
    std::vector<piece> pcs;
    int minlen = 1 << 30;
    for (int p = 0; p < npat; ++p)
        for (int i = 0; i <= D; ++i) {
            const int a = (int)((long)i * lens[p] / (D + 1)), b = (int)((long)(i + 1) * lens[p] / (D + 1));
            pcs.push_back({p, a, b - a});
            if (b - a < minlen) minlen = b - a;
        }
    const int npc = (int)pcs.size();

    const int fq = minlen < 4 ? minlen : 4;     // length of the probed q-gram
    const uint32_t qmask = fq == 4 ? 0xffffffffu : ((1u << (8 * fq)) - 1u);
    const uint32_t fold = nocase ? (0x20202020u & qmask) : 0u;
    // Probe stride: an entry of length L that occurs verbatim at text position j contains a 4-gram
    // at a text position divisible by S at one of its offsets 0..S-1 as soon as L >= S + 3.  So a
    // set whose shortest entry has >= 7 (>= 5) bytes is probed at every 4th (2nd) position only,
    // with the grams of offsets 0..S-1 of every entry in the table -- 4 (8) probes per 16 bytes
    // instead of 16, no unaligned extraction at S = 4.
    const int stride = minlen >= 7 ? 4 : (minlen >= 5 ? 2 : 1);
    // ... and with entries of >= 8 bytes the grams at offsets 0..3 can take a fifth byte: 26 x fewer
    // chance hits for the verifier, same four probes
    const bool q5 = minlen >= 8;
    q->mp_stride = stride;
    q->mp_q5 = q5;
    *fq_out = fq;
    *qmask_out = qmask;
    *fold_out = fold;
    *minlen_out = minlen;

    const uint32_t NB = 1u << AGH_MP_BUCKET_BITS;
    std::vector<uint32_t> bits((1u << AGH_MP_BITS) / 32, 0), off(npc + 1, 0);
    std::vector<uint8_t> pool;
    std::vector<uint32_t> bstart(NB + 1, 0);
    std::vector<char> usable(npc, 1);
    struct gram_item { uint32_t bucket, piece, o; };
    std::vector<gram_item> gi;
    for (int i = 0; i < npc; ++i) {
        const unsigned char *src = pats[pcs[i].owner] + pcs[i].po;
        off[i] = (uint32_t)pool.size();
        for (int t = 0; t < pcs[i].len; ++t) {
            unsigned char c = src[t];
            if (q->dlen == 1 && !q->delim_fold && c == delim0) usable[i] = 0;   // can never lie inside one record
            if (nocase && is_upper(c)) c += 32;
            pool.push_back(c);
        }
        if (q->dlen > 1 || q->delim_fold) {     // ... nor can an entry that holds the whole delimiter
            for (int t = 0; t + q->dlen <= pcs[i].len && usable[i]; ++t) {
                bool same = true;
                for (int j = 0; j < q->dlen && same; ++j) {
                    unsigned char c = src[t + j];
                    if (q->delim_fold && is_upper(c)) c += 32;
                    same = c == q->delim[j];
                }
                if (same) usable[i] = 0;
            }
        }
        if (!usable[i]) continue;
        for (int o = 0; o < stride; ++o) {
            uint32_t g = 0;
            for (int t = 0; t < fq; ++t) g |= (uint32_t)src[o + t] << (8 * t);     // o + fq <= len: len >= stride + 3
            g = (g & qmask) | fold;
            uint32_t hs = g;                    // what the sweep hashes: the gram, or the 5-byte mix
            if (q5) hs = agh_mix5(g, (uint32_t)src[o + 4] | (fold ? 0x20u : 0u));
            const uint32_t h = fq == 4 ? (agh_sample_prod_q4(hs) & ((1u << AGH_MP_BITS) - 1u)) : agh_sample_hash18_q3(hs);
            bits[h >> 5] |= 1u << (h & 31u);
            if (fq == 4) {                      // second Bloom probe (agh_multi.hip probe_chunk)
                const uint32_t h2 = agh_sample_hash18b_q4(hs);
                bits[h2 >> 5] |= 1u << (h2 & 31u);
            }
            gi.push_back({agh_mp_bucket(g), (uint32_t)i, (uint32_t)o});
            bstart[gi.back().bucket + 1]++;
        }
    }
    off[npc] = (uint32_t)pool.size();
    pool.resize(pool.size() + 16, 0);           // the exact verifier reads 16 bytes at any entry
    if (pool.size() >= (1u << 24)) return fail("pattern set too large (%zu bytes)", pool.size());
    if ((size_t)npc >= (1u << 28)) return fail("too many pattern pieces");
    for (uint32_t b = 0; b < NB; ++b) bstart[b + 1] += bstart[b];
    // bucket items (agh_launch.h): everything the verifier needs about an entry in one 16-byte load
    std::vector<agh_mp_item> items(gi.size() ? gi.size() : 1);
    memset(items.data(), 0, items.size() * sizeof(agh_mp_item));
    {
        std::vector<uint32_t> fill(bstart.begin(), bstart.end() - 1);
        for (const gram_item &x : gi) {
            const uint32_t at = fill[x.bucket]++;
            items[at].piece = x.piece | (x.o << 28);
            items[at].info = (off[x.piece] << 8) | (uint32_t)pcs[x.piece].len;
            items[at].owner = (uint32_t)pcs[x.piece].owner;
            items[at].pom = ((uint32_t)pcs[x.piece].po << 8) | (uint32_t)(lens[pcs[x.piece].owner] & 0xff);
        }
    }
    // per-pattern position masks for the verifying automaton (as agh_query_literal builds them)
    std::vector<uint32_t> omask;
    if (D > 0) {
        omask.assign((size_t)npat * 256, 0u);
        for (int p = 0; p < npat; ++p) {
            for (int t = 0; t < lens[p]; ++t) {
                int c = pats[p][t];
                if (nocase && is_upper(c)) c += 32;
                omask[(size_t)p * 256 + c] |= 1u << t;
                if (nocase && is_lower(c)) omask[(size_t)p * 256 + c - 32] |= 1u << t;
            }
        }
    }
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        HIP_TRY(hipMalloc(dst, bytes ? bytes : 4));
        if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return 0;
    };
    bstart.push_back(bstart.back());            // (the verifier reads [b] and [b + 1] as one pair)
    if (up(&q->d_mp_bits, bits.data(), bits.size() * 4) || up(&q->d_mp_bstart, bstart.data(), bstart.size() * 4) ||
        up(&q->d_mp_items, items.data(), items.size() * sizeof(agh_mp_item)) ||
        up(&q->d_mp_pool, pool.data(), pool.size()) ||
        up(&q->d_mp_omask, omask.data(), omask.size() * 4))
        return -1;

    // ---- tables of the one-pass count-only scan (agh_mscan.hip) ---------------------------------
    // Sets it takes: 4-byte grams, a one-byte delimiter, no -w / -x, and entries it can check out of two
    // 16-byte text loads -- k = 0: patterns of 4..15 bytes; k = 1: patterns of 8..14 bytes (two pieces of
    // 4..7 bytes; the side next to a verbatim piece has <= 7 bytes: side_within_one_edit).  Everything
    // else (and numbered scans of these sets) stays on k_sweep_multi + k_verify_multi.
    q->ms_ok = false;
    // ... and those probed at every or every second position: with entries of >= 7 bytes the strided sweep
    // (4 probes per 16 bytes) is the faster one (profiles/r04_perf_c5_b.log: 0.91 vs 1.00 ms per 4 GiB).
    bool ms = fq == 4 && q->dlen == 1 && !q->delim_fold && !q->guard && D <= 1 && q->multi && stride <= 2;
    {
        const char *e = getenv("AGH_MSCAN");
        if (e && e[0] == '0') ms = false;
    }
    for (int p = 0; p < npat && ms && D == 1; ++p) ms = lens[p] >= 8 && lens[p] <= 14;
    for (int i = 0; i < npc && ms && D == 0; ++i) ms = !usable[i] || pcs[i].len <= 15;
    if (ms) {
        struct ms_entry { uint32_t key; uint32_t w[4]; };
        std::vector<ms_entry> es;
        auto pb = [&](int i, int t) -> uint32_t { return t < pcs[i].len ? pool[off[i] + t] : 0u; };
        for (int i = 0; i < npc; ++i) {
            if (!usable[i]) continue;
            ms_entry e;
            e.w[0] = pb(i, 0) | pb(i, 1) << 8 | pb(i, 2) << 16 | pb(i, 3) << 24;
            e.key = e.w[0] | fold;
            if (D == 0) {
                e.w[1] = pb(i, 4) | pb(i, 5) << 8 | pb(i, 6) << 16 | pb(i, 7) << 24;
                e.w[2] = pb(i, 8) | pb(i, 9) << 8 | pb(i, 10) << 16 | pb(i, 11) << 24;
                e.w[3] = pb(i, 12) | pb(i, 13) << 8 | pb(i, 14) << 16 | (uint32_t)pcs[i].len << 24;
            } else {
                // the other side of the pattern, nearest byte first, in the case the pool has
                const int m = lens[pcs[i].owner], po = pcs[i].po, len = pcs[i].len;
                const bool before = po > 0;
                const int L = before ? po : m - len;
                uint8_t B[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int t = 0; t < L; ++t) {
                    unsigned char c = before ? pats[pcs[i].owner][po - 1 - t] : pats[pcs[i].owner][len + t];
                    if (nocase && is_upper(c)) c += 32;
                    B[t] = c;
                }
                const uint32_t meta = (uint32_t)L | (uint32_t)(len - 4) << 3 | (before ? 32u : 0u);
                e.w[1] = B[0] | B[1] << 8 | B[2] << 16 | (uint32_t)B[3] << 24;
                e.w[2] = B[4] | B[5] << 8 | B[6] << 16 | meta << 24;
                e.w[3] = pb(i, 4) | pb(i, 5) << 8 | pb(i, 6) << 16;
            }
            if (e.key == 0) ms = false;
            es.push_back(e);
        }
        std::stable_sort(es.begin(), es.end(), [](const ms_entry &a, const ms_entry &b) { return a.key < b.key; });
        uint32_t rb = 13;
        {
            const char *e = getenv("AGH_MSCAN_RB");
            if (e && e[0] == '1' && e[1] == '2') rb = 12;
        }
#if AGH_MS_NBF == 2
        rb = 12;                                    // (16 KiB of masks next to the gram table in LDS)
        // ms.gtab: AGH_MS_GSLOTS grams, then AGH_MS_GSLOTS masks of fifth bytes (agh_mscan.hip level 2)
        std::vector<uint32_t> ptab((size_t)2 << rb, 0), gtab((size_t)AGH_MS_GSLOTS * 2, 0), mdir(AGH_MS_GSLOTS, 0);
#elif AGH_MS_NBF
        // (queue B entries are 8 bytes in this build: with 2^13 rows the workgroup would sit exactly on the 160 KiB)
        rb = 12;
        // per gram slot: dir, MX (fifth byte of the entries of >= 5 bytes), ML / MB (the two nearest bytes of the
        // other side of the pattern behind / in front of a 4-byte piece); bit = byte & 31 (agh_mscan.hip stage_b)
        std::vector<uint32_t> ptab((size_t)2 << rb, 0), gtab(AGH_MS_GSLOTS, 0), mdir((size_t)AGH_MS_GSLOTS * 4, 0);
#else
        std::vector<uint32_t> ptab((size_t)2 << rb, 0), gtab(AGH_MS_GSLOTS, 0), mdir(AGH_MS_GSLOTS, 0);
#endif
        std::vector<uint32_t> ment(es.size() * 4 + 4, 0);
        size_t n_grams = 0;
        for (size_t a = 0; a < es.size() && ms;) {
            size_t b = a;
            while (b < es.size() && es[b].key == es[a].key) ++b;
            const uint32_t g = es[a].key;
            if (b - a > 255 || ++n_grams > AGH_MS_GSLOTS * 6 / 10) { ms = false; break; }
            const uint32_t k0 = g & 0xffu, k3 = g >> 24;
            ptab[2 * agh_ms_row(g >> 8, rb)] |= 1u << (k0 & 31u);            // pre of (k1 k2 k3)
            ptab[2 * agh_ms_row(g & 0xffffffu, rb) + 1] |= 1u << (k3 & 31u);     // suf of (k0 k1 k2)
            const uint32_t gh = agh_ms_ghash(g), b1 = AGH_MS_GB1(gh), b2 = AGH_MS_GB2(gh);
            auto load_of = [&](uint32_t bk) { int c = 0; while (c < 4 && gtab[4 * bk + c]) ++c; return c; };
            const int l1 = load_of(b1), l2 = load_of(b2);
            if (l1 >= 4 && l2 >= 4) { ms = false; break; }     // (both buckets full: the set stays on the two-kernel form)
            const uint32_t sl = l1 <= l2 ? 4 * b1 + (uint32_t)l1 : 4 * b2 + (uint32_t)l2;
            gtab[sl] = g;
#if AGH_MS_NBF == 2
            {
                uint32_t mx = 0;
                for (size_t i = a; i < b; ++i) {
                    const ms_entry &e = es[i];
                    if (D == 0) mx |= (e.w[3] >> 24) >= 5 ? 1u << (e.w[1] & 31u) : ~0u;
                    else mx |= ((e.w[2] >> 27) & 3u) >= 1 ? 1u << (e.w[3] & 31u) : ~0u;     // (meta >> 3) & 3 = len - 4
                }
                gtab[AGH_MS_GSLOTS + sl] = mx;
            }
            mdir[sl] = (uint32_t)a << 8 | (uint32_t)(b - a);
#elif AGH_MS_NBF
            {
                uint32_t mx = 0, ml = 0, mb = 0;
                for (size_t i = a; i < b; ++i) {
                    const ms_entry &e = es[i];
                    if (D == 0) {
                        const uint32_t len = e.w[3] >> 24;
                        mx |= len >= 5 ? 1u << (e.w[1] & 31u) : ~0u;           // 4 bytes: nothing more to ask
                    } else {
                        const uint32_t meta = e.w[2] >> 24, L = meta & 7u, tl = (meta >> 3) & 3u;
                        const uint32_t near2 = L >= 2 ? (1u << (e.w[1] & 31u)) | (1u << ((e.w[1] >> 8) & 31u)) : ~0u;
                        if (tl >= 1) mx |= 1u << (e.w[3] & 31u);
                        else if (meta & 32u) mb |= near2;
                        else ml |= near2;
                    }
                }
                mdir[4 * sl] = (uint32_t)a << 8 | (uint32_t)(b - a);
                mdir[4 * sl + 1] = mx;
                mdir[4 * sl + 2] = ml;
                mdir[4 * sl + 3] = mb;
            }
#else
            mdir[sl] = (uint32_t)a << 8 | (uint32_t)(b - a);
#endif
            a = b;
        }
        if (es.size() >= (1u << 24)) ms = false;
        for (size_t i = 0; i < es.size(); ++i) memcpy(&ment[4 * i], es[i].w, 16);
        if (ms) {
            if (up(&q->d_ms_ptab, ptab.data(), ptab.size() * 4) || up(&q->d_ms_gtab, gtab.data(), gtab.size() * 4) ||
                up(&q->d_ms_mdir, mdir.data(), mdir.size() * 4) || up(&q->d_ms_ment, ment.data(), ment.size() * 4))
                return -1;
            q->ms_ok = true;
            q->ms_rb = rb;
            const char *dbg = getenv("AGH_MSCAN_DBG");
            q->ms_dbg = dbg ? (uint32_t)strtoul(dbg, nullptr, 0) : 0u;
        }
    }
    return 0;
}


// -f patternfile: the role of prepf() (newmgrep.c:192-375).  D = 0 is the reference's behaviour
// (compat.c:34-37: "approximate matching is not supported with -f"); D > 0 is the union of the
// single-pattern k-error predicate over all patterns (BASELINE config 5), filtered through
// D+1 verbatim pieces per pattern (agh_multi.hip).
static agh_query *build_multi(const unsigned char *const *pats, const int *lens, int npat, int D,
                              int nocase, const unsigned char *delim, int dlen, int guard = 0)
{
    if (!pats || !lens || npat < 1) { fail("no patterns"); return nullptr; }
    if (guard && D > 0) { fail("-w / -x with a pattern file need exact matching"); return nullptr; }
    if (!delim || dlen < 1 || dlen > AGH_MAX_DELIM) {
        fail("delimiter length %d outside 1..%d", dlen, AGH_MAX_DELIM);
        return nullptr;
    }
    // delimiters of several bytes, and letters under -i (maskgen.c:259-266), take their record ends
    // from the delimiter bitmap like the single-pattern engines
    bool delim_letters = false;
    unsigned char dl[AGH_MAX_DELIM];
    for (int i = 0; i < dlen; ++i) {
        dl[i] = delim[i];
        delim_letters = delim_letters || is_upper(delim[i]) || is_lower(delim[i]);
        if (nocase && is_upper(dl[i])) dl[i] = (unsigned char)(dl[i] + 32);
    }
    const bool dfold = nocase && delim_letters;
    const unsigned char dlast = dl[dlen - 1];
    if (D < 0 || D > AGH_MAX_ERRORS) { fail("number of errors %d outside 0..%d", D, AGH_MAX_ERRORS); return nullptr; }
    for (int p = 0; p < npat; ++p) {
        if (lens[p] < 1 || lens[p] > 255) { fail("pattern %d: length %d outside 1..255", p, lens[p]); return nullptr; }
        if (D > 0 && (lens[p] > 32 || lens[p] <= D)) {
            fail("pattern %d: with %d errors the length %d must be in %d..32", p, D, lens[p], D + 1);
            return nullptr;
        }
        if (D > 0)
            for (int t = 0; t < lens[p]; ++t) {
                const unsigned char c = pats[p][t], cf = (dfold && is_upper(c)) ? (unsigned char)(c + 32) : c;
                if (cf == dlast || c == '\n') {
                    fail("pattern %d holds the byte that ends a record (not supported with errors)", p);
                    return nullptr;
                }
            }
    }
    if (agh_device_count() <= 0) { fail("no usable HIP device: libagrep_hip has no CPU path"); return nullptr; }

    agh_query *q = new agh_query();
    q->multi = true;
    q->npat = npat;
    q->guard = guard;
    q->k = D;
    q->dlen = dlen;
    memcpy(q->delim, dl, (size_t)dlen);
    q->delim_fold = dfold;
    memset(q->mask, 0, sizeof(q->mask));
    if (fill_multi_tables(q, pats, lens, npat, D, nocase, delim[0], &q->fq, &q->qmask, &q->fold,
                          &q->m) ||
        upload_common(q)) {
        agh_query_free(q);
        return nullptr;
    }
    q->fh = q->mp_stride;
    return q;
}

extern "C" agh_query *agh_query_multi(const unsigned char *const *pats, const int *lens, int npat,
                                      int nocase, const unsigned char *delim, int dlen)
{
    return build_multi(pats, lens, npat, 0, nocase, delim, dlen);
}

extern "C" agh_query *agh_query_multi_ex(const unsigned char *const *pats, const int *lens, int npat,
                                         unsigned qflags, const unsigned char *delim, int dlen)
{
    if ((qflags & AGH_Q_WORD) && (qflags & AGH_Q_WHOLELINE)) {          // agrep.c:2188-2196
        fail("illegal option combination (-x and -w)");
        return nullptr;
    }
    return build_multi(pats, lens, npat, 0, (qflags & AGH_Q_NOCASE) ? 1 : 0, delim, dlen,
                       (qflags & AGH_Q_WHOLELINE) ? 2 : ((qflags & AGH_Q_WORD) ? 1 : 0));
}

extern "C" agh_query *agh_query_multi_approx(const unsigned char *const *pats, const int *lens,
                                             int npat, int D, int nocase,
                                             const unsigned char *delim, int dlen)
{
    return build_multi(pats, lens, npat, D, nocase, delim, dlen);
}

// asearch1.c:42-44 / agrep.c:2680-2696 (-I# -S# -D#): non-unit edit costs.  Such queries run
// on the general automaton (full scan).
extern "C" int agh_query_set_costs(agh_query *q, int I, int S, int DD)
{
    if (!q) return fail("null query");
    if (q->multi) return fail("multi-pattern queries use unit costs");
    if (I < 1 || S < 1 || DD < 1)
        return fail("costs must be >= 1 (cost 0 turns every position into a self loop, asearch1.c:41)");
    q->ci = I;
    q->cs = S;
    q->cd = DD;
    // (the table engine runs asearch1.c's recurrence on its own tables: agh_table.hip feed_costs)
    if ((I != 1 || S != 1 || DD != 1) && !q->table) q->general = true;
    return 0;
}

extern "C" void agh_query_free(agh_query *q)
{
    if (!q) return;
    if (q->d_mask) (void)hipFree(q->d_mask);
    if (q->d_ftab) (void)hipFree(q->d_ftab);
    if (q->d_gtab) (void)hipFree(q->d_gtab);
    if (q->d_mp_bits) (void)hipFree(q->d_mp_bits);
    if (q->d_mp_bstart) (void)hipFree(q->d_mp_bstart);
    if (q->d_mp_items) (void)hipFree(q->d_mp_items);
    if (q->d_mp_pool) (void)hipFree(q->d_mp_pool);
    if (q->d_mp_omask) (void)hipFree(q->d_mp_omask);
    if (q->d_ms_ptab) (void)hipFree(q->d_ms_ptab);
    if (q->d_ms_gtab) (void)hipFree(q->d_ms_gtab);
    if (q->d_ms_mdir) (void)hipFree(q->d_ms_mdir);
    if (q->d_ms_ment) (void)hipFree(q->d_ms_ment);
    if (q->d_acc) (void)hipFree(q->d_acc);
    if (q->h_acc) (void)hipHostFree(q->h_acc);
    if (q->d_counters) (void)hipFree(q->d_counters);
    if (q->d_chunk_totals) (void)hipFree(q->d_chunk_totals);
    if (q->h_counters) (void)hipHostFree(q->h_counters);
    if (q->h_cuts) (void)hipHostFree(q->h_cuts);
    for (hipEvent_t e : q->dep_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : q->time_events) (void)hipEventDestroy(e);
    if (q->aux_stream) (void)hipStreamDestroy(q->aux_stream);
    q->cand_b.release();
    q->wave_cand_b.release();
    q->cuts.release();
    q->seg_copy.release();
    q->seg_dbm.release();
    q->tickets.release();
    q->giveups.release();
    if (q->ev0) (void)hipEventDestroy(q->ev0);
    if (q->ev1) (void)hipEventDestroy(q->ev1);
    if (q->ev2) (void)hipEventDestroy(q->ev2);
    if (q->ev3) (void)hipEventDestroy(q->ev3);
    q->strip_prefix.release();
    q->wave_totals.release();
    q->cand.release();
    q->wave_cand.release();
    q->bitmap.release();
    q->hashset.release();
    q->dbm.release();
    q->staging.release();
    q->match_pos.release();
    q->match_rec.release();
    q->match_start.release();
    q->match_end.release();
    q->match_off.release();
    q->gather.release();
    q->staging_b.release();
    for (int b = 0; b < AGH_PIN_RING; ++b) {
        if (q->pinned[b]) (void)hipHostFree(q->pinned[b]);
        if (q->pinned_ev[b]) (void)hipEventDestroy(q->pinned_ev[b]);
    }
    if (q->stage_stream) (void)hipStreamDestroy(q->stage_stream);
    delete q;
}

extern "C" int agh_query_info(const agh_query *q, int *m, int *D, int *filter_q, int *filter_h)
{
    if (!q) return fail("null query");
    if (m) *m = q->m;
    if (D) *D = q->k;
    if (filter_q) *filter_q = q->fq;
    if (filter_h) *filter_h = q->fh;
    return 0;
}

// (AGH_TIGHT_VERIFY=0: offset-blind verify windows, no false-positive rejection -- A/B runs; q->tune)

// Full scan, fast form (k_fullscan_fast + k_fullscan_replay): unit costs, a one-byte delimiter that
// no pattern position accepts.  The replay lists live in the candidate buffers a full scan does not
// use otherwise.  AGH_FS_FAST=0: the exact kernel (A/B runs).
#define AGH_FF_SLICE_HOST 256u      // = AGH_FF_SLICE (agh_fullscan.hip)
// The table engine's fast form walks its chunk serially: with 4 KiB per lane a wave needs ~0.28 ms however small
// the text, and the exact kernel (1 KiB per lane, floor 0.18 ms) won below ~320 MiB.  With the chunk chosen by the
// size of the text (1 KiB below 512 MiB: floor 0.11 ms) the fast form wins at every size
// (profiles/r04_perf_table_sizes.log), so there is no switch-over any more; AGH_TF_FAST_MIN_MB brings one back.
#ifndef AGH_TF_FAST_MIN_MB_DEFAULT
#define AGH_TF_FAST_MIN_MB_DEFAULT 0
#endif
#ifndef AGH_TF_CHUNK_DEFAULT
#define AGH_TF_CHUNK_DEFAULT 0          // bytes per lane of the fast form (1024 / 2048 / 4096); 0: by the size of the text
#endif
static bool fs_fast_ok(const agh_query *q)
{
    if (!q->tune.fs_fast || q->fs_fast_off || q->multi) return false;
    // table engine (k_tablescan_fast + k_table_replay): unit costs, one-byte delimiter
    if (q->table) return q->ci == 1 && q->cs == 1 && q->cd == 1 && !(q->dlen > 1 || q->delim_fold);
    // k = 0: one level, nothing to pack -- the one-kernel form is faster there (3.8 vs 3.2 TB/s)
    return q->k >= 1 && !q->general && !(q->dlen > 1 || q->delim_fold) && q->mask[q->delim[0]] == 0;
}

static int fs_fast_setup(agh_query *q, uint64_t n, agh_scan_args *va)
{
    va->fs_fast = 0;
    va->tf_chunk = 0;
    if (!fs_fast_ok(q)) return 0;
    if (q->table && n < (q->tune.tf_fast_min_mb << 20)) return 0;
    // (64 KiB tiles with 256 entries each; the table engine's 256 KiB tiles with 1024 entries need the same)
    const uint64_t n_tiles = (n + 65535) / 65536;
    if (q->cand.ensure((n_tiles + 4) * AGH_FF_SLICE_HOST * sizeof(uint64_t))) return -1;
    if (q->wave_cand.ensure((n_tiles + 8) * sizeof(uint32_t))) return -1;
    va->fs_fast = 1;
    va->tf_chunk = q->tune.tf_chunk;
    // table engine, M <= 15: two streams per lane (k_tablescan_fast2) -- the always-one bit M must be there
    if (q->table && q->tune.tf_pack2) {
        const unsigned M = (unsigned)q->m + (unsigned)q->dlen + 1u;
        if (M <= 15u && ((q->tab.Init0 >> M) & (q->tab.Init1 >> M) & 1u)) va->fs_fast = 2;
    }
    va->fs_replay = (uint64_t *)q->cand.p;
    va->fs_tile_cnt = (uint32_t *)q->wave_cand.p;
    return 0;
}

static agh_multi_dev multi_dev(const agh_query *q, const uint64_t *dbm)
{
    agh_multi_dev m;
    m.dbm = dbm;
    m.bits = (const uint32_t *)q->d_mp_bits;
    m.bucket_start = (const uint32_t *)q->d_mp_bstart;
    m.items = (const agh_mp_item *)q->d_mp_items;
    m.pool = (const uint8_t *)q->d_mp_pool;
    m.owner_mask = (const uint32_t *)q->d_mp_omask;
    return m;
}

// ---------------------------------------------------------------------------------------
// one segment (<= AGH_SEG_MAX bytes) resident in HBM
// ---------------------------------------------------------------------------------------
static const uint64_t AGH_SEG_MAX_DEFAULT = (uint64_t)8 << 30;   // nominal segment (plan_segments)
static const uint64_t AGH_LEAN_SEG_MAX = (uint64_t)64 << 30;     // lean scans: one launch per 64 GiB
// lean pipeline defaults (lean_run): part size in MiB (0: one launch per segment) and whether the
// verifier runs on a second stream; AGH_PART_MB / AGH_OVERLAP override (A/B runs)
#define AGH_PART_MB_DEFAULT 0
#define AGH_OVERLAP_DEFAULT 0
#define AGH_FUSED_DEFAULT 1
// ... from this segment size on (MiB; AGH_FUSED_MIN_MB overrides, the tests run with 0).  The
// persistent kernel pays ~70 us once (the candidates queued last are verified after the stream has
// ended, and 4096 waves drawing 256 KiB tickets finish less evenly than hardware-dispatched
// workgroups): measured against the two-kernel form (scripts/ab_fused.py, k = 2 / k = 0) it is
// -3 % / -3 % at 4 GiB, even at 8 GiB, +1.8 % / +1.6 % at 16 GiB, +5 % / 0 % at 64 GiB in round 2;
// round 3 (H = 2 samples at k = 2, small tickets at the end; profiles/r03_ab_headline.log): -3 % / -4 %
// at 4 GiB, +1.9 % / +0.2 % at 8 GiB, +3.5 % / +3.5 % at 16 GiB, +6.7 % / +3.7 % at 64 GiB; with the final
// grid (agh_fused.hip launch_fused; profiles/r03_ab_headline_final.log): -10 % / -13 % at 1 GiB,
// -4 % / -3 % at 2 GiB, +3.3 % / +2.5 % at 4 GiB, +4.9 % / +3.7 % at 8 GiB, +6 % / +4.8 % at 64 GiB.
#define AGH_FUSED_MIN_MB_DEFAULT 4096


static int get_events(std::vector<hipEvent_t> &pool, size_t want, unsigned evflags);

// Lean scans with a one-byte delimiter: room for the matches whose record starts further back than the
// verifier looks; resolved after the scan by k_resolve_giveups instead of a rerun of the segment.
#define AGH_GIVEUP_CAP 4096u
static int attach_giveups(agh_query *q, agh_marks *mk)
{
    mk->giveups = nullptr;
    mk->giveup_cap = 0;
    if (q_mb(q) || q->tune.giveup_cap == 0) return 0;
    const uint32_t cap = q->tune.giveup_cap < 0 ? AGH_GIVEUP_CAP : (uint32_t)q->tune.giveup_cap;
    if (q->giveups.ensure((size_t)cap * sizeof(uint64_t))) return -1;
    mk->giveups = (uint64_t *)q->giveups.p;
    mk->giveup_cap = cap;
    return 0;
}

struct seg_result {
    uint64_t matched = 0, records = 0, candidates = 0, stored = 0;
    uint32_t engine = 0, truncated = 0, lean_rerun = 0;
    float ms = 0.f, sweep_ms = 0.f;
};

static int scan_segment(agh_query *q, const void *d_text, uint64_t n, hipStream_t st,
                        unsigned flags, uint32_t head_byte, int tail_virtual,
                        uint64_t *d_match_pos, uint32_t *d_match_rec, uint32_t match_cap,
                        seg_result *out, const uint64_t *pre_dbm)
{
    *out = seg_result();
    if (n == 0) return 0;
    uint32_t lean_rerun = 0;
    if (((uintptr_t)d_text & 15u) != 0) return fail("device text must be 16-byte aligned");
    // piece engine of a single literal pattern (see attach_piece_engine)
    // -v with a record list needs the census arrays of the byte-parallel engines
    const bool invert = (flags & AGH_INVERT) != 0;
    const bool invert_list = invert && d_match_pos != nullptr;
    if (invert_list && q->multi) return fail("-v with record output is not supported for pattern files");
    const bool pe = q->piece_single && !q->general && !(flags & AGH_FORCE_FULLSCAN) && !invert_list;
    const bool multi = q->multi || pe;
    const uint64_t n_strips = (n + AGH_STRIP - 1) >> AGH_STRIP_SHIFT;
    const uint64_t nw = (n_strips + AGH_WAVE_STRIPS - 1) / AGH_WAVE_STRIPS;
    if (multi && n > ((uint64_t)4 << 30)) return fail("multi-pattern segments are limited to 4 GiB");
    if (n > ((uint64_t)16 << 30) - 4096) return fail("segments are limited to 16 GiB");
    if (q->fh == 2 && !multi && n > ((uint64_t)8 << 30) - 4096) return fail("segments of this query are limited to 8 GiB");
    if (multi && (flags & AGH_FORCE_FULLSCAN))
        return fail("multi-pattern queries have no full-scan engine");
    const bool want_filter = (q->fq > 0 || pe) && !(flags & AGH_FORCE_FULLSCAN) && !q->table && !invert_list &&
                             !(q->general && (q_mb(q) || pe));   // general verify: 1-byte delimiters
    if (!want_filter && (flags & AGH_FORCE_FILTER))
        return fail("the q-gram filter does not apply to this query (m=%d, k=%d)", q->m, q->k);
    if (q->strip_prefix.ensure((n_strips + 8) * sizeof(uint32_t))) return -1;
    if (q->wave_totals.ensure((nw + 8) * sizeof(uint32_t))) return -1;
    if (want_filter) {
        if (q->cand.ensure(nw * (multi ? AGH_MP_SLICE_CAP : AGH_SLICE_CAP) * sizeof(uint64_t))) return -1;
        if (q->wave_cand.ensure((nw + 8) * sizeof(uint32_t))) return -1;
    }

    agh_dev_query dq;
    dq.m = q->m;
    dq.k = q->k;
    dq.delim = q->delim[q->dlen - 1];           // the byte that completes a delimiter
    dq.dlen = (uint32_t)q->dlen;
    memset(dq.dbytes, 0, sizeof(dq.dbytes));
    memcpy(dq.dbytes, q->delim, (size_t)q->dlen);
    dq.dfold = q->delim_fold ? 1u : 0u;
    dq.mb = q_mb(q) ? 1u : 0u;
    dq.mp_q5 = (multi && q->mp_q5) ? 1u : 0u;
    dq.guard = q->multi ? (uint32_t)q->guard : 0u;
    dq.fq = pe ? q->pe_fq : q->fq;
    dq.fh = pe ? q->mp_stride : q->fh;                  // multi-pattern sweeps: the probe stride
    dq.qmask = pe ? q->pe_qmask : q->qmask;
    dq.fold = pe ? q->pe_fold : q->fold;
    dq.ci = (uint32_t)std::min(q->ci, q->k + 1);       // asearch1.c:42-44
    dq.cs = (uint32_t)std::min(q->cs, q->k + 1);
    dq.cd = (uint32_t)std::min(q->cd, q->k + 1);
    dq.no_err = q->no_err;
    dq.head_byte = head_byte;
    dq.tail_virtual = tail_virtual;

    // ---- multi-byte delimiter: mark where (selected) delimiter occurrences end --------------
    const uint64_t *d_dbm = pre_dbm;           // the caller marked the delimiters of the whole text
    if (q_mb(q) && !pre_dbm) {
        const uint64_t n_words = (n + 63) / 64 + 4;     // readers may touch a few words past n
        // (a copied segment of a longer text: the bitmap of the whole text stays intact for the
        // record bounds that are computed afterwards)
        dev_buf &dbm_buf = q->seg_dbm_active ? q->seg_dbm : q->dbm;
        if (dbm_buf.ensure(n_words * sizeof(uint64_t))) return -1;
        HIP_TRY(hipMemsetAsync(q->d_counters, 0, AGH_C_COUNT * sizeof(uint32_t), st));
        agh_launch_delim_bitmap(d_text, n, dq, (uint64_t *)dbm_buf.p, n_words, q->d_counters, st);
        HIP_TRY(hipMemcpyAsync(q->h_counters, q->d_counters, AGH_C_COUNT * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (q->h_counters[AGH_C_DELIM_CHAIN])
            return fail("a run of overlapping delimiter occurrences exceeds 4 KiB (unsupported)");
        d_dbm = (const uint64_t *)dbm_buf.p;
    }

    // ---- lean pipeline: count-only scans (-c, -l) of a filterable query -------------------
    // No delimiter census, no record numbers: the verifier identifies a matched record by the
    // offset of its first byte and de-duplicates through a hash set; the count is the number
    // of occupied slots.  Gives up (-> numbered pipeline below) when a record start lies more
    // than AGH_LEAN_BACK_CAP bytes in front of a match, the set fills up, or slices overflow.
    const bool lean_ok = want_filter && !d_match_pos && (flags & (AGH_COUNT | AGH_FILENAMEONLY)) &&
                         !(flags & AGH_FORCE_NUMBERED) && !invert;      // -v needs the record count
    if (lean_ok) {
        // 2^17 slots at least; without a hint from a previous scan (4 x matched) one slot per
        // 8 KiB of text, i.e. room for a match every 32 KiB at 25 % load -- denser texts fall
        // back to the numbered pipeline once and come back with a hint
        uint64_t slots = 1u << 17;
        if (!q->hashset_slots_hint) while (slots < (n >> 13) && slots < (1u << 26)) slots <<= 1;
        while (slots < q->hashset_slots_hint) slots <<= 1;
        {
            const size_t cap_before = q->hashset.cap;   // (a new block may reuse the old address)
            if (q->hashset.ensure(slots * sizeof(uint64_t))) return -1;
            if (q->hashset.cap != cap_before || q->hashset_dirty)
                HIP_TRY(hipMemsetAsync(q->hashset.p, 0, q->hashset.cap, st));
            q->hashset_dirty = true;
        }
        dq.tail_virtual = tail_virtual;
        if (flags & AGH_TIME_SCAN) HIP_TRY(hipEventRecord(q->ev0, st));
        HIP_TRY(hipMemsetAsync(q->d_counters, 0, AGH_C_COUNT * sizeof(uint32_t), st));
        agh_sweep_args sa;
        sa.text = d_text;
        sa.n = n;
        sa.q = dq;
        sa.ftab = multi ? (const uint8_t *)q->d_mp_bits : q->d_ftab;
        sa.strip_prefix = (uint32_t *)q->strip_prefix.p;
        sa.wave_totals = (uint32_t *)q->wave_totals.p;
        sa.cand = (uint64_t *)q->cand.p;
        sa.wave_cand = (uint32_t *)q->wave_cand.p;
        sa.counters = q->d_counters;
        sa.chunk_totals = q->d_chunk_totals;
        sa.dbm = d_dbm;
        sa.lean = 1;
        sa.ev_begin = q->ev2;
        sa.ev_end = q->ev3;
        if (!(flags & AGH_TIME_SWEEP)) sa.ev_begin = sa.ev_end = nullptr;   // two events cost ~11 us per scan
        agh_scan_args va;
        memset(&va, 0, sizeof(va));
        va.verify_blocks = q->tune.verify_blocks;
        va.mk.counters = q->d_counters;
        va.mk.hashset = (uint64_t *)q->hashset.p;
        va.mk.hashset_mask = (uint32_t)(slots - 1);
        if (attach_giveups(q, &va.mk)) return -1;
        va.text = d_text;
        va.n = n;
        va.q = dq;
        va.mask = q->d_mask;
        va.wide = q->wide;
        va.general = q->general;
        va.cand = (const uint64_t *)q->cand.p;
        va.wave_cand = (const uint32_t *)q->wave_cand.p;
        va.nw = (uint32_t)nw;
        va.wave_prefix = (const uint32_t *)q->wave_totals.p;
        va.dbm = d_dbm;
        va.gtab = q->tune.tight_verify ? q->d_gtab : nullptr;
        va.gram_spread = q->gram_spread;
        if (multi && q->multi_dense) {
            // dense hit set: probes and verification of the full strips in one kernel, nothing goes
            // through the slices but the partial last strip
            agh_launch_dense_multi(sa, multi_dev(q, d_dbm), va.mk, st);
            sa.tail_only = 1;
            agh_launch_sweep_multi(sa, st);
        } else if (multi) {
            // (sweep, then verify: verifying inside the sweep, verifying waves next to the sweeping ones and
            // the verifier of one part under the sweep of the next were all measured slower -- agh_multi.hip)
            agh_launch_sweep_multi(sa, st);
        } else {
            agh_launch_sweep(sa, q->fh, st);
        }
        if (multi) agh_launch_verify_multi(va, multi_dev(q, d_dbm), true, st);
        else agh_launch_verify_lean(va, st);
        agh_launch_resolve_giveups(d_text, dq.delim, va.mk, st);
        agh_launch_hashset_count((uint64_t *)q->hashset.p, (uint32_t)(q->hashset.cap / 8),
                                 (const uint32_t *)q->wave_cand.p, (uint32_t)nw, q->d_counters, st);
        HIP_TRY(hipGetLastError());
        if (flags & AGH_TIME_SCAN) HIP_TRY(hipEventRecord(q->ev1, st));
        HIP_TRY(hipMemcpyAsync(q->h_counters, q->d_counters, AGH_C_COUNT * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        q->hashset_dirty = false;
        if (multi && q->h_counters[AGH_C_OVERFLOW] && !q->multi_dense) {
            // dense hit set (many very short patterns): check hits inline from now on
            q->multi_dense = true;
            return scan_segment(q, d_text, n, st, flags, head_byte, tail_virtual, d_match_pos,
                                d_match_rec, match_cap, out, pre_dbm);
        }
        const bool gave_up = q->h_counters[AGH_C_LEAN_FALLBACK] || q->h_counters[AGH_C_OVERFLOW];
        if (!gave_up) {
            if (flags & AGH_TIME_SCAN) HIP_TRY(hipEventElapsedTime(&out->ms, q->ev0, q->ev1));
            if (sa.ev_begin) HIP_TRY(hipEventElapsedTime(&out->sweep_ms, q->ev2, q->ev3));
            out->matched = q->h_counters[AGH_C_MATCHED];
            out->candidates = q->h_counters[AGH_C_CAND];
            out->records = 0;                   // not computed by a lean scan
            out->engine = AGH_ENGINE_FILTER;
            q->hashset_slots_hint = 4ull * out->matched;
            return 0;
        }
        q->hashset_slots_hint = 8ull * (q->h_counters[AGH_C_MATCHED] + 1024);
        lean_rerun = 1;
        // fall through: the numbered pipeline is exact for every input
    }

    // ---- lean full scan: count-only scans of a query without a filter ------------------------
    // The same record identity (hash set of record starts) lets the automaton-over-every-byte
    // engine skip the census sweep: one pass over the text instead of two.
    const bool lean_fs_ok = !want_filter && !multi && !d_match_pos && !invert &&
                            (flags & (AGH_COUNT | AGH_FILENAMEONLY)) && !(flags & AGH_FORCE_NUMBERED) &&
                            !lean_rerun;
    if (lean_fs_ok) {
        uint64_t slots = 1u << 17;
        if (!q->hashset_slots_hint) while (slots < (n >> 13) && slots < (1u << 26)) slots <<= 1;
        while (slots < q->hashset_slots_hint) slots <<= 1;
        {
            const size_t cap_before = q->hashset.cap;
            if (q->hashset.ensure(slots * sizeof(uint64_t))) return -1;
            if (q->hashset.cap != cap_before || q->hashset_dirty)
                HIP_TRY(hipMemsetAsync(q->hashset.p, 0, q->hashset.cap, st));
            q->hashset_dirty = true;
        }
        if (flags & AGH_TIME_SCAN) HIP_TRY(hipEventRecord(q->ev0, st));
        HIP_TRY(hipMemsetAsync(q->d_counters, 0, AGH_C_COUNT * sizeof(uint32_t), st));
        agh_scan_args va;
        memset(&va, 0, sizeof(va));
        va.verify_blocks = q->tune.verify_blocks;
        va.text = d_text;
        va.n = n;
        va.q = dq;
        va.mask = q->d_mask;
        va.wide = q->wide;
        va.general = q->general;
        va.dbm = d_dbm;
        va.mk.counters = q->d_counters;
        va.mk.hashset = (uint64_t *)q->hashset.p;
        va.mk.hashset_mask = (uint32_t)(slots - 1);
        if (attach_giveups(q, &va.mk)) return -1;
        va.table = q->table;
        va.tab = q->tab;
        if (fs_fast_setup(q, n, &va)) return -1;
        const bool fs_fast = va.fs_fast != 0;
        if (q->table) agh_launch_tablescan(va, st);
        else agh_launch_fullscan(va, st);
        agh_launch_resolve_giveups(d_text, dq.delim, va.mk, st);
        agh_launch_hashset_count((uint64_t *)q->hashset.p, (uint32_t)(q->hashset.cap / 8), nullptr, 0u,
                                 q->d_counters, st);
        HIP_TRY(hipGetLastError());
        if (flags & AGH_TIME_SCAN) HIP_TRY(hipEventRecord(q->ev1, st));
        HIP_TRY(hipMemcpyAsync(q->h_counters, q->d_counters, AGH_C_COUNT * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        q->hashset_dirty = false;
        if (fs_fast && q->h_counters[AGH_C_OVERFLOW]) {
            // more pieces to replay than the lists hold (a match in every few records): this text
            // belongs to the kernel that does its bookkeeping on the way
            q->fs_fast_off = true;
            return scan_segment(q, d_text, n, st, flags, head_byte, tail_virtual, d_match_pos, d_match_rec,
                                match_cap, out, pre_dbm);
        }
        if (!q->h_counters[AGH_C_LEAN_FALLBACK]) {
            if (flags & AGH_TIME_SCAN) HIP_TRY(hipEventElapsedTime(&out->ms, q->ev0, q->ev1));
            out->matched = q->h_counters[AGH_C_MATCHED];
            out->records = 0;
            out->engine = AGH_ENGINE_FULLSCAN;
            q->hashset_slots_hint = 4ull * out->matched;
            return 0;
        }
        q->hashset_slots_hint = 8ull * (q->h_counters[AGH_C_MATCHED] + 1024);
        lean_rerun = 1;                         // numbered pipeline below: exact for every input
    }

    // Optimistic single-sync pipeline: the record bitmap is sized from a hint (the previous
    // scan of this query, or one record per 32 bytes) and everything -- census/filter sweep,
    // prefix scan, verify or full scan, population count -- is queued back to back.  The host
    // looks at the counters once at the end; only if a buffer turned out too small (candidate
    // slices: the filter is not selective on this text; bitmap: more records than guessed)
    // does it run the affected stage again.
    bool use_filter = want_filter;
    uint64_t bits_hint = std::max<uint64_t>(q->bitmap_bits_hint, n / 64 + 1024);
    bool swept = false;
    float total_ms = 0.f;
    for (int attempt = 0; attempt < 4; ++attempt) {
        const size_t bm_words = (size_t)(((bits_hint + 64 + 127) / 128) * 4);   // 16-byte units
        {
            const size_t cap_before = q->bitmap.cap;    // (a new block may reuse the old address)
            if (q->bitmap.ensure(bm_words * sizeof(uint32_t))) return -1;
            if (q->bitmap.cap != cap_before || q->bitmap_dirty) {
                // fresh allocation (or an aborted scan): zero everything once; afterwards
                // k_bitmap_count leaves the bitmap clean
                HIP_TRY(hipMemsetAsync(q->bitmap.p, 0, q->bitmap.cap, st));
            }
            q->bitmap_dirty = true;
        }

        if (flags & AGH_TIME_SCAN) HIP_TRY(hipEventRecord(q->ev0, st));
        if (!swept) HIP_TRY(hipMemsetAsync(q->d_counters, 0, AGH_C_COUNT * sizeof(uint32_t), st));
        else {
            // keep the census results (NDELIM, LASTBYTE, CAND), clear the rest
            HIP_TRY(hipMemsetAsync(q->d_counters + AGH_C_OVERFLOW, 0, sizeof(uint32_t), st));
            HIP_TRY(hipMemsetAsync(q->d_counters + AGH_C_MATCHED, 0, sizeof(uint32_t), st));
            HIP_TRY(hipMemsetAsync(q->d_counters + AGH_C_STORED, 0, sizeof(uint32_t), st));
            HIP_TRY(hipMemsetAsync(q->d_counters + AGH_C_BM_OVERFLOW, 0, sizeof(uint32_t), st));
        }
        if (!swept) {
            agh_sweep_args sa;
            sa.text = d_text;
            sa.n = n;
            sa.q = dq;
            sa.ftab = multi ? (const uint8_t *)q->d_mp_bits : q->d_ftab;
            sa.strip_prefix = (uint32_t *)q->strip_prefix.p;
            sa.wave_totals = (uint32_t *)q->wave_totals.p;
            sa.cand = (uint64_t *)q->cand.p;
            sa.wave_cand = (uint32_t *)q->wave_cand.p;
            sa.counters = q->d_counters;
            sa.chunk_totals = q->d_chunk_totals;
            sa.dbm = d_dbm;
            sa.lean = 0;
            sa.ev_begin = (flags & AGH_TIME_SWEEP) ? q->ev2 : nullptr;
            sa.ev_end = (flags & AGH_TIME_SWEEP) ? q->ev3 : nullptr;
            if (multi && q->multi_dense) {
                // census first (plain H=0 sweep + prefix scan), then the inline multi sweep
                // numbers records from that prefix and marks them directly
                agh_launch_sweep(sa, 0, st);
                agh_marks mk0;
                memset(&mk0, 0, sizeof(mk0));
                mk0.bitmap = (uint32_t *)q->bitmap.p;
                mk0.bitmap_bits = (uint32_t)std::min<uint64_t>(bm_words * 32, 0xffffffffu);
                mk0.counters = q->d_counters;
                mk0.match_pos = d_match_pos;
                mk0.match_rec = d_match_rec;
                mk0.match_cap = match_cap;
                sa.ev_begin = sa.ev_end = nullptr;
                agh_launch_dense_multi(sa, multi_dev(q, d_dbm), mk0, st);
                sa.tail_only = 1;
                agh_launch_sweep_multi(sa, st);
            } else if (multi) {
                agh_launch_sweep_multi(sa, st);
                agh_launch_census_scan(sa, true, st);
            } else {
                agh_launch_sweep(sa, use_filter ? q->fh : 0, st);
            }
            HIP_TRY(hipGetLastError());
        }
        agh_scan_args va;
        memset(&va, 0, sizeof(va));
        va.verify_blocks = q->tune.verify_blocks;
        va.text = d_text;
        va.n = n;
        va.q = dq;
        va.mask = q->d_mask;
        va.wide = q->wide;
        va.general = q->general;
        va.table = q->table;
        va.tab = q->tab;
        va.cand = (const uint64_t *)q->cand.p;
        va.wave_cand = (const uint32_t *)q->wave_cand.p;
        va.nw = (uint32_t)nw;
        va.strip_prefix = (const uint32_t *)q->strip_prefix.p;
        va.wave_prefix = (const uint32_t *)q->wave_totals.p;
        va.n_strips = (uint32_t)n_strips;
        va.dbm = d_dbm;
        va.mk.bitmap = (uint32_t *)q->bitmap.p;
        va.mk.bitmap_bits = (uint32_t)std::min<uint64_t>(bm_words * 32, 0xffffffffu);
        va.mk.counters = q->d_counters;
        va.mk.match_pos = invert_list ? nullptr : d_match_pos;     // -v: bits only, list below
        va.mk.match_rec = invert_list ? nullptr : d_match_rec;
        va.mk.match_cap = invert_list ? 0u : match_cap;
        va.mk.hashset = nullptr;
        va.mk.hashset_mask = 0;
        va.gtab = (q->tune.tight_verify && !multi) ? q->d_gtab : nullptr;
        va.gram_spread = q->gram_spread;
        if (!multi && !use_filter && fs_fast_setup(q, n, &va)) return -1;
        const bool fs_fast = va.fs_fast != 0;
        if (multi) agh_launch_verify_multi(va, multi_dev(q, d_dbm), false, st);
        else if (use_filter) agh_launch_verify(va, st);
        else if (q->table) agh_launch_tablescan(va, st);
        else agh_launch_fullscan(va, st);
        if (invert_list) {                      // the records whose bit stayed clear
            va.mk.match_pos = d_match_pos;
            va.mk.match_rec = d_match_rec;
            va.mk.match_cap = match_cap;
            agh_launch_unmatched(va, st);
        }
        agh_launch_bitmap_count((uint32_t *)q->bitmap.p, (uint32_t)(q->bitmap.cap / 4), q->d_counters, st);
        HIP_TRY(hipGetLastError());
        if (flags & AGH_TIME_SCAN) HIP_TRY(hipEventRecord(q->ev1, st));
        HIP_TRY(hipMemcpyAsync(q->h_counters, q->d_counters, AGH_C_COUNT * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        q->bitmap_dirty = false;
        float ms = 0.f;
        if (flags & AGH_TIME_SCAN) HIP_TRY(hipEventElapsedTime(&ms, q->ev0, q->ev1));
        total_ms += ms;
        if (!swept && (flags & AGH_TIME_SWEEP)) HIP_TRY(hipEventElapsedTime(&out->sweep_ms, q->ev2, q->ev3));
        swept = true;

        const uint32_t n_delims = q->h_counters[AGH_C_NDELIM];
        if (q->tune.debug)
            fprintf(stderr, "[agh] attempt %d multi %d dense %d filter %d: ndelim %u cand %u overflow %u "
                            "bm_overflow %u matched %u bm_bits %llu\n", attempt, (int)multi,
                    (int)q->multi_dense, (int)use_filter, n_delims, q->h_counters[AGH_C_CAND],
                    q->h_counters[AGH_C_OVERFLOW], q->h_counters[AGH_C_BM_OVERFLOW],
                    q->h_counters[AGH_C_MATCHED], (unsigned long long)bm_words * 32);
        q->bitmap_bits_hint = (uint64_t)n_delims + n_delims / 4 + 1024;
        if (fs_fast && q->h_counters[AGH_C_OVERFLOW]) {     // replay lists full: the exact kernel (see above)
            q->fs_fast_off = true;
            bits_hint = std::max<uint64_t>(bits_hint, (uint64_t)n_delims + 1024);
            continue;
        }
        const bool slice_overflow = use_filter && q->h_counters[AGH_C_OVERFLOW];
        const bool bm_overflow = q->h_counters[AGH_C_BM_OVERFLOW] != 0 ||
                                 (uint64_t)n_delims + 2 > (uint64_t)bm_words * 32;
        if (slice_overflow) {
            if (multi && !q->multi_dense) {
                q->multi_dense = true;          // dense hit set: check hits inline from now on
                swept = false;
                bits_hint = (uint64_t)n_delims + 1024;
                continue;
            }
            if ((flags & AGH_FORCE_FILTER) || multi)
                return fail("candidate slices overflowed (%u candidates)", q->h_counters[AGH_C_CAND]);
            use_filter = false;                 // not selective on this text: automaton everywhere
            swept = false;                      // the full scan needs the H=0 sweep's strip prefix
        }
        if (slice_overflow || bm_overflow) {
            bits_hint = (uint64_t)n_delims + 1024;
            if (multi && q->multi_dense) swept = false;     // the inline sweep does the marking
            continue;
        }
        out->records = (uint64_t)n_delims +
                       (q->h_counters[AGH_C_LASTBYTE] != q->delim[q->dlen - 1] ? 1u : 0u);
        out->candidates = use_filter ? q->h_counters[AGH_C_CAND] : 0;
        out->engine = use_filter ? AGH_ENGINE_FILTER : AGH_ENGINE_FULLSCAN;
        out->matched = q->h_counters[AGH_C_MATCHED];
        if (invert) out->matched = out->records - out->matched;
        out->stored = std::min<uint64_t>(q->h_counters[AGH_C_STORED], match_cap);
        out->truncated = q->h_counters[AGH_C_STORED] > match_cap;
        out->ms = total_ms;
        out->lean_rerun = lean_rerun;
        return 0;
    }
    return fail("internal error: scan did not converge");
}

// ---------------------------------------------------------------------------------------
// segments: inputs above the per-launch limit are cut where a record ends at a 16-byte aligned
// offset.  All cuts of a scan are found by one small kernel (k_find_cuts) and read back with one
// host sync.  Nominal boundaries sit `nominal` bytes apart; a cut lies in (previous boundary,
// boundary], so a segment is never longer than 2 x nominal (single patterns: 8 GiB nominal under
// the 16 GiB reach of the 32-bit dword index; -f / piece engine: 2 GiB under 32-bit byte offsets).
// ---------------------------------------------------------------------------------------
static uint64_t seg_nominal(const agh_query *q)
{
    if (q->tune.seg_max_mb) {                                       // AGH_SEG_MAX_MB (tests: tiny segments)
        // (H = 2 candidates are 32-bit halfword indices: 8 GiB reach, so 4 GiB nominal at most)
        const uint64_t mb = (q->fh == 2 && !q->multi && !q->piece_single) ? std::min<uint64_t>(q->tune.seg_max_mb, 4096)
                                                                         : q->tune.seg_max_mb;
        return mb << 20;
    }
    if (q->multi || q->piece_single) return (uint64_t)2 << 30;
    // H == 2 candidates are 32-bit HALFWORD indices in numbered scans: 8 GiB reach, 4 GiB nominal
    return q->fh == 2 ? ((uint64_t)4 << 30) : AGH_SEG_MAX_DEFAULT;
}

static int plan_segments(agh_query *q, const unsigned char *base, uint64_t len, hipStream_t st,
                         std::vector<uint64_t> *cuts, bool lean, const uint64_t *global_dbm)
{
    cuts->clear();
    cuts->push_back(0);
    // Lean (count-only) scans of a single pattern carry 64-bit dword indices in their candidate
    // entries and need no record numbers, so ONE kernel sequence covers up to 64 GiB (bounded only by
    // the candidate slices: 1/8 of the text); everything else is limited by 32-bit indices.
    uint64_t nominal = seg_nominal(q);
    if (lean && !q->tune.seg_max_mb) nominal = AGH_LEAN_SEG_MAX;
    if (len > nominal) {
        const uint64_t nb = (len - 1) / nominal;        // boundaries strictly inside the text
        if (nb + 1 > AGH_MAX_SEGS) return fail("input too large: more than %d segments", AGH_MAX_SEGS);
        if (q->cuts.ensure(3 * AGH_MAX_SEGS * sizeof(uint64_t))) return -1;
        uint64_t *h_bound = q->h_cuts, *h_lo = q->h_cuts + AGH_MAX_SEGS, *h_cut = q->h_cuts + 2 * AGH_MAX_SEGS;
        for (uint64_t i = 0; i < nb; ++i) {
            h_bound[i] = (i + 1) * nominal;             // multiples of 16: nominal is a MiB count
            h_lo[i] = i * nominal;
        }
        uint64_t *d = (uint64_t *)q->cuts.p;
        HIP_TRY(hipMemcpyAsync(d, h_bound, nb * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d + AGH_MAX_SEGS, h_lo, nb * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        // aligned cuts first (a segment that starts on a 16-byte -- bitmap delimiters: 64-byte --
        // boundary is scanned where it lies); where no record ends on such a boundary (fixed-width
        // records behind an odd header), any record end: that segment is copied to an aligned buffer
        auto find = [&](uint32_t step, uint64_t *host_out) -> int {
            if (global_dbm)
                agh_launch_find_cuts_dbm(global_dbm, d, d + AGH_MAX_SEGS, (uint32_t)nb, step, d + 2 * AGH_MAX_SEGS, st);
            else
                agh_launch_find_cuts(base, d, d + AGH_MAX_SEGS, (uint32_t)nb, q->delim[0], step, d + 2 * AGH_MAX_SEGS, st);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(host_out, d + 2 * AGH_MAX_SEGS, nb * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            return 0;
        };
        if (find(global_dbm ? 64u : 16u, h_cut)) return -1;
        bool missing = false;
        for (uint64_t i = 0; i < nb; ++i) missing = missing || !h_cut[i];
        if (missing && !q->tune.aligned_cuts_only) {
            std::vector<uint64_t> any(nb, 0);
            if (find(1u, any.data())) return -1;
            for (uint64_t i = 0; i < nb; ++i)
                if (!h_cut[i]) h_cut[i] = any[i];
        }
        for (uint64_t i = 0; i < nb; ++i) {
            if (!h_cut[i])
                return fail("no record ends between byte %llu and %llu (needed to cut an input above the "
                            "%llu-byte segment limit)", (unsigned long long)h_lo[i],
                            (unsigned long long)h_bound[i], (unsigned long long)nominal);
            cuts->push_back(h_cut[i]);
        }
    }
    cuts->push_back(len);
    return 0;
}

// ---------------------------------------------------------------------------------------
// lean pipeline over all segments of a count-only scan (-c, -l) of a single-pattern query with
// a sample filter and a one-byte delimiter -- the headline path.
//   * every segment is swept in PARTS (wave ranges of the same text, no record alignment
//     needed: candidates carry text offsets and the verifier reads the whole segment);
//   * AGH_OVERLAP: the verifier of part p runs on a second stream while part p+1 is swept
//     (candidate buffers and counter blocks alternate between two sets);
//   * nothing is read back until every segment is queued: one host sync per scan.  A segment
//     whose lean scan gave up (record start > 64 KiB back, hash set full, slice overflow) is
//     run again on the numbered pipeline afterwards; agh_result.lean_reruns counts those;
//   * AGH_FILENAMEONLY (-l, asearch.c:130-161 returns at the first match): parts grow from
//     64 MiB, the host looks at the hit flag after each and stops at the first part with a hit.
// ---------------------------------------------------------------------------------------
static uint64_t env_u64(const char *name, uint64_t dflt)
{
    const char *e = getenv(name);
    if (e && *e) return strtoull(e, nullptr, 10);
    return dflt;
}
static long env_long(const char *name)
{
    const char *e = getenv(name);
    return (e && *e) ? (long)strtoul(e, nullptr, 10) : -1;
}
static bool env_on(const char *name, bool dflt)
{
    const char *e = getenv(name);
    return e && *e ? e[0] != '0' : dflt;
}

// The environment switches, once per query (agh_tuning, agh_launch.h).
void agh_read_tuning(agh_tuning *t)
{
    t->live = env_on("AGH_ENV_LIVE", false);
    t->tight_verify = env_on("AGH_TIGHT_VERIFY", true);
    t->fs_fast = env_on("AGH_FS_FAST", true);
    t->tf_pack2 = env_on("AGH_TF_PACK2", true);
    t->tf_fast_min_mb = env_u64("AGH_TF_FAST_MIN_MB", AGH_TF_FAST_MIN_MB_DEFAULT);
    {
        const uint64_t c = env_u64("AGH_TF_CHUNK", AGH_TF_CHUNK_DEFAULT);
        t->tf_chunk = (c == 1024 || c == 2048 || c == 4096) ? (uint32_t)c : 0u;
    }
    t->fused = env_on("AGH_FUSED", AGH_FUSED_DEFAULT != 0);
    t->debug = getenv("AGH_DEBUG") != nullptr;
    t->aligned_cuts_only = getenv("AGH_ALIGNED_CUTS_ONLY") != nullptr;
    t->stream = env_u64("AGH_STREAM", 1) != 0;
    {
        const uint64_t mb = env_u64("AGH_SEG_MAX_MB", 0);
        t->seg_max_mb = (mb >= 1 && mb <= 8192) ? mb : 0;
    }
    t->part_mb = env_u64("AGH_PART_MB", AGH_PART_MB_DEFAULT);
    t->overlap = env_u64("AGH_OVERLAP", AGH_OVERLAP_DEFAULT) != 0;
    t->fused_min_mb = env_u64("AGH_FUSED_MIN_MB", AGH_FUSED_MIN_MB_DEFAULT);
    t->stream_seg_mb = std::max<uint64_t>(env_u64("AGH_STREAM_SEG_MB", 1024), 1);
    t->readers = (unsigned)env_u64("AGH_READERS", 0);
    t->fused_range_kb = env_long("AGH_FUSED_RANGE_KB");
    t->fused_tail_kb = env_long("AGH_FUSED_TAIL_KB");
    t->fused_tail_mb = env_long("AGH_FUSED_TAIL_MB");
    t->fused_blocks = env_long("AGH_FUSED_BLOCKS");
    t->verify_blocks = env_long("AGH_VERIFY_BLOCKS");
    t->giveup_cap = env_long("AGH_GIVEUP_CAP");
}

// CUs of the current device (the fused lean kernel launches persistent workgroups)
static uint32_t device_cus()
{
    static thread_local int cached_dev = -1;     // (agrep-hip --gpus N: one host thread per device)
    static thread_local uint32_t cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256u;
    if (dev != cached_dev) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cached = (uint32_t)v;
        cached_dev = dev;
    }
    return cached;
}

// (AGH_FUSED=0: count-only scans as two kernels -- k_sweep, then k_verify -- instead of the fused one; q->tune)

static bool lean_pipeline_ok(const agh_query *q, unsigned flags, bool want_list)
{
    const bool invert = (flags & AGH_INVERT) != 0;
    return q->fq > 0 && !q->multi && !q->table && !q_mb(q) && !want_list && !invert &&
           (flags & (AGH_COUNT | AGH_FILENAMEONLY)) &&
           !(flags & (AGH_FORCE_FULLSCAN | AGH_FORCE_NUMBERED)) && !q->general;
}

static int scan_segment(agh_query *q, const void *d_text, uint64_t n, hipStream_t st,
                        unsigned flags, uint32_t head_byte, int tail_virtual,
                        uint64_t *d_match_pos, uint32_t *d_match_rec, uint32_t match_cap,
                        seg_result *out, const uint64_t *pre_dbm = nullptr);


static int get_events(std::vector<hipEvent_t> &pool, size_t want, unsigned evflags)
{
    while (pool.size() < want) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, evflags));
        pool.push_back(e);
    }
    return 0;
}

static int lean_run(agh_query *q, const unsigned char *base, const std::vector<uint64_t> &cuts,
                    hipStream_t st, unsigned flags, agh_result *res, bool is_first, bool is_last)
{
    const int nseg = (int)cuts.size() - 1;
    const bool early = (flags & AGH_FILENAMEONLY) != 0;
    const bool timing = (flags & AGH_TIME_SWEEP) != 0;
    // part size: a multiple of 8 wave ranges (2 MiB) so verify workgroups never straddle parts
    const uint64_t part_unit = (uint64_t)AGH_WAVE_STRIPS * AGH_STRIP * 8u;
    uint64_t part_bytes = q->tune.part_mb << 20;
    part_bytes = part_bytes / part_unit * part_unit;
    uint64_t max_n = 0;
    for (int i = 0; i < nseg; ++i) max_n = std::max(max_n, cuts[i + 1] - cuts[i]);
    const bool overlap = q->tune.overlap && !early &&
                         (nseg > 1 || (part_bytes && part_bytes < max_n));
    const uint64_t max_strips = (max_n + AGH_STRIP - 1) >> AGH_STRIP_SHIFT;
    const uint64_t max_nw = (max_strips + AGH_WAVE_STRIPS - 1) / AGH_WAVE_STRIPS;

    // buffers for the largest segment, before anything is queued (a hipMalloc / hipFree in the
    // middle would serialise the streams)
    dev_buf *cand[2] = {&q->cand, overlap && nseg > 1 ? &q->cand_b : &q->cand};
    dev_buf *wcand[2] = {&q->wave_cand, overlap && nseg > 1 ? &q->wave_cand_b : &q->wave_cand};
    for (int b = 0; b < 2; ++b) {
        if (cand[b]->ensure(max_nw * AGH_SLICE_CAP * sizeof(uint64_t))) return -1;
        if (wcand[b]->ensure((max_nw + 8) * sizeof(uint32_t))) return -1;
    }
    if (q->strip_prefix.ensure(64)) return -1;          // (unused by lean sweeps, never NULL)
    if (q->wave_totals.ensure((max_nw + 8) * sizeof(uint32_t))) return -1;
    uint64_t slots = 1u << 17;
    if (!q->hashset_slots_hint) while (slots < (max_n >> 13) && slots < (1u << 26)) slots <<= 1;
    while (slots < q->hashset_slots_hint) slots <<= 1;
    {
        const size_t cap_before = q->hashset.cap;
        if (q->hashset.ensure(slots * sizeof(uint64_t))) return -1;
        if (q->hashset.cap != cap_before || q->hashset_dirty)
            HIP_TRY(hipMemsetAsync(q->hashset.p, 0, q->hashset.cap, st));
        q->hashset_dirty = true;
    }
    // work counters of the fused kernel: a line of their own each -- on the counters' line the
    // sweepers' ticket atomics would queue behind the verifier's ANYHIT stores
    const uint64_t fused_min = q->tune.fused_min_mb << 20;
    const bool may_fuse = !early && !overlap && !part_bytes && q->tune.fused && max_n >= fused_min &&
                          q->tune.tight_verify && q->d_gtab;
    if (may_fuse) {
        if (q->tickets.ensure((size_t)nseg * 256u)) return -1;
        HIP_TRY(hipMemsetAsync(q->tickets.p, 0, (size_t)nseg * 256u, st));
    }
    hipStream_t aux = st;
    if (overlap) {
        if (!q->aux_stream) HIP_TRY(hipStreamCreateWithFlags(&q->aux_stream, hipStreamNonBlocking));
        aux = q->aux_stream;
    }
    size_t n_dep = 0, n_time = 0;
    if (overlap && get_events(q->dep_events, 4, hipEventDisableTiming)) return -1;

    agh_dev_query dq;
    dq.m = q->m;
    dq.k = q->k;
    dq.delim = q->delim[0];
    dq.dlen = 1;
    memset(dq.dbytes, 0, sizeof(dq.dbytes));
    dq.dbytes[0] = q->delim[0];
    dq.dfold = 0;
    dq.mb = 0;
    dq.mp_q5 = 0;
    dq.guard = 0;
    dq.fq = q->fq;
    dq.fh = q->fh;
    dq.qmask = q->qmask;
    dq.fold = q->fold;
    dq.ci = dq.cs = dq.cd = 1;
    dq.no_err = q->no_err;

    struct seg_job { int first_time_ev, n_parts; bool done_early; };
    std::vector<seg_job> jobs((size_t)nseg);
    bool stop = false;
    uint64_t scanned = 0;
    int queued = 0;
    for (int i = 0; i < nseg && !stop; ++i, ++queued) {
        const unsigned char *text = base + cuts[i];
        const uint64_t n = cuts[i + 1] - cuts[i];
        const int slot = overlap ? (i & 1) : 0;
        uint32_t *d_cnt = q->d_counters + (size_t)(1 + slot) * AGH_C_COUNT;
        uint32_t *h_cnt = q->h_counters + (size_t)(1 + i) * AGH_C_COUNT;
        jobs[i].first_time_ev = (int)n_time;
        jobs[i].n_parts = 0;
        jobs[i].done_early = false;
        dq.head_byte = (i == 0 && is_first) ? '\n' : q->delim[0];
        dq.tail_virtual = (i == nseg - 1 && is_last) ? 1 : 0;
        const uint64_t n_strips = (n + AGH_STRIP - 1) >> AGH_STRIP_SHIFT;
        const uint32_t nw = (uint32_t)((n_strips + AGH_WAVE_STRIPS - 1) / AGH_WAVE_STRIPS);
        // the sweep of segment i reuses the buffers of segment i-2: its verifier must be done
        if (overlap && i >= 2) HIP_TRY(hipStreamWaitEvent(st, q->dep_events[2 + slot], 0));
        HIP_TRY(hipMemsetAsync(d_cnt, 0, AGH_C_COUNT * sizeof(uint32_t), st));

        agh_sweep_args sa;
        sa.text = text;
        sa.n = n;
        sa.q = dq;
        sa.ftab = q->d_ftab;
        sa.strip_prefix = (uint32_t *)q->strip_prefix.p;
        sa.wave_totals = (uint32_t *)q->wave_totals.p;
        sa.cand = (uint64_t *)cand[slot]->p;
        sa.wave_cand = (uint32_t *)wcand[slot]->p;
        sa.counters = d_cnt;
        sa.chunk_totals = q->d_chunk_totals;
        sa.dbm = nullptr;
        sa.lean = 1;
        agh_scan_args va;
        memset(&va, 0, sizeof(va));
        va.verify_blocks = q->tune.verify_blocks;
        va.mk.counters = d_cnt;
        va.mk.hashset = (uint64_t *)q->hashset.p;
        va.mk.hashset_mask = (uint32_t)(slots - 1);
        if (attach_giveups(q, &va.mk)) return -1;
        va.text = text;
        va.n = n;
        va.q = dq;
        va.mask = q->d_mask;
        va.wide = q->wide;
        va.general = 0;
        va.cand = (const uint64_t *)cand[slot]->p;
        va.wave_cand = (const uint32_t *)wcand[slot]->p;
        va.nw = nw;
        va.wave_prefix = (const uint32_t *)q->wave_totals.p;
        va.gtab = q->tune.tight_verify ? q->d_gtab : nullptr;
        va.gram_spread = q->gram_spread;

        // one kernel for sweep + verify where the query's shape has a fused instance; the partial
        // last strip (n % 1024 bytes) still goes through k_sweep_tail + k_verify
        bool fused = false;
        if (may_fuse && n >= fused_min) {
            agh_fused_args fa;
            fa.text = text;
            fa.n = n;
            fa.q = dq;
            fa.ftab = q->d_ftab;
            fa.mask = q->d_mask;
            fa.wide = q->wide;
            fa.gtab = va.gtab;
            fa.gram_spread = q->gram_spread;
            fa.mk = va.mk;
            fa.n_cu = device_cus();
            fa.ticket = (uint32_t *)((char *)q->tickets.p + (size_t)i * 256u);
            fa.tune = &q->tune;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (timing) {
                if (get_events(q->time_events, n_time + 2, hipEventDefault)) return -1;
                e0 = q->time_events[n_time];
                e1 = q->time_events[n_time + 1];
                HIP_TRY(hipEventRecord(e0, st));
            }
            fused = agh_launch_sweep_fused(fa, q->fh, st);
            if (fused) {
                res->fused_segments += 1;
                if (timing) {
                    HIP_TRY(hipEventRecord(e1, st));
                    n_time += 2;
                    ++jobs[i].n_parts;
                }
                HIP_TRY(hipMemsetAsync(wcand[slot]->p, 0, ((size_t)nw + 8) * sizeof(uint32_t), st));
                if (n & (AGH_STRIP - 1)) {
                    sa.tail_only = 1;
                    sa.w_begin = sa.w_end = 0;
                    sa.ev_begin = sa.ev_end = nullptr;
                    agh_launch_sweep(sa, q->fh, st);
                    va.w_begin = (uint32_t)((n >> AGH_STRIP_SHIFT) / AGH_WAVE_STRIPS) & ~7u;
                    va.w_end = 0;
                    agh_launch_verify_lean(va, st);
                }
                HIP_TRY(hipGetLastError());
            }
        }
        uint64_t pb = early ? ((uint64_t)64 << 20) : part_bytes;
        for (uint64_t off = fused ? n : 0; off < n;) {
            const uint64_t part_end = (pb && off + pb < n) ? off + pb : n;
            const bool last = part_end == n;
            sa.w_begin = va.w_begin = (uint32_t)(off / ((uint64_t)AGH_WAVE_STRIPS * AGH_STRIP));
            sa.w_end = va.w_end = last ? 0u : (uint32_t)(part_end / ((uint64_t)AGH_WAVE_STRIPS * AGH_STRIP));
            sa.ev_begin = sa.ev_end = nullptr;
            if (timing) {
                if (get_events(q->time_events, n_time + 2, hipEventDefault)) return -1;
                sa.ev_begin = q->time_events[n_time];
                sa.ev_end = q->time_events[n_time + 1];
                n_time += 2;
            }
            agh_launch_sweep(sa, q->fh, st);
            ++jobs[i].n_parts;
            if (overlap) {                      // the verifier waits for this part's candidates
                hipEvent_t e = q->dep_events[n_dep & 1];
                ++n_dep;
                HIP_TRY(hipEventRecord(e, st));
                HIP_TRY(hipStreamWaitEvent(aux, e, 0));
            }
            agh_launch_verify_lean(va, aux);
            HIP_TRY(hipGetLastError());
            off = part_end;
            if (early) {
                HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt, AGH_C_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                if (h_cnt[AGH_C_ANYHIT] || h_cnt[AGH_C_LEAN_FALLBACK] || h_cnt[AGH_C_OVERFLOW]) {
                    stop = true;
                    jobs[i].done_early = !last;
                    scanned += off;
                    break;
                }
                if (pb < ((uint64_t)4 << 30)) pb <<= 1;
            }
        }
        if (!stop) scanned += n;
        agh_launch_resolve_giveups(text, dq.delim, va.mk, aux);
        agh_launch_hashset_count((uint64_t *)q->hashset.p, (uint32_t)(q->hashset.cap / 8),
                                 (const uint32_t *)wcand[slot]->p, nw, d_cnt, aux);
        if (q->reduce_comm) agh_launch_accumulate_counts(d_cnt, q->d_acc, aux);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt, AGH_C_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, aux));
        if (overlap) HIP_TRY(hipEventRecord(q->dep_events[2 + slot], aux));
    }
    if (overlap) {                              // later work on the caller's stream comes after the verifier
        const int last_slot = (queued - 1) & 1;
        HIP_TRY(hipStreamWaitEvent(st, q->dep_events[2 + last_slot], 0));
        if (queued >= 2) HIP_TRY(hipStreamWaitEvent(st, q->dep_events[2 + (last_slot ^ 1)], 0));
    }
    if (q->reduce_comm && !stop) {
        // the counts never leave the device before they are summed over the ranks: the all-reduce is queued
        // behind the last segment's kernels, the host waits once for everything
        if (agh_comm_allreduce_dev(q->reduce_comm, q->d_acc, 3, st)) return -1;
        HIP_TRY(hipMemcpyAsync(q->h_acc, q->d_acc, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        q->reduce_done = true;
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (overlap) HIP_TRY(hipStreamSynchronize(aux));
    q->hashset_dirty = false;

    res->n_bytes = scanned;
    res->engine = AGH_ENGINE_FILTER;
    uint64_t max_matched = 0;
    for (int i = 0; i < queued; ++i) {
        const uint32_t *h = q->h_counters + (size_t)(1 + i) * AGH_C_COUNT;
        const bool gave_up = h[AGH_C_LEAN_FALLBACK] || h[AGH_C_OVERFLOW];
        if (timing)
            for (int p = 0; p < jobs[i].n_parts; ++p) {
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, q->time_events[jobs[i].first_time_ev + 2 * p],
                                            q->time_events[jobs[i].first_time_ev + 2 * p + 1]));
                res->sweep_ms += ms;
                res->sweep_launches += 1;
            }
        if (!gave_up) {
            res->n_matched += h[AGH_C_MATCHED];
            res->n_candidates += h[AGH_C_CAND];
            max_matched = std::max<uint64_t>(max_matched, h[AGH_C_MATCHED]);
            continue;
        }
        // exact for every input: the numbered pipeline on the whole segment (in pieces of its own
        // size limit)
        max_matched = std::max<uint64_t>(max_matched, 2ull * (h[AGH_C_MATCHED] + 1024));
        q->hashset_slots_hint = 4ull * max_matched;
        agh_result rr;
        if (agh_scan_device_impl(q, base + cuts[i], cuts[i + 1] - cuts[i], st, flags | AGH_FORCE_NUMBERED, &rr,
                             nullptr, nullptr, 0, i == 0 && is_first, i == nseg - 1 && is_last))
            return -1;
        if (jobs[i].done_early) res->n_bytes = cuts[i + 1];    // the rerun read the whole segment
        res->n_matched += rr.n_matched;
        res->n_candidates += rr.n_candidates;
        res->lean_reruns += 1;
    }
    res->n_segments = (uint32_t)queued;
    q->hashset_slots_hint = 4ull * max_matched;
    return 0;
}

// ---------------------------------------------------------------------------------------
// count-only (-c, -l) scans of a pattern file whose set the one-pass kernel takes (agh_mscan.hip,
// tables: fill_multi_tables): one launch per segment of up to 64 GiB, nothing read back until every
// segment is queued.  A segment whose scan gave up (a record start more than AGH_LEAN_BACK_CAP bytes
// in front of a match, hash set full) is run again on the numbered two-kernel pipeline.
// ---------------------------------------------------------------------------------------
static bool mscan_applies(const agh_query *q, unsigned flags, bool want_list)
{
    return q->multi && q->ms_ok && !want_list && (flags & (AGH_COUNT | AGH_FILENAMEONLY)) &&
           !(flags & (AGH_INVERT | AGH_FORCE_FULLSCAN | AGH_FORCE_NUMBERED));
}

static int mscan_run(agh_query *q, const unsigned char *base, const std::vector<uint64_t> &cuts,
                     hipStream_t st, unsigned flags, agh_result *res, bool is_first, bool is_last)
{
    const int nseg = (int)cuts.size() - 1;
    const bool timing = (flags & (AGH_TIME_SWEEP | AGH_TIME_SCAN)) != 0;
    uint64_t max_n = 0;
    for (int i = 0; i < nseg; ++i) max_n = std::max(max_n, cuts[i + 1] - cuts[i]);
    uint64_t slots = 1u << 17;
    if (!q->hashset_slots_hint) while (slots < (max_n >> 13) && slots < (1u << 26)) slots <<= 1;
    while (slots < q->hashset_slots_hint) slots <<= 1;
    {
        const size_t cap_before = q->hashset.cap;
        if (q->hashset.ensure(slots * sizeof(uint64_t))) return -1;
        if (q->hashset.cap != cap_before || q->hashset_dirty)
            HIP_TRY(hipMemsetAsync(q->hashset.p, 0, q->hashset.cap, st));
        q->hashset_dirty = true;
    }
    if (q->tickets.ensure((size_t)nseg * 256u)) return -1;
    HIP_TRY(hipMemsetAsync(q->tickets.p, 0, (size_t)nseg * 256u, st));
    if (timing && get_events(q->time_events, 3 * (size_t)nseg, hipEventDefault)) return -1;

    agh_dev_query dq;
    memset(&dq, 0, sizeof(dq));
    dq.m = q->m;
    dq.k = q->k;
    dq.delim = q->delim[0];
    dq.dlen = 1;
    dq.dbytes[0] = q->delim[0];
    dq.mp_q5 = q->mp_q5 ? 1u : 0u;
    dq.fq = q->fq;
    dq.fh = q->mp_stride;
    dq.qmask = q->qmask;
    dq.fold = q->fold;
    dq.ci = dq.cs = dq.cd = 1;
    dq.no_err = q->no_err;
    uint32_t *d_cnt = q->d_counters + AGH_C_COUNT;      // (block 0 belongs to scan_segment)
    for (int i = 0; i < nseg; ++i) {
        uint32_t *h_cnt = q->h_counters + (size_t)(1 + i) * AGH_C_COUNT;
        dq.head_byte = (i == 0 && is_first) ? '\n' : q->delim[0];
        dq.tail_virtual = (i == nseg - 1 && is_last) ? 1 : 0;
        HIP_TRY(hipMemsetAsync(d_cnt, 0, AGH_C_COUNT * sizeof(uint32_t), st));
        agh_mscan_args a;
        a.text = base + cuts[i];
        a.n = cuts[i + 1] - cuts[i];
        a.q = dq;
        a.ms.ptab = (const uint2 *)q->d_ms_ptab;
        a.ms.gtab = (const uint32_t *)q->d_ms_gtab;
        a.ms.mdir = (const uint32_t *)q->d_ms_mdir;
        a.ms.ment = (const uint4 *)q->d_ms_ment;
        a.ms.rb = q->ms_rb;
        a.mt = multi_dev(q, nullptr);
        memset(&a.mk, 0, sizeof(a.mk));
        a.mk.counters = d_cnt;
        a.mk.hashset = (uint64_t *)q->hashset.p;
        a.mk.hashset_mask = (uint32_t)(slots - 1);
        if (attach_giveups(q, &a.mk)) return -1;
        a.ticket = (uint32_t *)((char *)q->tickets.p + (size_t)i * 256u);
        a.n_cu = device_cus();
        a.dbg = q->ms_dbg;
        if (timing) HIP_TRY(hipEventRecord(q->time_events[3 * i], st));
        if (!agh_launch_mscan(a, st)) return fail("internal error: no one-pass kernel for this pattern set");
        if (timing) HIP_TRY(hipEventRecord(q->time_events[3 * i + 1], st));
        agh_launch_resolve_giveups(a.text, dq.delim, a.mk, st);
        agh_launch_hashset_count((uint64_t *)q->hashset.p, (uint32_t)(q->hashset.cap / 8), nullptr, 0u, d_cnt, st);
        if (q->reduce_comm) agh_launch_accumulate_counts(d_cnt, q->d_acc, st);
        HIP_TRY(hipGetLastError());
        if (timing) HIP_TRY(hipEventRecord(q->time_events[3 * i + 2], st));
        HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt, AGH_C_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    }
    if (q->reduce_comm) {
        if (agh_comm_allreduce_dev(q->reduce_comm, q->d_acc, 3, st)) return -1;
        HIP_TRY(hipMemcpyAsync(q->h_acc, q->d_acc, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        q->reduce_done = true;
    }
    HIP_TRY(hipStreamSynchronize(st));
    q->hashset_dirty = false;

    res->engine = AGH_ENGINE_FILTER;
    uint64_t max_matched = 0;
    for (int i = 0; i < nseg; ++i) {
        const uint32_t *h = q->h_counters + (size_t)(1 + i) * AGH_C_COUNT;
        if (timing) {
            float ms = 0.f, all = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, q->time_events[3 * i], q->time_events[3 * i + 1]));
            HIP_TRY(hipEventElapsedTime(&all, q->time_events[3 * i], q->time_events[3 * i + 2]));
            res->sweep_ms += ms;               // the one kernel that reads every byte (+ its edge positions)
            res->device_ms += all;
            res->sweep_launches += 1;
        }
        if (!(h[AGH_C_LEAN_FALLBACK] || h[AGH_C_OVERFLOW])) {
            res->n_matched += h[AGH_C_MATCHED];
            res->n_candidates += h[AGH_C_CAND];
            res->fused_segments += 1;
            max_matched = std::max<uint64_t>(max_matched, h[AGH_C_MATCHED]);
            continue;
        }
        max_matched = std::max<uint64_t>(max_matched, 2ull * (h[AGH_C_MATCHED] + 1024));
        agh_result rr;
        if (agh_scan_device_impl(q, base + cuts[i], cuts[i + 1] - cuts[i], st, flags | AGH_FORCE_NUMBERED, &rr,
                             nullptr, nullptr, 0, i == 0 && is_first, i == nseg - 1 && is_last))
            return -1;
        res->n_matched += rr.n_matched;
        res->n_candidates += rr.n_candidates;
        res->lean_reruns += 1;
    }
    res->n_segments = (uint32_t)nseg;
    q->hashset_slots_hint = 4ull * max_matched;
    return 0;
}

int agh_scan_device_impl(agh_query *q, const void *dev_text, uint64_t len, hipStream_t st,
                         unsigned flags, agh_result *res, uint64_t *d_match_pos,
                         uint32_t *d_match_rec, size_t match_cap, bool is_first, bool is_last)
{
    if (!q || !res) return fail("null argument");
    memset(res, 0, sizeof(*res));
    res->n_bytes = len;
    if (!len) return 0;
    if (((uintptr_t)dev_text & 15u) != 0) return fail("device text must be 16-byte aligned");
    const unsigned char *base = (const unsigned char *)dev_text;
    std::vector<uint64_t> cuts;
    const bool lean = lean_pipeline_ok(q, flags, d_match_pos != nullptr);
    // Delimiters that come from the delimiter bitmap and a text above one segment: the bitmap is
    // built once for the whole text (a selected occurrence depends on the ones in front of it, not on
    // where a segment starts); every segment then works on its part of it.
    const uint64_t *global_dbm = nullptr;
    if (q_mb(q) && len > seg_nominal(q)) {
        const uint64_t n_words = (len + 63) / 64 + 4;
        if (q->dbm.ensure(n_words * sizeof(uint64_t))) return -1;
        agh_dev_query dq;
        memset(&dq, 0, sizeof(dq));
        dq.delim = q->delim[q->dlen - 1];
        dq.dlen = (uint32_t)q->dlen;
        memcpy(dq.dbytes, q->delim, (size_t)q->dlen);
        dq.dfold = q->delim_fold ? 1u : 0u;
        dq.mb = 1u;
        dq.head_byte = is_first ? '\n' : q->delim[q->dlen - 1];
        HIP_TRY(hipMemsetAsync(q->d_counters, 0, AGH_C_COUNT * sizeof(uint32_t), st));
        agh_launch_delim_bitmap(base, len, dq, (uint64_t *)q->dbm.p, n_words, q->d_counters, st);
        HIP_TRY(hipMemcpyAsync(q->h_counters, q->d_counters, AGH_C_COUNT * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (q->h_counters[AGH_C_DELIM_CHAIN])
            return fail("a run of overlapping delimiter occurrences exceeds 4 KiB (unsupported)");
        global_dbm = (const uint64_t *)q->dbm.p;
    }
    const bool ms = mscan_applies(q, flags, d_match_pos != nullptr);
    if (plan_segments(q, base, len, st, &cuts, lean || ms, global_dbm)) return -1;
    if (lean || ms) {
        bool aligned = true;                    // (the segment STARTS: the last entry is the end of the text)
        for (size_t i = 0; i + 1 < cuts.size(); ++i) aligned = aligned && (cuts[i] & 15u) == 0;
        if (aligned && ms) return mscan_run(q, base, cuts, st, flags, res, is_first, is_last);
        if (aligned) return lean_run(q, base, cuts, st, flags, res, is_first, is_last);
        // a cut that is not 16-byte aligned: segment by segment through aligned copies (below), in
        // the segment sizes of that path
        if (plan_segments(q, base, len, st, &cuts, false, global_dbm)) return -1;
    }
    for (size_t i = 0; i + 1 < cuts.size(); ++i) {
        const uint64_t off = cuts[i], end = cuts[i + 1];
        seg_result sr;
        uint64_t stored = res->n_stored;
        uint32_t cap_left = (uint32_t)std::min<uint64_t>(match_cap - stored, 0xffffffffu);
        // a segment that does not start on an aligned offset (no record ended on one near the
        // boundary: plan_segments) is scanned from an aligned copy, with a delimiter bitmap of its own
        const unsigned char *seg_text = base + off;
        const uint64_t *seg_dbm = global_dbm ? global_dbm + off / 64 : nullptr;
        if (off & (global_dbm ? 63u : 15u)) {
            if (q->seg_copy.ensure(end - off + 64)) return -1;
            HIP_TRY(hipMemcpyAsync(q->seg_copy.p, base + off, end - off, hipMemcpyDeviceToDevice, st));
            seg_text = (const unsigned char *)q->seg_copy.p;
            seg_dbm = nullptr;
            q->seg_dbm_active = global_dbm != nullptr;
            res->copied_segments += 1;
        }
        const int seg_rc = scan_segment(q, seg_text, end - off, st, flags,
                                        (i == 0 && is_first) ? '\n' : q->delim[q->dlen - 1], end == len && is_last,
                                        d_match_pos ? d_match_pos + stored : nullptr,
                                        d_match_rec ? d_match_rec + stored : nullptr,
                                        d_match_pos ? cap_left : 0, &sr, seg_dbm);
        q->seg_dbm_active = false;
        if (seg_rc) return -1;
        if (d_match_pos && sr.stored && off > 0) {
            // the segment's kernels saw positions / record numbers relative to its own start
            agh_launch_offset_matches(d_match_pos + stored, d_match_rec ? d_match_rec + stored : nullptr,
                                      (uint32_t)sr.stored, off, (uint32_t)res->n_records, st);
            HIP_TRY(hipGetLastError());
        }
        res->n_matched += sr.matched;
        res->n_records += sr.records;
        res->n_candidates += sr.candidates;
        res->n_stored += sr.stored;
        res->truncated |= sr.truncated;
        res->device_ms += sr.ms;
        res->sweep_ms += sr.sweep_ms;
        if (sr.sweep_ms > 0.f) res->sweep_launches += 1;
        res->lean_reruns += sr.lean_rerun;
        res->engine = sr.engine;
        res->n_segments += 1;
    }
    return 0;
}

extern "C" int agh_scan_device(agh_query *q, const void *dev_text, size_t len, void *stream,
                               unsigned flags, agh_result *res, void *dev_match_pos,
                               size_t match_cap)
{
    if (q) agh_refresh_tuning(q);
    return agh_scan_device_impl(q, dev_text, len, (hipStream_t)stream, flags, res,
                            (uint64_t *)dev_match_pos, nullptr, dev_match_pos ? match_cap : 0, true, true);
}

// One step of a sharded count-only scan (the only exchange is the -c aggregate): the scan of this
// rank's shard and the all-reduce of the counts.  On the count-only pipelines the per-segment counts are
// summed on the device and ncclAllReduce is enqueued on the scan's stream right behind the kernels -- the
// host waits ONCE per step (round 3: scan sync, H2D of 16 bytes, all-reduce, D2H, sync).  Every rank issues
// the same collectives whatever path its own scan took: one all-reduce of (matched, records, gave-up), and
// -- only if some rank's count-only scan gave up and was rerun -- a second one with the final counts.
extern "C" int agh_scan_device_reduce(agh_query *q, agh_comm *c, const void *dev_text, size_t len, void *stream,
                                      unsigned flags, agh_result *res, uint64_t totals[2])
{
    if (!q || !c || !res || !totals) return fail("null argument");
    agh_refresh_tuning(q);
    if (!q->d_acc) {
        HIP_TRY(hipMalloc((void **)&q->d_acc, 4 * sizeof(uint64_t)));
        HIP_TRY(hipHostMalloc((void **)&q->h_acc, 4 * sizeof(uint64_t)));
    }
    HIP_TRY(hipMemsetAsync(q->d_acc, 0, 4 * sizeof(uint64_t), (hipStream_t)stream));
    q->reduce_comm = c;
    q->reduce_done = false;
    const int rc = agh_scan_device_impl(q, dev_text, len, (hipStream_t)stream, flags, res, nullptr, nullptr, 0, true, true);
    q->reduce_comm = nullptr;
    if (rc) return -1;
    uint64_t v[3] = {res->n_matched, res->n_records, 0};
    if (q->reduce_done) {
        v[0] = q->h_acc[0];
        v[1] = q->h_acc[1];
        v[2] = q->h_acc[2];
    } else if (agh_comm_allreduce_host(c, v, 3)) {      // (a path without the device-side sum: same collective)
        return -1;
    }
    if (v[2]) {                                         // some rank reran a segment: the final counts, once more
        v[0] = res->n_matched;
        v[1] = res->n_records;
        if (agh_comm_allreduce_host(c, v, 2)) return -1;
    }
    totals[0] = v[0];
    totals[1] = v[1];
    return 0;
}

// ---------------------------------------------------------------------------------------
// bench / test support
// ---------------------------------------------------------------------------------------
extern "C" int agh_corpus_fill_device(void *dev_out, uint64_t first_page, uint64_t n_pages,
                                      uint64_t seed, const unsigned char *variants,
                                      const uint32_t *vlen, uint32_t n_variants,
                                      uint32_t plant_period, uint32_t upper_permille,
                                      uint64_t *planted, void *stream)
{
    if (n_variants > 8) return fail("at most 8 variants");
    for (uint32_t i = 0; i < n_variants; ++i)
        if (vlen[i] > 80) return fail("variant %u longer than 80 bytes", i);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *d_planted = nullptr;
    HIP_TRY(hipMalloc((void **)&d_planted, 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(d_planted, 0, 8 * sizeof(unsigned long long), st));
    agh_launch_corpus(dev_out, first_page, n_pages, seed, variants, vlen, n_variants,
                      plant_period, upper_permille, d_planted, st);
    hipError_t e = hipGetLastError();
    unsigned long long h[8] = {0};
    if (e == hipSuccess)
        e = hipMemcpyAsync(h, d_planted, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_planted);
    if (e != hipSuccess) return fail("corpus generation failed: %s", hipGetErrorString(e));
    if (planted)
        for (int i = 0; i < 8; ++i) planted[i] = h[i];
    return 0;
}

// Streaming-read ceiling of the sweep's access pattern (bench support).  With -DAGH_WITH_EXP
// (make EXP=1) the library also carries the structural A/B variants of the sweep (agh_exp.hip)
// behind agh_probe_variant_ms -- diagnostics, not part of the product ABI or its header.
static int probe_ms(const void *dev_text, size_t len, void *stream, int exp, double *ms)
{
    hipStream_t st = (hipStream_t)stream;
    uint32_t *d_c = nullptr;
    hipEvent_t a, b;
    HIP_TRY(hipMalloc((void **)&d_c, AGH_C_COUNT * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(d_c, 0, AGH_C_COUNT * sizeof(uint32_t), st));
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    HIP_TRY(hipEventRecord(a, st));
#ifdef AGH_WITH_EXP
    if (exp >= 0) agh_launch_exp(exp, dev_text, len, d_c, st);
    else
#endif
        agh_launch_read_probe(dev_text, len, d_c, st);
    (void)exp;
    HIP_TRY(hipEventRecord(b, st));
    HIP_TRY(hipStreamSynchronize(st));
    float f = 0;
    HIP_TRY(hipEventElapsedTime(&f, a, b));
    if (ms) *ms = f;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(d_c);
    return 0;
}

extern "C" int agh_probe_read_ms(const void *dev_text, size_t len, void *stream, double *ms)
{
    return probe_ms(dev_text, len, stream, -1, ms);
}

#ifdef AGH_WITH_EXP
extern "C" int agh_probe_variant_ms(const void *dev_text, size_t len, void *stream, int exp,
                                    double *ms)
{
    return probe_ms(dev_text, len, stream, exp, ms);
}
#endif
