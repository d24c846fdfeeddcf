// agh_query.cpp -- query compilation of libagrep_hip.so (include/agrep_hip.h): errors, devices, the sample
// filter's shape (choose_filter), table uploads, and the builders -- agh_query_literal[_ex] (maskgen.c for
// literals, sgrep.c's guards), agh_query_from_maskgen (the reference's own tables; table engine), agh_query_multi*
// (prepf(), newmgrep.c:192-375, and the tables of the one-pass -f kernel), costs, free, info.  The pattern
// language: agh_pattern.cpp.  Scans: agh_api.cpp (resident text), agh_stage.cpp (buffers, files, records).
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

void agh_timeline(const char *what)
{
    static int on = -1;
    static double t0 = 0;
    if (on < 0) {
        const char *e = getenv("AGH_TIMELINE");
        on = (e && e[0] == '1') ? 1 : 0;
    }
    if (!on) return;
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    const double t = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    if (t0 == 0) t0 = t;
    fprintf(stderr, "[agh %8.2f ms] %s\n", t - t0, what);
}

extern "C" int agh_device_count(void)
{
    int n = 0;
    agh_timeline("hipGetDeviceCount ...");
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    agh_timeline("... runtime is up");
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
        // Round 5, delimiters of several bytes: WHICH positions a byte outside the text can reach.  The bytes of a real
        // delimiter occurrence are text like any other (sampled, and the verifier takes the verdict at the
        // occurrence's last byte as asearch.c does), so `-d 'From '` need not cost every pattern with an r, o or m in
        // it its filter.  Outside the text stand (a) the byte in front of a segment -- '\n' or the last delimiter byte,
        // re-fed after the reset (asearch.c:175-186) -- which only the first k + 1 positions can take, and (b) the dlen
        // bytes appended at the end, which an occurrence that ends inside them takes with its last positions: position
        // p sits at most (m - 1 - p) + k bytes in front of the occurrence's end, so p < m - dlen - k never gets there.
        // A one-byte delimiter keeps the old rule (every position).
        const bool positional = q->dlen > 1;
        for (uint8_t c : raw) {
            bool virt = c == '\n';
            for (int j = 0; j < q->dlen; ++j) {
                const bool is_d = c == q->delim[j] || (q->delim_fold && (c | 0x20) == q->delim[j]);
                if (!is_d) continue;
                if (!positional) virt = true;
                else if (p >= q->m - q->dlen - q->k) virt = true;                  // (b)
                else if (j == q->dlen - 1 && p <= q->k) virt = true;               // (a)
            }
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
    agh_timeline("query: device buffers, pinned counters, events ...");
    HIP_TRY(hipMalloc((void **)&q->d_counters, (AGH_LEAN_SLOTS + 1) * AGH_C_COUNT * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&q->d_chunk_totals, 128 * sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc((void **)&q->h_counters, (AGH_MAX_SEGS + 1) * AGH_C_COUNT * sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc((void **)&q->h_cuts, 3 * AGH_MAX_SEGS * sizeof(uint64_t)));
    HIP_TRY(hipEventCreate(&q->ev0));
    HIP_TRY(hipEventCreate(&q->ev1));
    HIP_TRY(hipEventCreate(&q->ev2));
    HIP_TRY(hipEventCreate(&q->ev3));
    agh_timeline("... query: common resources done");
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
        // entries: one per gram slot (the gram's first entry, its gram word replaced by (index of the further entries with
        // that gram << 8) | their number), then the further entries in the old 16-byte form
        std::vector<uint32_t> ptab((size_t)2 << rb, 0), gtab(AGH_MS_GSLOTS, 0);
        std::vector<uint32_t> ment((AGH_MS_GSLOTS + es.size()) * 4 + 4, 0);
        size_t n_more = 0;
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
            memcpy(&ment[4 * (size_t)sl], es[a].w, 16);
            ment[4 * (size_t)sl] = (uint32_t)n_more << 8 | (uint32_t)(b - a - 1);
            for (size_t i = a + 1; i < b; ++i, ++n_more) memcpy(&ment[4 * (AGH_MS_GSLOTS + n_more)], es[i].w, 16);
            a = b;
        }
        if (es.size() >= (1u << 24)) ms = false;
        if (ms) {
            if (up(&q->d_ms_ptab, ptab.data(), ptab.size() * 4) || up(&q->d_ms_gtab, gtab.data(), gtab.size() * 4) ||
                up(&q->d_ms_ment, ment.data(), ment.size() * 4))
                return -1;
            q->ms_ok = true;
            q->ms_rb = rb;
            const char *dbg = getenv("AGH_MSCAN_DBG");
            q->ms_dbg = dbg ? (uint32_t)strtoul(dbg, nullptr, 0) : 0u;
        }
    }
    // ---- tables of the tile kernel for dense sets (agh_mtile.hip): one error, patterns of 4..14 bytes some of which are too short
    // for the one-pass kernel's 4-byte grams (pieces of 2..3 bytes: every position is a candidate) ---------------
    q->mw_ok = false;
    // (sets whose pieces all have >= 4 bytes are selective: the filter kernels are several times faster there)
    bool mw = D == 1 && q->multi && q->dlen == 1 && !q->delim_fold && !q->guard && !q->ms_ok && minlen >= 2 && minlen < 4;
    {
        const char *e = getenv("AGH_MTILE");      // 0: such sets stay on k_dense_multi (A/B; read when the query is built)
        if (e && e[0] == '0') mw = false;
    }
    for (int p = 0; p < npat && mw; ++p) mw = lens[p] >= 4 && lens[p] <= 14;
    if (mw) {
        struct mw_entry { uint32_t slot; uint32_t w[4]; };    // slot: of the entry directory
        std::vector<mw_entry> es;
        for (int i = 0; i < npc; ++i) {
            if (!usable[i]) continue;
            const int m = lens[pcs[i].owner], po = pcs[i].po, len = pcs[i].len;
            const bool before = po > 0;
            const int L = before ? po : m - len;
            if (len > 7 || L > 7 || L < 1) { mw = false; break; }
            auto pb = [&](int t) -> uint32_t { return t < len ? pool[off[i] + t] : 0u; };
            uint8_t B[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int t = 0; t < L; ++t) {           // the other side, nearest byte first, in the case the pool has
                unsigned char c = before ? pats[pcs[i].owner][po - 1 - t] : pats[pcs[i].owner][len + t];
                if (nocase && is_upper(c)) c += 32;
                B[t] = c;
            }
            mw_entry e;
            e.w[0] = pb(0) | pb(1) << 8 | pb(2) << 16 | pb(3) << 24;
            e.w[1] = pb(4) | pb(5) << 8 | pb(6) << 16 | (uint32_t)len << 24;
            e.w[2] = B[0] | B[1] << 8 | B[2] << 16 | (uint32_t)B[3] << 24;
            e.w[3] = B[4] | B[5] << 8 | B[6] << 16 | ((uint32_t)L | (before ? 8u : 0u)) << 24;
            // the directory's slot: a longer piece under its first three bytes, a piece of two under the pair
            e.slot = len >= 3 ? agh_mw_slot3(e.w[0] & 0xffffffu) : agh_mw_slot(e.w[0] & 0xffffu);
            es.push_back(e);
        }
        if (es.empty() || es.size() > AGH_MW_MAX_ENT) mw = false;
        if (mw) {
            std::stable_sort(es.begin(), es.end(), [](const mw_entry &a, const mw_entry &b) { return a.slot < b.slot; });
            std::vector<uint32_t> dir(AGH_MW_DIR, 0), ent(es.size() * 4), fmask((size_t)AGH_MW_DIR * 4, 0), g4(AGH_MW_G4_WORDS, 0);
            for (size_t a = 0; a < es.size();) {
                size_t b = a;
                while (b < es.size() && es[b].slot == es[a].slot) ++b;
                dir[es[a].slot] = (uint32_t)a << 16 | (uint32_t)(b - a);
                a = b;
            }
            for (size_t i = 0; i < es.size(); ++i) {
                memcpy(&ent[4 * i], es[i].w, 16);
                // which bytes next to the pair this entry can accept at all (bit = byte & 31: the same for both cases
                // of a letter): a piece of >= 3 bytes its own third byte; a piece of two bytes, whose side stands
                // right behind (in front of) the pair, one of the side's two nearest bytes as the first or the second
                // byte there -- side_within_one_edit: the first mismatch is the missing, the replaced or the extra byte
                const uint32_t *w = es[i].w;
                const uint32_t pl = w[1] >> 24, meta = w[3] >> 24, L = meta & 7u;
                uint32_t *fm = &fmask[(size_t)agh_mw_slot(w[0] & 0xffffu) * 4];     // (the mask table goes by the pair)
                if (pl >= 4u) {                     // its first four bytes (the pool holds folded bytes under -i)
                    const uint32_t h = agh_sample_prod_q4(w[0]);
                    g4[(h >> 2) & (AGH_MW_G4_WORDS - 1u)] |= 1u << ((h >> 13) & 31u);
                } else if (pl == 3u) {
                    fm[0] |= 1u << ((w[0] >> 16) & 31u);
                } else {
                    const uint32_t near2 = L >= 2u ? (1u << (w[2] & 31u)) | (1u << ((w[2] >> 8) & 31u)) : ~0u;
                    if (meta & 8u) { fm[2] |= near2; fm[3] |= near2; }
                    else { fm[0] |= near2; fm[1] |= near2; }
                }
            }
            if (up(&q->d_mw_ent, ent.data(), ent.size() * 4) || up(&q->d_mw_dir, dir.data(), dir.size() * 4) ||
                up(&q->d_mw_fmask, fmask.data(), fmask.size() * 4) || up(&q->d_mw_g4, g4.data(), g4.size() * 4))
                return -1;
            q->mw_nent = (uint32_t)es.size();
            q->mw_ok = true;
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
    if (q->d_mw_ent) (void)hipFree(q->d_mw_ent);
    if (q->d_mw_dir) (void)hipFree(q->d_mw_dir);
    if (q->d_mw_fmask) (void)hipFree(q->d_mw_fmask);
    if (q->d_mw_g4) (void)hipFree(q->d_mw_g4);
    if (q->d_ms_ptab) (void)hipFree(q->d_ms_ptab);
    if (q->d_ms_gtab) (void)hipFree(q->d_ms_gtab);
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
    q->rec_pos.release();
    q->tf_cont.release();
    q->bm_blocks.release();
    q->match_out.release();
    if (q->h_emit) (void)hipHostFree(q->h_emit);
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

