// agh_api.cpp -- scans of text resident in HBM for libagrep_hip.so (include/agrep_hip.h): the segment planner,
// the lean / numbered / one-pass pipelines and their kernel orchestration, the tuning switches,
// agh_scan_device[_reduce], the corpus generator and the read probes.  Queries are built in agh_query.cpp,
// staging, files and record output live in agh_stage.cpp.  No CPU scan path exists here: if HIP is unusable
// every entry point fails with -1 / errno = 123.
#include "agh_internal.h"

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
    // table engine (k_tablescan_fast + k_table_replay); edit costs since round 5
    // (';' AND patterns flag a piece for every record end with ANY of their end bits: with costs -- one stream per lane --
    // the replays outweigh the fast kernel: 'match;approx' k = 2 0.71 against 0.80 TB/s, profiles/r05_perf_table_costs.log)
    // delimiters of several bytes / a folded letter (round 5): record ends from the delimiter-end bitmap, k <= 4
    if (q->table) return !((q->dlen > 1 || q->delim_fold) && q->k > 4) && !((q->ci != 1 || q->cs != 1 || q->cd != 1) && q->tab.AND);
    // k = 0: one level, nothing to pack -- the one-kernel form is faster there (3.8 vs 3.2 TB/s)
    return q->k >= 1 && !q->general && !(q->dlen > 1 || q->delim_fold) && q->mask[q->delim[0]] == 0;
}

static int fs_fast_setup(agh_query *q, uint64_t n, agh_scan_args *va)
{
    va->fs_fast = 0;
    va->tf_chunk = 0;
    va->tr_group = 0;
    va->tf_direct = 0;
    va->tf_cont = nullptr;
    va->tf_cont_cap = 0;
    va->tf_cont_at = 0;
    va->fs_streams = 0;
    if (!fs_fast_ok(q)) return 0;
    if (q->table && n < (q->tune.tf_fast_min_mb << 20)) return 0;
    // (64 KiB tiles with 256 entries each; the table engine's 256 KiB tiles with 1024 entries need the same)
    const uint64_t n_tiles = (n + 65535) / 65536;
    if (q->cand.ensure((n_tiles + 4) * AGH_FF_SLICE_HOST * sizeof(uint64_t))) return -1;
    if (q->wave_cand.ensure((n_tiles + 8) * sizeof(uint32_t))) return -1;
    va->fs_fast = 1;
    va->tf_chunk = q->tune.tf_chunk;
    va->tr_group = q->tune.tr_group;
    va->tf_direct = q->tune.tf_direct ? 1u : 0u;
    va->fs_streams = q->tune.fs_streams;
    // table engine, M <= 15: two streams per lane (k_tablescan_fast2) -- the always-one bit M must be there
    if (q->table && q->tune.tf_pack2) {
        const unsigned M = (unsigned)q->m + (unsigned)q->dlen + 1u;
        if (M <= 15u && ((q->tab.Init0 >> M) & (q->tab.Init1 >> M) & 1u)) va->fs_fast = 2;
    }
    va->fs_replay = (uint64_t *)q->cand.p;
    va->fs_tile_cnt = (uint32_t *)q->wave_cand.p;
    if (q->table && q->tune.tf_cont) {
        // at most tf_cont lanes x (one or two streams) of every 64-chunk tile hand a record over, and two streams per
        // lane come with two tiles per wave; the chunk is agh_launch_tablescan's choice (48 B per entry: 0.9 % of a
        // text of 1 GiB or more at the default of 48 lanes)
        const uint64_t chunk = agh_tf_chunk_for(n, q->tune.tf_chunk);
        const uint64_t cap = ((n + 64 * chunk - 1) / (64 * chunk) + 8) * q->tune.tf_cont;
        if (cap < ((uint64_t)1 << 31) && q->tf_cont.ensure(cap * 48u) == 0) {
            va->tf_cont = (uint4 *)q->tf_cont.p;
            va->tf_cont_cap = (uint32_t)cap;
            va->tf_cont_at = q->tune.tf_cont;
        }
    }
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
// what one numbered kernel sequence takes: 64 chunks of 1024 wave ranges in the census scan (agh_sweep.hip), 40-bit
// sample indices in the candidate entries, 32-bit record numbers
static const uint64_t AGH_SEG_HARD_MAX = (uint64_t)16 << 30;
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
    uint64_t matched = 0, records = 0, candidates = 0, stored = 0, rec_bytes = 0;
    uint32_t engine = 0, truncated = 0, lean_rerun = 0;
    float ms = 0.f, sweep_ms = 0.f;
};

static uint32_t device_cus();
static int scan_segment(agh_query *q, const void *d_text, uint64_t n, hipStream_t st,
                        unsigned flags, uint32_t head_byte, int tail_virtual,
                        const agh_list_out *list, seg_result *out, const uint64_t *pre_dbm)
{
    // list: this segment's part of the caller's record list (pointers already advanced, cap = what is left)
    uint64_t *const d_match_pos = list ? list->pos : nullptr;
    uint32_t *const d_match_rec = list ? list->rec : nullptr;
    const uint32_t match_cap = list ? (uint32_t)std::min<size_t>(list->cap, 0xffffffffu) : 0u;
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
    if (n > AGH_SEG_HARD_MAX) return fail("segments are limited to 16 GiB");
    if (multi && (flags & AGH_FORCE_FULLSCAN))
        return fail("multi-pattern queries have no full-scan engine");
    const bool want_filter = (q->fq > 0 || pe) && !(flags & AGH_FORCE_FULLSCAN) && !q->table && !invert_list &&
                             !(q->general && (q_mb(q) || pe));   // general verify: 1-byte delimiters
    if (!want_filter && (flags & AGH_FORCE_FILTER))
        return fail("the q-gram filter does not apply to this query (m=%d, k=%d)", q->m, q->k);
    // the automaton over every byte, the table engine, the multi-pattern kernels, k >= 5 and -v lists: the second library
    if ((multi || !want_filter || q->k >= 5 || invert_list) && agh_need_engines()) return -1;
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
        if (flags & AGH_TIME_SCAN) HIP_TRY(hipEventRecord(q->ev2, st));     // (device_ms includes the bitmap since round 5)
        agh_launch_delim_bitmap(d_text, n, dq, (uint64_t *)dbm_buf.p, n_words, q->d_counters, st);
        if (flags & AGH_TIME_SCAN) HIP_TRY(hipEventRecord(q->ev3, st));
        HIP_TRY(hipMemcpyAsync(q->h_counters, q->d_counters, AGH_C_COUNT * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (flags & AGH_TIME_SCAN) {
            float bm_ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&bm_ms, q->ev2, q->ev3));
            q->dbm_ms += bm_ms;
        }
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
            return scan_segment(q, d_text, n, st, flags, head_byte, tail_virtual, list, out, pre_dbm);
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
            return scan_segment(q, d_text, n, st, flags, head_byte, tail_virtual, list, out, pre_dbm);
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
    // numbered scans of a dense -f set with one error: the tile kernel marks records by number (round 5: k_dense_multi)
    const bool tile_numbered = multi && q->mw_ok && !d_dbm && q->k == 1 && q->tune.mtile_numbered;
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
            if (d_match_pos) {                  // the list: rec_pos[r] per bitmap bit, the compaction's block counts
                if (q->rec_pos.ensure(bm_words * 32 * sizeof(uint64_t))) return -1;
                if (q->bm_blocks.ensure((q->bitmap.cap / 4 / 1024 + 4) * sizeof(uint32_t))) return -1;
            }
        }

        if (flags & AGH_TIME_SCAN) HIP_TRY(hipEventRecord(q->ev0, st));
        if (!swept) HIP_TRY(hipMemsetAsync(q->d_counters, 0, AGH_C_COUNT * sizeof(uint32_t), st));
        else {
            // keep the census results (NDELIM, LASTBYTE, CAND), clear the rest
            HIP_TRY(hipMemsetAsync(q->d_counters + AGH_C_OVERFLOW, 0, sizeof(uint32_t), st));
            HIP_TRY(hipMemsetAsync(q->d_counters + AGH_C_MATCHED, 0, sizeof(uint32_t), st));
            HIP_TRY(hipMemsetAsync(q->d_counters + AGH_C_STORED, 0, sizeof(uint32_t), st));
            HIP_TRY(hipMemsetAsync(q->d_counters + AGH_C_BM_OVERFLOW, 0, sizeof(uint32_t), st));
            HIP_TRY(hipMemsetAsync(q->d_counters + AGH_C_RECBYTES_LO, 0, 2 * sizeof(uint32_t), st));
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
            if (tile_numbered) {
                // dense one-error sets (agh_mtile.hip): the census, then candidate bits per tile and the walk over
                // them, matched records marked by number -- no candidate slices, no second kernel
                agh_launch_sweep(sa, 0, st);
                agh_mwalk_args w;
                w.text = d_text;
                w.n = n;
                w.q = dq;
                w.mw.ent = (const uint4 *)q->d_mw_ent;
                w.mw.dir = (const uint32_t *)q->d_mw_dir;
                w.mw.fmask = (const uint4 *)q->d_mw_fmask;
                w.mw.g4 = (const uint32_t *)q->d_mw_g4;
                w.mw.n_ent = q->mw_nent;
                w.mt = multi_dev(q, nullptr);
                memset(&w.mk, 0, sizeof(w.mk));
                w.mk.bitmap = (uint32_t *)q->bitmap.p;
                w.mk.bitmap_bits = (uint32_t)std::min<uint64_t>(bm_words * 32, 0xffffffffu);
                w.mk.counters = q->d_counters;
                w.mk.rec_pos = (d_match_pos && !invert_list) ? (uint64_t *)q->rec_pos.p : nullptr;
                if (q->tickets.ensure(256u)) return -1;
                HIP_TRY(hipMemsetAsync(q->tickets.p, 0, 256u, st));
                w.ticket = (uint32_t *)q->tickets.p;
                w.n_cu = device_cus();
                w.ch = q->tune.mtile | q->tune.mtile_dbg << 8;
                w.wave_totals = (const uint32_t *)q->wave_totals.p;
                w.strip_prefix = (const uint32_t *)q->strip_prefix.p;
                sa.ev_begin = sa.ev_end = nullptr;
                if (!agh_launch_mtile(w, st)) return fail("internal error: no tile walk for this pattern set");
            } else if (multi && q->multi_dense) {
                // census first (plain H=0 sweep + prefix scan), then the inline multi sweep
                // numbers records from that prefix and marks them directly
                agh_launch_sweep(sa, 0, st);
                agh_marks mk0;
                memset(&mk0, 0, sizeof(mk0));
                mk0.bitmap = (uint32_t *)q->bitmap.p;
                mk0.bitmap_bits = (uint32_t)std::min<uint64_t>(bm_words * 32, 0xffffffffu);
                mk0.counters = q->d_counters;
                mk0.rec_pos = d_match_pos ? (uint64_t *)q->rec_pos.p : nullptr;
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
        va.mk.rec_pos = (d_match_pos && !invert_list) ? (uint64_t *)q->rec_pos.p : nullptr;   // -v: bits only, list below
        va.mk.hashset = nullptr;
        va.mk.hashset_mask = 0;
        va.gtab = (q->tune.tight_verify && !multi) ? q->d_gtab : nullptr;
        va.gram_spread = q->gram_spread;
        if (!multi && !use_filter && fs_fast_setup(q, n, &va)) return -1;
        const bool fs_fast = va.fs_fast != 0;
        if (tile_numbered) { /* marked by k_mtile */ }
        else if (multi) agh_launch_verify_multi(va, multi_dev(q, d_dbm), false, st);
        else if (use_filter) agh_launch_verify(va, st);
        else if (q->table) agh_launch_tablescan(va, st);
        else agh_launch_fullscan(va, st);
        if (invert_list) {                      // the records whose bit stayed clear
            va.mk.rec_pos = (uint64_t *)q->rec_pos.p;
            agh_launch_unmatched(va, st);
        }
        if (d_match_pos) {
            // the list in file order: ordered compaction of the bitmap (counts and clears it as well), then the
            // bounds of the listed records -- everything queued, the host reads the counters once below
            agh_launch_bitmap_list((uint32_t *)q->bitmap.p, (uint32_t)(q->bitmap.cap / 4),
                                   (uint32_t)std::min<uint64_t>(bm_words * 32, 0xffffffffu), (uint32_t *)q->bm_blocks.p,
                                   (const uint64_t *)q->rec_pos.p, invert_list ? 1 : 0, d_match_pos, d_match_rec, match_cap,
                                   q->d_counters, st);
            if (list->start)
                agh_launch_match_bounds(d_text, n, dq, d_dbm, d_match_pos, q->d_counters, match_cap,
                                        (uint32_t)std::min<uint64_t>(match_cap, bm_words * 32), list->start, list->end, st);
        } else {
            agh_launch_bitmap_count((uint32_t *)q->bitmap.p, (uint32_t)(q->bitmap.cap / 4), q->d_counters, st);
        }
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
        if (q->h_counters[AGH_C_BM_OVERFLOW] == 2u)
            return fail("more than 2^32 records in one segment of %llu bytes (record numbers are 32-bit): set AGH_SEG_MAX_MB",
                        (unsigned long long)n);
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
            if (agh_need_engines()) return -1;
        }
        if (slice_overflow || bm_overflow) {
            bits_hint = (uint64_t)n_delims + 1024;
            if ((multi && q->multi_dense) || tile_numbered) swept = false;     // the inline sweep / the tile kernel does the marking
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
        out->rec_bytes = (uint64_t)q->h_counters[AGH_C_RECBYTES_LO] | ((uint64_t)q->h_counters[AGH_C_RECBYTES_HI] << 32);
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
    if (q->tune.seg_max_mb) return q->tune.seg_max_mb << 20;        // AGH_SEG_MAX_MB (tests: tiny segments)
    if (q->multi || q->piece_single) return (uint64_t)2 << 30;
    return AGH_SEG_MAX_DEFAULT;
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
    // a text one kernel sequence can take is not cut at all (16 GiB: config C3 in one piece -- every segment pays
    // its launches, its census scan and a host synchronisation, ~80 us)
    const bool single = !q->tune.seg_max_mb && !q->multi && !q->piece_single && len <= AGH_SEG_HARD_MAX;
    if (len > nominal && !single) {
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
    t->fs_streams = (uint32_t)std::min<uint64_t>(env_u64("AGH_FS_STREAMS", 0), 4);
    t->tf_pack2 = env_on("AGH_TF_PACK2", true);
    t->tf_direct = env_on("AGH_TF_DIRECT", true);
    t->tf_cont = (uint32_t)std::min<uint64_t>(env_u64("AGH_TF_CONT", 48), 64);
    t->tf_fast_min_mb = env_u64("AGH_TF_FAST_MIN_MB", AGH_TF_FAST_MIN_MB_DEFAULT);
    {
        const uint64_t c = env_u64("AGH_TF_CHUNK", AGH_TF_CHUNK_DEFAULT);
        t->tf_chunk = (c == 1024 || c == 2048 || c == 4096 || c == 8192 || c == 16384 || c == 32768) ? (uint32_t)c : 0u;
        const uint64_t g = env_u64("AGH_TR_GROUP", 8);
        t->tr_group = (g == 1 || g == 2 || g == 4 || g == 8 || g == 16) ? (uint32_t)g : 8u;
        const uint64_t mtl = env_u64("AGH_MTILE", 2);
        t->mtile = (mtl == 1 || mtl == 2 || mtl == 4) ? (uint32_t)mtl : 2u;
        // (measurements: AGH_MTILE_DBG bits, AGH_MTILE_SHARE = lanes with candidates at which the rest is shared out)
        t->mtile_dbg = (uint32_t)(env_u64("AGH_MTILE_DBG", 0) & 0xff);
        t->mtile_numbered = env_on("AGH_MTILE_NUMBERED", true);
        if (getenv("AGH_MTILE_SHARE")) t->mtile_dbg |= (uint32_t)((env_u64("AGH_MTILE_SHARE", 32) & 0x7f) + 1) << 8;
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
                        const agh_list_out *list, seg_result *out, const uint64_t *pre_dbm = nullptr);


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
    agh_timeline("lean_run: start");
    if (q->k >= 5 && agh_need_engines()) return -1;       // (the verify kernels for k = 5..8 live in the second library)
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

    agh_timeline("lean_run: buffers ready");
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
    agh_timeline("lean_run: everything queued");
    HIP_TRY(hipStreamSynchronize(st));
    if (overlap) HIP_TRY(hipStreamSynchronize(aux));
    agh_timeline("lean_run: synchronised");
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
                                 nullptr, i == 0 && is_first, i == nseg - 1 && is_last))
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
    return q->multi && (q->ms_ok || q->mw_ok) && !want_list && (flags & (AGH_COUNT | AGH_FILENAMEONLY)) &&
           !(flags & (AGH_INVERT | AGH_FORCE_FULLSCAN | AGH_FORCE_NUMBERED));
}

static int mscan_run(agh_query *q, const unsigned char *base, const std::vector<uint64_t> &cuts,
                     hipStream_t st, unsigned flags, agh_result *res, bool is_first, bool is_last)
{
    const int nseg = (int)cuts.size() - 1;
    const bool timing = (flags & (AGH_TIME_SWEEP | AGH_TIME_SCAN)) != 0;
    uint64_t max_n = 0;
    for (int i = 0; i < nseg; ++i) max_n = std::max(max_n, cuts[i + 1] - cuts[i]);
    // the tile kernel over a dense set (agh_mtile.hip) counts the records inside a 4 KiB tile in registers; the set only
    // takes the records that cross a tile's bounds (two per tile at most) -- but on such sets most of those match: one
    // slot per 512 bytes of text (load <= 1/4)
    const bool walk = !q->ms_ok;
    uint64_t slots = 1u << 17;
    if (walk) while (slots < (max_n >> 9) && slots < (1u << 28)) slots <<= 1;
    else if (!q->hashset_slots_hint) while (slots < (max_n >> 13) && slots < (1u << 26)) slots <<= 1;
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
        if (walk) {
            agh_mwalk_args w;
            w.text = a.text;
            w.n = a.n;
            w.q = dq;
            w.mw.ent = (const uint4 *)q->d_mw_ent;
            w.mw.dir = (const uint32_t *)q->d_mw_dir;
            w.mw.fmask = (const uint4 *)q->d_mw_fmask;
            w.mw.g4 = (const uint32_t *)q->d_mw_g4;
            w.mw.n_ent = q->mw_nent;
            w.mt = a.mt;
            w.mk = a.mk;
            w.ticket = a.ticket;
            w.n_cu = a.n_cu;
            w.ch = q->tune.mtile | q->tune.mtile_dbg << 8;
            if (!agh_launch_mtile(w, st)) return fail("internal error: no tile walk for this pattern set");
        } else if (!agh_launch_mscan(a, st)) return fail("internal error: no one-pass kernel for this pattern set");
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
                                 nullptr, i == 0 && is_first, i == nseg - 1 && is_last))
            return -1;
        res->n_matched += rr.n_matched;
        res->n_candidates += rr.n_candidates;
        res->lean_reruns += 1;
    }
    res->n_segments = (uint32_t)nseg;
    if (!walk) q->hashset_slots_hint = 4ull * max_matched;
    return 0;
}

int agh_scan_device_impl(agh_query *q, const void *dev_text, uint64_t len, hipStream_t st,
                         unsigned flags, agh_result *res, agh_list_out *list, bool is_first, bool is_last)
{
    if (!q || !res) return fail("null argument");
    if (list && !list->pos) list = nullptr;
    uint64_t *const d_match_pos = list ? list->pos : nullptr;
    uint32_t *const d_match_rec = list ? list->rec : nullptr;
    const size_t match_cap = list ? list->cap : 0;
    if (list) list->rec_bytes = 0;
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
        if (flags & AGH_TIME_SCAN) HIP_TRY(hipEventRecord(q->ev2, st));
        agh_launch_delim_bitmap(base, len, dq, (uint64_t *)q->dbm.p, n_words, q->d_counters, st);
        if (flags & AGH_TIME_SCAN) HIP_TRY(hipEventRecord(q->ev3, st));
        HIP_TRY(hipMemcpyAsync(q->h_counters, q->d_counters, AGH_C_COUNT * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (flags & AGH_TIME_SCAN) {
            float bm_ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&bm_ms, q->ev2, q->ev3));
            res->device_ms += bm_ms;
        }
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
        agh_list_out seg_list;
        if (list) {
            seg_list.pos = d_match_pos + stored;
            seg_list.rec = d_match_rec ? d_match_rec + stored : nullptr;
            seg_list.start = list->start ? list->start + stored : nullptr;
            seg_list.end = list->end ? list->end + stored : nullptr;
            seg_list.cap = cap_left;
        }
        const int seg_rc = scan_segment(q, seg_text, end - off, st, flags,
                                        (i == 0 && is_first) ? '\n' : q->delim[q->dlen - 1], end == len && is_last,
                                        list ? &seg_list : nullptr, &sr, seg_dbm);
        q->seg_dbm_active = false;
        if (seg_rc) return -1;
        if (list) list->rec_bytes += sr.rec_bytes;
        if (d_match_pos && sr.stored && off > 0) {
            // the segment's kernels saw positions / record numbers relative to its own start
            agh_launch_offset_matches(seg_list.pos, seg_list.rec, seg_list.start, seg_list.end,
                                      (uint32_t)sr.stored, off, (uint32_t)res->n_records, st);
            HIP_TRY(hipGetLastError());
        }
        res->n_matched += sr.matched;
        res->n_records += sr.records;
        res->n_candidates += sr.candidates;
        res->n_stored += sr.stored;
        res->truncated |= sr.truncated;
        res->device_ms += sr.ms + q->dbm_ms;    // (the delimiter-end bitmap of the segment, if it needed one)
        q->dbm_ms = 0.f;
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
    agh_list_out list;
    list.pos = (uint64_t *)dev_match_pos;
    list.cap = dev_match_pos ? match_cap : 0;
    return agh_scan_device_impl(q, dev_text, len, (hipStream_t)stream, flags, res, dev_match_pos ? &list : nullptr, true, true);
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
    const int rc = agh_scan_device_impl(q, dev_text, len, (hipStream_t)stream, flags, res, nullptr, true, true);
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
