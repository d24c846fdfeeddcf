// agh_internal.h -- what the translation units of the host side of libagrep_hip.so share: the error
// helper, device buffers, the query object and the entry points of the scan orchestration
// (agh_query.cpp: queries; agh_api.cpp: segments, kernel sequences; agh_stage.cpp: host <-> HBM staging, files, pipes,
// record output).  Internal; the public boundary is include/agrep_hip.h.
#pragma once
#include <errno.h>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "../../include/agrep_hip.h"
#include "agh_device.h"
#include "agh_launch.h"


// errors: -1 / NULL + errno = AGH_ERRNO, the text kept per thread (agh_last_error)
__attribute__((visibility("hidden"))) int agh_fail(const char *fmt, ...);
#define fail agh_fail

// AGH_TIMELINE=1: time stamps of a call's milestones on stderr (milliseconds since the first stamp of the process;
// diagnostics of start-up cost, scripts/startup_r5.sh).  One getenv per process, nothing else when it is off.
__attribute__((visibility("hidden"))) void agh_timeline(const char *what);

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess)                                                            \
            return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                        __LINE__);                                                        \
    } while (0)

// ---------------------------------------------------------------------------------------
// query
// ---------------------------------------------------------------------------------------
struct dev_buf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 4096;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

#define AGH_PIN_RING 4        // pinned 16 MiB chunks between read() and the H2D copies (agh_stage.cpp)
#define AGH_LEAN_SLOTS 2      // segments of the lean pipeline in flight (sweep i+1 | verify i)
#define AGH_MAX_SEGS 256      // segments of one scan (8 GiB each: 2 TiB)

__attribute__((visibility("hidden"))) void agh_read_tuning(agh_tuning *t);

struct agh_query {
    agh_query() { agh_read_tuning(&tune); }
    int m = 0, k = 0, dlen = 1, wide = 0;
    bool delim_fold = false;            // -i with letters in a multi-byte delimiter
    unsigned char delim[AGH_MAX_DELIM] = {'\n'};
    uint64_t mask[256];                 // bit (p-1) set iff byte is in the class of position p
    int fq = 0, fh = 0;                 // filter sample shape (0: no filter)
    int run_a = 0, run_len = 0;         // the literal run of positions the samples are taken from
    uint32_t qmask = 0, fold = 0;
    // device-resident tables
    void *d_mask = nullptr;             // 256 x uint32_t or uint64_t
    uint8_t *d_ftab = nullptr;          // AGH_FT_SIZE bytes
    uint64_t *d_gtab = nullptr;         // AGH_FT_SIZE x (gram, first/last offset): tight verify windows
    uint32_t gram_spread = 0;           // max (last - first offset) over the grams in d_gtab
    // per-query workspace (grown lazily, reused across scans)
    dev_buf strip_prefix, wave_totals, cand, wave_cand, bitmap, hashset, dbm, staging, match_pos,
        match_rec, match_start, match_end, match_off, gather;
    dev_buf tf_cont;                    // table engine, fast form: open records handed from the scan to k_table_cont (48 B each)
    dev_buf rec_pos;                    // record lists: one byte offset per record number (8 B per bitmap bit)
    dev_buf bm_blocks;                  // ... scratch of the ordered compaction of the bitmap
    dev_buf match_out;                  // ... agh_match entries of the piece being emitted
    unsigned char *h_emit = nullptr;    // pinned: the matched records of one emit() call (entries + bytes)
    size_t h_emit_cap = 0;
    uint64_t staged_len = 0;            // bytes of the text currently held in `staging`
    unsigned char input_head[64] = {0}; // streamed scans: the first bytes of the input (agh_input_head)
    uint32_t input_head_len = 0;
    bool staged_first = true, staged_last = true;   // ... is the head / the tail of its file (agh_scan_fd_range)
    hipStream_t stage_stream = nullptr; // H2D copies of agh_scan_fd
    unsigned char *pinned[AGH_PIN_RING] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pinned_ev[AGH_PIN_RING] = {nullptr, nullptr, nullptr, nullptr};
    size_t pinned_cap[AGH_PIN_RING] = {0, 0, 0, 0};     // (allocated on first use, no larger than the input needs)
    dev_buf staging_b;                  // the second device segment of the streaming pipeline
    size_t match_cap_hint = 0;          // record output: matches of the previous segment (+25 %)
    uint32_t *d_counters = nullptr;     // AGH_LEAN_SLOTS + 1 counter blocks (block 0: everything but the pipeline)
    uint32_t *d_chunk_totals = nullptr; // scratch of the prefix scan
    uint32_t *h_counters = nullptr;     // pinned: block 0 + one block per pipelined segment
    // lean pipeline (lean_run): a second stream for the verifier, a second set of candidate
    // buffers, dependency / timing events, the device scratch of the segment cutter
    hipStream_t aux_stream = nullptr;
    dev_buf cand_b, wave_cand_b, cuts;
    dev_buf seg_copy;                   // aligned copy of a segment whose cut is not 16-byte aligned
    dev_buf seg_dbm;                    // ... and its own delimiter bitmap (q->dbm holds the whole text's)
    bool seg_dbm_active = false;
    dev_buf giveups;                    // lean scans: matches whose record start the verifier did not reach (k_resolve_giveups)
    dev_buf tickets;                    // fused lean kernel: one work counter (own 256-byte line) per segment
    std::vector<hipEvent_t> dep_events, time_events;
    uint64_t *h_cuts = nullptr;         // pinned: bounds, lower limits, cuts
    uint64_t bitmap_bits_hint = 0;      // records seen by the previous scan (+25 %)
    bool bitmap_dirty = false;          // a scan was queued but its count-and-clear did not finish
    uint64_t hashset_slots_hint = 0;    // lean scans: slots wanted by the previous scan
    bool hashset_dirty = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    float dbm_ms = 0.f;                         // time of the delimiter-end bitmaps built since the last segment result
    // multi-pattern (-f) queries
    // general automaton (asearch1.c costs, <exact> segments): full scan only
    bool general = false;
    bool table = false;                 // table engine: mask[] holds the reference's Mask[]
    agh_dev_tables tab;
    int ci = 1, cs = 1, cd = 1;
    uint64_t no_err = ~0ull;
    bool multi = false;
    // piece engine for a single literal pattern the sample filter cannot take (short pattern /
    // many errors): the multi-pattern tables hold its k+1 pieces (or the pattern itself, k = 0)
    bool piece_single = false;
    int pe_fq = 0, pe_minlen = 0;
    int guard = 0;                      // -f with -w (1) / -x (2): checked by the exact verifier
    int mp_stride = 1;                  // multi-pattern sweep: probe every 1 / 2 / 4 bytes (fill_multi_tables)
    bool mp_q5 = false;                 // ... with 5-byte grams (stride 4, entries of >= 8 bytes)
    uint32_t pe_qmask = 0, pe_fold = 0;
    bool multi_dense = false;           // hits are too dense for the candidate slices
    bool fs_fast_off = false;           // full scan: the replay lists overflowed once (match-dense text): exact kernel from now on
    int npat = 0;
    void *d_mp_bits = nullptr, *d_mp_bstart = nullptr, *d_mp_items = nullptr, *d_mp_pool = nullptr,
         *d_mp_omask = nullptr;
    agh_tuning tune;                    // the environment switches as they were when the query was created
    // agh_scan_device_reduce: the communicator of the step in progress, its totals on the device / pinned
    struct agh_comm *reduce_comm = nullptr;
    uint64_t *d_acc = nullptr, *h_acc = nullptr;        // matched, records, segments that gave up (+ 1 spare)
    bool reduce_done = false;
    // one-pass count-only -f scan (agh_mscan.hip): pair table, exact gram table, entry directory
    bool ms_ok = false;
    uint32_t ms_rb = 0, ms_dbg = 0;
    void *d_ms_ptab = nullptr, *d_ms_gtab = nullptr, *d_ms_ment = nullptr;
    // record walk over dense -f sets with one error (agh_mwalk.hip)
    bool mw_ok = false;
    uint32_t mw_nent = 0;
    void *d_mw_ent = nullptr, *d_mw_dir = nullptr, *d_mw_fmask = nullptr, *d_mw_g4 = nullptr;
};

// delimiter ends come from the delimiter bitmap: several bytes, or one letter under -i
static inline bool q_mb(const agh_query *q) { return q->dlen > 1 || q->delim_fold; }

// ---- agh_comm.cpp (internal) --------------------------------------------------------------------
extern "C" __attribute__((visibility("hidden"))) int agh_comm_allreduce_dev(struct agh_comm *c, uint64_t *d_buf, size_t count,
                                                                            hipStream_t st);
extern "C" __attribute__((visibility("hidden"))) int agh_comm_allreduce_host(struct agh_comm *c, uint64_t *v, size_t count);

// ---- agh_ext.cpp: the engines behind the filter live in libagrep_hip_engines.so ------------------------------------
__attribute__((visibility("hidden"))) int agh_need_engines();

// ---- agh_api.cpp ------------------------------------------------------------------------------
// at the start of every public scan call: AGH_ENV_LIVE=1 re-reads the switches (tests, A/B scripts)
static inline void agh_refresh_tuning(agh_query *q) { if (q->tune.live) agh_read_tuning(&q->tune); }
// The record list of a scan, on the device, in file order (agh_records.hip): per listed record one byte offset
// inside it (pos), its number (rec, optional) and -- if start / end are given -- its bounds [start, end).
// cap entries each; rec_bytes (out) = sum of the lengths of the stored records.
struct agh_list_out {
    uint64_t *pos = nullptr;
    uint32_t *rec = nullptr;
    uint64_t *start = nullptr, *end = nullptr;
    size_t cap = 0;
    uint64_t rec_bytes = 0;
};
// one scan of text resident in HBM, cut into segments as the query needs; list (optional): the matched
// records (-v: the others)
__attribute__((visibility("hidden"))) int agh_scan_device_impl(agh_query *q, const void *dev_text, uint64_t len, hipStream_t st,
                                                               unsigned flags, agh_result *res, agh_list_out *list,
                                                               bool is_first, bool is_last);
