// agh_launch.h -- internal C++ interface between the C-ABI layer (agh_query.cpp, agh_api.cpp) and the
// kernel translation units (agh_sweep / agh_scan / agh_multi / agh_table / agh_exp .hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "agh_device.h"

// Environment switches (DESIGN.md "Environment switches"), read ONCE when a query is created -- no getenv on
// a scan path.  AGH_ENV_LIVE=1 (the test suite, A/B scripts) re-reads them at the start of every scan call.
struct agh_tuning {
    bool live = false;              // AGH_ENV_LIVE
    bool tight_verify = true;       // AGH_TIGHT_VERIFY
    bool fs_fast = true;            // AGH_FS_FAST
    uint32_t fs_streams = 0;        // AGH_FS_STREAMS: full scan, fast form: at most 1..3 text streams per lane (0: by length; A/B)
    bool tf_pack2 = true;           // AGH_TF_PACK2: table engine, two streams per lane where M <= 15
    uint64_t tf_fast_min_mb = 0;    // AGH_TF_FAST_MIN_MB: table engine, fast form from this segment size on
    uint32_t tf_chunk = 0;          // AGH_TF_CHUNK: bytes per lane of the fast form (1024 .. 32768), 0 = by size (agh_tf_chunk_for)
    bool tf_direct = true;          // AGH_TF_DIRECT: table engine, count-only: the fast kernel counts pieces with one record end itself (0: every flagged piece is replayed; A/B)
    uint32_t tf_cont = 48;          // AGH_TF_CONT: table engine, fast form: the walk past a chunk's end hands its open records to k_table_cont once this few lanes still have one (0: every wave walks to its longest record's end)
    uint32_t tr_group = 8;          // AGH_TR_GROUP: tiles whose replay lists one wave of k_table_replay takes (1, 2, 4, 8, 16)
    uint32_t mtile = 2;             // AGH_MTILE: dense -f sets with one error: tiles a wave of k_mtile holds at a time (1, 2, 4)
    bool mtile_numbered = true;     // AGH_MTILE_NUMBERED: 0: numbered scans of such sets stay on k_dense_multi (A/B, tests)
    uint32_t mtile_dbg = 0;         // AGH_MTILE_DBG | (AGH_MTILE_SHARE + 1) << 8: measurement switches of k_mtile
    bool fused = true;              // AGH_FUSED
    bool debug = false;             // AGH_DEBUG
    bool aligned_cuts_only = false; // AGH_ALIGNED_CUTS_ONLY
    bool stream = true;             // AGH_STREAM
    uint64_t seg_max_mb = 0;        // AGH_SEG_MAX_MB (0: unset)
    uint64_t part_mb = 0;           // AGH_PART_MB
    bool overlap = false;           // AGH_OVERLAP
    uint64_t fused_min_mb = 4096;   // AGH_FUSED_MIN_MB
    uint64_t stream_seg_mb = 1024;  // AGH_STREAM_SEG_MB
    unsigned readers = 0;           // AGH_READERS (0: unset)
    long fused_range_kb = -1, fused_tail_kb = -1, fused_tail_mb = -1, fused_blocks = -1;   // AGH_FUSED_* (-1: unset)
    long verify_blocks = -1;        // AGH_VERIFY_BLOCKS (-1: unset)
    long giveup_cap = -1;           // AGH_GIVEUP_CAP: entries of the lean scans' give-up list (-1: 4096; 0: none -- a record
                                    // start further back than the verifier looks reruns the segment, as in round 3)
};

struct agh_marks {
    uint32_t *bitmap;        // one bit per record
    uint32_t bitmap_bits;    // capacity; larger record numbers raise AGH_C_BM_OVERFLOW
    uint32_t *counters;
    uint64_t *rec_pos;       // optional (record lists): rec_pos[r] = one byte offset inside record r, written by whoever
                             // sets r's bit first -- no shared counter (one hot L2 atomic takes ~90 updates/us: 104 197
                             // matches per 4 GiB cost 1.2 ms of a 0.7 ms sweep); the list comes out of the ordered
                             // compaction of the bitmap afterwards (agh_records.hip), in file order
    uint64_t *hashset;       // lean scans: open-addressing set of (record start + 1)
    uint32_t hashset_mask;   // slots - 1 (power of two)
    // lean scans, one-byte delimiters: positions of matches whose record start lies more than
    // AGH_LEAN_BACK_CAP bytes back; k_resolve_giveups finds those starts after the scan (NULL / 0: such a match
    // raises AGH_C_LEAN_FALLBACK and the host reruns the segment with record numbers)
    uint64_t *giveups;
    uint32_t giveup_cap;
};

struct agh_sweep_args {
    const void *text;
    uint64_t n;
    agh_dev_query q;
    const uint8_t *ftab;     // AGH_FT_SIZE bytes (device), may be NULL when H == 0
    uint32_t *strip_prefix;  // ceil(n / 1024) + 4 entries
    uint32_t *wave_totals;   // becomes wave_prefix after the sweep
    uint64_t *cand;          // nw slices of AGH_SLICE_CAP entries: (record count << 32) | dword
    uint32_t *wave_cand;     // nw candidate counts
    uint32_t *chunk_totals;  // 2 * 64 scratch words of the prefix scan
    const uint64_t *dbm;     // multi-byte delimiters: delimiter-end bitmap (else NULL)
    uint32_t *counters;
    int lean;                // 1: no delimiter census (count-only scans)
    hipEvent_t ev_begin;     // optional: recorded right before / after the k_sweep launch
    hipEvent_t ev_end;
    // part of the text swept by this launch: wave ranges [w_begin, w_end) (a wave range is
    // AGH_WAVE_STRIPS KiB); w_end == 0: the whole text.  Lean sweeps only: parts let the
    // verifier of one part run while the next part is swept, and let -l stop early.
    uint32_t w_begin = 0, w_end = 0;
    int tail_only = 0;       // 1: only the partial last strip (the fused kernel swept the rest)
};

// Bytes per lane of the fast table kernels (a tile is 64 such chunks): the walk past a chunk's end and the start-up of
// a stream cost the same whatever the chunk, but a wave walks its chunk serially, so small texts take small chunks
// (profiles/r06_ab_table_chunk.log: 8 KiB is worth 28 % at 8 GiB with 1.7 KB records and costs lines nothing there;
// at 4 GiB 16 % against 2 % of the lines' rate, at 2 GiB 14 % against 4 % -- lines decide).  forced: AGH_TF_CHUNK.
static inline uint32_t agh_tf_chunk_for(uint64_t n, uint32_t forced)
{
    if (forced) return forced;
    if (n >= ((uint64_t)8 << 30)) return 8192u;
    if (n >= ((uint64_t)1 << 30)) return 4096u;
    return n >= ((uint64_t)512 << 20) ? 2048u : 1024u;
}

struct agh_scan_args {
    const void *text;
    uint64_t n;
    agh_dev_query q;
    const void *mask;        // 256 x uint32_t or uint64_t (device)
    int wide;                // 1: 64-bit state words
    int general;             // 1: general automaton (costs / <exact>), full scan only
    int table;               // 1: table engine (mask = the reference's Mask[], tab = its scalars)
    agh_dev_tables tab;
    const uint64_t *cand;
    const uint32_t *wave_cand;
    uint32_t nw;
    const uint32_t *strip_prefix;
    const uint32_t *wave_prefix;
    const uint64_t *dbm;     // multi-byte delimiters: delimiter-end bitmap (else NULL)
    uint32_t n_strips;
    const uint64_t *gtab;    // lean verify: gram table (gram, first/last pattern offset) or NULL
    uint32_t gram_spread;
    agh_marks mk;
    uint32_t w_begin, w_end; // lean verify: slices [w_begin, w_end) only (w_end == 0: all nw)
    // full scan, fast form (agh_fullscan.hip k_fullscan_fast + k_fullscan_replay): list of pieces to
    // walk exactly (AGH_FF_SLICE entries per 64 KiB tile) and its per-tile counts
    int fs_fast;
    uint32_t tf_chunk;              // table engine, fast form: bytes per lane (1024 / 2048 / 4096), 0 = by size
    uint32_t tr_group;              // tiles per wave of k_table_replay (0: 8)
    uint4 *tf_cont;                 // table engine, fast form: entries of the open records handed over (3 x uint4 each), or null
    uint32_t tf_cont_cap;           // ... entries the buffer holds
    uint32_t tf_cont_at;            // ... lanes with an open record at which a wave hands over (0: never)
    uint32_t tf_direct;             // table engine, fast form, count-only: 1 = count pieces with one record end in the fast kernel
    uint32_t fs_streams;            // full scan, fast form: at most this many text streams per lane (0: by the pattern's length)
    uint64_t *fs_replay;
    uint32_t *fs_tile_cnt;
    long verify_blocks;      // AGH_VERIFY_BLOCKS: grid cap of k_verify (-1: the default)
};

void agh_launch_sweep(const agh_sweep_args &a, int H, hipStream_t st);

// count-only scans in one kernel (agh_fused.hip): sweep + verify of all full strips
struct agh_fused_args {
    const void *text;
    uint64_t n;
    agh_dev_query q;
    const uint8_t *ftab;
    const void *mask;        // 256 x uint32_t or uint64_t (device)
    int wide;                // 1: 64-bit state words
    const uint64_t *gtab;
    uint32_t gram_spread;
    agh_marks mk;            // hash set + counters
    uint32_t *ticket;        // zeroed work counter in a cache line of its own
    uint32_t n_cu;
    const agh_tuning *tune;  // AGH_FUSED_RANGE_KB / _TAIL_KB / _TAIL_MB / _BLOCKS overrides (A/B runs)
};
bool agh_launch_sweep_fused(const agh_fused_args &a, int H, hipStream_t st);
void agh_launch_verify(const agh_scan_args &a, hipStream_t st);
void agh_launch_fullscan(const agh_scan_args &a, hipStream_t st);
void agh_launch_tablescan(const agh_scan_args &a, hipStream_t st);
void agh_launch_unmatched(const agh_scan_args &a, hipStream_t st);
void agh_launch_bitmap_count(uint32_t *bitmap, uint32_t n_words, uint32_t *counters,
                             hipStream_t st);
void agh_launch_hashset_count(uint64_t *tab, uint32_t n_slots, const uint32_t *wave_cand,
                              uint32_t nw, uint32_t *counters, hipStream_t st);
// the record starts of the matches a lean scan left in mk.giveups (a look-back without a limit, one workgroup
// per match), entered into the scan's hash set
void agh_launch_resolve_giveups(const void *text, uint32_t delim, const agh_marks &mk, hipStream_t st);
void agh_launch_accumulate_counts(const uint32_t *counters, uint64_t *acc, hipStream_t st);
void agh_launch_verify_lean(const agh_scan_args &a, hipStream_t st);
void agh_launch_find_cuts(const void *text, const uint64_t *bound, const uint64_t *lo,
                          uint32_t n_bounds, uint32_t delim, uint32_t step, uint64_t *cut, hipStream_t st);
void agh_launch_find_cuts_dbm(const uint64_t *dbm, const uint64_t *bound, const uint64_t *lo,
                              uint32_t n_bounds, uint32_t step, uint64_t *cut, hipStream_t st);
void agh_launch_read_probe(const void *text, uint64_t n, uint32_t *counters, hipStream_t st);
void agh_launch_corpus(void *out, uint64_t first_page, uint64_t n_pages, uint64_t seed,
                       const unsigned char *variants, const uint32_t *vlen,
                       uint32_t n_variants, uint32_t plant_period, uint32_t upper_permille,
                       unsigned long long *planted_dev, hipStream_t st);
void agh_launch_exp(int exp, const void *text, uint64_t n, uint32_t *counters, hipStream_t st);
// ---- record output on the device (agh_records.hip) ----
// ordered compaction of the record bitmap: listed records (set bits; invert: clear bits below AGH_C_NREC) in file
// order -> out_pos[i] = rec_pos[r], out_rec[i] = r for the first `cap` of them; AGH_C_STORED = records listed,
// AGH_C_MATCHED = set bits; the bitmap is left zeroed.  bits: the bits of this scan (= entries of rec_pos; an inverted
// list names no record at or above it).  blk: n_words / 1024 + 2 scratch words
void agh_launch_bitmap_list(uint32_t *bitmap, uint32_t n_words, uint32_t bits, uint32_t *blk, const uint64_t *rec_pos, int invert,
                            uint64_t *out_pos, uint32_t *out_rec, uint32_t cap, uint32_t *counters, hipStream_t st);
// [start, end) of the records around pos[0 .. min(AGH_C_STORED, cap)); AGH_C_RECBYTES += the sum of their lengths
void agh_launch_match_bounds(const void *text, uint64_t n, const agh_dev_query &q, const uint64_t *dbm, const uint64_t *pos,
                             uint32_t *counters, uint32_t cap, uint32_t grid_entries, uint64_t *start, uint64_t *end,
                             hipStream_t st);
void agh_launch_offset_matches(uint64_t *pos, uint32_t *rec, uint64_t *start, uint64_t *end, uint32_t cnt, uint64_t pos_off,
                               uint32_t rec_off, hipStream_t st);
// What a listed record occupies in the output: [start - pre, end) + post_dlen bytes of delimiter behind it, pre =
// min(pre_dlen, bytes of the input in front of the record) -- pre_dlen / post_dlen are 0 or the delimiter length
// (AGH_EMIT_HEAD_DELIM / AGH_EMIT_TAIL_DELIM).  base_off: where text[0] lies in the input.
struct agh_gather_shape {
    uint64_t base_off;
    uint32_t pre_dlen, post_dlen;
    uint8_t dbytes[8];      // the delimiter: appended behind the last record when the input does not end with one
};
// blk[b] = output bytes of the records in front of block b (256 records each) of list entries [first, first + cnt),
// blk[n_blocks] = their total: (cnt + 255) / 256 + 1 words
void agh_launch_len_offsets(const uint64_t *start, const uint64_t *end, uint32_t first, uint32_t cnt, const agh_gather_shape &g,
                            uint64_t *blk, hipStream_t st);
// the bytes of entries [first, first + cnt) back to back at out (NULL: none) and their agh_match triples
// (start, end, index -- shifted by g.base_off / rec_off) at out_matches (NULL: none); blk as left by
// agh_launch_len_offsets, pointing at the block of `first`; n: length of the text
void agh_launch_gather_records(const void *text, uint64_t n, const uint64_t *start, const uint64_t *end, const uint32_t *rec,
                               const uint64_t *blk, uint32_t first, uint32_t cnt, const agh_gather_shape &g, uint64_t rec_off,
                               void *out, void *out_matches, hipStream_t st);
void agh_launch_delim_bitmap(const void *text, uint64_t n, const agh_dev_query &q, uint64_t *dbm,
                             uint64_t n_words, uint32_t *counters, hipStream_t st);

// multi-pattern (-f) device tables: 2^18-bit gram table, bucket directory, bucket items
struct agh_mp_item {
    uint32_t info;      // (pool offset << 8) | length of the entry (a pattern, or a piece of one)
    uint32_t piece;     // entry number | (offset of the probed gram inside the entry << 28)
    uint32_t owner;     // k-error queries: the pattern the piece was cut from
    uint32_t pom;       // ... (offset of the piece inside that pattern << 8) | pattern length
};
struct agh_multi_dev {
    const uint32_t *bits;          // 2^18-bit table of entry grams (two Bloom probes when q == 4)
    const uint32_t *bucket_start;  // (1 << AGH_MP_BUCKET_BITS) + 1 offsets into items
    const agh_mp_item *items;      // grouped by gram bucket
    const uint8_t *pool;           // entry bytes (lower-cased when the query folds case), padded by 16
    const uint32_t *owner_mask;    // k-error queries: [pattern][256] position masks (bit p-1 = position p)
    const uint64_t *dbm;           // this scan's delimiter-end bitmap (delimiters of several bytes, folded letters) or NULL
};
// a.ftab = the bit table; a.tail_only: only the partial last strip (the dense kernel took the rest)
void agh_launch_sweep_multi(const agh_sweep_args &a, hipStream_t st);
// dense hit sets: probes + verification of all full strips in one kernel
void agh_launch_dense_multi(const agh_sweep_args &a, const agh_multi_dev &m, const agh_marks &mk,
                            hipStream_t st);
void agh_launch_verify_multi(const agh_scan_args &a, const agh_multi_dev &m, bool lean,
                             hipStream_t st);
void agh_launch_census_scan(const agh_sweep_args &a, bool with_cand, hipStream_t st);

// one-pass count-only -f scan (agh_mscan.hip): every table it probes on the way lives in LDS
struct agh_mscan_dev {
    const uint2 *ptab;       // pair table, 1 << rb rows (agh_device.h)
    const uint32_t *gtab;    // AGH_MS_GSLOTS key grams (0 = empty), buckets of 4
    const uint4 *ment;       // entries: a whole pattern (k = 0) or a piece + the other side of its pattern (k = 1).  [slot]: the
                             // first entry with the gram of gtab[slot], its gram word = (index of the further ones << 8) | their
                             // number; [AGH_MS_GSLOTS + i]: the further ones
    uint32_t rb;             // log2 rows of ptab: 12 or 13
};
struct agh_mscan_args {
    const void *text;
    uint64_t n;
    agh_dev_query q;
    agh_mscan_dev ms;
    agh_multi_dev mt;        // the general tables: candidates next to the ends of the text
    agh_marks mk;            // hash set + counters
    uint32_t *ticket;        // zeroed work counter in a cache line of its own
    uint32_t n_cu;
    uint32_t dbg;            // AGH_MSCAN_DBG: measurement switches of the kernel (0 in production)
};
bool agh_launch_mscan(const agh_mscan_args &a, hipStream_t st);

// dense -f sets with one error (agh_mtile.hip): pieces of 2..7 bytes, the other side of the pattern <= 7 bytes
struct agh_mwalk_dev {
    const uint4 *ent;        // x: piece bytes 0..3; y: bytes 4..6 | piece length << 24; z: side bytes 0..3 (nearest first);
                             // w: side bytes 4..6 | (side length | 8 if the side lies in front of the piece) << 24
    const uint32_t *dir;     // AGH_MW_DIR slots, (first entry << 16) | entries: a piece of two bytes under agh_mw_slot of
                             // the pair, a longer one under agh_mw_slot3 of its first three bytes
    const uint4 *fmask;      // per agh_mw_slot of a pair, bit (byte & 31): x the byte behind the pair (t[j+2]), y t[j+3], z t[j-1],
                             // w t[j-2] that some entry of two or three bytes starting with the pair can accept at all
    const uint32_t *g4;      // AGH_MW_G4_WORDS: one bit per piece of >= 4 bytes, by its first four bytes
    uint32_t n_ent;
};
struct agh_mwalk_args {
    const void *text;
    uint64_t n;
    agh_dev_query q;
    agh_mwalk_dev mw;
    agh_multi_dev mt;        // the general tables: positions next to the ends of the text
    agh_marks mk;            // hash set + counters
    uint32_t *ticket;
    uint32_t n_cu;
    uint32_t ch;             // tiles a wave holds at a time (1, 2, 4; 0: 2); bits 8..: measurement switches
    const uint32_t *wave_totals = nullptr, *strip_prefix = nullptr;     // numbered scans: the census in front of every strip
};
bool agh_launch_mtile(const agh_mwalk_args &a, hipStream_t st);
// forces the load of the core library's code object (first launch: ~7 ms) -- for a thread that has time for it
void agh_warm_core_module();
