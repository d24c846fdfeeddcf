// agh_records.hip -- record output on the device (what output() / s_output() derive on the CPU while they
// print: agrep.c:3805-3956, sgrep.c:1274-1333): the list of matched records IN FILE ORDER, their bounds and
// their bytes, without a shared counter, a sort or a host round trip in between.
//
//   engines              set bit r of the record bitmap; whoever sets it first also writes rec_pos[r] = one
//                        byte offset inside the record (agh_verify_inl.h mark_record) -- no list, no counter
//   k_bm_block_counts    records to list per block of 32 768 bitmap bits (set bits; -v: clear bits below the
//   k_bm_offsets         record count), exclusive scan of those, totals into the counters
//   k_bm_compact         ordered compaction: the i-th listed record gets slot i (its rank among the bits) --
//                        record numbers ascend, so this IS file order; the bitmap is left zeroed
//   k_match_bounds       slot -> [start, end) of the record around rec_pos (16 bytes per step both ways)
//   k_len_block_sums /   exclusive scan of the record lengths: where each record's bytes go in the output
//   k_len_offsets
//   k_gather_records     the bytes, back to back, and the agh_match entries the caller's emit() receives
//
// Round 4 kept an unordered list behind one atomicAdd per matched record (104 197 updates of one L2 line per
// 4 GiB: 1.2 ms next to a 0.69 ms sweep), copied it to the host three times, sorted it there and sent it back
// for the gather.
#include <stdlib.h>

#include "agh_verify_inl.h"

#define AGH_BM_BLOCK 256u       // threads per block = 16-byte bitmap pieces per block (32 768 records)

__device__ __forceinline__ uint32_t popc4(uint4 v)
{
    return (uint32_t)(__popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w));
}

// records below `nrec` among the 128 of bitmap piece i (nrec = ~0: all of them)
__device__ __forceinline__ uint32_t valid_in_piece(uint64_t i, uint64_t nrec)
{
    const uint64_t lo = i * 128ull;
    return nrec > lo ? (uint32_t)(nrec - lo < 128ull ? nrec - lo : 128ull) : 0u;
}

// exclusive prefix of v over the 256 threads of a block (wave scan + four wave totals through LDS);
// *total = the block's sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wsum, uint32_t *total)
{
    const uint32_t inc = wave_sum_to_lane63(v);
    const uint32_t wv = threadIdx.x / WAVE;
    if (lane_id() == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t before = inc - v;
    for (uint32_t i = 0; i < wv; ++i) before += wsum[i];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return before;
}

// (-v lists) the records a list may name: those below AGH_C_NREC that the scan's bitmap -- and rec_pos, which has
// one entry per bit of it -- covers.  A first attempt on a text with more records than the hint foresaw has
// NREC > bits: the host sees AGH_C_BM_OVERFLOW and reruns, but this launch is already queued and must stay inside.
__device__ __forceinline__ uint64_t listable_records(const uint32_t *__restrict__ counters, uint32_t bits)
{
    const uint32_t nrec = counters[AGH_C_NREC];
    return nrec < bits ? nrec : bits;
}

__global__ __launch_bounds__(AGH_BM_BLOCK) void k_bm_block_counts(const uint4 *__restrict__ bitmap, uint32_t n_vec,
                                                                  uint32_t *__restrict__ blk, int invert,
                                                                  const uint32_t *__restrict__ counters, uint32_t bits)
{
    __shared__ uint32_t wsum[4];
    const uint64_t i = (uint64_t)blockIdx.x * AGH_BM_BLOCK + threadIdx.x;
    uint32_t c = i < n_vec ? popc4(bitmap[i]) : 0u;
    if (invert) c = (i < n_vec ? valid_in_piece(i, listable_records(counters, bits)) : 0u) - c;   // (no bit at or above NREC is ever set)
    uint32_t total;
    (void)block_excl_scan(c, wsum, &total);
    if (threadIdx.x == 0) blk[blockIdx.x] = total;
}

// blk[0..n_blocks) -> exclusive prefix in place, blk[n_blocks] = total; one workgroup
__global__ __launch_bounds__(1024) void k_bm_offsets(uint32_t *__restrict__ blk, uint32_t n_blocks,
                                                     uint32_t *__restrict__ counters, int invert)
{
    __shared__ uint32_t wsum[16];
    uint32_t carry = 0;
    const uint32_t wv = threadIdx.x / WAVE;
    for (uint32_t base = 0; base < n_blocks; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? blk[i] : 0u;
        const uint32_t inc = wave_sum_to_lane63(v);
        if (lane_id() == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t before = carry + inc - v, all = 0;
        for (uint32_t j = 0; j < 16u; ++j) {
            if (j < wv) before += wsum[j];
            all += wsum[j];
        }
        if (i < n_blocks) blk[i] = before;
        carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        blk[n_blocks] = carry;
        counters[AGH_C_STORED] = carry;                                         // records listed (the host clamps to its capacity)
        counters[AGH_C_MATCHED] = invert ? counters[AGH_C_NREC] - carry : carry;   // set bits either way
    }
}

__global__ __launch_bounds__(AGH_BM_BLOCK) void k_bm_compact(uint4 *__restrict__ bitmap, uint32_t n_vec,
                                                             const uint32_t *__restrict__ blk,
                                                             const uint64_t *__restrict__ rec_pos,
                                                             uint64_t *__restrict__ out_pos, uint32_t *__restrict__ out_rec,
                                                             uint32_t cap, int invert,
                                                             const uint32_t *__restrict__ counters, uint32_t bits)
{
    __shared__ uint32_t wsum[4];
    const uint32_t b0 = blk[blockIdx.x], b1 = blk[blockIdx.x + 1];
    const uint64_t i = (uint64_t)blockIdx.x * AGH_BM_BLOCK + threadIdx.x;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (i < n_vec) v = bitmap[i];
    const bool dirty = (v.x | v.y | v.z | v.w) != 0;
    if (dirty) bitmap[i] = make_uint4(0, 0, 0, 0);      // the next scan finds a clean bitmap
    if (b0 == b1) return;                               // (uniform) nothing to list in this block
    const uint64_t nrec = invert ? listable_records(counters, bits) : ~0ull;
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t c = 0;
    if (invert) {
        const uint32_t valid = i < n_vec ? valid_in_piece(i, nrec) : 0u;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t nv = valid > 32u * d ? (valid - 32u * d < 32u ? valid - 32u * d : 32u) : 0u;
            w[d] = ~w[d] & (nv == 32u ? 0xffffffffu : ((1u << nv) - 1u));
        }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) c += (uint32_t)__popc(w[d]);
    uint32_t total;
    uint32_t rank = b0 + block_excl_scan(c, wsum, &total);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t bits = w[d];
        while (bits) {
            const uint32_t bit = (uint32_t)__ffs((int)bits) - 1u;
            bits &= bits - 1u;
            const uint64_t r = i * 128ull + 32u * d + bit;
            if (rank < cap) {
                out_pos[rank] = rec_pos[r];
                if (out_rec) out_rec[rank] = (uint32_t)r;
            }
            ++rank;
        }
    }
}

// n_words: 32-bit words of the bitmap (allocated in 16-byte units; all of them are walked and cleared); bits: the bits
// this scan may have set = the entries of rec_pos; blk: n_words / 1024 + 2 scratch words
void agh_launch_bitmap_list(uint32_t *bitmap, uint32_t n_words, uint32_t bits, uint32_t *blk, const uint64_t *rec_pos, int invert,
                            uint64_t *out_pos, uint32_t *out_rec, uint32_t cap, uint32_t *counters, hipStream_t st)
{
    const uint32_t n_vec = n_words / 4u;
    const uint32_t n_blocks = (n_vec + AGH_BM_BLOCK - 1u) / AGH_BM_BLOCK;
    if (n_blocks)
        hipLaunchKernelGGL(k_bm_block_counts, dim3(n_blocks), dim3(AGH_BM_BLOCK), 0, st, (const uint4 *)bitmap, n_vec, blk,
                           invert, (const uint32_t *)counters, bits);
    hipLaunchKernelGGL(k_bm_offsets, dim3(1), dim3(1024), 0, st, blk, n_blocks, counters, invert);
    if (n_blocks)
        hipLaunchKernelGGL(k_bm_compact, dim3(n_blocks), dim3(AGH_BM_BLOCK), 0, st, (uint4 *)bitmap, n_vec,
                           (const uint32_t *)blk, rec_pos, out_pos, out_rec, cap, invert, (const uint32_t *)counters, bits);
}

// ---------------------------------------------------------------------------------------
// record bounds: [start, end) of the record around pos[i] (start = 1 + the last delimiter in front of pos,
// end = the first delimiter at or behind it), 16 text bytes per step; the sum of the lengths goes to the
// 64-bit counter pair AGH_C_RECBYTES.  The number of entries comes from the device (AGH_C_STORED), so the
// kernel is queued behind the compaction without a host round trip.
// ---------------------------------------------------------------------------------------
typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));

__device__ __forceinline__ uint32_t eq_bytes(uint32_t w, uint32_t dd)     // bit 7 of every byte of w that equals the delimiter
{
    const uint32_t x = w ^ dd;
    return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
}

__device__ __forceinline__ uint64_t record_start_1b(const uint8_t *__restrict__ text, uint64_t pos, uint32_t delim)
{
    const uint32_t dd = delim * 0x01010101u;
    while (pos >= 16) {
        const u32x4_a1 v = *reinterpret_cast<const u32x4_a1 *>(text + pos - 16);
#pragma unroll
        for (int d = 3; d >= 0; --d) {
            const uint32_t z = eq_bytes(v[d], dd);
            if (z) return pos - 16 + 4u * d + (uint32_t)((31 - __clz((int)z)) >> 3) + 1;
        }
        pos -= 16;
    }
    while (pos > 0) {
        if (text[pos - 1] == delim) return pos;
        --pos;
    }
    return 0;
}

__device__ __forceinline__ uint64_t record_end_1b(const uint8_t *__restrict__ text, uint64_t n, uint64_t pos, uint32_t delim)
{
    const uint32_t dd = delim * 0x01010101u;
    while (pos + 16 <= n) {
        const u32x4_a1 v = *reinterpret_cast<const u32x4_a1 *>(text + pos);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t z = eq_bytes(v[d], dd);
            if (z) return pos + 4u * d + (uint32_t)((__ffs((int)z) - 1) >> 3);
        }
        pos += 16;
    }
    while (pos < n) {
        if (text[pos] == delim) return pos;
        ++pos;
    }
    return n;
}

__global__ __launch_bounds__(256) void k_match_bounds(const uint8_t *__restrict__ text, uint64_t n, agh_dev_query q,
                                                      const uint64_t *__restrict__ dbm, const uint64_t *__restrict__ pos,
                                                      uint32_t *__restrict__ counters, uint32_t cap,
                                                      uint64_t *__restrict__ start, uint64_t *__restrict__ end)
{
    __shared__ unsigned long long part[4];
    const uint32_t cnt = counters[AGH_C_STORED] < cap ? counters[AGH_C_STORED] : cap;
    if ((uint64_t)blockIdx.x * 256u >= cnt) return;     // (uniform)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    unsigned long long len = 0;
    if (i < cnt) {
        uint64_t e = pos[i];
        if (e > n) e = n;
        uint64_t s = e, en = e;
        if (q.mb) {
            // record = (end of the last delimiter in front of e, start of the next delimiter]
            const int64_t d = dbm_prev(dbm, e, ~0ull);
            s = d >= 0 ? (uint64_t)d + 1 : 0;
            while (en < n && !dbm_bit(dbm, en)) ++en;      // en = end byte of the next delimiter
            if (en < n) en = en + 1 >= q.dlen ? en + 1 - q.dlen : 0;
            else en = virtual_close_start(text, n, q, dbm);   // closed by the appended delimiter
            if (en < s) en = s;
        } else {
            s = record_start_1b(text, e, q.delim);
            en = record_end_1b(text, n, e, q.delim);
        }
        start[i] = s;
        end[i] = en;
        len = en - s;
    }
    // 64-bit sum over the block: wave reduction through shuffles, one atomic per workgroup
    for (int o = 32; o > 0; o >>= 1) len += __shfl_down(len, o);
    if (lane_id() == 0) part[threadIdx.x / WAVE] = len;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(reinterpret_cast<unsigned long long *>(counters + AGH_C_RECBYTES_LO), t);
    }
}

void agh_launch_match_bounds(const void *text, uint64_t n, const agh_dev_query &q, const uint64_t *dbm, const uint64_t *pos,
                             uint32_t *counters, uint32_t cap, uint32_t grid_entries, uint64_t *start, uint64_t *end,
                             hipStream_t st)
{
    // grid_entries: an upper bound of the entries the host knows (cap, or fewer when it has a better one)
    if (!grid_entries) return;
    hipLaunchKernelGGL(k_match_bounds, dim3((grid_entries + 255u) / 256u), dim3(256), 0, st, (const uint8_t *)text, n, q, dbm,
                       pos, counters, cap, start, end);
}

// Matches of a later segment: positions, bounds and record numbers become absolute.
__global__ __launch_bounds__(256) void k_offset_matches(uint64_t *__restrict__ pos, uint32_t *__restrict__ rec,
                                                        uint64_t *__restrict__ start, uint64_t *__restrict__ end, uint32_t cnt,
                                                        uint64_t pos_off, uint32_t rec_off)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    pos[i] += pos_off;
    if (start) { start[i] += pos_off; end[i] += pos_off; }
    if (rec) rec[i] += rec_off;
}

void agh_launch_offset_matches(uint64_t *pos, uint32_t *rec, uint64_t *start, uint64_t *end, uint32_t cnt, uint64_t pos_off,
                               uint32_t rec_off, hipStream_t st)
{
    if (!cnt || (!pos_off && !rec_off)) return;
    hipLaunchKernelGGL(k_offset_matches, dim3((cnt + 255u) / 256u), dim3(256), 0, st, pos, rec, start, end, cnt, pos_off,
                       rec_off);
}

// ---------------------------------------------------------------------------------------
// where each record's bytes go: exclusive scan of the lengths in blocks of 256 records
// ---------------------------------------------------------------------------------------
// exclusive prefix of 256 64-bit values (one per thread) through LDS; *total = their sum
__device__ __forceinline__ uint64_t block_excl_scan64(uint64_t v, uint64_t *sh, uint64_t *total)
{
    const uint32_t t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (uint32_t o = 1; o < 256u; o <<= 1) {
        const uint64_t add = t >= o ? sh[t - o] : 0ull;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    const uint64_t inc = sh[t];
    *total = sh[255];
    __syncthreads();
    return inc - v;
}

// What a listed record occupies in the output: its bytes, optionally with the delimiter in front of it (as far
// as the input has one there: the first record of a file has none) and the delimiter behind it (always dlen
// bytes: the reference appends the delimiter at the end of the input, asearch.c:87-91) -- the buffer shape
// output() is handed by asearch.c:162-170.
__device__ __forceinline__ uint64_t emit_pre(const agh_gather_shape &g, uint64_t s)
{
    const uint64_t avail = g.base_off + s;              // bytes of the input in front of the record
    return avail < g.pre_dlen ? avail : g.pre_dlen;
}
__device__ __forceinline__ uint64_t emit_len(const agh_gather_shape &g, uint64_t s, uint64_t e)
{
    return emit_pre(g, s) + (e - s) + g.post_dlen;
}

__global__ __launch_bounds__(256) void k_len_block_sums(const uint64_t *__restrict__ start, const uint64_t *__restrict__ end,
                                                        uint32_t first, uint32_t cnt, agh_gather_shape g,
                                                        uint64_t *__restrict__ blk)
{
    __shared__ uint64_t sh[256];
    const uint32_t i = first + blockIdx.x * 256u + threadIdx.x;
    uint64_t total;
    (void)block_excl_scan64(i < first + cnt ? emit_len(g, start[i], end[i]) : 0ull, sh, &total);
    if (threadIdx.x == 0) blk[blockIdx.x] = total;
}

// blk[0..n_blocks) -> exclusive prefix in place, blk[n_blocks] = total; one workgroup
__global__ __launch_bounds__(256) void k_len_offsets(uint64_t *__restrict__ blk, uint32_t n_blocks)
{
    __shared__ uint64_t sh[256];
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n_blocks; base += 256u) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < n_blocks ? blk[i] : 0ull;
        uint64_t total;
        const uint64_t before = block_excl_scan64(v, sh, &total);
        if (i < n_blocks) blk[i] = carry + before;
        carry += total;
    }
    if (threadIdx.x == 0) blk[n_blocks] = carry;
}

// Records [first, first + cnt) of the list: bytes to out[blk-offset ...) back to back (one wave per record,
// 16 bytes per lane and step), and -- if wanted -- their agh_match entries (offsets / record numbers shifted to
// their place in the input).
// (sixteen waves per block of 256 records: the copies of one block's records run side by side -- with four waves a
// block took ~64 dependent load -> store rounds per wave, 36 us for 104 197 records)
#define AGH_GATHER_THREADS 1024u
__global__ __launch_bounds__(AGH_GATHER_THREADS) void k_gather_records(const uint8_t *__restrict__ text, uint64_t n,
                                                        const uint64_t *__restrict__ start,
                                                        const uint64_t *__restrict__ end, const uint32_t *__restrict__ rec,
                                                        const uint64_t *__restrict__ blk, uint32_t first, uint32_t cnt,
                                                        agh_gather_shape g, uint64_t rec_off, uint8_t *__restrict__ out,
                                                        uint64_t *__restrict__ out_matches)
{
    __shared__ uint64_t sh[256], s_start[256], s_len[256], s_off[256];
    const uint32_t t = threadIdx.x;
    const bool scanner = t < 256u;                      // the first four waves own one record each for the prefix sums
    const uint32_t i = first + blockIdx.x * 256u + (scanner ? t : 0u);
    const bool have = scanner && i < first + cnt;
    const uint64_t s = have ? start[i] : 0ull, e = have ? end[i] : 0ull;
    const uint64_t pre = have ? emit_pre(g, s) : 0ull;
    const uint64_t mine = have ? emit_len(g, s, e) : 0ull;
    if (scanner) sh[t] = mine;
    __syncthreads();
    for (uint32_t o = 1; o < 256u; o <<= 1) {
        const uint64_t add = (scanner && t >= o) ? sh[t - o] : 0ull;
        __syncthreads();
        if (scanner) sh[t] += add;
        __syncthreads();
    }
    if (scanner) {
        s_start[t] = s - pre;                           // (in front of the text pointer for the first record of a
        s_len[t] = pre + (e - s);                       //  later segment: the stream keeps those bytes there)
        s_off[t] = blk[blockIdx.x] - blk[0] + (sh[t] - mine);   // (blk[0]: the offset of this piece's first block)
    }
    if (have && out_matches) {
        uint64_t *m = out_matches + 3ull * (i - first);
        m[0] = s + g.base_off;
        m[1] = e + g.base_off;
        m[2] = (rec ? (uint64_t)rec[i] : 0ull) + rec_off;
    }
    __syncthreads();
    if (!out) return;
    const uint32_t lane = (uint32_t)lane_id(), wv = t / WAVE;
    const uint32_t in_block = cnt - blockIdx.x * 256u < 256u ? cnt - blockIdx.x * 256u : 256u;
    for (uint32_t r = wv; r < in_block; r += AGH_GATHER_THREADS / WAVE) {
        const uint8_t *src = text + (int64_t)s_start[r];
        uint8_t *dst = out + s_off[r];
        const uint64_t len = s_len[r];
        uint64_t b = (uint64_t)lane * 16u;
        for (; b + 16 <= len; b += 64u * 16u)
            *reinterpret_cast<u32x4_a1 *>(dst + b) = *reinterpret_cast<const u32x4_a1 *>(src + b);
        // the last len % 16 bytes
        const uint64_t tail = len & ~(uint64_t)15;
        if (lane < (uint32_t)(len - tail)) dst[tail + lane] = src[tail + lane];
        // the delimiter behind the record: from the text, or -- behind the last, unterminated record -- the one the
        // reference appends (the appended bytes continue whatever partial delimiter the text ended with)
        if (lane < g.post_dlen) {
            const uint64_t at = s_start[r] + len + lane;    // = end + lane
            dst[len + lane] = at < n ? text[at] : g.dbytes[(at - n) & 7u];
        }
    }
}

void agh_launch_len_offsets(const uint64_t *start, const uint64_t *end, uint32_t first, uint32_t cnt, const agh_gather_shape &g,
                            uint64_t *blk, hipStream_t st)
{
    const uint32_t n_blocks = (cnt + 255u) / 256u;
    if (n_blocks)
        hipLaunchKernelGGL(k_len_block_sums, dim3(n_blocks), dim3(256), 0, st, start, end, first, cnt, g, blk);
    hipLaunchKernelGGL(k_len_offsets, dim3(1), dim3(256), 0, st, blk, n_blocks);
}

// blk: the offsets agh_launch_len_offsets left for the SAME (first, cnt), or a sub-range of whole blocks of it
// (blk pointing at the sub-range's first block): the output then starts at that block's offset
void agh_launch_gather_records(const void *text, uint64_t n, const uint64_t *start, const uint64_t *end, const uint32_t *rec,
                               const uint64_t *blk, uint32_t first, uint32_t cnt, const agh_gather_shape &g, uint64_t rec_off,
                               void *out, void *out_matches, hipStream_t st)
{
    if (!cnt) return;
    hipLaunchKernelGGL(k_gather_records, dim3((cnt + 255u) / 256u), dim3(AGH_GATHER_THREADS), 0, st, (const uint8_t *)text, n, start, end, rec,
                       blk, first, cnt, g, rec_off, (uint8_t *)out, (uint64_t *)out_matches);
}
