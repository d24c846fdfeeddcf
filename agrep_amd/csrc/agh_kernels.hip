// agh_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the agrep record scanner.
//
// Data flow of one scan (all buffers in HBM, text read with 16 B/lane coalesced loads):
//
//   k_sweep<H>      streams every text byte once.  Per 1 KiB strip it counts the record
//                   delimiters (SWAR zero-byte test + v_bcnt) and, when the query admits the
//                   q-gram sample filter, probes one q-byte sample every H bytes against a
//                   32 KiB hash table held in LDS; samples that hit are appended to the
//                   candidate list (wave-aggregated atomics).  HBM-bound: this is the
//                   kernel the roofline is quoted on.
//   k_wave_scan     exclusive scan of the per-wave delimiter totals (tiny).
//   k_verify<W,K>   one lane per candidate: the Wu-Manber k-error shift-AND automaton
//                   (asearch.c:94-116 restated with left shifts, delimiter out of band) over
//                   the <= 2(m+k)+q bytes around the sample; every match is turned into a
//                   record number and marked in a one-bit-per-record bitmap, so a record is
//                   counted once however many windows or occurrences hit it.
//   k_fullscan<W,K> the same automaton over every byte (the asearch.c shape), for queries the
//                   filter cannot serve.  Text is staged through LDS so that each lane walks
//                   a contiguous 256 B chunk while global loads stay coalesced; a lane starts
//                   m+k+1 bytes early to rebuild the automaton state (bounded memory,
//                   SURVEY.md B.5).
//
// No MFMA anywhere: the work is byte/bitwise integer and the bound is HBM read bandwidth.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "agh_device.h"
#include "agh_launch.h"

#define WAVE 64

// ---------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (WAVE - 1)); }

// popcount of the "non-zero byte" mask of (w ^ dd): 28 fixed bits + one bit per byte that
// is NOT the delimiter.  zero bytes of one dword = 32 - result.
__device__ __forceinline__ uint32_t nz_popc(uint32_t w, uint32_t dd)
{
    uint32_t x = w ^ dd;
    uint32_t t = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu;
    return (uint32_t)__popc(t);
}

__device__ __forceinline__ uint32_t delims_in(uint4 v, uint32_t dd)
{
    return 128u - (nz_popc(v.x, dd) + nz_popc(v.y, dd) + nz_popc(v.z, dd) + nz_popc(v.w, dd));
}

// Replace the bytes of a 16-byte chunk at index >= keep by `fill` (used at the text end).
__device__ __forceinline__ uint4 mask_tail(uint4 v, int keep, uint32_t fill4)
{
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        int kb = keep - 4 * d;               // bytes of this dword to keep
        if (kb <= 0) w[d] = fill4;
        else if (kb < 4) {
            uint32_t m = (1u << (8 * kb)) - 1u;
            w[d] = (w[d] & m) | (fill4 & ~m);
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Wave-wide sum with DPP row shifts / row broadcasts (gfx9 family); total lands in lane 63.
__device__ __forceinline__ uint32_t wave_sum_to_lane63(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31
    return v;
}

// ---------------------------------------------------------------------------------------
// sweep: delimiter census + q-gram sample filter
// ---------------------------------------------------------------------------------------
template <int H>
__device__ __forceinline__ uint32_t probe(uint32_t w, const agh_dev_query &q,
                                          const uint8_t *ftab)
{
    uint32_t s = (w & q.qmask) | q.fold;
    return ftab[agh_sample_hash(s)];
}

// One 16-byte chunk: accumulate the non-delimiter popcount and the sample hit bits.
template <int H>
__device__ __forceinline__ void sweep_chunk(uint4 v, uint32_t dd, const agh_dev_query &q,
                                            const uint8_t *ftab, uint32_t &acc,
                                            uint32_t &hits, int bitbase)
{
    acc += nz_popc(v.x, dd) + nz_popc(v.y, dd) + nz_popc(v.z, dd) + nz_popc(v.w, dd);
    if (H > 0) {
        hits |= probe<H>(v.x, q, ftab) << bitbase;
        if (H <= 8) hits |= probe<H>(v.z, q, ftab) << (bitbase + 2);
        if (H <= 4) {
            hits |= probe<H>(v.y, q, ftab) << (bitbase + 1);
            hits |= probe<H>(v.w, q, ftab) << (bitbase + 3);
        }
    }
}

// Append the candidates of one wave to the wave's private slice of the candidate buffer (no
// atomics: one hot global counter saturates at ~90 updates/us on this chip and would cap the
// whole sweep).  hits: bit (4*u + d) of lane l = sample at dword d of the lane's chunk in
// strip s+u.  rc[u] = delimiters (inside this wave's range) in front of the lane's chunk of
// strip s+u -- stored with the candidate so that the verifier can number records without
// re-reading any text.  cnt is wave-uniform.
__device__ __forceinline__ void emit_candidates(uint32_t hits, uint64_t s, const uint32_t rc[4],
                                                uint64_t *__restrict__ slice, uint32_t &cnt,
                                                uint32_t *counters)
{
    uint64_t hm = __ballot(hits != 0);
    const int lane = lane_id();
    while (hm) {
        int l = __ffsll((long long)hm) - 1;
        hm &= hm - 1;
        uint32_t hbits = (uint32_t)__builtin_amdgcn_readlane((int)hits, l);
        uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)rc[0], l);
        uint32_t r1 = (uint32_t)__builtin_amdgcn_readlane((int)rc[1], l);
        uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)rc[2], l);
        uint32_t r3 = (uint32_t)__builtin_amdgcn_readlane((int)rc[3], l);
        int c = __popc(hbits);
        if (lane < c) {
            uint32_t t = hbits;
            for (int j = 0; j < lane; ++j) t &= t - 1;
            int b = __ffs((int)t) - 1;
            int u = b >> 2;
            uint32_t dw = (uint32_t)(((s + (uint64_t)u) * 64u + (uint64_t)l) * 4u +
                                     (uint64_t)(b & 3));
            uint32_t r = u == 0 ? r0 : (u == 1 ? r1 : (u == 2 ? r2 : r3));
            uint32_t idx = cnt + (uint32_t)lane;
            if (idx < AGH_SLICE_CAP) slice[idx] = ((uint64_t)r << 32) | dw;
            else counters[AGH_C_OVERFLOW] = 1u;
        }
        cnt += (uint32_t)c;
    }
}

// grid: ceil(n_waves / 4) workgroups of 256 threads; wave w owns strips
// [w*AGH_WAVE_STRIPS, min((w+1)*AGH_WAVE_STRIPS, n_full_strips)).
template <int H>
__global__ __launch_bounds__(256) void k_sweep(const uint4 *__restrict__ text,
                                               uint64_t n_full_strips, agh_dev_query q,
                                               const uint8_t *__restrict__ ftab_g,
                                               uint32_t *__restrict__ strip_prefix,
                                               uint32_t *__restrict__ wave_totals,
                                               uint64_t *__restrict__ cand,
                                               uint32_t *__restrict__ wave_cand,
                                               uint32_t *__restrict__ counters)
{
    __shared__ __attribute__((aligned(16))) uint8_t ftab[H > 0 ? AGH_FT_SIZE : 16];
    if (H > 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(ftab_g);
        uint4 *dst = reinterpret_cast<uint4 *>(ftab);
        for (uint32_t i = threadIdx.x; i < AGH_FT_SIZE / 16; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const int lane = lane_id();
    const uint64_t w = (uint64_t)blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE);
    const uint64_t s0 = w * AGH_WAVE_STRIPS;
    if (s0 >= n_full_strips) return;
    uint64_t s1 = s0 + AGH_WAVE_STRIPS;
    if (s1 > n_full_strips) s1 = n_full_strips;
    const uint32_t dd = q.delim * 0x01010101u;
    uint32_t run = 0;                           // delimiters before strip s inside this range
    uint32_t ncand = 0;                         // candidates in this wave's slice
    uint64_t *slice = cand + w * AGH_SLICE_CAP;
    uint64_t s = s0;

    for (; s + 4 <= s1; s += 4) {
        const uint4 *p = text + s * 64 + lane;
        uint4 v0 = p[0], v1 = p[64], v2 = p[128], v3 = p[192];   // 4 x 1 KiB per wave in flight
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, hits = 0;
        sweep_chunk<H>(v0, dd, q, ftab, a0, hits, 0);
        sweep_chunk<H>(v1, dd, q, ftab, a1, hits, 4);
        sweep_chunk<H>(v2, dd, q, ftab, a2, hits, 8);
        sweep_chunk<H>(v3, dd, q, ftab, a3, hits, 12);
        // per-strip delimiter totals: two packed 16-bit sums per DPP scan (the scan is a
        // full inclusive prefix over the 64 lanes; lane 63 holds the totals)
        const uint32_t own01 = a0 | (a1 << 16), own23 = a2 | (a3 << 16);
        const uint32_t sc01 = wave_sum_to_lane63(own01);
        const uint32_t sc23 = wave_sum_to_lane63(own23);
        const uint32_t p01 = (uint32_t)__builtin_amdgcn_readlane((int)sc01, 63);
        const uint32_t p23 = (uint32_t)__builtin_amdgcn_readlane((int)sc23, 63);
        uint32_t z0 = 8192u - (p01 & 0xffffu), z1 = 8192u - (p01 >> 16);
        uint32_t z2 = 8192u - (p23 & 0xffffu), z3 = 8192u - (p23 >> 16);
        if (lane == 0)
            *reinterpret_cast<uint4 *>(strip_prefix + s) =
                make_uint4(run, run + z0, run + z0 + z1, run + z0 + z1 + z2);
        if (H > 0 && __ballot(hits != 0)) {
            // delimiters in front of my chunk: 128*lane minus the non-delimiter popcounts of
            // the lanes before me (exclusive prefix = inclusive scan - own)
            const uint32_t ex01 = sc01 - own01, ex23 = sc23 - own23;
            const uint32_t lb = 128u * (uint32_t)lane;
            uint32_t rc[4];
            rc[0] = run + lb - (ex01 & 0xffffu);
            rc[1] = run + z0 + lb - (ex01 >> 16);
            rc[2] = run + z0 + z1 + lb - (ex23 & 0xffffu);
            rc[3] = run + z0 + z1 + z2 + lb - (ex23 >> 16);
            emit_candidates(hits, s, rc, slice, ncand, counters);
        }
        run += z0 + z1 + z2 + z3;
    }
    for (; s < s1; ++s) {                       // < 4 strips left in the range
        uint4 v0 = text[s * 64 + lane];
        uint32_t a0 = 0, hits = 0;
        sweep_chunk<H>(v0, dd, q, ftab, a0, hits, 0);
        const uint32_t sc0 = wave_sum_to_lane63(a0);
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)sc0, 63);
        if (lane == 0) strip_prefix[s] = run;
        if (H > 0 && __ballot(hits != 0)) {
            uint32_t rc[4];
            rc[0] = run + 128u * (uint32_t)lane - (sc0 - a0);
            rc[1] = rc[2] = rc[3] = 0;
            emit_candidates(hits, s, rc, slice, ncand, counters);
        }
        run += 8192u - p0;
    }
    if (lane == 0) {
        wave_totals[w] = run;
        if (H > 0) wave_cand[w] = ncand < AGH_SLICE_CAP ? ncand : AGH_SLICE_CAP;
    }
}

// The last, partial strip (n % 1024 != 0): one wave, bytes >= n masked to a non-delimiter.
// Runs after k_sweep on the same stream.
template <int H>
__global__ __launch_bounds__(64) void k_sweep_tail(const uint4 *__restrict__ text, uint64_t n,
                                                   agh_dev_query q,
                                                   const uint8_t *__restrict__ ftab_g,
                                                   uint32_t *__restrict__ strip_prefix,
                                                   uint32_t *__restrict__ wave_totals,
                                                   uint64_t *__restrict__ cand,
                                                   uint32_t *__restrict__ wave_cand,
                                                   uint32_t *__restrict__ counters)
{
    const int lane = lane_id();
    const uint64_t s = n >> AGH_STRIP_SHIFT;            // index of the partial strip
    const uint64_t off = (s << AGH_STRIP_SHIFT) + (uint64_t)lane * 16u;
    const uint32_t dd = q.delim * 0x01010101u;
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    uint4 v = make_uint4(fill4, fill4, fill4, fill4);
    if (off < n) {
        v = text[off >> 4];
        if (off + 16 > n) v = mask_tail(v, (int)(n - off), fill4);
    }
    uint32_t a0 = 0, hits = 0;
    sweep_chunk<H>(v, dd, q, ftab_g, a0, hits, 0);      // table straight from global/L2
    const uint32_t sc0 = wave_sum_to_lane63(a0);
    const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)sc0, 63);
    const uint32_t z = 8192u - p0;
    const uint64_t w = s / AGH_WAVE_STRIPS;
    const bool fresh = (s % AGH_WAVE_STRIPS) == 0;      // k_sweep never touched this range
    uint32_t before = fresh ? 0u : wave_totals[w];
    before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
    if (H > 0) {
        uint32_t ncand = fresh ? 0u : wave_cand[w];
        ncand = (uint32_t)__builtin_amdgcn_readfirstlane((int)ncand);
        if (__ballot(hits != 0)) {
            uint32_t rc[4];
            rc[0] = before + 128u * (uint32_t)lane - (sc0 - a0);
            rc[1] = rc[2] = rc[3] = 0;
            emit_candidates(hits, s, rc, cand + w * AGH_SLICE_CAP, ncand, counters);
        }
        if (lane == 0) wave_cand[w] = ncand < AGH_SLICE_CAP ? ncand : AGH_SLICE_CAP;
    }
    if (lane == 0) {
        strip_prefix[s] = before;
        wave_totals[w] = before + z;
    }
}

// Exclusive scans (one workgroup of 1024 threads, array staged in LDS so that the global
// traffic is two coalesced passes): wave_totals[0..nw) in place -> delimiter prefix per wave
// range; wave_cand[0..nw) -> total candidate count.  Also records the last text byte.
#define AGH_SCAN_MAX 32768u   // 8 GiB segment / 256 KiB per wave range

__device__ uint32_t block_exscan(const uint32_t *in, uint32_t *out, uint32_t nw, uint32_t *buf,
                                 uint32_t *part)
{
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nw + 1023u) / 1024u;
    for (uint32_t i = t; i < per * 1024u; i += 1024u) buf[i] = i < nw ? in[i] : 0u;
    __syncthreads();
    uint32_t sum = 0;
    for (uint32_t i = 0; i < per; ++i) sum += buf[t * per + i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {           // Hillis-Steele inclusive scan
        uint32_t v = (t >= d) ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t i = 0; i < per; ++i) {
        uint32_t v = buf[t * per + i];
        buf[t * per + i] = run;
        run += v;
    }
    const uint32_t total = part[1023];
    __syncthreads();
    if (out)
        for (uint32_t i = t; i < nw; i += 1024u) out[i] = buf[i];
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(1024) void k_wave_scan(uint32_t *__restrict__ wave_totals,
                                                    const uint32_t *__restrict__ wave_cand,
                                                    uint32_t nw, const uint8_t *text,
                                                    uint64_t n, uint32_t *__restrict__ counters)
{
    extern __shared__ uint32_t scan_lds[];
    uint32_t *part = scan_lds;
    uint32_t *buf = scan_lds + 1024;
    uint32_t nd = block_exscan(wave_totals, wave_totals, nw, buf, part);
    uint32_t nc = wave_cand ? block_exscan(wave_cand, nullptr, nw, buf, part) : 0u;
    if (threadIdx.x == 0) {
        counters[AGH_C_NDELIM] = nd;
        counters[AGH_C_CAND] = nc;
        counters[AGH_C_LASTBYTE] = n ? (uint32_t)text[n - 1] : 0xffffffffu;
    }
}

// ---------------------------------------------------------------------------------------
// record bookkeeping shared by verify and fullscan
// ---------------------------------------------------------------------------------------
// Record r has a match whose last byte is at e: set its bit.  The number of matched records
// is the population count of the bitmap (k_bitmap_count) -- a shared "matched" counter
// would serialise on one L2 atomic unit (~90 updates/us) and dominate the scan.
__device__ __forceinline__ void mark_record(const agh_marks &mk, uint32_t r, uint64_t e)
{
    const uint32_t bit = 1u << (r & 31u);
    uint32_t old = atomicOr(&mk.bitmap[r >> 5], bit);
    if (mk.match_pos && !(old & bit)) {
        uint32_t idx = atomicAdd(&mk.counters[AGH_C_STORED], 1u);
        if (idx < mk.match_cap) {
            mk.match_pos[idx] = e;
            if (mk.match_rec) mk.match_rec[idx] = r;
        }
    }
}

__global__ __launch_bounds__(256) void k_bitmap_count(const uint32_t *__restrict__ bitmap,
                                                      uint32_t n_words,
                                                      uint32_t *__restrict__ counters)
{
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words;
         i += gridDim.x * blockDim.x)
        acc += (uint32_t)__popc(bitmap[i]);
    acc = wave_sum_to_lane63(acc);
    __shared__ uint32_t part[4];
    if (lane_id() == 63) part[threadIdx.x / WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(&counters[AGH_C_MATCHED], t);
    }
}

// Number of delimiters at byte positions < e (= 0-based record number of position e).
__device__ uint32_t record_of(const uint8_t *__restrict__ text, uint64_t n, uint64_t e,
                              const uint32_t *__restrict__ strip_prefix,
                              const uint32_t *__restrict__ wave_prefix, uint32_t n_strips,
                              uint32_t total_delims, uint32_t delim)
{
    const uint64_t strip = e >> AGH_STRIP_SHIFT;
    if (strip >= n_strips) return total_delims;
    uint32_t r = wave_prefix[strip / AGH_WAVE_STRIPS] + strip_prefix[strip];
    const uint32_t dd = delim * 0x01010101u;
    const uint32_t fill4 = (~delim & 0xffu) * 0x01010101u;
    const uint64_t p = strip << AGH_STRIP_SHIFT;
    const uint4 *t4 = reinterpret_cast<const uint4 *>(text + p);
    const uint32_t span = (uint32_t)(e - p);
    const uint32_t full = span >> 4;
    for (uint32_t i = 0; i < full; ++i) r += delims_in(t4[i], dd);
    if (span & 15u) r += delims_in(mask_tail(t4[full], (int)(span & 15u), fill4), dd);
    return r;
}

// ---------------------------------------------------------------------------------------
// the k-error automaton (asearch.c:94-116 mirrored to left shifts, 1 = active)
//   R0' = ((R0 << 1) | 1) & CM
//   Re' = (((Re << 1) | 1) & CM) | R(e-1) | (((R(e-1) | R(e-1)') << 1) | 1)
// reset state = all zeros (asearch.c:175-186 resets to Init[0] = "nothing but the start
// state" and re-feeds the delimiter byte; the leading-deletion bits appear through the
// recurrence itself on that first step).
// ---------------------------------------------------------------------------------------
template <typename WT, int K>
struct Automaton {
    WT R[K + 1];
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int e = 0; e <= K; ++e) R[e] = 0;
    }
    __device__ __forceinline__ bool step(WT cm, WT finalbit)
    {
        WT po = R[0];
        WT pn = ((po << 1) | (WT)1) & cm;
        R[0] = pn;
#pragma unroll
        for (int e = 1; e <= K; ++e) {
            WT cur = R[e];
            WT ne = (((cur << 1) | (WT)1) & cm) | po | (((po | pn) << 1) | (WT)1);
            po = cur;
            pn = ne;
            R[e] = ne;
        }
        return (R[K] & finalbit) != 0;
    }
};

// ---------------------------------------------------------------------------------------
// verify: one hardware wave per sweep-wave slice, one lane per candidate sample
// ---------------------------------------------------------------------------------------
// The window [a0, we) around a sample is fetched with NCH aligned 16-byte loads issued
// together (one memory round trip), then walked byte by byte out of registers.  Up to four
// match events per window are kept in registers and resolved after the walk; a window with
// more (records of a few bytes) is re-walked by the byte-wise slow path.
template <typename WT, int K>
__device__ __noinline__ void verify_window_slow(const uint8_t *__restrict__ text, uint64_t n,
                                                const agh_dev_query &q, const WT *lmask,
                                                uint64_t ws, uint64_t we, uint64_t anchor,
                                                uint32_t rc_anchor, const agh_marks &mk,
                                                uint32_t total_delims)
{
    const WT finalbit = (WT)1 << (q.m - 1);
    // delimiters in [ws, anchor): the anchor's record number is known, ws's is derived
    uint32_t back = 0;
    for (uint64_t i = ws; i < anchor; ++i) back += (text[i] == q.delim);
    uint32_t rec = rc_anchor - back;
    Automaton<WT, K> A;
    A.reset();
    bool seen = false;
    if (ws == 0) A.step(lmask[q.head_byte], finalbit);
    for (uint64_t i = ws; i < we; ++i) {
        const uint32_t c = text[i];
        if (A.step(lmask[c], finalbit) && !seen) { seen = true; mark_record(mk, rec, i); }
        if (c == q.delim) {
            A.reset();
            ++rec;
            seen = false;
            if (A.step(lmask[c], finalbit)) { seen = true; mark_record(mk, rec, i + 1); }
        }
    }
    if (we == n && q.tail_virtual) {
        if (A.step(lmask[q.delim], finalbit) && !seen) mark_record(mk, rec, n);
        A.reset();
        if (A.step(lmask[q.delim], finalbit)) mark_record(mk, rec + 1u, n);
    }
    (void)total_delims;
}

template <typename WT, int K, int NCH>
__global__ __launch_bounds__(256) void k_verify(const uint8_t *__restrict__ text, uint64_t n,
                                                agh_dev_query q,
                                                const WT *__restrict__ mask_g,
                                                const uint64_t *__restrict__ cand,
                                                const uint32_t *__restrict__ wave_cand,
                                                const uint32_t *__restrict__ wave_prefix,
                                                uint32_t nw, agh_marks mk)
{
    __shared__ WT lmask[256];
    lmask[threadIdx.x] = mask_g[threadIdx.x];
    __syncthreads();
    const uint32_t total_delims = mk.counters[AGH_C_NDELIM];
    const WT finalbit = (WT)1 << (q.m - 1);
    const uint32_t L = (uint32_t)(q.m + q.k + 1);
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    const int lane = lane_id();

    for (uint32_t w = blockIdx.x * 4 + threadIdx.x / WAVE; w < nw; w += gridDim.x * 4) {
        const uint32_t cnt = wave_cand[w];
        const uint64_t *slice = cand + (uint64_t)w * AGH_SLICE_CAP;
        const uint32_t wp = wave_prefix[w];
        for (uint32_t base = 0; base < cnt; base += WAVE) {
            const uint32_t ci = base + (uint32_t)lane;
            if (ci >= cnt) continue;
            const uint64_t ent = slice[ci];
            const uint64_t j = (ent & 0xffffffffull) * 4u;
            if (j >= n) continue;
            const uint32_t rc_anchor = wp + (uint32_t)(ent >> 32);   // record number at `anchor`
            const uint64_t anchor = j & ~(uint64_t)15;               // the sample's chunk start
            uint64_t ws = j > L ? j - L : 0;
            if (anchor < ws) ws = anchor;
            const uint64_t a0 = ws & ~(uint64_t)15;
            uint64_t we = j + (uint64_t)q.fq + (uint64_t)(q.m + q.k);
            if (we > n) we = n;

            uint4 ch[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint64_t g = a0 + 16u * (uint64_t)c;
                ch[c] = (g < n16 && g < we) ? *reinterpret_cast<const uint4 *>(text + g)
                                            : make_uint4(fill4, fill4, fill4, fill4);
            }

            Automaton<WT, K> A;
            A.reset();
            bool seen = false;
            if (ws == 0) A.step(lmask[q.head_byte], finalbit);
            uint32_t cd = 0;                    // delimiters seen in [ws, i)
            uint32_t cd_anchor = 0;             // ... in [ws, anchor)
            uint32_t ev0 = 0, ev1 = 0, ev2 = 0, ev3 = 0, nev = 0;   // events: cd-relative record
            uint32_t ep0 = 0, ep1 = 0, ep2 = 0, ep3 = 0;            // and position (a0-relative)
#define AGH_PUSH_EVENT(RELREC, RELPOS)                       \
    do {                                                     \
        ev3 = ev2; ev2 = ev1; ev1 = ev0; ev0 = (RELREC);     \
        ep3 = ep2; ep2 = ep1; ep1 = ep0; ep0 = (RELPOS);     \
        ++nev;                                               \
    } while (0)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint32_t dws[4] = {ch[c].x, ch[c].y, ch[c].z, ch[c].w};
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const uint32_t rel = (uint32_t)(16 * c + b);
                    const uint64_t i = a0 + rel;
                    if (i >= ws && i < we) {
                        const uint32_t byte = (dws[b >> 2] >> (8 * (b & 3))) & 0xffu;
                        if (i == anchor) cd_anchor = cd;
                        if (A.step(lmask[byte], finalbit) && !seen) {
                            seen = true;
                            AGH_PUSH_EVENT(cd, rel);
                        }
                        if (byte == q.delim) {
                            A.reset();
                            ++cd;
                            seen = false;
                            if (A.step(lmask[byte], finalbit)) {
                                seen = true;
                                AGH_PUSH_EVENT(cd, rel + 1u);
                            }
                        }
                    }
                }
            }
            if (anchor >= we) cd_anchor = cd;   // (cannot happen: the sample lies in the window)
            if (we == n && q.tail_virtual) {    // asearch.c:87-91: delimiter appended at EOF
                if (A.step(lmask[q.delim], finalbit) && !seen)
                    AGH_PUSH_EVENT(cd, (uint32_t)(n - a0));
                A.reset();
                if (A.step(lmask[q.delim], finalbit)) AGH_PUSH_EVENT(cd + 1u, (uint32_t)(n - a0));
            }
#undef AGH_PUSH_EVENT
            if (nev > 4) {
                verify_window_slow<WT, K>(text, n, q, lmask, ws, we, anchor, rc_anchor, mk,
                                          total_delims);
            } else {
                const uint32_t r0 = rc_anchor - cd_anchor;
                if (nev > 0) mark_record(mk, r0 + ev0, a0 + ep0);
                if (nev > 1) mark_record(mk, r0 + ev1, a0 + ep1);
                if (nev > 2) mark_record(mk, r0 + ev2, a0 + ep2);
                if (nev > 3) mark_record(mk, r0 + ev3, a0 + ep3);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// fullscan: the automaton over every byte
// ---------------------------------------------------------------------------------------
template <typename WT, int K>
__global__ __launch_bounds__(AGH_FS_THREADS) void k_fullscan(
    const uint8_t *__restrict__ text, uint64_t n, agh_dev_query q,
    const WT *__restrict__ mask_g, const uint32_t *__restrict__ strip_prefix,
    const uint32_t *__restrict__ wave_prefix, uint32_t n_strips, agh_marks mk)
{
    // slot 0 = the 256 bytes in front of the tile (warm-up halo), slots 1..256 = lane chunks
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    WT *lmask = reinterpret_cast<WT *>(lds);
    uint8_t *tile = lds + 256 * sizeof(WT);
    lmask[threadIdx.x] = mask_g[threadIdx.x];

    const uint64_t tile_bytes = (uint64_t)AGH_FS_THREADS * AGH_FS_CHUNK;
    const uint64_t n_tiles = (n + tile_bytes - 1) / tile_bytes;
    const uint32_t total_delims = mk.counters[AGH_C_NDELIM];
    const WT finalbit = (WT)1 << (q.m - 1);
    const uint32_t dd = q.delim * 0x01010101u;
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    const uint32_t warm = ((uint32_t)(q.m + q.k + 1) + 15u) & ~15u;   // <= 80 bytes

    for (uint64_t tix = blockIdx.x; tix < n_tiles; tix += gridDim.x) {
        const uint64_t t0 = tix * tile_bytes;
        __syncthreads();
        // cooperative, coalesced global -> LDS of 257 chunks (halo + tile), 16 B pieces
        for (uint32_t pc = threadIdx.x; pc < (AGH_FS_THREADS + 1) * (AGH_FS_CHUNK / 16);
             pc += AGH_FS_THREADS) {
            const uint32_t slot = pc / (AGH_FS_CHUNK / 16), sub = pc % (AGH_FS_CHUNK / 16);
            const int64_t g = (int64_t)t0 - (int64_t)AGH_FS_CHUNK + (int64_t)pc * 16;
            uint4 v = make_uint4(fill4, fill4, fill4, fill4);
            if (g >= 0 && (uint64_t)g < n16) v = *reinterpret_cast<const uint4 *>(text + g);
            *reinterpret_cast<uint4 *>(tile + slot * AGH_FS_SLOT + sub * 16) = v;
        }
        __syncthreads();

        const uint64_t cs = t0 + (uint64_t)threadIdx.x * AGH_FS_CHUNK;
        const uint8_t *mine = tile + (threadIdx.x + 1) * AGH_FS_SLOT;
        uint32_t my_delims = 0;
        uint64_t ce = cs + AGH_FS_CHUNK;
        if (ce > n) ce = n;
        if (cs < n) {
            const uint32_t len = (uint32_t)(ce - cs);
            for (uint32_t i = 0; i < (len >> 4); ++i)
                my_delims += delims_in(*reinterpret_cast<const uint4 *>(mine + i * 16), dd);
            if (len & 15u)
                my_delims += delims_in(
                    mask_tail(*reinterpret_cast<const uint4 *>(mine + (len & ~15u)),
                              (int)(len & 15u), fill4), dd);
        }
        // delimiters of the preceding chunks of my 1 KiB strip (4 lanes per strip)
        uint32_t before = 0;
        {
            const int l4 = (int)(threadIdx.x & 3u);
            uint32_t v1 = (uint32_t)__shfl_up((int)my_delims, 1, 4);
            uint32_t v2 = (uint32_t)__shfl_up((int)my_delims, 2, 4);
            uint32_t v3 = (uint32_t)__shfl_up((int)my_delims, 3, 4);
            if (l4 >= 1) before += v1;
            if (l4 >= 2) before += v2;
            if (l4 >= 3) before += v3;
        }
        if (cs >= n) continue;

        const uint64_t strip = cs >> AGH_STRIP_SHIFT;
        uint32_t rec = (strip < n_strips)
                           ? wave_prefix[strip / AGH_WAVE_STRIPS] + strip_prefix[strip] + before
                           : total_delims;

        Automaton<WT, K> A;
        A.reset();
        if (cs == 0) {
            A.step(lmask[q.head_byte], finalbit); // asearch.c:69-78
        } else {
            const uint8_t *halo = tile + threadIdx.x * AGH_FS_SLOT + (AGH_FS_CHUNK - warm);
            for (uint32_t i = 0; i < warm; ++i) {
                const uint32_t c = halo[i];
                A.step(lmask[c], finalbit);
                if (c == q.delim) {
                    A.reset();
                    A.step(lmask[c], finalbit);
                }
            }
        }
        bool seen = false;
        const uint32_t len = (uint32_t)(ce - cs);
        for (uint32_t i = 0; i < len; ++i) {
            const uint32_t c = mine[i];
            bool hit = A.step(lmask[c], finalbit);
            if (hit && !seen) {
                seen = true;
                mark_record(mk, rec, cs + i);
            }
            if (c == q.delim) {
                A.reset();
                ++rec;
                seen = false;
                if (A.step(lmask[c], finalbit)) {
                    seen = true;
                    mark_record(mk, rec, cs + i + 1);
                }
            }
        }
        if (ce == n && q.tail_virtual) {        // asearch.c:87-91
            if (A.step(lmask[q.delim], finalbit) && !seen) mark_record(mk, rec, n);
            A.reset();
            if (A.step(lmask[q.delim], finalbit)) mark_record(mk, rec + 1u, n);
        }
    }
}

// ---------------------------------------------------------------------------------------
// bench support: read probe and synthetic corpus
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_read_probe(const uint4 *__restrict__ text,
                                                    uint64_t n_strips,
                                                    uint32_t *__restrict__ counters)
{
    const int lane = lane_id();
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x / WAVE);
    const uint64_t s0 = w * AGH_WAVE_STRIPS;
    if (s0 >= n_strips) return;
    uint64_t s1 = s0 + AGH_WAVE_STRIPS;
    if (s1 > n_strips) s1 = n_strips;
    uint32_t acc = 0;
    uint64_t s = s0;
    for (; s + 4 <= s1; s += 4) {
        const uint4 *p = text + s * 64 + lane;
        uint4 v0 = p[0], v1 = p[64], v2 = p[128], v3 = p[192];
        acc ^= v0.x ^ v0.y ^ v0.z ^ v0.w ^ v1.x ^ v1.y ^ v1.z ^ v1.w;
        acc ^= v2.x ^ v2.y ^ v2.z ^ v2.w ^ v3.x ^ v3.y ^ v3.z ^ v3.w;
    }
    for (; s < s1; ++s) {
        uint4 v0 = text[s * 64 + lane];
        acc ^= v0.x ^ v0.y ^ v0.z ^ v0.w;
    }
    if (acc == 0x9e3779b9u) counters[AGH_C_CHECK] = acc;   // keeps the loads alive
}

struct agh_corpus_params {
    uint64_t seed;
    uint32_t n_variants, plant_period, upper_permille;
    uint32_t vlen[8];
    uint8_t variants[8][80];
};

__device__ __forceinline__ uint64_t cg_next(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// One thread per 4096-byte page; bytes are produced strictly in order and stored 8 at a time.
// Twin of oracle/corpus_gen.c:cg_page (tests assert byte equality).
__global__ __launch_bounds__(64) void k_corpus(uint64_t *__restrict__ out, uint64_t first_page,
                                               uint64_t n_pages, agh_corpus_params p,
                                               unsigned long long *__restrict__ planted)
{
    const uint64_t pg = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pg >= n_pages) return;
    const char alpha[42] = "abcdefghijklmnopqrstuvwxyz      etaoinshr";
    const uint64_t page = first_page + pg;
    uint64_t s = p.seed ^ (page * 0x9E3779B97F4A7C15ull) ^ 0xA5A5A5A55A5A5A5Aull;
    uint64_t *dst = out + pg * 512;
    uint32_t pos = 0;
    uint64_t acc = 0;
    bool prev_planted = true;
    (void)cg_next(s);
    while (pos < 4096u) {
        uint64_t r = cg_next(s);
        uint32_t len = 40u + (uint32_t)((r & 0xffff) % 81u);
        uint32_t rem = 4096u - pos;
        uint32_t draw = (uint32_t)((r >> 16) & 0xffffff);
        uint32_t v = p.n_variants ? (uint32_t)((r >> 40) % p.n_variants) : 0u;
        if (rem < len + 1u + 41u) len = rem - 1u;
        bool plant = !prev_planted && p.n_variants && p.plant_period &&
                     (draw % p.plant_period == 0) && len >= 5u + p.vlen[v] + 5u;
        uint64_t x = 0, u = 0;
        for (uint32_t i = 0; i <= len; ++i) {
            uint32_t c;
            if (i == len) {
                c = '\n';
            } else {
                if ((i & 7u) == 0) { x = cg_next(s); u = cg_next(s); }
                c = (uint8_t)alpha[(((uint32_t)(x >> (8 * (i & 7u))) & 0xffu) * 41u) >> 8];
                if (plant && i >= 5u && i < 5u + p.vlen[v]) c = p.variants[v][i - 5u];
                if (p.upper_permille && c >= 'a' && c <= 'z' &&
                    ((((uint32_t)(u >> (8 * (i & 7u))) & 0xffu) * 1000u) >> 8) < p.upper_permille)
                    c -= 32u;
            }
            acc |= (uint64_t)c << (8 * (pos & 7u));
            if ((pos & 7u) == 7u) { dst[pos >> 3] = acc; acc = 0; }
            ++pos;
        }
        if (plant && planted) atomicAdd(&planted[v], 1ull);
        prev_planted = plant;
    }
}

// ---------------------------------------------------------------------------------------
// host-callable launchers (C++ linkage inside the library; the C-ABI is agh_api.cpp)
// ---------------------------------------------------------------------------------------
template <int H>
static void launch_sweep_t(const agh_sweep_args &a, hipStream_t st)
{
    const uint64_t n_full = a.n >> AGH_STRIP_SHIFT;
    const uint64_t n_waves = (n_full + AGH_WAVE_STRIPS - 1) / AGH_WAVE_STRIPS;
    if (n_waves) {
        const uint32_t blocks = (uint32_t)((n_waves + 3) / 4);
        hipLaunchKernelGGL(k_sweep<H>, dim3(blocks), dim3(256), 0, st,
                           (const uint4 *)a.text, n_full, a.q, a.ftab, a.strip_prefix,
                           a.wave_totals, a.cand, a.wave_cand, a.counters);
    }
    if (a.n & (AGH_STRIP - 1))
        hipLaunchKernelGGL(k_sweep_tail<H>, dim3(1), dim3(64), 0, st, (const uint4 *)a.text,
                           a.n, a.q, a.ftab, a.strip_prefix, a.wave_totals, a.cand,
                           a.wave_cand, a.counters);
    const uint64_t n_strips = (a.n + AGH_STRIP - 1) >> AGH_STRIP_SHIFT;
    const uint32_t nw = (uint32_t)((n_strips + AGH_WAVE_STRIPS - 1) / AGH_WAVE_STRIPS);
    const size_t scan_lds = (1024 + ((nw + 1023u) / 1024u) * 1024u) * sizeof(uint32_t);
    hipLaunchKernelGGL(k_wave_scan, dim3(1), dim3(1024), scan_lds, st, a.wave_totals,
                       H > 0 ? (const uint32_t *)a.wave_cand : (const uint32_t *)nullptr,
                       nw, (const uint8_t *)a.text, a.n, a.counters);
}

void agh_launch_sweep(const agh_sweep_args &a, int H, hipStream_t st)
{
    switch (H) {
    case 0: launch_sweep_t<0>(a, st); break;
    case 4: launch_sweep_t<4>(a, st); break;
    case 8: launch_sweep_t<8>(a, st); break;
    default: launch_sweep_t<16>(a, st); break;
    }
}

template <typename WT, int K, int NCH>
static void launch_verify_n(const agh_scan_args &a, hipStream_t st)
{
    uint32_t blocks = (a.nw + 3u) / 4u;
    if (!blocks) return;
    hipLaunchKernelGGL((k_verify<WT, K, NCH>), dim3(blocks), dim3(256), 0, st,
                       (const uint8_t *)a.text, a.n, a.q, (const WT *)a.mask, a.cand,
                       a.wave_cand, a.wave_prefix, a.nw, a.mk);
}

template <typename WT, int K>
static void launch_verify_t(const agh_scan_args &a, hipStream_t st)
{
    // window span: 2(m+k) + 1 + q plus up to 15 bytes of alignment slack in front
    const int span = 2 * (a.q.m + a.q.k) + 1 + a.q.fq + 15;
    const int nch = (span + 15) / 16;
    if (sizeof(WT) == 4) {
        if (nch <= 4) launch_verify_n<WT, K, 4>(a, st);
        else launch_verify_n<WT, K, 8>(a, st);
    } else {
        if (nch <= 8) launch_verify_n<WT, K, 8>(a, st);
        else launch_verify_n<WT, K, 12>(a, st);
    }
}

template <typename WT, int K>
static void launch_fullscan_t(const agh_scan_args &a, hipStream_t st)
{
    const uint64_t tile_bytes = (uint64_t)AGH_FS_THREADS * AGH_FS_CHUNK;
    uint64_t n_tiles = (a.n + tile_bytes - 1) / tile_bytes;
    if (!n_tiles) return;
    uint32_t blocks = n_tiles > 65536 ? 65536u : (uint32_t)n_tiles;
    const size_t lds = 256 * sizeof(WT) + (size_t)(AGH_FS_THREADS + 1) * AGH_FS_SLOT;
    hipLaunchKernelGGL((k_fullscan<WT, K>), dim3(blocks), dim3(AGH_FS_THREADS), lds, st,
                       (const uint8_t *)a.text, a.n, a.q, (const WT *)a.mask, a.strip_prefix,
                       a.wave_prefix, a.n_strips, a.mk);
}

template <typename WT>
static void dispatch_k(const agh_scan_args &a, bool full, hipStream_t st)
{
#define AGH_CASE(KK)                                            \
    case KK:                                                    \
        if (full) launch_fullscan_t<WT, KK>(a, st);             \
        else launch_verify_t<WT, KK>(a, st);                    \
        break;
    switch (a.q.k) {
        AGH_CASE(0) AGH_CASE(1) AGH_CASE(2) AGH_CASE(3) AGH_CASE(4)
        AGH_CASE(5) AGH_CASE(6) AGH_CASE(7) AGH_CASE(8)
    default: break;
    }
#undef AGH_CASE
}

void agh_launch_verify(const agh_scan_args &a, hipStream_t st)
{
    if (a.wide) dispatch_k<uint64_t>(a, false, st);
    else dispatch_k<uint32_t>(a, false, st);
}

void agh_launch_fullscan(const agh_scan_args &a, hipStream_t st)
{
    if (a.wide) dispatch_k<uint64_t>(a, true, st);
    else dispatch_k<uint32_t>(a, true, st);
}

void agh_launch_bitmap_count(const uint32_t *bitmap, uint32_t n_words, uint32_t *counters,
                             hipStream_t st)
{
    uint32_t blocks = (n_words + 256u * 16u - 1u) / (256u * 16u);
    if (blocks > 128u) blocks = 128u;
    if (!blocks) blocks = 1u;
    hipLaunchKernelGGL(k_bitmap_count, dim3(blocks), dim3(256), 0, st, bitmap, n_words, counters);
}

void agh_launch_read_probe(const void *text, uint64_t n, uint32_t *counters, hipStream_t st)
{
    const uint64_t n_strips = n >> AGH_STRIP_SHIFT;
    const uint64_t n_waves = (n_strips + AGH_WAVE_STRIPS - 1) / AGH_WAVE_STRIPS;
    if (!n_waves) return;
    hipLaunchKernelGGL(k_read_probe, dim3((uint32_t)((n_waves + 3) / 4)), dim3(256), 0, st,
                       (const uint4 *)text, n_strips, counters);
}

void agh_launch_corpus(void *out, uint64_t first_page, uint64_t n_pages, uint64_t seed,
                       const unsigned char *variants, const uint32_t *vlen,
                       uint32_t n_variants, uint32_t plant_period, uint32_t upper_permille,
                       unsigned long long *planted_dev, hipStream_t st)
{
    agh_corpus_params p;
    p.seed = seed;
    p.n_variants = n_variants;
    p.plant_period = plant_period;
    p.upper_permille = upper_permille;
    for (uint32_t i = 0; i < 8; ++i) {
        p.vlen[i] = i < n_variants ? vlen[i] : 0;
        for (uint32_t j = 0; j < 80; ++j)
            p.variants[i][j] = (i < n_variants && j < p.vlen[i]) ? variants[i * 80 + j] : 0;
    }
    if (!n_pages) return;
    hipLaunchKernelGGL(k_corpus, dim3((uint32_t)((n_pages + 63) / 64)), dim3(64), 0, st,
                       (uint64_t *)out, first_page, n_pages, p, planted_dev);
}
