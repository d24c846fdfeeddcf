// agh_sweep.hip -- the HBM-bound kernels of the agrep record scanner on gfx950 (MI355X / CDNA4):
// the sweep over every text byte, the prefix scans, the counting kernels, the corpus generator.
//
// Data flow of one scan (all buffers in HBM; DESIGN.md has the engine table):
//
//   k_sweep<H>      streams every text byte once (non-temporal 16 B/lane coalesced loads).  Per
//                   1 KiB strip it counts the record delimiters (SWAR zero-byte test + v_bcnt;
//                   skipped by lean = count-only scans) and probes one q-byte sample every H
//                   bytes against a 2^18-bit table (32 KiB) in LDS; samples that hit go through a
//                   per-wave LDS queue into the wave's private slice of the candidate buffer
//                   (no atomics).  This is the kernel the roofline is quoted on.
//   k_scan_local /  exclusive scan of the per-wave delimiter totals (numbered scans).
//   k_scan_fixup
//   k_verify        (agh_scan.hip) one lane per candidate: the k-error automaton over the window
//                   around the sample; matched records go into a one-bit-per-record bitmap
//                   (numbered) or a hash set of record starts (lean).
//   k_bitmap_count / k_hashset_count   count and clear those.
//   k_fullscan, k_tablescan, k_sweep_multi   the other engines (agh_scan.hip, agh_table.hip,
//                   agh_multi.hip).
//
// No MFMA anywhere: the work is byte/bitwise integer and the bound is HBM read bandwidth.
#include <stdlib.h>
#include <string.h>

#include <cstdlib>
#include "agh_device_inl.h"


#include "agh_sweep_inl.h"
#include "agh_verify_inl.h"      // lean_insert (k_resolve_giveups)

// One supertile = 4 consecutive strips (4 KiB) of one wave: census, probes, prefix, emit.
template <int H, int MODE>
__device__ __forceinline__ void sweep_supertile(uint4 v0, uint4 v1, uint4 v2, uint4 v3,
                                                uint2 db, uint64_t s, int lane, uint32_t dd,
                                                const agh_dev_query &q, const uint8_t *ftab,
                                                uint32_t *__restrict__ strip_prefix,
                                                uint64_t *cq, uint32_t &qn,
                                                uint64_t *__restrict__ slice, uint32_t &run,
                                                uint32_t &ncand, uint32_t *counters,
                                                uint32_t nx3 = 0)
{
    // nx3 (H == 2 only): first dword of strip s+4, which the last sample of lane 63 reaches into
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, hits = 0;
    uint32_t x0 = 0, x1 = 0, x2 = 0, x3 = 0;
    if (H == 2) {
        x0 = next_lane_dword(v0.x, (uint32_t)__builtin_amdgcn_readlane((int)v1.x, 0));
        x1 = next_lane_dword(v1.x, (uint32_t)__builtin_amdgcn_readlane((int)v2.x, 0));
        x2 = next_lane_dword(v2.x, (uint32_t)__builtin_amdgcn_readlane((int)v3.x, 0));
        x3 = next_lane_dword(v3.x, nx3);
    }
    sweep_chunk<H, MODE>(v0, dd, q, ftab, a0, hits, 0, db.x, x0);
    sweep_chunk<H, MODE>(v1, dd, q, ftab, a1, hits, 4, db.x >> 16, x1);
    sweep_chunk<H, MODE>(v2, dd, q, ftab, a2, hits, 8, db.y, x2);
    sweep_chunk<H, MODE>(v3, dd, q, ftab, a3, hits, 12, db.y >> 16, x3);
    if (MODE & 4) {
        if (__ballot(hits != 0)) {
            const uint32_t rc[4] = {0u, 0u, 0u, 0u};
            emit_candidates<H>(hits, s, rc, cq, qn, slice, ncand, counters);
        }
        return;
    }
    // per-strip delimiter totals: two packed 16-bit sums per DPP scan (the scan is a full
    // inclusive prefix over the 64 lanes; lane 63 holds the totals)
    const uint32_t own01 = a0 | (a1 << 16), own23 = a2 | (a3 << 16);
    const uint32_t sc01 = wave_sum_to_lane63(own01);
    const uint32_t sc23 = wave_sum_to_lane63(own23);
    const uint32_t p01 = (uint32_t)__builtin_amdgcn_readlane((int)sc01, 63);
    const uint32_t p23 = (uint32_t)__builtin_amdgcn_readlane((int)sc23, 63);
    const uint32_t z0 = 8192u - (p01 & 0xffffu), z1 = 8192u - (p01 >> 16);
    const uint32_t z2 = 8192u - (p23 & 0xffffu), z3 = 8192u - (p23 >> 16);
    if (H == 0 && lane == 0)                    // only the full-scan kernel reads strip_prefix
        *reinterpret_cast<uint4 *>(strip_prefix + s) =
            make_uint4(run, run + z0, run + z0 + z1, run + z0 + z1 + z2);
    if (H > 0 && __ballot(hits != 0)) {
        // delimiters in front of my chunk: 128*lane minus the non-delimiter popcounts of the
        // lanes before me (exclusive prefix = inclusive scan - own)
        const uint32_t ex01 = sc01 - own01, ex23 = sc23 - own23;
        const uint32_t lb = 128u * (uint32_t)lane;
        uint32_t rc[4];
        rc[0] = run + lb - (ex01 & 0xffffu);
        rc[1] = run + z0 + lb - (ex01 >> 16);
        rc[2] = run + z0 + z1 + lb - (ex23 & 0xffffu);
        rc[3] = run + z0 + z1 + z2 + lb - (ex23 >> 16);
        emit_candidates<H>(hits, s, rc, cq, qn, slice, ncand, counters);
    }
    run += z0 + z1 + z2 + z3;
}

// grid: ceil(n_waves / (BLOCK/64)) workgroups of BLOCK threads sharing one LDS copy of the
// filter table; wave w owns strips [w*AGH_WAVE_STRIPS, min((w+1)*AGH_WAVE_STRIPS, n_full)).
// PREFETCH: the loads of supertile i+1 are issued before supertile i is processed, so every
// wave keeps 4-8 KiB of HBM reads in flight while its VALU work runs.
template <int H, int MODE, int BLOCK, bool PREFETCH>
__global__ __launch_bounds__(BLOCK) void k_sweep(const uint4 *__restrict__ text,
                                                 uint64_t n_full_strips, uint32_t w_base,
                                                 agh_dev_query q,
                                                 const uint8_t *__restrict__ ftab_g,
                                                 uint32_t *__restrict__ strip_prefix,
                                                 uint32_t *__restrict__ wave_totals,
                                                 uint64_t *__restrict__ cand,
                                                 uint32_t *__restrict__ wave_cand,
                                                 uint32_t *__restrict__ counters,
                                                 const uint16_t *__restrict__ dbm16,
                                                 uint64_t n_dw)
{
    // n_dw: readable dwords of the text (H == 2 looks one dword past a strip)
    __shared__ __attribute__((aligned(16))) uint8_t ftab[H > 0 ? AGH_FT_SIZE : 16];
    __shared__ uint64_t cq_all[H > 0 ? (BLOCK / WAVE) * AGH_CQ_LEN : 1];
    if (H > 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(ftab_g);
        uint4 *dst = reinterpret_cast<uint4 *>(ftab);
        constexpr int PER = AGH_FT_SIZE / 16 / BLOCK;        // 16-byte pieces per thread
        uint4 tmp[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) tmp[i] = src[threadIdx.x + i * BLOCK];
#pragma unroll
        for (int i = 0; i < PER; ++i) dst[threadIdx.x + i * BLOCK] = tmp[i];
        __syncthreads();
    }
    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    // this launch sweeps the wave ranges w_base .. up to strip n_full_strips (a part of the text)
    const uint64_t w = (uint64_t)w_base + (uint64_t)blockIdx.x * (BLOCK / WAVE) + wib;
    const uint64_t s0 = w * AGH_WAVE_STRIPS;
    if (s0 >= n_full_strips) return;
    uint64_t s1 = s0 + AGH_WAVE_STRIPS;
    if (s1 > n_full_strips) s1 = n_full_strips;
    const uint32_t dd = q.delim * 0x01010101u;
    uint32_t run = 0;                           // delimiters before strip s inside this range
    uint32_t ncand = 0;                         // candidates already in this wave's slice
    uint32_t qn = 0;                            // candidates queued in LDS
    uint64_t *cq = cq_all + (H > 0 ? wib * AGH_CQ_LEN : 0);
    uint64_t *slice = cand + w * AGH_SLICE_CAP;
    uint64_t s = s0;

    // delimiter-end bits of the lane's four chunks (multi-byte delimiters + census only)
    auto dbits = [&](uint64_t st) -> uint2 {
        if (!(MODE & 8) || (MODE & 4)) return make_uint2(0u, 0u);
        const uint16_t *d = dbm16 + st * 64 + lane;
        return make_uint2((uint32_t)d[0] | ((uint32_t)d[64] << 16),
                          (uint32_t)d[128] | ((uint32_t)d[192] << 16));
    };
    // H == 2: the dword right behind strip st-1 (uniform; 0 past the readable text)
    auto first_dword_of = [&](uint64_t st) -> uint32_t {
        if (H != 2) return 0u;
        const uint64_t i = st * 256u;
        return i < n_dw ? reinterpret_cast<const uint32_t *>(text)[i] : 0u;
    };
    if (PREFETCH) {
        if (s + 4 <= s1) {
            const uint4 *p = text + s * 64 + lane;
            uint4 c0 = ld_stream(p), c1 = ld_stream(p + 64), c2 = ld_stream(p + 128), c3 = ld_stream(p + 192);
            uint2 cd = dbits(s);
            for (; s + 8 <= s1; s += 4) {
                const uint4 *pn = text + (s + 4) * 64 + lane;
                uint4 n0 = ld_stream(pn), n1 = ld_stream(pn + 64), n2 = ld_stream(pn + 128), n3 = ld_stream(pn + 192);
                uint2 nd = dbits(s + 4);
                sweep_supertile<H, MODE>(c0, c1, c2, c3, cd, s, lane, dd, q, ftab, strip_prefix,
                                         cq, qn, slice, run, ncand, counters,
                                         H == 2 ? (uint32_t)__builtin_amdgcn_readlane((int)n0.x, 0) : 0u);
                c0 = n0; c1 = n1; c2 = n2; c3 = n3;
                cd = nd;
            }
            sweep_supertile<H, MODE>(c0, c1, c2, c3, cd, s, lane, dd, q, ftab, strip_prefix, cq,
                                     qn, slice, run, ncand, counters, first_dword_of(s + 4));
            s += 4;
        }
    } else {
        for (; s + 4 <= s1; s += 4) {
            const uint4 *p = text + s * 64 + lane;
            uint4 v0 = ld_stream(p), v1 = ld_stream(p + 64), v2 = ld_stream(p + 128), v3 = ld_stream(p + 192);   // 4 x 1 KiB in flight
            sweep_supertile<H, MODE>(v0, v1, v2, v3, dbits(s), s, lane, dd, q, ftab,
                                     strip_prefix, cq, qn, slice, run, ncand, counters, first_dword_of(s + 4));
        }
    }
    for (; s < s1; ++s) {                       // < 4 strips left in the range
        uint4 v0 = ld_stream(text + s * 64 + lane);
        uint32_t a0 = 0, hits = 0;
        sweep_chunk<H, MODE>(v0, dd, q, ftab, a0, hits, 0,
                             ((MODE & 8) && !(MODE & 4)) ? (uint32_t)dbm16[s * 64 + lane] : 0u,
                             H == 2 ? next_lane_dword(v0.x, first_dword_of(s + 1)) : 0u);
        if (H == 2) hits >>= 24;                // eight pushes from the top: probe i at bit i
        if (MODE & 4) {
            if (__ballot(hits != 0)) {
                const uint32_t rc[4] = {0u, 0u, 0u, 0u};
                emit_candidates<H>(hits, s, rc, cq, qn, slice, ncand, counters);
            }
            continue;
        }
        const uint32_t sc0 = wave_sum_to_lane63(a0);
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)sc0, 63);
        if (H == 0 && lane == 0) strip_prefix[s] = run;
        if (H > 0 && __ballot(hits != 0)) {
            uint32_t rc[4];
            rc[0] = run + 128u * (uint32_t)lane - (sc0 - a0);
            rc[1] = rc[2] = rc[3] = 0;
            emit_candidates<H>(hits, s, rc, cq, qn, slice, ncand, counters);
        }
        run += 8192u - p0;
    }
    if (H > 0 && qn) flush_candidates(cq, qn, qn, slice, ncand, counters);
    if (lane == 0) {
        wave_totals[w] = run;
        if (H > 0) wave_cand[w] = ncand < AGH_SLICE_CAP ? ncand : AGH_SLICE_CAP;
    }
}

// The last, partial strip (n % 1024 != 0): one wave, bytes >= n masked to a non-delimiter.
// Runs after k_sweep on the same stream.
template <int H, int MODE>
__global__ __launch_bounds__(64) void k_sweep_tail(const uint4 *__restrict__ text, uint64_t n,
                                                   agh_dev_query q,
                                                   const uint8_t *__restrict__ ftab_g,
                                                   uint32_t *__restrict__ strip_prefix,
                                                   uint32_t *__restrict__ wave_totals,
                                                   uint64_t *__restrict__ cand,
                                                   uint32_t *__restrict__ wave_cand,
                                                   uint32_t *__restrict__ counters,
                                                   const uint16_t *__restrict__ dbm16)
{
    __shared__ uint64_t cq[AGH_CQ_LEN];
    uint32_t qn = 0;
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
    sweep_chunk<H, MODE>(v, dd, q, ftab_g, a0, hits, 0,       // table straight from global/L2
                         ((MODE & 8) && !(MODE & 4) && off < n) ? (uint32_t)dbm16[off >> 4] : 0u,
                         H == 2 ? next_lane_dword(v.x, fill4) : 0u);
    if (H == 2) {
        hits >>= 24;
        // samples that start at or behind the end of the text are no candidates
        if (off >= n) hits = 0;
        else if (off + 16 > n) hits &= (1u << ((n - off + 1) >> 1)) - 1u;
    }
    const uint32_t sc0 = (MODE & 4) ? 0u : wave_sum_to_lane63(a0);
    const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)sc0, 63);
    const uint32_t z = (MODE & 4) ? 0u : 8192u - p0;
    const uint64_t w = s / AGH_WAVE_STRIPS;
    const bool fresh = (s % AGH_WAVE_STRIPS) == 0;      // k_sweep never touched this range
    uint32_t before = fresh ? 0u : wave_totals[w];
    before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
    if (H > 0) {
        uint32_t ncand = fresh ? 0u : wave_cand[w];
        ncand = (uint32_t)__builtin_amdgcn_readfirstlane((int)ncand);
        if (__ballot(hits != 0)) {
            uint32_t rc[4];
            // lean entries are 64-bit dword indices: the record-count half must stay zero
            rc[0] = (MODE & 4) ? 0u : before + 128u * (uint32_t)lane - (sc0 - a0);
            rc[1] = rc[2] = rc[3] = 0;
            emit_candidates<H>(hits, s, rc, cq, qn, cand + w * AGH_SLICE_CAP, ncand, counters);
            if (qn) flush_candidates(cq, qn, qn, cand + w * AGH_SLICE_CAP, ncand, counters);
        }
        if (lane == 0) wave_cand[w] = ncand < AGH_SLICE_CAP ? ncand : AGH_SLICE_CAP;
    }
    if (lane == 0) {
        if (H == 0) strip_prefix[s] = before;
        wave_totals[w] = before + z;
    }
}

// ---------------------------------------------------------------------------------------
// Multi-byte delimiters (-d 'From ', -d '$$'): one pass marks where delimiter occurrences end,
// leftmost and non-overlapping exactly as the in-word delimiter automaton of the reference
// selects them (asearch.c:54-57, 175-186: after a detection the delimiter positions are
// cleared, so an occurrence overlapping a selected one is skipped).  One lane per 64 text
// bytes; a lane restarts the little shift-AND automaton at a point no occurrence straddles
// (for delimiters without a border that is simply dlen-1 bytes in front of its bytes).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool delim_occurs_at(const uint8_t *__restrict__ text, uint64_t n,
                                                uint64_t s, const agh_dev_query &q)
{
    if (s + q.dlen > n) return false;
    for (uint32_t j = 0; j < q.dlen; ++j)
        if (text[s + j] != q.dbytes[j]) return false;
    return true;
}

// Round 5: the bitmap was 5 of the 7 ms of a 4 GiB scan under -d 'xy' (a byte at a time through delim_class, one
// lane per 64 bytes).  Now a lane turns its 64 bytes (+ the 16 in front) into one "this byte equals delimiter byte j"
// bit mask per j with dword SWAR compares (folded first under -i), ANDs the shifted masks into "an occurrence ends
// here", and keeps all of them when no two of them overlap -- which is every lane of a text without runs of a
// delimiter that overlaps itself.  A lane that sees two overlapping occurrences (ends less than dlen apart), and the
// lane with the virtual head byte in front of it, selects leftmost / non-overlapping with the serial automaton as before.
__device__ __forceinline__ uint32_t delim_eq4(uint32_t x, uint32_t dd)
{
    const uint32_t y = x ^ dd;                  // bit 7 of every byte of x that equals the byte in dd, packed to 4 bits
    const uint32_t z = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);
    return ((z >> 7) * 0x01020408u) >> 24;      // bytes 0..3 -> bits 0..3 (no carries: all partial products distinct)
}
__device__ __forceinline__ uint32_t fold4(uint32_t x)
{
    const uint32_t t = x & 0x7f7f7f7fu;         // 'A'..'Z' -> 'a'..'z', four bytes at once
    const uint32_t up = (t + 0x3f3f3f3fu) & ~(t + 0x25252525u) & ~x & 0x80808080u;
    return x | (up >> 2);
}

__device__ uint64_t delim_bitmap_serial(const uint8_t *__restrict__ text, uint64_t n, const agh_dev_query &q,
                                        uint64_t b, uint32_t *__restrict__ counters)
{
    // restart point: no occurrence may start before r and end at or after r
    uint64_t r = b >= q.dlen - 1 ? b - (q.dlen - 1) : 0;
    uint32_t steps = 0;
    for (;;) {
        bool covered = false;
        if (r > 0) {
            const uint64_t lo = r >= q.dlen - 1 ? r - (q.dlen - 1) : 0;
            for (uint64_t st = lo; st < r && !covered; ++st)
                covered = delim_occurs_at(text, n, st, q) && st + q.dlen > r;
        }
        if (!covered) break;
        --r;
        if (++steps > 4096u) { counters[AGH_C_DELIM_CHAIN] = 1u; break; }
    }
    // the virtual byte in front of the text (asearch.c:69-78) takes part in delimiter matching
    uint32_t ds = r == 0 ? (delim_class(q, q.head_byte) & 1u) : 0u;
    uint64_t out = 0;
    const uint32_t endbit = 1u << (q.dlen - 1);
    const uint64_t hi = b + 64 < n ? b + 64 : n;
    for (uint64_t i = r; i < hi; ++i) {
        ds = ((ds << 1) | 1u) & delim_class(q, text[i]);
        if (ds & endbit) {
            if (i >= b) out |= 1ull << (i - b);
            ds = 0;
        }
    }
    return out;
}

#ifndef AGH_DBM_ITER
#define AGH_DBM_ITER 8u
#endif
__global__ __launch_bounds__(256) void k_delim_bitmap(const uint8_t *__restrict__ text,
                                                      uint64_t n, agh_dev_query q,
                                                      uint64_t *__restrict__ dbm,
                                                      uint64_t n_words,
                                                      uint32_t *__restrict__ counters)
{
    // (AGH_DBM_ITER words per lane, 256 apart: one word per lane made a 4 GiB text 262144 workgroups that lived for
    // a few microseconds each)
    for (uint32_t it = 0; it < AGH_DBM_ITER; ++it) {
    const uint64_t wi = ((uint64_t)blockIdx.x * AGH_DBM_ITER + it) * blockDim.x + threadIdx.x;
    if (wi >= n_words) return;
    const uint64_t b = wi * 64;
    if (b >= n) { dbm[wi] = 0; continue; }
    if (wi == 0) {                              // (the virtual head byte in front)
        dbm[0] = delim_bitmap_serial(text, n, q, 0, counters);
        continue;
    }
    // bytes b - 16 .. b + 63 as 20 dwords (pieces at or behind the end of the text: none; the piece that holds the
    // last byte is readable to its end, its bytes behind the text are masked below)
    uint32_t w[20];
#pragma unroll
    for (uint32_t p = 0; p < 5; ++p) {
        const uint64_t a = b - 16 + 16u * p;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (a < n) v = *reinterpret_cast<const uint4 *>(text + a);
        w[4 * p] = v.x; w[4 * p + 1] = v.y; w[4 * p + 2] = v.z; w[4 * p + 3] = v.w;
    }
    if (q.dfold) {
#pragma unroll
        for (uint32_t d = 0; d < 20; ++d) w[d] = fold4(w[d]);
    }
    // window bit i = text byte b - 16 + i; 80 bits as (lo: 64, hi: 16)
    const uint64_t live = n - (b - 16);         // bytes of the window inside the text (> 16)
    const uint64_t live_lo = live >= 64 ? ~0ull : ((1ull << live) - 1ull);
    const uint32_t live_hi = live >= 80 ? 0xffffu : (live > 64 ? ((1u << (live - 64)) - 1u) : 0u);
    uint64_t e_lo = ~0ull;
    uint32_t e_hi = 0xffffu;
    const uint32_t dlen = q.dlen;
    for (uint32_t j = 0; j < dlen; ++j) {
        const uint32_t dd = (uint32_t)q.dbytes[j] * 0x01010101u;
        uint64_t m_lo = 0;
        uint32_t m_hi = 0;
#pragma unroll
        for (uint32_t d = 0; d < 16; ++d) m_lo |= (uint64_t)delim_eq4(w[d], dd) << (4u * d);
#pragma unroll
        for (uint32_t d = 16; d < 20; ++d) m_hi |= delim_eq4(w[d], dd) << (4u * (d - 16));
        m_lo &= live_lo;
        m_hi &= live_hi;
        const uint32_t sh = dlen - 1u - j;      // an occurrence that ends at i has byte j at i - sh
        if (sh) {
            m_hi = ((m_hi << sh) | (uint32_t)(m_lo >> (64u - sh))) & 0xffffu;
            m_lo <<= sh;
        }
        e_lo &= m_lo;
        e_hi &= m_hi;
    }
    // two occurrences whose ends are less than dlen apart overlap: the serial selection decides
    uint64_t s_lo = e_lo, a_lo = 0;
    uint32_t s_hi = e_hi, a_hi = 0;
    for (uint32_t p = 1; p < dlen; ++p) {
        s_hi = ((s_hi << 1) | (uint32_t)(s_lo >> 63)) & 0xffffu;
        s_lo <<= 1;
        a_lo |= s_lo;
        a_hi |= s_hi;
    }
    const uint64_t out = (e_lo >> 16) | ((uint64_t)e_hi << 48);
    const uint64_t ov = ((e_lo & a_lo) >> 16) | ((uint64_t)(e_hi & a_hi) << 48);
    dbm[wi] = ov ? delim_bitmap_serial(text, n, q, b, counters) : out;
    }
}

// Round 6: the same selection with the sweeps' layout -- a wave takes 1 KiB strips with one coalesced 16-byte load per
// lane (the kernel above: five loads per lane, 64 bytes apart, every wave-instruction touching 32 cache lines), the 8
// bytes in front of a lane's chunk come from the lane before it (DPP; lane 0: the strip before, carried in scalars),
// and "an occurrence ends here" costs one zero-byte test per dword whatever the delimiter's length: the byte s in
// front of an end must be delimiter byte dlen - 1 - s, so y = OR over s of (the dword moved back by s bytes) ^ that
// byte is zero exactly in the bytes where an occurrence ends.  A lane writes the 16 bits of its chunk; four lanes make
// a word of the bitmap, and a word in which two occurrences overlap (and word 0: the virtual head byte) is selected by
// the serial automaton of the group's first lane as before.  A wave walks AGH_DBM_STRIPS consecutive strips; the one
// in front of them is computed for its carries only.
#ifndef AGH_DBM_STRIPS
#define AGH_DBM_STRIPS 16u
#endif
template <uint32_t DLEN>                        // (the delimiter's length at compile time: no branch per shifted compare)
__global__ __launch_bounds__(256) void k_delim_bitmap_strips(const uint8_t *__restrict__ text, uint64_t n, agh_dev_query q,
                                                             uint64_t *__restrict__ dbm, uint64_t n_words,
                                                             uint32_t *__restrict__ counters)
{
    const uint32_t lane = (uint32_t)lane_id();
    const uint64_t n_strips = (n_words * 64u + 1023u) / 1024u;
    const uint64_t s0 = ((uint64_t)blockIdx.x * 4u + threadIdx.x / WAVE) * AGH_DBM_STRIPS;
    if (s0 >= n_strips) return;
    const uint64_t s1 = s0 + AGH_DBM_STRIPS < n_strips ? s0 + AGH_DBM_STRIPS : n_strips;
    uint16_t *out = reinterpret_cast<uint16_t *>(dbm);
    constexpr uint32_t dlen = DLEN;
    uint64_t dall = 0;                          // the delimiter's bytes as one scalar
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) dall |= (uint64_t)q.dbytes[j] << (8u * j);
    uint32_t dds[8];                            // delimiter byte dlen - 1 - s in all four bytes of a dword
#pragma unroll
    for (uint32_t sft = 0; sft < 8; ++sft)
        dds[sft] = sft < dlen ? ((uint32_t)(dall >> (8u * (dlen - 1u - sft))) & 0xffu) * 0x01010101u : 0u;
    auto load = [&](uint64_t s) -> uint4 {
        const uint64_t a = s * 1024u + (uint64_t)lane * 16u;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (a < n) v = *reinterpret_cast<const uint4 *>(text + a);      // (the piece that holds the last byte is readable to its end)
        return v;
    };
    uint32_t c_e = 0, c_z = 0, c_w = 0;         // lane 63 of the strip before: its ends and its last two dwords
    const uint64_t first = s0 ? s0 - 1u : 0u;
    __shared__ uint16_t lists[4][64];
    uint16_t *list = lists[threadIdx.x / WAVE];
    uint32_t n_list = 0;                        // (uniform)
    auto strip = [&](uint64_t s, uint4 v, uint32_t slot) {
        const uint64_t a = s * 1024u + (uint64_t)lane * 16u;
        if (q.dfold) {
            v.x = fold4(v.x); v.y = fold4(v.y); v.z = fold4(v.z); v.w = fold4(v.w);
        }
        const uint32_t pz = (uint32_t)__builtin_amdgcn_update_dpp((int)c_z, (int)v.z, 0x138, 0xf, 0xf, false);   // wave_shr:1
        const uint32_t pw = (uint32_t)__builtin_amdgcn_update_dpp((int)c_w, (int)v.w, 0x138, 0xf, 0xf, false);
        const uint32_t W[6] = {pz, pw, v.x, v.y, v.z, v.w};
        uint32_t zb[4];
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
            uint32_t y = 0;
#pragma unroll
            for (uint32_t sft = 0; sft < 8; ++sft) {
                if (sft < dlen) {               // (compile time)
                    const uint32_t hi = d + 2u - (sft >> 2);
                    const uint32_t src = (sft & 3u) ? __builtin_amdgcn_alignbyte(W[hi], W[hi - 1u], 4u - (sft & 3u)) : W[hi];
                    y |= src ^ dds[sft];
                }
            }
            zb[d] = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);      // bit 7 of every byte where an occurrence ends
        }
        // two dwords' bits 7, 15, 23, 31 -> eight bits in text order
        auto pack8 = [](uint32_t z0, uint32_t z1) -> uint32_t {
            const uint32_t u = (z0 >> 7) | (z1 >> 3);
            return (u | (u >> 7) | (u >> 14) | (u >> 21)) & 0xffu;
        };
        uint32_t e16 = pack8(zb[0], zb[1]) | (pack8(zb[2], zb[3]) << 8);
        const bool inside = s * 1024u + 1024u <= n;                      // (uniform) the whole strip lies in the text
        if (!inside) {
            const uint64_t left = a < n ? n - a : 0u;                    // (an end at i < n has all its bytes inside the text)
            if (left < 16u) e16 &= (1u << left) - 1u;
        }
        // two occurrences whose ends are less than dlen apart overlap: the serial selection decides
        const uint32_t pe = (uint32_t)__builtin_amdgcn_update_dpp((int)c_e, (int)e16, 0x138, 0xf, 0xf, false);
        uint32_t near = 0;
#pragma unroll
        for (uint32_t p = 1; p < dlen; ++p) near |= (e16 << p) | (pe >> (16u - p));
        const bool ov = (e16 & near & 0xffffu) != 0u;
        c_e = (uint32_t)__builtin_amdgcn_readlane((int)e16, 63);
        c_z = (uint32_t)__builtin_amdgcn_readlane((int)v.z, 63);
        c_w = (uint32_t)__builtin_amdgcn_readlane((int)v.w, 63);
        if (s < s0) return;                     // (uniform) the strip in front: carries only
        // a word (four lanes) in which two occurrences overlap, and word 0 (the virtual head byte in front), is left to
        // the serial selection: listed here, and walked one word per lane after the wave's four strips (one lane of
        // every four walking while the wave waits made a text full of '   ' runs 6.7 ms against round 5's 4.8)
        const uint64_t nm = __ballot(ov || (s == 0 && lane < 4u));
        bool grp = false;
        if (nm) {
            grp = ((uint32_t)(nm >> (lane & ~3u)) & 0xfu) != 0u;
            const bool lead = grp && (lane & 3u) == 0u;
            const uint64_t lm = __ballot(lead);
            if (lead)
                list[n_list + __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u))] =
                    (uint16_t)((slot << 4) | (lane >> 2));
            n_list += (uint32_t)__popcll(lm);
        }
        if (!grp && (inside || a < n_words * 64u)) out[a >> 4] = (uint16_t)e16;
    };
    // four strips of a wave in flight while the four before are worked on (one strip ahead left the chip with 64 KiB
    // per CU under way: 1.4 ms per 4 GiB)
    uint4 buf[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) buf[i] = first + i < s1 ? load(first + i) : make_uint4(0, 0, 0, 0);
    for (uint64_t base = first; base < s1; base += 4) {
        uint4 nbuf[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) nbuf[i] = base + 4u + i < s1 ? load(base + 4u + i) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i)
            if (base + i < s1) strip(base + i, buf[i], i);
        if (n_list) {                           // (uniform) at most 4 x 16 words: one per lane
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane < n_list) {
                const uint32_t e = list[lane];
                const uint64_t word = (base + (e >> 4)) * 16u + (e & 15u);
                if (word < n_words) dbm[word] = word * 64u < n ? delim_bitmap_serial(text, n, q, word * 64u, counters) : 0ull;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            n_list = 0;
        }
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) buf[i] = nbuf[i];
    }
}

void agh_warm_core_module()
{
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(k_delim_bitmap));
}

void agh_launch_delim_bitmap(const void *text, uint64_t n, const agh_dev_query &q, uint64_t *dbm,
                             uint64_t n_words, uint32_t *counters, hipStream_t st)
{
    if (!n_words) return;
    static const bool by_words = [] { const char *e = getenv("AGH_DBM_WORDS"); return e && e[0] == '1'; }();    // (A/B: round 5's kernel)
    if (by_words) {
        hipLaunchKernelGGL(k_delim_bitmap, dim3((uint32_t)((n_words + 256 * AGH_DBM_ITER - 1) / (256 * AGH_DBM_ITER))), dim3(256), 0, st,
                           (const uint8_t *)text, n, q, dbm, n_words, counters);
        return;
    }
    const uint64_t n_strips = (n_words * 64u + 1023u) / 1024u;
    const uint64_t waves = (n_strips + AGH_DBM_STRIPS - 1u) / AGH_DBM_STRIPS;
    const dim3 grid((uint32_t)((waves + 3u) / 4u));
#define AGH_DBM_CASE(L)                                                                       \
    case L:                                                                                   \
        hipLaunchKernelGGL(k_delim_bitmap_strips<L>, grid, dim3(256), 0, st, (const uint8_t *)text, n, q, dbm, n_words, counters); \
        break;
    switch (q.dlen) {
        AGH_DBM_CASE(1) AGH_DBM_CASE(2) AGH_DBM_CASE(3) AGH_DBM_CASE(4) AGH_DBM_CASE(5) AGH_DBM_CASE(6) AGH_DBM_CASE(7)
        AGH_DBM_CASE(8)
    default:                                    // (no such query: dbytes holds eight)
        hipLaunchKernelGGL(k_delim_bitmap, dim3((uint32_t)((n_words + 256 * AGH_DBM_ITER - 1) / (256 * AGH_DBM_ITER))), dim3(256), 0, st,
                           (const uint8_t *)text, n, q, dbm, n_words, counters);
    }
#undef AGH_DBM_CASE
}

// Exclusive scan of the per-wave delimiter totals, two small multi-block kernels:
//   k_scan_local: every workgroup scans 1024 consecutive entries in place (4 per thread,
//                 wave DPP scan, 4 wave totals through LDS) and publishes its chunk total;
//                 it also reduces its 1024 candidate counts.
//   k_scan_fixup: every workgroup adds the totals of the chunks before it (<= 32 values) to
//                 its entries; workgroup 0 writes the grand totals and the last text byte.
#define AGH_SCAN_CHUNK 1024u
#define AGH_SCAN_MAXCHUNKS 64u   // 64 * 1024 wave ranges * 256 KiB = 16 GiB >= one segment

__global__ __launch_bounds__(256) void k_scan_local(uint32_t *__restrict__ wave_totals,
                                                    const uint32_t *__restrict__ wave_cand,
                                                    uint32_t nw,
                                                    uint32_t *__restrict__ chunk_totals)
{
    __shared__ uint32_t wsum[4], csum[4];
    const int lane = lane_id();
    const uint32_t wv = threadIdx.x / WAVE;
    const uint32_t base = blockIdx.x * AGH_SCAN_CHUNK + threadIdx.x * 4u;
    uint32_t v[4], c = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = base + i < nw ? wave_totals[base + i] : 0u;
        if (wave_cand) c += base + i < nw ? wave_cand[base + i] : 0u;
    }
    const uint32_t own = v[0] + v[1] + v[2] + v[3];
    const uint32_t inc = wave_sum_to_lane63(own);       // inclusive scan over the wave
    const uint32_t csc = wave_sum_to_lane63(c);
    if (lane == 63) { wsum[wv] = inc; csum[wv] = csc; }
    __syncthreads();
    uint32_t before = inc - own;
    for (uint32_t i = 0; i < wv; ++i) before += wsum[i];
    if (base < nw) wave_totals[base] = before;
    if (base + 1 < nw) wave_totals[base + 1] = before + v[0];
    if (base + 2 < nw) wave_totals[base + 2] = before + v[0] + v[1];
    if (base + 3 < nw) wave_totals[base + 3] = before + v[0] + v[1] + v[2];
    if (threadIdx.x == 0) {
        chunk_totals[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        chunk_totals[AGH_SCAN_MAXCHUNKS + blockIdx.x] = csum[0] + csum[1] + csum[2] + csum[3];
    }
}

__global__ __launch_bounds__(256) void k_scan_fixup(uint32_t *__restrict__ wave_totals,
                                                    uint32_t nw,
                                                    const uint32_t *__restrict__ chunk_totals,
                                                    uint32_t n_chunks, const uint8_t *text,
                                                    uint64_t n, uint32_t *__restrict__ counters,
                                                    const uint64_t *__restrict__ dbm)
{
    __shared__ uint32_t sh_before;
    if (threadIdx.x < WAVE) {
        const uint32_t i = threadIdx.x;
        uint32_t d = i < n_chunks ? chunk_totals[i] : 0u;
        uint32_t c = i < n_chunks ? chunk_totals[AGH_SCAN_MAXCHUNKS + i] : 0u;
        const uint32_t dpre = wave_sum_to_lane63(i < blockIdx.x ? d : 0u);
        const uint32_t dall = wave_sum_to_lane63(d);
        const uint32_t dhi = wave_sum_to_lane63(d >> 16);   // (record numbers are 32-bit: a segment with more records is refused)
        const uint32_t call = wave_sum_to_lane63(c);
        if (i == 63) {
            sh_before = dpre;
            if (blockIdx.x == 0) {
                if (dhi >= 65000u) counters[AGH_C_BM_OVERFLOW] = 2u;
                counters[AGH_C_NDELIM] = dall;
                counters[AGH_C_CAND] = call;
                // does the text end with a delimiter?  (multi-byte: a selected occurrence)
                if (!n) counters[AGH_C_LASTBYTE] = 0xffffffffu;
                else if (dbm) counters[AGH_C_LASTBYTE] = dbm_bit(dbm, n - 1) ? (uint32_t)text[n - 1] : 0xffffffffu;
                else counters[AGH_C_LASTBYTE] = (uint32_t)text[n - 1];
            }
        }
    }
    __syncthreads();
    const uint32_t before = sh_before;
    if (before) {
        const uint32_t base = blockIdx.x * AGH_SCAN_CHUNK + threadIdx.x * 4u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (base + i < nw) wave_totals[base + i] += before;
    }
}

// ---------------------------------------------------------------------------------------
// matched-record count = population count of the record bitmap
// ---------------------------------------------------------------------------------------
// Counts the set bits of the record bitmap and leaves it zeroed for the next scan (so a scan
// never pays a separate memset for a bitmap that is almost entirely zero already).
__global__ __launch_bounds__(256) void k_bitmap_count(uint4 *__restrict__ bitmap,
                                                      uint32_t n_vec,
                                                      uint32_t *__restrict__ counters)
{
    // few, fat workgroups: the matched counter gets one atomic per workgroup (a hot counter
    // takes only ~90 updates/us)
    uint32_t acc = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n_vec; i += 4 * stride) {
        uint4 b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = bitmap[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (b[u].x | b[u].y | b[u].z | b[u].w) {
                acc += (uint32_t)(__popc(b[u].x) + __popc(b[u].y) + __popc(b[u].z) + __popc(b[u].w));
                bitmap[i + u * stride] = make_uint4(0, 0, 0, 0);
            }
    }
    for (; i < n_vec; i += stride) {
        const uint4 b = bitmap[i];
        if (b.x | b.y | b.z | b.w) {
            acc += (uint32_t)(__popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w));
            bitmap[i] = make_uint4(0, 0, 0, 0);
        }
    }
    acc = wave_sum_to_lane63(acc);
    __shared__ uint32_t part[4];
    if (lane_id() == 63) part[threadIdx.x / WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(&counters[AGH_C_MATCHED], t);
    }
}

// Lean scans: the set of matched records is a hash set of record-start offsets (+1); count the
// occupied slots, clear them, and total the candidate counters while at it.
__global__ __launch_bounds__(256) void k_hashset_count(uint64_t *__restrict__ tab,
                                                       uint32_t n_slots,
                                                       const uint32_t *__restrict__ wave_cand,
                                                       uint32_t nw,
                                                       uint32_t *__restrict__ counters)
{
    uint32_t acc = 0, cacc = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n2 = n_slots / 2;            // two slots per 16-byte access
    uint4 *t4 = reinterpret_cast<uint4 *>(tab);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const uint4 b = t4[i];
        if (b.x | b.y | b.z | b.w) {
            acc += ((b.x | b.y) ? 1u : 0u) + ((b.z | b.w) ? 1u : 0u);
            t4[i] = make_uint4(0, 0, 0, 0);
        }
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += stride) cacc += wave_cand[i];
    acc = wave_sum_to_lane63(acc);
    cacc = wave_sum_to_lane63(cacc);
    __shared__ uint32_t part[8];
    if (lane_id() == 63) { part[threadIdx.x / WAVE] = acc; part[4 + threadIdx.x / WAVE] = cacc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = part[0] + part[1] + part[2] + part[3];
        const uint32_t c = part[4] + part[5] + part[6] + part[7];
        if (t) atomicAdd(&counters[AGH_C_MATCHED], t);
        if (c) atomicAdd(&counters[AGH_C_CAND], c);
    }
}

// ---------------------------------------------------------------------------------------
// segment cuts: inputs above the segment limit are cut where a record ends at a 16-byte aligned
// offset (every kernel wants an aligned base).  One workgroup per nominal boundary walks back
// from it, 256 aligned offsets (4 KiB) per round; cut[i] = the largest p <= bound[i], p > lo[i],
// p % 16 == 0 with text[p-1] == delimiter, or 0 if there is none.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_find_cuts(const uint8_t *__restrict__ text,
                                                   const uint64_t *__restrict__ bound,
                                                   const uint64_t *__restrict__ lo,
                                                   uint32_t delim, uint32_t step,
                                                   uint64_t *__restrict__ cut)
{
    // step = 16: cuts at 16-byte aligned offsets (the kernels take an aligned base as it is);
    // step = 1: any record end (the host copies such a segment to an aligned buffer)
    __shared__ uint32_t best;
    const uint64_t b = bound[blockIdx.x], l = lo[blockIdx.x];
    uint64_t found = 0;
    for (uint64_t base = b; base > l;) {
        if (threadIdx.x == 0) best = 0xffffffffu;
        __syncthreads();
        const uint64_t back = (uint64_t)step * threadIdx.x;
        if (base >= back && base - back > l && text[base - back - 1] == delim)
            atomicMin(&best, threadIdx.x);
        __syncthreads();
        const uint32_t t = best;
        __syncthreads();
        if (t != 0xffffffffu) { found = base - (uint64_t)step * t; break; }
        if (base < 256ull * step + l) break;
        base -= 256ull * step;
    }
    if (threadIdx.x == 0) cut[blockIdx.x] = found;
}

// The same for delimiters that come from the delimiter bitmap (several bytes, or a folded letter):
// step = 64: a cut that is a multiple of 64, so that a segment's part of the bitmap starts on a
// word; step = 1: any selected delimiter end (the segment is copied and gets a bitmap of its own).
// cut[i] = the largest p <= bound[i], p > lo[i], p % step == 0, with a selected delimiter occurrence
// ending at byte p - 1, or 0.
__global__ __launch_bounds__(256) void k_find_cuts_dbm(const uint64_t *__restrict__ dbm,
                                                       const uint64_t *__restrict__ bound,
                                                       const uint64_t *__restrict__ lo,
                                                       uint32_t step, uint64_t *__restrict__ cut)
{
    __shared__ uint32_t best;
    const uint64_t b = bound[blockIdx.x] & ~(uint64_t)(step - 1), l = lo[blockIdx.x];
    uint64_t found = 0;
    for (uint64_t base = b; base > l;) {
        if (threadIdx.x == 0) best = 0xffffffffu;
        __syncthreads();
        const uint64_t back = (uint64_t)step * threadIdx.x;
        if (base >= back + 1 && base - back > l && dbm_bit(dbm, base - back - 1))
            atomicMin(&best, threadIdx.x);
        __syncthreads();
        const uint32_t t = best;
        __syncthreads();
        if (t != 0xffffffffu) { found = base - (uint64_t)step * t; break; }
        if (base < 256ull * step + l) break;
        base -= 256ull * step;
    }
    if (threadIdx.x == 0) cut[blockIdx.x] = found;
}

void agh_launch_find_cuts_dbm(const uint64_t *dbm, const uint64_t *bound, const uint64_t *lo,
                              uint32_t n_bounds, uint32_t step, uint64_t *cut, hipStream_t st)
{
    if (!n_bounds) return;
    hipLaunchKernelGGL(k_find_cuts_dbm, dim3(n_bounds), dim3(256), 0, st, dbm, bound, lo, step, cut);
}

void agh_launch_find_cuts(const void *text, const uint64_t *bound, const uint64_t *lo,
                          uint32_t n_bounds, uint32_t delim, uint32_t step, uint64_t *cut, hipStream_t st)
{
    if (!n_bounds) return;
    hipLaunchKernelGGL(k_find_cuts, dim3(n_bounds), dim3(256), 0, st, (const uint8_t *)text, bound,
                       lo, delim, step, cut);
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
        uint4 v0 = ld_stream(p), v1 = ld_stream(p + 64), v2 = ld_stream(p + 128), v3 = ld_stream(p + 192);
        acc ^= v0.x ^ v0.y ^ v0.z ^ v0.w ^ v1.x ^ v1.y ^ v1.z ^ v1.w;
        acc ^= v2.x ^ v2.y ^ v2.z ^ v2.w ^ v3.x ^ v3.y ^ v3.z ^ v3.w;
    }
    for (; s < s1; ++s) {
        uint4 v0 = ld_stream(text + s * 64 + lane);
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
// host-callable launchers
// ---------------------------------------------------------------------------------------
// Launch geometry chosen from within-process A/B rounds on MI355X (scripts/sweep_variants.py,
// DESIGN.md): 256-thread workgroups, next supertile prefetched.
#define AGH_SWEEP_BLOCK 256

// exclusive scan of the per-wave delimiter totals (+ candidate total, last byte)
void agh_launch_census_scan(const agh_sweep_args &a, bool with_cand, hipStream_t st)
{
    const uint64_t n_strips = (a.n + AGH_STRIP - 1) >> AGH_STRIP_SHIFT;
    const uint32_t nw = (uint32_t)((n_strips + AGH_WAVE_STRIPS - 1) / AGH_WAVE_STRIPS);
    const uint32_t n_chunks = (nw + AGH_SCAN_CHUNK - 1) / AGH_SCAN_CHUNK;
    hipLaunchKernelGGL(k_scan_local, dim3(n_chunks), dim3(256), 0, st, a.wave_totals,
                       with_cand ? (const uint32_t *)a.wave_cand : (const uint32_t *)nullptr, nw,
                       a.chunk_totals);
    hipLaunchKernelGGL(k_scan_fixup, dim3(n_chunks), dim3(256), 0, st, a.wave_totals, nw,
                       (const uint32_t *)a.chunk_totals, n_chunks, (const uint8_t *)a.text, a.n,
                       a.counters, (const uint64_t *)a.dbm);
}

template <int H, int MODE>
static void launch_sweep_hm(const agh_sweep_args &a, hipStream_t st)
{
    const uint64_t n_full_all = a.n >> AGH_STRIP_SHIFT;
    // a part [w_begin, w_end) of the wave ranges, or everything
    const bool to_end = a.w_end == 0 || (uint64_t)a.w_end * AGH_WAVE_STRIPS >= n_full_all;
    const uint64_t n_full = to_end ? n_full_all : (uint64_t)a.w_end * AGH_WAVE_STRIPS;
    const uint64_t w_hi = (n_full + AGH_WAVE_STRIPS - 1) / AGH_WAVE_STRIPS;
    const uint64_t n_waves = w_hi > a.w_begin ? w_hi - a.w_begin : 0;
    if (a.ev_begin) (void)hipEventRecord(a.ev_begin, st);
    if (n_waves && !a.tail_only) {
        const uint32_t wpb = AGH_SWEEP_BLOCK / 64;
        const uint32_t blocks = (uint32_t)((n_waves + wpb - 1) / wpb);
        hipLaunchKernelGGL((k_sweep<H, MODE, AGH_SWEEP_BLOCK, true>), dim3(blocks),
                           dim3(AGH_SWEEP_BLOCK), 0, st, (const uint4 *)a.text, n_full,
                           a.w_begin, a.q, a.ftab, a.strip_prefix, a.wave_totals, a.cand,
                           a.wave_cand, a.counters, (const uint16_t *)a.dbm,
                           (uint64_t)(((a.n + 15) & ~(uint64_t)15) / 4));
    }
    if (a.ev_end) (void)hipEventRecord(a.ev_end, st);
    if (!to_end) return;                        // parts are lean: the tail belongs to the last one
    if (a.n & (AGH_STRIP - 1))
        hipLaunchKernelGGL((k_sweep_tail<H, MODE>), dim3(1), dim3(64), 0, st,
                           (const uint4 *)a.text, a.n, a.q, a.ftab, a.strip_prefix,
                           a.wave_totals, a.cand, a.wave_cand, a.counters,
                           (const uint16_t *)a.dbm);
    if (MODE & 4) return;                       // lean: no record numbering, nothing to scan
    agh_launch_census_scan(a, H > 0, st);
}

template <int H>
static void launch_sweep_t(const agh_sweep_args &a, hipStream_t st)
{
    // lean sweeps never look at delimiters, so bit 3 only matters for census sweeps
    const int mode = (a.q.fold ? 1 : 0) | (a.q.fq == 4 ? 2 : 0) | (a.lean ? 4 : 0) |
                     ((a.q.mb && !a.lean) ? 8 : 0);
    switch (mode) {
    case 0: launch_sweep_hm<H, 0>(a, st); break;
    case 1: launch_sweep_hm<H, 1>(a, st); break;
    case 2: launch_sweep_hm<H, 2>(a, st); break;
    case 3: launch_sweep_hm<H, 3>(a, st); break;
    case 4: launch_sweep_hm<H, 4>(a, st); break;
    case 5: launch_sweep_hm<H, 5>(a, st); break;
    case 6: launch_sweep_hm<H, 6>(a, st); break;
    case 7: launch_sweep_hm<H, 7>(a, st); break;
    case 8: launch_sweep_hm<H, 8>(a, st); break;
    case 9: launch_sweep_hm<H, 9>(a, st); break;
    case 10: launch_sweep_hm<H, 10>(a, st); break;
    default: launch_sweep_hm<H, 11>(a, st); break;
    }
}

void agh_launch_sweep(const agh_sweep_args &a, int H, hipStream_t st)
{
    switch (H) {
    case 0:
        if (a.q.mb) launch_sweep_hm<0, 8>(a, st);
        else launch_sweep_hm<0, 0>(a, st);
        break;
    case 2: launch_sweep_t<2>(a, st); break;
    case 4: launch_sweep_t<4>(a, st); break;
    case 8: launch_sweep_t<8>(a, st); break;
    default: launch_sweep_t<16>(a, st); break;
    }
}

void agh_launch_bitmap_count(uint32_t *bitmap, uint32_t n_words, uint32_t *counters,
                             hipStream_t st)
{
    const uint32_t n_vec = n_words / 4;          // the bitmap is allocated in 16-byte units
    if (!n_vec) return;
    uint32_t blocks = (n_vec + 256u * 8u - 1u) / (256u * 8u);
    if (blocks > 256u) blocks = 256u;
    hipLaunchKernelGGL(k_bitmap_count, dim3(blocks), dim3(256), 0, st, (uint4 *)bitmap, n_vec,
                       counters);
}

void agh_launch_hashset_count(uint64_t *tab, uint32_t n_slots, const uint32_t *wave_cand,
                              uint32_t nw, uint32_t *counters, hipStream_t st)
{
    uint32_t blocks = (n_slots / 2 + 256u * 8u - 1u) / (256u * 8u);
    if (blocks > 256u) blocks = 256u;
    if (!blocks) blocks = 1u;
    hipLaunchKernelGGL(k_hashset_count, dim3(blocks), dim3(256), 0, st, tab, n_slots, wave_cand,
                       nw, counters);
}

// Matches of a lean scan whose record starts more than AGH_LEAN_BACK_CAP bytes back (agh_marks.giveups holds the
// position the verifier's own look-back stopped at): one workgroup per entry walks back 4 KiB per step until a
// delimiter shows (or the text begins) and enters the record start into the hash set.  Round 3 reran the whole
// segment -- up to 64 GiB -- on the numbered pipeline for one such record.
__global__ __launch_bounds__(256) void k_resolve_giveups(const uint8_t *__restrict__ text, uint32_t delim, agh_marks mk)
{
    __shared__ unsigned long long best;
    const uint32_t n = mk.counters[AGH_C_GIVEUPS] < mk.giveup_cap ? mk.counters[AGH_C_GIVEUPS] : mk.giveup_cap;
    const uint32_t dd = delim * 0x01010101u;
    for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
        uint64_t pos = mk.giveups[e];           // no delimiter in [pos, the match)
        uint64_t start = 0;                     // ... none at all: the record opens the text
        while (pos > 0) {
            if (threadIdx.x == 0) best = 0ull;
            __syncthreads();
            const uint64_t lo = pos > 4096u ? pos - 4096u : 0;
            // 16 bytes per thread, highest address first; unaligned 16-byte loads stay inside [lo, pos)
            const uint64_t back = (uint64_t)threadIdx.x * 16u;      // (pos - lo <= 4096: no thread reaches below lo)
            const uint64_t hi = back < pos - lo ? pos - back : lo;
            if (hi > lo) {
                const uint64_t at = hi >= lo + 16u ? hi - 16u : lo;
                unsigned long long found = 0;
                if (hi - at == 16u) {
                    typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
                    const u32x4_a1 v = *reinterpret_cast<const u32x4_a1 *>(text + at);
#pragma unroll
                    for (int d = 3; d >= 0 && !found; --d) {
                        const uint32_t x = v[d] ^ dd;
                        const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
                        if (z) found = at + 4u * (uint32_t)d + (uint32_t)((31 - __clz((int)z)) >> 3) + 1u;
                    }
                } else {
                    for (uint64_t i = hi; i > at && !found; --i)
                        if (text[i - 1] == delim) found = i;
                }
                if (found) atomicMax(&best, found);
            }
            __syncthreads();
            if (best) { start = best; break; }
            __syncthreads();
            pos = lo;
        }
        if (threadIdx.x == 0) lean_insert(mk, start);
        __syncthreads();
    }
}

void agh_launch_resolve_giveups(const void *text, uint32_t delim, const agh_marks &mk, hipStream_t st)
{
    if (!mk.giveups || !mk.giveup_cap) return;
    hipLaunchKernelGGL(k_resolve_giveups, dim3(64), dim3(256), 0, st, (const uint8_t *)text, delim, mk);
}

// A sharded count-only scan leaves its totals on the device for the all-reduce that follows on the same
// stream: acc[0] += matched, acc[1] += records (0: lean scans do not number records), acc[2] += "this
// segment's count-only scan gave up" (the host reruns it and the ranks reduce once more).
__global__ void k_accumulate_counts(const uint32_t *__restrict__ counters, unsigned long long *__restrict__ acc)
{
    acc[0] += counters[AGH_C_MATCHED];
    acc[2] += (counters[AGH_C_LEAN_FALLBACK] | counters[AGH_C_OVERFLOW]) ? 1ull : 0ull;
}

void agh_launch_accumulate_counts(const uint32_t *counters, uint64_t *acc, hipStream_t st)
{
    hipLaunchKernelGGL(k_accumulate_counts, dim3(1), dim3(1), 0, st, counters, (unsigned long long *)acc);
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
