// agh_scan.hip -- the k-error automaton kernels (verify on candidate windows, full scan over
// every byte).  See agh_sweep.hip for the data flow of a scan.
#include "agh_verify_inl.h"

#define AGH_VGROUP 8u   // sweep-wave slices verified by one workgroup

// ---------------------------------------------------------------------------------------
// verify: one workgroup per AGH_VGROUP sweep-wave slices, one lane per candidate sample
// ---------------------------------------------------------------------------------------
// GEN: the general automaton (edit costs, <exact> segments, -w / -x guards) on the windows.
template <typename WT, int K, int NCH, bool LEAN, bool MB, bool GEN>
__global__ __launch_bounds__(256) void k_verify(const uint8_t *__restrict__ text, uint64_t n,
                                                agh_dev_query q,
                                                const WT *__restrict__ mask_g,
                                                const uint64_t *__restrict__ cand,
                                                const uint32_t *__restrict__ wave_cand,
                                                const uint32_t *__restrict__ wave_prefix,
                                                uint32_t nw, agh_marks mk,
                                                const uint64_t *__restrict__ dbm,
                                                const uint64_t *__restrict__ gtab,
                                                uint32_t tspan, uint32_t g_base)
{
    __shared__ WT lmask[256];
    __shared__ uint32_t pre[AGH_VGROUP + 1];
    lmask[threadIdx.x] = mask_g[threadIdx.x];
    const uint32_t g0 = g_base + blockIdx.x * AGH_VGROUP;   // first slice of this workgroup
    if (threadIdx.x < AGH_VGROUP)               // the 8 counts arrive in one round trip
        pre[threadIdx.x + 1] = (g0 + threadIdx.x < nw) ? wave_cand[g0 + threadIdx.x] : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        pre[0] = 0;
        for (uint32_t i = 1; i <= AGH_VGROUP; ++i) {
            run += pre[i];
            pre[i] = run;
        }
    }
    __syncthreads();
    const uint32_t total = pre[AGH_VGROUP];
    if (total == 0) return;
    VerifyCtx<WT, K> c;
    verify_ctx_init<WT, K, GEN>(c, text, n, q, lmask, mk, dbm);
    c.gtab = gtab;
    c.tspan = tspan;

    for (uint32_t ci = threadIdx.x; ci < total; ci += 256) {
        uint32_t sl = 0;
#pragma unroll
        for (uint32_t i = 1; i < AGH_VGROUP; ++i) sl += (pre[i] <= ci) ? 1u : 0u;
        const uint32_t w = g0 + sl;
        const uint64_t ent = cand[(uint64_t)w * AGH_SLICE_CAP + (ci - pre[sl])];
        verify_candidate<WT, K, NCH, LEAN, MB, GEN>(c, ent, LEAN ? 0u : wave_prefix[w]);
    }
}

// ---------------------------------------------------------------------------------------
// fullscan: the automaton over every byte
// ---------------------------------------------------------------------------------------
// 16 consecutive text bytes (one LDS b128 read) through the automaton, branch-free: returns
// the 16-bit masks of new-match positions (h16) and delimiter-end positions (d16; an input in
// MB mode, where delimiter ends come from the bitmap).
template <typename WT, int K, bool MB, bool GEN>
__device__ __forceinline__ void fullscan_piece(uint4 v, const WT *lmask, WT finalbit,
                                               const agh_dev_query &q, const Automaton<WT, K> &RF,
                                               uint32_t rf_hit, Automaton<WT, K> &A,
                                               uint32_t &seen, uint32_t &h16, uint32_t &d16)
{
    const uint32_t dws[4] = {v.x, v.y, v.z, v.w};
    uint32_t h = 0, d = MB ? d16 : 0u;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const uint32_t byte = (dws[b >> 2] >> (8 * (b & 3))) & 0xffu;
        const uint32_t hit = A.template step_q<GEN>(lmask[byte], finalbit, q) ? 1u : 0u;
        const uint32_t isd = MB ? (d >> b) & 1u : ((byte == q.delim) ? 1u : 0u);
        h |= (hit & ~seen) << b;
        if (!MB) d |= isd << b;
        seen |= hit;
        if (isd) {
#pragma unroll
            for (int e = 0; e <= K; ++e) A.R[e] = RF.R[e];
            seen = rf_hit;
        }
    }
    h16 = h;
    d16 = d;
}

// GEN: the general automaton (non-unit costs / <exact> segments) instead of the unit-cost one.
//
// Data feeding.  The automaton is serial over bytes, so the parallelism is one CHUNK per lane
// (AGH_FS_CHUNK = 1 KiB = one census strip; a wave owns a 64 KiB tile), and every lane needs ITS
// next bytes while a coalesced load hands consecutive bytes to consecutive lanes.  The transpose
// goes through a small per-wave LDS ring: each round the wave gathers the next 64 bytes of all 64
// chunks with four dwordx4 loads (4 lanes x 16 B per chunk: 64-byte segments, every text byte
// fetched once), writes them into the ring, and every lane reads back its own 64 bytes (row
// stride 80 B: conflict-free b128 reads).  The next round's loads are in flight while the current
// 64 bytes run through the automaton.  5 KiB of LDS per wave -> 7 workgroups per CU instead of
// the 2 a 64 KiB tile allowed, and the warm-up replay (m+k+1 bytes of halo per chunk)
// is 8 % of a 1 KiB chunk instead of 31 % of a 256-byte one.
template <typename WT, int K, bool MB, bool GEN>
__global__ __launch_bounds__(AGH_FS_THREADS) void k_fullscan(
    const uint8_t *__restrict__ text, uint64_t n, agh_dev_query q,
    const WT *__restrict__ mask_g, const uint32_t *__restrict__ strip_prefix,
    const uint32_t *__restrict__ wave_prefix, uint32_t n_strips, agh_marks mk,
    const uint64_t *__restrict__ dbm)
{
    __shared__ WT lmask[256];
    __shared__ __attribute__((aligned(16))) uint8_t ring_all[(AGH_FS_THREADS / WAVE) * WAVE * AGH_FS_ROW];
    lmask[threadIdx.x] = mask_g[threadIdx.x];
    __syncthreads();

    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    uint8_t *ring = ring_all + wib * (WAVE * AGH_FS_ROW);
    const uint64_t tile_bytes = (uint64_t)WAVE * AGH_FS_CHUNK;
    const uint64_t n_tiles = (n + tile_bytes - 1) / tile_bytes;
    const uint32_t total_delims = mk.counters[AGH_C_NDELIM];
    const WT finalbit = (WT)1 << (q.m - 1);
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    const uint32_t warm = ((uint32_t)(q.m + q.k + 1) + 15u) & ~15u;   // <= 80 bytes

    // state right after a record boundary: reset + re-fed delimiter byte (asearch.c:175-186)
    Automaton<WT, K> RF;
    RF.reset();
    const uint32_t rf_hit = RF.template step_q<GEN>(lmask[q.delim], finalbit, q) ? 1u : 0u;

    // my part of the cooperative gather: 16 bytes of chunk (lane / 4 + 16 i), i = 0..3
    const uint32_t seg_lo = (uint32_t)lane >> 2, part = (uint32_t)lane & 3u;
    uint8_t *ring_w = ring + seg_lo * AGH_FS_ROW + part * 16u;
    const uint8_t *ring_r = ring + (uint32_t)lane * AGH_FS_ROW;

    for (uint64_t tile = (uint64_t)blockIdx.x * (AGH_FS_THREADS / WAVE) + wib; tile < n_tiles;
         tile += (uint64_t)gridDim.x * (AGH_FS_THREADS / WAVE)) {
        const uint64_t t0 = tile * tile_bytes;
        const uint64_t cs = t0 + (uint64_t)lane * AGH_FS_CHUNK;
        const bool mine = cs < n;
        uint64_t ce = cs + AGH_FS_CHUNK;
        if (ce > n) ce = n;
        const uint32_t len = mine ? (uint32_t)(ce - cs) : 0u;
        auto gather = [&](uint32_t r, uint4 (&g)[4]) {
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint64_t a = t0 + (uint64_t)(seg_lo + 16u * i) * AGH_FS_CHUNK + r * AGH_FS_ROUND + part * 16u;
                g[i] = a < n16 ? ld_stream(reinterpret_cast<const uint4 *>(text + a))
                               : make_uint4(fill4, fill4, fill4, fill4);
            }
        };
        uint4 g[4];
        gather(0, g);

        uint32_t rec = 0;
        Automaton<WT, K> A;
        A.reset();
        uint32_t seen = 0;
        if (mine) {
            const uint64_t strip = cs >> AGH_STRIP_SHIFT;       // chunk == strip
            rec = strip < n_strips ? wave_prefix[strip / AGH_WAVE_STRIPS] + strip_prefix[strip] : total_delims;
            if (cs == 0) {
                A.template step_q<GEN>(lmask[q.head_byte], finalbit, q); // asearch.c:69-78
            } else {
                // rebuild the state from the m+k+1 (rounded to 16) bytes in front of the chunk
                for (uint32_t t = 0; t < warm / 16; ++t) {
                    uint32_t h16 = 0, d16 = 0;
                    if (MB) d16 = (uint32_t)dbm_bits64(dbm, cs - warm + 16u * t) & 0xffffu;
                    fullscan_piece<WT, K, MB, GEN>(
                        *reinterpret_cast<const uint4 *>(text + cs - warm + 16u * t), lmask, finalbit, q,
                        RF, rf_hit, A, seen, h16, d16);
                }
                seen = 0;                       // matches before cs belong to the previous lane
            }
        }
        bool seenb = false;
        for (uint32_t r = 0; r < AGH_FS_CHUNK / AGH_FS_ROUND; ++r) {
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i)
                *reinterpret_cast<uint4 *>(ring_w + 16u * i * AGH_FS_ROW) = g[i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the whole wave's rows are in
            __builtin_amdgcn_wave_barrier();
            uint4 v[4];
#pragma unroll
            for (uint32_t p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const uint4 *>(ring_r + 16u * p);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // read before the next round overwrites
            __builtin_amdgcn_wave_barrier();
            if (r + 1 < AGH_FS_CHUNK / AGH_FS_ROUND) gather(r + 1, g);   // in flight during the walk
#pragma unroll
            for (uint32_t p = 0; p < 4; ++p) {
                const uint32_t off = r * AGH_FS_ROUND + 16u * p;
                if (off + 16u <= len) {
                    uint32_t h16 = 0, d16 = 0;
                    if (MB) d16 = (uint32_t)dbm_bits64(dbm, cs + off) & 0xffffu;
                    fullscan_piece<WT, K, MB, GEN>(v[p], lmask, finalbit, q, RF, rf_hit, A, seen, h16, d16);
                    uint32_t ev = h16 | (rf_hit ? d16 : 0u);
                    while (ev) {                        // rare: a record matched in this piece
                        const uint32_t b = (uint32_t)__ffs((int)ev) - 1u;
                        ev &= ev - 1u;
                        const uint32_t below = (uint32_t)__popc(d16 & ((1u << b) - 1u));
                        if ((h16 >> b) & 1u) mark_record(mk, rec + below, cs + off + b);
                        if (rf_hit && ((d16 >> b) & 1u)) mark_record(mk, rec + below + 1u, cs + off + b + 1u);
                    }
                    rec += (uint32_t)__popc(d16);
                } else if (off < len) {                 // the last, partial piece of the text
                    const uint32_t dws[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
                    seenb = seen != 0;
                    for (uint32_t i = 0; off + i < len; ++i) {
                        const uint32_t c = (dws[i >> 2] >> (8u * (i & 3u))) & 0xffu;
                        const bool hit = A.template step_q<GEN>(lmask[c], finalbit, q);
                        if (hit && !seenb) {
                            seenb = true;
                            mark_record(mk, rec, cs + off + i);
                        }
                        if (MB ? dbm_bit(dbm, cs + off + i) != 0 : c == q.delim) {
                            A.reset();
                            ++rec;
                            seenb = false;
                            if (A.template step_q<GEN>(lmask[c], finalbit, q)) {
                                seenb = true;
                                mark_record(mk, rec, cs + off + i + 1);
                            }
                        }
                    }
                    seen = seenb ? 1u : 0u;
                }
            }
        }
        if (mine && ce == n && q.tail_virtual)          // asearch.c:87-91
            feed_virtual_tail<WT, K, false, GEN>(text, n, q, lmask, dbm, A, seen != 0, rec, 0, mk);
    }
}

// ---------------------------------------------------------------------------------------
// record output: bounds of matched records and their bytes, straight from the staged text
// (what output()/s_output() derive on the CPU: agrep.c:3805-3956, sgrep.c:1274-1333)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_match_bounds(const uint8_t *__restrict__ text,
                                                      uint64_t n, agh_dev_query q,
                                                      const uint64_t *__restrict__ dbm,
                                                      const uint64_t *__restrict__ pos,
                                                      uint32_t cnt, uint64_t *__restrict__ start,
                                                      uint64_t *__restrict__ end)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    uint64_t e = pos[i];
    if (e > n) e = n;
    uint64_t s = e, en = e;
    if (q.dlen > 1) {
        // record = (end of the last delimiter in front of e, start of the next delimiter]
        const int64_t d = dbm_prev(dbm, e, ~0ull);
        s = d >= 0 ? (uint64_t)d + 1 : 0;
        while (en < n && !dbm_bit(dbm, en)) ++en;      // en = end byte of the next delimiter
        if (en < n) en = en + 1 >= q.dlen ? en + 1 - q.dlen : 0;
        else en = virtual_close_start(text, n, q, dbm);   // closed by the appended delimiter
        if (en < s) en = s;
    } else {
        while (s > 0 && text[s - 1] != q.delim) --s;   // 1 + last delimiter in front of e
        while (en < n && text[en] != q.delim) ++en;    // first delimiter at or after e
    }
    start[i] = s;
    end[i] = en;
}

// One wave per record: out[off[i] .. off[i] + len) = text[start[i] .. end[i]).
__global__ __launch_bounds__(256) void k_gather_records(const uint8_t *__restrict__ text,
                                                        const uint64_t *__restrict__ start,
                                                        const uint64_t *__restrict__ end,
                                                        const uint64_t *__restrict__ off,
                                                        uint32_t cnt, uint8_t *__restrict__ out)
{
    const uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    if (r >= cnt) return;
    const uint64_t s = start[r], len = end[r] - s, o = off[r];
    for (uint64_t b = (uint64_t)lane_id(); b < len; b += WAVE) out[o + b] = text[s + b];
}

// Matches of a later segment: positions and record numbers become absolute.
__global__ __launch_bounds__(256) void k_offset_matches(uint64_t *__restrict__ pos,
                                                        uint32_t *__restrict__ rec, uint32_t cnt,
                                                        uint64_t pos_off, uint32_t rec_off)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    pos[i] += pos_off;
    if (rec) rec[i] += rec_off;
}

void agh_launch_offset_matches(uint64_t *pos, uint32_t *rec, uint32_t cnt, uint64_t pos_off,
                               uint32_t rec_off, hipStream_t st)
{
    if (!cnt || (!pos_off && !rec_off)) return;
    hipLaunchKernelGGL(k_offset_matches, dim3((cnt + 255u) / 256u), dim3(256), 0, st, pos, rec, cnt,
                       pos_off, rec_off);
}

void agh_launch_match_bounds(const void *text, uint64_t n, const agh_dev_query &q,
                             const uint64_t *dbm, const uint64_t *pos, uint32_t cnt,
                             uint64_t *start, uint64_t *end, hipStream_t st)
{
    if (!cnt) return;
    hipLaunchKernelGGL(k_match_bounds, dim3((cnt + 255u) / 256u), dim3(256), 0, st,
                       (const uint8_t *)text, n, q, dbm, pos, cnt, start, end);
}

void agh_launch_gather_records(const void *text, const uint64_t *start, const uint64_t *end,
                               const uint64_t *off, uint32_t cnt, void *out, hipStream_t st)
{
    if (!cnt) return;
    hipLaunchKernelGGL(k_gather_records, dim3((cnt + 3u) / 4u), dim3(256), 0, st,
                       (const uint8_t *)text, start, end, off, cnt, (uint8_t *)out);
}

// ---------------------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------------------
template <typename WT, int K, int NCH, bool LEAN>
static void launch_verify_n(const agh_scan_args &a, const uint64_t *gtab, uint32_t tspan,
                            hipStream_t st)
{
    // slices [w_begin, w_end) of a lean part (w_begin is a multiple of AGH_VGROUP), else all
    const uint32_t w_hi = (a.w_end && a.w_end < a.nw) ? a.w_end : a.nw;
    const uint32_t g_base = a.w_begin;
    if (w_hi <= g_base) return;
    uint32_t blocks = (w_hi - g_base + AGH_VGROUP - 1u) / AGH_VGROUP;
    if (a.general)                          // single-byte delimiters only (the host checks)
        hipLaunchKernelGGL((k_verify<WT, K, NCH, LEAN, false, true>), dim3(blocks), dim3(256), 0, st,
                           (const uint8_t *)a.text, a.n, a.q, (const WT *)a.mask, a.cand,
                           a.wave_cand, a.wave_prefix, w_hi, a.mk, a.dbm, gtab, tspan, g_base);
    else if (a.q.dlen > 1)
        hipLaunchKernelGGL((k_verify<WT, K, NCH, LEAN, true, false>), dim3(blocks), dim3(256), 0, st,
                           (const uint8_t *)a.text, a.n, a.q, (const WT *)a.mask, a.cand,
                           a.wave_cand, a.wave_prefix, w_hi, a.mk, a.dbm, gtab, tspan, g_base);
    else
        hipLaunchKernelGGL((k_verify<WT, K, NCH, LEAN, false, false>), dim3(blocks), dim3(256), 0, st,
                           (const uint8_t *)a.text, a.n, a.q, (const WT *)a.mask, a.cand,
                           a.wave_cand, a.wave_prefix, w_hi, a.mk, a.dbm, gtab, tspan, g_base);
}

template <typename WT, int K, bool LEAN>
static void launch_verify_t(const agh_scan_args &a, hipStream_t st)
{
    // window span = max(m+k+1, 16) + q + m + k bytes, fetched as ceil(span/16) pieces
    const int lw = a.q.m + a.q.k + 1 > 16 ? a.q.m + a.q.k + 1 : 16;
    const int nch = (lw + a.q.fq + a.q.m + a.q.k + 15) / 16;
    if (a.gtab && (LEAN || (int)(((uint32_t)(a.q.m + 2 * a.q.k + 1) + a.gram_spread + 15u + 15u) / 16u) < nch)) {
        // gram offsets known: one warm-up byte + (m + 2k + spread) bytes per candidate; numbered
        // scans may start up to 15 bytes earlier (at the sample's 16-byte chunk)
        const uint32_t tspan = (uint32_t)(a.q.m + 2 * a.q.k + 1) + a.gram_spread + (LEAN ? 0u : 15u);
        const int tn = (int)((tspan + 15u) / 16u);
        // (numbered scans come here only when the window gets shorter by at least one 16-byte
        // piece: the table costs two dependent loads per candidate)
        if (sizeof(WT) == 4) {
            if (tn <= 2 && LEAN) launch_verify_n<WT, K, LEAN ? 2 : 3, LEAN>(a, a.gtab, tspan, st);
            else if (tn <= 3) launch_verify_n<WT, K, 3, LEAN>(a, a.gtab, tspan, st);
            else launch_verify_n<WT, K, 6, LEAN>(a, a.gtab, tspan, st);
        } else {
            if (tn <= 4 && LEAN) launch_verify_n<WT, K, LEAN ? 4 : 7, LEAN>(a, a.gtab, tspan, st);
            else if (tn <= 7) launch_verify_n<WT, K, 7, LEAN>(a, a.gtab, tspan, st);
            else launch_verify_n<WT, K, 10, LEAN>(a, a.gtab, tspan, st);
        }
        return;
    }
    if (sizeof(WT) == 4) {                  // m <= 32: span <= 85
        if (nch <= 3) launch_verify_n<WT, K, 3, LEAN>(a, nullptr, 0u, st);
        else launch_verify_n<WT, K, 6, LEAN>(a, nullptr, 0u, st);
    } else {                                // m <= 64: span <= 149
        if (nch <= 7) launch_verify_n<WT, K, 7, LEAN>(a, nullptr, 0u, st);
        else launch_verify_n<WT, K, 10, LEAN>(a, nullptr, 0u, st);
    }
}

template <typename WT, int K>
static void launch_fullscan_t(const agh_scan_args &a, hipStream_t st)
{
    const uint64_t tile_bytes = (uint64_t)WAVE * AGH_FS_CHUNK;          // one wave, 64 KiB
    const uint64_t n_tiles = (a.n + tile_bytes - 1) / tile_bytes;
    if (!n_tiles) return;
    const uint64_t want = (n_tiles + (AGH_FS_THREADS / WAVE) - 1) / (AGH_FS_THREADS / WAVE);
    const uint32_t blocks = want > 16384 ? 16384u : (uint32_t)want;    // grid-stride beyond that
#define AGH_FS_LAUNCH(MBV, GENV)                                                              \
    hipLaunchKernelGGL((k_fullscan<WT, K, MBV, GENV>), dim3(blocks), dim3(AGH_FS_THREADS), 0,    \
                       st, (const uint8_t *)a.text, a.n, a.q, (const WT *)a.mask,                \
                       a.strip_prefix, a.wave_prefix, a.n_strips, a.mk, a.dbm)
    const bool mbv = a.q.dlen > 1, genv = a.general != 0;
    if (mbv && genv) AGH_FS_LAUNCH(true, true);
    else if (mbv) AGH_FS_LAUNCH(true, false);
    else if (genv) AGH_FS_LAUNCH(false, true);
    else AGH_FS_LAUNCH(false, false);
#undef AGH_FS_LAUNCH
}

template <typename WT>
static void dispatch_k(const agh_scan_args &a, int what, hipStream_t st)
{
#define AGH_CASE(KK)                                            \
    case KK:                                                    \
        if (what == 0) launch_verify_t<WT, KK, false>(a, st);   \
        else if (what == 1) launch_fullscan_t<WT, KK>(a, st);   \
        else launch_verify_t<WT, KK, true>(a, st);              \
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
    if (a.wide) dispatch_k<uint64_t>(a, 0, st);
    else dispatch_k<uint32_t>(a, 0, st);
}

void agh_launch_fullscan(const agh_scan_args &a, hipStream_t st)
{
    if (a.wide) dispatch_k<uint64_t>(a, 1, st);
    else dispatch_k<uint32_t>(a, 1, st);
}

void agh_launch_verify_lean(const agh_scan_args &a, hipStream_t st)
{
    if (a.wide) dispatch_k<uint64_t>(a, 2, st);
    else dispatch_k<uint32_t>(a, 2, st);
}

