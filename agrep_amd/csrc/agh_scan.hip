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
template <typename WT, int K, bool MB, bool GEN>
__global__ __launch_bounds__(AGH_FS_THREADS) void k_fullscan(
    const uint8_t *__restrict__ text, uint64_t n, agh_dev_query q,
    const WT *__restrict__ mask_g, const uint32_t *__restrict__ strip_prefix,
    const uint32_t *__restrict__ wave_prefix, uint32_t n_strips, agh_marks mk,
    const uint64_t *__restrict__ dbm)
{
    // slot 0 = the 256 bytes in front of the tile (warm-up halo), slots 1..256 = lane chunks
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    WT *lmask = reinterpret_cast<WT *>(lds);
    uint8_t *tile = lds + 256 * sizeof(WT);
    lmask[threadIdx.x] = mask_g[threadIdx.x];
    __syncthreads();

    const uint64_t tile_bytes = (uint64_t)AGH_FS_THREADS * AGH_FS_CHUNK;
    const uint64_t n_tiles = (n + tile_bytes - 1) / tile_bytes;
    const uint32_t total_delims = mk.counters[AGH_C_NDELIM];
    const WT finalbit = (WT)1 << (q.m - 1);
    const uint32_t dd = q.delim * 0x01010101u;
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    const uint32_t warm = ((uint32_t)(q.m + q.k + 1) + 15u) & ~15u;   // <= 80 bytes

    // state right after a record boundary: reset + re-fed delimiter byte (asearch.c:175-186)
    Automaton<WT, K> RF;
    RF.reset();
    const uint32_t rf_hit = RF.template step_q<GEN>(lmask[q.delim], finalbit, q) ? 1u : 0u;

    for (uint64_t tix = blockIdx.x; tix < n_tiles; tix += gridDim.x) {
        const uint64_t t0 = tix * tile_bytes;
        __syncthreads();
        // cooperative, coalesced global -> LDS of 257 chunks (halo + tile), 16 B pieces
        for (uint32_t pc = threadIdx.x; pc < (AGH_FS_THREADS + 1) * (AGH_FS_CHUNK / 16);
             pc += AGH_FS_THREADS) {
            const uint32_t slot = pc / (AGH_FS_CHUNK / 16), sub = pc % (AGH_FS_CHUNK / 16);
            const int64_t g = (int64_t)t0 - (int64_t)AGH_FS_CHUNK + (int64_t)pc * 16;
            uint4 v = make_uint4(fill4, fill4, fill4, fill4);
            if (g >= 0 && (uint64_t)g < n16) v = ld_stream(reinterpret_cast<const uint4 *>(text + g));
            *reinterpret_cast<uint4 *>(tile + slot * AGH_FS_SLOT + sub * 16) = v;
        }
        __syncthreads();

        const uint64_t cs = t0 + (uint64_t)threadIdx.x * AGH_FS_CHUNK;
        const uint8_t *mine = tile + (threadIdx.x + 1) * AGH_FS_SLOT;
        uint32_t my_delims = 0;
        uint64_t ce = cs + AGH_FS_CHUNK;
        if (ce > n) ce = n;
        if (cs < n && MB) {
            my_delims = dbm_count(dbm, cs, ce);
        } else if (cs < n) {
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
        uint32_t seen = 0;
        if (cs == 0) {
            A.template step_q<GEN>(lmask[q.head_byte], finalbit, q); // asearch.c:69-78
        } else {
            // rebuild the state from the m+k+1 (rounded to 16) bytes in front of the chunk
            const uint8_t *halo = tile + threadIdx.x * AGH_FS_SLOT + (AGH_FS_CHUNK - warm);
            for (uint32_t t = 0; t < warm / 16; ++t) {
                uint32_t h16 = 0, d16 = 0;
                if (MB) d16 = (uint32_t)dbm_bits64(dbm, cs - warm + 16u * t) & 0xffffu;
                fullscan_piece<WT, K, MB, GEN>(*reinterpret_cast<const uint4 *>(halo + 16 * t),
                                               lmask, finalbit, q, RF, rf_hit, A, seen, h16, d16);
            }
            seen = 0;                           // matches before cs belong to the previous lane
        }
        const uint32_t len = (uint32_t)(ce - cs);
        const uint32_t full = len >> 4;
        for (uint32_t t = 0; t < full; ++t) {
            uint32_t h16 = 0, d16 = 0;
            if (MB) d16 = (uint32_t)dbm_bits64(dbm, cs + 16u * t) & 0xffffu;
            fullscan_piece<WT, K, MB, GEN>(*reinterpret_cast<const uint4 *>(mine + 16 * t), lmask,
                                           finalbit, q, RF, rf_hit, A, seen, h16, d16);
            uint32_t ev = h16 | (rf_hit ? d16 : 0u);
            while (ev) {                        // rare: a record matched in this piece
                const uint32_t b = (uint32_t)__ffs((int)ev) - 1u;
                ev &= ev - 1u;
                const uint32_t below = (uint32_t)__popc(d16 & ((1u << b) - 1u));
                if ((h16 >> b) & 1u) mark_record(mk, rec + below, cs + 16u * t + b);
                if (rf_hit && ((d16 >> b) & 1u)) mark_record(mk, rec + below + 1u, cs + 16u * t + b + 1u);
            }
            rec += (uint32_t)__popc(d16);
        }
        bool seenb = seen != 0;
        for (uint32_t i = full * 16; i < len; ++i) {     // the last, partial piece of the text
            const uint32_t c = mine[i];
            bool hit = A.template step_q<GEN>(lmask[c], finalbit, q);
            if (hit && !seenb) {
                seenb = true;
                mark_record(mk, rec, cs + i);
            }
            if (MB ? dbm_bit(dbm, cs + i) != 0 : c == q.delim) {
                A.reset();
                ++rec;
                seenb = false;
                if (A.template step_q<GEN>(lmask[c], finalbit, q)) {
                    seenb = true;
                    mark_record(mk, rec, cs + i + 1);
                }
            }
        }
        if (ce == n && q.tail_virtual)          // asearch.c:87-91
            feed_virtual_tail<WT, K, false, GEN>(text, n, q, lmask, dbm, A, seenb, rec, 0, mk);
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
    const uint64_t tile_bytes = (uint64_t)AGH_FS_THREADS * AGH_FS_CHUNK;
    uint64_t n_tiles = (a.n + tile_bytes - 1) / tile_bytes;
    if (!n_tiles) return;
    uint32_t blocks = n_tiles > 65536 ? 65536u : (uint32_t)n_tiles;
    const size_t lds = 256 * sizeof(WT) + (size_t)(AGH_FS_THREADS + 1) * AGH_FS_SLOT;
#define AGH_FS_LAUNCH(MBV, GENV)                                                              \
    hipLaunchKernelGGL((k_fullscan<WT, K, MBV, GENV>), dim3(blocks), dim3(AGH_FS_THREADS), lds,  \
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

