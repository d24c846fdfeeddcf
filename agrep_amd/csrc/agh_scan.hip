// agh_scan.hip -- the k-error automaton on candidate windows (k_verify).  The automaton over every byte lives in agh_fullscan.hip (its own translation unit:
// the two kernel families compile in parallel).  See agh_sweep.hip for the data flow of a scan.
//
// Compiled ten times: once per number of errors (-DAGH_SCAN_K=0..8: the verify kernels of that k, both
// word widths -- a code object of its own each, so a process loads the one its query needs and not 7 MB
// of all of them) and once without the define (the dispatch by k).  Record output: agh_records.hip.
#include <stdlib.h>

#include "agh_verify_inl.h"

#define AGH_VGROUP 8u   // sweep-wave slices verified by one workgroup

#define AGH_SK_CAT2(a, b) a##b
#define AGH_SK_CAT(a, b) AGH_SK_CAT2(a, b)

#ifdef AGH_SCAN_K
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
    __syncthreads();
    VerifyCtx<WT, K> c;
    verify_ctx_init<WT, K, GEN>(c, text, n, q, lmask, mk, dbm);
    c.gtab = gtab;
    c.tspan = tspan;
    // grid-stride over the groups of AGH_VGROUP slices: a 64 GiB scan has 32768 groups of ~540
    // candidates; the mask table and the context are set up once per workgroup, not once per group
    const uint32_t n_groups = (nw - g_base + AGH_VGROUP - 1u) / AGH_VGROUP;
    for (uint32_t gi = blockIdx.x; gi < n_groups; gi += gridDim.x) {
        const uint32_t g0 = g_base + gi * AGH_VGROUP;       // first slice of this group
        __syncthreads();                        // everybody is done with pre[] of the last group
        if (threadIdx.x < AGH_VGROUP)           // the 8 counts arrive in one round trip
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
        // (uniform trip count: the lane-to-lane window comparison needs all 64 lanes)
        for (uint32_t c0 = 0; c0 < total; c0 += 256) {
            const uint32_t ci = c0 + threadIdx.x;
            const bool valid = ci < total;
            uint32_t sl = 0;
#pragma unroll
            for (uint32_t i = 1; i < AGH_VGROUP; ++i) sl += (pre[i] <= ci) ? 1u : 0u;
            const uint32_t w = g0 + sl;
            const uint64_t ent = valid ? cand[(uint64_t)w * AGH_SLICE_CAP + (ci - pre[sl])] : 0;
            VerifyWin win;
            win.j = win.ws = 0;
            win.span = win.mode = 0;
            if (valid) win = verify_locate<WT, K, NCH, LEAN>(c, ent);
            // (the same window finds the same records, numbered or not)
            if (verify_same_window_as_prev_lane(win)) win.mode = 0u;
            if (win.mode) verify_walk<WT, K, NCH, LEAN, MB, GEN>(c, ent, LEAN ? 0u : wave_prefix[w], win);
        }
    }
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
    {   // grid cap (workgroups loop over the groups); AGH_VERIFY_BLOCKS=0: one workgroup per group
        uint32_t cap = 16384u;                  // A/B on 64 GiB: 0 / 4096 / 8192 / 16384 all within 0.5 %
        if (a.verify_blocks >= 0) cap = (uint32_t)a.verify_blocks;
        if (cap && blocks > cap) blocks = cap;
    }
    if (a.general)                          // single-byte delimiters only (the host checks)
        hipLaunchKernelGGL((k_verify<WT, K, NCH, LEAN, false, true>), dim3(blocks), dim3(256), 0, st,
                           (const uint8_t *)a.text, a.n, a.q, (const WT *)a.mask, a.cand,
                           a.wave_cand, a.wave_prefix, w_hi, a.mk, a.dbm, gtab, tspan, g_base);
    else if (a.q.mb)
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
    if (a.gtab) {
        // gram offsets known: one warm-up byte + (m + 2k + spread) bytes per candidate, for count-only and
        // numbered scans alike (round 4 let a numbered window start at the sample's 16-byte chunk: 15 bytes
        // more, and no two samples of an occurrence shared a window -- 204 us per 8 GiB of config C3)
        const uint32_t tspan = (uint32_t)(a.q.m + 2 * a.q.k + 1) + a.gram_spread;
        const int tn = (int)((tspan + 15u) / 16u);
        if (sizeof(WT) == 4) {
            if (tn <= 2) launch_verify_n<WT, K, 2, LEAN>(a, a.gtab, tspan, st);
            else if (tn <= 3) launch_verify_n<WT, K, 3, LEAN>(a, a.gtab, tspan, st);
            else launch_verify_n<WT, K, 6, LEAN>(a, a.gtab, tspan, st);
        } else {
            if (tn <= 4) launch_verify_n<WT, K, 4, LEAN>(a, a.gtab, tspan, st);
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

// what: 0 numbered, 2 lean (count-only)
void AGH_SK_CAT(agh_launch_verify_k, AGH_SCAN_K)(const agh_scan_args &a, int what, hipStream_t st)
{
    if (a.wide) {
        if (what == 0) launch_verify_t<uint64_t, AGH_SCAN_K, false>(a, st);
        else launch_verify_t<uint64_t, AGH_SCAN_K, true>(a, st);
    } else {
        if (what == 0) launch_verify_t<uint32_t, AGH_SCAN_K, false>(a, st);
        else launch_verify_t<uint32_t, AGH_SCAN_K, true>(a, st);
    }
}

#else   // ---- the object without a k: dispatch ----------------------------------------

// ---------------------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------------------
void agh_launch_verify_k0(const agh_scan_args &, int, hipStream_t);
void agh_launch_verify_k1(const agh_scan_args &, int, hipStream_t);
void agh_launch_verify_k2(const agh_scan_args &, int, hipStream_t);
void agh_launch_verify_k3(const agh_scan_args &, int, hipStream_t);
void agh_launch_verify_k4(const agh_scan_args &, int, hipStream_t);
void agh_launch_verify_k5(const agh_scan_args &, int, hipStream_t);
void agh_launch_verify_k6(const agh_scan_args &, int, hipStream_t);
void agh_launch_verify_k7(const agh_scan_args &, int, hipStream_t);
void agh_launch_verify_k8(const agh_scan_args &, int, hipStream_t);

static void dispatch_k(const agh_scan_args &a, int what, hipStream_t st)
{
    switch (a.q.k) {
    case 0: agh_launch_verify_k0(a, what, st); break;
    case 1: agh_launch_verify_k1(a, what, st); break;
    case 2: agh_launch_verify_k2(a, what, st); break;
    case 3: agh_launch_verify_k3(a, what, st); break;
    case 4: agh_launch_verify_k4(a, what, st); break;
    case 5: agh_launch_verify_k5(a, what, st); break;
    case 6: agh_launch_verify_k6(a, what, st); break;
    case 7: agh_launch_verify_k7(a, what, st); break;
    case 8: agh_launch_verify_k8(a, what, st); break;
    default: break;
    }
}

void agh_launch_verify(const agh_scan_args &a, hipStream_t st) { dispatch_k(a, 0, st); }

void agh_launch_verify_lean(const agh_scan_args &a, hipStream_t st) { dispatch_k(a, 2, st); }
#endif
