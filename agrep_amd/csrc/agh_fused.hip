// agh_fused.hip -- count-only (lean) scans in ONE kernel: six or eight sweeping waves and two verifying
// waves per workgroup.
//
// The two-kernel lean pipeline (k_sweep, then k_verify) spends ~7 % of a 64 GiB scan in k_verify,
// and running the verifier on a second stream only displaces sweep workgroups.  Here the verifiers
// are two EXTRA waves of each sweep workgroup: they take no LDS or wave slot away from the
// sweepers (16 or 10 of a CU's 32 wave slots in all), their line fetches ride along
// the stream, and candidates never go through HBM -- they are handed over 64 at a time through a
// small LDS ring.  Measured (scripts/ab_fused.py, same process, same corpus): 64 GiB, k = 2:
// 11.0-11.3 ms for the two kernels, 10.6-10.8 ms fused; with the verifier switched off the kernel
// takes 10.1-10.4 ms, so about two thirds of the verifier's cost stay visible -- its ~4 % of extra,
// random HBM traffic is not free next to a stream that already runs at the HBM ceiling.
// Tried on top and dropped (slower or equal, scripts/ab_fused.py): static range assignment (-6 ... -10 %:
// under saturation the workgroups of XCDs 0, 1, 6, 7 stream a range in ~88 us and those of XCDs
// 2-5 in 130-170 us (scripts/fused_trace.py; inside a workgroup the four waves differ by 2 us) --
// the ticket counter is what balances them); other ticket sizes (AGH_FUSED_RANGE_KB: 96 KiB -8 %,
// 128 KiB 0 ... -2 %, 512 KiB -1 ... -3 %, 1024 KiB +0.5 % at 64 GiB and -5 % at 16 GiB; 64 KiB and
// below saturate the counter near 70 requests/us: 14.7 ms); streaming across range boundaries with
// the ticket read deferred behind the stream (+1 %: the boundary bubble is not what costs); 1, 3 or
// 4 verifying waves (within 0.1 %).  Round 2 launched four workgroups of 4 + 2 waves per CU, of which only
// three are resident (4 x 38 KiB of LDS do not fit next to whatever the CU keeps for itself, k_sweep's
// 4 x 35 KiB do; the trace shows a quarter of them starting when the first ones leave).  Round 3 measured
// the shape of the grid -- sweeping waves per workgroup x workgroups per CU -- and ships 2 x (6 + 2) waves
// per CU for H = 2 and 1 x (8 + 2) for the lighter sample shapes (launch_fused has the numbers).  A 36 KiB
// layout (2^17 filter bits: make FT_BITS=14) with 16 sweeping waves on a CU was measured as well: not
// faster (more chance candidates, and twelve streaming waves saturate the HBM).
// Two things mattered on the way (kept in mind for any kernel built like this one):
//   * the work counter needs a cache line of its own: on the scan counters' line the sweepers'
//     ticket atomics queued behind the verifiers' stores and the kernel took 14 ms;
//   * the verifier takes the address of the query struct, which moves the kernel-argument copy
//     to scratch: the sweepers read their two fields from a register copy.
//
//   sweeping wave:  persistent; takes the next 256 KiB wave range from a global ticket counter
//                   (requested one range ahead), streams it exactly like k_sweep<H, lean>, queues
//                   hits in its private LDS queue; when 64 are queued it takes a ring chunk
//                   (LDS ticket), copies them and publishes the chunk.
//   verifying wave: takes the next chunk number, waits for that chunk, verifies it one lane per
//                   candidate with the same verify_locate() / verify_walk() the stand-alone
//                   k_verify runs; matched records go into the hash set of record starts.  Leaves
//                   when the four sweepers are done and its chunk number was never handed out.
//                   Sweeping waves that have run out of ranges turn into verifying waves.
//
// Reference semantics are those of the verifier (asearch.c:66-324 on the candidate windows); this
// file only changes where the work runs.
//
// Compiled once per k (-DAGH_FU_K=0..3, four objects in parallel): every instance carries the
// fully unrolled automaton of its k, as in agh_scan.hip.
#include <stdlib.h>

#include "agh_sweep_inl.h"
#include "agh_verify_inl.h"

#ifndef AGH_FU_K
#error "compile with -DAGH_FU_K=0..3"
#endif
// Verifying waves per workgroup (next to 6 or 8 sweeping ones, launch_fused; the kernels use 53-67 VGPRs /
// ~106 SGPRs).  Measured on the 64 GiB bench corpus in round 2 with four sweeping waves per workgroup,
// NV = 1, 2, 3, 4 were all within 0.1 % (a build-time A/B hook, since removed): the kernel sits at the HBM
// ceiling either way; two keeps headroom for candidate-dense text.
#define AGH_FU_NV 2
#define AGH_FU_CHUNKS 4u                 // LDS ring: 4 chunks of 64 candidates (2 KiB)
// a wave's private queue: handed over at 64, one emit round adds at most 16 (one lane's hit bits;
// 32 with H == 2)
#define AGH_FU_CQ_LEN 96u
// Waits poll LDS every ~0.5 us (s_sleep 16).  The longest legitimate wait is the kernel's own run
// time (a verifier whose sweepers find nothing): tens of ms.  After ~4 s a wait gives up and
// raises AGH_C_LEAN_FALLBACK, which makes the host redo the segment with the numbered pipeline.
#define AGH_FU_SPIN_LIMIT (1u << 23)
// small tickets at the end of the text (see launch_fused): MiB of text handed out in tickets of ... KiB
#ifndef AGH_FU_TAIL_MB_DEFAULT
#define AGH_FU_TAIL_MB_DEFAULT 512u
#endif
#define AGH_FU_TAIL_KB_DEFAULT 128u

#ifdef AGH_FU_TRACE
// diagnostics build (make EXP=1): when each wave stopped sweeping / left the kernel, in 100 MHz ticks
__device__ uint64_t g_fu_trace[4 * 8192];
#define AGH_FU_STAMP(slot) do { if (lane == 0) g_fu_trace[(slot) * 8192 + (blockIdx.x * 8u + wib) % 8192u] = wall_clock64(); } while (0)
#else
#define AGH_FU_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ uint32_t lds_peek(const uint32_t *p)
{
    return __atomic_load_n(p, __ATOMIC_RELAXED);
}

template <typename WT, int H, int MODE, int K, int NCH, int NV, int NS>
__global__ __launch_bounds__(64 * (NS + NV)) void k_sweep_fused(
    const uint4 *__restrict__ text, uint64_t n, uint64_t n_full_strips, agh_dev_query q,
    const uint8_t *__restrict__ ftab_g, const WT *__restrict__ mask_g, agh_marks mk,
    const uint64_t *__restrict__ gtab, uint32_t tspan, uint32_t n_ranges,
    uint32_t *__restrict__ work, uint32_t range_strips, uint32_t n_big, uint32_t tail_strips)
{
    static_assert((MODE & 4) && !(MODE & 8), "lean sweeps with one-byte delimiters only");
        __shared__ __attribute__((aligned(16))) uint8_t ftab[AGH_FT_SIZE];
    __shared__ uint64_t cq_all[NS * AGH_FU_CQ_LEN];
    __shared__ uint64_t ring[AGH_FU_CHUNKS * 64];
    __shared__ WT lmask[256];
    __shared__ uint32_t ring_ready[AGH_FU_CHUNKS];   // ticket + 1 of the chunk that is published
    __shared__ uint32_t ring_freed[AGH_FU_CHUNKS];   // ticket + 1 of the chunk that was consumed last
    __shared__ uint32_t ring_count[AGH_FU_CHUNKS];
    __shared__ uint32_t tickets, done, next_chunk;

    if (threadIdx.x < 256) {
        const uint4 *src = reinterpret_cast<const uint4 *>(ftab_g);
        uint4 *dst = reinterpret_cast<uint4 *>(ftab);
        constexpr int PER = AGH_FT_SIZE / 16 / 256;
        uint4 tmp[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) tmp[i] = src[threadIdx.x + i * 256];
#pragma unroll
        for (int i = 0; i < PER; ++i) dst[threadIdx.x + i * 256] = tmp[i];
    } else {
        const uint32_t t = threadIdx.x - 256u;
        for (uint32_t i = t; i < 256u; i += 64u * (NS + NV - 4)) lmask[i] = mask_g[i];
        if (t < AGH_FU_CHUNKS) ring_ready[t] = ring_freed[t] = 0u;
        if (t == 0) tickets = done = next_chunk = 0u;
    }
    __syncthreads();
    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    AGH_FU_STAMP(0);

    if (wib < (uint32_t)NS) {
        // ---------------------------------------------------------------- a sweeping wave
        // (the verifier takes the address of q, which sends the kernel-argument copy to scratch: the
        // sweepers probe with a register copy of the two fields they read)
        agh_dev_query qs;
        qs.qmask = q.qmask;
        qs.fold = q.fold;
        uint64_t *cq = cq_all + wib * AGH_FU_CQ_LEN;
        uint32_t qn = 0;
        // hand the first `take` queued candidates to the verifier, keep the rest
        auto hand_over = [&](uint32_t take) {
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(&tickets, 1u);
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            const uint32_t slot = t % AGH_FU_CHUNKS;
            if (t >= AGH_FU_CHUNKS) {               // the chunk's previous contents (ticket t - CHUNKS)
                uint32_t spins = 0;
                while (lds_peek(&ring_freed[slot]) != t - AGH_FU_CHUNKS + 1u) {
                    if (++spins > AGH_FU_SPIN_LIMIT) {  // (see the verifier's wait)
                        mk.counters[AGH_C_LEAN_FALLBACK] = 1u;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(16);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if ((uint32_t)lane < take) ring[slot * 64u + (uint32_t)lane] = cq[lane];
            if (lane == 0) ring_count[slot] = take;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __atomic_store_n(&ring_ready[slot], t + 1u, __ATOMIC_RELAXED);
            const uint32_t rest = qn - take;        // < 32: move it to the front
            uint64_t keep = 0;
            if ((uint32_t)lane < rest) keep = cq[take + (uint32_t)lane];
            if ((uint32_t)lane < rest) cq[lane] = keep;
            qn = rest;
        };
        const uint32_t rc0[4] = {0u, 0u, 0u, 0u};
        // nx3 (H == 2): first dword of strip s+4 (the last sample of lane 63 reaches into it)
        auto supertile = [&](uint4 v0, uint4 v1, uint4 v2, uint4 v3, uint64_t s, uint32_t nx3) {
            uint32_t a = 0, hits = 0;
            uint32_t x0 = 0, x1 = 0, x2 = 0, x3 = 0;
            if (H == 2) {
                x0 = next_lane_dword(v0.x, (uint32_t)__builtin_amdgcn_readlane((int)v1.x, 0));
                x1 = next_lane_dword(v1.x, (uint32_t)__builtin_amdgcn_readlane((int)v2.x, 0));
                x2 = next_lane_dword(v2.x, (uint32_t)__builtin_amdgcn_readlane((int)v3.x, 0));
                x3 = next_lane_dword(v3.x, nx3);
            }
            sweep_chunk<H, MODE>(v0, 0u, qs, ftab, a, hits, 0, 0u, x0);
            sweep_chunk<H, MODE>(v1, 0u, qs, ftab, a, hits, 4, 0u, x1);
            sweep_chunk<H, MODE>(v2, 0u, qs, ftab, a, hits, 8, 0u, x2);
            sweep_chunk<H, MODE>(v3, 0u, qs, ftab, a, hits, 12, 0u, x3);
            if (__ballot(hits != 0))
                emit_candidates_to<H>(hits, s, rc0, cq, qn, [&]() { hand_over(64u); });
        };
        const uint64_t n_dw = ((n + 15) & ~(uint64_t)15) / 4;      // readable dwords of the text
        auto first_dword_of = [&](uint64_t st) -> uint32_t {       // (uniform; 0 past the text)
            if (H != 2) return 0u;
            const uint64_t i = st * 256u;
            return i < n_dw ? reinterpret_cast<const uint32_t *>(text)[i] : 0u;
        };

        // the first range is the wave's own number (4096 waves asking one counter at the same
        // moment would wait ~45 us for the last answer); the counter hands out the rest
        const uint32_t first_dynamic = gridDim.x * (uint32_t)NS;
        uint32_t r = blockIdx.x * (uint32_t)NS + wib;
        uint32_t n_done = 0;
        while (r < n_ranges) {
            ++n_done;
            uint32_t r_next = 0;
            bool have_next = false;                 // (lane 0's view)
            // tickets 0 .. n_big-1 are ranges of range_strips KiB; the text behind them is handed out
            // in smaller tickets of tail_strips KiB, so that the waves finish closer together
            const uint64_t s0 = r < n_big ? (uint64_t)r * range_strips
                                          : (uint64_t)n_big * range_strips + (uint64_t)(r - n_big) * tail_strips;
            uint64_t s1 = s0 + (r < n_big ? range_strips : tail_strips);
            if (s1 > n_full_strips) s1 = n_full_strips;
            const uint64_t s_half = s0 + (((s1 - s0) >> 1) & ~(uint64_t)3);
            uint64_t s = s0;
            if (s + 4 <= s1) {
                const uint4 *p = text + s * 64 + lane;
                uint4 c0 = ld_stream(p), c1 = ld_stream(p + 64), c2 = ld_stream(p + 128), c3 = ld_stream(p + 192);
                for (; s + 8 <= s1; s += 4) {
                    const uint4 *pn = text + (s + 4) * 64 + lane;
                    uint4 n0 = ld_stream(pn), n1 = ld_stream(pn + 64), n2 = ld_stream(pn + 128), n3 = ld_stream(pn + 192);
                    // the next range's ticket is requested mid-range: in flight behind the stream
                    if (s == s_half && lane == 0) { r_next = first_dynamic + atomicAdd(work, 1u); have_next = true; }
                    supertile(c0, c1, c2, c3, s, H == 2 ? (uint32_t)__builtin_amdgcn_readlane((int)n0.x, 0) : 0u);
                    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
                }
                supertile(c0, c1, c2, c3, s, first_dword_of(s + 4));
                s += 4;
            }
            for (; s < s1; ++s) {                   // < 4 strips left in the range
                const uint4 v0 = ld_stream(text + s * 64 + lane);
                uint32_t a = 0, hits = 0;
                sweep_chunk<H, MODE>(v0, 0u, qs, ftab, a, hits, 0, 0u,
                                     H == 2 ? next_lane_dword(v0.x, first_dword_of(s + 1)) : 0u);
                if (H == 2) hits >>= 24;
                if (__ballot(hits != 0))
                    emit_candidates_to<H>(hits, s, rc0, cq, qn, [&]() { hand_over(64u); });
            }
            if (!have_next && lane == 0) r_next = first_dynamic + atomicAdd(work, 1u);   // (short last range)
            r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r_next);
        }
        AGH_FU_STAMP(1);
#ifdef AGH_FU_TRACE
        if (lane == 0) g_fu_trace[3 * 8192 + (blockIdx.x * 8u + wib) % 8192u] = n_done;
#endif
        while (qn) hand_over(qn < 64u ? qn : 64u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) atomicAdd(&done, 1u);
    }

    // ---------------------------------------------------------------------- verifying:
    // the two verifying waves from the start, and every sweeping wave once it has run out of
    // ranges (so the chunks handed over last do not wait in line for two waves while four idle).
    // A consumer takes the next chunk number, waits until that chunk is published, verifies it
    // one lane per candidate and frees the ring slot.
    VerifyCtx<WT, K> c;
    verify_ctx_init<WT, K, false>(c, reinterpret_cast<const uint8_t *>(text), n, q, lmask, mk,
                                  nullptr);
    c.gtab = gtab;
    c.tspan = tspan;
    uint32_t total = 0;
    for (;;) {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(&next_chunk, 1u);
        v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        const uint32_t slot = v % AGH_FU_CHUNKS;
        bool more = true;
        for (uint32_t spins = 0;; ++spins) {
            if (lds_peek(&ring_ready[slot]) == v + 1u) break;
            if (lds_peek(&done) == (uint32_t)NS && lds_peek(&tickets) <= v) { more = false; break; }
            if (spins > AGH_FU_SPIN_LIMIT) {        // never seen; a stuck protocol must not hang the GPU
                mk.counters[AGH_C_LEAN_FALLBACK] = 1u;
                more = false;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        if (!more) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t cnt = lds_peek(&ring_count[slot]);
        total += cnt;
        const bool valid = (uint32_t)lane < cnt;
        const uint64_t ent = valid ? ring[slot * 64u + (uint32_t)lane] : 0;
        VerifyWin win;
        win.j = win.ws = 0;
        win.span = win.mode = 0;
        if (valid) win = verify_locate<WT, K, NCH, true>(c, ent);
        if (verify_same_window_as_prev_lane(win)) win.mode = 0u;
        if (win.mode) verify_walk<WT, K, NCH, true, false, false>(c, ent, 0u, win);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __atomic_store_n(&ring_freed[slot], v + 1u, __ATOMIC_RELAXED);
    }
    if (lane == 0 && total) atomicAdd(&mk.counters[AGH_C_CAND], total);
    AGH_FU_STAMP(2);
}

template <typename WT, int H, int MODE, int NCH>
static void launch_fused(const agh_fused_args &a, uint32_t tspan, hipStream_t st)
{
    constexpr int K = AGH_FU_K;
    constexpr int NV = AGH_FU_NV;
    constexpr int NS = H == 2 ? 6 : 8;          // sweeping waves per workgroup (below)
    const uint64_t n_full = a.n >> AGH_STRIP_SHIFT;
    // KiB per ticket (a multiple of 8: the ticket for the next range is requested half-way); the
    // diagnostics overrides are clamped -- a zero grid or a wrapped range size must not reach the launch
    uint32_t range_strips = AGH_WAVE_STRIPS;
    if (a.tune && a.tune->fused_range_kb >= 0) {
        const unsigned long v = (unsigned long)a.tune->fused_range_kb;
        range_strips = (uint32_t)((v > 65536ul ? 65536ul : v) + 7ul) & ~7u;
    }
    if (range_strips < 16u) range_strips = 16u;
    // the last AGH_FUSED_TAIL_MB of the text go out in tickets of AGH_FUSED_TAIL_KB: the waves stop
    // within a small ticket's time of each other instead of a large one's.  scripts/ab_round3.py,
    // 64 GiB k = 2 / k = 0: no tail 10.61 / 10.15 ms, 512 MiB in 128 KiB tickets 10.48 / 10.05,
    // 256 MiB in 64 KiB 10.49 / 10.03, 512 MiB in 32 KiB 10.71 / 10.16; at 8 GiB 128 KiB tickets are
    // neutral (1.413 -> 1.419 ms) and 64 KiB ones cost 1-5 %
    uint32_t tail_strips = AGH_FU_TAIL_KB_DEFAULT;
    uint64_t tail_total = (uint64_t)AGH_FU_TAIL_MB_DEFAULT << 10;            // in strips (KiB)
    if (a.tune && a.tune->fused_tail_kb >= 0) {
        const unsigned long v = (unsigned long)a.tune->fused_tail_kb;
        tail_strips = (uint32_t)((v > 65536ul ? 65536ul : v) + 7ul) & ~7u;
    }
    if (a.tune && a.tune->fused_tail_mb >= 0) {
        const unsigned long v = (unsigned long)a.tune->fused_tail_mb;
        tail_total = (uint64_t)(v > (1ul << 20) ? (1ul << 20) : v) << 10;
    }
    if (tail_strips < 8u) tail_strips = 8u;
    if (tail_strips >= range_strips) tail_total = 0;                        // nothing smaller to hand out
    if (tail_total > n_full / 2) tail_total = n_full / 2;                   // short texts: half of it at most
    const uint64_t n_big64 = (n_full - tail_total) / range_strips;          // whole large tickets
    const uint64_t rest = n_full - n_big64 * range_strips;
    const uint64_t n_small = tail_total ? (rest + tail_strips - 1) / tail_strips
                                        : (rest + range_strips - 1) / range_strips;
    const uint32_t n_big = tail_total ? (uint32_t)n_big64 : (uint32_t)(n_big64 + n_small);
    if (!tail_total) tail_strips = range_strips;
    const uint32_t n_ranges = (uint32_t)(n_big64 + n_small);
    if (!n_ranges) return;
    // Shape of the persistent grid.  Sweeping waves per CU decide: H = 2 (eight probes per 16-byte chunk)
    // needs twelve -- eight leave the VALU short (10.2 -> 11.1-11.4 ms per 64 GiB) --, the lighter shapes
    // (H >= 4: k = 0, k = 1, config C3) want SIX TO EIGHT: more streams only disturb each other at the HBM
    // (k = 0: 8 sweeping waves 9.96 ms, 12 10.08-10.13, 16 10.11-10.15, 24 10.16-10.31).  And for the same
    // number of sweeping waves fewer, larger workgroups are better: 2 x 6 sweepers against 3 x 4 at k = 2
    // 10.14 vs 10.46 ms per 64 GiB and 1.360 vs 1.415 per 8 GiB in the same call, 1 x 8 against 2 x 4 at
    // k = 0 9.96 vs 10.14 and 1.310 vs 1.335 (profiles/r03_ab_sweepers_per_workgroup.log; fewer copies of
    // the filter table, fewer verifying waves polling).  History: round 2 launched four workgroups of 4 + 2
    // waves per CU, of which three were resident (4 x 38 KiB of LDS do not fit next to what the CU keeps for
    // itself) -- the fourth started when another one left and swept its statically assigned first ranges
    // alone at the end (r03_ab_headline_workgroups.log: 4 -> 3 per CU 1.407 -> 1.371 ms per 8 GiB).
    constexpr uint32_t WG_PER_CU = H == 2 ? 2u : 1u;
    uint32_t blocks = a.n_cu * WG_PER_CU;
    const uint32_t need = (n_ranges + (uint32_t)NS - 1u) / (uint32_t)NS;
    if (blocks > need) blocks = need;
    if (a.tune && a.tune->fused_blocks >= 0) {                               // (A/B runs)
        const unsigned long v = (unsigned long)a.tune->fused_blocks;
        blocks = (uint32_t)(v < 1ul ? 1ul : (v > (unsigned long)a.n_cu * 8ul ? (unsigned long)a.n_cu * 8ul : v));
    }
    hipLaunchKernelGGL((k_sweep_fused<WT, H, MODE, K, NCH, NV, NS>), dim3(blocks), dim3(64 * (NS + NV)), 0,
                       st, (const uint4 *)a.text, a.n, n_full, a.q, a.ftab, (const WT *)a.mask,
                       a.mk, a.gtab, tspan, n_ranges, a.ticket, range_strips, n_big, tail_strips);
}

template <typename WT, int H, int NCH>
static void launch_fused_m(const agh_fused_args &a, uint32_t tspan, hipStream_t st)
{
    switch ((a.q.fold ? 1 : 0) | (a.q.fq == 4 ? 2 : 0)) {
    case 0: launch_fused<WT, H, 4, NCH>(a, tspan, st); break;
    case 1: launch_fused<WT, H, 5, NCH>(a, tspan, st); break;
    case 2: launch_fused<WT, H, 6, NCH>(a, tspan, st); break;
    default: launch_fused<WT, H, 7, NCH>(a, tspan, st); break;
    }
}

template <typename WT, int NCH>
static void launch_fused_h(const agh_fused_args &a, int H, uint32_t tspan, hipStream_t st)
{
    switch (H) {
    case 2:                                     // H == 2 samples have four bytes (MODE bit 1)
        if (a.q.fold) launch_fused<WT, 2, 7, NCH>(a, tspan, st);
        else launch_fused<WT, 2, 6, NCH>(a, tspan, st);
        break;
    case 4: launch_fused_m<WT, 4, NCH>(a, tspan, st); break;
    case 8: launch_fused_m<WT, 8, NCH>(a, tspan, st); break;
    default: launch_fused_m<WT, 16, NCH>(a, tspan, st); break;
    }
}

// The instances of this object's k.  Returns false when the window does not fit one (the caller
// runs k_sweep + k_verify): windows of 2 or 3 16-byte pieces for 32-bit automata (m <= 32), 4 or 7
// for 64-bit ones -- the same geometry launch_verify_t picks for lean scans with a gram table.
#define AGH_FU_CAT2(a, b) a##b
#define AGH_FU_CAT(a, b) AGH_FU_CAT2(a, b)
bool AGH_FU_CAT(agh_launch_sweep_fused_k, AGH_FU_K)(const agh_fused_args &a, int H, hipStream_t st)
{
    const uint32_t tspan = (uint32_t)(a.q.m + 2 * a.q.k + 1) + a.gram_spread;
    const int tn = (int)((tspan + 15u) / 16u);
    if (!a.wide) {
        if (tn <= 2) launch_fused_h<uint32_t, 2>(a, H, tspan, st);
        else if (tn <= 3) launch_fused_h<uint32_t, 3>(a, H, tspan, st);
        else return false;
    } else {
        if (tn <= 4) launch_fused_h<uint64_t, 4>(a, H, tspan, st);
        else if (tn <= 7) launch_fused_h<uint64_t, 7>(a, H, tspan, st);
        else return false;
    }
    return true;
}

#if defined(AGH_FU_TRACE) && AGH_FU_K == 2
extern "C" int agh_debug_fused_trace(uint64_t *out)      // 4 x 8192 stamps of the last k = 2 launch
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fu_trace), sizeof(uint64_t) * 4 * 8192) == hipSuccess ? 0 : -1;
}
#endif

#if AGH_FU_K == 0
bool agh_launch_sweep_fused_k1(const agh_fused_args &a, int H, hipStream_t st);
bool agh_launch_sweep_fused_k2(const agh_fused_args &a, int H, hipStream_t st);
bool agh_launch_sweep_fused_k3(const agh_fused_args &a, int H, hipStream_t st);

bool agh_launch_sweep_fused(const agh_fused_args &a, int H, hipStream_t st)
{
    if (a.q.mb || !a.gtab || (H != 2 && H != 4 && H != 8 && H != 16)) return false;
    switch (a.q.k) {
    case 0: return agh_launch_sweep_fused_k0(a, H, st);
    case 1: return agh_launch_sweep_fused_k1(a, H, st);
    case 2: return agh_launch_sweep_fused_k2(a, H, st);
    case 3: return agh_launch_sweep_fused_k3(a, H, st);
    default: return false;
    }
}
#endif
