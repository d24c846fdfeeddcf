// agh_multi.hip -- multi-pattern scans (-f patternfile; the role of newmgrep.c: prepf() tables +
// monkey1() hash-and-compare, newmgrep.c:192-375, 839-1012), exact and -- beyond the reference,
// which ignores -# together with -f (compat.c:34-37) -- with k errors (BASELINE config 5).
//
// A record matches iff it contains any pattern verbatim (k = 0) or a substring within edit distance
// k of any pattern.  Same shape as the single-pattern scan with the filter turned around: instead of
// a few aligned text samples against all q-grams of one pattern, every S-th text position's q-byte
// gram (q = min(4, shortest entry); S = 1, 2 or 4 by the shortest entry, fill_multi_tables) is
// probed against the grams of the table ENTRIES -- whole patterns, or the k+1 disjoint pieces of
// every pattern (an occurrence with <= k errors contains one of them verbatim) -- in a 2^18-bit
// table held in LDS.
//
//   k_sweep_multi     streams the text like k_sweep.  Per probe: v_alignbyte (unaligned gram),
//                     v_dot2_u32_u16 (hash, agh_sample_prod_q4), the aligned dword of the bit table,
//                     v_lshrrev by the low five bits of the product, v_alignbit to push the bit
//                     into the hit word -- five VALU operations.  First-level hits take a second,
//                     independent Bloom probe (q = 4); what survives is queued in LDS, every lane
//                     writing its own hits (one per round), and goes to the wave's private slice.
//   k_verify_multi    one lane per candidate, one workgroup per slice: the bucket of entries with
//                     that gram is walked, an entry that occurs verbatim is a match (k = 0) or sends
//                     its pattern's k-error automaton over the window the occurrence can cover;
//                     record bookkeeping only after a hit (delimiter / hit masks as in verify_walk).
//   k_dense_multi     sets whose hits do not fit the slices (hundreds of 1..3-byte entries; -f with
//                     errors over 4-byte patterns): the same probes, but the queue is verified on
//                     the spot, 64 candidates at a time, one per lane -- nothing can overflow.
#include <string.h>
#include "agh_multi_inl.h"

// ---------------------------------------------------------------------------------------
// the sweep
// ---------------------------------------------------------------------------------------
// One wave per 256 KiB range, 4 KiB supertiles, next supertile prefetched -- as k_sweep.
// MODE: bit 0 fold case, bit 1 q == 4, bit 2 lean (no census).
// Hits go to the wave's private slice of the candidate buffer; k_verify_multi reads them back.
// Measured and dropped (round 3; profiles/r03_perf_multi_verifier_waves*.log, r03_pmc_sweep_multi*_fused.json):
// verifying a full queue inside the sweeping wave (1.78 vs 1.44 ms per 4 GiB, exact 4..12 B; k = 1: 2.60
// vs 1.87), two verifying waves per workgroup fed through an LDS ring as in k_sweep_fused (1.55 vs 1.45;
// k = 1: 1.65 vs 1.65; 8..12 B exact: 1.09 vs 0.93) and the verifier of one part of the text on a second
// stream under the sweep of the next (1.52 / 1.84 / 2.84 ms in 2 / 4 / 8 parts) -- the sweep is bound by
// VALU issue and LDS reads, and whatever shares its SIMDs costs it what it would have cost alone.
template <int MODE, int STRIDE, bool Q5>
__global__ __launch_bounds__(256) void k_sweep_multi(const uint4 *__restrict__ text, uint64_t n,
                                                     uint64_t n_full_strips, agh_dev_query q,
                                                     const uint32_t *__restrict__ bits_g,
                                                     uint32_t *__restrict__ wave_totals,
                                                     uint64_t *__restrict__ cand,
                                                     uint32_t *__restrict__ wave_cand,
                                                     uint32_t *__restrict__ counters,
                                                     const uint16_t *__restrict__ dbm16, uint32_t w_base)
{
    // this launch sweeps the wave ranges w_base .. up to strip n_full_strips (a part of the text)
    __shared__ __attribute__((aligned(16))) uint32_t tab[AGH_MP_WORDS];
    __shared__ uint64_t cq_all[4 * AGH_MP_CQ_LEN];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(bits_g);
        uint4 *dst = reinterpret_cast<uint4 *>(tab);
        constexpr int PER = AGH_MP_WORDS / 4 / 256;
        uint4 tmp[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) tmp[i] = src[threadIdx.x + i * 256];
#pragma unroll
        for (int i = 0; i < PER; ++i) dst[threadIdx.x + i * 256] = tmp[i];
        __syncthreads();
    }
    const uint8_t *tab8 = reinterpret_cast<const uint8_t *>(tab);
    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const uint64_t w = (uint64_t)w_base + (uint64_t)blockIdx.x * 4 + wib;
    const uint64_t s0 = w * AGH_WAVE_STRIPS;
    if (s0 >= n_full_strips) return;
    uint64_t s1 = s0 + AGH_WAVE_STRIPS;
    if (s1 > n_full_strips) s1 = n_full_strips;
    const uint32_t dd = q.delim * 0x01010101u;
    const uint32_t fold4 = (MODE & 1) ? 0x20202020u : 0u;
    const uint64_t n_dw = ((n + 15) & ~(uint64_t)15) / 4;      // readable dwords
    uint32_t run = 0, ncand = 0, qn = 0;
    uint64_t *cq = cq_all + wib * AGH_MP_CQ_LEN;
    uint64_t *slice = cand + w * AGH_MP_SLICE_CAP;
    auto flush64 = [&]() { flush_candidates<AGH_MP_SLICE_CAP>(cq, qn, 64u, slice, ncand, counters); };
    // the dword right behind strip st-1 (uniform; 0 past the readable text)
    auto first_dword_of = [&](uint64_t st) -> uint32_t {
        const uint64_t i = st * 256u;
        return i < n_dw ? reinterpret_cast<const uint32_t *>(text)[i] : 0u;
    };
    constexpr uint32_t NB = 16u / STRIDE;
    // one supertile: strips s .. s+3 (v0..v3), nx3 = the first dword of strip s+4
    auto supertile = [&](uint4 v0, uint4 v1, uint4 v2, uint4 v3, uint64_t s, uint32_t nx3) {
        uint32_t w0[5] = {v0.x | fold4, v0.y | fold4, v0.z | fold4, v0.w | fold4,
                          next_lane_dword(v0.x, (uint32_t)__builtin_amdgcn_readlane((int)v1.x, 0)) | fold4};
        uint32_t w1[5] = {v1.x | fold4, v1.y | fold4, v1.z | fold4, v1.w | fold4,
                          next_lane_dword(v1.x, (uint32_t)__builtin_amdgcn_readlane((int)v2.x, 0)) | fold4};
        uint32_t w2[5] = {v2.x | fold4, v2.y | fold4, v2.z | fold4, v2.w | fold4,
                          next_lane_dword(v2.x, (uint32_t)__builtin_amdgcn_readlane((int)v3.x, 0)) | fold4};
        uint32_t w3[5] = {v3.x | fold4, v3.y | fold4, v3.z | fold4, v3.w | fold4,
                          next_lane_dword(v3.x, nx3) | fold4};
        // census (numbered scans): delimiters in front of the lane's chunk of every strip.  In front
        // of the probes: behind them its bitmap branch splits the block, and the compiler parks all 64
        // table reads of a supertile in registers across it (181 VGPRs, two waves per SIMD)
        uint32_t rc[4] = {0u, 0u, 0u, 0u};
        if (!(MODE & 4)) {
            // ("128 minus the delimiters of the lane's chunk"; bitmap delimiters: 16 bits per chunk)
            uint32_t a0, a1, a2, a3;
            if (q.mb) {
                const uint16_t *d = dbm16 + s * 64 + lane;
                a0 = 128u - (uint32_t)__popc((uint32_t)d[0]);
                a1 = 128u - (uint32_t)__popc((uint32_t)d[64]);
                a2 = 128u - (uint32_t)__popc((uint32_t)d[128]);
                a3 = 128u - (uint32_t)__popc((uint32_t)d[192]);
            } else {
                a0 = nz_popc(v0.x, dd) + nz_popc(v0.y, dd) + nz_popc(v0.z, dd) + nz_popc(v0.w, dd);
                a1 = nz_popc(v1.x, dd) + nz_popc(v1.y, dd) + nz_popc(v1.z, dd) + nz_popc(v1.w, dd);
                a2 = nz_popc(v2.x, dd) + nz_popc(v2.y, dd) + nz_popc(v2.z, dd) + nz_popc(v2.w, dd);
                a3 = nz_popc(v3.x, dd) + nz_popc(v3.y, dd) + nz_popc(v3.z, dd) + nz_popc(v3.w, dd);
            }
            const uint32_t own01 = a0 | (a1 << 16), own23 = a2 | (a3 << 16);
            const uint32_t sc01 = wave_sum_to_lane63(own01), sc23 = wave_sum_to_lane63(own23);
            const uint32_t p01 = (uint32_t)__builtin_amdgcn_readlane((int)sc01, 63);
            const uint32_t p23 = (uint32_t)__builtin_amdgcn_readlane((int)sc23, 63);
            const uint32_t z0 = 8192u - (p01 & 0xffffu), z1 = 8192u - (p01 >> 16);
            const uint32_t z2 = 8192u - (p23 & 0xffffu), z3 = 8192u - (p23 >> 16);
            const uint32_t ex01 = sc01 - own01, ex23 = sc23 - own23, lb = 128u * (uint32_t)lane;
            rc[0] = run + lb - (ex01 & 0xffffu);
            rc[1] = run + z0 + lb - (ex01 >> 16);
            rc[2] = run + z0 + z1 + lb - (ex23 & 0xffffu);
            rc[3] = run + z0 + z1 + z2 + lb - (ex23 >> 16);
            run += z0 + z1 + z2 + z3;
        }
        uint32_t lo = 0, hi = 0;
        probe_chunk_l1<MODE, STRIDE, Q5>(w0, q, tab8, lo);
        probe_chunk_l1<MODE, STRIDE, Q5>(w1, q, tab8, lo);
        if (STRIDE == 1) {
            probe_chunk_l1<MODE, STRIDE, Q5>(w2, q, tab8, hi);
            probe_chunk_l1<MODE, STRIDE, Q5>(w3, q, tab8, hi);
        } else {
            probe_chunk_l1<MODE, STRIDE, Q5>(w2, q, tab8, lo);
            probe_chunk_l1<MODE, STRIDE, Q5>(w3, q, tab8, lo);
            if (STRIDE == 4) lo >>= 16;         // 16 pushes only
        }
        if (!__ballot((lo | hi) != 0u)) return;
        // second Bloom probe, strip by strip: the gram of a run-time position wants an indexed read,
        // which registers do not have, and a loop per strip keeps the indices static.  (Measured: the
        // supertile parked in LDS and ONE loop over all its first-level hits -- 5 rounds instead of 8 --
        // is within the noise of this, 3 % either way by set; it is not where the time goes: 368 of the
        // 396 VALU instructions of a stride-1 supertile are the 64 first-level probes.)
        if (MODE & 2) {
            constexpr uint32_t M = NB == 16 ? 0xffffu : (NB == 8 ? 0xffu : 0xfu);
            const uint32_t h0 = lo & M, h1 = (lo >> NB) & M;
            const uint32_t h2 = STRIDE == 1 ? (hi & M) : ((lo >> (2 * NB)) & M);
            const uint32_t h3 = STRIDE == 1 ? ((hi >> NB) & M) : ((lo >> (3 * NB)) & M);
            uint32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0;
            if (__ballot(h0 != 0u)) k0 = probe_chunk_l2<STRIDE, Q5>(h0, w0[0], w0[1], w0[2], w0[3], w0[4], tab8);
            if (__ballot(h1 != 0u)) k1 = probe_chunk_l2<STRIDE, Q5>(h1, w1[0], w1[1], w1[2], w1[3], w1[4], tab8);
            if (__ballot(h2 != 0u)) k2 = probe_chunk_l2<STRIDE, Q5>(h2, w2[0], w2[1], w2[2], w2[3], w2[4], tab8);
            if (__ballot(h3 != 0u)) k3 = probe_chunk_l2<STRIDE, Q5>(h3, w3[0], w3[1], w3[2], w3[3], w3[4], tab8);
            if (STRIDE == 1) { lo = k0 | (k1 << NB); hi = k2 | (k3 << NB); }
            else lo = k0 | (k1 << NB) | (k2 << (2 * NB)) | (k3 << (3 * NB));
        }
        emit_rounds<STRIDE>(lo, hi, s, rc, cq, qn, flush64);
    };
    // a single strip (fewer than four left in the range)
    auto single = [&](uint4 v, uint64_t st, uint32_t nx) {
        uint32_t w0[5] = {v.x | fold4, v.y | fold4, v.z | fold4, v.w | fold4, next_lane_dword(v.x, nx) | fold4};
        uint32_t lo = 0;
        probe_chunk_l1<MODE, STRIDE, Q5>(w0, q, tab8, lo);
        lo >>= 32u - NB;
        uint32_t rc[4] = {0u, 0u, 0u, 0u};
        if (!(MODE & 4)) {
            const uint32_t a0 = q.mb ? 128u - (uint32_t)__popc((uint32_t)dbm16[st * 64 + lane])
                                     : nz_popc(v.x, dd) + nz_popc(v.y, dd) + nz_popc(v.z, dd) + nz_popc(v.w, dd);
            const uint32_t sc = wave_sum_to_lane63(a0);
            rc[0] = run + 128u * (uint32_t)lane - (sc - a0);
            run += 8192u - (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
        }
        if (!__ballot(lo != 0u)) return;
        if (MODE & 2) lo = probe_chunk_l2<STRIDE, Q5>(lo, w0[0], w0[1], w0[2], w0[3], w0[4], tab8);
        emit_rounds<STRIDE>(lo, 0u, st, rc, cq, qn, flush64);
    };
    uint64_t s = s0;
    if (s + 4 <= s1) {
        const uint4 *p = text + s * 64 + lane;
        uint4 c0 = ld_stream(p), c1 = ld_stream(p + 64), c2 = ld_stream(p + 128), c3 = ld_stream(p + 192);
        for (; s + 8 <= s1; s += 4) {
            const uint4 *pn = text + (s + 4) * 64 + lane;
            uint4 n0 = ld_stream(pn), n1 = ld_stream(pn + 64), n2 = ld_stream(pn + 128), n3 = ld_stream(pn + 192);
            supertile(c0, c1, c2, c3, s, (uint32_t)__builtin_amdgcn_readlane((int)n0.x, 0));
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        }
        supertile(c0, c1, c2, c3, s, first_dword_of(s + 4));
        s += 4;
    }
    for (; s < s1; ++s) single(ld_stream(text + s * 64 + lane), s, first_dword_of(s + 1));
    if (qn) flush_candidates<AGH_MP_SLICE_CAP>(cq, qn, qn, slice, ncand, counters);
    if (lane == 0) {
        wave_totals[w] = run;
        wave_cand[w] = ncand < AGH_MP_SLICE_CAP ? ncand : AGH_MP_SLICE_CAP;
    }
}

// The last, partial strip: one wave, bytes >= n masked to a non-delimiter filler.
template <int MODE, int STRIDE, bool Q5>
__global__ __launch_bounds__(64) void k_sweep_multi_tail(const uint4 *__restrict__ text,
                                                         uint64_t n, agh_dev_query q,
                                                         const uint32_t *__restrict__ bits_g,
                                                         uint32_t *__restrict__ wave_totals,
                                                         uint64_t *__restrict__ cand,
                                                         uint32_t *__restrict__ wave_cand,
                                                         uint32_t *__restrict__ counters,
                                                         const uint32_t *__restrict__ strip_prefix,
                                                         const uint16_t *__restrict__ dbm16)
{
    // strip_prefix != NULL: a census pass already ran (dense numbered scans); wave_totals holds
    // the exclusive prefix per range and strip_prefix the per-strip offsets -- read, not written
    __shared__ uint64_t cq[AGH_MP_CQ_LEN];
    const int lane = lane_id();
    const uint64_t s = n >> AGH_STRIP_SHIFT;
    const uint64_t off = (s << AGH_STRIP_SHIFT) + (uint64_t)lane * 16u;
    const uint32_t dd = q.delim * 0x01010101u;
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    const uint32_t fold4 = (MODE & 1) ? 0x20202020u : 0u;
    uint4 v = make_uint4(fill4, fill4, fill4, fill4);
    if (off < n) {
        v = text[off >> 4];
        if (off + 16 > n) v = mask_tail(v, (int)(n - off), fill4);
    }
    uint32_t acc = 0;
    if (!(MODE & 4)) {
        if (q.mb) acc = 128u - (off < n ? (uint32_t)__popc((uint32_t)dbm16[off >> 4]) : 0u);
        else acc = nz_popc(v.x, dd) + nz_popc(v.y, dd) + nz_popc(v.z, dd) + nz_popc(v.w, dd);
    }
    uint32_t w0[5] = {v.x | fold4, v.y | fold4, v.z | fold4, v.w | fold4, next_lane_dword(v.x, fill4) | fold4};
    constexpr uint32_t NB = 16u / STRIDE;
    uint32_t hits = 0;
    probe_chunk_l1<MODE, STRIDE, Q5>(w0, q, reinterpret_cast<const uint8_t *>(bits_g), hits);   // table straight from global/L2
    hits >>= 32u - NB;
    // positions inside the text only
    if (off >= n) hits = 0;
    else if (off + 16 > n) hits &= (1u << ((uint32_t)(n - off + STRIDE - 1) / STRIDE)) - 1u;
    if ((MODE & 2) && hits)
        hits = probe_chunk_l2<STRIDE, Q5>(hits, w0[0], w0[1], w0[2], w0[3], w0[4], reinterpret_cast<const uint8_t *>(bits_g));
    const uint32_t sc = (MODE & 4) ? 0u : wave_sum_to_lane63(acc);
    const uint32_t z = (MODE & 4) ? 0u : 8192u - (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
    const uint64_t w = s / AGH_WAVE_STRIPS;
    const bool fresh = (s % AGH_WAVE_STRIPS) == 0;
    // delimiters of this wave's range in front of the strip (the verifier adds the prefix of
    // the ranges before it)
    uint32_t before = strip_prefix ? strip_prefix[s] : (fresh ? 0u : wave_totals[w]);
    before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
    uint32_t ncand = fresh ? 0u : wave_cand[w];
    ncand = (uint32_t)__builtin_amdgcn_readfirstlane((int)ncand);
    uint32_t qn = 0;
    uint64_t *slice = cand + w * AGH_MP_SLICE_CAP;
    uint32_t rc[4] = {0u, 0u, 0u, 0u};
    if (!(MODE & 4)) rc[0] = before + 128u * (uint32_t)lane - (sc - acc);
    emit_rounds<STRIDE>(hits, 0u, s, rc, cq, qn,
                        [&]() { flush_candidates<AGH_MP_SLICE_CAP>(cq, qn, 64u, slice, ncand, counters); });
    if (qn) flush_candidates<AGH_MP_SLICE_CAP>(cq, qn, qn, slice, ncand, counters);
    if (lane == 0) {
        wave_cand[w] = ncand < AGH_MP_SLICE_CAP ? ncand : AGH_MP_SLICE_CAP;
        if (!strip_prefix) wave_totals[w] = before + z;
    }
}

// ---------------------------------------------------------------------------------------
// verify: one workgroup per slice, one lane per candidate position
// ---------------------------------------------------------------------------------------
template <bool LEAN, int K>
__global__ __launch_bounds__(256) void k_verify_multi(const uint8_t *__restrict__ text,
                                                      uint64_t n, agh_dev_query q,
                                                      agh_multi_tables mt,
                                                      const uint64_t *__restrict__ cand,
                                                      const uint32_t *__restrict__ wave_cand,
                                                      const uint32_t *__restrict__ wave_prefix,
                                                      uint32_t w_begin, uint32_t nw, agh_marks mk)
{
    for (uint32_t w = w_begin + blockIdx.x; w < nw; w += gridDim.x) {
        const uint32_t cnt = wave_cand[w];
        const uint64_t *slice = cand + (uint64_t)w * AGH_MP_SLICE_CAP;
        const uint32_t wp = LEAN ? 0u : wave_prefix[w];
        for (uint32_t ci = threadIdx.x; ci < cnt; ci += 256u) {
            const uint64_t ent = slice[ci];
            const uint64_t j = ent & 0xffffffffull;
            if (j >= n) continue;
            mp_verify_at<LEAN, K>(text, n, q, mt, j, wp + (uint32_t)(ent >> 32), mk);
        }
    }
}

// ---------------------------------------------------------------------------------------
// dense hit sets: probe and verify in one kernel
// ---------------------------------------------------------------------------------------
// Every position is probed (whatever stride the tables were built for: a gram found at an offset
// the stride would have skipped still names real entries), hits are queued as in the sweep, and
// a full queue is verified on the spot -- one candidate per lane, so the lanes stay busy however
// the hits are spread over the text.  Run-time mode (fold, q, 5-byte grams): the probes are not
// what this kernel spends its time on.  Numbered scans: wave_totals / strip_prefix hold the
// exclusive prefixes of a census pass (k_sweep<0> + scan) that ran in front of this kernel.
// A wave takes AGH_DENSE_STRIPS strips (16 KiB), not a whole 256 KiB range: the kernel waits on
// the verifier's dependent loads, and 256 MiB in 256 KiB ranges are one wave per SIMD.
#define AGH_DENSE_STRIPS 16u
template <bool LEAN, int K>
__global__ __launch_bounds__(256) void k_dense_multi(const uint4 *__restrict__ text, uint64_t n,
                                                     uint64_t n_full_strips, agh_dev_query q,
                                                     agh_multi_tables mt,
                                                     const uint32_t *__restrict__ wave_totals,
                                                     const uint32_t *__restrict__ strip_prefix,
                                                     uint32_t *__restrict__ wave_cand, agh_marks mk)
{
    // The bit table stays in global memory (32 KiB: L2 / L1 resident): this kernel waits on the
    // verifier's dependent loads, not on probes, and without the table in LDS seven waves per SIMD are
    // resident instead of four
    __shared__ uint64_t cq_all[4 * AGH_MP_CQ_LEN];
    const uint8_t *tab8 = reinterpret_cast<const uint8_t *>(mt.bits);
    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const uint64_t s0 = ((uint64_t)blockIdx.x * 4 + wib) * AGH_DENSE_STRIPS;
    if (s0 >= n_full_strips) return;
    const uint64_t w = s0 / AGH_WAVE_STRIPS;    // the 256 KiB range the census numbers are kept for
    uint64_t s1 = s0 + AGH_DENSE_STRIPS;
    if (s1 > n_full_strips) s1 = n_full_strips;
    const uint8_t *text8 = reinterpret_cast<const uint8_t *>(text);
    const uint32_t dd = q.delim * 0x01010101u;
    const uint32_t fold4 = q.fold ? 0x20202020u : 0u;
    const bool q4 = q.fq == 4, q5 = q.mp_q5 != 0;
    const uint64_t n_dw = ((n + 15) & ~(uint64_t)15) / 4;
    uint32_t run = LEAN ? 0u : wave_totals[w] + strip_prefix[s0], qn = 0;
    uint64_t *cq = cq_all + wib * AGH_MP_CQ_LEN;
    // verify the first `take` queued candidates, one per lane; keep the rest
    auto verify_queue = [&](uint32_t take) {
        if ((uint32_t)lane < take) {
            const uint64_t ent = cq[lane];
            mp_verify_at<LEAN, K>(text8, n, q, mt, ent & 0xffffffffull, (uint32_t)(ent >> 32), mk);
        }
        const uint32_t rest = qn - take;
        uint64_t keep = 0;
        if ((uint32_t)lane < rest) keep = cq[take + (uint32_t)lane];
        if ((uint32_t)lane < rest) cq[lane] = keep;
        qn = rest;
    };
    for (uint64_t s = s0; s < s1; ++s) {
        const uint4 v = ld_stream(text + s * 64 + lane);
        const uint64_t i_nx = (s + 1) * 256u;
        const uint32_t wrap = i_nx < n_dw ? reinterpret_cast<const uint32_t *>(text)[i_nx] : 0u;
        const uint32_t w0[5] = {v.x | fold4, v.y | fold4, v.z | fold4, v.w | fold4, next_lane_dword(v.x, wrap) | fold4};
        uint32_t hits = 0;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int d = p >> 2, sh = p & 3;
            uint32_t g = sh ? __builtin_amdgcn_alignbyte(w0[d + 1], w0[d], sh) : w0[d];
            if (q5) g = agh_mix5(g, w0[d + 1] >> (8 * sh));     // the fifth byte (agh_mix5 takes the low one)
            const uint32_t idx = q4 ? agh_sample_prod_q4(g) : agh_sample_hash18_q3(g & q.qmask);
            hits = __builtin_amdgcn_alignbit(mp_bit(tab8, idx), hits, 1);
        }
        hits >>= 16;
        uint32_t rc[4] = {0u, 0u, 0u, 0u};
        if (!LEAN) {
            const uint32_t a0 = q.mb ? 128u - (uint32_t)__popc((uint32_t)reinterpret_cast<const uint16_t *>(mt.dbm)[s * 64 + lane])
                                     : nz_popc(v.x, dd) + nz_popc(v.y, dd) + nz_popc(v.z, dd) + nz_popc(v.w, dd);
            const uint32_t sc = wave_sum_to_lane63(a0);
            rc[0] = run + 128u * (uint32_t)lane - (sc - a0);
            run += 8192u - (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
        }
        if (!__ballot(hits != 0u)) continue;
        emit_rounds<1>(hits, 0u, s, rc, cq, qn, [&]() { verify_queue(64u); });
    }
    if (qn) verify_queue(qn);
    if (lane == 0 && s0 % AGH_WAVE_STRIPS == 0) wave_cand[w] = 0u;      // nothing went through the slices
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
template <int MODE, int STRIDE, bool Q5>
static void launch_sweep_multi_ms(const agh_sweep_args &a, hipStream_t st)
{
    // a part [w_begin, w_end) of the wave ranges (count-only scans: the verifier of one part runs
    // under the sweep of the next), or everything
    const uint64_t n_full_all = a.n >> AGH_STRIP_SHIFT;
    const bool to_end = a.w_end == 0 || (uint64_t)a.w_end * AGH_WAVE_STRIPS >= n_full_all;
    const uint64_t n_full = to_end ? n_full_all : (uint64_t)a.w_end * AGH_WAVE_STRIPS;
    const uint64_t w_hi = (n_full + AGH_WAVE_STRIPS - 1) / AGH_WAVE_STRIPS;
    const uint64_t n_waves = w_hi > a.w_begin ? w_hi - a.w_begin : 0;
    if (a.ev_begin) (void)hipEventRecord(a.ev_begin, st);
    if (n_waves && !a.tail_only)
        hipLaunchKernelGGL((k_sweep_multi<MODE, STRIDE, Q5>), dim3((uint32_t)((n_waves + 3) / 4)), dim3(256), 0, st,
                           (const uint4 *)a.text, a.n, n_full, a.q, (const uint32_t *)a.ftab, a.wave_totals,
                           a.cand, a.wave_cand, a.counters, (const uint16_t *)a.dbm, a.w_begin);
    if (a.ev_end) (void)hipEventRecord(a.ev_end, st);
    if (!to_end) return;                        // the partial last strip belongs to the last part
    if (a.n & (AGH_STRIP - 1))
        hipLaunchKernelGGL((k_sweep_multi_tail<MODE, STRIDE, Q5>), dim3(1), dim3(64), 0, st,
                           (const uint4 *)a.text, a.n, a.q, (const uint32_t *)a.ftab,
                           a.wave_totals, a.cand, a.wave_cand, a.counters,
                           (a.tail_only && !a.lean) ? (const uint32_t *)a.strip_prefix
                                                    : (const uint32_t *)nullptr,
                           (const uint16_t *)a.dbm);
}

// a.q.fh = the probe stride chosen by the host (fill_multi_tables): 1, 2 or 4; strides > 1 imply q == 4
template <int MODE>
static void launch_sweep_multi_m(const agh_sweep_args &a, hipStream_t st)
{
    if ((MODE & 2) && a.q.fh == 4 && a.q.mp_q5) launch_sweep_multi_ms<MODE, 4, true>(a, st);
    else if ((MODE & 2) && a.q.fh == 4) launch_sweep_multi_ms<MODE, 4, false>(a, st);
    else if ((MODE & 2) && a.q.fh == 2) launch_sweep_multi_ms<MODE, 2, false>(a, st);
    else launch_sweep_multi_ms<MODE, 1, false>(a, st);
}

// Multi-pattern sweep; a.ftab = the 2^18-bit table.  The prefix scan of the census (numbered scans)
// is launched by the caller through agh_launch_census_scan().  a.tail_only: only the partial last
// strip (the dense kernel handled the full strips).
void agh_launch_sweep_multi(const agh_sweep_args &a, hipStream_t st)
{
    const int mode = (a.q.fold ? 1 : 0) | (a.q.fq == 4 ? 2 : 0) | (a.lean ? 4 : 0);
    switch (mode) {
    case 0: launch_sweep_multi_m<0>(a, st); break;
    case 1: launch_sweep_multi_m<1>(a, st); break;
    case 2: launch_sweep_multi_m<2>(a, st); break;
    case 3: launch_sweep_multi_m<3>(a, st); break;
    case 4: launch_sweep_multi_m<4>(a, st); break;
    case 5: launch_sweep_multi_m<5>(a, st); break;
    case 6: launch_sweep_multi_m<6>(a, st); break;
    default: launch_sweep_multi_m<7>(a, st); break;
    }
}

#define AGH_K_SWITCH(MACRO)                                                                   \
    switch (a.q.k) {                                                                          \
        MACRO(0) MACRO(1) MACRO(2) MACRO(3) MACRO(4) MACRO(5) MACRO(6) MACRO(7) MACRO(8)      \
    default: break;                                                                           \
    }

// Dense hit sets: probe + verify of all full strips in one kernel (numbered scans: after a census
// pass); the partial last strip goes through agh_launch_sweep_multi(tail_only) + agh_launch_verify_multi.
void agh_launch_dense_multi(const agh_sweep_args &a, const agh_multi_dev &m, const agh_marks &mk,
                            hipStream_t st)
{
    const uint64_t n_full = a.n >> AGH_STRIP_SHIFT;
    const uint64_t n_waves = (n_full + AGH_DENSE_STRIPS - 1) / AGH_DENSE_STRIPS;
    if (!n_waves) return;
    const uint32_t blocks = (uint32_t)((n_waves + 3) / 4);
    const bool lean = a.lean != 0;
#define AGH_DM_CASE(KK)                                                                       \
    case KK:                                                                                  \
        if (lean)                                                                             \
            hipLaunchKernelGGL((k_dense_multi<true, KK>), dim3(blocks), dim3(256), 0, st,     \
                               (const uint4 *)a.text, a.n, n_full, a.q, m, a.wave_totals,     \
                               (const uint32_t *)a.strip_prefix, a.wave_cand, mk);            \
        else                                                                                  \
            hipLaunchKernelGGL((k_dense_multi<false, KK>), dim3(blocks), dim3(256), 0, st,    \
                               (const uint4 *)a.text, a.n, n_full, a.q, m, a.wave_totals,     \
                               (const uint32_t *)a.strip_prefix, a.wave_cand, mk);            \
        break;
    AGH_K_SWITCH(AGH_DM_CASE)
#undef AGH_DM_CASE
}

void agh_launch_verify_multi(const agh_scan_args &a, const agh_multi_dev &m, bool lean,
                             hipStream_t st)
{
    // slices [w_begin, w_end) of a part, else all
    const uint32_t w_hi = (a.w_end && a.w_end < a.nw) ? a.w_end : a.nw;
    if (w_hi <= a.w_begin) return;
    const uint32_t blocks = w_hi - a.w_begin > 65536u ? 65536u : w_hi - a.w_begin;
#define AGH_VM_CASE(KK)                                                                       \
    case KK:                                                                                  \
        if (lean)                                                                             \
            hipLaunchKernelGGL((k_verify_multi<true, KK>), dim3(blocks), dim3(256), 0, st,    \
                               (const uint8_t *)a.text, a.n, a.q, m, a.cand, a.wave_cand,     \
                               a.wave_prefix, a.w_begin, w_hi, a.mk);                         \
        else                                                                                  \
            hipLaunchKernelGGL((k_verify_multi<false, KK>), dim3(blocks), dim3(256), 0, st,   \
                               (const uint8_t *)a.text, a.n, a.q, m, a.cand, a.wave_cand,     \
                               a.wave_prefix, a.w_begin, w_hi, a.mk);                         \
        break;
    AGH_K_SWITCH(AGH_VM_CASE)
#undef AGH_VM_CASE
}
