// agh_sweep_inl.h -- the per-chunk pieces of the sweep (sample probes, candidate queue) shared by
// k_sweep (agh_sweep.hip) and the fused sweep + verify kernel (agh_fused.hip).
#pragma once
#include "agh_device_inl.h"

// ---------------------------------------------------------------------------------------
// sweep: delimiter census + q-gram sample filter
// ---------------------------------------------------------------------------------------
// MODE bit 0: the query folds ASCII case (OR 0x20 into every sampled byte);
// MODE bit 1: 4-byte samples (no mask needed, 32-bit hash) instead of <= 3-byte samples;
// MODE bit 2: lean sweep -- no delimiter census (count-only scans identify a record by the
//             offset of its first byte, found by the verifier, instead of by its number);
// MODE bit 3: multi-byte delimiter -- the census reads the delimiter-end bitmap.
template <int MODE>
__device__ __forceinline__ uint32_t probe(uint32_t w, const agh_dev_query &q,
                                          const uint8_t *ftab)
{
    if (MODE & 2) {
        const uint32_t s = (MODE & 1) ? (w | q.fold) : w;
        const uint32_t p = agh_sample_prod_q4(s);
        return ((uint32_t)ftab[AGH_Q4_SLOT(p)] >> AGH_Q4_BIT(p)) & 1u;
    } else {
        const uint32_t s = (MODE & 1) ? ((w & q.qmask) | q.fold) : (w & q.qmask);
        const uint32_t p = agh_sample_prod_q3(s);
        return ((uint32_t)ftab[AGH_Q3_SLOT(p)] >> AGH_Q3_BIT(p)) & 1u;
    }
}

// H == 2: a 4-byte sample at every even byte offset (overlapping samples; the lemma then needs
// floor((m-k-q+1)/h) >= 2k+1, choose_filter).  Eight probes per 16-byte chunk, so every instruction
// counts: one v_dot2_u32_u16 hashes the sample, the table is read as the aligned dword that holds
// the bit, v_lshrrev takes the bit number from the low five bits of the product as they are, and
// v_alignbit pushes bit 0 of the result into the hit word from the top.  After the 32 pushes of a
// supertile (4 chunks, in order) probe i of chunk u sits at bit 8u + i.
__device__ __forceinline__ void probe_push_q4(uint32_t g, const uint8_t *ftab, uint32_t &hits)
{
    const uint32_t p = agh_sample_prod_q4(g);
    const uint32_t val = *reinterpret_cast<const uint32_t *>(ftab + ((p >> 3) & (AGH_FT_SIZE - 4u)));
    hits = __builtin_amdgcn_alignbit(val >> (p & 31u), hits, 1);
}

// nx: the dword that follows the chunk in the text (the sample at byte 14 reaches into it)
template <int MODE>
__device__ __forceinline__ void sweep_chunk_h2(uint4 v, uint32_t nx, const agh_dev_query &q,
                                               const uint8_t *ftab, uint32_t &hits)
{
    uint32_t w[5] = {v.x, v.y, v.z, v.w, nx};
    if (MODE & 1) {
#pragma unroll
        for (int d = 0; d < 5; ++d) w[d] |= q.fold;      // (q == 4: fold is 0x20 in all four bytes)
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        probe_push_q4(w[d], ftab, hits);
        probe_push_q4(__builtin_amdgcn_alignbyte(w[d + 1], w[d], 2), ftab, hits);
    }
}

// first dword of the following lane's chunk (DPP wave_shl:1); lane 63 takes `wrap` (uniform)
__device__ __forceinline__ uint32_t next_lane_dword(uint32_t x, uint32_t wrap)
{
    const uint32_t sh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, false);   // wave_shl:1
    return lane_id() == 63 ? wrap : sh;
}

// One 16-byte chunk: accumulate the non-delimiter popcount and the sample hit bits.
template <int H, int MODE>
__device__ __forceinline__ void sweep_chunk(uint4 v, uint32_t dd, const agh_dev_query &q,
                                            const uint8_t *ftab, uint32_t &acc,
                                            uint32_t &hits, int bitbase, uint32_t dbits16 = 0,
                                            uint32_t nx = 0)
{
    if (MODE & 8) {
        // multi-byte delimiter: the chunk's 16 delimiter-end bits come from the bitmap; keep
        // the "128 minus delimiters" convention of the SWAR census
        if (!(MODE & 4)) acc += 128u - (uint32_t)__popc(dbits16 & 0xffffu);
    } else if (!(MODE & 4)) {
        acc += nz_popc(v.x, dd) + nz_popc(v.y, dd) + nz_popc(v.z, dd) + nz_popc(v.w, dd);
    }
    if (H == 2) sweep_chunk_h2<MODE>(v, nx, q, ftab, hits);     // (pushes from the top: bitbase unused)
    if (H > 2) {
        hits |= probe<MODE>(v.x, q, ftab) << bitbase;
        if (H <= 8) hits |= probe<MODE>(v.z, q, ftab) << (bitbase + 2);
        if (H <= 4) {
            hits |= probe<MODE>(v.y, q, ftab) << (bitbase + 1);
            hits |= probe<MODE>(v.w, q, ftab) << (bitbase + 3);
        }
    }
}

// Candidates of one wave.  They are queued in LDS (cq, private to the wave) and written to the
// wave's private slice of the candidate buffer 64 at a time with one coalesced store:
//   * no atomics -- one hot global counter saturates at ~90 updates/us on this chip and would
//     cap the whole sweep;
//   * no global store per hit -- vmcnt also counts stores, so a store issued between the
//     prefetch and its use stalls the wave until that store has completed (measured: ~7 % of
//     the sweep).
// hits: bit (4*u + d) of lane l = sample at dword d of the lane's chunk in strip s+u.
// rc[u] = delimiters (inside this wave's range) in front of the lane's chunk of strip s+u --
// stored with the candidate so that the verifier can number records without re-reading text.
// qn (queued) and cnt (already in the slice) are wave-uniform.
// H == 2: hit bit 8u + i = the sample at byte 2i of the lane's chunk in strip s+u, and the entry
// is a HALFWORD index (j = entry * 2).
template <int H, typename OnFull>
__device__ __forceinline__ void emit_candidates_to(uint32_t hits, uint64_t s,
                                                   const uint32_t rc[4], uint64_t *cq,
                                                   uint32_t &qn, OnFull on_full)
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
            int u = H == 2 ? b >> 3 : b >> 2;
            // dword index of the sample: 40 bits inside a numbered segment (<= 16 GiB), the record count in
            // the 24 bits above (agh_device.h); lean sweeps (r == 0) use all 64 bits, so one launch can
            // cover any text length
            const uint64_t dw = H == 2 ? ((s + (uint64_t)u) * 64u + (uint64_t)l) * 8u + (uint64_t)(b & 7)
                                       : ((s + (uint64_t)u) * 64u + (uint64_t)l) * 4u + (uint64_t)(b & 3);
            uint32_t r = u == 0 ? r0 : (u == 1 ? r1 : (u == 2 ? r2 : r3));
            cq[qn + (uint32_t)lane] = ((uint64_t)r << AGH_CAND_IDX_BITS) | dw;
        }
        qn += (uint32_t)c;
        if (qn >= 64u) on_full();
    }
}

template <int H>
__device__ __forceinline__ void emit_candidates(uint32_t hits, uint64_t s, const uint32_t rc[4],
                                                uint64_t *cq, uint32_t &qn,
                                                uint64_t *__restrict__ slice, uint32_t &cnt,
                                                uint32_t *counters)
{
    emit_candidates_to<H>(hits, s, rc, cq, qn,
                          [&]() { flush_candidates(cq, qn, 64u, slice, cnt, counters); });
}

