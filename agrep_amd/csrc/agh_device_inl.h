// agh_device_inl.h -- device-side helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "agh_device.h"
#include "agh_launch.h"

#define WAVE 64

// ---------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (WAVE - 1)); }

// Streaming read of 16 text bytes: non-temporal (global_load_dwordx4 ... nt).  The sweeps read
// every byte exactly once; without the hint the same loop measures 6.1 TB/s, with it 6.9 TB/s
// on MI355X (scripts/exp_variants.py).
__device__ __forceinline__ uint4 ld_stream(const uint4 *p)
{
    typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
    const u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt *>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}

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


// ---- multi-byte delimiters: bit i of the delimiter bitmap <=> a (selected, i.e. leftmost
// non-overlapping: asearch.c:54-57 D_Mask) delimiter occurrence ENDS at text byte i ----------
__device__ __forceinline__ uint32_t dbm_bit(const uint64_t *__restrict__ dbm, uint64_t pos)
{
    return (uint32_t)(dbm[pos >> 6] >> (pos & 63)) & 1u;
}

// 64 bitmap bits starting at bit position pos (any alignment)
__device__ __forceinline__ uint64_t dbm_bits64(const uint64_t *__restrict__ dbm, uint64_t pos)
{
    const uint64_t w = pos >> 6;
    const uint32_t sh = (uint32_t)(pos & 63);
    const uint64_t lo = dbm[w];
    if (sh == 0) return lo;
    return (lo >> sh) | (dbm[w + 1] << (64 - sh));
}

// positions j of the delimiter whose byte equals c (bit j)
__device__ __forceinline__ uint32_t delim_class(const agh_dev_query &q, uint32_t c)
{
    uint32_t m = 0;
    // -i: maskgen.c:259-266 aliases the upper-case rows of Mask[] for the delimiter positions
    // as well, so "FROM " ends a record of -d 'From ' (dbytes are lower-cased by the host then)
    if (q.dfold && c >= 'A' && c <= 'Z') c += 32u;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j)
        if (j < q.dlen && q.dbytes[j] == c) m |= 1u << j;
    return m;
}

// number of set bitmap bits at positions [a, b)
__device__ __forceinline__ uint32_t dbm_count(const uint64_t *__restrict__ dbm, uint64_t a,
                                              uint64_t b)
{
    uint32_t c = 0;
    while (a < b) {
        const uint64_t w = dbm[a >> 6] >> (a & 63);
        const uint64_t take = 64 - (a & 63) < b - a ? 64 - (a & 63) : b - a;
        c += (uint32_t)__popcll(take == 64 ? w : (w & ((1ull << take) - 1ull)));
        a += take;
    }
    return c;
}

// position of the last set bit below pos, looking back at most cap bits; -1: none down to
// bit 0; -2: none within cap
__device__ __forceinline__ int64_t dbm_prev(const uint64_t *__restrict__ dbm, uint64_t pos,
                                            uint64_t cap)
{
    const uint64_t stop = pos > cap ? pos - cap : 0;
    while (pos > stop) {
        const uint64_t wi = (pos - 1) >> 6;
        uint64_t w = dbm[wi];
        const uint32_t top = (uint32_t)((pos - 1) & 63);        // highest bit still in range
        if (top < 63) w &= (1ull << (top + 1)) - 1ull;
        if (w) {
            const uint64_t at = (wi << 6) + 63 - (uint64_t)__clzll((long long)w);
            return at >= stop ? (int64_t)at : (stop ? -2 : -1);
        }
        pos = wi << 6;
    }
    return stop ? -2 : -1;
}

// ---- per-wave LDS candidate queue, flushed 64 entries at a time with one coalesced store ---
#define AGH_CQ_LEN 96

template <uint32_t CAP = AGH_SLICE_CAP>
__device__ __forceinline__ void flush_candidates(uint64_t *cq, uint32_t &qn, uint32_t take,
                                                 uint64_t *__restrict__ slice, uint32_t &cnt,
                                                 uint32_t *counters)
{
    const uint32_t lane = (uint32_t)lane_id();
    if (lane < take) {
        const uint32_t idx = cnt + lane;
        if (idx < CAP) slice[idx] = cq[lane];
        else counters[AGH_C_OVERFLOW] = 1u;
    }
    cnt += take;
    const uint32_t rest = qn - take;            // < 32: move it to the front
    uint64_t keep = 0;
    if (lane < rest) keep = cq[take + lane];
    if (lane < rest) cq[lane] = keep;
    qn = rest;
}

