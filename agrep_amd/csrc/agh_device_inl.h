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

