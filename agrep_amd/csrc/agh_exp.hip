// agh_exp.hip -- timing experiments on the sweep's structure (diagnostics, not product path).
// Each variant reads the text exactly like k_sweep<4> (per-wave 256 KiB ranges, 4 KiB
// supertiles, prefetch) and differs in ONE aspect, so that within-process A/B rounds attribute
// the gap between the pure read probe and the real sweep.
#include <stdio.h>
#include <stdlib.h>

#include "agh_device_inl.h"

// EXP bits: 1 = hash VALU work, 2 = LDS table lookups, 4 = allocate the 32 KiB LDS table,
//           8 = census VALU work, 16 = prefetch, 32 = 16 KiB table (index masked),
//           64 = non-temporal loads (global_load ... nt)
//           128 = two lanes re-read a chained pair of 128-byte lines `lag` supertiles behind the
//                 wave's stream position (lag 255: far away, another wave's range) -- what a
//                 verifier trailing the sweep would fetch; tells whether L2 / MALL still hold it.
template <int EXP>
__global__ __launch_bounds__(256) void k_sweep_exp(const uint4 *__restrict__ text,
                                                   uint64_t n_full_strips, uint32_t qmask,
                                                   uint32_t *__restrict__ counters, uint32_t lag)
{
    __shared__ __attribute__((aligned(16))) uint8_t ftab[(EXP & 4) ? ((EXP & 32) ? 16384 : 32768) : 16];
    if (EXP & 4) {
        for (uint32_t i = threadIdx.x; i < sizeof(ftab) / 4; i += 256)
            reinterpret_cast<uint32_t *>(ftab)[i] = 0;
        __syncthreads();
    }
    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const uint64_t w = (uint64_t)blockIdx.x * 4 + wib;
    const uint64_t s0 = w * AGH_WAVE_STRIPS;
    if (s0 >= n_full_strips) return;
    uint64_t s1 = s0 + AGH_WAVE_STRIPS;
    if (s1 > n_full_strips) s1 = n_full_strips;
    uint32_t acc = 0, hits = 0;
    uint32_t pend_v = 0, pend_b = 0, pend_ln = 0;
    const uint32_t *pend_base = reinterpret_cast<const uint32_t *>(text);
    const uint32_t dd = 0x0a0a0a0au;
    auto chunk = [&](uint4 v) {
        const uint32_t dws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (EXP & 8) acc += nz_popc(dws[d], dd);
            if (EXP & 1) {
                uint32_t h = agh_sample_hash_q3(dws[d] & qmask);
                if (EXP & 32) h &= 16383u;
                if (EXP & 2) hits |= ftab[h];
                else hits |= h;
            } else {
                hits |= dws[d];
            }
        }
    };
    uint64_t s = s0;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto ld = [&](const uint4 *q) -> uint4 {
        if (EXP & 64) {
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(q));
            return make_uint4(v[0], v[1], v[2], v[3]);
        }
        return *q;
    };
    if (EXP & 16) {
        if (s + 4 <= s1) {
            const uint4 *p = text + s * 64 + lane;
            uint4 c0 = ld(p), c1 = ld(p + 64), c2 = ld(p + 128), c3 = ld(p + 192);
            for (; s + 8 <= s1; s += 4) {
                const uint4 *pn = text + (s + 4) * 64 + lane;
                uint4 n0 = ld(pn), n1 = ld(pn + 64), n2 = ld(pn + 128), n3 = ld(pn + 192);
                if ((EXP & 128) && lane < 2) {
                    // software-pipelined: the first line of the pair is requested one step before
                    // its value picks the second, so the chain's latency is not in the loop's path
                    // (a dedicated verifier wave would not stall the sweepers either).
                    uint64_t bs = (lag == 255) ? ((n_full_strips - 4 - s) & ~(uint64_t)3)
                                               : (s >= s0 + 4ull * lag ? s - 4ull * lag : s0);
                    const uint32_t *b = reinterpret_cast<const uint32_t *>(text + bs * 64);
                    const uint32_t ln = (uint32_t)((s >> 2) * 2654435761u + lane * 13u) >> 27;   // 0..31
                    acc += pend_b;
                    pend_b = pend_base[((pend_ln + (pend_v & 1) + 1) & 31) * 32 + 9];
                    pend_v = b[ln * 32 + 5];
                    pend_base = b;
                    pend_ln = ln;
                }
                chunk(c0); chunk(c1); chunk(c2); chunk(c3);
                c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            }
            chunk(c0); chunk(c1); chunk(c2); chunk(c3);
        }
    } else {
        for (; s + 4 <= s1; s += 4) {
            const uint4 *p = text + s * 64 + lane;
            uint4 c0 = ld(p), c1 = ld(p + 64), c2 = ld(p + 128), c3 = ld(p + 192);
            chunk(c0); chunk(c1); chunk(c2); chunk(c3);
        }
    }
    acc += pend_b + pend_v;
    if ((hits ^ acc) == 0x9e3779b9u) counters[AGH_C_CHECK] = hits;
}

template <int EXP>
static void launch_exp(const void *text, uint64_t n, uint32_t *counters, hipStream_t st, uint32_t lag = 0)
{
    const uint64_t n_full = n >> AGH_STRIP_SHIFT;
    const uint64_t n_waves = (n_full + AGH_WAVE_STRIPS - 1) / AGH_WAVE_STRIPS;
    hipLaunchKernelGGL(k_sweep_exp<EXP>, dim3((uint32_t)((n_waves + 3) / 4)), dim3(256), 0, st,
                       (const uint4 *)text, n_full, 0xffffffu, counters, lag);
}

void agh_launch_exp(int exp, const void *text, uint64_t n, uint32_t *counters, hipStream_t st)
{
    const uint32_t lag = (uint32_t)exp >> 8;
    exp &= 255;
    switch (exp) {
    case 0: launch_exp<0>(text, n, counters, st); break;
    case 16: launch_exp<16>(text, n, counters, st); break;
    case 4 + 16: launch_exp<4 + 16>(text, n, counters, st); break;
    case 1 + 16: launch_exp<1 + 16>(text, n, counters, st); break;
    case 1 + 4 + 16: launch_exp<1 + 4 + 16>(text, n, counters, st); break;
    case 1 + 2 + 4 + 16: launch_exp<1 + 2 + 4 + 16>(text, n, counters, st); break;
    case 1 + 2 + 4 + 16 + 32: launch_exp<1 + 2 + 4 + 16 + 32>(text, n, counters, st); break;
    case 1 + 2 + 4: launch_exp<1 + 2 + 4>(text, n, counters, st); break;
    case 8 + 16: launch_exp<8 + 16>(text, n, counters, st); break;
    case 1 + 2 + 4 + 8 + 16: launch_exp<1 + 2 + 4 + 8 + 16>(text, n, counters, st); break;
    case 64: launch_exp<64>(text, n, counters, st); break;
    case 64 + 16: launch_exp<64 + 16>(text, n, counters, st); break;
    case 64 + 1 + 2 + 4 + 16: launch_exp<64 + 1 + 2 + 4 + 16>(text, n, counters, st); break;
    case 128 + 64 + 1 + 2 + 4 + 16: launch_exp<128 + 64 + 1 + 2 + 4 + 16>(text, n, counters, st, lag); break;
    case 128 + 1 + 2 + 4 + 16: launch_exp<128 + 1 + 2 + 4 + 16>(text, n, counters, st, lag); break;
    default: fprintf(stderr, "agh_launch_exp: no variant %d\n", exp); break;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) fprintf(stderr, "agh_launch_exp(%d): %s\n", exp, hipGetErrorString(e));
}
