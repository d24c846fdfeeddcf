// micro-benchmark: cycles per ds_read_b32 at byte offsets 0..3 (gfx950 unaligned LDS access), scattered rows
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32_a1 __attribute__((aligned(1)));
__global__ void k(uint32_t *out, uint64_t *cyc, int off, int stride)
{
    __shared__ __attribute__((aligned(16))) unsigned char sh[65536];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<uint32_t *>(sh)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a = (threadIdx.x * stride + off) & 0xffff, acc = 0;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 4096; ++it) {
        uint32_t v = *reinterpret_cast<const u32_a1 *>(sh + (a & 0xfffc) + off);
        acc += v;
        a = (a + (v & 0x3c) + 64) & 0xffff;      // dependent chain: latency; & 0x3c keeps the base 4-aligned
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void kt(uint32_t *out, uint64_t *cyc, int off, int stride)
{
    // throughput: 8 independent reads per iteration
    __shared__ __attribute__((aligned(16))) unsigned char sh[65536];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<uint32_t *>(sh)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a = (threadIdx.x * stride) & 0xfffc, acc = 0;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 1024; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += *reinterpret_cast<const u32_a1 *>(sh + ((a + u * 772) & 0xfffc) + off);
        a = (a + 36) & 0xfffc;
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    uint32_t *out; uint64_t *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
    for (int stride = 4; stride <= 52; stride += 48)
        for (int off = 0; off < 4; ++off) {
            uint64_t h[2];
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, off, stride);
            hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(kt, dim3(1), dim3(256), 0, 0, out, cyc, off, stride);
            hipMemcpy(h + 1, cyc, 8, hipMemcpyDeviceToHost);
            printf("lds ds_read_b32 byte offset %d, lane stride %d B: latency chain %.1f cycles/read, 4 waves x 8 reads: %.1f cycles per wave-read\n",
                   off, stride, h[0] / 4096.0, h[1] / (1024.0 * 8));
        }
    return 0;
}
