// agh_scan.hip -- the k-error automaton kernels (verify on candidate windows, full scan over
// every byte).  See agh_sweep.hip for the data flow of a scan.
#include "agh_device_inl.h"

// ---------------------------------------------------------------------------------------
// record bookkeeping shared by verify and fullscan
// ---------------------------------------------------------------------------------------
// Record r has a match whose last byte is at e: set its bit.  The number of matched records
// is the population count of the bitmap (k_bitmap_count) -- a shared "matched" counter
// would serialise on one L2 atomic unit (~90 updates/us) and dominate the scan.
__device__ __forceinline__ void mark_record(const agh_marks &mk, uint32_t r, uint64_t e)
{
    if (r >= mk.bitmap_bits) {                  // the host retries with a larger bitmap
        mk.counters[AGH_C_BM_OVERFLOW] = 1u;
        return;
    }
    const uint32_t bit = 1u << (r & 31u);
    uint32_t old = atomicOr(&mk.bitmap[r >> 5], bit);
    if (mk.match_pos && !(old & bit)) {
        uint32_t idx = atomicAdd(&mk.counters[AGH_C_STORED], 1u);
        if (idx < mk.match_cap) {
            mk.match_pos[idx] = e;
            if (mk.match_rec) mk.match_rec[idx] = r;
        }
    }
}

// ---- lean scans: a record is identified by the offset of its first byte ---------------------
__device__ __forceinline__ void lean_insert(const agh_marks &mk, uint64_t rec_start)
{
    const uint64_t key = rec_start + 1;         // 0 = empty slot
    uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & mk.hashset_mask;
    for (int probe = 0; probe < 64; ++probe) {
        const uint64_t old = atomicCAS((unsigned long long *)&mk.hashset[slot], 0ull,
                                       (unsigned long long)key);
        if (old == 0 || old == key) return;
        slot = (slot + 1) & mk.hashset_mask;
    }
    mk.counters[AGH_C_LEAN_FALLBACK] = 1u;      // table too full: the host re-runs numbered
}

// 1 + position of the last delimiter at a byte offset < pos (0 if there is none), looking
// back at most AGH_LEAN_BACK_CAP bytes; ~0 and the fallback flag if it is further away.
__device__ uint64_t lean_record_start(const uint8_t *__restrict__ text, uint64_t pos,
                                      uint32_t delim, const agh_marks &mk)
{
    typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
    const uint32_t dd = delim * 0x01010101u;
    const uint64_t stop = pos > AGH_LEAN_BACK_CAP ? pos - AGH_LEAN_BACK_CAP : 0;
    while (pos >= stop + 16) {
        const u32x4_a1 v = *reinterpret_cast<const u32x4_a1 *>(text + pos - 16);
#pragma unroll
        for (int d = 3; d >= 0; --d) {
            const uint32_t x = v[d] ^ dd;
            // bit 7 of every byte that equals the delimiter
            const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
            if (z) return pos - 16 + 4u * d + (uint32_t)((31 - __clz((int)z)) >> 3) + 1;
        }
        pos -= 16;
    }
    while (pos > stop) {
        if (text[pos - 1] == delim) return pos;
        --pos;
    }
    if (stop == 0) return 0;
    mk.counters[AGH_C_LEAN_FALLBACK] = 1u;
    return ~0ull;
}

// Number of delimiters at byte positions < e (= 0-based record number of position e).
__device__ uint32_t record_of(const uint8_t *__restrict__ text, uint64_t n, uint64_t e,
                              const uint32_t *__restrict__ strip_prefix,
                              const uint32_t *__restrict__ wave_prefix, uint32_t n_strips,
                              uint32_t total_delims, uint32_t delim)
{
    const uint64_t strip = e >> AGH_STRIP_SHIFT;
    if (strip >= n_strips) return total_delims;
    uint32_t r = wave_prefix[strip / AGH_WAVE_STRIPS] + strip_prefix[strip];
    const uint32_t dd = delim * 0x01010101u;
    const uint32_t fill4 = (~delim & 0xffu) * 0x01010101u;
    const uint64_t p = strip << AGH_STRIP_SHIFT;
    const uint4 *t4 = reinterpret_cast<const uint4 *>(text + p);
    const uint32_t span = (uint32_t)(e - p);
    const uint32_t full = span >> 4;
    for (uint32_t i = 0; i < full; ++i) r += delims_in(t4[i], dd);
    if (span & 15u) r += delims_in(mask_tail(t4[full], (int)(span & 15u), fill4), dd);
    return r;
}

// ---------------------------------------------------------------------------------------
// the k-error automaton (asearch.c:94-116 mirrored to left shifts, 1 = active)
//   R0' = ((R0 << 1) | 1) & CM
//   Re' = (((Re << 1) | 1) & CM) | R(e-1) | (((R(e-1) | R(e-1)') << 1) | 1)
// reset state = all zeros (asearch.c:175-186 resets to Init[0] = "nothing but the start
// state" and re-feeds the delimiter byte; the leading-deletion bits appear through the
// recurrence itself on that first step).
// ---------------------------------------------------------------------------------------
template <typename WT, int K>
struct Automaton {
    WT R[K + 1];
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int e = 0; e <= K; ++e) R[e] = 0;
    }
    __device__ __forceinline__ bool step(WT cm, WT finalbit)
    {
        WT po = R[0];
        WT pn = ((po << 1) | (WT)1) & cm;
        R[0] = pn;
#pragma unroll
        for (int e = 1; e <= K; ++e) {
            WT cur = R[e];
            WT ne = (((cur << 1) | (WT)1) & cm) | po | (((po | pn) << 1) | (WT)1);
            po = cur;
            pn = ne;
            R[e] = ne;
        }
        return (R[K] & finalbit) != 0;
    }
};

// ---------------------------------------------------------------------------------------
// verify: one workgroup per AGH_VGROUP sweep-wave slices, one lane per candidate sample
// ---------------------------------------------------------------------------------------
// Byte-wise reference walk of one window: used for windows at the head / tail of the text
// (virtual head byte, appended delimiter) where the register fast path does not apply.
template <typename WT, int K, bool LEAN>
__device__ __noinline__ void verify_window_slow(const uint8_t *__restrict__ text, uint64_t n,
                                                const agh_dev_query &q, const WT *lmask,
                                                uint64_t ws, uint64_t we, uint64_t anchor,
                                                uint32_t rc_anchor, const agh_marks &mk)
{
    const WT finalbit = (WT)1 << (q.m - 1);
    uint32_t rec = 0;
    uint64_t rstart = 0;                        // LEAN: first byte of the current record
    if (LEAN) {
        rstart = lean_record_start(text, ws, q.delim, mk);
        if (rstart == ~0ull) return;
    } else {
        // delimiters in [ws, anchor): the anchor's record number is known, ws's is derived
        uint32_t back = 0;
        for (uint64_t i = ws; i < anchor; ++i) back += (text[i] == q.delim);
        rec = rc_anchor - back;
    }
    Automaton<WT, K> A;
    A.reset();
    bool seen = false;
    if (ws == 0) A.step(lmask[q.head_byte], finalbit);
    for (uint64_t i = ws; i < we; ++i) {
        const uint32_t c = text[i];
        if (A.step(lmask[c], finalbit) && !seen) {
            seen = true;
            if (LEAN) lean_insert(mk, rstart); else mark_record(mk, rec, i);
        }
        if (c == q.delim) {
            A.reset();
            ++rec;
            rstart = i + 1;
            seen = false;
            if (A.step(lmask[c], finalbit)) {
                seen = true;
                if (LEAN) lean_insert(mk, rstart); else mark_record(mk, rec, i + 1);
            }
        }
    }
    if (we == n && q.tail_virtual) {
        if (A.step(lmask[q.delim], finalbit) && !seen) {
            if (LEAN) lean_insert(mk, rstart); else mark_record(mk, rec, n);
        }
        A.reset();
        if (A.step(lmask[q.delim], finalbit)) {
            if (LEAN) lean_insert(mk, n + 1); else mark_record(mk, rec + 1u, n);
        }
    }
}

// Unaligned 16-byte view of the text (gfx9+ global loads accept any byte address).
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));

#define AGH_VGROUP 8u   // sweep-wave slices verified by one workgroup

// Fast path geometry, identical for every lane: the window starts Lw = max(m+k+1, 16) bytes in
// front of the sample and spans Lw + q + m + k bytes; it is fetched with NCH unaligned 16-byte
// loads issued together and walked branch-free out of registers.  Match positions and
// delimiter positions are collected as bit masks; record numbers are derived from them after
// the walk.  Windows that touch the head or the tail of the text take the byte-wise path.
template <typename WT, int K, int NCH, bool LEAN>
__global__ __launch_bounds__(256) void k_verify(const uint8_t *__restrict__ text, uint64_t n,
                                                agh_dev_query q,
                                                const WT *__restrict__ mask_g,
                                                const uint64_t *__restrict__ cand,
                                                const uint32_t *__restrict__ wave_cand,
                                                const uint32_t *__restrict__ wave_prefix,
                                                uint32_t nw, agh_marks mk)
{
    constexpr int NMW = (NCH * 16 + 63) / 64;           // 64-bit words per position mask
    __shared__ WT lmask[256];
    __shared__ uint32_t pre[AGH_VGROUP + 1];
    lmask[threadIdx.x] = mask_g[threadIdx.x];
    const uint32_t g0 = blockIdx.x * AGH_VGROUP;        // first slice of this workgroup
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < AGH_VGROUP; ++i) {
            pre[i] = run;
            run += (g0 + i < nw) ? wave_cand[g0 + i] : 0u;
        }
        pre[AGH_VGROUP] = run;
    }
    __syncthreads();
    const uint32_t total = pre[AGH_VGROUP];
    if (total == 0) return;
    const WT finalbit = (WT)1 << (q.m - 1);
    const uint32_t Lw = (uint32_t)(q.m + q.k + 1) > 16u ? (uint32_t)(q.m + q.k + 1) : 16u;
    const uint32_t tailw = (uint32_t)(q.fq + q.m + q.k);
    const uint32_t span = Lw + tailw;                   // <= 16 * NCH by construction
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;

    // state right after a record boundary: reset + re-fed delimiter byte (asearch.c:175-186)
    Automaton<WT, K> RF;
    RF.reset();
    const bool rf_hit = RF.step(lmask[q.delim], finalbit);

    for (uint32_t ci = threadIdx.x; ci < total; ci += 256) {
        uint32_t sl = 0;
#pragma unroll
        for (uint32_t i = 1; i < AGH_VGROUP; ++i) sl += (pre[i] <= ci) ? 1u : 0u;
        const uint32_t w = g0 + sl;
        const uint64_t ent = cand[(uint64_t)w * AGH_SLICE_CAP + (ci - pre[sl])];
        const uint64_t j = (ent & 0xffffffffull) * 4u;
        if (j >= n) continue;
        const uint32_t rc_anchor = LEAN ? 0u : wave_prefix[w] + (uint32_t)(ent >> 32);  // record no. at anchor
        const uint64_t anchor = j & ~(uint64_t)15;                           // sample's chunk start

        const bool fast = j >= Lw && j + tailw < n && (j - Lw) + 16u * NCH <= n16;
        if (!fast) {
            const uint64_t ws = j > Lw ? j - Lw : 0;
            uint64_t we = j + tailw;
            if (we > n) we = n;
            verify_window_slow<WT, K, LEAN>(text, n, q, lmask, ws, we, anchor, rc_anchor, mk);
            continue;
        }
        const uint64_t ws = j - Lw;
        u32x4_u ch[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            ch[c] = *reinterpret_cast<const u32x4_u *>(text + ws + 16 * c);

        Automaton<WT, K> A;
        A.reset();
        uint32_t seen = 0;
        uint64_t hitm[NMW], hit2m[NMW], dm[NMW];
#pragma unroll
        for (int i = 0; i < NMW; ++i) hitm[i] = hit2m[i] = dm[i] = 0;
#pragma unroll
        for (int p = 0; p < NCH * 16; ++p) {
            if ((uint32_t)p >= span) break;             // wave-uniform
            const uint32_t dwv = ch[p >> 4][(p >> 2) & 3];
            const uint32_t byte = (dwv >> (8 * (p & 3))) & 0xffu;
            const uint32_t hit = A.step(lmask[byte], finalbit) ? 1u : 0u;
            const uint32_t isd = (byte == q.delim) ? 1u : 0u;
            hitm[p >> 6] |= (uint64_t)(hit & ~seen) << (p & 63);
            dm[p >> 6] |= (uint64_t)isd << (p & 63);
            seen |= hit;
            if (isd) {                                  // select, no branch: see RF above
#pragma unroll
                for (int e = 0; e <= K; ++e) A.R[e] = RF.R[e];
                seen = rf_hit ? 1u : 0u;
            }
        }
        if (rf_hit) {
#pragma unroll
            for (int i = 0; i < NMW; ++i) hit2m[i] = dm[i];   // a match right after every delimiter
        }
        bool any = false;
#pragma unroll
        for (int i = 0; i < NMW; ++i) any |= (hitm[i] | hit2m[i]) != 0;
        if (any && LEAN) {
            // record start = 1 + last delimiter in front of the event: taken from the window's
            // delimiter mask when it is there, else found by looking back from the window
            auto last_delim_below = [&](uint32_t x) -> int {   // relative index or -1
                int best = -1;
#pragma unroll
                for (int i = 0; i < NMW; ++i) {
                    const int lo = i * 64;
                    uint64_t m = dm[i];
                    if ((int)x < lo + 64) m &= (int)x > lo ? ((1ull << (x - lo)) - 1ull) : 0ull;
                    if (m) best = lo + 63 - __clzll((long long)m);
                }
                return best;
            };
            uint64_t before_ws = ~1ull;                         // lazily computed
            auto start_of = [&](uint32_t x) -> uint64_t {
                const int d = last_delim_below(x);
                if (d >= 0) return ws + (uint64_t)d + 1;
                if (before_ws == ~1ull) before_ws = lean_record_start(text, ws, q.delim, mk);
                return before_ws;
            };
#pragma unroll
            for (int i = 0; i < NMW; ++i) {
                uint64_t hm = hitm[i];
                while (hm) {
                    const uint32_t p = (uint32_t)(i * 64 + __ffsll((long long)hm) - 1);
                    hm &= hm - 1;
                    const uint64_t st = start_of(p);
                    if (st != ~0ull) lean_insert(mk, st);
                }
                uint64_t h2 = hit2m[i];
                while (h2) {
                    const uint32_t p = (uint32_t)(i * 64 + __ffsll((long long)h2) - 1);
                    h2 &= h2 - 1;
                    lean_insert(mk, ws + p + 1);                // the record right after delimiter p
                }
            }
        } else if (any) {
            // delimiters in [ws, x) from the delimiter mask
            auto delims_before = [&](uint32_t x) {
                uint32_t c = 0;
#pragma unroll
                for (int i = 0; i < NMW; ++i) {
                    const int lo = i * 64;
                    if ((int)x >= lo + 64) c += (uint32_t)__popcll(dm[i]);
                    else if ((int)x > lo) c += (uint32_t)__popcll(dm[i] & ((1ull << (x - lo)) - 1ull));
                }
                return c;
            };
            const uint32_t r0 = rc_anchor - delims_before((uint32_t)(anchor - ws));
#pragma unroll
            for (int i = 0; i < NMW; ++i) {
                uint64_t hm = hitm[i];
                while (hm) {
                    const uint32_t p = (uint32_t)(i * 64 + __ffsll((long long)hm) - 1);
                    hm &= hm - 1;
                    mark_record(mk, r0 + delims_before(p), ws + p);
                }
                uint64_t h2 = hit2m[i];
                while (h2) {
                    const uint32_t p = (uint32_t)(i * 64 + __ffsll((long long)h2) - 1);
                    h2 &= h2 - 1;
                    mark_record(mk, r0 + delims_before(p + 1u), ws + p + 1u);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// fullscan: the automaton over every byte
// ---------------------------------------------------------------------------------------
template <typename WT, int K>
__global__ __launch_bounds__(AGH_FS_THREADS) void k_fullscan(
    const uint8_t *__restrict__ text, uint64_t n, agh_dev_query q,
    const WT *__restrict__ mask_g, const uint32_t *__restrict__ strip_prefix,
    const uint32_t *__restrict__ wave_prefix, uint32_t n_strips, agh_marks mk)
{
    // slot 0 = the 256 bytes in front of the tile (warm-up halo), slots 1..256 = lane chunks
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    WT *lmask = reinterpret_cast<WT *>(lds);
    uint8_t *tile = lds + 256 * sizeof(WT);
    lmask[threadIdx.x] = mask_g[threadIdx.x];

    const uint64_t tile_bytes = (uint64_t)AGH_FS_THREADS * AGH_FS_CHUNK;
    const uint64_t n_tiles = (n + tile_bytes - 1) / tile_bytes;
    const uint32_t total_delims = mk.counters[AGH_C_NDELIM];
    const WT finalbit = (WT)1 << (q.m - 1);
    const uint32_t dd = q.delim * 0x01010101u;
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    const uint32_t warm = ((uint32_t)(q.m + q.k + 1) + 15u) & ~15u;   // <= 80 bytes

    for (uint64_t tix = blockIdx.x; tix < n_tiles; tix += gridDim.x) {
        const uint64_t t0 = tix * tile_bytes;
        __syncthreads();
        // cooperative, coalesced global -> LDS of 257 chunks (halo + tile), 16 B pieces
        for (uint32_t pc = threadIdx.x; pc < (AGH_FS_THREADS + 1) * (AGH_FS_CHUNK / 16);
             pc += AGH_FS_THREADS) {
            const uint32_t slot = pc / (AGH_FS_CHUNK / 16), sub = pc % (AGH_FS_CHUNK / 16);
            const int64_t g = (int64_t)t0 - (int64_t)AGH_FS_CHUNK + (int64_t)pc * 16;
            uint4 v = make_uint4(fill4, fill4, fill4, fill4);
            if (g >= 0 && (uint64_t)g < n16) v = *reinterpret_cast<const uint4 *>(text + g);
            *reinterpret_cast<uint4 *>(tile + slot * AGH_FS_SLOT + sub * 16) = v;
        }
        __syncthreads();

        const uint64_t cs = t0 + (uint64_t)threadIdx.x * AGH_FS_CHUNK;
        const uint8_t *mine = tile + (threadIdx.x + 1) * AGH_FS_SLOT;
        uint32_t my_delims = 0;
        uint64_t ce = cs + AGH_FS_CHUNK;
        if (ce > n) ce = n;
        if (cs < n) {
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
        if (cs == 0) {
            A.step(lmask[q.head_byte], finalbit); // asearch.c:69-78
        } else {
            const uint8_t *halo = tile + threadIdx.x * AGH_FS_SLOT + (AGH_FS_CHUNK - warm);
            for (uint32_t i = 0; i < warm; ++i) {
                const uint32_t c = halo[i];
                A.step(lmask[c], finalbit);
                if (c == q.delim) {
                    A.reset();
                    A.step(lmask[c], finalbit);
                }
            }
        }
        bool seen = false;
        const uint32_t len = (uint32_t)(ce - cs);
        for (uint32_t i = 0; i < len; ++i) {
            const uint32_t c = mine[i];
            bool hit = A.step(lmask[c], finalbit);
            if (hit && !seen) {
                seen = true;
                mark_record(mk, rec, cs + i);
            }
            if (c == q.delim) {
                A.reset();
                ++rec;
                seen = false;
                if (A.step(lmask[c], finalbit)) {
                    seen = true;
                    mark_record(mk, rec, cs + i + 1);
                }
            }
        }
        if (ce == n && q.tail_virtual) {        // asearch.c:87-91
            if (A.step(lmask[q.delim], finalbit) && !seen) mark_record(mk, rec, n);
            A.reset();
            if (A.step(lmask[q.delim], finalbit)) mark_record(mk, rec + 1u, n);
        }
    }
}

// ---------------------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------------------
template <typename WT, int K, int NCH, bool LEAN>
static void launch_verify_n(const agh_scan_args &a, hipStream_t st)
{
    uint32_t blocks = (a.nw + AGH_VGROUP - 1u) / AGH_VGROUP;
    if (!blocks) return;
    hipLaunchKernelGGL((k_verify<WT, K, NCH, LEAN>), dim3(blocks), dim3(256), 0, st,
                       (const uint8_t *)a.text, a.n, a.q, (const WT *)a.mask, a.cand,
                       a.wave_cand, a.wave_prefix, a.nw, a.mk);
}

template <typename WT, int K, bool LEAN>
static void launch_verify_t(const agh_scan_args &a, hipStream_t st)
{
    // window span = max(m+k+1, 16) + q + m + k bytes, fetched as ceil(span/16) pieces
    const int lw = a.q.m + a.q.k + 1 > 16 ? a.q.m + a.q.k + 1 : 16;
    const int nch = (lw + a.q.fq + a.q.m + a.q.k + 15) / 16;
    if (sizeof(WT) == 4) {                  // m <= 32: span <= 85
        if (nch <= 3) launch_verify_n<WT, K, 3, LEAN>(a, st);
        else launch_verify_n<WT, K, 6, LEAN>(a, st);
    } else {                                // m <= 64: span <= 149
        if (nch <= 7) launch_verify_n<WT, K, 7, LEAN>(a, st);
        else launch_verify_n<WT, K, 10, LEAN>(a, st);
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
    hipLaunchKernelGGL((k_fullscan<WT, K>), dim3(blocks), dim3(AGH_FS_THREADS), lds, st,
                       (const uint8_t *)a.text, a.n, a.q, (const WT *)a.mask, a.strip_prefix,
                       a.wave_prefix, a.n_strips, a.mk);
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

