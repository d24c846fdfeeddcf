// agh_mwalk.hip -- count-only -f scans with one error over DENSE sets (BASELINE config 5 as SURVEY 8d words it:
// 1024 patterns of 4..12 bytes, k = 1): with pieces of two bytes every text position is a candidate and four
// records in five match, so a filter has nothing to filter.  What the reference's own multi-pattern engine does on
// such input is what pays here too: stop at the first hit of a record and go on behind its end
// (newmgrep.c:858-905: monkey1() returns to the record loop after the first verified entry; sgrep.c:1186-1204 jumps
// to the record end).
//
// One LANE per 1 KiB of text, walking it position by position:
//   examine   the entries whose piece starts with the two bytes at j (directory + entries in LDS): piece verbatim at
//             j, then the other side of its pattern within one edit of the <= 8 text bytes next to it
//             (side_within_one_edit: two 64-bit words, no automaton) -- the predicate of the round-3 verifier
//             (agh_multi_inl.h mp_verify_at<K = 1>), the union over the patterns of the k-error predicate
//   skip      after a hit: 16 bytes per step to the delimiter that ends the record
// Records that lie inside one lane's kilobyte are counted by that lane alone (a register); a record that crosses
// into the next lane's text goes into the scan's hash set of record starts like in every other count-only engine,
// whoever finds the hit -- one record in ~13, not four in five of 53 M.
// The first 8 and the last 24 positions of the text: k_mwalk_edges through the general verifier.
#include "agh_multi_inl.h"

#define MW_CH 1024u                 // text bytes per lane
#define MW_WAVES 8u                 // waves per workgroup (one copy of the tables: 64 KiB of LDS, two workgroups per CU)

typedef uint64_t u64_a1 __attribute__((aligned(1)));

template <bool FOLD>
__global__ __launch_bounds__(MW_WAVES * 64) void k_mwalk(const uint8_t *__restrict__ text, uint64_t n, uint32_t delim,
                                                         agh_mwalk_dev mw, agh_marks mk, uint32_t *__restrict__ ticket,
                                                         uint32_t n_tiles)
{
    __shared__ uint4 ent[AGH_MW_MAX_ENT];
    __shared__ uint32_t dir[AGH_MW_DIR];
    for (uint32_t i = threadIdx.x; i < mw.n_ent; i += MW_WAVES * 64) ent[i] = mw.ent[i];
    for (uint32_t i = threadIdx.x; i < AGH_MW_DIR; i += MW_WAVES * 64) dir[i] = mw.dir[i];
    __syncthreads();
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const uint32_t total_waves = gridDim.x * MW_WAVES;
    const uint32_t dd = delim * 0x01010101u;
    const uint64_t lo_lim = 8, hi_lim = n - 24;           // (n >= 32: the launcher checks)
    uint32_t local = 0;                                   // matched records that lie inside my kilobytes
    uint32_t t = blockIdx.x * MW_WAVES + wib;
    while (t < n_tiles) {
        const uint64_t cs = (uint64_t)t * (64u * MW_CH) + (uint64_t)lane * MW_CH;
        uint64_t j = cs < lo_lim ? lo_lim : cs;
        const uint64_t end = cs + MW_CH < hi_lim ? cs + MW_CH : hi_lim;
        bool active = j < end;
        // the start of the record at j: known if a delimiter stands right in front of my text
        uint64_t rstart = (active && text[j - 1] == delim) ? j : ~0ull;
        bool own = rstart != ~0ull;                       // the record at j starts inside my text
        bool skipping = false;
        while (__ballot(active)) {
            if (active) {
                uint64_t P = *reinterpret_cast<const u64_a1 *>(text + j - 8);
                uint64_t F0 = *reinterpret_cast<const u64_a1 *>(text + j);
                uint64_t F1 = *reinterpret_cast<const u64_a1 *>(text + j + 8);
                if (skipping) {
                    // the delimiter that ends the matched record, 16 bytes per step (on the raw bytes)
                    const uint32_t z0 = (uint32_t)(((((uint32_t)F0 ^ dd) & 0x7f7f7f7fu) + 0x7f7f7f7fu) | ((uint32_t)F0 ^ dd) | 0x7f7f7f7fu);
                    const uint32_t z1 = (uint32_t)((((((uint32_t)(F0 >> 32)) ^ dd) & 0x7f7f7f7fu) + 0x7f7f7f7fu) | (((uint32_t)(F0 >> 32)) ^ dd) | 0x7f7f7f7fu);
                    const uint32_t z2 = (uint32_t)(((((uint32_t)F1 ^ dd) & 0x7f7f7f7fu) + 0x7f7f7f7fu) | ((uint32_t)F1 ^ dd) | 0x7f7f7f7fu);
                    const uint32_t z3 = (uint32_t)((((((uint32_t)(F1 >> 32)) ^ dd) & 0x7f7f7f7fu) + 0x7f7f7f7fu) | (((uint32_t)(F1 >> 32)) ^ dd) | 0x7f7f7f7fu);
                    // bit 7 of a byte of ~z: that byte is the delimiter
                    uint32_t d = 16u;
                    if (~z0) d = (uint32_t)(__ffs((int)~z0) - 1) >> 3;
                    else if (~z1) d = 4u + ((uint32_t)(__ffs((int)~z1) - 1) >> 3);
                    else if (~z2) d = 8u + ((uint32_t)(__ffs((int)~z2) - 1) >> 3);
                    else if (~z3) d = 12u + ((uint32_t)(__ffs((int)~z3) - 1) >> 3);
                    if (d < 16u && j + d < end) {         // the record ends inside my text
                        if (own) ++local;                 // ... and began there: mine alone
                        else if (rstart != ~0ull) lean_insert(mk, rstart);
                        j += d + 1u;
                        rstart = j;
                        own = true;
                        skipping = false;
                    } else if (d < 16u || j + 16u >= end) {
                        // it crosses into the next lane's text: the set of record starts sorts out who counts it
                        if (rstart != ~0ull) lean_insert(mk, rstart);
                        j = end;
                    } else {
                        j += 16u;
                    }
                } else {
                    if (FOLD) {
                        P = (uint64_t)swar_lower((uint32_t)P) | ((uint64_t)swar_lower((uint32_t)(P >> 32)) << 32);
                        F0 = (uint64_t)swar_lower((uint32_t)F0) | ((uint64_t)swar_lower((uint32_t)(F0 >> 32)) << 32);
                        F1 = (uint64_t)swar_lower((uint32_t)F1) | ((uint64_t)swar_lower((uint32_t)(F1 >> 32)) << 32);
                    }
                    const uint32_t lo = (uint32_t)F0, hi = (uint32_t)(F0 >> 32);
                    if ((lo & 0xffu) == delim) {          // (no entry holds the delimiter byte)
                        ++j;
                        rstart = j;
                        own = true;
                    } else {
                        const uint32_t dr = dir[agh_mw_slot(lo & 0xffffu)];
                        const uint32_t first = dr >> 16, cnt = dr & 0xffffu;
                        bool matched = false;
                        for (uint32_t i = 0; i < cnt && !matched; ++i) {
                            const uint4 e = ent[first + i];
                            const uint32_t pl = e.y >> 24;                      // piece length 2..7
                            uint32_t diff = (lo ^ e.x) & (pl >= 4u ? 0xffffffffu : ((1u << (8u * pl)) - 1u));
                            if (pl > 4u) diff |= (hi ^ e.y) & ((1u << (8u * (pl - 4u))) - 1u);
                            if (diff) continue;
                            const uint32_t meta = e.w >> 24, L = meta & 7u;
                            const uint64_t B = (uint64_t)e.z | ((uint64_t)(e.w & 0xffffffu) << 32);
                            uint64_t S;
                            if (meta & 8u) S = __builtin_bswap64(P);           // the head of the pattern in front of the piece
                            else S = (F0 >> (8u * pl)) | (F1 << (64u - 8u * pl));   // the rest behind it
                            matched = side_within_one_edit(S, B, L, delim);
                        }
                        if (matched) {
                            if (rstart == ~0ull) rstart = lean_record_start(text, j, delim, mk);   // (~0: noted as a give-up)
                            mk.counters[AGH_C_ANYHIT] = 1u;
                            skipping = true;              // (from j: the delimiter search starts here)
                        } else {
                            ++j;
                        }
                    }
                }
                if (j >= end) {
                    // a matched record still open at the end of my text belongs to the set as well
                    if (skipping && rstart != ~0ull) lean_insert(mk, rstart);
                    active = false;
                }
            }
        }
        uint32_t tk = 0;
        if (lane == 0) tk = atomicAdd(ticket, 1u);
        t = total_waves + (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
    }
    local = wave_sum_to_lane63(local);
    if (lane == 63 && local) atomicAdd(&mk.counters[AGH_C_MATCHED], local);
}

// The positions k_mwalk leaves out: fewer than 8 bytes in front of them or fewer than 24 behind.  Their records
// cross the walk's limits, so whatever both kernels find there meets in the hash set.
__global__ __launch_bounds__(64) void k_mwalk_edges(const uint8_t *__restrict__ text8, uint64_t n, agh_dev_query q,
                                                    agh_multi_dev mt, agh_marks mk)
{
    const uint32_t lane = (uint32_t)lane_id();
    uint64_t j = ~0ull;
    if (lane < 8u) j = lane;
    else if (lane < 32u && n >= 24u + 8u) j = n - 24u + (lane - 8u);
    else if (lane < 32u && lane < n) j = lane;            // (texts below 32 bytes: all of it)
    if (j < n) mp_verify_at<true, 1>(text8, n, q, mt, j, 0u, mk);
}

// false: no instance for this query / text -- the caller takes the general multi-pattern kernels
bool agh_launch_mwalk(const agh_mwalk_args &a, hipStream_t st)
{
    if (a.q.k != 1 || a.q.mb || !a.n || !a.mw.n_ent || a.mw.n_ent > AGH_MW_MAX_ENT) return false;
    if (a.n >= 32u) {
        const uint64_t n_tiles = (a.n + 64u * MW_CH - 1u) / (64u * MW_CH);
        if (n_tiles > 0xffffffffull - 65536ull) return false;
        uint32_t blocks = (a.n_cu ? a.n_cu : 256u) * 2u;
        const uint32_t need = (uint32_t)((n_tiles + MW_WAVES - 1u) / MW_WAVES);
        if (blocks > need) blocks = need;
        if (a.q.fold)
            hipLaunchKernelGGL((k_mwalk<true>), dim3(blocks), dim3(MW_WAVES * 64), 0, st, (const uint8_t *)a.text, a.n, a.q.delim,
                               a.mw, a.mk, a.ticket, (uint32_t)n_tiles);
        else
            hipLaunchKernelGGL((k_mwalk<false>), dim3(blocks), dim3(MW_WAVES * 64), 0, st, (const uint8_t *)a.text, a.n, a.q.delim,
                               a.mw, a.mk, a.ticket, (uint32_t)n_tiles);
    }
    hipLaunchKernelGGL(k_mwalk_edges, dim3(1), dim3(64), 0, st, (const uint8_t *)a.text, a.n, a.q, a.mt, a.mk);
    return true;
}
