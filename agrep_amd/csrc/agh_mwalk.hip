// agh_mwalk.hip -- count-only -f scans with one error over DENSE sets (BASELINE config 5 as SURVEY 8d words it:
// 1024 patterns of 4..12 bytes, k = 1): with pieces of two bytes every text position is a candidate and four
// records in five match, so a filter has nothing to filter.  What the reference's own multi-pattern engine does on
// such input is what pays here too: stop at the first hit of a record and go on behind its end
// (newmgrep.c:858-905: monkey1() returns to the record loop after the first verified entry; sgrep.c:1186-1204 jumps
// to the record end).
//
// One LANE per 1 .. 4 KiB of text (by the size of the segment: the launcher's caller picks), walking it position by
// position, in three kinds of steps:
//   advance   (cheap, up to eight positions in a row per lane) one 8-byte load around j; the two bytes at j select a
//             directory slot whose four 32-bit masks say which bytes next to them some entry could accept at all --
//             the byte behind the pair (the third byte of a longer piece, or one of the two nearest bytes of the other
//             side of a two-byte piece's pattern), the byte after that, and the two bytes in front (what
//             side_within_one_edit can accept: the first mismatch is the missing, the replaced or the extra byte).
//             Lossless; ~1 position in 5 stays a candidate.  Lanes stop at their candidate and wait for the others.
//   examine   (most lanes at a candidate now) the entries of the slot: piece verbatim at j and the same necessary
//             condition on the entry's own bytes inside the loop; the full test of the side -- two 64-bit words, the
//             round-3 verifier's predicate (agh_multi_inl.h mp_verify_at<K = 1>: the union over the patterns of the
//             k-error predicate) -- once per lane behind the loop
//   skip      after a hit: 16 bytes per step to the delimiter that ends the record
// (First version: every position walked its slot's entries with the full test inside the loop -- a wave pays the
// longest list and the test for every entry some lane passes: 4 GiB in 42.8 ms; profiles/r05_perf_c5_worded.log.)
// Records that lie inside one lane's kilobyte are counted by that lane alone (a register); a record that crosses
// into the next lane's text goes into the scan's hash set of record starts like in every other count-only engine,
// whoever finds the hit -- one record in ~13, not four in five of 53 M.
// The first 8 and the last 24 positions of the text: k_mwalk_edges through the general verifier.
#include "agh_multi_inl.h"

#define MW_CH 1024u                 // text bytes per lane
#ifndef MW_WAVES
#define MW_WAVES 16u                // waves per workgroup = per CU (one copy of the tables)
#endif

typedef uint64_t u64_a1 __attribute__((aligned(1)));

#define MW_ADVANCE 8                // positions a lane may advance before the wave examines the candidates it has

__device__ __forceinline__ uint32_t mw_delim_offset16(uint64_t F0, uint64_t F1, uint32_t dd)
{
    // offset of the first delimiter byte among 16, or 16
    auto nz = [&](uint32_t w) -> uint32_t { const uint32_t x = w ^ dd; return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu); };
    const uint32_t z0 = nz((uint32_t)F0), z1 = nz((uint32_t)(F0 >> 32)), z2 = nz((uint32_t)F1), z3 = nz((uint32_t)(F1 >> 32));
    if (z0) return (uint32_t)(__ffs((int)z0) - 1) >> 3;
    if (z1) return 4u + ((uint32_t)(__ffs((int)z1) - 1) >> 3);
    if (z2) return 8u + ((uint32_t)(__ffs((int)z2) - 1) >> 3);
    if (z3) return 12u + ((uint32_t)(__ffs((int)z3) - 1) >> 3);
    return 16u;
}

template <bool FOLD>
__global__ __launch_bounds__(MW_WAVES * 64) void k_mwalk(const uint8_t *__restrict__ text, uint64_t n, uint32_t delim,
                                                         agh_mwalk_dev mw, agh_marks mk, uint32_t *__restrict__ ticket,
                                                         uint32_t n_tiles, uint32_t ch)
{
    __shared__ uint4 fmask[AGH_MW_DIR];                   // 64 KiB
    __shared__ uint32_t dir[AGH_MW_DIR];                  // 16 KiB
    __shared__ uint4 ent[AGH_MW_MAX_ENT];                 // 48 KiB
    for (uint32_t i = threadIdx.x; i < AGH_MW_DIR; i += MW_WAVES * 64) {
        fmask[i] = mw.fmask[i];
        dir[i] = mw.dir[i];
    }
    for (uint32_t i = threadIdx.x; i < mw.n_ent; i += MW_WAVES * 64) ent[i] = mw.ent[i];
    __syncthreads();
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const uint32_t total_waves = gridDim.x * MW_WAVES;
    const uint32_t dd = delim * 0x01010101u;
    const uint64_t lo_lim = 8, hi_lim = n - 24;           // (n >= 32: the launcher checks)
    uint32_t local = 0;                                   // matched records that lie inside my kilobytes
    uint32_t t = blockIdx.x * MW_WAVES + wib;
    while (t < n_tiles) {
        const uint64_t cs = (uint64_t)t * (64u * ch) + (uint64_t)lane * ch;
        uint64_t j = cs < lo_lim ? lo_lim : cs;
        const uint64_t end = cs + ch < hi_lim ? cs + ch : hi_lim;
        bool active = j < end;
        // the start of the record at j: known if a delimiter stands right in front of my text
        uint64_t rstart = (active && text[j - 1] == delim) ? j : ~0ull;
        bool own = rstart != ~0ull;                       // the record at j starts inside my text
        bool skipping = false, cand = false;
        uint32_t cslot = 0;                               // directory slot of the candidate at j
        while (__ballot(active)) {
            // ---- skip, then advance: cheap steps, every lane for itself -------------------------------
            // (a lane's advance steps of one round read t[j0 - 2 .. j0 + 14) once: the window of step p is that
            // 128-bit word shifted by p bytes -- one pair of loads per round instead of one load per position)
            // skip first, in a loop of its own (a record's tail is three or four steps; inside the advance loop every
            // one of its eight steps paid for this path as well)
            while (__ballot(active && skipping)) {
                if (active && skipping) {
                    // the delimiter that ends the matched record, 16 bytes per step
                    const uint64_t F0 = *reinterpret_cast<const u64_a1 *>(text + j);
                    const uint64_t F1 = *reinterpret_cast<const u64_a1 *>(text + j + 8);
                    const uint32_t d = mw_delim_offset16(F0, F1, dd);
                    if (d < 16u && j + d < end) {         // the record ends inside my text
                        if (own) ++local;                 // ... and began there: mine alone
                        else if (rstart != ~0ull) lean_insert(mk, rstart);
                        j += d + 1u;
                        rstart = j;
                        own = true;
                        skipping = false;
                        if (j >= end) active = false;
                    } else if (d < 16u || j + 16u >= end) {
                        // it crosses into the next lane's text: the set of record starts sorts out who counts it
                        if (rstart != ~0ull) lean_insert(mk, rstart);
                        active = false;
                    } else {
                        j += 16u;
                    }
                }
            }
            const uint64_t j0 = j;
            const bool may_adv = active && !cand;
            uint64_t W0 = 0, W1 = 0;
            if (may_adv) {
                W0 = *reinterpret_cast<const u64_a1 *>(text + j - 2);
                W1 = *reinterpret_cast<const u64_a1 *>(text + j + 6);
            }
#pragma clang loop unroll(disable)
            for (int step = 0; step < MW_ADVANCE; ++step) {
                const bool adv = may_adv && active && !cand;
                if (!__ballot(adv)) break;
                if (adv) {
                    const uint32_t p8 = (uint32_t)(j - j0) * 8u;            // 0, 8, .. 56
                    const uint64_t W = p8 ? ((W0 >> p8) | (W1 << (64u - p8))) : W0;      // t[j-2 .. j+6)
                    uint32_t pair = (uint32_t)(W >> 16) & 0xffffu;
                    if ((pair & 0xffu) == delim) {        // (no entry holds the delimiter byte)
                        ++j;
                        rstart = j;
                        own = true;
                    } else {
                        if (FOLD) pair = swar_lower(pair);
                        const uint32_t slot = agh_mw_slot(pair);
                        const uint4 fm = fmask[slot];     // (all zero where no entry starts with this pair)
                        const uint32_t hit = (fm.x >> ((uint32_t)(W >> 32) & 31u)) | (fm.y >> ((uint32_t)(W >> 40) & 31u)) |
                                             (fm.z >> ((uint32_t)(W >> 8) & 31u)) | (fm.w >> ((uint32_t)W & 31u));
                        if (hit & 1u) { cand = true; cslot = slot; }
                        else ++j;
                    }
                    if (j >= end) active = false;
                }
            }
            // ---- examine: the candidates -------------------------------------------------------------
            if (__ballot(cand)) {
                bool matched = false;
                uint64_t P = 0, F0 = 0, F1 = 0;
                uint32_t first = 0, cnt = 0;
                if (cand) {
                    P = *reinterpret_cast<const u64_a1 *>(text + j - 8);
                    F0 = *reinterpret_cast<const u64_a1 *>(text + j);
                    F1 = *reinterpret_cast<const u64_a1 *>(text + j + 8);
                    if (FOLD) {
                        P = (uint64_t)swar_lower((uint32_t)P) | ((uint64_t)swar_lower((uint32_t)(P >> 32)) << 32);
                        F0 = (uint64_t)swar_lower((uint32_t)F0) | ((uint64_t)swar_lower((uint32_t)(F0 >> 32)) << 32);
                        F1 = (uint64_t)swar_lower((uint32_t)F1) | ((uint64_t)swar_lower((uint32_t)(F1 >> 32)) << 32);
                    }
                    const uint32_t dr = dir[cslot];
                    first = dr >> 16;
                    cnt = dr & 0xffffu;
                }
                const uint32_t lo = (uint32_t)F0, hi = (uint32_t)(F0 >> 32);
                const uint64_t Ph = __builtin_bswap64(P);               // the bytes in front of j, nearest first
                uint32_t i = 0;
                while (__ballot(i < cnt)) {
                    // the next entry whose piece stands at j and whose side passes the necessary condition ...
                    bool pend = false;
                    uint64_t S = 0, B = 0;
                    uint32_t L = 0;
                    while (i < cnt && !pend) {
                        const uint4 e = ent[first + i];
                        ++i;
                        const uint32_t pl = e.y >> 24;                  // piece length 2..7
                        uint32_t diff = (lo ^ e.x) & (pl >= 4u ? 0xffffffffu : ((1u << (8u * pl)) - 1u));
                        if (pl > 4u) diff |= (hi ^ e.y) & ((1u << (8u * (pl - 4u))) - 1u);
                        if (diff) continue;
                        const uint32_t meta = e.w >> 24;
                        L = meta & 7u;
                        B = (uint64_t)e.z | ((uint64_t)(e.w & 0xffffffu) << 32);
                        S = (meta & 8u) ? Ph : ((F0 >> (8u * pl)) | (F1 << (64u - 8u * pl)));
                        // S0 in {B0, B1} or S1 in {B0, B1} (L == 1: anything goes -- the one byte may be the missing one)
                        const uint32_t s0 = (uint32_t)S & 0xffu, s1 = (uint32_t)(S >> 8) & 0xffu;
                        const uint32_t b0 = (uint32_t)B & 0xffu, b1 = (uint32_t)(B >> 8) & 0xffu;
                        pend = L < 2u || s0 == b0 || s0 == b1 || s1 == b0 || s1 == b1;
                    }
                    // ... gets the full test, once per lane and round
                    if (pend && side_within_one_edit(S, B, L, delim)) {
                        matched = true;
                        i = cnt;
                    }
                }
                if (cand) {
                    cand = false;
                    if (matched) {
                        if (rstart == ~0ull) rstart = lean_record_start(text, j, delim, mk);   // (~0: noted as a give-up)
                        mk.counters[AGH_C_ANYHIT] = 1u;
                        skipping = true;                  // (from j: the delimiter search starts here)
                    } else {
                        ++j;
                        if (j >= end) active = false;
                    }
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
        const uint32_t ch = a.ch ? a.ch : MW_CH;
        const uint64_t n_tiles = (a.n + 64u * ch - 1u) / (64u * ch);
        if (n_tiles > 0xffffffffull - 65536ull) return false;
        // masks, directory and up to 3072 entries: 128 KiB of LDS -- one workgroup per CU
        uint32_t blocks = a.n_cu ? a.n_cu : 256u;
        const uint32_t need = (uint32_t)((n_tiles + MW_WAVES - 1u) / MW_WAVES);
        if (blocks > need) blocks = need;
        if (a.q.fold)
            hipLaunchKernelGGL((k_mwalk<true>), dim3(blocks), dim3(MW_WAVES * 64), 0, st, (const uint8_t *)a.text, a.n, a.q.delim,
                               a.mw, a.mk, a.ticket, (uint32_t)n_tiles, ch);
        else
            hipLaunchKernelGGL((k_mwalk<false>), dim3(blocks), dim3(MW_WAVES * 64), 0, st, (const uint8_t *)a.text, a.n, a.q.delim,
                               a.mw, a.mk, a.ticket, (uint32_t)n_tiles, ch);
    }
    hipLaunchKernelGGL(k_mwalk_edges, dim3(1), dim3(64), 0, st, (const uint8_t *)a.text, a.n, a.q, a.mt, a.mk);
    return true;
}
