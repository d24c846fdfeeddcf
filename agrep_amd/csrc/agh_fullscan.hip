// agh_fullscan.hip -- the k-error automaton over EVERY text byte (the asearch.c:94-199 shape):
// the engine of queries the sample filter cannot serve (wide classes with short literal cores,
// 1-byte pieces, multi-byte delimiters together with costs, -v with record output).
#include "agh_verify_inl.h"

// The ring feeder reads the two 64-byte halves of a 128-byte line in consecutive rounds: plain
// loads, so that the second half still finds the line (non-temporal loads, which buy the sweeps
// their last 10 %, cost here: 4 GiB count-only, k = 0 / 1: 3.55 / 2.94 -> 3.82 / 3.17 TB/s;
// make VARIANT=fsnt VARFLAGS=-DAGH_FS_NT=1 builds the other form for A/B, profiles/r03_perf_fullscan*.log)
#ifndef AGH_FS_NT
#define AGH_FS_NT 0
#endif
__device__ __forceinline__ uint4 fs_load(const uint4 *p)
{
#if AGH_FS_NT
    return ld_stream(p);
#else
    return *p;
#endif
}

// ---------------------------------------------------------------------------------------
// fullscan: the automaton over every byte
// ---------------------------------------------------------------------------------------
// 16 consecutive text bytes (one LDS b128 read) through the automaton, branch-free: returns
// the 16-bit masks of new-match positions (h16) and delimiter-end positions (d16; an input in
// MB mode, where delimiter ends come from the bitmap).
template <typename WT, int K, bool MB, bool GEN>
__device__ __forceinline__ void fullscan_piece(uint4 v, const WT *lmask, WT finalbit,
                                               const agh_dev_query &q, const Automaton<WT, K> &RF,
                                               uint32_t rf_hit, Automaton<WT, K> &A,
                                               uint32_t &seen, uint32_t &h16, uint32_t &d16)
{
    const uint32_t dws[4] = {v.x, v.y, v.z, v.w};
    uint32_t h = 0, d = MB ? d16 : 0u;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const uint32_t byte = (dws[b >> 2] >> (8 * (b & 3))) & 0xffu;
        const uint32_t hit = A.template step_q<GEN>(lmask[byte], finalbit, q) ? 1u : 0u;
        const uint32_t isd = MB ? (d >> b) & 1u : ((byte == q.delim) ? 1u : 0u);
        h |= (hit & ~seen) << b;
        if (!MB) d |= isd << b;
        seen |= hit;
        if (isd) {
#pragma unroll
            for (int e = 0; e <= K; ++e) A.R[e] = RF.R[e];
            seen = rf_hit;
        }
    }
    h16 = h;
    d16 = d;
}

// The common case made cheap: unit costs, a one-byte delimiter that is not a member of any pattern
// position (Mask[delim] == 0 -- literal patterns cannot hold the record delimiter).  Then the
// state right after a boundary is R_e = 2^e - 1 (level e has its e leading deletions, nothing
// else), and the ordinary step on the delimiter byte (CM = 0) already produces a superset of it:
// R_e' = R_(e-1) | ((R_(e-1) | R_(e-1)') << 1 | 1) >= 2^e - 1.  So the boundary is one AND per level
// with (kb | 2^e - 1), kb = 0 for the delimiter byte and ~0 otherwise, read from LDS together with
// the byte's mask (one b64 / b128 lookup) and folded into the step by v_bitop3: no compare, no
// select, no branch.  Matches are not located here: the piece only ORs up the top level; a piece
// whose OR reaches the final bit (one in thousands) is replayed by the exact path from the state
// saved at its start.  17-18 VALU instructions per byte at k = 2 instead of 31.
template <typename WT>
struct MaskKill {
    WT cm, kb;
};

template <typename WT, int K>
__device__ __forceinline__ WT fullscan_piece_fast(uint4 v, const MaskKill<WT> *tab, Automaton<WT, K> &A)
{
    const uint32_t dws[4] = {v.x, v.y, v.z, v.w};
    WT any = 0;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const uint32_t byte = (dws[b >> 2] >> (8 * (b & 3))) & 0xffu;
        const MaskKill<WT> e = tab[byte];
        WT po = A.R[0];
        WT pn = ((po << 1) | (WT)1) & e.cm;
        A.R[0] = pn;
#pragma unroll
        for (int l = 1; l <= K; ++l) {
            const WT cur = A.R[l];
            const WT ne = ((((cur << 1) | (WT)1) & e.cm) | po | (((po | pn) << 1) | (WT)1)) &
                          (e.kb | (((WT)1 << l) - (WT)1));
            po = cur;
            pn = ne;
            A.R[l] = ne;
        }
        any |= A.R[K];
    }
    return any;
}

// 0x80 in every byte of w that equals the delimiter (dd = delimiter in all four bytes)
__device__ __forceinline__ uint32_t delim_bits(uint32_t w, uint32_t dd)
{
    const uint32_t x = w ^ dd;
    return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
}

// GEN: the general automaton (non-unit costs / <exact> segments) instead of the unit-cost one.
//
// Data feeding.  The automaton is serial over bytes, so the parallelism is one CHUNK per lane
// (AGH_FS_CHUNK = 1 KiB = one census strip; a wave owns a 64 KiB tile), and every lane needs ITS
// next bytes while a coalesced load hands consecutive bytes to consecutive lanes.  The transpose
// goes through a small per-wave LDS ring: each round the wave gathers the next 64 bytes of all 64
// chunks with four dwordx4 loads (4 lanes x 16 B per chunk: 64-byte segments, every text byte
// fetched once), writes them into the ring, and every lane reads back its own 64 bytes (row
// stride 80 B: conflict-free b128 reads).  The next round's loads are in flight while the current
// 64 bytes run through the automaton.  5 KiB of LDS per wave -> 7 workgroups per CU instead of
// the 2 a 64 KiB tile allowed, and the warm-up replay (m+k+1 bytes of halo per chunk)
// is 8 % of a 1 KiB chunk instead of 31 % of a 256-byte one.
// LEAN (count-only scans): no delimiter census in front of this kernel and no record numbers --
// a matched record is identified by the offset of its first byte (known from the delimiters the
// lane has walked over, else found by looking back from the chunk start) and goes into the hash set.
template <typename WT, int K, bool MB, bool GEN, bool LEAN>
__global__ __launch_bounds__(AGH_FS_THREADS) void k_fullscan(
    const uint8_t *__restrict__ text, uint64_t n, agh_dev_query q,
    const WT *__restrict__ mask_g, const uint32_t *__restrict__ strip_prefix,
    const uint32_t *__restrict__ wave_prefix, uint32_t n_strips, agh_marks mk,
    const uint64_t *__restrict__ dbm)
{
    __shared__ WT lmask[256];
    __shared__ MaskKill<WT> ktab[(MB || GEN) ? 1 : 256];
    __shared__ __attribute__((aligned(16))) uint8_t ring_all[(AGH_FS_THREADS / WAVE) * WAVE * AGH_FS_ROW];
    lmask[threadIdx.x] = mask_g[threadIdx.x];
    if (!MB && !GEN) {
        ktab[threadIdx.x].cm = mask_g[threadIdx.x];
        ktab[threadIdx.x].kb = threadIdx.x == q.delim ? (WT)0 : ~(WT)0;
    }
    __syncthreads();
    // the cheap boundary handling applies (see fullscan_piece_fast); uniform over the grid
    const bool fastd = !MB && !GEN && lmask[q.delim & 0xffu] == (WT)0;
    const uint32_t dd = q.delim * 0x01010101u;

    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    uint8_t *ring = ring_all + wib * (WAVE * AGH_FS_ROW);
    const uint64_t tile_bytes = (uint64_t)WAVE * AGH_FS_CHUNK;
    const uint64_t n_tiles = (n + tile_bytes - 1) / tile_bytes;
    const uint32_t total_delims = LEAN ? 0u : mk.counters[AGH_C_NDELIM];
    const WT finalbit = (WT)1 << (q.m - 1);
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    const uint32_t warm = ((uint32_t)(q.m + q.k + 1) + 15u) & ~15u;   // <= 80 bytes

    // state right after a record boundary: reset + re-fed delimiter byte (asearch.c:175-186)
    Automaton<WT, K> RF;
    RF.reset();
    const uint32_t rf_hit = RF.template step_q<GEN>(lmask[q.delim], finalbit, q) ? 1u : 0u;

    // my part of the cooperative gather: 16 bytes of chunk (lane / 4 + 16 i), i = 0..3
    const uint32_t seg_lo = (uint32_t)lane >> 2, part = (uint32_t)lane & 3u;
    uint8_t *ring_w = ring + seg_lo * AGH_FS_ROW + part * 16u;
    const uint8_t *ring_r = ring + (uint32_t)lane * AGH_FS_ROW;

    for (uint64_t tile = (uint64_t)blockIdx.x * (AGH_FS_THREADS / WAVE) + wib; tile < n_tiles;
         tile += (uint64_t)gridDim.x * (AGH_FS_THREADS / WAVE)) {
        const uint64_t t0 = tile * tile_bytes;
        const uint64_t cs = t0 + (uint64_t)lane * AGH_FS_CHUNK;
        const bool mine = cs < n;
        uint64_t ce = cs + AGH_FS_CHUNK;
        if (ce > n) ce = n;
        const uint32_t len = mine ? (uint32_t)(ce - cs) : 0u;
        auto gather = [&](uint32_t r, uint4 (&g)[4]) {
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint64_t a = t0 + (uint64_t)(seg_lo + 16u * i) * AGH_FS_CHUNK + r * AGH_FS_ROUND + part * 16u;
                g[i] = a < n16 ? fs_load(reinterpret_cast<const uint4 *>(text + a))
                               : make_uint4(fill4, fill4, fill4, fill4);
            }
        };
        uint4 g[4];
        gather(0, g);

        uint32_t rec = 0;
        uint64_t rstart = cs == 0 ? 0ull : ~0ull;       // LEAN: first byte of the current record (~0: not known yet)
        auto rstart_get = [&]() -> uint64_t {
            if (rstart == ~0ull)
                rstart = MB ? lean_record_start_mb(dbm, cs, mk) : lean_record_start(text, cs, q.delim, mk);
            return rstart;
        };
        Automaton<WT, K> A;
        A.reset();
        uint32_t seen = 0;
        if (mine) {
            const uint64_t strip = cs >> AGH_STRIP_SHIFT;       // chunk == strip
            if (!LEAN)
                rec = strip < n_strips ? wave_prefix[strip / AGH_WAVE_STRIPS] + strip_prefix[strip] : total_delims;
            if (cs == 0) {
                A.template step_q<GEN>(lmask[q.head_byte], finalbit, q); // asearch.c:69-78
            } else {
                // rebuild the state from the m+k+1 (rounded to 16) bytes in front of the chunk
                for (uint32_t t = 0; t < warm / 16; ++t) {
                    uint32_t h16 = 0, d16 = 0;
                    if (MB) d16 = (uint32_t)dbm_bits64(dbm, cs - warm + 16u * t) & 0xffffu;
                    fullscan_piece<WT, K, MB, GEN>(
                        *reinterpret_cast<const uint4 *>(text + cs - warm + 16u * t), lmask, finalbit, q,
                        RF, rf_hit, A, seen, h16, d16);
                }
                seen = 0;                       // matches before cs belong to the previous lane
            }
        }
        bool seenb = false;
        for (uint32_t r = 0; r < AGH_FS_CHUNK / AGH_FS_ROUND; ++r) {
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i)
                *reinterpret_cast<uint4 *>(ring_w + 16u * i * AGH_FS_ROW) = g[i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the whole wave's rows are in
            __builtin_amdgcn_wave_barrier();
            uint4 v[4];
#pragma unroll
            for (uint32_t p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const uint4 *>(ring_r + 16u * p);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // read before the next round overwrites
            __builtin_amdgcn_wave_barrier();
            if (r + 1 < AGH_FS_CHUNK / AGH_FS_ROUND) gather(r + 1, g);   // in flight during the walk
#pragma unroll
            for (uint32_t p = 0; p < 4; ++p) {
                const uint32_t off = r * AGH_FS_ROUND + 16u * p;
                if (off + 16u <= len) {
                    bool exact = true;
                    if (!MB && !GEN && fastd) {
                        const Automaton<WT, K> at_start = A;
                        const WT any = fullscan_piece_fast<WT, K>(v[p], ktab, A);
                        if (any & finalbit) {
                            A = at_start;               // rare: a match ends in this piece -> exact replay
                        } else {
                            exact = false;
                            const uint32_t z0 = delim_bits(v[p].x, dd), z1 = delim_bits(v[p].y, dd);
                            const uint32_t z2 = delim_bits(v[p].z, dd), z3 = delim_bits(v[p].w, dd);
                            if (z0 | z1 | z2 | z3) {
                                seen = 0;               // the open record starts inside this piece
                                if (LEAN) {
                                    // one past the last delimiter of the piece
                                    const uint32_t zl = z3 ? z3 : (z2 ? z2 : (z1 ? z1 : z0));
                                    const uint32_t dwi = z3 ? 3u : (z2 ? 2u : (z1 ? 1u : 0u));
                                    rstart = cs + off + 4u * dwi + ((31u - (uint32_t)__clz((int)zl)) >> 3) + 1u;
                                } else {
                                    rec += (uint32_t)(__popc(z0) + __popc(z1) + __popc(z2) + __popc(z3));
                                }
                            }
                        }
                    }
                    if (exact) {
                    uint32_t h16 = 0, d16 = 0;
                    if (MB) d16 = (uint32_t)dbm_bits64(dbm, cs + off) & 0xffffu;
                    fullscan_piece<WT, K, MB, GEN>(v[p], lmask, finalbit, q, RF, rf_hit, A, seen, h16, d16);
                    uint32_t ev = h16 | (rf_hit ? d16 : 0u);
                    while (ev) {                        // rare: a record matched in this piece
                        const uint32_t b = (uint32_t)__ffs((int)ev) - 1u;
                        ev &= ev - 1u;
                        const uint32_t dl = d16 & ((1u << b) - 1u);     // delimiters in front of b
                        if (LEAN) {
                            if ((h16 >> b) & 1u) {
                                const uint64_t st = dl ? cs + off + (32u - (uint32_t)__clz((int)dl)) : rstart_get();
                                if (st != ~0ull) lean_insert(mk, st);
                            }
                            if (rf_hit && ((d16 >> b) & 1u)) lean_insert(mk, cs + off + b + 1u);
                        } else {
                            const uint32_t below = (uint32_t)__popc(dl);
                            if ((h16 >> b) & 1u) mark_record(mk, rec + below, cs + off + b);
                            if (rf_hit && ((d16 >> b) & 1u)) mark_record(mk, rec + below + 1u, cs + off + b + 1u);
                        }
                    }
                    if (LEAN) {
                        if (d16) rstart = cs + off + (32u - (uint32_t)__clz((int)d16));
                    } else {
                        rec += (uint32_t)__popc(d16);
                    }
                    }
                } else if (off < len) {                 // the last, partial piece of the text
                    const uint32_t dws[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
                    seenb = seen != 0;
                    for (uint32_t i = 0; off + i < len; ++i) {
                        const uint32_t c = (dws[i >> 2] >> (8u * (i & 3u))) & 0xffu;
                        const bool hit = A.template step_q<GEN>(lmask[c], finalbit, q);
                        if (hit && !seenb) {
                            seenb = true;
                            if (LEAN) {
                                const uint64_t st = rstart_get();
                                if (st != ~0ull) lean_insert(mk, st);
                            } else {
                                mark_record(mk, rec, cs + off + i);
                            }
                        }
                        if (MB ? dbm_bit(dbm, cs + off + i) != 0 : c == q.delim) {
                            A.reset();
                            ++rec;
                            rstart = cs + off + i + 1;
                            seenb = false;
                            if (A.template step_q<GEN>(lmask[c], finalbit, q)) {
                                seenb = true;
                                if (LEAN) lean_insert(mk, rstart); else mark_record(mk, rec, cs + off + i + 1);
                            }
                        }
                    }
                    seen = seenb ? 1u : 0u;
                }
            }
        }
        if (mine && ce == n && q.tail_virtual) {        // asearch.c:87-91
            const uint64_t st = LEAN ? rstart_get() : 0ull;
            if (!LEAN || st != ~0ull)
                feed_virtual_tail<WT, K, LEAN, GEN>(text, n, q, lmask, dbm, A, seen != 0, rec, st, mk);
        }
    }
}

// ---------------------------------------------------------------------------------------
// fast full scan: the hot loop without any record bookkeeping, two text streams per lane
// ---------------------------------------------------------------------------------------
// For the queries fullscan_piece_fast serves (unit costs, one-byte delimiter outside every pattern
// class) the kernel above still carries the exact path -- seen / record number / record start /
// h16 / d16, 146 VGPRs -- although one piece in thousands needs it.  Here the hot kernel ONLY runs
// the branch-free step and ORs up the top level; a piece whose OR reaches the final bit (or that
// holds the last byte of the text) is written to a per-tile list, and k_fullscan_replay walks those
// pieces exactly afterwards: state rebuilt from the m+k+1 bytes in front of the piece (SURVEY B.5,
// the same warm-up a chunk start uses), records identified after the fact.
// PACK: patterns of m <= 16 positions -- every literal query that ends up here with k <= 4 has
// m < 3(k+1) -- run TWO text streams per lane in the halves of one 32-bit word: stream B's state is
// R << 16.  The left shift carries bit 15 of stream A into bit 16, which is stream B's position 1
// and is forced to 1 by the "| 1" of every shift anyway, so the halves never disturb each other.
// A wave then owns a PAIR of 64 KiB tiles; per byte pair: two table reads, cm = cmA | cmB,
// kb = kbA & kbB, one automaton step -- 9-10 VALU instructions per byte at k = 2 instead of 21.
#define AGH_FF_SLICE 256u       // replay entries per 64 KiB tile (6 % of its pieces; more: the exact kernel)

template <typename WT, int K>
__device__ __forceinline__ WT ff_step(Automaton<WT, K> &A, WT cm, WT kb, WT ones)
{
    WT po = A.R[0];
    WT aprev = (po << 1) | ones;
    WT pn = aprev & cm;
    A.R[0] = pn;
#pragma unroll
    for (int l = 1; l <= K; ++l) {
        const WT cur = A.R[l];
        const WT al = (cur << 1) | ones;
        const WT b = (pn << 1) | aprev;                     // ((R(l-1) | R(l-1)') << 1) | 1
        const WT lvl = (WT)((((WT)1 << l) - (WT)1) * ones); // 2^l - 1 in every stream
        const WT ne = ((al & cm) | po | b) & (kb | lvl);
        po = cur;
        pn = ne;
        aprev = al;
        A.R[l] = ne;
    }
    return A.R[K];
}

// 16 bytes of one stream through the automaton (m <= 32 / 64: one stream per lane); -> OR of the top level
template <typename WT, int K>
__device__ __forceinline__ WT ff_piece(uint4 va, const MaskKill<WT> *tab, Automaton<WT, K> &A)
{
    const uint32_t da[4] = {va.x, va.y, va.z, va.w};
    WT any = 0;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const MaskKill<WT> e = tab[(da[b >> 2] >> (8 * (b & 3))) & 0xffu];
        any |= ff_step<WT, K>(A, e.cm, e.kb, (WT)1);
    }
    return any;
}

// ... and of two streams in the halves of a 32-bit word.  One table of 32-bit entries serves both:
// low half = the byte's position mask (m <= 16), high half = 0xffff unless the byte is the delimiter;
// two v_perm_b32 put the halves of the two entries where the packed step wants them
// (cm = maskA | maskB << 16, kb = killA | killB << 16).
template <int K>
__device__ __forceinline__ uint32_t ff_piece2(uint4 va, uint4 vb, const uint32_t *tab, Automaton<uint32_t, K> &A)
{
    const uint32_t da[4] = {va.x, va.y, va.z, va.w}, db[4] = {vb.x, vb.y, vb.z, vb.w};
    uint32_t any = 0;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const uint32_t ta = tab[(da[b >> 2] >> (8 * (b & 3))) & 0xffu];
        const uint32_t tb = tab[(db[b >> 2] >> (8 * (b & 3))) & 0xffu];
        const uint32_t cm = __builtin_amdgcn_perm(tb, ta, 0x05040100u);
        const uint32_t kb = __builtin_amdgcn_perm(tb, ta, 0x07060302u);
        any |= ff_step<uint32_t, K>(A, cm, kb, 0x00010001u);
    }
    return any;
}

// ... and of THREE streams in the 10-bit fields of a 32-bit word (m <= 10; round 6: the shape of agrep's everyday
// query -- a word with two or three errors -- lands here: `matching` -2 has pieces of two bytes and no filter).  The
// left shift carries a field's top bit into the next field's position 1, which the `| ones` of every shift sets
// anyway; what stands above position m in a field only ever moves upward.  One table per stream, its entries already
// in the stream's field: x = the byte's position mask, y = the field's ones unless the byte is the delimiter -- two
// v_or3 put the three entries together (a perm cannot: the fields are not bytes).
// Measured and dropped (profiles/r06_ab_fullscan_streams.log): four streams in the bytes (m <= 8) -- their ring
// leaves six waves per CU and they lose to three streams at every k (k = 2: 1.40 vs 1.26 ms per 4 GiB); rounds of 32
// instead of 64 bytes to shrink the ring -- the gather then reads half sectors, 2.4 ms whatever k.
template <int K, int NS>
__device__ __forceinline__ uint32_t ff_pieceN(const uint4 (&v)[NS], const uint2 *tab, Automaton<uint32_t, K> &A)
{
    static_assert(NS == 3, "three streams in 10-bit fields");
    constexpr uint32_t ones = 0x00100401u;
    uint32_t any = 0;
    // the table entries of four steps at a time, the next four in flight while these are walked (left to itself the
    // compiler reads a step's entries and waits for them: sixteen LDS round trips per piece)
    auto fetch = [&](int g, uint2 (&t)[4][NS]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = 4 * g + i;
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const uint32_t d = b >> 2 == 0 ? v[st].x : (b >> 2 == 1 ? v[st].y : (b >> 2 == 2 ? v[st].z : v[st].w));
                t[i][st] = tab[st * 256 + ((d >> (8 * (b & 3))) & 0xffu)];
            }
        }
    };
    auto walk = [&](const uint2 (&t)[4][NS]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t cm = t[i][0].x | t[i][1].x | t[i][2].x, kb = t[i][0].y | t[i][1].y | t[i][2].y;
            any |= ff_step<uint32_t, K>(A, cm, kb, ones);
        }
    };
    uint2 ta[4][NS], tb[4][NS];
    fetch(0, ta);
    fetch(1, tb);
    __builtin_amdgcn_sched_barrier(0);
    walk(ta);
    __builtin_amdgcn_sched_barrier(0);
    fetch(2, ta);
    __builtin_amdgcn_sched_barrier(0);
    walk(tb);
    __builtin_amdgcn_sched_barrier(0);
    fetch(3, tb);
    __builtin_amdgcn_sched_barrier(0);
    walk(ta);
    __builtin_amdgcn_sched_barrier(0);
    walk(tb);
    return any;
}

#define AGH_FF_THREADS 128u     // two waves per workgroup: 2 x 10 KiB of ring for the two-stream form

// NS: text streams per lane -- 1 (m <= 32 / 64), 2 (m <= 16, 16-bit halves), 3 (m <= 10, 10-bit fields)
template <typename WT, int K, int NS>
__global__ __launch_bounds__(AGH_FF_THREADS) void k_fullscan_fast(
    const uint8_t *__restrict__ text, uint64_t n, agh_dev_query q, const WT *__restrict__ mask_g,
    uint64_t *__restrict__ replay, uint32_t *__restrict__ tile_cnt, uint32_t *__restrict__ counters)
{
    constexpr uint32_t ROUND = AGH_FS_ROUND, ROW = ROUND + 16u, PARTS = ROUND / 16u;
    constexpr uint32_t FW = 32u / (NS >= 2 ? NS : 1), FMASK = NS >= 2 ? (1u << FW) - 1u : 0xffffffffu;
    __shared__ MaskKill<WT> tabA[NS == 1 ? 256 : 1];
    __shared__ uint32_t tab2[NS == 2 ? 256 : 1];
    __shared__ uint2 tabN[NS >= 3 ? NS * 256 : 1];
    __shared__ __attribute__((aligned(16))) uint8_t ring_all[(AGH_FF_THREADS / WAVE) * NS * WAVE * ROW];
    for (uint32_t c = threadIdx.x; c < 256u; c += AGH_FF_THREADS) {
        const WT cm = mask_g[c];
        const bool isd = c == q.delim;
        if constexpr (NS == 2) {
            tab2[c] = ((uint32_t)cm & 0xffffu) | (isd ? 0u : 0xffff0000u);
        } else if constexpr (NS >= 3) {
#pragma unroll
            for (int st = 0; st < NS; ++st)
                tabN[st * 256 + c] = make_uint2(((uint32_t)cm & FMASK) << (FW * st), isd ? 0u : FMASK << (FW * st));
        } else {
            tabA[c].cm = cm;
            tabA[c].kb = isd ? (WT)0 : ~(WT)0;
        }
    }
    __syncthreads();
    const WT finalA = (WT)1 << (q.m - 1);
    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    uint8_t *ring = ring_all + wib * (NS * WAVE * ROW);
    const uint64_t tile_bytes = (uint64_t)WAVE * AGH_FS_CHUNK;
    const uint64_t n_tiles = (n + tile_bytes - 1) / tile_bytes;
    const uint64_t n_units = (n_tiles + NS - 1) / NS;   // tiles, or groups of NS tiles
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    const uint32_t warm = ((uint32_t)(q.m + q.k + 1) + 15u) & ~15u;   // <= 80 bytes
    // a round's bytes of a chunk are loaded by PARTS neighbouring lanes, 16 bytes each: 64 / PARTS chunks per load
    const uint32_t seg_lo = (uint32_t)lane / PARTS, part = (uint32_t)lane % PARTS;
    uint8_t *ring_w = ring + seg_lo * ROW + part * 16u;
    const uint8_t *ring_r = ring + (uint32_t)lane * ROW;
    auto piece = [&](const uint4 (&v)[NS], Automaton<WT, K> &A) -> WT {
        if constexpr (NS == 2) return ff_piece2<K>(v[0], v[1], tab2, A);
        else if constexpr (NS >= 3) return ff_pieceN<K, NS>(v, tabN, A);
        else return ff_piece<WT, K>(v[0], tabA, A);
    };

    for (uint64_t unit = (uint64_t)blockIdx.x * (AGH_FF_THREADS / WAVE) + wib; unit < n_units;
         unit += (uint64_t)gridDim.x * (AGH_FF_THREADS / WAVE)) {
        uint64_t t0[NS], cs[NS];
        uint32_t len[NS], cnt[NS];
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            t0[st] = (unit * NS + (uint64_t)st) * tile_bytes;
            cs[st] = t0[st] + (uint64_t)lane * AGH_FS_CHUNK;
            len[st] = cs[st] < n ? (uint32_t)(n - cs[st] < AGH_FS_CHUNK ? n - cs[st] : AGH_FS_CHUNK) : 0u;
            cnt[st] = 0;
        }
        auto gather = [&](uint32_t r, uint4 (&g)[NS][PARTS]) {
#pragma unroll
            for (int st = 0; st < NS; ++st)
#pragma unroll
                for (uint32_t i = 0; i < PARTS; ++i) {
                    const uint64_t a = t0[st] + (uint64_t)(seg_lo + (64u / PARTS) * i) * AGH_FS_CHUNK + r * ROUND + part * 16u;
                    g[st][i] = a < n16 ? fs_load(reinterpret_cast<const uint4 *>(text + a))
                                       : make_uint4(fill4, fill4, fill4, fill4);
                }
        };
        uint4 g[NS][PARTS];
        gather(0, g);

        // state at the chunk starts: the m+k+1 (rounded to 16) bytes in front of them, or the
        // virtual head byte at the start of the text (asearch.c:69-78)
        Automaton<WT, K> A;
        A.reset();
        {
            const uint4 fillv = make_uint4(fill4, fill4, fill4, fill4);
            for (uint32_t t = 0; t < warm / 16; ++t) {
                uint4 wv[NS];
#pragma unroll
                for (int st = 0; st < NS; ++st) {
                    wv[st] = fillv;
                    if (len[st] && cs[st] >= warm) wv[st] = *reinterpret_cast<const uint4 *>(text + cs[st] - warm + 16u * t);
                }
                (void)piece(wv, A);
            }
            if (cs[0] == 0) {                   // (only stream 0 of the first unit starts the text)
                Automaton<WT, K> H;
                H.reset();
                const uint32_t hb = q.head_byte & 0xffu;
                const WT hcm = mask_g[hb];
                (void)ff_step<WT, K>(H, NS >= 2 ? (WT)(hcm & (WT)FMASK) : hcm, hb == q.delim ? (WT)0 : ~(WT)0, (WT)1);
#pragma unroll
                for (int l = 0; l <= K; ++l)
                    A.R[l] = NS >= 2 ? (WT)((A.R[l] & ~(WT)FMASK) | (H.R[l] & (WT)FMASK)) : H.R[l];
            }
        }
        for (uint32_t r = 0; r < AGH_FS_CHUNK / ROUND; ++r) {
#pragma unroll
            for (int st = 0; st < NS; ++st)
#pragma unroll
                for (uint32_t i = 0; i < PARTS; ++i)
                    *reinterpret_cast<uint4 *>(ring_w + st * (WAVE * ROW) + (64u / PARTS) * i * ROW) = g[st][i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the whole wave's rows are in
            __builtin_amdgcn_wave_barrier();
            if (r + 1 < AGH_FS_CHUNK / ROUND) gather(r + 1, g);   // in flight during the walk
            uint32_t flags = 0;                 // bit 4*st + p: piece p of stream st goes to the replay list
            // (not unrolled: across four pieces the scheduler hoists table reads until the two-stream
            // instances need 134-138 VGPRs -- one wave per SIMD less -- and a register cap spills)
#pragma unroll 1
            for (uint32_t p = 0; p < PARTS; ++p) {
                const uint32_t off = r * ROUND + 16u * p;
                // the lane's own 16 bytes of every stream, read when they are needed (the ring is not
                // written again before the next round: eight VGPRs of text instead of thirty-two)
                uint4 v[NS];
#pragma unroll
                for (int st = 0; st < NS; ++st)
                    v[st] = *reinterpret_cast<const uint4 *>(ring_r + st * (WAVE * ROW) + 16u * p);
                const WT any = piece(v, A);
#pragma unroll
                for (int st = 0; st < NS; ++st) {
                    const WT fb = NS >= 2 ? (WT)(finalA << (FW * st)) : finalA;
                    // a match may end in the piece, or the piece holds the last byte of the text
                    // (partial pieces and the appended delimiter, asearch.c:87-91, are the replay's)
                    const bool in_text = off < len[st];
                    const bool last = in_text && cs[st] + off + 16u >= n;
                    if (in_text && (((any & fb) != 0) || last)) flags |= 1u << (4 * st + p);
                }
            }
            if (__ballot(flags != 0u)) {
#pragma unroll
                for (int st = 0; st < NS; ++st)
#pragma unroll
                    for (uint32_t p = 0; p < PARTS; ++p) {
                        const bool f = (flags >> (4 * st + p)) & 1u;
                        const uint64_t fm = __ballot(f);
                        if (!fm) continue;
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                        const uint64_t tile = unit * NS + (uint64_t)st;
                        if (f) {
                            const uint32_t at = cnt[st] + rank;
                            if (at < AGH_FF_SLICE) replay[tile * AGH_FF_SLICE + at] = cs[st] + r * ROUND + 16u * p;
                            else counters[AGH_C_OVERFLOW] = 1u;
                        }
                        cnt[st] += (uint32_t)__popcll(fm);
                    }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const uint64_t tile = unit * NS + (uint64_t)st;
                if (tile < n_tiles) tile_cnt[tile] = cnt[st] < AGH_FF_SLICE ? cnt[st] : AGH_FF_SLICE;
            }
        }
    }
}

// The listed pieces, exactly: one lane per piece, one wave per tile.
template <typename WT, int K, bool LEAN>
__global__ __launch_bounds__(256) void k_fullscan_replay(
    const uint8_t *__restrict__ text, uint64_t n, agh_dev_query q, const WT *__restrict__ mask_g,
    const uint64_t *__restrict__ replay, const uint32_t *__restrict__ tile_cnt, uint32_t n_tiles,
    const uint32_t *__restrict__ strip_prefix, const uint32_t *__restrict__ wave_prefix,
    uint32_t n_strips, agh_marks mk)
{
    __shared__ WT lmask[256];
    lmask[threadIdx.x] = mask_g[threadIdx.x];
    __syncthreads();
    const WT finalbit = (WT)1 << (q.m - 1);
    const uint32_t warm = ((uint32_t)(q.m + q.k + 1) + 15u) & ~15u;
    const uint32_t total_delims = LEAN ? 0u : mk.counters[AGH_C_NDELIM];
    for (uint32_t tile = blockIdx.x * 4u + threadIdx.x / WAVE; tile < n_tiles; tile += gridDim.x * 4u) {
        const uint32_t cnt = tile_cnt[tile];
        for (uint32_t e = (uint32_t)lane_id(); e < cnt; e += WAVE) {
            const uint64_t P = replay[(uint64_t)tile * AGH_FF_SLICE + e];
            const uint64_t start = P >= warm ? P - warm : 0;
            uint64_t end = P + 16;
            if (end > n) end = n;
            Automaton<WT, K> A;
            A.reset();
            if (start == 0) A.step(lmask[q.head_byte], finalbit);
            uint64_t last_delim = ~0ull;        // last delimiter at an offset in [start, i)
            bool seen = false;
            uint32_t rec = 0;
            if (!LEAN) {
                // record number of byte P: census of its strip + the delimiters from the strip start
                const uint64_t strip = P >> AGH_STRIP_SHIFT;
                rec = strip < n_strips ? wave_prefix[strip / AGH_WAVE_STRIPS] + strip_prefix[strip] : total_delims;
                const uint32_t dd = q.delim * 0x01010101u;
                for (uint64_t i = strip << AGH_STRIP_SHIFT; i < P; i += 16)
                    rec += delims_in(*reinterpret_cast<const uint4 *>(text + i), dd);
            }
            for (uint64_t i = start; i < end; ++i) {
                const uint32_t c = text[i];
                const bool hit = A.step(lmask[c], finalbit);
                if (hit && !seen && i >= P) {
                    seen = true;
                    if (LEAN) {
                        const uint64_t st = last_delim != ~0ull ? last_delim + 1 : lean_record_start(text, start, q.delim, mk);
                        if (st != ~0ull) lean_insert(mk, st);
                    } else {
                        mark_record(mk, rec, i);
                    }
                }
                if (c == q.delim) {             // (outside every pattern class: the re-fed step cannot hit)
                    A.reset();
                    A.step(lmask[c], finalbit);
                    seen = false;
                    last_delim = i;
                    if (!LEAN && i >= P) ++rec;
                }
            }
            if (end == n && q.tail_virtual) {   // asearch.c:87-91
                uint64_t st = 0;
                if (LEAN) st = last_delim != ~0ull ? last_delim + 1 : lean_record_start(text, start, q.delim, mk);
                if (!LEAN || st != ~0ull)
                    feed_virtual_tail<WT, K, LEAN, false>(text, n, q, lmask, nullptr, A, seen, rec, st, mk);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// host-callable launcher
// ---------------------------------------------------------------------------------------
template <typename WT, int K>
static void launch_fullscan_t(const agh_scan_args &a, hipStream_t st)
{
    const uint64_t tile_bytes = (uint64_t)WAVE * AGH_FS_CHUNK;          // one wave, 64 KiB
    const uint64_t n_tiles = (a.n + tile_bytes - 1) / tile_bytes;
    if (!n_tiles) return;
    const uint64_t want = (n_tiles + (AGH_FS_THREADS / WAVE) - 1) / (AGH_FS_THREADS / WAVE);
    const uint32_t blocks = want > 16384 ? 16384u : (uint32_t)want;    // grid-stride beyond that
    if (a.fs_fast) {
        // unit costs, one-byte delimiter outside every pattern class (the host checked): the lean hot
        // kernel + the exact replay of the pieces it listed
        const bool leanv = a.mk.hashset != nullptr;
        // text streams per lane by the pattern's length (32-bit words): three for m <= 10, two for m <= 16
        // (a.fs_streams: 0 = by length; 1..3 caps it -- A/B, tests)
        int ns = 1;
        if (sizeof(WT) == 4) ns = a.q.m <= 10 ? 3 : (a.q.m <= 16 ? 2 : 1);
        if (a.fs_streams && (int)a.fs_streams < ns) ns = (int)a.fs_streams;
        const uint64_t units = (n_tiles + (uint64_t)ns - 1) / (uint64_t)ns;
        const uint64_t wantf = (units + (AGH_FF_THREADS / WAVE) - 1) / (AGH_FF_THREADS / WAVE);
        const uint32_t blocksf = wantf > 32768 ? 32768u : (uint32_t)wantf;
#define AGH_FF_LAUNCH(WTT, NSV)                                                                                      \
    hipLaunchKernelGGL((k_fullscan_fast<WTT, K, NSV>), dim3(blocksf), dim3(AGH_FF_THREADS), 0, st, (const uint8_t *)a.text, \
                       a.n, a.q, (const WTT *)a.mask, a.fs_replay, a.fs_tile_cnt, a.mk.counters)
        if constexpr (sizeof(WT) == 4) {
            if (ns == 3) AGH_FF_LAUNCH(uint32_t, 3);
            else if (ns == 2) AGH_FF_LAUNCH(uint32_t, 2);
        }
        if (ns == 1) AGH_FF_LAUNCH(WT, 1);
#undef AGH_FF_LAUNCH
        const uint32_t nt = (uint32_t)n_tiles;
        const uint32_t rblocks = (nt + 3u) / 4u > 16384u ? 16384u : (nt + 3u) / 4u;
        if (leanv)
            hipLaunchKernelGGL((k_fullscan_replay<WT, K, true>), dim3(rblocks), dim3(256), 0, st,
                               (const uint8_t *)a.text, a.n, a.q, (const WT *)a.mask, (const uint64_t *)a.fs_replay,
                               (const uint32_t *)a.fs_tile_cnt, nt, a.strip_prefix, a.wave_prefix, a.n_strips, a.mk);
        else
            hipLaunchKernelGGL((k_fullscan_replay<WT, K, false>), dim3(rblocks), dim3(256), 0, st,
                               (const uint8_t *)a.text, a.n, a.q, (const WT *)a.mask, (const uint64_t *)a.fs_replay,
                               (const uint32_t *)a.fs_tile_cnt, nt, a.strip_prefix, a.wave_prefix, a.n_strips, a.mk);
        return;
    }
#define AGH_FS_LAUNCH(MBV, GENV, LEANV)                                                       \
    hipLaunchKernelGGL((k_fullscan<WT, K, MBV, GENV, LEANV>), dim3(blocks), dim3(AGH_FS_THREADS), \
                       0, st, (const uint8_t *)a.text, a.n, a.q, (const WT *)a.mask,             \
                       a.strip_prefix, a.wave_prefix, a.n_strips, a.mk, a.dbm)
    const bool mbv = a.q.mb != 0, genv = a.general != 0, leanv = a.mk.hashset != nullptr;
    if (leanv) {                                // count-only: hash set of record starts, no census
        if (mbv && genv) AGH_FS_LAUNCH(true, true, true);
        else if (mbv) AGH_FS_LAUNCH(true, false, true);
        else if (genv) AGH_FS_LAUNCH(false, true, true);
        else AGH_FS_LAUNCH(false, false, true);
    } else if (mbv && genv) AGH_FS_LAUNCH(true, true, false);
    else if (mbv) AGH_FS_LAUNCH(true, false, false);
    else if (genv) AGH_FS_LAUNCH(false, true, false);
    else AGH_FS_LAUNCH(false, false, false);
#undef AGH_FS_LAUNCH
}

template <typename WT>
static void dispatch_fullscan(const agh_scan_args &a, hipStream_t st)
{
    switch (a.q.k) {
    case 0: launch_fullscan_t<WT, 0>(a, st); break;
    case 1: launch_fullscan_t<WT, 1>(a, st); break;
    case 2: launch_fullscan_t<WT, 2>(a, st); break;
    case 3: launch_fullscan_t<WT, 3>(a, st); break;
    case 4: launch_fullscan_t<WT, 4>(a, st); break;
    case 5: launch_fullscan_t<WT, 5>(a, st); break;
    case 6: launch_fullscan_t<WT, 6>(a, st); break;
    case 7: launch_fullscan_t<WT, 7>(a, st); break;
    case 8: launch_fullscan_t<WT, 8>(a, st); break;
    default: break;
    }
}

// Compiled twice (32-bit state words: m <= 32, 64-bit: m <= 64; -DAGH_FS_WIDE=0/1) so that the two
// halves of the instantiation set build in parallel.
#if AGH_FS_WIDE
void agh_launch_fullscan_wide(const agh_scan_args &a, hipStream_t st) { dispatch_fullscan<uint64_t>(a, st); }
#else
void agh_launch_fullscan_wide(const agh_scan_args &a, hipStream_t st);
void agh_launch_fullscan(const agh_scan_args &a, hipStream_t st)
{
    if (a.wide) agh_launch_fullscan_wide(a, st);
    else dispatch_fullscan<uint32_t>(a, st);
}
#endif
