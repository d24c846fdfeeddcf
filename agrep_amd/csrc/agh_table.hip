// agh_table.hip -- the table engine: asearch.c's recurrence run literally on the reference's own
// maskgen tables (Mask[], Init[0], Init1, NO_ERR_MASK, endposition, D_endpos; right-shifting
// words, position 1 at the MSB side, agrep.c:281-282), for the part of the pattern language
// whose state is not a function of the last m+k+1 bytes: '#' wildcards and -p (sticky positions
// through Init1, maskgen.c:231-232), ';' AND / ',' OR (several end bits, verdict taken when the
// record closes, asearch.c:119-128).
//
// Parallel over records instead of over bytes: a lane owns the records whose first byte lies
// in its 1 KiB chunk and runs each of them from its start (reset + re-fed delimiter,
// asearch.c:175-186) to the delimiter that closes it, wherever that is.  Record numbers come
// from the same delimiter census the full scan uses (count-only scans need neither).
// Delimiters of several bytes (and letters under -i): the tables carry the delimiter in their first
// D_length positions (preproce.c:181-224) and level 0 finds its leftmost non-overlapping occurrences
// -- but only for a run that started in front of them.  So WHERE records end is read from the
// delimiter bitmap the other engines use; a lane that learns of a boundary its own state missed takes
// the state a boundary leaves (TableAutomaton::force_boundary) and is exact from there.  k_tablescan
// only; the fast form below keeps to one byte.
#include <type_traits>

#include "agh_verify_inl.h"

#define AGH_TS_CHUNK 256u       // k_unmatched: bytes per lane

template <int K>
struct TableAutomaton {
    uint32_t B[K + 1];

    __device__ __forceinline__ void reset(const agh_dev_tables &T)
    {
#pragma unroll
        for (int e = 0; e <= K; ++e) B[e] = T.Init0;           // asearch.c:63-64
    }
    // One text byte.  Returns bit 0 = the byte closed a record, bit 1 = that record matched.
    __device__ __forceinline__ uint32_t feed(uint32_t CM, const agh_dev_tables &T)
    {
        uint32_t A[K + 1];
        A[0] = ((B[0] >> 1) & CM) | (T.Init1 & B[0]);          // asearch.c:94-116
#pragma unroll
        for (int e = 1; e <= K; ++e)
            A[e] = ((B[e] >> 1) & CM) | (T.Init1 & B[e]) | B[e - 1] |
                   (((A[e - 1] | B[e - 1]) >> 1) & T.NO_ERR);
        uint32_t ret = 0;
        if (A[0] & T.D_endpos) {                               // asearch.c:119
            const uint32_t r1 = A[K] & T.endposition;
            const bool hit = T.AND ? (r1 == T.endposition) : (r1 != 0u);   // asearch.c:128
            ret = 1u | (hit ? 2u : 0u);
            // asearch.c:175-186: every level back to Init[0], the same byte consumed again,
            // level 0 masked so that the delimiter is not detected twice
            A[0] = (((T.Init0 >> 1) & CM) | (T.Init0 & T.Init1)) & T.D_Mask;
#pragma unroll
            for (int e = 1; e <= K; ++e)
                A[e] = ((T.Init0 >> 1) & CM) | (T.Init1 & T.Init0) | T.Init0 |
                       (((A[e - 1] | T.Init0) >> 1) & T.NO_ERR);
        }
#pragma unroll
        for (int e = 0; e <= K; ++e) B[e] = A[e];
        return ret;
    }
    // The same with edit costs (-I# -S# -D#, asearch1.c:88-97 on these tables): level e takes the
    // insertion from level e - ci, the substitution from e - cs, the deletion from the NEW level
    // e - cd; source levels below 0 are asearch1's zero words (A[k], B[k] for k < D).
    __device__ __forceinline__ uint32_t feed_costs(uint32_t CM, const agh_dev_tables &T, uint32_t ci,
                                                   uint32_t cs, uint32_t cd)
    {
        uint32_t A[K + 1];
        A[0] = ((B[0] >> 1) & CM) | (T.Init1 & B[0]);
#pragma unroll
        for (int e = 1; e <= K; ++e) {
            uint32_t ins = 0, via = 0;
#pragma unroll
            for (int s = 0; s < e; ++s) {
                const uint32_t d = (uint32_t)(e - s);
                if (d == ci) ins = B[s];
                if (d == cs) via |= B[s];
                if (d == cd) via |= A[s];
            }
            A[e] = ((B[e] >> 1) & CM) | (T.Init1 & B[e]) | ins | ((via >> 1) & T.NO_ERR);
        }
        uint32_t ret = 0;
        if (A[0] & T.D_endpos) {
            const uint32_t r1 = A[K] & T.endposition;
            const bool hit = T.AND ? (r1 == T.endposition) : (r1 != 0u);
            ret = 1u | (hit ? 2u : 0u);
            // asearch1.c:150-159: all levels Init[0], the byte again, level 0 masked
            A[0] = (((T.Init0 >> 1) & CM) | (T.Init0 & T.Init1)) & T.D_Mask;
#pragma unroll
            for (int e = 1; e <= K; ++e) {
                uint32_t ins = 0, via = 0;
#pragma unroll
                for (int s = 0; s < e; ++s) {
                    const uint32_t d = (uint32_t)(e - s);
                    if (d == ci) ins = T.Init0;
                    if (d == cs) via |= T.Init0;
                    if (d == cd) via |= A[s];
                }
                A[e] = ((T.Init0 >> 1) & CM) | (T.Init1 & T.Init0) | ins | ((via >> 1) & T.NO_ERR);
            }
        }
#pragma unroll
        for (int e = 0; e <= K; ++e) B[e] = A[e];
        return ret;
    }
    template <bool COSTS>
    __device__ __forceinline__ uint32_t feed_q(uint32_t CM, const agh_dev_tables &T, const agh_dev_query &q)
    {
        if (COSTS) return feed_costs(CM, T, q.ci, q.cs, q.cd);
        return feed(CM, T);
    }
    // The state a record boundary at a byte with mask CM leaves (the reset branch of feed / feed_costs),
    // for a lane that has just learnt from the delimiter bitmap that a delimiter of several bytes ended
    // here: it started inside that delimiter, so its own level 0 could not know.
    template <bool COSTS>
    __device__ __forceinline__ void force_boundary(uint32_t CM, const agh_dev_tables &T, const agh_dev_query &q)
    {
        B[0] = (((T.Init0 >> 1) & CM) | (T.Init0 & T.Init1)) & T.D_Mask;
#pragma unroll
        for (int e = 1; e <= K; ++e) {
            if (COSTS) {
                uint32_t ins = 0, via = 0;
#pragma unroll
                for (int s = 0; s < e; ++s) {
                    const uint32_t d = (uint32_t)(e - s);
                    if (d == q.ci) ins = T.Init0;
                    if (d == q.cs) via |= T.Init0;
                    if (d == q.cd) via |= B[s];
                }
                B[e] = ((T.Init0 >> 1) & CM) | (T.Init1 & T.Init0) | ins | ((via >> 1) & T.NO_ERR);
            } else {
                B[e] = ((T.Init0 >> 1) & CM) | (T.Init1 & T.Init0) | T.Init0 |
                       (((B[e - 1] | T.Init0) >> 1) & T.NO_ERR);
            }
        }
    }
};

// Text feeding as in k_fullscan (agh_fullscan.hip): a lane's chunk is one 1 KiB census strip, a wave
// owns a 64 KiB tile and gathers the next 64 bytes of all 64 chunks per round (64-byte segments,
// four dwordx4 loads) through a 5 KiB per-wave LDS ring.  A lane runs the records that START in its
// chunk, so after the 16 rounds it walks on alone (plain loads, the neighbour's chunk is in L2) to
// the delimiter that closes its last record.
// LEAN (count-only): no census pass in front -- a record is identified by the offset of its first
// byte, which the owning lane knows exactly, and goes into the hash set of record starts.
template <int K, bool LEAN, bool COSTS, bool MB>
__global__ __launch_bounds__(AGH_FS_THREADS) void k_tablescan(
    const uint8_t *__restrict__ text, uint64_t n, agh_dev_query q, agh_dev_tables T,
    const uint32_t *__restrict__ mask_g, const uint32_t *__restrict__ strip_prefix,
    const uint32_t *__restrict__ wave_prefix, uint32_t n_strips, agh_marks mk,
    const uint64_t *__restrict__ dbm)
{
    // q.mb (delimiters of several bytes, letters under -i): WHERE a record ends comes from the delimiter
    // bitmap (the leftmost non-overlapping occurrences, which is also what level 0 of the automaton
    // selects when it runs from the start of the text) -- a lane that starts inside a delimiter cannot
    // know from its own state.  What the boundary does to the state stays the automaton's business.
    constexpr bool mb = MB;
    __shared__ uint32_t lmask[256];
    __shared__ __attribute__((aligned(16))) uint8_t ring_all[(AGH_FS_THREADS / WAVE) * WAVE * AGH_FS_ROW];
    lmask[threadIdx.x] = mask_g[threadIdx.x];
    __syncthreads();
    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    uint8_t *ring = ring_all + wib * (WAVE * AGH_FS_ROW);
    const uint64_t tile_bytes = (uint64_t)WAVE * AGH_FS_CHUNK;
    const uint64_t n_tiles = (n + tile_bytes - 1) / tile_bytes;
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
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
                g[i] = a < n16 ? ld_stream(reinterpret_cast<const uint4 *>(text + a))
                               : make_uint4(fill4, fill4, fill4, fill4);
            }
        };
        uint4 g[4];
        gather(0, g);

        // number of the record that contains byte cs (chunk == strip)
        uint32_t rec = 0;
        if (mine && !LEAN) {
            const uint64_t strip = cs >> AGH_STRIP_SHIFT;
            rec = strip < n_strips ? wave_prefix[strip / AGH_WAVE_STRIPS] + strip_prefix[strip] : 0u;
        }
        uint64_t rstart = 0;                    // first byte of the record being run (once active)
        TableAutomaton<K> A;
        A.reset(T);
        bool active = false, done = !mine;
        if (mine && cs == 0) {
            (void)A.template feed_q<COSTS>(lmask[q.head_byte], T, q);   // asearch.c:69-78 (never a hit: host check)
            active = true;
        }
        auto step = [&](uint32_t c, uint64_t pos, uint32_t dbit) {
            uint32_t r = A.template feed_q<COSTS>(lmask[c], T, q);
            if (mb) {
                if (dbit && !(r & 1u)) {        // (only before my first boundary: I started inside the delimiter)
                    A.template force_boundary<COSTS>(lmask[c], T, q);
                    r = 1u;
                } else if (!dbit) {
                    r = 0u;                     // (an occurrence my untrusted state selected differently)
                }
            }
            if (r & 1u) {
                if (active && (r & 2u)) {
                    if (LEAN) lean_insert(mk, rstart); else mark_record(mk, rec, pos);
                }
                if (pos >= ce) done = true;     // that closed my last record
                active = true;
                ++rec;
                rstart = pos + 1;
            }
        };
        for (uint32_t r = 0; r < AGH_FS_CHUNK / AGH_FS_ROUND; ++r) {
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i)
                *reinterpret_cast<uint4 *>(ring_w + 16u * i * AGH_FS_ROW) = g[i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            uint4 v[4];
#pragma unroll
            for (uint32_t p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const uint4 *>(ring_r + 16u * p);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (r + 1 < AGH_FS_CHUNK / AGH_FS_ROUND) gather(r + 1, g);
#pragma unroll
            for (uint32_t p = 0; p < 4; ++p) {
                const uint32_t off = r * AGH_FS_ROUND + 16u * p;
                const uint32_t dws[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
                const uint32_t d16 = (mb && off < len) ? (uint32_t)dbm_bits64(dbm, cs + off) & 0xffffu : 0u;
                if (off + 16u <= len) {
#pragma unroll
                    for (uint32_t b = 0; b < 16; ++b)
                        step((dws[b >> 2] >> (8u * (b & 3u))) & 0xffu, cs + off + b, (d16 >> b) & 1u);
                } else if (off < len) {
                    for (uint32_t b = 0; off + b < len; ++b)
                        step((dws[b >> 2] >> (8u * (b & 3u))) & 0xffu, cs + off + b, (d16 >> b) & 1u);
                }
            }
        }
        if (!active) done = true;               // no record starts in my chunk
        // on alone to the delimiter that closes my last record (ce is 16-byte aligned unless ce == n)
        for (uint64_t p0 = ce; !done && p0 < n; p0 += 16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(text + p0);
            const uint32_t dws[4] = {v.x, v.y, v.z, v.w};
            const uint32_t d16 = mb ? (uint32_t)dbm_bits64(dbm, p0) & 0xffffu : 0u;
            for (uint32_t b = 0; b < 16 && !done && p0 + b < n; ++b)
                step((dws[b >> 2] >> (8u * (b & 3u))) & 0xffu, p0 + b, (d16 >> b) & 1u);
        }
        if (!done && active && q.tail_virtual) {            // asearch.c:87-91: the delimiter appended at EOF
            uint32_t r = 0;
            for (uint32_t jd = 0; jd < q.dlen && !(r & 1u); ++jd)
                r = A.template feed_q<COSTS>(lmask[q.dbytes[jd]], T, q);
            if ((r & 3u) == 3u) {
                if (LEAN) lean_insert(mk, rstart); else mark_record(mk, rec, n);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// fast form: branch-free hot loop + exact replay of the records that may match
// ---------------------------------------------------------------------------------------
// k_tablescan spends 23 VALU instructions per byte at k = 0 (profiles/r03_pmc_tablescan.json: issue
// bound) on a loop that is mostly bookkeeping: is this byte a delimiter, does a record of mine close
// here, which record is it.  Here the hot kernel only advances the reference's recurrence --
// the reset at a delimiter is a select against the constant state a reset leaves (RF), read together
// with the byte's mask as (CM, kill) from LDS -- and ORs up "the top level holds an end bit at a
// record end".  A 16-byte piece where that happened goes to the tile's replay list; k_table_replay
// then runs asearch.c's recurrence over exactly those records, from their first byte.  Ownership is
// as in k_tablescan: a lane speaks for the records whose preceding delimiter lies in its 1 KiB chunk
// (its state is only trusted behind the first delimiter it has seen) and walks on past the chunk's
// end to the first delimiter there.  ';' AND patterns are flagged like ',' OR ones (any end bit): the
// replay decides.  Unit costs, one-byte delimiter.
template <int K>
struct TableFast {
    uint32_t B[K + 1];
    // one byte: (CM, kb) = (mask of the byte, 0 for the delimiter else ~0) -> A[K] BEFORE the reset
    // COSTS (round 5): asearch1.c's levels by accumulated cost (TableAutomaton::feed_costs) in the same branch-free
    // form -- the sources of a level are selected by the three costs, the reset stays a select
    template <bool COSTS = false>
    __device__ __forceinline__ uint32_t feed(uint32_t CM, uint32_t kb, const agh_dev_tables &T,
                                             const uint32_t (&RF)[K + 1], uint32_t ci = 1u, uint32_t cs = 1u,
                                             uint32_t cd = 1u)
    {
        uint32_t A[K + 1];
        A[0] = ((B[0] >> 1) & CM) | (T.Init1 & B[0]);
#pragma unroll
        for (int e = 1; e <= K; ++e) {
            if (COSTS) {
                uint32_t ins = 0, via = 0;
#pragma unroll
                for (int s = 0; s < e; ++s) {
                    const uint32_t d = (uint32_t)(e - s);
                    if (d == ci) ins = B[s];
                    if (d == cs) via |= B[s];
                    if (d == cd) via |= A[s];
                }
                A[e] = ((B[e] >> 1) & CM) | (T.Init1 & B[e]) | ins | ((via >> 1) & T.NO_ERR);
            } else {
                A[e] = ((B[e] >> 1) & CM) | (T.Init1 & B[e]) | B[e - 1] |
                       (((A[e - 1] | B[e - 1]) >> 1) & T.NO_ERR);
            }
        }
        const uint32_t top = A[K];
#pragma unroll
        for (int e = 0; e <= K; ++e) B[e] = (A[e] & kb) | (RF[e] & ~kb);     // v_bfi: the reset is a select
        return top;
    }
};

// the state asearch.c:175-186 leaves behind a delimiter (all levels Init[0], the delimiter consumed
// again, level 0 masked)
template <int K, bool COSTS = false>
__device__ __forceinline__ void table_reset_state(const agh_dev_tables &T, uint32_t CMd, uint32_t (&RF)[K + 1],
                                                  uint32_t ci = 1u, uint32_t cs = 1u, uint32_t cd = 1u)
{
    RF[0] = (((T.Init0 >> 1) & CMd) | (T.Init0 & T.Init1)) & T.D_Mask;
#pragma unroll
    for (int e = 1; e <= K; ++e) {
        if (COSTS) {                            // asearch1.c:150-159 (TableAutomaton::feed_costs' reset branch)
            uint32_t ins = 0, via = 0;
#pragma unroll
            for (int s = 0; s < e; ++s) {
                const uint32_t d = (uint32_t)(e - s);
                if (d == ci) ins = T.Init0;
                if (d == cs) via |= T.Init0;
                if (d == cd) via |= RF[s];
            }
            RF[e] = ((T.Init0 >> 1) & CMd) | (T.Init1 & T.Init0) | ins | ((via >> 1) & T.NO_ERR);
        } else {
            RF[e] = ((T.Init0 >> 1) & CMd) | (T.Init1 & T.Init0) | T.Init0 | (((RF[e - 1] | T.Init0) >> 1) & T.NO_ERR);
        }
    }
}

// A lane's chunk is 4 KiB here (k_tablescan: 1 KiB): the walk past the chunk's end costs a wave the
// LONGEST of its 64 lanes' walks (~300 bytes with 80-byte records), a third of a 1 KiB chunk's work
// but a twelfth of this one's.  Record numbers do not depend on it (the replay takes them from the
// census strips).
// Round 4: the chunk is a launch parameter (1, 2, 4 or -- round 6, from 8 GiB on -- 8 KiB: agh_tf_chunk_for picks it
// by the size of the text -- a wave walks its chunk serially, 0.28 ms at 4 KiB, so small texts take small chunks); a
// tile is 64 chunks and its replay slice holds one entry per 256 bytes of text whatever the chunk.
#define AGH_TF_SLICE_OF(chunk) ((chunk) / 4u)      // replay entries per tile (64 chunks)

// MB (round 5): delimiters of several bytes / a folded letter -- WHERE a record ends is read from the delimiter-end
// bitmap (16 bits per 16-byte piece, one 2-byte load), so the reset select is driven by the bitmap's bit instead of
// the byte's table entry; what a boundary does to the state (RF: every level Init[0], the delimiter's last byte
// consumed again, level 0 masked) is the one-byte case's.  A lane that starts inside a delimiter needs no special
// case here: its state is untrusted until the first bitmap bit anyway.
// Count-only scans of patterns without ';' (round 6): a flagged piece in which exactly ONE record ends needs no
// replay -- the fast recurrence is asearch.c's from the first delimiter a stream has seen (the reset leaves a state
// that does not depend on what came before), so "an end bit on the top level at a trusted record end" IS that
// record's verdict (asearch.c:119-128 without AND), and a record ends in one piece only: it is counted in a
// register here.  Pieces with several record ends and the piece at the text's end still go to the replay, which
// walks every record that ends in a listed piece -- a piece is either counted here or listed, never both.
// (With 1.7 KB records the replay -- one lane per record, from its first byte -- was 2.3 of 11 ms.)
template <bool MB>
__device__ __forceinline__ bool tf_one_end(uint4 v, uint32_t nb, uint32_t d16, uint32_t delim)
{
    if (MB) return __popc(d16 & ((1u << nb) - 1u)) == 1;
    return delims_in(mask_tail(v, (int)nb, (~delim & 0xffu) * 0x01010101u), (delim & 0xffu) * 0x01010101u) == 1u;
}
__device__ __forceinline__ void tf_add_direct(uint32_t ndirect, uint32_t *__restrict__ counters)
{
    if (!__ballot(ndirect != 0u)) return;
    const uint32_t t = wave_sum_to_lane63(ndirect);
    if (lane_id() == 63) {
        atomicAdd(&counters[AGH_C_MATCHED], t);
        counters[AGH_C_ANYHIT] = 1u;
    }
}

// The walk past a chunk's end costs a wave the LONGEST of its lanes' walks: ~5 x the mean record with records of
// random length (1.7 KB records: 9 KB behind every 4 KiB chunk, 8.1 of 11 ms -- profiles/r06_table_delims.log).  Once
// only `at` lanes still have an open record the wave hands those over -- position, tile and the K + 1 state words of
// the stream -- and goes on to its next tile; k_table_cont walks them to their delimiters one lane per record, every
// lane taking the next entry as soon as its record has closed.
// (Not before the walk is AGH_TF_CONT_MIN bytes long: with 80-byte lines the longest of 128 walks ends at ~400 bytes,
// the hand-over and the second kernel cost more than those last rounds.)
#ifndef AGH_TF_CONT_MIN
#define AGH_TF_CONT_MIN 512u
#endif
struct tf_cont_args {
    uint4 *ent;
    uint32_t cap, at;
};
template <int K>
__device__ __forceinline__ void tf_cont_push(const tf_cont_args &c, uint32_t *__restrict__ counters, bool open,
                                             uint64_t pos, uint32_t tile, const uint32_t (&S)[K + 1])
{
    const uint64_t m = __ballot(open);
    if (!open) return;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    uint32_t base = 0;
    if (rank == 0) base = atomicAdd(&counters[AGH_C_CONT_N], (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);         // (the first active lane is the one of rank 0)
    const uint32_t at = base + rank;
    if (at >= c.cap) {
        counters[AGH_C_OVERFLOW] = 1u;
        return;
    }
    uint32_t w[12];
    w[0] = (uint32_t)pos;
    w[1] = (uint32_t)(pos >> 32);
    w[2] = tile;
#pragma unroll
    for (int e = 0; e < 9; ++e) w[3 + e] = e <= K ? S[e <= K ? e : 0] : 0u;
    c.ent[3u * at] = make_uint4(w[0], w[1], w[2], w[3]);
    c.ent[3u * at + 1u] = make_uint4(w[4], w[5], w[6], w[7]);
    c.ent[3u * at + 2u] = make_uint4(w[8], w[9], w[10], w[11]);
}

template <int K, bool COSTS, bool MB>
__global__ __launch_bounds__(AGH_FS_THREADS) void k_tablescan_fast(
    const uint8_t *__restrict__ text, uint64_t n, agh_dev_query q, agh_dev_tables T,
    const uint32_t *__restrict__ mask_g, uint64_t *__restrict__ replay,
    uint32_t *__restrict__ tile_cnt, uint32_t *__restrict__ counters, uint32_t tf_chunk,
    const uint16_t *__restrict__ dbm16, uint32_t direct, tf_cont_args cont)
{
    const uint32_t tf_slice = AGH_TF_SLICE_OF(tf_chunk);
    uint32_t ndirect = 0;                       // (direct) records counted here: see tf_one_end
    struct MK { uint32_t cm, kb; };
    __shared__ MK tab[256];
    __shared__ __attribute__((aligned(16))) uint8_t ring_all[(AGH_FS_THREADS / WAVE) * WAVE * AGH_FS_ROW];
    tab[threadIdx.x].cm = mask_g[threadIdx.x];
    tab[threadIdx.x].kb = (!MB && threadIdx.x == q.delim) ? 0u : ~0u;
    __syncthreads();
    uint32_t RF[K + 1];
    const uint32_t ci = q.ci, cs_ = q.cs, cd = q.cd;
    table_reset_state<K, COSTS>(T, mask_g[q.delim & 0xffu], RF, ci, cs_, cd);
    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    uint8_t *ring = ring_all + wib * (WAVE * AGH_FS_ROW);
    const uint64_t tile_bytes = (uint64_t)WAVE * tf_chunk;
    const uint64_t n_tiles = (n + tile_bytes - 1) / tile_bytes;
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    const uint32_t seg_lo = (uint32_t)lane >> 2, part = (uint32_t)lane & 3u;
    uint8_t *ring_w = ring + seg_lo * AGH_FS_ROW + part * 16u;
    const uint8_t *ring_r = ring + (uint32_t)lane * AGH_FS_ROW;

    for (uint64_t tile = (uint64_t)blockIdx.x * (AGH_FS_THREADS / WAVE) + wib; tile < n_tiles;
         tile += (uint64_t)gridDim.x * (AGH_FS_THREADS / WAVE)) {
        const uint64_t t0 = tile * tile_bytes;
        const uint64_t cs = t0 + (uint64_t)lane * tf_chunk;
        uint64_t ce = cs + tf_chunk;
        if (ce > n) ce = n;
        const uint32_t len = cs < n ? (uint32_t)(ce - cs) : 0u;
        auto gather = [&](uint32_t r, uint4 (&g)[4]) {
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint64_t a = t0 + (uint64_t)(seg_lo + 16u * i) * tf_chunk + r * AGH_FS_ROUND + part * 16u;
                g[i] = a < n16 ? *reinterpret_cast<const uint4 *>(text + a) : make_uint4(fill4, fill4, fill4, fill4);
            }
        };
        uint4 g[4];
        gather(0, g);
        TableFast<K> A;
#pragma unroll
        for (int e = 0; e <= K; ++e) A.B[e] = T.Init0;
        // trusted: the lane has seen the start of the record it is in (the text's first byte, or a
        // delimiter inside its chunk)
        uint32_t trusted = 0u, cnt = 0;
        if (cs == 0 && len) {                   // the virtual head byte, asearch.c:69-78
            const MK e = tab[q.head_byte & 0xffu];
            (void)A.template feed<COSTS>(e.cm, e.kb, T, RF, ci, cs_, cd);
            trusted = ~0u;
        }
        // 16 bytes: -> "some trusted record end in the piece shows an end bit on the top level"
        uint32_t dseen = 0;                     // ~0 once a delimiter went through piece()
        auto piece = [&](uint4 v, uint32_t nbytes, uint32_t d16) -> uint32_t {
            const uint32_t dws[4] = {v.x, v.y, v.z, v.w};
            const uint32_t nd16 = ~d16;         // (MB) bit b clear: a delimiter ends at byte b
            uint32_t flag = 0;
#pragma unroll
            for (uint32_t b = 0; b < 16; ++b) {
                if (b < nbytes) {
                    const MK e = tab[(dws[b >> 2] >> (8u * (b & 3u))) & 0xffu];
                    const uint32_t kb = MB ? (uint32_t)__builtin_amdgcn_sbfe((int)nd16, b, 1u) : e.kb;
                    const uint32_t top = A.template feed<COSTS>(e.cm, kb, T, RF, ci, cs_, cd);
                    flag |= top & ~kb & trusted;
                    trusted |= ~kb;
                    dseen |= ~kb;
                }
            }
            return flag & T.endposition;
        };
        auto emit = [&](bool f, uint64_t pos) {  // (uniform call sites: every lane of the wave comes here)
            const uint64_t fm = __ballot(f);
            if (!fm) return;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
            if (f) {
                const uint32_t at = cnt + rank;
                if (at < tf_slice) replay[tile * tf_slice + at] = pos;
                else counters[AGH_C_OVERFLOW] = 1u;
            }
            cnt += (uint32_t)__popcll(fm);
        };
        // (MB) the delimiter-end bits of two rounds (128 text bytes): one 16-byte load per lane, two rounds ahead.
        // (8-byte loads per round cost as much L1 time as the text itself -- 64 lanes, 64 cache lines per
        // instruction: 5.9 ms on 4 GiB; 16 bytes per two rounds: see profiles/r05_perf_table_mb.log)
        auto dload = [&](uint64_t at) -> uint4 {
            return (MB && at < n) ? *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(dbm16) + (at >> 3))
                                  : make_uint4(0, 0, 0, 0);
        };
        uint4 dv = make_uint4(0, 0, 0, 0), dnext = dload(cs);
        for (uint32_t r = 0; r < tf_chunk / AGH_FS_ROUND; ++r) {
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i)
                *reinterpret_cast<uint4 *>(ring_w + 16u * i * AGH_FS_ROW) = g[i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (MB && !(r & 1u)) {
                dv = dnext;
                dnext = dload(cs + (uint64_t)(r + 2) * AGH_FS_ROUND);
            }
            const uint64_t dcur = (r & 1u) ? ((uint64_t)dv.w << 32 | dv.z) : ((uint64_t)dv.y << 32 | dv.x);
            if (r + 1 < tf_chunk / AGH_FS_ROUND) gather(r + 1, g);
#pragma unroll 1
            for (uint32_t p = 0; p < 4; ++p) {
                const uint32_t off = r * AGH_FS_ROUND + 16u * p;
                const uint4 v = *reinterpret_cast<const uint4 *>(ring_r + 16u * p);
                const uint32_t nb = off + 16u <= len ? 16u : (off < len ? len - off : 0u);
                const uint32_t d16 = (uint32_t)(dcur >> (16u * p)) & 0xffffu;
                uint32_t flag = nb ? piece(v, nb, d16) : 0u;
                const bool last = cs + off + 16u >= n;
                if (direct && __ballot(flag != 0u) && flag && !last && tf_one_end<MB>(v, nb, d16, q.delim)) {
                    ++ndirect;
                    flag = 0u;
                }
                // the piece that holds the last byte of the text: the appended delimiter is the replay's
                if (nb && last && trusted) flag = 1u;
                emit(flag != 0u, cs + off);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // all reads done before the next round's writes
            __builtin_amdgcn_wave_barrier();
        }
        // on alone to the delimiter that closes my last record (ce is 16-byte aligned unless ce == n)
        // (a record that starts exactly at the chunk's end is mine as well: I saw the delimiter in
        // front of it, the next lane trusts its state only behind the first delimiter IT sees)
        bool open = len == tf_chunk && trusted != 0u && ce < n;
        dseen = 0;
        // (the next piece is loaded while this one runs: with records of a KiB or two the walk is most of the kernel)
        uint4 vw = make_uint4(0, 0, 0, 0);
        uint32_t dw16 = 0;
        if (open) {
            vw = *reinterpret_cast<const uint4 *>(text + ce);
            dw16 = MB ? (uint32_t)dbm16[ce >> 4] : 0u;
        }
        for (uint64_t p0 = ce;; p0 += 16) {
            const uint64_t om = __ballot(open);
            if (!om) break;
            if (cont.at && p0 - ce >= AGH_TF_CONT_MIN && (uint32_t)__popcll(om) <= cont.at) {   // the stragglers: k_table_cont's
                tf_cont_push<K>(cont, counters, open, p0, (uint32_t)tile, A.B);
                break;
            }
            // whole 16-byte pieces until one holds a delimiter: what it flags behind that delimiter
            // belongs to the next lane, which flags it as well -- the replay does not mind
            uint32_t flag = 0;
            const bool mine = open;
            if (open) {
                const uint4 v = vw;
                const uint32_t d16 = dw16;
                if (p0 + 16 < n) {
                    vw = *reinterpret_cast<const uint4 *>(text + p0 + 16);
                    dw16 = MB ? (uint32_t)dbm16[(p0 + 16) >> 4] : 0u;
                }
                const uint32_t nb = p0 + 16 <= n ? 16u : (uint32_t)(n - p0);
                flag = piece(v, nb, d16);
                if (direct && flag && p0 + 16 < n && tf_one_end<MB>(v, nb, d16, q.delim)) {
                    ++ndirect;
                    flag = 0u;
                }
                if (dseen) open = false;
                else if (p0 + 16 >= n) {         // the text ends inside my record: the last piece
                    flag = 1u;
                    open = false;
                }
            }
            // (into MY tile's list, whatever tile the piece lies in: the replay only needs the position)
            emit(mine && flag != 0u, p0);
        }
        if (lane == 0) tile_cnt[tile] = cnt < tf_slice ? cnt : tf_slice;
    }
    tf_add_direct(ndirect, counters);
}

// ---------------------------------------------------------------------------------------
// fast form, two streams per lane (tables of M <= 15 positions)
// ---------------------------------------------------------------------------------------
// k_tablescan_fast issues ~10 VALU instructions per byte and most of them work on words whose upper half
// is idle: the recurrence only needs bits 0..M (bit M = the always-one bit above position 1 that Init[0]
// and Init1 keep set, maskgen.c:224).  With M + 1 <= 16 two chunks share every state word -- chunk c of a
// 256 KiB tile in the low halves, chunk c of the NEXT tile in the high halves -- and one pass of the
// recurrence, the reset select and the flag logic serves two text bytes.  The shift brings bit 16 into
// bit 15: with M = 15 that is the always-one bit (already set), with M < 15 a bit above M, and nothing
// above M reaches the positions (every constant is cut to bits 0..M; the only term that moves bits down,
// ">> 1", moves bit M + 1 into bit M, which is one anyway).  What a lane extracts per byte (table index,
// LDS read) stays per stream; the LDS entry is (mask, kill) in 16 + 16 bits and two v_perm put the two
// streams' halves together.  Tiles, replay lists and counters are the unpacked kernel's: the low stream
// writes tile 2t, the high stream tile 2t + 1 -- k_table_replay does not know.
// ring rows: 64 text bytes + 16 of padding (bank spread for the b128 reads), or AGH_TF2_SWZ: no padding, the
// 16-byte column XORed with bits 1..2 of the row (8 consecutive rows cover the 32 banks once) -- 33 KiB instead
// of 41 KiB per workgroup, i.e. four workgroups per CU instead of three
#ifdef AGH_TF2_SWZ
#define AGH_TF2_ROW 64u
#define AGH_TF2_COL(row, col) (((col) ^ (((row) >> 1) & 3u)) * 16u)
#else
#define AGH_TF2_ROW AGH_FS_ROW
#define AGH_TF2_COL(row, col) ((col) * 16u)
#endif
#ifdef AGH_TF2_OCC
#define AGH_TF2_ATTR __attribute__((amdgpu_waves_per_eu(AGH_TF2_OCC, AGH_TF2_OCC)))
#else
#define AGH_TF2_ATTR
#endif
template <int K, bool COSTS, bool MB>
__global__ __launch_bounds__(AGH_FS_THREADS) AGH_TF2_ATTR void k_tablescan_fast2(
    const uint8_t *__restrict__ text, uint64_t n, agh_dev_query q, agh_dev_tables T,
    const uint32_t *__restrict__ mask_g, uint64_t *__restrict__ replay,
    uint32_t *__restrict__ tile_cnt, uint32_t *__restrict__ counters, uint32_t M, uint32_t n_tiles1,
    uint32_t tf_chunk, const uint16_t *__restrict__ dbm16, uint32_t direct, tf_cont_args cont)
{
    const uint32_t tf_slice = AGH_TF_SLICE_OF(tf_chunk);
    uint32_t ndirect = 0;                       // (direct) records counted here: tf_one_end
    __shared__ uint32_t tab[256];               // mask (bits 0..M-1) | kill << 16 (0 for the delimiter, else 0xffff)
    __shared__ __attribute__((aligned(16))) uint8_t ring_all[(AGH_FS_THREADS / WAVE) * 2 * WAVE * AGH_TF2_ROW];
    const uint32_t keep = (2u << M) - 1u;       // bits 0..M
    tab[threadIdx.x] = (mask_g[threadIdx.x] & keep & 0xffffu) | ((!MB && threadIdx.x == q.delim) ? 0u : 0xffff0000u);
    __syncthreads();
    auto dup = [&](uint32_t x) -> uint32_t { x &= keep; return x | (x << 16); };
    agh_dev_tables Tp = T;
    Tp.Init0 = dup(T.Init0);
    Tp.Init1 = dup(T.Init1);
    Tp.NO_ERR = dup(T.NO_ERR);
    Tp.endposition = dup(T.endposition);
    uint32_t RF[K + 1];
    const uint32_t ci = q.ci, cs_ = q.cs, cd = q.cd;        // (COSTS: asearch1.c's levels -- the same bitwise recurrence shape,
    table_reset_state<K, COSTS>(T, mask_g[q.delim & 0xffu], RF, ci, cs_, cd);   //  so the two halves keep out of each other's way)
#pragma unroll
    for (int e = 0; e <= K; ++e) RF[e] = dup(RF[e]);
    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    uint8_t *ring = ring_all + wib * (2 * WAVE * AGH_TF2_ROW);
    const uint64_t tile_bytes = (uint64_t)WAVE * tf_chunk;        // one stream's tile
    const uint64_t n_tiles2 = ((uint64_t)n_tiles1 + 1) / 2;
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    const uint32_t seg_lo = (uint32_t)lane >> 2, part = (uint32_t)lane & 3u;
    uint8_t *ring_w = ring + seg_lo * AGH_TF2_ROW + AGH_TF2_COL(seg_lo, part);     // (rows seg_lo + 16 i: the same swizzle)
    const uint8_t *ring_ra = ring + (uint32_t)lane * AGH_TF2_ROW;
    const uint8_t *ring_rb = ring + (WAVE + (uint32_t)lane) * AGH_TF2_ROW;

    for (uint64_t tile2 = (uint64_t)blockIdx.x * (AGH_FS_THREADS / WAVE) + wib; tile2 < n_tiles2;
         tile2 += (uint64_t)gridDim.x * (AGH_FS_THREADS / WAVE)) {
        const uint64_t t0 = tile2 * 2 * tile_bytes;
        // stream 0 (low halves): chunk `lane` of tile 2 * tile2; stream 1: the same chunk of tile 2 * tile2 + 1
        uint64_t cs[2], ce[2];
        uint32_t len[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            cs[s2] = t0 + (uint64_t)s2 * tile_bytes + (uint64_t)lane * tf_chunk;
            ce[s2] = cs[s2] + tf_chunk;
            if (ce[s2] > n) ce[s2] = n;
            len[s2] = cs[s2] < n ? (uint32_t)(ce[s2] - cs[s2]) : 0u;
        }
        auto gather = [&](uint32_t r, uint32_t first, uint4 (&g)[4]) {      // rows first .. first + 63, 16 apart
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint64_t a = t0 + (uint64_t)(first + seg_lo + 16u * i) * tf_chunk + r * AGH_FS_ROUND + part * 16u;
                g[i] = a < n16 ? *reinterpret_cast<const uint4 *>(text + a) : make_uint4(fill4, fill4, fill4, fill4);
            }
        };
        uint4 ga[4], gb[4];
        gather(0, 0u, ga);
        gather(0, WAVE, gb);
        TableFast<K> A;
#pragma unroll
        for (int e = 0; e <= K; ++e) A.B[e] = Tp.Init0;
        // per half: 0xffff once the stream has seen the start of the record it is in
        uint32_t trusted = 0u, cnt0 = 0, cnt1 = 0;
        if (cs[0] == 0 && len[0]) {             // the virtual head byte, asearch.c:69-78 (stream 1: not trusted yet)
            const uint32_t e = tab[q.head_byte & 0xffu];
            (void)A.template feed<COSTS>(__builtin_amdgcn_perm(e, e, 0x05040100u), __builtin_amdgcn_perm(e, e, 0x07060302u), Tp, RF,
                                         ci, cs_, cd);
            trusted = 0xffffu;
        }
        uint32_t dseen = 0;                     // per half: 0xffff once a delimiter went through piece()
        // 16 bytes of either stream: -> per half, "some trusted record end in the piece shows an end bit on
        // the top level".  live: 0xffff per half for the bytes that exist (FULL: all 16 of both)
        // mode 0: bytes beyond a stream's end are masked (live); 1: all 16 bytes of both streams exist; 2: and
        // every stream of the wave is trusted already (the common case after a record or two): no bookkeeping
        auto piece = [&](uint4 va, uint4 vb, uint32_t nba, uint32_t nbb, uint32_t d16a, uint32_t d16b, auto mode) -> uint32_t {
            constexpr int MODE = decltype(mode)::value;
            const uint32_t da[4] = {va.x, va.y, va.z, va.w}, db[4] = {vb.x, vb.y, vb.z, vb.w};
            // (MB: the kill halves come from the delimiter-end bitmap of either stream's piece; bits behind the text are clear)
            const uint32_t nda = ~d16a, ndb = ~d16b;
            uint32_t flag = 0;
#pragma unroll
            for (uint32_t b = 0; b < 16; ++b) {
                const uint32_t ea = tab[(da[b >> 2] >> (8u * (b & 3u))) & 0xffu];
                const uint32_t eb = tab[(db[b >> 2] >> (8u * (b & 3u))) & 0xffu];
                const uint32_t cm = __builtin_amdgcn_perm(eb, ea, 0x05040100u);
                const uint32_t kb = MB ? __builtin_amdgcn_perm((uint32_t)__builtin_amdgcn_sbfe((int)ndb, b, 1u),
                                                               (uint32_t)__builtin_amdgcn_sbfe((int)nda, b, 1u), 0x05040100u)
                                       : __builtin_amdgcn_perm(eb, ea, 0x07060302u);
                const uint32_t top = A.template feed<COSTS>(cm, kb, Tp, RF, ci, cs_, cd);
                if (MODE == 2) {
                    flag |= top & ~kb;
                } else if (MODE == 1) {
                    flag |= top & ~kb & trusted;
                    trusted |= ~kb;
                    dseen |= ~kb;
                } else {
                    const uint32_t live = (b < nba ? 0xffffu : 0u) | (b < nbb ? 0xffff0000u : 0u);
                    flag |= top & ~kb & trusted & live;
                    trusted |= ~kb & live;
                    dseen |= ~kb & live;
                }
            }
            return flag & Tp.endposition;
        };
        auto emit = [&](bool f, uint64_t pos, uint64_t tile1, uint32_t &cnt) {   // (uniform call sites)
            const uint64_t fm = __ballot(f);
            if (!fm) return;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
            if (f) {
                const uint32_t at = cnt + rank;
                if (at < tf_slice) replay[tile1 * tf_slice + at] = pos;
                else counters[AGH_C_OVERFLOW] = 1u;
            }
            cnt += (uint32_t)__popcll(fm);
        };
        // (MB) the delimiter-end bits of two rounds and either stream: one 16-byte load each, two rounds ahead
        // (k_tablescan_fast has the reason)
        auto dload = [&](uint64_t at) -> uint4 {
            return (MB && at < n) ? *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(dbm16) + (at >> 3))
                                  : make_uint4(0, 0, 0, 0);
        };
        uint4 dva = make_uint4(0, 0, 0, 0), dvb = dva, dna = dload(cs[0]), dnb = dload(cs[1]);
        for (uint32_t r = 0; r < tf_chunk / AGH_FS_ROUND; ++r) {
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                *reinterpret_cast<uint4 *>(ring_w + 16u * i * AGH_TF2_ROW) = ga[i];
                *reinterpret_cast<uint4 *>(ring_w + (WAVE + 16u * i) * AGH_TF2_ROW) = gb[i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (MB && !(r & 1u)) {
                dva = dna;
                dvb = dnb;
                dna = dload(cs[0] + (uint64_t)(r + 2) * AGH_FS_ROUND);
                dnb = dload(cs[1] + (uint64_t)(r + 2) * AGH_FS_ROUND);
            }
            const uint64_t dca = (r & 1u) ? ((uint64_t)dva.w << 32 | dva.z) : ((uint64_t)dva.y << 32 | dva.x);
            const uint64_t dcb = (r & 1u) ? ((uint64_t)dvb.w << 32 | dvb.z) : ((uint64_t)dvb.y << 32 | dvb.x);
            if (r + 1 < tf_chunk / AGH_FS_ROUND) {
                gather(r + 1, 0u, ga);
                gather(r + 1, WAVE, gb);
            }
#pragma unroll 1
            for (uint32_t p = 0; p < 4; ++p) {
                const uint32_t off = r * AGH_FS_ROUND + 16u * p;
                const uint4 va = *reinterpret_cast<const uint4 *>(ring_ra + AGH_TF2_COL((uint32_t)lane, p));
                const uint4 vb = *reinterpret_cast<const uint4 *>(ring_rb + AGH_TF2_COL((uint32_t)lane, p));
                const uint32_t nba = off + 16u <= len[0] ? 16u : (off < len[0] ? len[0] - off : 0u);
                const uint32_t nbb = off + 16u <= len[1] ? 16u : (off < len[1] ? len[1] - off : 0u);
                const bool full = __ballot(nba != 16u || nbb != 16u) == 0ull;
                const bool settled = full && __ballot(trusted != ~0u) == 0ull;
                const uint32_t posa = (uint32_t)(dca >> (16u * p)) & 0xffffu, posb = (uint32_t)(dcb >> (16u * p)) & 0xffffu;
                const uint32_t flag = settled ? piece(va, vb, nba, nbb, posa, posb, std::integral_constant<int, 2>{})
                                      : full  ? piece(va, vb, nba, nbb, posa, posb, std::integral_constant<int, 1>{})
                                              : piece(va, vb, nba, nbb, posa, posb, std::integral_constant<int, 0>{});
                bool fa = nba && (flag & 0xffffu), fb = nbb && (flag >> 16);
                const bool lasta = cs[0] + off + 16u >= n, lastb = cs[1] + off + 16u >= n;
                if (direct && __ballot(fa || fb)) {
                    if (fa && !lasta && tf_one_end<MB>(va, nba, posa, q.delim)) { ++ndirect; fa = false; }
                    if (fb && !lastb && tf_one_end<MB>(vb, nbb, posb, q.delim)) { ++ndirect; fb = false; }
                }
                // the piece that holds the last byte of the text: the appended delimiter is the replay's
                if (nba && lasta && (trusted & 0xffffu)) fa = true;
                if (nbb && lastb && (trusted >> 16)) fb = true;
                emit(fa, cs[0] + off, tile2 * 2, cnt0);
                emit(fb, cs[1] + off, tile2 * 2 + 1, cnt1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // all reads done before the next round's writes
            __builtin_amdgcn_wave_barrier();
        }
        // on alone to the delimiter that closes a stream's last record (k_tablescan_fast: the same walk)
        bool opa = len[0] == tf_chunk && (trusted & 0xffffu) != 0u && ce[0] < n;
        bool opb = len[1] == tf_chunk && (trusted >> 16) != 0u && ce[1] < n;
        dseen = 0;
        // (the next pieces are loaded while these run: with records of a KiB or two the walk is most of the kernel)
        const uint4 vfill = make_uint4(fill4, fill4, fill4, fill4);
        uint4 wa = vfill, wb = vfill;
        uint32_t wda = 0, wdb = 0;
        if (opa) {
            wa = *reinterpret_cast<const uint4 *>(text + ce[0]);
            wda = MB ? (uint32_t)dbm16[ce[0] >> 4] : 0u;
        }
        if (opb) {
            wb = *reinterpret_cast<const uint4 *>(text + ce[1]);
            wdb = MB ? (uint32_t)dbm16[ce[1] >> 4] : 0u;
        }
        for (uint64_t step = 0;; step += 16) {
            const uint64_t om = __ballot(opa || opb);
            if (!om) break;
            if (cont.at && step >= AGH_TF_CONT_MIN && (uint32_t)__popcll(om) <= cont.at) {       // the stragglers: k_table_cont's
                uint32_t S[K + 1];
#pragma unroll
                for (int e = 0; e <= K; ++e) S[e] = A.B[e] & 0xffffu;
                tf_cont_push<K>(cont, counters, opa, ce[0] + step, (uint32_t)(tile2 * 2), S);
#pragma unroll
                for (int e = 0; e <= K; ++e) S[e] = A.B[e] >> 16;
                tf_cont_push<K>(cont, counters, opb, ce[1] + step, (uint32_t)(tile2 * 2 + 1), S);
                break;
            }
            const uint64_t pa = ce[0] + step, pb = ce[1] + step;
            const bool ma = opa, mb = opb;
            uint4 va = vfill, vb = vfill;
            uint32_t nba = 0, nbb = 0, da16 = 0, db16 = 0;
            if (opa) {
                va = wa;
                da16 = wda;
                nba = pa + 16 <= n ? 16u : (uint32_t)(n - pa);
                if (pa + 16 < n) {
                    wa = *reinterpret_cast<const uint4 *>(text + pa + 16);
                    wda = MB ? (uint32_t)dbm16[(pa + 16) >> 4] : 0u;
                }
            }
            if (opb) {
                vb = wb;
                db16 = wdb;
                nbb = pb + 16 <= n ? 16u : (uint32_t)(n - pb);
                if (pb + 16 < n) {
                    wb = *reinterpret_cast<const uint4 *>(text + pb + 16);
                    wdb = MB ? (uint32_t)dbm16[(pb + 16) >> 4] : 0u;
                }
            }
            const uint32_t flag = piece(va, vb, nba, nbb, da16, db16, std::integral_constant<int, 0>{});
            bool fa = ma && (flag & 0xffffu), fb = mb && (flag >> 16);
            if (direct && __ballot(fa || fb)) {
                if (fa && pa + 16 < n && tf_one_end<MB>(va, nba, da16, q.delim)) { ++ndirect; fa = false; }
                if (fb && pb + 16 < n && tf_one_end<MB>(vb, nbb, db16, q.delim)) { ++ndirect; fb = false; }
            }
            if (opa) {
                if (dseen & 0xffffu) opa = false;
                else if (pa + 16 >= n) { fa = true; opa = false; }      // the text ends inside my record
            }
            if (opb) {
                if (dseen >> 16) opb = false;
                else if (pb + 16 >= n) { fb = true; opb = false; }
            }
            emit(fa, pa, tile2 * 2, cnt0);
            emit(fb, pb, tile2 * 2 + 1, cnt1);
        }
        if (lane == 0) {
            tile_cnt[tile2 * 2] = cnt0 < tf_slice ? cnt0 : tf_slice;
            if (tile2 * 2 + 1 < n_tiles1) tile_cnt[tile2 * 2 + 1] = cnt1 < tf_slice ? cnt1 : tf_slice;
        }
    }
    tf_add_direct(ndirect, counters);
}

#ifndef AGH_TF_CONT_PER_LANE
#define AGH_TF_CONT_PER_LANE 4u
#endif
// The open records the fast kernels handed over (tf_cont_push): one lane per record, 16 bytes at a time up to the
// delimiter that closes it, the walk's rules at that piece -- flagged and the only record end of its piece (count-only,
// no ';'): counted here; flagged otherwise, or the piece at the text's end: onto the replay list of the tile the record
// came from.  A lane whose record has closed takes the next entry (one ticket per wave and refill).
template <int K, bool COSTS, bool MB>
__global__ __launch_bounds__(256) void k_table_cont(
    const uint8_t *__restrict__ text, uint64_t n, agh_dev_query q, agh_dev_tables T,
    const uint32_t *__restrict__ mask_g, const uint4 *__restrict__ ent, uint32_t cap,
    uint64_t *__restrict__ replay, uint32_t *__restrict__ tile_cnt, uint32_t *__restrict__ counters,
    uint32_t tf_slice, const uint16_t *__restrict__ dbm16, uint32_t direct, uint32_t M)
{
    __shared__ uint32_t lmask[256];
    lmask[threadIdx.x] = mask_g[threadIdx.x];
    __syncthreads();
    uint32_t total = counters[AGH_C_CONT_N];
    if (total > cap) total = cap;
    // four entries per lane or so: a lane that takes the next entry when its record closes evens out the records'
    // lengths, a lane per entry would leave every wave waiting for its longest one
    if (((uint64_t)blockIdx.x * 256u + (threadIdx.x & ~63u)) * AGH_TF_CONT_PER_LANE >= total && (blockIdx.x | (threadIdx.x >> 6))) return;
    const uint32_t keep = M >= 31u ? ~0u : (2u << M) - 1u;      // (a half of a two-stream word: nothing above bit M)
    const uint32_t ci = q.ci, cs_ = q.cs, cd = q.cd;
    uint32_t RF[K + 1];                                           // (unused: no byte resets the state here)
#pragma unroll
    for (int e = 0; e <= K; ++e) RF[e] = 0u;
    const uint32_t dd = (q.delim & 0xffu) * 0x01010101u;
    uint32_t ndirect = 0;
    bool have = false, dry = false;
    uint64_t pos = 0;
    uint32_t tile = 0;
    TableFast<K> A;
#pragma unroll
    for (int e = 0; e <= K; ++e) A.B[e] = 0u;
    uint4 vcur = make_uint4(0, 0, 0, 0);
    uint32_t dcur = 0;
    uint32_t wnext = 0, wend = 0;                 // (uniform) the wave's block of entries: 64 per ticket
    for (;;) {
        // (a ticket per lane and record made every round of every wave wait for an L2 atomic: 1.7 ms for 260 000 records)
        const uint64_t need = __ballot(!have);
        if (need) {
            if (wnext == wend && !dry) {
                uint32_t b = 0;
                if (lane_id() == 0) b = atomicAdd(&counters[AGH_C_CONT_NEXT], 64u);
                b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                if (b >= total) {
                    dry = true;
                } else {
                    wnext = b;
                    wend = b + 64u < total ? b + 64u : total;
                }
            }
            if (wnext < wend) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                const uint32_t idx = wnext + rank;
                if (!have && idx < wend) {
                    const uint4 e0 = ent[3u * idx], e1 = ent[3u * idx + 1u], e2 = ent[3u * idx + 2u];
                    const uint32_t w[12] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, e2.z, e2.w};
                    pos = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
                    tile = w[2];
#pragma unroll
                    for (int e = 0; e <= K; ++e) A.B[e] = w[3 + e] & keep;
                    have = true;
                    vcur = *reinterpret_cast<const uint4 *>(text + pos);
                    dcur = MB ? (uint32_t)dbm16[pos >> 4] : 0u;
                }
                const uint32_t took = (uint32_t)__popcll(need);
                wnext = wnext + took < wend ? wnext + took : wend;
            }
        }
        if (!__ballot(have)) {
            if (dry) break;
            continue;                           // (the block ran out in this round: the next one brings a new ticket)
        }
        if (have) {
            const uint4 v = vcur;
            uint32_t m16 = dcur;
            if (pos + 16 < n) {                 // the next piece is on its way while this one runs
                vcur = *reinterpret_cast<const uint4 *>(text + pos + 16);
                dcur = MB ? (uint32_t)dbm16[(pos + 16) >> 4] : 0u;
            }
            const uint32_t nb = pos + 16 <= n ? 16u : (uint32_t)(n - pos);
            const uint32_t dws[4] = {v.x, v.y, v.z, v.w};
            if (!MB) {
                m16 = 0;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t x = dws[d] ^ dd;
                    const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
                    m16 |= ((((z >> 7) & 0x01010101u) * 0x10204080u) >> 28) << (4 * d);
                }
            }
            m16 &= (1u << nb) - 1u;
            const uint32_t first = m16 ? (uint32_t)__builtin_ctz(m16) : 16u;
            // all sixteen bytes, whatever closes in between: behind the delimiter (and behind the text's end) the
            // state is not used again, and a lane that stops early saves the wave nothing -- sixteen table reads in
            // flight and no branch instead (a branch per byte made a piece 1.6 us: profiles/r06_ab_table_cont.log)
            uint32_t cm[16];
#pragma unroll
            for (uint32_t b = 0; b < 16; ++b) cm[b] = lmask[(dws[b >> 2] >> (8u * (b & 3u))) & 0xffu];
            uint32_t vtop = 0;
#pragma unroll
            for (uint32_t b = 0; b < 16; ++b) {
                const uint32_t top = A.template feed<COSTS>(cm[b], ~0u, T, RF, ci, cs_, cd);
                vtop = b == first ? top : vtop;
            }
            const bool last = pos + 16 >= n;
            bool list = false;
            if (first < 16u) {                  // my record closes here
                if (vtop & T.endposition) {
                    if (direct && !last && __popc(m16) == 1) ++ndirect;
                    else list = true;
                }
                have = false;
            } else if (last) {                  // the text ends inside my record: the replay's (the appended delimiter)
                list = true;
                have = false;
            } else {
                pos += 16;
            }
            if (list) {
                const uint32_t slot = atomicAdd(&tile_cnt[tile], 1u);
                if (slot < tf_slice) {
                    replay[(uint64_t)tile * tf_slice + slot] = pos;
                } else {
                    counters[AGH_C_OVERFLOW] = 1u;
                    atomicMin(&tile_cnt[tile], tf_slice);
                }
            }
        }
    }
    tf_add_direct(ndirect, counters);
}

// Exact: for every listed piece, every record that ENDS in it (a delimiter inside the piece, or the
// end of the text) is run through asearch.c's recurrence from its first byte.
template <int K, bool LEAN, bool COSTS>
__global__ __launch_bounds__(256) void k_table_replay(
    const uint8_t *__restrict__ text, uint64_t n, agh_dev_query q, agh_dev_tables T,
    const uint32_t *__restrict__ mask_g, const uint64_t *__restrict__ replay,
    const uint32_t *__restrict__ tile_cnt, uint32_t n_tiles,
    const uint32_t *__restrict__ strip_prefix, const uint32_t *__restrict__ wave_prefix,
    uint32_t n_strips, agh_marks mk, uint32_t tf_slice, const uint64_t *__restrict__ dbm, uint32_t group)
{
    __shared__ uint32_t lmask[256];
    lmask[threadIdx.x] = mask_g[threadIdx.x];
    __syncthreads();
    const uint32_t dd = q.delim * 0x01010101u;
    const bool mb = q.mb != 0;                  // record ends from the delimiter-end bitmap (as in k_tablescan)
    // A wave takes the lists of `group` consecutive tiles at once (round 5; 8 by default): a tile lists five pieces or
    // so (a match every 500 lines) and a VALU instruction costs a wave its four cycles whatever the number of busy lanes
    // -- but a wave per SIMD has nothing to hide its latencies behind, so not the fullest waves either
    // (profiles/r05_perf_table_mb.log).
    const uint32_t n_groups = (n_tiles + group - 1u) / group;
    for (uint32_t grp = blockIdx.x * 4u + threadIdx.x / WAVE; grp < n_groups; grp += gridDim.x * 4u) {
        const uint32_t tl = grp * group + (uint32_t)lane_id();
        const uint32_t mine = ((uint32_t)lane_id() < group && tl < n_tiles) ? tile_cnt[tl] : 0u;
        const uint32_t incl = wave_sum_to_lane63(mine);     // inclusive scan over the lanes
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        for (uint32_t e = (uint32_t)lane_id(); e < cnt; e += WAVE) {
            uint32_t tsel = 0, tbase = 0;
#pragma unroll
            for (int t = 0; t < 15; ++t) {
                const uint32_t it = (uint32_t)__builtin_amdgcn_readlane((int)incl, t);
                if (it <= e) { tsel = (uint32_t)t + 1u; tbase = it; }
            }
            const uint64_t P = replay[(uint64_t)(grp * group + tsel) * tf_slice + (e - tbase)];
            uint64_t pend = P + 16;
            if (pend > n) pend = n;
            // records that end in [P, pend): one per delimiter there, plus the open one at the text's end
            uint64_t rs = mb ? lean_record_start_mb(dbm, P, mk)         // start of the record that holds byte P
                             : lean_record_start(text, P, q.delim, mk);
            if (rs == ~0ull) {                  // more than 1 MiB back: this text belongs to the exact kernel
                mk.counters[AGH_C_OVERFLOW] = 1u;
                continue;
            }
            uint32_t rec = 0;
            if (!LEAN) {
                const uint64_t strip = rs >> AGH_STRIP_SHIFT;
                rec = strip < n_strips ? wave_prefix[strip / AGH_WAVE_STRIPS] + strip_prefix[strip] : 0u;
                if (mb) {
                    rec += dbm_count(dbm, strip << AGH_STRIP_SHIFT, rs);
                } else {
                    for (uint64_t i = strip << AGH_STRIP_SHIFT; i + 16 <= rs; i += 16)
                        rec += delims_in(*reinterpret_cast<const uint4 *>(text + i), dd);
                    for (uint64_t i = rs & ~(uint64_t)15; i < rs; ++i) rec += text[i] == q.delim;
                }
            }
            TableAutomaton<K> A;
            A.reset(T);
            // the state at the record's start: what the delimiter (or the virtual head byte) in front
            // of it left behind
            if (rs == 0) (void)A.template feed_q<COSTS>(lmask[q.head_byte], T, q);
            else if (mb) A.template force_boundary<COSTS>(lmask[text[rs - 1]], T, q);
            else (void)A.template feed_q<COSTS>(lmask[q.delim], T, q);
            // 16-byte pieces (aligned; the next one is on its way while this one runs): a byte load and -- under
            // delimiters from the bitmap -- an 8-byte load per text byte, plus three branches, made this loop ~200
            // cycles a byte: 1.4 ms of a 4.5 ms scan (15 of 24 ms with 1.7 KB records), five lanes of a wave busy.
            // No record ends in [rs, P) (rs is the start of the record that holds byte P, P is 16-byte aligned), so the
            // whole pieces in front of P only advance the recurrence: 16 masks from LDS, 16 steps, nothing else.
            const uint64_t pb_end = (pend + 15) & ~(uint64_t)15;
            uint64_t pb = rs & ~(uint64_t)15;
            uint4 vn = make_uint4(0, 0, 0, 0);
            uint32_t dn = 0;
            if (pb < pb_end) {
                vn = *reinterpret_cast<const uint4 *>(text + pb);
                dn = mb ? (uint32_t)reinterpret_cast<const uint16_t *>(dbm)[pb >> 4] : 0u;
            }
            for (; pb < pb_end; pb += 16) {
                const uint4 v = vn;
                const uint32_t d16 = dn;
                if (pb + 16 < pb_end) {
                    vn = *reinterpret_cast<const uint4 *>(text + pb + 16);
                    dn = mb ? (uint32_t)reinterpret_cast<const uint16_t *>(dbm)[(pb + 16) >> 4] : 0u;
                }
                const uint32_t dws[4] = {v.x, v.y, v.z, v.w};
                if (pb >= rs && pb + 16 <= P) {             // a whole piece inside the record
                    uint32_t cm[16];
#pragma unroll
                    for (uint32_t bi = 0; bi < 16; ++bi) cm[bi] = lmask[(dws[bi >> 2] >> (8u * (bi & 3u))) & 0xffu];
#pragma unroll
                    for (uint32_t bi = 0; bi < 16; ++bi) (void)A.template feed_q<COSTS>(cm[bi], T, q);
                    continue;
                }
                const uint32_t b0 = pb < rs ? (uint32_t)(rs - pb) : 0u;           // (rs only moves forward inside the loop: fixed here)
                const uint32_t b1 = pb + 16 <= pend ? 16u : (uint32_t)(pend - pb);
#pragma unroll
                for (uint32_t bi = 0; bi < 16; ++bi) {
                    if (bi < b0 || bi >= b1) continue;
                    const uint64_t i = pb + bi;
                    const uint32_t c = (dws[bi >> 2] >> (8u * (bi & 3u))) & 0xffu;
                    uint32_t r = A.template feed_q<COSTS>(lmask[c], T, q);
                    if (mb) {                   // k_tablescan's step(): the bitmap says where, the automaton what
                        const uint32_t dbit = (d16 >> bi) & 1u;
                        if (dbit && !(r & 1u)) {
                            A.template force_boundary<COSTS>(lmask[c], T, q);
                            r = 1u;
                        } else if (!dbit) {
                            r = 0u;
                        }
                    }
                    if (r & 1u) {               // a record closes at i
                        if ((r & 2u) && i >= P) {
                            if (LEAN) lean_insert(mk, rs); else mark_record(mk, rec, i);
                        }
                        ++rec;
                        rs = i + 1;
                    }
                }
            }
            if (pend == n && q.tail_virtual && rs < n) {    // asearch.c:87-91: the open record at the end
                uint32_t r = 0;
                if (mb) {
                    for (uint32_t jd = 0; jd < q.dlen && !(r & 1u); ++jd)
                        r = A.template feed_q<COSTS>(lmask[q.dbytes[jd]], T, q);
                } else {
                    r = A.template feed_q<COSTS>(lmask[q.delim], T, q);
                }
                if ((r & 3u) == 3u) {
                    if (LEAN) lean_insert(mk, rs); else mark_record(mk, rec, n);
                }
            }
        }
    }
}

void agh_launch_tablescan(const agh_scan_args &a, hipStream_t st)
{
    const uint64_t tile_bytes = (uint64_t)WAVE * AGH_FS_CHUNK;
    const uint64_t n_tiles = (a.n + tile_bytes - 1) / tile_bytes;
    if (!n_tiles) return;
    const uint64_t want = (n_tiles + (AGH_FS_THREADS / WAVE) - 1) / (AGH_FS_THREADS / WAVE);
    const uint32_t blocks = want > 16384 ? 16384u : (uint32_t)want;
    const bool lean = a.mk.hashset != nullptr;  // count-only: hash set of record starts, no census
    const bool costs = a.q.ci != 1u || a.q.cs != 1u || a.q.cd != 1u;     // asearch1.c instead of asearch.c
    if (a.fs_fast) {                            // branch-free hot kernel + exact replay (the host checked)
        // chunk per lane: 8 KiB from 8 GiB on, 4 KiB from 1 GiB, 2 KiB from 512 MiB, else 1 KiB (a.tf_chunk forces one)
        const uint32_t tf_chunk = agh_tf_chunk_for(a.n, a.tf_chunk);
        const uint32_t tf_slice = AGH_TF_SLICE_OF(tf_chunk);
        const uint64_t tf_tile = (uint64_t)WAVE * tf_chunk;             // 64 / 128 / 256 KiB tiles
        const uint32_t nt = (uint32_t)((a.n + tf_tile - 1) / tf_tile);
        const uint32_t fblocks = (nt + 3u) / 4u > 16384u ? 16384u : (nt + 3u) / 4u;
        const uint32_t group = a.tr_group ? a.tr_group : 8u;
        const uint32_t rblocks = ((nt + group - 1u) / group + 3u) / 4u;    // (k_table_replay: a wave per `group` tiles)
        const uint32_t nt2 = (nt + 1u) / 2u;
        const uint32_t fblocks2 = (nt2 + 3u) / 4u > 16384u ? 16384u : (nt2 + 3u) / 4u;
        const uint32_t M = (uint32_t)a.q.m + a.q.dlen + 1u;                  // maskgen's M (agh_query_from_maskgen)
        // count-only, no ';': pieces in which one record ends are counted by the fast kernel itself (tf_one_end)
        const uint32_t direct = (lean && !a.tab.AND && a.tf_direct) ? 1u : 0u;
        tf_cont_args cont;
        cont.ent = a.tf_cont;
        cont.cap = a.tf_cont_cap;
        cont.at = a.tf_cont ? a.tf_cont_at : 0u;
        // (k_table_cont: a lane per handed-over record, refilled; at most `at` lanes of every tile hand one or two over)
        // (the hand-over count and its ticket start at zero for every launch: a scan that is repeated -- a bigger record
        // bitmap -- does not zero the scan's counters again)
        if (cont.at) (void)hipMemsetAsync(a.mk.counters + AGH_C_CONT_N, 0, 2 * sizeof(uint32_t), st);
        const uint64_t cwant = ((uint64_t)nt * (cont.at ? cont.at : 1u) + 255u) / 256u;
        const uint32_t cblocks = cwant > 2048u ? 2048u : (cwant ? (uint32_t)cwant : 1u);
#define AGH_TF_REPLAY(KK, LEANV, COSTV)                                                       \
            hipLaunchKernelGGL((k_table_replay<KK, LEANV, COSTV>), dim3(rblocks), dim3(256), 0, st, \
                               (const uint8_t *)a.text, a.n, a.q, a.tab, (const uint32_t *)a.mask, \
                               (const uint64_t *)a.fs_replay, (const uint32_t *)a.fs_tile_cnt, nt, \
                               a.strip_prefix, a.wave_prefix, a.n_strips, a.mk, tf_slice, a.dbm, group)
#define AGH_TF_FAST(KK, COSTV, MBV)                                                           \
            if (a.fs_fast == 2)                                                               \
                hipLaunchKernelGGL((k_tablescan_fast2<KK, COSTV, MBV>), dim3(fblocks2), dim3(AGH_FS_THREADS), 0, st, \
                                   (const uint8_t *)a.text, a.n, a.q, a.tab, (const uint32_t *)a.mask, \
                                   a.fs_replay, a.fs_tile_cnt, a.mk.counters, M, nt, tf_chunk, (const uint16_t *)a.dbm, direct, cont); \
            else                                                                              \
                hipLaunchKernelGGL((k_tablescan_fast<KK, COSTV, MBV>), dim3(fblocks), dim3(AGH_FS_THREADS), 0, st, \
                                   (const uint8_t *)a.text, a.n, a.q, a.tab, (const uint32_t *)a.mask, \
                                   a.fs_replay, a.fs_tile_cnt, a.mk.counters, tf_chunk, (const uint16_t *)a.dbm, direct, cont); \
            if (cont.at)                                                                      \
                hipLaunchKernelGGL((k_table_cont<KK, COSTV, MBV>), dim3(cblocks), dim3(256), 0, st, \
                                   (const uint8_t *)a.text, a.n, a.q, a.tab, (const uint32_t *)a.mask, \
                                   (const uint4 *)cont.ent, cont.cap, a.fs_replay, a.fs_tile_cnt, a.mk.counters, tf_slice, \
                                   (const uint16_t *)a.dbm, direct, M)
#define AGH_TF_TAIL(KK, COSTV)                                                                \
            if (lean) AGH_TF_REPLAY(KK, true, COSTV); else AGH_TF_REPLAY(KK, false, COSTV);   \
            break
// (delimiters of several bytes: fast forms up to k = 4 -- the host keeps more errors on the exact kernel)
#define AGH_TF_CASE_MB(KK)                                                                    \
    case KK:                                                                                  \
        if (costs && a.q.mb) { AGH_TF_FAST(KK, true, true); AGH_TF_TAIL(KK, true); }          \
        if (costs) { AGH_TF_FAST(KK, true, false); AGH_TF_TAIL(KK, true); }                   \
        if (a.q.mb) { AGH_TF_FAST(KK, false, true); AGH_TF_TAIL(KK, false); }                 \
        { AGH_TF_FAST(KK, false, false); AGH_TF_TAIL(KK, false); }
#define AGH_TF_CASE(KK)                                                                       \
    case KK:                                                                                  \
        if (costs) { AGH_TF_FAST(KK, true, false); AGH_TF_TAIL(KK, true); }                   \
        { AGH_TF_FAST(KK, false, false); AGH_TF_TAIL(KK, false); }
        switch (a.q.k) {
            AGH_TF_CASE_MB(0) AGH_TF_CASE_MB(1) AGH_TF_CASE_MB(2) AGH_TF_CASE_MB(3) AGH_TF_CASE_MB(4)
            AGH_TF_CASE(5) AGH_TF_CASE(6) AGH_TF_CASE(7) AGH_TF_CASE(8)
        default: break;
        }
#undef AGH_TF_CASE_MB
#undef AGH_TF_FAST
#undef AGH_TF_TAIL
#undef AGH_TF_CASE
#undef AGH_TF_REPLAY
        return;
    }
#define AGH_TS_LAUNCH(KK, LEANV, COSTV, MBV)                                                  \
    hipLaunchKernelGGL((k_tablescan<KK, LEANV, COSTV, MBV>), dim3(blocks), dim3(AGH_FS_THREADS), 0, st, \
                       (const uint8_t *)a.text, a.n, a.q, a.tab, (const uint32_t *)a.mask,    \
                       a.strip_prefix, a.wave_prefix, a.n_strips, a.mk, a.dbm)
#define AGH_CASE_MB(KK, MBV)                                                                  \
        if (lean && costs) AGH_TS_LAUNCH(KK, true, true, MBV);                                \
        else if (lean) AGH_TS_LAUNCH(KK, true, false, MBV);                                   \
        else if (costs) AGH_TS_LAUNCH(KK, false, true, MBV);                                  \
        else AGH_TS_LAUNCH(KK, false, false, MBV);
#define AGH_CASE(KK)                                                                          \
    case KK:                                                                                  \
        if (a.q.mb) { AGH_CASE_MB(KK, true) } else { AGH_CASE_MB(KK, false) }                 \
        break;
    switch (a.q.k) {
        AGH_CASE(0) AGH_CASE(1) AGH_CASE(2) AGH_CASE(3) AGH_CASE(4)
        AGH_CASE(5) AGH_CASE(6) AGH_CASE(7) AGH_CASE(8)
    default: break;
    }
#undef AGH_CASE_MB
#undef AGH_CASE
#undef AGH_TS_LAUNCH
}

// ---------------------------------------------------------------------------------------
// -v (INVERSE, asearch.c:128): the records whose bit is NOT set in the record bitmap.  Text-
// parallel like k_tablescan: a lane looks at the delimiters of its 256-byte chunk; the record
// a delimiter closes has the number "delimiters in front of it".  Leaves rec_pos[r] = position of the
// closing delimiter for every unmatched record r and the record count in AGH_C_NREC; the list itself is
// the ordered compaction of the inverted bitmap (agh_records.hip).
// ---------------------------------------------------------------------------------------
// Delimiters of several bytes (or a folded letter): the delimiter ends come from the bitmap.
__global__ __launch_bounds__(256) void k_unmatched(const uint8_t *__restrict__ text, uint64_t n,
                                                   agh_dev_query q,
                                                   const uint32_t *__restrict__ strip_prefix,
                                                   const uint32_t *__restrict__ wave_prefix,
                                                   uint32_t n_strips, agh_marks mk,
                                                   const uint64_t *__restrict__ dbm)
{
    const bool mb = q.mb != 0;
    auto is_delim_end = [&](uint64_t p) -> bool { return mb ? dbm_bit(dbm, p) != 0 : text[p] == q.delim; };
    const uint64_t n_chunks = (n + AGH_TS_CHUNK - 1) / AGH_TS_CHUNK;
    const uint32_t dd = q.delim * 0x01010101u;
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    for (uint64_t base = (uint64_t)blockIdx.x * 256u; base < n_chunks;
         base += (uint64_t)gridDim.x * 256u) {
        const uint64_t cs = (base + threadIdx.x) * AGH_TS_CHUNK;
        uint64_t ce = cs + AGH_TS_CHUNK;
        if (ce > n) ce = n;
        uint32_t my_delims = 0;
        if (cs < n && mb) {
            my_delims = dbm_count(dbm, cs, ce);
        } else if (cs < n) {
            const uint32_t len = (uint32_t)(ce - cs);
            for (uint32_t i = 0; i < (len >> 4); ++i)
                my_delims += delims_in(*reinterpret_cast<const uint4 *>(text + cs + i * 16), dd);
            if (len & 15u)
                my_delims += delims_in(
                    mask_tail(*reinterpret_cast<const uint4 *>(text + cs + (len & ~15u)),
                              (int)(len & 15u), fill4), dd);
        }
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
        const uint64_t strip = cs >> AGH_STRIP_SHIFT;
        uint32_t rec = (cs < n && strip < n_strips)
                           ? wave_prefix[strip / AGH_WAVE_STRIPS] + strip_prefix[strip] + before
                           : 0u;
        // the last, unterminated record is closed by the delimiter appended at EOF
        const bool open_tail = cs < n && ce == n && q.tail_virtual && !is_delim_end(n - 1);
        // every record of mine whose bit stayed clear: rec_pos[r] = where it ends (the ordered compaction of the
        // inverted bitmap turns these into the list, agh_records.hip)
        if (cs < n) {
            uint32_t r = rec;
            for (uint64_t p = cs; p < ce; ++p)
                if (is_delim_end(p)) {
                    if (r < mk.bitmap_bits && !((mk.bitmap[r >> 5] >> (r & 31u)) & 1u)) mk.rec_pos[r] = p;
                    ++r;
                }
            if (open_tail && r < mk.bitmap_bits && !((mk.bitmap[r >> 5] >> (r & 31u)) & 1u)) mk.rec_pos[r] = n;
            if (ce == n) mk.counters[AGH_C_NREC] = r + (open_tail ? 1u : 0u);   // records of the text
        }
    }
}

void agh_launch_unmatched(const agh_scan_args &a, hipStream_t st)
{
    const uint64_t n_chunks = (a.n + AGH_TS_CHUNK - 1) / AGH_TS_CHUNK;
    if (!n_chunks) return;
    uint64_t want = (n_chunks + 255) / 256;
    const uint32_t blocks = want > 65536 ? 65536u : (uint32_t)want;
    hipLaunchKernelGGL(k_unmatched, dim3(blocks), dim3(256), 0, st, (const uint8_t *)a.text, a.n,
                       a.q, a.strip_prefix, a.wave_prefix, a.n_strips, a.mk, a.dbm);
}
