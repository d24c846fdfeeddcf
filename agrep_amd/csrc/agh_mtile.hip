// agh_mtile.hip -- count-only -f scans with one error over DENSE sets (BASELINE config 5 as SURVEY 8d words it:
// 1024 patterns of 4..12 bytes, k = 1: pieces of two bytes, every fifth position a candidate, four records in five
// match) in two phases per tile, inside one persistent kernel.  (Role of newmgrep.c:463-691 mgrep() + :839-1012
// monkey1(); what carries over from the reference is its early exit: after a record's first verified entry it goes on
// behind the record's end, newmgrep.c:858-905.)
//
// Round 5's record walk (k_mwalk, one lane per 1..4 KiB; removed after the A/B in profiles/r06_ab_mtile_v1.log) had the
// early exit but the wrong layout: every lane streamed its own cache lines (EA reads 2.3-2.8 x the text, 43 % of wave
// time on memory) and walked its slot's entries while its 63 neighbours waited (124 lane-instructions per byte):
// 207 GB/s.  Here the two kinds of work are separated:
//
//   phase A  position-parallel, the sweeps' chunk-per-lane layout (a wave loads 1 KiB strips with one coalesced
//            dwordx4 per lane): for each of a chunk's 16 positions the pair t[j], t[j+1] selects a directory slot
//            whose four 32-bit masks say which bytes next to the pair some entry of two or three bytes could accept at
//            all (round 5's lossless necessary condition), the four bytes at j select a bit that says whether a piece of
//            >= 4 bytes starts with them (fill_multi_tables) -- a CANDIDATE bit; and a DELIMITER bit per
//            byte.  Both go through 1 KiB of LDS per wave into natural bit order: lane l then holds the 64 candidate
//            and the 64 delimiter bits of positions [64 l, 64 l + 64) of the tile.
//   phase B  every lane walks the candidate bits of its own word(s) in rounds, lowest first: the entries of the
//            candidate's slot against the text at j (the walk's examine step: piece verbatim, the necessary condition
//            on the entry's own bytes, side_within_one_edit once per round).  A hit clears every candidate bit of its
//            record's part of the word (the run between the neighbouring delimiter bits): the early exit.  The text
//            comes from the lines the wave has just streamed (L1 / L2 hits, neighbouring lanes 64 bytes apart).
//   count    a record lies in the tile that holds the delimiter ending it; it matched iff a matched bit stands in the
//            run of non-delimiter bits below that delimiter -- ~D + M carries exactly those runs into their delimiter's
//            bit, and the same trick on the ballots of "my top run matched" / "no delimiter in my word" carries a hit
//            across the lanes of the tile (carry look-ahead as one scalar addition).  Records that lie inside one tile
//            are counted in a register; the first and the last record part of a tile (records that cross tile bounds,
//            one in ~25) go into the scan's hash set of record starts like in every count-only engine.
//
// The first 8 and the last 24 positions of the text go through the general verifier (k_mtile_edges), and records
// that reach into those stretches are never counted in a register -- what both kernels find there meets in the
// hash set.
#include "agh_multi_inl.h"

#define MT_WAVES 16u                // waves per workgroup = per CU (one copy of the tables)
#define MT_TILE 4096u               // 64 lanes x 64 positions
#define MT_RANGE_TILES 64u          // tiles per ticket (256 KiB) on large texts; fewer on small ones: the launcher
#define MT_LIST 512u                // entries of a wave's list of left-over candidates
#define MT_SHARE_AT 32u             // ... shared out once so few lanes still have a candidate of their own

typedef uint64_t u64_a1 __attribute__((aligned(1)));

__device__ __forceinline__ uint32_t mt_uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// bit i (0..3) = byte i of w equals the delimiter
__device__ __forceinline__ uint32_t mt_delim_nibble(uint32_t w, uint32_t dd)
{
    const uint32_t x = w ^ dd;
    const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);     // bit 7 of every zero byte
    return (((z >> 7) & 0x01010101u) * 0x10204080u) >> 28;
}

// bits [a, b) of a 64-bit word (0 <= a, b <= 64; a >= b: none)
__device__ __forceinline__ uint64_t mt_bits(uint32_t a, uint32_t b)
{
    if (a >= b) return 0ull;
    const uint64_t hi = b >= 64u ? ~0ull : ((1ull << b) - 1ull);
    return hi & ~((1ull << a) - 1ull);
}

// the bits of a word that belong to positions [lo, hi) when the word's bit 0 is position `base`
__device__ __forceinline__ uint64_t mt_valid(uint64_t base, uint64_t lo, uint64_t hi)
{
    const uint64_t a = lo > base ? lo - base : 0, b = hi > base ? hi - base : 0;
    return mt_bits(a > 64 ? 64u : (uint32_t)a, b > 64 ? 64u : (uint32_t)b);
}

struct mt_shared {
    uint4 fmask[AGH_MW_DIR];                    // 64 KiB
    uint32_t dir[AGH_MW_DIR];                   // 16 KiB
    uint4 ent[AGH_MW_MAX_ENT];                  // 48 KiB
    uint32_t g4[AGH_MW_G4_WORDS];               // 8 KiB: the first four bytes of the pieces of >= 4 bytes, one bit each
    uint16_t scratch[MT_WAVES][MT_LIST];        // 16 KiB, per wave: phase A: a tile's candidate bits (16 per lane and
                                                // strip) and, behind them, its delimiter bits; phase B: the list of the
                                                // candidates the wave shares out at the end
};

// NW: tiles a wave holds at a time (a lane walks the candidate words of NW tiles in the same rounds: the more words,
// the less a wave waits for its unluckiest lane)
// NUM: numbered scans (record lists, -n): the census of the numbered pipeline ran in front (wave_totals / strip_prefix:
// delimiters in front of every 1 KiB strip), a record's number is the number of delimiters in front of it, and matched
// records are marked in the record bitmap -- first and last parts of a tile like every other record, no hash set.
template <bool FOLD, int NW, bool NUM>
__global__ __launch_bounds__(MT_WAVES * 64) void k_mtile(const uint8_t *__restrict__ text, uint64_t n, uint32_t delim,
                                                         agh_mwalk_dev mw, agh_marks mk, uint32_t *__restrict__ ticket,
                                                         uint32_t n_ranges, uint32_t range_tiles, uint32_t dbg,
                                                         const uint32_t *__restrict__ wave_totals,
                                                         const uint32_t *__restrict__ strip_prefix)
{
    // dbg (AGH_MTILE_DBG, measurements only): 1 no walk over the candidate bits, 2 the walk without its text loads,
    // 4 AGH_C_CAND counts the rounds of the walk (per wave and tile group) instead of the candidates examined;
    // bits 8..: 1 + the number of lanes with candidates at which the rest is shared out (0: MT_SHARE_AT)
    __shared__ __attribute__((aligned(16))) mt_shared sh;
    for (uint32_t i = threadIdx.x; i < AGH_MW_DIR; i += MT_WAVES * 64) {
        sh.fmask[i] = mw.fmask[i];
        sh.dir[i] = mw.dir[i];
    }
    for (uint32_t i = threadIdx.x; i < mw.n_ent; i += MT_WAVES * 64) sh.ent[i] = mw.ent[i];
    for (uint32_t i = threadIdx.x; i < AGH_MW_G4_WORDS; i += MT_WAVES * 64) sh.g4[i] = mw.g4[i];
    __syncthreads();
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t wib = mt_uni(threadIdx.x / WAVE);
    const uint32_t total_waves = gridDim.x * MT_WAVES;
    const uint32_t dd = delim * 0x01010101u;
    const uint64_t lo_lim = 8, hi_lim = n - 24;           // (n >= 32: the launcher checks)
    const uint64_t n_pad = (n + 15) & ~(uint64_t)15;      // readable bytes
    const uint64_t n_tiles = (n + MT_TILE - 1) / MT_TILE;
    const uint8_t *fmask8 = reinterpret_cast<const uint8_t *>(sh.fmask), *g4b = reinterpret_cast<const uint8_t *>(sh.g4);
    uint16_t *cb16 = sh.scratch[wib], *db16 = cb16 + 256, *list = cb16;
    const uint64_t *cb64 = reinterpret_cast<const uint64_t *>(cb16), *db64 = reinterpret_cast<const uint64_t *>(db16);
    uint32_t local = 0;                                   // matched records that lie inside one tile (per lane)
    uint32_t n_exam = 0;                                  // candidates examined by this wave (dbg & 4: rounds)
    const uint32_t share_at = (dbg >> 8) ? (dbg >> 8) - 1u : MT_SHARE_AT;     // (A/B: AGH_MTILE_SHARE + 1 in bits 8..)

    auto load_strip = [&](uint64_t off) -> uint4 {        // 16 bytes at off + 16 lane; behind the text: delimiters
        const uint64_t o = off + (uint64_t)lane * 16u;
        uint4 v = make_uint4(dd, dd, dd, dd);
        if (o < n_pad) v = *reinterpret_cast<const uint4 *>(text + o);
        return v;
    };
    auto load_dword = [&](uint64_t off) -> uint32_t {     // (off a multiple of four) wave-uniform
        return off < n_pad ? *reinterpret_cast<const uint32_t *>(text + off) : dd;
    };

    // ---- phase A: candidate and delimiter bits of one strip (16 positions per lane) ---------------------------
    // W = t[c - 4 .. c + 20): the dword in front of the chunk, the chunk, the dword behind it
    auto strip_bits = [&](const uint32_t (&W)[6], uint32_t &cand16, uint32_t &delim16) {
        delim16 = mt_delim_nibble(W[1], dd) | mt_delim_nibble(W[2], dd) << 4 | mt_delim_nibble(W[3], dd) << 8 |
                  mt_delim_nibble(W[4], dd) << 12;
        auto byte_at = [&](int idx) -> uint32_t { return (W[idx >> 2] >> (8 * (idx & 3))) & 0xffu; };   // idx: offset in W
        uint32_t acc = 0;
        // four positions at a time; the table reads of the next four are in flight while these are tested (left alone
        // the scheduler issues all sixteen reads of the chunk first: 64 result registers, spills in the hot loop)
        // per position two reads: the pair's masks (pieces of two and three bytes) and the bit of the four bytes at j
        // (pieces of >= 4 bytes: with the third byte alone -- round 6's first version -- 7 % of all positions were
        // candidates of such a piece and 96 % of those failed on the fourth byte in the walk)
        auto fetch = [&](int g, uint4 (&fm)[4], uint32_t (&gw)[4], uint32_t (&gs)[4]) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int o = 4 + 4 * g + p;                                // offset of t[j] in W
                uint32_t gram = (o & 3) ? __builtin_amdgcn_alignbyte(W[(o >> 2) + 1], W[o >> 2], o & 3) : W[o >> 2];
                if (FOLD) gram = swar_lower(gram);
                // slot * 16 = the byte offset of the slot's masks: (pair * 40503 >> 4 & 4095) << 4
                const uint32_t a = ((gram & 0xffffu) * 40503u) & ((AGH_MW_DIR - 1u) << 4);
                fm[p] = *reinterpret_cast<const uint4 *>(fmask8 + a);
                const uint32_t h = agh_sample_prod_q4(gram);                // word: bits 2..12, bit: bits 13..17
                gw[p] = *reinterpret_cast<const uint32_t *>(g4b + (h & ((AGH_MW_G4_WORDS - 1u) << 2)));
                gs[p] = h >> 13;
            }
        };
        auto test = [&](int g, const uint4 (&fm)[4], const uint32_t (&gw)[4], const uint32_t (&gs)[4]) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int o = 4 + 4 * g + p;
                const uint32_t hit = (fm[p].x >> (byte_at(o + 2) & 31u)) | (fm[p].y >> (byte_at(o + 3) & 31u)) |
                                     (fm[p].z >> (byte_at(o - 1) & 31u)) | (fm[p].w >> (byte_at(o - 2) & 31u)) |
                                     (gw[p] >> (gs[p] & 31u));
                acc = __builtin_amdgcn_alignbit(hit, acc, 1);               // bit 0 of hit -> bit 31, the rest moves down
            }
            asm volatile("" : "+v"(acc));
        };
        uint4 fa[4], fb[4];
        uint32_t ga[4], gb[4], sa[4], sb[4];
        fetch(0, fa, ga, sa);
        fetch(1, fb, gb, sb);
        __builtin_amdgcn_sched_barrier(0);
        test(0, fa, ga, sa);
        __builtin_amdgcn_sched_barrier(0);
        fetch(2, fa, ga, sa);
        __builtin_amdgcn_sched_barrier(0);
        test(1, fb, gb, sb);
        __builtin_amdgcn_sched_barrier(0);
        fetch(3, fb, gb, sb);
        __builtin_amdgcn_sched_barrier(0);
        test(2, fa, ga, sa);
        __builtin_amdgcn_sched_barrier(0);
        test(3, fb, gb, sb);
        __builtin_amdgcn_sched_barrier(0);
        cand16 = (acc >> 16) & ~delim16;                                    // (no entry holds the delimiter byte)
    };

    // ---- phase B, one candidate: does some entry match at j?  Two lists: the pieces of two bytes under the pair at j,
    // the longer ones under the three bytes at j (agh_mw_slot / agh_mw_slot3) -----------------------------------------
    auto examine = [&](bool cand, uint64_t j) -> bool {
        bool matched = false;
        uint64_t P = 0, F0 = 0, F1 = 0;
        uint32_t first2 = 0, cnt2 = 0, first3 = 0, cnt = 0;
        if (cand && !(dbg & 2u)) {
            P = *reinterpret_cast<const u64_a1 *>(text + j - 8);
            F0 = *reinterpret_cast<const u64_a1 *>(text + j);
            F1 = *reinterpret_cast<const u64_a1 *>(text + j + 8);
            if (FOLD) {
                P = (uint64_t)swar_lower((uint32_t)P) | ((uint64_t)swar_lower((uint32_t)(P >> 32)) << 32);
                F0 = (uint64_t)swar_lower((uint32_t)F0) | ((uint64_t)swar_lower((uint32_t)(F0 >> 32)) << 32);
                F1 = (uint64_t)swar_lower((uint32_t)F1) | ((uint64_t)swar_lower((uint32_t)(F1 >> 32)) << 32);
            }
        }
        if (cand) {
            const uint32_t d2 = sh.dir[agh_mw_slot((uint32_t)F0 & 0xffffu)], d3 = sh.dir[agh_mw_slot3((uint32_t)F0 & 0xffffffu)];
            first2 = d2 >> 16;
            cnt2 = d2 & 0xffffu;
            first3 = (d3 >> 16) - cnt2;                                     // (entry i >= cnt2 of the walk: first3 + i)
            cnt = cnt2 + (d3 & 0xffffu);
        }
        const uint64_t Ph = __builtin_bswap64(P);                           // the bytes in front of j, nearest first
        uint32_t i = 0;
        while (__ballot(i < cnt)) {
            // the next entry whose piece stands at j and whose side passes the necessary condition ...
            bool pend = false;
            uint64_t S = 0, B = 0;
            uint32_t L = 0;
            while (i < cnt && !pend) {
                const uint4 e = sh.ent[(i < cnt2 ? first2 : first3) + i];
                ++i;
                const uint32_t pl = e.y >> 24;                              // piece length 2..7
                // the piece's pl bytes against t[j ..): what differs above them is shifted out
                const uint64_t E = (uint64_t)e.x | ((uint64_t)(e.y & 0xffffffu) << 32);
                if ((F0 ^ E) << (64u - 8u * pl)) continue;
                const uint32_t meta = e.w >> 24;
                L = meta & 7u;
                B = (uint64_t)e.z | ((uint64_t)(e.w & 0xffffffu) << 32);
                S = (meta & 8u) ? Ph : ((F0 >> (8u * pl)) | (F1 << (64u - 8u * pl)));
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
        return matched;
    };

    uint32_t r = blockIdx.x * MT_WAVES + wib;
    while (r < n_ranges) {
        const uint64_t t0 = (uint64_t)r * range_tiles;
        uint64_t t1 = t0 + range_tiles;
        if (t1 > n_tiles) t1 = n_tiles;
        // the first tile of the range is in flight before the loop, every further one before the walk of the tiles
        // in front of it
        uint4 nv0 = load_strip(t0 * MT_TILE), nv1 = load_strip(t0 * MT_TILE + 1024u), nv2 = load_strip(t0 * MT_TILE + 2048u),
              nv3 = load_strip(t0 * MT_TILE + 3072u);
        for (uint64_t tg = t0; tg < t1; tg += NW) {
            uint64_t C[NW], D[NW], M[NW];
            uint32_t s0[NW];                                                // the tile starts where a record starts
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                C[k] = D[k] = M[k] = 0;
                s0[k] = 0;
                const uint64_t tb = (tg + (uint64_t)k) * MT_TILE;
                if (tg + (uint64_t)k >= t1) continue;                       // (uniform) the range ends inside the group
                const uint4 v0 = nv0, v1 = nv1, v2 = nv2, v3 = nv3;
                const uint32_t pre = tb >= 4u ? load_dword(tb - 4u) : dd;     // t[tb - 4 .. tb)
                const uint32_t post = load_dword(tb + MT_TILE);              // t[tb + 4096 .. tb + 4100)
                if (tg + (uint64_t)k + 1u < t1) {                           // the next tile of the range
                    nv0 = load_strip(tb + MT_TILE);
                    nv1 = load_strip(tb + MT_TILE + 1024u);
                    nv2 = load_strip(tb + MT_TILE + 2048u);
                    nv3 = load_strip(tb + MT_TILE + 3072u);
                }
                s0[k] = (tb >= lo_lim && (pre >> 24) == delim) ? 1u : 0u;
                // the dword in front of every chunk (the lane before me: DPP wave_shr:1; lane 0: the strip before) and
                // behind it (wave_shl:1; lane 63: the strip behind)
                auto prev_w = [&](uint32_t w, uint32_t wrap) -> uint32_t {
                    return (uint32_t)__builtin_amdgcn_update_dpp((int)wrap, (int)w, 0x138, 0xf, 0xf, false);
                };
                auto next_x = [&](uint32_t x, uint32_t wrap) -> uint32_t {
                    return (uint32_t)__builtin_amdgcn_update_dpp((int)wrap, (int)x, 0x130, 0xf, 0xf, false);
                };
                const uint32_t l63_0 = (uint32_t)__builtin_amdgcn_readlane((int)v0.w, 63), l63_1 = (uint32_t)__builtin_amdgcn_readlane((int)v1.w, 63),
                               l63_2 = (uint32_t)__builtin_amdgcn_readlane((int)v2.w, 63);
                const uint32_t l0_1 = (uint32_t)__builtin_amdgcn_readlane((int)v1.x, 0), l0_2 = (uint32_t)__builtin_amdgcn_readlane((int)v2.x, 0),
                               l0_3 = (uint32_t)__builtin_amdgcn_readlane((int)v3.x, 0);
                uint32_t c16, d16;
                {
                    const uint32_t W[6] = {prev_w(v0.w, pre), v0.x, v0.y, v0.z, v0.w, next_x(v0.x, l0_1)};
                    strip_bits(W, c16, d16);
                    cb16[lane] = (uint16_t)c16;
                    db16[lane] = (uint16_t)d16;
                }
                {
                    const uint32_t W[6] = {prev_w(v1.w, l63_0), v1.x, v1.y, v1.z, v1.w, next_x(v1.x, l0_2)};
                    strip_bits(W, c16, d16);
                    cb16[64u + lane] = (uint16_t)c16;
                    db16[64u + lane] = (uint16_t)d16;
                }
                {
                    const uint32_t W[6] = {prev_w(v2.w, l63_1), v2.x, v2.y, v2.z, v2.w, next_x(v2.x, l0_3)};
                    strip_bits(W, c16, d16);
                    cb16[128u + lane] = (uint16_t)c16;
                    db16[128u + lane] = (uint16_t)d16;
                }
                {
                    const uint32_t W[6] = {prev_w(v3.w, l63_2), v3.x, v3.y, v3.z, v3.w, next_x(v3.x, post)};
                    strip_bits(W, c16, d16);
                    cb16[192u + lane] = (uint16_t)c16;
                    db16[192u + lane] = (uint16_t)d16;
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const uint64_t wbase = tb + (uint64_t)lane * 64u;
                C[k] = (dbg & 1u) ? 0ull : cb64[lane] & mt_valid(wbase, lo_lim, hi_lim);
                // (count-only: a record is counted in a register only between delimiters the walk can see on both
                // sides; numbered: every delimiter of the text numbers a record)
                D[k] = db64[lane] & (NUM ? mt_valid(wbase, 0, n) : mt_valid(wbase, lo_lim - 1u, hi_lim));
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // (the next tile writes the same scratch)
            }

            // ---- phase B: rounds over the candidate bits ---------------------------------------------------------
            // (a lane takes its words one after the other and a word's candidates lowest first: below a candidate
            // nothing of its record's run is left to clear)
            for (;;) {
                int kk = -1;
                uint64_t Ck = 0, Dk = 0;
#pragma unroll
                for (int k = NW - 1; k >= 0; --k)
                    if (C[k]) { kk = k; Ck = C[k]; Dk = D[k]; }
                const bool cand = kk >= 0;
                const uint64_t cb = __ballot(cand);
                if (!cb) break;
                if ((uint32_t)__popcll(cb) <= share_at) {
                    // Few lanes are left with candidates of their own -- those whose records do not match walk every
                    // one of them (a record that matches is done after ~4), and the wave would wait for the longest
                    // of them with most of its lanes idle (first version: 11.4 rounds per tile at 40 % of the lanes,
                    // profiles/r06_ab_mtile_v3.log).  The rest is shared out: every candidate that is left goes
                    // into a list, the lanes take 64 at a time, the owners read the hits back.  (No early exit among
                    // those: a hit bit next to another one of the same record costs a test, not the count.)
                    for (;;) {
                        uint32_t mine = 0;
#pragma unroll
                        for (int k = 0; k < NW; ++k) mine += (uint32_t)__popcll(C[k]);
                        if (!__ballot(mine != 0u)) break;
                        const uint32_t incl = wave_sum_to_lane63(mine);
                        const uint32_t total = mt_uni((uint32_t)__builtin_amdgcn_readlane((int)incl, 63));
                        const uint32_t my_first = incl - mine;
                        uint32_t at = my_first;
#pragma unroll
                        for (int k = 0; k < NW; ++k) {
                            uint64_t c = C[k];
                            while (c && at < MT_LIST) {
                                const uint32_t bb = (uint32_t)__builtin_ctzll(c);
                                c &= c - 1ull;
                                list[at++] = (uint16_t)((uint32_t)k << 12 | lane << 6 | bb);
                            }
                            C[k] = c;                                       // (what found no room: the next pass)
                        }
                        const uint32_t my_n = at - my_first;
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        const uint32_t listed = total < MT_LIST ? total : MT_LIST;
                        for (uint32_t i0 = 0; i0 < listed; i0 = mt_uni(i0 + 64u)) {
                            const uint32_t idx = i0 + lane;
                            const bool has = idx < listed;
                            const uint32_t e = has ? list[idx] : 0u;
                            const uint64_t jj = (tg + (uint64_t)((e >> 12) & 3u)) * MT_TILE + (uint64_t)(e & 0xfffu);
                            n_exam = mt_uni(n_exam + ((dbg & 4u) ? 1u : (uint32_t)__popcll(__ballot(has))));
                            if (examine(has, jj) && has) list[idx] = (uint16_t)(e | 0x8000u);
                        }
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        for (uint32_t i = 0; i < my_n; ++i) {
                            const uint32_t e = list[my_first + i];
                            if (e & 0x8000u) {
#pragma unroll
                                for (int k = 0; k < NW; ++k)
                                    if ((int)((e >> 12) & 3u) == k) M[k] |= 1ull << (e & 63u);
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (the next pass writes the list again)
                    }
                    break;
                }
                n_exam = mt_uni(n_exam + ((dbg & 4u) ? 1u : (uint32_t)__popcll(cb)));
                const uint32_t b = cand ? (uint32_t)__builtin_ctzll(Ck) : 0u;
                const uint64_t j = (tg + (uint64_t)(cand ? kk : 0)) * MT_TILE + (uint64_t)lane * 64u + b;
                const bool hit = examine(cand, j);
                if (cand) {
                    const uint64_t bit = 1ull << b;
                    uint64_t clr = bit;
                    if (hit) {
                        // every candidate of this record from b to the next delimiter: + bit carries through the run of
                        // non-delimiter bits above b and stops in the delimiter's zero
                        const uint64_t S = ~Dk;
                        clr = ((S + bit) ^ S) & S;
                    }
#pragma unroll
                    for (int k = 0; k < NW; ++k)
                        if (k == kk) {
                            C[k] &= ~clr;
                            if (hit) M[k] |= bit;
                        }
                }
            }

            // ---- count: per tile, which delimiters end a matched record ---------------------------------------------
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                if (tg + (uint64_t)k >= t1) continue;
                const uint64_t tb = (tg + (uint64_t)k) * MT_TILE;
                const uint64_t Dk = D[k], Mk = M[k];
                const uint64_t S = ~Dk, Ms = Mk & S;
                const uint64_t sum = S + Ms;
                const bool g = sum < S;                                     // a matched bit in the run that reaches bit 63
                const bool pz = Dk == 0;
                uint64_t R = sum & Dk;                                      // delimiters with a matched bit in the run below them
                const uint64_t G = __ballot(g), P = __ballot(pz), ND = ~P;   // ND: lanes that hold a delimiter
                const uint64_t A = G, Bv = G | P, sum2 = A + Bv;
                const uint64_t carry_into = sum2 ^ A ^ Bv;                  // bit l: a hit reaches lane l from the lanes below
                const bool cout = ((A & Bv) | ((A | Bv) & ~sum2)) >> 63;     // ... and leaves lane 63: the tile's last part
                const uint64_t lowD = Dk & (0ull - Dk);
                if (((carry_into >> lane) & 1ull) && Dk) R |= lowD;
                const bool any_delim = ND != 0ull;
                const uint32_t l = any_delim ? 63u - (uint32_t)__builtin_clzll(ND) : 64u;      // last lane with a delimiter
                // my matched bits in the tile's last part (behind its last delimiter)
                uint64_t Mt = 0;
                if (any_delim) {
                    const uint32_t top = Dk ? 63u - (uint32_t)__builtin_clzll(Dk) : 0u;
                    Mt = lane > l ? Mk : (lane == l ? (top == 63u ? 0ull : Mk & ~((2ull << top) - 1ull)) : 0ull);
                }
                if constexpr (NUM) {
                    // record number = delimiters in front: the census up to the tile's first strip, the lanes below me,
                    // the bits below the delimiter in my word
                    const uint64_t strip0 = tb >> AGH_STRIP_SHIFT;
                    const uint32_t base = wave_totals[strip0 / AGH_WAVE_STRIPS] + strip_prefix[strip0];
                    const uint32_t mine = (uint32_t)__popcll(Dk);
                    const uint32_t incl = wave_sum_to_lane63(mine);
                    const uint32_t total = mt_uni((uint32_t)__builtin_amdgcn_readlane((int)incl, 63));
                    const uint32_t rec0 = base + incl - mine;
                    while (R) {
                        const uint32_t d = (uint32_t)__builtin_ctzll(R);
                        R &= R - 1ull;
                        mark_record(mk, rec0 + (uint32_t)__popcll(Dk & ((1ull << d) - 1ull)), tb + (uint64_t)lane * 64u + d);
                    }
                    if (!any_delim) Mt = Mk;                                // (no delimiter in the tile: one record part)
                    const uint64_t tbm = __ballot(Mt != 0ull);
                    if (cout && tbm && lane == (uint32_t)__builtin_ctzll(tbm))
                        mark_record(mk, base + total, tb + (uint64_t)lane * 64u + (uint32_t)__builtin_ctzll(Mt));
                } else {
                    const uint32_t f = any_delim ? (uint32_t)__builtin_ctzll(ND) : 64u;  // first lane with a delimiter
                    // the record that ends at the tile's first delimiter began in front of the tile unless the tile
                    // starts a record: not mine to count
                    bool head_hit = false;
                    if (!s0[k] && lane == f) {
                        head_hit = (R & lowD) != 0ull;
                        R &= ~lowD;
                    }
                    local += (uint32_t)__popcll(R);
                    // the tile's first and last record parts, if they matched: into the set of record starts (by a look
                    // back from one matched position -- the last part's start lies inside the tile, a few bytes away)
                    const uint64_t head_any = __ballot(head_hit);
                    uint64_t Mh = 0;                                        // my matched bits in the first part
                    if (!s0[k]) Mh = lane < f ? Mk : (lane == f ? Mk & (lowD - 1ull) : 0ull);
                    if (!any_delim && s0[k]) Mt = Mk;                       // one record from the tile's first byte on
                    const bool want_head = !s0[k] && (any_delim ? head_any != 0ull : cout);
                    const bool want_tail = (any_delim || s0[k]) && cout;
                    const uint64_t hb = __ballot(Mh != 0ull), tbm = __ballot(Mt != 0ull);
                    if (want_head && hb && lane == (uint32_t)__builtin_ctzll(hb)) {
                        const uint64_t st = lean_record_start(text, tb + (uint64_t)lane * 64u + (uint32_t)__builtin_ctzll(Mh), delim, mk);
                        if (st != ~0ull) lean_insert(mk, st);
                    }
                    if (want_tail && tbm && lane == (uint32_t)__builtin_ctzll(tbm)) {
                        const uint64_t st = lean_record_start(text, tb + (uint64_t)lane * 64u + (uint32_t)__builtin_ctzll(Mt), delim, mk);
                        if (st != ~0ull) lean_insert(mk, st);
                    }
                }
            }
        }
        uint32_t tk = 0;
        if (lane == 0) tk = atomicAdd(ticket, 1u);
        r = total_waves + mt_uni(tk);
    }
    if (!NUM) local = wave_sum_to_lane63(local);
    if (!NUM && lane == 63 && local) {
        atomicAdd(&mk.counters[AGH_C_MATCHED], local);
        mk.counters[AGH_C_ANYHIT] = 1u;
    }
    if (lane == 0 && n_exam) atomicAdd(&mk.counters[AGH_C_CAND], n_exam);
}

// The positions k_mtile leaves out: fewer than 8 bytes in front of them or fewer than 24 behind.  Their records
// reach into those stretches, so whatever both kernels find there meets in the hash set.
__global__ __launch_bounds__(64) void k_mtile_edges(const uint8_t *__restrict__ text8, uint64_t n, agh_dev_query q,
                                                    agh_multi_dev mt, agh_marks mk)
{
    const uint32_t lane = (uint32_t)lane_id();
    uint64_t j = ~0ull;
    if (lane < 8u) j = lane;
    else if (lane < 32u && n >= 24u + 8u) j = n - 24u + (lane - 8u);
    else if (lane < 32u && lane < n) j = lane;            // (texts below 32 bytes: all of it)
    if (j < n) mp_verify_at<true, 1>(text8, n, q, mt, j, 0u, mk);
}

// ... numbered: the record count in front of a position's 16-byte chunk comes from the census (the strip's prefix) and
// the delimiters between the strip's first byte and the chunk
__global__ __launch_bounds__(64) void k_mtile_edges_numbered(const uint8_t *__restrict__ text8, uint64_t n, agh_dev_query q,
                                                             agh_multi_dev mt, agh_marks mk,
                                                             const uint32_t *__restrict__ wave_totals,
                                                             const uint32_t *__restrict__ strip_prefix)
{
    const uint32_t lane = (uint32_t)lane_id();
    uint64_t j = ~0ull;
    if (lane < 8u) j = lane;
    else if (lane < 32u && n >= 24u + 8u) j = n - 24u + (lane - 8u);
    else if (lane < 32u && lane < n) j = lane;
    if (j >= n) return;
    const uint64_t chunk = j & ~(uint64_t)15, strip = chunk >> AGH_STRIP_SHIFT;
    uint32_t rc = wave_totals[strip / AGH_WAVE_STRIPS] + strip_prefix[strip];
    for (uint64_t i = strip << AGH_STRIP_SHIFT; i < chunk; ++i) rc += text8[i] == q.delim;
    mp_verify_at<false, 1>(text8, n, q, mt, j, rc, mk);
}

template <int NW>
static void launch_mtile(const agh_mwalk_args &a, uint32_t blocks, uint32_t n_ranges, uint32_t range_tiles, hipStream_t st)
{
#define AGH_MT_LAUNCH(F, NUMB)                                                                                        \
    hipLaunchKernelGGL((k_mtile<F, NW, NUMB>), dim3(blocks), dim3(MT_WAVES * 64), 0, st, (const uint8_t *)a.text, a.n, \
                       a.q.delim, a.mw, a.mk, a.ticket, n_ranges, range_tiles, a.ch >> 8, a.wave_totals, a.strip_prefix)
    if (a.wave_totals) { if (a.q.fold) AGH_MT_LAUNCH(true, true); else AGH_MT_LAUNCH(false, true); }
    else { if (a.q.fold) AGH_MT_LAUNCH(true, false); else AGH_MT_LAUNCH(false, false); }
#undef AGH_MT_LAUNCH
}

// false: no instance for this query / text -- the caller takes the general multi-pattern kernels.
// a.ch: tiles a wave holds at a time (1, 2 or 4; 0: the default); bits 8..: measurement switches.
// a.wave_totals / a.strip_prefix (the census of the numbered pipeline): matched records go into the record bitmap by
// number instead of being counted
bool agh_launch_mtile(const agh_mwalk_args &a, hipStream_t st)
{
    if (a.q.k != 1 || a.q.mb || !a.n || !a.mw.n_ent || a.mw.n_ent > AGH_MW_MAX_ENT) return false;
    if (a.n >= 32u) {
        const uint32_t nw = (a.ch & 0xffu) == 1u ? 1u : ((a.ch & 0xffu) == 4u ? 4u : 2u);
        const uint32_t cus = a.n_cu ? a.n_cu : 256u;
        const uint64_t n_tiles = (a.n + MT_TILE - 1u) / MT_TILE;
        // tiles per ticket: 64 where that still leaves every wave of the chip eight tickets, else fewer (a multiple of
        // the tiles a wave holds at a time) -- 64 MiB are 16 384 tiles for 4096 waves
        uint64_t range_tiles = n_tiles / ((uint64_t)cus * MT_WAVES * 8u);
        range_tiles = range_tiles / nw * nw;
        if (range_tiles < nw) range_tiles = nw;
        if (range_tiles > MT_RANGE_TILES) range_tiles = MT_RANGE_TILES;
        const uint64_t n_ranges = (n_tiles + range_tiles - 1u) / range_tiles;
        if (n_ranges > 0xffffffffull - 65536ull) return false;
        uint32_t blocks = cus;
        const uint32_t need = (uint32_t)((n_ranges + MT_WAVES - 1u) / MT_WAVES);
        if (blocks > need) blocks = need;
        switch (nw) {
        case 1: launch_mtile<1>(a, blocks, (uint32_t)n_ranges, (uint32_t)range_tiles, st); break;
        case 4: launch_mtile<4>(a, blocks, (uint32_t)n_ranges, (uint32_t)range_tiles, st); break;
        default: launch_mtile<2>(a, blocks, (uint32_t)n_ranges, (uint32_t)range_tiles, st); break;
        }
    }
    if (a.wave_totals)
        hipLaunchKernelGGL(k_mtile_edges_numbered, dim3(1), dim3(64), 0, st, (const uint8_t *)a.text, a.n, a.q, a.mt, a.mk,
                           a.wave_totals, a.strip_prefix);
    else
        hipLaunchKernelGGL(k_mtile_edges, dim3(1), dim3(64), 0, st, (const uint8_t *)a.text, a.n, a.q, a.mt, a.mk);
    return true;
}
