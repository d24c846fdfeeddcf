// agh_mscan.hip -- count-only -f scans (-c / -l with a pattern file; BASELINE config 5) in ONE pass:
// the text is read once and everything up to "this record matches" happens in the sweeping wave,
// out of tables that live in LDS.  (Role of newmgrep.c:463-691 mgrep() + :839-1012 monkey1(); with
// one error per pattern beyond the reference, which ignores -# together with -f, compat.c:34-37.)
//
// Why a kernel of its own: the two-kernel form (k_sweep_multi + k_verify_multi, agh_multi.hip) probes
// one 2^18-bit table per text position -- 5.75 VALU operations and one random ds_read_b32 each (32 banks,
// 3.2 cycles per lane group under conflicts: the LDS is 71 % busy at 3.5 TB/s, profiles/r03_pmc_sweep_multi_k1.json)
// -- and then re-reads the text around 10 M candidate positions per 4 GiB from HBM in a second launch.
//
//   level 1   pair table (agh_device.h): ONE ds_read_b64 answers two neighbouring positions, the row
//             comes from v_mul_u32_u24 (the 24-bit multiply drops the byte that does not belong to the
//             shared 3-gram) + v_and_b32_sdwa, the two bit tests are v_lshrrev_b32_sdwa with the
//             prefix / suffix byte selected in place, v_alignbit pushes the results: 6.5 VALU
//             operations and one LDS read per TWO positions.
//   queue A   lanes whose 16-byte chunk had a level-1 hit queue the chunk itself: its 20 text bytes
//             (16 + the dword behind them), where it lies and the 16 hit bits -- one 24-byte entry per
//             chunk, no per-hit loop, and level 2 needs no second look at the text.
//   level 2   64 queued chunks at a time, one per lane: the 4-gram at a hit position is cut out of the
//             entry's registers and looked up EXACTLY in the gram table (buckets of four 32-bit grams,
//             one ds_read_b128 -- the only dependent LDS access); what survives is a real occurrence of
//             an entry's prefix.  (First version: positions only, the grams re-read from a staged copy
//             of the supertile -- four dependent LDS round trips per hit, as slow as level 1 itself:
//             profiles/r04_perf_c5_breakdown_a.log.)
//   level 3   64 survivors at a time: the entries with that gram (16 bytes each, L2-resident) against
//             the text next to the position -- k = 0: the rest of the pattern; k = 1: the rest of the
//             piece and the other side of the pattern within one edit (side_within_one_edit).  The
//             only global reads on the path are 32 text bytes + the entry, issued by 64 lanes at once.
//   marks     matched positions are queued again; 32..64 at a time look back for the start of their
//             record and enter it into the scan's hash set (lean_insert) like every count-only engine.
//
// Persistent grid: one workgroup per CU (the tables fill most of the LDS), waves draw 256 KiB ranges
// from a ticket counter.  A wave covers positions 1..N of its range (position 0 of a chunk is
// position 16 of the lane before it); positions 0..7 and the last 23 of the text: k_mscan_edges.
#include "agh_multi_inl.h"

// the three queues of a wave are rings of 128 entries: drained 64 at a time, a step adds at most 64
#define MS_RING 128u
#define MS_RING_B 128u
#define MS_QA_DW 6u                 // queue A entry: w0 w1 | w2 w3 | w4 meta (three ds_write_b64)

// Measured and removed (round 5, one box, profiles/r05_ab_c5_variants.log; 4 GiB, config 5: shipped 1.27-1.30 ms):
// a neighbour-byte filter in front of level 3 (1.39-1.48), a fifth-byte mask in level 2 with 2^12 rows (1.50-1.53),
// level 3 in two halves with one batch in flight (1.29-1.31), 2^12 rows alone (1.44-1.48).
typedef uint32_t ms_qb_t;

template <int WAVES, int RB>
struct ms_shared {
    uint2 ptab[1u << RB];
    uint32_t gtab[AGH_MS_GSLOTS];
    uint2 qa[WAVES][MS_RING * MS_QA_DW / 2];
    ms_qb_t qb[WAVES][MS_RING_B];
    uint64_t qm[WAVES][MS_RING];
};

// The sixteen positions 1..16 of one 16-byte chunk (w[4] = the dword behind it).  Pair i shares the
// 3-gram at byte s = 2i + 2: position s - 1 is (t[s-1], 3-gram), position s is (3-gram, t[s+3]).
// Two steps, so that the table reads of the next chunk are in flight while this one is evaluated:
// ms_probe_issue computes the eight rows and starts the reads, ms_probe_take tests the sixteen bits.
template <int RB>
__device__ __forceinline__ void ms_probe_issue(const uint32_t (&w)[5], const uint8_t *ptab8, uint32_t (&y)[8], uint2 (&W)[8])
{
    constexpr uint32_t amask = ((1u << RB) - 1u) << 3;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int s = 2 * i + 2, d = s >> 2;
        y[i] = (s & 2) ? __builtin_amdgcn_alignbyte(w[d + 1], w[d], 2) : w[d];
        const uint32_t a = (__umul24(y[i], AGH_MS_C) >> 16) & amask;
        W[i] = *reinterpret_cast<const uint2 *>(ptab8 + a);
    }
}

__device__ __forceinline__ void ms_probe_take(const uint32_t (&w)[5], const uint32_t (&y)[8], const uint2 (&W)[8], uint32_t &acc)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int s = 2 * i + 2, d = s >> 2;
        uint32_t r1;
        if (s & 2)      // the byte in front of the 3-gram is byte 1 of the same dword
            asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
                : "=v"(r1) : "v"(w[d]), "v"(W[i].x));
        else            // ... or the last byte of the dword before.  (& 31: the hardware takes five bits
            r1 = W[i].x >> ((w[d - 1] >> 24) & 31u);     // anyway, but a C shift by >= 32 is undefined -- with
        const uint32_t r2 = W[i].y >> ((y[i] >> 24) & 31u); // bytes known to be >= 0x20 (-i) the tests were folded away)
        acc = __builtin_amdgcn_alignbit(r1, acc, 1);
        acc = __builtin_amdgcn_alignbit(r2, acc, 1);
    }
}

typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));

// wave-uniform bookkeeping (queue lengths) pinned to scalar registers: left to itself the compiler
// treats the flat loop below as divergent and runs it on exec masks
__device__ __forceinline__ uint32_t ms_uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// first dword of the following lane's chunk (DPP wave_shl:1; lane 63 has no source lane and keeps
// `wrap`, the dword behind the strip)
__device__ __forceinline__ uint32_t ms_next_lane_dword(uint32_t x, uint32_t wrap)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)wrap, (int)x, 0x130, 0xf, 0xf, false);
}

// k = 0: does the entry (a whole pattern of 4..15 bytes) stand at j?  T0 = text[j-8, j+8), T1 = text[j+8, j+24)
__device__ __forceinline__ bool ms_match_k0(const uint4 ent, const uint32_t (&T)[8])
{
    const uint32_t len = ent.w >> 24;
    auto mask_of = [](uint32_t nb) -> uint32_t { return nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u); };
    uint32_t diff = T[2] ^ ent.x;
    diff |= (T[3] ^ ent.y) & mask_of(len - 4u);
    diff |= (T[4] ^ ent.z) & mask_of(len > 8u ? len - 8u : 0u);
    diff |= (T[5] ^ ent.w) & mask_of(len > 12u ? len - 12u : 0u);
    return diff == 0u;
}

// k = 1: the piece (4..7 bytes) stands at j and the other side of its pattern (L <= 7 bytes, stored
// nearest byte first) lies within one edit of the text next to it
__device__ __forceinline__ bool ms_match_k1(const uint4 ent, const uint32_t (&T)[8], uint32_t delim)
{
    const uint32_t meta = ent.z >> 24, L = meta & 7u, tl = (meta >> 3) & 3u;
    uint32_t diff = T[2] ^ ent.x;
    diff |= (T[3] ^ ent.w) & ((1u << (8u * tl)) - 1u);          // bytes 4 .. len-1 of the piece
    if (diff) return false;
    const uint64_t B = (uint64_t)ent.y | ((uint64_t)(ent.z & 0xffffffu) << 32);
    uint64_t S;
    if (meta & 32u) {                           // the head of the pattern in front of the piece
        S = __builtin_bswap64((uint64_t)T[0] | ((uint64_t)T[1] << 32));
    } else {                                    // the rest of the pattern behind it: text[j + 4 + tl ...)
        const uint32_t lo = __builtin_amdgcn_alignbyte(T[4], T[3], tl);
        const uint32_t hi = __builtin_amdgcn_alignbyte(T[5], T[4], tl);
        S = (uint64_t)lo | ((uint64_t)hi << 32);
    }
    return side_within_one_edit(S, B, L, delim);
}

// strip st of the text, bytes at and behind n replaced by the filler; strips behind the text: filler
__device__ __noinline__ uint4 ms_load_strip(const uint4 *text, uint64_t n, uint64_t st, uint32_t fill4)
{
    const uint64_t off = (st << AGH_STRIP_SHIFT) + (uint64_t)lane_id() * 16u;
    uint4 v = make_uint4(fill4, fill4, fill4, fill4);
    if (off < n) {
        v = text[off >> 4];
        if (off + 16 > n) v = mask_tail(v, (int)(n - off), fill4);
    }
    return v;
}

template <int WAVES, int RB, bool FOLD, int K>
__global__ __launch_bounds__(WAVES * 64) void k_mscan(const uint4 *__restrict__ text, uint64_t n, uint32_t delim,
                                                      agh_mscan_dev ms, agh_marks mk,
                                                      uint32_t *__restrict__ ticket, uint32_t n_ranges, uint32_t dbg)
{
    // dbg (AGH_MSCAN_DBG, measurements only): 1 drop level-3 candidates, 2 level 3 without its text loads,
    // 4 drop level-1 chunks, 8 level 3 after every supertile, 16 non-temporal loads
    __shared__ __attribute__((aligned(16))) ms_shared<WAVES, RB> sh;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(ms.ptab);
        uint4 *dst = reinterpret_cast<uint4 *>(sh.ptab);
        for (uint32_t i = threadIdx.x; i < (8u << RB) / 16u; i += WAVES * 64) dst[i] = src[i];
        const uint4 *gs = reinterpret_cast<const uint4 *>(ms.gtab);
        uint4 *gd = reinterpret_cast<uint4 *>(sh.gtab);
        for (uint32_t i = threadIdx.x; i < AGH_MS_GSLOTS / 4u; i += WAVES * 64) gd[i] = gs[i];
        __syncthreads();
    }
    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const uint8_t *ptab8 = reinterpret_cast<const uint8_t *>(sh.ptab);
    const uint4 *gt4 = reinterpret_cast<const uint4 *>(sh.gtab);
    uint2 *qa = sh.qa[wib];
    ms_qb_t *qb = sh.qb[wib];
    uint64_t *qm = sh.qm[wib];
    const uint8_t *text8 = reinterpret_cast<const uint8_t *>(text);
    const uint64_t n_strips = (n + AGH_STRIP - 1) >> AGH_STRIP_SHIFT, n_full = n >> AGH_STRIP_SHIFT;
    const uint64_t n_dw = ((n + 15) & ~(uint64_t)15) / 4;      // readable dwords
    const uint32_t fold4 = FOLD ? 0x20202020u : 0u;
    const uint32_t fill4 = delim * 0x01010101u;               // no entry holds the delimiter byte
    // ring heads and lengths (wave-uniform)
    uint32_t hA = 0, qnA = 0, hB = 0, qnB = 0, hM = 0, qnM = 0, ncand = 0;
    uint64_t range_base = 0;
    auto rank_of = [](uint64_t mask) -> uint32_t {
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    };

    // ---- matched positions -> record starts -> hash set -------------------------------------------
    auto stage_m = [&]() {
        const uint32_t take = ms_uni(qnM < 64u ? qnM : 64u);
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane < take) {
            const uint64_t st = lean_record_start(text8, qm[(hM + (uint32_t)lane) & (MS_RING - 1u)], delim, mk);
            if (st != ~0ull) lean_insert(mk, st);
        }
        hM = ms_uni((hM + take) & (MS_RING - 1u));
        qnM = ms_uni(qnM - take);
    };

    // ---- level 3: real occurrences of an entry's first four bytes --------------------------------
    // (positions closer than 8 bytes to the start or 24 to the end of the text: k_mscan_edges)
    auto stage_b = [&]() {
        const uint32_t take = ms_uni(qnB < 64u ? qnB : 64u);
        __builtin_amdgcn_wave_barrier();
        bool matched = false;
        uint64_t j = 0;
        if ((uint32_t)lane < take) {
            const uint32_t e = qb[(hB + (uint32_t)lane) & (MS_RING_B - 1u)];
            j = range_base + (e >> 12);
            if (j >= 8u && j + 24u <= n && !(dbg & 1u)) {
                // the gram's first entry sits at the gram's own slot (its gram word says where further entries with
                // the same gram lie and how many): ONE load decides most candidates -- round 5 read a directory word
                // first and the entry behind it, two dependent trips to the L2 with the wave's stream waiting
                uint4 ent = ms.ment[e & (AGH_MS_GSLOTS - 1u)];
                u32x4_a1 t0 = {0u, 0u, 0u, 0u}, t1 = {0u, 0u, 0u, 0u};
                if (!(dbg & 2u)) {
                    t0 = *reinterpret_cast<const u32x4_a1 *>(text8 + j - 8);
                    t1 = *reinterpret_cast<const u32x4_a1 *>(text8 + j + 8);
                }
                uint32_t T[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
                if (FOLD) {
#pragma unroll
                    for (int d = 0; d < 8; ++d) T[d] = swar_lower(T[d]);
                }
                const uint32_t more = ent.x;                            // (further entries << 8) | how many
                ent.x = T[2];                                           // (level 2 found the gram itself)
                matched = K == 0 ? ms_match_k0(ent, T) : ms_match_k1(ent, T, delim);
                const uint32_t first = AGH_MS_GSLOTS + (more >> 8), cnt = more & 0xffu;
                for (uint32_t i = 0; i < cnt && !matched; ++i) {
                    const uint4 en = ms.ment[first + i];
                    matched = K == 0 ? ms_match_k0(en, T) : ms_match_k1(en, T, delim);
                }
            }
        }
        hB = ms_uni((hB + take) & (MS_RING_B - 1u));
        qnB = ms_uni(qnB - take);
        ncand = ms_uni(ncand + take);
        const uint64_t mb = __ballot(matched);
        if (mb) {
            if (matched) qm[(hM + qnM + rank_of(mb)) & (MS_RING - 1u)] = j;
            qnM = ms_uni(qnM + (uint32_t)__popcll(mb));
        }
    };

    // ---- one supertile = strips s .. s+3, nx3 = the first dword of strip s+4; st_rel = its number
    // inside the wave's range ------------------------------------------------------------------------
    auto supertile = [&](uint4 v0, uint4 v1, uint4 v2, uint4 v3, uint32_t nx3, uint32_t st_rel, bool range_ends) {
        uint32_t lo = 0, hi = 0;
        const uint32_t w0[5] = {v0.x | fold4, v0.y | fold4, v0.z | fold4, v0.w | fold4,
                                ms_next_lane_dword(v0.x, (uint32_t)__builtin_amdgcn_readlane((int)v1.x, 0)) | fold4};
        const uint32_t w1[5] = {v1.x | fold4, v1.y | fold4, v1.z | fold4, v1.w | fold4,
                                ms_next_lane_dword(v1.x, (uint32_t)__builtin_amdgcn_readlane((int)v2.x, 0)) | fold4};
        const uint32_t w2[5] = {v2.x | fold4, v2.y | fold4, v2.z | fold4, v2.w | fold4,
                                ms_next_lane_dword(v2.x, (uint32_t)__builtin_amdgcn_readlane((int)v3.x, 0)) | fold4};
        const uint32_t w3[5] = {v3.x | fold4, v3.y | fold4, v3.z | fold4, v3.w | fold4,
                                ms_next_lane_dword(v3.x, nx3) | fold4};
        // ---- level 1.  The reads of chunk c+1 are issued before chunk c is evaluated.  The hit word is
        // pinned after every chunk: left alone, the pushes sink behind all 32 table reads of the
        // supertile, whose 64 result registers then stay live (147 VGPRs instead of ~100).
        {
            uint32_t ya[8], yb[8];
            uint2 Wa[8], Wb[8];
            ms_probe_issue<RB>(w0, ptab8, ya, Wa);
            ms_probe_issue<RB>(w1, ptab8, yb, Wb);
            __builtin_amdgcn_sched_barrier(0);
            ms_probe_take(w0, ya, Wa, lo);
            asm volatile("" : "+v"(lo));
            ms_probe_issue<RB>(w2, ptab8, ya, Wa);
            __builtin_amdgcn_sched_barrier(0);
            ms_probe_take(w1, yb, Wb, lo);
            asm volatile("" : "+v"(lo));
            ms_probe_issue<RB>(w3, ptab8, yb, Wb);
            __builtin_amdgcn_sched_barrier(0);
            ms_probe_take(w2, ya, Wa, hi);
            asm volatile("" : "+v"(hi));
            __builtin_amdgcn_sched_barrier(0);
            ms_probe_take(w3, yb, Wb, hi);
            asm volatile("" : "+v"(hi));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (dbg & 4u) lo = hi = 0;
        // ---- queue A: the chunks of a strip that have a hit, straight-line; level 2 runs whenever 64
        // chunks are waiting (about every second supertile), level 3 whenever 64 survivors are ----------
        auto push_strip = [&](const uint32_t (&w)[5], uint32_t m16, uint32_t u) {
            const uint64_t bal = __ballot(m16 != 0u);
            if (bal) {
                if (m16) {
                    uint2 *dst = qa + ((hA + qnA + rank_of(bal)) & (MS_RING - 1u)) * (MS_QA_DW / 2u);
                    const uint32_t meta = (((st_rel << 2 | u) << 6 | (uint32_t)lane) << 16) | m16;
                    dst[0] = make_uint2(w[0], w[1]);
                    dst[1] = make_uint2(w[2], w[3]);
                    dst[2] = make_uint2(w[4], meta);
                }
                qnA = ms_uni(qnA + (uint32_t)__popcll(bal));
            }
        };
        // level 2 on (up to) 64 queued chunks, one per lane: every hit position's 4-gram is cut out of the
        // chunk and looked up exactly; survivors go to queue B
        auto level2 = [&]() {
            const uint32_t take = ms_uni(qnA < 64u ? qnA : 64u);
            __builtin_amdgcn_wave_barrier();
            uint32_t m = 0, pos_rel = 0, e[5] = {0u, 0u, 0u, 0u, 0u};
            if ((uint32_t)lane < take) {
                const uint2 *src = qa + ((hA + (uint32_t)lane) & (MS_RING - 1u)) * (MS_QA_DW / 2u);
                const uint2 a0 = src[0], a1 = src[1], a2 = src[2];
                e[0] = a0.x; e[1] = a0.y; e[2] = a1.x; e[3] = a1.y; e[4] = a2.x;
                m = a2.y & 0xffffu;
                pos_rel = (a2.y >> 16) << 4;            // (supertile, strip, lane) = the chunk's number in the range
            }
            hA = ms_uni((hA + take) & (MS_RING - 1u));
            qnA = ms_uni(qnA - take);
            do {
                if (qnB > MS_RING_B - 64u) {            // room for the survivors of this round
                    stage_b();
                    if (qnM >= 64u) stage_m();
                }
                const bool act = m != 0u;
                const uint32_t p = act ? (uint32_t)__ffs((int)m) : 1u;      // position 1..16 in the chunk
                m &= m - 1u;
                // bytes [p, p+4) of the 20: the dword pair by bits 3 and 2 of p, the bytes by v_alignbyte
                const bool p8 = (p & 8u) != 0u, p4 = (p & 4u) != 0u, p16 = (p & 16u) != 0u;
                const uint32_t x0 = p8 ? e[2] : e[0], x1 = p8 ? e[3] : e[1], x2 = p8 ? e[4] : e[2];
                const uint32_t glo = p16 ? e[4] : (p4 ? x1 : x0), ghi = p4 ? x2 : x1;
                const uint32_t g = __builtin_amdgcn_alignbyte(ghi, glo, p & 3u);    // (entries hold folded text)
                // two-choice table: the gram sits in one of two buckets, both read at once -- one LDS round
                // trip per hit, no chain of full buckets to follow
                const uint32_t gh = agh_ms_ghash(g), b1 = AGH_MS_GB1(gh), b2 = AGH_MS_GB2(gh);
                const uint4 G1 = gt4[b1], G2 = gt4[b2];
                const int h1 = G1.x == g ? 0 : (G1.y == g ? 1 : (G1.z == g ? 2 : (G1.w == g ? 3 : -1)));
                const int h2 = G2.x == g ? 0 : (G2.y == g ? 1 : (G2.z == g ? 2 : (G2.w == g ? 3 : -1)));
                const bool found = act && (h1 >= 0 || h2 >= 0);
                const uint64_t fb = __ballot(found);
                if (fb) {
                    const uint32_t slot = h1 >= 0 ? b1 * 4u + (uint32_t)h1 : b2 * 4u + (uint32_t)h2;
                    if (found) qb[(hB + qnB + rank_of(fb)) & (MS_RING_B - 1u)] = ((pos_rel + p) << 12) | slot;
                    qnB = ms_uni(qnB + (uint32_t)__popcll(fb));
                }
            } while (__ballot(m != 0u));
        };
        // (a loop that stays a loop: unrolled, level 2 and 3 would be in the code four times)
#pragma clang loop unroll(disable)
        for (uint32_t u = 0; u < 4u; u = ms_uni(u + 1u)) {
            switch (u) {
            case 0: push_strip(w0, lo & 0xffffu, 0u); break;
            case 1: push_strip(w1, lo >> 16, 1u); break;
            case 2: push_strip(w2, hi & 0xffffu, 2u); break;
            default: push_strip(w3, hi >> 16, 3u); break;
            }
            while (qnA >= 64u || (u == 3u && range_ends && qnA)) level2();
        }
        while (qnB >= 64u || ((range_ends || (dbg & 8u)) && qnB)) {
            stage_b();
            if (qnM >= 64u) stage_m();
        }
    };

    auto load_strip = [&](uint64_t st) -> uint4 { return ms_load_strip(text, n, st, fill4); };
    auto first_dword_of = [&](uint64_t st) -> uint32_t {
        const uint64_t i = st * 256u;
        return i < n_dw ? reinterpret_cast<const uint32_t *>(text)[i] : fill4;
    };

    const uint32_t total_waves = gridDim.x * (uint32_t)WAVES;
    uint32_t r = blockIdx.x * (uint32_t)WAVES + wib;
    while (r < n_ranges) {
        const uint64_t s0 = (uint64_t)r * AGH_WAVE_STRIPS;
        uint64_t s1 = s0 + AGH_WAVE_STRIPS;
        if (s1 > n_strips) s1 = n_strips;
        range_base = s0 << AGH_STRIP_SHIFT;
        // full strips, a multiple of four (every range but the last one of the text): straight loads;
        // else bytes behind the text become filler.  The next supertile is in flight either way.
        const bool plain = s1 <= n_full && ((s1 - s0) & 3u) == 0u;
        // TWO supertiles in flight (round 6): level 1 of supertile s needs the first dword of supertile s + 4 (lane 63's
        // last positions reach into it).  With one supertile ahead that dword came out of loads issued a few
        // instructions earlier -- s_waitcnt vmcnt(3) right behind them, a full memory latency per supertile with the
        // wave's own stream stalled (the ISA of round 5's kernel: profiles/r06_mscan_prefetch.log).  Now it was
        // loaded a whole supertile ago.
        auto load4 = [&](uint64_t st, uint4 &a0, uint4 &a1, uint4 &a2, uint4 &a3) {
            if (plain) {
                const uint4 *p = text + st * 64 + lane;
                // Plain loads, not the non-temporal ones of the other sweeps: level 3 reads the text around ~10 M
                // candidates per 4 GiB again a few microseconds later, and lines that came in non-temporally are
                // gone by then (HBM traffic 1.30 x the text, 1.31 vs 1.15 ms per 4 GiB: profiles/r04_perf_c5_dbg_a.log)
                if (!(dbg & 16u)) { a0 = p[0]; a1 = p[64]; a2 = p[128]; a3 = p[192]; }
                else { a0 = ld_stream(p); a1 = ld_stream(p + 64); a2 = ld_stream(p + 128); a3 = ld_stream(p + 192); }
            } else {
                a0 = load_strip(st); a1 = load_strip(st + 1); a2 = load_strip(st + 2); a3 = load_strip(st + 3);
            }
        };
        uint4 c0, c1, c2, c3, n0, n1, n2, n3;
        load4(s0, c0, c1, c2, c3);
        n0 = c0; n1 = c1; n2 = c2; n3 = c3;
        if (s0 + 4 < s1) load4(s0 + 4, n0, n1, n2, n3);
        for (uint64_t s = s0; s < s1; s += 4) {
            uint4 m0 = n0, m1 = n1, m2 = n2, m3 = n3;
            if (s + 8 < s1) load4(s + 8, m0, m1, m2, m3);
            const uint32_t nx3 = s + 4 < s1 ? (uint32_t)__builtin_amdgcn_readlane((int)n0.x, 0) : first_dword_of(s + 4);
            supertile(c0, c1, c2, c3, nx3, (uint32_t)((s - s0) >> 2), s + 4 >= s1);
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            n0 = m0; n1 = m1; n2 = m2; n3 = m3;
        }
        // (queues A and B are empty here: their positions are relative to the range)
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(ticket, 1u);
        r = total_waves + (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    }
    while (qnM) stage_m();
    if (lane == 0 && ncand) atomicAdd(&mk.counters[AGH_C_CAND], ncand);
}

// The positions k_mscan leaves out: fewer than 8 bytes in front of them or fewer than 24 behind (its
// text window would leave the text).  One lane per position through the general verifier of the
// two-kernel form (bounds, the virtual bytes around the text, byte-wise windows).
template <int K>
__global__ __launch_bounds__(64) void k_mscan_edges(const uint8_t *__restrict__ text8, uint64_t n, agh_dev_query q,
                                                    agh_multi_dev mt, agh_marks mk)
{
    const uint32_t lane = (uint32_t)lane_id();
    uint64_t j = ~0ull;
    if (lane < 8u) j = lane;
    else if (lane < 31u && n >= 23u + 8u) j = n - 23u + (lane - 8u);     // (texts below 31 bytes: all of it above)
    else if (lane < 31u && lane < n) j = lane;
    if (j < n) mp_verify_at<true, K>(text8, n, q, mt, j, 0u, mk);
}

// ---------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------
template <int WAVES, int RB>
static void launch_mscan_cfg(const agh_mscan_args &a, uint32_t n_ranges, hipStream_t st)
{
    uint32_t blocks = a.n_cu ? a.n_cu : 256u;
    const uint32_t need = (n_ranges + WAVES - 1) / WAVES;
    if (blocks > need) blocks = need;
    const bool fold = a.q.fold != 0;
#define AGH_MS_LAUNCH(F, KK)                                                                             \
    hipLaunchKernelGGL((k_mscan<WAVES, RB, F, KK>), dim3(blocks), dim3(WAVES * 64), 0, st, (const uint4 *)a.text, \
                       a.n, a.q.delim, a.ms, a.mk, a.ticket, n_ranges, a.dbg)
    if (a.q.k == 0) { if (fold) AGH_MS_LAUNCH(true, 0); else AGH_MS_LAUNCH(false, 0); }
    else { if (fold) AGH_MS_LAUNCH(true, 1); else AGH_MS_LAUNCH(false, 1); }
#undef AGH_MS_LAUNCH
    if (a.q.k == 0)
        hipLaunchKernelGGL((k_mscan_edges<0>), dim3(1), dim3(64), 0, st, (const uint8_t *)a.text, a.n, a.q, a.mt, a.mk);
    else
        hipLaunchKernelGGL((k_mscan_edges<1>), dim3(1), dim3(64), 0, st, (const uint8_t *)a.text, a.n, a.q, a.mt, a.mk);
}

// false: no instance for this query (k > 1, delimiter bitmap) -- the caller takes the two-kernel form
bool agh_launch_mscan(const agh_mscan_args &a, hipStream_t st)
{
    if (a.q.k > 1 || a.q.mb || a.q.fq != 4 || !a.n) return false;
    const uint64_t n_strips = (a.n + AGH_STRIP - 1) >> AGH_STRIP_SHIFT;
    const uint64_t n_ranges = (n_strips + AGH_WAVE_STRIPS - 1) / AGH_WAVE_STRIPS;
    if (n_ranges > 0xffffffffull - 65536ull) return false;
    if (a.ms.rb == 13u) launch_mscan_cfg<16, 13>(a, (uint32_t)n_ranges, st);
    else
    if (a.ms.rb == 12u) launch_mscan_cfg<16, 12>(a, (uint32_t)n_ranges, st);
    else return false;
    return true;
}
