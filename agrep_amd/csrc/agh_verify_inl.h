// agh_verify_inl.h -- the k-error automaton and the per-candidate window verification, shared
// used by the verify kernels (agh_scan.hip).
#pragma once
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
        atomicMax(&mk.counters[AGH_C_BM_OVERFLOW], 1u);     // (2: more than 2^32 records, k_scan_fixup)
        return;
    }
    const uint32_t bit = 1u << (r & 31u);
    // (dense hit sets mark the same record again and again: a plain load settles those)
    if (__atomic_load_n(&mk.bitmap[r >> 5], __ATOMIC_RELAXED) & bit) return;
    uint32_t old = atomicOr(&mk.bitmap[r >> 5], bit);
    if (mk.rec_pos && !(old & bit)) mk.rec_pos[r] = e;
}

// ---- lean scans: a record is identified by the offset of its first byte ---------------------
__device__ __forceinline__ void lean_insert(const agh_marks &mk, uint64_t rec_start)
{
    const uint64_t key = rec_start + 1;         // 0 = empty slot
    uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & mk.hashset_mask;
    for (int probe = 0; probe < 64; ++probe) {
        // most inserts repeat a record already in the set (several candidate windows per
        // occurrence): a plain load settles those without an L2 atomic
        uint64_t old = __atomic_load_n(&mk.hashset[slot], __ATOMIC_RELAXED);
        if (old == key) return;
        if (old == 0) {
            old = atomicCAS((unsigned long long *)&mk.hashset[slot], 0ull, (unsigned long long)key);
            if (old == 0) {
                mk.counters[AGH_C_ANYHIT] = 1u; // -l scans stop at the first part with a hit
                return;
            }
            if (old == key) return;
        }
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
    // a record of more than AGH_LEAN_BACK_CAP bytes in front of a match: noted for k_resolve_giveups (a look-back
    // without a limit after the scan) -- only a full list sends the whole segment to the numbered pipeline
    if (mk.giveups) {
        const uint32_t idx = atomicAdd(&mk.counters[AGH_C_GIVEUPS], 1u);
        if (idx < mk.giveup_cap) {
            mk.giveups[idx] = stop;             // nothing ends a record in [stop, pos): the search goes on from here
            return ~0ull;
        }
    }
    mk.counters[AGH_C_LEAN_FALLBACK] = 1u;
    return ~0ull;
}

// Multi-byte delimiters: the same through the delimiter-end bitmap.
__device__ __forceinline__ uint64_t lean_record_start_mb(const uint64_t *__restrict__ dbm,
                                                         uint64_t pos, const agh_marks &mk)
{
    const int64_t d = dbm_prev(dbm, pos, AGH_LEAN_BACK_CAP);
    if (d >= 0) return (uint64_t)d + 1;
    if (d == -1) return 0;
    mk.counters[AGH_C_LEAN_FALLBACK] = 1u;
    return ~0ull;
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
    // asearch1.c:88-97: levels are accumulated cost; insertion comes from level e-ci,
    // substitution from e-cs, deletion from the NEW level e-cd; positions whose no_err bit is
    // clear cannot be entered through an error (maskgen.c:80-95 "<...>").  The start state is
    // shifted in only when a source level exists (the reference's dummy levels are zero words).
    __device__ __forceinline__ bool step_general(WT cm, WT finalbit, uint32_t ci, uint32_t cs,
                                                 uint32_t cd, WT no_err)
    {
        WT old[K + 1], nw[K + 1];
#pragma unroll
        for (int e = 0; e <= K; ++e) old[e] = R[e];
        nw[0] = ((old[0] << 1) | (WT)1) & cm;
#pragma unroll
        for (int e = 1; e <= K; ++e) {
            WT ins = 0, via = 0;
            bool have = false;
#pragma unroll
            for (int s = 0; s < e; ++s) {
                const uint32_t d = (uint32_t)(e - s);
                if (d == ci) ins = old[s];
                if (d == cs) { via |= old[s]; have = true; }
                if (d == cd) { via |= nw[s]; have = true; }
            }
            const WT err = have ? (((via << 1) | (WT)1) & no_err) : (WT)0;
            nw[e] = (((old[e] << 1) | (WT)1) & cm) | ins | err;
        }
#pragma unroll
        for (int e = 0; e <= K; ++e) R[e] = nw[e];
        return (R[K] & finalbit) != 0;
    }
    template <bool GEN>
    __device__ __forceinline__ bool step_q(WT cm, WT finalbit, const agh_dev_query &q)
    {
        if (GEN) return step_general(cm, finalbit, q.ci, q.cs, q.cd, (WT)q.no_err);
        return step(cm, finalbit);
    }
};

// ---------------------------------------------------------------------------------------
// verify: one workgroup per AGH_VGROUP sweep-wave slices, one lane per candidate sample
// ---------------------------------------------------------------------------------------
// Byte-wise reference walk of one window: used for windows at the head / tail of the text
// (virtual head byte, appended delimiter) where the register fast path does not apply.
// Where the last, unterminated record ends once the delimiter is appended at EOF: the appended
// bytes may complete a partial delimiter the real text ended with (text "...x\n" with
// delimiter "\n\n": the record ends in front of that "\n").  Equals n for 1-byte delimiters.
__device__ __forceinline__ uint64_t virtual_close_start(const uint8_t *__restrict__ text,
                                                        uint64_t n, const agh_dev_query &q,
                                                        const uint64_t *__restrict__ dbm)
{
    if (q.dlen <= 1) return n;
    uint32_t ds = 0;
    uint64_t from = n >= q.dlen - 1 ? n - (q.dlen - 1) : 0;
    const int64_t last = dbm_prev(dbm, n, q.dlen);
    if (last >= 0 && (uint64_t)last + 1 > from) from = (uint64_t)last + 1;
    if (from == 0) ds = delim_class(q, q.head_byte) & 1u;   // the virtual byte in front
    for (uint64_t i = from; i < n; ++i) ds = ((ds << 1) | 1u) & delim_class(q, text[i]);
    const uint32_t endbit = 1u << (q.dlen - 1);
    for (uint32_t j = 0; j < q.dlen; ++j) {
        ds = ((ds << 1) | 1u) & delim_class(q, q.dbytes[j]);
        if (ds & endbit) {
            const uint64_t endpos = n + j + 1;              // one past the completing byte
            return endpos >= q.dlen ? endpos - q.dlen : 0;
        }
    }
    return n;
}

// The delimiter appended at EOF (asearch.c:87-91), byte by byte, for any delimiter length: the
// automaton sees the bytes as text, the little delimiter automaton `ds` completes whatever
// partial delimiter the real text ended with.  A / seen / rec / rstart continue from the walk.
template <typename WT, int K, bool LEAN, bool GEN = false>
__device__ __forceinline__ void feed_virtual_tail(const uint8_t *__restrict__ text, uint64_t n,
                                                  const agh_dev_query &q, const WT *lmask,
                                                  const uint64_t *__restrict__ dbm,
                                                  Automaton<WT, K> &A, bool seen, uint32_t rec,
                                                  uint64_t rstart, const agh_marks &mk)
{
    const WT finalbit = (WT)1 << (q.m - 1);
    uint32_t ds = 0;
    if (q.dlen > 1) {
        // progress of a partial delimiter at the end of the real text: only the last dlen-1
        // bytes after the last selected delimiter end matter
        uint64_t from = n >= q.dlen - 1 ? n - (q.dlen - 1) : 0;
        const int64_t last = dbm_prev(dbm, n, q.dlen);
        if (last >= 0 && (uint64_t)last + 1 > from) from = (uint64_t)last + 1;
        if (from == 0) ds = delim_class(q, q.head_byte) & 1u;   // the virtual byte in front
        for (uint64_t i = from; i < n; ++i) ds = ((ds << 1) | 1u) & delim_class(q, text[i]);
    }
    const uint32_t endbit = 1u << (q.dlen - 1);
    for (uint32_t j = 0; j < q.dlen; ++j) {
        const uint32_t c = q.dbytes[j];
        if (A.template step_q<GEN>(lmask[c], finalbit, q) && !seen) {
            seen = true;
            if (LEAN) lean_insert(mk, rstart); else mark_record(mk, rec, n);
        }
        ds = ((ds << 1) | 1u) & delim_class(q, c);
        if (ds & endbit) {
            ds = 0;
            A.reset();
            ++rec;
            rstart = n + j + 1;
            seen = false;
            if (A.template step_q<GEN>(lmask[c], finalbit, q)) {
                seen = true;
                if (LEAN) lean_insert(mk, rstart); else mark_record(mk, rec, n);
            }
        }
    }
}

template <typename WT, int K, bool LEAN, bool GEN = false>
__device__ __noinline__ void verify_window_slow(const uint8_t *__restrict__ text, uint64_t n,
                                                const agh_dev_query &q, const WT *lmask,
                                                const uint64_t *__restrict__ dbm,
                                                uint64_t ws, uint64_t we, uint64_t anchor,
                                                uint32_t rc_anchor, const agh_marks &mk)
{
    const WT finalbit = (WT)1 << (q.m - 1);
    const bool mb = q.mb != 0;
    uint32_t rec = 0;
    uint64_t rstart = 0;                        // LEAN: first byte of the current record
    if (LEAN) {
        rstart = mb ? lean_record_start_mb(dbm, ws, mk) : lean_record_start(text, ws, q.delim, mk);
        if (rstart == ~0ull) return;
    } else {
        // delimiters in [ws, anchor): the anchor's record number is known, ws's is derived
        uint32_t back = 0;
        if (mb) back = dbm_count(dbm, ws, anchor);
        else for (uint64_t i = ws; i < anchor; ++i) back += (text[i] == q.delim);
        rec = rc_anchor - back;
    }
    Automaton<WT, K> A;
    A.reset();
    bool seen = false;
    if (ws == 0) A.template step_q<GEN>(lmask[q.head_byte], finalbit, q);
    for (uint64_t i = ws; i < we; ++i) {
        const uint32_t c = text[i];
        if (A.template step_q<GEN>(lmask[c], finalbit, q) && !seen) {
            seen = true;
            if (LEAN) lean_insert(mk, rstart); else mark_record(mk, rec, i);
        }
        if (mb ? dbm_bit(dbm, i) != 0 : c == q.delim) {
            A.reset();
            ++rec;
            rstart = i + 1;
            seen = false;
            if (A.template step_q<GEN>(lmask[c], finalbit, q)) {
                seen = true;
                if (LEAN) lean_insert(mk, rstart); else mark_record(mk, rec, i + 1);
            }
        }
    }
    if (we == n && q.tail_virtual)
        feed_virtual_tail<WT, K, LEAN, GEN>(text, n, q, lmask, dbm, A, seen, rec, rstart, mk);
}

// Unaligned 16-byte view of the text (gfx9+ global loads accept any byte address).
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));


// Everything a lane needs to verify candidates of one query (built once per kernel).
template <typename WT, int K>
struct VerifyCtx {
    const uint8_t *text;
    uint64_t n, n16;
    const agh_dev_query *q;  // points at the kernel argument (kept out of scratch)
    const WT *lmask;         // LDS copy of the 256 position masks
    const uint64_t *dbm;     // multi-byte delimiters: delimiter-end bitmap
    WT finalbit;
    uint32_t Lw, tailw, span;
    uint32_t jshift;         // candidate entry -> byte offset: 2 (dword index), 1 for H == 2 samples
    const uint64_t *gtab;    // lean scans: per hash slot (gram, first/last offset) or NULL
    uint32_t tspan;          // window length when the gram's offset is known: m + 2k + spread
    Automaton<WT, K> RF;     // state right after a record boundary (reset + re-fed delimiter)
    bool rf_hit;
    const agh_marks *mk;     // likewise
};

template <typename WT, int K, bool GEN = false>
__device__ __forceinline__ void verify_ctx_init(VerifyCtx<WT, K> &c, const uint8_t *text,
                                                uint64_t n, const agh_dev_query &q,
                                                const WT *lmask, const agh_marks &mk,
                                                const uint64_t *dbm)
{
    c.text = text;
    c.dbm = dbm;
    c.n = n;
    c.n16 = (n + 15) & ~(uint64_t)15;
    c.q = &q;
    c.lmask = lmask;
    c.mk = &mk;
    c.finalbit = (WT)1 << (q.m - 1);
    c.Lw = (uint32_t)(q.m + q.k + 1) > 16u ? (uint32_t)(q.m + q.k + 1) : 16u;
    c.tailw = (uint32_t)(q.fq + q.m + q.k);
    c.span = c.Lw + c.tailw;                    // <= 16 * NCH by construction
    c.jshift = q.fh == 2 ? 1u : 2u;
    c.gtab = nullptr;
    c.tspan = 0;
    c.RF.reset();
    c.rf_hit = c.RF.template step_q<GEN>(lmask[q.delim], c.finalbit, q);   // asearch.c:175-186
}

// Where the window of a candidate lies.  mode 0: nothing to do (beyond the text, or the sample's
// gram is not one of the pattern's); 1: fast path, window [ws, ws + span); 2: byte-wise path.
struct VerifyWin {
    uint64_t j, ws;
    uint32_t span, mode;
};

template <typename WT, int K, int NCH, bool LEAN>
__device__ __forceinline__ VerifyWin verify_locate(const VerifyCtx<WT, K> &c, uint64_t ent)
{
    VerifyWin w;
    w.ws = 0;
    w.span = 0;
    w.mode = 0;
    // lean entries: 64-bit dword index (halfword index for H == 2 samples: jshift 1)
    const uint64_t j = (LEAN ? ent : (ent & AGH_CAND_IDX_MASK)) << c.jshift;
    w.j = j;
    if (j >= c.n) return w;

    // Lean scans with a gram table: the sample's q-gram says where in the pattern it sits.
    // A gram that is not the pattern's (hash false positive) is dropped here; otherwise an
    // occurrence containing it starts in [j-o-k, j-o+k] and ends before j-o+m+k, so a fresh
    // automaton over [j - o_last - k - 1, j - o_first + m + k) finds it -- about half the bytes
    // of the offset-blind window.  (The extra leading byte gives level e its e leading
    // deletions, exactly as the re-fed delimiter does at a record start.)
    uint32_t lw = c.Lw, span = c.span;
    bool fast = true;
    if (c.gtab) {
        typedef uint32_t u32_a2 __attribute__((aligned(2)));
        const uint32_t dw = *reinterpret_cast<const u32_a2 *>(c.text + j);     // j is 4-aligned (2 with H == 2)
        const uint32_t g = (dw & c.q->qmask) | c.q->fold;
        const uint64_t e = c.gtab[c.q->fq == 4 ? agh_sample_hash_q4(g) : agh_sample_hash_q3(g)];
        if (e & AGH_GT_AMBIGUOUS) {
            fast = false;                       // full window, byte-wise
        } else {
            if ((uint32_t)e != g) return w;
            lw = (uint32_t)((e >> 40) & 0xffu) + (uint32_t)c.q->k + 1u;   // + one warm-up byte
            span = c.tspan;
            // (numbered scans: the window may start behind the start of the sample's chunk, whose record number
            // the candidate carries -- verify_walk counts the delimiters in between only if something matched)
        }
    }
    fast = fast && j >= lw && (j - lw) + span < c.n && (j - lw) + 16u * NCH <= c.n16;
    w.mode = fast ? 1u : 2u;
    w.ws = fast ? j - lw : 0;
    w.span = span;
    return w;
}

// Candidates in neighbouring lanes are neighbours in the text, and the sampled grams of ONE
// occurrence (3-4 of them for m = 16, H = 4) all point at the same window when no indel sits
// between them: the lane whose fast-path window equals its predecessor's has nothing to add.
// Must be called by all 64 lanes (lanes without a candidate pass mode 0).
__device__ __forceinline__ bool verify_same_window_as_prev_lane(const VerifyWin &w)
{
    const uint32_t lo = (uint32_t)w.ws, hi = (uint32_t)(w.ws >> 32) | (w.mode << 8);
    const uint32_t plo = (uint32_t)__shfl_up((int)lo, 1), phi = (uint32_t)__shfl_up((int)hi, 1);
    return lane_id() > 0 && w.mode == 1u && plo == lo && phi == hi;
}

// Fast path geometry, identical for every lane: the window starts Lw = max(m+k+1, 16) bytes in
// front of the sample and spans Lw + q + m + k bytes; it is fetched with NCH unaligned 16-byte
// loads issued together and walked branch-free out of registers.  Match positions and
// delimiter positions are collected as bit masks; record numbers are derived from them after
// the walk.  Windows that touch the head or the tail of the text take the byte-wise path.
template <typename WT, int K, int NCH, bool LEAN, bool MB, bool GEN = false>
__device__ __forceinline__ void verify_walk(const VerifyCtx<WT, K> &c, uint64_t ent,
                                            uint32_t wave_base, const VerifyWin &win)
{
    constexpr int NMW = (NCH * 16 + 63) / 64;           // 64-bit words per position mask
    if (win.mode == 0u) return;
    const uint64_t j = win.j;
    const uint32_t rc_anchor = LEAN ? 0u : wave_base + (uint32_t)(ent >> AGH_CAND_IDX_BITS);  // record no. at anchor
    const uint64_t anchor = j & ~(uint64_t)15;                           // sample's chunk start
    const uint32_t span = win.span;
    if (win.mode == 2u) {
        const uint64_t ws = j > c.Lw ? j - c.Lw : 0;
        uint64_t we = j + c.tailw;
        if (we > c.n) we = c.n;
        verify_window_slow<WT, K, LEAN, GEN>(c.text, c.n, *c.q, c.lmask, c.dbm, ws, we, anchor, rc_anchor,
                                        *c.mk);
        return;
    }
    const uint64_t ws = win.ws;
    u32x4_u ch[NCH];
#pragma unroll
    for (int ic = 0; ic < NCH; ++ic)
        ch[ic] = *reinterpret_cast<const u32x4_u *>(c.text + ws + 16 * ic);

    Automaton<WT, K> A;
    A.reset();
    uint32_t seen = 0;
    uint64_t hitm[NMW], hit2m[NMW], dm[NMW];
#pragma unroll
    for (int i = 0; i < NMW; ++i) hitm[i] = hit2m[i] = dm[i] = 0;
    if (MB) {
        // delimiter ends inside the window, straight from the bitmap (bits >= span cleared)
#pragma unroll
        for (int i = 0; i < NMW; ++i) {
            uint64_t w = dbm_bits64(c.dbm, ws + 64u * (uint32_t)i);
            const int lo = i * 64;
            if ((int)span <= lo) w = 0;
            else if ((int)span < lo + 64) w &= (1ull << (span - lo)) - 1ull;
            dm[i] = w;
        }
    }
#pragma unroll
    for (int p = 0; p < NCH * 16; ++p) {
        if ((uint32_t)p >= span) break;               // uniform but for dropped lanes
        const uint32_t dwv = ch[p >> 4][(p >> 2) & 3];
        const uint32_t byte = (dwv >> (8 * (p & 3))) & 0xffu;
        const uint32_t hit = A.template step_q<GEN>(c.lmask[byte], c.finalbit, *c.q) ? 1u : 0u;
        const uint32_t isd = MB ? (uint32_t)(dm[p >> 6] >> (p & 63)) & 1u
                                : ((byte == c.q->delim) ? 1u : 0u);
        hitm[p >> 6] |= (uint64_t)(hit & ~seen) << (p & 63);
        if (!MB) dm[p >> 6] |= (uint64_t)isd << (p & 63);
        seen |= hit;
        if (isd) {                                  // select, no branch: see RF above
#pragma unroll
            for (int e = 0; e <= K; ++e) A.R[e] = c.RF.R[e];
            seen = c.rf_hit ? 1u : 0u;
        }
    }
    if (c.rf_hit) {
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
            if (before_ws == ~1ull)
                before_ws = MB ? lean_record_start_mb(c.dbm, ws, *c.mk)
                               : lean_record_start(c.text, ws, c.q->delim, *c.mk);
            return before_ws;
        };
#pragma unroll
        for (int i = 0; i < NMW; ++i) {
            uint64_t hm = hitm[i];
            while (hm) {
                const uint32_t p = (uint32_t)(i * 64 + __ffsll((long long)hm) - 1);
                hm &= hm - 1;
                const uint64_t st = start_of(p);
                if (st != ~0ull) lean_insert(*c.mk, st);
            }
            uint64_t h2 = hit2m[i];
            while (h2) {
                const uint32_t p = (uint32_t)(i * 64 + __ffsll((long long)h2) - 1);
                h2 &= h2 - 1;
                lean_insert(*c.mk, ws + p + 1);                // the record right after delimiter p
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
        // record number at the window start: the candidate carries the one at the start of its sample's chunk
        // (anchor); the window starts in front of it, or -- tight windows -- up to 15 bytes behind it
        uint32_t r0;
        if (ws <= anchor) {
            r0 = rc_anchor - delims_before((uint32_t)(anchor - ws));
        } else if (MB) {
            r0 = rc_anchor + dbm_count(c.dbm, anchor, ws);
        } else {
            const uint4 av = *reinterpret_cast<const uint4 *>(c.text + anchor);
            r0 = rc_anchor + delims_in(mask_tail(av, (int)(ws - anchor), (~c.q->delim & 0xffu) * 0x01010101u),
                                       c.q->delim * 0x01010101u);
        }
#pragma unroll
        for (int i = 0; i < NMW; ++i) {
            uint64_t hm = hitm[i];
            while (hm) {
                const uint32_t p = (uint32_t)(i * 64 + __ffsll((long long)hm) - 1);
                hm &= hm - 1;
                mark_record(*c.mk, r0 + delims_before(p), ws + p);
            }
            uint64_t h2 = hit2m[i];
            while (h2) {
                const uint32_t p = (uint32_t)(i * 64 + __ffsll((long long)h2) - 1);
                h2 &= h2 - 1;
                mark_record(*c.mk, r0 + delims_before(p + 1u), ws + p + 1u);
            }
        }
    }
}

template <typename WT, int K, int NCH, bool LEAN, bool MB, bool GEN = false>
__device__ __forceinline__ void verify_candidate(const VerifyCtx<WT, K> &c, uint64_t ent,
                                                 uint32_t wave_base)
{
    const VerifyWin w = verify_locate<WT, K, NCH, LEAN>(c, ent);
    verify_walk<WT, K, NCH, LEAN, MB, GEN>(c, ent, wave_base, w);
}
