// agh_multi.hip -- multi-pattern scans (-f patternfile; the role of newmgrep.c: prepf() tables +
// monkey1() hash-and-compare, newmgrep.c:192-375, 839-1012), exact and -- beyond the reference,
// which ignores -# together with -f (compat.c:34-37) -- with k errors (BASELINE config 5).
//
// A record matches iff it contains any pattern verbatim (k = 0) or a substring within edit distance
// k of any pattern.  Same shape as the single-pattern scan with the filter turned around: instead of
// a few aligned text samples against all q-grams of one pattern, every S-th text position's q-byte
// gram (q = min(4, shortest entry); S = 1, 2 or 4 by the shortest entry, fill_multi_tables) is
// probed against the grams of the table ENTRIES -- whole patterns, or the k+1 disjoint pieces of
// every pattern (an occurrence with <= k errors contains one of them verbatim) -- in a 2^18-bit
// table held in LDS.
//
//   k_sweep_multi     streams the text like k_sweep.  Per probe: v_alignbyte (unaligned gram),
//                     v_dot2_u32_u16 (hash, agh_sample_prod_q4), the aligned dword of the bit table,
//                     v_lshrrev by the low five bits of the product, v_alignbit to push the bit
//                     into the hit word -- five VALU operations.  First-level hits take a second,
//                     independent Bloom probe (q = 4); what survives is queued in LDS, every lane
//                     writing its own hits (one per round), and goes to the wave's private slice.
//   k_verify_multi    one lane per candidate, one workgroup per slice: the bucket of entries with
//                     that gram is walked, an entry that occurs verbatim is a match (k = 0) or sends
//                     its pattern's k-error automaton over the window the occurrence can cover;
//                     record bookkeeping only after a hit (delimiter / hit masks as in verify_walk).
//   k_dense_multi     sets whose hits do not fit the slices (hundreds of 1..3-byte entries; -f with
//                     errors over 4-byte patterns): the same probes, but the queue is verified on
//                     the spot, 64 candidates at a time, one per lane -- nothing can overflow.
#include <string.h>

#include "agh_verify_inl.h"
#include "agh_sweep_inl.h"

typedef agh_multi_dev agh_multi_tables;

#define AGH_MP_WORDS ((1u << AGH_MP_BITS) / 32u)
#define AGH_MP_CQ_LEN 128u          // a round adds at most 64 entries to fewer than 64 queued ones

// ---------------------------------------------------------------------------------------
// probes
// ---------------------------------------------------------------------------------------
// bit number of a gram in the table; g carries the case fold already (OR 0x20 into every byte)
template <int MODE>   // bit 1: q == 4
__device__ __forceinline__ uint32_t mp_index(uint32_t g, const agh_dev_query &q)
{
    if (MODE & 2) return agh_sample_prod_q4(g);              // the low 18 bits count
    return agh_sample_hash18_q3(g & q.qmask);
}

__device__ __forceinline__ uint32_t mp_bit(const uint8_t *tab8, uint32_t idx)
{
    const uint32_t val = *reinterpret_cast<const uint32_t *>(tab8 + ((idx >> 3) & ((1u << (AGH_MP_BITS - 3)) - 4u)));
    return val >> (idx & 31u);                               // bit 0 = the table bit
}

// The probed positions of one 16-byte chunk (w[4] = the dword that follows it): every STRIDE-th
// byte; 16 / STRIDE result bits are pushed into acc from the top, first probe first.
template <int MODE, int STRIDE, bool Q5>
__device__ __forceinline__ void probe_chunk_l1(const uint32_t (&w)[5], const agh_dev_query &q,
                                               const uint8_t *tab8, uint32_t &acc)
{
#pragma unroll
    for (int p = 0; p < 16; p += STRIDE) {
        const int d = p >> 2, sh = p & 3;
        uint32_t g = sh ? __builtin_amdgcn_alignbyte(w[d + 1], w[d], sh) : w[d];
        if (Q5) g = agh_mix5(g, w[d + 1]);                   // stride 4: the fifth byte opens the next dword
        acc = __builtin_amdgcn_alignbit(mp_bit(tab8, mp_index<MODE>(g, q)), acc, 1);
    }
}

// the gram at byte p (0..15, a run-time value) of a chunk.  The five dwords come BY VALUE: selects
// over an array passed by reference are folded into one indexed load, which sends the caller's
// arrays to scratch memory (128 bytes per lane stored per supertile: the sweep lost a quarter).
template <bool Q5>
__device__ __forceinline__ uint32_t gram_at(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3,
                                            uint32_t w4, uint32_t p)
{
    const uint32_t d = p >> 2;
    const uint32_t lo = d == 0 ? w0 : (d == 1 ? w1 : (d == 2 ? w2 : w3));
    const uint32_t hi = d == 0 ? w1 : (d == 1 ? w2 : (d == 2 ? w3 : w4));
    uint32_t g = __builtin_amdgcn_alignbyte(hi, lo, p & 3u);
    if (Q5) g = agh_mix5(g, hi);
    return g;
}

// q == 4: the table is a two-probe Bloom filter.  First-level hits (with ~2000 entries 0.8 % of all
// positions by chance plus the real prefix occurrences) take the second probe; what survives is
// almost only real occurrences of an entry's gram.  h: compact hit bits (bit i = byte i * STRIDE).
template <int STRIDE, bool Q5>
__device__ __forceinline__ uint32_t probe_chunk_l2(uint32_t h, uint32_t w0, uint32_t w1, uint32_t w2,
                                                   uint32_t w3, uint32_t w4, const uint8_t *tab8)
{
    uint32_t keep = 0;
    while (h) {
        const uint32_t i = (uint32_t)__ffs((int)h) - 1u;
        h &= h - 1u;
        const uint32_t g = gram_at<Q5>(w0, w1, w2, w3, w4, i * STRIDE);
        keep |= (mp_bit(tab8, agh_sample_hash18b_q4(g)) & 1u) << i;
    }
    return keep;
}

__device__ __forceinline__ bool dev_isalnum(uint32_t c)      // isalnum() of the C locale
{
    return (c - '0' < 10u) || ((c | 0x20u) - 'a' < 26u);
}

// ASCII upper -> lower in four bytes at once (newmgrep.c: tr[] folds case under -i).
__device__ __forceinline__ uint32_t swar_lower(uint32_t t)
{
    const uint32_t x = t & 0x7f7f7f7fu;
    const uint32_t ge = x + 0x3f3f3f3fu;            // bit 7 of a byte <=> byte >= 'A'
    const uint32_t gt = x + 0x25252525u;            // bit 7 <=> byte > 'Z'
    return t | (((ge & ~gt & ~t) & 0x80808080u) >> 2);
}

// Queue the hits of a supertile: every lane writes its own hits, one per round, at the rank of its
// lane among the lanes that still have one.  lo/hi: compact hit bits, strip u at bits [u*NB, (u+1)*NB).
// Entry = (delimiters in front of the lane's chunk << 32) | byte offset.
template <int STRIDE, typename OnFull>
__device__ __forceinline__ void emit_rounds(uint32_t lo, uint32_t hi, uint64_t s, const uint32_t (&rc)[4],
                                            uint64_t *cq, uint32_t &qn, OnFull on_full)
{
    constexpr uint32_t NB = 16u / STRIDE, NBS = NB == 16 ? 4u : (NB == 8 ? 3u : 2u);
    const uint32_t lane = (uint32_t)lane_id();
    uint64_t hm;
    while ((hm = __ballot((lo | hi) != 0u)) != 0ull) {
        const bool has = (lo | hi) != 0u;
        uint32_t i;
        if (lo) { i = (uint32_t)__ffs((int)lo) - 1u; lo &= lo - 1u; }
        else { i = 32u + (uint32_t)__ffs((int)hi) - 1u; hi &= hi - 1u; }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
        if (has) {
            const uint32_t u = i >> NBS, pi = i & (NB - 1u);
            const uint32_t off = (uint32_t)((s + u) * AGH_STRIP) + lane * 16u + pi * STRIDE;
            const uint32_t r = u == 0 ? rc[0] : (u == 1 ? rc[1] : (u == 2 ? rc[2] : rc[3]));
            cq[qn + rank] = ((uint64_t)r << 32) | off;
        }
        qn += (uint32_t)__popcll(hm);
        if (qn >= 64u) on_full();
    }
}

// ---------------------------------------------------------------------------------------
// verification of one candidate position
// ---------------------------------------------------------------------------------------
// Does the table entry (len bytes at pool) occur verbatim at text position s?  (Used when the
// probed gram is not the entry's prefix; the prefix case compares against the window at j.)
__device__ __forceinline__ bool multi_entry_at(const uint8_t *__restrict__ text, uint64_t n, bool fold,
                                               const uint8_t *__restrict__ pool, uint32_t len, uint64_t s)
{
    if (s + len > n) return false;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    uint32_t t = 0;
    if (s + 16 <= n16) {
        const u32x4_u v = *reinterpret_cast<const u32x4_u *>(text + s);
        const u32x4_u pv = *reinterpret_cast<const u32x4_u *>(pool);
        const uint32_t head = len < 16u ? len : 16u;
        uint32_t diff = 0;
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
            const uint32_t nb = head > 4u * d ? head - 4u * d : 0u;
            const uint32_t m = nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u);
            diff |= ((fold ? swar_lower(v[d]) : v[d]) ^ pv[d]) & m;
        }
        if (diff) return false;
        t = 16;
    }
    for (; t < len; ++t) {
        uint32_t c = text[s + t];
        if (fold && c >= 'A' && c <= 'Z') c += 32u;
        if (c != pool[t]) return false;
    }
    return true;
}

// 16 text bytes at j (lower-cased when the query folds) + the probed gram
struct mp_window {
    uint32_t tw[4];
    uint32_t g;
};

__device__ __forceinline__ mp_window mp_fetch(const uint8_t *__restrict__ text, uint64_t n,
                                              const agh_dev_query &q, uint64_t j)
{
    mp_window w;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    if (j + 16 <= n16) {
        const u32x4_u v = *reinterpret_cast<const u32x4_u *>(text + j);
        w.tw[0] = v[0]; w.tw[1] = v[1]; w.tw[2] = v[2]; w.tw[3] = v[3];
    } else {                                        // the last bytes of the text
        w.tw[0] = w.tw[1] = w.tw[2] = w.tw[3] = 0;
        for (uint32_t t = 0; t < 16 && j + t < n; ++t) w.tw[t >> 2] |= (uint32_t)text[j + t] << (8 * (t & 3));
    }
    w.g = (w.tw[0] & q.qmask) | q.fold;             // the probed q-gram at j
    if (q.fold) {
#pragma unroll
        for (int d = 0; d < 4; ++d) w.tw[d] = swar_lower(w.tw[d]);
    }
    return w;
}

// Does bucket item `it` occur at the candidate position?  -> start of the entry's occurrence, or ~0
__device__ __forceinline__ uint64_t mp_item_occurs(const uint8_t *__restrict__ text, uint64_t n,
                                                   const agh_dev_query &q, const agh_multi_tables &mt,
                                                   const agh_mp_item &item, const mp_window &w, uint64_t j)
{
    const uint32_t o = item.info >> 8, len = item.info & 0xffu;
    const uint32_t go = item.piece >> 28;               // the gram sits at this offset of the entry
    if (go) {                                           // strided probing: the entry starts in front of j
        if (j < go) return ~0ull;
        return multi_entry_at(text, n, q.fold != 0, mt.pool + o, len, j - go) ? j - go : ~0ull;
    }
    if (j + len > n) return ~0ull;
    const u32x4_u pv = *reinterpret_cast<const u32x4_u *>(mt.pool + o);
    const uint32_t head = len < 16u ? len : 16u;
    uint32_t diff = 0;
#pragma unroll
    for (uint32_t d = 0; d < 4; ++d) {
        const uint32_t nb = head > 4u * d ? head - 4u * d : 0u;      // bytes of this dword in play
        const uint32_t m = nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u);
        diff |= (w.tw[d] ^ pv[d]) & m;
    }
    if (diff) return ~0ull;
    for (uint32_t t = 16; t < len; ++t) {               // entries longer than 16 bytes: the rest
        uint32_t c = text[j + t];
        if (q.fold && c >= 'A' && c <= 'Z') c += 32u;
        if (c != mt.pool[o + t]) return ~0ull;
    }
    return j;
}

// A verified exact occurrence at j: count its record once.
template <bool LEAN>
__device__ __forceinline__ void multi_mark(const uint8_t *__restrict__ text, const agh_dev_query &q,
                                           const agh_marks &mk, uint64_t j, uint32_t rc_chunk,
                                           const uint64_t *__restrict__ dbm)
{
    if (LEAN) {
        const uint64_t st = q.mb ? lean_record_start_mb(dbm, j, mk) : lean_record_start(text, j, q.delim, mk);
        if (st != ~0ull) lean_insert(mk, st);
    } else {
        // record number = delimiters in front of the chunk + delimiters in [chunk, j)
        uint32_t rec = rc_chunk;
        if (q.mb) rec += dbm_count(dbm, j & ~(uint64_t)15, j);
        else for (uint64_t i = j & ~(uint64_t)15; i < j; ++i) rec += (text[i] == q.delim);
        mark_record(mk, rec, j);
    }
}

// -f with errors: the k-error automaton of one pattern (position masks pmask, length m <= 32) over
// the window [ws, we) -- at most 64 bytes: m + 2k <= 48, plus up to 15 bytes of lead when the window
// has to start at the candidate's 16-byte chunk (numbered scans count delimiters from there).
// Delimiters and first hits per record are collected as bit masks during the walk; records are
// resolved afterwards, and only if something matched: the look-back for the record start (lean) or
// the delimiter count (numbered) costs nothing for the candidates that do not match.
template <bool LEAN, int K>
__device__ __forceinline__ void approx_window_k(const uint8_t *__restrict__ text, uint64_t n,
                                                const agh_dev_query &q,
                                                const uint32_t *__restrict__ pmask, uint32_t m,
                                                uint64_t ws, uint64_t we, uint64_t anchor,
                                                uint32_t rc_anchor, const agh_marks &mk,
                                                const uint64_t *__restrict__ dbm)
{
    const uint32_t finalbit = 1u << (m - 1);
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    const bool mb = q.mb != 0;                  // delimiter ends from the bitmap (several bytes / folded)
    auto before_window = [&]() -> uint64_t {
        return mb ? lean_record_start_mb(dbm, ws, mk) : lean_record_start(text, ws, q.delim, mk);
    };
    Automaton<uint32_t, K> A;
    A.reset();
    uint32_t seen = 0;
    uint64_t hitm = 0, dm = 0;
    if (ws == 0) seen = A.step(pmask[q.head_byte], finalbit) ? 1u : 0u;   // (never: patterns hold no delimiter byte)
    for (uint32_t b0 = 0; ws + b0 < we; b0 += 16) {
        const uint64_t i0 = ws + b0;
        const uint32_t nb = we - i0 < 16 ? (uint32_t)(we - i0) : 16u;
        uint32_t dws[4] = {0u, 0u, 0u, 0u};
        if (i0 + 16 <= n16) {
            const u32x4_u v = *reinterpret_cast<const u32x4_u *>(text + i0);
            dws[0] = v[0]; dws[1] = v[1]; dws[2] = v[2]; dws[3] = v[3];
        } else {
            for (uint32_t t = 0; t < nb; ++t) dws[t >> 2] |= (uint32_t)text[i0 + t] << (8 * (t & 3));
        }
        uint32_t cms[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) cms[t] = (uint32_t)t < nb ? pmask[(dws[t >> 2] >> (8 * (t & 3))) & 0xffu] : 0u;
        const uint32_t d16 = mb ? (uint32_t)dbm_bits64(dbm, i0) & 0xffffu : 0u;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if ((uint32_t)t < nb) {
                const uint32_t c = (dws[t >> 2] >> (8 * (t & 3))) & 0xffu;
                const uint32_t hit = A.step(cms[t], finalbit) ? 1u : 0u;
                hitm |= (uint64_t)(hit & ~seen) << (b0 + (uint32_t)t);
                seen |= hit;
                if (mb ? ((d16 >> t) & 1u) != 0u : c == q.delim) {   // patterns never hold the delimiter byte: cm == 0 and
                    dm |= 1ull << (b0 + (uint32_t)t);   // the re-fed step leaves level e with its e deletions
                    A.reset();
                    A.step(cms[t], finalbit);
                    seen = 0;
                }
            }
        }
    }
    if (we == n && q.tail_virtual) {            // asearch.c:87-91: the appended delimiter is the 65th position at most
        bool tail_hit = false;
        for (uint32_t jd = 0; jd < q.dlen && !tail_hit; ++jd) tail_hit = A.step(pmask[q.dbytes[jd]], finalbit);
        if (tail_hit && !seen) {
            // the record that is open at the end of the text
            const uint32_t span = (uint32_t)(we - ws);
            const uint64_t below = span >= 64 ? dm : (dm & ((1ull << span) - 1ull));
            if (LEAN) {
                const uint64_t st = below ? ws + 64 - (uint64_t)__clzll((long long)below) : before_window();
                if (st != ~0ull) lean_insert(mk, st);
            } else {
                const uint32_t a = (uint32_t)(anchor - ws);
                const uint32_t back = (uint32_t)__popcll(a >= 64 ? dm : (dm & ((1ull << a) - 1ull)));
                mark_record(mk, rc_anchor - back + (uint32_t)__popcll(below), n);
            }
        }
    }
    if (!hitm) return;
    if (LEAN) {
        uint64_t before_ws = ~1ull;             // record start in front of the window: looked up at most once
        while (hitm) {
            const uint32_t p = (uint32_t)__ffsll((long long)hitm) - 1u;
            hitm &= hitm - 1ull;
            const uint64_t below = dm & ((1ull << p) - 1ull);
            uint64_t st;
            if (below) st = ws + 64 - (uint64_t)__clzll((long long)below);
            else {
                if (before_ws == ~1ull) before_ws = before_window();
                st = before_ws;
            }
            if (st != ~0ull) lean_insert(mk, st);
        }
    } else {
        const uint32_t a = (uint32_t)(anchor - ws);     // ws <= anchor: delimiters in [ws, anchor)
        const uint32_t r0 = rc_anchor - (uint32_t)__popcll(a >= 64 ? dm : (dm & ((1ull << a) - 1ull)));
        while (hitm) {
            const uint32_t p = (uint32_t)__ffsll((long long)hitm) - 1u;
            hitm &= hitm - 1ull;
            mark_record(mk, r0 + (uint32_t)__popcll(dm & ((1ull << p) - 1ull)), ws + p);
        }
    }
}

// -f with ONE error, patterns of two pieces: with one piece verbatim the other side has to lie
// within one edit of the text next to it -- a question about <= 8 bytes that two 64-bit words answer
// (no automaton, no per-byte mask gathers: the verifier waits on its dependent loads).
// S = the text bytes next to the piece, nearest first; B = the pattern bytes of the other side in the
// same order; L <= 7 of them.  Up to the first mismatch i both agree; one edit there and the rest has
// to agree again: the pattern byte is missing in the text (B[i+1..] == S[i..]), replaced
// (B[i+1..] == S[i+1..]) or a text byte stands in front of it (B[i..] == S[i+1..]).  A delimiter can
// only be the replaced or the extra text byte (patterns hold none): the automaton resets there, so
// that is no match.
__device__ __forceinline__ bool side_within_one_edit(uint64_t S, uint64_t B, uint32_t L, uint32_t delim)
{
    const uint64_t maskL = (1ull << (8u * L)) - 1ull;
    const uint64_t x = (S ^ B) & maskL;
    if (!x) return true;
    const uint32_t i8 = (uint32_t)__builtin_ctzll(x) & ~7u;         // 8 * (first mismatching byte)
    const uint64_t tail = maskL >> i8;                              // bytes i .. L-1, moved down
    if (!(((S ^ (B >> 8)) >> i8) & (tail >> 8))) return true;      // the pattern byte is missing
    if (((uint32_t)(S >> i8) & 0xffu) == delim) return false;
    if (!((x >> i8) >> 8)) return true;                             // replaced
    return !((((S >> 8) ^ B) >> i8) & tail);                        // an extra text byte
}

typedef uint64_t u64_u __attribute__((aligned(1)));

// Everything that can match at candidate position j: the bucket of entries with the gram at j.
// K = 0: an entry that occurs is a match; K > 0: a verbatim PIECE at text position js sends its
// pattern's automaton over [js - po - K, js + (m - po) + K).
template <bool LEAN, int K>
__device__ __forceinline__ void mp_verify_at(const uint8_t *__restrict__ text, uint64_t n,
                                             const agh_dev_query &q, const agh_multi_tables &mt,
                                             uint64_t j, uint32_t rc_chunk, const agh_marks &mk)
{
    const mp_window w = mp_fetch(text, n, q, j);
    const uint32_t b = agh_mp_bucket(w.g);
    typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
    const u32x2_a4 be = *reinterpret_cast<const u32x2_a4 *>(mt.bucket_start + b);     // [b], [b + 1]
    for (uint32_t it = be.x; it < be.y; ++it) {
        const uint4 raw = reinterpret_cast<const uint4 *>(mt.items)[it];
        agh_mp_item item;
        item.info = raw.x; item.piece = raw.y; item.owner = raw.z; item.pom = raw.w;
        const uint64_t js = mp_item_occurs(text, n, q, mt, item, w, j);
        if (js == ~0ull) continue;
        if (K == 0 && q.guard) {
            // -w / -x with -f (newmgrep.c:869-872, :835-840): the bytes next to the occurrence; the
            // virtual byte in front of the text and the delimiter appended behind it count
            const uint32_t len = item.info & 0xffu;
            const uint32_t before = js ? text[js - 1] : q.head_byte;
            const uint32_t after = js + len < n ? text[js + len] : q.dbytes[0];
            const bool ok = q.guard == 2u ? (before == '\n' && after == '\n')
                                          : !(dev_isalnum(before) || dev_isalnum(after));
            if (!ok) continue;
        }
        if (K == 0) {
            multi_mark<LEAN>(text, q, mk, j, rc_chunk, mt.dbm);
            return;                             // one verbatim entry is enough for the record
        }
        const uint32_t po = item.pom >> 8, m = item.pom & 0xffu;
        if constexpr (K == 1) {
            // two pieces: [0, len) and [po, m); the side that is not the piece has L bytes
            const uint32_t len = item.info & 0xffu, L = po ? po : m - len;
            if (!q.mb && L <= 7u && (po ? js >= 8u : js + len + 8u <= n)) {
                const uint8_t *pat = mt.pool + (item.info >> 8) - po;       // the pieces of a pattern lie in a row
                uint64_t S, B;
                if (po == 0) {                  // the rest of the pattern behind the piece
                    S = *reinterpret_cast<const u64_u *>(text + js + len);
                    B = *reinterpret_cast<const u64_u *>(pat + len);
                } else {                        // the head of the pattern in front of it: nearest byte first
                    S = __builtin_bswap64(*reinterpret_cast<const u64_u *>(text + js - 8));
                    B = __builtin_bswap64(*reinterpret_cast<const u64_u *>(pat) << (8u * (8u - L)));
                }
                if (q.fold) S = (uint64_t)swar_lower((uint32_t)S) | ((uint64_t)swar_lower((uint32_t)(S >> 32)) << 32);
                if (side_within_one_edit(S, B, L, q.delim)) {
                    multi_mark<LEAN>(text, q, mk, j, rc_chunk, mt.dbm);
                    return;                     // the record of j is counted: nothing else to find here
                }
                continue;
            }
        }
        const uint64_t anchor = j & ~(uint64_t)15;          // rc_chunk = delimiters in front of it
        const uint64_t back = (uint64_t)po + K;
        uint64_t ws = js > back ? js - back : 0;
        if (!LEAN && ws > anchor) ws = anchor;
        uint64_t we = js + (m - po) + K;
        if (we > n) we = n;
        approx_window_k<LEAN, K>(text, n, q, mt.owner_mask + (size_t)item.owner * 256u, m, ws, we, anchor,
                                 rc_chunk, mk, mt.dbm);
    }
}

// ---------------------------------------------------------------------------------------
// the sweep
// ---------------------------------------------------------------------------------------
// One wave per 256 KiB range, 4 KiB supertiles, next supertile prefetched -- as k_sweep.
// MODE: bit 0 fold case, bit 1 q == 4, bit 2 lean (no census).
// Hits go to the wave's private slice of the candidate buffer; k_verify_multi reads them back.
// Measured and dropped (round 3; profiles/r03_perf_multi_verifier_waves*.log, r03_pmc_sweep_multi*_fused.json):
// verifying a full queue inside the sweeping wave (1.78 vs 1.44 ms per 4 GiB, exact 4..12 B; k = 1: 2.60
// vs 1.87), two verifying waves per workgroup fed through an LDS ring as in k_sweep_fused (1.55 vs 1.45;
// k = 1: 1.65 vs 1.65; 8..12 B exact: 1.09 vs 0.93) and the verifier of one part of the text on a second
// stream under the sweep of the next (1.52 / 1.84 / 2.84 ms in 2 / 4 / 8 parts) -- the sweep is bound by
// VALU issue and LDS reads, and whatever shares its SIMDs costs it what it would have cost alone.
template <int MODE, int STRIDE, bool Q5>
__global__ __launch_bounds__(256) void k_sweep_multi(const uint4 *__restrict__ text, uint64_t n,
                                                     uint64_t n_full_strips, agh_dev_query q,
                                                     const uint32_t *__restrict__ bits_g,
                                                     uint32_t *__restrict__ wave_totals,
                                                     uint64_t *__restrict__ cand,
                                                     uint32_t *__restrict__ wave_cand,
                                                     uint32_t *__restrict__ counters,
                                                     const uint16_t *__restrict__ dbm16, uint32_t w_base)
{
    // this launch sweeps the wave ranges w_base .. up to strip n_full_strips (a part of the text)
    __shared__ __attribute__((aligned(16))) uint32_t tab[AGH_MP_WORDS];
    __shared__ uint64_t cq_all[4 * AGH_MP_CQ_LEN];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(bits_g);
        uint4 *dst = reinterpret_cast<uint4 *>(tab);
        constexpr int PER = AGH_MP_WORDS / 4 / 256;
        uint4 tmp[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) tmp[i] = src[threadIdx.x + i * 256];
#pragma unroll
        for (int i = 0; i < PER; ++i) dst[threadIdx.x + i * 256] = tmp[i];
        __syncthreads();
    }
    const uint8_t *tab8 = reinterpret_cast<const uint8_t *>(tab);
    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const uint64_t w = (uint64_t)w_base + (uint64_t)blockIdx.x * 4 + wib;
    const uint64_t s0 = w * AGH_WAVE_STRIPS;
    if (s0 >= n_full_strips) return;
    uint64_t s1 = s0 + AGH_WAVE_STRIPS;
    if (s1 > n_full_strips) s1 = n_full_strips;
    const uint32_t dd = q.delim * 0x01010101u;
    const uint32_t fold4 = (MODE & 1) ? 0x20202020u : 0u;
    const uint64_t n_dw = ((n + 15) & ~(uint64_t)15) / 4;      // readable dwords
    uint32_t run = 0, ncand = 0, qn = 0;
    uint64_t *cq = cq_all + wib * AGH_MP_CQ_LEN;
    uint64_t *slice = cand + w * AGH_MP_SLICE_CAP;
    auto flush64 = [&]() { flush_candidates<AGH_MP_SLICE_CAP>(cq, qn, 64u, slice, ncand, counters); };
    // the dword right behind strip st-1 (uniform; 0 past the readable text)
    auto first_dword_of = [&](uint64_t st) -> uint32_t {
        const uint64_t i = st * 256u;
        return i < n_dw ? reinterpret_cast<const uint32_t *>(text)[i] : 0u;
    };
    constexpr uint32_t NB = 16u / STRIDE;
    // one supertile: strips s .. s+3 (v0..v3), nx3 = the first dword of strip s+4
    auto supertile = [&](uint4 v0, uint4 v1, uint4 v2, uint4 v3, uint64_t s, uint32_t nx3) {
        uint32_t w0[5] = {v0.x | fold4, v0.y | fold4, v0.z | fold4, v0.w | fold4,
                          next_lane_dword(v0.x, (uint32_t)__builtin_amdgcn_readlane((int)v1.x, 0)) | fold4};
        uint32_t w1[5] = {v1.x | fold4, v1.y | fold4, v1.z | fold4, v1.w | fold4,
                          next_lane_dword(v1.x, (uint32_t)__builtin_amdgcn_readlane((int)v2.x, 0)) | fold4};
        uint32_t w2[5] = {v2.x | fold4, v2.y | fold4, v2.z | fold4, v2.w | fold4,
                          next_lane_dword(v2.x, (uint32_t)__builtin_amdgcn_readlane((int)v3.x, 0)) | fold4};
        uint32_t w3[5] = {v3.x | fold4, v3.y | fold4, v3.z | fold4, v3.w | fold4,
                          next_lane_dword(v3.x, nx3) | fold4};
        // census (numbered scans): delimiters in front of the lane's chunk of every strip.  In front
        // of the probes: behind them its bitmap branch splits the block, and the compiler parks all 64
        // table reads of a supertile in registers across it (181 VGPRs, two waves per SIMD)
        uint32_t rc[4] = {0u, 0u, 0u, 0u};
        if (!(MODE & 4)) {
            // ("128 minus the delimiters of the lane's chunk"; bitmap delimiters: 16 bits per chunk)
            uint32_t a0, a1, a2, a3;
            if (q.mb) {
                const uint16_t *d = dbm16 + s * 64 + lane;
                a0 = 128u - (uint32_t)__popc((uint32_t)d[0]);
                a1 = 128u - (uint32_t)__popc((uint32_t)d[64]);
                a2 = 128u - (uint32_t)__popc((uint32_t)d[128]);
                a3 = 128u - (uint32_t)__popc((uint32_t)d[192]);
            } else {
                a0 = nz_popc(v0.x, dd) + nz_popc(v0.y, dd) + nz_popc(v0.z, dd) + nz_popc(v0.w, dd);
                a1 = nz_popc(v1.x, dd) + nz_popc(v1.y, dd) + nz_popc(v1.z, dd) + nz_popc(v1.w, dd);
                a2 = nz_popc(v2.x, dd) + nz_popc(v2.y, dd) + nz_popc(v2.z, dd) + nz_popc(v2.w, dd);
                a3 = nz_popc(v3.x, dd) + nz_popc(v3.y, dd) + nz_popc(v3.z, dd) + nz_popc(v3.w, dd);
            }
            const uint32_t own01 = a0 | (a1 << 16), own23 = a2 | (a3 << 16);
            const uint32_t sc01 = wave_sum_to_lane63(own01), sc23 = wave_sum_to_lane63(own23);
            const uint32_t p01 = (uint32_t)__builtin_amdgcn_readlane((int)sc01, 63);
            const uint32_t p23 = (uint32_t)__builtin_amdgcn_readlane((int)sc23, 63);
            const uint32_t z0 = 8192u - (p01 & 0xffffu), z1 = 8192u - (p01 >> 16);
            const uint32_t z2 = 8192u - (p23 & 0xffffu), z3 = 8192u - (p23 >> 16);
            const uint32_t ex01 = sc01 - own01, ex23 = sc23 - own23, lb = 128u * (uint32_t)lane;
            rc[0] = run + lb - (ex01 & 0xffffu);
            rc[1] = run + z0 + lb - (ex01 >> 16);
            rc[2] = run + z0 + z1 + lb - (ex23 & 0xffffu);
            rc[3] = run + z0 + z1 + z2 + lb - (ex23 >> 16);
            run += z0 + z1 + z2 + z3;
        }
        uint32_t lo = 0, hi = 0;
        probe_chunk_l1<MODE, STRIDE, Q5>(w0, q, tab8, lo);
        probe_chunk_l1<MODE, STRIDE, Q5>(w1, q, tab8, lo);
        if (STRIDE == 1) {
            probe_chunk_l1<MODE, STRIDE, Q5>(w2, q, tab8, hi);
            probe_chunk_l1<MODE, STRIDE, Q5>(w3, q, tab8, hi);
        } else {
            probe_chunk_l1<MODE, STRIDE, Q5>(w2, q, tab8, lo);
            probe_chunk_l1<MODE, STRIDE, Q5>(w3, q, tab8, lo);
            if (STRIDE == 4) lo >>= 16;         // 16 pushes only
        }
        if (!__ballot((lo | hi) != 0u)) return;
        // second Bloom probe, strip by strip: the gram of a run-time position wants an indexed read,
        // which registers do not have, and a loop per strip keeps the indices static.  (Measured: the
        // supertile parked in LDS and ONE loop over all its first-level hits -- 5 rounds instead of 8 --
        // is within the noise of this, 3 % either way by set; it is not where the time goes: 368 of the
        // 396 VALU instructions of a stride-1 supertile are the 64 first-level probes.)
        if (MODE & 2) {
            constexpr uint32_t M = NB == 16 ? 0xffffu : (NB == 8 ? 0xffu : 0xfu);
            const uint32_t h0 = lo & M, h1 = (lo >> NB) & M;
            const uint32_t h2 = STRIDE == 1 ? (hi & M) : ((lo >> (2 * NB)) & M);
            const uint32_t h3 = STRIDE == 1 ? ((hi >> NB) & M) : ((lo >> (3 * NB)) & M);
            uint32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0;
            if (__ballot(h0 != 0u)) k0 = probe_chunk_l2<STRIDE, Q5>(h0, w0[0], w0[1], w0[2], w0[3], w0[4], tab8);
            if (__ballot(h1 != 0u)) k1 = probe_chunk_l2<STRIDE, Q5>(h1, w1[0], w1[1], w1[2], w1[3], w1[4], tab8);
            if (__ballot(h2 != 0u)) k2 = probe_chunk_l2<STRIDE, Q5>(h2, w2[0], w2[1], w2[2], w2[3], w2[4], tab8);
            if (__ballot(h3 != 0u)) k3 = probe_chunk_l2<STRIDE, Q5>(h3, w3[0], w3[1], w3[2], w3[3], w3[4], tab8);
            if (STRIDE == 1) { lo = k0 | (k1 << NB); hi = k2 | (k3 << NB); }
            else lo = k0 | (k1 << NB) | (k2 << (2 * NB)) | (k3 << (3 * NB));
        }
        emit_rounds<STRIDE>(lo, hi, s, rc, cq, qn, flush64);
    };
    // a single strip (fewer than four left in the range)
    auto single = [&](uint4 v, uint64_t st, uint32_t nx) {
        uint32_t w0[5] = {v.x | fold4, v.y | fold4, v.z | fold4, v.w | fold4, next_lane_dword(v.x, nx) | fold4};
        uint32_t lo = 0;
        probe_chunk_l1<MODE, STRIDE, Q5>(w0, q, tab8, lo);
        lo >>= 32u - NB;
        uint32_t rc[4] = {0u, 0u, 0u, 0u};
        if (!(MODE & 4)) {
            const uint32_t a0 = q.mb ? 128u - (uint32_t)__popc((uint32_t)dbm16[st * 64 + lane])
                                     : nz_popc(v.x, dd) + nz_popc(v.y, dd) + nz_popc(v.z, dd) + nz_popc(v.w, dd);
            const uint32_t sc = wave_sum_to_lane63(a0);
            rc[0] = run + 128u * (uint32_t)lane - (sc - a0);
            run += 8192u - (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
        }
        if (!__ballot(lo != 0u)) return;
        if (MODE & 2) lo = probe_chunk_l2<STRIDE, Q5>(lo, w0[0], w0[1], w0[2], w0[3], w0[4], tab8);
        emit_rounds<STRIDE>(lo, 0u, st, rc, cq, qn, flush64);
    };
    uint64_t s = s0;
    if (s + 4 <= s1) {
        const uint4 *p = text + s * 64 + lane;
        uint4 c0 = ld_stream(p), c1 = ld_stream(p + 64), c2 = ld_stream(p + 128), c3 = ld_stream(p + 192);
        for (; s + 8 <= s1; s += 4) {
            const uint4 *pn = text + (s + 4) * 64 + lane;
            uint4 n0 = ld_stream(pn), n1 = ld_stream(pn + 64), n2 = ld_stream(pn + 128), n3 = ld_stream(pn + 192);
            supertile(c0, c1, c2, c3, s, (uint32_t)__builtin_amdgcn_readlane((int)n0.x, 0));
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        }
        supertile(c0, c1, c2, c3, s, first_dword_of(s + 4));
        s += 4;
    }
    for (; s < s1; ++s) single(ld_stream(text + s * 64 + lane), s, first_dword_of(s + 1));
    if (qn) flush_candidates<AGH_MP_SLICE_CAP>(cq, qn, qn, slice, ncand, counters);
    if (lane == 0) {
        wave_totals[w] = run;
        wave_cand[w] = ncand < AGH_MP_SLICE_CAP ? ncand : AGH_MP_SLICE_CAP;
    }
}

// The last, partial strip: one wave, bytes >= n masked to a non-delimiter filler.
template <int MODE, int STRIDE, bool Q5>
__global__ __launch_bounds__(64) void k_sweep_multi_tail(const uint4 *__restrict__ text,
                                                         uint64_t n, agh_dev_query q,
                                                         const uint32_t *__restrict__ bits_g,
                                                         uint32_t *__restrict__ wave_totals,
                                                         uint64_t *__restrict__ cand,
                                                         uint32_t *__restrict__ wave_cand,
                                                         uint32_t *__restrict__ counters,
                                                         const uint32_t *__restrict__ strip_prefix,
                                                         const uint16_t *__restrict__ dbm16)
{
    // strip_prefix != NULL: a census pass already ran (dense numbered scans); wave_totals holds
    // the exclusive prefix per range and strip_prefix the per-strip offsets -- read, not written
    __shared__ uint64_t cq[AGH_MP_CQ_LEN];
    const int lane = lane_id();
    const uint64_t s = n >> AGH_STRIP_SHIFT;
    const uint64_t off = (s << AGH_STRIP_SHIFT) + (uint64_t)lane * 16u;
    const uint32_t dd = q.delim * 0x01010101u;
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    const uint32_t fold4 = (MODE & 1) ? 0x20202020u : 0u;
    uint4 v = make_uint4(fill4, fill4, fill4, fill4);
    if (off < n) {
        v = text[off >> 4];
        if (off + 16 > n) v = mask_tail(v, (int)(n - off), fill4);
    }
    uint32_t acc = 0;
    if (!(MODE & 4)) {
        if (q.mb) acc = 128u - (off < n ? (uint32_t)__popc((uint32_t)dbm16[off >> 4]) : 0u);
        else acc = nz_popc(v.x, dd) + nz_popc(v.y, dd) + nz_popc(v.z, dd) + nz_popc(v.w, dd);
    }
    uint32_t w0[5] = {v.x | fold4, v.y | fold4, v.z | fold4, v.w | fold4, next_lane_dword(v.x, fill4) | fold4};
    constexpr uint32_t NB = 16u / STRIDE;
    uint32_t hits = 0;
    probe_chunk_l1<MODE, STRIDE, Q5>(w0, q, reinterpret_cast<const uint8_t *>(bits_g), hits);   // table straight from global/L2
    hits >>= 32u - NB;
    // positions inside the text only
    if (off >= n) hits = 0;
    else if (off + 16 > n) hits &= (1u << ((uint32_t)(n - off + STRIDE - 1) / STRIDE)) - 1u;
    if ((MODE & 2) && hits)
        hits = probe_chunk_l2<STRIDE, Q5>(hits, w0[0], w0[1], w0[2], w0[3], w0[4], reinterpret_cast<const uint8_t *>(bits_g));
    const uint32_t sc = (MODE & 4) ? 0u : wave_sum_to_lane63(acc);
    const uint32_t z = (MODE & 4) ? 0u : 8192u - (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
    const uint64_t w = s / AGH_WAVE_STRIPS;
    const bool fresh = (s % AGH_WAVE_STRIPS) == 0;
    // delimiters of this wave's range in front of the strip (the verifier adds the prefix of
    // the ranges before it)
    uint32_t before = strip_prefix ? strip_prefix[s] : (fresh ? 0u : wave_totals[w]);
    before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
    uint32_t ncand = fresh ? 0u : wave_cand[w];
    ncand = (uint32_t)__builtin_amdgcn_readfirstlane((int)ncand);
    uint32_t qn = 0;
    uint64_t *slice = cand + w * AGH_MP_SLICE_CAP;
    uint32_t rc[4] = {0u, 0u, 0u, 0u};
    if (!(MODE & 4)) rc[0] = before + 128u * (uint32_t)lane - (sc - acc);
    emit_rounds<STRIDE>(hits, 0u, s, rc, cq, qn,
                        [&]() { flush_candidates<AGH_MP_SLICE_CAP>(cq, qn, 64u, slice, ncand, counters); });
    if (qn) flush_candidates<AGH_MP_SLICE_CAP>(cq, qn, qn, slice, ncand, counters);
    if (lane == 0) {
        wave_cand[w] = ncand < AGH_MP_SLICE_CAP ? ncand : AGH_MP_SLICE_CAP;
        if (!strip_prefix) wave_totals[w] = before + z;
    }
}

// ---------------------------------------------------------------------------------------
// verify: one workgroup per slice, one lane per candidate position
// ---------------------------------------------------------------------------------------
template <bool LEAN, int K>
__global__ __launch_bounds__(256) void k_verify_multi(const uint8_t *__restrict__ text,
                                                      uint64_t n, agh_dev_query q,
                                                      agh_multi_tables mt,
                                                      const uint64_t *__restrict__ cand,
                                                      const uint32_t *__restrict__ wave_cand,
                                                      const uint32_t *__restrict__ wave_prefix,
                                                      uint32_t w_begin, uint32_t nw, agh_marks mk)
{
    for (uint32_t w = w_begin + blockIdx.x; w < nw; w += gridDim.x) {
        const uint32_t cnt = wave_cand[w];
        const uint64_t *slice = cand + (uint64_t)w * AGH_MP_SLICE_CAP;
        const uint32_t wp = LEAN ? 0u : wave_prefix[w];
        for (uint32_t ci = threadIdx.x; ci < cnt; ci += 256u) {
            const uint64_t ent = slice[ci];
            const uint64_t j = ent & 0xffffffffull;
            if (j >= n) continue;
            mp_verify_at<LEAN, K>(text, n, q, mt, j, wp + (uint32_t)(ent >> 32), mk);
        }
    }
}

// ---------------------------------------------------------------------------------------
// dense hit sets: probe and verify in one kernel
// ---------------------------------------------------------------------------------------
// Every position is probed (whatever stride the tables were built for: a gram found at an offset
// the stride would have skipped still names real entries), hits are queued as in the sweep, and
// a full queue is verified on the spot -- one candidate per lane, so the lanes stay busy however
// the hits are spread over the text.  Run-time mode (fold, q, 5-byte grams): the probes are not
// what this kernel spends its time on.  Numbered scans: wave_totals / strip_prefix hold the
// exclusive prefixes of a census pass (k_sweep<0> + scan) that ran in front of this kernel.
// A wave takes AGH_DENSE_STRIPS strips (16 KiB), not a whole 256 KiB range: the kernel waits on
// the verifier's dependent loads, and 256 MiB in 256 KiB ranges are one wave per SIMD.
#define AGH_DENSE_STRIPS 16u
template <bool LEAN, int K>
__global__ __launch_bounds__(256) void k_dense_multi(const uint4 *__restrict__ text, uint64_t n,
                                                     uint64_t n_full_strips, agh_dev_query q,
                                                     agh_multi_tables mt,
                                                     const uint32_t *__restrict__ wave_totals,
                                                     const uint32_t *__restrict__ strip_prefix,
                                                     uint32_t *__restrict__ wave_cand, agh_marks mk)
{
    // The bit table stays in global memory (32 KiB: L2 / L1 resident): this kernel waits on the
    // verifier's dependent loads, not on probes, and without the table in LDS seven waves per SIMD are
    // resident instead of four
    __shared__ uint64_t cq_all[4 * AGH_MP_CQ_LEN];
    const uint8_t *tab8 = reinterpret_cast<const uint8_t *>(mt.bits);
    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const uint64_t s0 = ((uint64_t)blockIdx.x * 4 + wib) * AGH_DENSE_STRIPS;
    if (s0 >= n_full_strips) return;
    const uint64_t w = s0 / AGH_WAVE_STRIPS;    // the 256 KiB range the census numbers are kept for
    uint64_t s1 = s0 + AGH_DENSE_STRIPS;
    if (s1 > n_full_strips) s1 = n_full_strips;
    const uint8_t *text8 = reinterpret_cast<const uint8_t *>(text);
    const uint32_t dd = q.delim * 0x01010101u;
    const uint32_t fold4 = q.fold ? 0x20202020u : 0u;
    const bool q4 = q.fq == 4, q5 = q.mp_q5 != 0;
    const uint64_t n_dw = ((n + 15) & ~(uint64_t)15) / 4;
    uint32_t run = LEAN ? 0u : wave_totals[w] + strip_prefix[s0], qn = 0;
    uint64_t *cq = cq_all + wib * AGH_MP_CQ_LEN;
    // verify the first `take` queued candidates, one per lane; keep the rest
    auto verify_queue = [&](uint32_t take) {
        if ((uint32_t)lane < take) {
            const uint64_t ent = cq[lane];
            mp_verify_at<LEAN, K>(text8, n, q, mt, ent & 0xffffffffull, (uint32_t)(ent >> 32), mk);
        }
        const uint32_t rest = qn - take;
        uint64_t keep = 0;
        if ((uint32_t)lane < rest) keep = cq[take + (uint32_t)lane];
        if ((uint32_t)lane < rest) cq[lane] = keep;
        qn = rest;
    };
    for (uint64_t s = s0; s < s1; ++s) {
        const uint4 v = ld_stream(text + s * 64 + lane);
        const uint64_t i_nx = (s + 1) * 256u;
        const uint32_t wrap = i_nx < n_dw ? reinterpret_cast<const uint32_t *>(text)[i_nx] : 0u;
        const uint32_t w0[5] = {v.x | fold4, v.y | fold4, v.z | fold4, v.w | fold4, next_lane_dword(v.x, wrap) | fold4};
        uint32_t hits = 0;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int d = p >> 2, sh = p & 3;
            uint32_t g = sh ? __builtin_amdgcn_alignbyte(w0[d + 1], w0[d], sh) : w0[d];
            if (q5) g = agh_mix5(g, w0[d + 1] >> (8 * sh));     // the fifth byte (agh_mix5 takes the low one)
            const uint32_t idx = q4 ? agh_sample_prod_q4(g) : agh_sample_hash18_q3(g & q.qmask);
            hits = __builtin_amdgcn_alignbit(mp_bit(tab8, idx), hits, 1);
        }
        hits >>= 16;
        uint32_t rc[4] = {0u, 0u, 0u, 0u};
        if (!LEAN) {
            const uint32_t a0 = q.mb ? 128u - (uint32_t)__popc((uint32_t)reinterpret_cast<const uint16_t *>(mt.dbm)[s * 64 + lane])
                                     : nz_popc(v.x, dd) + nz_popc(v.y, dd) + nz_popc(v.z, dd) + nz_popc(v.w, dd);
            const uint32_t sc = wave_sum_to_lane63(a0);
            rc[0] = run + 128u * (uint32_t)lane - (sc - a0);
            run += 8192u - (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
        }
        if (!__ballot(hits != 0u)) continue;
        emit_rounds<1>(hits, 0u, s, rc, cq, qn, [&]() { verify_queue(64u); });
    }
    if (qn) verify_queue(qn);
    if (lane == 0 && s0 % AGH_WAVE_STRIPS == 0) wave_cand[w] = 0u;      // nothing went through the slices
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
template <int MODE, int STRIDE, bool Q5>
static void launch_sweep_multi_ms(const agh_sweep_args &a, hipStream_t st)
{
    // a part [w_begin, w_end) of the wave ranges (count-only scans: the verifier of one part runs
    // under the sweep of the next), or everything
    const uint64_t n_full_all = a.n >> AGH_STRIP_SHIFT;
    const bool to_end = a.w_end == 0 || (uint64_t)a.w_end * AGH_WAVE_STRIPS >= n_full_all;
    const uint64_t n_full = to_end ? n_full_all : (uint64_t)a.w_end * AGH_WAVE_STRIPS;
    const uint64_t w_hi = (n_full + AGH_WAVE_STRIPS - 1) / AGH_WAVE_STRIPS;
    const uint64_t n_waves = w_hi > a.w_begin ? w_hi - a.w_begin : 0;
    if (a.ev_begin) (void)hipEventRecord(a.ev_begin, st);
    if (n_waves && !a.tail_only)
        hipLaunchKernelGGL((k_sweep_multi<MODE, STRIDE, Q5>), dim3((uint32_t)((n_waves + 3) / 4)), dim3(256), 0, st,
                           (const uint4 *)a.text, a.n, n_full, a.q, (const uint32_t *)a.ftab, a.wave_totals,
                           a.cand, a.wave_cand, a.counters, (const uint16_t *)a.dbm, a.w_begin);
    if (a.ev_end) (void)hipEventRecord(a.ev_end, st);
    if (!to_end) return;                        // the partial last strip belongs to the last part
    if (a.n & (AGH_STRIP - 1))
        hipLaunchKernelGGL((k_sweep_multi_tail<MODE, STRIDE, Q5>), dim3(1), dim3(64), 0, st,
                           (const uint4 *)a.text, a.n, a.q, (const uint32_t *)a.ftab,
                           a.wave_totals, a.cand, a.wave_cand, a.counters,
                           (a.tail_only && !a.lean) ? (const uint32_t *)a.strip_prefix
                                                    : (const uint32_t *)nullptr,
                           (const uint16_t *)a.dbm);
}

// a.q.fh = the probe stride chosen by the host (fill_multi_tables): 1, 2 or 4; strides > 1 imply q == 4
template <int MODE>
static void launch_sweep_multi_m(const agh_sweep_args &a, hipStream_t st)
{
    if ((MODE & 2) && a.q.fh == 4 && a.q.mp_q5) launch_sweep_multi_ms<MODE, 4, true>(a, st);
    else if ((MODE & 2) && a.q.fh == 4) launch_sweep_multi_ms<MODE, 4, false>(a, st);
    else if ((MODE & 2) && a.q.fh == 2) launch_sweep_multi_ms<MODE, 2, false>(a, st);
    else launch_sweep_multi_ms<MODE, 1, false>(a, st);
}

// Multi-pattern sweep; a.ftab = the 2^18-bit table.  The prefix scan of the census (numbered scans)
// is launched by the caller through agh_launch_census_scan().  a.tail_only: only the partial last
// strip (the dense kernel handled the full strips).
void agh_launch_sweep_multi(const agh_sweep_args &a, hipStream_t st)
{
    const int mode = (a.q.fold ? 1 : 0) | (a.q.fq == 4 ? 2 : 0) | (a.lean ? 4 : 0);
    switch (mode) {
    case 0: launch_sweep_multi_m<0>(a, st); break;
    case 1: launch_sweep_multi_m<1>(a, st); break;
    case 2: launch_sweep_multi_m<2>(a, st); break;
    case 3: launch_sweep_multi_m<3>(a, st); break;
    case 4: launch_sweep_multi_m<4>(a, st); break;
    case 5: launch_sweep_multi_m<5>(a, st); break;
    case 6: launch_sweep_multi_m<6>(a, st); break;
    default: launch_sweep_multi_m<7>(a, st); break;
    }
}

#define AGH_K_SWITCH(MACRO)                                                                   \
    switch (a.q.k) {                                                                          \
        MACRO(0) MACRO(1) MACRO(2) MACRO(3) MACRO(4) MACRO(5) MACRO(6) MACRO(7) MACRO(8)      \
    default: break;                                                                           \
    }

// Dense hit sets: probe + verify of all full strips in one kernel (numbered scans: after a census
// pass); the partial last strip goes through agh_launch_sweep_multi(tail_only) + agh_launch_verify_multi.
void agh_launch_dense_multi(const agh_sweep_args &a, const agh_multi_dev &m, const agh_marks &mk,
                            hipStream_t st)
{
    const uint64_t n_full = a.n >> AGH_STRIP_SHIFT;
    const uint64_t n_waves = (n_full + AGH_DENSE_STRIPS - 1) / AGH_DENSE_STRIPS;
    if (!n_waves) return;
    const uint32_t blocks = (uint32_t)((n_waves + 3) / 4);
    const bool lean = a.lean != 0;
#define AGH_DM_CASE(KK)                                                                       \
    case KK:                                                                                  \
        if (lean)                                                                             \
            hipLaunchKernelGGL((k_dense_multi<true, KK>), dim3(blocks), dim3(256), 0, st,     \
                               (const uint4 *)a.text, a.n, n_full, a.q, m, a.wave_totals,     \
                               (const uint32_t *)a.strip_prefix, a.wave_cand, mk);            \
        else                                                                                  \
            hipLaunchKernelGGL((k_dense_multi<false, KK>), dim3(blocks), dim3(256), 0, st,    \
                               (const uint4 *)a.text, a.n, n_full, a.q, m, a.wave_totals,     \
                               (const uint32_t *)a.strip_prefix, a.wave_cand, mk);            \
        break;
    AGH_K_SWITCH(AGH_DM_CASE)
#undef AGH_DM_CASE
}

void agh_launch_verify_multi(const agh_scan_args &a, const agh_multi_dev &m, bool lean,
                             hipStream_t st)
{
    // slices [w_begin, w_end) of a part, else all
    const uint32_t w_hi = (a.w_end && a.w_end < a.nw) ? a.w_end : a.nw;
    if (w_hi <= a.w_begin) return;
    const uint32_t blocks = w_hi - a.w_begin > 65536u ? 65536u : w_hi - a.w_begin;
#define AGH_VM_CASE(KK)                                                                       \
    case KK:                                                                                  \
        if (lean)                                                                             \
            hipLaunchKernelGGL((k_verify_multi<true, KK>), dim3(blocks), dim3(256), 0, st,    \
                               (const uint8_t *)a.text, a.n, a.q, m, a.cand, a.wave_cand,     \
                               a.wave_prefix, a.w_begin, w_hi, a.mk);                         \
        else                                                                                  \
            hipLaunchKernelGGL((k_verify_multi<false, KK>), dim3(blocks), dim3(256), 0, st,   \
                               (const uint8_t *)a.text, a.n, a.q, m, a.cand, a.wave_cand,     \
                               a.wave_prefix, a.w_begin, w_hi, a.mk);                         \
        break;
    AGH_K_SWITCH(AGH_VM_CASE)
#undef AGH_VM_CASE
}
