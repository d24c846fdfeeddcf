// agh_multi_inl.h -- device helpers of the multi-pattern (-f) engines shared by agh_multi.hip (sweep /
// verify / dense kernels) and agh_mscan.hip (the one-pass count-only scan): table probes and the
// verification of one candidate position (exact entries, k = 1 side check, k-error window walk).
#pragma once
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

