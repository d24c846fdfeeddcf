// agh_multi.hip -- exact multi-pattern scan (-f patternfile; the role of newmgrep.c:
// prepf() tables + monkey1() hash-and-compare, newmgrep.c:192-375, 839-1012).
//
// A record matches iff it contains any pattern verbatim.  Same shape as the single-pattern
// scan, with the filter turned around: instead of a few aligned text samples against all
// q-grams of one pattern, EVERY text position's q-byte prefix (q = min(4, shortest pattern))
// is probed against the set of pattern prefixes -- a 2^18-bit table held in LDS (32 KiB), so
// 1024 patterns fill 0.4 % of it.  Hits go through the same per-wave LDS queue and private
// slices; k_verify_multi then walks the bucket of patterns sharing that prefix and compares
// bytes.  Census / lean record identity / counting are the single-pattern machinery.
#include "agh_verify_inl.h"

struct agh_multi_tables {
    const uint32_t *bits;          // 2^18-bit prefix table (global copy, 32 KiB)
    const uint32_t *bucket_start;  // (1 << AGH_MP_BUCKET_BITS) + 1 offsets into bucket_items
    const uint32_t *bucket_items;  // pattern numbers grouped by prefix bucket
    const uint32_t *pat_off;       // npat + 1 offsets into pool
    const uint8_t *pool;           // pattern bytes (lower-cased when the query folds case)
    // k-error queries (agh_query_multi_approx): the table entries are PIECES of patterns
    const uint32_t *piece_owner;   // pattern number a piece was cut from
    const uint8_t *piece_po;       // its offset inside that pattern
    const uint8_t *owner_len;      // pattern lengths (<= 32)
    const uint32_t *owner_mask;    // [pattern][256] position masks (bit p-1 = position p)
    const uint32_t *item_info;     // per bucket item: (pool offset << 8) | length -- saves the exact
                                   // verifier two dependent loads (bucket_items -> pat_off -> pool)
};

#define AGH_MP_WORDS ((1u << AGH_MP_BITS) / 32u)

template <int MODE>   // bit 0: fold case, bit 1: q == 4, bit 2: lean (no census)
__device__ __forceinline__ uint32_t probe_bit(uint32_t g, const agh_dev_query &q,
                                              const uint32_t *tab)
{
    // the table read as bytes: byte index = hash >> 3 (one v_bfe of the product), bit = hash & 7
    const uint8_t *tab8 = reinterpret_cast<const uint8_t *>(tab);
    if (MODE & 2) {
        const uint32_t p = agh_sample_prod18_q4((MODE & 1) ? (g | q.fold) : g);
        // v_bfe with a register offset: one instruction for (byte >> bit) & 1
        return __builtin_amdgcn_ubfe((uint32_t)tab8[p >> 17], (p >> 14) & 7u, 1u);
    }
    const uint32_t h = agh_sample_hash18_q3((MODE & 1) ? ((g & q.qmask) | q.fold) : (g & q.qmask));
    return __builtin_amdgcn_ubfe((uint32_t)tab8[h >> 3], h & 7u, 1u);
}

// the probed positions of one 16-byte chunk: every STRIDE-th byte (nx = the 4 bytes that follow it)
template <int MODE, int STRIDE, bool Q5>
__device__ __forceinline__ uint32_t probe_chunk(uint4 v, uint32_t nx, const agh_dev_query &q,
                                                const uint32_t *tab)
{
    const uint32_t w[5] = {v.x, v.y, v.z, v.w, nx};
    uint32_t hits = 0;
#pragma unroll
    for (int p = 0; p < 16; p += STRIDE) {
        const int d = p >> 2, sh = p & 3;
        uint32_t g = sh ? __builtin_amdgcn_alignbyte(w[d + 1], w[d], sh) : w[d];
        if (Q5) {                               // stride 4: the fifth byte is the next dword's first
            if (MODE & 1) g |= q.fold;
            g = agh_mix5(g, (MODE & 1) ? (w[d + 1] | 0x20u) : w[d + 1]);
        }
        hits = (probe_bit<Q5 ? (MODE & ~1) : MODE>(g, q, tab) << p) | hits;       // v_lshl_or_b32
    }
    if ((MODE & 2) && __ballot(hits != 0)) {
        // q == 4: the table is a two-probe Bloom filter.  First-level hits (0.4-0.8 % of all
        // positions with ~1000 patterns, mostly hash false positives) take the second probe;
        // what survives is almost only real prefix occurrences.
        uint32_t h = hits, keep = 0;
        while (h) {
            const uint32_t p = (uint32_t)__ffs((int)h) - 1u;
            h &= h - 1u;
            const uint32_t d = p >> 2, sh = p & 3u;
            const uint32_t lo = d == 0 ? w[0] : (d == 1 ? w[1] : (d == 2 ? w[2] : w[3]));
            const uint32_t hi = d == 0 ? w[1] : (d == 1 ? w[2] : (d == 2 ? w[3] : w[4]));
            uint32_t g = __builtin_amdgcn_alignbyte(hi, lo, sh);
            if (MODE & 1) g |= q.fold;
            if (Q5) g = agh_mix5(g, (MODE & 1) ? (hi | 0x20u) : hi);
            const uint32_t h2 = agh_sample_hash18b_q4(g);
            keep |= ((tab[h2 >> 5] >> (h2 & 31u)) & 1u) << p;
        }
        hits = keep;
    }
    return hits;
}

// hits16: bit p of lane l = text position (strip*1024 + l*16 + p); rc = delimiters in front of
// the lane's chunk (census scans).  Entry = (rc << 32) | byte offset.
template <typename OnFull>
__device__ __forceinline__ void emit_positions(uint32_t hits16, uint64_t strip, uint32_t rc,
                                               uint64_t *cq, uint32_t &qn, OnFull on_full)
{
    uint64_t hm = __ballot(hits16 != 0);
    const int lane = lane_id();
    while (hm) {
        const int l = __ffsll((long long)hm) - 1;
        hm &= hm - 1;
        const uint32_t hbits = (uint32_t)__builtin_amdgcn_readlane((int)hits16, l);
        const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)rc, l);
        const int c = __popc(hbits);
        if (lane < c) {
            uint32_t t = hbits;
            for (int j = 0; j < lane; ++j) t &= t - 1;
            const uint32_t b = (uint32_t)__ffs((int)t) - 1u;
            const uint32_t off = (uint32_t)(strip * AGH_STRIP + (uint64_t)l * 16u + b);
            cq[qn + (uint32_t)lane] = ((uint64_t)r << 32) | off;
        }
        qn += (uint32_t)c;
        if (qn >= 64u) on_full();
    }
}

// ASCII upper -> lower in four bytes at once (newmgrep.c: tr[] folds case under -i).
__device__ __forceinline__ uint32_t swar_lower(uint32_t t)
{
    const uint32_t x = t & 0x7f7f7f7fu;
    const uint32_t ge = x + 0x3f3f3f3fu;            // bit 7 of a byte <=> byte >= 'A'
    const uint32_t gt = x + 0x25252525u;            // bit 7 <=> byte > 'Z'
    return t | (((ge & ~gt & ~t) & 0x80808080u) >> 2);
}

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

// Does any pattern occur at text position j?  Walks the bucket of patterns sharing the prefix;
// the text window is fetched once (16 unaligned bytes), patterns of up to 16 bytes are compared
// as four masked dwords (the pool is padded so that 16 bytes can always be read).
__device__ __forceinline__ bool multi_match_at(const uint8_t *__restrict__ text, uint64_t n,
                                               const agh_dev_query &q, const agh_multi_tables &mt,
                                               uint64_t j)
{
    const bool fold = q.fold != 0;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    uint32_t tw[4];
    if (j + 16 <= n16) {
        const u32x4_u v = *reinterpret_cast<const u32x4_u *>(text + j);
        tw[0] = v[0]; tw[1] = v[1]; tw[2] = v[2]; tw[3] = v[3];
    } else {                                        // the last bytes of the text
        tw[0] = tw[1] = tw[2] = tw[3] = 0;
        for (uint32_t t = 0; t < 16 && j + t < n; ++t) tw[t >> 2] |= (uint32_t)text[j + t] << (8 * (t & 3));
    }
    const uint32_t g = (tw[0] & q.qmask) | q.fold;  // the probed q-gram at j
    if (fold) {
#pragma unroll
        for (int d = 0; d < 4; ++d) tw[d] = swar_lower(tw[d]);
    }
    const uint32_t b = agh_mp_bucket(g);
    for (uint32_t it = mt.bucket_start[b]; it < mt.bucket_start[b + 1]; ++it) {
        const uint32_t info = mt.item_info[it];
        const uint32_t o = info >> 8, len = info & 0xffu;
        const uint32_t go = mt.bucket_items[it] >> 28;     // the gram sits at this offset of the entry
        if (go) {                                          // strided probing: the entry starts in front of j
            if (j < go) continue;
            if (multi_entry_at(text, n, fold, mt.pool + o, len, j - go)) return true;
            continue;
        }
        if (j + len > n) continue;
        const u32x4_u pv = *reinterpret_cast<const u32x4_u *>(mt.pool + o);
        const uint32_t head = len < 16u ? len : 16u;
        uint32_t diff = 0;
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
            const uint32_t nb = head > 4u * d ? head - 4u * d : 0u;      // bytes of this dword in play
            const uint32_t m = nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u);
            diff |= (tw[d] ^ pv[d]) & m;
        }
        if (diff) continue;
        uint32_t t = 16;
        for (; t < len; ++t) {                      // patterns longer than 16 bytes: the rest
            uint32_t c = text[j + t];
            if (fold && c >= 'A' && c <= 'Z') c += 32u;
            if (c != mt.pool[o + t]) break;
        }
        if (t >= len) return true;
    }
    return false;
}

// ---- -f with errors ---------------------------------------------------------------------
// A record matches iff it holds a substring within edit distance k of ANY pattern (union of the
// single-pattern predicate; the reference itself ignores -# together with -f, compat.c:34-37).
// Partition filter: an occurrence with <= k errors contains at least one of k+1 disjoint pieces
// of its pattern verbatim.  The pieces are the entries of the exact multi-pattern tables; a
// verbatim piece at text position j sends the k-error automaton of its pattern over the bytes
// the occurrence can cover: [j - po - k, j + (m - po) + k).
struct AutomatonRT {                    // k is a run-time value here (one kernel for every k)
    uint32_t R[AGH_MAX_ERRORS_DEV + 1];
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int e = 0; e <= AGH_MAX_ERRORS_DEV; ++e) R[e] = 0;
    }
    __device__ __forceinline__ bool step(uint32_t cm, uint32_t finalbit, int k)
    {
        uint32_t po = R[0];
        uint32_t pn = ((po << 1) | 1u) & cm;
        R[0] = pn;
        uint32_t top = pn;
#pragma unroll
        for (int e = 1; e <= AGH_MAX_ERRORS_DEV; ++e) {
            if (e <= k) {
                const uint32_t cur = R[e];
                const uint32_t ne = (((cur << 1) | 1u) & cm) | po | (((po | pn) << 1) | 1u);
                po = cur;
                pn = ne;
                R[e] = ne;
                top = ne;
            }
        }
        return (top & finalbit) != 0;
    }
};

// The automaton of one pattern over [ws, we); same record bookkeeping as verify_window_slow.
template <bool LEAN>
__device__ __noinline__ void approx_window(const uint8_t *__restrict__ text, uint64_t n,
                                           const agh_dev_query &q,
                                           const uint32_t *__restrict__ pmask, uint32_t m,
                                           uint64_t ws, uint64_t we, uint64_t anchor,
                                           uint32_t rc_anchor, const agh_marks &mk)
{
    const uint32_t finalbit = 1u << (m - 1);
    const int k = (int)q.k;
    uint32_t rec = 0;
    uint64_t rstart = 0;
    if (LEAN) {
        rstart = lean_record_start(text, ws, q.delim, mk);
        if (rstart == ~0ull) return;
    } else {
        uint32_t back = 0;                      // delimiters in [ws, anchor)
        for (uint64_t i = ws; i < anchor; ++i) back += (text[i] == q.delim);
        rec = rc_anchor - back;
    }
    AutomatonRT A;
    A.reset();
    bool seen = false;
    if (ws == 0) A.step(pmask[q.head_byte], finalbit, k);
    for (uint64_t i = ws; i < we; ++i) {
        const uint32_t c = text[i];
        if (A.step(pmask[c], finalbit, k) && !seen) {
            seen = true;
            if (LEAN) lean_insert(mk, rstart); else mark_record(mk, rec, i);
        }
        if (c == q.delim) {
            A.reset();
            ++rec;
            rstart = i + 1;
            seen = false;
            A.step(pmask[c], finalbit, k);      // patterns never hold the delimiter byte
        }
    }
    if (we == n && q.tail_virtual) {            // asearch.c:87-91
        if (A.step(pmask[q.delim], finalbit, k) && !seen) {
            if (LEAN) lean_insert(mk, rstart); else mark_record(mk, rec, n);
        }
    }
}

// Every piece that occurs verbatim at text position j gets its pattern verified.
template <bool LEAN>
__device__ __noinline__ void multi_approx_at(const uint8_t *__restrict__ text, uint64_t n,
                                             const agh_dev_query &q, const agh_multi_tables &mt,
                                             uint64_t j, uint32_t rc_chunk, const agh_marks &mk)
{
    const uint32_t fold = q.fold ? 0x20u : 0u;
    uint32_t g = 0;
    for (uint32_t t = 0; t < (uint32_t)q.fq && j + t < n; ++t) g |= (uint32_t)text[j + t] << (8 * t);
    g = (g & q.qmask) | q.fold;
    const uint32_t b = agh_mp_bucket(g);
    for (uint32_t it = mt.bucket_start[b]; it < mt.bucket_start[b + 1]; ++it) {
        const uint32_t pc = mt.bucket_items[it] & 0x0fffffffu, go = mt.bucket_items[it] >> 28;
        const uint32_t o = mt.pat_off[pc], len = mt.pat_off[pc + 1] - o;
        if (j < go) continue;
        const uint64_t js = j - go;                         // where the piece starts
        if (!multi_entry_at(text, n, fold != 0, mt.pool + o, len, js)) continue;
        const uint32_t owner = mt.piece_owner[pc], po = mt.piece_po[pc], m = mt.owner_len[owner];
        const uint64_t anchor = j & ~(uint64_t)15;          // rc_chunk = delimiters in front of it
        const uint64_t back = (uint64_t)po + q.k;
        uint64_t ws = js > back ? js - back : 0;
        if (ws > anchor) ws = anchor;
        uint64_t we = js + (m - po) + q.k;
        if (we > n) we = n;
        approx_window<LEAN>(text, n, q, mt.owner_mask + (size_t)owner * 256u, m, ws, we, anchor,
                            rc_chunk, mk);
    }
}

// ---- the same with the number of errors known at compile time (k_verify_multi<LEAN, K>) --------
// What made the run-time version slow: eight predicated levels per byte whatever k is, one
// dependent global load (the pattern's mask of the text byte) in front of every automaton step,
// byte-wise piece compares.  Here the piece is compared as masked dwords against the window that is
// fetched once, the automaton has exactly K+1 levels, and the masks of 16 text bytes are fetched
// together (16 independent loads in flight) before the 16 steps run out of registers.
template <bool LEAN, int K>
__device__ __forceinline__ void approx_window_k(const uint8_t *__restrict__ text, uint64_t n,
                                                const agh_dev_query &q,
                                                const uint32_t *__restrict__ pmask, uint32_t m,
                                                uint64_t ws, uint64_t we, uint64_t anchor,
                                                uint32_t rc_anchor, const agh_marks &mk)
{
    const uint32_t finalbit = 1u << (m - 1);
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    uint32_t rec = 0;
    uint64_t rstart = 0;
    if (LEAN) {
        rstart = lean_record_start(text, ws, q.delim, mk);
        if (rstart == ~0ull) return;
    } else {
        uint32_t back = 0;                      // delimiters in [ws, anchor)
        for (uint64_t i = ws; i < anchor; ++i) back += (text[i] == q.delim);
        rec = rc_anchor - back;
    }
    Automaton<uint32_t, K> A;
    A.reset();
    bool seen = false;
    if (ws == 0) A.step(pmask[q.head_byte], finalbit);
    for (uint64_t i0 = ws; i0 < we; i0 += 16) {
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
        for (int t = 0; t < 16; ++t) cms[t] = pmask[(dws[t >> 2] >> (8 * (t & 3))) & 0xffu];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if ((uint32_t)t < nb) {
                const uint32_t c = (dws[t >> 2] >> (8 * (t & 3))) & 0xffu;
                if (A.step(cms[t], finalbit) && !seen) {
                    seen = true;
                    if (LEAN) lean_insert(mk, rstart); else mark_record(mk, rec, i0 + t);
                }
                if (c == q.delim) {
                    A.reset();
                    ++rec;
                    rstart = i0 + t + 1;
                    seen = false;
                    A.step(cms[t], finalbit);   // patterns never hold the delimiter byte
                }
            }
        }
    }
    if (we == n && q.tail_virtual) {            // asearch.c:87-91
        if (A.step(pmask[q.delim], finalbit) && !seen) {
            if (LEAN) lean_insert(mk, rstart); else mark_record(mk, rec, n);
        }
    }
}

template <bool LEAN, int K>
__device__ __forceinline__ void multi_approx_at_k(const uint8_t *__restrict__ text, uint64_t n,
                                                  const agh_dev_query &q, const agh_multi_tables &mt,
                                                  uint64_t j, uint32_t rc_chunk, const agh_marks &mk)
{
    const bool fold = q.fold != 0;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    uint32_t tw[4];
    if (j + 16 <= n16) {
        const u32x4_u v = *reinterpret_cast<const u32x4_u *>(text + j);
        tw[0] = v[0]; tw[1] = v[1]; tw[2] = v[2]; tw[3] = v[3];
    } else {
        tw[0] = tw[1] = tw[2] = tw[3] = 0;
        for (uint32_t t = 0; t < 16 && j + t < n; ++t) tw[t >> 2] |= (uint32_t)text[j + t] << (8 * (t & 3));
    }
    const uint32_t g = (tw[0] & q.qmask) | q.fold;
    if (fold) {
#pragma unroll
        for (int d = 0; d < 4; ++d) tw[d] = swar_lower(tw[d]);
    }
    const uint32_t b = agh_mp_bucket(g);
    for (uint32_t it = mt.bucket_start[b]; it < mt.bucket_start[b + 1]; ++it) {
        const uint32_t info = mt.item_info[it];
        const uint32_t o = info >> 8, len = info & 0xffu;
        const uint32_t pc = mt.bucket_items[it] & 0x0fffffffu, go = mt.bucket_items[it] >> 28;
        if (j < go) continue;
        const uint64_t js = j - go;                         // where the piece starts
        if (go) {
            if (!multi_entry_at(text, n, fold, mt.pool + o, len, js)) continue;
        } else {
            if (j + len > n) continue;
            const u32x4_u pv = *reinterpret_cast<const u32x4_u *>(mt.pool + o);
            const uint32_t head = len < 16u ? len : 16u;
            uint32_t diff = 0;
#pragma unroll
            for (uint32_t d = 0; d < 4; ++d) {
                const uint32_t nb = head > 4u * d ? head - 4u * d : 0u;
                const uint32_t mk4 = nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u);
                diff |= (tw[d] ^ pv[d]) & mk4;
            }
            if (diff) continue;
            uint32_t t = 16;
            for (; t < len; ++t) {
                uint32_t c = text[j + t];
                if (fold && c >= 'A' && c <= 'Z') c += 32u;
                if (c != mt.pool[o + t]) break;
            }
            if (t < len) continue;
        }
        const uint32_t owner = mt.piece_owner[pc], po = mt.piece_po[pc], m = mt.owner_len[owner];
        const uint64_t anchor = j & ~(uint64_t)15;          // rc_chunk = delimiters in front of it
        const uint64_t back = (uint64_t)po + K;
        uint64_t ws = js > back ? js - back : 0;
        if (ws > anchor) ws = anchor;
        uint64_t we = js + (m - po) + K;
        if (we > n) we = n;
        approx_window_k<LEAN, K>(text, n, q, mt.owner_mask + (size_t)owner * 256u, m, ws, we, anchor,
                                 rc_chunk, mk);
    }
}

// A verified occurrence at j: count its record once.
template <bool LEAN>
__device__ __forceinline__ void multi_mark(const uint8_t *__restrict__ text, const agh_dev_query &q,
                                           const agh_marks &mk, uint64_t j, uint32_t rc_chunk)
{
    if (LEAN) {
        const uint64_t st = lean_record_start(text, j, q.delim, mk);
        if (st != ~0ull) lean_insert(mk, st);
    } else {
        // record number = delimiters in front of the chunk + delimiters in [chunk, j)
        uint32_t rec = rc_chunk;
        for (uint64_t i = j & ~(uint64_t)15; i < j; ++i) rec += (text[i] == q.delim);
        mark_record(mk, rec, j);
    }
}

// One wave per 256 KiB range, 4 KiB supertiles, next supertile prefetched -- as k_sweep.
// INLINE: dense hit sets (many 1..3-byte patterns) overflow the candidate slices; then every
// hit is checked on the spot by its own lane (slow, but no buffer can overflow).  Only lean /
// count-only bookkeeping is done inline.
template <int MODE, bool INLINE, int STRIDE, bool Q5>
__global__ __launch_bounds__(256) void k_sweep_multi(const uint4 *__restrict__ text, uint64_t n,
                                                     uint64_t n_full_strips, agh_dev_query q,
                                                     const uint32_t *__restrict__ bits_g,
                                                     uint32_t *__restrict__ wave_totals,
                                                     uint64_t *__restrict__ cand,
                                                     uint32_t *__restrict__ wave_cand,
                                                     uint32_t *__restrict__ counters,
                                                     agh_multi_tables mt, agh_marks mk)
{
    // INLINE numbered scans: wave_totals already holds the exclusive prefix of a census pass
    __shared__ __attribute__((aligned(16))) uint32_t tab[AGH_MP_WORDS];
    __shared__ uint64_t cq_all[4 * (AGH_CQ_LEN + 16)];
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
    const int lane = lane_id();
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const uint64_t w = (uint64_t)blockIdx.x * 4 + wib;
    const uint64_t s0 = w * AGH_WAVE_STRIPS;
    if (s0 >= n_full_strips) return;
    uint64_t s1 = s0 + AGH_WAVE_STRIPS;
    if (s1 > n_full_strips) s1 = n_full_strips;
    const uint32_t dd = q.delim * 0x01010101u;
    const uint32_t *text32 = reinterpret_cast<const uint32_t *>(text);
    const uint64_t n_dw = ((n + 15) & ~(uint64_t)15) / 4;      // readable dwords
    uint32_t run = (INLINE && !(MODE & 4)) ? wave_totals[w] : 0u, ncand = 0, qn = 0;
    uint64_t *cq = cq_all + wib * (AGH_CQ_LEN + 16);
    uint64_t *slice = cand + w * AGH_MP_SLICE_CAP;
    auto flush64 = [&]() { flush_candidates<AGH_MP_SLICE_CAP>(cq, qn, 64u, slice, ncand, counters); };
    // the dword right behind chunk (strip st, this lane)
    auto next_dw = [&](uint64_t st) -> uint32_t {
        const uint64_t i = (st * 64 + (uint64_t)lane) * 4 + 4;
        return i < n_dw ? text32[i] : 0u;
    };
    auto strip_work = [&](uint4 v, uint32_t nx, uint64_t st) {
        uint32_t acc = 0;
        if (!(MODE & 4)) acc = nz_popc(v.x, dd) + nz_popc(v.y, dd) + nz_popc(v.z, dd) + nz_popc(v.w, dd);
        const uint32_t hits = probe_chunk<MODE, STRIDE, Q5>(v, nx, q, tab);
        uint32_t rc = 0, z = 0;
        if (!(MODE & 4)) {
            const uint32_t sc = wave_sum_to_lane63(acc);
            z = 8192u - (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
            rc = run + 128u * (uint32_t)lane - (sc - acc);
        }
        if (INLINE) {
            uint32_t h = hits;
            const uint64_t base = st * AGH_STRIP + (uint64_t)lane * 16u;
            while (h) {
                const uint32_t b = (uint32_t)__ffs((int)h) - 1u;
                h &= h - 1u;
                if (q.k)
                    multi_approx_at<(MODE & 4) != 0>(reinterpret_cast<const uint8_t *>(text), n, q,
                                                     mt, base + b, rc, mk);
                else if (multi_match_at(reinterpret_cast<const uint8_t *>(text), n, q, mt, base + b))
                    multi_mark<(MODE & 4) != 0>(reinterpret_cast<const uint8_t *>(text), q, mk,
                                                base + b, rc);
            }
        } else if (__ballot(hits != 0)) {
            emit_positions(hits, st, rc, cq, qn, flush64);
        }
        run += z;
    };
    uint64_t s = s0;
    if (s + 4 <= s1) {
        const uint4 *p = text + s * 64 + lane;
        uint4 c0 = ld_stream(p), c1 = ld_stream(p + 64), c2 = ld_stream(p + 128), c3 = ld_stream(p + 192);
        uint32_t x0 = next_dw(s), x1 = next_dw(s + 1), x2 = next_dw(s + 2), x3 = next_dw(s + 3);
        for (; s + 8 <= s1; s += 4) {
            const uint4 *pn = text + (s + 4) * 64 + lane;
            uint4 n0 = ld_stream(pn), n1 = ld_stream(pn + 64), n2 = ld_stream(pn + 128), n3 = ld_stream(pn + 192);
            uint32_t y0 = next_dw(s + 4), y1 = next_dw(s + 5), y2 = next_dw(s + 6), y3 = next_dw(s + 7);
            strip_work(c0, x0, s); strip_work(c1, x1, s + 1);
            strip_work(c2, x2, s + 2); strip_work(c3, x3, s + 3);
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        }
        strip_work(c0, x0, s); strip_work(c1, x1, s + 1);
        strip_work(c2, x2, s + 2); strip_work(c3, x3, s + 3);
        s += 4;
    }
    for (; s < s1; ++s) strip_work(text[s * 64 + lane], next_dw(s), s);
    if (!INLINE && qn) flush_candidates<AGH_MP_SLICE_CAP>(cq, qn, qn, slice, ncand, counters);
    if (lane == 0 && !INLINE) {
        wave_totals[w] = run;
        wave_cand[w] = ncand < AGH_MP_SLICE_CAP ? ncand : AGH_MP_SLICE_CAP;
    }
    if (lane == 0 && INLINE) wave_cand[w] = 0u;
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
                                                         const uint32_t *__restrict__ strip_prefix)
{
    // strip_prefix != NULL: a census pass already ran (dense numbered scans); wave_totals holds
    // the exclusive prefix per range and strip_prefix the per-strip offsets -- read, not written
    __shared__ uint64_t cq[AGH_CQ_LEN + 16];
    const int lane = lane_id();
    const uint64_t s = n >> AGH_STRIP_SHIFT;
    const uint64_t off = (s << AGH_STRIP_SHIFT) + (uint64_t)lane * 16u;
    const uint32_t dd = q.delim * 0x01010101u;
    const uint32_t fill4 = (~q.delim & 0xffu) * 0x01010101u;
    const uint64_t n16 = (n + 15) & ~(uint64_t)15;
    uint4 v = make_uint4(fill4, fill4, fill4, fill4);
    uint32_t nx = fill4;
    if (off < n) {
        v = text[off >> 4];
        if (off + 16 > n) v = mask_tail(v, (int)(n - off), fill4);
        if (off + 16 < n16) {
            nx = reinterpret_cast<const uint32_t *>(text)[(off >> 2) + 4];
            if (off + 20 > n) {
                const int keep = (int)(n > off + 16 ? n - off - 16 : 0);
                nx = keep >= 4 ? nx : ((nx & ((1u << (8 * keep)) - 1u)) | (fill4 & ~((1u << (8 * keep)) - 1u)));
            }
        }
    }
    uint32_t acc = 0;
    if (!(MODE & 4)) acc = nz_popc(v.x, dd) + nz_popc(v.y, dd) + nz_popc(v.z, dd) + nz_popc(v.w, dd);
    uint32_t hits = probe_chunk<MODE, STRIDE, Q5>(v, nx, q, bits_g);    // table straight from global/L2
    if (off >= n) hits = 0;
    else if (off + 16 > n) hits &= (1u << (n - off)) - 1u;      // positions inside the text only
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
    if (__ballot(hits != 0)) {
        const uint32_t rc = (MODE & 4) ? 0u : before + 128u * (uint32_t)lane - (sc - acc);
        emit_positions(hits, s, rc, cq, qn,
                       [&]() { flush_candidates<AGH_MP_SLICE_CAP>(cq, qn, 64u, slice, ncand, counters); });
        if (qn) flush_candidates<AGH_MP_SLICE_CAP>(cq, qn, qn, slice, ncand, counters);
    }
    if (lane == 0) {
        wave_cand[w] = ncand < AGH_MP_SLICE_CAP ? ncand : AGH_MP_SLICE_CAP;
        if (!strip_prefix) wave_totals[w] = before + z;
    }
}

// One lane per candidate position: walk the bucket of patterns with that prefix, compare.
// K = 0: exact patterns; K = 1..8: -f with K errors (the candidate is a verbatim piece).
template <bool LEAN, int K>
__global__ __launch_bounds__(256) void k_verify_multi(const uint8_t *__restrict__ text,
                                                      uint64_t n, agh_dev_query q,
                                                      agh_multi_tables mt,
                                                      const uint64_t *__restrict__ cand,
                                                      const uint32_t *__restrict__ wave_cand,
                                                      const uint32_t *__restrict__ wave_prefix,
                                                      uint32_t nw, agh_marks mk)
{
    const uint32_t w = blockIdx.x * 4 + threadIdx.x / WAVE;
    if (w >= nw) return;
    const uint32_t cnt = wave_cand[w];
    const uint64_t *slice = cand + (uint64_t)w * AGH_MP_SLICE_CAP;
    const uint32_t wp = LEAN ? 0u : wave_prefix[w];
    for (uint32_t ci = (uint32_t)lane_id(); ci < cnt; ci += WAVE) {
        const uint64_t ent = slice[ci];
        const uint64_t j = ent & 0xffffffffull;
        if (j >= n) continue;
        if (K) multi_approx_at_k<LEAN, K>(text, n, q, mt, j, wp + (uint32_t)(ent >> 32), mk);
        else if (multi_match_at(text, n, q, mt, j)) multi_mark<LEAN>(text, q, mk, j, wp + (uint32_t)(ent >> 32));
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
template <int MODE, int STRIDE, bool Q5>
static void launch_sweep_multi_ms(const agh_sweep_args &a, const agh_multi_dev &m,
                                  const agh_marks &mk, bool inl, hipStream_t st)
{
    const uint64_t n_full = a.n >> AGH_STRIP_SHIFT;
    const uint64_t n_waves = (n_full + AGH_WAVE_STRIPS - 1) / AGH_WAVE_STRIPS;
    if (a.ev_begin) (void)hipEventRecord(a.ev_begin, st);
    agh_multi_tables mt;
    mt.bits = m.bits;
    mt.bucket_start = m.bucket_start;
    mt.bucket_items = m.bucket_items;
    mt.pat_off = m.pat_off;
    mt.pool = m.pool;
    mt.piece_owner = m.piece_owner;
    mt.piece_po = m.piece_po;
    mt.owner_len = m.owner_len;
    mt.owner_mask = m.owner_mask;
    mt.item_info = m.item_info;
    if (n_waves && inl)
        hipLaunchKernelGGL((k_sweep_multi<MODE, true, STRIDE, Q5>), dim3((uint32_t)((n_waves + 3) / 4)),
                           dim3(256), 0, st, (const uint4 *)a.text, a.n, n_full, a.q,
                           (const uint32_t *)a.ftab, a.wave_totals, a.cand, a.wave_cand,
                           a.counters, mt, mk);
    else if (n_waves)
        hipLaunchKernelGGL((k_sweep_multi<MODE, false, STRIDE, Q5>), dim3((uint32_t)((n_waves + 3) / 4)),
                           dim3(256), 0, st, (const uint4 *)a.text, a.n, n_full, a.q,
                           (const uint32_t *)a.ftab, a.wave_totals, a.cand, a.wave_cand,
                           a.counters, mt, mk);
    if (a.ev_end) (void)hipEventRecord(a.ev_end, st);
    if (a.n & (AGH_STRIP - 1))
        hipLaunchKernelGGL((k_sweep_multi_tail<MODE, STRIDE, Q5>), dim3(1), dim3(64), 0, st,
                           (const uint4 *)a.text, a.n, a.q, (const uint32_t *)a.ftab,
                           a.wave_totals, a.cand, a.wave_cand, a.counters,
                           (inl && !a.lean) ? (const uint32_t *)a.strip_prefix
                                            : (const uint32_t *)nullptr);
}

// a.q.fh = the probe stride chosen by the host (fill_multi_tables): 1, 2 or 4; strides > 1 imply q == 4
template <int MODE>
static void launch_sweep_multi_m(const agh_sweep_args &a, const agh_multi_dev &m,
                                 const agh_marks &mk, bool inl, hipStream_t st)
{
    if ((MODE & 2) && a.q.fh == 4 && a.q.mp_q5) launch_sweep_multi_ms<MODE, 4, true>(a, m, mk, inl, st);
    else if ((MODE & 2) && a.q.fh == 4) launch_sweep_multi_ms<MODE, 4, false>(a, m, mk, inl, st);
    else if ((MODE & 2) && a.q.fh == 2) launch_sweep_multi_ms<MODE, 2, false>(a, m, mk, inl, st);
    else launch_sweep_multi_ms<MODE, 1, false>(a, m, mk, inl, st);
}

// Multi-pattern sweep; a.ftab = the 2^18-bit prefix table.  The prefix scan of the census
// (numbered scans) is launched by the caller through agh_launch_census_scan().
// inl: check every hit on the spot (dense hit sets); the tail strip always goes through slices.
void agh_launch_sweep_multi(const agh_sweep_args &a, const agh_multi_dev &m, const agh_marks &mk,
                            bool inl, hipStream_t st)
{
    const int mode = (a.q.fold ? 1 : 0) | (a.q.fq == 4 ? 2 : 0) | (a.lean ? 4 : 0);
    switch (mode) {
    case 0: launch_sweep_multi_m<0>(a, m, mk, inl, st); break;
    case 1: launch_sweep_multi_m<1>(a, m, mk, inl, st); break;
    case 2: launch_sweep_multi_m<2>(a, m, mk, inl, st); break;
    case 3: launch_sweep_multi_m<3>(a, m, mk, inl, st); break;
    case 4: launch_sweep_multi_m<4>(a, m, mk, inl, st); break;
    case 5: launch_sweep_multi_m<5>(a, m, mk, inl, st); break;
    case 6: launch_sweep_multi_m<6>(a, m, mk, inl, st); break;
    default: launch_sweep_multi_m<7>(a, m, mk, inl, st); break;
    }
}

void agh_launch_verify_multi(const agh_scan_args &a, const agh_multi_dev &m, bool lean,
                             hipStream_t st)
{
    agh_multi_tables mt;
    mt.bits = m.bits;
    mt.bucket_start = m.bucket_start;
    mt.bucket_items = m.bucket_items;
    mt.pat_off = m.pat_off;
    mt.pool = m.pool;
    mt.piece_owner = m.piece_owner;
    mt.piece_po = m.piece_po;
    mt.owner_len = m.owner_len;
    mt.owner_mask = m.owner_mask;
    mt.item_info = m.item_info;
    const uint32_t blocks = (a.nw + 3u) / 4u;
    if (!blocks) return;
#define AGH_VM(LEANV, KK)                                                                     \
    hipLaunchKernelGGL((k_verify_multi<LEANV, KK>), dim3(blocks), dim3(256), 0, st,           \
                       (const uint8_t *)a.text, a.n, a.q, mt, a.cand, a.wave_cand,            \
                       a.wave_prefix, a.nw, a.mk)
#define AGH_VM_CASE(KK) case KK: if (lean) AGH_VM(true, KK); else AGH_VM(false, KK); break;
    switch (a.q.k) {
        AGH_VM_CASE(0) AGH_VM_CASE(1) AGH_VM_CASE(2) AGH_VM_CASE(3) AGH_VM_CASE(4)
        AGH_VM_CASE(5) AGH_VM_CASE(6) AGH_VM_CASE(7) AGH_VM_CASE(8)
    default: break;
    }
#undef AGH_VM_CASE
#undef AGH_VM
}
