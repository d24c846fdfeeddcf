// agh_device.h -- structures and constants shared by the host side of libagrep_hip.so and
// its gfx950 kernels.  Internal; the public boundary is include/agrep_hip.h.
#pragma once
#include <stdint.h>

// A "strip" is 1 KiB of text = one coalesced wave-wide 16 B/lane load (64 lanes x 16 B).
// The sweep kernel publishes, per strip, the number of record delimiters that precede it
// inside its wave's range; that is what turns a match position into a record number.
#define AGH_STRIP_SHIFT 10
#define AGH_STRIP (1u << AGH_STRIP_SHIFT)
// Contiguous strips owned by one wavefront of the sweep kernel (256 KiB of text).
#define AGH_WAVE_STRIPS 256u
// Candidate slots owned by one sweep wave (one per 64 text bytes; more means the filter is
// not selective on this text and the scan falls back to the full automaton).
#define AGH_SLICE_CAP (AGH_WAVE_STRIPS * 16u)
// Candidate entry of a numbered single-pattern sweep: (delimiters in front of the sample's 16-byte chunk inside its
// wave's range) << 40 | index of the sample (dwords; halfwords for H == 2 samples).  The count is below 2^19 (a
// range is 256 KiB), the index below 2^34 in a 16 GiB segment.  (Round 4 packed 32 + 32 bits: H == 2 queries were
// limited to 8 GiB per segment, every other one just below 16.)  Lean entries are the bare 64-bit index.
#define AGH_CAND_IDX_BITS 40
#define AGH_CAND_IDX_MASK ((1ull << AGH_CAND_IDX_BITS) - 1ull)
// multi-pattern scans probe every byte position: room for one hit per 16 bytes
#define AGH_MP_SLICE_CAP (AGH_WAVE_STRIPS * 64u)
// Lean scans: how far back the verifier looks for the start of a matched record before the
// scan falls back to the numbered (census) mode.
#define AGH_LEAN_BACK_CAP (1024u * 1024u)
// q-gram filter table: one byte per hash bucket, resident in LDS (32 KiB / workgroup).
#ifndef AGH_FT_BITS
#define AGH_FT_BITS 15         // filter table: 2^15 bytes = 2^18 bits (make FT_BITS=14 builds a 16 KiB variant for A/B)
#endif
#define AGH_FT_SIZE (1u << AGH_FT_BITS)
// Full-scan kernel: bytes per lane chunk (= one census strip), bytes per lane per refill of the
// per-wave LDS ring, the ring's row stride (+16 B keeps the b128 reads conflict-free), lanes per
// workgroup.
#define AGH_FS_CHUNK 1024u
#define AGH_FS_ROUND 64u
#define AGH_FS_ROW (AGH_FS_ROUND + 16u)
#define AGH_FS_THREADS 256u

enum agh_counter {
    AGH_C_CAND = 0,      // candidate windows emitted by the filter
    AGH_C_OVERFLOW = 1,  // candidate / match buffer overflow flag
    AGH_C_MATCHED = 2,   // distinct matched records
    AGH_C_NDELIM = 3,    // delimiters in the text
    AGH_C_STORED = 4,    // records the ordered compaction of the record bitmap listed (k_bm_offsets)
    AGH_C_LASTBYTE = 5,  // text[n-1]
    AGH_C_CHECK = 6,     // read-probe checksum sink
    AGH_C_BM_OVERFLOW = 7, // a record number did not fit the record bitmap
    AGH_C_LEAN_FALLBACK = 8,
    AGH_C_DELIM_CHAIN = 9, // multi-byte delimiter: overlapping occurrences chain > 4 KiB (unsupported) // lean scan gave up (record start too far back / hash set full)
    AGH_C_ANYHIT = 10,   // lean scans: some record matched (what -l needs to stop early)
    AGH_C_GIVEUPS = 11,  // lean scans: matches whose record starts further back than the verifier looks (agh_marks.giveups)
    AGH_C_RECBYTES_LO = 12, // record output: bytes of the listed records (one 64-bit sum: k_match_bounds)
    AGH_C_RECBYTES_HI = 13,
    AGH_C_NREC = 14,     // -v record lists: records k_unmatched walked over (the compaction inverts below this)
    AGH_C_CONT_N = 15,   // table engine, fast form: open records handed to k_table_cont
    AGH_C_CONT_NEXT = 16, // ... and the ticket its lanes take them with
    AGH_C_COUNT = 20
};

// The reference's own query tables for the table engine (agh_table.hip), maskgen.c layout.
#define AGH_GT_AMBIGUOUS ((uint64_t)1 << 63)   // gram table entry: no offset information
#define AGH_MAX_ERRORS_DEV 8   // = AGH_MAX_ERRORS of the C-ABI (agrep.h:44 MaxError)

struct agh_dev_tables {
    uint32_t Init0, Init1, NO_ERR, endposition, D_endpos, D_Mask, AND;
};

struct agh_dev_query {
    int32_t m;          // pattern positions
    int32_t k;          // errors
    uint32_t delim;     // the delimiter byte (last byte of a multi-byte delimiter)
    uint32_t dlen;      // delimiter length in bytes; > 1: delimiter ends come from the bitmap
    uint8_t dbytes[8];  // the delimiter
    uint32_t dfold;     // 1: delimiter bytes match case-insensitively (-i with letters in the delimiter)
    uint32_t mb;        // 1: delimiter ends come from the delimiter bitmap (dlen > 1, or a folded letter)
    uint32_t mp_q5;     // multi-pattern sweep at stride 4: the probed grams have 5 bytes (entries >= 8 bytes)
    uint32_t guard;     // -f: 1 = -w (non-alphanumeric bytes around an occurrence), 2 = -x (whole line)
    int32_t fq;         // filter: sample length in bytes (1..4), 0 = no filter
    int32_t fh;         // filter: sample stride in bytes (4, 8 or 16)
    uint32_t qmask;     // low fq bytes
    uint32_t fold;      // 0x20 in every sampled byte when the query folds ASCII case
    uint32_t head_byte; // byte fed in front of the segment: '\n' for the first segment of a
                        // file (asearch.c:69-78), the delimiter for later segments
    int32_t tail_virtual; // 1: the delimiter is appended at the segment end (asearch.c:87-91)
    // general automaton (asearch1.c costs, <exact> segments): used by the full-scan kernel only
    uint32_t ci, cs, cd;  // cost of insertion / substitution / deletion (1..k+1)
    uint64_t no_err;      // bit p-1 clear: position p may not be entered through an error
};

// Multi-pattern (-f) scans: 2^18-bit table of pattern-prefix q-grams (32 KiB, LDS) and the
// bucket directory the verifier walks.
#define AGH_MP_BITS 18
#define AGH_MP_BUCKET_BITS 14

// Hash of one text/pattern sample (already masked and folded) into the filter table.
// Identical on host (table construction) and device (probe).  v_mul_u32_u24 is full rate.
#if defined(__HIPCC__)
#define AGH_HD __host__ __device__ __forceinline__
#else
#define AGH_HD static inline
#endif
// The filter table holds 2^18 BITS in 2^15 bytes: the byte a sample selects (its 15-bit slot, which
// is also its slot in the gram table) and the bit inside it come from one 32-bit product.  Eight
// times fewer chance hits than one flag per byte at the same LDS size, for two more VALU operations
// per probe (v_bfe for the bit number, v_bfe to take the bit).
// q == 4: the sample has 32 significant bits.  One v_dot2_u32_u16: low16 * A + high16 * B; the LOW
// 18 bits of that sum are what selects the bit -- for ASCII text the information sits in the low bits
// of every byte and a multiplication only carries it upward, so the low bits of the sum see all four
// bytes (measured on the bench alphabet and on random lower-case text: chance-hit rate within 5 % of
// an ideal hash; the top bits of the same sum are 7 x worse).  Bit index = prod & 0x3ffff, i.e. byte
// slot = (prod >> 3) & (size - 1), bit = prod & 7 -- the dword holding it is at byte address
// (prod >> 3) & (size - 4) and the bit inside that dword is prod & 31, which v_lshrrev takes from the
// register as it is.
#define AGH_DOT_A 0x9E37u
#define AGH_DOT_B 0x79B9u
AGH_HD uint32_t agh_sample_prod_q4(uint32_t s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short agh_u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(agh_u16x2, s),
                                  __builtin_bit_cast(agh_u16x2, (uint32_t)(AGH_DOT_A | (AGH_DOT_B << 16))), 0u, false);
#else
    return (s & 0xffffu) * AGH_DOT_A + (s >> 16) * AGH_DOT_B;
#endif
}
#define AGH_Q4_SLOT(p) (((p) >> 3) & (AGH_FT_SIZE - 1u))
#define AGH_Q4_BIT(p) ((p) & 7u)
AGH_HD uint32_t agh_sample_hash_q4(uint32_t s)
{
    return AGH_Q4_SLOT(agh_sample_prod_q4(s));
}
// 18-bit variant for the multi-pattern bit table, probed at EVERY text position (16 probes per
// 16 bytes), so every instruction counts: one shift + one v_mad_u32_u24 (the 24-bit multiply
// ignores the top byte by itself; the shifted copy folds it back in through the addend).
AGH_HD uint32_t agh_sample_prod18_q4(uint32_t s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(s, 0x9E3779u) + (s >> 13);
#else
    return (s & 0xffffffu) * 0x9E3779u + (s >> 13);
#endif
}
AGH_HD uint32_t agh_sample_hash18_q4(uint32_t s)
{
    return (agh_sample_prod18_q4(s) >> 14) & ((1u << AGH_MP_BITS) - 1u);
}
// 5-byte grams (strided multi-pattern sweep over entries of >= 8 bytes): the fifth byte is spread
// over the dword (v_perm_b32) and XORed in; both Bloom hashes then take the mix.
AGH_HD uint32_t agh_mix5(uint32_t g4, uint32_t next_dword)
{
    return g4 ^ ((next_dword & 0xffu) * 0x01010101u);
}
// second, independent 18-bit hash of a 4-byte prefix: the multi-pattern bit table is a Bloom
// filter with two probes when q == 4 (the second probe runs only on first-level hits).  The same
// one-instruction form as the first hash with another pair of multipliers (a full 32-bit multiply
// is quarter rate); low 18 bits of the sum, chance-hit rate of the pair within 10 % of two ideal hashes.
AGH_HD uint32_t agh_sample_hash18b_q4(uint32_t s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short agh_u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(agh_u16x2, s),
                                  __builtin_bit_cast(agh_u16x2, (uint32_t)(0xC2B3u | (0x5BD1u << 16))), 0u, false) &
           ((1u << AGH_MP_BITS) - 1u);
#else
    return ((s & 0xffffu) * 0xC2B3u + (s >> 16) * 0x5BD1u) & ((1u << AGH_MP_BITS) - 1u);
#endif
}
AGH_HD uint32_t agh_sample_hash18_q3(uint32_t s)
{
    uint32_t p = (s & 0xffffffu) * 0x85EBCAu;
    return (p >> 13) & ((1u << AGH_MP_BITS) - 1u);
}
// bucket of a pattern prefix in the verifier's directory
AGH_HD uint32_t agh_mp_bucket(uint32_t s)
{
    uint32_t x = s * 0x9E3779B1u;
    return (x ^ (x >> 15)) >> (32 - AGH_MP_BUCKET_BITS);
}
// ---- one-pass -f scan (agh_mscan.hip): pair table + exact gram table --------------------------
// Two neighbouring text positions p, p+1 hold the 4-grams (a, xyz) and (xyz, d) around the SAME
// 3-gram xyz = t[p+1..p+3], so ONE 8-byte row -- selected by a hash of xyz -- answers both:
//   row.x bit (a & 31): some key gram is a followed by xyz   ("pre" role of xyz)
//   row.y bit (d & 31): some key gram is xyz followed by d   ("suf" role of xyz)
// Every key gram (k0 k1 k2 k3) is entered twice: pre of (k1 k2 k3) with a = k0, suf of (k0 k1 k2)
// with d = k3.  Half the LDS reads of one probe per position, and ds_read_b64 spreads over 64 banks.
// Row of a 3-gram held in the low 24 bits of y (the top byte is ignored by the 24-bit multiply):
// the middle bits [31:19] of y * C; the device takes them as a byte address with
// ((v_mul_u32_u24 y, C) >> 16) & ((rows - 1) << 3)  (one v_and_b32_sdwa src0_sel:WORD_1).
#define AGH_MS_C 0xC2B2AEu
#define AGH_MS_RB_MAX 13u                   // rows <= 2^13 (64 KiB)
AGH_HD uint32_t agh_ms_row(uint32_t y, uint32_t rb)
{
    const uint32_t h = (uint32_t)((uint64_t)(y & 0xffffffu) * AGH_MS_C);
    return (h >> 19) & ((1u << rb) - 1u);
}
// exact gram table: AGH_MS_GSLOTS 32-bit grams (0 = empty) in buckets of four, two-choice hashing: a
// gram lives in one of the two buckets its hash names (the host puts it into the emptier one), a lookup
// reads both
#define AGH_MS_GSLOTS 4096u
#define AGH_MS_GBUCKETS (AGH_MS_GSLOTS / 4u)
AGH_HD uint32_t agh_ms_ghash(uint32_t g)
{
    return g * 0x9E3779B1u;
}
#define AGH_MS_GB1(h) ((h) >> 22)                    // 10 bits each
#define AGH_MS_GB2(h) (((h) >> 12) & (AGH_MS_GBUCKETS - 1u))
// ---- dense -f sets with one error (agh_mtile.hip): pieces of 2..7 bytes --------------------------------------------
#define AGH_MW_DIR 4096u                    // slots of the mask table (by the pair) and of the entry directory
#define AGH_MW_G4_WORDS 2048u               // 2^16 bits: the first four bytes of the pieces of >= 4 bytes (word: bits 2..12 of
                                            // agh_sample_prod_q4, bit: bits 13..17)
#define AGH_MW_MAX_ENT 3072u                // entries (16 bytes each) next to the directory, the masks and the bit table: 152 KiB of LDS, one
                                            // workgroup per CU
// the mask table's slot: the first two bytes of a piece
AGH_HD uint32_t agh_mw_slot(uint32_t bigram)
{
    return ((bigram & 0xffffu) * 40503u >> 4) & (AGH_MW_DIR - 1u);
}
// the entry directory's slot of a piece of >= 3 bytes: its first three bytes (pieces of two bytes: agh_mw_slot of the
// pair).  A text position looks into both slots; with the pair alone a slot of the 4..12-byte set held 2.8 entries on
// average and a wave waited for the longest of its 64 lists in every round (87 lane-instructions per byte,
// profiles/r06_pmc_mtile_v1.json)
AGH_HD uint32_t agh_mw_slot3(uint32_t trigram)
{
    return ((trigram & 0xffffffu) * 0x9E3779B1u >> 20) & (AGH_MW_DIR - 1u);
}
// q <= 3: the sample already fits 24 bits.
AGH_HD uint32_t agh_sample_prod_q3(uint32_t s)
{
    return (s & 0xffffffu) * 0x85EBCAu;
}
#define AGH_Q3_SLOT(p) (((p) >> 13) & (AGH_FT_SIZE - 1u))
#define AGH_Q3_BIT(p) (((p) >> 10) & 7u)
AGH_HD uint32_t agh_sample_hash_q3(uint32_t s)
{
    return AGH_Q3_SLOT(agh_sample_prod_q3(s));
}
