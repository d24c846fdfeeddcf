/*
 * corpus_gen.h -- TEST / BENCH INFRASTRUCTURE: CPU side of the synthetic corpus generator.
 *
 * Spec (restated in DESIGN.md "Synthetic corpus"): the corpus is a sequence
 * of independent 4096-byte pages; page b is a pure function of (seed, b), so CPU and GPU
 * produce identical bytes for any shard without storing the whole corpus.  Each page is a
 * run of newline-terminated records of 40..120 bytes (the last record of a page absorbs the
 * remainder, <= 161 bytes) drawn from a 41-symbol English-skewed lowercase multiset; one
 * record in `plant_period` carries one of the caller-supplied pattern variants at record
 * offset 5.  Quirk-free by construction: <= 1 planted occurrence per
 * record (Q4), never two planted records in a row (Q2), pages end with '\n' so no record
 * straddles byte 49152 (Q1), records <= 1024 B (Q10), trailing newline (Q7).
 *
 * The device twin is agh_corpus_fill_device() in agrep_amd/csrc/agh_corpus.hip; tests assert
 * both produce the same bytes.
 */
#ifndef CORPUS_GEN_H
#define CORPUS_GEN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CG_PAGE 4096
#define CG_MAX_VARIANTS 8
#define CG_MAX_VLEN 80

typedef struct {
    uint64_t seed;
    uint32_t n_variants;                 /* 0 = nothing planted */
    uint32_t plant_period;               /* 1 record in plant_period is planted (500) */
    uint32_t upper_permille;             /* letters upper-cased with this probability (0 or 500) */
    uint32_t vlen[CG_MAX_VARIANTS];
    uint8_t  variants[CG_MAX_VARIANTS][CG_MAX_VLEN];
} cg_params;

/* Fill out[0 .. n_pages*4096) with pages first_page .. first_page+n_pages-1.
 * planted[v] (may be NULL) is incremented once per planted record of variant v. */
void cg_fill(const cg_params *p, uint64_t first_page, uint64_t n_pages, uint8_t *out,
             uint64_t *planted);

#ifdef __cplusplus
}
#endif
#endif
