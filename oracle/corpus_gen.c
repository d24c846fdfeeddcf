/* corpus_gen.c -- TEST / BENCH INFRASTRUCTURE (see corpus_gen.h). */
#include "corpus_gen.h"

static const char CG_ALPHA[42] = "abcdefghijklmnopqrstuvwxyz      etaoinshr";

static inline uint64_t cg_next(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static void cg_page(const cg_params *p, uint64_t page, uint8_t *out, uint64_t *planted)
{
    uint64_t s = p->seed ^ (page * 0x9E3779B97F4A7C15ull) ^ 0xA5A5A5A55A5A5A5Aull;
    uint32_t pos = 0;
    int prev_planted = 1;
    (void)cg_next(&s);
    while (pos < CG_PAGE) {
        uint64_t r = cg_next(&s);
        uint32_t len = 40u + (uint32_t)((r & 0xffff) % 81u);
        uint32_t rem = CG_PAGE - pos;
        uint32_t draw = (uint32_t)((r >> 16) & 0xffffff);
        uint32_t v = p->n_variants ? (uint32_t)((r >> 40) % p->n_variants) : 0;
        uint32_t i;
        uint64_t x = 0, u = 0;
        int plant;
        if (rem < len + 1u + 41u) len = rem - 1u;
        plant = !prev_planted && p->n_variants && p->plant_period &&
                (draw % p->plant_period == 0) && len >= 5u + p->vlen[v] + 5u;
        for (i = 0; i < len; i++) {
            uint32_t c;
            if ((i & 7u) == 0) { x = cg_next(&s); u = cg_next(&s); }
            c = (uint8_t)CG_ALPHA[(((uint32_t)(x >> (8 * (i & 7u))) & 0xffu) * 41u) >> 8];
            if (plant && i >= 5u && i < 5u + p->vlen[v]) c = p->variants[v][i - 5u];
            if (p->upper_permille && c >= 'a' && c <= 'z' &&
                ((((uint32_t)(u >> (8 * (i & 7u))) & 0xffu) * 1000u) >> 8) < p->upper_permille)
                c -= 32u;
            out[pos + i] = (uint8_t)c;
        }
        out[pos + len] = '\n';
        pos += len + 1u;
        if (plant && planted) planted[v]++;
        prev_planted = plant;
    }
}

void cg_fill(const cg_params *p, uint64_t first_page, uint64_t n_pages, uint8_t *out,
             uint64_t *planted)
{
    uint64_t b;
    for (b = 0; b < n_pages; b++) cg_page(p, first_page + b, out + b * CG_PAGE, planted);
}
