/*
 * csrc/corpus_gen.h -- the non-uniform synthetic corpora, one definition for the host (corpus.c) and the
 * device (corpus_text_kind_kernel in wm_kernels.inc).
 *
 * The reference's own data sets are E.coli, A.thaliana, swiss-prot and world192 (main.c:39-109; the files are not
 * in the upstream repository): repeats, low-complexity runs, skewed symbol frequencies.  The i.i.d. uniform text
 * of smh_corpus_text_* is the BEST case of every filter engine of this library, so the engine choice has to be
 * exercised on text that is not (round 4).  Three more kinds, all regenerable slice by slice on either side:
 *
 *   SMH_CORPUS_DNA_REPEATS  alphabet 4.  1 KiB blocks: 70 % order-3 Markov text (one of four fixed probabilities
 *                           0.50 / 0.25 / 0.15 / 0.10 per symbol and context: 1.74 bits per symbol), 15 % copies of one
 *                           of 64 library blocks (the "repeated, pattern-bearing segments": a pattern sampled from
 *                           one recurs wherever that block does), 5 % tandem repeats of a 2..31-symbol unit, 10 %
 *                           low-complexity runs (poly-A in three of four, 1 symbol in 32 is noise)
 *   SMH_CORPUS_SKEWED       any alphabet.  i.i.d. symbols of a Zipf-like distribution -- weight 1 / (k + 4) for
 *                           alphabets up to 32 (20 symbols: 13 % .. 2.3 %, a protein's spread), 1 / (k + 4)^2 above
 *                           (256 symbols: 22 %, 14 %, 10 %, ...: 4.3 bits per symbol, natural-language text) -- with
 *                           15 % library blocks and 5 % low-complexity runs
 *   SMH_CORPUS_PLANTED      any alphabet.  Uniform text in which ONE 32-symbol word recurs in every 64-byte cell (at
 *                           a cell-dependent offset 0..32): a pattern set that holds the word's first m symbols
 *                           matches every <= 64 columns
 *
 * Random numbers: 16 bits per symbol, four symbols per splitmix64 value, indexed by the symbol's position in the
 * stream its block shows (a library block shows another stream), so a block is a function of (seed, block index)
 * alone.  Inside a block the generator is sequential (Markov context, tandem unit): a slice is produced by
 * generating the blocks it touches from their first symbol.
 */
#ifndef SMH_CORPUS_GEN_H
#define SMH_CORPUS_GEN_H

#include <stdint.h>

#if defined(__HIPCC__)
#define SMH_CG __host__ __device__ static inline
#else
#define SMH_CG static inline
#endif

#define SMH_CORPUS_BLOCK 1024u
#define SMH_CORPUS_CELL 64u
#define SMH_CORPUS_WORD 32u
#define SMH_CORPUS_LIB 64u

struct smh_corpus_tabs {
    uint16_t markov[64][4]; /* DNA_REPEATS: cumulative 16-bit thresholds of symbols 0, 1, 2 after the context's three symbols ([3] unused) */
    uint8_t quant[1024];    /* SKEWED: the symbol at quantile i / 1024 */
    uint8_t word[SMH_CORPUS_WORD]; /* PLANTED: the word */
};

SMH_CG uint64_t smh_cg_mix(uint64_t seed, uint64_t index) /* == smh_splitmix64_at (corpus.c) */
{
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
SMH_CG uint32_t smh_cg_r16(uint64_t seed, uint64_t i) { return (uint32_t)(smh_cg_mix(seed, i >> 2) >> (16u * (uint32_t)(i & 3u))) & 0xFFFFu; }

/* what block b of the text shows */
struct smh_cg_block {
    uint64_t content; /* index of the block whose stream is shown (a library block shows one of SMH_CORPUS_LIB) */
    uint64_t seed;
    uint32_t type;    /* 0 plain, 1 tandem repeat of the first `unit` symbols, 2 low-complexity run of `run` */
    uint32_t unit;
    uint32_t run;
    uint32_t noise;   /* type 2: a symbol is noise when its second random number is below this */
};

SMH_CG void smh_cg_block_of(uint64_t seed, uint64_t b, int kind, const struct smh_corpus_tabs *T, struct smh_cg_block *B)
{
    const uint64_t h = smh_cg_mix(seed ^ 0xB10C5EEDULL, b);
    const uint32_t r = (uint32_t)(h % 100u);
    B->content = b; B->seed = seed; B->type = 0; B->unit = 0; B->run = 0; B->noise = 0;
    if (kind == 1) {
        if (r >= 70u && r < 85u) { B->content = (h >> 8) % SMH_CORPUS_LIB; B->seed = seed ^ 0x11B2A2FULL; }
        else if (r >= 85u && r < 90u) { B->type = 1; B->unit = 2u + (uint32_t)((h >> 16) % 30u); }
        else if (r >= 90u) { B->type = 2; B->run = ((h >> 24) & 3u) == 3u ? (uint32_t)((h >> 26) & 3u) : 0u; B->noise = 2048u; }
    } else if (kind == 2) {
        if (r >= 80u && r < 95u) { B->content = (h >> 8) % SMH_CORPUS_LIB; B->seed = seed ^ 0x11B2A2FULL; }
        else if (r >= 95u) { B->type = 2; B->run = T->quant[(h >> 24) & 1023u]; B->noise = 4096u; }
    }
}

/* symbol p of a block (p = 0, 1, 2, ... in order); ctx = the Markov context, unit = the tandem unit seen so far */
SMH_CG uint32_t smh_cg_next(const struct smh_corpus_tabs *T, const struct smh_cg_block *B, int kind, uint32_t alphabet, uint64_t b,
                            uint32_t p, uint32_t *ctx, uint8_t *unit /* [32] */)
{
    const uint64_t gi = B->content * SMH_CORPUS_BLOCK + p;
    uint32_t sym;
    if (kind == 1) {
        const uint32_t r = smh_cg_r16(B->seed, gi);
        const uint16_t *t = T->markov[*ctx & 63u];
        sym = (uint32_t)(r >= t[0]) + (uint32_t)(r >= t[1]) + (uint32_t)(r >= t[2]);
    } else if (kind == 2) {
        sym = T->quant[smh_cg_r16(B->seed, gi) & 1023u];
    } else { /* planted: uniform text, the word at the cell's offset */
        const uint64_t i = b * SMH_CORPUS_BLOCK + p, cell = i / SMH_CORPUS_CELL;
        const uint32_t in = (uint32_t)(i % SMH_CORPUS_CELL);
        const uint32_t off = (uint32_t)(smh_cg_mix(B->seed ^ 0x9CE11ULL, cell) % (SMH_CORPUS_CELL - SMH_CORPUS_WORD + 1u));
        sym = in >= off && in < off + SMH_CORPUS_WORD ? T->word[in - off] : (uint32_t)(smh_cg_mix(B->seed, i) % alphabet);
    }
    if (B->type == 1) {
        if (p < B->unit) unit[p] = (uint8_t)sym;
        else sym = unit[p % B->unit];
    } else if (B->type == 2) {
        if (smh_cg_r16(B->seed ^ 0x2015EULL, gi) >= B->noise) sym = B->run;
    }
    *ctx = ((*ctx << 2) | (sym & 3u)) & 63u;
    return sym;
}

#endif
