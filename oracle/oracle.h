/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's Aho-Corasick and Wu-Manber hot path
 * (reference files ac/ac.c, wu/wu.c, caller conventions in main.c).  Nothing
 * under oracle/ is part of the shipped engine: only tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py may load it, and only
 * as the checker.  The product library (libsmatcher_hip.so) never links or
 * calls into this directory.
 *
 * Parity pin: every function here is checked against the reference's own
 * compiled ac/ac.c + wu/wu.c (oracle/_ref/libref.so, built by oracle/Makefile
 * from the sources where they lie under /root/reference) and against the
 * golden counts / table digests committed under tests/golden/.
 *
 * Extensions over the reference signatures (documented, not semantic):
 *   - text length is int64_t and counts are uint64_t (reference: int n,
 *     unsigned matches -- smatcher.h:90,105), so that the 1 GiB / 32 GiB
 *     configurations can be checked without wrap-around;
 *   - the Wu-Manber globals m_nBitsInShift / shiftsize (smatcher.h:71,73) are
 *     passed explicitly so the oracle is re-entrant.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic corpus (SURVEY.md 8c generator: splitmix64, symbol = z % sigma) ---- */
uint64_t ora_splitmix64_at(uint64_t seed, uint64_t index);
void ora_gen_text(uint8_t *out, uint64_t n, uint64_t offset, uint64_t seed, int sigma);
/* uniform random patterns, pattern-major, flat p*m bytes */
void ora_gen_patterns_uniform(uint8_t *out, int m, int p, uint64_t seed, int sigma);
/* every `from_text_every`-th pattern (j % from_text_every == 0) is a substring of
 * the synthetic text (text_seed, n_text) at a splitmix-chosen offset, the rest
 * uniform; from_text_every <= 0 means all uniform. */
void ora_gen_patterns_mixed(uint8_t *out, int m, int p, uint64_t seed, int sigma,
                            uint64_t text_seed, uint64_t n_text, int from_text_every);

/* ---- Aho-Corasick (ac/ac.c) ---- */
/* Caller-owned flat tables sized (m*p+1)[*sigma], pre-initialised as main.c:410-420 does:
 * state_transition <- -1, state_supply <- 0, state_final <- 0. */
typedef struct ora_ac_table ora_ac_table;
ora_ac_table *ora_preproc_ac(const uint8_t *const *pattern, int m, int p_size, int alphabet,
                             int32_t *state_transition, uint32_t *state_supply,
                             uint32_t *state_final);
uint32_t ora_ac_idcounter(const ora_ac_table *t);
uint32_t ora_ac_patterncounter(const ora_ac_table *t);
uint64_t ora_search_ac(const uint8_t *text, int64_t n, const ora_ac_table *t);
/* same count from the flat goto / supply / final tables alone (what cuda_ac.cu:563-592 walks) */
uint64_t ora_search_ac_tables(const uint8_t *text, int64_t n, int alphabet,
                              const int32_t *state_transition, const uint32_t *state_supply,
                              const uint32_t *state_final);
void ora_free_ac(ora_ac_table *t);

/* ---- Set-Horspool (sh/sh.c) ---- */
/* tables caller-owned and pre-initialised as for AC: state_transition <- -1, state_final <- 0 */
void ora_preproc_sh(const uint8_t *const *pattern, int m, int p_size, int alphabet,
                    int32_t *state_transition, uint32_t *state_final,
                    uint32_t *idcounter_out, uint32_t *patterncounter_out);
/* the bad-character table main.c:173 gets from the (missing) helper's preBmBc */
void ora_pre_bmbc(const uint8_t *const *pattern, int m, int p_size, int alphabet, int32_t *bmBc);
uint64_t ora_search_sh(int m, const uint8_t *text, int64_t n, int alphabet,
                       const int32_t *state_transition, const uint32_t *state_final, const int32_t *bmBc);

/* ---- Set Backward Oracle Matching (sbom/sbom.c) ---- */
/* state_transition <- -1 as for AC; state_final_multi: (m*p+1) rows of 200 entries, zeroed (main.c:422-425) */
void ora_preproc_sbom(const uint8_t *const *pattern, int m, int p_size, int alphabet,
                      int32_t *state_transition, uint32_t *state_final_multi,
                      uint32_t *idcounter_out, uint32_t *patterncounter_out);
uint64_t ora_search_sbom(const uint8_t *pattern_flat, int m, const uint8_t *text, int64_t n, int alphabet,
                         const int32_t *state_transition, const uint32_t *state_final_multi);

/* ---- SOG, 8-byte patterns (sog/sog8.c) ---- */
/* T8: 2^24 bytes, scanner_hs / scanner_index: p_size entries, scanner_hs2: 8192 bytes (main.c:495-515).  The
 * 2-level bitmap is set from the pattern's real hash (the reference reads an uninitialised variable there,
 * sog/sog8.c:124,135); everything else as the reference. */
void ora_preproc_sog8(uint8_t *T8, uint32_t *scanner_hs, int32_t *scanner_index, uint8_t *scanner_hs2,
                      const uint8_t *const *pattern, int p_size);
uint64_t ora_search_sog8(const uint8_t *T8, const uint32_t *scanner_hs, const int32_t *scanner_index,
                         const uint8_t *scanner_hs2, const uint8_t *const *pattern, const uint8_t *text, int64_t n,
                         int p_size);

/* ---- Wu-Manber (wu/wu.c) ---- */
/* returns the table length for an alphabet, 0 if unsupported (reference calls fail()) */
uint32_t ora_wu_determine_shiftsize(int alphabet);
/* caller pre-fills SHIFT[i] = m - B + 1 and PREFIX_size[i] = 0 (main.c:444-449) */
void ora_preproc_wu(const uint8_t *const *pattern, int m, int p_size, int alphabet, int B,
                    int nbits, int32_t *SHIFT, int32_t *PREFIX_value, int32_t *PREFIX_index,
                    int32_t *PREFIX_size);
void ora_preproc_wu2(const uint8_t *pattern_flat, int m, int p_size, int alphabet, int B,
                     int nbits, int32_t *SHIFT, int32_t *PREFIX_value, int32_t *PREFIX_index,
                     int32_t *PREFIX_size);
uint64_t ora_search_wu(const uint8_t *const *pattern, int m, int p_size, const uint8_t *text,
                       int64_t n, int nbits, const int32_t *SHIFT, const int32_t *PREFIX_value,
                       const int32_t *PREFIX_index, const int32_t *PREFIX_size);
uint64_t ora_search_wu2(const uint8_t *pattern_flat, int m, int p_size, const uint8_t *text,
                        int64_t n, int nbits, const int32_t *SHIFT, const int32_t *PREFIX_value,
                        const int32_t *PREFIX_index, const int32_t *PREFIX_size);

/* compressed-row form of the same tables (the dense PREFIX arrays are 2 x 2.1 GB at alphabet 256 /
 * 100 000 patterns): rows in wu/wu.c's append order; SHIFT pre-filled with m - B + 1; bucket_off has
 * shiftsize + 1 entries, bucket_val / bucket_idx p_size entries */
void ora_preproc_wu_csr(const uint8_t *pattern_flat, int m, int p_size, int B, int nbits, uint32_t shiftsize,
                        int32_t *SHIFT, uint32_t *bucket_off, int32_t *bucket_val, int32_t *bucket_idx);
uint64_t ora_search_wu_csr(const uint8_t *pattern_flat, int m, const uint8_t *text, int64_t n, int nbits,
                           const int32_t *SHIFT, const uint32_t *bucket_off, const int32_t *bucket_val,
                           const int32_t *bucket_idx);

/* ---- byte-range sharding (main.c:375-378, 464-477) ---- */
/* shard i of R over a text of n bytes: [begin, end) with the m-1 halo, true length (no padding) */
void ora_shard_range(int64_t n, int R, int i, int m, int64_t *begin, int64_t *end);

/* ---- definition-level checker: |{e : text[e-m+1..e] in set(patterns)}| by brute force ---- */
uint64_t ora_count_bruteforce(const uint8_t *pattern_flat, int m, int p_size,
                              const uint8_t *text, int64_t n);
/* sorted list of match END columns (reference only has commented printf's for these:
 * ac/ac.c:217, wu/wu.c:93); writes at most cap entries, returns the total found */
uint64_t ora_positions_bruteforce(const uint8_t *pattern_flat, int m, int p_size,
                                  const uint8_t *text, int64_t n, int64_t *out, uint64_t cap);

/* FNV-1a 64 digest of a buffer -- used to pin big tables in small golden files */
uint64_t ora_fnv1a64(const void *buf, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
