/*
 * include/smatcher.h -- drop-in host API of the MI355X multi-pattern matcher.
 *
 * This header replaces the reference's smatcher.h for the Aho-Corasick and
 * Wu-Manber path.  Every declaration below keeps the reference's name,
 * argument order, argument meaning, ownership and error behaviour so that the
 * reference driver (main.c) compiles and links against libsmatcher_hip.so
 * unchanged; the citation after each item is the reference line it replaces.
 *
 * What is different underneath: preproc_* build the same caller-owned tables
 * on the host (bit-identical contents), search_* and cuda_* run hand-written
 * gfx950 kernels on the current HIP device and return the same match count
 * the reference's CPU loops (ac/ac.c:198-222, wu/wu.c:49-107) return.
 * There is no CPU search fallback: without a usable GPU these entry points
 * print a message and exit(1), which is the reference's own error convention
 * (cuda/cuda.h:26-47, fail()).
 *
 * Of the reference's sibling algorithms, Set-Horspool and SBOM are declared further down with the reference's
 * shapes (smatcher.h:55-69,93-99), and so is SOG (smatcher.h:75-80,108-109); KMP and BM (smatcher.h:131-133) are
 * single-pattern helpers outside this library.
 *
 * 64-bit text lengths / counts, resident-text handles, streams and the
 * multi-GPU shard helpers live in smatcher_hip.h.
 */
#ifndef SMATCHER_H
#define SMATCHER_H

/* the system headers the reference header gives its includers (smatcher.h:20-29): main.c relies on them
 * (ceil, stat, ...) without including them itself */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>
#include <sys/types.h>
#include <sys/stat.h>
#include <inttypes.h>
#include <math.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: ../helper2.h (absent upstream) supplied these to every caller of smatcher.h */
#ifndef MIN
#define MIN(a, b) (((a) < (b)) ? (a) : (b))
#endif
#ifndef MAX
#define MAX(a, b) (((a) > (b)) ? (a) : (b))
#endif
void fail(const char *msg); /* print msg, exit(1) */

/* smatcher.h:41-47 -- trie node as the reference declares it.  Callers only ever
 * hold it through struct ac_table.zerostate; this library keeps one root node
 * there (id 0, fail = itself, every next[] edge a self loop). */
struct ac_state {
    unsigned int id;
    unsigned int keywordline;
    unsigned char *output;
    struct ac_state *fail;
    struct ac_state **next;
};

/* smatcher.h:49-53 -- idcounter = number of automaton states, patterncounter =
 * number of distinct patterns, as the reference leaves them after preproc_ac. */
struct ac_table {
    unsigned int idcounter;
    unsigned int patterncounter;
    struct ac_state *zerostate;
};

/* smatcher.h:71,73 -- the reference DEFINES these in the header (needs -fcommon);
 * here they are declared and defined once inside the library. */
extern unsigned short m_nBitsInShift; /* must be 2 (main.c:431); preset to 2 */
extern unsigned int shiftsize;        /* set by wu_determine_shiftsize */

/* ---- Aho-Corasick: smatcher.h:89-91, ac/ac.c:224-252 ----
 * pattern: p_size pointers to m symbols each (values < alphabet).
 * state_transition[(m*p_size+1)*alphabet] pre-filled with -1, state_supply and
 * state_final [(m*p_size+1)] pre-filled with 0 by the caller (main.c:410-420);
 * filled exactly as the reference fills them.  Returned table is released by free_ac. */
struct ac_table *preproc_ac(unsigned char **pattern, int m, int p_size, int alphabet,
                            int *state_transition, unsigned int *state_supply,
                            unsigned int *state_final);
/* number of text end positions whose AC state is accepting (ac/ac.c:198-222), computed on the GPU */
unsigned search_ac(unsigned char *text, int n, struct ac_table *table);
void free_ac(struct ac_table *table, int alphabet);

/* ---- Wu-Manber: smatcher.h:101-106, wu/wu.c ----
 * SHIFT[shiftsize] pre-filled with m-B+1, PREFIX_size[shiftsize] with 0,
 * PREFIX_value / PREFIX_index [shiftsize*p_size] (main.c:429-449); B must be 3. */
void wu_determine_shiftsize(int alphabet);
void preproc_wu(unsigned char **pattern, int m, int p_size, int alphabet, int B, int *SHIFT,
                int *PREFIX_value, int *PREFIX_index, int *PREFIX_size);
void preproc_wu2(unsigned char *pattern_flat, int m, int p_size, int alphabet, int B, int *SHIFT,
                 int *PREFIX_value, int *PREFIX_index, int *PREFIX_size);
unsigned int search_wu(unsigned char **pattern, int m, int p_size, unsigned char *text, int n,
                       int *SHIFT, int *PREFIX_value, int *PREFIX_index, int *PREFIX_size);
unsigned int search_wu2(unsigned char *pattern_flat, int m, int p_size, unsigned char *text, int n,
                        int *SHIFT, int *PREFIX_value, int *PREFIX_index, int *PREFIX_size);

/* ---- GPU entry points with the reference's shapes ----
 * cuda/cuda_ac.cu:594,691,788,885,983: print "Kernel K matches \t%i\t time \t%f\n".
 * cuda/cuda_wm.cu:183,438,652,854,1060: return the count, *gpuTime = kernel seconds.
 * K = 1,2 run the table-faithful kernels (goto/supply/final rows, dense-bucket
 * scan); K = 3,4,5 run the tuned kernels (LDS-resident DFA / block filter).
 * Unlike the reference there is no minimum text size and no dropped tail. */
void cuda_ac1(int m, unsigned char *text, int n, int p_size, int alphabet, int *state_transition,
              unsigned int *state_supply, unsigned int *state_final);
void cuda_ac2(int m, unsigned char *text, int n, int p_size, int alphabet, int *state_transition,
              unsigned int *state_supply, unsigned int *state_final);
void cuda_ac3(int m, unsigned char *text, int n, int p_size, int alphabet, int *state_transition,
              unsigned int *state_supply, unsigned int *state_final);
void cuda_ac4(int m, unsigned char *text, int n, int p_size, int alphabet, int *state_transition,
              unsigned int *state_supply, unsigned int *state_final);
void cuda_ac5(int m, unsigned char *text, int n, int p_size, int alphabet, int *state_transition,
              unsigned int *state_supply, unsigned int *state_final);
int cuda_wm1(unsigned char *pattern_flat, int m, unsigned char *text, int n, int p_size,
             int alphabet, int B, int *SHIFT, int *PREFIX_value, int *PREFIX_index,
             int *PREFIX_size, double *gpuTime);
int cuda_wm2(unsigned char *pattern_flat, int m, unsigned char *text, int n, int p_size,
             int alphabet, int B, int *SHIFT, int *PREFIX_value, int *PREFIX_index,
             int *PREFIX_size, double *gpuTime);
int cuda_wm3(unsigned char *pattern_flat, int m, unsigned char *text, int n, int p_size,
             int alphabet, int B, int *SHIFT, int *PREFIX_value, int *PREFIX_index,
             int *PREFIX_size, double *gpuTime);
int cuda_wm4(unsigned char *pattern_flat, int m, unsigned char *text, int n, int p_size,
             int alphabet, int B, int *SHIFT, int *PREFIX_value, int *PREFIX_index,
             int *PREFIX_size, double *gpuTime);
int cuda_wm5(unsigned char *pattern_flat, int m, unsigned char *text, int n, int p_size,
             int alphabet, int B, int *SHIFT, int *PREFIX_value, int *PREFIX_index,
             int *PREFIX_size, double *gpuTime);

/* ---- Set-Horspool (smatcher.h:93-95, sh/sh.c) ----
 * preproc_sh fills the caller's state_transition / state_final (pre-initialised like the AC tables:
 * -1 / 0, main.c:410-420) with the REVERSED trie of the patterns, numbered as sh/sh.c:82-149 does.
 * preBmBc is the bad-character table main.c:173 obtains from the reference's missing helper.
 * search_sh / cuda_sh1..5 (cuda/cuda_sh.cu:110,289,429,558,687) count on the GPU; cuda_shK print
 * "Kernel K matches \t%i\t time \t%f\n" (cuda/cuda_sh.cu:191).  K = 1,2 walk the reversed trie as
 * given with the bmBc skip loop, K = 3,4,5 and search_sh run the tuned kernels. */
void preBmBc(unsigned char **pattern, int m, int p_size, int alphabet, int *bmBc);
struct ac_table *preproc_sh(unsigned char **pattern, int m, int p_size, int alphabet,
                            int *state_transition, unsigned int *state_final);
unsigned search_sh(int m, unsigned char *text, int n, struct ac_table *table, int *bmBc);
void free_sh(struct ac_table *table, int alphabet);
void cuda_sh1(int m, unsigned char *text, int n, int p_size, int alphabet, int *state_transition,
              unsigned int *state_final, int *bmBc);
void cuda_sh2(int m, unsigned char *text, int n, int p_size, int alphabet, int *state_transition,
              unsigned int *state_final, int *bmBc);
void cuda_sh3(int m, unsigned char *text, int n, int p_size, int alphabet, int *state_transition,
              unsigned int *state_final, int *bmBc);
void cuda_sh4(int m, unsigned char *text, int n, int p_size, int alphabet, int *state_transition,
              unsigned int *state_final, int *bmBc);
void cuda_sh5(int m, unsigned char *text, int n, int p_size, int alphabet, int *state_transition,
              unsigned int *state_final, int *bmBc);

/* ---- Set Backward Oracle Matching (smatcher.h:55-69,97-99, sbom/sbom.c) ----
 * preproc_sbom fills the caller's state_transition (pre-initialised to -1) with the factor oracle of
 * the reversed patterns -- trie edges and external transitions -- and state_final_multi (zeroed,
 * (m*p_size+1) rows of 200 entries, main.c:422-425) with {count, pattern ids...} per state, as
 * sbom/sbom.c:52-126 does.  pointer_array is the reference's header global; its driver allocates and
 * frees it around preproc_sbom / free_sbom (main.c:208,229); this library does not use it.
 * cuda_sbomK print "Kernel K matches \t%i\t time \t%f\n" (cuda/cuda_sbom.cu:212); K = 1,2 walk the
 * oracle and the lists as given, K = 3,4,5 and search_sbom run the tuned kernels. */
struct sbom_state {
    unsigned int id;
    unsigned int *F;
    unsigned int num;
    struct sbom_state *fail;
    struct sbom_state **next;
};
struct sbom_table {
    unsigned int idcounter;
    unsigned int patterncounter;
    struct sbom_state *zerostate; /* NULL here: the pointer graph is not materialised */
};
extern struct sbom_state **pointer_array;
struct sbom_table *preproc_sbom(unsigned char **pattern, int m, int p_size, int alphabet,
                                int *state_transition, unsigned int *state_final_multi);
unsigned search_sbom(unsigned char **pattern, int m, unsigned char *text, int n, struct sbom_table *table);
void free_sbom(struct sbom_table *table, int m);
void cuda_sbom1(unsigned char *pattern, int m, unsigned char *text, int n, int p_size, int alphabet,
                int *state_transition, unsigned int *state_final_multi);
void cuda_sbom2(unsigned char *pattern, int m, unsigned char *text, int n, int p_size, int alphabet,
                int *state_transition, unsigned int *state_final_multi);
void cuda_sbom3(unsigned char *pattern, int m, unsigned char *text, int n, int p_size, int alphabet,
                int *state_transition, unsigned int *state_final_multi);
void cuda_sbom4(unsigned char *pattern, int m, unsigned char *text, int n, int p_size, int alphabet,
                int *state_transition, unsigned int *state_final_multi);
void cuda_sbom5(unsigned char *pattern, int m, unsigned char *text, int n, int p_size, int alphabet,
                int *state_transition, unsigned int *state_final_multi);

/* ------------------------------------------------------------------ SOG (shift-or with 3-grams, m = 8)
 * smatcher.h:75-80,108-109; sog/sog8.c; cuda/cuda_sog.cu:221-831.  Tables are caller-owned (main.c:495-515):
 * T8 SIZE_3GRAM_TABLE bytes, scanner_hs and scanner_index p_size entries, scanner_hs2 32 * 256 bytes.
 * preproc_sog8 fills T8, scanner_hs and scanner_index exactly as the reference does; scanner_hs2 gets DEFINED
 * contents -- the bit search_sog8 tests, from the pattern's real hash -- where the reference computes it from
 * an uninitialised variable (sog/sog8.c:124,135), which makes its own match count depend on stack contents.
 * search_sog8 / cuda_sog1..5 return / print the number of 8-byte windows of the text that equal a pattern. */
#define SIZE_3GRAM_TABLE 0x1000000
#define CHAR_WIDTH_3GRAM 8
#define GET3GRAM(address) ((((uint32_t)(address)[0])) + (((uint32_t)((address)[1])) << CHAR_WIDTH_3GRAM) + (((uint32_t)((address)[2])) << (CHAR_WIDTH_3GRAM << 1)))
void preproc_sog8(uint8_t *T8, uint32_t *scanner_hs, int *scanner_index, uint8_t *scanner_hs2, unsigned char **pattern,
                  int m, unsigned char *text, int n, int p_size, int B);
unsigned int search_sog8(uint8_t *T8, uint32_t *scanner_hs, int *scanner_index, uint8_t *scanner_hs2,
                         unsigned char **pattern, int m, unsigned char *text, int n, int p_size, int B);
/* cuda/cuda_sog.cu: `pattern` is the flat p_size x m array (main.c:457-459); prints "Kernel K matches \t%i\t time \t%f\n" */
void cuda_sog1(uint8_t *T8, uint32_t *scanner_hs, int *scanner_index, uint8_t *scanner_hs2, unsigned char *pattern,
               int m, unsigned char *text, int n, int p_size, int B);
void cuda_sog2(uint8_t *T8, uint32_t *scanner_hs, int *scanner_index, uint8_t *scanner_hs2, unsigned char *pattern,
               int m, unsigned char *text, int n, int p_size, int B);
void cuda_sog3(uint8_t *T8, uint32_t *scanner_hs, int *scanner_index, uint8_t *scanner_hs2, unsigned char *pattern,
               int m, unsigned char *text, int n, int p_size, int B);
void cuda_sog4(uint8_t *T8, uint32_t *scanner_hs, int *scanner_index, uint8_t *scanner_hs2, unsigned char *pattern,
               int m, unsigned char *text, int n, int p_size, int B);
void cuda_sog5(uint8_t *T8, uint32_t *scanner_hs, int *scanner_index, uint8_t *scanner_hs2, unsigned char *pattern,
               int m, unsigned char *text, int n, int p_size, int B);

#ifdef __cplusplus
}
#endif
#endif
