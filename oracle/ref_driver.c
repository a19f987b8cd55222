/*
 * oracle/ref_driver.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin C-ABI around the reference's OWN compiled ac/ac.c and wu/wu.c
 * (linked from /root/reference by oracle/Makefile into oracle/_ref/libref.so).
 * It does what main.c does around the hot path and nothing more:
 *   - table allocation / initialisation conventions  main.c:410-420, 429-449
 *   - m_nBitsInShift = 2                             main.c:431
 *   - preproc then search, timed separately          main.c:125-157, 268-298
 * Pattern buffers are m+1 bytes, zero padded, because ac_addstring evaluates
 * string[m] after a fully existing path (ac/ac.c:136-143).
 */
#include "smatcher.h" /* the reference's header, found with -I/root/reference */
#include <time.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

unsigned int ref_shiftsize(int alphabet)
{
    wu_determine_shiftsize(alphabet);
    return shiftsize;
}

/* tables are caller-owned, sized as main.c:410-420; they are initialised here */
unsigned long long ref_run_ac(const unsigned char *pat_flat, int m, int p_size, int alphabet,
                              const unsigned char *text, int n, int *state_transition,
                              unsigned int *state_supply, unsigned int *state_final,
                              unsigned int *idcounter, unsigned int *patterncounter,
                              double *t_preproc, double *t_search)
{
    size_t rows = (size_t)m * p_size + 1;
    memset(state_transition, -1, rows * alphabet * sizeof(int));
    memset(state_supply, 0, rows * sizeof(unsigned int));
    memset(state_final, 0, rows * sizeof(unsigned int));

    unsigned char **pattern = (unsigned char **)malloc((size_t)p_size * sizeof(unsigned char *));
    for (int j = 0; j < p_size; j++) {
        pattern[j] = (unsigned char *)calloc((size_t)m + 1, 1);
        memcpy(pattern[j], pat_flat + (size_t)j * m, (size_t)m);
    }
    double t0 = now_s();
    struct ac_table *table = preproc_ac(pattern, m, p_size, alphabet, state_transition,
                                        state_supply, state_final);
    double t1 = now_s();
    unsigned int matches = text ? search_ac((unsigned char *)text, n, table) : 0;
    double t2 = now_s();
    if (idcounter) *idcounter = table->idcounter;
    if (patterncounter) *patterncounter = table->patterncounter;
    if (t_preproc) *t_preproc = t1 - t0;
    if (t_search) *t_search = t2 - t1;
    free_ac(table, alphabet);
    for (int j = 0; j < p_size; j++) free(pattern[j]);
    free(pattern);
    return matches;
}

/* which: 0 = preproc_wu/search_wu (char**), 1 = preproc_wu2/search_wu2 (flat) */
unsigned long long ref_run_wu(const unsigned char *pat_flat, int m, int p_size, int alphabet,
                              const unsigned char *text, int n, int *SHIFT, int *PREFIX_value,
                              int *PREFIX_index, int *PREFIX_size, int which,
                              double *t_preproc, double *t_search)
{
    const int B = 3; /* main.c:335 */
    wu_determine_shiftsize(alphabet);
    m_nBitsInShift = 2;
    for (unsigned int i = 0; i < shiftsize; i++) {
        SHIFT[i] = m - B + 1;
        PREFIX_size[i] = 0;
    }
    unsigned int matches = 0;
    double t0, t1, t2;
    if (which == 0) {
        unsigned char **pattern = (unsigned char **)malloc((size_t)p_size * sizeof(unsigned char *));
        for (int j = 0; j < p_size; j++) {
            pattern[j] = (unsigned char *)malloc((size_t)m);
            memcpy(pattern[j], pat_flat + (size_t)j * m, (size_t)m);
        }
        t0 = now_s();
        preproc_wu(pattern, m, p_size, alphabet, B, SHIFT, PREFIX_value, PREFIX_index, PREFIX_size);
        t1 = now_s();
        if (text)
            matches = search_wu(pattern, m, p_size, (unsigned char *)text, n, SHIFT, PREFIX_value,
                                PREFIX_index, PREFIX_size);
        t2 = now_s();
        for (int j = 0; j < p_size; j++) free(pattern[j]);
        free(pattern);
    } else {
        t0 = now_s();
        preproc_wu2((unsigned char *)pat_flat, m, p_size, alphabet, B, SHIFT, PREFIX_value,
                    PREFIX_index, PREFIX_size);
        t1 = now_s();
        if (text)
            matches = search_wu2((unsigned char *)pat_flat, m, p_size, (unsigned char *)text, n,
                                 SHIFT, PREFIX_value, PREFIX_index, PREFIX_size);
        t2 = now_s();
    }
    if (t_preproc) *t_preproc = t1 - t0;
    if (t_search) *t_search = t2 - t1;
    return matches;
}

/* Set-Horspool, main.c:158-196 (multish): preproc_sh then search_sh with the caller's bmBc (the
 * reference gets it from preBmBc in its missing helper; the tests pass the oracle's ora_pre_bmbc) */
unsigned long long ref_run_sh(const unsigned char *pat_flat, int m, int p_size, int alphabet,
                              const unsigned char *text, int n, int *state_transition,
                              unsigned int *state_final, int *bmBc,
                              unsigned int *idcounter, unsigned int *patterncounter,
                              double *t_preproc, double *t_search)
{
    size_t rows = (size_t)m * p_size + 1;
    memset(state_transition, -1, rows * alphabet * sizeof(int));
    memset(state_final, 0, rows * sizeof(unsigned int));
    unsigned char **pattern = (unsigned char **)malloc((size_t)p_size * sizeof(unsigned char *));
    for (int j = 0; j < p_size; j++) {
        pattern[j] = (unsigned char *)calloc((size_t)m + 1, 1);
        memcpy(pattern[j], pat_flat + (size_t)j * m, (size_t)m);
    }
    double t0 = now_s();
    struct ac_table *table = preproc_sh(pattern, m, p_size, alphabet, state_transition, state_final);
    double t1 = now_s();
    unsigned int matches = text ? search_sh(m, (unsigned char *)text, n, table, bmBc) : 0;
    double t2 = now_s();
    if (idcounter) *idcounter = table->idcounter;
    if (patterncounter) *patterncounter = table->patterncounter;
    if (t_preproc) *t_preproc = t1 - t0;
    if (t_search) *t_search = t2 - t1;
    free_sh(table, alphabet);
    for (int j = 0; j < p_size; j++) free(pattern[j]);
    free(pattern);
    return matches;
}

/* Set Backward Oracle Matching, main.c:197-231 (multisbom): pointer_array (a global of the reference's
 * header, smatcher.h:55) is allocated by the caller, main.c:208; state_final_multi holds up to 199
 * pattern ids per state (main.c:422-425), zeroed by the caller */
unsigned long long ref_run_sbom(const unsigned char *pat_flat, int m, int p_size, int alphabet,
                                const unsigned char *text, int n, int *state_transition,
                                unsigned int *state_final_multi, unsigned int *idcounter,
                                unsigned int *patterncounter, double *t_preproc, double *t_search)
{
    size_t rows = (size_t)m * p_size + 1;
    memset(state_transition, -1, rows * alphabet * sizeof(int));
    memset(state_final_multi, 0, rows * 200 * sizeof(unsigned int));
    unsigned char **pattern = (unsigned char **)malloc((size_t)p_size * sizeof(unsigned char *));
    for (int j = 0; j < p_size; j++) {
        pattern[j] = (unsigned char *)calloc((size_t)m + 1, 1);
        memcpy(pattern[j], pat_flat + (size_t)j * m, (size_t)m);
    }
    pointer_array = calloc((size_t)p_size * m, sizeof(struct sbom_state *));
    double t0 = now_s();
    struct sbom_table *table = preproc_sbom(pattern, m, p_size, alphabet, state_transition, state_final_multi);
    double t1 = now_s();
    unsigned int matches = text ? search_sbom(pattern, m, (unsigned char *)text, n, table) : 0;
    double t2 = now_s();
    if (idcounter) *idcounter = table->idcounter;
    if (patterncounter) *patterncounter = table->patterncounter;
    if (t_preproc) *t_preproc = t1 - t0;
    if (t_search) *t_search = t2 - t1;
    free_sbom(table, m);
    free(pointer_array);
    pointer_array = NULL;
    for (int j = 0; j < p_size; j++) free(pattern[j]);
    free(pattern);
    return matches;
}

/* SOG, main.c:300-322 (multisog): the tables are caller-owned (main.c:495-515).  Only preproc_sog8's T8,
 * scanner_hs and scanner_index are deterministic in the reference (its scanner_hs2 depends on an uninitialised
 * variable, sog/sog8.c:124,135); the count search_sog8 returns with that bitmap is reported as well, for the
 * record -- tests pin the tables, not this count. */
unsigned long long ref_run_sog8(const unsigned char *pat_flat, int p_size, const unsigned char *text, int n,
                                uint8_t *T8, uint32_t *scanner_hs, int *scanner_index, uint8_t *scanner_hs2)
{
    const int m = 8, B = 3;
    unsigned char **pattern = (unsigned char **)malloc((size_t)p_size * sizeof(unsigned char *));
    for (int j = 0; j < p_size; j++) {
        pattern[j] = (unsigned char *)calloc((size_t)m + 1, 1);
        memcpy(pattern[j], pat_flat + (size_t)j * m, (size_t)m);
    }
    preproc_sog8(T8, scanner_hs, scanner_index, scanner_hs2, pattern, m, (unsigned char *)text, n, p_size, B);
    unsigned int matches = text ? search_sog8(T8, scanner_hs, scanner_index, scanner_hs2, pattern, m, (unsigned char *)text, n, p_size, B) : 0;
    for (int j = 0; j < p_size; j++) free(pattern[j]);
    free(pattern);
    return matches;
}
