/*
 * tests/c_driver/legacy_driver.c -- a plain C caller of include/smatcher.h.
 *
 * Does what the reference driver does around the hot path, in the reference's order and with the
 * reference's allocation / initialisation conventions, single rank:
 *   table setup            main.c:410-420 (AC), main.c:429-449 (WM)
 *   multiac                main.c:125-157   preproc_ac -> search_ac -> free_ac
 *   multiwm2               main.c:268-298   preproc_wu2 -> search_wu2
 *   GPU variants           main.c:582-648   cuda_ac1..5, cuda_wm1..5
 * It is compiled with gcc against libsmatcher_hip.so by tests/test_c_driver.py to show that a C
 * program written against the reference's API builds and links unchanged.
 * usage: legacy_driver <m> <p_size> <n> <alphabet>     (synthetic corpus, text seed 42, pattern seed 7)
 */
#include "smatcher.h"
#include "smatcher_hip.h" /* only for the corpus generator */

int main(int argc, char **argv)
{
    if (argc < 5) fail("usage: legacy_driver m p_size n alphabet\n");
    int m = atoi(argv[1]), p_size = atoi(argv[2]), n = atoi(argv[3]), alphabet = atoi(argv[4]), B = 3;
    int i, j;

    unsigned char *text = (unsigned char *)malloc((size_t)n);
    unsigned char *pattern2 = (unsigned char *)malloc((size_t)m * p_size);
    unsigned char **pattern = (unsigned char **)malloc(p_size * sizeof(unsigned char *));
    smh_corpus_text_host(text, (uint64_t)n, 0, 42, alphabet);
    smh_corpus_patterns(pattern2, m, p_size, 7, alphabet, 42, (uint64_t)n, 2);
    for (j = 0; j < p_size; j++) {
        pattern[j] = (unsigned char *)calloc((size_t)m + 1, 1);
        memcpy(pattern[j], pattern2 + (size_t)j * m, (size_t)m);
    }

    /* main.c:410-420 */
    int *state_transition = (int *)malloc((size_t)(m * p_size + 1) * alphabet * sizeof(int));
    memset(state_transition, -1, (size_t)(m * p_size + 1) * alphabet * sizeof(int));
    unsigned int *state_supply = (unsigned int *)malloc((m * p_size + 1) * sizeof(unsigned int));
    memset(state_supply, 0, (m * p_size + 1) * sizeof(unsigned int));
    unsigned int *state_final = (unsigned int *)malloc((m * p_size + 1) * sizeof(unsigned int));
    memset(state_final, 0, (m * p_size + 1) * sizeof(unsigned int));

    /* main.c:429-449 */
    wu_determine_shiftsize(alphabet);
    m_nBitsInShift = 2;
    int *SHIFT = (int *)malloc(shiftsize * sizeof(int));
    int *PREFIX_value = (int *)malloc((size_t)shiftsize * p_size * sizeof(int));
    int *PREFIX_index = (int *)malloc((size_t)shiftsize * p_size * sizeof(int));
    int *PREFIX_size = (int *)malloc(shiftsize * sizeof(int));
    for (i = 0; i < (int)shiftsize; i++) {
        SHIFT[i] = m - B + 1;
        PREFIX_size[i] = 0;
    }

    /* multiac */
    struct ac_table *table = preproc_ac(pattern, m, p_size, alphabet, state_transition, state_supply, state_final);
    printf("preproc_ac states \t%u\t patterns \t%u\n", table->idcounter, table->patterncounter);
    fflush(stdout);
    int matches = search_ac(text, n, table);
    free_ac(table, alphabet);
    printf("search_ac matches \t%i\n", matches);

    /* multiwm2 */
    preproc_wu2(pattern2, m, p_size, alphabet, B, SHIFT, PREFIX_value, PREFIX_index, PREFIX_size);
    matches = search_wu2(pattern2, m, p_size, text, n, SHIFT, PREFIX_value, PREFIX_index, PREFIX_size);
    printf("search_wm2 matches \t%i\n", matches);

    /* GPU variants, main.c:582-648 */
    cuda_ac1(m, text, n, p_size, alphabet, state_transition, state_supply, state_final);
    cuda_ac5(m, text, n, p_size, alphabet, state_transition, state_supply, state_final);
    double gpuTime[2];
    int r1 = cuda_wm1(pattern2, m, text, n, p_size, alphabet, B, SHIFT, PREFIX_value, PREFIX_index, PREFIX_size, &gpuTime[0]);
    int r5 = cuda_wm5(pattern2, m, text, n, p_size, alphabet, B, SHIFT, PREFIX_value, PREFIX_index, PREFIX_size, &gpuTime[1]);
    printf("cuda_wm1 matches \t%i\t cuda_wm5 matches \t%i\n", r1, r5);

    for (j = 0; j < p_size; j++) free(pattern[j]);
    free(pattern); free(pattern2); free(text);
    free(state_transition); free(state_supply); free(state_final);
    free(SHIFT); free(PREFIX_value); free(PREFIX_index); free(PREFIX_size);
    fflush(stdout);
    return 0;
}
