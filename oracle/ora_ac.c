/*
 * oracle/ora_ac.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the reference's Aho-Corasick CPU path, ac/ac.c, with the
 * caller conventions of main.c.  Same observable results (state numbering,
 * flat tables, match count); the node pool is index-based instead of one
 * malloc per node, and the BFS uses an array queue instead of the reference's
 * O(len) list_append (ac/list.h:57-74), which changes running time only.
 */
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NONE 0xFFFFFFFFu

struct ora_ac_table {
    /* struct ac_table (smatcher.h:49-53): idcounter, patterncounter, zerostate */
    uint32_t idcounter;
    uint32_t patterncounter;
    int alphabet;
    uint32_t cap;      /* allocated states */
    uint32_t *next;    /* cap * alphabet child ids, NONE = NULL pointer (struct ac_state.next) */
    uint32_t *fail;    /* struct ac_state.fail */
    uint8_t *output;   /* struct ac_state.output != NULL */
};

static void ora_die(const char *msg)
{
    fprintf(stderr, "%s", msg);
    exit(1);
}

/* ac/ac.c:37-63 ac_init: root node, all of its edges NULL, and row 0 of the
 * flat transition table set to 0 (every other row stays at the caller's -1). */
static void ora_ac_init(ora_ac_table *g, int alphabet, uint32_t cap, int32_t *state_transition)
{
    g->alphabet = alphabet;
    g->cap = cap;
    g->next = (uint32_t *)malloc((size_t)cap * alphabet * sizeof(uint32_t));
    g->fail = (uint32_t *)malloc((size_t)cap * sizeof(uint32_t));
    g->output = (uint8_t *)calloc(cap, 1);
    if (!g->next || !g->fail || !g->output) ora_die("Could not allocate memory\n");
    for (int c = 0; c < alphabet; ++c) g->next[c] = NONE;
    g->fail[0] = 0;
    g->patterncounter = 0;
    g->idcounter = 1;
    for (int c = 0; c < alphabet; ++c) state_transition[c] = 0;
}

/* ac/ac.c:127-196 ac_addstring: follow existing edges, then create one state
 * per remaining symbol (id = creation order) writing
 * state_transition[parent*alphabet + symbol] = child (ac/ac.c:162); mark the
 * end state final once (ac/ac.c:183-194), duplicates do not bump patterncounter.
 *
 * Reference quirk not restated: its follow loop evaluates string[m] after a
 * fully existing path (ac/ac.c:136-143).  With all patterns of length m the
 * depth-m node has no children, so that read never changes the outcome; the
 * oracle simply stops at j == m. */
static void ora_ac_addstring(ora_ac_table *g, const uint8_t *string, int m,
                             int32_t *state_transition, uint32_t *state_final)
{
    const int A = g->alphabet;
    uint32_t state = 0;
    int j = 0;
    while (j < m) {
        uint32_t nx = g->next[(size_t)state * A + string[j]];
        if (nx == NONE) break;
        state = nx;
        ++j;
    }
    for (; j < m; ++j) {
        uint32_t id = g->idcounter++;
        if (id >= g->cap) ora_die("Could not allocate memory\n");
        for (int c = 0; c < A; ++c) g->next[(size_t)id * A + c] = NONE;
        g->output[id] = 0;
        g->fail[id] = 0;
        state_transition[(size_t)state * A + string[j]] = (int32_t)id;
        g->next[(size_t)state * A + string[j]] = id;
        state = id;
    }
    if (!g->output[state]) {
        state_final[state] = 1;
        g->output[state] = 1;
        g->patterncounter++;
    }
}

/* ac/ac.c:79-124 ac_maketree: NULL edges of the root become self loops
 * (ac/ac.c:86-88); breadth-first, fail(child) = goto(fail*(parent), symbol)
 * (ac/ac.c:107-112); state_supply[] is written for depth >= 2 only
 * (ac/ac.c:114) -- depth-1 states rely on the caller's zero fill.  No output
 * merging along fail links ("Join outputs missing", ac/ac.c:118). */
static void ora_ac_maketree(ora_ac_table *g, uint32_t *state_supply)
{
    const int A = g->alphabet;
    uint32_t *queue = (uint32_t *)malloc((size_t)g->idcounter * sizeof(uint32_t));
    size_t head = 0, tail = 0;
    for (int c = 0; c < A; ++c) {
        uint32_t s = g->next[c];
        if (s == NONE) {
            g->next[c] = 0;
        } else {
            queue[tail++] = s;
            g->fail[s] = 0;
        }
    }
    while (head < tail) {
        uint32_t cur = queue[head++];
        for (int c = 0; c < A; ++c) {
            uint32_t s = g->next[(size_t)cur * A + c];
            if (s == NONE) continue;
            queue[tail++] = s;
            uint32_t state = g->fail[cur];
            while (g->next[(size_t)state * A + c] == NONE) state = g->fail[state];
            g->fail[s] = g->next[(size_t)state * A + c];
            state_supply[s] = g->fail[s];
        }
    }
    free(queue);
}

/* ac/ac.c:224-245 preproc_ac */
ora_ac_table *ora_preproc_ac(const uint8_t *const *pattern, int m, int p_size, int alphabet,
                             int32_t *state_transition, uint32_t *state_supply,
                             uint32_t *state_final)
{
    ora_ac_table *t = (ora_ac_table *)calloc(1, sizeof(*t));
    if (!t) ora_die("Could not initialize table\n");
    ora_ac_init(t, alphabet, (uint32_t)((size_t)m * p_size + 1), state_transition);
    for (int i = 0; i < p_size; ++i)
        ora_ac_addstring(t, pattern[i], m, state_transition, state_final);
    ora_ac_maketree(t, state_supply);
    return t;
}

uint32_t ora_ac_idcounter(const ora_ac_table *t) { return t->idcounter; }
uint32_t ora_ac_patterncounter(const ora_ac_table *t) { return t->patterncounter; }

/* ac/ac.c:198-222 search_ac: one goto per text byte, walking fail links while
 * the goto is undefined; a position counts when the state reached is itself
 * an accepting state (r->output != NULL, ac/ac.c:215-216). */
uint64_t ora_search_ac(const uint8_t *text, int64_t n, const ora_ac_table *t)
{
    const int A = t->alphabet;
    uint32_t r = 0;
    uint64_t matches = 0;
    for (int64_t column = 0; column < n; ++column) {
        uint32_t s;
        while ((s = t->next[(size_t)r * A + text[column]]) == NONE) r = t->fail[r];
        r = s;
        if (t->output[r]) ++matches;
    }
    return matches;
}

/* The same walk over the exported flat tables, as the reference's GPU
 * kernels do it (cuda/cuda_ac.cu:584-591): -1 = no edge, row 0 holds 0. */
uint64_t ora_search_ac_tables(const uint8_t *text, int64_t n, int alphabet,
                              const int32_t *state_transition, const uint32_t *state_supply,
                              const uint32_t *state_final)
{
    int32_t r = 0, s;
    uint64_t matches = 0;
    for (int64_t column = 0; column < n; ++column) {
        while ((s = state_transition[(size_t)r * alphabet + text[column]]) == -1)
            r = (int32_t)state_supply[r];
        r = s;
        matches += state_final[r];
    }
    return matches;
}

/* ac/ac.c:247-252 free_ac */
void ora_free_ac(ora_ac_table *t)
{
    if (!t) return;
    free(t->next);
    free(t->fail);
    free(t->output);
    free(t);
}
