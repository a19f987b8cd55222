/*
 * csrc/acm_host.c -- ONE automaton for a set of patterns of mixed lengths.
 *
 * The reference's automaton cannot take such a set: ac_maketree never merges outputs along the supply links
 * ("Join outputs missing", ac/ac.c:118), search_ac counts a position only when its DEEPEST state is itself a
 * leaf (ac/ac.c:215-216), and ac_addstring mis-marks a pattern that is a prefix of an earlier one
 * (ac/ac.c:136-143).  The defined result of a mixed set is the length-class decomposition (one reference run
 * per length, counts summed; SURVEY 8 preamble) = sum over text positions e of the number of DISTINCT patterns
 * that end at e.  That is what an Aho-Corasick automaton with suffix-closed output COUNTS computes in one pass:
 *
 *     out(s) = [a pattern ends at s] + out(supply(s))          (the joined outputs ac/ac.c:118 leaves out)
 *     count  = sum over positions of out(state after the position)
 *
 * Device form: the automaton cut at depth K -- the deepest cut whose table fits LDS -- as a complete DFA whose
 * entries carry, beside the next row, out(next) and a candidate bit "the K-symbol prefix of a LONGER pattern
 * ends here".  Patterns of length <= K are counted by the scan alone; a candidate is walked down the goto trie
 * in HBM from its start and every pattern end deeper than K on that path counts once (each occurrence of a long
 * pattern has exactly one such start).  End ownership: a lane counts the positions of its own segment and warms
 * its state up over the K-1 bytes in front of it.
 */
#include "smh_internal.h"

#include <stdlib.h>
#include <string.h>

#define SMH_ACM_LDS_BUDGET (156u * 1024u)

void smh_acm_free(struct smh_acm *a)
{
    if (!a) return;
    if (a->dev) smh_acm_dev_free(a->dev);
    free(a->scan);
    free(a->g_goto);
    free(a->g_final);
    a->magic = 0;
    free(a);
}

struct smh_acm *smh_acm_compile(const unsigned char *patterns, const uint32_t *lengths, int p_size, int alphabet)
{
    if (!patterns || !lengths || p_size < 1 || alphabet < 1 || alphabet > 256) {
        smh_set_error("smh_acm_compile: bad arguments");
        return NULL;
    }
    const size_t A = (size_t)alphabet;
    uint64_t total = 0;
    uint32_t max_len = 0;
    for (int j = 0; j < p_size; ++j) {
        if (lengths[j] < 1 || lengths[j] > 65535u) { smh_set_error("smh_acm_compile: pattern length out of range"); return NULL; }
        total += lengths[j];
        if (lengths[j] > max_len) max_len = lengths[j];
    }
    if (total + 1 >= (1u << 24)) { smh_set_error("smh_acm_compile: more than 2^24 trie nodes"); return NULL; }
    const uint32_t cap = (uint32_t)total + 1;
    /* 1. trie in insertion order */
    uint32_t *child = (uint32_t *)calloc((size_t)cap * A, sizeof(uint32_t));
    uint8_t *fin = (uint8_t *)calloc(cap, 1);
    uint32_t *depth = (uint32_t *)calloc(cap, sizeof(uint32_t));
    uint32_t *fail = (uint32_t *)calloc(cap, sizeof(uint32_t));
    uint32_t *order = (uint32_t *)malloc((size_t)cap * sizeof(uint32_t));
    uint32_t *newid = (uint32_t *)malloc((size_t)cap * sizeof(uint32_t));
    uint32_t *out = (uint32_t *)calloc(cap, sizeof(uint32_t));
    uint32_t *first = NULL, *delta = NULL;
    struct smh_acm *a = (struct smh_acm *)calloc(1, sizeof *a);
    if (!child || !fin || !depth || !fail || !order || !newid || !out || !a) goto oom;
    uint32_t nodes = 1;
    {
        uint64_t off = 0;
        for (int j = 0; j < p_size; ++j) {
            uint32_t s = 0;
            for (uint32_t i = 0; i < lengths[j]; ++i) {
                const unsigned c = patterns[off + i];
                if ((int)c >= alphabet) { smh_set_error("smh_acm_compile: symbol %u >= alphabet %d", c, alphabet); goto bad; }
                uint32_t t = child[(size_t)s * A + c];
                if (!t) {
                    t = nodes++;
                    child[(size_t)s * A + c] = t;
                    depth[t] = depth[s] + 1;
                }
                s = t;
            }
            fin[s] = 1; /* duplicates of a pattern mark the same node: counted once, as in every class of the decomposition */
            off += lengths[j];
        }
    }
    /* 2. breadth-first order, supply links, joined output counts */
    {
        uint32_t head = 0, tail = 0;
        order[tail++] = 0;
        while (head < tail) {
            const uint32_t s = order[head++];
            for (size_t c = 0; c < A; ++c) {
                const uint32_t t = child[(size_t)s * A + c];
                if (!t) continue;
                if (s == 0) {
                    fail[t] = 0;
                } else {
                    uint32_t f = fail[s];
                    while (f && !child[(size_t)f * A + c]) f = fail[f];
                    const uint32_t g = child[(size_t)f * A + c];
                    fail[t] = (g && g != t) ? g : 0;
                }
                out[t] = (uint32_t)fin[t] + out[fail[t]];
                order[tail++] = t;
            }
        }
        for (uint32_t k = 0; k < nodes; ++k) newid[order[k]] = k;
    }
    /* 3. first[d] = first BFS id with depth >= d */
    first = (uint32_t *)malloc(((size_t)max_len + 3) * sizeof(uint32_t));
    if (!first) goto oom;
    {
        uint32_t d = 0;
        for (uint32_t k = 0; k < nodes; ++k)
            while (d <= depth[order[k]]) first[d++] = k;
        while (d <= max_len + 2) first[d++] = nodes;
    }
    /* 4. the deepest cut that fits: 16-bit entries (candidate | count:2 | row:13) or 32-bit (candidate | count:7 | row:24) */
    int K = 0, eb = 0;
    for (int k = (int)(max_len < 17u ? max_len : 17u); k >= 1 && !K; --k) { /* the fast path warms up over at most 16 bytes */
        const uint32_t rows = first[k + 1];
        uint32_t maxout = 0;
        for (uint32_t r = 0; r < rows; ++r)
            if (out[order[r]] > maxout) maxout = out[order[r]];
        /* 32-bit entries wherever they fit: their layout makes the scan 3.5 VALU per byte against 8 (acm_lane.h) */
        if (maxout <= 127u && (uint64_t)rows * A * 4u <= SMH_ACM_LDS_BUDGET) { K = k; eb = 4; }
        else if (rows <= 8192u && maxout <= 3u && (uint64_t)rows * A * 2u <= SMH_ACM_LDS_BUDGET) { K = k; eb = 2; }
    }
    if (!K) {
        smh_set_error("smh_acm_compile: no cut of the automaton fits %u bytes of LDS (alphabet %d)", SMH_ACM_LDS_BUDGET, alphabet);
        goto bad;
    }
    const uint32_t rows = first[K + 1];
    if (K < (int)max_len) {
        /* how often would a random position be a candidate?  A cut that is too shallow (large alphabets: a row costs
         * alphabet entries) makes every other position one, and each costs a walk through HBM: not worth one pass */
        double cands = 0.0, space = 1.0;
        for (uint32_t r = first[K]; r < rows; ++r) {
            const uint32_t s = order[r];
            for (size_t c = 0; c < A; ++c)
                if (child[(size_t)s * A + c]) { cands += 1.0; break; }
        }
        for (int i = 0; i < K; ++i) space *= (double)alphabet;
        if (cands / space > 2e-3) {
            smh_set_error("smh_acm_compile: the deepest cut that fits LDS (depth %d) would make %.2f %% of the positions candidates",
                          K, 100.0 * cands / space);
            goto bad;
        }
    }
    /* 5. complete transition function over the kept rows (BFS ids), entries that would leave depth K bent to supply */
    delta = (uint32_t *)malloc((size_t)rows * A * sizeof(uint32_t)); /* old-id targets of the FULL automaton, then cut */
    a->scan = calloc((size_t)rows * A * (size_t)eb + 16, 1);
    if (!delta || !a->scan) goto oom;
    for (uint32_t r = 0; r < rows; ++r) {
        const uint32_t s = order[r];
        for (size_t c = 0; c < A; ++c) {
            uint32_t t = child[(size_t)s * A + c];
            if (!t && s) t = delta[(size_t)newid[fail[s]] * A + c]; /* supply row comes earlier in BFS order */
            if (t && depth[t] > (uint32_t)K) t = fail[t];           /* a goto edge out of a depth-K row: one supply step reaches depth <= K */
            while (depth[t] > (uint32_t)K) t = fail[t];
            delta[(size_t)r * A + c] = t;
            int has_child = 0;
            if (depth[t] == (uint32_t)K)
                for (size_t cc = 0; cc < A && !has_child; ++cc) has_child = child[(size_t)t * A + cc] != 0;
            const uint32_t row = newid[t], cnt = out[t], cand = has_child ? 1u : 0u;
            if (eb == 2) ((uint16_t *)a->scan)[(size_t)r * A + c] = (uint16_t)(row | (cnt << 13) | (cand << 15));
            else ((uint32_t *)a->scan)[(size_t)r * A + c] = cand | (row * (uint32_t)A * 4u) | (cnt << 24); /* acm_lane.h smh_acm_entry<uint32_t> */
        }
    }
    /* 6. goto trie in BFS ids for the walk of the candidates */
    a->g_goto = (uint32_t *)calloc((size_t)nodes * A + 4, sizeof(uint32_t));
    a->g_final = (uint8_t *)calloc((size_t)nodes + 16, 1);
    if (!a->g_goto || !a->g_final) goto oom;
    for (uint32_t k = 0; k < nodes; ++k) {
        const uint32_t s = order[k];
        a->g_final[k] = fin[s];
        for (size_t c = 0; c < A; ++c) {
            const uint32_t t = child[(size_t)s * A + c];
            a->g_goto[(size_t)k * A + c] = t ? newid[t] : 0u;
        }
    }
    a->magic = SMH_MAGIC_ACM;
    a->alphabet = alphabet;
    a->max_len = (int)max_len;
    a->K = K;
    a->nodes = nodes;
    a->scan_rows = rows;
    a->entry_bytes = eb;
    a->scan_bytes = (uint32_t)(((size_t)rows * A * (size_t)eb + 15) & ~(size_t)15);
    a->exact = K >= (int)max_len;
    free(child); free(fin); free(depth); free(fail); free(order); free(newid); free(out); free(first); free(delta);
    return a;
oom:
    smh_set_error("smh_acm_compile: out of memory");
bad:
    free(child); free(fin); free(depth); free(fail); free(order); free(newid); free(out); free(first); free(delta);
    if (a) { free(a->scan); free(a->g_goto); free(a->g_final); free(a); }
    return NULL;
}
