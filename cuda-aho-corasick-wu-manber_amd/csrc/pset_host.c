/*
 * csrc/pset_host.c -- pattern sets with mixed lengths (include/smatcher_hip.h "pattern sets").
 *
 * The reference cannot express them: preproc_ac / preproc_wu take one m (smatcher.h:89,101), and
 * ac_addstring fed a pattern that is a prefix of an earlier, longer one marks the wrong state
 * final (ac/ac.c:136-143,183-186).  The defined result is the length-class decomposition -- one
 * reference run per distinct length, counts summed -- and that is what a set handle executes on
 * the device: one smh_ac / smh_wm per length class over the same resident text.
 * Pure host code on top of the public entry points; no kernel of its own.
 */
#include "smh_internal.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

static double smh_wall_seconds(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

#define SMH_MAGIC_PSET 0x50534554u /* "PSET" */

struct smh_pset_class {
    uint32_t length;
    uint32_t patterns;
    smh_ac *ac; /* exactly one of ac / wm is set */
    smh_wm *wm;
};

struct smh_pset {
    uint32_t magic;
    int alphabet;
    int algorithm;
    uint32_t patterns;
    uint32_t n_classes;
    struct smh_pset_class *cls;
    /* SMH_ALGO_WM sets with 2..32 lengths, all >= 3: ONE pass.  `suffix` is a handle over the patterns'
     * last min-length symbols: its block filter proposes END columns for every length at once, and a
     * surviving column is verified against each class's table (wm_lane.h smh_wm_verify_class).  The
     * count is the same sum over classes; the text is read once instead of once per length. */
    smh_wm *suffix;
    smh_wm **class_wm;
    /* SMH_ALGO_AC sets with 2 or more lengths: ONE pass with one automaton whose states carry joined output
     * counts (acm_host.c); NULL when no cut of it fits LDS, then one scan per class */
    struct smh_acm *acm;
    /* split form (SMH_ALGO_WM, 4-letter alphabet, TWO passes): `suffix` / `class_wm` cover the classes from index
     * split_first on (patterns of SMH_GRAM_PAIR2_SPLIT symbols and more: grouped pair-gram filter, next to no
     * candidates), `acm` the shorter patterns (whose matches are frequent and which the automaton counts in line).
     * 0 when suffix / acm cover the whole set. */
    uint32_t split_first;
};

static int cmp_u32(const void *a, const void *b)
{
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

void smh_pset_free(smh_pset *set)
{
    if (!set || set->magic != SMH_MAGIC_PSET) return;
    for (uint32_t i = 0; i < set->n_classes; ++i) {
        smh_ac_free(set->cls[i].ac);
        smh_wm_free(set->cls[i].wm);
    }
    smh_wm_free(set->suffix);
    smh_acm_free(set->acm);
    free(set->class_wm);
    free(set->cls);
    set->magic = 0;
    free(set);
}

smh_pset *smh_pset_compile(const unsigned char *patterns, const uint32_t *lengths, int p_size, int alphabet,
                           int algorithm)
{
    if (!patterns || !lengths || p_size < 1 || alphabet < 1 || alphabet > 256 ||
        (algorithm != SMH_ALGO_AC && algorithm != SMH_ALGO_WM)) {
        smh_set_error("smh_pset_compile: bad arguments");
        return NULL;
    }
    uint64_t total = 0;
    for (int j = 0; j < p_size; ++j) {
        if (lengths[j] < 1 || lengths[j] > 65535u) {
            smh_set_error("smh_pset_compile: pattern %d has length %u (1..65535 supported)", j, lengths[j]);
            return NULL;
        }
        total += lengths[j];
    }
    for (uint64_t k = 0; k < total; ++k)
        if ((int)patterns[k] >= alphabet) {
            smh_set_error("smh_pset_compile: symbol %u >= alphabet %d", patterns[k], alphabet);
            return NULL;
        }
    uint32_t *sorted = (uint32_t *)malloc((size_t)p_size * sizeof(uint32_t));
    smh_pset *set = (smh_pset *)calloc(1, sizeof *set);
    unsigned char *flat = (unsigned char *)malloc(total ? total : 1);
    if (!sorted || !set || !flat) goto oom;
    memcpy(sorted, lengths, (size_t)p_size * sizeof(uint32_t));
    qsort(sorted, (size_t)p_size, sizeof(uint32_t), cmp_u32);
    uint32_t n_classes = 0;
    for (int j = 0; j < p_size; ++j)
        if (j == 0 || sorted[j] != sorted[j - 1]) sorted[n_classes++] = sorted[j];
    set->magic = SMH_MAGIC_PSET;
    set->alphabet = alphabet;
    set->algorithm = algorithm;
    set->patterns = (uint32_t)p_size;
    set->cls = (struct smh_pset_class *)calloc(n_classes, sizeof *set->cls);
    if (!set->cls) goto oom;
    for (uint32_t c = 0; c < n_classes; ++c) {
        /* gather the class in the order given (the reference's state / bucket numbering follows it) */
        const uint32_t L = sorted[c];
        uint32_t count = 0;
        uint64_t off = 0;
        for (int j = 0; j < p_size; ++j) {
            if (lengths[j] == L) memcpy(flat + (size_t)count++ * L, patterns + off, L);
            off += lengths[j];
        }
        struct smh_pset_class *k = &set->cls[c];
        k->length = L;
        k->patterns = count;
        set->n_classes = c + 1; /* so that a failure below frees what exists */
        if (algorithm == SMH_ALGO_WM && L >= 3)
            k->wm = smh_wm_compile(flat, (int)L, (int)count, alphabet);
        else
            k->ac = smh_ac_compile_patterns(flat, (int)L, (int)count, alphabet);
        if (!k->wm && !k->ac) { /* the class compiler has set the error text */
            free(sorted); free(flat);
            smh_pset_free(set);
            return NULL;
        }
    }
    /* ---- fewer passes than one per class.  SMH_ALGO_WM sets, in this order: (A) grouped pair-gram filter over the
     * full patterns, (B) the automaton with joined output counts, (C) block filter over the patterns' last min-length
     * symbols while it is sparse, (D) split form: (A) for the long patterns + (B) for the short ones.  SMH_ALGO_AC
     * sets: (B). ---- */
    const int no_gram = smh_tune_has(SMH_TUNE_PSET, "nogram");
    const int no_acm = smh_tune_has(SMH_TUNE_PSET, "classes");
    int grouped = 1;
    if (algorithm == SMH_ALGO_WM && n_classes >= 2 && n_classes <= SMH_PSET_MAX_ONE_PASS_CLASSES && set->cls[0].length >= 3) {
        const uint32_t Lmin = set->cls[0].length;
        uint64_t off = 0;
        for (int j = 0; j < p_size; ++j) { /* every pattern's last Lmin symbols */
            memcpy(flat + (size_t)j * Lmin, patterns + off + lengths[j] - Lmin, Lmin);
            off += lengths[j];
        }
        set->suffix = smh_wm_compile(flat, (int)Lmin, p_size, alphabet);
        set->class_wm = (smh_wm **)malloc(n_classes * sizeof(smh_wm *));
        if (set->class_wm)
            for (uint32_t c = 0; c < n_classes; ++c) set->class_wm[c] = set->cls[c].wm;
        /* (A) 4-letter alphabet: every pattern contributes all the planes its length allows, so candidates are rare even
         * where the min-length suffix is not selective -- as long as the SHORT patterns are few (smh_internal.h) */
        if (set->suffix && set->class_wm && !no_gram) {
            grouped = smh_wm_build_gram_mixed(set->suffix, patterns, lengths, p_size);
            if (grouped < 0) goto oom;
        }
    }
    if (n_classes >= 2 && grouped != 0 && !no_acm) /* (B); NULL: no cut of the automaton fits LDS with few candidates */
        set->acm = smh_acm_compile(patterns, lengths, p_size, alphabet);
    if (set->suffix && grouped != 0) {
        /* (C) worth it only while survivors are rare: each one costs a verify per class, a scan per class ~0.25 ms/GiB;
         * the automaton, where it exists, is the better single pass (0.29-0.38 ms/GiB whatever the set) */
        if (set->acm || !set->class_wm || set->suffix->filter_density >= SMH_PSET_ONE_PASS_DENSITY) {
            smh_wm_free(set->suffix);
            set->suffix = NULL;
        }
    }
    if (algorithm == SMH_ALGO_WM && !set->suffix && !set->acm && alphabet == 4 && n_classes >= 2 && !no_gram && !no_acm &&
        set->cls[0].length < SMH_GRAM_PAIR2_SPLIT && set->cls[n_classes - 1].length >= SMH_GRAM_PAIR2_SPLIT) {
        /* (D) thousands of patterns over a wide range of lengths: neither filter is selective (many short patterns) and
         * the whole automaton does not fit.  Two passes: long patterns through (A), short ones through (B). */
        uint32_t first_long = 0;
        while (set->cls[first_long].length < SMH_GRAM_PAIR2_SPLIT) ++first_long;
        const uint32_t n_long = n_classes - first_long, Lmin = set->cls[first_long].length;
        unsigned char *part = (unsigned char *)malloc(total ? total : 1);
        uint32_t *plen = (uint32_t *)malloc((size_t)p_size * sizeof(uint32_t));
        if (!part || !plen) { free(part); free(plen); goto oom; }
        if (n_long <= SMH_PSET_MAX_ONE_PASS_CLASSES) {
            int np = 0;
            uint64_t off = 0, fill = 0;
            for (int j = 0; j < p_size; ++j) { /* the long patterns, and their last Lmin symbols */
                if (lengths[j] >= SMH_GRAM_PAIR2_SPLIT) {
                    memcpy(part + fill, patterns + off, lengths[j]);
                    memcpy(flat + (size_t)np * Lmin, patterns + off + lengths[j] - Lmin, Lmin);
                    fill += lengths[j];
                    plen[np++] = lengths[j];
                }
                off += lengths[j];
            }
            set->suffix = smh_wm_compile(flat, (int)Lmin, np, alphabet);
            free(set->class_wm);
            set->class_wm = (smh_wm **)malloc(n_long * sizeof(smh_wm *));
            int ok = set->suffix && set->class_wm && smh_wm_build_gram_mixed(set->suffix, part, plen, np) == 0;
            if (ok) {
                for (uint32_t c = 0; c < n_long; ++c) set->class_wm[c] = set->cls[first_long + c].wm;
                np = 0; off = 0; fill = 0;
                for (int j = 0; j < p_size; ++j) { /* the short patterns */
                    if (lengths[j] < SMH_GRAM_PAIR2_SPLIT) {
                        memcpy(part + fill, patterns + off, lengths[j]);
                        fill += lengths[j];
                        plen[np++] = lengths[j];
                    }
                    off += lengths[j];
                }
                set->acm = smh_acm_compile(part, plen, np, alphabet);
                ok = set->acm != NULL;
            }
            if (ok) {
                set->split_first = first_long;
            } else { /* one scan per class */
                smh_wm_free(set->suffix);
                set->suffix = NULL;
                smh_acm_free(set->acm);
                set->acm = NULL;
            }
        }
        free(part);
        free(plen);
    }
    free(sorted);
    free(flat);
    return set;
oom:
    free(sorted); free(flat);
    if (set) { set->magic = SMH_MAGIC_PSET; smh_pset_free(set); }
    smh_set_error("smh_pset_compile: out of memory");
    return NULL;
}

static int pset_ok(const smh_pset *set, const char *who)
{
    if (set && set->magic == SMH_MAGIC_PSET) return 1;
    smh_set_error("%s: bad handle", who);
    return 0;
}

int smh_pset_get_info(const smh_pset *set, smh_pset_info *out)
{
    if (!pset_ok(set, "smh_pset_get_info") || !out) return SMH_EINVAL;
    memset(out, 0, sizeof *out);
    out->alphabet = (uint32_t)set->alphabet;
    out->algorithm = (uint32_t)set->algorithm;
    out->classes = set->n_classes;
    out->patterns = set->patterns;
    out->min_length = set->cls[0].length;
    out->max_length = set->cls[set->n_classes - 1].length;
    out->one_pass = (set->suffix != NULL || set->acm != NULL) && set->split_first == 0;
    out->passes = set->split_first ? 2u : (out->one_pass ? 1u : set->n_classes);
    return SMH_OK;
}

int smh_pset_get_class(const smh_pset *set, uint32_t i, uint32_t *length, uint32_t *patterns)
{
    if (!pset_ok(set, "smh_pset_get_class")) return SMH_EINVAL;
    if (i >= set->n_classes) {
        smh_set_error("smh_pset_get_class: class %u of %u", i, set->n_classes);
        return SMH_EINVAL;
    }
    if (length) *length = set->cls[i].length;
    if (patterns) *patterns = set->cls[i].patterns;
    return SMH_OK;
}

int smh_pset_scan(smh_pset *set, const unsigned char *d_text, uint64_t n, uint64_t *d_count, void *stream)
{
    if (!pset_ok(set, "smh_pset_scan")) return SMH_EINVAL;
    if (((uintptr_t)d_text & 15u) == 0 && (set->suffix || set->acm)) {
        /* one pass, or (split form) the long patterns' filter pass and the short patterns' automaton pass */
        int rc = SMH_OK;
        if (set->suffix)
            rc = smh_wm_scan_multi(set->suffix, set->class_wm, (int)(set->n_classes - set->split_first), d_text, n, d_count, stream);
        if (rc == SMH_OK && set->acm) rc = smh_acm_scan(set->acm, d_text, n, d_count, stream);
        return rc;
    }
    for (uint32_t c = 0; c < set->n_classes; ++c) {
        struct smh_pset_class *k = &set->cls[c];
        const int rc = k->wm ? smh_wm_scan(k->wm, d_text, n, d_count, SMH_VARIANT_TUNED, stream)
                             : smh_ac_scan(k->ac, d_text, n, d_count, SMH_VARIANT_TUNED, stream);
        if (rc != SMH_OK) return rc;
    }
    return SMH_OK;
}

int smh_pset_positions(smh_pset *set, const unsigned char *d_text, uint64_t n, uint64_t *d_positions,
                       uint64_t capacity, uint64_t *d_cursor, void *stream)
{
    if (!pset_ok(set, "smh_pset_positions")) return SMH_EINVAL;
    uint32_t per_class_end = set->n_classes;
    if (set->suffix && ((uintptr_t)d_text & 15u) == 0) {
        const int rc = smh_wm_positions_multi(set->suffix, set->class_wm, (int)(set->n_classes - set->split_first), d_text, n,
                                              d_positions, capacity, d_cursor, stream);
        if (rc != SMH_OK || set->split_first == 0) return rc;
        per_class_end = set->split_first; /* split form: the short classes one by one (the automaton only counts) */
    }
    for (uint32_t c = 0; c < per_class_end; ++c) {
        struct smh_pset_class *k = &set->cls[c];
        const int rc = k->wm ? smh_wm_positions(k->wm, d_text, n, d_positions, capacity, d_cursor, stream)
                             : smh_ac_positions(k->ac, d_text, n, d_positions, capacity, d_cursor, stream);
        if (rc != SMH_OK) return rc;
    }
    return SMH_OK;
}

int smh_pset_count_host(smh_pset *set, const unsigned char *text, uint64_t n, uint64_t *count,
                        double *kernel_seconds)
{
    if (!pset_ok(set, "smh_pset_count_host")) return SMH_EINVAL;
    if (!count || (n && !text)) {
        smh_set_error("smh_pset_count_host: bad arguments");
        return SMH_EINVAL;
    }
    *count = 0;
    if (kernel_seconds) *kernel_seconds = 0.0;
    /* one upload, every class scans the resident copy */
    unsigned char *d_text = NULL;
    uint64_t *d_count = NULL;
    int rc = smh_device_malloc((void **)&d_text, ((n + 15) / 16) * 16 + 64);
    if (rc == SMH_OK) rc = smh_device_malloc((void **)&d_count, 16);
    if (rc == SMH_OK && n) rc = smh_copy_to_device(d_text, text, n, NULL);
    if (rc == SMH_OK) rc = smh_device_memset(d_count, 0, 16, NULL);
    if (rc == SMH_OK) rc = smh_stream_synchronize(NULL);
    if (rc == SMH_OK) {
        const double t0 = smh_wall_seconds();
        rc = smh_pset_scan(set, d_text, n, d_count, NULL);
        if (rc == SMH_OK) rc = smh_stream_synchronize(NULL);
        if (kernel_seconds) *kernel_seconds = smh_wall_seconds() - t0;
    }
    if (rc == SMH_OK) rc = smh_copy_to_host(count, d_count, 8, NULL);
    if (rc == SMH_OK) rc = smh_stream_synchronize(NULL);
    if (d_text) smh_device_free(d_text);
    if (d_count) smh_device_free(d_count);
    return rc;
}
