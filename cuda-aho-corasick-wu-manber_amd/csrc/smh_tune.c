/* The testing library's knob store (smh_tune.h).  NOT part of libsmatcher_hip.so: the Makefile compiles this file into
 * tests/emu/libsmatcher_hip_testing.so only, and without -DSMH_TESTING it is empty. */
#ifdef SMH_TESTING
#include "smh_tune.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static struct {
    pthread_rwlock_t lock;
    int loaded;
    char *s[SMH_TUNE_N];
} g_tune = { PTHREAD_RWLOCK_INITIALIZER, 0, { 0 } };

static const char *const g_tune_env[SMH_TUNE_N] = { "SMH_WM_TUNE", "SMH_AC_TUNE", "SMH_HASH_TUNE", "SMH_KEY_TUNE", "SMH_PSET_TUNE" };

/* first use: one snapshot of the environment (command-line tools set the variables before they start) */
static void tune_load_locked(void)
{
    if (g_tune.loaded) return;
    for (int i = 0; i < SMH_TUNE_N; ++i) {
        const char *e = getenv(g_tune_env[i]);
        g_tune.s[i] = e && *e ? strdup(e) : NULL;
    }
    g_tune.loaded = 1;
}

static void tune_ensure(void)
{
    pthread_rwlock_rdlock(&g_tune.lock);
    const int loaded = g_tune.loaded;
    pthread_rwlock_unlock(&g_tune.lock);
    if (loaded) return;
    pthread_rwlock_wrlock(&g_tune.lock);
    tune_load_locked();
    pthread_rwlock_unlock(&g_tune.lock);
}

int smh_tune_has(int which, const char *word)
{
    if (which < 0 || which >= SMH_TUNE_N) return 0;
    tune_ensure();
    pthread_rwlock_rdlock(&g_tune.lock);
    const int r = g_tune.s[which] && strstr(g_tune.s[which], word) != NULL;
    pthread_rwlock_unlock(&g_tune.lock);
    return r;
}

int smh_tune_int(int which, const char *key, int dflt)
{
    if (which < 0 || which >= SMH_TUNE_N) return dflt;
    tune_ensure();
    int r = dflt;
    pthread_rwlock_rdlock(&g_tune.lock);
    if (g_tune.s[which]) {
        const char *at = strstr(g_tune.s[which], key);
        if (at) r = atoi(at + strlen(key));
    }
    pthread_rwlock_unlock(&g_tune.lock);
    return r;
}

int smh_test_tune_set(int which, const char *str)
{
    if (which < 0 || which >= SMH_TUNE_N) return -1;
    pthread_rwlock_wrlock(&g_tune.lock);
    tune_load_locked();
    free(g_tune.s[which]);
    g_tune.s[which] = str && *str ? strdup(str) : NULL;
    pthread_rwlock_unlock(&g_tune.lock);
    return 0;
}
#else
typedef int smh_tune_not_in_the_product; /* ISO C forbids an empty translation unit */
#endif
