/* Development knobs -- compiled ONLY into tests/emu/libsmatcher_hip_testing.so (-DSMH_TESTING).
 *
 * The product library (libsmatcher_hip.so) is built without SMH_TESTING: every accessor below is then a constant, the
 * compiler drops the knob names with the dead branches, and no code path of a scan, a launch or a table build reads the
 * environment.  (What the product does read, once per process, is listed in include/smatcher_hip.h "Environment": SMH_ADAPT,
 * SMH_HOST_PIECE_KIB, SMH_MULTI_SHARE_DEVICE -- none of them can change a count.)  Some knobs exist for timing experiments
 * and DO change the count ("nohalo=1", "stmin=-1", "drop=1"): that is why none of this may ship.
 *
 * Testing build: the five knob strings live in one struct (smh_tune.c), filled from SMH_{WM,AC,HASH,KEY,PSET}_TUNE the first
 * time any knob is asked for and replaced by smh_test_tune_set(); readers take a read lock, so launches on several threads and
 * a test that changes a knob do not race the way getenv / setenv would. */
#ifndef SMH_TUNE_H
#define SMH_TUNE_H

enum { SMH_TUNE_WM = 0, SMH_TUNE_AC, SMH_TUNE_HASH, SMH_TUNE_KEY, SMH_TUNE_PSET, SMH_TUNE_N };

#ifdef SMH_TESTING
#ifdef __cplusplus
extern "C" {
#endif
/* 1 when the knob string `which` contains `word` */
int smh_tune_has(int which, const char *word);
/* the integer behind `key` ("wg=" -> atoi of what follows) or dflt when the key is absent */
int smh_tune_int(int which, const char *key, int dflt);
/* testing library's exported setter: str == NULL or "" clears the knob string; returns 0, -1 for a bad `which` */
int smh_test_tune_set(int which, const char *str);
#ifdef __cplusplus
}
#endif
#else
#define smh_tune_has(which, word) 0
#define smh_tune_int(which, key, dflt) (dflt)
#endif

#endif /* SMH_TUNE_H */
