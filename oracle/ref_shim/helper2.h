/*
 * oracle/ref_shim/helper2.h -- TEST INFRASTRUCTURE ONLY.
 *
 * The upstream repository includes "../helper2.h" from smatcher.h:31 but does
 * not ship that file (nor ../helper.c, Makefile:47-48).  The two hot-path
 * translation units, ac/ac.c and wu/wu.c, need exactly three names from it:
 * MIN (wu/wu.c:131), and fail() (ac/ac.c:46,155,235; wu/wu.c:46); MAX is
 * defined for symmetry.  This shim supplies those and nothing else, so that
 * oracle/Makefile can compile the reference's own ac/ac.c and wu/wu.c where
 * they lie.  No algorithmic code lives here.
 */
#ifndef ORACLE_REF_SHIM_HELPER2_H
#define ORACLE_REF_SHIM_HELPER2_H
#include <stdio.h>
#include <stdlib.h>
#ifndef MIN
#define MIN(a, b) (((a) < (b)) ? (a) : (b))
#endif
#ifndef MAX
#define MAX(a, b) (((a) > (b)) ? (a) : (b))
#endif
static inline void fail(const char *msg)
{
    fputs(msg, stderr);
    exit(1);
}
#endif
