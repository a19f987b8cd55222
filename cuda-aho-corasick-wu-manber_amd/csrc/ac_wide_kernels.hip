/*
 * csrc/ac_wide_kernels.hip -- the counting Aho-Corasick kernels for automata with 32-bit table entries
 * (more than 32768 rows in the LDS image).  Same source as ac_kernels.hip (ac_kernels.inc); a separate
 * translation unit so that the template instantiations compile in parallel.
 */
#define SMH_TU_POSITIONS 0
#define SMH_TU_WIDE 1
#include "ac_kernels.inc"
