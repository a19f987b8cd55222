/*
 * csrc/ac_pos_kernels.hip -- the tuned Aho-Corasick kernels instantiated in positions mode
 * (smh_launch_ac_dfa_positions).  Same source as ac_kernels.hip (ac_kernels.inc); a separate
 * translation unit so that the two sets of template instantiations compile in parallel.
 */
#define SMH_TU_POSITIONS 1
#define SMH_TU_WIDE 0
#include "ac_kernels.inc"
