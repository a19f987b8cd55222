/*
 * csrc/wm_pos_kernels.hip -- the tuned Wu-Manber kernels instantiated in positions mode
 * (smh_launch_wm_block_positions).  Same source as wm_kernels.hip (wm_kernels.inc).
 */
#define SMH_TU_POSITIONS 1
#include "wm_kernels.inc"
