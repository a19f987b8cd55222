/*
 * csrc/wm_kernels.hip -- Wu-Manber scan kernels for gfx950 (MI355X).
 *
 * wm_block_kernel  tuned path: the device SHIFT table (block filter bit set) is
 *                  staged in LDS once per workgroup; every END column of a
 *                  lane's 64-byte segment is tested against it with a rolling
 *                  block code; survivors go through the HBM HASH/PREFIX verify
 *                  table.  Replaces wm_kernel3..5 (cuda/cuda_wm.cu:60-650).
 * wm_table_kernel  the reference tables as given: SHIFT staged in LDS
 *                  (16-bit), per-lane skip loop, CSR bucket scan, byte compare.
 *                  Replaces wm_kernel1/2 (cuda/cuda_wm.cu:786-1058).
 *
 * Roofline: HBM read, 1 byte per text symbol.  No MFMA.
 */
#define SMH_TU_POSITIONS 0
#include "wm_kernels.inc"
