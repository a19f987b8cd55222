/*
 * csrc/ac_kernels.hip -- Aho-Corasick scan kernels for gfx950 (MI355X).
 *
 * ac_dfa_kernel    tuned path: the depth-K automaton staged in LDS once per
 *                  workgroup (stride 1 or 2 symbols per lookup), 64-byte text
 *                  segments streamed straight into registers, NCH automata per
 *                  lane, candidates compacted into a per-wave queue (ballot +
 *                  prefix count) and verified against the full DFA in HBM,
 *                  wave-level reduction and one 64-bit atomic per wave.
 *                  Replaces ac_kernel3..5b (cuda/cuda_ac.cu:23-532) -- no
 *                  textures, no per-thread counters copied back to the host
 *                  (cuda/cuda_ac.cu:667-673).
 * ac_table_kernel  walks the reference-layout goto/supply/final tables from
 *                  HBM/L2 as given; replaces ac_kernel1/2 (cuda/cuda_ac.cu:535-592).
 *
 * Roofline: HBM read, 1 byte per text symbol.  No MFMA: the work
 * is one dependent table lookup per byte.
 */
#define SMH_TU_POSITIONS 0
#define SMH_TU_WIDE 0
#include "ac_kernels.inc"
