// Routing flags and tunables of a context that exist for the parity tests (tests/) and the A/B scripts (profiles/): every route is
// bit-identical to the default one, none is needed by a caller of the bin_matrix / CSSCode surface, so they are not part of
// include/gf2hip.h.  They go through the same two entry points, gf2_ctx_set_flags and gf2_ctx_set_option; the numbers are stable
// (quantum_css_codes_amd/_native.py states them again, tests/test_abi.py compares).
#pragma once

#include "gf2hip.h"

// ---- flags (bits 5, 8 and 10 are public: GF2_F_MC_DENSE, GF2_F_RREF_SEQUENTIAL, GF2_F_NORMALIZE_SEQUENTIAL) -------------------------
#define GF2_F_SPARSE_GATHER        (1u << 0)   /* gf2_syndrome_sparse_dev: wavefront-per-sample column gather            */
#define GF2_F_SPARSE_SLABS         (1u << 1)   /* gf2_syndrome_sparse_dev: LDS row-slab pipeline also for small batches  */
#define GF2_F_NO_REDO              (1u << 2)   /* slab pipeline: no column is left to the redo pass                     */
#define GF2_F_GATHER_GENERIC       (1u << 3)   /* slab pipeline: compiler-scheduled gather kernel                       */
#define GF2_F_MC_UNFUSED           (1u << 4)   /* gf2_mc_run: sampler, then the syndrome calls, on one stream           */
#define GF2_F_MC_FUSED             (1u << 6)   /* gf2_mc_run: sampler fused into the column-gather kernel               */
#define GF2_F_MC_PIPELINE          (1u << 7)   /* gf2_mc_run on small codes: sampler + syndrome + histogram kernels     */
#define GF2_F_RREF_NO_SMALL        (1u << 9)   /* gf2_rref*: no wavefront-per-matrix kernel                             */
#define GF2_F_SAMPLER_GENERIC      (1u << 11)  /* gf2_sample_errors_dev: lane-per-segment kernel                        */
#define GF2_F_DIAG_CLOCKS          (1u << 12)  /* slab pipeline: print wall-clock stamps of its kernels to stderr       */
#define GF2_F_DIAG_MC_TIMES        (1u << 13)  /* gf2_mc_run: print the host's phases to stderr                         */
#define GF2_F_MC_ROWS              (1u << 14)  /* gf2_mc_run: packed rows from the sampler, records by the compact kernel */
#define GF2_F_COMBINE_FOLDED       (1u << 15)  /* slab pipeline: the combine step of a pass inside the next pass' compact kernel */
#define GF2_F_RREF_NO_LOOKAHEAD    (1u << 16)  /* gf2_rref* on more than 8192 rows: the next pair's panels after, not under, the trailing pass */
#define GF2_F_RREF_LOOKAHEAD       (1u << 17)  /* ... under it whatever the size (default: from 128 MiB of matrix on)              */
#define GF2_F_COMBINE_SEPARATE     (1u << 18)  /* slab pipeline: a combine kernel after every pass (default since round 4: the combine step of a pass rides in the next pass' gather kernel) */
#define GF2_F_ALL                  ((1u << 19) - 1u)   /* every defined flag; gf2_ctx_set_flags refuses other bits              */

// ---- tunables (0 and 4 are public: GF2_OPT_SLAB_PASS_LOG2, GF2_OPT_MC_CHUNK_LOG2) --------------------------------------------------
#define GF2_OPT_COMBINE_BLOCKS  1   /* slab pipeline: workgroups of the combine kernel (default 128 * 1024 / threads)        */
#define GF2_OPT_GATHER_REVERSE   2   /* slab pipeline: 1 (default) = the gather kernel walks the records last tile first  */
#define GF2_OPT_REDO_BLOCKS_PER_CU 3 /* slab pipeline: workgroups per CU of the redo kernel, 1..64 (default 8)                 */
#define GF2_OPT_COMBINE_THREADS 5   /* slab pipeline: threads per workgroup of the combine kernel, 64 / 128 / 256 / 512 / 1024 (default 1024) */
#define GF2_OPT_GATHER_CROSS    6   /* slab pipeline: 1 = a gather step takes ranks 4k..4k+3 of four sorted tiles, 0 (default) = a quartile of one */
#define GF2_OPT_GATHER_OVER     7   /* slab pipeline: gather workgroups per CU over a launch, 1..8 (default 1)                            */
#define GF2_OPT_RREF_SMALL_BCAST 8  /* wavefront-per-matrix RREF: how the pivot rows reach the other rows.  One at a time: 0 = through LDS, 1 = through v_readlane; 2 = four at a time through a table of their sums in LDS (contiguous rows of whole 16-byte pieces, at most 32 words per lane; otherwise as the default).  Default: 2 for rows of 16 words, half LDS half readlane for at most 64 rows of at most 8 words, else 0 */
#define GF2_OPT_MC_SAMPLER_WAVES 9  /* gf2_mc_run at n <= 4096, sparse rates: record-sampler wavefronts per CU, 1..9 (default 8)          */
#define GF2_OPT_MC_TAIL_CAP     10  /* gf2_mc_run at n <= 4096, sparse rates: erroneous qubits of a 512-qubit segment that the record sampler's lanes take in step; a sample with more in a segment is finished by a lane of its own later.  0 (all in step), 2, 4, 6 or 8 (default: by the rate) */
#define GF2_OPT_RREF_STREAM_VARIANT 11 /* blocked RREF: development switches -- above 4096 rows those of launch_rref_sweeps_streamed (gf2_elim.hip); up to 4096 rows 1 = keep the pivot-row snapshots for large batches */
#define GF2_OPT_RREF_ROWS_WG    12  /* blocked RREF: rows per workgroup of the trailing pass (>= 64; default: by the batch)            */
#define GF2_OPT_RREF_SWEEP_K    13  /* blocked RREF: panels per sweep, 4 or 2 up to 4096 rows (default: by the shape), 4 with the rows streamed above; 0 = the round-4 pair kernels */
#define GF2_OPT_COUNT           14
