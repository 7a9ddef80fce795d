/*
 * gf2hip.h -- C ABI of libgf2hip.so, the MI355X (gfx950) GF(2) engine behind the bin_matrix /
 * CSSCode hot path of jimpo/quantum-css-codes.
 *
 * The reference has no FFI layer: its boundary is the Python module surface (bin_matrix.py,
 * css_code.py).  The Python package quantum_css_codes_amd keeps that surface and binds these entry
 * points with ctypes (quantum_css_codes_amd/_native.py; INTEGRATION.md shows the stub a reference
 * maintainer would add).  Each entry point names the reference lines it replaces; paths are
 * relative to the reference repository.
 *
 * Conventions
 *   - Every function returns an int: GF2_OK or a negative GF2_E_* code.  gf2_last_error() returns
 *     a thread-local human-readable message for the last failure on the calling thread.
 *   - Packed layout: row-major uint64_t words, column j in word j>>6 at bit j&63, `ld` words per
 *     row (ld >= ceil(n/64)).  Pad bits (columns >= n) must be zero on input and are zero on output.
 *   - "host" pointers are caller-owned host memory; the library never keeps them past return.
 *     "dev" pointers are device memory from gf2_dev_alloc (or any hipMalloc'ed / torch-allocated
 *     buffer on the context's device).
 *   - A gf2_ctx owns one HIP stream.  Host-buffer entry points are synchronous.  `_dev` entry points
 *     enqueue on the context's stream and return; gf2_ctx_sync() waits.  A context is not
 *     thread-safe; distinct contexts are independent.
 *   - There is no CPU fallback: without a usable GPU, compute entry points fail with GF2_E_HIP.
 */
#ifndef GF2HIP_H
#define GF2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GF2_OK            0
#define GF2_E_ARG        (-1)  /* bad argument                                                   */
#define GF2_E_COLUMNS    (-2)  /* ValueError("not enough columns"), css_code.py:811-812          */
#define GF2_E_DEPENDENT  (-3)  /* InvalidCodeError("rows are not independent"), css_code.py:825  */
#define GF2_E_HIP        (-4)  /* HIP runtime error / no device                                  */
#define GF2_E_NOMEM      (-5)  /* allocation failure                                             */
#define GF2_E_NOTCSS     (-6)  /* NotImplementedError("only handles CSS codes"), css_code.py:762-763 */
#define GF2_E_RCCL       (-7)  /* RCCL error (gf2_comm_*, gf2_hist_allreduce)                    */

#define GF2_LAYOUT_SAMPLE_MAJOR 0  /* E: B rows of lde words (one error per row); S: B rows of lds words */
#define GF2_LAYOUT_BIT_SLICED   1  /* E: n rows of ceil(B/64) words (word b of row q = qubit q of samples
                                      64b..64b+63); S: r rows likewise.  Requires n <= 64 and r <= 64. */
#define GF2_LAYOUT_TILED        2  /* Device-native error layout of the large-n syndrome kernel.  Samples are
                                      grouped in tiles of 64; with ldt = gf2_tiled_ld(n) (even) words per sample,
                                      word w of sample b sits at word offset
                                          (b>>6)*64*ldt + (w>>1)*128 + (b&63)*2 + (w&1)
                                      so the 64 lanes of a wavefront read one 16-byte piece each from 1 KiB of
                                      contiguous memory.  The buffer holds ceil(B/64) whole tiles
                                      (gf2_tiled_words).  Syndromes of tiled errors are SLAB-MAJOR: word s
                                      (rows 64s..64s+63) of sample b at s*lds + b with lds >= B, so a wavefront
                                      stores 512 contiguous bytes. */

#define GF2_HIST_FULL    0     /* bins indexed by vec_to_int(syndrome) (bin_matrix.py:36-43), 2^r bins  */
#define GF2_HIST_WEIGHT  1     /* bins indexed by the syndrome's Hamming weight, r+1 bins               */

typedef struct gf2_ctx gf2_ctx;
typedef struct gf2_check gf2_check;   /* a parity-check matrix prepared on the device */

/* ---- library / context ------------------------------------------------------------------------ */
int gf2_version(void);
const char* gf2_last_error(void);
int gf2_device_count(int* count_out);
int gf2_ctx_create(int device, gf2_ctx** ctx_out);
int gf2_ctx_destroy(gf2_ctx* ctx);
int gf2_ctx_sync(gf2_ctx* ctx);

/* Routing flags of a context.  Several entry points have more than one implementation behind them, all bit-identical; 0 (the
 * default) leaves the choice to the library.  The routes a caller of the bin_matrix / CSSCode surface may want to force: */
#define GF2_F_MC_DENSE             (1u << 5)   /* gf2_mc_run: dense table kernel whatever the error rate                */
#define GF2_F_RREF_SEQUENTIAL      (1u << 8)   /* gf2_rref*: one pivot per step                                         */
#define GF2_F_NORMALIZE_SEQUENTIAL (1u << 10)  /* gf2_normalize*: one pivot per step                                    */
/* (The other bits name routes that exist for the parity tests and the A/B scripts under profiles/ -- quantum_css_codes_amd/csrc/
 * gf2_tuning.h lists them; gf2_ctx_set_flags refuses bits no route is defined for.  The library never reads the environment per
 * call; gf2_ctx_create reads GF2_FLAGS once as the initial value.) */
int gf2_ctx_set_flags(gf2_ctx* ctx, uint32_t flags);
int gf2_ctx_get_flags(gf2_ctx* ctx, uint32_t* flags_out);
/* Tunables of a context (value < 0 restores the default).  The two that size device memory: */
#define GF2_OPT_SLAB_PASS_LOG2  0   /* slab pipeline: 2^k samples per pass through the workspace, 12 <= k <= 24 (default 22) */
#define GF2_OPT_MC_CHUNK_LOG2   4   /* gf2_mc_run at n <= 4096, sparse rates: 2^k samples per chunk, 16 <= k <= 22 (default 22; 21 with GF2_F_MC_ROWS) */
/* (Further option numbers, used by the A/B scripts: quantum_css_codes_amd/csrc/gf2_tuning.h.) */
int gf2_ctx_set_option(gf2_ctx* ctx, int option, int64_t value);

/* Device memory and stream-ordered copies on the context's stream (copies are synchronous). */
int gf2_dev_alloc(gf2_ctx* ctx, size_t bytes, void** dev_out);
int gf2_dev_free(gf2_ctx* ctx, void* dev);
int gf2_dev_zero(gf2_ctx* ctx, void* dev, size_t bytes);
int gf2_h2d(gf2_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
int gf2_d2h(gf2_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);

/* HIP-event timing on the context's stream: bracket any sequence of _dev calls. */
int gf2_timer_start(gf2_ctx* ctx);
int gf2_timer_stop(gf2_ctx* ctx, float* elapsed_ms_out);   /* synchronises on the stop event */
/* Accumulated HIP-event time of one kernel family since the last reset (see GF2_K_*); each launch of
 * that family is bracketed by its own pair of events when profiling is enabled. */
#define GF2_K_SYNDROME 0
#define GF2_K_HIST     1
#define GF2_K_SAMPLER  2
#define GF2_K_ELIM     3
#define GF2_K_COUNT    4
int gf2_profile_enable(gf2_ctx* ctx, int on);
int gf2_profile_reset(gf2_ctx* ctx);
int gf2_profile_get(gf2_ctx* ctx, int kernel_family, double* total_ms_out, int64_t* launches_out);

/* Memory-bandwidth probe: streams `bytes` (a multiple of 16) from src_dev with 16-byte loads; dst_dev null = read only,
 * else the bytes are also stored there (copy).  Asynchronous on the context's stream; bracket it with gf2_timer_*.  It is
 * what bench.py quotes roofline fractions against next to the 8 TB/s specification.  sink_dev: one device word. */
int gf2_membw_probe_dev(gf2_ctx* ctx, const void* src_dev, void* dst_dev, size_t bytes, uint64_t* sink_dev);

/* ---- host-side packing (pure host code, no GPU needed) -------------------------------------------
 * Dense integer arrays <-> packed words.  Packing applies `& 1` (the reference reduces lazily with
 * np.mod(.,2), bin_matrix.py:34, css_code.py:39-40).  Strides are in elements. */
int gf2_pack_rows_u8(const uint8_t* src, int64_t m, int64_t n, int64_t src_stride, uint64_t* dst, int64_t ld);
int gf2_pack_rows_i64(const int64_t* src, int64_t m, int64_t n, int64_t src_stride, uint64_t* dst, int64_t ld);
/* The same with CSSCode's input test (css_code.py:39-44, "parity check matrix must be binary") made on the way: *other_out = 1
 * when some entry is neither 0 nor 1 (the packed rows are still entries & 1). */
int gf2_pack_rows_binary_u8(const uint8_t* src, int64_t m, int64_t n, int64_t src_stride, uint64_t* dst, int64_t ld, int* other_out);
int gf2_pack_rows_binary_i64(const int64_t* src, int64_t m, int64_t n, int64_t src_stride, uint64_t* dst, int64_t ld, int* other_out);
int gf2_unpack_rows_u8(const uint64_t* src, int64_t m, int64_t n, int64_t ld, uint8_t* dst, int64_t dst_stride);
int gf2_unpack_rows_i64(const uint64_t* src, int64_t m, int64_t n, int64_t ld, int64_t* dst, int64_t dst_stride);

/* ---- GF(2) linear algebra on host buffers -------------------------------------------------------- */

/* bin_matrix.reduced_row_echelon_form (bin_matrix.py:8-34).  In place on the packed matrix.
 * pivots_out (capacity min(m,n)) receives the pivot column of each of the first *rank_out rows. */
int gf2_rref(gf2_ctx* ctx, uint64_t* a, int64_t m, int64_t n, int64_t ld,
             int64_t* pivots_out, int64_t* rank_out);

/* Batched form: `batch` independent m x n matrices, matrix b at a + b*m*ld.  One workgroup per
 * matrix.  pivots_out: batch x min(m,n); rank_out: batch. */
int gf2_rref_batch(gf2_ctx* ctx, uint64_t* a, int64_t batch, int64_t m, int64_t n, int64_t ld,
                   int64_t* pivots_out, int64_t* rank_out);

/* Device-resident forms (asynchronous on the context's stream).  pivots_dev: batch x min(m,n) int64
 * (may be null); rank_dev: batch int64.  gf2_normalize_dev: swaps_dev capacity 2*r int64, nswaps_dev one
 * int64, status_dev one int (0 = ok, 1 = rows are not independent). */
int gf2_rref_batch_dev(gf2_ctx* ctx, uint64_t* a_dev, int64_t batch, int64_t m, int64_t n, int64_t ld,
                       int64_t* pivots_dev, int64_t* rank_dev);
int gf2_normalize_dev(gf2_ctx* ctx, uint64_t* h_dev, int64_t r, int64_t n, int64_t ld, int64_t offset,
                      int64_t* swaps_dev, int64_t* nswaps_dev, int* status_dev);

/* [build-defined; anchored on bin_matrix.py:8-34 + css_code.py:124-161]  Canonical
 * nullspace basis read off the RREF: row t has a 1 at free column F[t] and R[i,F[t]] at pivot column
 * P[i].  n_out needs capacity n rows of ldn words (at most n - rank are written); *rows_out = n-rank. */
int gf2_nullspace(gf2_ctx* ctx, const uint64_t* a, int64_t m, int64_t n, int64_t ld,
                  uint64_t* n_out, int64_t ldn, int64_t* rows_out);

/* css_code.normalize_parity_check (css_code.py:809-836).  In place; identity block lands in columns
 * offset..offset+r-1; swaps_out (capacity 2*r) receives (column, column) pairs in order.
 * Returns GF2_E_COLUMNS / GF2_E_DEPENDENT for the two reference exceptions. */
int gf2_normalize(gf2_ctx* ctx, uint64_t* h, int64_t r, int64_t n, int64_t ld, int64_t offset,
                  int64_t* swaps_out, int64_t* nswaps_out);

/* css_code.swap_columns (css_code.py:783-785).  In place. */
int gf2_swap_columns(gf2_ctx* ctx, uint64_t* a, int64_t m, int64_t n, int64_t ld, int64_t i, int64_t j);

/* np.mod(np.matmul(A, B.T), 2): the commutation check of css_code.py:47.  C is ra x rb packed
 * (ldc >= ceil(rb/64)). */
int gf2_matmul_abt(gf2_ctx* ctx, const uint64_t* a, int64_t ra, int64_t lda,
                   const uint64_t* b, int64_t rb, int64_t ldb, int64_t n,
                   uint64_t* c, int64_t ldc);

/* css_code.syndrome_table (css_code.py:715-735) for n <= 64, r <= 24 [SURVEY.md 8f item 2].  h_rows: r words, qubit j =
 * bit j.  table_out: 2^r words indexed by bin_matrix.vec_to_int(syndrome) (row 0 = most significant bit,
 * bin_matrix.py:36-43); an entry is the packed error of weight <= t with that syndrome, or all ones.  *t_out is the
 * decoding threshold the reference returns: the classes 0..t have pairwise distinct syndromes and class t + 1 does not
 * (or t = n, or t = max_weight when max_weight >= 0 [build-defined cap] is reached first).  *entries_out (may be null)
 * = number of filled entries. */
int gf2_syndrome_table(gf2_ctx* ctx, const uint64_t* h_rows, int64_t r, int64_t n, int64_t max_weight,
                       uint64_t* table_out, int64_t* t_out, int64_t* entries_out);

/* The same search for 64 < n <= 128 (SURVEY.md 8f item 2 names n = 23..127).  h_rows: r rows of two words.  A slot of table_out
 * (2^r words) holds (weight << 32) | rank -- the error's rank inside its weight class in the combinatorial number system
 * (positions c_1 < ... < c_w have rank C(c_1, 1) + ... + C(c_w, w)) -- or all ones; the caller unranks.  t_out, entries_out and
 * max_weight as above. */
int gf2_syndrome_table_wide(gf2_ctx* ctx, const uint64_t* h_rows, int64_t r, int64_t n, int64_t max_weight,
                            uint64_t* table_out, int64_t* t_out, int64_t* entries_out);

/* The same search for 128 < n <= 8192 (still r <= 24: the table has 2^r slots; beyond that the Python host enumerates the classes
 * and only the syndromes come from the device).  h_rows: r packed rows of ld words.  Errors are enumerated as position lists
 * and keyed by the XOR of their columns' keys; a class is enumerated only if it can fit the table (C(n, w) <= 2^r), which
 * keeps w <= 8 for every n > 128.  Table slots, t_out, entries_out and max_weight as in gf2_syndrome_table_wide. */
int gf2_syndrome_table_cols(gf2_ctx* ctx, const uint64_t* h_rows, int64_t r, int64_t n, int64_t ld, int64_t max_weight,
                            uint64_t* table_out, int64_t* t_out, int64_t* entries_out);

/* The same search beyond 24 checks [SURVEY.md 8f item 2: a k = 1 CSS code has r_1 + r_2 = n - 1, so from n = 51 on one of its
 * checks has more than 24 rows]: an open-addressing hash table on the device, sized from the weight classes it holds instead
 * of 2^r.  1 <= r <= 127, 1 <= n <= 8192, h_rows: r packed rows of ld words.  Keys are vec_to_int(syndrome) exactly (row 0 =
 * most significant bit): one word per entry for r <= 63, two (low word first) for 64 <= r <= 127 -- the reference's own keys
 * wrap beyond 63 bits (bin_matrix.py:40-43).  Output: *entries_out entries (the classes 0 .. t), in no particular order --
 * keys_out (1 or 2 words each) and vals_out = (weight << 32) | rank in the class (combinatorial number system, as above); they
 * are written only if capacity >= *entries_out (call again with larger buffers otherwise; capacity 0 just counts).  At most
 * 2^28 errors are enumerated in all: GF2_E_NOMEM if no collision has shown by then and max_weight (>= 0) does not stop the
 * search earlier. */
int gf2_syndrome_table_hashed(gf2_ctx* ctx, const uint64_t* h_rows, int64_t r, int64_t n, int64_t ld, int64_t max_weight,
                              uint64_t* keys_out, uint64_t* vals_out, int64_t capacity, int64_t* t_out, int64_t* entries_out);

/* css_code.transform_stabilisers (css_code.py:737-781) [SURVEY.md 8f item 3].  mat: k rows of ld words holding the k x 2n
 * stabiliser matrix [X | Z] (column j = bit j), rewritten in place.  gates: ngates rows of three int32 (kind, a, b):
 * kind 0 = H on qubit a (conjugate_h_with_check_mat, :757-767), kind 1 = CNOT control a target b
 * (conjugate_cnot_with_check_mat, :769-781).  Gates apply in order.  *stop_out = -1 and GF2_OK when all applied.
 * Otherwise mat holds the result of gates[0 : *stop_out] and the call returns GF2_E_NOTCSS (that gate is an H on a
 * qubit where some row has X and Z) or GF2_E_ARG (unknown kind / qubit outside [0, n): the reference's ValueErrors,
 * :747-755).  2n <= 20480. */
int gf2_conjugate_gates(gf2_ctx* ctx, uint64_t* mat, int64_t k, int64_t n, int64_t ld, const int32_t* gates,
                        int64_t ngates, int64_t* stop_out);

/* Row Hamming weights: np.sum(mat, axis=1) of css_code.is_doubly_even (css_code.py:846-850). */
int gf2_row_weights(gf2_ctx* ctx, const uint64_t* a, int64_t m, int64_t n, int64_t ld, uint32_t* weights_out);

/* ---- syndrome extraction -------------------------------------------------------------------------
 * np.mod(np.matmul(parity_check, e), 2) of css_code.py:728 for B errors at once
 * [build-defined batching, SURVEY.md 8a x2]. */

/* Uploads H and builds its device-side lookup tables.  The handle belongs to ctx. */
int gf2_check_create(gf2_ctx* ctx, const uint64_t* h, int64_t r, int64_t n, int64_t ld, gf2_check** check_out);
int gf2_check_destroy(gf2_ctx* ctx, gf2_check* check);

/* Tiled layout helpers: words per sample (even, >= ceil(n/64)) and words of a buffer for `batch` samples. */
int64_t gf2_tiled_ld(int64_t n);
int64_t gf2_tiled_words(int64_t n, int64_t batch);
/* Sample-major (batch x lde) -> tiled, both on the device; asynchronous. */
int gf2_retile_dev(gf2_ctx* ctx, const uint64_t* e_dev, int64_t batch, int64_t lde, int64_t n, uint64_t* tiled_dev);

/* Host buffers, synchronous. */
int gf2_syndrome_batch(gf2_ctx* ctx, const uint64_t* h, int64_t r, int64_t n, int64_t ldh,
                       const uint64_t* e, int64_t batch, int64_t lde, int layout,
                       uint64_t* s_out, int64_t lds);

/* Device buffers, asynchronous on the context's stream.
 * Sample-major: e_dev is batch x lde, s_dev is batch x lds (lds >= ceil(r/64)).  For n > 64 the errors are
 *               first re-tiled into context workspace (one extra streaming pass); keep resident data in
 *               GF2_LAYOUT_TILED to avoid it.
 * Tiled:        e_dev as described at GF2_LAYOUT_TILED (lde ignored); s_dev is slab-major: ceil(r/64) rows
 *               of lds >= batch words.
 * Bit-sliced:   e_dev is n x lde with lde >= ceil(batch/64); s_dev is r x lds, lds >= ceil(batch/64). */
int gf2_syndrome_dev(gf2_ctx* ctx, const gf2_check* check, const uint64_t* e_dev, int64_t batch, int64_t lde,
                     int layout, uint64_t* s_dev, int64_t lds);

/* Sparse-error path: work proportional to the weight of each error.  Sample-major errors (batch x lde).  Produces the
 * sample-major syndromes (s_dev, batch x lds; may be null) and/or accumulates the syndrome-weight histogram (hist_dev
 * with r+1 uint64 bins; may be null) without materialising the syndromes.  Identical results to gf2_syndrome_dev;
 * faster when errors are sparse (DESIGN.md gives the crossover).  Fails for small checks (n, r <= 64) and r > 8192.
 * Three implementations behind it: batches of at least 32768 samples on a check with r <= 2048 and
 * n - r <= 2400 take the LDS row-slab pipeline (compact -> gather -> combine, plus a redo pass for columns left out of the
 * records), with or without syndromes stored (since round 4 every gather workgroup stores its slab's 64-byte piece); checks
 * with n <= 512 and r <= 256 the lane-per-sample kernel; everything else one wavefront per sample gathering columns of the
 * transposed check from L2.  Words of a syndrome row past ceil(r/64) (lds larger than needed) are either left as they were
 * or zeroed. */
int gf2_syndrome_sparse_dev(gf2_ctx* ctx, const gf2_check* check, const uint64_t* e_dev, int64_t batch, int64_t lde,
                            uint64_t* s_dev, int64_t lds, uint64_t* hist_dev, int64_t nbins);

/* Histogram of packed syndromes (device), accumulated into hist_dev (uint64 bins).  layout
 * GF2_LAYOUT_SAMPLE_MAJOR: s_dev is batch x lds; GF2_LAYOUT_TILED: slab-major as written by gf2_syndrome_dev for
 * tiled errors (lds >= batch).  mode GF2_HIST_FULL needs r <= 24 and nbins == 2^r; GF2_HIST_WEIGHT needs
 * nbins == r+1. */
int gf2_histogram_dev(gf2_ctx* ctx, const uint64_t* s_dev, int64_t batch, int64_t lds, int layout, int64_t r,
                      int mode, uint64_t* hist_dev, int64_t nbins);

/* ---- Monte-Carlo --------------------------------------------------------------------------------
 * [build-defined, SURVEY.md 8a x3]  Sample `i` (global index) is a pure function of (seed, i); the
 * generator is specified in DESIGN.md ("Sampler") and restated in oracle/.  X errors are caught by
 * parity_check_c2, Z errors by parity_check_c1 (css_code.py:457-470). */

/* Writes packed errors for samples first_sample .. first_sample+count-1.  layout GF2_LAYOUT_SAMPLE_MAJOR:
 * count x lde; GF2_LAYOUT_TILED: gf2_tiled_words(n, count) words (lde ignored; pad samples are zero). */
int gf2_sample_errors_dev(gf2_ctx* ctx, int64_t n, uint64_t seed, int64_t first_sample, int64_t count,
                          double p_x, double p_y, double p_z,
                          uint64_t* ex_dev, uint64_t* ez_dev, int64_t lde, int layout);

/* Full pipeline: sample -> syndromes -> histograms, chunked through device workspace owned by ctx.
 * hist_z (from H1 . e_z) and hist_x (from H2 . e_x) are host uint64 arrays, overwritten.  The same histograms whatever the
 * route: n <= 64: one fused kernel; mid-size checks: sampler + lane-per-sample kernel; n <= 4096 at sparse
 * rates with both checks in standard form: the record sampler (no packed rows) + gather / combine kernels of the LDS row-slab
 * pipeline; otherwise packed rows from the sampler through the sparse or the dense syndrome kernels.  The first call of a size
 * allocates the workspaces. */
int gf2_mc_run(gf2_ctx* ctx, const gf2_check* check_c1, const gf2_check* check_c2,
               uint64_t seed, int64_t first_sample, int64_t count,
               double p_x, double p_y, double p_z, int mode,
               uint64_t* hist_z, int64_t nbins_z, uint64_t* hist_x, int64_t nbins_x);

/* Table decode + logical-error tally [build-defined, SURVEY.md 8f item 1]: the classical content of
 * quil_classical_correct (css_code.py:649-685) and noisy_measure (css_code.py:599-646) applied to sampled errors.
 * Per sample: s_x = H2.e_x; if vec_to_int(s_x) is in the C2 syndrome table the correction is XOR-ed in, otherwise the
 * error is left unchanged (css_code.py:655-657); the logical Z measurement flips iff z_operator . residual_x is odd
 * (css_code.py:640-646).  Likewise for Z errors with H1, the C1 table and x_operator.  Small codes only (n <= 63,
 * r_1, r_2 <= 20).  table_c1 / table_c2: 2^r_1 / 2^r_2 host words indexed by vec_to_int(syndrome): the packed
 * correction, or ~0 where the table has no entry.  counts_out[5] = { logical X flips, logical Z flips, samples with
 * either flip, samples whose X syndrome has no table entry, samples whose Z syndrome has no table entry }. */
int gf2_mc_decode(gf2_ctx* ctx, const gf2_check* check_c1, const gf2_check* check_c2,
                  const uint64_t* table_c1, const uint64_t* table_c2, uint64_t x_operator, uint64_t z_operator,
                  uint64_t seed, int64_t first_sample, int64_t count, double p_x, double p_y, double p_z,
                  uint64_t* counts_out);

/* The same tally for codes of up to 128 qubits through hashed tables [SURVEY.md 8f item 1 for the mid-size codes of item 2]:
 * checks of up to 127 rows, tables given by their entries instead of 2^r dense words.  h1 / h2: packed rows (ld words each) of
 * parity_check_c1 / parity_check_c2; keys: vec_to_int(syndrome) of every table entry, one word each for r <= 63 and two (low
 * word first) beyond; corr: the entry's packed error, two words; x_operator / z_operator: two words each.  counts_out[5] as
 * above.  GF2_E_ARG if a key occurs twice. */
int gf2_mc_decode_hashed(gf2_ctx* ctx, int64_t n, int64_t ld, const uint64_t* h1, int64_t r1, const uint64_t* keys1,
                         const uint64_t* corr1, int64_t entries1, const uint64_t* h2, int64_t r2, const uint64_t* keys2,
                         const uint64_t* corr2, int64_t entries2, const uint64_t* x_operator, const uint64_t* z_operator,
                         uint64_t seed, int64_t first_sample, int64_t count, double p_x, double p_y, double p_z,
                         uint64_t* counts_out);

/* ---- multi-GPU: the histogram all-reduce -------------------------------------------------------------
 * [build-defined, SURVEY.md 8e]  The Monte-Carlo run shards by sample range (sample i = f(seed, i)); ranks never exchange
 * anything on the data path.  The one collective is the sum of the histograms -- keys as css_code.py:729, X errors against
 * parity_check_c2 and Z errors against parity_check_c1 (css_code.py:457-470) -- over RCCL (xGMI inside a node).  librccl is
 * loaded on first use; without it these calls fail with GF2_E_RCCL and nothing else is affected. */
typedef struct gf2_comm gf2_comm;
#define GF2_COMM_ID_BYTES 128
/* One process per GPU: one rank makes an id, every rank receives it out of band (the launcher's store) and joins with the
 * context whose device and stream the collective runs on.  Collective: returns when all `nranks` ranks have called it. */
int gf2_comm_unique_id(void* id_out, size_t bytes);
int gf2_comm_create(gf2_ctx* ctx, const void* id, int nranks, int rank, gf2_comm** comm_out);
/* One process, `count` contexts on `count` different devices (ncclCommInitAll). */
int gf2_comm_create_all(gf2_ctx* const* ctxs, int count, gf2_comm** comm_out);
int gf2_comm_size(const gf2_comm* comm, int* nranks_out, int* nlocal_out);
int gf2_comm_destroy(gf2_comm* comm);
/* In-place sum over all ranks of `nbins` uint64 bins in device memory; hist_dev[i] lives on the device of the i-th context the
 * communicator was created with (one entry for gf2_comm_create).  Enqueued on each context's stream behind whatever produced
 * the bins; synchronous at return. */
int gf2_hist_allreduce(gf2_comm* comm, uint64_t* const* hist_dev, int64_t nbins);
int gf2_rccl_version(int* version_out);

#ifdef __cplusplus
}
#endif
#endif /* GF2HIP_H */
