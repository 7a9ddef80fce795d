// Internal definitions shared by the translation units of libgf2hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "gf2hip.h"
#include "gf2_tuning.h"

#define GF2_VERSION_NUMBER 100

typedef unsigned long long u64;   // HIP's 64-bit atomics/intrinsics are declared on this type

// ---- error plumbing ------------------------------------------------------------------------------
void gf2_set_error(const char* fmt, ...);

#define GF2_FAIL(code, ...)          \
    do {                             \
        gf2_set_error(__VA_ARGS__);  \
        return (code);               \
    } while (0)

#define GF2_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t gf2_err_ = (expr);                                                         \
        if (gf2_err_ != hipSuccess) {                                                         \
            gf2_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(gf2_err_),        \
                          __FILE__, __LINE__);                                                \
            return GF2_E_HIP;                                                                 \
        }                                                                                     \
    } while (0)

#define GF2_TRY(expr)                 \
    do {                              \
        int gf2_rc_ = (expr);         \
        if (gf2_rc_ != GF2_OK) return gf2_rc_; \
    } while (0)

// ---- context -------------------------------------------------------------------------------------
struct gf2_ctx {
    int device;
    int num_cus;
    hipStream_t stream;
    hipStream_t side[2];          // side streams of the pipelined Monte-Carlo (gf2_mc_run): one per Pauli component
    hipStream_t hi;               // highest-priority stream: the look-ahead panels of the streamed RREF (gf2_elim.hip)
    hipEvent_t side_ev[7];        // sampled[2], done_z[2], done_x[2], start
    hipEvent_t t0, t1;            // gf2_timer_*
    // per-kernel-family profiling: ring of event pairs resolved lazily
    int profile_on;
    double prof_ms[GF2_K_COUNT];
    int64_t prof_launches[GF2_K_COUNT];
    static const int kProfSlots = 256;
    hipEvent_t prof_ev[kProfSlots][2];
    int prof_family[kProfSlots];
    int prof_used;
    // workspaces grown on demand: slot 0 = Monte-Carlo pipeline, slot 1 = re-tiling of sample-major errors,
    // slot 2 = records and partial weights of the LDS-slab sparse pipeline, slot 3 = the same for the second side stream
    void* ws[4];
    size_t ws_bytes[4];
    // large dynamic-LDS opt-in (hipFuncSetAttribute) done for this context's device: [0] syndrome_tiled_kernel,
    // [1] table_class_hash_kernel, [2] slab_gather_kernel, [3] conjugate_kernel, [4] rref_update_pair_kernel
    bool lds_optin[9];            // ... [5], [6] rref_sweep_update_kernel<2>, <4>, [7] sweep_finish_kernel, [8] rref_sweep_update_flat_kernel
    // routing flags (GF2_F_*) and tunables (GF2_OPT_*, -1 = default): gf2_ctx_set_flags / gf2_ctx_set_option
    uint32_t flags;
    int64_t opt[GF2_OPT_COUNT];
    // the sampler's two inverse-CDF tables (whole segments, last segment) in device memory, and what they were made for
    uint64_t* seg_cdf_dev;
    uint64_t seg_key_t;
    int seg_tail_cap;              // record sampler: qubits of a segment the lanes take in step (from the tables' rate; 0: all)
    int seg_key_nb;
};
static inline bool gf2_flag(const gf2_ctx* ctx, uint32_t f) { return (ctx->flags & f) != 0; }

int gf2_ctx_activate(gf2_ctx* ctx);
int gf2_ws_reserve(gf2_ctx* ctx, int slot, size_t bytes);
// Profiling brackets around one kernel launch of `family` (no-ops unless enabled).
int gf2_prof_begin(gf2_ctx* ctx, int family);
int gf2_prof_end(gf2_ctx* ctx);
int gf2_prof_drain(gf2_ctx* ctx);

// ---- prepared parity check -------------------------------------------------------------------------
// Rows are grouped in slabs of 64 (one output word per slab); columns in pairs of 128 (one 16-byte piece of
// the tiled error layout), each pair in 32 groups of 4.  For slab s only its "active" pairs are tabulated
// (pair_list / npairs); tables_dev holds for the t-th active pair of slab s, group j and nibble value v
//     tables[((s * max_pairs + t) * 32 + j) * 16 + v]  bit i  =  XOR_{c : v bit c} H[64 s + i][128 q + 4 j + c]
// with the columns of an identity block H[:, ident_off : ident_off + r] == I left out (taken from the error
// word directly by the kernel).
struct gf2_check {
    int64_t r, n, ld;
    int64_t slabs;        // ceil(r / 64)
    int64_t ldt;          // words per sample in the tiled layout (even)
    int64_t ident_off;    // -1: none
    int64_t max_pairs;    // stride of pair_list / tables per slab
    int small;            // n <= 64 and r <= 64: streaming kernels, rows_small
    uint64_t* h_dev;      // r x ld packed rows
    uint64_t* tables_dev;
    int32_t* pair_list_dev;
    int32_t* npairs_dev;
    // transposed check for the sparse-error kernel: (n + 1) columns of 64 * ht_k dwords (column n is zero, the
    // identity block's columns are zero); null when unsupported (small check, r > 8192)
    uint32_t* ht_dev;
    int ht_k;
    // row-slab tables of the LDS sparse pipeline (gf2_slabs.hip): nslabs512 x slab_cols entries of 64 bytes (512 rows of
    // one non-identity column; the last entry of a slab is zero); null when the check does not qualify
    void* slab_tab_dev;
    // column table of the lane-per-sample kernel (gf2_lane.hip): n entries of ceil(r/64) words; null unless 64 < n <= 512 or
    // 64 < r <= 256 fits (n <= 512, r <= 256, not `small`)
    void* lane_tab_dev;
    int slab_cols, slab_null, nslabs512;   // entries per row-part plane, index of a zero entry, row slabs
    uint64_t rows_small[64];
    // n <= 4096: bit j = column j of the check has a 1 somewhere (a column of zeros changes no syndrome: the slab pipeline neither
    // lists it nor sends the samples that have it through the redo pass)
    uint64_t col_any[64];
};

int gf2_stream_wait(hipStream_t stream);            // polls, then blocks (gf2_ctx.hip)
int gf2_build_columns(gf2_ctx* ctx, gf2_check* ck);
int gf2_build_slab_table(gf2_ctx* ctx, gf2_check* ck);
int gf2_build_lane_table(gf2_ctx* ctx, gf2_check* ck);
bool gf2_lane_ok(const gf2_check* ck);
int gf2_syndrome_lane(gf2_ctx* ctx, const gf2_check* ck, const uint64_t* e_dev, int64_t batch, int64_t lde, uint64_t* s_dev,
                      int64_t lds, uint64_t* hist_dev, hipStream_t stream);
bool gf2_slabs_ok(const gf2_check* ck);
// stream: the context's stream or one of its side streams; ws_slot: 2 or 3 (one per stream that may run concurrently)
// hist_dev (r + 1 bins, accumulated into) and / or s_dev (syndromes, lds words per sample): either may be null
int gf2_syndrome_slabs(gf2_ctx* ctx, const gf2_check* ck, const uint64_t* e_dev, int64_t batch, int64_t lde,
                       uint64_t* hist_dev, hipStream_t stream, int ws_slot, uint64_t* s_dev = nullptr, int64_t lds = 0);
int gf2_slabs_reserve(gf2_ctx* ctx, const gf2_check* ck, int64_t batch, int ws_slot);
// gf2_mc_run's path without packed rows: the sampler writes the records and the identity words itself (gf2_slabs.hip)
struct SegTables;
bool gf2_mc_records_ok(const gf2_check* c1, const gf2_check* c2);
size_t gf2_mc_records_bytes(int64_t n, int64_t pass);
int gf2_mc_records_run(gf2_ctx* ctx, const gf2_check* c1, const gf2_check* c2, uint64_t seed, int64_t first_sample, int64_t count,
                       int64_t pass, const SegTables& th, void* buf, uint64_t* hz_dev, uint64_t* hx_dev, hipStream_t stream);
bool gf2_mc_sparse_fused_ok(const gf2_check* c1, const gf2_check* c2);
int gf2_mc_sparse_fused(gf2_ctx* ctx, const gf2_check* c1, const gf2_check* c2, uint64_t seed, int64_t first_sample,
                        int64_t count, double p_x, double p_y, double p_z, uint64_t* hz_dev, uint64_t* hx_dev);

static inline int64_t gf2_words(int64_t bits) { return (bits + 63) >> 6; }
static inline int64_t gf2_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Quantised probability threshold in [0, 2^32] (DESIGN.md "Sampler").
static inline uint64_t gf2_quantise(double x) {
    double t = __builtin_floor(x * 4294967296.0 + 0.5);
    if (!(t > 0.0)) return 0;
    if (t >= 4294967296.0) return 4294967296ull;
    return (uint64_t)t;
}
