// GF(2) elimination on packed words (gfx950): reduced row echelon form, the reference's order-sensitive
// parity-check normalisation, nullspace, column swap and row weights.
//
//   bin_matrix.reduced_row_echelon_form   bin_matrix.py:8-34     -> eliminate_kernel<ELIM_RREF>
//   css_code.normalize_parity_check       css_code.py:809-836    -> eliminate_kernel<ELIM_NORMALIZE>
//   css_code.swap_columns                 css_code.py:783-785    -> swap_columns_kernel
//   css_code.is_doubly_even (row sums)    css_code.py:846-850    -> row_weights_kernel
//   nullspace [build-defined, x1]                                 -> nullspace_kernel
//
// One workgroup of 1024 lanes owns one matrix (a batch of matrices is one workgroup each); the matrix
// stays in global memory and, at the sizes of interest (1 MiB packed at 2048 x 4096), in the XCD's L2.
// Pivots are processed strictly one after another.  The RREF is unique, so for it the kernel is free to
// swap rows; the normalisation is not: there the kernel performs the reference's
// operations in the reference's order -- first odd row at or below the diagonal is XOR-ed into the
// diagonal row, otherwise the first odd column of the diagonal row's current state is swapped in -- so
// the column swaps and the result are bit-identical.
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "gf2_internal.h"

#define ELIM_RREF 0
#define ELIM_NORMALIZE 1
#define ELIM_THREADS 1024
#define ELIM_WAVES (ELIM_THREADS / 64)
#define ELIM_MAX_LD 2048          // pivot row staged in LDS: 16 KiB

#define ELIM_STATUS_OK 0
#define ELIM_STATUS_DEPENDENT 1

// State shared by the kernels of the blocked eliminations (one per matrix, in global scratch).
struct RrefState {                                  // per matrix, in global scratch
    int64_t rank;                                   // RREF: pivots so far.  normalisation: the next diagonal index
    int64_t first_free;                             // RREF: first column seen without a pivot
    int64_t skip_lo, skip_hi;                       // columns [skip_lo, skip_hi) cannot change in the current update
    int64_t zero_lo, zero_hi;                       // rows [zero_lo, zero_hi) are rebuilt from zero (normalisation)
    int32_t t;                                      // pivots of the current panel
    int32_t stalled;                                // normalisation: the panel stopped early, a single step must follow
    int32_t pending;                                // streamed RREF panel: the last round's table waits in tabs (panel_finish_kernel)
    int32_t tg[2];                                  // RREF: pivots of the two panels of the current pair (one trailing update per pair)
};

template <int MODE>
__global__ __launch_bounds__(ELIM_THREADS) void eliminate_kernel(u64* __restrict__ base, int64_t m, int64_t n,
                                                                 int64_t ld, int64_t offset, int64_t* __restrict__ pivots_base,
                                                                 int64_t pivots_stride, int64_t* __restrict__ rank_base,
                                                                 int64_t* __restrict__ swaps, int64_t* __restrict__ nswaps,
                                                                 int* __restrict__ status, RrefState* __restrict__ single) {
    // `single` (normalisation only): do exactly the step single->rank, and only if the blocked panel stalled there
    __shared__ u64 pivot_row[ELIM_MAX_LD];
    __shared__ int found;          // first row (phase A) / first column (swap search), or INT_MAX
    u64* a = base + (int64_t)blockIdx.x * m * ld;
    int64_t* pivots = pivots_base ? pivots_base + (int64_t)blockIdx.x * pivots_stride : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    int64_t lead = 0;              // RREF: next pivot row.  NORMALIZE: the diagonal index i.
    int64_t swap_count = 0;
    int64_t step_begin = 0, steps = MODE == ELIM_RREF ? n : m;
    if (single) {                  // stalled == 1: one step; stalled == 2: every remaining step
        if (!single->stalled || single->rank >= m || status[blockIdx.x] != 0) return;
        step_begin = single->rank;
        steps = single->stalled == 2 ? m : step_begin + 1;
        swap_count = nswaps[blockIdx.x];
    }
    for (int64_t step = step_begin; step < steps; ++step) {
        const int64_t col = MODE == ELIM_RREF ? step : step + offset;
        if (MODE == ELIM_RREF && lead >= m) break;
        if (MODE == ELIM_NORMALIZE) lead = step;
        const int64_t cw = col >> 6;
        const int cb = (int)(col & 63);

        // ---- A: first row >= lead with an odd entry in this column ------------------------------------------
        if (tid == 0) found = 0x7fffffff;
        __syncthreads();
        int first = 0x7fffffff;
        for (int64_t r0 = lead; r0 < m; r0 += ELIM_THREADS) {
            const int64_t row = r0 + tid;
            const bool odd = row < m && ((a[row * ld + cw] >> cb) & 1ull);
            const u64 mask = __ballot(odd);
            if (mask && lane == 0) atomicMin(&found, (int)(r0 + wave * 64 + __ffsll((long long)mask) - 1));
            __syncthreads();
            first = found;
            __syncthreads();                               // everyone has read `found` before it changes again
            if (first != 0x7fffffff) break;
        }
        const int64_t donor = first == 0x7fffffff ? -1 : first;

        // ---- B: bring a pivot to row `lead` ---------------------------------------------------------------------
        if (MODE == ELIM_RREF) {
            if (donor < 0) continue;                         // no pivot in this column
            if (donor != lead)
                for (int64_t w = tid; w < ld; w += ELIM_THREADS) {
                    const u64 x = a[lead * ld + w], y = a[donor * ld + w];
                    a[lead * ld + w] = y;
                    a[donor * ld + w] = x;
                }
            if (tid == 0 && pivots) pivots[lead] = col;
        } else if (donor >= 0) {
            if (donor != lead)                               // diagonal entry is even: add the donor row
                for (int64_t w = tid; w < ld; w += ELIM_THREADS) a[lead * ld + w] ^= a[donor * ld + w];
        } else {
            // no odd row: swap in the first odd column of the diagonal row (its state right now)
            if (tid == 0) found = 0x7fffffff;
            __syncthreads();
            for (int64_t w = cw + tid; w < ld; w += ELIM_THREADS) {
                u64 v = a[lead * ld + w];
                if (w == cw) v &= ~0ull << cb;
                if (v) atomicMin(&found, (int)(w * 64 + __ffsll((long long)v) - 1));
            }
            __syncthreads();
            const int64_t other = found;
            __syncthreads();
            if (other == 0x7fffffff) {
                if (tid == 0) {
                    status[blockIdx.x] = ELIM_STATUS_DEPENDENT;
                    if (nswaps) nswaps[blockIdx.x] = swap_count;
                }
                return;
            }
            if (tid == 0 && swaps) {
                swaps[2 * swap_count] = col;
                swaps[2 * swap_count + 1] = other;
            }
            swap_count += 1;
            const int64_t ow = other >> 6;
            const int ob = (int)(other & 63);
            for (int64_t row = tid; row < m; row += ELIM_THREADS) {
                u64 x = a[row * ld + cw], y = a[row * ld + ow];
                const u64 diff = ((x >> cb) ^ (y >> ob)) & 1ull;
                if (cw == ow) {
                    x ^= (diff << cb) | (diff << ob);
                    a[row * ld + cw] = x;
                } else {
                    a[row * ld + cw] = x ^ (diff << cb);
                    a[row * ld + ow] = y ^ (diff << ob);
                }
            }
        }
        __syncthreads();

        // ---- C: clear the column in every other row ---------------------------------------------------------------
        for (int64_t w = tid; w < ld; w += ELIM_THREADS) pivot_row[w] = a[lead * ld + w];
        __syncthreads();
        for (int64_t r0 = (int64_t)wave * 64; r0 < m; r0 += ELIM_THREADS) {
            const int64_t row = r0 + lane;
            const bool odd = row < m && row != lead && ((a[row * ld + cw] >> cb) & 1ull);
            u64 mask = __ballot(odd);
            while (mask) {
                const int k = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                u64* target = a + (r0 + k) * ld;
                for (int64_t w = lane; w < ld; w += 64) target[w] ^= pivot_row[w];
            }
        }
        __syncthreads();
        if (MODE == ELIM_RREF) lead += 1;
    }
    if (tid == 0) {
        if (MODE == ELIM_RREF && rank_base) rank_base[blockIdx.x] = lead;
        if (MODE == ELIM_NORMALIZE && nswaps) nswaps[blockIdx.x] = swap_count;
        if (single) {
            single->rank = steps;
            single->stalled = 0;
        }
    }
}

// ---- blocked RREF -----------------------------------------------------------------------------------------------------
//
// The reduced row echelon form is unique, so the elimination may pick any pivot row and move rows whenever it
// likes.  Rows stay in place and the matrix is processed in panels of 64 columns, two kernels per panel:
//
//   rref_panel_kernel (one workgroup per matrix)
//     1. panel factorisation.  A lane owns rows tid, tid+1024, .. and holds their panel words w_i and 64-bit
//        coefficients d_i (new_i = old_i ^ d_i . OLDPIV, OLDPIV = the chosen pivot rows as they stand at the start of
//        the panel).  Up to 128 unused rows with a bit in a still-unresolved panel column form a window in LDS; ONE wavefront
//        runs Gauss-Jordan on the window and on 64 probe rows e_j with __ballot / readlane only (no barriers: window_round).
//        Every other row is then finished with two byte-table lookups made from the probe rows (elimination is linear in the
//        row).  If a column found no pivot inside the window but rows outside it still carry the bit, another round follows;
//        for random matrices one round resolves 64 columns.
//     2. d_i and a snapshot of the OLDPIV rows go to global scratch.
//   rref_update_pair_kernel (grid: row blocks x 32-word column chunks x matrices -- the whole GPU), once per PAIR of panels
//     3. A[i] ^= dA_i . OLDPIV_A ^ dB_i . OLDPIV_B', Method of Four Russians: for each group of 4 pivots of either panel the 16
//        XOR combinations of their rows sit in LDS (2 x 64 KiB); a wavefront moves four rows per slot (16 lanes = 256
//        contiguous bytes each), a lookup is one SDWA instruction that drops a nibble of d into the address + one
//        ds_read_b128.  The second panel of a pair runs BEFORE the first panel's update has been applied: it brings its own
//        column up to date on the way in (one byte-table lookup per row) and the table build of this kernel does the same
//        for its pivot rows (OLDPIV_B' = OLDPIV_B ^ fix . OLDPIV_A), so the matrix makes one trip through HBM per 128
//        columns.  Chunks left of the pair are skipped while no pivot-free column has been seen there (they cannot change).
//        (The normalisation uses the same kernel with its panel as the first of a pair that has no second.)
//   gather_rows_kernel: after the last panel the pivot rows are gathered into rows 0..rank-1, the rest zeroed.
#define RB_THREADS 1024
#define RB_WIN 128


__device__ __forceinline__ u64 readlane64(u64 v, int src) {
    return ((u64)(unsigned int)__builtin_amdgcn_readlane((int)(v >> 32), src) << 32) |
           (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, src);
}

// 64 x 64 bit transpose across a wavefront: lane i holds row i, on return lane j holds column j (six butterfly stages).
template <int S>
__device__ __forceinline__ u64 transpose_stage(u64 x, int lane, u64 m) {
    const u64 y = __shfl_xor(x, S);
    return (lane & S) ? ((x & ~m) | ((y >> S) & m)) : ((x & m) | ((y << S) & ~m));
}
__device__ __forceinline__ u64 transpose64(u64 x, int lane) {
    x = transpose_stage<32>(x, lane, 0x00000000FFFFFFFFull);
    x = transpose_stage<16>(x, lane, 0x0000FFFF0000FFFFull);
    x = transpose_stage<8>(x, lane, 0x00FF00FF00FF00FFull);
    x = transpose_stage<4>(x, lane, 0x0F0F0F0F0F0F0F0Full);
    x = transpose_stage<2>(x, lane, 0x3333333333333333ull);
    x = transpose_stage<1>(x, lane, 0x5555555555555555ull);
    return x;
}

// One round of the panel factorisation on the window (wavefront 0 of the panel kernels), Gauss-Jordan inside one wavefront.
// In: win_w / win_d / win_row (nwin <= 128 rows: panel word, coefficients so far, row), the unresolved panel columns, t pivots so
// far.  Out: for every new pivot p its column and its row (pbit, prow_l); fin_w / fin_d / win_piv for the window rows that
// became pivots; DP / WP, the probe rows; misc = {pivots now, new columns lo, hi}.
//
// Coefficients are kept in their FINAL form: new_i = old_i ^ d_i . OLDPIV, OLDPIV_p = pivot p's row as it stands at the start of
// the panel.  The pivot row of p, when chosen, is P_p = OLDPIV_p ^ d_r . OLDPIV, so a row that takes P_p takes e_p ^ d_r into its
// coefficients -- there is no separate matrix V to invert afterwards.  64 PROBE rows e_0 .. e_63 are eliminated along: row e_j
// ends with the word and the coefficients any row gets for having bit j, and elimination is linear in the row, so every row
// outside the window is finished with two table lookups instead of replaying the pivots.
//
// Lane l holds window rows l and l + 64 and probe row e_l (word and coefficients in registers, as 32-bit halves).  A pivot step
// (round 5): the rows that have column b are per-lane masks (v_bfe_i32 of the half that holds the column); one ballot finds the
// first window row of the first 64 that has the bit and is no pivot yet (the other 64 only when there is none); its word and
// coefficients come by readlane; every row of the column but the pivot row itself takes them, x ^= P & mask, one v_bitop3 per
// dword.  ~125 ns per pivot, a dependent chain: 24 vector instructions, one branch per pivot and one TAKEN branch per four (a taken
// branch costs a lone wavefront ~15 ns).  (Before: three ballots as the lane masks of `if (bit) x ^= P`, which the compiler turned
// into selects -- 44 vector instructions, 170 ns.  Rounds 1 and 2 went through a row-sliced round with ballots over the rows and
// five readlanes per pivot, and a column-sliced one -- lane = panel column, twelve v_writelane per pivot -- at 325 and 280 ns;
// eliminating strips of 2, 4 or 8 columns as bit vectors in scalar registers first and applying their pivots afterwards was no
// faster: 0.99, 1.03, 1.13 ms for the 2048 x 4096 matrix against 0.98.)
__device__ __forceinline__ void window_round(int lane, int nwin, int t, int64_t rank, int64_t m, u64 unresolved, const u64* win_w,
                                             const u64* win_d, const int* win_row, int* win_piv, u64* fin_w, u64* fin_d,
                                             int* pbit, int* prow_l, u64* DP, u64* WP, int* misc, int* win_q = nullptr) {
    u64 w0 = lane < nwin ? win_w[lane] : 0ull, w1 = lane + 64 < nwin ? win_w[lane + 64] : 0ull, w2 = 1ull << lane;
    u64 d0 = (t > 0 && lane < nwin) ? win_d[lane] : 0ull;             // uniform: coefficients of earlier rounds
    u64 d1 = (t > 0 && lane + 64 < nwin) ? win_d[lane + 64] : 0ull, d2 = 0ull;
    u64 piv0 = 0, piv1 = 0, newbits = 0;                               // (uniform) window rows that are pivots; columns resolved here
    unsigned int my_pbit = 0, my_prow = 0;                             // lane p: pivot p's column and window row
    int tt = t;
    const int tmax = m - rank < 64 ? (int)(m - rank) : 64;             // (a 32-bit scalar compare per pivot: `rank + tt < m` was a 64-bit vector one)
    // The pivot step on 32-bit halves, with the rows that take the pivot row as a per-lane MASK (all ones / zero: one v_bfe_i32 of
    // the half that holds the column), so that a row's update is  x ^= P & mask  -- one v_bitop3 per dword with the pivot row's dword
    // in a scalar register -- and only the two ballots of the candidate search are left.  (Written as `if (inverse_ballot(c)) w ^= P`
    // the compiler selected 0 or P per lane first -- v_cndmask + v_xor, and a v_mov per scalar value: 44 vector instructions per pivot,
    // ~170 ns; now 24.)  The columns of the low half first, then those of the high half: the half is a compile-time constant.
    unsigned int w0l = (unsigned int)w0, w0h = (unsigned int)(w0 >> 32), w1l = (unsigned int)w1, w1h = (unsigned int)(w1 >> 32);
    unsigned int w2l = (unsigned int)w2, w2h = (unsigned int)(w2 >> 32);
    unsigned int d0l = (unsigned int)d0, d0h = (unsigned int)(d0 >> 32), d1l = (unsigned int)d1, d1h = (unsigned int)(d1 >> 32);
    unsigned int d2l = 0, d2h = 0;
    auto half_steps = [&](auto half_constant, unsigned int todo32) {
        constexpr int HALF = decltype(half_constant)::value;
        // W: the halves that hold this half's columns, X: the other halves
        unsigned int &W0 = HALF ? w0h : w0l, &W1 = HALF ? w1h : w1l, &W2 = HALF ? w2h : w2l;
        unsigned int &X0 = HALF ? w0l : w0h, &X1 = HALF ? w1l : w1h, &X2 = HALF ? w2l : w2h;
        auto note_pivot = [&](int b5, int r) {
            // lane tt keeps the pivot's column and window row (v_writelane_b32 with the lane in M0)
            asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0"
                         : "+v"(my_pbit), "+v"(my_prow)
                         : "s"(tt), "s"((unsigned int)(b5 + 32 * HALF)), "s"((unsigned int)r)
                         : "m0");
            newbits |= 1ull << (b5 + 32 * HALF);
        };
        todo32 = (unsigned int)__builtin_amdgcn_readfirstlane((int)todo32);     // (uniform, and now the compiler knows)
        if (tt >= tmax) todo32 = 0;                                    // (uniform)
        while (todo32) {
            int b5 = __ffs((int)todo32) - 1;
            todo32 &= todo32 - 1;
            int m0 = __builtin_amdgcn_sbfe((int)W0, b5, 1);            // all ones: the row has the column
            u64 cand0 = __ballot(m0 != 0) & ~piv0;
            if (__builtin_expect(cand0 != 0, 1)) {
                // The usual case, a pivot among the first 64 window rows, as a loop of its own with ONE branch per pivot (a branch
                // costs the lone wavefront ~10 ns): the halves that hold the columns are updated first, then the NEXT column's test
                // and ballot are issued, and the other nine updates run while that ballot is on its way to the scalar unit.
                unsigned int rest;
                int b5n, m0n;
                u64 cand0n;
                bool go;
                auto step = [&]() {
                    const int r = __ffsll((long long)cand0) - 1;
                    piv0 |= 1ull << r;
                    asm("v_writelane_b32 %0, 0, %1" : "+v"(m0) : "s"(r));           // the pivot row does not take itself (no builtin for v_writelane)
                    unsigned int Pl = (unsigned int)__builtin_amdgcn_readlane((int)w0l, r), Ph = (unsigned int)__builtin_amdgcn_readlane((int)w0h, r);
                    unsigned int Vl = (unsigned int)__builtin_amdgcn_readlane((int)d0l, r), Vh = (unsigned int)__builtin_amdgcn_readlane((int)d0h, r);
                    {
                        const u64 V = (((u64)Vh << 32) | Vl) ^ (1ull << tt);     // (scalar: a shift and an XOR of 64 bits)
                        Vl = (unsigned int)V, Vh = (unsigned int)(V >> 32);
                    }
                    const int m1 = __builtin_amdgcn_sbfe((int)W1, b5, 1), m2 = __builtin_amdgcn_sbfe((int)W2, b5, 1);
                    const unsigned int PW = HALF ? Ph : Pl, PX = HALF ? Pl : Ph;
                    // x ^= P & m: bitop3 truth table 0x78 = a ^ (b & c)
                    W0 = __builtin_amdgcn_bitop3_b32(W0, PW, (unsigned int)m0, 0x78);
                    W1 = __builtin_amdgcn_bitop3_b32(W1, PW, (unsigned int)m1, 0x78);
                    W2 = __builtin_amdgcn_bitop3_b32(W2, PW, (unsigned int)m2, 0x78);
                    __builtin_amdgcn_sched_barrier(0);
                    note_pivot(b5, r);
                    tt += 1;
                    rest = tt < tmax ? todo32 : 0u;                    // (a scalar select)
                    b5n = __builtin_ctz(rest | 0x80000000u);           // (no column left: any offset will do, `go` is false)
                    m0n = __builtin_amdgcn_sbfe((int)W0, b5n, 1);
                    cand0n = __ballot(m0n != 0) & ~piv0;
                    __builtin_amdgcn_sched_barrier(0);                 // (the next column's test stays ahead of the other nine updates)
                    X0 = __builtin_amdgcn_bitop3_b32(X0, PX, (unsigned int)m0, 0x78);
                    d0l = __builtin_amdgcn_bitop3_b32(d0l, Vl, (unsigned int)m0, 0x78), d0h = __builtin_amdgcn_bitop3_b32(d0h, Vh, (unsigned int)m0, 0x78);
                    X1 = __builtin_amdgcn_bitop3_b32(X1, PX, (unsigned int)m1, 0x78);
                    d1l = __builtin_amdgcn_bitop3_b32(d1l, Vl, (unsigned int)m1, 0x78), d1h = __builtin_amdgcn_bitop3_b32(d1h, Vh, (unsigned int)m1, 0x78);
                    X2 = __builtin_amdgcn_bitop3_b32(X2, PX, (unsigned int)m2, 0x78);
                    d2l = __builtin_amdgcn_bitop3_b32(d2l, Vl, (unsigned int)m2, 0x78), d2h = __builtin_amdgcn_bitop3_b32(d2h, Vh, (unsigned int)m2, 0x78);
                    todo32 = rest & (rest - 1u);
                    go = rest != 0 && cand0n != 0;
                    b5 = b5n, m0 = m0n, cand0 = cand0n;
                };
                // (four steps per backward branch: a TAKEN branch costs the lone wavefront ~15 ns, one that falls through next to nothing)
                for (;;) {
                    step();
                    if (!go) break;
                    step();
                    if (!go) break;
                    step();
                    if (!go) break;
                    step();
                    if (!go) break;
                }
                if (rest == 0) break;                                 // no column left (or no row)
                // column b5 has no pivot among the first 64 window rows: below
            }
            // the second half of the window (rare)
            const int m1 = __builtin_amdgcn_sbfe((int)W1, b5, 1);
            const u64 cand1 = __ballot(m1 != 0) & ~piv1;
            if (cand1) {
                int m1c = m1;
                int r = __ffsll((long long)cand1) - 1;
                piv1 |= 1ull << r;
                asm("v_writelane_b32 %0, 0, %1" : "+v"(m1c) : "s"(r));
                unsigned int Pl = (unsigned int)__builtin_amdgcn_readlane((int)w1l, r), Ph = (unsigned int)__builtin_amdgcn_readlane((int)w1h, r);
                unsigned int Vl = (unsigned int)__builtin_amdgcn_readlane((int)d1l, r), Vh = (unsigned int)__builtin_amdgcn_readlane((int)d1h, r);
                {
                    const u64 V = (((u64)Vh << 32) | Vl) ^ (1ull << tt);
                    Vl = (unsigned int)V, Vh = (unsigned int)(V >> 32);
                }
                const int m2 = __builtin_amdgcn_sbfe((int)W2, b5, 1);
                w0l = __builtin_amdgcn_bitop3_b32(w0l, Pl, (unsigned int)m0, 0x78), w0h = __builtin_amdgcn_bitop3_b32(w0h, Ph, (unsigned int)m0, 0x78);
                d0l = __builtin_amdgcn_bitop3_b32(d0l, Vl, (unsigned int)m0, 0x78), d0h = __builtin_amdgcn_bitop3_b32(d0h, Vh, (unsigned int)m0, 0x78);
                w1l = __builtin_amdgcn_bitop3_b32(w1l, Pl, (unsigned int)m1c, 0x78), w1h = __builtin_amdgcn_bitop3_b32(w1h, Ph, (unsigned int)m1c, 0x78);
                d1l = __builtin_amdgcn_bitop3_b32(d1l, Vl, (unsigned int)m1c, 0x78), d1h = __builtin_amdgcn_bitop3_b32(d1h, Vh, (unsigned int)m1c, 0x78);
                w2l = __builtin_amdgcn_bitop3_b32(w2l, Pl, (unsigned int)m2, 0x78), w2h = __builtin_amdgcn_bitop3_b32(w2h, Ph, (unsigned int)m2, 0x78);
                d2l = __builtin_amdgcn_bitop3_b32(d2l, Vl, (unsigned int)m2, 0x78), d2h = __builtin_amdgcn_bitop3_b32(d2h, Vh, (unsigned int)m2, 0x78);
                note_pivot(b5, r + 64);
                tt += 1;
                if (tt >= tmax) todo32 = 0;
            }
        }
    };
    half_steps(std::integral_constant<int, 0>{}, (unsigned int)unresolved);
    half_steps(std::integral_constant<int, 1>{}, (unsigned int)(unresolved >> 32));
    w0 = ((u64)w0h << 32) | w0l, w1 = ((u64)w1h << 32) | w1l, w2 = ((u64)w2h << 32) | w2l;
    d0 = ((u64)d0h << 32) | d0l, d1 = ((u64)d1h << 32) | d1l, d2 = ((u64)d2h << 32) | d2l;
    // the new pivot rows as they stand at the end of the round
    if ((piv0 >> lane) & 1ull) {
        fin_w[lane] = w0;
        fin_d[lane] = d0;
        win_piv[lane] = 1;
    }
    if ((piv1 >> lane) & 1ull) {
        fin_w[lane + 64] = w1;
        fin_d[lane + 64] = d1;
        win_piv[lane + 64] = 1;
    }
    if (lane >= t && lane < tt) {
        pbit[lane] = (int)my_pbit;
        prow_l[lane] = win_row[my_prow];
        if (win_q) win_q[my_prow] = lane;                              // (uniform) the window row's number among the panel's pivots
    }
    DP[lane] = d2;                                                     // row j: the coefficients a row takes for having bit j
    WP[lane] = w2;                                                     // row j: what becomes of bit j
    if (lane == 0) {
        misc[0] = tt;
        misc[1] = (int)(unsigned int)newbits;
        misc[2] = (int)(unsigned int)(newbits >> 32);
    }
}

// One byte table, T[g * 256 + v] = XOR of rows64[8 g + c] over the bits c of v (8 groups of 256 entries from 64 rows), by 128 lanes:
// lane q = (g, low nibble l) makes the combination of the group's low four rows that l names once and then walks the sixteen high
// nibbles in Gray-code order, one XOR and one store per entry -- 16 lanes of a group store 128 contiguous bytes, so the
// ds_write_b64 groups are conflict-free.  (Every entry made from scratch -- eight selects and XORs each, 2048 entries over the
// workgroup -- was 1.4 us per table in the panel kernels: more than the lookups it serves.)
__device__ __forceinline__ void gray_byte_table(int q, const u64* rows64, u64* T) {
    const int g = q >> 4, l = q & 15;
    const u64* r = rows64 + 8 * g;
    u64 cur = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) cur ^= r[c] & (0ull - (u64)((l >> c) & 1));
    u64 hi[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) hi[c] = r[4 + c];
    u64* out = T + g * 256 + l;
    out[0] = cur;
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        cur ^= hi[__builtin_ctz(i)];                                   // (compile-time index: the loop is unrolled)
        out[(i ^ (i >> 1)) * 16] = cur;                                // entry (high nibble = Gray(i), low nibble l)
    }
}

// Byte tables of the probe rows: VT of their coefficients (DP), TW of their words (only when another round follows).
__device__ __forceinline__ void round_tables(int tid, const u64* CP, const u64* WP, u64* VT, u64* TW, bool again) {
    if (tid < 128)
        gray_byte_table(tid, CP, VT);
    else if (tid < 256 && again)
        gray_byte_table(tid - 128, WP, TW);
}

__device__ __forceinline__ u64 byte_lookup(const u64* T, u64 w) {
    u64 x = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) x ^= T[g * 256 + (int)((w >> (8 * g)) & 255ull)];
    return x;
}

template <int RPT>
__global__ __launch_bounds__(RB_THREADS) void rref_panel_kernel(u64* __restrict__ base, int64_t m, int64_t n, int64_t ld,
                                                               int64_t pw, int64_t* __restrict__ pivots_base, int64_t cap,
                                                               int32_t* __restrict__ pivrow_base, RrefState* __restrict__ states,
                                                               unsigned char* __restrict__ used_base, u64* __restrict__ d_base,
                                                               u64* __restrict__ snap_base, int member,
                                                               const u64* __restrict__ dprev_base, const u64* __restrict__ snapprev_base,
                                                               u64* __restrict__ fix_base) {
    // member: 0 = first panel of a pair (the matrix is up to date), 1 = second panel: the first panel's update has not been
    // applied yet (one trailing pass serves both), so this panel's column is brought up to date on the way in:
    // word ^= d_prev[row] . (column pw of the first panel's pivot-row snapshot), through a byte table
    __shared__ u64 VT[2048];                                            // byte tables of the probe rows' coefficients (per round)
    __shared__ u64 TW[2048];                                            // byte tables of the probe rows' words (rounds that are followed by another)
    __shared__ u64 win_w[RB_WIN], win_d[RB_WIN], fin_w[RB_WIN], fin_d[RB_WIN], DP[64], WP[64];
    __shared__ int win_row[RB_WIN], win_piv[RB_WIN], pbit[64], prow_l[64], wave_tot[RB_THREADS / 64], misc[4];

    const int64_t mat = blockIdx.x;
    u64* a = base + mat * m * ld;
    RrefState* st = states + mat;
    unsigned char* used = used_base + mat * m;
    u64* dout = d_base + mat * m;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t rank = st->rank;
    int64_t first_free = st->first_free;
    if (rank >= m || pw * 64 >= n) {                                   // nothing left to do for this matrix
        if (tid == 0) {
            st->t = 0;
            st->tg[member] = 0;
            if (member == 0) st->tg[1] = 0;
        }
        return;
    }
    int64_t* pivots = pivots_base ? pivots_base + mat * cap : nullptr;
    int32_t* pivrow = pivrow_base + mat * cap;
    const int t_prev = member ? st->tg[0] : 0;
    const u64* dprev = dprev_base + mat * m;

    // ---- 1. panel factorisation ------------------------------------------------------------------------------------------
    u64 w[RPT], d[RPT];                                                 // panel word and coefficients (new = old ^ d . OLDPIV) of this lane's rows
    int slot[RPT];
    unsigned int usedmask = 0, usedmask0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int64_t row = tid + (int64_t)RB_THREADS * k;
        w[k] = row < m ? a[row * ld + pw] : 0ull;
        d[k] = 0;
        if (row < m && used[row]) usedmask |= 1u << k;
    }
    if (t_prev > 0) {                                                  // uniform: the pair's first panel left an update behind
        if (tid < 64) DP[tid] = tid < t_prev ? snapprev_base[(mat * 64 + tid) * ld + pw] : 0ull;
        __syncthreads();
        for (int idx = tid; idx < 2048; idx += RB_THREADS) {
            const int g = idx >> 8, vv = idx & 255;
            u64 x = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) x ^= DP[8 * g + k] & (0ull - (u64)((vv >> k) & 1));
            TW[idx] = x;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int64_t row = tid + (int64_t)RB_THREADS * k;
            if (row < m) w[k] ^= byte_lookup(TW, dprev[row]);
        }
        __syncthreads();
    }
    usedmask0 = usedmask;
    const int64_t cols_here = n - pw * 64;
    const u64 panel_cols = cols_here >= 64 ? ~0ull : ((1ull << cols_here) - 1ull);
    u64 unresolved = panel_cols;
    int t = 0;
    while (unresolved && t < 64 && rank + t < m) {
        // window: the first (up to) RB_WIN unused rows with a bit in an unresolved column
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            slot[k] = -1;
            if (!((usedmask >> k) & 1u) && (w[k] & unresolved)) cnt += 1;
        }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < RB_THREADS / 64; ++i) {
            const int v = wave_tot[i];
            if (i < wave) before += v;
            total += v;
        }
        if (total == 0) break;                                        // the unresolved columns have no pivot
        int pos = before + incl - cnt;
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (!((usedmask >> k) & 1u) && (w[k] & unresolved)) {
                if (pos < RB_WIN) {
                    slot[k] = pos;
                    win_row[pos] = tid + RB_THREADS * k;
                    win_w[pos] = w[k];
                    win_d[pos] = d[k];
                    win_piv[pos] = 0;
                }
                pos += 1;
            }
        __syncthreads();
        const int nwin = total < RB_WIN ? total : RB_WIN;
        if (wave == 0)
            window_round(lane, nwin, t, rank, m, unresolved, win_w, win_d, win_row, win_piv, fin_w, fin_d, pbit, prow_l, DP, WP, misc);
        __syncthreads();
        const int t_new = misc[0];
        const u64 newbits = ((u64)(unsigned int)misc[2] << 32) | (unsigned int)misc[1];
        // another round may follow (uniform): only then are the rows' words needed again
        const bool again = (unresolved & ~newbits) != 0 && t_new < 64 && rank + t_new < m;
        round_tables(tid, DP, WP, VT, TW, again);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            if (slot[k] >= 0 && win_piv[slot[k]]) {                   // a new pivot row: as the wavefront left it
                w[k] = fin_w[slot[k]];
                d[k] = fin_d[slot[k]];
                usedmask |= 1u << k;
            } else {                                                  // every other row: linear in its word
                const u64 w0 = w[k];
                d[k] ^= byte_lookup(VT, w0);
                if (again) w[k] = byte_lookup(TW, w0);
            }
        }
        unresolved &= ~newbits;
        t = t_new;
        __syncthreads();                                              // window arrays are reused by the next round
    }
    if (unresolved) {                                                 // whatever is left has no pivot
        const int64_t fc = pw * 64 + (__ffsll((long long)unresolved) - 1);
        if (fc < first_free) first_free = fc;
    }
    if (tid == 0) {
        st->t = t;
        st->tg[member] = t;
        st->rank = rank + t;
        if (member == 0) {                                            // the pair's update skips what its FIRST panel allows
            st->tg[1] = 0;
            st->skip_lo = 0;
            st->skip_hi = st->first_free < pw * 64 ? st->first_free : pw * 64;     // first_free as it was BEFORE this panel
        }
        st->first_free = first_free;
    }
    if (t == 0) return;
#pragma unroll
    for (int k = 0; k < RPT; ++k)
        if (((usedmask ^ usedmask0) >> k) & 1u) used[tid + RB_THREADS * k] = 1;
    // the update kernel brings this panel's pivot rows up to date before it uses them: what the first panel adds to them
    if (member && tid < t) fix_base[mat * 64 + tid] = t_prev > 0 ? dprev[prow_l[tid]] : 0ull;
    if (wave == 0 && lane < t) {
        // global pivot lists in ascending column order (a later round may have resolved an earlier column): the
        // position of a pivot is the number of resolved panel columns below its own
        const u64 resolved = panel_cols & ~unresolved;
        const int pos = __popcll(resolved & ((1ull << pbit[lane]) - 1ull));
        pivrow[rank + pos] = prow_l[lane];
        if (pivots) pivots[rank + pos] = pw * 64 + pbit[lane];
    }
    // snapshot of the pivot rows as they stand now (the update kernel overwrites them)
    u64* snap = snap_base + mat * 64 * ld;
    for (int64_t idx = tid; idx < (int64_t)t * ld; idx += RB_THREADS) {
        const int p = (int)(idx / ld);
        const int64_t wd = idx - (int64_t)p * ld;
        snap[idx] = a[(int64_t)prow_l[p] * ld + wd];
    }

    // ---- 2. the coefficients are final as they are (window_round keeps them in terms of the pivot rows' snapshot)
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int64_t row = tid + (int64_t)RB_THREADS * k;
        if (row < m) dout[row] = d[k];
    }
}


// A chore of the streamed panel step (m > 8192) that ONE workgroup is slow at and the whole chip does in microseconds: the
// panel's column (one word out of every row: strided by the row pitch) into the contiguous round-0 state.  (The other two --
// the rows' coefficients and the snapshot of the pivot rows -- are panel_finish_kernel.)
__global__ __launch_bounds__(256) void panel_column_kernel(const u64* __restrict__ base, int64_t m, int64_t ld, int64_t pw,
                                                           u64* __restrict__ wpan_base, u64* __restrict__ cco_base,
                                                           int32_t* __restrict__ slot_base, int member,
                                                           const RrefState* __restrict__ states, const u64* __restrict__ dprev_base,
                                                           const u64* __restrict__ snapprev_base) {
    __shared__ u64 colp[64], TP[2048];
    const int64_t mat = blockIdx.y, row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int t_prev = member ? states[mat].tg[0] : 0;
    if (t_prev > 0) {                                                  // second panel of a pair: see rref_panel_kernel
        const int tid = threadIdx.x;
        if (tid < 64) colp[tid] = tid < t_prev ? snapprev_base[(mat * 64 + tid) * ld + pw] : 0ull;
        __syncthreads();
        for (int idx = tid; idx < 2048; idx += 256) {
            const int g = idx >> 8, vv = idx & 255;
            u64 x = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) x ^= colp[8 * g + k] & (0ull - (u64)((vv >> k) & 1));
            TP[idx] = x;
        }
        __syncthreads();
    }
    if (row >= m) return;
    u64 w = base[(mat * m + row) * ld + pw];
    if (t_prev > 0) w ^= byte_lookup(TP, dprev_base[mat * m + row]);
    wpan_base[mat * m + row] = w;
    cco_base[mat * m + row] = 0;
    slot_base[mat * m + row] = -1;
}

// After the streamed panel kernel, one launch for its two leftovers: (1) the coefficients of every row the panel kernel has not
// settled itself (slot -2) take the last round's probe-row table applied to the row's word, one row per lane on as many
// workgroups as there are rows for (`do_coeff`); (2) the last 64 workgroups copy the panel's pivot rows (64 rows of ld words,
// words [w_lo, w_hi) but [hole_lo, hole_hi)) to the snapshot.  With look-ahead (launch_rref_blocked) the copy comes in two
// launches: the chunk that holds the pair's own columns at once, the rest when the previous pair's trailing pass has finished;
// by then the state's `t` is the other panel's, so the copy goes by tg[member].
__global__ __launch_bounds__(1024) void panel_finish_kernel(const u64* __restrict__ base, int64_t m, int64_t ld,
                                                            const RrefState* __restrict__ states, const u64* __restrict__ wpan_base,
                                                            const u64* __restrict__ cco_base, const int32_t* __restrict__ slot_base,
                                                            const u64* __restrict__ tabs_base, u64* __restrict__ d_base,
                                                            const int32_t* __restrict__ prow_base, u64* __restrict__ snap_base,
                                                            int member, int do_coeff, int64_t w_lo, int64_t w_hi, int64_t hole_lo,
                                                            int64_t hole_hi) {
    __shared__ u64 TC[2048];
    const int64_t mat = blockIdx.y;
    const RrefState st = states[mat];
    const int t = st.tg[member];
    if (t == 0) return;
    const int coeff_blocks = (int)gridDim.x - 64;
    if ((int)blockIdx.x >= coeff_blocks) {
        const int p = (int)blockIdx.x - coeff_blocks;
        if (p >= t) return;
        const u64* src = base + (mat * m + prow_base[mat * 64 + p]) * ld;
        u64* dst = snap_base + (mat * 64 + p) * ld;
        for (int64_t wd = w_lo + threadIdx.x; wd < w_hi; wd += blockDim.x)
            if (wd < hole_lo || wd >= hole_hi) dst[wd] = src[wd];
        return;
    }
    if (!do_coeff) return;
    for (int idx = threadIdx.x; idx < 2048; idx += 1024) TC[idx] = st.pending ? tabs_base[mat * 2048 + idx] : 0ull;
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (row >= m) return;
    u64 d = cco_base[mat * m + row];
    if (st.pending && slot_base[mat * m + row] != -2) d ^= byte_lookup(TC, wpan_base[mat * m + row]);
    d_base[mat * m + row] = d;
}

// Look-ahead: what the two panels of a pair could not copy while the previous pair's trailing pass was running -- their pivot
// rows outside the chunk [hole_lo, hole_hi) that was up to date already.  Workgroups 0 .. 63: first panel, 64 .. 127: second.
__global__ __launch_bounds__(1024) void panel_snapshot_rest_kernel(const u64* __restrict__ base, int64_t m, int64_t ld,
                                                                   const RrefState* __restrict__ states,
                                                                   const int32_t* __restrict__ prow_a, const int32_t* __restrict__ prow_b,
                                                                   u64* __restrict__ snap_a, u64* __restrict__ snap_b, int64_t hole_lo,
                                                                   int64_t hole_hi) {
    const int64_t mat = blockIdx.y;
    const int member = (int)blockIdx.x >> 6, p = (int)blockIdx.x & 63;
    if (p >= states[mat].tg[member]) return;
    const int32_t* prow = member ? prow_b : prow_a;
    const u64* src = base + (mat * m + prow[mat * 64 + p]) * ld;
    u64* dst = (member ? snap_b : snap_a) + (mat * 64 + p) * ld;
    for (int64_t wd = threadIdx.x; wd < ld; wd += blockDim.x)
        if (wd < hole_lo || wd >= hole_hi) dst[wd] = src[wd];
}

// The same panel step for matrices with more than 8192 rows: rows are streamed instead of held in registers.  The
// current panel word and coefficient of every row live in global scratch (wpan, cco); the window is filled through an
// LDS counter (any unused rows with a bit in an unresolved column will do -- the RREF does not depend on the choice).
__global__ __launch_bounds__(RB_THREADS) void rref_panel_stream_kernel(u64* __restrict__ base, int64_t m, int64_t n, int64_t ld,
                                                                      int64_t pw, int64_t* __restrict__ pivots_base, int64_t cap,
                                                                      int32_t* __restrict__ pivrow_base, RrefState* __restrict__ states,
                                                                      unsigned char* __restrict__ used_base, u64* __restrict__ d_base,
                                                                      int32_t* __restrict__ prow_base, u64* __restrict__ wpan_base,
                                                                      u64* __restrict__ cco_base, int32_t* __restrict__ slot_base,
                                                                      u64* __restrict__ tabs_base, int member,
                                                                      const u64* __restrict__ dprev_base, u64* __restrict__ fix_base) {
    __shared__ u64 VT[2048], TW[2048];
    __shared__ u64 win_w[RB_WIN], win_d[RB_WIN], fin_w[RB_WIN], fin_d[RB_WIN], DP[64], WP[64];
    __shared__ int win_row[RB_WIN], win_piv[RB_WIN], pbit[64], prow_l[64], misc[4];
    __shared__ int win_count;

    const int64_t mat = blockIdx.x;
    u64* a = base + mat * m * ld;
    RrefState* st = states + mat;
    unsigned char* used = used_base + mat * m;
    (void)d_base;
    u64* wpan = wpan_base + mat * m;
    u64* cco = cco_base + mat * m;
    int32_t* slot_of = slot_base + mat * m;                             // -1, or the row's window slot in this round
    (void)a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t rank = st->rank;
    int64_t first_free = st->first_free;
    if (rank >= m || pw * 64 >= n) {
        if (tid == 0) {
            st->t = 0;
            st->tg[member] = 0;
            if (member == 0) st->tg[1] = 0;
        }
        return;
    }
    const int t_prev = member ? st->tg[0] : 0;
    int64_t* pivots = pivots_base ? pivots_base + mat * cap : nullptr;
    int32_t* pivrow = pivrow_base + mat * cap;
    const int64_t cols_here = n - pw * 64;
    const u64 panel_cols = cols_here >= 64 ? ~0ull : ((1ull << cols_here) - 1ull);
    u64 unresolved = panel_cols;
    int t = 0, pending = 0;                                            // pending: the last round's rows are left to panel_finish_kernel
    // round 0 state (wpan = the panel's column, cco = 0, slot_of = -1): panel_column_kernel, launched before this one
    while (unresolved && t < 64 && rank + t < m) {
        if (tid == 0) win_count = 0;
        __syncthreads();
        for (int64_t r0 = 0; r0 < m; r0 += RB_THREADS) {                // fill the window; stop scanning once it is full
            const int64_t row = r0 + tid;
            if (row < m && !used[row] && (wpan[row] & unresolved)) {
                const int pos = atomicAdd(&win_count, 1);
                if (pos < RB_WIN) {
                    slot_of[row] = pos;
                    win_row[pos] = (int)row;
                    win_w[pos] = wpan[row];
                    win_d[pos] = cco[row];
                    win_piv[pos] = 0;
                }
            }
            __syncthreads();
            if (win_count >= RB_WIN) break;
        }
        const int total = win_count;
        if (total == 0) break;
        const int nwin = total < RB_WIN ? total : RB_WIN;
        if (wave == 0)
            window_round(lane, nwin, t, rank, m, unresolved, win_w, win_d, win_row, win_piv, fin_w, fin_d, pbit, prow_l, DP, WP, misc);
        __syncthreads();
        const int t_new = misc[0];
        const u64 newbits = ((u64)(unsigned int)misc[2] << 32) | (unsigned int)misc[1];
        const bool again = (unresolved & ~newbits) != 0 && t_new < 64 && rank + t_new < m;    // uniform: another round may follow
        round_tables(tid, DP, WP, VT, TW, again);
        __syncthreads();
        if (!again) {
            // the last round: only its pivot rows are settled here (coefficients final, marked -2); every other row takes its
            // coefficients from the probe-row table in panel_finish_kernel, on the whole chip instead of in this one workgroup
            if (tid < nwin && win_piv[tid]) {
                const int row = win_row[tid];
                cco[row] = fin_d[tid];
                used[row] = 1;
                slot_of[row] = -2;
            }
            for (int idx = tid; idx < 2048; idx += RB_THREADS) tabs_base[mat * 2048 + idx] = VT[idx];
            pending = 1;
            unresolved &= ~newbits;
            t = t_new;
            __syncthreads();
            break;
        }
        for (int64_t row = tid; row < m; row += RB_THREADS) {
            const int sl = slot_of[row];
            if (sl >= 0) slot_of[row] = -1;
            if (sl >= 0 && win_piv[sl]) {                             // a new pivot row: as the wavefront left it
                wpan[row] = fin_w[sl];
                cco[row] = fin_d[sl];
                used[row] = 1;
            } else {                                                  // every other row: linear in its word
                const u64 w0 = wpan[row];
                cco[row] ^= byte_lookup(VT, w0);
                wpan[row] = byte_lookup(TW, w0);
            }
        }
        unresolved &= ~newbits;
        t = t_new;
        __syncthreads();
    }
    if (unresolved) {
        const int64_t fc = pw * 64 + (__ffsll((long long)unresolved) - 1);
        if (fc < first_free) first_free = fc;
    }
    if (tid == 0) st->pending = pending;
    if (tid == 0) {
        st->t = t;
        st->tg[member] = t;
        st->rank = rank + t;
        if (member == 0) {
            st->tg[1] = 0;
            st->skip_lo = 0;
            st->skip_hi = st->first_free < pw * 64 ? st->first_free : pw * 64;
        }
        st->first_free = first_free;
    }
    if (t == 0) return;
    if (member && tid < t) fix_base[mat * 64 + tid] = t_prev > 0 ? dprev_base[mat * m + prow_l[tid]] : 0ull;
    if (wave == 0 && lane < t) {
        const u64 resolved = panel_cols & ~unresolved;
        const int pos = __popcll(resolved & ((1ull << pbit[lane]) - 1ull));
        pivrow[rank + pos] = prow_l[lane];
        if (pivots) pivots[rank + pos] = pw * 64 + pbit[lane];
    }
    if (tid < t) prow_base[mat * 64 + tid] = prow_l[tid];               // for panel_finish_kernel, which follows
}

// addr.byte1 = low / high nibble of byte B of d, the other bytes of addr kept (SDWA): one instruction turns the table's base
// address into the address of entry (nibble) -- entries are 256 bytes apart, byte 0 holds the lane's word offset.
template <int B>
__device__ __forceinline__ void nibble_lo_to_byte1(unsigned int& addr, unsigned int d, unsigned int c0f) {
    if constexpr (B == 0) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(addr) : "v"(d), "v"(c0f));
    if constexpr (B == 1) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(addr) : "v"(d), "v"(c0f));
    if constexpr (B == 2) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:DWORD" : "+v"(addr) : "v"(d), "v"(c0f));
    if constexpr (B == 3) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(addr) : "v"(d), "v"(c0f));
}
template <int B>
__device__ __forceinline__ void nibble_hi_to_byte1(unsigned int& addr, unsigned int d) {
    if constexpr (B == 0) asm("v_lshrrev_b32_sdwa %0, 4, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_0" : "+v"(addr) : "v"(d));
    if constexpr (B == 1) asm("v_lshrrev_b32_sdwa %0, 4, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_1" : "+v"(addr) : "v"(d));
    if constexpr (B == 2) asm("v_lshrrev_b32_sdwa %0, 4, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2" : "+v"(addr) : "v"(d));
    if constexpr (B == 3) asm("v_lshrrev_b32_sdwa %0, 4, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_3" : "+v"(addr) : "v"(d));
}

// One byte (two groups of 4 pivots) of both panels' coefficients for two slots (a lane's two words of two rows): 8 table reads of
// 16 bytes issued together, then the XORs.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <int BYTE>
__device__ __forceinline__ void pair_lookups2(u32x4_t* x, const u64* dA, const u64* dB, unsigned int* pas, unsigned int* pbs, unsigned int c0f) {
    typedef const __attribute__((address_space(3))) u32x4_t* lds_v4_ptr;
    constexpr unsigned int off = (unsigned int)BYTE * 8192u;          // groups 2*BYTE and 2*BYTE + 1: 4096 bytes each
    u32x4_t t[8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned int a = BYTE < 4 ? (unsigned int)dA[u] : (unsigned int)(dA[u] >> 32);
        const unsigned int b = BYTE < 4 ? (unsigned int)dB[u] : (unsigned int)(dB[u] >> 32);
        unsigned int pa = pas[u], pb = pbs[u];                         // byte 1 is rewritten by every lookup, the rest stays
        nibble_lo_to_byte1<BYTE & 3>(pa, a, c0f);
        t[4 * u] = *(lds_v4_ptr)(uintptr_t)(pa + off);
        nibble_hi_to_byte1<BYTE & 3>(pa, a);
        t[4 * u + 1] = *(lds_v4_ptr)(uintptr_t)(pa + off + 4096u);
        nibble_lo_to_byte1<BYTE & 3>(pb, b, c0f);
        t[4 * u + 2] = *(lds_v4_ptr)(uintptr_t)(pb + off);
        nibble_hi_to_byte1<BYTE & 3>(pb, b);
        t[4 * u + 3] = *(lds_v4_ptr)(uintptr_t)(pb + off + 4096u);
        pas[u] = pa, pbs[u] = pb;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        u32x4_t r;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned int y = __builtin_amdgcn_bitop3_b32(x[u][c], t[4 * u][c], t[4 * u + 1][c], 0x96);
            r[c] = __builtin_amdgcn_bitop3_b32(y, t[4 * u + 2][c], t[4 * u + 3][c], 0x96);
        }
        // pins the XORs here: they have no side effects, and instruction selection otherwise places all of a row block's after
        // its last read, with every table word spilled in between
        asm volatile("" : "+v"(r));
        x[u] = r;
    }
}

// The RREF's trailing update, one pass of the matrix for a PAIR of panels (grid: row blocks, column chunks of 32 words,
// matrices; block 1024; 128 KiB of dynamic LDS: two Four-Russians tables of 16 groups x 16 entries x 32 words).
//   new_i = old_i ^ dA_i . SA ^ dB_i . SB',   SB'_q = SB_q ^ fix_q . SA
// SA: the first panel's pivot rows as they stood before it; SB: the second panel's pivot rows as they stand in memory, that
// is without the first panel's update, which the table build adds (fix_q = dA of that row).  A wavefront moves four rows
// per slot, 16 lanes = 256 contiguous bytes each; a lookup is one SDWA instruction (a nibble of d into byte 1 of the address) + one
// ds_read_b128 with an immediate offset.
#define U2_CW 32
__global__ __launch_bounds__(RB_THREADS) void rref_update_pair_kernel(u64* base, int64_t m, int64_t ld,
                                                                     int64_t rows_per_wg, const RrefState* __restrict__ states,
                                                                     const u64* __restrict__ da_base, const u64* __restrict__ db_base,
                                                                     const u64* __restrict__ snapa_base, const u64* __restrict__ snapb_base,
                                                                     const u64* __restrict__ fix_base, int chunk_base, int chunk_skip,
                                                                     u64* out_base) {
    // out_base: where the rows go.  The matrix itself (in place), or another buffer of the same shape: then EVERY word is written,
    // changed or not (the blocked RREF's first pass moves the batch into the workspace this way, so that the row gather at the end
    // can write straight into the caller's buffer: one copy of the batch less).
    extern __shared__ __attribute__((aligned(16))) u64 T[];           // [2][16 groups][16 entries][32 words]
    const int64_t mat = blockIdx.z;
    const RrefState st = states[mat];
    const int ta = st.tg[0], tb = st.tg[1];
    const bool moving = out_base != base;                              // uniform
    // the launch covers chunks chunk_base .. chunk_base + gridDim.y - 1 but chunk_skip (look-ahead: the chunk of the next pair's
    // columns goes first, in a launch of its own)
    if ((int)blockIdx.y + chunk_base == chunk_skip) return;
    const int64_t cw0 = ((int64_t)blockIdx.y + chunk_base) * U2_CW;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc_n = ld - cw0 < U2_CW ? (int)(ld - cw0) : U2_CW;
    if ((ta | tb) == 0 || (cw0 * 64 >= st.skip_lo && (cw0 + U2_CW) * 64 <= st.skip_hi)) {
        if (moving) {                                                  // nothing to add here: the rows move as they are
            const int64_t r_lo = (int64_t)blockIdx.x * rows_per_wg;
            const int64_t r_hi = r_lo + rows_per_wg < m ? r_lo + rows_per_wg : m;
            const u64* src = base + mat * m * ld;
            u64* dst = out_base + mat * m * ld;
            for (int64_t idx = tid; idx < (r_hi - r_lo) * U2_CW; idx += RB_THREADS) {
                const int64_t row = r_lo + idx / U2_CW;
                const int wd = (int)(idx % U2_CW);
                if (wd < wc_n) dst[row * ld + cw0 + wd] = src[row * ld + cw0 + wd];
            }
        }
        return;
    }
    typedef const __attribute__((address_space(3))) u64* lds_u64_ptr;
    // 16 lookups of one table, entry (g, nibble g of d) at byte (g*16 + nibble)*256 + word*8 from `at`
    auto lookup16 = [](unsigned int at, u64 d) -> u64 {
        u64 x = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const unsigned int nib = (unsigned int)(d >> (4 * g)) & 15u;
            x ^= *(lds_u64_ptr)(uintptr_t)(at + nib * 256u + (unsigned int)g * 4096u);
        }
        return x;
    };
    // the 16 XOR combinations of a group's four rows: a lane takes (group, word) and the entries with bit 3 clear or set, reads the
    // four single rows and writes its eight entries (4 LDS reads and 8 writes per lane instead of 32 reads)
    auto combos = [&](u64* tab) {
        const int wd = tid & (U2_CW - 1), g = (tid / U2_CW) & 15, top = tid / (16 * U2_CW);
        u64* e = tab + (g * 16) * U2_CW + wd;
        const u64 r0 = e[1 * U2_CW], r1 = e[2 * U2_CW], r2 = e[4 * U2_CW], r3 = top ? e[8 * U2_CW] : 0ull;
        const u64 c3 = r3, c13 = r0 ^ r3, c23 = r1 ^ r3, c123 = r0 ^ r1 ^ r3;
        u64* o = e + (top ? 8 * U2_CW : 0);
        if (!top) o[0] = 0ull;                                         // entry 8 is a single row and stays
        if (top) o[1 * U2_CW] = c13, o[2 * U2_CW] = c23;               // entries 1, 2, 4 are single rows
        o[3 * U2_CW] = c123;
        if (top) o[4 * U2_CW] = r2 ^ c3;
        o[5 * U2_CW] = r2 ^ c13;
        o[6 * U2_CW] = r2 ^ c23;
        o[7 * U2_CW] = r2 ^ c123;
    };
    u64* a = base + mat * m * ld;
    u64* a_out = out_base + mat * m * ld;
    const u64* da = da_base + mat * m;
    const u64* db = db_base + mat * m;
    // A lane moves two words (16 bytes) of a row, sixteen lanes a row's 256 bytes of the chunk, a wavefront four rows per slot: a
    // lookup is a ds_read_b128, half as many read and address instructions per word as with 8-byte lookups (the pass is bound by
    // issuing them: with its loads and stores removed it ran 13 % faster, no more).
    const int quarter = lane >> 4, hw = (lane & 15) * 2;               // row of the slot, first of this lane's two words
    const bool valid0 = hw < wc_n, valid1 = hw + 1 < wc_n;
    auto skippable = [&](int wd) { return (cw0 + wd) * 64 >= st.skip_lo && (cw0 + wd + 1) * 64 <= st.skip_hi; };
    const bool lane_live = valid0 && (moving || !(skippable(hw) && (!valid1 || skippable(hw + 1))));     // (a word that cannot change is XORed with zeros)
    const unsigned int at_a = (unsigned int)hw * 8u, at_b = at_a + 65536u, c0f = 0x0fu;
    const int64_t row_end = ((int64_t)blockIdx.x + 1) * rows_per_wg < m ? ((int64_t)blockIdx.x + 1) * rows_per_wg : m;
    constexpr int NW = RB_THREADS / 64;
    // addresses = a wavefront-uniform base (scalar registers) + one 32-bit lane offset that serves all rows of a lane:
    // per-lane 64-bit pointers would take dozens of registers
    const int rl = 4 * wave + quarter;                                 // this lane's row among the 64 of a slot
    const unsigned int lane_word = (unsigned int)rl * (unsigned int)ld + (unsigned int)hw;
    // Two slots (8 rows of a wavefront, 128 of the workgroup) are worked on while the next two are on their way from memory.
    constexpr int STEP = 4 * NW * 2;
    auto load2 = [&](int64_t rb, u32x4_t* x, u64* dA, u64* dB) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t rs = rb + (int64_t)u * 4 * NW;               // uniform
            const bool in = rs + rl < row_end;
            const u64* dau = da + rs;
            const u64* dbu = db + rs;
            const u64* au = a + rs * ld + cw0;
            dA[u] = (in && ta > 0) ? dau[(unsigned int)rl] : 0ull;
            dB[u] = (in && tb > 0) ? dbu[(unsigned int)rl] : 0ull;
            u32x4_t v = {0u, 0u, 0u, 0u};
            const bool from_zero = rs + rl >= st.zero_lo && rs + rl < st.zero_hi;   // normalisation: rebuilt rows start from 0
            if (in && lane_live && !from_zero) {                       // (not made to wait for d: rows with d = 0 are rare)
                if (valid1) {
                    v = *reinterpret_cast<const u32x4_t*>(au + lane_word);
                } else {
                    const u64 one = au[lane_word];
                    v[0] = (unsigned int)one, v[1] = (unsigned int)(one >> 32);
                }
            }
            x[u] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    u32x4_t x0[2], x1[2];
    u64 dA0[2], dB0[2], dA1[2], dB1[2];
    const int64_t rb0 = (int64_t)blockIdx.x * rows_per_wg;
    load2(rb0, x0, dA0, dB0);                                          // on their way while the tables are built
    u64* TA = T;
    u64* TB = T + 16 * 16 * U2_CW;
    u64 sb[2], fixv[2];
    {
        u64 sa[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {                               // 64 x 32 words / 1024 lanes; everything the build reads from memory now
            const int idx = tid + RB_THREADS * it, p = idx / U2_CW, wd = idx & (U2_CW - 1);
            sa[it] = (p < ta && wd < wc_n) ? snapa_base[(mat * 64 + p) * ld + cw0 + wd] : 0ull;
            sb[it] = (p < tb && wd < wc_n) ? snapb_base[(mat * 64 + p) * ld + cw0 + wd] : 0ull;
            fixv[it] = (ta > 0 && p < tb) ? fix_base[mat * 64 + p] : 0ull;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + RB_THREADS * it, p = idx / U2_CW, wd = idx & (U2_CW - 1);
            TA[((p >> 2) * 16 + (1 << (p & 3))) * U2_CW + wd] = sa[it];
        }
    }
    __syncthreads();
    combos(TA);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + RB_THREADS * it, p = idx / U2_CW, wd = idx & (U2_CW - 1);
        u64 x = sb[it];
        if (ta > 0 && p < tb) x ^= lookup16((unsigned int)wd * 8u, fixv[it]);
        TB[((p >> 2) * 16 + (1 << (p & 3))) * U2_CW + wd] = x;
    }
    __syncthreads();
    combos(TB);
    __syncthreads();
    auto work2 = [&](int64_t rb, u32x4_t* x, const u64* dA, const u64* dB) {
        unsigned int pas[2] = {at_a, at_a}, pbs[2] = {at_b, at_b};
        // (scheduling barriers: left alone, the scheduler hoists all the reads of a row block and spills)
#define GF2_PAIR_BYTE(B) pair_lookups2<B>(x, dA, dB, pas, pbs, c0f); __builtin_amdgcn_sched_barrier(0)
        GF2_PAIR_BYTE(0); GF2_PAIR_BYTE(1); GF2_PAIR_BYTE(2); GF2_PAIR_BYTE(3);
        GF2_PAIR_BYTE(4); GF2_PAIR_BYTE(5); GF2_PAIR_BYTE(6); GF2_PAIR_BYTE(7);
#undef GF2_PAIR_BYTE
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t rs = rb + (int64_t)u * 4 * NW;
            u64* au = a_out + rs * ld + cw0;
            if (((dA[u] | dB[u]) || (moving && rs + rl < row_end)) && lane_live) {
                if (valid1)
                    *reinterpret_cast<u32x4_t*>(au + lane_word) = x[u];
                else
                    au[lane_word] = ((u64)x[u][1] << 32) | x[u][0];
            }
        }
    };
    for (int64_t rb = rb0; rb < row_end; rb += 2 * STEP) {
        load2(rb + STEP, x1, dA1, dB1);
        work2(rb, x0, dA0, dB0);
        load2(rb + 2 * STEP, x0, dA0, dB0);
        if (rb + STEP < row_end) work2(rb + STEP, x1, dA1, dB1);       // uniform
    }
}

// ---- blocked RREF, K panels per sweep (round 5) ------------------------------------------------------------------------------
//
// The pair scheme above makes one trip of the matrix through HBM per 128 columns.  The same idea with K = 4 panels (256
// columns) per trip halves the trips; the Four-Russians lookups per trip double (their total stays), and the tables of four panels
// fit the same 128 KiB because a workgroup then owns 16 words (128 bytes) of a row instead of 32.  Per sweep, two kernels:
//
//   rref_sweep_panel_kernel<K, RPT> (one workgroup per matrix): the K panel factorisations one after another in ONE launch,
//     RIGHT-LOOKING: a lane holds the K column words of its rows and, after panel l, brings the later columns up to date
//     (w_j ^= d_l . column j of panel l's pivot rows as they stand -- the pivot rows' own w_j -- one byte-table lookup), and it
//     keeps every row's coefficients FOLDED onto the pivot rows as they stand in MEMORY (e_{l2} ^= d_l . E_{l,l2}, E's rows = e_{l2}
//     of panel l's pivot rows), so that the trailing pass needs no correction of the second to fourth panel's pivot rows.  The K
//     column words of every row come from a compact side buffer (8 K bytes per row, written by the previous sweep's trailing pass)
//     instead of one 128-byte line per row and word; without one (first sweep, a sweep after one without pivots) from the rows.
//   rref_sweep_update_kernel<K> (grid: row blocks x chunks of 64 / K words x matrices): new_i = old_i ^ sum_l e_{l,i} . S_l with K
//     Four-Russians tables made of the pivot-row snapshots as they are; writes the next sweep's K column words of its rows to the
//     side buffer on the way out.  A sweep with at most 8 pivots (the sweeps after the one that reached full rank) takes the pivot
//     rows one by one instead.
//
// Built with -DGF2_SWEEP_DIAG=1 (profiles/r05_diag.sh) the two kernels stamp their phases with the shader clock and the constant
// 100 MHz clock; with GF2_RREF_DIAG=1 in the environment the launcher prints the averages: where the numbers in DESIGN.md come from.
#ifndef GF2_SWEEP_DIAG
#define GF2_SWEEP_DIAG 0
#endif
#if GF2_SWEEP_DIAG
__device__ u64 g_sweep_diag[8];                     // cycles / wall ticks of the trailing pass' phases (wavefront 0 of every workgroup)
__device__ u64 g_panel_diag[8];                     // wall ticks of the panel kernel's phases (workgroup 0)
__device__ u64 g_sweep_wg[4 * 4096];                // per workgroup of ONE pass launch: start, tables done, end (wall ticks), hardware id
#endif
struct SweepState {                                 // per matrix, in global scratch
    int64_t rank;                                   // pivots so far
    int64_t first_free;                             // first column seen without a pivot
    int64_t skip_hi;                                // words wholly below this column cannot change in the current sweep
    int64_t colw_pw;                                // the side buffer holds words colw_pw .. colw_pw + K - 1 as they stand (-1: nothing)
    int32_t tg[4];                                  // pivots of the sweep's panels
    int32_t pending, scan_lo;                       // streamed panels: the last round's table waits in `tabs` for sweep_finish_kernel;
                                                    // every row below scan_lo is a pivot row already (where the window fill starts)
};

__device__ __forceinline__ void byte_table(int tid, const u64* rows64, u64* T) {
    if (tid < 128) gray_byte_table(tid, rows64, T);
}

template <int K, int RPT>
__global__ __launch_bounds__(RB_THREADS) void rref_sweep_panel_kernel(const u64* __restrict__ base, int64_t m, int64_t n, int64_t ld,
                                                                     int64_t pw0, int64_t* __restrict__ pivots_base, int64_t cap,
                                                                     int32_t* __restrict__ pivrow_base, SweepState* __restrict__ states,
                                                                     unsigned char* __restrict__ used_base,
                                                                     const u64* __restrict__ colw_base, u64* __restrict__ d_base,
                                                                     int64_t dstride, u64* __restrict__ snap_base, int64_t sstride,
                                                                     int32_t* __restrict__ prow_out) {
    // d_base: [K][dstride] coefficients (dstride >= batch * m); snap_base: [K][sstride] pivot-row snapshots (batch x 64 x ld each),
    // or nullptr: no snapshots, the pivot rows' numbers go to prow_out [batch][K][64] and the pass reads the rows themselves
    constexpr int NT = K - 1 < 2 ? 2 : K - 1;                           // byte tables held at a time
    __shared__ u64 TT[NT * 2048];
    __shared__ u64 win_w[RB_WIN], win_d[RB_WIN], fin_w[RB_WIN], fin_d[RB_WIN], DP[64], WP[64];
    __shared__ u64 pub[K - 1][64];                                      // what a panel's new pivot rows show the other rows (see below)
    __shared__ int win_row[RB_WIN], win_piv[RB_WIN], win_q[RB_WIN], pbit[64], prow_l[64], misc[4];
    __shared__ int win_count;
    __shared__ int prow_all[K][64], tj[K];

    const int64_t mat = blockIdx.x;
#if GF2_SWEEP_DIAG
    u64 stamp_prev = wall_clock64();
#define GF2_STAMP(i)                                                  \
    do {                                                              \
        if (threadIdx.x == 0 && blockIdx.x == 0) {                    \
            const u64 now_ = wall_clock64();                          \
            atomicAdd(&g_panel_diag[i], now_ - stamp_prev);           \
            stamp_prev = now_;                                        \
        }                                                             \
    } while (0)
#else
#define GF2_STAMP(i) do { } while (0)
#endif
    const u64* a = base + mat * m * ld;
    SweepState* st = states + mat;
    unsigned char* used = used_base + mat * m;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t rank0 = st->rank;
    int64_t first_free = st->first_free;
    if (rank0 >= m || pw0 * 64 >= n) {                                 // nothing left to do for this matrix
        if (tid < K) st->tg[tid] = 0;
        return;
    }
    const bool have_colw = colw_base != nullptr && st->colw_pw == pw0;  // uniform
    const u64* colw = colw_base + mat * m * K;
    int64_t* pivots = pivots_base ? pivots_base + mat * cap : nullptr;
    int32_t* pivrow = pivrow_base + mat * cap;

    // The K column words of this lane's rows, kept up to date as the sweep's panels are factorised one after another
    // (right-looking): after panel l every row takes, for each later column j,  w_j ^= d_l . (column j of panel l's pivot rows AS
    // THEY STAND, i.e. those rows' own w_j) -- one byte-table lookup.  The coefficients are kept FOLDED onto the pivot rows as they
    // stand in memory: panel l's pivot row q has itself taken e_{l2,q} . S_{l2} from the earlier panels l2 < l (S: memory state), so
    // a row that takes d_l . (current pivot rows of l) takes d_l . S_l and, for every l2 < l, (d_l . E_{l,l2}) . S_{l2}, where row q
    // of the 64 x 64 matrix E_{l,l2} is e_{l2} of pivot row q:  e_{l2} ^= d_l . E_{l,l2}, another byte-table lookup.  The trailing
    // pass then is  new_i = old_i ^ sum_l e_{l,i} . S_l  with tables made of pivot rows as the snapshot holds them, nothing to fix.
    u64 w_all[K][RPT], e_all[K][RPT];
    unsigned int usedmask = 0, usedmask0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int64_t row = tid + (int64_t)RB_THREADS * k;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            w_all[j][k] = (row < m && (pw0 + j) * 64 < n) ? (have_colw ? colw[row * K + j] : a[row * ld + pw0 + j]) : 0ull;
            e_all[j][k] = 0;
        }
        if (row < m && used[row]) usedmask |= 1u << k;
    }
    usedmask0 = usedmask;
    GF2_STAMP(0);                                                      // prologue: state, column words, used flags
    int64_t rank = rank0;
    const int64_t first_free_in = first_free;
    if (tid < K) tj[tid] = 0;
    if (tid == 0) win_count = 0;
    int t_sum = 0;
#pragma unroll
    for (int l = 0; l < K; ++l) {
        const int64_t pw = pw0 + l;
        __syncthreads();                                               // (a panel that found nothing left its last round without one)
        if (rank >= m || pw * 64 >= n) continue;                       // uniform: nothing left for this panel
        if (tid < 64 * (K - 1)) pub[tid >> 6][tid & 63] = 0ull;         // (read after the rounds' barriers)
        // ---- panel factorisation on column l (as rref_panel_kernel) ------------------------------------------------------------
        u64 w[RPT], d[RPT];
        int slot[RPT], qidx[RPT];
        unsigned int fresh = 0;                                         // rows of this lane that became pivots of this panel
#pragma unroll
        for (int k = 0; k < RPT; ++k) w[k] = w_all[l][k], d[k] = 0, qidx[k] = 0;
        const int64_t cols_here = n - pw * 64;
        const u64 panel_cols = cols_here >= 64 ? ~0ull : ((1ull << cols_here) - 1ull);
        u64 unresolved = panel_cols;
        int t = 0;
        while (unresolved && t < 64 && rank + t < m) {
            // window: up to RB_WIN unused rows with a bit in an unresolved column, in any order (the RREF does not depend on which rows
            // become the pivots).  A wavefront counts its candidates with one ballot per register row and reserves its places with
            // ONE LDS atomic: the inclusive scan over the lanes and the sum of the wavefronts' totals behind a barrier that this
            // replaces were six ds_bpermute and a barrier per round.
            u64 cb[RPT];
            int wave_cnt = 0;
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                slot[k] = -1;
                cb[k] = __ballot(!((usedmask >> k) & 1u) && (w[k] & unresolved) != 0);
                wave_cnt += __popcll(cb[k]);
            }
            int base_pos = 0;
            if (wave_cnt > 0) {                                        // uniform per wavefront
                if (lane == 0) base_pos = atomicAdd(&win_count, wave_cnt);
                base_pos = __builtin_amdgcn_readfirstlane(base_pos);
            }
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                if ((cb[k] >> lane) & 1ull) {
                    const int pos = base_pos + __builtin_amdgcn_mbcnt_hi((unsigned int)(cb[k] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)cb[k], 0u));
                    if (pos < RB_WIN) {
                        slot[k] = pos;
                        win_row[pos] = tid + RB_THREADS * k;
                        win_w[pos] = w[k];
                        win_d[pos] = d[k];
                        win_piv[pos] = 0;
                    }
                }
                base_pos += __popcll(cb[k]);
            }
            __syncthreads();
            const int total = win_count;
            if (total == 0) break;                                    // the unresolved columns have no pivot (nobody added: the counter stays 0)
            __syncthreads();
            GF2_STAMP(1);                                              // window fill
            const int nwin = total < RB_WIN ? total : RB_WIN;
            if (wave == 0)
                window_round(lane, nwin, t, rank, m, unresolved, win_w, win_d, win_row, win_piv, fin_w, fin_d, pbit, prow_l, DP, WP, misc, win_q);
            __syncthreads();
            if (tid == 0) win_count = 0;                               // (everyone has read the total; the next round's atomics come two barriers later)
            GF2_STAMP(2);                                              // window_round
            const int t_new = misc[0];
            const u64 newbits = ((u64)(unsigned int)misc[2] << 32) | (unsigned int)misc[1];
            const bool again = (unresolved & ~newbits) != 0 && t_new < 64 && rank + t_new < m;   // uniform
            round_tables(tid, DP, WP, TT, TT + 2048, again);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                if (slot[k] >= 0 && win_piv[slot[k]]) {               // a new pivot row: as the wavefront left it
                    w[k] = fin_w[slot[k]];
                    d[k] = fin_d[slot[k]];
                    qidx[k] = win_q[slot[k]];
                    usedmask |= 1u << k;
                    fresh |= 1u << k;
                } else {                                              // every other row: linear in its word
                    const u64 w0 = w[k];
                    d[k] ^= byte_lookup(TT, w0);
                    if (again) w[k] = byte_lookup(TT + 2048, w0);
                }
            }
            unresolved &= ~newbits;
            t = t_new;
            __syncthreads();                                          // the window arrays and tables are reused
            GF2_STAMP(3);                                              // round tables + every row's lookups
        }
        if (unresolved) {                                             // whatever is left has no pivot
            const int64_t fc = pw * 64 + (__ffsll((long long)unresolved) - 1);
            if (fc < first_free) first_free = fc;
        }
        if (t == 0) continue;                                         // uniform
        if (tid == 0) tj[l] = t;
        t_sum += t;
        if (wave == 0 && lane < t) {
            // global pivot lists in ascending column order (a later round may have resolved an earlier column)
            const u64 resolved = panel_cols & ~unresolved;
            const int pos = __popcll(resolved & ((1ull << pbit[lane]) - 1ull));
            pivrow[rank + pos] = prow_l[lane];
            if (pivots) pivots[rank + pos] = pw * 64 + pbit[lane];
            prow_all[l][lane] = prow_l[lane];
        }
        rank += t;
        // ---- the new pivot rows show the other rows their later columns and their coefficients of the earlier panels ---------------
        // pub[v]: v < K - 1 - l: column l + 1 + v;  v >= K - 1 - l: coefficients of panel v - (K - 1 - l)
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if ((fresh >> k) & 1u) {
#pragma unroll
                for (int j = l + 1; j < K; ++j) pub[j - l - 1][qidx[k]] = w_all[j][k];
#pragma unroll
                for (int l2 = 0; l2 < l; ++l2) pub[K - 1 - l + l2][qidx[k]] = e_all[l2][k];
            }
        __syncthreads();
        // K - 1 byte tables at once, 128 lanes each
        if (tid < (K - 1) * 128) gray_byte_table(tid & 127, pub[tid >> 7], TT + (tid >> 7) * 2048);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const u64 dl = d[k];
            e_all[l][k] = dl;
#pragma unroll
            for (int j = l + 1; j < K; ++j) w_all[j][k] ^= byte_lookup(TT + (j - l - 1) * 2048, dl);
#pragma unroll
            for (int l2 = 0; l2 < l; ++l2) e_all[l2][k] ^= byte_lookup(TT + (K - 1 - l + l2) * 2048, dl);
        }
        GF2_STAMP(4);                                                  // pivot lists, publish, K - 1 tables, lookups
    }
    __syncthreads();
    if (tid < K) st->tg[tid] = tj[tid];
    if (tid == 0) {
        st->rank = rank;
        st->skip_hi = first_free_in < pw0 * 64 ? first_free_in : pw0 * 64;     // first_free as it was BEFORE this sweep
        st->first_free = first_free;
    }
    if (t_sum == 0) return;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int64_t row = tid + (int64_t)RB_THREADS * k;
        if (((usedmask ^ usedmask0) >> k) & 1u) used[row] = 1;
        if (row < m) {
#pragma unroll
            for (int l = 0; l < K; ++l) d_base[(int64_t)l * dstride + mat * m + row] = e_all[l][k];   // (zero for a panel that found nothing)
        }
    }
    if (snap_base == nullptr) {                                        // (uniform)
        if (tid < K * 64) prow_out[(mat * K + (tid >> 6)) * 64 + (tid & 63)] = (tid & 63) < tj[tid >> 6] ? prow_all[tid >> 6][tid & 63] : 0;
        GF2_STAMP(5);
#if GF2_SWEEP_DIAG
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&g_panel_diag[7], 1ull);
#endif
        return;
    }
    // snapshots of the pivot rows as they stand in memory (the trailing pass overwrites them).  The K panels' rows as ONE list of
    // 16-byte pieces, four loads in flight per lane before the first store: written as a load-store loop per panel this was sixteen
    // dependent round trips to memory for a 2048 x 4096 matrix (17 of the launch's 104 us with 256 matrices in flight, 4 of 36 alone).
    if ((ld & 1) == 0 && (int64_t)K * 64 * (ld >> 1) < (1ll << 31)) {
        const unsigned int pl = (unsigned int)(ld >> 1), per_panel = 64u * pl, total = (unsigned int)K * per_panel;
        for (unsigned int base = tid; base < total; base += 4 * RB_THREADS) {
            u32x4_t v[4];
            unsigned int dst[4];
            bool on[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned int idx = base + (unsigned int)u * RB_THREADS;
                const unsigned int l = idx / per_panel, rest = idx - l * per_panel, q = rest / pl, pc = rest - q * pl;
                on[u] = idx < total && (int)q < tj[l < (unsigned int)K ? l : 0];
                dst[u] = idx;
                if (on[u]) v[u] = *reinterpret_cast<const u32x4_t*>(a + (int64_t)prow_all[l][q] * ld + 2 * pc);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!on[u]) continue;
                const unsigned int l = dst[u] / per_panel, rest = dst[u] - l * per_panel;
                *reinterpret_cast<u32x4_t*>(snap_base + (int64_t)l * sstride + mat * 64 * ld + 2 * (int64_t)rest) = v[u];
            }
        }
    } else {
#pragma unroll 1
        for (int l = 0; l < K; ++l) {
            u64* snap = snap_base + (int64_t)l * sstride + mat * 64 * ld;
            const int tl = tj[l];
            for (int64_t idx = tid; idx < (int64_t)tl * ld; idx += RB_THREADS) {
                const int q = (int)(idx / ld);
                const int64_t wd = idx - (int64_t)q * ld;
                snap[idx] = a[(int64_t)prow_all[l][q] * ld + wd];
            }
        }
    }
    GF2_STAMP(5);                                                      // state, coefficients out, snapshots
#if GF2_SWEEP_DIAG
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&g_panel_diag[7], 1ull);
#endif
#undef GF2_STAMP
}

// The trailing pass of a sweep of K panels (grid: row blocks, chunks of SW_CW = 64 / K words, matrices; block 1024; 128 KiB of
// dynamic LDS).  Table entry (panel j, group g, nibble v) holds SW_CW words and entries are 256 bytes apart, so that a lookup stays
// one SDWA instruction (the nibble into byte 1 of the address) + one ds_read_b128:
//   K = 2: entries of 32 words, panel j at j * 64 KiB (the pair kernel's layout);
//   K = 4: entries of 16 words, the two panels of a pair side by side in one 256-byte slot (panel 2i at +0, 2i+1 at +128), pair i
//          at i * 64 KiB.  A wavefront moves eight rows per slot (eight lanes = 128 contiguous bytes each); the sixteen lanes that
//          a ds_read_b128 serves together are {0-3, 12-15, 20-27} ..., four quarter-rows of FOUR different rows: all on one panel's
//          half they would meet two and two on the same banks, so lanes 16-31 and 48-63 take the panels of a pair in the other
//          order (XOR does not care) and every group covers the 64 banks once.
template <int K, int TH>
__device__ __forceinline__ void sweep_update_unit(u64* T, const int64_t mat, const int64_t chunk, const int64_t r_lo, const int64_t row_end,
                                                  const unsigned int unit, u64* base, int64_t m, int64_t ld,
                                                  const SweepState* __restrict__ states, SweepState* __restrict__ live,
                                                  const u64* __restrict__ d_base,
                                                  int64_t dstride, const u64* __restrict__ snap_base,
                                                  int64_t sstride, int64_t pw0,
                                                  u64* __restrict__ colw_base, u64* out_base, const int32_t* __restrict__ prow_base = nullptr) {
    // snap_base == nullptr: the pivot rows are read where they lie (prow_base: their numbers) -- only for a workgroup that owns ALL
    // rows of its chunk, which reads them for its tables before it writes any row
    // states: what this sweep's panels left (with look-ahead a COPY: the next sweep's panels are writing the state by now);
    // live: the state the next panel kernel reads -- it learns from there that the side buffer holds its column words
    static_assert(K == 2 || K == 4, "two or four panels per sweep");
#if GF2_SWEEP_DIAG
    const u64 diag_wg0 = wall_clock64();
#endif
    constexpr int CW = 64 / K;                                          // words of a row per workgroup
    constexpr int LPR = CW / 2;                                         // lanes per row (16 bytes each)
    constexpr int RPS = 64 / LPR;                                       // rows of a wavefront's slot
    constexpr int ITER = 64 * CW / TH;                          // pivot-row words per lane and panel in the table build
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tg[K], t_all = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        tg[j] = states[mat].tg[j];
        t_all |= tg[j];
    }
    const int64_t skip_hi = states[mat].skip_hi;
    const bool moving = out_base != base;                              // uniform
    const int64_t cw0 = chunk * CW;
    const int wc_n = ld - cw0 < CW ? (int)(ld - cw0) : CW;
    u64* a = base + mat * m * ld;
    u64* a_out = out_base + mat * m * ld;
    // the next sweep's K column words, if this chunk holds them: they leave for the side buffer with the rows
    const int64_t pw_next = pw0 + K;
    const bool emits = colw_base != nullptr && pw_next < ld && pw_next >= cw0 && pw_next < cw0 + CW;     // uniform
    u64* colw = colw_base + mat * m * K;
    if (t_all == 0 || (cw0 + CW) * 64 <= skip_hi) {
        // nothing to add here (no pivots, or the chunk lies left of everything that can change); `emits` implies the chunk is to
        // the right of the sweep, so only the first reason applies to an emitting workgroup: the words go out as they stand
        if (moving || emits) {
            for (int64_t idx = tid; idx < (row_end - r_lo) * CW; idx += TH) {
                const int64_t row = r_lo + idx / CW;
                const int wd = (int)(idx % CW);
                if (wd >= wc_n) continue;
                const u64 v = a[row * ld + cw0 + wd];
                if (moving) a_out[row * ld + cw0 + wd] = v;
                if (emits && cw0 + wd >= pw_next && cw0 + wd < pw_next + K) colw[row * K + (cw0 + wd - pw_next)] = v;
            }
            if (emits && tid == 0 && r_lo == 0) live[mat].colw_pw = pw_next;
        }
        return;
    }
    // A handful of pivots (the sweeps after the one that brought most matrices to full rank: a random 2048 x 4096 matrix lacks one
    // to three pivots after its first 2048 columns with probability 0.71): no tables, a row takes each pivot row it has the bit for --
    // 16 K lookups per 16-byte piece would be spent on coefficients that are zero but for a few bits.
    int t_sum = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) t_sum += tg[j];
    if (t_sum <= 8) {
        u64* P = T;                                                    // [t_sum][CW] the pivot rows' words of this chunk
        for (int idx = tid; idx < 8 * CW; idx += TH) {
            int p = idx / CW, j = 0;
            const int wd = idx % CW;
            u64 v = 0;
            if (p < t_sum) {
#pragma unroll
                for (int jj = 0; jj < K - 1; ++jj)
                    if (j == jj && p >= tg[jj]) p -= tg[jj], j = jj + 1;
                if (wd < wc_n)
                    v = snap_base ? snap_base[(int64_t)j * sstride + (mat * 64 + p) * ld + cw0 + wd]
                                  : a[(int64_t)prow_base[(mat * K + j) * 64 + p] * ld + cw0 + wd];
            }
            P[idx] = v;
        }
        __syncthreads();
        const int emit_lo = emits ? (int)(pw_next - cw0) : -1;
        for (int64_t idx = tid; idx < (row_end - r_lo) * LPR; idx += TH) {
            const int64_t row = r_lo + idx / LPR;
            const int hw2 = (int)(idx % LPR) * 2;
            if (hw2 >= wc_n) continue;
            const bool two = hw2 + 1 < wc_n;
            const bool dead = !moving && (cw0 + hw2 + 1) * 64 <= skip_hi && (!two || (cw0 + hw2 + 2) * 64 <= skip_hi);
            if (dead) continue;                                        // (left of everything that can change; never a word of the next sweep)
            u64 x0 = a[row * ld + cw0 + hw2], x1 = two ? a[row * ld + cw0 + hw2 + 1] : 0ull;
            u64 any = 0;
            int p0 = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (tg[j] == 0) continue;                              // uniform
                const u64 dj = d_base[(int64_t)j * dstride + mat * m + row];
                any |= dj;
                for (int q = 0; q < tg[j]; ++q) {
                    const u64 hit = 0ull - ((dj >> q) & 1ull);
                    x0 ^= P[(p0 + q) * CW + hw2] & hit;
                    x1 ^= P[(p0 + q) * CW + hw2 + 1] & hit;         // (hw2 + 1 < CW: CW is even)
                }
                p0 += tg[j];
            }
            if (any || moving) {
                a_out[row * ld + cw0 + hw2] = x0;
                if (two) a_out[row * ld + cw0 + hw2 + 1] = x1;
            }
            if (emit_lo >= 0 && hw2 >= emit_lo && hw2 < emit_lo + K) {
                colw[row * K + (hw2 - emit_lo)] = x0;
                colw[row * K + (hw2 - emit_lo) + 1] = x1;
            }
        }
        if (emits && tid == 0 && r_lo == 0) live[mat].colw_pw = pw_next;
        return;
    }
    // byte offset of panel j's table (see above)
    auto tbase = [](int j) -> unsigned int { return K == 2 ? (unsigned int)j * 65536u : (unsigned int)(j >> 1) * 65536u + (unsigned int)(j & 1) * 128u; };
    // the 16 XOR combinations of a group's four rows (entries 1, 2, 4, 8 are the single rows): a lane takes (group, word) and the
    // entries with bit 3 clear or set -- 32 * CW lanes
    auto combos = [&](int j) {
      for (int c = tid; c < 32 * CW; c += TH) {
        const int wd = c & (CW - 1), g = (c / CW) & 15, top = c / (16 * CW);
        u64* e = T + tbase(j) / 8 + (g * 16) * 32 + wd;                // entries are 32 words (256 bytes) apart
        const u64 r0 = e[1 * 32], r1 = e[2 * 32], r2 = e[4 * 32], r3 = top ? e[8 * 32] : 0ull;
        const u64 c3 = r3, c13 = r0 ^ r3, c23 = r1 ^ r3, c123 = r0 ^ r1 ^ r3;
        u64* o = e + (top ? 8 * 32 : 0);
        if (!top) o[0] = 0ull;
        if (top) o[1 * 32] = c13, o[2 * 32] = c23;
        o[3 * 32] = c123;
        if (top) o[4 * 32] = r2 ^ c3;
        o[5 * 32] = r2 ^ c13;
        o[6 * 32] = r2 ^ c23;
        o[7 * 32] = r2 ^ c123;
      }
    };
    const int sub = lane / LPR, hw = (lane % LPR) * 2;                  // row of the slot, first of this lane's two words
    const bool valid0 = hw < wc_n, valid1 = hw + 1 < wc_n;
    auto skippable = [&](int wd) { return (cw0 + wd + 1) * 64 <= skip_hi; };
    const bool lane_live = valid0 && (moving || !(skippable(hw) && (!valid1 || skippable(hw + 1))));
    // lanes 16-31 and 48-63 take the panels of a pair in the other order (K = 4: see above)
    const int swp = K == 4 ? (lane >> 3) & 1 : 0;
    const unsigned int c0f = 0x0fu;
    unsigned int at_first[K / 2], at_second[K / 2];
    int64_t d_first[K / 2], d_second[K / 2];                            // offsets into d_base of the panels this lane takes first / second
#pragma unroll
    for (int i = 0; i < K / 2; ++i) {
        at_first[i] = tbase(2 * i + swp) + (unsigned int)hw * 8u;
        at_second[i] = tbase(2 * i + 1 - swp) + (unsigned int)hw * 8u;
        d_first[i] = (int64_t)(2 * i + swp) * dstride + mat * m;
        d_second[i] = (int64_t)(2 * i + 1 - swp) * dstride + mat * m;
    }
    constexpr int NW = TH / 64;
    const int rl = RPS * wave + sub;                                   // this lane's row among the RPS * NW of a slot
    const unsigned int lane_word = (unsigned int)rl * (unsigned int)ld + (unsigned int)hw;
    constexpr int SROWS = RPS * NW;                                    // rows of the workgroup's slot
    constexpr int STEP = SROWS * 2;                                    // two slots are worked on while the next two are on their way
    const int emit_at = emits ? (int)(pw_next - cw0) : -1;             // first word of the chunk that goes to the side buffer
    struct Two {
        u32x4_t x[2];
        u64 dF[K / 2][2], dS[K / 2][2];
    };
    auto load2 = [&](int64_t rb, Two& s) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t rs = rb + (int64_t)u * SROWS;                // uniform
            const bool in = rs + rl < row_end;
            const u64* au = a + rs * ld + cw0;
#pragma unroll
            for (int i = 0; i < K / 2; ++i) {
                s.dF[i][u] = in ? d_base[d_first[i] + rs + rl] : 0ull;
                s.dS[i][u] = in ? d_base[d_second[i] + rs + rl] : 0ull;
            }
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (in && lane_live) {
                if (valid1) {
                    v = *reinterpret_cast<const u32x4_t*>(au + lane_word);
                } else {
                    const u64 one = au[lane_word];
                    v[0] = (unsigned int)one, v[1] = (unsigned int)(one >> 32);
                }
            }
            s.x[u] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    Two s0, s1;
    const int64_t rb0 = r_lo;
#if GF2_SWEEP_DIAG
    const u64 diag_c0 = clock64(), diag_w0 = wall_clock64();
#endif
    load2(rb0, s0);                                                    // on their way while the tables are built
    // ---- tables: the K panels' pivot rows as the snapshot holds them (the coefficients are folded onto those: nothing to fix) -----
    {
        u64 sv[K][ITER];
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int idx = tid + TH * it, p = idx / CW, wd = idx & (CW - 1);
                sv[j][it] = !(p < tg[j] && wd < wc_n) ? 0ull
                            : snap_base ? snap_base[(int64_t)j * sstride + (mat * 64 + p) * ld + cw0 + wd]
                                        : a[(int64_t)prow_base[(mat * K + j) * 64 + p] * ld + cw0 + wd];
            }
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int idx = tid + TH * it, p = idx / CW, wd = idx & (CW - 1);
                T[tbase(j) / 8 + ((p >> 2) * 16 + (1 << (p & 3))) * 32 + wd] = sv[j][it];
            }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < K; ++j) combos(j);
        __syncthreads();
    }
#if GF2_SWEEP_DIAG
    const u64 diag_c1 = clock64(), diag_w1 = wall_clock64();
#endif
    auto work2 = [&](int64_t rb, Two& s) {
        unsigned int pf[K / 2][2], ps[K / 2][2];
#pragma unroll
        for (int i = 0; i < K / 2; ++i) pf[i][0] = pf[i][1] = at_first[i], ps[i][0] = ps[i][1] = at_second[i];
#define GF2_SWEEP_BYTE(B)                                                    \
    _Pragma("unroll") for (int i = 0; i < K / 2; ++i) {                       \
        pair_lookups2<B>(s.x, s.dF[i], s.dS[i], pf[i], ps[i], c0f);          \
        __builtin_amdgcn_sched_barrier(0);                                    \
    }
        GF2_SWEEP_BYTE(0) GF2_SWEEP_BYTE(1) GF2_SWEEP_BYTE(2) GF2_SWEEP_BYTE(3)
        GF2_SWEEP_BYTE(4) GF2_SWEEP_BYTE(5) GF2_SWEEP_BYTE(6) GF2_SWEEP_BYTE(7)
#undef GF2_SWEEP_BYTE
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t rs = rb + (int64_t)u * SROWS;
            const bool in = rs + rl < row_end;
            u64* au = a_out + rs * ld + cw0;
            u64 any = 0;
#pragma unroll
            for (int i = 0; i < K / 2; ++i) any |= s.dF[i][u] | s.dS[i][u];
            if ((any || (moving && in)) && lane_live) {
                if (valid1)
                    *reinterpret_cast<u32x4_t*>(au + lane_word) = s.x[u];
                else
                    au[lane_word] = ((u64)s.x[u][1] << 32) | s.x[u][0];
            }
            if (emit_at >= 0 && in && hw >= emit_at && hw < emit_at + K)
                *reinterpret_cast<u32x4_t*>(colw + (rs + rl) * K + (hw - emit_at)) = s.x[u];
        }
    };
    for (int64_t rb = rb0; rb < row_end; rb += 2 * STEP) {
        load2(rb + STEP, s1);
        work2(rb, s0);
        load2(rb + 2 * STEP, s0);
        if (rb + STEP < row_end) work2(rb + STEP, s1);                 // uniform
    }
    if (emits && tid == 0 && r_lo == 0) live[mat].colw_pw = pw_next;
#if GF2_SWEEP_DIAG
    if (tid == 0) {
        const u64 diag_c2 = clock64(), diag_w2 = wall_clock64();
        atomicAdd(&g_sweep_diag[0], diag_c1 - diag_c0);
        atomicAdd(&g_sweep_diag[1], diag_w1 - diag_w0);
        atomicAdd(&g_sweep_diag[2], diag_c2 - diag_c1);
        atomicAdd(&g_sweep_diag[3], diag_w2 - diag_w1);
        atomicAdd(&g_sweep_diag[4], 1ull);
        if (pw0 == (int64_t)g_sweep_diag[7]) {                        // the sweep whose workgroups are logged
            const unsigned int wg = unit;
            if (wg < 4096) {
                unsigned int hw, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                g_sweep_wg[4 * wg] = diag_wg0;
                g_sweep_wg[4 * wg + 1] = diag_w1;
                g_sweep_wg[4 * wg + 2] = diag_w2;
                g_sweep_wg[4 * wg + 3] = ((u64)xcc << 32) | hw;
            }
        }
    }
#endif
}

// (A persistent form -- one workgroup per CU taking units off a counter -- was built and measured: 220 - 279 us per sweep of
// 256 matrices against 195 - 223 for this grid.  The per-workgroup clocks that prompted it, profiles/r05_pass_timeline.md, had shown
// gaps of 20 - 30 us between one workgroup's end and the next one's start on the same CU; they are the workgroup's own later
// wavefronts -- the LDS serves the oldest wavefront first, so wavefront 0, which took the stamps, is done long before the last.)
template <int K, int TH>
__global__ __launch_bounds__(TH) void rref_sweep_update_kernel(u64* base, int64_t m, int64_t ld, int64_t rows_per_wg,
                                                               const SweepState* __restrict__ states, SweepState* __restrict__ live,
                                                               const u64* __restrict__ d_base,
                                                               int64_t dstride, const u64* __restrict__ snap_base,
                                                               int64_t sstride, int64_t pw0,
                                                               u64* __restrict__ colw_base, u64* out_base, int chunk_base, int chunk_skip,
                                                               const int32_t* __restrict__ prow_base = nullptr) {
    // the launch covers chunks chunk_base .. chunk_base + gridDim.y - 1 but chunk_skip (look-ahead of the streamed path: the chunk
    // of the next sweep's columns goes first, in a launch of its own)
    extern __shared__ __attribute__((aligned(16))) u64 T[];           // (the lookups address the tables from LDS byte 0: the kernel's only LDS)
    const int chunk = (int)blockIdx.y + chunk_base;
    if (chunk == chunk_skip) return;
    const unsigned int unit = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int64_t r_lo = (int64_t)blockIdx.x * rows_per_wg;
    sweep_update_unit<K, TH>(T, blockIdx.z, chunk, r_lo, r_lo + rows_per_wg < m ? r_lo + rows_per_wg : m, unit, base, m, ld, states, live, d_base,
                             dstride, snap_base, sstride, pw0, colw_base, out_base, prow_base);
}

// The streamed path's big launch as a ONE-dimensional grid: workgroups 0 .. blocks * n_live * batch - 1 are the live ones (chunks from
// c_first on, but chunk_skip), the chunks left of c_first -- which return at once unless their matrix has seen a pivot-free
// column -- come after them.  Workgroups go to the XCDs in turn, in grid order: with the live ones first and in one run every XCD
// holds at most ceil(live workgroups / 8) of them, and the launcher can count the CUs each XCD has left for the next sweep's panels.
// (As a grid of row blocks x all chunks x matrices, the dead chunks' workgroups took part in the rotation: with four row blocks the
// even chunks' workgroups all went to XCDs 0-3, and a finish workgroup sent to a full XCD waited for the whole pass, 143 us.)
template <int K, int TH>
__global__ __launch_bounds__(TH) void rref_sweep_update_flat_kernel(u64* base, int64_t m, int64_t ld, int64_t rows_per_wg, int blocks,
                                                                    int chunks, int c_first, int chunk_skip,
                                                                    const SweepState* __restrict__ states, SweepState* __restrict__ live,
                                                                    const u64* __restrict__ d_base, int64_t dstride,
                                                                    const u64* __restrict__ snap_base, int64_t sstride, int64_t pw0,
                                                                    u64* __restrict__ colw_base, u64* out_base, int batch) {
    extern __shared__ __attribute__((aligned(16))) u64 T[];
    const int n_live = chunks - c_first - 1;                           // (chunk_skip lies in [c_first, chunks))
    const unsigned int live_wgs = (unsigned int)(blocks * n_live * batch);
    unsigned int id = blockIdx.x;
    int x, chunk, mat;
    if (id < live_wgs) {
        x = (int)(id % (unsigned int)blocks);
        const unsigned int yz = id / (unsigned int)blocks;
        chunk = c_first + (int)(yz % (unsigned int)n_live);
        if (chunk >= chunk_skip) ++chunk;
        mat = (int)(yz / (unsigned int)n_live);
    } else {
        id -= live_wgs;
        x = (int)(id % (unsigned int)blocks);
        const unsigned int yz = id / (unsigned int)blocks;
        chunk = (int)(yz % (unsigned int)c_first);
        mat = (int)(yz / (unsigned int)c_first);
    }
    const int64_t r_lo = (int64_t)x * rows_per_wg;
    if (r_lo >= m) return;
    sweep_update_unit<K, TH>(T, mat, chunk, r_lo, r_lo + rows_per_wg < m ? r_lo + rows_per_wg : m, blockIdx.x, base, m, ld, states, live, d_base,
                             dstride, snap_base, sstride, pw0, colw_base, out_base);
}

// ---- the same sweeps for matrices of more than 4096 rows: rows streamed, not held in registers -----------------------------------
//
// A lane cannot hold the K column words and coefficients of more than four of its rows, so above 4096 rows the right-looking sweep
// keeps them in memory: `colw` (K words per row, the side buffer itself) and the folded coefficients e_l (the arrays the pass reads).
// Per panel l of a sweep, two launches:
//   sweep_stream_panel_kernel (one workgroup per matrix): as rref_panel_stream_kernel -- window filled through an LDS counter,
//     window_round, the rare further rounds applied to every row by this workgroup, the last round's rows left to the chip -- on
//     column l of colw; then it publishes, for the panel's pivot rows, their later column words and their coefficients of the
//     sweep's earlier panels (`pub`, K - 1 vectors of 64 words).
//   sweep_finish_kernel (the whole chip, one row per lane): d = coefficients of the last round (table lookup of the row's word);
//     e_l = d;  w_j ^= d . pub_j for the later columns;  e_{l2} ^= d . pub_{l2} for the earlier panels (rref_sweep_panel_kernel's
//     right-looking step); resets the per-round scratch for the next panel.
// sweep_column_kernel fills colw from the rows when the pass has not (first sweep, a sweep after one without pivots);
// sweep_snapshot_kernel copies the sweep's pivot rows for the pass.  launch_rref_sweeps_streamed runs the next sweep's panels on a
// second stream UNDER this sweep's pass (look-ahead), as launch_rref_blocked does for pairs.
__global__ __launch_bounds__(256) void sweep_column_kernel(const u64* __restrict__ base, int64_t m, int64_t ld, int64_t pw0, int K,
                                                           const SweepState* __restrict__ states, u64* __restrict__ colw_base,
                                                           u64* __restrict__ cco_base, int32_t* __restrict__ slot_base) {
    const int64_t mat = blockIdx.y, row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m) return;
    if (states[mat].colw_pw != pw0) {                                  // uniform per matrix
        const u64* a = base + (mat * m + row) * ld;
        for (int j = 0; j < K; ++j) colw_base[(mat * m + row) * K + j] = pw0 + j < ld ? a[pw0 + j] : 0ull;
    }
    cco_base[mat * m + row] = 0;
    slot_base[mat * m + row] = -1;
}

template <int K>
__global__ __launch_bounds__(RB_THREADS) void sweep_stream_panel_kernel(int64_t m, int64_t n, int64_t ld, int64_t pw0, int l,
                                                                       int64_t* __restrict__ pivots_base, int64_t cap,
                                                                       int32_t* __restrict__ pivrow_base, SweepState* __restrict__ states,
                                                                       unsigned char* __restrict__ used_base,
                                                                       int32_t* __restrict__ prow_base, u64* __restrict__ colw_base,
                                                                       u64* __restrict__ cco_base, int32_t* __restrict__ slot_base,
                                                                       u64* __restrict__ tabs_base, const u64* __restrict__ e_base,
                                                                       int64_t dstride, u64* __restrict__ pub_base) {
    __shared__ u64 VT[2048], TW[2048];
    __shared__ u64 win_w[RB_WIN], win_d[RB_WIN], fin_w[RB_WIN], fin_d[RB_WIN], DP[64], WP[64];
    __shared__ int win_row[RB_WIN], win_piv[RB_WIN], pbit[64], prow_l[64], misc[4];
    __shared__ int win_count, lo_min[2];
    const int64_t mat = blockIdx.x;
    SweepState* st = states + mat;
    unsigned char* used = used_base + mat * m;
    u64* colw = colw_base + mat * m * K;                                // the panel's column: word l of every row's K
    u64* cco = cco_base + mat * m;
    int32_t* slot_of = slot_base + mat * m;                             // -1, or the row's window slot in this round; -2: settled here
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t pw = pw0 + l;
    const int64_t rank = st->rank;
    int64_t first_free = st->first_free;
    if (l == 0 && tid == 0) {                                           // the sweep's first panel: what its pass may skip, nothing found yet
        st->skip_hi = first_free < pw0 * 64 ? first_free : pw0 * 64;    // first_free as it was BEFORE this sweep
        for (int j = 1; j < K; ++j) st->tg[j] = 0;
    }
    if (rank >= m || pw * 64 >= n) {                                    // nothing left to do for this panel
        if (tid == 0) st->tg[l] = 0, st->pending = 0;
        return;
    }
    int64_t* pivots = pivots_base ? pivots_base + mat * cap : nullptr;
    int32_t* pivrow = pivrow_base + mat * cap;
    const int64_t cols_here = n - pw * 64;
    const u64 panel_cols = cols_here >= 64 ? ~0ull : ((1ull << cols_here) - 1ull);
    u64 unresolved = panel_cols;
    int t = 0, pending = 0;
    // The window takes the unused rows in ascending order, so the pivot rows gather at the low indices: after a hundred sweeps of a
    // 32768-row matrix the fill waded through 25000 used rows, 1024 per barrier, before it met a candidate (27 us per panel against
    // 16 in the first sweeps).  scan_lo = a row below which every row is used; the first round of a panel moves it up.
    const int64_t scan_lo = st->scan_lo;
    if (tid < 2) lo_min[tid] = 0x7fffffff;                             // (two, taken in turn: a block's minimum is read while the next block's is made)
    bool lo_open = true;                                               // uniform: no unused row seen yet in this panel's first round
    int lo_found = 0x7fffffff;
    while (unresolved && t < 64 && rank + t < m) {
        if (tid == 0) win_count = 0;
        __syncthreads();
        int block = 0;
        for (int64_t r0 = scan_lo; r0 < m; r0 += RB_THREADS, ++block) {  // fill the window; stop scanning once it is full
            const int64_t row = r0 + tid;
            const bool unused = row < m && !used[row];
            if (lo_open) {                                              // uniform
                const u64 any = __ballot(unused);
                if (any && lane == __ffsll((long long)any) - 1) atomicMin(&lo_min[block & 1], (int)row);
            }
            {
                // (one LDS atomic per wavefront for its candidates' places, not one per row on the one counter)
                const u64 wv = unused ? colw[row * K + l] : 0ull;
                const u64 cb = __ballot((wv & unresolved) != 0);
                int base_pos = 0;
                if (cb) {                                               // uniform per wavefront
                    if (lane == 0) base_pos = atomicAdd(&win_count, (int)__popcll(cb));
                    base_pos = __builtin_amdgcn_readfirstlane(base_pos);
                }
                if ((cb >> lane) & 1ull) {
                    const int pos = base_pos + __builtin_amdgcn_mbcnt_hi((unsigned int)(cb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)cb, 0u));
                    if (pos < RB_WIN) {
                        slot_of[row] = pos;
                        win_row[pos] = (int)row;
                        win_w[pos] = wv;
                        win_d[pos] = cco[row];
                        win_piv[pos] = 0;
                    }
                }
            }
            __syncthreads();
            if (lo_open) {
                lo_found = lo_min[block & 1];
                lo_open = lo_found == 0x7fffffff;
            }
            if (win_count >= RB_WIN) break;
        }
        if (lo_open) {                                                  // the scan ran to the end without an unused row (uniform)
            lo_found = (int)m;
            lo_open = false;
        }
        const int total = win_count;
        if (total == 0) break;
        const int nwin = total < RB_WIN ? total : RB_WIN;
        if (wave == 0)
            window_round(lane, nwin, t, rank, m, unresolved, win_w, win_d, win_row, win_piv, fin_w, fin_d, pbit, prow_l, DP, WP, misc);
        __syncthreads();
        const int t_new = misc[0];
        const u64 newbits = ((u64)(unsigned int)misc[2] << 32) | (unsigned int)misc[1];
        const bool again = (unresolved & ~newbits) != 0 && t_new < 64 && rank + t_new < m;    // uniform: another round may follow
        round_tables(tid, DP, WP, VT, TW, again);
        __syncthreads();
        if (!again) {
            // the last round: only its pivot rows are settled here (coefficients final, marked -2); every other row takes its
            // coefficients from the probe-row table in sweep_finish_kernel, on the whole chip instead of in this one workgroup
            if (tid < nwin) {
                const int row = win_row[tid];
                if (win_piv[tid]) {
                    cco[row] = fin_d[tid];
                    used[row] = 1;
                    slot_of[row] = -2;
                } else {
                    slot_of[row] = -1;
                }
            }
            for (int idx = tid; idx < 2048; idx += RB_THREADS) tabs_base[mat * 2048 + idx] = VT[idx];
            pending = 1;
            unresolved &= ~newbits;
            t = t_new;
            __syncthreads();
            break;
        }
        for (int64_t row = tid; row < m; row += RB_THREADS) {
            const int sl = slot_of[row];
            if (sl >= 0) slot_of[row] = -1;
            if (sl >= 0 && sl < RB_WIN && win_row[sl] == (int)row && win_piv[sl]) {       // a new pivot row: as the wavefront left it
                colw[row * K + l] = fin_w[sl];
                cco[row] = fin_d[sl];
                used[row] = 1;
            } else {                                                  // every other row: linear in its word
                const u64 w0 = colw[row * K + l];
                cco[row] ^= byte_lookup(VT, w0);
                colw[row * K + l] = byte_lookup(TW, w0);
            }
        }
        unresolved &= ~newbits;
        t = t_new;
        __syncthreads();
    }
    if (unresolved) {
        const int64_t fc = pw * 64 + (__ffsll((long long)unresolved) - 1);
        if (fc < first_free) first_free = fc;
    }
    if (tid == 0) {
        if (lo_found != 0x7fffffff) st->scan_lo = lo_found;
        st->pending = pending;
        st->tg[l] = t;
        st->rank = rank + t;
        st->first_free = first_free;
    }
    if (t == 0) return;
    if (wave == 0 && lane < t) {
        const u64 resolved = panel_cols & ~unresolved;
        const int pos = __popcll(resolved & ((1ull << pbit[lane]) - 1ull));
        pivrow[rank + pos] = prow_l[lane];
        if (pivots) pivots[rank + pos] = pw * 64 + pbit[lane];
        prow_base[(mat * K + l) * 64 + lane] = prow_l[lane];            // for the snapshot of the sweep's pivot rows
    }
    // what the panel's pivot rows show the other rows: their later column words, their coefficients of the sweep's earlier panels
    if (tid < 64 * (K - 1)) {
        const int v = tid >> 6, q = tid & 63;
        u64 x = 0;
        if (q < t) {
            const int64_t prow = prow_l[q];
            x = v < K - 1 - l ? colw[prow * K + (l + 1 + v)] : e_base[(int64_t)(v - (K - 1 - l)) * dstride + mat * m + prow];
        }
        pub_base[(mat * (K - 1) + v) * 64 + q] = x;
    }
}

// The rest of a streamed panel, a row per lane and trip, on as many workgroups as the launcher finds CUs for (see above).  Dynamic LDS: K tables.
template <int K>
__global__ __launch_bounds__(1024) void sweep_finish_kernel(int64_t m, int l, const SweepState* __restrict__ states,
                                                            u64* __restrict__ colw_base, u64* __restrict__ cco_base,
                                                            int32_t* __restrict__ slot_base, const u64* __restrict__ tabs_base,
                                                            u64* __restrict__ e_base, int64_t dstride, const u64* __restrict__ pub_base) {
    extern __shared__ __attribute__((aligned(16))) u64 FT[];           // [0]: the last round's table, [1 .. K - 1]: the published vectors'
    const int64_t mat = blockIdx.y;
    const int64_t row0 = (int64_t)blockIdx.x * 1024 + threadIdx.x, stride = (int64_t)gridDim.x * 1024;   // rows row0, row0 + stride, ...
    const int t = states[mat].tg[l], pending = states[mat].pending;
    u64* el = e_base + (int64_t)l * dstride + mat * m;
    if (t == 0) {                                                       // the pass reads the panel's coefficients all the same
        for (int64_t row = row0; row < m; row += stride) el[row] = 0ull;
        return;
    }
    __shared__ u64 pv_s[(K - 1) * 64];
    if (threadIdx.x < (K - 1) * 64) pv_s[threadIdx.x] = pub_base[mat * (K - 1) * 64 + threadIdx.x];
    for (int idx = threadIdx.x; idx < 2048; idx += 1024) FT[idx] = pending ? tabs_base[mat * 2048 + idx] : 0ull;
    __syncthreads();
    if (threadIdx.x < (K - 1) * 128) gray_byte_table(threadIdx.x & 127, pv_s + (threadIdx.x >> 7) * 64, FT + 2048 + (threadIdx.x >> 7) * 2048);
    __syncthreads();
    for (int64_t row = row0; row < m; row += stride) {
        u64* w = colw_base + (mat * m + row) * K;
        u64 d = cco_base[mat * m + row];
        if (pending && slot_base[mat * m + row] != -2) d ^= byte_lookup(FT, w[l]);
        el[row] = d;
        for (int j = l + 1; j < K; ++j) w[j] ^= byte_lookup(FT + 2048 * (j - l), d);
        for (int l2 = 0; l2 < l; ++l2) e_base[(int64_t)l2 * dstride + mat * m + row] ^= byte_lookup(FT + 2048 * (K - l + l2), d);
        cco_base[mat * m + row] = 0;
        slot_base[mat * m + row] = -1;
    }
}

// The sweep's pivot rows as they stand in memory, for the pass: workgroup (64 l + q, matrix) copies pivot row q of panel l.
__global__ __launch_bounds__(1024) void sweep_snapshot_kernel(const u64* __restrict__ base, int64_t m, int64_t ld, int K,
                                                              const SweepState* __restrict__ states, const int32_t* __restrict__ prow_base,
                                                              u64* __restrict__ snap_base, int64_t sstride) {
    const int64_t mat = blockIdx.y;
    const int l = (int)blockIdx.x >> 6, q = (int)blockIdx.x & 63;
    if (q >= states[mat].tg[l]) return;
    const u64* src = base + (mat * m + prow_base[(mat * K + l) * 64 + q]) * ld;
    u64* dst = snap_base + (int64_t)l * sstride + (mat * 64 + q) * ld;
    for (int64_t wd = threadIdx.x; wd < ld; wd += blockDim.x) dst[wd] = src[wd];
}

// ---- blocked normalisation (css_code.py:809-836), bit-exact ------------------------------------------------------------
//
// The reference adds the first odd row at or below the diagonal INTO the diagonal row (never swaps rows) and clears the
// column in every other row; when no row has the bit it swaps in the first odd column of the diagonal row's current
// state.  The outcome depends on that order, so the blocked form replays exactly those
// operations, regrouped: steps i0 .. i0+63 form a panel on the 64 columns c0 = i0+offset ..; one wavefront simulates
// them on a window of the rows i0 .. i0+127 in position order (the donor of step i is the first window entry at or after
// the diagonal with the bit set).  With P_p the vector the reference adds to the other rows at step p,
//     P_p = B_p ^ sum_{q<p} csel_p[q] P_q,   B_p = old_{i0+p} ^ [diagonal was even] old_{donor_p},
// every other row ends as old_j ^ d_j . B and diagonal row p as d_p . B -- the same trailing update as the RREF, rebuilt rows
// starting from zero.  The coefficients d are kept in that final form while the steps are simulated (a row that takes P_p takes
// d of the rebuilt diagonal row, e_p ^ d_diagonal ^ [diagonal was even] d_donor), as in the RREF's window_round.  When the window holds no donor for
// a step (the donor is further down, or a column swap is due) the panel stops there, its steps are applied, and that one
// step is done by the sequential kernel on the materialised matrix.
template <int RPT>
__global__ __launch_bounds__(RB_THREADS) void norm_panel_kernel(u64* __restrict__ a, int64_t r, int64_t n, int64_t ld,
                                                               int64_t offset, RrefState* __restrict__ st,
                                                               const int* __restrict__ status, u64* __restrict__ dout,
                                                               u64* __restrict__ snap) {
    __shared__ u64 VT[2048];                                            // byte tables of the probe rows' coefficients
    __shared__ u64 win_w[RB_WIN], win_d[RB_WIN], DP[64];
    __shared__ int donor[64], misc[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i0 = st->rank;
    if (i0 >= r || status[0] != 0) {
        if (tid == 0) {
            st->t = 0;
            st->tg[0] = st->tg[1] = 0;
            st->stalled = 0;
        }
        return;
    }
    const int64_t c0 = i0 + offset, cw = c0 >> 6;
    const int sh = (int)(c0 & 63);
    const int steps = r - i0 < 64 ? (int)(r - i0) : 64;
    const u64 stepmask = steps >= 64 ? ~0ull : ((1ull << steps) - 1ull);

    u64 w[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int64_t row = tid + (int64_t)RB_THREADS * k;
        w[k] = 0;
        if (row < r) {
            u64 v = a[row * ld + cw] >> sh;
            if (sh && cw + 1 < ld) v |= a[row * ld + cw + 1] << (64 - sh);
            w[k] = v & stepmask;
            if (row >= i0 && row < i0 + RB_WIN) win_w[row - i0] = w[k];
        }
    }
    if (tid < RB_WIN && i0 + tid >= r) win_w[tid] = 0;
    __syncthreads();
    if (wave == 0) {
        // Coefficients in their final form, as in the RREF's window_round: the rebuilt diagonal row of step p is
        // (e_p ^ d_diag ^ [diagonal even] d_donor) . B from zero, a row that takes it takes those coefficients, and 64 probe rows
        // e_j, eliminated along, give every row outside the window its coefficients by table lookup (such a row is never a
        // diagonal row or a donor of this panel, so what happens to it is linear in its word).
        // (The step on 32-bit halves with the rows that take P as per-lane masks -- x ^= P & mask, one v_bitop3 per dword -- and the
        // diagonal row written by v_writelane, as in window_round: the form with 64-bit shifts, ballots for every row set and
        // `if (bit) x ^= P` compiled to selects was ~2x the vector instructions.)
        const u64 ww0 = win_w[lane], ww1 = win_w[lane + 64];
        unsigned int e0l = (unsigned int)ww0, e0h = (unsigned int)(ww0 >> 32), e1l = (unsigned int)ww1, e1h = (unsigned int)(ww1 >> 32);
        unsigned int e2l = lane < 32 ? 1u << lane : 0u, e2h = lane >= 32 ? 1u << (lane - 32) : 0u;
        unsigned int f0l = 0, f0h = 0, f1l = 0, f1h = 0, f2l = 0, f2h = 0;            // the coefficients
        int tt = 0;
        auto half_steps = [&](auto half_constant, int s_lo, int s_hi) {
            constexpr int HALF = decltype(half_constant)::value;
            for (int sidx = s_lo; sidx < s_hi; ++sidx) {
                const int s5 = sidx - 32 * HALF;
                int m0 = __builtin_amdgcn_sbfe((int)(HALF ? e0h : e0l), s5, 1);      // all ones: the row has the step's column
                const int m1 = __builtin_amdgcn_sbfe((int)(HALF ? e1h : e1l), s5, 1);
                const int m2 = __builtin_amdgcn_sbfe((int)(HALF ? e2h : e2l), s5, 1);
                const u64 bal0 = __ballot(m0 != 0) & (~0ull << sidx);                // at or below the diagonal
                u64 bal1 = 0;
                if (__builtin_expect(bal0 == 0, 0)) {
                    bal1 = __ballot(m1 != 0);
                    if (!bal1) return false;                               // no donor inside the window
                }
                const int h = bal0 ? 0 : 1;
                const int src = __ffsll((long long)(bal0 ? bal0 : bal1)) - 1;
                unsigned int pwl = (unsigned int)__builtin_amdgcn_readlane((int)e0l, sidx), pwh = (unsigned int)__builtin_amdgcn_readlane((int)e0h, sidx);
                unsigned int pdl = (unsigned int)__builtin_amdgcn_readlane((int)f0l, sidx), pdh = (unsigned int)__builtin_amdgcn_readlane((int)f0h, sidx);
                int dn = -1;
                if (!(h == 0 && src == sidx)) {                             // diagonal entry is even: add the donor
                    if (h == 0) {
                        pwl ^= (unsigned int)__builtin_amdgcn_readlane((int)e0l, src), pwh ^= (unsigned int)__builtin_amdgcn_readlane((int)e0h, src);
                        pdl ^= (unsigned int)__builtin_amdgcn_readlane((int)f0l, src), pdh ^= (unsigned int)__builtin_amdgcn_readlane((int)f0h, src);
                    } else {
                        pwl ^= (unsigned int)__builtin_amdgcn_readlane((int)e1l, src), pwh ^= (unsigned int)__builtin_amdgcn_readlane((int)e1h, src);
                        pdl ^= (unsigned int)__builtin_amdgcn_readlane((int)f1l, src), pdh ^= (unsigned int)__builtin_amdgcn_readlane((int)f1h, src);
                    }
                    dn = (int)(i0 + src + 64 * h);
                }
                if (sidx < 32) pdl ^= 1u << sidx; else pdh ^= 1u << (sidx - 32);      // (sidx's half is the loop's: a constant condition)
                asm("v_writelane_b32 %0, 0, %1" : "+v"(m0) : "s"(sidx));              // the diagonal row is written, not added to
                e0l = __builtin_amdgcn_bitop3_b32(e0l, pwl, (unsigned int)m0, 0x78), e0h = __builtin_amdgcn_bitop3_b32(e0h, pwh, (unsigned int)m0, 0x78);
                f0l = __builtin_amdgcn_bitop3_b32(f0l, pdl, (unsigned int)m0, 0x78), f0h = __builtin_amdgcn_bitop3_b32(f0h, pdh, (unsigned int)m0, 0x78);
                e1l = __builtin_amdgcn_bitop3_b32(e1l, pwl, (unsigned int)m1, 0x78), e1h = __builtin_amdgcn_bitop3_b32(e1h, pwh, (unsigned int)m1, 0x78);
                f1l = __builtin_amdgcn_bitop3_b32(f1l, pdl, (unsigned int)m1, 0x78), f1h = __builtin_amdgcn_bitop3_b32(f1h, pdh, (unsigned int)m1, 0x78);
                e2l = __builtin_amdgcn_bitop3_b32(e2l, pwl, (unsigned int)m2, 0x78), e2h = __builtin_amdgcn_bitop3_b32(e2h, pwh, (unsigned int)m2, 0x78);
                f2l = __builtin_amdgcn_bitop3_b32(f2l, pdl, (unsigned int)m2, 0x78), f2h = __builtin_amdgcn_bitop3_b32(f2h, pdh, (unsigned int)m2, 0x78);
                // (the lane in M0: a scalar value and a scalar lane select in one instruction are one constant-bus operand too many)
                asm volatile("s_mov_b32 m0, %8\n\tv_writelane_b32 %0, %4, m0\n\tv_writelane_b32 %1, %5, m0\n\tv_writelane_b32 %2, %6, m0\n\t"
                             "v_writelane_b32 %3, %7, m0"
                             : "+v"(e0l), "+v"(e0h), "+v"(f0l), "+v"(f0h)
                             : "s"(pwl), "s"(pwh), "s"(pdl), "s"(pdh), "s"(sidx)
                             : "m0");
                if (lane == 0) donor[sidx] = dn;
                tt += 1;
            }
            return true;
        };
        if (half_steps(std::integral_constant<int, 0>{}, 0, steps < 32 ? steps : 32) && steps > 32)
            (void)half_steps(std::integral_constant<int, 1>{}, 32, steps);
        u64 ed[3] = {((u64)f0h << 32) | f0l, ((u64)f1h << 32) | f1l, ((u64)f2h << 32) | f2l};
        win_d[lane] = ed[0];
        win_d[lane + 64] = ed[1];
        DP[lane] = ed[2];
        if (lane == 0) misc[0] = tt;
    }
    __syncthreads();
    const int t = misc[0];
    if (tid == 0) {
        st->t = t;
        st->tg[0] = t;                                               // the trailing pass takes this panel as the first of a pair without a second
        st->tg[1] = 0;
        st->stalled = t < steps ? 1 : 0;
        st->rank = i0 + t;
        st->skip_lo = offset;
        st->skip_hi = c0;
        st->zero_lo = i0;
        st->zero_hi = i0 + t;
    }
    if (t == 0) return;
    // B_p = old diagonal row (+ old donor row)
    for (int64_t idx = tid; idx < (int64_t)t * ld; idx += RB_THREADS) {
        const int p = (int)(idx / ld);
        const int64_t wd = idx - (int64_t)p * ld;
        u64 v = a[(i0 + p) * ld + wd];
        if (donor[p] >= 0) v ^= a[(int64_t)donor[p] * ld + wd];
        snap[idx] = v;
    }
    for (int idx = tid; idx < 2048; idx += RB_THREADS) {
        const int g = idx >> 8, vv = idx & 255;
        u64 x = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) x ^= DP[8 * g + k] & (0ull - (u64)((vv >> k) & 1));
        VT[idx] = x;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int64_t row = tid + (int64_t)RB_THREADS * k;
        if (row < r) dout[row] = (row >= i0 && row < i0 + RB_WIN) ? win_d[row - i0] : byte_lookup(VT, w[k]);
    }
}

// first_free = n (no pivot-free column seen yet) in the zeroed per-matrix states.
__global__ void rref_state_init_kernel(RrefState* __restrict__ states, int64_t batch, int64_t n) {
    const int64_t mat = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (mat < batch) states[mat].first_free = n;
}

__global__ void sweep_state_init_kernel(SweepState* __restrict__ states, int64_t batch, int64_t n) {
    const int64_t mat = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (mat < batch) {
        states[mat].first_free = n;
        states[mat].colw_pw = -1;
    }
}

// out row k (k < rank) = in row pivrow[k]; rows >= rank are zero.  grid (ceil(m / 16), batch), block 256: sixteen rows per
// workgroup, moved as 16-byte pieces when the row pitch is even (8-byte words otherwise).
#define GATHER_ROWS 16
template <typename State>
__global__ __launch_bounds__(256) void gather_rows_kernel(const u64* __restrict__ in, u64* __restrict__ out,
                                                          const int32_t* __restrict__ pivrow, const State* __restrict__ states,
                                                          int64_t* __restrict__ rank_out, int64_t m, int64_t ld, int64_t cap) {
    typedef unsigned int v4 __attribute__((ext_vector_type(4)));
    __shared__ int64_t src_row[GATHER_ROWS];
    const int64_t k0 = (int64_t)blockIdx.x * GATHER_ROWS, mat = blockIdx.y;
    const int64_t rank = states[mat].rank;
    if (blockIdx.x == 0 && threadIdx.x == 0) rank_out[mat] = rank;
    if (threadIdx.x < GATHER_ROWS) {
        const int64_t k = k0 + threadIdx.x;
        src_row[threadIdx.x] = k < rank ? (int64_t)pivrow[mat * cap + k] : -1;
    }
    __syncthreads();
    const u64* src = in + mat * m * ld;
    u64* dst = out + (mat * m + k0) * ld;
    const int64_t rows = m - k0 < GATHER_ROWS ? m - k0 : GATHER_ROWS;
    if ((ld & 1) == 0) {
        const int64_t ppr = ld >> 1;                                   // 16-byte pieces per row
        for (int64_t idx = threadIdx.x; idx < rows * ppr; idx += 256) {
            const int64_t r = idx / ppr, piece = idx - r * ppr;
            v4 v = {0u, 0u, 0u, 0u};
            if (src_row[r] >= 0) v = *reinterpret_cast<const v4*>(src + src_row[r] * ld + 2 * piece);
            *reinterpret_cast<v4*>(dst + r * ld + 2 * piece) = v;
        }
    } else {
        for (int64_t idx = threadIdx.x; idx < rows * ld; idx += 256) {
            const int64_t r = idx / ld, wd = idx - r * ld;
            dst[r * ld + wd] = src_row[r] >= 0 ? src[src_row[r] * ld + wd] : 0ull;
        }
    }
}

__global__ void swap_columns_kernel(u64* __restrict__ a, int64_t m, int64_t ld, int64_t i, int64_t j) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m) return;
    const int64_t wi = i >> 6, wj = j >> 6;
    const int bi = (int)(i & 63), bj = (int)(j & 63);
    u64 x = a[row * ld + wi], y = a[row * ld + wj];
    const u64 diff = ((x >> bi) ^ (y >> bj)) & 1ull;
    if (wi == wj) {
        a[row * ld + wi] = x ^ ((diff << bi) | (diff << bj));
    } else {
        a[row * ld + wi] = x ^ (diff << bi);
        a[row * ld + wj] = y ^ (diff << bj);
    }
}

// one wave per row
__global__ __launch_bounds__(256) void row_weights_kernel(const u64* __restrict__ a, int64_t m, int64_t ld,
                                                          uint32_t* __restrict__ weights) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= m) return;
    unsigned int acc = 0;
    for (int64_t w = lane; w < ld; w += 64) acc += __popcll(a[row * ld + w]);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) weights[row] = acc;
}

// Output word (t, w): column c is pivot column P[i] -> bit R[i, F[t]]; c == F[t] -> 1; other free columns -> 0.
__global__ void nullspace_kernel(const u64* __restrict__ red, int64_t n, int64_t ld, const int32_t* __restrict__ free_cols,
                                 int64_t nfree, const int32_t* __restrict__ col_pivot_row, u64* __restrict__ out,
                                 int64_t ldn) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t t = blockIdx.y;
    if (w >= ldn || t >= nfree) return;
    const int64_t fc = free_cols[t];
    const int64_t fw = fc >> 6;
    const int fb = (int)(fc & 63);
    u64 acc = 0;
    for (int b = 0; b < 64; ++b) {
        const int64_t c = w * 64 + b;
        if (c >= n) break;
        const int32_t pr = col_pivot_row[c];
        u64 bit;
        if (pr >= 0)
            bit = (red[(int64_t)pr * ld + fw] >> fb) & 1ull;
        else
            bit = c == fc ? 1ull : 0ull;
        acc |= bit << b;
    }
    out[t * ldn + w] = acc;
}

// ---- host side -----------------------------------------------------------------------------------------------------

// ---- RREF of small matrices, one wavefront per matrix --------------------------------------------------------------------
//
// Up to 64 * RPL rows of up to 64 * LD columns: lane l holds rows l, l + 64, ... in registers (RPL rows of 2 LD dwords).  The
// columns are walked left to right; a column that still has a 1 in an unused row gets that row as its pivot (the RREF
// does not depend on which one: the lowest lane of the lowest register row here) and the pivot row is added to every other row
// that has the bit.  The matrix is read once and written once (rows go out in pivot order, the zero rows after them), so a
// large batch of small matrices streams at what the walk allows, and the walk is what round 4 rewrote (round 1's form: 70 vector
// instructions per pivot, 850 GB/s on 64 x 512 matrices):
//   * the dword that holds the column is a compile-time register (the walk is unrolled over the 2 LD dwords of a row, a runtime
//     loop over the 32 bits inside), so there is no select chain over the words of a row: a column's test is an AND and a compare;
//   * which rows are still unused, and which rows have the bit, are 64-bit lane masks in scalar registers, and the rows that
//     take the pivot row are the exec mask as it stands (inverse_ballot): no per-lane predicate arithmetic;
//   * the pivot row travels to scalar registers by v_readlane (one row per lane) or through LDS (more: its lane writes it, every
//     lane reads it back at one address), and only from the 16-byte piece that holds the column on -- an unused row is zero to
//     the left of the column, so the words before it cannot change;
//   * a matrix whose rows are contiguous (ld == LD) is loaded and stored in 16-byte pieces, every lane its own rows.  (Whole
//     1 KiB pieces per instruction, handed to and from the row-per-lane layout through an LDS staging area, were built and
//     measured: 0.377 against 0.342 ms for 256 MiB of 64 x 512 matrices -- the 5 KiB of LDS per wavefront cost a workgroup per CU,
//     and the access pattern was never the limit: PMC traffic is 1.06x the algorithmic bytes either way.)
//   * the column loop has ONE exit: with `continue` and two `break`s the compiler made a state machine of forty scalar
//     instructions and eight branches per pivot out of it (0.43 -> 0.38 ms).
// About 30 vector and 20 scalar instructions per pivot are left.  Grid-stride over the batch.
#define SMALL_WAVES 4
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int RPL, int LD, int RL>
__global__ __launch_bounds__(64 * SMALL_WAVES) void rref_small_kernel(u64* __restrict__ base, int64_t batch, int m, int n, int64_t ld,
                                                                      int64_t* __restrict__ pivots_base, int64_t cap,
                                                                      int64_t* __restrict__ rank_out) {
    constexpr int DW = 2 * LD;                                      // dwords per row
    constexpr bool LANE_PRED = RPL >= 2;                            // see the column loop
    constexpr bool PIECES = LD >= 2;                                // rows of whole 16-byte pieces
    __shared__ __align__(16) unsigned int bcast_all[SMALL_WAVES][DW < 4 ? 4 : DW];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned int* const bcast = bcast_all[wv];
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const bool contiguous = PIECES && ld == LD && (reinterpret_cast<uintptr_t>(base) & 15) == 0;   // uniform
    for (int64_t mat = wave; mat < batch; mat += nwaves) {
        u64* a = base + mat * m * ld;
        unsigned int w[RPL][DW];
        int pivcol[RPL];                                            // >= 0 once this row has become a pivot row: its column
        int myrank[RPL];
        u64 unused[RPL];                                            // lane masks (scalar): rows that exist and are not pivot rows yet
        unsigned int live[RPL];                                     // LANE_PRED: the same per lane, all ones or zero
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            pivcol[q] = -1;
            myrank[q] = 0;
            unused[q] = __ballot(lane + 64 * q < m);
            live[q] = lane + 64 * q < m ? 0xFFFFFFFFu : 0u;
        }
        if (contiguous) {
            // every lane its own rows, 16 bytes per load
            if constexpr (PIECES) {
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    const int row = lane + 64 * q;
                    const u32x4* src = reinterpret_cast<const u32x4*>(a + (int64_t)row * LD);
#pragma unroll
                    for (int j = 0; j < LD / 2; ++j) {
                        u32x4 v = {0, 0, 0, 0};
                        if (row < m) v = src[j];
                        w[q][4 * j] = v.x, w[q][4 * j + 1] = v.y, w[q][4 * j + 2] = v.z, w[q][4 * j + 3] = v.w;
                    }
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                const int row = lane + 64 * q;
#pragma unroll
                for (int k = 0; k < LD; ++k) {
                    const u64 v = (row < m && k < ld) ? a[(int64_t)row * ld + k] : 0ull;
                    w[q][2 * k] = (unsigned int)v;
                    w[q][2 * k + 1] = (unsigned int)(v >> 32);
                }
            }
        }
        int rank = 0;
#pragma unroll
        for (int d = 0; d < DW; ++d) {                              // the dword of the row that holds the column: a constant below
            if (d * 32 >= n || rank >= m) break;                    // uniform
            const int d4 = d & ~3;                                  // the pivot row is zero before its column: pieces from here on
            // (one exit, no `continue`: the compiler turns a loop with several exits into a state machine of scalar moves and
            // compares -- forty scalar instructions and eight branches per pivot in the first form of this loop)
            const int nbits = n - d * 32 < 32 ? n - d * 32 : 32;    // uniform
            int limit = nbits;                                      // (0 once every row is a pivot row: ONE condition for the loop to test)
#pragma unroll 1
            for (int bb = 0; bb < limit; ++bb) {
                const int col = d * 32 + bb;
                const unsigned int bit = 1u << bb;
                u64 has[RPL];
                unsigned int t[RPL];                                // LANE_PRED: the row's bit of the column (zero: the row takes nothing)
                int src_q = -1;
                u64 cand = 0;
                if constexpr (LANE_PRED) {
                    // Two or more rows per lane: which rows have the column and which are still unused stay per-lane values.  As
                    // 64-bit scalar masks (one ballot, its AND with `unused` and a place in the pick-the-first chain per register row,
                    // inverse ballots for the XORs) the walk took 54 / 115 scalar instructions per pivot at two / four rows per lane,
                    // and the CU's ONE scalar unit bound the kernel (profiles/r05_rref_small_floor.md).
#pragma unroll
                    for (int q = 0; q < RPL; ++q) t[q] = w[q][d] & bit;
#pragma unroll
                    for (int q = 0; q < RPL; ++q)
                        if (src_q < 0) {                            // uniform; the lowest register row that has an unused row with the bit
                            const u64 c = __ballot((t[q] & live[q]) != 0);
                            if (c) src_q = q, cand = c;
                        }
                } else {
#pragma unroll
                    for (int q = 0; q < RPL; ++q) {
                        has[q] = __ballot((w[q][d] & bit) != 0);
                        if (src_q < 0 && (has[q] & unused[q])) {
                            src_q = q;
                            cand = has[q] & unused[q];
                        }
                    }
                }
                if (src_q >= 0) {                                   // (else: no unused row has this column: not a pivot column)
                const int src_lane = __ffsll((long long)cand) - 1;
                const u64 src_bit = 1ull << src_lane;
                // The pivot row's way to the other rows.  Two ways, and the kernel uses both: dwords d4 .. split - 1 by v_readlane into
                // scalar registers (a vector instruction each), the rest through LDS (written by the pivot's lane, read back by all
                // at one address; the LDS serves a wavefront's operations in order, so no wait between the two) -- the LDS round trip
                // runs under the readlanes and the XORs of the first part.  All by readlane is bound by issuing them, all through LDS by
                // the round trip and the LDS itself (RL = dwords by readlane, counted from d4).
                const int split = d4 + RL < DW ? (d4 + RL + 3) & ~3 : DW;         // (whole 16-byte pieces go through LDS; a constant after unrolling)
                unsigned int pr[DW];
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    if (q != src_q) continue;                       // uniform
                    if (__builtin_amdgcn_inverse_ballot_w64(src_bit)) {
                        if constexpr (DW >= 4) {
#pragma unroll
                            for (int dd = 0; dd < DW; dd += 4) {
                                if (dd < split) continue;
                                *reinterpret_cast<u32x4*>(bcast + dd) = u32x4{w[q][dd], w[q][dd + 1], w[q][dd + 2], w[q][dd + 3]};
                            }
                        }
                        pivcol[q] = col;
                        myrank[q] = rank;
                        if constexpr (LANE_PRED) {
                            live[q] = 0;                            // a pivot row from here on ...
                            t[q] = 0;                               // ... which does not take itself
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if constexpr (DW >= 4) {
#pragma unroll
                    for (int dd = 0; dd < DW; dd += 4)
                        if (dd >= split) {
                            const u32x4 v = *reinterpret_cast<const u32x4*>(bcast + dd);
                            pr[dd] = v.x, pr[dd + 1] = v.y, pr[dd + 2] = v.z, pr[dd + 3] = v.w;
                        }
                }
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    if (q != src_q) continue;                       // uniform
#pragma unroll
                    for (int dd = 0; dd < DW; ++dd)
                        if (dd >= d4 && dd < split) pr[dd] = (unsigned int)__builtin_amdgcn_readlane((int)w[q][dd], src_lane);
                }
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    bool takes;
                    if constexpr (LANE_PRED) {
                        takes = t[q] != 0;
                    } else {
                        const u64 take = q == src_q ? has[q] & ~src_bit : has[q];
                        takes = __builtin_amdgcn_inverse_ballot_w64(take);
                    }
                    if (takes) {                                             // (the readlane part first: the LDS part is still on its way)
#pragma unroll
                        for (int dd = 0; dd < DW; ++dd)
                            if (dd >= d4 && dd < split) w[q][dd] ^= pr[dd];
#pragma unroll
                        for (int dd = 0; dd < DW; ++dd)
                            if (dd >= split) w[q][dd] ^= pr[dd];
                    }
                }
                if constexpr (!LANE_PRED) unused[src_q] ^= src_bit; // (the bit is set: the pivot came from there)
                rank += 1;
                if (rank >= m) limit = 0;                           // uniform
                __builtin_amdgcn_wave_barrier();                    // (the next pivot row is written after every lane has read this one)
                }
            }
        }
        // rows out in pivot order; everything from row `rank` on is zero
        if (contiguous) {
            if constexpr (PIECES) {
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    const int row = lane + 64 * q;
                    if (pivcol[q] >= 0) {
                        u32x4* dst = reinterpret_cast<u32x4*>(a + (int64_t)myrank[q] * LD);
#pragma unroll
                        for (int j = 0; j < LD / 2; ++j) dst[j] = u32x4{w[q][4 * j], w[q][4 * j + 1], w[q][4 * j + 2], w[q][4 * j + 3]};
                        if (pivots_base) pivots_base[mat * cap + myrank[q]] = pivcol[q];
                    }
                    if (row >= rank && row < m) {
                        u32x4* dst = reinterpret_cast<u32x4*>(a + (int64_t)row * LD);
#pragma unroll
                        for (int j = 0; j < LD / 2; ++j) dst[j] = u32x4{0, 0, 0, 0};
                    }
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                const int row = lane + 64 * q;
                if (pivcol[q] >= 0) {
#pragma unroll
                    for (int k = 0; k < LD; ++k)
                        if (k < ld) a[(int64_t)myrank[q] * ld + k] = ((u64)w[q][2 * k + 1] << 32) | w[q][2 * k];
                    if (pivots_base) pivots_base[mat * cap + myrank[q]] = pivcol[q];
                }
                if (row >= rank && row < m) {
#pragma unroll
                    for (int k = 0; k < LD; ++k)
                        if (k < ld) a[(int64_t)row * ld + k] = 0ull;
                }
            }
        }
        if (lane == 0) rank_out[mat] = rank;
    }
}

// The same walk with the pivot rows taken FOUR AT A TIME (the method of the four Russians, per wavefront).  Adding a pivot row to
// the rows that have its column costs a broadcast and an XOR per dword of the row whatever the number of rows that take it; so the
// walk finds four pivots on the ONE dword that holds their columns (a copy x of it: test, pick, one readlane, one masked XOR per
// pivot), remembers for each which rows took it (four lane masks), and only then touches the whole rows: the four pivot rows go
// to LDS as they are found, are brought to the state the sequential walk would have left them in (row i takes rows j < i
// according to the masks: uniform conditions), their sixteen sums are made by the lanes in a (dword, quarter) arrangement --
// T[c] = sum of the rows in c -- and every row adds the ONE sum its four mask bits name: a 16-byte LDS read and four XORs per piece
// for four pivots.  A block ends with the dword, and with the last unused row.  Contiguous rows of whole 16-byte pieces only
// (ld == LD >= 2, 16-byte aligned): everything else takes rref_small_kernel.  Bit-identical to it (RREF is unique; the pivot
// rows are the same ones: lowest lane of the lowest register row).  Which pivots of the block a row has taken is a per-lane
// value (`mul`), and what a pivot's row took of the earlier ones is its `mul` when it became one -- its lane leaves it in LDS
// next to the row: no scalar bookkeeping per pivot (selecting "the masks of step i" by a counter cost twenty scalar instructions
// per pivot in the first form, writing the four steps out as many for their flags in the second).
template <int RPL, int LD>
__global__ __launch_bounds__(64 * SMALL_WAVES) __attribute__((amdgpu_waves_per_eu(RPL * LD <= 8 ? 8 : (RPL * LD <= 16 ? 4 : 2), 8))) void rref_small_m4r_kernel(u64* __restrict__ base, int64_t batch, int m, int n,
                                                                          int64_t* __restrict__ pivots_base, int64_t cap,
                                                                          int64_t* __restrict__ rank_out) {
    constexpr int DW = 2 * LD;                                      // dwords per row (a multiple of 4)
    static_assert(DW >= 4 && DW <= 64 && (DW & (DW - 1)) == 0, "rows of 1 .. 16 whole 16-byte pieces");
    constexpr int GROUPS = 64 / DW;                                 // quarters of the sixteen sums a pass of the lanes makes
    constexpr int PASSES = GROUPS >= 4 ? 1 : 4 / GROUPS;
    // per wavefront: five row slots (the block's four pivot rows as they stood when it began, written when it ends -- every lane
    // stores its rows, a pivot's lane into the pivot's slot and everybody else into the fifth: no exec region, no one-lane stores),
    // the sixteen sums, and per pivot of the block what its row took of the earlier ones
    __shared__ __align__(16) unsigned int lds_all[SMALL_WAVES][5 * DW + 16 * DW + 4];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned int* const prow = lds_all[wv];
    unsigned int* const sums = lds_all[wv] + 5 * DW;
    unsigned int* const meta = lds_all[wv] + 21 * DW;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int my_dd = lane & (DW - 1), my_group = lane / DW;
    // (one wavefront: the LDS serves its operations in order, so what orders a write before another lane's read is the order of
    // the instructions -- which is all this has to keep)
    auto lds_sync = []() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };
    for (int64_t mat = wave; mat < batch; mat += nwaves) {
        u64* a = base + mat * m * LD;
        unsigned int w[RPL][DW];
        int pivcol[RPL], myrank[RPL];
        u64 unused[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            pivcol[q] = -1;
            myrank[q] = 0;
            unused[q] = __ballot(lane + 64 * q < m);
            const int row = lane + 64 * q;
            const u32x4* src = reinterpret_cast<const u32x4*>(a + (int64_t)row * LD);
#pragma unroll
            for (int j = 0; j < LD / 2; ++j) {
                u32x4 v = {0, 0, 0, 0};
                if (row < m) v = src[j];
                w[q][4 * j] = v.x, w[q][4 * j + 1] = v.y, w[q][4 * j + 2] = v.z, w[q][4 * j + 3] = v.w;
            }
        }
        int rank = 0;
#pragma unroll
        for (int d = 0; d < DW; ++d) {
            if (d * 32 >= n || rank >= m) break;                    // uniform
            const int d4 = d & ~3;                                  // a pivot row is zero before its column: pieces from here on
            const int nbits = n - d * 32 < 32 ? n - d * 32 : 32;    // uniform
            unsigned int x[RPL];                                    // the dword of the columns, ahead of the rows by the block's pivots
            unsigned int mul[RPL];                                  // which pivots of the block the row has taken (bit i: pivot i)
            unsigned int slot[RPL];                                 // the row's slot in LDS: the pivot's number, 4 for everybody else
#pragma unroll
            for (int q = 0; q < RPL; ++q) x[q] = w[q][d], mul[q] = 0, slot[q] = 4;
            int npiv = 0;                                           // pivots in the block so far (uniform)
            auto flush = [&]() {
                // the rows have not changed since the block began: its pivots' rows as the block found them
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    unsigned int* const mine = prow + slot[q] * DW;
#pragma unroll
                    for (int dd = 0; dd < DW; dd += 4) {
                        if (dd < d4) continue;
                        *reinterpret_cast<u32x4*>(mine + dd) = u32x4{w[q][dd], w[q][dd + 1], w[q][dd + 2], w[q][dd + 3]};
                    }
                    slot[q] = 4;
                }
                lds_sync();
                const u32x4 under = *reinterpret_cast<const u32x4*>(meta);              // (one address: the same in every lane)
                const unsigned int p0 = prow[my_dd];
                unsigned int p1 = prow[DW + my_dd], p2 = prow[2 * DW + my_dd], p3 = prow[3 * DW + my_dd];
                // a ^ (b & mask), mask = all ones where the bit is set (slots the block did not fill name older rows: nobody takes
                // them, `mul` has no bit for them)
                auto add_if = [](unsigned int acc, unsigned int row, unsigned int bits, int bit) {
                    const unsigned int mask = 0u - ((bits >> bit) & 1u);
                    return __builtin_amdgcn_bitop3_b32(acc, row, mask, 0x78);
                };
                p1 = add_if(p1, p0, under.y, 0);
                p2 = add_if(p2, p0, under.z, 0);
                p2 = add_if(p2, p1, under.z, 1);
                p3 = add_if(p3, p0, under.w, 0);
                p3 = add_if(p3, p1, under.w, 1);
                p3 = add_if(p3, p2, under.w, 2);
#pragma unroll
                for (int pass = 0; pass < PASSES; ++pass) {
                    const int cg = (pass * GROUPS + my_group) & 3;
                    const unsigned int high = ((cg & 1) ? p2 : 0u) ^ ((cg & 2) ? p3 : 0u);
                    unsigned int* const out = sums + (4 * cg) * DW + my_dd;
                    out[0] = high;
                    out[DW] = high ^ p0;
                    out[2 * DW] = high ^ p1;
                    out[3 * DW] = high ^ p0 ^ p1;
                }
                lds_sync();
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    const unsigned int* const mine = sums + mul[q] * DW;
#pragma unroll
                    for (int dd = 0; dd < DW; dd += 4) {
                        if (dd < d4) continue;
                        const u32x4 v = *reinterpret_cast<const u32x4*>(mine + dd);
                        w[q][dd] ^= v.x, w[q][dd + 1] ^= v.y, w[q][dd + 2] ^= v.z, w[q][dd + 3] ^= v.w;
                    }
                    x[q] = w[q][d];
                    mul[q] = 0;
                }
                rank += npiv;
                npiv = 0;
                lds_sync();                                         // (the next block's rows and sums are written after these reads)
            };
            int limit = nbits;                                      // (0 once every row is a pivot row: one condition for the loop)
#pragma unroll 1
            for (int bb = 0; bb < limit; ++bb) {
                const unsigned int bit = 1u << bb;
                u64 has[RPL];
                int src_q = -1;
                u64 cand = 0;
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    has[q] = __ballot((x[q] & bit) != 0);
                    if (src_q < 0 && (has[q] & unused[q])) {
                        src_q = q;
                        cand = has[q] & unused[q];
                    }
                }
                if (src_q >= 0) {                                   // a pivot column
                    const int src_lane = __ffsll((long long)cand) - 1;
                    const u64 src_bit = 1ull << src_lane;
                    unsigned int px = 0, took = 0;
#pragma unroll
                    for (int q = 0; q < RPL; ++q) {
                        if (q != src_q) continue;                   // uniform
                        px = (unsigned int)__builtin_amdgcn_readlane((int)x[q], src_lane);
                        took = (unsigned int)__builtin_amdgcn_readlane((int)mul[q], src_lane);
                        const bool me = lane == src_lane;
                        pivcol[q] = me ? d * 32 + bb : pivcol[q];
                        myrank[q] = me ? rank + npiv : myrank[q];
                        slot[q] = me ? (unsigned int)npiv : slot[q];
                    }
                    meta[npiv] = took;                              // (every lane the same value)
                    const unsigned int mine = 1u << npiv;
#pragma unroll
                    for (int q = 0; q < RPL; ++q) {
                        const u64 tk = q == src_q ? has[q] & ~src_bit : has[q];
                        if (__builtin_amdgcn_inverse_ballot_w64(tk)) {
                            x[q] ^= px;
                            mul[q] |= mine;
                        }
                    }
                    unused[src_q] ^= src_bit;                       // (the bit is set: the pivot came from there)
                    npiv += 1;
                    if (rank + npiv >= m) limit = 0;                // uniform
                }
                // a block ends with its fourth pivot, with the dword and with the last unused row (ONE place: the code of a block's
                // end is as long as the rest of the loop)
                if (npiv == 4 || (npiv > 0 && bb + 1 >= limit)) flush();       // uniform
            }
        }
        // rows out in pivot order; everything from row `rank` on is zero
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            const int row = lane + 64 * q;
            if (pivcol[q] >= 0) {
                u32x4* dst = reinterpret_cast<u32x4*>(a + (int64_t)myrank[q] * LD);
#pragma unroll
                for (int j = 0; j < LD / 2; ++j) dst[j] = u32x4{w[q][4 * j], w[q][4 * j + 1], w[q][4 * j + 2], w[q][4 * j + 3]};
                if (pivots_base) pivots_base[mat * cap + myrank[q]] = pivcol[q];
            }
            if (row >= rank && row < m) {
                u32x4* dst = reinterpret_cast<u32x4*>(a + (int64_t)row * LD);
#pragma unroll
                for (int j = 0; j < LD / 2; ++j) dst[j] = u32x4{0, 0, 0, 0};
            }
        }
        if (lane == 0) rank_out[mat] = rank;
    }
}

template <int RPL, int LD>
static int launch_rref_small(gf2_ctx* ctx, u64* a_dev, int64_t batch, int64_t m, int64_t n, int64_t ld, int64_t* pivots_dev,
                             int64_t cap, int64_t* rank_dev) {
    int64_t blocks = gf2_cdiv(batch, SMALL_WAVES);
    if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_ELIM));
    // dwords of the pivot row by v_readlane, the rest through LDS (see the kernel): GF2_OPT_RREF_SMALL_BCAST 0 = all through LDS,
    // 1 = all by readlane where the registers allow (up to two rows per lane), default = half and half
    const int64_t how = ctx->opt[GF2_OPT_RREF_SMALL_BCAST];
    const dim3 grid((unsigned)blocks), block(64 * SMALL_WAVES);
    // Four pivots at a time (rows that are contiguous whole 16-byte pieces, at most 32 words of them per lane: registers): the
    // default for rows of 16 words -- 256 MiB of 64 x 1024 matrices in 0.36 instead of 0.76 ms, of 128 x 1024 in 0.64 instead of
    // 11.5 (one pivot at a time spills there).  Rows of 8 words are a draw (64 x 512: 0.346 against 0.333 ms), two or four of
    // them per lane and narrower rows lose (128 x 512: 0.65 against 0.60, 256 x 512: 1.15 against 1.04, 64 x 128: 1.03 against
    // 0.76 -- profiles/r04_rref_small_m4r.log: the block's fixed cost against what the row is worth);
    // GF2_OPT_RREF_SMALL_BCAST = 2 takes it wherever it is built.
    if constexpr (LD >= 2 && RPL * LD <= 32) {
        if (((how < 0 && LD == 16) || how == 2) && ld == LD && (reinterpret_cast<uintptr_t>(a_dev) & 15) == 0) {
            hipLaunchKernelGGL((rref_small_m4r_kernel<RPL, LD>), grid, block, 0, ctx->stream, a_dev, batch, (int)m, (int)n, pivots_dev, cap,
                               rank_dev);
            GF2_TRY(gf2_prof_end(ctx));
            GF2_HIP(hipGetLastError());
            return GF2_OK;
        }
    }
    if (LD < 2)                                                     // (a row of two dwords is no 16-byte piece: by readlane)
        hipLaunchKernelGGL((rref_small_kernel<RPL, LD, 2 * LD>), grid, block, 0, ctx->stream, a_dev, batch, (int)m, (int)n, ld, pivots_dev,
                           cap, rank_dev);
    else if (how == 0)
        hipLaunchKernelGGL((rref_small_kernel<RPL, LD, 0>), grid, block, 0, ctx->stream, a_dev, batch, (int)m, (int)n, ld, pivots_dev, cap,
                           rank_dev);
    else if (how == 1 && RPL <= 2)
        hipLaunchKernelGGL((rref_small_kernel<RPL, LD, 2 * LD>), grid, block, 0, ctx->stream, a_dev, batch, (int)m, (int)n, ld, pivots_dev,
                           cap, rank_dev);
    else
        // one row of at most 16 dwords per lane: half and half (256 MiB of 64 x 512 matrices: 0.43 ms; all by readlane 0.49, all
        // through LDS 0.58); longer rows or more rows per lane: all through LDS (64 x 1024: 0.83 against 1.10 and 1.22 ms; 128 x 512:
        // 0.72 against 0.76 and 0.79) -- gpurun_out/r04/rref_small_hybrid.log
        hipLaunchKernelGGL((rref_small_kernel<RPL, LD, (RPL == 1 && LD <= 8 ? LD : 0)>), grid, block, 0, ctx->stream, a_dev, batch, (int)m,
                           (int)n, ld, pivots_dev, cap, rank_dev);
    GF2_TRY(gf2_prof_end(ctx));
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

static int launch_eliminate(gf2_ctx* ctx, int mode, u64* a_dev, int64_t batch, int64_t m, int64_t n, int64_t ld,
                            int64_t offset, int64_t* pivots_dev, int64_t pivots_stride, int64_t* rank_dev,
                            int64_t* swaps_dev, int64_t* nswaps_dev, int* status_dev) {
    if (ld > ELIM_MAX_LD) GF2_FAIL(GF2_E_ARG, "elimination supports at most %d columns (ld=%lld)", ELIM_MAX_LD * 64, (long long)ld);
    if (m >= 0x7fffffffLL || n >= 0x7fffffffLL) GF2_FAIL(GF2_E_ARG, "elimination: matrix too large");
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_ELIM));
    if (mode == ELIM_RREF)
        hipLaunchKernelGGL(eliminate_kernel<ELIM_RREF>, dim3((unsigned)batch), dim3(ELIM_THREADS), 0, ctx->stream, a_dev, m, n,
                           ld, offset, pivots_dev, pivots_stride, rank_dev, swaps_dev, nswaps_dev, status_dev,
                           (RrefState*)nullptr);
    else
        hipLaunchKernelGGL(eliminate_kernel<ELIM_NORMALIZE>, dim3((unsigned)batch), dim3(ELIM_THREADS), 0, ctx->stream, a_dev,
                           m, n, ld, offset, pivots_dev, pivots_stride, rank_dev, swaps_dev, nswaps_dev, status_dev,
                           (RrefState*)nullptr);
    GF2_TRY(gf2_prof_end(ctx));
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

// Blocked path for matrices of at most 4096 rows, K panels per sweep (rref_sweep_panel_kernel / rref_sweep_update_kernel), all on
// the context's stream.  Workspace: a copy of the batch (the first pass moves the batch there, the row gather writes it back in
// place), pivot-row lists, states, used flags, per panel of a sweep the folded coefficients and the pivot-row snapshots, and the
// side buffer of the next sweep's column words.
// (Cutting a batch into up to four groups of matrices on streams of their own, so that the panels of one group -- one workgroup per
// matrix, mostly one wavefront of it at work -- run under the trailing passes of the others, was built and measured, also with the
// panel kernel held to 64 registers and the pass at 512 threads so that one of each fits a CU: 3.0 - 4.3 ms against 3.1 for 256
// matrices of 2048 x 4096.  The pass is bound by the LDS of its CU and a panel workgroup next to it loses more than the overlap
// gives; profiles/r05_groups.md.)
template <int K>
static int launch_rref_sweeps(gf2_ctx* ctx, u64* a_dev, int64_t batch, int64_t m, int64_t n, int64_t ld, int64_t* pivots_dev,
                              int64_t cap, int64_t* rank_dev) {
    constexpr int CW = 64 / K, TH = RB_THREADS;
    const int rpt = (int)gf2_cdiv(m, RB_THREADS);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t abytes = al((size_t)batch * m * ld * 8), pbytes = al((size_t)batch * cap * 4);
    const size_t sbytes = al((size_t)batch * sizeof(SweepState)), ubytes = al((size_t)batch * m);
    const size_t dbytes = al((size_t)batch * m * 8), nbytes = al((size_t)batch * 64 * ld * 8);
    const size_t cbytes = al((size_t)batch * m * K * 8);
    GF2_TRY(gf2_ws_reserve(ctx, 1, abytes + pbytes + sbytes + ubytes + K * (dbytes + nbytes) + cbytes));
    char* q = (char*)ctx->ws[1];
    u64* tmp = (u64*)q; q += abytes;
    int32_t* pivrow = (int32_t*)q; q += pbytes;
    SweepState* states = (SweepState*)q; q += sbytes;
    unsigned char* used = (unsigned char*)q; q += ubytes;
    u64* dco = (u64*)q; q += K * dbytes;
    u64* snap = (u64*)q; q += K * nbytes;
    u64* colw = (u64*)q;
    const int64_t dstride = (int64_t)(dbytes / 8), sstride = (int64_t)(nbytes / 8);
    hipStream_t on = ctx->stream;
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_ELIM));
    GF2_HIP(hipMemsetAsync(states, 0, sbytes + ubytes, on));                  // rank = 0, used = 0 ...
    hipLaunchKernelGGL(sweep_state_init_kernel, dim3((unsigned)gf2_cdiv(batch, 256)), dim3(256), 0, on, states, batch, n);
    if (!ctx->lds_optin[K == 2 ? 5 : 6]) {
        GF2_HIP(hipFuncSetAttribute((const void*)rref_sweep_update_kernel<K, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ctx->lds_optin[K == 2 ? 5 : 6] = true;
    }
    const int64_t panels = gf2_words(n) < ld ? gf2_words(n) : ld;
    const int64_t chunks = gf2_cdiv(ld, CW);
    const int64_t sweeps = gf2_cdiv(panels, K);
    // rows per update workgroup: a workgroup owns its CU (128 KiB of tables) and builds the tables before it moves a row -- as long
    // as moving 300 rows -- so: one round of the chip if the batch is small (128 rows at least), whole matrices (2048 rows or more) if it is large
    // (one matrix of 2048 x 4096: 0.85 / 0.78 / 0.75 / 0.75 ms with 512 / 256 / 128 / 64 rows; eight of them 0.85 / 0.82 / 0.90 with 256 / 128 / 64)
    int64_t rows_wg = gf2_cdiv(gf2_cdiv(m * chunks * batch, (int64_t)ctx->num_cus), 128) * 128;
    if (rows_wg > 128) rows_wg = gf2_cdiv(m, gf2_cdiv(m, rows_wg));
    if (rows_wg < 128) rows_wg = 128;
    if (ctx->opt[GF2_OPT_RREF_ROWS_WG] >= 64) rows_wg = ctx->opt[GF2_OPT_RREF_ROWS_WG];
    // A pass workgroup that owns ALL rows of its chunk (large batches) reads the sweep's pivot rows where they lie, for its tables,
    // before it writes a row: the panel kernel then leaves their numbers instead of copies of them (256 matrices of 2048 x 4096:
    // 128 KiB read and written per matrix and sweep less; GF2_OPT_RREF_STREAM_VARIANT = 1 keeps the snapshots).
    const bool direct = rows_wg >= m && ctx->opt[GF2_OPT_RREF_STREAM_VARIANT] != 1;
    int32_t* const prow = (int32_t*)snap;                              // (the snapshots' place: [batch][K][64] numbers)
    // Every row may have its pivot once m columns have been seen, and a random matrix is done right there or a few columns later:
    // from then on the ranks are read back after every sweep for two sweeps, then after every other one (a stream synchronisation,
    // but it saves the launches of the sweeps that would find nothing left to do -- half of them for a 2048 x 4096 matrix).
    auto all_done = [&](int64_t pw_last, bool* done) -> int {
        *done = false;
        const int64_t past = (pw_last + 1) * 64 - m;
        if (!(past >= 0 && pw_last + 1 < panels && (past < 128 * K || past % (128 * K) < 64 * K))) return GF2_OK;
        std::vector<SweepState> now((size_t)batch);
        GF2_HIP(hipMemcpyAsync(now.data(), states, (size_t)batch * sizeof(SweepState), hipMemcpyDeviceToHost, on));
        GF2_TRY(gf2_stream_wait(on));
        *done = true;
        for (const auto& st : now) *done = *done && st.rank >= m;
        return GF2_OK;
    };
    // (A panel workgroup alone on its CU -- 40 or 90 KiB of unused dynamic LDS so that no second one fits -- changes nothing: 2.61 -
    // 2.65 ms against 2.64 - 2.66 for 256 matrices of 2048 x 4096; the panel kernel's phases are latency, not contention.)
    for (int64_t s = 0; s < sweeps; ++s) {
        const int64_t pw0 = s * K;
        u64* work = s == 0 ? a_dev : tmp;                              // the first pass takes the batch to the workspace copy
#define GF2_SP_LAUNCH(RPT)                                                                                                      \
    hipLaunchKernelGGL((rref_sweep_panel_kernel<K, RPT>), dim3((unsigned)batch), dim3(RB_THREADS), 0, on, (const u64*)work, m, n, ld, pw0, \
                       pivots_dev, cap, pivrow, states, used, (const u64*)colw, dco, dstride, direct ? (u64*)nullptr : snap, sstride, prow)
        if (rpt <= 1)
            GF2_SP_LAUNCH(1);
        else if (rpt <= 2)
            GF2_SP_LAUNCH(2);
        else
            GF2_SP_LAUNCH(4);
#undef GF2_SP_LAUNCH
        const dim3 grid((unsigned)gf2_cdiv(m, rows_wg), (unsigned)chunks, (unsigned)batch);
        hipLaunchKernelGGL((rref_sweep_update_kernel<K, TH>), grid, dim3(TH), 128 * 1024, on, work, m, ld, rows_wg, (const SweepState*)states,
                           states, (const u64*)dco, dstride, direct ? (const u64*)nullptr : (const u64*)snap, sstride, pw0, colw, tmp, 0, -1,
                           (const int32_t*)prow);
        GF2_HIP(hipGetLastError());
        bool done;
        GF2_TRY(all_done(pw0 + K - 1, &done));
        if (done) break;
    }
    hipLaunchKernelGGL(gather_rows_kernel<SweepState>, dim3((unsigned)gf2_cdiv(m, GATHER_ROWS), (unsigned)batch), dim3(256), 0, on,
                       (const u64*)tmp, a_dev, (const int32_t*)pivrow, (const SweepState*)states, rank_dev, m, ld, cap);
    GF2_HIP(hipGetLastError());
    GF2_TRY(gf2_prof_end(ctx));
#if GF2_SWEEP_DIAG
    if (getenv("GF2_RREF_DIAG")) {
        u64 dg[8];
        GF2_HIP(hipStreamSynchronize(on));
        GF2_HIP(hipMemcpyFromSymbol(dg, HIP_SYMBOL(g_sweep_diag), sizeof(dg)));
        const double n_wg = (double)(dg[4] ? dg[4] : 1);
        fprintf(stderr, "sweep pass K=%d: %.0f workgroups; table build %.0f cycles / %.2f us (%.2f GHz); rows %.0f cycles / %.2f us (%.2f GHz), wavefront 0\n",
                K, n_wg, dg[0] / n_wg, dg[1] / n_wg / 100.0, dg[1] ? dg[0] / (dg[1] * 10.0) : 0.0, dg[2] / n_wg, dg[3] / n_wg / 100.0,
                dg[3] ? dg[2] / (dg[3] * 10.0) : 0.0);
        u64 pd[8];
        GF2_HIP(hipMemcpyFromSymbol(pd, HIP_SYMBOL(g_panel_diag), sizeof(pd)));
        const double np = (double)(pd[7] ? pd[7] : 1);
        fprintf(stderr, "sweep panel K=%d: %.0f launches; us per launch: prologue %.2f, window fill %.2f, window_round %.2f, round tables %.2f, "
                "publish + later columns %.2f, epilogue %.2f\n", K, np, pd[0] / np / 100, pd[1] / np / 100, pd[2] / np / 100, pd[3] / np / 100,
                pd[4] / np / 100, pd[5] / np / 100);
        memset(pd, 0, sizeof(pd));
        GF2_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_panel_diag), pd, sizeof(pd)));
        if (getenv("GF2_RREF_DIAG_WG")) {
            std::vector<u64> wg(4 * 4096);
            GF2_HIP(hipMemcpyFromSymbol(wg.data(), HIP_SYMBOL(g_sweep_wg), wg.size() * 8));
            FILE* f = fopen(getenv("GF2_RREF_DIAG_WG"), "w");
            if (f) {
                for (int i = 0; i < 4096; ++i)
                    if (wg[4 * i]) fprintf(f, "%d %llu %llu %llu %llx\n", i, wg[4 * i], wg[4 * i + 1], wg[4 * i + 2], wg[4 * i + 3]);
                fclose(f);
            }
        }
        memset(dg, 0, sizeof(dg));
        dg[7] = getenv("GF2_RREF_DIAG_SWEEP") ? (u64)atoll(getenv("GF2_RREF_DIAG_SWEEP")) : 8;
        GF2_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_sweep_diag), dg, sizeof(dg)));
    }
#endif
    return GF2_OK;
}

// Blocked path for matrices of more than 4096 rows, four panels per sweep, rows streamed (sweep_stream_panel_kernel & co. above),
// with LOOK-AHEAD: the next sweep's panels run on the high-priority side stream under this sweep's trailing pass.
//   main | snapshot s | pass s, chunk c only (128-row workgroups) |  pass s, every other chunk                       | snapshot s+1 ...
//   side |                                        wait . . . . . . | column, (panel, finish) x 4 of sweep s+1, state copy |
// c = the chunk of 16 words that holds the next sweep's four columns: the pass writes them to the side buffer on the way, and they
// are all the panels need of the matrix.  The pass reads a COPY of the state taken after its sweep's panels, and the two sweeps in
// flight use two sets of coefficients, snapshots and pivot-row lists.
static int launch_rref_sweeps_streamed(gf2_ctx* ctx, u64* a_dev, int64_t batch, int64_t m, int64_t n, int64_t ld, int64_t* pivots_dev,
                                       int64_t cap, int64_t* rank_dev) {
    constexpr int K = 4, CW = 64 / K, TH = RB_THREADS;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t abytes = al((size_t)batch * m * ld * 8), pbytes = al((size_t)batch * cap * 4);
    const size_t sbytes = al((size_t)batch * sizeof(SweepState)), ubytes = al((size_t)batch * m);
    const size_t dbytes = al((size_t)batch * m * 8), nbytes = al((size_t)batch * 64 * ld * 8);
    const size_t cbytes = al((size_t)batch * m * K * 8), lbytes = al((size_t)batch * m * 4);
    const size_t tbytes = al((size_t)batch * 2048 * 8), ubbytes = al((size_t)batch * (K - 1) * 64 * 8), rbytes = al((size_t)batch * K * 64 * 4);
    const size_t set_bytes = K * (dbytes + nbytes) + rbytes + sbytes;
    GF2_TRY(gf2_ws_reserve(ctx, 1, abytes + pbytes + sbytes + ubytes + cbytes + dbytes + lbytes + tbytes + ubbytes + 2 * set_bytes));
    char* q = (char*)ctx->ws[1];
    u64* tmp = (u64*)q; q += abytes;
    int32_t* pivrow = (int32_t*)q; q += pbytes;
    SweepState* states = (SweepState*)q; q += sbytes;
    unsigned char* used = (unsigned char*)q; q += ubytes;
    u64* colw = (u64*)q; q += cbytes;
    u64* cco = (u64*)q; q += dbytes;
    int32_t* slot_of = (int32_t*)q; q += lbytes;
    u64* tabs = (u64*)q; q += tbytes;
    u64* pub = (u64*)q; q += ubbytes;
    struct SweepSet {
        u64* e;                                                        // [K][dstride] folded coefficients
        u64* snap;                                                     // [K][sstride] pivot-row snapshots
        int32_t* prow;                                                 // [batch][K][64]
        SweepState* state_copy;                                        // the state after the sweep's last panel
    } sets[2];
    for (int k = 0; k < 2; ++k) {
        sets[k].e = (u64*)q; q += K * dbytes;
        sets[k].snap = (u64*)q; q += K * nbytes;
        sets[k].prow = (int32_t*)q; q += rbytes;
        sets[k].state_copy = (SweepState*)q; q += sbytes;
    }
    const int64_t dstride = (int64_t)(dbytes / 8), sstride = (int64_t)(nbytes / 8);
    hipStream_t s1 = ctx->stream, s2 = ctx->hi;
    hipEvent_t e_first = ctx->side_ev[0], e_ready = ctx->side_ev[1], e_panels = ctx->side_ev[2];
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_ELIM));
    GF2_HIP(hipMemsetAsync(states, 0, sbytes + ubytes, s1));                  // rank = 0, used = 0 ...
    hipLaunchKernelGGL(sweep_state_init_kernel, dim3((unsigned)gf2_cdiv(batch, 256)), dim3(256), 0, s1, states, batch, n);
    if (!ctx->lds_optin[6]) {
        GF2_HIP(hipFuncSetAttribute((const void*)rref_sweep_update_kernel<K, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ctx->lds_optin[6] = true;
    }
    if (!ctx->lds_optin[7]) {
        GF2_HIP(hipFuncSetAttribute((const void*)sweep_finish_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, K * 2048 * 8));
        ctx->lds_optin[7] = true;
    }
    const int64_t panels = gf2_words(n) < ld ? gf2_words(n) : ld;
    const int64_t chunks = gf2_cdiv(ld, CW);
    const int64_t sweeps = gf2_cdiv(panels, K);
    // Rows per update workgroup of a sweep's big launch: ONE round of the chip less some CUs, over the chunks that are still live
    // (a pass workgroup owns its CU -- 128 KiB of LDS, every register -- so the panel and finish kernels of the next sweep, on the
    // side stream, run only where no pass workgroup sits: with two rounds of workgroups of 4096 rows on every CU the look-ahead
    // gained nothing, 30.6 ms with it and 30.8 without).  The chunks left of
    // the sweep return at once while no pivot-free column has been seen, so the count of live chunks falls as the sweeps go on.
    const bool ahead = !gf2_flag(ctx, GF2_F_RREF_NO_LOOKAHEAD) && chunks >= 2 && s2 != nullptr;
    const bool ahead_on = ahead;
    // development switches (GF2_OPT_RREF_STREAM_VARIANT): bit 0 = the big launch waits until the side stream has started (as first
    // built: 27.0 ms for the 256 MiB matrix against 25.8 without the wait -- with CUs left free the one-workgroup panel kernel finds
    // its place anyway, and the wait is a second cross-queue hand-over of ~20 us per sweep), bits 1-2 = which of the two takes the
    // side stream (below), bits 8.. = CUs left to the side stream + 1
    const int64_t variant = ctx->opt[GF2_OPT_RREF_STREAM_VARIANT] < 0 ? 0 : ctx->opt[GF2_OPT_RREF_STREAM_VARIANT];
    const bool wait_ready = (variant & 1) != 0;
    const int swap_mode = (int)((variant >> 1) & 3);                  // 0: by the estimate, 1: panels always on the side stream, 2: the pass always
    const int reserve_forced = (variant >> 8) > 0 ? (int)(variant >> 8) - 1 : -1;
    // The big launch of sweep s: ONE round of workgroups of whole row blocks (every workgroup of a row block walks the same rows at the
    // same time: HBM sees whole rows), each owning its CU, on all CUs but `reserve`; the next sweep's panels run on the CUs it leaves.
    // The reserve is chosen per sweep from a small model (18 ns per row and workgroup + 10 us of tables; a panel 17 us; a finish
    // launch 6 us + 2.5 us per trip of its lanes over the rows; 20 us for whichever of the two waits on the other queue): few CUs
    // left free mean fewer, longer finish workgroups, and pay while the pass is the longer of the two.  (First build, 40 CUs fixed
    // and 32 finish workgroups: 4 / 12 / 24 / 40 / 56 / 72 / 96 CUs gave 31.5 / 30.1 / 28.0 / 26.4 / 27.5 / 28.7 / 32.2 ms for the
    // 256 MiB matrix -- profiles/r05_streamed_sweeps.md.)
    struct Plan {
        int64_t rows, blocks;
        int64_t finish_wgs;                                            // per matrix; 0: as many as there are rows for
        bool chain_here;                                               // the panels' chain stays on the main stream, the big launch goes aside
        bool flat;                                                     // the big launch as rref_sweep_update_flat_kernel
    };
    const int xcds = ctx->num_cus % 8 == 0 ? 8 : 1, cus_xcd = ctx->num_cus / xcds;
    auto plan_for = [&](int64_t s) -> Plan {
        Plan best{m, 1, 0, false, false};
        const int64_t max_blocks = gf2_cdiv(m, 1024);                  // (no fewer than 1024 rows each: the tables are built per workgroup)
        const int64_t live = (chunks - s * K / CW - (ahead_on ? 1 : 0)) * batch;
        if (ctx->opt[GF2_OPT_RREF_ROWS_WG] >= 64 || !ahead_on || live <= 0) {
            int64_t blocks = live > 0 && ctx->num_cus > live ? ctx->num_cus / live : 1;
            if (blocks > max_blocks) blocks = max_blocks;
            best.rows = ctx->opt[GF2_OPT_RREF_ROWS_WG] >= 64 ? ctx->opt[GF2_OPT_RREF_ROWS_WG] : gf2_cdiv(m, blocks);
            best.blocks = gf2_cdiv(m, best.rows);
            best.chain_here = swap_mode == 2;
            return best;
        }
        int64_t best_cost = -1;
        // CUs every XCD keeps for the other stream: 5 of 32.  Fewer lose although the pass gets shorter (252 workgroups: 158 us against
        // 200 with 189): the panel kernel or a finish workgroup then finds no CU on its XCD and waits for the whole pass (150 us) --
        // 1 / 2 / 3 / 5 / 7 per XCD: 27.6 / 26.9 / 25.7 / 23.7 / 25.1 ms for the 256 MiB matrix, also with the live workgroups first
        // in a one-dimensional grid (the flat kernel) and finish launches cut to the CUs left.
        for (int left = 5; left <= 5; ++left) {
            if (reserve_forced >= 0) left = reserve_forced / xcds > 0 ? reserve_forced / xcds : 1;
            const int64_t cap_wgs = (int64_t)xcds * (cus_xcd - left);
            int64_t blocks = cap_wgs > live ? cap_wgs / live : 1;
            if (blocks > max_blocks) blocks = max_blocks;
            const int64_t rows = gf2_cdiv(m, blocks);
            blocks = gf2_cdiv(m, rows);
            const int64_t wgs = blocks * live, per_xcd = gf2_cdiv(wgs, (int64_t)xcds);
            const int64_t left_now = per_xcd < cus_xcd ? cus_xcd - per_xcd : 0;     // free CUs of the fullest XCD
            const int64_t rounds = left_now > 0 ? 1 : gf2_cdiv(wgs, (int64_t)ctx->num_cus);
            const int64_t pass_us = rounds * (10 + rows * 18 / 1000);
            const int64_t fin_all = gf2_cdiv(m, 1024) * batch;
            int64_t fin_wgs = fin_all, trips = 1;
            if (left_now > 0) {
                if (fin_wgs > 2 * left_now * xcds) fin_wgs = 2 * left_now * xcds;       // (1024 lanes, 66 KiB of LDS: two to a CU)
                trips = gf2_cdiv(fin_all, fin_wgs);
            } else {
                trips = 8;                                              // (they wait for pass workgroups to leave)
            }
            const int64_t chain_us = K * (17 + 6 + 5 * trips / 2) + 5;
            const int64_t aside = pass_us + 7 > chain_us + 20 ? pass_us + 7 : chain_us + 20;       // panels on the side stream
            const int64_t here = pass_us + 20 > chain_us ? pass_us + 20 : chain_us;               // panels on the main stream
            const bool chain_here = swap_mode == 2 || (swap_mode == 0 && here < aside);
            const int64_t cost = chain_here ? here : aside;
            if (best_cost < 0 || cost <= best_cost)
                best_cost = cost, best = Plan{rows, blocks, left_now > 0 ? (fin_wgs / batch > 0 ? fin_wgs / batch : 1) : 0, chain_here, true};
            if (reserve_forced >= 0) break;
        }
        return best;
    };
    if (!ctx->lds_optin[8]) {
        GF2_HIP(hipFuncSetAttribute((const void*)rref_sweep_update_flat_kernel<K, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ctx->lds_optin[8] = true;
    }
    // the K panels of sweep `s` into set `ps`, on stream `on`; `work`: where the batch lives (only read when the side buffer is stale)
    // (The column kernel runs for the first sweep only: afterwards the pass of sweep s always leaves the column words of sweep
    // s + 1 in the side buffer -- also for a matrix whose sweep found no pivot -- and sweep_finish_kernel resets the scratch.)
    auto launch_panels = [&](const SweepSet& ps, int64_t s, const u64* work, hipStream_t on, int64_t finish_plan) {
        const int64_t pw0 = s * K;
        // finish workgroups per matrix: one round on the CUs the pass leaves (the plan), or one per 1024 rows
        const int64_t finish_wgs = finish_plan > 0 && finish_plan < gf2_cdiv(m, 1024) ? finish_plan : gf2_cdiv(m, 1024);
        if (s == 0)
            hipLaunchKernelGGL(sweep_column_kernel, dim3((unsigned)gf2_cdiv(m, 256), (unsigned)batch), dim3(256), 0, on, work, m, ld, pw0, K,
                               (const SweepState*)states, colw, cco, slot_of);
        for (int l = 0; l < K; ++l) {
            hipLaunchKernelGGL(sweep_stream_panel_kernel<K>, dim3((unsigned)batch), dim3(RB_THREADS), 0, on, m, n, ld, pw0, l, pivots_dev, cap,
                               pivrow, states, used, ps.prow, colw, cco, slot_of, tabs, (const u64*)ps.e, dstride, pub);
            hipLaunchKernelGGL(sweep_finish_kernel<K>, dim3((unsigned)finish_wgs, (unsigned)batch), dim3(1024), K * 2048 * 8, on, m, l,
                               (const SweepState*)states, colw, cco, slot_of, (const u64*)tabs, ps.e, dstride, (const u64*)pub);
        }
        return hipMemcpyAsync(ps.state_copy, states, (size_t)batch * sizeof(SweepState), hipMemcpyDeviceToDevice, on);
    };
    auto launch_pass = [&](const SweepSet& ps, int64_t s, u64* work, int64_t c_lo, int64_t c_hi, int64_t skip, int64_t rows, hipStream_t on) {
        if (c_hi <= c_lo || (c_hi - c_lo == 1 && skip == c_lo)) return;
        const dim3 grid((unsigned)gf2_cdiv(m, rows), (unsigned)(c_hi - c_lo), (unsigned)batch);
        hipLaunchKernelGGL((rref_sweep_update_kernel<K, TH>), grid, dim3(TH), 128 * 1024, on, work, m, ld, rows, (const SweepState*)ps.state_copy,
                           states, (const u64*)ps.e, dstride, (const u64*)ps.snap, sstride, s * K, colw, tmp, (int)c_lo, (int)skip);
    };
    auto all_done = [&](int64_t pw_last, bool* done) -> int {
        *done = false;
        const int64_t past = (pw_last + 1) * 64 - m;
        if (!(past >= 0 && pw_last + 1 < panels && (past < 128 * K || past % (128 * K) < 64 * K))) return GF2_OK;
        std::vector<SweepState> now((size_t)batch);
        GF2_HIP(hipMemcpyAsync(now.data(), states, (size_t)batch * sizeof(SweepState), hipMemcpyDeviceToHost, s1));
        GF2_TRY(gf2_stream_wait(s1));
        *done = true;
        for (const auto& st : now) *done = *done && st.rank >= m;
        return GF2_OK;
    };
    // sweep 0: nothing to overlap with
    GF2_HIP(launch_panels(sets[0], 0, a_dev, s1, 0));
    int64_t last = sweeps - 1;                                          // the last sweep that has work (lowered once every rank is m)
    for (int64_t s = 0; s <= last; ++s) {
        const SweepSet& cur = sets[s & 1];
        const SweepSet& nxt = sets[(s + 1) & 1];
        u64* work = s == 0 ? a_dev : tmp;                              // the first pass takes the batch to the workspace copy
        hipLaunchKernelGGL(sweep_snapshot_kernel, dim3((unsigned)(K * 64), (unsigned)batch), dim3(1024), 0, s1, (const u64*)work, m, ld, K,
                           (const SweepState*)cur.state_copy, (const int32_t*)cur.prow, cur.snap, sstride);
        const Plan plan = plan_for(s);
        if (s == last) {
            launch_pass(cur, s, work, 0, chunks, -1, plan.rows, s1);
            break;
        }
        const int64_t cnext = (s + 1) * K / CW;                        // the chunk of the next sweep's columns
        if (ahead) {
            launch_pass(cur, s, work, cnext, cnext + 1, -1, 128, s1);
            GF2_HIP(hipEventRecord(e_first, s1));
            GF2_HIP(hipStreamWaitEvent(s2, e_first, 0));
            // Whatever goes to the other queue starts ~20 us late (the hand-over).  While the pass is the longer of the two, that is the
            // panels; once the panels' chain (K x (panel + finish) ~ 110 us) outlasts the pass -- the later sweeps of a big matrix, every
            // sweep of a batch of matrices of 8192 rows -- the chain stays on this stream and the big launch takes the hand-over.
            const int64_t rows_big = plan.rows;
            const bool chain_here = plan.chain_here;
            hipStream_t s_chain = chain_here ? s1 : s2, s_pass = chain_here ? s2 : s1;
            if (wait_ready && !chain_here) GF2_HIP(hipEventRecord(e_ready, s2));
            GF2_HIP(launch_panels(nxt, s + 1, tmp, s_chain, plan.finish_wgs));
            if (!chain_here) GF2_HIP(hipEventRecord(e_panels, s2));
            if (wait_ready && !chain_here) GF2_HIP(hipStreamWaitEvent(s1, e_ready, 0));
            if (plan.flat && s * K / CW <= cnext) {
                const int c_first = (int)(s * K / CW);                  // chunks left of it return at once (no pivot-free column so far)
                const int64_t wgs = plan.blocks * (chunks - 1) * batch;
                hipLaunchKernelGGL((rref_sweep_update_flat_kernel<K, TH>), dim3((unsigned)wgs), dim3(TH), 128 * 1024, s_pass, work, m, ld, rows_big,
                                   (int)plan.blocks, (int)chunks, c_first, (int)cnext, (const SweepState*)cur.state_copy, states, (const u64*)cur.e,
                                   dstride, (const u64*)cur.snap, sstride, s * K, colw, tmp, (int)batch);
            } else {
                launch_pass(cur, s, work, 0, chunks, cnext, rows_big, s_pass);
            }
            if (chain_here) GF2_HIP(hipEventRecord(e_panels, s2));
            GF2_HIP(hipStreamWaitEvent(s1, e_panels, 0));
        } else {
            launch_pass(cur, s, work, 0, chunks, -1, plan.rows, s1);
            GF2_HIP(launch_panels(nxt, s + 1, tmp, s1, 0));
        }
        GF2_HIP(hipGetLastError());
        // (the next sweep's panels have run when the ranks are read: the read-back waits for s1, which has waited for them)
        bool done;
        GF2_TRY(all_done((s + 1) * K + K - 1, &done));
        if (done) last = s + 1;                                        // its panels found the last pivots (or nothing): its pass is still due
    }
    hipLaunchKernelGGL(gather_rows_kernel<SweepState>, dim3((unsigned)gf2_cdiv(m, GATHER_ROWS), (unsigned)batch), dim3(256), 0, s1,
                       (const u64*)tmp, a_dev, (const int32_t*)pivrow, (const SweepState*)states, rank_dev, m, ld, cap);
    GF2_HIP(hipGetLastError());
    GF2_TRY(gf2_prof_end(ctx));
    return GF2_OK;
}

extern "C" {

// Blocked path.  Workspace: a copy of the batch for the row gather, pivot-row lists, per-matrix state, used flags, and per SET
// (two of them: with look-ahead the next pair's panels fill one while the trailing pass of this pair reads the other) the
// coefficients d and pivot-row snapshots of the pair's two panels, `fix`, the pivot rows' indices and a copy of the state as the
// pass wants it; for m > 8192 also the streamed panel words / coefficients / slots.
static int launch_rref_blocked(gf2_ctx* ctx, u64* a_dev, int64_t batch, int64_t m, int64_t n, int64_t ld,
                               int64_t* pivots_dev, int64_t cap, int64_t* rank_dev) {
    const int rpt = (int)gf2_cdiv(m, RB_THREADS);
    // up to 4096 rows: K panels per sweep with the rows' column words and coefficients in registers (round 5; eight rows per lane do
    // not fit: 15 ms against 7.8 for four 8192 x 16384 matrices, hence the streamed form above)
    // more than 4096 rows: four panels per sweep with the rows streamed (32768 x 65536: 23.4 - 24.3 ms and half the traffic against the
    // pair kernels' 39.3; 8192 x 16384 x 4: 3.8 against 7.5); GF2_OPT_RREF_SWEEP_K = 0 keeps the pair kernels below
    if (rpt > 4 && ctx->opt[GF2_OPT_RREF_SWEEP_K] != 0)
        return launch_rref_sweeps_streamed(ctx, a_dev, batch, m, n, ld, pivots_dev, cap, rank_dev);
    if (rpt <= 4 && ctx->opt[GF2_OPT_RREF_SWEEP_K] != 0) {
        // Four panels per sweep where the batch streams from HBM (half the trips: 2.67 against 2.71 ms and half the traffic for 256
        // matrices of 2048 x 4096) and a lane holds at most two rows (with four, the four panels' words and coefficients spill:
        // 2.38 against 2.14 ms for eight matrices of 4096 x 8192); two panels per sweep otherwise (one matrix of 2048 x 4096: 0.78
        // against 0.80 ms) -- profiles/r05_rref_dev.log
        int64_t k = ctx->opt[GF2_OPT_RREF_SWEEP_K];
        if (k < 0) k = (rpt <= 2 && batch * m * ld * 8 >= (64ll << 20)) ? 4 : 2;
        if (k == 2) return launch_rref_sweeps<2>(ctx, a_dev, batch, m, n, ld, pivots_dev, cap, rank_dev);
        return launch_rref_sweeps<4>(ctx, a_dev, batch, m, n, ld, pivots_dev, cap, rank_dev);
    }
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t abytes = al((size_t)batch * m * ld * 8), pbytes = al((size_t)batch * cap * 4);
    const size_t sbytes = al((size_t)batch * sizeof(RrefState)), ubytes = al((size_t)batch * m);
    const size_t dbytes = al((size_t)batch * m * 8), nbytes = al((size_t)batch * 64 * ld * 8);
    const bool stream = rpt > 8;
    const int nsets = stream ? 2 : 1;
    const size_t fbytes = al((size_t)batch * 64 * 8), rbytes = al((size_t)batch * 64 * 4);
    const size_t set_bytes = 2 * dbytes + 2 * nbytes + fbytes + 2 * rbytes + sbytes;
    const size_t xbytes = stream ? 2 * dbytes + al((size_t)batch * m * 4) + (size_t)batch * 2048 * 8 : 0;
    GF2_TRY(gf2_ws_reserve(ctx, 1, abytes + pbytes + sbytes + ubytes + nsets * set_bytes + xbytes));
    char* q = (char*)ctx->ws[1];
    u64* tmp = (u64*)q; q += abytes;
    int32_t* pivrow = (int32_t*)q; q += pbytes;
    RrefState* states = (RrefState*)q; q += sbytes;
    unsigned char* used = (unsigned char*)q; q += ubytes;
    struct PairSet {
        u64* dco[2];                                                   // per panel of the pair: coefficients d, pivot-row snapshot
        u64* snap[2];
        u64* fix;
        int32_t* panel_rows[2];
        RrefState* state_copy;                                         // the state after the pair's second panel
    } sets[2];
    for (int k = 0; k < nsets; ++k) {
        PairSet& ps = sets[k];
        ps.dco[0] = (u64*)q; q += dbytes;
        ps.dco[1] = (u64*)q; q += dbytes;
        ps.snap[0] = (u64*)q; q += nbytes;
        ps.snap[1] = (u64*)q; q += nbytes;
        ps.fix = (u64*)q; q += fbytes;
        ps.panel_rows[0] = (int32_t*)q; q += rbytes;
        ps.panel_rows[1] = (int32_t*)q; q += rbytes;
        ps.state_copy = (RrefState*)q; q += sbytes;
    }
    u64* wpan = (u64*)q;
    u64* cco = (u64*)(q + dbytes);
    int32_t* slot_of = (int32_t*)(q + 2 * dbytes);
    u64* tabs = (u64*)(q + 2 * dbytes + al((size_t)batch * m * 4));
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_ELIM));
    GF2_HIP(hipMemsetAsync(states, 0, sbytes + ubytes, ctx->stream));          // rank = 0, used = 0 ...
    hipLaunchKernelGGL(rref_state_init_kernel, dim3((unsigned)gf2_cdiv(batch, 256)), dim3(256), 0, ctx->stream, states, batch, n);
    if (!ctx->lds_optin[4]) {
        GF2_HIP(hipFuncSetAttribute((const void*)rref_update_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ctx->lds_optin[4] = true;
    }
    const int64_t panels = gf2_words(n) < ld ? gf2_words(n) : ld;
    // rows per update workgroup: a lone matrix wants many workgroups, a large batch wants the table build amortised
    const int64_t chunks = gf2_cdiv(ld, U2_CW);
    // (a workgroup owns its CU -- 128 KiB of tables -- and building them takes as long as streaming 300 rows: two workgroups
    // per CU over the whole launch, but no fewer than 128 rows each)
    int64_t rows_per_wg = gf2_cdiv(gf2_cdiv(m * chunks * batch, 2 * (int64_t)ctx->num_cus), 256) * 256;
    if (rows_per_wg > 256) rows_per_wg = gf2_cdiv(m, gf2_cdiv(m, rows_per_wg));
    if (rows_per_wg < 128) rows_per_wg = 128;
    if (m * chunks * batch < 256 * (int64_t)ctx->num_cus) rows_per_wg = 128;
    // The trailing pass of a pair over chunks [c_lo, c_hi) but `skip`, reading set `ps`.
    // `work`: where the batch lives.  The caller's buffer at first; without streamed panels the FIRST pass writes every row into
    // the workspace copy instead of back in place, the reduction goes on there, and the row gather at the end writes straight into
    // the caller's buffer -- instead of gathering into the workspace and copying the batch back (0.1 of 3.3 ms for 256 matrices
    // of 2048 x 4096).
    u64* work = a_dev;
    auto launch_pass = [&](const PairSet& ps, const RrefState* st, int64_t c_lo, int64_t c_hi, int64_t skip, int64_t rows_wg,
                           hipStream_t on) {
        if (c_hi <= c_lo || (c_hi - c_lo == 1 && skip == c_lo)) return;
        const bool move = !stream && work == a_dev;
        const dim3 grid((unsigned)gf2_cdiv(m, rows_wg), (unsigned)(c_hi - c_lo), (unsigned)batch);
        hipLaunchKernelGGL(rref_update_pair_kernel, grid, dim3(RB_THREADS), 128 * 1024, on, work, m, ld, rows_wg, st,
                           (const u64*)ps.dco[0], (const u64*)ps.dco[1], (const u64*)ps.snap[0], (const u64*)ps.snap[1],
                           (const u64*)ps.fix, (int)c_lo, (int)skip, move ? tmp : work);
        if (move) work = tmp;
    };
    // Every row may have its pivot once m columns have been seen, and a random matrix is done right there: from then on the
    // ranks are read back now and then (a stream synchronisation, but it saves the launches of the panels that would find
    // nothing left to do -- half of them for a 2048 x 4096 matrix).
    auto all_done = [&](int64_t pw, bool* done) -> int {
        *done = false;
        if (!((pw + 1) * 64 >= m && pw + 1 < panels && ((pw + 1) * 64 - m) % 512 < 128)) return GF2_OK;
        std::vector<RrefState> now((size_t)batch);
        GF2_HIP(hipMemcpyAsync(now.data(), states, (size_t)batch * sizeof(RrefState), hipMemcpyDeviceToHost, ctx->stream));
        GF2_TRY(gf2_stream_wait(ctx->stream));
        *done = true;
        for (const auto& st : now) *done = *done && st.rank >= m;
        return GF2_OK;
    };
    // Panels go in PAIRS with one trailing pass of the matrix per pair (rref_update_pair_kernel): the second panel brings its
    // own column up to date on the way in and leaves the correction of its pivot rows to the table build of that pass.
    if (!stream) {
        const PairSet& ps = sets[0];
        for (int64_t pw = 0; pw < panels; ++pw) {
            const int member = (int)(pw & 1);
#define GF2_RP_LAUNCH(RPT)                                                                                              \
    hipLaunchKernelGGL((rref_panel_kernel<RPT>), dim3((unsigned)batch), dim3(RB_THREADS), 0, ctx->stream, work, m, n, ld, \
                       pw, pivots_dev, cap, pivrow, states, used, ps.dco[member], ps.snap[member], member, (const u64*)ps.dco[0], \
                       (const u64*)ps.snap[0], ps.fix)
            if (rpt <= 1)
                GF2_RP_LAUNCH(1);
            else if (rpt <= 2)
                GF2_RP_LAUNCH(2);
            else if (rpt <= 4)
                GF2_RP_LAUNCH(4);
            else
                GF2_RP_LAUNCH(8);
#undef GF2_RP_LAUNCH
            if (member == 0 && pw + 1 < panels) continue;             // the pair's second panel first
            launch_pass(ps, states, 0, chunks, -1, rows_per_wg, ctx->stream);
            bool done;
            GF2_TRY(all_done(pw, &done));
            if (done) break;
        }
    } else {
        // m > 8192: the panel step is three kernels (column, the one-workgroup factorisation, finish), 30 us per panel with 255 CUs
        // idle during the middle one -- two thirds of a trailing pass when a single matrix is being reduced.  LOOK-AHEAD: the next
        // pair's panels run on a high-priority side stream UNDER this pair's pass:
        //   main  | pass, chunk c only (128-row workgroups)  . . wait . . | pass, every other chunk (one launch)      | rest of the
        //   side  |                     wait . . . . . . . . | column A' | panel A', finish A', column B', panel B', finish B' | snapshots
        // c is the chunk of 32 words that holds the next pair's two columns: the panels need those up to date, nothing else of the
        // matrix.  The one-workgroup panel kernel (39 KiB of LDS, 96 registers) cannot share a CU with a pass workgroup (128 KiB),
        // so the big launch starts when panel A' is about to (behind its column kernel), and without chunk c it is 16 workgroups
        // short of two rounds of the chip: panel B' finds a CU too.  What is left for after the pass is the snapshot of the new
        // pivot rows outside chunk c, which has to see the finished pass.  The pass reads a COPY of the state taken after its
        // pair's second panel (the look-ahead panels write the state), and the two pairs in flight use two sets of coefficients,
        // snapshots and pivot-row indices.
        hipStream_t s1 = ctx->stream, s2 = ctx->hi;
        hipEvent_t e_first = ctx->side_ev[0], e_column = ctx->side_ev[1], e_panels = ctx->side_ev[2];
        auto launch_column = [&](const PairSet& ps, int64_t pw, int member, hipStream_t on) {
            hipLaunchKernelGGL(panel_column_kernel, dim3((unsigned)gf2_cdiv(m, 256), (unsigned)batch), dim3(256), 0, on,
                               (const u64*)a_dev, m, ld, pw, wpan, cco, slot_of, member, (const RrefState*)states,
                               (const u64*)ps.dco[0], (const u64*)ps.snap[0]);
        };
        // the factorisation and finish kernels of panel pw (member of its pair) into set ps; snapshot words [w_lo, w_hi) now
        auto launch_panel = [&](const PairSet& ps, int64_t pw, int member, int64_t w_lo, int64_t w_hi, hipStream_t on) {
            hipLaunchKernelGGL(rref_panel_stream_kernel, dim3((unsigned)batch), dim3(RB_THREADS), 0, on, a_dev, m, n, ld, pw,
                               pivots_dev, cap, pivrow, states, used, ps.dco[member], ps.panel_rows[member], wpan, cco, slot_of, tabs,
                               member, (const u64*)ps.dco[0], ps.fix);
            hipLaunchKernelGGL(panel_finish_kernel, dim3((unsigned)gf2_cdiv(m, 1024) + 64, (unsigned)batch), dim3(1024), 0, on,
                               (const u64*)a_dev, m, ld, (const RrefState*)states, (const u64*)wpan, (const u64*)cco,
                               (const int32_t*)slot_of, (const u64*)tabs, ps.dco[member], (const int32_t*)ps.panel_rows[member],
                               ps.snap[member], member, 1, w_lo, w_hi, (int64_t)0, (int64_t)0);
        };
        const int64_t npairs = gf2_cdiv(panels, 2);
        // (it pays when a pass outlasts the two panels: 43.1 -> 38.1 ms for one 32768 x 65536 matrix, but 11.9 -> 14.4 ms for
        // 16384 x 32768, whose passes take 30 us and whose panels 64: from 128 MiB of matrix on -- profiles/r03_rref_lookahead.log)
        const bool ahead = !gf2_flag(ctx, GF2_F_RREF_NO_LOOKAHEAD) && chunks >= 2 && s2 != nullptr &&
                           ((chunks >= 8 && m * ld * batch >= (1ll << 24)) || gf2_flag(ctx, GF2_F_RREF_LOOKAHEAD));
        // pair 0: nothing to overlap with
        launch_column(sets[0], 0, 0, s1);
        launch_panel(sets[0], 0, 0, 0, ld, s1);
        if (panels > 1) {
            launch_column(sets[0], 1, 1, s1);
            launch_panel(sets[0], 1, 1, 0, ld, s1);
        }
        GF2_HIP(hipMemcpyAsync(sets[0].state_copy, states, (size_t)batch * sizeof(RrefState), hipMemcpyDeviceToDevice, s1));
        for (int64_t p = 0; p < npairs; ++p) {
            const PairSet& cur = sets[p & 1];
            const PairSet& nxt = sets[(p + 1) & 1];
            if (p + 1 >= npairs) {
                launch_pass(cur, cur.state_copy, 0, chunks, -1, rows_per_wg, s1);
                break;
            }
            const int64_t pa = 2 * (p + 1), pb = pa + 1 < panels ? pa + 1 : -1;   // the next pair's panels
            const int64_t cnext = pa / U2_CW;                                     // both in one chunk (pa is even)
            const int64_t w_lo = cnext * U2_CW, w_hi = (cnext + 1) * U2_CW < ld ? (cnext + 1) * U2_CW : ld;
            if (ahead) {
                launch_pass(cur, cur.state_copy, cnext, cnext + 1, -1, 128, s1);
                GF2_HIP(hipEventRecord(e_first, s1));
                GF2_HIP(hipStreamWaitEvent(s2, e_first, 0));
                launch_column(nxt, pa, 0, s2);
                GF2_HIP(hipEventRecord(e_column, s2));
                launch_panel(nxt, pa, 0, w_lo, w_hi, s2);
                if (pb >= 0) {
                    launch_column(nxt, pb, 1, s2);
                    launch_panel(nxt, pb, 1, w_lo, w_hi, s2);
                }
                GF2_HIP(hipMemcpyAsync(nxt.state_copy, states, (size_t)batch * sizeof(RrefState), hipMemcpyDeviceToDevice, s2));
                GF2_HIP(hipEventRecord(e_panels, s2));
                GF2_HIP(hipStreamWaitEvent(s1, e_column, 0));
                launch_pass(cur, cur.state_copy, 0, chunks, cnext, rows_per_wg, s1);
                GF2_HIP(hipStreamWaitEvent(s1, e_panels, 0));
                hipLaunchKernelGGL(panel_snapshot_rest_kernel, dim3(128, (unsigned)batch), dim3(1024), 0, s1, (const u64*)a_dev, m, ld,
                                   (const RrefState*)states, (const int32_t*)nxt.panel_rows[0], (const int32_t*)nxt.panel_rows[1],
                                   nxt.snap[0], nxt.snap[1], w_lo, w_hi);
            } else {
                launch_pass(cur, cur.state_copy, 0, chunks, -1, rows_per_wg, s1);
                launch_column(nxt, pa, 0, s1);
                launch_panel(nxt, pa, 0, 0, ld, s1);
                if (pb >= 0) {
                    launch_column(nxt, pb, 1, s1);
                    launch_panel(nxt, pb, 1, 0, ld, s1);
                }
                GF2_HIP(hipMemcpyAsync(nxt.state_copy, states, (size_t)batch * sizeof(RrefState), hipMemcpyDeviceToDevice, s1));
            }
            // (the look-ahead panels have run when the ranks are read: the read-back waits for s1, which has waited for them)
            bool done;
            GF2_TRY(all_done(pb >= 0 ? pb : pa, &done));
            if (done) {
                // the next pair's panels found the last pivots (or nothing): their pass is still due
                launch_pass(nxt, nxt.state_copy, 0, chunks, -1, rows_per_wg, s1);
                break;
            }
        }
    }
    GF2_HIP(hipGetLastError());
    hipLaunchKernelGGL(gather_rows_kernel<RrefState>, dim3((unsigned)gf2_cdiv(m, GATHER_ROWS), (unsigned)batch), dim3(256), 0, ctx->stream,
                       (const u64*)work, work == a_dev ? tmp : a_dev, (const int32_t*)pivrow, (const RrefState*)states, rank_dev, m, ld, cap);
    GF2_HIP(hipGetLastError());
    if (work == a_dev) GF2_HIP(hipMemcpyAsync(a_dev, tmp, (size_t)batch * m * ld * 8, hipMemcpyDeviceToDevice, ctx->stream));
    GF2_TRY(gf2_prof_end(ctx));
    return GF2_OK;
}

int gf2_rref_batch_dev(gf2_ctx* ctx, uint64_t* a_dev, int64_t batch, int64_t m, int64_t n, int64_t ld,
                       int64_t* pivots_dev, int64_t* rank_dev) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch_dev: null context");
    if (batch < 0 || m < 0 || n < 0 || ld < gf2_words(n)) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch_dev: bad shape");
    if (!rank_dev) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch_dev: null rank buffer");
    GF2_TRY(gf2_ctx_activate(ctx));
    if (batch == 0) return GF2_OK;
    if (m == 0 || n == 0) return gf2_dev_zero(ctx, rank_dev, (size_t)batch * 8);
    if (!a_dev) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch_dev: null matrix");
    const int64_t cap = m < n ? m : n;
    if (!gf2_flag(ctx, GF2_F_RREF_SEQUENTIAL) && !gf2_flag(ctx, GF2_F_RREF_NO_SMALL) && m <= 256 && ld <= 16 &&
        gf2_cdiv(m, 64) * ld <= 32) {
        // small matrices: one wavefront each, rows in registers (RPL rows of LD words per lane, at most 64 registers)
#define GF2_SMALL(RPL, LD) return launch_rref_small<RPL, LD>(ctx, (u64*)a_dev, batch, m, n, ld, pivots_dev, cap, rank_dev)
        const int rpl = (int)gf2_cdiv(m, 64);
        if (rpl <= 1) {
            if (ld <= 1) GF2_SMALL(1, 1);
            if (ld <= 2) GF2_SMALL(1, 2);
            if (ld <= 4) GF2_SMALL(1, 4);
            if (ld <= 8) GF2_SMALL(1, 8);
            GF2_SMALL(1, 16);
        } else if (rpl <= 2) {
            if (ld <= 4) GF2_SMALL(2, 4);
            if (ld <= 8) GF2_SMALL(2, 8);
            GF2_SMALL(2, 16);
        } else {
            if (ld <= 4) GF2_SMALL(4, 4);
            GF2_SMALL(4, 8);
        }
#undef GF2_SMALL
    }
    if (m < 0x7fffffffLL && batch <= 65535 && gf2_cdiv(m, 128) <= 65535 && !gf2_flag(ctx, GF2_F_RREF_SEQUENTIAL))
        return launch_rref_blocked(ctx, (u64*)a_dev, batch, m, n, ld, pivots_dev, cap, rank_dev);
    return launch_eliminate(ctx, ELIM_RREF, (u64*)a_dev, batch, m, n, ld, 0, pivots_dev, cap, rank_dev, nullptr, nullptr,
                            nullptr);
}

int gf2_rref_batch(gf2_ctx* ctx, uint64_t* a, int64_t batch, int64_t m, int64_t n, int64_t ld, int64_t* pivots_out,
                   int64_t* rank_out) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch: null context");
    if (batch < 0 || m < 0 || n < 0 || ld < gf2_words(n)) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch: bad shape");
    if (!rank_out) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch: null rank output");
    if (batch == 0) return GF2_OK;
    if (m == 0 || n == 0) {
        for (int64_t b = 0; b < batch; ++b) rank_out[b] = 0;
        return GF2_OK;
    }
    if (!a) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch: null matrix");
    const int64_t cap = m < n ? m : n;
    const size_t abytes = (size_t)batch * m * ld * 8;
    uint64_t* a_dev = nullptr;
    int64_t *piv_dev = nullptr, *rank_dev = nullptr;
    int rc = gf2_dev_alloc(ctx, abytes, (void**)&a_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)batch * cap * 8, (void**)&piv_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)batch * 8, (void**)&rank_dev);
    if (rc == GF2_OK) rc = gf2_h2d(ctx, a_dev, a, abytes);
    if (rc == GF2_OK) rc = gf2_dev_zero(ctx, piv_dev, (size_t)batch * cap * 8);
    if (rc == GF2_OK) rc = gf2_rref_batch_dev(ctx, a_dev, batch, m, n, ld, piv_dev, rank_dev);
    if (rc == GF2_OK) rc = gf2_d2h(ctx, a, a_dev, abytes);
    if (rc == GF2_OK && pivots_out) rc = gf2_d2h(ctx, pivots_out, piv_dev, (size_t)batch * cap * 8);
    if (rc == GF2_OK) rc = gf2_d2h(ctx, rank_out, rank_dev, (size_t)batch * 8);
    gf2_dev_free(ctx, a_dev);
    gf2_dev_free(ctx, piv_dev);
    gf2_dev_free(ctx, rank_dev);
    return rc;
}

int gf2_rref(gf2_ctx* ctx, uint64_t* a, int64_t m, int64_t n, int64_t ld, int64_t* pivots_out, int64_t* rank_out) {
    return gf2_rref_batch(ctx, a, 1, m, n, ld, pivots_out, rank_out);
}

int gf2_normalize_dev(gf2_ctx* ctx, uint64_t* h_dev, int64_t r, int64_t n, int64_t ld, int64_t offset,
                      int64_t* swaps_dev, int64_t* nswaps_dev, int* status_dev) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_normalize_dev: null context");
    if (r < 0 || n < 0 || offset < 0 || ld < gf2_words(n)) GF2_FAIL(GF2_E_ARG, "gf2_normalize_dev: bad shape");
    if (n < offset + r) GF2_FAIL(GF2_E_COLUMNS, "not enough columns");
    if (!nswaps_dev || !status_dev) GF2_FAIL(GF2_E_ARG, "gf2_normalize_dev: null output");
    GF2_TRY(gf2_ctx_activate(ctx));
    GF2_TRY(gf2_dev_zero(ctx, nswaps_dev, 8));
    GF2_TRY(gf2_dev_zero(ctx, status_dev, 4));
    if (r == 0) return GF2_OK;
    if (!h_dev) GF2_FAIL(GF2_E_ARG, "gf2_normalize_dev: null matrix");
    if (r > 8 * RB_THREADS || ld > ELIM_MAX_LD || gf2_flag(ctx, GF2_F_NORMALIZE_SEQUENTIAL))
        return launch_eliminate(ctx, ELIM_NORMALIZE, (u64*)h_dev, 1, r, n, ld, offset, nullptr, 0, nullptr, swaps_dev,
                                nswaps_dev, status_dev);
    // blocked: panel -> update -> (single sequential step if the panel stalled); the host looks at the state every 8 rounds
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t sbytes = al(sizeof(RrefState)), dbytes = al((size_t)r * 8), nbytes = al((size_t)64 * ld * 8);
    GF2_TRY(gf2_ws_reserve(ctx, 1, sbytes + dbytes + nbytes));
    char* q = (char*)ctx->ws[1];
    RrefState* st = (RrefState*)q; q += sbytes;
    u64* dco = (u64*)q; q += dbytes;
    u64* snap = (u64*)q;
    GF2_HIP(hipMemsetAsync(st, 0, sizeof(RrefState), ctx->stream));
    if (!ctx->lds_optin[4]) {
        GF2_HIP(hipFuncSetAttribute((const void*)rref_update_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ctx->lds_optin[4] = true;
    }
    const int rpt = (int)gf2_cdiv(r, RB_THREADS);
    const int64_t rows_per_wg = gf2_cdiv(ld, U2_CW) >= 16 ? 256 : 128;
    const dim3 ugrid((unsigned)gf2_cdiv(r, rows_per_wg), (unsigned)gf2_cdiv(ld, U2_CW), 1);
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_ELIM));
    // The host looks at the state when the work can be over at the earliest -- after ceil(remaining rows / 64) rounds -- and not
    // every 8 rounds (a Steane- or Reed-Muller-sized check is done after one round; seven more would be 21 empty launches).
    auto rounds = [&]() -> int {
        int64_t last_rank = 0, last_round = -1, next_check = gf2_cdiv(r, 64) - 1;
        for (int64_t round = 0;; ++round) {
#define GF2_NP_LAUNCH(RPT)                                                                                              \
    hipLaunchKernelGGL((norm_panel_kernel<RPT>), dim3(1), dim3(RB_THREADS), 0, ctx->stream, (u64*)h_dev, r, n, ld, offset, \
                       st, status_dev, dco, snap)
            if (rpt <= 1)
                GF2_NP_LAUNCH(1);
            else if (rpt <= 2)
                GF2_NP_LAUNCH(2);
            else if (rpt <= 4)
                GF2_NP_LAUNCH(4);
            else
                GF2_NP_LAUNCH(8);
#undef GF2_NP_LAUNCH
            // the RREF's trailing pass with this panel as the first of a pair that has no second (diagonal rows rebuilt from zero)
            hipLaunchKernelGGL(rref_update_pair_kernel, ugrid, dim3(RB_THREADS), 128 * 1024, ctx->stream, (u64*)h_dev, r, ld, rows_per_wg,
                               (const RrefState*)st, (const u64*)dco, (const u64*)dco, (const u64*)snap, (const u64*)snap, (const u64*)dco,
                               0, -1, (u64*)h_dev);
            hipLaunchKernelGGL(eliminate_kernel<ELIM_NORMALIZE>, dim3(1), dim3(ELIM_THREADS), 0, ctx->stream, (u64*)h_dev, r, n, ld,
                               offset, (int64_t*)nullptr, (int64_t)0, (int64_t*)nullptr, swaps_dev, nswaps_dev, status_dev, st);
            GF2_HIP(hipGetLastError());
            if (round < next_check) continue;
            RrefState host;
            int status = 0;
            GF2_HIP(hipMemcpyAsync(&host, st, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
            GF2_HIP(hipMemcpyAsync(&status, status_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
            GF2_HIP(hipStreamSynchronize(ctx->stream));
            if (status != 0 || (host.rank >= r && !host.stalled)) return GF2_OK;
            if (host.rank - last_rank < 8 * (round - last_round)) {
                // mostly stalls (a matrix that needs a column swap at nearly every step: fewer than 8 rows per round): the panels
                // do not pay; let the sequential kernel take every remaining step from here
                host.stalled = 2;
                GF2_HIP(hipMemcpyAsync(st, &host, sizeof(host), hipMemcpyHostToDevice, ctx->stream));
                GF2_HIP(hipStreamSynchronize(ctx->stream));
                hipLaunchKernelGGL(eliminate_kernel<ELIM_NORMALIZE>, dim3(1), dim3(ELIM_THREADS), 0, ctx->stream, (u64*)h_dev, r,
                                   n, ld, offset, (int64_t*)nullptr, (int64_t)0, (int64_t*)nullptr, swaps_dev, nswaps_dev,
                                   status_dev, st);
                GF2_HIP(hipGetLastError());
                return GF2_OK;
            }
            last_rank = host.rank;
            last_round = round;
            const int64_t more = gf2_cdiv(r - host.rank, 64);
            next_check = round + (more < 1 ? 1 : (more > 8 ? 8 : more));
        }
    };
    const int rc = rounds();
    const int rc_prof = gf2_prof_end(ctx);                            // the profile slot is closed on the error paths too
    return rc != GF2_OK ? rc : rc_prof;
}

int gf2_normalize(gf2_ctx* ctx, uint64_t* h, int64_t r, int64_t n, int64_t ld, int64_t offset, int64_t* swaps_out,
                  int64_t* nswaps_out) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_normalize: null context");
    if (r < 0 || n < 0 || offset < 0 || ld < gf2_words(n)) GF2_FAIL(GF2_E_ARG, "gf2_normalize: bad shape");
    if (!nswaps_out) GF2_FAIL(GF2_E_ARG, "gf2_normalize: null output");
    *nswaps_out = 0;
    if (n < offset + r) GF2_FAIL(GF2_E_COLUMNS, "not enough columns");
    if (r == 0) return GF2_OK;
    if (!h || !swaps_out) GF2_FAIL(GF2_E_ARG, "gf2_normalize: null buffer");
    const size_t hbytes = (size_t)r * ld * 8;
    uint64_t* h_dev = nullptr;
    int64_t *swaps_dev = nullptr, *nswaps_dev = nullptr;
    int* status_dev = nullptr;
    int status = 0;
    int rc = gf2_dev_alloc(ctx, hbytes, (void**)&h_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)2 * r * 8, (void**)&swaps_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, 8, (void**)&nswaps_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, 4, (void**)&status_dev);
    if (rc == GF2_OK) rc = gf2_h2d(ctx, h_dev, h, hbytes);
    if (rc == GF2_OK) rc = gf2_normalize_dev(ctx, h_dev, r, n, ld, offset, swaps_dev, nswaps_dev, status_dev);
    if (rc == GF2_OK) rc = gf2_d2h(ctx, &status, status_dev, 4);
    if (rc == GF2_OK) rc = gf2_d2h(ctx, nswaps_out, nswaps_dev, 8);
    if (rc == GF2_OK) rc = gf2_d2h(ctx, h, h_dev, hbytes);
    if (rc == GF2_OK && *nswaps_out > 0) rc = gf2_d2h(ctx, swaps_out, swaps_dev, (size_t)2 * (*nswaps_out) * 8);
    gf2_dev_free(ctx, h_dev);
    gf2_dev_free(ctx, swaps_dev);
    gf2_dev_free(ctx, nswaps_dev);
    gf2_dev_free(ctx, status_dev);
    if (rc == GF2_OK && status == ELIM_STATUS_DEPENDENT) GF2_FAIL(GF2_E_DEPENDENT, "rows are not independent");
    return rc;
}

int gf2_swap_columns(gf2_ctx* ctx, uint64_t* a, int64_t m, int64_t n, int64_t ld, int64_t i, int64_t j) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_swap_columns: null context");
    if (m < 0 || n < 0 || ld < gf2_words(n)) GF2_FAIL(GF2_E_ARG, "gf2_swap_columns: bad shape");
    if (i < 0 || j < 0 || i >= n || j >= n) GF2_FAIL(GF2_E_ARG, "gf2_swap_columns: column index out of range");
    if (m == 0 || i == j) return GF2_OK;
    if (!a) GF2_FAIL(GF2_E_ARG, "gf2_swap_columns: null matrix");
    const size_t bytes = (size_t)m * ld * 8;
    uint64_t* a_dev = nullptr;
    int rc = gf2_dev_alloc(ctx, bytes, (void**)&a_dev);
    if (rc == GF2_OK) rc = gf2_h2d(ctx, a_dev, a, bytes);
    if (rc == GF2_OK) {
        hipLaunchKernelGGL(swap_columns_kernel, dim3((unsigned)gf2_cdiv(m, 256)), dim3(256), 0, ctx->stream, (u64*)a_dev, m,
                           ld, i, j);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) {
            gf2_set_error("swap_columns_kernel failed: %s", hipGetErrorString(err));
            rc = GF2_E_HIP;
        }
    }
    if (rc == GF2_OK) rc = gf2_d2h(ctx, a, a_dev, bytes);
    gf2_dev_free(ctx, a_dev);
    return rc;
}

int gf2_row_weights(gf2_ctx* ctx, const uint64_t* a, int64_t m, int64_t n, int64_t ld, uint32_t* weights_out) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_row_weights: null context");
    if (m < 0 || n < 0 || ld < gf2_words(n)) GF2_FAIL(GF2_E_ARG, "gf2_row_weights: bad shape");
    if (m == 0) return GF2_OK;
    if (!weights_out) GF2_FAIL(GF2_E_ARG, "gf2_row_weights: null output");
    if (n == 0 || ld == 0) {
        for (int64_t i = 0; i < m; ++i) weights_out[i] = 0;
        return GF2_OK;
    }
    if (!a) GF2_FAIL(GF2_E_ARG, "gf2_row_weights: null matrix");
    const size_t bytes = (size_t)m * ld * 8;
    uint64_t* a_dev = nullptr;
    uint32_t* w_dev = nullptr;
    int rc = gf2_dev_alloc(ctx, bytes, (void**)&a_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)m * 4, (void**)&w_dev);
    if (rc == GF2_OK) rc = gf2_h2d(ctx, a_dev, a, bytes);
    if (rc == GF2_OK) {
        hipLaunchKernelGGL(row_weights_kernel, dim3((unsigned)gf2_cdiv(m, 4)), dim3(256), 0, ctx->stream, (const u64*)a_dev,
                           m, ld, w_dev);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) {
            gf2_set_error("row_weights_kernel failed: %s", hipGetErrorString(err));
            rc = GF2_E_HIP;
        }
    }
    if (rc == GF2_OK) rc = gf2_d2h(ctx, weights_out, w_dev, (size_t)m * 4);
    gf2_dev_free(ctx, a_dev);
    gf2_dev_free(ctx, w_dev);
    return rc;
}

int gf2_nullspace(gf2_ctx* ctx, const uint64_t* a, int64_t m, int64_t n, int64_t ld, uint64_t* n_out, int64_t ldn,
                  int64_t* rows_out) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_nullspace: null context");
    if (m < 0 || n < 0 || ld < gf2_words(n) || ldn < gf2_words(n)) GF2_FAIL(GF2_E_ARG, "gf2_nullspace: bad shape");
    if (!rows_out) GF2_FAIL(GF2_E_ARG, "gf2_nullspace: null output");
    *rows_out = 0;
    if (n == 0) return GF2_OK;
    if (!n_out || (!a && m > 0)) GF2_FAIL(GF2_E_ARG, "gf2_nullspace: null buffer");
    if (n >= 0x7fffffffLL) GF2_FAIL(GF2_E_ARG, "gf2_nullspace: too many columns");
    const int64_t cap = m < n ? m : n;
    const size_t abytes = (size_t)(m > 0 ? m : 1) * ld * 8;
    uint64_t *a_dev = nullptr, *out_dev = nullptr;
    int64_t *piv_dev = nullptr, *rank_dev = nullptr;
    int32_t *free_dev = nullptr, *map_dev = nullptr;
    int64_t rank = 0;
    std::vector<int64_t> pivots((size_t)(cap > 0 ? cap : 1));
    std::vector<int32_t> col_pivot_row((size_t)n, -1), free_cols;
    int rc = gf2_dev_alloc(ctx, abytes, (void**)&a_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)(cap > 0 ? cap : 1) * 8, (void**)&piv_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, 8, (void**)&rank_dev);
    if (rc == GF2_OK && m > 0) rc = gf2_h2d(ctx, a_dev, a, (size_t)m * ld * 8);
    if (rc == GF2_OK) rc = gf2_rref_batch_dev(ctx, a_dev, 1, m, n, ld, piv_dev, rank_dev);
    if (rc == GF2_OK) rc = gf2_d2h(ctx, &rank, rank_dev, 8);
    if (rc == GF2_OK && rank > 0) rc = gf2_d2h(ctx, pivots.data(), piv_dev, (size_t)rank * 8);
    if (rc == GF2_OK) {
        for (int64_t i = 0; i < rank; ++i) col_pivot_row[(size_t)pivots[(size_t)i]] = (int32_t)i;
        for (int64_t c = 0; c < n; ++c)
            if (col_pivot_row[(size_t)c] < 0) free_cols.push_back((int32_t)c);
    }
    const int64_t nfree = (int64_t)free_cols.size();
    if (rc == GF2_OK && nfree > 0) {
        rc = gf2_dev_alloc(ctx, (size_t)nfree * ldn * 8, (void**)&out_dev);
        if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)nfree * 4, (void**)&free_dev);
        if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)n * 4, (void**)&map_dev);
        if (rc == GF2_OK) rc = gf2_h2d(ctx, free_dev, free_cols.data(), (size_t)nfree * 4);
        if (rc == GF2_OK) rc = gf2_h2d(ctx, map_dev, col_pivot_row.data(), (size_t)n * 4);
        if (rc == GF2_OK) {
            dim3 grid((unsigned)gf2_cdiv(ldn, 64), (unsigned)nfree);
            hipLaunchKernelGGL(nullspace_kernel, grid, dim3(64), 0, ctx->stream, (const u64*)a_dev, n, ld, free_dev, nfree,
                               map_dev, (u64*)out_dev, ldn);
            hipError_t err = hipGetLastError();
            if (err != hipSuccess) {
                gf2_set_error("nullspace_kernel failed: %s", hipGetErrorString(err));
                rc = GF2_E_HIP;
            }
        }
        if (rc == GF2_OK) rc = gf2_d2h(ctx, n_out, out_dev, (size_t)nfree * ldn * 8);
    }
    if (rc == GF2_OK) *rows_out = nfree;
    gf2_dev_free(ctx, a_dev);
    gf2_dev_free(ctx, piv_dev);
    gf2_dev_free(ctx, rank_dev);
    gf2_dev_free(ctx, out_dev);
    gf2_dev_free(ctx, free_dev);
    gf2_dev_free(ctx, map_dev);
    return rc;
}

}  // extern "C"
