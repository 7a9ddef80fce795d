// Sparse-error syndromes for mid-size checks (n <= 512, r <= 256): one lane per sample (gfx950).
//
// Same product as np.mod(np.matmul(parity_check, e), 2) (css_code.py:728).  The machinery of the n = 4096 path (records,
// row slabs, a wavefront or four lanes per sample) costs a fixed amount per sample that dwarfs the work when an error row
// is 16 to 64 bytes.  Here the whole transposed check sits in LDS (one entry of ceil(r/64) words per column, 16 KiB at
// 512 x 256), a lane loads its sample's packed error row (consecutive lanes read consecutive rows), walks its set bits
// and XORs the listed columns into a syndrome held in registers; the weight goes to an LDS-privatised histogram and / or
// the syndrome row is stored.  A wavefront runs as long as its heaviest sample, which at sparse rates is a handful of
// columns; the kernel is then bound by reading the errors.
#include "gf2_internal.h"

#define LANE_MAX_N 512
#define LANE_MAX_R 256
#define LANE_THREADS 256

// tab[col * SW + k]: bit i = H[64 k + i][col]
__global__ void build_lane_table_kernel(const u64* __restrict__ h, int r, int n, int64_t ld, int sw, u64* __restrict__ tab) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * sw) return;
    const int col = idx / sw, k = idx % sw;
    u64 word = 0;
    for (int i = 0; i < 64; ++i) {
        const int row = 64 * k + i;
        if (row < r) word |= ((h[(int64_t)row * ld + (col >> 6)] >> (col & 63)) & 1ull) << i;
    }
    tab[idx] = word;
}

template <int EW, int SW>
__global__ __launch_bounds__(LANE_THREADS) void syndrome_lane_kernel(const u64* __restrict__ tab, int n, int r,
                                                                     const u64* __restrict__ e, int64_t batch, int64_t lde,
                                                                     u64* __restrict__ s_out, int64_t lds, u64* __restrict__ hist) {
    __shared__ u64 T[LANE_MAX_N * SW];
    __shared__ unsigned int bins[LANE_MAX_R + 1];
    for (int i = threadIdx.x; i < n * SW; i += LANE_THREADS) T[i] = tab[i];
    if (hist)
        for (int i = threadIdx.x; i <= r; i += LANE_THREADS) bins[i] = 0;
    __syncthreads();
    const int words = (n + 63) >> 6;
    const int64_t stride = (int64_t)gridDim.x * LANE_THREADS;
    for (int64_t sample = (int64_t)blockIdx.x * LANE_THREADS + threadIdx.x; sample < batch; sample += stride) {
        const u64* row = e + sample * lde;
        u64 x[EW];
#pragma unroll
        for (int w = 0; w < EW; ++w) x[w] = w < words ? row[w] : 0ull;
        u64 s[SW];
#pragma unroll
        for (int k = 0; k < SW; ++k) s[k] = 0;
#pragma unroll
        for (int w = 0; w < EW; ++w) {
            u64 v = x[w];
            while (v) {
                const int col = w * 64 + (__ffsll((long long)v) - 1);
                v &= v - 1;
#pragma unroll
                for (int k = 0; k < SW; ++k) s[k] ^= T[col * SW + k];
            }
        }
        if (s_out) {
#pragma unroll
            for (int k = 0; k < SW; ++k)
                if (k < lds) s_out[sample * lds + k] = s[k];
        }
        if (hist) {
            unsigned int wt = 0;
#pragma unroll
            for (int k = 0; k < SW; ++k) wt += (unsigned int)__popcll(s[k]);
            atomicAdd(&bins[wt], 1u);
        }
    }
    if (hist) {
        __syncthreads();
        for (int i = threadIdx.x; i <= r; i += LANE_THREADS)
            if (bins[i]) atomicAdd(&hist[i], (u64)bins[i]);
    }
}

// words per table entry: the kernel is instantiated for 1, 2 and 4 (rows past r are zero)
static int lane_sw(int words) { return words <= 1 ? 1 : (words <= 2 ? 2 : 4); }

int gf2_build_lane_table(gf2_ctx* ctx, gf2_check* ck) {
    ck->lane_tab_dev = nullptr;
    if (ck->small || ck->r == 0 || ck->n == 0 || ck->n > LANE_MAX_N || ck->r > LANE_MAX_R) return GF2_OK;
    const int sw = lane_sw((int)gf2_cdiv(ck->r, 64));
    GF2_TRY(gf2_dev_alloc(ctx, (size_t)ck->n * sw * 8, (void**)&ck->lane_tab_dev));
    hipLaunchKernelGGL(build_lane_table_kernel, dim3((unsigned)gf2_cdiv(ck->n * sw, 256)), dim3(256), 0, ctx->stream,
                       (const u64*)ck->h_dev, (int)ck->r, (int)ck->n, ck->ld, sw, (u64*)ck->lane_tab_dev);
    GF2_HIP(hipGetLastError());
    GF2_HIP(hipStreamSynchronize(ctx->stream));
    return GF2_OK;
}

bool gf2_lane_ok(const gf2_check* ck) { return ck->lane_tab_dev != nullptr; }

// s_dev (batch x lds words) and / or hist_dev (r + 1 bins, accumulated into) may be null.
int gf2_syndrome_lane(gf2_ctx* ctx, const gf2_check* ck, const uint64_t* e_dev, int64_t batch, int64_t lde, uint64_t* s_dev,
                      int64_t lds, uint64_t* hist_dev, hipStream_t stream) {
    const int words = (int)gf2_words(ck->n), sw = lane_sw((int)gf2_cdiv(ck->r, 64));
    int64_t blocks = gf2_cdiv(batch, LANE_THREADS * 4);
    if (blocks > (int64_t)ctx->num_cus * 8) blocks = (int64_t)ctx->num_cus * 8;
    if (blocks < 1) blocks = 1;
#define GF2_LANE(EW, SW)                                                                                              \
    hipLaunchKernelGGL((syndrome_lane_kernel<EW, SW>), dim3((unsigned)blocks), dim3(LANE_THREADS), 0, stream,           \
                       (const u64*)ck->lane_tab_dev, (int)ck->n, (int)ck->r, (const u64*)e_dev, batch, lde, (u64*)s_dev, lds, \
                       (u64*)hist_dev)
#define GF2_LANE_EW(SW)       \
    if (words <= 2)           \
        GF2_LANE(2, SW);      \
    else if (words <= 4)      \
        GF2_LANE(4, SW);      \
    else                      \
        GF2_LANE(8, SW)
    if (sw <= 1) {
        GF2_LANE_EW(1);
    } else if (sw <= 2) {
        GF2_LANE_EW(2);
    } else {
        GF2_LANE_EW(4);
    }
#undef GF2_LANE_EW
#undef GF2_LANE
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}
