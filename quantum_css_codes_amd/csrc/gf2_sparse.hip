// Sparse-error syndrome extraction (gfx950): s = XOR of the parity-check columns selected by the set bits of e.
//
// Same product as np.mod(np.matmul(parity_check, e), 2) (css_code.py:728), organised for the regime the
// Monte-Carlo workload lives in: at depolarising p = 0.01 an error of n = 4096 qubits has about 27 set bits per
// Pauli component, so 97 % of the dense kernel's table lookups fetch the zero entry.  Here one wavefront owns one
// sample at a time:
//
//   1. the 64 lanes load the sample's packed error row (sample-major, 512 B contiguous at n = 4096);
//   2. the columns of an identity block H[:, off:off+r] = I (standard forms, css_code.py:51-61) are taken straight
//      from the error bits; their positions are then masked out;
//   3. the remaining set bits are compacted into a per-wave LDS list, layer by layer (__ballot + mbcnt);
//   4. for every listed column j the wavefront loads column j of H, stored transposed (256 B per 2048 rows,
//      lane d holds rows 32d..32d+31), eight columns in flight at a time, and XORs it into its accumulator;
//   5. the syndrome row is stored (256 B contiguous, sample-major) and/or its weight goes to the histogram.
//
// Work is proportional to the error weight; the transposed check (1 MiB at 2048 x 4096, half of it never touched
// in standard form) stays in L2.  The kernel is bound by L2 gather bandwidth: about 14 random 256-byte column
// reads per sample and component at the benchmark's p = 0.01 (measured ~50-70 GB/s per CU, the chip's ceiling for
// L2-resident row gathers).  A 16-lanes-per-sample variant (4 samples per wavefront, 1 KiB per wavefront load,
// 2.3 instead of 5 instructions per column) measured the same time, which is how the bound was identified.
// The dense Four-Russians kernel (gf2_syndrome.hip) remains the data-independent path; DESIGN.md gives the crossover.
// Histogram-only calls on large batches go to the LDS-slab pipeline instead (gf2_slabs.hip), which uses this file's
// wavefront-per-sample routine only for the rare samples that do not fit a record.
#include "gf2_internal.h"
#include "gf2_sampler.h"
#include "gf2_sparse_dev.h"

#define SPARSE_WAVES 8

// grid (slabs, words of n), block 64.  Lane i holds row 64*slab+i; column 64*word+b of the slab is the ballot of
// bit b.  cols[(64*word + b) * ldc64 + slab] (u64 = 64 rows).  Identity-block columns are left zero.
__global__ void build_columns_kernel(const uint64_t* __restrict__ h, int64_t r, int64_t n, int64_t ld, int64_t ident_off,
                                     int64_t ldc64, u64* __restrict__ cols) {
    const int lane = threadIdx.x;
    const int64_t slab = blockIdx.x, word = blockIdx.y;
    const int64_t row = slab * 64 + lane;
    u64 w = (row < r && word < ld) ? h[row * ld + word] : 0ull;
    if (ident_off >= 0) {
        const int64_t lo = ident_off - word * 64, hi = ident_off + r - word * 64;
        if (hi > 0 && lo < 64) {
            u64 m = ~0ull;
            if (lo > 0) m &= ~0ull << lo;
            if (hi < 64) m &= ~(~0ull << hi);
            w &= ~m;
        }
    }
    u64 mine = 0;
    for (int b = 0; b < 64; ++b) {
        const u64 m = __ballot((w >> b) & 1ull);
        if (lane == b) mine = m;
    }
    const int64_t col = word * 64 + lane;
    if (col < n) cols[col * ldc64 + slab] = mine;
}

// K = dwords of the column per lane (rows 2048*k + 32*lane ..).  ht: (n + 1) columns of 64*K dwords, column n is zero.
template <int K, bool WRITE_S, bool HIST>
__global__ __launch_bounds__(64 * SPARSE_WAVES) void syndrome_sparse_kernel(
    const uint32_t* __restrict__ ht, int col_shift, int64_t n, int64_t r, int64_t ident_off,
    const uint64_t* __restrict__ e, int64_t batch, int64_t lde, uint32_t* __restrict__ s_out, int64_t lds32,
    u64* __restrict__ hist, int nbins) {
    __shared__ unsigned int list[SPARSE_WAVES][SPARSE_LIST_CAP + 8];
    __shared__ unsigned int bins[HIST ? 4096 : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned int* mylist = list[wave];
    const bool priv = HIST && nbins <= 4096;
    if (priv) {
        for (int i = threadIdx.x; i < nbins; i += blockDim.x) bins[i] = 0;
        __syncthreads();
    }
    const int64_t words = (n + 63) >> 6;
    const unsigned int lane4 = lane * 4u, lane8 = lane * 8u;
    const unsigned int zero_col = (unsigned int)n;
    const int64_t total_waves = (int64_t)gridDim.x * SPARSE_WAVES;
    const char* htb = reinterpret_cast<const char*>(ht);
    const char* eb = reinterpret_cast<const char*>(e);
    const int64_t row_bytes = lde * 8;

    // per-lane constants of the identity block: dword d = lane + 64k covers rows 32d..32d+31 <-> error bits
    // ident_off + 32d ..; id_off = byte offset of the first error dword, id_sh = shift, id_keep = valid-row mask
    unsigned int id_off[K], id_keep[K];
    bool id_two[K];
    const int id_sh = ident_off >= 0 ? (int)(ident_off & 31) : 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int64_t row0 = 32 * (int64_t)(lane + 64 * k);
        id_off[k] = 0;
        id_keep[k] = 0;
        id_two[k] = false;
        if (ident_off >= 0 && row0 < r) {
            const int64_t dw = (ident_off + row0) >> 5;
            id_off[k] = (unsigned int)(dw * 4);
            id_keep[k] = r - row0 < 32 ? ~(~0u << (r - row0)) : ~0u;
            id_two[k] = id_sh != 0 && dw + 1 < lde * 2;
        }
    }
    const u64 idmask0 = ident_mask(ident_off, r, lane);             // identity columns inside this lane's first word

    int64_t sample = (int64_t)blockIdx.x * SPARSE_WAVES + wave;
    u64 w_next = 0;
    if (sample < batch && lane < words) w_next = *reinterpret_cast<const u64*>(eb + sample * row_bytes + lane8);
    for (; sample < batch; sample += total_waves) {
        const char* rowp = eb + sample * row_bytes;                 // uniform: stays in SGPRs
        u64 w = w_next;
        const int64_t nxt = sample + total_waves;
        if (nxt < batch && lane < words) w_next = *reinterpret_cast<const u64*>(eb + nxt * row_bytes + lane8);

        unsigned int acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            unsigned int v = 0;
            if (id_keep[k]) {
                v = *reinterpret_cast<const uint32_t*>(rowp + id_off[k]) >> id_sh;
                if (id_two[k]) v |= *reinterpret_cast<const uint32_t*>(rowp + id_off[k] + 4) << (32 - id_sh);
                v &= id_keep[k];
            }
            acc[k] = v;
        }
        for (int64_t wb = 0; wb < words; wb += 64) {
            if (wb) {
                const int64_t wi = wb + lane;
                w = wi < words ? *reinterpret_cast<const u64*>(rowp + wi * 8) : 0ull;
                w &= ~ident_mask(ident_off, r, wi);
            } else {
                w &= ~idmask0;                                      // identity columns are already accounted for
            }
            // compact the set-bit positions into the LDS list, one layer (t-th set bit of every lane) at a time
            u64 x = w;
            unsigned int total = 0;
            for (;;) {
                const u64 active = __ballot(x != 0);
                if (!active) break;
                if (x) {
                    const int b = __ffsll((long long)x) - 1;
                    x &= x - 1;
                    const unsigned int pos = total + __builtin_amdgcn_mbcnt_hi((unsigned int)(active >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((unsigned int)active, 0u));
                    if (pos < SPARSE_LIST_CAP) mylist[pos] = (unsigned int)(((wb + lane) << 6) + b);
                }
                total += (unsigned int)__popcll(active);
            }
            if (total == 0) continue;
            if (total <= SPARSE_LIST_CAP) {
                if (lane < 8) mylist[total + lane] = zero_col;        // pad to a multiple of 8 with the zero column
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (unsigned int k0 = 0; k0 < total; k0 += 8) {
                    unsigned int v[8][K];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const unsigned int off = (mylist[k0 + i] << col_shift) | lane4;
#pragma unroll
                        for (int k = 0; k < K; ++k) v[i][k] = *reinterpret_cast<const uint32_t*>(htb + off + 256u * k);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int k = 0; k < K; ++k) acc[k] ^= v[i][k];
                }
                __builtin_amdgcn_wave_barrier();
            } else {
                // dense sample: walk the lanes' words one by one (uniform loops)
                u64 nz = __ballot(w != 0);
                while (nz) {
                    const int src = __ffsll((long long)nz) - 1;
                    nz &= nz - 1;
                    u64 word = ((u64)(unsigned int)__builtin_amdgcn_readlane((int)(w >> 32), src) << 32) |
                               (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)w, src);
                    while (word) {
                        const int b = __ffsll((long long)word) - 1;
                        word &= word - 1;
                        const unsigned int off = ((unsigned int)(((wb + src) << 6) + b) << col_shift) | lane4;
#pragma unroll
                        for (int k = 0; k < K; ++k) acc[k] ^= *reinterpret_cast<const uint32_t*>(htb + off + 256u * k);
                    }
                }
            }
        }
        if (WRITE_S) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int64_t d = lane + 64 * k;
                if (d < lds32) s_out[sample * lds32 + d] = acc[k];
            }
        }
        if (HIST) {
            unsigned int wt = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) wt += __popc(acc[k]);
            wt = wave_total(wt);
            if (lane == 0) {
                if (priv)
                    atomicAdd(&bins[wt], 1u);
                else
                    atomicAdd(&hist[wt], 1ull);
            }
        }
    }
    if (priv) {
        __syncthreads();
        for (int i = threadIdx.x; i < nbins; i += blockDim.x)
            if (bins[i]) atomicAdd(&hist[i], (u64)bins[i]);
    }
}

// ---- fused Monte-Carlo: sampler + both syndromes + both weight histograms, no error words in memory -------------------
//
// n <= 4096 and r_1, r_2 <= 2048: a wavefront owns a sample, lane = 64-qubit word = syndrome dword.  The first lanes run
// the sample's (at most 8) segments into an LDS image, every lane picks up its words (e_x, e_z) from there, then the sparse
// column accumulation runs once per component
// (e_z against H1, e_x against H2, css_code.py:457-470).  The identity-block bits, which the stand-alone kernel
// re-reads from memory, come from the neighbouring lanes' registers here.
__global__ __launch_bounds__(64 * SPARSE_WAVES) void mc_sparse_fused_kernel(SparseSide side_z, SparseSide side_x, int64_t n,
                                                                           u64 seed, int64_t first_sample, int64_t count,
                                                                           SegTables th) {
    __shared__ unsigned int list[SPARSE_WAVES][SPARSE_LIST_CAP + 8];
    __shared__ unsigned int bins_z[2052], bins_x[2052];
    __shared__ u64 cdf_lds[2 * GF2_SEG_CDF];
    __shared__ unsigned int img[SPARSE_WAVES][8 * 33];                   // per segment 32 dwords (x: 0..15, z: 16..31) + 1 of padding
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 2052; i += blockDim.x) {
        bins_z[i] = 0;
        bins_x[i] = 0;
    }
    stage_seg_cdf(th, cdf_lds);
    const int words = (int)((n + 63) >> 6);
    const bool last = lane == th.nseg - 1;
    const int64_t total_waves = (int64_t)gridDim.x * SPARSE_WAVES;
    for (int64_t i = (int64_t)blockIdx.x * SPARSE_WAVES + wave; i < count; i += total_waves) {
        u64 ex = 0, ez = 0;
        if (lane < th.nseg)
            sample_segment(sample_key(seed, (u64)(first_sample + i)), lane, last ? th.nb_last : GF2_SEG_BITS,
                           cdf_lds + (last ? GF2_SEG_CDF : 0), th.t_1, th.t_2, img[wave] + lane * 33);
        __builtin_amdgcn_wave_barrier();
        if (lane < words) {
            const unsigned int* seg = img[wave] + (lane >> 3) * 33 + (lane & 7) * 2;
            ex = ((u64)seg[1] << 32) | seg[0];
            ez = ((u64)seg[17] << 32) | seg[16];
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned int wz = sparse_component_weight(ez, side_z, n, lane, list[wave]);
        const unsigned int wx = sparse_component_weight(ex, side_x, n, lane, list[wave]);
        if (lane == 0) {
            atomicAdd(&bins_z[wz], 1u);
            atomicAdd(&bins_x[wx], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < side_z.nbins; i += blockDim.x)
        if (bins_z[i]) atomicAdd(&side_z.hist[i], (u64)bins_z[i]);
    for (int i = threadIdx.x; i < side_x.nbins; i += blockDim.x)
        if (bins_x[i]) atomicAdd(&side_x.hist[i], (u64)bins_x[i]);
}

// ---- host side ---------------------------------------------------------------------------------------------------

int gf2_build_columns(gf2_ctx* ctx, gf2_check* ck) {
    ck->ht_dev = nullptr;
    ck->ht_k = 0;
    if (ck->small || ck->r == 0 || ck->n == 0) return GF2_OK;
    const int64_t k = gf2_cdiv(ck->r, 2048);
    if (k > 4 || ck->n >= (1 << 24)) return GF2_OK;                // not supported: callers fall back to the dense kernel
    const int64_t kk = k == 3 ? 4 : k;                              // instantiated for 1, 2, 4
    const int64_t ldc32 = 64 * kk, ldc64 = ldc32 / 2;
    const size_t bytes = (size_t)(ck->n + 1) * ldc32 * 4;
    if (bytes > (1ull << 32)) return GF2_OK;
    GF2_TRY(gf2_dev_alloc(ctx, bytes, (void**)&ck->ht_dev));
    GF2_TRY(gf2_dev_zero(ctx, ck->ht_dev, bytes));
    dim3 grid((unsigned)ck->slabs, (unsigned)gf2_words(ck->n));
    hipLaunchKernelGGL(build_columns_kernel, grid, dim3(64), 0, ctx->stream, ck->h_dev, ck->r, ck->n, ck->ld, ck->ident_off,
                       ldc64, (u64*)ck->ht_dev);
    GF2_HIP(hipGetLastError());
    GF2_HIP(hipStreamSynchronize(ctx->stream));
    ck->ht_k = (int)kk;
    return GF2_OK;
}

template <bool WRITE_S, bool HIST>
static void launch_sparse(gf2_ctx* ctx, const gf2_check* ck, const uint64_t* e_dev, int64_t batch, int64_t lde,
                          uint32_t* s32, int64_t lds32, u64* hist, int nbins) {
    int64_t blocks = gf2_cdiv(batch, SPARSE_WAVES * 4);
    const int64_t cap = (int64_t)ctx->num_cus * 4;
    if (blocks > cap) blocks = cap;
    const int col_shift = ck->ht_k == 1 ? 8 : (ck->ht_k == 2 ? 9 : 10);   // log2 of the bytes per column
    dim3 grid((unsigned)blocks), block(64 * SPARSE_WAVES);
#define GF2_SPARSE_LAUNCH(KK)                                                                                          \
    hipLaunchKernelGGL((syndrome_sparse_kernel<KK, WRITE_S, HIST>), grid, block, 0, ctx->stream,                        \
                       (const uint32_t*)ck->ht_dev, col_shift, ck->n, ck->r, ck->ident_off, e_dev, batch, lde, s32, lds32, \
                       hist, nbins)
    if (ck->ht_k == 1)
        GF2_SPARSE_LAUNCH(1);
    else if (ck->ht_k == 2)
        GF2_SPARSE_LAUNCH(2);
    else
        GF2_SPARSE_LAUNCH(4);
#undef GF2_SPARSE_LAUNCH
}

// Fused sparse Monte-Carlo (internal; gf2_mc_run uses it).  hist buffers must be zeroed by the caller.
int gf2_mc_sparse_fused(gf2_ctx* ctx, const gf2_check* c1, const gf2_check* c2, uint64_t seed, int64_t first_sample,
                        int64_t count, double p_x, double p_y, double p_z, uint64_t* hz_dev, uint64_t* hx_dev) {
    SegTables th;
    GF2_TRY(gf2_seg_tables(ctx, p_x, p_y, p_z, c1->n, &th));
    if (count == 0) return GF2_OK;
    SparseSide sz = {c1->ht_dev, c1->r, c1->ident_off, (u64*)hz_dev, (int)(c1->r + 1)};
    SparseSide sx = {c2->ht_dev, c2->r, c2->ident_off, (u64*)hx_dev, (int)(c2->r + 1)};
    int64_t blocks = gf2_cdiv(count, SPARSE_WAVES * 8);
    const int64_t cap = (int64_t)ctx->num_cus * 4;
    if (blocks > cap) blocks = cap;
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_SYNDROME));
    hipLaunchKernelGGL(mc_sparse_fused_kernel, dim3((unsigned)blocks), dim3(64 * SPARSE_WAVES), 0, ctx->stream, sz, sx, c1->n,
                       (u64)seed, first_sample, count, th);
    GF2_TRY(gf2_prof_end(ctx));
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

bool gf2_mc_sparse_fused_ok(const gf2_check* c1, const gf2_check* c2) {
    return c1->ht_dev && c2->ht_dev && c1->ht_k == 1 && c2->ht_k == 1 && c1->n <= 4096 && c1->n == c2->n;
}

extern "C" {

int gf2_syndrome_sparse_dev(gf2_ctx* ctx, const gf2_check* ck, const uint64_t* e_dev, int64_t batch, int64_t lde,
                            uint64_t* s_dev, int64_t lds, uint64_t* hist_dev, int64_t nbins) {
    if (!ctx || !ck) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_sparse_dev: null argument");
    if (batch < 0) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_sparse_dev: negative batch");
    if (!ck->ht_dev) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_sparse_dev: this check has no transposed columns (small check or r > 8192)");
    if (!s_dev && !hist_dev) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_sparse_dev: nothing to produce");
    if (s_dev && lds < ck->slabs) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_sparse_dev: lds too small");
    if (hist_dev && nbins != ck->r + 1) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_sparse_dev: weight histogram needs r+1 bins");
    if (lde < gf2_words(ck->n) || lde < 1) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_sparse_dev: lde too small");
    if (batch == 0) return GF2_OK;
    if (!e_dev) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_sparse_dev: null errors");
    GF2_TRY(gf2_ctx_activate(ctx));
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_SYNDROME));
    // histogram only: the LDS-slab pipeline (gf2_slabs.hip) once the batch amortises its three launches and the table
    // loads; GF2_SPARSE_SLABS / GF2_SPARSE_GATHER force one or the other (same results)
    if (gf2_lane_ok(ck) && !gf2_flag(ctx, GF2_F_SPARSE_GATHER) && !gf2_flag(ctx, GF2_F_SPARSE_SLABS))
        GF2_TRY(gf2_syndrome_lane(ctx, ck, e_dev, batch, lde, s_dev, lds, hist_dev, ctx->stream));   // n <= 512, r <= 256
    else if (gf2_slabs_ok(ck) && !gf2_flag(ctx, GF2_F_SPARSE_GATHER) && (batch >= 32768 || gf2_flag(ctx, GF2_F_SPARSE_SLABS)))
        // (round 4: with the syndromes stored too -- every gather workgroup its slab's 64-byte piece -- instead of the
        // column-gather kernel for any call that wanted them)
        GF2_TRY(gf2_syndrome_slabs(ctx, ck, e_dev, batch, lde, hist_dev, ctx->stream, 2, s_dev, lds));
    else if (s_dev && hist_dev)
        launch_sparse<true, true>(ctx, ck, e_dev, batch, lde, (uint32_t*)s_dev, lds * 2, (u64*)hist_dev, (int)nbins);
    else if (s_dev)
        launch_sparse<true, false>(ctx, ck, e_dev, batch, lde, (uint32_t*)s_dev, lds * 2, nullptr, 0);
    else
        launch_sparse<false, true>(ctx, ck, e_dev, batch, lde, nullptr, 0, (u64*)hist_dev, (int)nbins);
    GF2_TRY(gf2_prof_end(ctx));
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

}  // extern "C"
