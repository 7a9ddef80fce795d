// Syndrome extraction S = E . H^T over GF(2) on packed words, and syndrome histograms (gfx950).
//
// Replaces np.mod(np.matmul(parity_check, e), 2) of css_code.py:728 (and the commutation check of
// css_code.py:47) for a whole batch of errors.
//
//   * syndrome_tiled_kernel    Method-of-Four-Russians tables in LDS.  A workgroup owns one slab of 64
//                              parity-check rows and a chunk of samples.  For every group of 4 columns the
//                              slab's table holds the 16 XOR-combinations of its four column words
//                              (16 x 8 B); a lane owns a sample and each nibble of its error word selects
//                              one entry with a ds_read_b64.  The 16 entries of a group span 32 distinct
//                              banks and equal addresses broadcast, so the reads are conflict-free.
//                              Errors arrive in the tiled layout (gf2hip.h): a wavefront reads one 16-byte
//                              piece per lane from 1 KiB of contiguous memory.
//                              Only "active" 128-column pairs of a slab are tabulated: pairs whose columns are
//                              all zero in the slab are skipped, and the columns of an identity block
//                              H[:, off:off+r] = I (the reference's standard forms, css_code.py:51-61) are
//                              taken straight from the error word instead of through the table.
//                              Tables are staged at LDS offset 0 in chunks of at most 16 pairs (64 KiB), so a
//                              lookup address is nibble*8 plus an instruction immediate; two workgroups share a CU.
//   * retile_kernel            sample-major -> tiled, through LDS (coalesced both ways).
//   * syndrome_small_kernel    n <= 64, r <= 64, sample-major: one lane per sample, rows in SGPRs,
//                              parity by AND + popcount.  Pure streaming.
//   * syndrome_sliced_kernel   n <= 64, r <= 64, bit-sliced: one lane per 64 samples, S[i] = XOR of the
//                              E[q] selected by row i.  Pure streaming at n + r bits per sample.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "gf2_internal.h"

// ---- table construction ---------------------------------------------------------------------------------

// grid (slabs, max_pairs), block 64.  Lane i holds row 64*slab+i.  Entry layout per (slab, t):
//   tables[((slab * max_pairs + t) * 32 + j) * 16 + v],  j = group within the pair, v = nibble value.
__global__ void build_tables_kernel(const uint64_t* __restrict__ h, int64_t r, int64_t ld, int64_t ident_off,
                                    const int32_t* __restrict__ pair_list, const int32_t* __restrict__ npairs,
                                    int64_t max_pairs, u64* __restrict__ tables) {
    const int lane = threadIdx.x;
    const int64_t slab = blockIdx.x, t = blockIdx.y;
    const int64_t row = slab * 64 + lane;
    u64* out = tables + (slab * max_pairs + t) * 512;
    if (t >= npairs[slab]) {
        for (int i = lane; i < 512; i += 64) out[i] = 0;
        return;
    }
    const int64_t q = pair_list[slab * max_pairs + t];
    for (int half = 0; half < 2; ++half) {
        const int64_t word = 2 * q + half;
        u64 w = 0;
        if (row < r && word < ld) w = h[row * ld + word];
        if (ident_off >= 0) {                                   // drop the identity block's columns
            const int64_t lo = ident_off - word * 64, hi = ident_off + r - word * 64;
            if (hi > 0 && lo < 64) {
                u64 m = ~0ull;
                if (lo > 0) m &= ~0ull << lo;
                if (hi < 64) m &= ~(~0ull << hi);
                w &= ~m;
            }
        }
        for (int g = 0; g < 16; ++g) {
            u64 col[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) col[c] = __ballot((w >> (4 * g + c)) & 1ull);
            if (lane < 16) {
                u64 acc = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if ((lane >> c) & 1) acc ^= col[c];
                out[(half * 16 + g) * 16 + lane] = acc;
            }
        }
    }
}

// ---- Four-Russians syndrome kernel ------------------------------------------------------------------------

#define SYN_THREADS 768                             // 12 wavefronts; two workgroups per CU -> 6 waves/SIMD, 80 VGPRs
#define SYN_WAVES (SYN_THREADS / 64)
#define SYN_TPW_MIN 5                               // tiles (of 64 samples) per wave: at least 60 tiles = 3840 samples per workgroup
#define SYN_CHUNK_PAIRS 19                          // 19 pairs x 4 KiB = 76 KiB of LDS per staging, 2 workgroups/CU
#define SYN_UNROLL_PAIRS 16                         // pairs whose table offsets fit the 16-bit DS immediate

typedef const __attribute__((address_space(3))) u64* lds_u64_ptr;

// The kernel has no static LDS, so its dynamic LDS starts at LDS address 0 and a table entry is addressed by
// its plain byte offset: nibble*8 in a VGPR plus an instruction immediate (at most 65535, i.e. the first 16 pairs;
// the up to 3 pairs beyond that take a plain loop with the pair's base added to the VGPR).
template <int OFFSET>
__device__ __forceinline__ u64 lds_entry(unsigned int nib8) {
    static_assert(OFFSET >= 0 && OFFSET <= 65536 - 8, "table offset must fit the DS immediate");
    return *(lds_u64_ptr)(uintptr_t)(nib8 + (unsigned)OFFSET);
}

// (byte B of x) & mask in one VALU op (SDWA byte select).
template <int B>
__device__ __forceinline__ unsigned int byte_and(unsigned int x, unsigned int mask) {
    unsigned int out;
    if (B == 0) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(out) : "v"(x), "v"(mask));
    if (B == 1) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(out) : "v"(x), "v"(mask));
    if (B == 2) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(out) : "v"(x), "v"(mask));
    if (B == 3) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(out) : "v"(x), "v"(mask));
    return out;
}

__device__ __forceinline__ u64 xor3_u64(u64 a, u64 b, u64 c) {
    const unsigned int lo = __builtin_amdgcn_bitop3_b32((unsigned int)a, (unsigned int)b, (unsigned int)c, 0x96);
    const unsigned int hi = __builtin_amdgcn_bitop3_b32((unsigned int)(a >> 32), (unsigned int)(b >> 32), (unsigned int)(c >> 32), 0x96);
    return ((u64)hi << 32) | lo;
}

// 8 table reads for the 8 nibbles of one dword; BASE = byte offset of the dword's first group.
template <int BASE>
__device__ __forceinline__ u64 lookup_dword(const unsigned int d, u64 a, unsigned int mask78) {
    const unsigned int lo = d << 3, hi = d >> 1;                // nibble*8 sits in bits 3..6 of each byte
    // two entries per v_bitop3_b32 (0x96: three-way XOR) and half of the accumulator: one instead of two XORs per lookup
    a = xor3_u64(a, lds_entry<BASE + 0 * 128>(byte_and<0>(lo, mask78)), lds_entry<BASE + 1 * 128>(byte_and<0>(hi, mask78)));
    a = xor3_u64(a, lds_entry<BASE + 2 * 128>(byte_and<1>(lo, mask78)), lds_entry<BASE + 3 * 128>(byte_and<1>(hi, mask78)));
    a = xor3_u64(a, lds_entry<BASE + 4 * 128>(byte_and<2>(lo, mask78)), lds_entry<BASE + 5 * 128>(byte_and<2>(hi, mask78)));
    a = xor3_u64(a, lds_entry<BASE + 6 * 128>(byte_and<3>(lo, mask78)), lds_entry<BASE + 7 * 128>(byte_and<3>(hi, mask78)));
    return a;
}

// 32 table reads for the 32 nibbles of one 16-byte error piece; BASE = byte offset of the pair's tables.
template <int BASE>
__device__ __forceinline__ u64 lookup_piece(const uint4 v, u64 a, unsigned int mask78) {
    // 8 reads in flight at a time: with 32 wavefronts per CU that is enough to keep the LDS busy, and it keeps the
    // kernel inside its 64-VGPR budget without spilling
    a = lookup_dword<BASE + 0 * 1024>(v.x, a, mask78);
    a = lookup_dword<BASE + 1 * 1024>(v.y, a, mask78);
    __builtin_amdgcn_sched_barrier(0);
    a = lookup_dword<BASE + 2 * 1024>(v.z, a, mask78);
    a = lookup_dword<BASE + 3 * 1024>(v.w, a, mask78);
    return a;
}

// Pairs t = 0 .. min(np, 16)-1 of the staged chunk, fully unrolled so every table offset is an immediate.  The piece
// of pair t+1 is requested before the 32 lookups of pair t (explicit one-deep prefetch; the scheduling barrier
// keeps the compiler from hoisting all 19 loads and spilling).
template <int T>
struct PairUnroll {
    static __device__ __forceinline__ void run(const uint4* __restrict__ tile, int lane, const int32_t* __restrict__ pairs,
                                               int np, const uint4 cur, u64& a, unsigned int mask78) {
        constexpr int t = SYN_UNROLL_PAIRS - T;
        uint4 next = cur;
        if (t + 1 < np) next = tile[(int64_t)pairs[t + 1] * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        a = lookup_piece<t * 4096>(cur, a, mask78);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < np) PairUnroll<T - 1>::run(tile, lane, pairs, np, next, a, mask78);
    }
};
template <>
struct PairUnroll<0> {
    static __device__ __forceinline__ void run(const uint4*, int, const int32_t*, int, const uint4, u64&, unsigned int) {}
};

// Blocks are dealt round-robin over the 8 XCDs, so blocks b and b+8 share an L2.  The blocks of one XCD get
// the same sample chunk and different slabs: the chunk's errors come from HBM once per XCD and are served
// from its L2 to the other slabs.
__global__ __launch_bounds__(SYN_THREADS, 6) void syndrome_tiled_kernel(
    const u64* __restrict__ tables, const int32_t* __restrict__ pair_list, const int32_t* __restrict__ npairs,
    int64_t max_pairs, int64_t slabs, int64_t r, int64_t ident_off, const uint64_t* __restrict__ e, int64_t batch,
    int64_t ldt, uint64_t* __restrict__ s, int64_t sstride, int64_t chunks, int tpw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tab[];
    const int64_t b = blockIdx.x;
    const int64_t xcd = b & 7, q = b >> 3;
    const int64_t slab = q % slabs;
    const int64_t chunk = (q / slabs) * 8 + xcd;
    if (chunk >= chunks) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: tile addresses stay in SGPRs
    const int64_t tiles = (batch + 63) >> 6;
    const int64_t tile0 = chunk * (int64_t)(SYN_WAVES * tpw) + wave;
    const int np_total = npairs[slab];
    const int32_t* pairs = pair_list + slab * max_pairs;
    unsigned int mask78 = 0x78u;
    asm volatile("" : "+v"(mask78));                            // keep the mask in a VGPR for the SDWA operand

    // Rows 64*slab.. of an identity block correspond to error bits ident_off + 64*slab ..
    const int64_t ibit = ident_off + slab * 64;
    const int64_t iw0 = ibit >> 6;
    const int ish = (int)(ibit & 63);
    const int64_t irows = r - slab * 64;
    const u64 imask = irows < 64 ? ~(~0ull << irows) : ~0ull;

    int p0 = 0;
    do {                                                        // one pass per staging of at most 19 pairs
        const int np = np_total - p0 < SYN_CHUNK_PAIRS ? np_total - p0 : SYN_CHUNK_PAIRS;
        if (p0) __syncthreads();
        {   // stage the tables of pairs p0 .. p0+np-1 at LDS offset 0 (16-byte copies, coalesced)
            const uint4* src = reinterpret_cast<const uint4*>(tables + (slab * max_pairs + p0) * 512);
            uint4* dst = reinterpret_cast<uint4*>(tab);
            for (int i = threadIdx.x; i < np * 256; i += SYN_THREADS) dst[i] = src[i];
        }
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < tpw; ++k) {
            const int64_t tile = tile0 + (int64_t)k * SYN_WAVES;
            if (tile >= tiles) break;
            const uint64_t* tbase = e + tile * 64 * ldt;
            u64 a = 0;
            if (np > 0) {
                const uint4* tptr = reinterpret_cast<const uint4*>(tbase);
                const uint4 first = tptr[(int64_t)pairs[p0] * 64 + lane];
                const int npu = np < SYN_UNROLL_PAIRS ? np : SYN_UNROLL_PAIRS;
                PairUnroll<SYN_UNROLL_PAIRS>::run(tptr, lane, pairs + p0, npu, first, a, mask78);
#pragma unroll 1
                for (int t = SYN_UNROLL_PAIRS; t < np; ++t) {              // the few pairs beyond the immediate range
                    const uint4 v = tptr[(int64_t)pairs[p0 + t] * 64 + lane];
                    const unsigned int d[4] = {v.x, v.y, v.z, v.w};
                    const unsigned int base = (unsigned int)t * 4096u;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned int lo = d[k] << 3, hi = d[k] >> 1;
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            a ^= *(lds_u64_ptr)(uintptr_t)(base + (k * 8 + 2 * b) * 128 + ((lo >> (8 * b)) & 0x78u));
                            a ^= *(lds_u64_ptr)(uintptr_t)(base + (k * 8 + 2 * b + 1) * 128 + ((hi >> (8 * b)) & 0x78u));
                        }
                    }
                }
            }
            // everything that is only needed once per tile comes after the lookups, so that no 64-bit address or
            // mask has to stay live across the unrolled body (spills there cost HBM traffic through scratch)
            const int64_t sample = tile * 64 + lane;
            if (sample < batch) {
                if (p0) {
                    a ^= s[slab * sstride + sample];
                } else if (ident_off >= 0) {
                    const uint64_t* lp = tbase + lane * 2;
                    u64 idv = lp[(iw0 >> 1) * 128 + (iw0 & 1)] >> ish;
                    if (ish && iw0 + 1 < ldt) idv |= lp[((iw0 + 1) >> 1) * 128 + ((iw0 + 1) & 1)] << (64 - ish);
                    a ^= idv & imask;
                }
                s[slab * sstride + sample] = a;                          // slab-major: 512 B per wavefront store
            }
        }
        p0 += SYN_CHUNK_PAIRS;
    } while (p0 < np_total);
}

// ---- sample-major -> tiled ----------------------------------------------------------------------------------

// grid (tiles), block 256.  Word w of sample b goes to (b>>6)*64*ldt + (w>>1)*128 + (b&63)*2 + (w&1).
__global__ __launch_bounds__(256) void retile_kernel(const uint64_t* __restrict__ src, int64_t batch, int64_t lde,
                                                     int64_t words, int64_t ldt, uint64_t* __restrict__ dst) {
    __shared__ u64 buf[64 * 65];
    const int64_t tile = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t w0 = 0; w0 < ldt; w0 += 64) {
        if (w0) __syncthreads();
        for (int sidx = wave; sidx < 64; sidx += 4) {
            const int64_t sample = tile * 64 + sidx;
            const int64_t w = w0 + lane;
            buf[sidx * 65 + lane] = (sample < batch && w < words) ? src[sample * lde + w] : 0ull;
        }
        __syncthreads();
        const int64_t pairs_here = (ldt - w0 < 64 ? ldt - w0 : 64) >> 1;
        for (int64_t qi = wave; qi < pairs_here; qi += 4) {
            ulonglong2 v;
            v.x = buf[lane * 65 + 2 * qi];
            v.y = buf[lane * 65 + 2 * qi + 1];
            reinterpret_cast<ulonglong2*>(dst + tile * 64 * ldt + ((w0 >> 1) + qi) * 128)[lane] = v;
        }
    }
}

// Slab-major syndromes (word s of sample b at in[s * in_stride + b]) -> sample-major (out[b * lds_out + s]).
// grid (ceil(batch / 64)), block 256; through LDS so that both sides are coalesced.
__global__ __launch_bounds__(256) void slab_to_sample_kernel(const uint64_t* __restrict__ in, int64_t in_stride,
                                                             int64_t batch, int64_t slabs, uint64_t* __restrict__ out,
                                                             int64_t lds_out) {
    __shared__ u64 buf[64 * 65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    for (int64_t s0 = 0; s0 < slabs; s0 += 64) {
        if (s0) __syncthreads();
        for (int si = wave; si < 64; si += 4) {
            const int64_t slab = s0 + si, sample = b0 + lane;
            buf[si * 65 + lane] = (slab < slabs && sample < batch) ? in[slab * in_stride + sample] : 0ull;
        }
        __syncthreads();
        for (int bi = wave; bi < 64; bi += 4) {
            const int64_t sample = b0 + bi, slab = s0 + lane;
            if (sample < batch && slab < slabs) out[sample * lds_out + slab] = buf[lane * 65 + bi];
        }
    }
}

// ---- small codes: n <= 64, r <= 64 -----------------------------------------------------------------------

struct SmallRows {
    u64 row[64];
};

__global__ __launch_bounds__(256) void syndrome_small_kernel(SmallRows rows, int r, const uint64_t* __restrict__ e,
                                                             int64_t batch, int64_t lde,
                                                             uint64_t* __restrict__ s, int64_t lds_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (lde == 1 && lds_out == 1 && ((((uintptr_t)e) | ((uintptr_t)s)) & 15) == 0) {
        // one word per sample on both sides: two samples per lane, 16-byte accesses
        const int64_t pairs = (batch + 1) >> 1;
        for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < pairs; p += stride) {
            const bool two = 2 * p + 1 < batch;
            ulonglong2 v;
            if (two) {
                v = reinterpret_cast<const ulonglong2*>(e)[p];
            } else {
                v.x = e[2 * p];
                v.y = 0;
            }
            ulonglong2 out = make_ulonglong2(0ull, 0ull);
            for (int k = 0; k < r; ++k) {
                out.x |= (u64)(__popcll(rows.row[k] & v.x) & 1) << k;
                out.y |= (u64)(__popcll(rows.row[k] & v.y) & 1) << k;
            }
            if (two)
                reinterpret_cast<ulonglong2*>(s)[p] = out;
            else
                s[2 * p] = out.x;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < batch; i += stride) {
        const u64 v = e[i * lde];
        u64 out = 0;
        for (int k = 0; k < r; ++k) out |= (u64)(__popcll(rows.row[k] & v) & 1) << k;
        s[i * lds_out] = out;
    }
}

// Two adjacent sample words per lane: every access is 16 bytes per lane, 1 KiB per wavefront instruction.
template <int NMAX>
__global__ __launch_bounds__(256) void syndrome_sliced_kernel(SmallRows rows, int r, int n,
                                                              const uint64_t* __restrict__ e, int64_t words,
                                                              int64_t lde, uint64_t* __restrict__ s,
                                                              int64_t lds_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t pairs = (words + 1) >> 1;
    const bool wide = ((lde | lds_out) & 1) == 0 && ((((uintptr_t)e) | ((uintptr_t)s)) & 15) == 0;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < pairs; b += stride) {
        const bool two = 2 * b + 1 < words;
        ulonglong2 v[NMAX];
#pragma unroll
        for (int q = 0; q < NMAX; ++q) {
            v[q] = make_ulonglong2(0ull, 0ull);
            if (q < n) {
                const uint64_t* src = e + (int64_t)q * lde + 2 * b;
                if (wide && two) {
                    v[q] = *reinterpret_cast<const ulonglong2*>(src);
                } else {
                    v[q].x = src[0];
                    if (two) v[q].y = src[1];
                }
            }
        }
        for (int k = 0; k < r; ++k) {
            const u64 row = rows.row[k];
            ulonglong2 acc = make_ulonglong2(0ull, 0ull);
#pragma unroll
            for (int q = 0; q < NMAX; ++q) {
                const u64 m = 0ull - ((row >> q) & 1ull);
                acc.x ^= v[q].x & m;
                acc.y ^= v[q].y & m;
            }
            uint64_t* dst = s + (int64_t)k * lds_out + 2 * b;
            if (wide && two) {
                *reinterpret_cast<ulonglong2*>(dst) = acc;
            } else {
                dst[0] = acc.x;
                if (two) dst[1] = acc.y;
            }
        }
    }
}

// ---- histograms ------------------------------------------------------------------------------------------

#define HIST_LDS_BINS 8192

// mode 0: key = big-endian integer of the r syndrome bits (row 0 most significant, bin_matrix.py:36-43);
// mode 1: key = Hamming weight.  Bins privatised in LDS when they fit, one global atomic per bin and block.
// slab_major: word w of sample i at s[w * lds_in + i] (lane-coalesced), else at s[i * lds_in + w].
__global__ __launch_bounds__(256) void histogram_kernel(const uint64_t* __restrict__ s, int64_t batch, int64_t lds_in,
                                                        int r, int mode, int slab_major, u64* __restrict__ hist,
                                                        int64_t nbins) {
    __shared__ unsigned int bins[HIST_LDS_BINS];
    const bool priv = nbins <= HIST_LDS_BINS;
    if (priv) {
        for (int i = threadIdx.x; i < nbins; i += blockDim.x) bins[i] = 0;
        __syncthreads();
    }
    const int words = (r + 63) >> 6;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < batch; i += stride) {
        const uint64_t* row = slab_major ? s + i : s + i * lds_in;
        const int64_t step = slab_major ? lds_in : 1;
        u64 key;
        if (mode == GF2_HIST_FULL) {
            key = r ? (__brevll(row[0]) >> (64 - r)) : 0ull;
        } else {
            key = 0;
            for (int w = 0; w < words; ++w) key += __popcll(row[w * step]);
        }
        if (priv)
            atomicAdd(&bins[key], 1u);
        else
            atomicAdd(&hist[key], 1ull);
    }
    if (priv) {
        __syncthreads();
        for (int i = threadIdx.x; i < nbins; i += blockDim.x)
            if (bins[i]) atomicAdd(&hist[i], (u64)bins[i]);
    }
}

// ---- host side -------------------------------------------------------------------------------------------

static inline int host_bit(const uint64_t* row, int64_t j) { return (int)((row[j >> 6] >> (j & 63)) & 1ull); }

// Offset of an identity block H[:, off:off+r] == I, or -1.  Candidates are the set bits of row 0 whose column
// is a unit vector; each survivor is verified in full.
static int64_t find_identity_block(const uint64_t* h, int64_t r, int64_t n, int64_t ld) {
    if (r == 0 || r > n) return -1;
    for (int64_t c = 0; c + r <= n; ++c) {
        if (!host_bit(h, c)) continue;
        bool ok = true;
        for (int64_t i = 0; i < r && ok; ++i)                           // diagonal set, column c clear below row 0
            ok = host_bit(h + i * ld, c + i) && (i == 0 || !host_bit(h + i * ld, c));
        for (int64_t i = 0; i < r && ok; ++i) {                         // row i restricted to the block == unit vector
            const uint64_t* row = h + i * ld;
            for (int64_t w = c >> 6; w <= (c + r - 1) >> 6 && ok; ++w) {
                u64 m = ~0ull;
                const int64_t lo = c - w * 64, hi = c + r - w * 64;
                if (lo > 0) m &= ~0ull << lo;
                if (hi < 64) m &= ~(~0ull << hi);
                const int64_t d = c + i - w * 64;
                const u64 want = (d >= 0 && d < 64) ? (1ull << d) : 0ull;
                ok = (row[w] & m) == want;
            }
        }
        if (ok) return c;
    }
    return -1;
}

extern "C" {

int64_t gf2_tiled_ld(int64_t n) {
    const int64_t w = gf2_words(n) > 0 ? gf2_words(n) : 1;
    return (w + 1) & ~(int64_t)1;
}

int64_t gf2_tiled_words(int64_t n, int64_t batch) { return gf2_cdiv(batch > 0 ? batch : 1, 64) * 64 * gf2_tiled_ld(n); }

int gf2_check_create(gf2_ctx* ctx, const uint64_t* h, int64_t r, int64_t n, int64_t ld, gf2_check** check_out) {
    if (!ctx || !check_out) GF2_FAIL(GF2_E_ARG, "gf2_check_create: null argument");
    *check_out = nullptr;
    if (r < 0 || n < 0 || ld < gf2_words(n) || (!h && r > 0 && n > 0))
        GF2_FAIL(GF2_E_ARG, "gf2_check_create: bad shape r=%lld n=%lld ld=%lld", (long long)r, (long long)n, (long long)ld);
    GF2_TRY(gf2_ctx_activate(ctx));
    gf2_check* ck = (gf2_check*)calloc(1, sizeof(gf2_check));
    if (!ck) GF2_FAIL(GF2_E_NOMEM, "gf2_check_create: out of host memory");
    ck->r = r;
    ck->n = n;
    ck->ld = ld > 0 ? ld : 1;
    ck->slabs = gf2_cdiv(r, 64);
    ck->ldt = gf2_tiled_ld(n);
    ck->ident_off = -1;
    ck->small = n <= 64 && r <= 64;
    if (ck->small && n > 0)
        for (int64_t i = 0; i < r; ++i) ck->rows_small[i] = h[i * ld];
    int rc = GF2_OK;
    const size_t hbytes = (size_t)(r > 0 ? r : 1) * ck->ld * 8;
    if ((rc = gf2_dev_alloc(ctx, hbytes, (void**)&ck->h_dev)) != GF2_OK) goto fail;
    if (r > 0 && n > 0 && (rc = gf2_h2d(ctx, ck->h_dev, h, (size_t)r * ld * 8)) != GF2_OK) goto fail;

    if (n <= 4096 && n > 0)
        for (int64_t i = 0; i < r; ++i)
            for (int64_t w = 0; w < gf2_words(n); ++w) ck->col_any[w] |= h[i * ld + w];
    if (!ck->small && r > 0 && n > 0) {
        ck->ident_off = find_identity_block(h, r, n, ld);
        // per-slab list of 128-column pairs with a non-zero column outside the identity block
        const int64_t pairs_total = ck->ldt / 2;
        std::vector<std::vector<int32_t>> lists((size_t)ck->slabs);
        int64_t max_pairs = 1;
        for (int64_t s = 0; s < ck->slabs; ++s) {
            for (int64_t q = 0; q < pairs_total; ++q) {
                u64 any = 0;
                for (int half = 0; half < 2; ++half) {
                    const int64_t w = 2 * q + half;
                    if (w >= ld) continue;
                    u64 colmask = 0;
                    for (int64_t i = s * 64; i < r && i < s * 64 + 64; ++i) colmask |= h[i * ld + w];
                    if (ck->ident_off >= 0) {
                        const int64_t lo = ck->ident_off - w * 64, hi = ck->ident_off + r - w * 64;
                        if (hi > 0 && lo < 64) {
                            u64 m = ~0ull;
                            if (lo > 0) m &= ~0ull << lo;
                            if (hi < 64) m &= ~(~0ull << hi);
                            colmask &= ~m;
                        }
                    }
                    any |= colmask;
                }
                if (any) lists[(size_t)s].push_back((int32_t)q);
            }
            if ((int64_t)lists[(size_t)s].size() > max_pairs) max_pairs = (int64_t)lists[(size_t)s].size();
        }
        ck->max_pairs = max_pairs;
        std::vector<int32_t> flat((size_t)(ck->slabs * max_pairs), 0), counts((size_t)ck->slabs, 0);
        for (int64_t s = 0; s < ck->slabs; ++s) {
            counts[(size_t)s] = (int32_t)lists[(size_t)s].size();
            for (size_t t = 0; t < lists[(size_t)s].size(); ++t) flat[(size_t)(s * max_pairs) + t] = lists[(size_t)s][t];
        }
        const size_t tbytes = (size_t)ck->slabs * max_pairs * 512 * 8;
        if ((rc = gf2_dev_alloc(ctx, flat.size() * 4, (void**)&ck->pair_list_dev)) != GF2_OK) goto fail;
        if ((rc = gf2_dev_alloc(ctx, counts.size() * 4, (void**)&ck->npairs_dev)) != GF2_OK) goto fail;
        if ((rc = gf2_dev_alloc(ctx, tbytes, (void**)&ck->tables_dev)) != GF2_OK) goto fail;
        if ((rc = gf2_h2d(ctx, ck->pair_list_dev, flat.data(), flat.size() * 4)) != GF2_OK) goto fail;
        if ((rc = gf2_h2d(ctx, ck->npairs_dev, counts.data(), counts.size() * 4)) != GF2_OK) goto fail;
        dim3 grid((unsigned)ck->slabs, (unsigned)max_pairs);
        hipLaunchKernelGGL(build_tables_kernel, grid, dim3(64), 0, ctx->stream, ck->h_dev, r, ck->ld, ck->ident_off,
                           ck->pair_list_dev, ck->npairs_dev, max_pairs, (u64*)ck->tables_dev);
        hipError_t err = hipGetLastError();
        if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
        if (err != hipSuccess) {
            gf2_set_error("build_tables_kernel failed: %s", hipGetErrorString(err));
            rc = GF2_E_HIP;
            goto fail;
        }
    }
    if ((rc = gf2_build_columns(ctx, ck)) != GF2_OK) goto fail;
    if ((rc = gf2_build_slab_table(ctx, ck)) != GF2_OK) goto fail;
    if ((rc = gf2_build_lane_table(ctx, ck)) != GF2_OK) goto fail;
    *check_out = ck;
    return GF2_OK;
fail:
    if (ck->lane_tab_dev) (void)hipFree(ck->lane_tab_dev);
    if (ck->slab_tab_dev) (void)hipFree(ck->slab_tab_dev);
    if (ck->ht_dev) (void)hipFree(ck->ht_dev);
    if (ck->h_dev) (void)hipFree(ck->h_dev);
    if (ck->tables_dev) (void)hipFree(ck->tables_dev);
    if (ck->pair_list_dev) (void)hipFree(ck->pair_list_dev);
    if (ck->npairs_dev) (void)hipFree(ck->npairs_dev);
    free(ck);
    return rc;
}

int gf2_check_destroy(gf2_ctx* ctx, gf2_check* check) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_check_destroy: null context");
    if (!check) return GF2_OK;
    GF2_TRY(gf2_dev_free(ctx, check->h_dev));
    GF2_TRY(gf2_dev_free(ctx, check->tables_dev));
    GF2_TRY(gf2_dev_free(ctx, check->pair_list_dev));
    GF2_TRY(gf2_dev_free(ctx, check->npairs_dev));
    GF2_TRY(gf2_dev_free(ctx, check->ht_dev));
    GF2_TRY(gf2_dev_free(ctx, check->slab_tab_dev));
    GF2_TRY(gf2_dev_free(ctx, check->lane_tab_dev));
    free(check);
    return GF2_OK;
}

int gf2_retile_dev(gf2_ctx* ctx, const uint64_t* e_dev, int64_t batch, int64_t lde, int64_t n, uint64_t* tiled_dev) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_retile_dev: null context");
    if (batch < 0 || n < 0 || lde < gf2_words(n) || lde < 1) GF2_FAIL(GF2_E_ARG, "gf2_retile_dev: bad shape");
    if (batch == 0) return GF2_OK;
    if (!e_dev || !tiled_dev) GF2_FAIL(GF2_E_ARG, "gf2_retile_dev: null buffer");
    GF2_TRY(gf2_ctx_activate(ctx));
    const int64_t tiles = gf2_cdiv(batch, 64);
    if (tiles > 0x7fffffffLL) GF2_FAIL(GF2_E_ARG, "gf2_retile_dev: batch too large");
    hipLaunchKernelGGL(retile_kernel, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, e_dev, batch, lde, gf2_words(n),
                       gf2_tiled_ld(n), tiled_dev);
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

// Writes slab-major syndromes: word s of sample b at s_dev[s * sstride + b].
static int launch_tiled(gf2_ctx* ctx, const gf2_check* ck, const uint64_t* e_tiled, int64_t batch, uint64_t* s_dev,
                        int64_t sstride) {
    const int64_t tiles = gf2_cdiv(batch, 64);
    // Every workgroup stages its slab's tables (64 - 76 KiB) before it touches a sample: with 3840 samples per workgroup a launch
    // of 2^20 samples read 594 MB of tables for 537 MB of errors (round 2's "1.86 x algorithmic" traffic was mostly this).  Large
    // batches therefore get as many tiles per wavefront as leave about four workgroups per CU over the launch (two are resident).
    int64_t chunks_target = gf2_cdiv(4 * (int64_t)ctx->num_cus, ck->slabs > 0 ? ck->slabs : 1);
    chunks_target = gf2_cdiv(chunks_target, 8) * 8;
    int64_t tpw = gf2_cdiv(tiles, chunks_target * SYN_WAVES);
    if (tpw < SYN_TPW_MIN) tpw = SYN_TPW_MIN;
    const int64_t chunks = gf2_cdiv(tiles, SYN_WAVES * tpw);
    const int64_t chunks8 = gf2_cdiv(chunks, 8) * 8;
    const int64_t blocks = chunks8 * ck->slabs;
    if (blocks > 0x7fffffffLL) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: batch too large for one launch");
    const int64_t stage_pairs = ck->max_pairs < SYN_CHUNK_PAIRS ? ck->max_pairs : SYN_CHUNK_PAIRS;
    const size_t shmem = (size_t)stage_pairs * 4096;
    if (!ctx->lds_optin[0]) {
        GF2_HIP(hipFuncSetAttribute((const void*)syndrome_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    SYN_CHUNK_PAIRS * 4096));
        ctx->lds_optin[0] = true;
    }
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_SYNDROME));
    hipLaunchKernelGGL(syndrome_tiled_kernel, dim3((unsigned)blocks), dim3(SYN_THREADS), shmem, ctx->stream,
                       (const u64*)ck->tables_dev, ck->pair_list_dev, ck->npairs_dev, ck->max_pairs, ck->slabs, ck->r,
                       ck->ident_off, e_tiled, batch, ck->ldt, s_dev, sstride, chunks, (int)tpw);
    GF2_TRY(gf2_prof_end(ctx));
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

int gf2_syndrome_dev(gf2_ctx* ctx, const gf2_check* ck, const uint64_t* e_dev, int64_t batch, int64_t lde,
                     int layout, uint64_t* s_dev, int64_t lds) {
    if (!ctx || !ck) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: null argument");
    if (batch < 0) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: negative batch");
    if (batch == 0 || ck->r == 0) return GF2_OK;
    if (!e_dev || !s_dev) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: null buffer");
    GF2_TRY(gf2_ctx_activate(ctx));

    if (layout == GF2_LAYOUT_BIT_SLICED) {
        if (!ck->small) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: bit-sliced layout needs n <= 64 and r <= 64");
        const int64_t words = gf2_words(batch);
        if (lde < words || lds < words) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: bit-sliced strides too small");
        SmallRows rows;
        memcpy(rows.row, ck->rows_small, sizeof(rows.row));
        int64_t blocks = gf2_cdiv(gf2_cdiv(words, 2), 256);
        if (blocks > 8192) blocks = 8192;
        dim3 grid((unsigned)blocks), block(256);
        GF2_TRY(gf2_prof_begin(ctx, GF2_K_SYNDROME));
        const int n = (int)ck->n, r = (int)ck->r;
        if (n <= 8)
            hipLaunchKernelGGL(syndrome_sliced_kernel<8>, grid, block, 0, ctx->stream, rows, r, n, e_dev, words, lde, s_dev, lds);
        else if (n <= 16)
            hipLaunchKernelGGL(syndrome_sliced_kernel<16>, grid, block, 0, ctx->stream, rows, r, n, e_dev, words, lde, s_dev, lds);
        else if (n <= 32)
            hipLaunchKernelGGL(syndrome_sliced_kernel<32>, grid, block, 0, ctx->stream, rows, r, n, e_dev, words, lde, s_dev, lds);
        else
            hipLaunchKernelGGL(syndrome_sliced_kernel<64>, grid, block, 0, ctx->stream, rows, r, n, e_dev, words, lde, s_dev, lds);
        GF2_TRY(gf2_prof_end(ctx));
        GF2_HIP(hipGetLastError());
        return GF2_OK;
    }
    if (layout == GF2_LAYOUT_TILED) {
        if (ck->small) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: the tiled layout is for n > 64 or r > 64");
        if (lds < batch) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: slab-major output needs lds >= batch");
        return launch_tiled(ctx, ck, e_dev, batch, s_dev, lds);
    }
    if (layout != GF2_LAYOUT_SAMPLE_MAJOR) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: unknown layout %d", layout);
    if (lds < ck->slabs) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: lds too small");
    if (lde < gf2_words(ck->n) || lde < 1) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: lde too small");

    if (ck->small) {
        SmallRows rows;
        memcpy(rows.row, ck->rows_small, sizeof(rows.row));
        int64_t blocks = gf2_cdiv(batch, 256);
        if (blocks > 8192) blocks = 8192;
        GF2_TRY(gf2_prof_begin(ctx, GF2_K_SYNDROME));
        hipLaunchKernelGGL(syndrome_small_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, rows, (int)ck->r,
                           e_dev, batch, lde, s_dev, lds);
        GF2_TRY(gf2_prof_end(ctx));
        GF2_HIP(hipGetLastError());
        return GF2_OK;
    }
    // sample-major in and out around the table kernel: one streaming pass into the tiled layout first, and the
    // slab-major result transposed back at the end
    const size_t tbytes = ((size_t)gf2_tiled_words(ck->n, batch) * 8 + 255) & ~(size_t)255;
    const int64_t sstride = gf2_cdiv(batch, 64) * 64;
    GF2_TRY(gf2_ws_reserve(ctx, 1, tbytes + (size_t)ck->slabs * sstride * 8));
    uint64_t* tiled = (uint64_t*)ctx->ws[1];
    uint64_t* slabbed = (uint64_t*)((char*)ctx->ws[1] + tbytes);
    GF2_TRY(gf2_retile_dev(ctx, e_dev, batch, lde, ck->n, tiled));
    GF2_TRY(launch_tiled(ctx, ck, tiled, batch, slabbed, sstride));
    hipLaunchKernelGGL(slab_to_sample_kernel, dim3((unsigned)gf2_cdiv(batch, 64)), dim3(256), 0, ctx->stream, slabbed,
                       sstride, batch, ck->slabs, s_dev, lds);
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

int gf2_syndrome_batch(gf2_ctx* ctx, const uint64_t* h, int64_t r, int64_t n, int64_t ldh, const uint64_t* e,
                       int64_t batch, int64_t lde, int layout, uint64_t* s_out, int64_t lds) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_batch: null context");
    if (batch < 0 || r < 0 || n < 0) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_batch: negative size");
    if (batch == 0 || r == 0) return GF2_OK;
    if (!e || !s_out || !h) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_batch: null buffer");
    // every pitch is checked before a host buffer is touched (the density probe below reads e)
    if (ldh < gf2_words(n) || ldh < 1) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_batch: ldh too small");
    if (layout == GF2_LAYOUT_SAMPLE_MAJOR && (lde < gf2_words(n) || lde < 1)) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_batch: lde too small");
    if (layout == GF2_LAYOUT_BIT_SLICED && (lde < gf2_cdiv(batch, 64) || lds < gf2_cdiv(batch, 64) || n > 64 || r > 64))
        GF2_FAIL(GF2_E_ARG, "gf2_syndrome_batch: bit-sliced layout needs n, r <= 64 and lde, lds >= ceil(batch / 64)");
    if (layout == GF2_LAYOUT_TILED && lds < batch) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_batch: slab-major syndromes need lds >= batch");
    if (layout != GF2_LAYOUT_SAMPLE_MAJOR && layout != GF2_LAYOUT_BIT_SLICED && layout != GF2_LAYOUT_TILED)
        GF2_FAIL(GF2_E_ARG, "gf2_syndrome_batch: unknown layout %d", layout);
    int64_t e_words, s_rows;
    if (layout == GF2_LAYOUT_BIT_SLICED) {
        e_words = n * lde;
        s_rows = r;
    } else if (layout == GF2_LAYOUT_TILED) {
        e_words = gf2_tiled_words(n, batch);
        s_rows = gf2_cdiv(r, 64);                          // slab-major output: ceil(r/64) rows of lds >= batch words
    } else {
        e_words = batch * lde;
        s_rows = batch;
    }
    if (layout == GF2_LAYOUT_SAMPLE_MAJOR && lds < gf2_cdiv(r, 64)) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_batch: lds too small");
    gf2_check* ck = nullptr;
    uint64_t *e_dev = nullptr, *s_dev = nullptr;
    int rc = gf2_check_create(ctx, h, r, n, ldh, &ck);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)e_words * 8, (void**)&e_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)s_rows * lds * 8, (void**)&s_dev);
    if (rc == GF2_OK) rc = gf2_h2d(ctx, e_dev, e, (size_t)e_words * 8);
    if (rc == GF2_OK) rc = gf2_dev_zero(ctx, s_dev, (size_t)s_rows * lds * 8);
    // Sample-major host errors: their density is known for free here, so sparse batches (at most ~160 set bits per
    // error) go to the column kernel, everything else to the table kernel.
    bool sparse = false;
    if (rc == GF2_OK && layout == GF2_LAYOUT_SAMPLE_MAJOR && ck->ht_dev) {
        const int64_t probe = batch < 4096 ? batch : 4096, step = batch / probe;
        int64_t bits = 0;
        for (int64_t i = 0; i < probe; ++i)
            for (int64_t w = 0; w < gf2_words(n); ++w) bits += __builtin_popcountll(e[i * step * lde + w]);
        sparse = (double)bits / (double)probe <= 160.0;
    }
    if (rc == GF2_OK) {
        if (sparse)
            rc = gf2_syndrome_sparse_dev(ctx, ck, e_dev, batch, lde, s_dev, lds, nullptr, 0);
        else
            rc = gf2_syndrome_dev(ctx, ck, e_dev, batch, lde, layout, s_dev, lds);
    }
    if (rc == GF2_OK) rc = gf2_d2h(ctx, s_out, s_dev, (size_t)s_rows * lds * 8);
    gf2_dev_free(ctx, e_dev);
    gf2_dev_free(ctx, s_dev);
    gf2_check_destroy(ctx, ck);
    return rc;
}

int gf2_matmul_abt(gf2_ctx* ctx, const uint64_t* a, int64_t ra, int64_t lda, const uint64_t* b, int64_t rb,
                   int64_t ldb, int64_t n, uint64_t* c, int64_t ldc) {
    // C[i][j] = <A_i, B_j>: the rows of A are the "errors", B is the "parity check".
    if (ldc < gf2_cdiv(rb, 64)) GF2_FAIL(GF2_E_ARG, "gf2_matmul_abt: ldc too small");
    return gf2_syndrome_batch(ctx, b, rb, n, ldb, a, ra, lda, GF2_LAYOUT_SAMPLE_MAJOR, c, ldc);
}

int gf2_histogram_dev(gf2_ctx* ctx, const uint64_t* s_dev, int64_t batch, int64_t lds, int layout, int64_t r, int mode,
                      uint64_t* hist_dev, int64_t nbins) {
    if (!ctx || !hist_dev) GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: null argument");
    if (batch < 0 || r < 0) GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: negative size");
    if (mode == GF2_HIST_FULL) {
        if (r > 24) GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: full histogram needs r <= 24 (r=%lld)", (long long)r);
        if (nbins != (1ll << r)) GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: full histogram needs 2^r bins");
    } else if (mode == GF2_HIST_WEIGHT) {
        if (nbins != r + 1) GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: weight histogram needs r+1 bins");
    } else {
        GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: unknown mode %d", mode);
    }
    if (batch == 0) return GF2_OK;
    if (layout != GF2_LAYOUT_SAMPLE_MAJOR && layout != GF2_LAYOUT_TILED)
        GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: layout must be sample-major or tiled (slab-major)");
    const bool slab_major = layout == GF2_LAYOUT_TILED;
    if (!s_dev || lds < 1 || (slab_major ? lds < batch : lds < gf2_cdiv(r, 64)))
        GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: bad syndrome buffer");
    GF2_TRY(gf2_ctx_activate(ctx));
    int64_t blocks = gf2_cdiv(batch, 256 * 8);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_HIST));
    hipLaunchKernelGGL(histogram_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, s_dev, batch, lds, (int)r,
                       mode, slab_major ? 1 : 0, (u64*)hist_dev, nbins);
    GF2_TRY(gf2_prof_end(ctx));
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

}  // extern "C"
