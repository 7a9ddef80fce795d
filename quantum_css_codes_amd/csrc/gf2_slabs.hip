// Sparse-error syndrome weights with the transposed check resident in LDS (gfx950).
//
// Same product as np.mod(np.matmul(parity_check, e), 2) (css_code.py:728), reduced to the weight histogram the
// Monte-Carlo workload keeps.  The wavefront-per-sample kernel (gf2_sparse.hip) gathers 256-byte columns of the
// transposed check from L2 and is bound by L2 gather bandwidth (about 14 column reads per sample and component at
// the benchmark's p = 0.01).  Here the check never moves: the rows are cut into slabs of 512, every CU keeps ONE slab
// of the transposed check in LDS for the whole launch (64 bytes per non-identity column, 128 KiB at n = 4096), and
// the samples visit the slabs.  Three kernels on the context's stream:
//
//   compact   wavefront = 64 samples at a time, lane = 64-bit word of the packed error row (512 contiguous bytes per
//             load; words inside the identity block are not read).  Non-zero words are compacted (ballot + mbcnt)
//             into a per-wave item list; then lane = item, and each set bit appends its column's ordinal to the
//             sample's 64-byte record (slot 0 = count, up to 31 columns, slots taken with an LDS atomic; order is
//             irrelevant to an XOR).  Records go to a workspace, 4 KiB contiguous per wavefront.  A sample with more
//             columns than a record holds is finished on the spot by the wavefront-per-sample routine.
//   gather    workgroup = (slab, share of the samples), 16 wavefronts, no barrier after the table is loaded.  Four
//             lanes per sample, 128 rows each: start from the identity-block bits of the error row (64 contiguous
//             bytes per sample and slab, read straight from the packed errors), XOR one ds_read_b128 per listed column
//             (unused slots point at a zero entry, so there is no per-lane predicate), popcount, 2 DPP adds, one
//             16-bit partial weight per sample and slab.
//   combine   lane = sample: adds the partial weights and counts the total in an LDS-privatised histogram.
//
// Every byte of the errors is read once from HBM (non-identity words by compact, identity words by gather); the
// records add 64 bytes written and 64 bytes read per slab and sample.  A lookup is 64 bytes wide per sample, so a
// group of 16 lanes holds 4 samples on 4 sixteen-bank quarters: 2.1 LDS cycles per group on average with random
// columns.
#include <type_traits>

#include "gf2_internal.h"

// Cache policy of the streams that are read exactly once (round 5): the compact kernel's packed error rows and
// the gather kernel's identity words.  GF2_SLAB_NT: 0 = default policy, 1 = rows non-temporal, 2 = identity words non-temporal,
// 3 = both (profiles/r05_nt.sh builds the four libraries and alternates them on one box; profiles/r05_nt.md holds what came of it).
#ifndef GF2_SLAB_NT
#define GF2_SLAB_NT 0
#endif
#if GF2_SLAB_NT & 1
#define GF2_ROWS_POLICY " nt"
#else
#define GF2_ROWS_POLICY ""
#endif
#if GF2_SLAB_NT & 2
#define GF2_IDENT_POLICY " nt"
#else
#define GF2_IDENT_POLICY ""
#endif
#include "gf2_sparse_dev.h"
#include "gf2_sampler.h"

#define CMP_WAVES 4
#define CMP_THREADS (64 * CMP_WAVES)
#define CMP_SUB 8                             // samples per compaction sub-pass
#define REC_SLOTS 32                          // 16-bit slots per record: 31 columns (slots 0 .. 30) + the header in the LAST slot
#define REC_OVER 0xFFu                        // count byte of a sample that was finished by the slow routine
#define REC_DONE 0x4000u                      // header of a sample finished by the slow routine whose count byte still stands
#define REC_FLAG 0xFFFFu                      // its partial weights
#define REC_STRAY 0xFF80u                     // | tile-local sample index: partial weight of a sample that has a column compact left out
#define GAT_THREADS 1024
#define GAT_WAVES 16
#define SLAB_ROWS 512
#define SLAB_MAX_COLS 2400                    // (cols + 1) * 64 bytes of LDS <= 150 KiB
#define SLAB_MAX_BINS 2304
#define SLAB_MAX_BATCH (1 << 24)              // most samples per pass through the workspace (record positions and buffer offsets are 32-bit)
#define SLAB_GATHER_LDS_MAX (160 * 1024)      // what a workgroup may ask for: the slab's table, the ticket counter and, when the previous pass'
                                              // combine step rides along, its counters
#define SLAB_DEFAULT_BATCH (1 << 22)              // (2^21 until round 3: 2^22 is 5 % faster on two streams, 2^23 and 2^24 no more: profiles/r03_sweep_pass.log)

// Records of a tile of 64 samples, sorted by count: the 32 shortest as 32-byte records (15 columns + header), the 32 longest
// as 64-byte records (31 columns + header) -- 3 KiB per tile instead of 4.  Seven samples in ten list at most 15 columns at the
// benchmark's rate, so the lower half of a sorted tile fits (a tile where it does not -- more than 32 samples with 16 or more
// columns -- has the misfits finished on the spot like samples beyond 31 columns).  Group g = tile * 4 + quartile holds 16 records.
// The header (count | tile-local sample << 8 | flags) is the record's LAST slot (15 or 31), the columns start at slot 0: the
// gather kernels look up ceil(count / 2) slot pairs and never the header (round 2 kept it in slot 0 and paid a lookup of the
// zero entry per record for it).
#define TILE_REC_BYTES 3072
#define SHORT_SLOTS 16
__device__ __forceinline__ unsigned int record_quarter_offset(unsigned int g, unsigned int lane_rec, unsigned int part) {
    const unsigned int tile = g >> 2, q = g & 3u;
    return tile * TILE_REC_BYTES + (q < 2 ? q * 512u + lane_rec * 32u + (part & 1u) * 16u : (q - 1u) * 1024u + lane_rec * 64u + part * 16u);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u32x4* lds_u32x4_ptr;

__device__ __forceinline__ unsigned int wave_max(unsigned int v) {
    v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xF, 0xF, true));
    v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xF, 0xF, true));
    v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0u, v, 0x141, 0xF, 0xF, true));
    v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0u, v, 0x140, 0xF, 0xF, true));
    v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0u, v, 0x142, 0xA, 0xF, true));
    v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0u, v, 0x143, 0xC, 0xF, true));
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ unsigned int xor3(unsigned int a, unsigned int b, unsigned int c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// ---- compact ---------------------------------------------------------------------------------------------------------

struct CompactArgs {
    const u64* e;
    const uint32_t* ht;        // transposed check of the wavefront-per-sample routine (samples beyond a record)
    u32x4* rec;                // ceil(batch / 64) * 64 records of 64 bytes
    u64* hist;                 // null: syndromes only
    uint32_t* syn;             // syndromes out (null: none): samples finished here store theirs, lds32 dwords per sample
    int64_t lds32;
    int64_t batch, lde;
    int r, n, ident_off, null_ord;
    u64 skip_words;            // words whose few non-identity columns are left to the redo pass (bit w = word w)
    u64* clk;                  // debugging (GF2_F_DIAG_CLOCKS): earliest entry / latest exit of the workgroups, else null
    // the PREVIOUS pass' partial weights, summed into the histogram on the way in (null: nothing to combine) -- the combine kernel's
    // work without its launch between two passes of a call
    const unsigned short* cmb_pw;
    int64_t cmb_positions, cmb_pad;
    unsigned int cmb_sample0;  // index of the previous pass' first sample in the call's batch (redo list entries are call-wide)
    int cmb_nslabs, cmb_nbins;
    unsigned int* redo_count;
    unsigned int* redo_list;
};

// LDS per wavefront.  Non-zero (sample, word) pairs wait in a ring until 64 of them are there, so every pass over them has all
// lanes busy; a pass writes the FIRST column of each word and puts words with more columns back at the ring's tail, with the
// slot their next column goes to (round 3; before, they went on a list of their own that was worked off by a second routine
// with a loop over the bits of each lane's word).  The 64 records of the tile are 64 contiguous bytes each (slot k of sample j
// at 64 j + 2 k); a column beyond slot 30 is written to the record's own slot 31, the header's place, which is filled in last.
#define CMP_RING 256
#define CMP_CONT 0x40000u                 // ring_m: the word has been here before, bits 12..17 = the slot of its next column
struct alignas(16) CompactWaveLds {
    u64 ring_v[CMP_RING];                 // word (the columns still to write)
    unsigned int ring_m[CMP_RING];        // word index | tile-local sample << 6 | next slot << 12 | CMP_CONT
    unsigned char rec[4096 + 16];
    unsigned int cnt[64];
};
static_assert(CMP_RING * 12 >= (SPARSE_LIST_CAP + 8) * 4, "the slow routine's list reuses the ring");

// The combine step for the positions this workgroup takes (every 4 * blockDim * gridDim): adds the (up to four) slabs' partial
// weights of a record position and counts the total in `bins` (LDS, nbins counters, zeroed here), which it then adds to the
// histogram -- one global atomic per non-empty bin and workgroup.  Positions a slab has flagged REC_STRAY go on the redo list
// (as sample indices + sample0) through `staging` (1024 entries) instead.  Called by all threads of the workgroup.
__device__ __forceinline__ void combine_positions(const unsigned short* __restrict__ pw, int64_t positions, int64_t batch_pad, int nslabs,
                                                  u64* __restrict__ hist, int nbins, unsigned int* redo_count,
                                                  unsigned int* __restrict__ redo_list, unsigned int sample0, unsigned int* bins,
                                                  unsigned int* staging, unsigned int* shared2) {
    unsigned int& redo_n = shared2[0];
    unsigned int& redo_base = shared2[1];
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) bins[i] = 0;
    if (threadIdx.x == 0) redo_n = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; s < positions; s += stride) {
        u64 four[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) four[k] = *reinterpret_cast<const u64*>(pw + (int64_t)(k < nslabs ? k : 0) * batch_pad + s);
        unsigned int w[4] = {0, 0, 0, 0};
        bool skip[4], stray[4];
        unsigned int local[4] = {0, 0, 0, 0};                      // tile-local sample index of a flagged position
#pragma unroll
        for (int t = 0; t < 4; ++t) skip[t] = stray[t] = false;   // finished and out-of-batch records carry REC_FLAG
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < nslabs) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const unsigned int v = (unsigned int)(four[k] >> (16 * t)) & 0xFFFFu;
                    skip[t] |= v == REC_FLAG;
                    if ((v & 0xFFC0u) == REC_STRAY) {
                        stray[t] = true;
                        local[t] = v & 63u;
                    }
                    w[t] += v;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (skip[t]) continue;
            if (stray[t]) {
                const unsigned int sample = sample0 + (unsigned int)((s + t) & ~(int64_t)63) + local[t];
                const unsigned int at = atomicAdd(&redo_n, 1u);
                if (at < 1024)
                    staging[at] = sample;
                else
                    redo_list[atomicAdd(redo_count, 1u)] = sample;                       // more than the staging area holds
            } else
                atomicAdd(&bins[w[t]], 1u);
        }
    }
    __syncthreads();
    if (hist)
        for (int i = threadIdx.x; i < nbins; i += blockDim.x)
            if (bins[i]) atomicAdd(&hist[i], (u64)bins[i]);
    const unsigned int mine = redo_n < 1024 ? redo_n : 1024;
    if (mine) {                                                     // uniform
        if (threadIdx.x == 0) redo_base = atomicAdd(redo_count, mine);
        __syncthreads();
        for (unsigned int i = threadIdx.x; i < mine; i += blockDim.x) redo_list[redo_base + i] = staging[i];
    }
    __syncthreads();
}

// A sub-pass scans 8 samples.  Only the words that hold non-identity columns matter (half of them in a standard form), so
// the (sample, word) pairs of a sub-pass are dealt out densely: pair p = t * 64 + lane of round t is word
// wlist[p % NW] of sample p / NW, NW = number of such words.  T = ceil(8 * NW / 64) rounds instead of 8 (4 for
// H = [I | A] at n = 4096).  Lanes beyond the last pair get mask 0 and a harmless address, so no load is predicated.
template <int T>
struct PairMap {
    unsigned int jw[T];        // (sample in sub-pass) << 6 | word
    unsigned int off[T];       // byte offset of that word from the sub-pass' first row
    u64 mask[T];               // the word's non-identity columns
};

// Requests the words of the sub-pass that starts at row s_first (T loads, whatever the path: the wait counts in the kernel
// rely on it).  Inline assembly: the compiler's wait insertion does not see these loads, the kernel waits by hand.
template <int T>
__device__ __forceinline__ void load_pairs(const CompactArgs& a, int64_t s_first, const PairMap<T>& pm, u64 (&w)[T]) {
    if (s_first + CMP_SUB <= a.batch) {                             // uniform: all 8 rows exist, one scalar base for the loads
        const u64* base = a.e + s_first * a.lde;
#pragma unroll
        for (int t = 0; t < T; ++t)
            asm volatile("global_load_dwordx2 %0, %1, %2" GF2_ROWS_POLICY : "=v"(w[t]) : "v"(pm.off[t]), "s"(base) : "memory");
    } else {
        // rows past the batch are clamped into it (the caller masks); offsets from the sub-pass' first row (itself clamped), so
        // that they stay small whatever the size of a pass (the lane offset is an unsigned 32-bit addend of the scalar base)
        const int64_t last = a.batch - 1;
        const int64_t first = s_first < last ? s_first : last;
        const u64* base = a.e + first * a.lde;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            int64_t sample = s_first + (int64_t)(pm.jw[t] >> 6);
            if (sample > last) sample = last;
            const unsigned int off = (unsigned int)(sample - first) * (unsigned int)a.lde * 8u + (pm.jw[t] & 63u) * 8u;
            asm volatile("global_load_dwordx2 %0, %1, %2" GF2_ROWS_POLICY : "=v"(w[t]) : "v"(off), "s"(base) : "memory");
        }
    }
}

// Waits until at most N vector memory operations are outstanding and makes `w` depend on the wait.
template <int N, int T>
__device__ __forceinline__ void wait_pairs(u64 (&w)[T]) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#pragma unroll
    for (int t = 0; t < T; ++t) asm volatile("" : "+v"(w[t]));
}

// FULL: every (sample, word) pair of a sub-pass is live and every scanned word consists of non-identity columns only (H = [I | A]
// and [A' | I | c] at n = 4096): no mask is applied (and none kept in registers).
template <int T, bool FULL>
__global__ __launch_bounds__(CMP_THREADS, T <= 4 ? 5 : (T <= 5 ? 4 : 3)) void slab_compact_kernel(CompactArgs a) {
    __shared__ CompactWaveLds lds_all[CMP_WAVES];
    __shared__ unsigned int wlist[64];
    __shared__ u64 wmask[64];
    __shared__ unsigned int nw_shared;
    if (a.clk && threadIdx.x == 0) atomicMin(&a.clk[0], (u64)wall_clock64());
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    CompactWaveLds& L = lds_all[wave];

    const int words = (a.n + 63) >> 6;
    u64 amask = 0;                                     // the non-identity columns of word `lane`
    if (lane < words) {
        amask = ~ident_mask(a.ident_off, a.r, lane);
        if (lane == words - 1 && (a.n & 63)) amask &= ~(~0ull << (a.n & 63));
        if ((a.skip_words >> lane) & 1ull) amask = 0;
    }
    if (a.cmb_pw) {
        // the previous pass' combine step first (nothing of this pass is in flight yet); its counters use the waves' LDS
        static_assert(sizeof(lds_all) >= (SLAB_MAX_BINS + 1024 + 2) * 4, "the combine step's counters and staging area fit the compact kernel's LDS");
        unsigned int* const scratch = reinterpret_cast<unsigned int*>(lds_all);
        combine_positions(a.cmb_pw, a.cmb_positions, a.cmb_pad, a.cmb_nslabs, a.hist, a.cmb_nbins, a.redo_count, a.redo_list,
                          a.cmb_sample0, scratch, scratch + SLAB_MAX_BINS, scratch + SLAB_MAX_BINS + 1024);
    }
    if (wave == 0) {
        const u64 act = __ballot(amask != 0);
        if (amask) wlist[__builtin_amdgcn_mbcnt_hi((unsigned int)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)act, 0u))] = lane;
        wmask[lane] = amask;
        if (lane == 0) nw_shared = (unsigned int)__popcll(act);
    }
    __syncthreads();
    const unsigned int nw = nw_shared;
    PairMap<T> pm;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const unsigned int p = t * 64 + lane;
        const bool live = nw && p < CMP_SUB * nw;
        const unsigned int j = live ? p / nw : 0u, wi = live ? wlist[p - j * nw] : 0u;
        pm.jw[t] = (j << 6) | wi;
        pm.off[t] = (j * (unsigned int)a.lde + wi) * 8u;
        pm.mask[t] = live ? wmask[wi] : 0ull;
    }
    const unsigned int null_ent = (unsigned int)a.null_ord << 4;        // entries are LDS byte offsets inside a row-part plane
    const unsigned int null2 = (unsigned int)__builtin_amdgcn_readfirstlane((int)(null_ent | (null_ent << 16)));   // stays scalar
    const SparseSide side = {a.ht, (int64_t)a.r, (int64_t)a.ident_off, nullptr, 0};
    // ordinal of column c: c below the identity block, c - r above it.  A word holds columns of one side only unless the
    // block is narrower than a word (`mixed`); otherwise the word's base carries the subtraction (16-bit wrap-around is fine).
    const bool mixed = a.ident_off >= 0 && a.r < 64;                    // uniform
    auto col_base16 = [&](unsigned int col0, unsigned int first_bit) -> unsigned int {
        if (mixed) return col0 << 4;
        return ((a.ident_off >= 0 && (int)(col0 + first_bit) >= a.ident_off) ? col0 - (unsigned int)a.r : col0) << 4;
    };
    auto ord16 = [&](unsigned int cb, unsigned int bit) -> unsigned int {
        if (!mixed) return cb + (bit << 4);
        const unsigned int col = (cb >> 4) + bit;
        return ((int)col >= a.ident_off ? col - (unsigned int)a.r : col) << 4;
    };
    unsigned int head = 0, tail = 0;                                    // ring positions (uniform, running)
    // One pass over `count` <= 64 ring entries starting at `first`: a word that comes for the first time reserves the slots of all
    // its columns with one LDS atomic, every word writes its lowest column and, if it has more, goes back to the tail.
    auto process = [&](unsigned int first, unsigned int count) {
        bool more = false;
        u64 v = 0;
        unsigned int meta = 0;
        if ((unsigned int)lane < count) {
            const unsigned int at = (first + lane) & (CMP_RING - 1);
            v = L.ring_v[at];
            const unsigned int m = L.ring_m[at];
            const unsigned int j = (m >> 6) & 63u, col0 = (m & 63u) << 6;
            const bool fresh = (m & CMP_CONT) == 0;
            const unsigned int reserved = atomicAdd(&L.cnt[j], fresh ? (unsigned int)__popcll(v) : 0u);
            const unsigned int slot = fresh ? reserved : (m >> 12) & 63u;
            const unsigned int bit = (unsigned int)(__ffsll((long long)v) - 1);
            const unsigned int cb = col_base16(col0, bit);
            unsigned char* const rj = L.rec + j * 64;
            *reinterpret_cast<unsigned short*>(rj + (slot < REC_SLOTS - 1 ? slot * 2 : 62u)) = (unsigned short)ord16(cb, bit);
            v &= v - 1;
            more = v != 0;
            meta = (m & 0xFFFu) | ((slot + 1 < REC_SLOTS - 1 ? slot + 1 : (unsigned int)(REC_SLOTS - 1)) << 12) | CMP_CONT;
        }
        const u64 mact = __ballot(more);
        if (more) {
            const unsigned int q = (tail + __builtin_amdgcn_mbcnt_hi((unsigned int)(mact >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mact, 0u))) &
                                   (CMP_RING - 1);
            L.ring_v[q] = v;
            L.ring_m[q] = meta;
        }
        tail += (unsigned int)__popcll(mact);
    };

    const int64_t ntiles = (a.batch + 63) >> 6;
    const int64_t total_waves = (int64_t)gridDim.x * CMP_WAVES;
    // The words of two sub-passes are in flight while a third is compacted: two register buffers take turns (the sub-pass loop
    // is unrolled by two), each refilled for the sub-pass after next as soon as its words have been masked.  Vector memory
    // operations complete in order, so a sub-pass waits until all but the other buffer's T loads (and, in the first two
    // sub-passes of a tile, the previous tile's four record stores) have completed.
    u64 wa[T], wb[T];
    int64_t tile = (int64_t)blockIdx.x * CMP_WAVES + wave;
    load_pairs<T>(a, tile * 64, pm, wa);
    load_pairs<T>(a, tile * 64 + CMP_SUB, pm, wb);
    wait_pairs<0, T>(wa);
    wait_pairs<0, T>(wb);
    auto sub_pass = [&](int64_t s0, int sub, u64 (&wbuf)[T], int64_t next_tile) {
        if (sub < 2)
            wait_pairs<T + 4, T>(wbuf);
        else
            wait_pairs<T, T>(wbuf);
        u64 w[T];
        if (s0 + (sub + 1) * CMP_SUB <= a.batch) {                  // uniform
#pragma unroll
            for (int t = 0; t < T; ++t) w[t] = FULL ? wbuf[t] : wbuf[t] & pm.mask[t];
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t) w[t] = s0 + sub * CMP_SUB + (pm.jw[t] >> 6) < a.batch ? wbuf[t] & pm.mask[t] : 0ull;
        }
        load_pairs<T>(a, sub + 2 < 64 / CMP_SUB ? s0 + (sub + 2) * CMP_SUB : next_tile * 64 + (sub + 2 - 64 / CMP_SUB) * CMP_SUB, pm,
                      wbuf);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const bool nz = w[t] != 0;
            const u64 act = __ballot(nz);
            if (nz) {
                const unsigned int pos = (tail + __builtin_amdgcn_mbcnt_hi((unsigned int)(act >> 32),
                                                                             __builtin_amdgcn_mbcnt_lo((unsigned int)act, 0u))) &
                                         (CMP_RING - 1);
                L.ring_v[pos] = w[t];
                L.ring_m[pos] = pm.jw[t] + ((unsigned int)(sub * CMP_SUB) << 6);
            }
            tail += (unsigned int)__popcll(act);
            while (tail - head >= 64) {                                 // uniform; at most 63 entries stay behind (a pass adds fewer than it takes)
                wave_lds_sync();
                process(head, 64);
                head += 64;
            }
        }
    };
#pragma unroll 1
    for (; tile < ntiles; tile += total_waves) {
        const int64_t s0 = tile * 64;
        {
            u32x4 null4;                            // made here every time: hoisted out of the loop it gets spilled, and a
            unsigned int n0, n1, n2, n3;            // scratch reload makes the compiler wait for every load in flight
            asm volatile("v_mov_b32 %0, %1" : "=v"(n0) : "s"(null2));
            asm volatile("v_mov_b32 %0, %1" : "=v"(n1) : "s"(null2));
            asm volatile("v_mov_b32 %0, %1" : "=v"(n2) : "s"(null2));
            asm volatile("v_mov_b32 %0, %1" : "=v"(n3) : "s"(null2));
            null4.x = n0, null4.y = n1, null4.z = n2, null4.w = n3;
#pragma unroll
            for (int q = 0; q < 4; ++q) reinterpret_cast<u32x4*>(L.rec)[lane * 4 + q] = null4;
        }
        L.cnt[lane] = 0;
#pragma unroll 1
        for (int sub = 0; sub < 64 / CMP_SUB; sub += 2) {
            sub_pass(s0, sub, wa, tile + total_waves);
            sub_pass(s0, sub + 1, wb, tile + total_waves);
        }
        while (tail != head) {                                          // uniform: what is left, and what that leaves
            wave_lds_sync();
            const unsigned int count = tail - head < 64 ? tail - head : 64;
            process(head, count);
            head += count;
        }
        wave_lds_sync();

        // lane = sample from here on
        const unsigned int c = L.cnt[lane];
        const bool over = c >= REC_SLOTS;
        // Records leave the tile sorted by count (counting sort over the 64 samples): the gather kernel walks 16 records
        // per step up to the largest count among them, and the lower half of the tile is stored as 32-byte records.
        // Slot 0 = count (REC_OVER: finished by the slow routine) | tile-local sample << 8.
        const unsigned int key = over ? 0u : c;
        unsigned int* const sort_bins = reinterpret_cast<unsigned int*>(&L);          // 32 bins + 32 offsets, item list is free
        if (lane < 32) sort_bins[lane] = 0;
        wave_lds_sync();
        const unsigned int in_bucket = atomicAdd(&sort_bins[key], 1u);
        wave_lds_sync();
        {
            unsigned int run = lane < 32 ? sort_bins[lane] : 0u;                      // inclusive scan over the bins
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) {
                const unsigned int up = __shfl_up(run, d);
                if (lane >= d) run += up;
            }
            if (lane < 32) sort_bins[32 + lane] = run - sort_bins[lane];
        }
        wave_lds_sync();
        const unsigned int rank = sort_bins[32 + key] + in_bucket;
        wave_lds_sync();
        // finished here and now: more columns than a record holds, or too many for the 32-byte record its rank gives it
        const bool misfit = over || (rank < 32 && c >= SHORT_SLOTS);
        {
            u64 todo = __ballot(misfit && s0 + lane < a.batch);
            unsigned int* const slow_list = reinterpret_cast<unsigned int*>(&L);    // the item list is free now
            while (todo) {
                const int j = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const u64 ww = lane < words ? a.e[(s0 + j) * a.lde + lane] : 0ull;
                unsigned int sdw;
                const unsigned int wt = sparse_component_weight(ww, side, a.n, lane, slow_list, &sdw);
                if (lane == 0 && a.hist) atomicAdd(&a.hist[wt], 1ull);
                if (a.syn && lane < ((a.r + 63) >> 6) * 2) a.syn[(s0 + j) * a.lds32 + lane] = sdw;       // (the gather kernel skips finished samples)
            }
            wave_lds_sync();
        }
        u32x4 R[4];                                                 // the records themselves were not touched by the sort or the slow routine
#pragma unroll
        for (int q = 0; q < 4; ++q) R[q] = reinterpret_cast<const u32x4*>(L.rec)[lane * 4 + q];
        // a misfit of the lower half keeps its place and its (truncated) count, so that the last record of its step still
        // carries the step's largest count; the gather kernel does its lookups and throws the result away
        {
            const unsigned int header = ((over ? REC_OVER : (misfit ? REC_DONE | (SHORT_SLOTS - 1) : c)) | ((unsigned int)lane << 8)) << 16;
            if (rank < 32)
                R[1].w = (R[1].w & 0xFFFFu) | header;               // slot 15 of a 32-byte record
            else
                R[3].w = (R[3].w & 0xFFFFu) | header;               // slot 31
        }
        // four store instructions per tile whatever the data (the wait counts above): two for every record, two more for the long ones
        char* const tile_out = reinterpret_cast<char*>(a.rec) + tile * TILE_REC_BYTES;
        u32x4* const out = reinterpret_cast<u32x4*>(tile_out + (rank < 32 ? rank * 32u : 1024u + (rank - 32u) * 64u));
        out[0] = R[0];
        out[1] = R[1];
        if (rank >= 32) {
            out[2] = R[2];
            out[3] = R[3];
        }
        wave_lds_sync();
    }
    if (a.clk && lane == 0) atomicMax(&a.clk[1], (u64)wall_clock64());
}

// ---- records straight from the sampler (gf2_mc_run) ---------------------------------------------------------------------------
//
// The Monte-Carlo run draws its errors itself, and what the gather kernel wants of a sample is not its packed row but (1) the
// record of its non-identity columns and (2) the words under the identity block.  This kernel makes exactly those from the
// sampler's definition (gf2_sampler.h): no row is written to be read back and taken apart by the compact kernel (1 KiB written
// and 0.5 KiB read per sample and two compact launches per pass less).
// Wavefront = tile of 64 samples, lane = SAMPLE, for all its segments in turn: a lane takes the segment's draw and count, then
// its erroneous qubits one after the other -- a mix64, Floyd's rule through one returning LDS atomic on a 512-bit map of its
// own, and per component either a bit in the lane's image of the identity words or a 16-bit slot appended to its record (the
// slot count is a register: nobody else writes this record).  The lanes of a wavefront do nearly the same amount of work (41 +- 6
// erroneous qubits at n = 4096, p = 0.01), so there is no flat re-layout to pay for.  After each segment the identity words
// leave as 64-byte pieces; after the last, the tile's records are sorted by count and stored as the compact kernel stores them.
// A sample whose record overflows (or that misses the short half of its tile) goes on a list: slab_misfit_kernel draws it
// again, as a packed row, and computes its weight with the wavefront-per-sample routine.
struct RecSide {
    u32x4* rec;                  // the tile records (TILE_REC_BYTES per tile)
    u64* eident;                 // error rows of lde words of which only the words under the identity block are written
    unsigned int* misfit_count;  // (this chunk's slot of the two)
    unsigned int* misfit_list;   // sample index inside the pass
    int ident_off, r, null_ord;
    int seg_lo, seg_hi;          // segments whose words the gather kernel may read
};
struct RecSamplerArgs {
    u64 seed;
    int64_t first_sample, count, lde;
    SegTables th;
    RecSide side[2];             // [0]: X component (e_x, against H2), [1]: Z component (e_z, against H1)
    int n;
    int cap;                     // erroneous qubits of a segment a lane takes in step with the others (even, 2 .. RS_CAP_MAX; 0: all)
};
#define RS_STRIDE 17             // dwords per lane of a 512-bit map: 16 + 1 (odd: the lanes' dwords spread over the banks)
#define RS_TAILS 64
#define RS_CAP_MAX 8
struct alignas(16) RecTail {     // 32 bytes
    u64 d;                       // the segment's draw
    unsigned int meta;           // lane | segment << 6 | K << 10
    unsigned int pad;
    unsigned short pos[RS_CAP_MAX];   // the positions taken so far
};
struct alignas(16) RecWaveLds {
    unsigned char rec[2][4096 + 16];
    unsigned int taken[64 * RS_STRIDE];
    unsigned int img[64 * RS_STRIDE];     // identity words of the one component that has any in the current segment (gf2_mc_records_ok)
    unsigned int bins[64];
    unsigned short sink[64];              // where a lane's record store goes when there is nothing to store
    RecTail tails[RS_TAILS];              // (sample, segment) pairs with qubits left over (see the kernel)
    unsigned int early[64 * RS_CAP_MAX / 2];   // a lane's first positions of the current segment, two per dword; while the leftovers
                                               // are worked off: the lanes' slot counts (2 x 64)
};
static_assert(sizeof(RecWaveLds) <= 160 * 1024 / 8, "eight record-sampler wavefronts per CU");

__global__ __launch_bounds__(64) void slab_record_sampler_kernel(RecSamplerArgs a) {
    __shared__ RecWaveLds L;
    const int lane = threadIdx.x;
    const int64_t ntiles = (a.count + 63) >> 6;
    unsigned int* const my_taken = L.taken + lane * RS_STRIDE;
    // A lane's record is 64 bytes = a quarter of the banks, and the lanes' counts run nearly in step: sixteen lanes would store
    // to one bank.  So lane l keeps its record turned by l / 4 dwords (slot k at slot (k + 2 (l / 4)) mod 32 of its 64 bytes):
    // equal slots of the 64 lanes are 64 banks.  Turned back when the record leaves.
    const unsigned int turn = (unsigned int)(lane >> 2) * 2u;
    u64 cdf16[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) cdf16[k] = a.th.cdf[k];
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t i = tile * 64 + lane;
        const bool live = i < a.count;
        const u64 ks = sample_key(a.seed, (u64)(a.first_sample + i));
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const unsigned int null_ent = (unsigned int)a.side[c].null_ord << 4;
            const unsigned int n2 = null_ent | (null_ent << 16);
            const u32x4 null4 = {n2, n2, n2, n2};
#pragma unroll
            for (int q = 0; q < 4; ++q) reinterpret_cast<u32x4*>(L.rec[c])[lane * 4 + q] = null4;
        }
        unsigned int cnt[2] = {0, 0};
        // Leftovers.  The lanes walk a segment in step, as many trips as the LARGEST count among the 64 samples asks for (11 or 12
        // qubits where the mean is 5): most of the later trips' lanes idle.  So the walk stops after a.cap qubits, and a lane with
        // more leaves (draw, count, the positions it has taken) in L.tails.  At the end of the tile -- or when 64 pairs are
        // waiting -- the pairs are worked off one per LANE: the lane marks the positions again in a map of its own, goes on with
        // Floyd's rule where the sample's lane stopped, and puts what it finds where that lane would have: record slots through a
        // counter in LDS, identity bits by an atomic OR on the words the segment's flush has stored.
        unsigned int ntails = 0;                                    // uniform
        auto run_tails = [&]() {
            unsigned int* const counts = L.early;
            wave_lds_sync();
            counts[lane] = cnt[0];
            counts[64 + lane] = cnt[1];
#pragma unroll
            for (int q = 0; q < RS_STRIDE; ++q) L.taken[q * 64 + lane] = 0;
            wave_lds_sync();
            const bool act = (unsigned int)lane < ntails;
            u64 td = 0;
            unsigned int meta = 0;
            u32x4 early4 = {0, 0, 0, 0};
            if (act) {
                td = L.tails[lane].d;
                meta = L.tails[lane].meta;
                early4 = *reinterpret_cast<const u32x4*>(L.tails[lane].pos);
            }
            const unsigned int owner = meta & 63u;
            const int ts = (int)((meta >> 6) & 15u), tK = (int)(meta >> 10);
            const int tbase = ts * GF2_SEG_BITS, tnb = ts == a.th.nseg - 1 ? a.th.nb_last : GF2_SEG_BITS;
            const unsigned int owner_turn = (owner >> 2) * 2u;
            const int64_t owner_row = (tile * 64 + (int64_t)owner) * a.lde;
            const unsigned int early_dw[4] = {early4.x, early4.y, early4.z, early4.w};
#pragma unroll
            for (int q = 0; q < RS_CAP_MAX; ++q) {
                if (q < a.cap) {                                    // uniform
                    const unsigned int p = (early_dw[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
                    atomicOr(&my_taken[p >> 5], act ? 1u << (p & 31u) : 0u);
                }
            }
            const unsigned int t1_lo = (unsigned int)a.th.t_1, t2_lo = (unsigned int)a.th.t_2;
            const bool t1_top = (a.th.t_1 >> 32) != 0, t2_top = (a.th.t_2 >> 32) != 0;
            auto put = [&](bool on, unsigned int pos, unsigned int kind) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const RecSide& sd = a.side[c];
                    if (!(on && ((kind >> c) & 1u))) continue;
                    const int rel = (int)pos - (sd.ident_off - tbase);
                    if ((unsigned int)rel < (unsigned int)sd.r) {
                        unsigned int* const row = reinterpret_cast<unsigned int*>(sd.eident + owner_row);
                        atomicOr(row + (((unsigned int)tbase + pos) >> 5), 1u << (pos & 31u));
                    } else {
                        const unsigned int slot = atomicAdd(&counts[c * 64 + owner], 1u);
                        const unsigned int ord = (unsigned int)((int)pos + (rel < 0 ? tbase : tbase - sd.r)) << 4;
                        if (slot + 1u < REC_SLOTS)
                            *reinterpret_cast<unsigned short*>(L.rec[c] + owner * 64 + ((slot + owner_turn) & 31u) * 2) = (unsigned short)ord;
                    }
                }
            };
            u64 x = td + GF2_GOLDEN * (u64)(a.cap + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the flushes' stores are done: the ORs below meet their words
            for (int k = a.cap; __ballot(k < tK) != 0; ++k, x += GF2_GOLDEN) {
                const bool on = k < tK;
                const u64 v = mix64(x);
                const unsigned int j = (unsigned int)(tnb - tK + k);
                unsigned int t = __umulhi((unsigned int)(v >> 32), j + 1u);
                const unsigned int c32 = (unsigned int)v;
                const unsigned int kind = ((t2_top || c32 < t2_lo) ? 1u : 0u) | ((!t1_top && c32 >= t1_lo) ? 2u : 0u);
                if (!on) t = 0;
                const unsigned int old = atomicOr(&my_taken[t >> 5], on ? 1u << (t & 31u) : 0u);
                const bool hit = on && ((old >> (t & 31u)) & 1u);
                const unsigned int jj = j & 511u;
                if (hit) atomicOr(&my_taken[jj >> 5], 1u << (jj & 31u));
                put(on, hit ? jj : t, kind);
            }
            wave_lds_sync();
            cnt[0] = counts[lane];
            cnt[1] = counts[64 + lane];
            wave_lds_sync();
            ntails = 0;
        };
#pragma unroll 1
        for (int s = 0; s < a.th.nseg; ++s) {
            const bool last = s == a.th.nseg - 1;
            const int nb = last ? a.th.nb_last : GF2_SEG_BITS;
            const int base = s * GF2_SEG_BITS;
            const bool flush0 = s >= a.side[0].seg_lo && s < a.side[0].seg_hi, flush1 = s >= a.side[1].seg_lo && s < a.side[1].seg_hi;
            {
                // the maps, 16 bytes per lane and store: L.taken and, behind it, L.img (one uniform branch; written as a loop over both
                // maps' dwords this was 34 stores and as many branches)
                static_assert(offsetof(RecWaveLds, taken) % 16 == 0 && offsetof(RecWaveLds, img) == offsetof(RecWaveLds, taken) + sizeof(L.taken) &&
                                  sizeof(L.taken) == 272 * 16,
                              "the two maps are one run of 544 16-byte pieces");
                u32x4* const maps = reinterpret_cast<u32x4*>(L.taken);
                const u32x4 zero4 = {0, 0, 0, 0};
                if (flush0 || flush1) {                                     // uniform
#pragma unroll
                    for (int q = 0; q < 8; ++q) maps[q * 64 + lane] = zero4;
                    if (lane < 32) maps[512 + lane] = zero4;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) maps[q * 64 + lane] = zero4;
                    if (lane < 16) maps[256 + lane] = zero4;
                }
            }
            const u64 d = live ? segment_draw(ks, (u64)s) : 0;
            int K = 0;
            {
                // the count: how many of the table's thresholds the draw reaches.  The whole segments' first 16 thresholds are in
                // registers since the kernel began (every segment used to wait for eight 16-byte loads of them here)
                const u64 u = d >> 32;
                if (!last) {                                                // uniform
#pragma unroll
                    for (int k = 0; k < 16; ++k) K += u >= cdf16[k] ? 1 : 0;
                } else {
                    const u64* const tab = a.th.cdf + GF2_SEG_CDF;
#pragma unroll
                    for (int k = 0; k < 16; ++k) K += u >= tab[k] ? 1 : 0;
                }
                if (K == 16) {
                    const u64* const tab = a.th.cdf + (last ? GF2_SEG_CDF : 0);
                    while (K < nb && u >= tab[K]) K += 1;
                }
                if (!live) K = 0;
            }
            __builtin_amdgcn_wave_barrier();
            // What this segment holds of each component's check (uniform): columns under the identity block (lo <= position < lo + r,
            // bits of the identity image) and / or other columns (slots of the record).  In a CSS code's standard forms a segment is
            // one or the other, except the last one of H2 = [A' | I | c].
            int lo[2];
            bool has_id[2], has_rec[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                lo[c] = a.side[c].ident_off - base;
                has_id[c] = lo[c] < nb && lo[c] + a.side[c].r > 0;
                has_rec[c] = lo[c] > 0 || lo[c] + a.side[c].r < nb;
            }
            // The segment's erroneous qubits, MASK = what the segment holds of the two checks as compile-time constants (bit 0 / 1:
            // identity columns / other columns of component 0, bits 2 / 3: of component 1), so that the walk below has no uniform
            // branches in it (round 3: a CSS code's segments are "records for one component, identity bits for the other").
            auto walk = [&](auto mask_constant) {
                constexpr int MASK = decltype(mask_constant)::value;
                constexpr bool HID[2] = {(MASK & 1) != 0, (MASK & 4) != 0}, HREC[2] = {(MASK & 2) != 0, (MASK & 8) != 0};
                // One erroneous qubit into the outputs, without a divergent branch: an identity column is a bit in the image (an
                // LDS OR of zero otherwise), any other column a 16-bit slot at the end of the lane's record (stored to the lane's
                // sink otherwise).
                auto emit = [&](bool on, unsigned int pos, unsigned int kind) {
                    unsigned int id_bit = 0;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const RecSide& sd = a.side[c];
                        const bool mine = on && ((kind >> c) & 1u);
                        const bool under = (unsigned int)((int)pos - lo[c]) < (unsigned int)sd.r;
                        if (HID[c]) {
                            if (mine && (!HREC[c] || under)) id_bit = 1u << (pos & 31u);
                        }
                        if (HREC[c]) {
                            const bool is_rec = mine && (!HID[c] || !under);
                            cnt[c] += is_rec ? 1u : 0u;
                            const unsigned int ord = (unsigned int)((int)pos + ((int)pos < lo[c] ? base : base - sd.r)) << 4;
                            unsigned short* const at = (is_rec && cnt[c] < REC_SLOTS)
                                                           ? reinterpret_cast<unsigned short*>(L.rec[c] + lane * 64 + ((cnt[c] - 1u + turn) & 31u) * 2)
                                                           : &L.sink[lane];
                            *at = (unsigned short)ord;
                        }
                    }
                    if (HID[0] || HID[1]) atomicOr(&L.img[lane * RS_STRIDE + (pos >> 5)], id_bit);
                };
                // two erroneous qubits per trip: their draws are independent chains of multiplies, and the second one's atomic is
                // issued behind the first one's without waiting for it (the LDS serves a wavefront's operations in order)
                // (error_draw of gf2_sampler.h with its argument d + G (k + 1) carried along -- two additions per trip instead of
                // two 64-bit multiplies -- and the kind thresholds, which are 2^32 at most, compared as 32-bit numbers)
                u64 x0 = d + GF2_GOLDEN, x1 = x0 + GF2_GOLDEN;
                const unsigned int t1_lo = (unsigned int)a.th.t_1, t2_lo = (unsigned int)a.th.t_2;
                const bool t1_top = (a.th.t_1 >> 32) != 0, t2_top = (a.th.t_2 >> 32) != 0;       // threshold = 2^32: every c is below it
                auto draw = [&](u64 x, int k, unsigned int* t_out, unsigned int* kind) {
                    const u64 v = mix64(x);
                    const unsigned int j = (unsigned int)(nb - K + k);
                    *t_out = __umulhi((unsigned int)(v >> 32), j + 1u);
                    const unsigned int c = (unsigned int)v;
                    *kind = ((t2_top || c < t2_lo) ? 1u : 0u) | ((!t1_top && c >= t1_lo) ? 2u : 0u);
                };
                const int stop = a.cap ? a.cap : GF2_SEG_BITS;           // (even)
                for (int k = 0; __ballot(k < K) != 0 && k < stop; k += 2, x0 += 2 * GF2_GOLDEN, x1 += 2 * GF2_GOLDEN) {
                    const bool on0 = k < K, on1 = k + 1 < K;
                    unsigned int t0, kind0, t1, kind1;
                    draw(x0, k, &t0, &kind0);
                    draw(x1, k + 1, &t1, &kind1);
                    if (!on0) t0 = 0;
                    if (!on1) t1 = 0;
                    const unsigned int old0 = atomicOr(&my_taken[t0 >> 5], on0 ? 1u << (t0 & 31u) : 0u);
                    const unsigned int old1 = atomicOr(&my_taken[t1 >> 5], on1 ? 1u << (t1 & 31u) : 0u);
                    const unsigned int j0 = (unsigned int)(nb - K + k) & 511u, j1 = (j0 + 1u) & 511u;
                    // Floyd: a position taken already gives way to j (never taken before); the second qubit's candidate may be the
                    // j the first one has just moved to
                    const bool hit0 = on0 && ((old0 >> (t0 & 31u)) & 1u);
                    const bool hit1 = on1 && (((old1 >> (t1 & 31u)) & 1u) || (hit0 && t1 == j0));
                    const unsigned int pos0 = hit0 ? j0 : t0, pos1 = hit1 ? j1 : t1;
                    atomicOr(&my_taken[j0 >> 5], hit0 ? 1u << (j0 & 31u) : 0u);
                    atomicOr(&my_taken[j1 >> 5], hit1 ? 1u << (j1 & 31u) : 0u);
                    L.early[lane * (RS_CAP_MAX / 2) + ((k >> 1) & (RS_CAP_MAX / 2 - 1))] = pos0 | (pos1 << 16);     // (no branch in the trip)
                    emit(on0, pos0, kind0);
                    emit(on1, pos1, kind1);
                }
            };
            switch ((has_id[0] ? 1 : 0) | (has_rec[0] ? 2 : 0) | (has_id[1] ? 4 : 0) | (has_rec[1] ? 8 : 0)) {        // uniform
                case 6: walk(std::integral_constant<int, 6>{}); break;       // records for component 0, identity bits for 1
                case 9: walk(std::integral_constant<int, 9>{}); break;       // the other way round
                case 10: walk(std::integral_constant<int, 10>{}); break;     // records for both
                case 5: walk(std::integral_constant<int, 5>{}); break;       // identity bits for both
                case 7: walk(std::integral_constant<int, 7>{}); break;
                case 11: walk(std::integral_constant<int, 11>{}); break;
                case 13: walk(std::integral_constant<int, 13>{}); break;
                case 14: walk(std::integral_constant<int, 14>{}); break;
                default: walk(std::integral_constant<int, 15>{}); break;
            }
            __builtin_amdgcn_wave_barrier();
            // the segment's identity words out: four lanes per sample, 16 bytes each, sixteen samples per store instruction
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const RecSide& sd = a.side[c];
                if (s < sd.seg_lo || s >= sd.seg_hi) continue;             // uniform
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int smp = it * 16 + (lane >> 2), dq = (lane & 3) * 4;
                    const unsigned int* src = L.img + smp * RS_STRIDE + dq;
                    const u32x4 v = {src[0], src[1], src[2], src[3]};
                    const int64_t word = (int64_t)s * GF2_SEG_WORDS + (dq >> 1);
                    if (tile * 64 + smp < a.count && word + 2 <= a.lde)
                        *reinterpret_cast<u32x4*>(sd.eident + (tile * 64 + smp) * a.lde + word) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (a.cap) {                                              // uniform
                const bool left = K > a.cap;
                const u64 lm = __ballot(left);
                if (lm) {                                             // uniform
                    u32x4 early4 = {0, 0, 0, 0};
                    if (left) early4 = *reinterpret_cast<const u32x4*>(L.early + lane * (RS_CAP_MAX / 2));
                    const unsigned int more = (unsigned int)__popcll(lm);
                    if (ntails + more > RS_TAILS) run_tails();       // uniform (this segment's flush is behind us as well)
                    if (left) {
                        RecTail& T = L.tails[ntails + __builtin_amdgcn_mbcnt_hi((unsigned int)(lm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)lm, 0u))];
                        T.d = d;
                        T.meta = (unsigned int)lane | ((unsigned int)s << 6) | ((unsigned int)K << 10);
                        *reinterpret_cast<u32x4*>(T.pos) = early4;
                    }
                    ntails += more;
                }
            }
        }
        if (ntails) run_tails();                                      // uniform
        // records out, per component, as the compact kernel leaves them: sorted by count, the 32 shortest as 32-byte records
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const RecSide& sd = a.side[c];
            const unsigned int cc = cnt[c];
            const bool over = cc >= REC_SLOTS;
            const unsigned int key = over ? 0u : cc;
            if (lane < 32) L.bins[lane] = 0;
            wave_lds_sync();
            const unsigned int in_bucket = atomicAdd(&L.bins[key], 1u);
            wave_lds_sync();
            {
                unsigned int run = lane < 32 ? L.bins[lane] : 0u;
#pragma unroll
                for (int dd = 1; dd < 32; dd <<= 1) {
                    const unsigned int up = __shfl_up(run, dd);
                    if (lane >= dd) run += up;
                }
                if (lane < 32) L.bins[32 + lane] = run - L.bins[lane];
            }
            wave_lds_sync();
            const unsigned int rank = L.bins[32 + key] + in_bucket;
            wave_lds_sync();
            const bool misfit = over || (rank < 32 && cc >= SHORT_SLOTS);
            if (misfit && live) sd.misfit_list[atomicAdd(sd.misfit_count, 1u)] = (unsigned int)(tile * 64 + lane);
            u32x4 R[4];
            {
                const unsigned int* const mine = reinterpret_cast<const unsigned int*>(L.rec[c]) + lane * 16;
                unsigned int dw[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) dw[q] = mine[(q + (turn >> 1)) & 15];
#pragma unroll
                for (int q = 0; q < 4; ++q) R[q] = u32x4{dw[4 * q], dw[4 * q + 1], dw[4 * q + 2], dw[4 * q + 3]};
            }
            {
                const unsigned int header = ((over ? REC_OVER : (misfit ? REC_DONE | (SHORT_SLOTS - 1) : cc)) | ((unsigned int)lane << 8)) << 16;
                if (rank < 32)
                    R[1].w = (R[1].w & 0xFFFFu) | header;
                else
                    R[3].w = (R[3].w & 0xFFFFu) | header;
            }
            char* const tile_out = reinterpret_cast<char*>(sd.rec) + tile * TILE_REC_BYTES;
            u32x4* const out = reinterpret_cast<u32x4*>(tile_out + (rank < 32 ? rank * 32u : 1024u + (rank - 32u) * 64u));
            out[0] = R[0];
            out[1] = R[1];
            if (rank >= 32) {
                out[2] = R[2];
                out[3] = R[3];
            }
            wave_lds_sync();
        }
    }
}

// The listed samples once more, as packed rows of their component, and their weights by the wavefront-per-sample routine.  One
// launch for both components (the first half of the workgroups: component 0).  The lists' counters come in two slots that the
// chunks of a run take in turn: this launch reads slot `parity` and clears the other one for the next chunk's sampler (nobody
// else touches it meanwhile) -- round 3 cleared them with two memsets per chunk and ran two launches.
#define MISFIT_WAVES 4
struct MisfitArgs {
    const unsigned int* count[2];    // per component: two counter slots
    const unsigned int* list[2];
    SparseSide side[2];
    unsigned int* clear[2];          // the other slot of each component
};
__global__ __launch_bounds__(64 * MISFIT_WAVES) void slab_misfit_kernel(MisfitArgs a, int parity, u64 seed, int64_t first_sample, SegTables th,
                                                                       int64_t n) {
    __shared__ unsigned int img[MISFIT_WAVES][8 * 33];
    __shared__ unsigned int cols[MISFIT_WAVES][SPARSE_LIST_CAP + 8];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = gridDim.x >> 1;
    const int comp = (int)blockIdx.x >= half ? 1 : 0;
    const int block = (int)blockIdx.x - comp * half;
    if (block == 0 && threadIdx.x == 0) *a.clear[comp] = 0;
    const unsigned int total = a.count[comp][parity];
    const unsigned int* const list = a.list[comp];
    const SparseSide side = a.side[comp];
    const int words = (int)((n + 63) >> 6);
    const bool last = lane == th.nseg - 1;
    for (unsigned int e = block * MISFIT_WAVES + wave; e < total; e += half * MISFIT_WAVES) {
        const u64 sample = (u64)(first_sample + (int64_t)list[e]);
        if (lane < th.nseg)
            sample_segment(sample_key(seed, sample), lane, last ? th.nb_last : GF2_SEG_BITS, th.cdf + (last ? GF2_SEG_CDF : 0), th.t_1,
                           th.t_2, img[wave] + lane * 33);
        __builtin_amdgcn_wave_barrier();
        u64 w = 0;
        if (lane < words) {
            const unsigned int* seg = img[wave] + (lane >> 3) * 33 + (comp ? 16 : 0) + (lane & 7) * 2;
            w = ((u64)seg[1] << 32) | seg[0];
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned int wt = sparse_component_weight(w, side, n, lane, cols[wave]);
        if (lane == 0) atomicAdd(&side.hist[wt], 1ull);
    }
}

// ---- gather ----------------------------------------------------------------------------------------------------------

struct GatherArgs {
    const u32x4* tab;          // nslabs x tab_cols x 4 vectors: entry (slab, ord) = rows 512 slab .. + 511 of column ord
    const u64* e;
    const u32x4* rec;
    unsigned short* pw;        // nslabs x batch_pad partial weights
    char* syn;                 // syndromes out (null: none): every workgroup stores its slab's 64-byte piece per sample, syn_pitch
    int64_t syn_pitch;         // bytes apart (round 4: before, a call that wanted syndromes went to the column-gather kernel)
    int syn_row;               // bytes of a syndrome: 8 ceil(r / 64); nothing is written past them
    char* syn_sink;            // 1 KiB that lanes with nothing to store write to (hand-scheduled kernel: every step issues its store)
    int64_t batch, batch_pad, lde;   // batch_pad: stride of the partial weights; records exist for ceil(batch / 64) * 64 positions
    int tab_stride, nslabs, r, ident_off, null_ord;   // tab_stride: entries per row-part plane (= 4 mod 16)
    int stray_n, stray_col[2];   // columns compact left out (hand-scheduled kernel only): a sample that has one is flagged REC_STRAY
    int reverse;               // hand-scheduled kernel: walk the records from the last tile to the first (see gf2_syndrome_slabs)
    int cross;                 // hand-scheduled kernel: a step takes ranks 4k .. 4k + 3 of FOUR tiles instead of a quartile of one
    u64* clk;                  // debugging (GF2_GATHER_CLOCK): earliest entry / latest exit of the workgroups at [2], [3], else null
    // the PREVIOUS pass' partial weights (its own buffer: the passes' weights alternate between two), summed into the histogram
    // by this launch's workgroups while nothing of their own is in flight yet (hand-scheduled kernel; null: nothing to combine):
    // the combine kernel's work without its launch, and without the two kernel boundaries around it, between two passes
    const unsigned short* cmb_pw;
    int64_t cmb_positions, cmb_pad;
    unsigned int cmb_sample0;
    int cmb_nbins, cmb_nslabs;   // (of the check those weights belong to: the Monte-Carlo run's two components ride in each other's launches)
    u64* cmb_hist;
    unsigned int* redo_count;
    unsigned int* redo_list;
};

__device__ __forceinline__ unsigned int keep_rows(int r, int row0) {
    const int left = r - row0;
    return left >= 32 ? ~0u : (left <= 0 ? 0u : ~(~0u << left));
}

// LDS offset of a table entry: plane base of this lane's row part + the 16-bit entry (one SDWA add per lookup).
__device__ __forceinline__ unsigned int entry_addr_lo(unsigned int part_base, unsigned int pair) {
    unsigned int out;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(out) : "v"(part_base), "v"(pair));
    return out;
}
__device__ __forceinline__ unsigned int entry_addr_hi(unsigned int part_base, unsigned int pair) {
    unsigned int out;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(out) : "v"(part_base), "v"(pair));
    return out;
}

// Quarter `part` (16 bytes = 8 slots) of the record at position `pos`: the four lanes of a record load one quarter each
// and hand the slots to each other with DPP quad broadcasts -- one memory instruction per step instead of four (address
// processing of a wave-wide load with scattered addresses costs about as much as the whole lookup loop).  Positions past
// the end read zeros.
__device__ __forceinline__ u32x4 fetch_record(__amdgpu_buffer_rsrc_t rec_rsrc, unsigned int group, unsigned int lane_rec, unsigned int part) {
    return __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, record_quarter_offset(group, lane_rec, part), 0, 0);
}

// Value of `v` in lane Q of this lane's quad.
template <int Q>
__device__ __forceinline__ unsigned int quad_bcast(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, Q * 0x55, 0xF, 0xF, true);
}

// The five error dwords that hold the identity-block bits of this lane's 128 rows of sample `sample`; addresses clamped into
// the row, the caller masks what lies outside.  No select on loaded values, so the loads stay in flight.
__device__ __forceinline__ void fetch_ident(const GatherArgs& a, unsigned int sample, unsigned int row_bytes, unsigned int dw0,
                                            unsigned int row_dwords, bool aligned16, int id_sh, bool last_part,
                                            unsigned int (&iw)[5]) {
    if (a.ident_off < 0) {
#pragma unroll
        for (int t = 0; t < 5; ++t) iw[t] = 0;
        return;
    }
    const unsigned int* rowp32 = reinterpret_cast<const unsigned int*>(
        reinterpret_cast<const char*>(a.e) + (u64)(sample < (unsigned int)a.batch ? sample : 0u) * row_bytes);
    if (aligned16) {
        const u32x4 x = *reinterpret_cast<const u32x4*>(rowp32 + dw0);
        iw[0] = x.x;
        iw[1] = x.y;
        iw[2] = x.z;
        iw[3] = x.w;
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) iw[t] = rowp32[dw0 + t < row_dwords ? dw0 + t : 0u];     // clamped; masked at use
    }
    iw[4] = 0;
    if (id_sh) {
        // the fifth dword is the neighbouring row part's first one; only the last part of the slab loads it
        if (last_part || !aligned16) iw[4] = rowp32[dw0 + 4 < row_dwords ? dw0 + 4 : 0u];
    }
}

// Value of `v` in lane q of this lane's quad (q is a constant after unrolling).
__device__ __forceinline__ unsigned int quad_pick(unsigned int v, int q) {
    switch (q) {
        case 0: return quad_bcast<0>(v);
        case 1: return quad_bcast<1>(v);
        case 2: return quad_bcast<2>(v);
        default: return quad_bcast<3>(v);
    }
}

// XORs the table entries of slots 0 .. 2 NH - 1 of a record into X.  Pair h (slots 2h, 2h + 1) is dword h mod 4 of record
// quarter h / 4, i.e. of quad lane h / 4.  The header sits in the high half of dword 7 of a 32-byte record (`mask7` = 0xFFFF
// there, all ones for a 64-byte record, whose dword 7 is two columns) and of dword 15 of a 64-byte one: it is replaced by the
// zero entry.  Straight-line code: the two ds_read_b128 of pair h are issued before pair h - AHEAD is consumed, so LDS latency
// (and its bank conflicts) overlaps the XORs of the same wavefront instead of relying on the other three wavefronts of the SIMD.
__device__ __forceinline__ unsigned int record_dword(const u32x4& R, int k) {
    return k == 0 ? R.x : (k == 1 ? R.y : (k == 2 ? R.z : R.w));
}
template <int NH>
__device__ __forceinline__ void lookup_halves(unsigned int part_base, const u32x4& R, unsigned int mask7, unsigned int nullpair,
                                              unsigned int (&X)[4]) {
    constexpr int AHEAD = 4;
    u32x4 v[NH][2];
#pragma unroll
    for (int h = 0; h < NH + AHEAD; ++h) {
        if (h < NH) {
            unsigned int pair = quad_pick(record_dword(R, h & 3), h >> 2);
            if (h == 7) pair = (pair & mask7) | (nullpair & ~mask7);
            if (h == 15) pair = (pair & 0xFFFFu) | (nullpair & 0xFFFF0000u);
            v[h][0] = *(lds_u32x4_ptr)(uintptr_t)entry_addr_lo(part_base, pair);
            v[h][1] = *(lds_u32x4_ptr)(uintptr_t)entry_addr_hi(part_base, pair);
        }
        if (h >= AHEAD && h - AHEAD < NH) {
            const int c = h - AHEAD;
            X[0] = xor3(X[0], v[c][0].x, v[c][1].x);
            X[1] = xor3(X[1], v[c][0].y, v[c][1].y);
            X[2] = xor3(X[2], v[c][0].z, v[c][1].z);
            X[3] = xor3(X[3], v[c][0].w, v[c][1].w);
        }
    }
}

// Sample a record belongs to: records are permuted inside their tile of 64 (the header carries the tile-local index).
__device__ __forceinline__ unsigned int record_sample(unsigned int pos, unsigned int header) {
    return (pos & ~63u) + ((header >> 8) & 63u);
}
// The header of this lane's record: the last slot, i.e. the high half of dword w of quarter 3 (a 32-byte record's quarters
// 2 and 3 are loaded as copies of 0 and 1).
__device__ __forceinline__ unsigned int record_header(const u32x4& R) { return quad_bcast<3>(R.w) >> 16; }

__global__ __launch_bounds__(GAT_THREADS, 4) void slab_gather_kernel(GatherArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];            // the only LDS: the table starts at LDS address 0
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroups b and b + 8 run on the same XCD (round-robin dispatch).  The nslabs workgroups that walk the same
    // samples are put on one XCD, so that a record (and the 128-byte lines of the error rows that two slabs share) is
    // fetched from HBM once and found in that XCD's L2 by the other slabs.
    int slab, share;
    const int shares = gridDim.x / a.nslabs;
    if ((shares & 7) == 0) {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        slab = local % a.nslabs;
        share = (local / a.nslabs) * 8 + xcd;
    } else {
        slab = blockIdx.x % a.nslabs;
        share = blockIdx.x / a.nslabs;
    }
    {
        const u32x4* src = a.tab + (int64_t)slab * a.tab_stride * 4;
        for (int v = threadIdx.x; v < a.tab_stride * 4; v += GAT_THREADS) reinterpret_cast<u32x4*>(lds)[v] = src[v];
    }
    __syncthreads();

    const int part = lane & 3;                                      // rows 128 part .. 128 part + 127 of the slab
    const unsigned int part_base = (unsigned int)part * (unsigned int)a.tab_stride * 16u;
    const int row0 = slab * SLAB_ROWS + part * 128;
    unsigned int keep[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) keep[q] = keep_rows(a.r, row0 + 32 * q);
    const int id_sh = a.ident_off >= 0 ? (a.ident_off & 31) : 0;
    // first identity dword of the slab (uniform) and of this lane: rows are multiples of 128, so the split is exact
    const unsigned int slab_dw0 = a.ident_off >= 0 ? (unsigned int)(a.ident_off >> 5) + slab * (SLAB_ROWS / 32) : 0u;
    const unsigned int dw0 = slab_dw0 + part * 4;
    const unsigned int row_dwords = (unsigned int)a.lde * 2u, row_bytes = (unsigned int)a.lde * 8u;
    const bool aligned16 = (a.lde & 1) == 0 && (reinterpret_cast<uintptr_t>(a.e) & 15) == 0 && (slab_dw0 & 3) == 0 &&
                           slab_dw0 + 16 <= row_dwords;      // uniform: one 16-byte load per lane
    unsigned int inrow[5];                                          // error dwords past the end of the row read as zero
#pragma unroll
    for (int t = 0; t < 5; ++t) inrow[t] = (a.ident_off >= 0 && dw0 + t < row_dwords && (t < 4 || id_sh)) ? ~0u : 0u;
    const unsigned int null_ent = (unsigned int)a.null_ord << 4;
    const bool from_neighbour = id_sh != 0 && aligned16;            // uniform
    bool mask_ident = a.ident_off >= 0 && (slab + 1) * SLAB_ROWS > a.r;              // uniform: rows past r in this slab
    if (a.ident_off >= 0 && slab_dw0 + 17 > row_dwords) mask_ident = true;           // or dwords past the end of the row
    const __amdgpu_buffer_rsrc_t rec_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a.rec), 0, (int)(((a.batch + 63) >> 6) * TILE_REC_BYTES), 0x00020000);
    unsigned short* const pw = a.pw + (int64_t)slab * a.batch_pad;

    const unsigned int ngroups = (unsigned int)(((a.batch + 63) >> 6) << 2);   // 16 records per wavefront step, whole tiles
    const unsigned int stride = (unsigned int)shares * GAT_WAVES;
    const unsigned int lane_rec = lane >> 2;

    // Software pipeline, two stages: the records of step i + 2 and the identity words of step i + 1 (their address needs
    // the record) are in flight while the lookups of step i run.  Three record and three identity buffers take turns
    // (the loop is unrolled by three), so nothing is copied between stages.
    u32x4 RA, RB, RC;
    unsigned int iwA[5], iwB[5], iwC[5];
    unsigned int grp = (unsigned int)share * GAT_WAVES + wave, k = 0;
    // A tile's four groups of 16 records ascend by count and a wavefront's group index keeps its residue mod 4 (the stride is a
    // multiple of 4), so without a rotation wavefront w would meet quartile w mod 4 of every tile -- the same quarter of the
    // wavefronts (one SIMD's, as they are dealt out) would get all the long records.  Step k takes quartile (w + k) mod 4.
    auto rec_grp = [&](unsigned int g, unsigned int rot) { return (g & ~3u) | ((g + rot) & 3u); };
    auto rec_pos = [&](unsigned int g, unsigned int rot) { return rec_grp(g, rot) * 16 + lane_rec; };
    RA = fetch_record(rec_rsrc, rec_grp(grp, 0), lane_rec, (unsigned int)part);
    RB = fetch_record(rec_rsrc, rec_grp(grp + stride, 1), lane_rec, (unsigned int)part);
    fetch_ident(a, record_sample(rec_pos(grp, 0), record_header(RA)), row_bytes, dw0, row_dwords, aligned16, id_sh,
                part == 3, iwA);

    auto step = [&](const u32x4& R, unsigned int (&iw)[5], const u32x4& Rnext, u32x4& Rfar, unsigned int (&iwnext)[5]) {
        const unsigned int pos = rec_pos(grp, k);
        Rfar = fetch_record(rec_rsrc, rec_grp(grp + 2 * stride, k + 2), lane_rec, (unsigned int)part);
        fetch_ident(a, record_sample(rec_pos(grp + stride, k + 1), record_header(Rnext)), row_bytes, dw0, row_dwords, aligned16, id_sh,
                    part == 3, iwnext);
        const unsigned int slot0 = record_header(R);                // count, tile-local sample, flags
        const bool is_short = (rec_grp(grp, k) & 3u) < 2u;          // uniform: 32-byte records, header in dword 7
        // finished by the compact kernel: REC_OVER sorts first (count 0); a REC_DONE record sits where its count put it, at
        // the top of the tile's lower half, and keeps the count byte, because the step's last record sets the lookups of all
        const bool flagged = (slot0 & 0xFFu) == REC_OVER || (slot0 & REC_DONE) != 0;
        const bool valid = record_sample(pos, slot0) < (unsigned int)a.batch;
        const unsigned int c = (slot0 & 0xFFu) == REC_OVER || !valid ? 0u : (slot0 & 0xFFu);
        if (from_neighbour) {                                       // quad_perm [1,2,3,3]: lane p takes lane p + 1's first dword
            const unsigned int nb = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)iw[0], 0xF9, 0xF, 0xF, false);
            if (part != 3) iw[4] = nb;
        }
        if (mask_ident) {                                           // uniform: only a slab that ends the rows or the row needs it
#pragma unroll
            for (int t = 0; t < 5; ++t) iw[t] &= inrow[t];
        }
        unsigned int X[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) X[q] = __builtin_amdgcn_alignbit(iw[q + 1], iw[q], id_sh);
        if (mask_ident) {
#pragma unroll
            for (int q = 0; q < 4; ++q) X[q] &= keep[q];
        }
        // the tile is sorted by count (finished and out-of-batch samples first, as 0): the last record has the most slots
        const unsigned int mx = (unsigned int)__builtin_amdgcn_readlane((int)c, 63);
        const unsigned int null_hi = null_ent << 16;
        auto lookups = [&](unsigned int lo, unsigned int hi) {      // four slots = two record dwords
            const u32x4 v0 = *(lds_u32x4_ptr)(uintptr_t)entry_addr_lo(part_base, lo);
            const u32x4 v1 = *(lds_u32x4_ptr)(uintptr_t)entry_addr_hi(part_base, lo);
            const u32x4 v2 = *(lds_u32x4_ptr)(uintptr_t)entry_addr_lo(part_base, hi);
            const u32x4 v3 = *(lds_u32x4_ptr)(uintptr_t)entry_addr_hi(part_base, hi);
            // v_bitop3_b32 (truth table 0x96 = three-way XOR): two instructions per dword and four columns instead of four
            X[0] = xor3(xor3(X[0], v0.x, v1.x), v2.x, v3.x);
            X[1] = xor3(xor3(X[1], v0.y, v1.y), v2.y, v3.y);
            X[2] = xor3(xor3(X[2], v0.z, v1.z), v2.z, v3.z);
            X[3] = xor3(xor3(X[3], v0.w, v1.w), v2.w, v3.w);
        };
        // slots 4g .. 4g + 3 are dwords (x, y) or (z, w) of quarter g / 2; the header (last slot) becomes a zero entry
        if (mx > 0) lookups(quad_bcast<0>(R.x), quad_bcast<0>(R.y));
        if (mx > 4) lookups(quad_bcast<0>(R.z), quad_bcast<0>(R.w));
        if (mx > 8) lookups(quad_bcast<1>(R.x), quad_bcast<1>(R.y));
        if (mx > 12) lookups(quad_bcast<1>(R.z), is_short ? (quad_bcast<1>(R.w) & 0xFFFFu) | null_hi : quad_bcast<1>(R.w));
        if (mx > 16) lookups(quad_bcast<2>(R.x), quad_bcast<2>(R.y));
        if (mx > 20) lookups(quad_bcast<2>(R.z), quad_bcast<2>(R.w));
        if (mx > 24) lookups(quad_bcast<3>(R.x), quad_bcast<3>(R.y));
        if (mx > 28) lookups(quad_bcast<3>(R.z), (quad_bcast<3>(R.w) & 0xFFFFu) | null_hi);
        unsigned int wt = __popc(X[0]) + __popc(X[1]) + __popc(X[2]) + __popc(X[3]);
        wt += __builtin_amdgcn_update_dpp(0u, wt, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
        wt += __builtin_amdgcn_update_dpp(0u, wt, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
        // partial weights are stored by record position (contiguous), which is all the histogram needs
        if (part == 0) pw[pos] = (unsigned short)(flagged || !valid ? REC_FLAG : wt);
        if (a.syn && !flagged && valid) {                           // this lane's 16 bytes of the sample's syndrome, what the row holds of them
            unsigned int* const out =
                reinterpret_cast<unsigned int*>(a.syn + (u64)record_sample(pos, slot0) * (u64)a.syn_pitch + slab * 64 + part * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (slab * 64 + part * 16 + q * 4 + 4 <= a.syn_row) out[q] = X[q];
        }
    };
#pragma unroll 1
    while (grp < ngroups) {
        step(RA, iwA, RB, RC, iwB);
        grp += stride, ++k;
        if (grp >= ngroups) break;
        step(RB, iwB, RC, RA, iwC);
        grp += stride, ++k;
        if (grp >= ngroups) break;
        step(RC, iwC, RA, RB, iwA);
        grp += stride, ++k;
    }
}

// ---- gather, hand-scheduled variant -----------------------------------------------------------------------------------
//
// Same work as slab_gather_kernel for the common shape: a standard-form check whose identity words are 16-byte pieces of the
// error rows (ident_off mod 128 < 32, even row pitch), which is what the n = 4096 checks are.  The generic kernel leaves the
// memory waits to the compiler, which ends every step with s_waitcnt vmcnt(0): whatever the ring depth, a request has one
// step (about 1.7 us, no more than an HBM round trip under load) to come back.  Here the loads are issued through inline
// assembly, which the compiler's wait insertion does not see, and the waits are written by hand: the record of step i + 4 and
// the identity words of step i + 2 are requested at the end of step i, and step i waits for exactly the requests it needs
// (vector memory operations complete in order, so "all but the last N").
typedef int i32x4 __attribute__((ext_vector_type(4)));

// EXTRA: ident_off is not a multiple of 32, a fifth dword per row part; CROSS: GatherArgs::cross (a template parameter: the
// step's address arithmetic is scalar code on its critical path)
// SYN: the syndromes are stored too (GatherArgs::syn): one more store per step, the counted waits allow for it
template <bool EXTRA, bool CROSS, bool SYN>
__global__ __launch_bounds__(GAT_THREADS, 4) void slab_gather_fast_kernel(GatherArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];            // the only LDS: the table starts at LDS address 0
    const int lane = threadIdx.x & 63;
    if (a.clk && threadIdx.x == 0) atomicMin(&a.clk[2], (u64)wall_clock64());
    int slab, share;                                                // as in slab_gather_kernel: the slabs of a share on one XCD
    const int shares = gridDim.x / a.nslabs;
    if ((shares & 7) == 0) {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        slab = local % a.nslabs;
        share = (local / a.nslabs) * 8 + xcd;
    } else {
        slab = blockIdx.x % a.nslabs;
        share = blockIdx.x / a.nslabs;
    }
    if (a.cmb_pw) {
        // The previous pass' combine step FIRST, its counters and staging area where the table is about to go: the launch then asks
        // for the table and the ticket counter only (131 KiB at n = 4096), and the 29 KiB that are left of a CU's LDS take a compact
        // workgroup of the other stream (30 KiB; 4 x 96 + 96 registers per SIMD lane fit too).  (With the counters behind the table --
        // 144 KiB -- the step read 0.5053 against 0.5064 of the HBM peak in the same process: gpurun_out/r04/ab5.log.)
        unsigned int* const scratch = reinterpret_cast<unsigned int*>(lds);
        combine_positions(a.cmb_pw, a.cmb_positions, a.cmb_pad, a.cmb_nslabs, a.cmb_hist, a.cmb_nbins, a.redo_count, a.redo_list,
                          a.cmb_sample0, scratch, scratch + SLAB_MAX_BINS, scratch + SLAB_MAX_BINS + 1024);
    }
    {
        // the slab's table: 9 loads per lane in flight (2400 columns x 4 parts = 9.4 rounds of 1024 lanes), then the LDS stores
        const u32x4* src = a.tab + (int64_t)slab * a.tab_stride * 4;
        const int total = a.tab_stride * 4;
        for (int v0 = threadIdx.x; v0 < total; v0 += 9 * GAT_THREADS) {
            u32x4 t[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int v = v0 + k * GAT_THREADS;
                t[k] = src[v < total ? v : v0];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int v = v0 + k * GAT_THREADS;
                if (v < total) reinterpret_cast<u32x4*>(lds)[v] = t[k];
            }
        }
    }
    // Steps are handed out from a counter behind the table.  The SIMD arbiter favours its oldest wavefront: with a fixed split the
    // four wavefronts of a SIMD finish one after the other (the first after 55 % of the kernel's duration) and the last
    // quarter of the work runs on one wavefront per SIMD with nothing to hide its latencies behind.
    unsigned int* const next_step = reinterpret_cast<unsigned int*>(lds + (size_t)a.tab_stride * 64);
    if (threadIdx.x == 0) *next_step = 0;
    __syncthreads();

    const int part = lane & 3;
    const unsigned int part_base = (unsigned int)part * (unsigned int)a.tab_stride * 16u;
    const int row0 = slab * SLAB_ROWS + part * 128;
    unsigned int keep[4];                                           // rows past r (all ones elsewhere: applied always)
#pragma unroll
    for (int q = 0; q < 4; ++q) keep[q] = keep_rows(a.r, row0 + 32 * q);
    const int id_sh = a.ident_off & 31;
    const unsigned int slab_dw0 = (unsigned int)(a.ident_off >> 5) + slab * (SLAB_ROWS / 32);
    const unsigned int dw0 = slab_dw0 + part * 4;
    const unsigned int row_dwords = (unsigned int)a.lde * 2u, row_bytes = (unsigned int)a.lde * 8u;
    unsigned int inrow[5];                                          // error dwords past the end of the row read as zero
#pragma unroll
    for (int t = 0; t < 5; ++t) inrow[t] = (dw0 + t < row_dwords && (t < 4 || EXTRA)) ? ~0u : 0u;
    if (!EXTRA) {
#pragma unroll
        for (int q = 0; q < 4; ++q) keep[q] &= inrow[q];            // the identity dwords are the rows' bits as they are
    }
    const unsigned int null_ent = (unsigned int)a.null_ord << 4;
    unsigned short* const pw = a.pw + (int64_t)slab * a.batch_pad;
    // columns next to the identity block that compact left out sit in the identity dwords of one row part: the bits to test
    unsigned int sm[4] = {0, 0, 0, 0};
    for (int t = 0; t < a.stray_n; ++t) {
        const unsigned int dc = (unsigned int)a.stray_col[t] >> 5;
        if (dc >= dw0 && dc < dw0 + 4) sm[dc - dw0] |= 1u << (a.stray_col[t] & 31);
    }
    const bool has_stray = __ballot((sm[0] | sm[1] | sm[2] | sm[3]) != 0) != 0;      // uniform
    // raw buffer over the records: positions past the end read zeros
    const u64 rec_base = reinterpret_cast<u64>(a.rec);
    const i32x4 rsrc = {__builtin_amdgcn_readfirstlane((int)(unsigned int)rec_base),
                        __builtin_amdgcn_readfirstlane((int)((unsigned int)(rec_base >> 32) & 0xFFFFu)),
                        __builtin_amdgcn_readfirstlane((int)(((a.batch + 63) >> 6) * TILE_REC_BYTES)), 0x00020000};
    const unsigned int syn_lane = (unsigned int)slab * 64u + (unsigned int)part * 16u;          // this lane's piece of a syndrome row
    const char* const ident_base = reinterpret_cast<const char*>(a.e) + dw0 * 4u;
    const unsigned int fifth_off = ((dw0 + 4 < row_dwords ? dw0 + 4 : 0u) - dw0) * 4u;       // from ident_base, wraps

    // A step looks up as many slot pairs as its longest record has.  The tiles are sorted by count, so the 16 records of a step
    // should sit close together in that order: a quartile of one tile spans a quarter of the tile's spread, the four records of
    // ranks 4k .. 4k + 3 taken from FOUR consecutive tiles span a sixteenth (`cross`, GF2_OPT_GATHER_CROSS; 16.4 -> 15.4 lookups
    // per record at the benchmark's rate).  Group g of a super-tile of four tiles: k = g mod 16; a step's records are 32-byte
    // ones for k < 8.  Measured (profiles/r03_*): the lookups it saves are paid back by its four short runs of records and
    // partial weights per step -- 2 % slower on one stream, equal on two -- so it is off by default.
    const unsigned int ntiles = (unsigned int)((a.batch + 63) >> 6);
    const unsigned int ngroups = CROSS ? ((ntiles + 3u) >> 2) << 4 : ntiles << 2;
    const unsigned int stride = (unsigned int)shares * GAT_WAVES;
    const unsigned int lane_rec = lane >> 2;
    const unsigned int part16 = (unsigned int)part * 16u;
    const unsigned int positions = ntiles << 6;

    // reverse: group g stands for group ngroups - 1 - g (groups past the end stay past the end: their records read as zeros)
    const unsigned int last_group = a.reverse ? ngroups - 1u : 0u;
    // the record quarter of this lane in its group: 32-byte records in the lower half of a tile, 64-byte ones above
    const unsigned int lane_tile = CROSS ? (lane_rec >> 2) : 0u, lane_sub = CROSS ? (lane_rec & 3u) : lane_rec;
    const unsigned int short_lane = lane_tile * TILE_REC_BYTES + lane_sub * 32u + ((unsigned int)part & 1u) * 16u;
    const unsigned int long_lane = lane_tile * TILE_REC_BYTES + lane_sub * 64u + part16;
    const unsigned int pos_lane = CROSS ? lane_tile * 64u + lane_sub : lane_rec;
    auto the_group = [&](unsigned int g) { return a.reverse && g < ngroups ? last_group - g : g; };
    auto issue_record = [&](u32x4& R, unsigned int g) {
        const unsigned int group = the_group(g);
        unsigned int off;
        if (CROSS) {
            const unsigned int k = group & 15u;
            off = (group >> 4) * (4u * TILE_REC_BYTES) + (k < 8 ? k * 128u + short_lane : 1024u + (k - 8u) * 256u + long_lane);
        } else {
            const unsigned int q = group & 3u;
            off = (group >> 2) * TILE_REC_BYTES + (q < 2 ? q * 512u + short_lane : (q - 1u) * 1024u + long_lane);
        }
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(R) : "v"(off), "s"(rsrc) : "memory");
    };
    auto issue_ident = [&](u32x4& I, unsigned int& E, unsigned int pos, unsigned int header) {
        const unsigned int sample = record_sample(pos, header);
        const char* p = ident_base + (u64)(sample < (unsigned int)a.batch ? sample : 0u) * row_bytes;
        asm volatile("global_load_dwordx4 %0, %1, off" GF2_IDENT_POLICY : "=v"(I) : "v"(p) : "memory");
        if (EXTRA) {
            // the fifth dword is the neighbouring row part's first one; only the last part of the slab loads it
            if (part == 3) asm volatile("global_load_dword %0, %1, off" GF2_IDENT_POLICY : "=v"(E) : "v"(p + (int)fifth_off) : "memory");
        }
    };

    u32x4 R0, R1, R2, R3, I0, I1, I2, I3;
    unsigned int E0 = 0, E1 = 0, E2 = 0, E3 = 0;
    // ticket t = step t / 16 of the fixed split's wavefront t mod 16: group share * 16 + t mod 16 + (t / 16) * stride
    // Tickets are taken FOUR at a time (one LDS atomic and its round trip per four steps instead of one per step; round 3): a
    // wavefront's four steps are then the four quartile groups of one tile, short records and long.
    unsigned int ticket_base = 0, ticket_used = 4;
    auto take = [&]() {
        if (ticket_used == 4) {                                     // uniform
            unsigned int t = 0;
            if (lane == 0) t = atomicAdd(next_step, 4u);
            ticket_base = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
            ticket_used = 0;
        }
        const unsigned int t = ticket_base + ticket_used++;
        return (unsigned int)share * GAT_WAVES + (t & (GAT_WAVES - 1)) + (t / GAT_WAVES) * stride;
    };
    // position of this lane's record (tile * 64 + rank in the sorted tile)
    auto rec_pos = [&](unsigned int g) {
        const unsigned int group = the_group(g);
        return (CROSS ? ((group >> 4) << 8) + ((group & 15u) << 2) : group << 4) + pos_lane;
    };
    auto short_records = [&](unsigned int g) { return CROSS ? (the_group(g) & 15u) < 8u : (the_group(g) & 3u) < 2u; };
    const unsigned int nullpair = null_ent | (null_ent << 16);
    unsigned int G0 = take(), G1 = take(), G2 = take(), G3 = take();
    issue_record(R0, G0);
    issue_record(R1, G1);
    issue_record(R2, G2);
    issue_record(R3, G3);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(R0), "+v"(R1), "+v"(R2), "+v"(R3)::"memory");
    issue_ident(I0, E0, rec_pos(G0), record_header(R0));
    issue_ident(I1, E1, rec_pos(G1), record_header(R1));
    issue_ident(I2, E2, rec_pos(G2), record_header(R2));
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(I0), "+v"(I1), "+v"(I2), "+v"(E0), "+v"(E1), "+v"(E2)::"memory");

    // One step: R, I, E = record and identity words of this step -- the record was requested four steps ago, the identity words
    // THREE (round 2: two; a memory round trip under load is about as long as a step, and with one step of lead the wavefront
    // sat in its wait: with the identity loads served from L2 the kernel ran 10 % faster, profiles/r03_gather_what_if.log).
    // Their address needs the header of the record of step i + 3, requested one step ago, so the step's ONE counted wait sits
    // at its end, between the request for the record of step i + 4 and the request for the identity words of step i + 3: all
    // but the last three operations (the identity words of step i + 2, this step's store, the record just requested) are
    // complete behind it -- record i + 3, and everything the next step reads.
    // (Rn, In, En: the next step's registers, valid behind the wait like Rp3 -- and only those are named in it: a register
    // whose load is still in flight must not appear as an operand, or the compiler may touch it.)
    auto step = [&](u32x4& R, u32x4& I, unsigned int& E, u32x4& Rn, u32x4& In, unsigned int& En, u32x4& Rp3, u32x4& Ip3,
                    unsigned int& Ep3, unsigned int& G, unsigned int Gp3) {
        const unsigned int pos = rec_pos(G);
        const unsigned int slot0 = record_header(R);                // count, tile-local sample, flags
        // finished by the compact kernel: REC_OVER sorts first (count 0); a REC_DONE record sits where its count put it, at
        // the top of the tile's lower half, and keeps the count byte, because the longest record sets the lookups of all
        const bool flagged = (slot0 & 0xFFu) == REC_OVER || (slot0 & REC_DONE) != 0;
        const bool valid = record_sample(pos, slot0) < (unsigned int)a.batch;
        const unsigned int c = (slot0 & 0xFFu) == REC_OVER || !valid ? 0u : (slot0 & 0xFFu);
        unsigned int stray = 0;
        if (has_stray) {
            stray = (I.x & sm[0]) | (I.y & sm[1]) | (I.z & sm[2]) | (I.w & sm[3]);
            stray |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)stray, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
            stray |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)stray, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
        }
        unsigned int iw[5] = {I.x, I.y, I.z, I.w, 0u};
        if (EXTRA) {                                                // quad_perm [1,2,3,3]: lane p takes lane p + 1's first dword
            const unsigned int nb = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)I.x, 0xF9, 0xF, 0xF, false);
            iw[4] = part != 3 ? nb : E;
        }
        unsigned int X[4];
        if (EXTRA) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                X[q] = __builtin_amdgcn_alignbit(iw[q + 1] & inrow[q + 1], iw[q] & inrow[q], id_sh) & keep[q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) X[q] = iw[q] & keep[q];
        }
        // the tiles are sorted by count (finished and out-of-batch samples first, as 0): the longest record of the step is the
        // last one of a tile's run, i.e. of lanes 15, 31, 47 or 63 (one run of 16 without `cross`: lane 63)
        unsigned int cmax = (unsigned int)__builtin_amdgcn_readlane((int)c, 63);
        cmax = max(cmax, (unsigned int)__builtin_amdgcn_readlane((int)c, 47));
        cmax = max(cmax, (unsigned int)__builtin_amdgcn_readlane((int)c, 31));
        cmax = max(cmax, (unsigned int)__builtin_amdgcn_readlane((int)c, 15));
        const unsigned int mask7 = short_records(G) ? 0xFFFFu : ~0u;          // uniform
        switch ((cmax + 1u) >> 1) {
            case 0: break;
            case 1: lookup_halves<1>(part_base, R, mask7, nullpair, X); break;
            case 2: lookup_halves<2>(part_base, R, mask7, nullpair, X); break;
            case 3: lookup_halves<3>(part_base, R, mask7, nullpair, X); break;
            case 4: lookup_halves<4>(part_base, R, mask7, nullpair, X); break;
            case 5: lookup_halves<5>(part_base, R, mask7, nullpair, X); break;
            case 6: lookup_halves<6>(part_base, R, mask7, nullpair, X); break;
            case 7: lookup_halves<7>(part_base, R, mask7, nullpair, X); break;
            case 8: lookup_halves<8>(part_base, R, mask7, nullpair, X); break;
            case 9: lookup_halves<9>(part_base, R, mask7, nullpair, X); break;
            case 10: lookup_halves<10>(part_base, R, mask7, nullpair, X); break;
            case 11: lookup_halves<11>(part_base, R, mask7, nullpair, X); break;
            case 12: lookup_halves<12>(part_base, R, mask7, nullpair, X); break;
            case 13: lookup_halves<13>(part_base, R, mask7, nullpair, X); break;
            case 14: lookup_halves<14>(part_base, R, mask7, nullpair, X); break;
            case 15: lookup_halves<15>(part_base, R, mask7, nullpair, X); break;
            default: lookup_halves<16>(part_base, R, mask7, nullpair, X); break;
        }
        unsigned int wt = __popc(X[0]) + __popc(X[1]) + __popc(X[2]) + __popc(X[3]);
        wt += __builtin_amdgcn_update_dpp(0u, wt, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
        wt += __builtin_amdgcn_update_dpp(0u, wt, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
        // (a super-tile's last tiles may lie past the batch: their records read as zeros and nothing is stored for them)
        if (SYN) {
            // this lane's 16 bytes of the sample's syndrome (a sample finished by the compact kernel has stored its own; one that has a
            // column compact left out is stored again, whole, by the redo kernel).  One store instruction per step whatever the data.
            // A lane with nothing to store writes to its own 16 bytes of a sink instead: no branch, so the instruction is issued in
            // every step (a step whose sixteen records are all finished or past the batch would otherwise issue one operation fewer
            // than the wait below counts).
            const u32x4 xs = {X[0], X[1], X[2], X[3]};
            const bool mine = !flagged && valid && pos < positions;
            char* const out = mine ? a.syn + (u64)record_sample(pos, slot0) * (u64)a.syn_pitch + syn_lane : a.syn_sink + lane * 16;
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(out), "v"(xs) : "memory");
        }
        if (part == 0 && pos < positions)
            pw[pos] = (unsigned short)(flagged || !valid ? REC_FLAG : (stray ? REC_STRAY | ((slot0 >> 8) & 63u) : wt));
        // refill the buffers this step has emptied: exactly one store (SYN: two), one record load and one (EXTRA: two) identity
        // loads per step, in this order -- the wait counts above depend on it
        asm volatile("" ::: "memory");
        G = take();
        issue_record(R, G);
        if (EXTRA && SYN)
            asm volatile("s_waitcnt vmcnt(5)" : "+v"(Rn), "+v"(In), "+v"(En), "+v"(Rp3)::"memory");
        else if (EXTRA || SYN)
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(Rn), "+v"(In), "+v"(En), "+v"(Rp3)::"memory");
        else
            asm volatile("s_waitcnt vmcnt(3)" : "+v"(Rn), "+v"(In), "+v"(En), "+v"(Rp3)::"memory");
        issue_ident(Ip3, Ep3, rec_pos(Gp3), record_header(Rp3));
    };
#pragma unroll 1
    while (G0 < ngroups) {                                          // a wavefront's tickets ascend
        step(R0, I0, E0, R1, I1, E1, R3, I3, E3, G0, G3);
        if (G1 >= ngroups) break;
        step(R1, I1, E1, R2, I2, E2, R0, I0, E0, G1, G0);
        if (G2 >= ngroups) break;
        step(R2, I2, E2, R3, I3, E3, R1, I1, E1, G2, G1);
        if (G3 >= ngroups) break;
        step(R3, I3, E3, R0, I0, E0, R2, I2, E2, G3, G2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.clk && lane == 0) atomicMax(&a.clk[3], (u64)wall_clock64());
}

// ---- combine ---------------------------------------------------------------------------------------------------------

// Four consecutive record positions per lane and step: one 8-byte load per slab (at most four slabs: r <= 2048), all issued before
// any is used.  Positions a slab has flagged REC_STRAY go on the redo list instead of into the histogram.
__global__ __launch_bounds__(1024) void slab_combine_kernel(const unsigned short* __restrict__ pw, int64_t positions, int64_t batch_pad,
                                                           int nslabs, u64* __restrict__ hist, int nbins, unsigned int* redo_count,
                                                           unsigned int* __restrict__ redo_list, unsigned int sample0, u64* clk) {
    __shared__ unsigned int bins[SLAB_MAX_BINS];
    __shared__ unsigned int redo_local[1024];                   // this workgroup's redo positions, handed over in one piece
    __shared__ unsigned int two[2];
    if (clk && threadIdx.x == 0) atomicMin(&clk[0], (u64)wall_clock64());
    combine_positions(pw, positions, batch_pad, nslabs, hist, nbins, redo_count, redo_list, sample0, bins, redo_local, two);
    if (clk && threadIdx.x == 0) atomicMax(&clk[1], (u64)wall_clock64());
}

// The samples on the redo list (they have a column that compact left out), one wavefront each, from the packed row.  The next
// listed sample's row is requested before this one's columns are gathered, and the weights are counted in LDS first: one global
// atomic per listed sample (round 3) made 900 thousand of them per call of 2^27 samples on the sixty or so bins that the weights
// of one check fall into.
__global__ __launch_bounds__(256) void slab_redo_kernel(const u64* __restrict__ e, int64_t batch, int64_t lde,
                                                        const unsigned int* __restrict__ redo_count,
                                                        const unsigned int* __restrict__ redo_list, const uint32_t* __restrict__ ht, int r,
                                                        int n, int ident_off, u64* __restrict__ hist, uint32_t* __restrict__ syn,
                                                        int64_t lds32) {
    __shared__ unsigned int lists[4][SPARSE_LIST_CAP + 8];
    __shared__ unsigned int bins[SLAB_MAX_BINS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned int count = *redo_count;
    if (blockIdx.x * 4u >= count) return;                           // (uniform over the workgroup: nothing listed for it)
    if (hist) {
        for (int i = threadIdx.x; i <= r; i += blockDim.x) bins[i] = 0;
        __syncthreads();
    }
    const SparseSide side = {ht, (int64_t)r, (int64_t)ident_off, nullptr, 0};
    const int words = (n + 63) >> 6;
    const unsigned int stride = gridDim.x * 4u;
    unsigned int i = blockIdx.x * 4u + wave;
    auto row_of = [&](unsigned int at) -> int64_t {
        const int64_t sample = at < count ? (int64_t)redo_list[at] : batch;
        return sample < batch ? sample : -1;
    };
    int64_t sample = row_of(i);
    u64 w_next = (sample >= 0 && lane < words) ? e[sample * lde + lane] : 0ull;
    for (; i < count; i += stride) {
        const u64 ww = w_next;
        const int64_t mine = sample;
        sample = row_of(i + stride);
        w_next = (sample >= 0 && lane < words) ? e[sample * lde + lane] : 0ull;
        if (mine < 0) continue;                                     // uniform
        unsigned int sdw;
        const unsigned int wt = sparse_component_weight(ww, side, n, lane, lists[wave], &sdw);
        if (lane == 0 && hist) atomicAdd(&bins[wt], 1u);
        if (syn && lane < ((r + 63) >> 6) * 2) syn[mine * lds32 + lane] = sdw;       // over what the gather kernel stored without the column
    }
    if (hist) {
        __syncthreads();
        for (int k = threadIdx.x; k <= r; k += blockDim.x)
            if (bins[k]) atomicAdd(&hist[k], (u64)bins[k]);
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------

// tab[slab][part][ord] (16 bytes) = rows 512 slab + 128 part .. + 127 of the ord-th non-identity column; ht has 64 dwords
// per column.  Entries ncols .. stride - 1 of every plane are zero.
__global__ void build_slab_table_kernel(const uint32_t* __restrict__ ht, int n, int r, int ident_off, int ncols, int stride,
                                        int nslabs, u32x4* __restrict__ tab) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;        // (part, ord)
    const int slab = blockIdx.y;
    if (idx >= stride * 4 || slab >= nslabs) return;
    const int part = idx / stride, ord = idx % stride;
    u32x4 v = {0, 0, 0, 0};
    if (ord < ncols) {
        const int col = (ident_off >= 0 && ord >= ident_off) ? ord + r : ord;
        if (col < n) v = *reinterpret_cast<const u32x4*>(ht + (int64_t)col * 64 + 16 * slab + 4 * part);
    }
    tab[(int64_t)slab * stride * 4 + idx] = v;
}

int gf2_build_slab_table(gf2_ctx* ctx, gf2_check* ck) {
    ck->slab_tab_dev = nullptr;
    ck->slab_cols = 0;
    ck->nslabs512 = 0;
    if (!ck->ht_dev || ck->ht_k != 1 || ck->n > 4096) return GF2_OK;
    const int64_t ncols = ck->ident_off >= 0 ? ck->n - ck->r : ck->n;
    if (ncols > SLAB_MAX_COLS || ck->r + 1 > SLAB_MAX_BINS) return GF2_OK;
    const int nslabs = (int)gf2_cdiv(ck->r, SLAB_ROWS);
    // entries per plane: the columns, at least one zero entry, padded to 4 mod 16 so that the four row parts of one entry
    // sit on four different quarters of the LDS banks
    int stride = (int)ncols + 1;
    while ((stride & 15) != 4) ++stride;
    GF2_TRY(gf2_dev_alloc(ctx, (size_t)nslabs * stride * 64, (void**)&ck->slab_tab_dev));
    dim3 grid((unsigned)gf2_cdiv(stride * 4, 256), (unsigned)nslabs);
    hipLaunchKernelGGL(build_slab_table_kernel, grid, dim3(256), 0, ctx->stream, (const uint32_t*)ck->ht_dev, (int)ck->n,
                       (int)ck->r, (int)ck->ident_off, (int)ncols, stride, nslabs, (u32x4*)ck->slab_tab_dev);
    GF2_HIP(hipGetLastError());
    GF2_HIP(hipStreamSynchronize(ctx->stream));
    ck->slab_cols = stride;
    ck->slab_null = (int)ncols;
    ck->nslabs512 = nslabs;
    return GF2_OK;
}

bool gf2_slabs_ok(const gf2_check* ck) { return ck->slab_tab_dev != nullptr; }

// Words at the edges of the identity block whose few non-identity columns cost compact a whole scan round (H2 of a CSS code
// with k logical qubits has r1 + k non-identity columns: k beyond the 32 words of H = [A | I | c]).  They may be left out
// of the records when (1) leaving them out saves a round, (2) they are at most two columns, and (3) every one of them lies
// in an identity dword that the hand-scheduled gather kernel loads anyway: that kernel flags the samples that have such a
// column (0.7 % per column at p = 0.01) and the redo kernel computes those from the packed row.
struct StrayPlan {
    u64 skip_words;
    int n_cols, col[2];
};

static int non_identity_words(const gf2_check* ck, u64 skip) {
    int nw = 0;
    for (int64_t wd = 0; wd < gf2_words(ck->n); ++wd) {
        const int64_t lo = wd * 64, hi = lo + 64 < ck->n ? lo + 64 : ck->n;
        const bool inside = ck->ident_off >= 0 && lo >= ck->ident_off && hi <= ck->ident_off + ck->r;
        nw += (inside || ((skip >> wd) & 1ull)) ? 0 : 1;
    }
    return nw;
}

static StrayPlan plan_stray(const gf2_ctx* ctx, const gf2_check* ck) {
    StrayPlan none = {0, 0, {0, 0}}, plan = none;
    if (ck->ident_off < 0 || ck->n > 4096 || gf2_flag(ctx, GF2_F_NO_REDO)) return none;
    const int64_t lo = ck->ident_off, hi = ck->ident_off + ck->r;           // identity columns [lo, hi)
    const int64_t first_dw = lo >> 5, last_dw = first_dw + (int64_t)ck->nslabs512 * (SLAB_ROWS / 32);
    int64_t cand[2][2] = {{(lo >> 6) * 64, lo}, {hi, ((hi >> 6) + 1) * 64 < ck->n ? ((hi >> 6) + 1) * 64 : ck->n}};
    for (int side = 0; side < 2; ++side) {
        const int64_t c0 = cand[side][0], c1 = cand[side][1];              // non-identity columns of the edge word
        if ((side == 0 && (lo & 63) == 0) || (side == 1 && ((hi & 63) == 0 || hi >= ck->n))) continue;
        // (a column of zeros adds nothing to any syndrome: it needs neither a place in the records nor the redo pass -- the last
        // column of H2 = first n - r_1 - 1 rows of [A^T | I] is one, BASELINE.json configs[3])
        auto nonzero = [&](int64_t c) { return ((ck->col_any[c >> 6] >> (c & 63)) & 1ull) != 0; };
        int64_t live = 0;
        for (int64_t c = c0; c < c1; ++c) live += nonzero(c) ? 1 : 0;
        if (c1 - c0 < 1 || plan.n_cols + live > 2) continue;
        bool visible = true;
        for (int64_t c = c0; c < c1; ++c) visible = visible && (!nonzero(c) || ((c >> 5) >= first_dw && (c >> 5) < last_dw));
        if (!visible) continue;
        // the word must not hold non-identity columns on its other side too (r < 64)
        if ((side == 0 ? hi : lo) > (c0 >> 6) * 64 && (side == 0 ? hi : lo) < (c0 >> 6) * 64 + 64 && ck->r < 64) continue;
        for (int64_t c = c0; c < c1; ++c)
            if (nonzero(c)) plan.col[plan.n_cols++] = (int)c;
        plan.skip_words |= 1ull << (c0 >> 6);
    }
    if (!plan.skip_words) return none;
    const int rounds_all = (CMP_SUB * non_identity_words(ck, 0) + 63) / 64;
    const int rounds_cut = (CMP_SUB * non_identity_words(ck, plan.skip_words) + 63) / 64;
    return rounds_cut < rounds_all ? plan : none;
}

// Samples per pass through the workspace (records, partial weights, redo list).
static int64_t slab_pass(const gf2_ctx* ctx, int64_t batch) {
    const int64_t cap = ctx->opt[GF2_OPT_SLAB_PASS_LOG2] > 0 ? (1ll << ctx->opt[GF2_OPT_SLAB_PASS_LOG2]) : SLAB_DEFAULT_BATCH;
    return batch < cap ? batch : cap;
}

static int slab_lds_optin(gf2_ctx* ctx) {
    if (!ctx->lds_optin[2]) {
        GF2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(slab_gather_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (SLAB_MAX_COLS + 16) * 64));
#define GF2_GATHER_OPTIN(E, C, S)                                                                                  \
    GF2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(slab_gather_fast_kernel<E, C, S>),                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, SLAB_GATHER_LDS_MAX))
        GF2_GATHER_OPTIN(false, false, false);
        GF2_GATHER_OPTIN(true, false, false);
        GF2_GATHER_OPTIN(false, true, false);
        GF2_GATHER_OPTIN(true, true, false);
        GF2_GATHER_OPTIN(false, false, true);
        GF2_GATHER_OPTIN(true, false, true);
#undef GF2_GATHER_OPTIN
        ctx->lds_optin[2] = true;
    }
    return GF2_OK;
}

// The gather kernel of one pass: records, identity words (rows of lde words at e) and the workspace are in place.
// syn (null: none): where the pass' first sample's syndrome goes, pitch bytes per sample; sink: 1 KiB for the lanes that store nothing
static int launch_gather(gf2_ctx* ctx, const gf2_check* ck, const uint64_t* e, u32x4* rec, unsigned short* pw, int64_t count, int64_t pad,
                         int64_t lde, bool fast, const StrayPlan& stray, hipStream_t stream, u64* clk_dev, char* syn = nullptr,
                         int64_t syn_pitch = 0, char* syn_sink = nullptr, const GatherArgs* combine_of = nullptr) {
    const size_t lds_bytes = (size_t)ck->slab_cols * 64;
    GatherArgs ga;
    ga.tab = (const u32x4*)ck->slab_tab_dev;
    ga.e = (const u64*)e;
    ga.rec = rec;
    ga.pw = pw;
    ga.batch = count;
    ga.batch_pad = pad;
    ga.lde = lde;
    ga.tab_stride = ck->slab_cols;
    ga.nslabs = ck->nslabs512;
    ga.r = (int)ck->r;
    ga.ident_off = (int)ck->ident_off;
    ga.null_ord = ck->slab_null;
    ga.stray_n = stray.n_cols;
    ga.stray_col[0] = stray.col[0];
    ga.stray_col[1] = stray.col[1];
    ga.clk = clk_dev;
    ga.syn = syn;
    ga.syn_pitch = syn_pitch;
    ga.syn_sink = syn_sink;
    ga.syn_row = (int)gf2_words(ck->r) * 8;
    ga.cmb_pw = nullptr;
    ga.cmb_positions = ga.cmb_pad = 0;
    ga.cmb_sample0 = 0;
    ga.cmb_nbins = ga.cmb_nslabs = 0;
    ga.cmb_hist = nullptr;
    ga.redo_count = ga.redo_list = nullptr;
    size_t extra_lds = 16;
    if (combine_of && fast) {                                          // the previous pass' combine step rides in this launch
        ga.cmb_pw = combine_of->cmb_pw;
        ga.cmb_positions = combine_of->cmb_positions;
        ga.cmb_pad = combine_of->cmb_pad;
        ga.cmb_sample0 = combine_of->cmb_sample0;
        ga.cmb_nbins = combine_of->cmb_nbins;
        ga.cmb_nslabs = combine_of->cmb_nslabs;
        ga.cmb_hist = combine_of->cmb_hist;
        ga.redo_count = combine_of->redo_count;
        ga.redo_list = combine_of->redo_list;
        const size_t scratch = (SLAB_MAX_BINS + 1024 + 2) * 4;             // (in the table's place: no LDS of its own ...
        if (lds_bytes + extra_lds < scratch) extra_lds = scratch - lds_bytes;   // ... unless the table is smaller: a check of few columns)
    }
    ga.reverse = ctx->opt[GF2_OPT_GATHER_REVERSE] == 0 ? 0 : 1;
    ga.cross = ctx->opt[GF2_OPT_GATHER_CROSS] == 1 && !syn ? 1 : 0;       // (the cross-tile grouping has no variant that stores syndromes)
    // GF2_OPT_GATHER_OVER workgroups per CU over the launch (default 1): more of them let a CU that is done early take another
    // share, at the price of one more copy of the slab's table into LDS each
    const int64_t over = ctx->opt[GF2_OPT_GATHER_OVER] > 0 ? ctx->opt[GF2_OPT_GATHER_OVER] : 1;
    int64_t shares = ctx->num_cus / ck->nslabs512 * over;
    const int64_t max_shares = gf2_cdiv(gf2_cdiv(count, 16), GAT_WAVES);   // (a few steps more with `cross`: they read zeros)
    if (shares > max_shares) shares = max_shares;
    if (shares < 1) shares = 1;
    const dim3 ggrid((unsigned)(shares * ck->nslabs512));
    const bool extra = (ck->ident_off & 31) != 0;
    if (fast && syn && extra)
        hipLaunchKernelGGL((slab_gather_fast_kernel<true, false, true>), ggrid, dim3(GAT_THREADS), lds_bytes + extra_lds, stream, ga);
    else if (fast && syn)
        hipLaunchKernelGGL((slab_gather_fast_kernel<false, false, true>), ggrid, dim3(GAT_THREADS), lds_bytes + extra_lds, stream, ga);
    else if (fast && extra && ga.cross)
        hipLaunchKernelGGL((slab_gather_fast_kernel<true, true, false>), ggrid, dim3(GAT_THREADS), lds_bytes + extra_lds, stream, ga);
    else if (fast && extra)
        hipLaunchKernelGGL((slab_gather_fast_kernel<true, false, false>), ggrid, dim3(GAT_THREADS), lds_bytes + extra_lds, stream, ga);
    else if (fast && ga.cross)
        hipLaunchKernelGGL((slab_gather_fast_kernel<false, true, false>), ggrid, dim3(GAT_THREADS), lds_bytes + extra_lds, stream, ga);
    else if (fast)
        hipLaunchKernelGGL((slab_gather_fast_kernel<false, false, false>), ggrid, dim3(GAT_THREADS), lds_bytes + extra_lds, stream, ga);
    else
        hipLaunchKernelGGL(slab_gather_kernel, ggrid, dim3(GAT_THREADS), lds_bytes, stream, ga);
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

// The combine kernel of one pass (sample0: the pass' first sample in the call's batch, for the redo list).
static int launch_combine(gf2_ctx* ctx, const gf2_check* ck, const unsigned short* pw, unsigned int* redo_count, unsigned int* redo_list,
                          int64_t count, int64_t pad, unsigned int sample0, uint64_t* hist_dev, hipStream_t stream, u64* clk_dev) {
    const int nbins = (int)ck->r + 1;
    // few large workgroups: every workgroup ends with one global atomic per non-empty bin.  (Workgroups of 256, 128 or 64 threads,
    // which would fit a CU beside a gather workgroup of the other stream instead of waiting for one to drain, make the two-stream
    // step 4, 6 and 7 % SLOWER: profiles/r02_sweep_combine.log.)
    const int cthreads = ctx->opt[GF2_OPT_COMBINE_THREADS] > 0 ? (int)ctx->opt[GF2_OPT_COMBINE_THREADS] : 1024;
    int64_t mblocks = gf2_cdiv(count, 4 * cthreads);
    const int64_t mb_cap = ctx->opt[GF2_OPT_COMBINE_BLOCKS] > 0 ? ctx->opt[GF2_OPT_COMBINE_BLOCKS] : 128 * (1024 / cthreads);
    if (mblocks > mb_cap) mblocks = mb_cap;
    hipLaunchKernelGGL(slab_combine_kernel, dim3((unsigned)mblocks), dim3((unsigned)cthreads), 0, stream, pw, gf2_cdiv(count, 64) * 64, pad,
                       ck->nslabs512, (u64*)hist_dev, nbins, redo_count, redo_list, sample0, clk_dev ? clk_dev + 4 : nullptr);
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

// The redo kernel over the whole call's batch: the samples the combine steps have listed (they have a column compact left out).
static int launch_redo(gf2_ctx* ctx, const gf2_check* ck, const uint64_t* e_dev, int64_t batch, int64_t lde, unsigned int* redo_count,
                       unsigned int* redo_list, uint64_t* hist_dev, uint64_t* s_dev, int64_t lds, hipStream_t stream) {
    const int64_t per_cu = ctx->opt[GF2_OPT_REDO_BLOCKS_PER_CU] > 0 ? ctx->opt[GF2_OPT_REDO_BLOCKS_PER_CU] : 8;
    hipLaunchKernelGGL(slab_redo_kernel, dim3((unsigned)(ctx->num_cus * per_cu)), dim3(256), 0, stream, (const u64*)e_dev, batch, lde,
                       (const unsigned int*)redo_count, (const unsigned int*)redo_list, ck->ht_dev, (int)ck->r, (int)ck->n,
                       (int)ck->ident_off, (u64*)hist_dev, (uint32_t*)s_dev, lds * 2);
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

// One call's worth of constants of the pipeline on one check: workspace pieces, kernel variant, columns left to the redo pass.
struct SlabCall {
    const gf2_check* ck;
    const uint64_t* e_dev;
    uint64_t* hist_dev;            // null: syndromes only
    uint64_t* s_dev;               // syndromes out, lds words per sample (null: histogram only)
    int64_t lds;
    char* syn_sink;
    int64_t batch, lde, pass, pad;
    u32x4* rec;
    unsigned short* pw;
    unsigned int* redo_count;
    unsigned int* redo_list;
    bool fast, fold;
    bool gfold;                    // the combine step of a pass rides in the next pass' gather kernel (two buffers of partial weights)
    StrayPlan stray;
};

static int slab_call_setup(gf2_ctx* ctx, const gf2_check* ck, const uint64_t* e_dev, int64_t batch, int64_t lde, uint64_t* hist_dev,
                           uint64_t* s_dev, int64_t lds, int ws_slot, SlabCall* c) {
    GF2_TRY(slab_lds_optin(ctx));
    c->ck = ck;
    c->e_dev = e_dev;
    c->hist_dev = hist_dev;
    c->s_dev = s_dev;
    c->lds = lds;
    c->batch = batch;
    c->lde = lde;
    c->pass = slab_pass(ctx, batch);
    c->pad = gf2_cdiv(c->pass, 64) * 64;
    // records and partial weights of one pass; the redo list is the call's (one redo launch at the end)
    const size_t rec_bytes = (size_t)c->pad * 64, pw_bytes = 2 * (size_t)ck->nslabs512 * c->pad * 2;     // (two passes' partial weights)
    const size_t redo_bytes = (size_t)(batch > c->pad ? gf2_cdiv(batch, 64) * 64 : c->pad) * 4 + 256;
    GF2_TRY(gf2_ws_reserve(ctx, ws_slot, rec_bytes + pw_bytes + redo_bytes + 1024));
    c->rec = (u32x4*)ctx->ws[ws_slot];
    c->pw = (unsigned short*)((char*)ctx->ws[ws_slot] + rec_bytes);
    c->redo_count = (unsigned int*)((char*)ctx->ws[ws_slot] + rec_bytes + pw_bytes);
    c->redo_list = c->redo_count + 64;
    c->syn_sink = (char*)ctx->ws[ws_slot] + rec_bytes + pw_bytes + redo_bytes;
    // the hand-scheduled gather kernel needs every slab's identity words to be whole 16-byte pieces inside the row
    const int64_t first_dw = ck->ident_off >= 0 ? ck->ident_off >> 5 : 0;
    c->fast = ck->ident_off >= 0 && (lde & 1) == 0 && (reinterpret_cast<uintptr_t>(e_dev) & 15) == 0 && (first_dw & 3) == 0 &&
              first_dw + (int64_t)ck->nslabs512 * (SLAB_ROWS / 32) <= lde * 2 && !gf2_flag(ctx, GF2_F_GATHER_GENERIC);
    // ... and, when it stores syndromes, every slab's 64-byte piece to lie inside a syndrome row
    if (s_dev && ((int64_t)ck->nslabs512 * 64 > gf2_words(ck->r) * 8 || (reinterpret_cast<uintptr_t>(s_dev) & 15) != 0 || (lds & 1) != 0)) c->fast = false;
    const StrayPlan none = {0, 0, {0, 0}};
    c->stray = c->fast ? plan_stray(ctx, ck) : none;
    // GF2_F_COMBINE_FOLDED: between two passes of a call the combine step rides in the next pass' compact kernel (its first
    // instructions, before any of its own loads) -- 14 launches fewer per call of 8 passes.  Built because the small kernels sit
    // on the stream's critical path and are stretched by the other stream's big ones; measured: no change on two streams, 5 %
    // slower on one (1280 workgroups' worth of histogram atomics instead of 128: profiles/r02_sweep_fold.log).  Off by default.
    c->fold = gf2_flag(ctx, GF2_F_COMBINE_FOLDED) && !gf2_flag(ctx, GF2_F_DIAG_CLOCKS);
    // Round 4: by default the combine step of pass k rides in the gather kernel of pass k + 1 instead (its workgroups do it first
    // of all, in the LDS their table goes into afterwards; GF2_F_COMBINE_SEPARATE keeps the kernel after every pass): the
    // hand-scheduled kernel only.
    c->gfold = c->fast && !c->fold && !gf2_flag(ctx, GF2_F_COMBINE_SEPARATE) && !gf2_flag(ctx, GF2_F_DIAG_CLOCKS) &&
               (hist_dev || c->stray.n_cols);
    return GF2_OK;
}

// The compact kernel of the pass that starts at sample `first` (with the previous pass' combine step in it when folded).
static int launch_compact(gf2_ctx* ctx, const SlabCall& c, int64_t first, hipStream_t stream, u64* clk_dev) {
    const gf2_check* ck = c.ck;
    const int64_t count = c.batch - first < c.pass ? c.batch - first : c.pass;
    CompactArgs ca;
    ca.clk = clk_dev;
    ca.e = (const u64*)(c.e_dev + first * c.lde);
    ca.ht = ck->ht_dev;
    ca.rec = c.rec;
    ca.hist = (u64*)c.hist_dev;
    ca.syn = c.s_dev ? (uint32_t*)(c.s_dev + first * c.lds) : nullptr;
    ca.lds32 = c.lds * 2;
    ca.batch = count;
    ca.lde = c.lde;
    ca.r = (int)ck->r;
    ca.n = (int)ck->n;
    ca.ident_off = (int)ck->ident_off;
    ca.null_ord = ck->slab_null;
    ca.skip_words = c.stray.skip_words;
    ca.cmb_pw = c.fold && first > 0 ? c.pw : nullptr;
    ca.cmb_positions = c.pad;                                          // (every pass before the last is a full one)
    ca.cmb_pad = c.pad;
    ca.cmb_sample0 = (unsigned int)(first - c.pass);
    ca.cmb_nslabs = ck->nslabs512;
    ca.cmb_nbins = (int)ck->r + 1;
    ca.redo_count = c.redo_count;
    ca.redo_list = c.redo_list;
    int64_t cblocks = gf2_cdiv(gf2_cdiv(count, 64), CMP_WAVES);
    // rounds per sub-pass: ceil(8 * words with non-identity columns / 64); workgroups per CU as the variant's registers allow
    const int rounds = (CMP_SUB * non_identity_words(ck, c.stray.skip_words) + 63) / 64;
    const int per_cu = rounds <= 4 ? 5 : (rounds <= 5 ? 4 : 3);
    if (cblocks > (int64_t)ctx->num_cus * per_cu) cblocks = (int64_t)ctx->num_cus * per_cu;
    const dim3 cgrid((unsigned)cblocks), cblock(CMP_THREADS);
    // all pairs live and all scanned words whole: no masks (the benchmark's two checks)
    bool full = CMP_SUB * non_identity_words(ck, c.stray.skip_words) == 64 * 4 && gf2_words(ck->n) * 64 == ck->n;
    for (int64_t wd = 0; full && wd < gf2_words(ck->n); ++wd) {
        const int64_t lo = wd * 64, hi = lo + 64;
        const bool inside = ck->ident_off >= 0 && lo >= ck->ident_off && hi <= ck->ident_off + ck->r;
        const bool outside = ck->ident_off < 0 || hi <= ck->ident_off || lo >= ck->ident_off + ck->r;
        full = inside || outside || ((c.stray.skip_words >> wd) & 1ull);
    }
    if (rounds <= 4 && full)
        hipLaunchKernelGGL((slab_compact_kernel<4, true>), cgrid, cblock, 0, stream, ca);
    else if (rounds <= 4)
        hipLaunchKernelGGL((slab_compact_kernel<4, false>), cgrid, cblock, 0, stream, ca);
    else if (rounds <= 5)
        hipLaunchKernelGGL((slab_compact_kernel<5, false>), cgrid, cblock, 0, stream, ca);
    else if (rounds <= 6)
        hipLaunchKernelGGL((slab_compact_kernel<6, false>), cgrid, cblock, 0, stream, ca);
    else
        hipLaunchKernelGGL((slab_compact_kernel<8, false>), cgrid, cblock, 0, stream, ca);
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

// gather and (unless it rides in the next compact or gather kernel) combine of the pass that starts at `first`.
static int launch_rest_of_pass(gf2_ctx* ctx, const SlabCall& c, int64_t first, hipStream_t stream, u64* clk_dev) {
    const int64_t count = c.batch - first < c.pass ? c.batch - first : c.pass;
    const int64_t index = first / c.pass;
    const size_t pw_one = (size_t)c.ck->nslabs512 * c.pad;                         // partial weights of one pass (16-bit words)
    unsigned short* const pw = c.pw + (c.gfold ? (size_t)(index & 1) * pw_one : 0);
    GatherArgs prev;
    const bool ride = c.gfold && first > 0;
    if (ride) {
        prev.cmb_pw = c.pw + (size_t)((index - 1) & 1) * pw_one;
        prev.cmb_positions = c.pad;                                                 // (every pass before the last is a full one)
        prev.cmb_pad = c.pad;
        prev.cmb_sample0 = (unsigned int)(first - c.pass);
        prev.cmb_nbins = (int)c.ck->r + 1;
        prev.cmb_nslabs = c.ck->nslabs512;
        prev.cmb_hist = (u64*)c.hist_dev;
        prev.redo_count = c.redo_count;
        prev.redo_list = c.redo_list;
    }
    GF2_TRY(launch_gather(ctx, c.ck, c.e_dev + first * c.lde, c.rec, pw, count, c.pad, c.lde, c.fast, c.stray, stream, clk_dev,
                          c.s_dev ? (char*)(c.s_dev + first * c.lds) : nullptr, c.lds * 8, c.syn_sink, ride ? &prev : nullptr));
    if (!c.hist_dev && !c.stray.n_cols) return GF2_OK;               // syndromes only and nothing to list: no combine step
    const bool last = first + c.pass >= c.batch;
    if ((!c.fold && !c.gfold) || last)
        GF2_TRY(launch_combine(ctx, c.ck, pw, c.redo_count, c.redo_list, count, c.pad, (unsigned int)first, c.hist_dev, stream, clk_dev));
    return GF2_OK;
}

// Resident sample-major errors through the pipeline.
// Weight histogram (hist_dev: r + 1 bins, accumulated into; null: none) and / or syndromes (s_dev: lds words per sample; null: none).
int gf2_syndrome_slabs(gf2_ctx* ctx, const gf2_check* ck, const uint64_t* e_dev, int64_t batch, int64_t lde,
                       uint64_t* hist_dev, hipStream_t stream, int ws_slot, uint64_t* s_dev, int64_t lds) {
    SlabCall c;
    GF2_TRY(slab_call_setup(ctx, ck, e_dev, batch, lde, hist_dev, s_dev, lds, ws_slot, &c));
    if (c.fold && !hist_dev) c.fold = false;
    GF2_HIP(hipMemsetAsync(c.redo_count, 0, 4, stream));
    for (int64_t first = 0; first < batch; first += c.pass) {
        u64* clk_dev = nullptr;
        if (gf2_flag(ctx, GF2_F_DIAG_CLOCKS)) {
            static const u64 init[6] = {~0ull, 0, ~0ull, 0, ~0ull, 0};
            GF2_HIP(hipMalloc((void**)&clk_dev, 64));
            GF2_HIP(hipMemcpy(clk_dev, init, 48, hipMemcpyHostToDevice));
        }
        GF2_TRY(launch_compact(ctx, c, first, stream, clk_dev));
        GF2_TRY(launch_rest_of_pass(ctx, c, first, stream, clk_dev));
        if (clk_dev) {
            u64 t[6];
            GF2_HIP(hipStreamSynchronize(stream));
            GF2_HIP(hipMemcpy(t, clk_dev, 48, hipMemcpyDeviceToHost));
            fprintf(stderr, "pipeline: compact %.1f .. %.1f us, combine %.1f .. %.1f us (from compact's first workgroup)\n", 0.0,
                    (double)(t[1] - t[0]) / 100.0, (double)(t[4] - t[0]) / 100.0, (double)(t[5] - t[0]) / 100.0);
            fprintf(stderr, "pipeline: gather %.1f .. %.1f us\n", (double)((int64_t)(t[2] - t[0])) / 100.0,
                    (double)((int64_t)(t[3] - t[0])) / 100.0);
            GF2_HIP(hipFree(clk_dev));
        }
    }
    if (c.stray.n_cols) GF2_TRY(launch_redo(ctx, ck, e_dev, batch, lde, c.redo_count, c.redo_list, hist_dev, s_dev, lds, stream));
    return GF2_OK;
}

// Grows the workspace of `ws_slot` for batches of this size now (growing synchronises the streams), so that a later
// gf2_syndrome_slabs on a side stream does not.
int gf2_slabs_reserve(gf2_ctx* ctx, const gf2_check* ck, int64_t batch, int ws_slot) {
    const int64_t pass = slab_pass(ctx, batch);
    const int64_t pad = gf2_cdiv(pass, 64) * 64;
    return gf2_ws_reserve(ctx, ws_slot, (size_t)pad * 64 + 2 * (size_t)ck->nslabs512 * pad * 2 + (size_t)pad * 4 + 256 + 1024);
}

// ---- the Monte-Carlo run's own way in: records from the sampler ---------------------------------------------------------------

// Both checks have row slabs, an identity block and at most 4096 columns; the caller looks at the error rate.
static void identity_segments(const gf2_check* ck, int nseg, int* lo, int* hi) {
    // the dwords the gather kernel loads: 16 per slab from the identity block's first, one more when the block is not dword-aligned
    const int64_t first_dw = ck->ident_off >> 5;
    const int64_t last_dw = first_dw + (int64_t)ck->nslabs512 * (SLAB_ROWS / 32) + ((ck->ident_off & 31) ? 4 : 0);
    *lo = (int)(first_dw / 16);
    *hi = (int)gf2_cdiv(last_dw, 16);
    if (*hi > nseg) *hi = nseg;
}

// ... and no segment of 512 columns holds identity words of both (the record sampler keeps one image of identity words per
// segment): true of a CSS code's standard forms H1 = [I | A], H2 = [A' | I | c] with r1 a multiple of 512.
bool gf2_mc_records_ok(const gf2_check* c1, const gf2_check* c2) {
    if (!(gf2_slabs_ok(c1) && gf2_slabs_ok(c2) && c1->ident_off >= 0 && c2->ident_off >= 0 && c1->n == c2->n && c1->n <= 4096 &&
          c1->n >= 64))
        return false;
    const int nseg = (int)gf2_cdiv(c1->n, GF2_SEG_BITS);
    int lo1, hi1, lo2, hi2;
    identity_segments(c1, nseg, &lo1, &hi1);
    identity_segments(c2, nseg, &lo2, &hi2);
    return hi1 <= lo2 || hi2 <= lo1;
}

// Bytes of one buffer set for `pass` samples: per component the identity rows, the records and the misfit list.
size_t gf2_mc_records_bytes(int64_t n, int64_t pass) {
    const size_t pad = (size_t)gf2_cdiv(pass, 64) * 64;
    const size_t row = (size_t)pad * gf2_words(n) * 8, rec = pad / 64 * TILE_REC_BYTES, mis = 256 + pad * 4;
    return 2 * (row + ((rec + 255) & ~(size_t)255) + mis);
}

// The Monte-Carlo run's chunks (gf2_mc_run at n <= 4096, sparse rates; buf: one buffer set of gf2_mc_records_bytes(n, pass) bytes,
// hz / hx: the two weight histograms on the device, accumulated into), all on `stream`.  Per chunk of at most `pass` samples:
//   the record sampler (records, identity words and misfit lists of both components);
//   gather Z (against c1), with the combine step of the PREVIOUS chunk's X component in its workgroups' prologue;
//   gather X (against c2), with the combine step of this chunk's Z component in its prologue;
//   one launch of the misfit kernel for both components.
// Four launches per chunk (round 3: nine operations -- two memsets of the misfit counters, the sampler, and gather, combine, misfit per
// component); the last chunk's X component is combined by a launch of its own at the end.
int gf2_mc_records_run(gf2_ctx* ctx, const gf2_check* c1, const gf2_check* c2, uint64_t seed, int64_t first_sample, int64_t count,
                       int64_t pass, const SegTables& th, void* buf, uint64_t* hz_dev, uint64_t* hx_dev, hipStream_t stream) {
    GF2_TRY(slab_lds_optin(ctx));
    const int64_t n = c1->n, lde = gf2_words(n);
    const size_t pad = (size_t)gf2_cdiv(pass, 64) * 64;
    const size_t row = (size_t)pad * lde * 8, rec_b = ((pad / 64 * TILE_REC_BYTES) + 255) & ~(size_t)255, mis = 256 + pad * 4;
    // component 0: X errors against c2, component 1: Z errors against c1 (css_code.py:457-470)
    const gf2_check* cks[2] = {c2, c1};
    uint64_t* hists[2] = {hx_dev, hz_dev};
    const int ws_slots[2] = {3, 2};
    struct Side {
        const uint64_t* eident;
        u32x4* rec;
        unsigned int* counters;      // two slots, then (from word 64 on) the list
        unsigned short* pw;
        unsigned int* redo_count;
        bool fast, can_carry;
    } side[2];
    for (int c = 0; c < 2; ++c) {
        const gf2_check* ck = cks[c];
        char* q = (char*)buf + (size_t)c * (row + rec_b + mis);
        side[c].eident = (const uint64_t*)q;
        side[c].rec = (u32x4*)(q + row);
        side[c].counters = (unsigned int*)(q + row + rec_b);
        const size_t pw_bytes = (size_t)ck->nslabs512 * pad * 2, redo_bytes = (size_t)pad * 4 + 256;
        GF2_TRY(gf2_ws_reserve(ctx, ws_slots[c], (size_t)pad * 64 + pw_bytes + redo_bytes));
        side[c].pw = (unsigned short*)((char*)ctx->ws[ws_slots[c]] + (size_t)pad * 64);
        side[c].redo_count = (unsigned int*)((char*)ctx->ws[ws_slots[c]] + (size_t)pad * 64 + pw_bytes);
        const int64_t first_dw = ck->ident_off >> 5;
        side[c].fast = (lde & 1) == 0 && (first_dw & 3) == 0 && first_dw + (int64_t)ck->nslabs512 * (SLAB_ROWS / 32) <= lde * 2 &&
                       !gf2_flag(ctx, GF2_F_GATHER_GENERIC);
        // its launch can carry the other component's combine step: the hand-scheduled kernel
        side[c].can_carry = side[c].fast && !gf2_flag(ctx, GF2_F_COMBINE_SEPARATE);
        GF2_HIP(hipMemsetAsync(side[c].counters, 0, 256, stream));          // both slots, once per run
    }
    const StrayPlan none = {0, 0, {0, 0}};
    GatherArgs pending;                                               // the combine step nobody has done yet
    int pending_of = -1;
    int64_t pending_count = 0;
    auto flush = [&]() -> int {                                       // ... by a launch of its own
        if (pending_of < 0) return GF2_OK;
        const Side& sd = side[pending_of];
        GF2_TRY(launch_combine(ctx, cks[pending_of], sd.pw, sd.redo_count, sd.redo_count + 64, pending_count, (int64_t)pad, 0u,
                               hists[pending_of], stream, nullptr));
        pending_of = -1;
        return GF2_OK;
    };
    int64_t chunk = 0;
    for (int64_t done = 0; done < count; done += pass, ++chunk) {
        const int64_t now = count - done < pass ? count - done : pass;
        const int parity = (int)(chunk & 1);
        RecSamplerArgs a;
        a.seed = seed;
        a.first_sample = first_sample + done;
        a.count = now;
        a.lde = lde;
        a.th = th;
        a.n = (int)n;
        a.cap = ctx->opt[GF2_OPT_MC_TAIL_CAP] >= 0 ? (int)ctx->opt[GF2_OPT_MC_TAIL_CAP] : ctx->seg_tail_cap;
        for (int c = 0; c < 2; ++c) {
            RecSide& sd = a.side[c];
            sd.eident = const_cast<u64*>((const u64*)side[c].eident);
            sd.rec = side[c].rec;
            sd.misfit_count = side[c].counters + parity;
            sd.misfit_list = side[c].counters + 64;
            sd.ident_off = (int)cks[c]->ident_off;
            sd.r = (int)cks[c]->r;
            sd.null_ord = cks[c]->slab_null;
            identity_segments(cks[c], th.nseg, &sd.seg_lo, &sd.seg_hi);
        }
        int64_t blocks = gf2_cdiv(now, 64);
        // one wavefront per workgroup, 17 KiB of LDS each: 9 fit a CU, 8 -- two per SIMD, and 2^22 / 64 tiles divide evenly among
        // 256 x 8 of them -- are faster (end to end 2.37 against 2.26 at 9 and 2.20 at 7 x 10^9 samples/s:
        // profiles/r04_mc_sampler_waves.log; the kernel is bound by what its wavefronts issue).  GF2_OPT_MC_SAMPLER_WAVES: 1..9
        const int64_t per_cu = ctx->opt[GF2_OPT_MC_SAMPLER_WAVES] > 0 ? ctx->opt[GF2_OPT_MC_SAMPLER_WAVES] : 8;
        if (blocks > (int64_t)ctx->num_cus * per_cu) blocks = (int64_t)ctx->num_cus * per_cu;
        GF2_TRY(gf2_prof_begin(ctx, GF2_K_SAMPLER));
        hipLaunchKernelGGL(slab_record_sampler_kernel, dim3((unsigned)blocks), dim3(64), 0, stream, a);
        GF2_TRY(gf2_prof_end(ctx));
        GF2_HIP(hipGetLastError());
        for (int c = 1; c >= 0; --c) {                                // Z first, then X
            const Side& sd = side[c];
            const bool carry = pending_of >= 0 && sd.can_carry;
            if (pending_of >= 0 && !carry) GF2_TRY(flush());
            GF2_TRY(launch_gather(ctx, cks[c], sd.eident, sd.rec, sd.pw, now, (int64_t)pad, lde, sd.fast, none, stream, nullptr, nullptr, 0,
                                  nullptr, carry ? &pending : nullptr));
            pending.cmb_pw = sd.pw;                                   // this component's own combine step is due now
            pending.cmb_positions = gf2_cdiv(now, 64) * 64;
            pending.cmb_pad = (int64_t)pad;
            pending.cmb_sample0 = 0;
            pending.cmb_nbins = (int)cks[c]->r + 1;
            pending.cmb_nslabs = cks[c]->nslabs512;
            pending.cmb_hist = (u64*)hists[c];
            pending.redo_count = sd.redo_count;
            pending.redo_list = sd.redo_count + 64;
            pending_of = c;
            pending_count = now;
        }
        MisfitArgs ma;
        for (int c = 0; c < 2; ++c) {
            ma.count[c] = side[c].counters;
            ma.list[c] = side[c].counters + 64;
            ma.side[c] = SparseSide{cks[c]->ht_dev, cks[c]->r, cks[c]->ident_off, (u64*)hists[c], (int)(cks[c]->r + 1)};
            ma.clear[c] = side[c].counters + (parity ^ 1);
        }
        hipLaunchKernelGGL(slab_misfit_kernel, dim3(128), dim3(64 * MISFIT_WAVES), 0, stream, ma, parity, (u64)seed, first_sample + done, th, n);
        GF2_HIP(hipGetLastError());
    }
    return flush();
}
