// Device helpers shared by the sparse-error kernels (gf2_sparse.hip, gf2_slabs.hip); gfx950 only.
#pragma once

#include "gf2_internal.h"

#define SPARSE_LIST_CAP 512

// Sum over the 64 lanes (returned uniformly): DPP butterfly inside each row of 16, then row broadcasts.
__device__ __forceinline__ unsigned int wave_total(unsigned int v) {
    v += __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0u, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0u, v, 0x140, 0xF, 0xF, true);   // row_mirror
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xA, 0xF, true);   // row_bcast15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xC, 0xF, true);   // row_bcast31 into rows 2 and 3
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ u64 ident_mask(int64_t ident_off, int64_t r, int64_t word) {
    if (ident_off < 0) return 0ull;
    const int64_t lo = ident_off - word * 64, hi = ident_off + r - word * 64;
    if (hi <= 0 || lo >= 64) return 0ull;
    u64 m = ~0ull;
    if (lo > 0) m &= ~0ull << lo;
    if (hi < 64) m &= ~(~0ull << hi);
    return m;
}

// One parity check as the wavefront-per-sample routines see it.
struct SparseSide {
    const uint32_t* ht;        // transposed check, 64 dwords per column, column n is zero
    int64_t r, ident_off;
    u64* hist;
    int nbins;
};

// Returns the syndrome's weight; *syn_out (if given) = this lane's syndrome dword (rows 32 lane .. 32 lane + 31, zero past r).
__device__ __forceinline__ unsigned int sparse_component_weight(u64 w, const SparseSide& side, int64_t n, int lane,
                                                               unsigned int* mylist, unsigned int* syn_out = nullptr) {
    // identity block: dword `lane` covers rows 32*lane.. <-> error bits ident_off + 32*lane ..
    unsigned int acc = 0;
    if (side.ident_off >= 0) {
        const int64_t bit = side.ident_off + 32 * (int64_t)lane;
        const int src = (int)(bit >> 6), sh = (int)(bit & 63);
        const u64 lo = __shfl(w, src & 63), hi = __shfl(w, (src + 1) & 63);
        u64 v = src < 64 ? lo >> sh : 0ull;
        if (sh && src + 1 < 64) v |= hi << (64 - sh);
        const int64_t row0 = 32 * (int64_t)lane;
        unsigned int keep = row0 < side.r ? (side.r - row0 < 32 ? ~(~0u << (side.r - row0)) : ~0u) : 0u;
        acc = (unsigned int)v & keep;
        w &= ~ident_mask(side.ident_off, side.r, lane);
    }
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
            if (pos < SPARSE_LIST_CAP) mylist[pos] = (unsigned int)((lane << 6) + b);
        }
        total += (unsigned int)__popcll(active);
    }
    const char* htb = reinterpret_cast<const char*>(side.ht);
    const unsigned int lane4 = lane * 4u;
    if (total && total <= SPARSE_LIST_CAP) {
        if (lane < 8) mylist[total + lane] = (unsigned int)n;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (unsigned int k0 = 0; k0 < total; k0 += 8) {
            unsigned int v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const uint32_t*>(htb + ((mylist[k0 + i] << 8) | lane4));
#pragma unroll
            for (int i = 0; i < 8; ++i) acc ^= v[i];
        }
        __builtin_amdgcn_wave_barrier();
    } else if (total) {                                             // dense sample: walk the words one by one
        u64 nz = __ballot(w != 0);
        while (nz) {
            const int src = __ffsll((long long)nz) - 1;
            nz &= nz - 1;
            u64 word = ((u64)(unsigned int)__builtin_amdgcn_readlane((int)(w >> 32), src) << 32) |
                       (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)w, src);
            while (word) {
                const int b = __ffsll((long long)word) - 1;
                word &= word - 1;
                acc ^= *reinterpret_cast<const uint32_t*>(htb + (((unsigned int)((src << 6) + b) << 8) | lane4));
            }
        }
    }
    if (syn_out) *syn_out = acc;
    return wave_total((unsigned int)__popc(acc));
}

