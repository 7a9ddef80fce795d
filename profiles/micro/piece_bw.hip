// Micro-benchmark (round 5): in-place update of a row-major matrix by workgroups that each own a COLUMN CHUNK of `piece` bytes over a
// row block -- the access pattern of the RREF's trailing pass -- for 128-byte and 64-byte chunks.  Does HBM serve 64-byte pieces
// 8 KiB apart as well as 128-byte ones?   hipcc --offload-arch=gfx950 -O3 piece_bw.hip -o piece_bw && ./piece_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int LPR>   // lanes per row (16 bytes each): 8 = 128-byte chunk, 4 = 64-byte chunk
__global__ __launch_bounds__(1024) void touch(u32x4* a, long m, long ld16, long rows_wg, int work) {
    const int tid = threadIdx.x, p = tid % LPR;
    const long chunk = blockIdx.y, r_lo = blockIdx.x * rows_wg, r_hi = r_lo + rows_wg < m ? r_lo + rows_wg : m;
    constexpr int RPW = 1024 / LPR;                 // rows per step of the workgroup
    for (long r = r_lo + tid / LPR; r < r_hi; r += 4 * RPW) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (r + u * RPW < r_hi) v[u] = a[(r + u * RPW) * ld16 + chunk * LPR + p];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            u32x4 x = v[u];
            for (int k = 0; k < work; ++k) x = x * 3u + 1u;    // stand-in for the lookups' time
            if (r + u * RPW < r_hi) a[(r + u * RPW) * ld16 + chunk * LPR + p] = x;
        }
    }
}
int main() {
    const long m = 32768, ld = 1024, ld16 = ld / 2;            // 256 MiB
    u32x4* a;
    hipMalloc(&a, m * ld * 8);
    hipMemset(a, 1, m * ld * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int work : {0, 40, 80}) for (int blocks : {3, 4}) for (int lpr : {8, 4}) {
        const long chunks = ld16 / lpr, rows_wg = (m + blocks - 1) / blocks;
        // as many chunks as keep the grid at about one round: 63 (128-byte) / 126 (64-byte) chunk columns would be two rounds for the
        // narrow ones, so the narrow grid takes half the row block count... here simply: every chunk, `blocks` row blocks, several rounds
        dim3 grid((unsigned)blocks, (unsigned)chunks);
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (lpr == 8) hipLaunchKernelGGL(touch<8>, grid, dim3(1024), 0, 0, a, m, ld16, rows_wg, work);
            else hipLaunchKernelGGL(touch<4>, grid, dim3(1024), 0, 0, a, m, ld16, rows_wg, work);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("work %2d  row blocks %d  chunk %3d B (%4ld chunk columns): %.3f ms = %.2f TB/s read + write\n", work, blocks, lpr * 16, chunks, best,
               2.0 * m * ld * 8 / best / 1e9);
    }
    return 0;
}
