// Micro-benchmark: streaming 2^20 rows of 512 bytes, reading the whole row / one 256-byte half / 64-byte pieces.
// Build: hipcc --offload-arch=gfx950 -O3 halfrow_read.hip -o halfrow_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;

// lane = 8-byte word; `first`..`first+count` words of every row are read (count <= 64)
__global__ void read_rows(const u64* __restrict__ e, int64_t rows, int first, int count, u64* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    u64 acc = 0;
    for (int64_t r0 = wave * 8; r0 < rows; r0 += nwaves * 8) {
        u64 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0;
        if (lane >= first && lane < first + count) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = e[(r0 + j) * 64 + lane];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
    }
    if (acc == 0x123456789ull) sink[0] = acc;
}

int main() {
    const int64_t rows = 1 << 20;
    u64 *e, *sink;
    hipMalloc(&e, rows * 512);
    hipMalloc(&sink, 8);
    hipMemset(e, 1, rows * 512);
    hipEvent_t t0, t1;
    hipEventCreate(&t0);
    hipEventCreate(&t1);
    const int cases[][2] = {{0, 64}, {0, 32}, {32, 32}, {0, 16}, {16, 16}, {0, 8}, {24, 8}, {0, 33}};
    for (auto& c : cases) {
        for (int rep = 0; rep < 3; ++rep) read_rows<<<256 * 8, 256>>>(e, rows, c[0], c[1], sink);
        hipEventRecord(t0);
        for (int rep = 0; rep < 10; ++rep) read_rows<<<256 * 8, 256>>>(e, rows, c[0], c[1], sink);
        hipEventRecord(t1);
        hipEventSynchronize(t1);
        float ms;
        hipEventElapsedTime(&ms, t0, t1);
        ms /= 10;
        printf("words [%2d, %2d) of every 512-byte row: %.1f us, %.0f GB/s of requested bytes\n", c[0], c[0] + c[1], ms * 1e3,
               rows * c[1] * 8.0 / ms / 1e6);
    }
    return 0;
}
