// Micro-test (round 5): do CU-masked streams (hipExtStreamCreateWithCUMask) confine workgroups, and how do mask bits map to XCDs?
// Each workgroup (1024 lanes, 128 KiB of LDS: one per CU) records its XCC and HW_ID; the host prints how many distinct CUs each
// XCD used for a few masks, and whether two kernels on disjoint masks run side by side.
//   hipcc --offload-arch=gfx950 -O3 cu_mask.hip -o cu_mask && ./cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <set>
#include <map>
__global__ __launch_bounds__(1024) void where(unsigned* out, int spin) {
    extern __shared__ unsigned pad[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; pad[0] = hw; }
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin * 100ull) { }       // spin microseconds (100 MHz clock)
}
static void report(const char* what, const std::vector<unsigned>& v, int n) {
    std::map<unsigned, std::set<unsigned>> cus;
    for (int i = 0; i < n; ++i) {
        const unsigned hw = v[2 * i], xcc = v[2 * i + 1] & 15u;
        cus[xcc].insert((hw >> 8) & 0xFFu);                                    // cu_id [11:8], sh_id [12], se_id [15:13]
    }
    printf("%s: %d workgroups;", what, n);
    for (auto& kv : cus) printf(" xcc%u:%zu", kv.first, kv.second.size());
    printf("\n");
}
int main() {
    hipFuncSetAttribute((const void*)where, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    unsigned* out; hipMalloc(&out, 4096 * 8);
    std::vector<unsigned> host(8192);
    const int words = 8;                                                      // 256 bits
    auto run_mask = [&](const char* what, const std::vector<unsigned>& mask, int wgs) {
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data()) != hipSuccess) { printf("%s: mask refused\n", what); return; }
        hipMemsetAsync(out, 0, 4096 * 8, s);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(where, dim3(wgs), dim3(1024), 128 * 1024, s, out, 50);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(host.data(), out, wgs * 8, hipMemcpyDeviceToHost);
        char buf[160]; snprintf(buf, sizeof buf, "%s (%.0f us for 50 us of spinning per workgroup)", what, ms * 1000);
        report(buf, host, wgs);
        hipStreamDestroy(s);
    };
    std::vector<unsigned> all(words, 0xFFFFFFFFu), low8(words, 0), low32(words, 0), rest(words, 0xFFFFFFFFu), every8(words, 0);
    low8[0] = 0xFFu; low32[0] = 0xFFFFFFFFu; rest[0] = 0xFFFFFF00u;
    for (int w = 0; w < words; ++w) every8[w] = 0x01010101u;                   // bits 0, 8, 16, ...
    run_mask("all 256 bits, 256 workgroups", all, 256);
    run_mask("bits 0-7, 8 workgroups", low8, 8);
    run_mask("bits 0-7, 16 workgroups", low8, 16);
    run_mask("bits 0-31, 32 workgroups", low32, 32);
    run_mask("bits 8-255, 248 workgroups", rest, 248);
    run_mask("bits 8-255, 256 workgroups", rest, 256);
    run_mask("every 8th bit, 32 workgroups", every8, 32);
    // two kernels side by side on disjoint masks: 248 long workgroups on bits 8-255, then 8 short ones on bits 0-7
    hipStream_t big, small;
    hipExtStreamCreateWithCUMask(&big, words, rest.data());
    hipExtStreamCreateWithCUMask(&small, words, low8.data());
    unsigned* out2; hipMalloc(&out2, 4096 * 8);
    hipEvent_t b0, b1, s0, s1; hipEventCreate(&b0); hipEventCreate(&b1); hipEventCreate(&s0); hipEventCreate(&s1);
    hipDeviceSynchronize();
    hipEventRecord(b0, big);
    hipLaunchKernelGGL(where, dim3(248), dim3(1024), 128 * 1024, big, out, 300);
    hipEventRecord(b1, big);
    hipEventRecord(s0, small);
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(where, dim3(8), dim3(1024), 64 * 1024, small, out2, 20);
    hipEventRecord(s1, small);
    hipDeviceSynchronize();
    float tb, ts; hipEventElapsedTime(&tb, b0, b1); hipEventElapsedTime(&ts, s0, s1);
    printf("side by side: 248 x 300 us on bits 8-255 took %.0f us; 4 launches of 8 x 20 us on bits 0-7 took %.0f us\n", tb * 1000, ts * 1000);
    // how many one-CU workgroups fit ONE round of a masked stream?  (248 on 248 CUs took two above)
    {
        std::vector<unsigned> rest16(words, 0xFFFFFFFFu);
        rest16[0] = 0xFFFF0000u;
        hipStream_t big16;
        hipExtStreamCreateWithCUMask(&big16, words, rest16.data());
        const int counts[] = {200, 216, 224, 232, 236, 240, 244, 248};
        for (int pass = 0; pass < 2; ++pass)
            for (int wgs : counts) {
                hipStream_t on = pass == 0 ? big : big16;
                if (pass == 1 && wgs > 240) continue;
                hipDeviceSynchronize();
                hipEventRecord(b0, on);
                hipLaunchKernelGGL(where, dim3(wgs), dim3(1024), 128 * 1024, on, out, 300);
                hipEventRecord(b1, on);
                hipDeviceSynchronize();
                hipEventElapsedTime(&tb, b0, b1);
                hipMemcpy(host.data(), out, wgs * 8, hipMemcpyDeviceToHost);
                char buf[128]; snprintf(buf, sizeof buf, "%s, %d x 300 us: %.0f us", pass == 0 ? "bits 8-255" : "bits 16-255", wgs, tb * 1000);
                report(buf, host, wgs);
            }
        // unmasked, for comparison
        for (int wgs : {216, 232, 240, 248, 256}) {
            hipDeviceSynchronize();
            hipEventRecord(b0, 0);
            hipLaunchKernelGGL(where, dim3(wgs), dim3(1024), 128 * 1024, 0, out, 300);
            hipEventRecord(b1, 0);
            hipDeviceSynchronize();
            hipEventElapsedTime(&tb, b0, b1);
            hipMemcpy(host.data(), out, wgs * 8, hipMemcpyDeviceToHost);
            char buf[128]; snprintf(buf, sizeof buf, "no mask, %d x 300 us: %.0f us", wgs, tb * 1000);
            report(buf, host, wgs);
        }
    }
    // the same without masks: the small kernels queue behind the big one's workgroups?
    hipStream_t p, q; hipStreamCreateWithFlags(&p, hipStreamNonBlocking); hipStreamCreateWithPriority(&q, hipStreamNonBlocking, -1);
    hipDeviceSynchronize();
    hipEventRecord(b0, p);
    hipLaunchKernelGGL(where, dim3(248), dim3(1024), 128 * 1024, p, out, 300);
    hipEventRecord(b1, p);
    hipEventRecord(s0, q);
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(where, dim3(8), dim3(1024), 64 * 1024, q, out2, 20);
    hipEventRecord(s1, q);
    hipDeviceSynchronize();
    hipEventElapsedTime(&tb, b0, b1); hipEventElapsedTime(&ts, s0, s1);
    printf("no masks:     248 x 300 us took %.0f us; 4 launches of 8 x 20 us on a high-priority stream took %.0f us\n", tb * 1000, ts * 1000);
    return 0;
}
