// css_code.transform_stabilisers (css_code.py:737-781) on packed stabiliser matrices (gfx950).
//
// The reference walks a gate list and rewrites a k x 2n matrix [X | Z] column by column: H on qubit q swaps the X and Z
// entry of q in every row (and refuses a row that has both: css_code.py:762-763), CNOT(c, t) adds X_c into X_t and Z_t
// into Z_c in every row (css_code.py:775-781).  Gates depend on each other, rows do not.  So: lane = row, a wavefront
// owns 64 rows for the whole gate list and keeps them in LDS as [word][row] (consecutive lanes on consecutive banks,
// every access conflict-free, 64 KiB at n = 4096); the gate stream is wave-uniform (64 gates per vector load, handed out
// with readlane), each gate is two to four LDS word accesses and a few bit operations.  One block per 64 rows; nothing
// else is shared.
//
// A gate the reference would refuse ends the walk: the lowest such gate index over all blocks is reported, and the
// host reruns the accepted prefix so that the matrix it hands back is the state the reference leaves behind before
// the refused gate.
//
// Runs of CNOTs with one control are folded on the host before they reach the kernel: CNOT(c, t_1), CNOT(c, t_2), ... commute
// (X_c and the Z_t do not change during the run), and all targets inside one 64-column word w are one operation,
//     X_w ^= X_c ? mask : 0,     Z_c ^= parity(Z_w & mask),
// a "word gate" (kind 2).  The encoders of css_code.py:203-312 are an H on a generator's pivot followed by CNOTs from it to
// every other qubit of the generator: 2.1 million gates at n = 4096 become about 130 thousand.
#include <algorithm>
#include <vector>

#include "gf2_internal.h"

#define CNJ_MAX_WORDS 320                      // words of 2n bits: 320 * 64 lanes * 8 bytes = 160 KiB of LDS

// gates: ngates rows of CNJ_GATE_INTS ints (kind, a, b, mask lo, mask hi, index in the caller's list): kind 0 = H on a,
// 1 = CNOT control a target b, 2 = CNOT control a onto the targets `mask` of word b of the X half; all indices validated.
#define CNJ_GATE_INTS 6
__global__ __launch_bounds__(64) void conjugate_kernel(u64* __restrict__ mat, int64_t k, int n, int64_t ld, int words,
                                                       const int* __restrict__ gates, int64_t ngates,
                                                       unsigned long long* __restrict__ first_refused) {
    extern __shared__ u64 rows_lds[];                               // [word][lane]
    const int lane = threadIdx.x;
    const int64_t row = (int64_t)blockIdx.x * 64 + lane;
    for (int w = 0; w < words; ++w) rows_lds[w * 64 + lane] = row < k ? mat[row * ld + w] : 0ull;
    // Gates arrive 64 at a time through vector loads (lane l holds gate g0 + l; the next 64 are in flight meanwhile) and are
    // handed out with readlane: scalar loads would share the LDS's wait counter and put a memory round trip into every
    // step of what is already a chain of LDS round trips.
    int kind_v = 1, a_v = 0, b_v = 0, ml_v = 0, mh_v = 0, at_v = 0;
    if (lane < ngates) {
        const int* g = gates + CNJ_GATE_INTS * lane;
        kind_v = g[0], a_v = g[1], b_v = g[2], ml_v = g[3], mh_v = g[4], at_v = g[5];
    }
    bool refused = false;
    for (int64_t g0 = 0; g0 < ngates && !refused; g0 += 64) {
        const int kind_c = kind_v, a_c = a_v, b_c = b_v, ml_c = ml_v, mh_c = mh_v, at_c = at_v;
        if (g0 + 64 + lane < ngates) {
            const int* g = gates + CNJ_GATE_INTS * (g0 + 64 + lane);
            kind_v = g[0], a_v = g[1], b_v = g[2], ml_v = g[3], mh_v = g[4], at_v = g[5];
        }
        const int cnt = ngates - g0 < 64 ? (int)(ngates - g0) : 64;
        for (int i = 0; i < cnt; ++i) {
            const int kind = __builtin_amdgcn_readlane(kind_c, i), a = __builtin_amdgcn_readlane(a_c, i),
                      b = __builtin_amdgcn_readlane(b_c, i);
            if (kind == 0) {
                const int zq = n + a;
                const int wx = a >> 6, bx = a & 63, wz = zq >> 6, bz = zq & 63;
                const u64 x = rows_lds[wx * 64 + lane], z = rows_lds[wz * 64 + lane];
                const u64 xb = (x >> bx) & 1ull, zb = (z >> bz) & 1ull;
                if (__ballot(xb & zb)) {                            // some row carries a Y on this qubit
                    if (lane == 0) atomicMin(first_refused, (unsigned long long)(unsigned int)__builtin_amdgcn_readlane(at_c, i));
                    refused = true;
                    break;
                }
                const u64 d = xb ^ zb;
                if (wx != wz) {
                    rows_lds[wx * 64 + lane] = x ^ (d << bx);
                    rows_lds[wz * 64 + lane] = z ^ (d << bz);
                } else {
                    rows_lds[wx * 64 + lane] = x ^ (d << bx) ^ (d << bz);
                }
            } else if (kind == 2) {
                // one control, the targets `mask` of X word b: X_b ^= X_a ? mask : 0; Z_a ^= parity(Z columns of the targets)
                const u64 mask = ((u64)(unsigned int)__builtin_amdgcn_readlane(mh_c, i) << 32) |
                                 (unsigned int)__builtin_amdgcn_readlane(ml_c, i);
                const int zc = n + a, z0 = n + 64 * b, sh = z0 & 63, wz0 = z0 >> 6;
                const u64 xa = (rows_lds[(a >> 6) * 64 + lane] >> (a & 63)) & 1ull;
                u64 zsel = rows_lds[wz0 * 64 + lane] & (mask << sh);
                if (sh && wz0 + 1 < words) zsel ^= rows_lds[(wz0 + 1) * 64 + lane] & (mask >> (64 - sh));
                rows_lds[b * 64 + lane] ^= (0ull - xa) & mask;
                rows_lds[(zc >> 6) * 64 + lane] ^= (u64)(__popcll(zsel) & 1) << (zc & 63);
            } else {
                const int zc = n + a, zt = n + b;
                const int w_xc = a >> 6, w_xt = b >> 6, w_zt = zt >> 6, w_zc = zc >> 6;
                if (w_xt != w_zt && w_xt != w_zc && w_xc != w_zc) {
                    // the X and the Z update touch different words: all four reads go out together
                    const u64 xc = rows_lds[w_xc * 64 + lane], xt = rows_lds[w_xt * 64 + lane];
                    const u64 ztw = rows_lds[w_zt * 64 + lane], zcw = rows_lds[w_zc * 64 + lane];
                    rows_lds[w_xt * 64 + lane] = xt ^ (((xc >> (a & 63)) & 1ull) << (b & 63));     // X: control -> target
                    rows_lds[w_zc * 64 + lane] = zcw ^ (((ztw >> (zt & 63)) & 1ull) << (zc & 63));  // Z: target -> control
                } else {
                    const u64 xc = (rows_lds[w_xc * 64 + lane] >> (a & 63)) & 1ull;
                    rows_lds[w_xt * 64 + lane] ^= xc << (b & 63);
                    const u64 ztb = (rows_lds[w_zt * 64 + lane] >> (zt & 63)) & 1ull;
                    rows_lds[w_zc * 64 + lane] ^= ztb << (zc & 63);
                }
            }
        }
    }
    if (row < k)
        for (int w = 0; w < words; ++w) mat[row * ld + w] = rows_lds[w * 64 + lane];
}

extern "C" int gf2_conjugate_gates(gf2_ctx* ctx, uint64_t* mat, int64_t k, int64_t n, int64_t ld, const int32_t* gates,
                                   int64_t ngates, int64_t* stop_out) {
    if (!ctx || !stop_out) GF2_FAIL(GF2_E_ARG, "gf2_conjugate_gates: null argument");
    *stop_out = -1;
    if (k < 0 || n < 0 || ngates < 0) GF2_FAIL(GF2_E_ARG, "gf2_conjugate_gates: negative size");
    const int64_t words = gf2_words(2 * n);
    if (ld < words) GF2_FAIL(GF2_E_ARG, "gf2_conjugate_gates: ld too small for 2n = %lld columns", (long long)(2 * n));
    if (words > CNJ_MAX_WORDS) GF2_FAIL(GF2_E_ARG, "gf2_conjugate_gates: 2n = %lld columns exceed %d", (long long)(2 * n), CNJ_MAX_WORDS * 64);
    if ((k > 0 && !mat) || (ngates > 0 && !gates)) GF2_FAIL(GF2_E_ARG, "gf2_conjugate_gates: null data");
    // the reference checks a gate when it reaches it (css_code.py:747-755): everything before the first bad one applies
    int64_t limit = ngates;
    for (int64_t g = 0; g < ngates; ++g) {
        const int32_t kind = gates[3 * g], a = gates[3 * g + 1], b = gates[3 * g + 2];
        const bool bad = (kind != 0 && kind != 1) || a < 0 || a >= n || (kind == 1 && (b < 0 || b >= n));
        if (bad) {
            limit = g;
            break;
        }
    }
    // Fold runs of CNOTs with one control into word gates (see the top of the file).  A run ends at an H, at another control,
    // at a CNOT whose target is the run's control and at a target inside the control's own X word (X_c must not change
    // during the run; the single-gate path takes such gates one by one).  Every entry carries the index of its first gate in
    // the caller's list: the device reports a refused H by it, and a prefix of the caller's list is a prefix of this one.
    std::vector<int32_t> folded;
    folded.reserve((size_t)(limit < 4096 ? limit : limit / 4) * CNJ_GATE_INTS);
    auto push = [&](int32_t kind, int32_t a, int32_t b, uint64_t mask, int64_t at) {
        const int32_t e[CNJ_GATE_INTS] = {kind, a, b, (int32_t)(uint32_t)mask, (int32_t)(uint32_t)(mask >> 32), (int32_t)at};
        folded.insert(folded.end(), e, e + CNJ_GATE_INTS);
    };
    std::vector<int64_t> entry_first;                                  // first caller gate of folded entry i (for prefix replays)
    {
        std::vector<uint64_t> run_mask((size_t)gf2_words(n) + 1, 0);
        std::vector<int32_t> run_words;
        int32_t run_c = -1;
        int64_t run_at = 0;
        auto flush = [&]() {
            for (int32_t w : run_words) {
                if (run_mask[w]) {
                    push(2, run_c, w, run_mask[w], run_at);
                    entry_first.push_back(run_at);
                }
                run_mask[w] = 0;
            }
            run_words.clear();
            run_c = -1;
        };
        for (int64_t g = 0; g < limit; ++g) {
            const int32_t kind = gates[3 * g], a = gates[3 * g + 1], b = gates[3 * g + 2];
            const bool foldable = kind == 1 && (b >> 6) != (a >> 6);
            if (!foldable || a != run_c) flush();
            if (foldable) {
                if (run_c < 0) run_c = a, run_at = g;
                if (!run_mask[b >> 6]) {
                    bool listed = false;
                    for (int32_t w : run_words) listed = listed || w == (b >> 6);
                    if (!listed) run_words.push_back(b >> 6);
                }
                run_mask[b >> 6] ^= 1ull << (b & 63);                  // a target named twice cancels
            } else {
                push(kind, a, b, 0, g);
                entry_first.push_back(g);
            }
        }
        flush();
    }
    const int64_t nfolded = (int64_t)entry_first.size();
    int rc = GF2_OK;
    int64_t refused = -1;
    if (k > 0 && words > 0 && limit > 0) {
        GF2_TRY(gf2_ctx_activate(ctx));
        if (!ctx->lds_optin[3]) {
            GF2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conjugate_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, CNJ_MAX_WORDS * 512));
            ctx->lds_optin[3] = true;
        }
        const size_t mat_bytes = (size_t)k * ld * 8, gate_bytes = (size_t)nfolded * CNJ_GATE_INTS * 4;
        u64 *work = nullptr, *orig = nullptr;
        int* gates_dev = nullptr;
        unsigned long long* flag_dev = nullptr;
        rc = gf2_dev_alloc(ctx, mat_bytes, (void**)&work);
        if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, mat_bytes, (void**)&orig);
        if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, gate_bytes, (void**)&gates_dev);
        if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, 8, (void**)&flag_dev);
        if (rc == GF2_OK) {
            const unsigned long long none = ~0ull;
            unsigned long long flag = none;
            const dim3 grid((unsigned)gf2_cdiv(k, 64)), block(64);
            const size_t lds = (size_t)words * 512;
            bool ok = hipMemcpyAsync(orig, mat, mat_bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                      hipMemcpyAsync(work, orig, mat_bytes, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
                      hipMemcpyAsync(gates_dev, folded.data(), gate_bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                      hipMemcpyAsync(flag_dev, &none, 8, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
            if (ok) {
                hipLaunchKernelGGL(conjugate_kernel, grid, block, lds, ctx->stream, work, k, (int)n, ld, (int)words, gates_dev,
                                   nfolded, flag_dev);
                ok = hipGetLastError() == hipSuccess &&
                     hipMemcpyAsync(&flag, flag_dev, 8, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                     hipStreamSynchronize(ctx->stream) == hipSuccess;
            }
            if (ok && flag != none) {                               // replay the accepted prefix on the original
                refused = (int64_t)flag;
                ok = hipMemcpyAsync(work, orig, mat_bytes, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess;
                // (a refused gate is an H, and an H ends a run: the entries that start before it are exactly the caller's prefix)
                const int64_t prefix = std::lower_bound(entry_first.begin(), entry_first.end(), refused) - entry_first.begin();
                if (ok && prefix > 0) {
                    hipLaunchKernelGGL(conjugate_kernel, grid, block, lds, ctx->stream, work, k, (int)n, ld, (int)words,
                                       gates_dev, prefix, flag_dev);
                    ok = hipGetLastError() == hipSuccess;
                }
            }
            if (ok)
                ok = hipMemcpyAsync(mat, work, mat_bytes, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                     hipStreamSynchronize(ctx->stream) == hipSuccess;
            if (!ok) {
                gf2_set_error("gf2_conjugate_gates: HIP call failed: %s", hipGetErrorString(hipGetLastError()));
                rc = GF2_E_HIP;
            }
        }
        (void)gf2_dev_free(ctx, work);
        (void)gf2_dev_free(ctx, orig);
        (void)gf2_dev_free(ctx, gates_dev);
        (void)gf2_dev_free(ctx, flag_dev);
        if (rc != GF2_OK) return rc;
    }
    if (refused >= 0) {
        *stop_out = refused;
        GF2_FAIL(GF2_E_NOTCSS, "gf2_conjugate_gates: gate %lld is an H on a qubit where a row has both X and Z", (long long)refused);
    }
    if (limit < ngates) {
        *stop_out = limit;
        GF2_FAIL(GF2_E_ARG, "gf2_conjugate_gates: gate %lld has an unknown kind or a qubit outside [0, n)", (long long)limit);
    }
    return GF2_OK;
}
