// Host-only part of libgf2hip.so: the thread-local error message and the packing of NumPy-style arrays (uint8 / int64, one entry
// per bit) into packed uint64 rows and back (gf2_pack_rows_*, gf2_unpack_rows_* of include/gf2hip.h: what css_code.py:39-44 and
// every np.mod(..., 2) of the reference do on dense arrays).  Plain C++, no HIP: `make tsan` / `make asan` build this translation
// unit alone for the CPU box (build/libgf2host_{tsan,asan}.so) and tests/test_host_sanitizers.py runs the packing round trips
// of tests/test_abi.py through them.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "gf2hip.h"

static thread_local char g_error[512] = "";

void gf2_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

extern "C" const char* gf2_last_error(void) { return g_error; }

#define GF2_FAIL(code, ...)          \
    do {                             \
        gf2_set_error(__VA_ARGS__);  \
        return (code);               \
    } while (0)

static inline int64_t gf2_words(int64_t bits) { return (bits + 63) >> 6; }

// Rows are independent: large arrays are cut into row ranges, one host thread each (a 2048 x 4096 int64 array is 64 MiB, more
// than one core streams in the time the elimination itself takes; gf2_rref on it spent 6 of its 7 ms here on one thread).
// At most GF2_HOST_THREADS threads (default 16), no more than the host's cores divided among the ranks that share it
// (LOCAL_WORLD_SIZE, as torch.distributed.run and bench.py's own launcher set it).  std::thread's constructor throws when the
// process or container is out of threads: nothing may leave an extern "C" entry point, so whatever could not be started runs
// on the calling thread, after the ranges that did start have been handed out, and everything started is joined.
static int64_t host_thread_cap() {
    static const int64_t cap = []() {                                   // read once
        const char* env = getenv("GF2_HOST_THREADS");
        const long v = env ? strtol(env, nullptr, 10) : 0;
        int64_t threads = v >= 1 && v <= 256 ? v : 16;
        const unsigned int hw = std::thread::hardware_concurrency();
        const char* lws = getenv("LOCAL_WORLD_SIZE");
        const long ranks = lws ? strtol(lws, nullptr, 10) : 1;
        int64_t share = hw ? (int64_t)hw / (ranks >= 1 && ranks <= 1024 ? ranks : 1) : 1;
        if (share < 1) share = 1;
        return threads < share ? threads : share;
    }();
    return cap;
}

#ifdef GF2_HOST_TEST_HOOKS
// sanitizer builds only (never in libgf2hip.so): makes the t-th thread creation of a call fail like an exhausted thread limit
static std::atomic<int> g_fail_after(-1);
extern "C" void gf2_host_test_fail_after(int started) { g_fail_after.store(started); }
#endif

template <typename F>
static void host_rows_parallel(int64_t rows, int64_t bytes_per_row, F body) {
    const int64_t total = rows * bytes_per_row;
    int64_t threads = host_thread_cap();
    if (threads > total >> 20) threads = total >> 20;                   // at least 1 MiB per thread
    if (threads > rows) threads = rows;
    if (threads <= 1) {
        body((int64_t)0, rows);
        return;
    }
    std::vector<std::thread> pool;
    const int64_t per = (rows + threads - 1) / threads;
    int64_t started_to = per < rows ? per : rows;                       // rows [0, per) are the caller's; [per, started_to) have a thread
    try {
        pool.reserve((size_t)threads);
        for (int64_t t = 1; t < threads; ++t) {
            const int64_t lo = t * per, hi = lo + per < rows ? lo + per : rows;
            if (lo >= hi) break;
#ifdef GF2_HOST_TEST_HOOKS
            if (g_fail_after >= 0 && t > g_fail_after) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
#endif
            pool.emplace_back([=]() { body(lo, hi); });
            started_to = hi;
        }
    } catch (const std::system_error&) {                                // out of threads: the rest is done here
    } catch (const std::bad_alloc&) {
    }
    body((int64_t)0, per < rows ? per : rows);
    if (started_to < rows) body(started_to, rows);
    for (auto& th : pool) th.join();
}

// `other_out` (may be null): set to 1 when some entry is not 0 or 1 -- css_code.py:39-44's "must be binary" test, made on the way
// through the array instead of in three further passes over it.
template <typename T>
static int pack_rows_host(const T* src, int64_t m, int64_t n, int64_t src_stride, uint64_t* dst, int64_t ld, int* other_out = nullptr) {
    if ((!src || !dst) && m > 0 && n > 0) GF2_FAIL(GF2_E_ARG, "pack: null buffer");
    if (m < 0 || n < 0 || ld < gf2_words(n) || src_stride < n) GF2_FAIL(GF2_E_ARG, "pack: bad shape");
    std::atomic<int> other(0);
    std::atomic<int>* const other_p = &other;
    host_rows_parallel(m, n * (int64_t)sizeof(T), [=](int64_t lo, int64_t hi) {
        T seen = 0;
        for (int64_t i = lo; i < hi; ++i) {
            const T* row = src + i * src_stride;
            uint64_t* out = dst + i * ld;
            for (int64_t w = 0; w < ld; ++w) {
                uint64_t acc = 0;
                const int64_t base = w * 64;
                const int64_t lim = n - base < 64 ? n - base : 64;
                for (int64_t b = 0; b < lim; ++b) {
                    acc |= (uint64_t)(row[base + b] & 1) << b;
                    seen |= row[base + b];
                }
                out[w] = acc;
            }
        }
        if (seen & ~(T)1) other_p->store(1, std::memory_order_relaxed);
    });
    if (other_out) *other_out = other.load();
    return GF2_OK;
}

template <typename T>
static int unpack_rows_host(const uint64_t* src, int64_t m, int64_t n, int64_t ld, T* dst, int64_t dst_stride) {
    if ((!src || !dst) && m > 0 && n > 0) GF2_FAIL(GF2_E_ARG, "unpack: null buffer");
    if (m < 0 || n < 0 || ld < gf2_words(n) || dst_stride < n) GF2_FAIL(GF2_E_ARG, "unpack: bad shape");
    host_rows_parallel(m, n * (int64_t)sizeof(T), [=](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const uint64_t* row = src + i * ld;
            T* out = dst + i * dst_stride;
            const int64_t full = n >> 6;
            for (int64_t w = 0; w < full; ++w) {                  /* whole words: a fixed-length loop the compiler vectorises */
                const uint64_t v = row[w];
                T* o = out + w * 64;
                for (int b = 0; b < 64; ++b) o[b] = (T)((v >> b) & 1);
            }
            for (int64_t j = full * 64; j < n; ++j) out[j] = (T)((row[j >> 6] >> (j & 63)) & 1);
        }
    });
    return GF2_OK;
}

extern "C" {

int gf2_pack_rows_u8(const uint8_t* src, int64_t m, int64_t n, int64_t src_stride, uint64_t* dst, int64_t ld) {
    return pack_rows_host<uint8_t>(src, m, n, src_stride, dst, ld);
}

int gf2_pack_rows_i64(const int64_t* src, int64_t m, int64_t n, int64_t src_stride, uint64_t* dst, int64_t ld) {
    return pack_rows_host<int64_t>(src, m, n, src_stride, dst, ld);
}

int gf2_pack_rows_binary_u8(const uint8_t* src, int64_t m, int64_t n, int64_t src_stride, uint64_t* dst, int64_t ld, int* other_out) {
    if (!other_out) GF2_FAIL(GF2_E_ARG, "pack: null output");
    return pack_rows_host<uint8_t>(src, m, n, src_stride, dst, ld, other_out);
}

int gf2_pack_rows_binary_i64(const int64_t* src, int64_t m, int64_t n, int64_t src_stride, uint64_t* dst, int64_t ld, int* other_out) {
    if (!other_out) GF2_FAIL(GF2_E_ARG, "pack: null output");
    return pack_rows_host<int64_t>(src, m, n, src_stride, dst, ld, other_out);
}

int gf2_unpack_rows_u8(const uint64_t* src, int64_t m, int64_t n, int64_t ld, uint8_t* dst, int64_t dst_stride) {
    return unpack_rows_host<uint8_t>(src, m, n, ld, dst, dst_stride);
}

int gf2_unpack_rows_i64(const uint64_t* src, int64_t m, int64_t n, int64_t ld, int64_t* dst, int64_t dst_stride) {
    return unpack_rows_host<int64_t>(src, m, n, ld, dst, dst_stride);
}

}  // extern "C"

