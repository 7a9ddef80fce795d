// Context, device memory, timing and host-side bit packing of libgf2hip.so.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <time.h>

#include "gf2_internal.h"

// Waits for a stream.  hipStreamSynchronize blocks on an interrupt after a short poll and was seen to wake up milliseconds
// late (a 12 ms gf2_mc_run returned after 20 ms every other call): poll hipStreamQuery for the first 50 ms instead, then block.
int gf2_stream_wait(hipStream_t stream) {
    timespec t0, t;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned int spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) return GF2_OK;
        if (q != hipErrorNotReady) GF2_HIP(q);
        if ((spins & 63u) == 63u) {
            clock_gettime(CLOCK_MONOTONIC, &t);
            if ((t.tv_sec - t0.tv_sec) * 1000000000ll + (t.tv_nsec - t0.tv_nsec) > 50000000ll) break;
        }
    }
    GF2_HIP(hipStreamSynchronize(stream));
    return GF2_OK;
}

// Streams `vecs` 16-byte pieces: four independent loads per lane and trip, XOR-ed into one word per workgroup (so that
// nothing is optimised away); COPY also stores them.  The measured ceiling that roofline fractions are quoted against next to
// the 8 TB/s specification.
typedef unsigned int probe_vec __attribute__((ext_vector_type(4)));
template <bool COPY>
__global__ __launch_bounds__(512) void membw_probe_kernel(const probe_vec* __restrict__ src, probe_vec* __restrict__ dst, int64_t vecs,
                                                          u64* __restrict__ sink) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    probe_vec acc = {0, 0, 0, 0};
    for (; i + 3 * stride < vecs; i += 4 * stride) {
        const probe_vec a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        const probe_vec c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        if (COPY) {
            __builtin_nontemporal_store(a, dst + i);
            __builtin_nontemporal_store(b, dst + i + stride);
            __builtin_nontemporal_store(c, dst + i + 2 * stride);
            __builtin_nontemporal_store(d, dst + i + 3 * stride);
        }
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < vecs; i += stride) {
        const probe_vec a = src[i];
        if (COPY) dst[i] = a;
        acc ^= a;
    }
    const unsigned int folded = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (folded == 0x9E3779B9u) atomicXor(sink, (u64)folded);       // keeps the loads alive; practically never taken
}

// The context's streams and events.  On failure the caller destroys whatever was created (every handle starts out null).
static int create_streams_and_events(gf2_ctx* ctx) {
    GF2_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) GF2_HIP(hipStreamCreateWithFlags(&ctx->side[k], hipStreamNonBlocking));
    {
        int least = 0, greatest = 0;                               // (numerically lower = higher priority)
        GF2_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        GF2_HIP(hipStreamCreateWithPriority(&ctx->hi, hipStreamNonBlocking, greatest));
    }
    for (int k = 0; k < 7; ++k) GF2_HIP(hipEventCreateWithFlags(&ctx->side_ev[k], hipEventDisableTiming));
    GF2_HIP(hipEventCreate(&ctx->t0));
    GF2_HIP(hipEventCreate(&ctx->t1));
    for (int i = 0; i < gf2_ctx::kProfSlots; ++i) {
        GF2_HIP(hipEventCreate(&ctx->prof_ev[i][0]));
        GF2_HIP(hipEventCreate(&ctx->prof_ev[i][1]));
    }
    return GF2_OK;
}

static void destroy_streams_and_events(gf2_ctx* ctx) {
    for (int i = 0; i < gf2_ctx::kProfSlots; ++i) {
        if (ctx->prof_ev[i][0]) (void)hipEventDestroy(ctx->prof_ev[i][0]);
        if (ctx->prof_ev[i][1]) (void)hipEventDestroy(ctx->prof_ev[i][1]);
    }
    if (ctx->t0) (void)hipEventDestroy(ctx->t0);
    if (ctx->t1) (void)hipEventDestroy(ctx->t1);
    for (int k = 0; k < 7; ++k)
        if (ctx->side_ev[k]) (void)hipEventDestroy(ctx->side_ev[k]);
    for (int k = 0; k < 2; ++k)
        if (ctx->side[k]) (void)hipStreamDestroy(ctx->side[k]);
    if (ctx->hi) (void)hipStreamDestroy(ctx->hi);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
}

extern "C" {

int gf2_version(void) { return GF2_VERSION_NUMBER; }

int gf2_device_count(int* count_out) {
    if (!count_out) GF2_FAIL(GF2_E_ARG, "gf2_device_count: null output");
    int count = 0;
    hipError_t err = hipGetDeviceCount(&count);
    if (err != hipSuccess) {
        *count_out = 0;
        GF2_FAIL(GF2_E_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(err));
    }
    *count_out = count;
    return GF2_OK;
}

int gf2_ctx_create(int device, gf2_ctx** ctx_out) {
    if (!ctx_out) GF2_FAIL(GF2_E_ARG, "gf2_ctx_create: null output");
    *ctx_out = nullptr;
    int count = 0;
    GF2_TRY(gf2_device_count(&count));
    if (device < 0 || device >= count)
        GF2_FAIL(GF2_E_HIP, "gf2_ctx_create: device %d not available (%d visible)", device, count);
    GF2_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    GF2_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        GF2_FAIL(GF2_E_HIP, "gf2_ctx_create: device %d is %s; this library is built for gfx950 only",
                 device, prop.gcnArchName);
    gf2_ctx* ctx = (gf2_ctx*)calloc(1, sizeof(gf2_ctx));
    if (!ctx) GF2_FAIL(GF2_E_NOMEM, "gf2_ctx_create: out of host memory");
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    for (int k = 0; k < GF2_OPT_COUNT; ++k) ctx->opt[k] = -1;
    const char* env_flags = getenv("GF2_FLAGS");                    // read once, here: the initial routing flags
    ctx->flags = env_flags ? (uint32_t)strtoul(env_flags, nullptr, 0) & GF2_F_ALL : 0u;     // unknown bits are dropped
    const int rc = create_streams_and_events(ctx);
    if (rc != GF2_OK) {                                             // the message of the failing call stays in g_error
        destroy_streams_and_events(ctx);
        free(ctx);
        return rc;
    }
    *ctx_out = ctx;
    return GF2_OK;
}

int gf2_ctx_set_flags(gf2_ctx* ctx, uint32_t flags) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_ctx_set_flags: null context");
    if (flags & ~GF2_F_ALL) GF2_FAIL(GF2_E_ARG, "gf2_ctx_set_flags: unknown flag bits 0x%x", flags & ~GF2_F_ALL);
    ctx->flags = flags;
    return GF2_OK;
}

int gf2_ctx_get_flags(gf2_ctx* ctx, uint32_t* flags_out) {
    if (!ctx || !flags_out) GF2_FAIL(GF2_E_ARG, "gf2_ctx_get_flags: null argument");
    *flags_out = ctx->flags;
    return GF2_OK;
}

int gf2_ctx_set_option(gf2_ctx* ctx, int option, int64_t value) {
    if (!ctx || option < 0 || option >= GF2_OPT_COUNT) GF2_FAIL(GF2_E_ARG, "gf2_ctx_set_option: bad argument");
    if (value >= 0) {
        if (option == GF2_OPT_SLAB_PASS_LOG2 && (value < 12 || value > 24))
            GF2_FAIL(GF2_E_ARG, "gf2_ctx_set_option: GF2_OPT_SLAB_PASS_LOG2 must be in 12..24");
        if (option == GF2_OPT_MC_CHUNK_LOG2 && (value < 16 || value > 22))
            GF2_FAIL(GF2_E_ARG, "gf2_ctx_set_option: GF2_OPT_MC_CHUNK_LOG2 must be in 16..22");
        if (option == GF2_OPT_COMBINE_THREADS && value != 64 && value != 128 && value != 256 && value != 512 && value != 1024)
            GF2_FAIL(GF2_E_ARG, "gf2_ctx_set_option: GF2_OPT_COMBINE_THREADS must be 64, 128, 256, 512 or 1024");
        if (option == GF2_OPT_COMBINE_BLOCKS && (value < 1 || value > 65535))
            GF2_FAIL(GF2_E_ARG, "gf2_ctx_set_option: GF2_OPT_COMBINE_BLOCKS must be in 1..65535");
        if (option == GF2_OPT_REDO_BLOCKS_PER_CU && (value < 1 || value > 64))
            GF2_FAIL(GF2_E_ARG, "gf2_ctx_set_option: GF2_OPT_REDO_BLOCKS_PER_CU must be in 1..64");
        if (option == GF2_OPT_GATHER_OVER && (value < 1 || value > 8))
            GF2_FAIL(GF2_E_ARG, "gf2_ctx_set_option: GF2_OPT_GATHER_OVER must be in 1..8");
        if ((option == GF2_OPT_GATHER_REVERSE || option == GF2_OPT_GATHER_CROSS) && value > 1)
            GF2_FAIL(GF2_E_ARG, "gf2_ctx_set_option: GF2_OPT_GATHER_REVERSE and GF2_OPT_GATHER_CROSS must be 0 or 1");
        if (option == GF2_OPT_MC_SAMPLER_WAVES && (value < 1 || value > 9))
            GF2_FAIL(GF2_E_ARG, "gf2_ctx_set_option: GF2_OPT_MC_SAMPLER_WAVES must be in 1..9");
        if (option == GF2_OPT_MC_TAIL_CAP && (value > 8 || (value & 1)))
            GF2_FAIL(GF2_E_ARG, "gf2_ctx_set_option: GF2_OPT_MC_TAIL_CAP must be 0, 2, 4, 6 or 8");
        if (option == GF2_OPT_RREF_SMALL_BCAST && value > 2) GF2_FAIL(GF2_E_ARG, "gf2_ctx_set_option: GF2_OPT_RREF_SMALL_BCAST must be 0, 1 or 2");
    }
    ctx->opt[option] = value < 0 ? -1 : value;
    return GF2_OK;
}

int gf2_ctx_destroy(gf2_ctx* ctx) {
    if (!ctx) return GF2_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int k = 0; k < 2; ++k) (void)hipStreamSynchronize(ctx->side[k]);
    (void)hipStreamSynchronize(ctx->hi);
    for (int k = 0; k < 4; ++k)
        if (ctx->ws[k]) (void)hipFree(ctx->ws[k]);
    if (ctx->seg_cdf_dev) (void)hipFree(ctx->seg_cdf_dev);
    destroy_streams_and_events(ctx);
    free(ctx);
    return GF2_OK;
}

int gf2_ctx_sync(gf2_ctx* ctx) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_ctx_sync: null context");
    GF2_TRY(gf2_ctx_activate(ctx));
    GF2_TRY(gf2_stream_wait(ctx->stream));
    return GF2_OK;
}

int gf2_dev_alloc(gf2_ctx* ctx, size_t bytes, void** dev_out) {
    if (!ctx || !dev_out) GF2_FAIL(GF2_E_ARG, "gf2_dev_alloc: null argument");
    GF2_TRY(gf2_ctx_activate(ctx));
    *dev_out = nullptr;
    hipError_t err = hipMalloc(dev_out, bytes ? bytes : 8);
    if (err == hipErrorOutOfMemory) GF2_FAIL(GF2_E_NOMEM, "gf2_dev_alloc: %zu bytes: out of device memory", bytes);
    GF2_HIP(err);
    return GF2_OK;
}

int gf2_dev_free(gf2_ctx* ctx, void* dev) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_dev_free: null context");
    if (!dev) return GF2_OK;
    GF2_TRY(gf2_ctx_activate(ctx));
    GF2_HIP(hipStreamSynchronize(ctx->stream));
    GF2_HIP(hipFree(dev));
    return GF2_OK;
}

int gf2_dev_zero(gf2_ctx* ctx, void* dev, size_t bytes) {
    if (!ctx || (!dev && bytes)) GF2_FAIL(GF2_E_ARG, "gf2_dev_zero: null argument");
    GF2_TRY(gf2_ctx_activate(ctx));
    if (bytes) GF2_HIP(hipMemsetAsync(dev, 0, bytes, ctx->stream));
    return GF2_OK;
}

int gf2_h2d(gf2_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes) {
    if (!ctx || ((!dev_dst || !host_src) && bytes)) GF2_FAIL(GF2_E_ARG, "gf2_h2d: null argument");
    GF2_TRY(gf2_ctx_activate(ctx));
    if (bytes) {
        GF2_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        GF2_HIP(hipStreamSynchronize(ctx->stream));
    }
    return GF2_OK;
}

int gf2_d2h(gf2_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
    if (!ctx || ((!host_dst || !dev_src) && bytes)) GF2_FAIL(GF2_E_ARG, "gf2_d2h: null argument");
    GF2_TRY(gf2_ctx_activate(ctx));
    if (bytes) {
        GF2_TRY(gf2_stream_wait(ctx->stream));          // a copy to pageable memory waits for the stream first: do it by polling
        GF2_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        GF2_TRY(gf2_stream_wait(ctx->stream));
    }
    return GF2_OK;
}

int gf2_timer_start(gf2_ctx* ctx) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_timer_start: null context");
    GF2_TRY(gf2_ctx_activate(ctx));
    GF2_HIP(hipEventRecord(ctx->t0, ctx->stream));
    return GF2_OK;
}

int gf2_timer_stop(gf2_ctx* ctx, float* elapsed_ms_out) {
    if (!ctx || !elapsed_ms_out) GF2_FAIL(GF2_E_ARG, "gf2_timer_stop: null argument");
    GF2_TRY(gf2_ctx_activate(ctx));
    GF2_HIP(hipEventRecord(ctx->t1, ctx->stream));
    GF2_TRY(gf2_stream_wait(ctx->stream));              // polled (see gf2_stream_wait); the event is then complete
    GF2_HIP(hipEventSynchronize(ctx->t1));
    GF2_HIP(hipEventElapsedTime(elapsed_ms_out, ctx->t0, ctx->t1));
    return GF2_OK;
}

int gf2_profile_enable(gf2_ctx* ctx, int on) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_profile_enable: null context");
    GF2_TRY(gf2_prof_drain(ctx));
    ctx->profile_on = on ? 1 : 0;
    return GF2_OK;
}

int gf2_profile_reset(gf2_ctx* ctx) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_profile_reset: null context");
    GF2_TRY(gf2_prof_drain(ctx));
    for (int k = 0; k < GF2_K_COUNT; ++k) {
        ctx->prof_ms[k] = 0.0;
        ctx->prof_launches[k] = 0;
    }
    return GF2_OK;
}

int gf2_profile_get(gf2_ctx* ctx, int kernel_family, double* total_ms_out, int64_t* launches_out) {
    if (!ctx || kernel_family < 0 || kernel_family >= GF2_K_COUNT)
        GF2_FAIL(GF2_E_ARG, "gf2_profile_get: bad argument");
    GF2_TRY(gf2_prof_drain(ctx));
    if (total_ms_out) *total_ms_out = ctx->prof_ms[kernel_family];
    if (launches_out) *launches_out = ctx->prof_launches[kernel_family];
    return GF2_OK;
}

// ---- memory bandwidth probe ---------------------------------------------------------------------------

int gf2_membw_probe_dev(gf2_ctx* ctx, const void* src_dev, void* dst_dev, size_t bytes, uint64_t* sink_dev) {
    if (!ctx || !src_dev || !sink_dev || (bytes & 15)) GF2_FAIL(GF2_E_ARG, "gf2_membw_probe_dev: bad argument (bytes must be a multiple of 16)");
    GF2_TRY(gf2_ctx_activate(ctx));
    const int64_t vecs = (int64_t)(bytes >> 4);
    const dim3 grid((unsigned)(ctx->num_cus * 8)), block(512);
    if (dst_dev)
        hipLaunchKernelGGL(membw_probe_kernel<true>, grid, block, 0, ctx->stream, (const probe_vec*)src_dev, (probe_vec*)dst_dev, vecs,
                           (u64*)sink_dev);
    else
        hipLaunchKernelGGL(membw_probe_kernel<false>, grid, block, 0, ctx->stream, (const probe_vec*)src_dev, (probe_vec*)nullptr, vecs,
                           (u64*)sink_dev);
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

// (host-side packing -- gf2_pack_rows_*, gf2_unpack_rows_* -- and the error message plumbing live in gf2_host.cpp: a plain C++
// translation unit without HIP, so that the CPU box can build it with -fsanitize=thread / address: `make tsan`, `make asan`)
}   // extern "C"

// ---- internals --------------------------------------------------------------------------------------

int gf2_ctx_activate(gf2_ctx* ctx) {
    GF2_HIP(hipSetDevice(ctx->device));
    return GF2_OK;
}

int gf2_ws_reserve(gf2_ctx* ctx, int slot, size_t bytes) {
    if (ctx->ws_bytes[slot] >= bytes) return GF2_OK;
    GF2_HIP(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 2; ++k) GF2_HIP(hipStreamSynchronize(ctx->side[k]));
    GF2_HIP(hipStreamSynchronize(ctx->hi));
    if (ctx->ws[slot]) GF2_HIP(hipFree(ctx->ws[slot]));
    ctx->ws[slot] = nullptr;
    ctx->ws_bytes[slot] = 0;
    hipError_t err = hipMalloc(&ctx->ws[slot], bytes);
    if (err == hipErrorOutOfMemory) GF2_FAIL(GF2_E_NOMEM, "workspace of %zu bytes: out of device memory", bytes);
    GF2_HIP(err);
    ctx->ws_bytes[slot] = bytes;
    return GF2_OK;
}

int gf2_prof_drain(gf2_ctx* ctx) {
    if (ctx->prof_used == 0) return GF2_OK;
    GF2_TRY(gf2_ctx_activate(ctx));
    GF2_HIP(hipEventSynchronize(ctx->prof_ev[ctx->prof_used - 1][1]));
    for (int i = 0; i < ctx->prof_used; ++i) {
        float ms = 0.f;
        GF2_HIP(hipEventElapsedTime(&ms, ctx->prof_ev[i][0], ctx->prof_ev[i][1]));
        ctx->prof_ms[ctx->prof_family[i]] += ms;
        ctx->prof_launches[ctx->prof_family[i]] += 1;
    }
    ctx->prof_used = 0;
    return GF2_OK;
}

// A slot is taken by gf2_prof_end, not by gf2_prof_begin: an entry point that returns with an error between the two leaves
// nothing open -- the next gf2_prof_begin records into the same slot again.
int gf2_prof_begin(gf2_ctx* ctx, int family) {
    if (!ctx->profile_on) return GF2_OK;
    if (ctx->prof_used == gf2_ctx::kProfSlots) GF2_TRY(gf2_prof_drain(ctx));
    ctx->prof_family[ctx->prof_used] = family;
    GF2_HIP(hipEventRecord(ctx->prof_ev[ctx->prof_used][0], ctx->stream));
    return GF2_OK;
}

int gf2_prof_end(gf2_ctx* ctx) {
    if (!ctx->profile_on) return GF2_OK;
    GF2_HIP(hipEventRecord(ctx->prof_ev[ctx->prof_used][1], ctx->stream));
    ctx->prof_used += 1;
    return GF2_OK;
}
