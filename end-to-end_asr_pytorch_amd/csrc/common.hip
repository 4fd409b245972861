// Error string + version for libasr_hip.so.
#include <stdarg.h>

#include <stdlib.h>

#include "asr_common.h"

static thread_local char g_err[512] = "";

void asr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int asr_version(void) { return 102; }   // 101: asr_vocab_proj_lse / asr_ctc_loss_fwd_lse removed (round 5), asr_launch_budget_current added; 102: asr_attn_ffn_fwd added
extern "C" const char* asr_last_error(void) { return g_err; }

// deterministic mode: kernels that would otherwise combine partial results with float atomics in arrival order (the forward GEMMs'
// split-K, the weight gradient's bias side product) take their single-writer form
static int g_deterministic = -1;
int asr_deterministic() {
    if (g_deterministic < 0) { const char* e = getenv("ASR_AMD_DETERMINISTIC"); g_deterministic = (e && atoi(e) != 0) ? 1 : 0; }
    return g_deterministic;
}
// launch budget (asr_hip.h: asr_launch_budget): per host thread, read by the launches named there
static thread_local int g_launch_budget = 0;
int asr_launch_budget_current() { return g_launch_budget; }
extern "C" int asr_launch_budget(int cus) { const int old = g_launch_budget; g_launch_budget = cus > 0 ? cus : 0; return old; }

extern "C" int asr_set_deterministic(int on) { const int old = asr_deterministic(); g_deterministic = on ? 1 : 0; return old; }

// ---- test support: leave every CU's LDS full of bf16 / f32 NaN patterns -----------------------------------------------------------
// A kernel that reads an LDS byte before anything it waits for has written it usually goes unnoticed - the bytes are what the previous
// workgroup of the SAME kernel left there, finite and plausible - until another process shares the GPU (round 3: the pipelined
// attention forward read K tile 1 one wait too early; two ranks on one GPU turned that into NaN parameters).  The parity tests run the
// LDS-staged kernels once more behind this launch and require bit-identical results.
namespace {
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned lds_dyn[];
    const int words = 160 * 1024 / 4;
    for (int i = threadIdx.x; i < words; i += 256) lds_dyn[i] = pattern;
    __syncthreads();
    if (lds_dyn[(threadIdx.x * 37) % words] != pattern) sink[0] = 1;     // (keeps the stores)
}
}  // namespace

extern "C" int asr_debug_poison_lds(void* stream, void* scratch4) {
    ASR_REQUIRE(scratch4, ASR_ERR_ARG, "debug_poison_lds: a 4-byte device scratch word is required");
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { asr_set_error("debug_poison_lds: %s", hipGetErrorString(e)); return (int)e; }
        attr = true;
    }
    hipLaunchKernelGGL(poison_lds_kernel, dim3(2048), dim3(256), 160 * 1024, static_cast<hipStream_t>(stream), 0x7fc07fc0u,
                       reinterpret_cast<unsigned*>(scratch4));
    ASR_LAUNCH_CHECK("debug_poison_lds");
    return 0;
}

// ---- which streams share a hardware queue ---------------------------------------------------------------------------------------------
// The HIP runtime multiplexes every stream of the process onto a few hardware queues (4 by default; 8 or 16 made every training step
// slower here) and kernels of two streams on one queue do not overlap.  Which streams share is decided by creation order across the
// whole process: the trainer's weight-gradient stream landing on the launch stream's queue cost up to 3 ms of a 7 ms step.  This probe
// answers the question for one pair: a kernel that spins ~80 us on `a`, a kernel that stamps the time on `b`; on a shared queue the
// stamp comes after the spin has ended.  Synchronises both streams: for set-up time, not for the step.
namespace {
__global__ void queue_probe_spin_kernel(unsigned long long* out, unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) out[0] = __builtin_amdgcn_s_memrealtime();
}
__global__ void queue_probe_stamp_kernel(unsigned long long* out) {
    if (threadIdx.x == 0) out[1] = __builtin_amdgcn_s_memrealtime();
}
}  // namespace

// Stream-ordering events without HIP's default system-scope fence at every record (asr_hip.h): for edges between streams of one device.
extern "C" int asr_event_create(void** out_event) {
    ASR_REQUIRE(out_event, ASR_ERR_ARG, "asr_event_create: null argument");
    hipEvent_t e = nullptr;
    ASR_REQUIRE(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess, ASR_ERR_ARG,
                "asr_event_create: hipEventCreateWithFlags failed");
    *out_event = e;
    return 0;
}
extern "C" int asr_stream_order_after(void* later_stream, void* earlier_stream, void* event) {
    ASR_REQUIRE(event, ASR_ERR_ARG, "asr_stream_order_after: null event");
    if (hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(earlier_stream)) != hipSuccess ||
        hipStreamWaitEvent(static_cast<hipStream_t>(later_stream), static_cast<hipEvent_t>(event), 0) != hipSuccess) {
        asr_set_error("asr_stream_order_after: record / wait failed");
        return ASR_ERR_ARG;
    }
    return 0;
}
extern "C" int asr_event_destroy(void* event) {
    if (event) (void)hipEventDestroy(static_cast<hipEvent_t>(event));
    return 0;
}

// Timing events without the system-scope fence (hip_runtime_api.h on hipEventDisableSystemFence: "can improve the accuracy of timing
// measurements by avoiding the cost of cache writeback and invalidation, and the performance impact of those actions on the execution
// of following work"): what bench.py's per-op pass brackets every op with.
extern "C" int asr_timer_create(void** out_event) {
    ASR_REQUIRE(out_event, ASR_ERR_ARG, "asr_timer_create: null argument");
    hipEvent_t e = nullptr;
    ASR_REQUIRE(hipEventCreateWithFlags(&e, hipEventDisableSystemFence) == hipSuccess, ASR_ERR_ARG, "asr_timer_create: hipEventCreateWithFlags failed");
    *out_event = e;
    return 0;
}
extern "C" int asr_timer_record(void* event, void* stream) {
    ASR_REQUIRE(event, ASR_ERR_ARG, "asr_timer_record: null event");
    ASR_REQUIRE(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)) == hipSuccess, ASR_ERR_ARG, "asr_timer_record failed");
    return 0;
}
extern "C" int asr_timer_elapsed_ms(void* start, void* stop, float* ms) {
    ASR_REQUIRE(start && stop && ms, ASR_ERR_ARG, "asr_timer_elapsed_ms: null argument");
    ASR_REQUIRE(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)) == hipSuccess, ASR_ERR_ARG,
                "asr_timer_elapsed_ms: events not complete (synchronise first)");
    return 0;
}

extern "C" int asr_streams_share_queue(void* stream_a, void* stream_b, int* shared) {
    ASR_REQUIRE(shared, ASR_ERR_ARG, "streams_share_queue: null result pointer");
    if (stream_a == stream_b) { *shared = 1; return 0; }
    static thread_local unsigned long long* buf = nullptr;
    hipError_t e;
    if (!buf && (e = hipMalloc(reinterpret_cast<void**>(&buf), 16)) != hipSuccess) { asr_set_error("streams_share_queue: %s", hipGetErrorString(e)); return (int)e; }
    hipStream_t a = static_cast<hipStream_t>(stream_a), b = static_cast<hipStream_t>(stream_b);
    if ((e = hipStreamSynchronize(a)) != hipSuccess || (e = hipStreamSynchronize(b)) != hipSuccess) { asr_set_error("streams_share_queue: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(queue_probe_spin_kernel, dim3(1), dim3(64), 0, a, buf, 8000ull);      // 80 us of the 100 MHz counter
    hipLaunchKernelGGL(queue_probe_stamp_kernel, dim3(1), dim3(64), 0, b, buf);
    ASR_LAUNCH_CHECK("streams_share_queue");
    unsigned long long h[2] = {0, 0};
    if ((e = hipStreamSynchronize(a)) != hipSuccess || (e = hipStreamSynchronize(b)) != hipSuccess ||
        (e = hipMemcpy(h, buf, 16, hipMemcpyDeviceToHost)) != hipSuccess) { asr_set_error("streams_share_queue: %s", hipGetErrorString(e)); return (int)e; }
    *shared = h[1] >= h[0] ? 1 : 0;
    return 0;
}
