#include "common.h"
thread_local char fva_err_buf[512] = "";
extern "C" const char* fva_last_error(void) { return fva_err_buf; }
extern "C" int fva_version(void) { return 1; }
static thread_local const char* g_last_kernel = "";
void fva_note_kernel(const char* name) { g_last_kernel = name; }
extern "C" const char* fva_conv_last_kernel(void) { return g_last_kernel; }

// ---- profiling spans --------------------------------------------------------------------------------------------------
#include <vector>
namespace {
struct Span { hipEvent_t e0, e1; int cls; double flop; };
std::vector<Span> g_spans;     // events are created by fva_profile_start, i.e. outside whatever the caller is timing
int g_used = 0;
bool g_on = false;
unsigned g_mask = ~0u;         // classes that are bracketed while armed (bit = class)
int g_stride = 1, g_seen = 0;  // of the eligible calls every g_stride-th is bracketed
}  // namespace

FvaProfileSpan::FvaProfileSpan(int cls, double flop, hipStream_t s) : slot(-1), stream(s) {
    if (!g_on || !((g_mask >> (cls & 0xff)) & 1u) || (g_seen++ % g_stride) != 0 || g_used >= (int)g_spans.size()) return;   // bits 8.. of cls: kernel size
    slot = g_used++;
    g_spans[slot].cls = cls;
    g_spans[slot].flop = flop;
    (void)hipEventRecord(g_spans[slot].e0, stream);
}
FvaProfileSpan::~FvaProfileSpan() {
    if (slot >= 0) (void)hipEventRecord(g_spans[slot].e1, stream);
}

extern "C" int fva_profile_start(int32_t max_spans) {
    if (max_spans < 0) return fva_fail(FVA_ERR_ARG, "fva_profile_start: negative span count");
    while ((int)g_spans.size() < max_spans) {
        Span sp{};
        if (hipEventCreate(&sp.e0) != hipSuccess || hipEventCreate(&sp.e1) != hipSuccess) return fva_fail(FVA_ERR_LAUNCH, "fva_profile_start: hipEventCreate failed");
        // record once now: the runtime allocates an event's backing signal on first use
        (void)hipEventRecord(sp.e0, nullptr);
        (void)hipEventRecord(sp.e1, nullptr);
        g_spans.push_back(sp);
    }
    (void)hipDeviceSynchronize();
    g_used = 0;
    g_on = true;
    return FVA_OK;
}

extern "C" int fva_profile_classes(uint32_t mask, int32_t stride) {
    if (stride < 1) return fva_fail(FVA_ERR_ARG, "fva_profile_classes: stride must be >= 1");
    g_mask = mask;
    g_stride = stride;
    g_seen = 0;
    return FVA_OK;
}

extern "C" int32_t fva_profile_stop(int32_t* cls, double* flop, float* ms, int32_t cap) {
    g_on = false;
    (void)hipDeviceSynchronize();
    int n = 0;
    for (int i = 0; i < g_used && n < cap; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_spans[i].e0, g_spans[i].e1) != hipSuccess) continue;
        cls[n] = g_spans[i].cls;
        flop[n] = g_spans[i].flop;
        ms[n] = t;
        ++n;
    }
    g_used = 0;
    return n;
}

// ---- side stream ------------------------------------------------------------------------------------------------------
// The weight gradient of a layer has no consumer before the optimizer, so it can run beside the rest of the backward pass:
// a low-priority stream whose blocks fill the CUs that the partly empty last round of a dgrad launch (256x256 tiles: 800,
// 400 or 200 tiles on 256 CUs) and the HBM-bound BatchNorm passes leave idle.  fork: the side stream waits for everything
// enqueued on `main` so far; join: `main` waits for everything enqueued on the side stream so far.
namespace {
hipStream_t g_side = nullptr;
int g_side_dev = -1;
hipEvent_t g_side_ev[64];
bool g_side_ev_ready = false;
int g_side_next = 0;
hipEvent_t side_event() { return g_side_ev[g_side_next++ & 63]; }

// a fresh lowest-priority stream (and, once per process, the fork / join events); the library's side stream is replaced only on success
int side_stream_create(hipStream_t* out, const char* who) {
    int least = 0, greatest = 0;
    hipStream_t s = nullptr;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) != hipSuccess)
        return fva_fail(FVA_ERR_LAUNCH, "%s: cannot create the side stream", who);
    if (!g_side_ev_ready) {
        for (auto& e : g_side_ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
                (void)hipStreamDestroy(s);
                return fva_fail(FVA_ERR_LAUNCH, "%s: hipEventCreate failed", who);
            }
        g_side_ev_ready = true;
    }
    *out = s;
    return FVA_OK;
}
}  // namespace

extern "C" int fva_side_stream_fork(void* main_stream, void** side_stream) {
    if (!side_stream) return fva_fail(FVA_ERR_ARG, "fva_side_stream_fork: null pointer");
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (g_side && dev != g_side_dev)   // one process per GPU is the model; a second device in the same process stays on its launch stream
        return fva_fail(FVA_ERR_ARG, "fva_side_stream_fork: the side stream belongs to device %d, current device is %d", g_side_dev, dev);
    if (!g_side) {
        hipStream_t s = nullptr;
        const int rc = side_stream_create(&s, "fva_side_stream_fork");
        if (rc) return rc;
        g_side = s;
        g_side_dev = dev;
    }
    hipEvent_t e = side_event();
    if (hipEventRecord(e, (hipStream_t)main_stream) != hipSuccess || hipStreamWaitEvent(g_side, e, 0) != hipSuccess)
        return fva_fail(FVA_ERR_LAUNCH, "fva_side_stream_fork: event record / wait failed");
    *side_stream = (void*)g_side;
    return FVA_OK;
}

// Drop the side stream for a fresh one.  How a HIP stream maps onto the hardware queues is the runtime's choice at creation; on some
// boxes / processes the low-priority stream lands where it no longer yields to the launch stream (the eager two-stream step then takes
// 32-35 ms instead of 28.9, profiles/r03_experiments.md).  ops.autotune_wgrad_side_stream() times the step and asks for another stream
// when the first one loses.  The old stream is drained and then DESTROYED -- unless a capture is in progress on it (refused: a captured
// graph names its streams only while capturing; replays do not use them).  The new stream is created first and swapped in on success.
extern "C" int fva_side_stream_renew(void) {
    if (!g_side) return FVA_OK;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(g_side, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
        return fva_fail(FVA_ERR_ARG, "fva_side_stream_renew: the side stream is being captured");
    if (hipStreamSynchronize(g_side) != hipSuccess) return fva_fail(FVA_ERR_LAUNCH, "fva_side_stream_renew: synchronize failed");
    hipStream_t s = nullptr;
    const int rc = side_stream_create(&s, "fva_side_stream_renew");
    if (rc) return rc;               // the old stream stays in place
    (void)hipStreamDestroy(g_side);
    g_side = s;
    return FVA_OK;
}

extern "C" int fva_side_stream_join(void* main_stream) {
    if (!g_side) return FVA_OK;
    hipEvent_t e = side_event();
    if (hipEventRecord(e, g_side) != hipSuccess || hipStreamWaitEvent((hipStream_t)main_stream, e, 0) != hipSuccess)
        return fva_fail(FVA_ERR_LAUNCH, "fva_side_stream_join: event record / wait failed");
    return FVA_OK;
}

// ---- halo border ------------------------------------------------------------------------------------------------------
namespace {
// zero border of a halo NHWC buffer [B][H+2p][W+2p][C] (16-byte pieces): one block per padded row
__global__ __launch_bounds__(256) void halo_border_zero_kernel(uint4* __restrict__ z, int H, int W, int cpp, int pad) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int yp = blockIdx.x % Hp;
    uint4* row = z + (int64_t)blockIdx.x * Wp * cpp;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    if (yp < pad || yp >= H + pad) {
        for (int i = threadIdx.x; i < Wp * cpp; i += 256) row[i] = zero;
    } else {
        for (int i = threadIdx.x; i < 2 * pad * cpp; i += 256) {
            const int side = i / (pad * cpp), j = i - side * pad * cpp;
            row[(side ? (W + pad) * cpp : 0) + j] = zero;
        }
    }
}
}  // namespace

int fva_zero_halo_border(void* z, int B, int H, int W, int cpp, int pad, hipStream_t stream) {
    if (pad <= 0) return FVA_OK;
    hipLaunchKernelGGL(halo_border_zero_kernel, dim3(B * (H + 2 * pad)), dim3(256), 0, stream, (uint4*)z, H, W, cpp, pad);
    FVA_LAUNCH_CHECK("halo_border_zero_kernel");
    return FVA_OK;
}
