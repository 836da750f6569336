#include "common.h"
thread_local char fva_err_buf[512] = "";
extern "C" const char* fva_last_error(void) { return fva_err_buf; }
extern "C" int fva_version(void) { return 1; }

// ---- profiling spans --------------------------------------------------------------------------------------------------
#include <vector>
namespace {
struct Span { hipEvent_t e0, e1; int cls; double flop; };
std::vector<Span> g_spans;     // events are created by fva_profile_start, i.e. outside whatever the caller is timing
int g_used = 0;
bool g_on = false;
unsigned g_mask = ~0u;         // classes that are bracketed while armed (bit = class)
}  // namespace

FvaProfileSpan::FvaProfileSpan(int cls, double flop, hipStream_t s) : slot(-1), stream(s) {
    if (!g_on || !((g_mask >> cls) & 1u) || g_used >= (int)g_spans.size()) return;
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

extern "C" int fva_profile_classes(uint32_t mask) {
    g_mask = mask;
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
