// Multi-tensor Adam for gfx950 with torch.optim.Adam's exact update order (single-tensor path of
// torch/optim/adam.py): L2 decay folded into the gradient, lerp for exp_avg, sqrt(v)/sqrt(bc2) + eps.
// HBM-bound: 16 B read + 12 B written per parameter; one launch for all tensors (blockIdx.y = tensor).
//
// Replaces torch.optim.Adam(..., betas=(0.937, 0.999), weight_decay=5e-4).step() as built by the reference
// (demos/yolov3_u/train.py:66-70).
#include <math.h>

#include "common.h"

namespace {

// Device-resident step state (fva_adam_step_dev): the step count and the learning rate live in device memory, so that a
// captured HIP graph of the whole training step replays with the right bias corrections and follows an LR schedule.
// state[0] = step count (as a double: exact to 2^53), state[1] = lr / (1 - beta1^step), state[2] = sqrt(1 - beta2^step)
__global__ void adam_tick_kernel(double* state, const float* __restrict__ lr, double beta1, double beta2) {
    const double step = state[0] + 1.0;
    state[0] = step;
    state[1] = (double)*lr / (1.0 - pow(beta1, step));
    state[2] = sqrt(1.0 - pow(beta2, step));
}

__global__ __launch_bounds__(256) void adam_kernel(const void* const* __restrict__ ptrs, const int64_t* __restrict__ sizes, int n,
                                                   float step_size, float bc2_sqrt, float beta1, float beta2, float eps, float wd,
                                                   float gscale, const double* __restrict__ dev_state) {
    if (dev_state) {
        step_size = (float)dev_state[1];
        bc2_sqrt = (float)dev_state[2];
    }
    const int t = blockIdx.y;
    const int64_t size = sizes[t];
    float* p = (float*)ptrs[t];
    const float* g = (const float*)ptrs[n + t];
    float* m = (float*)ptrs[2 * n + t];
    float* v = (float*)ptrs[3 * n + t];
    if (g == nullptr) return;  // parameter without a gradient this step (torch skips it too)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < size; i += (int64_t)gridDim.x * blockDim.x) {
        float gr = g[i] * gscale;
        const float pv = p[i];
        if (wd != 0.f) gr = gr + wd * pv;
        float mi = m[i], vi = v[i];
        mi = mi + (gr - mi) * (1.f - beta1);
        vi = vi * beta2 + (1.f - beta2) * gr * gr;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pv - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}

// dst[offs[t] + i] = (bf16 | f32) src_t[i] for the n tensors of a gradient bucket in ONE launch (blockIdx.y = tensor): the bucket
// copies of parallel.GradientReducer were 222 torch copy launches issued from per-parameter autograd hooks (~8 ms of host time per
// step); a null source pointer or a zero size skips the tensor.
__global__ __launch_bounds__(256) void gather_cast_kernel(const int64_t* __restrict__ table, int n, void* __restrict__ dst, int to_bf16) {
    const int t = blockIdx.y;
    const float* __restrict__ src = (const float*)table[t];
    const int64_t size = table[n + t], off = table[2 * n + t];
    if (src == nullptr || size <= 0) return;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (((size | off) & 3) == 0 && ((uintptr_t)src & 15) == 0) {
        const f32x4* s4 = (const f32x4*)src;
        if (to_bf16) {
            bf16x4* d4 = (bf16x4*)((bf16_t*)dst + off);
            for (int64_t i = i0; i < size / 4; i += stride) {
                const f32x4 v = s4[i];
                d4[i] = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            }
        } else {
            f32x4* d4 = (f32x4*)((float*)dst + off);
            for (int64_t i = i0; i < size / 4; i += stride) d4[i] = s4[i];
        }
    } else if (to_bf16) {
        bf16_t* d = (bf16_t*)dst + off;
        for (int64_t i = i0; i < size; i += stride) d[i] = (bf16_t)src[i];
    } else {
        float* d = (float*)dst + off;
        for (int64_t i = i0; i < size; i += stride) d[i] = src[i];
    }
}

}  // namespace

extern "C" int fva_gather_cast(const int64_t* table_dev, int32_t n, int64_t max_size, void* dst, int dst_dtype, void* stream) {
    if (!table_dev || n < 1 || !dst || (dst_dtype != FVA_F32 && dst_dtype != FVA_BF16)) return fva_fail(FVA_ERR_ARG, "fva_gather_cast: bad argument");
    int64_t gx = (max_size / 4 + 255) / 256;
    if (gx > 256) gx = 256;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(gather_cast_kernel, dim3((int)gx, n), dim3(256), 0, (hipStream_t)stream, table_dev, n, dst, dst_dtype == FVA_BF16 ? 1 : 0);
    FVA_LAUNCH_CHECK("gather_cast_kernel");
    return FVA_OK;
}

extern "C" int fva_adam_step(const void* const* ptrs, const int64_t* sizes, int32_t n, int64_t max_size, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
    if (!ptrs || !sizes || n < 1 || step < 1) return fva_fail(FVA_ERR_ARG, "fva_adam_step: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    int64_t gx = (max_size + 1023) / 1024;
    if (gx > 128) gx = 128;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((int)gx, n), dim3(256), 0, (hipStream_t)stream, ptrs, sizes, n, step_size, bc2_sqrt, beta1,
                       beta2, eps, weight_decay, grad_scale, (const double*)nullptr);
    FVA_LAUNCH_CHECK("adam_kernel");
    return FVA_OK;
}

extern "C" int fva_adam_step_dev(const void* const* ptrs, const int64_t* sizes, int32_t n, int64_t max_size, const float* lr_dev, float beta1,
                                 float beta2, float eps, float weight_decay, double* state_dev, float grad_scale, void* stream) {
    if (!ptrs || !sizes || n < 1 || !lr_dev || !state_dev) return fva_fail(FVA_ERR_ARG, "fva_adam_step_dev: bad argument");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state_dev, lr_dev, (double)beta1, (double)beta2);
    FVA_LAUNCH_CHECK("adam_tick_kernel");
    int64_t gx = (max_size + 1023) / 1024;
    if (gx > 128) gx = 128;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((int)gx, n), dim3(256), 0, (hipStream_t)stream, ptrs, sizes, n, 0.f, 1.f, beta1, beta2, eps,
                       weight_decay, grad_scale, (const double*)state_dev);
    FVA_LAUNCH_CHECK("adam_kernel");
    return FVA_OK;
}
