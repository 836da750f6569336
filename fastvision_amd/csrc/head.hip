// Detection-head backward glue for gfx950: turns the fp32 head gradient [B*H*W][N] (N = A*(5+C) = 255, odd
// pitch) into the halo NHWC operand the MFMA dgrad/wgrad kernels consume (N padded to Npad, zero border), applies
// the upstream scalar gradient, and reduces the bias gradient deterministically.  HBM-bound, small.
//
// Replaces the autograd backward of the biased 1x1 nn.Conv2d heads (reference detection/head/yolov3head.py:50,60).
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void head_prepare_kernel(const float* __restrict__ dhead, const float* __restrict__ gscale,
                                                           T* __restrict__ dy, int B, int H, int W, int N, int Npad) {
    const int Hp = H + 2, Wp = W + 2;
    const int64_t total = (int64_t)B * Hp * Wp * Npad;
    const float g = gscale ? *gscale : 1.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % Npad);
        const int64_t pix = i / Npad;
        const int xp = (int)(pix % Wp), yp = (int)((pix / Wp) % Hp), b = (int)(pix / ((int64_t)Wp * Hp));
        const int y = yp - 1, x = xp - 1;
        float v = 0.f;
        if (n < N && y >= 0 && y < H && x >= 0 && x < W) v = g * dhead[(((int64_t)b * H + y) * W + x) * N + n];
        dy[i] = from_f<T>(v);
    }
}

// partial[blk][n] = sum over the block's rows of dhead[m][n]
__global__ __launch_bounds__(256) void head_bias_partial_kernel(const float* __restrict__ dhead, float* __restrict__ partial,
                                                                int64_t M, int N, int rows_per_block) {
    const int64_t m0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t m1 = m0 + rows_per_block;
    if (m1 > M) m1 = M;
    for (int n = threadIdx.x; n < N; n += 256) {
        float s = 0.f;
        for (int64_t m = m0; m < m1; ++m) s += dhead[m * N + n];
        partial[(int64_t)blockIdx.x * N + n] = s;
    }
}

// one block per output channel: 256 lanes share the partial rows, fixed-order tree in LDS (deterministic)
__global__ __launch_bounds__(256) void head_bias_final_kernel(const float* __restrict__ partial, const float* __restrict__ gscale,
                                                              float* __restrict__ dbias, int N, int nblocks, int accumulate) {
    __shared__ double red[256];
    const int n = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += (double)partial[(int64_t)b * N + n];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float v = (float)red[0] * (gscale ? *gscale : 1.f);
        dbias[n] = accumulate ? dbias[n] + v : v;
    }
}

}  // namespace

extern "C" int fva_head_bwd_prepare(int dtype, const float* dhead, const float* grad_scale, void* dy, float* dbias, int accumulate,
                                    void* workspace, int B, int H, int W, int N, int Npad, void* stream) {
    if (!dhead || !dy || !dbias || !workspace) return fva_fail(FVA_ERR_ARG, "fva_head_bwd_prepare: null pointer");
    if (Npad < N || Npad % 8) return fva_fail(FVA_ERR_ARG, "fva_head_bwd_prepare: bad Npad %d for N %d", Npad, N);
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * (H + 2) * (W + 2) * Npad;
    int64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL(head_prepare_kernel<bf16_t>, dim3((int)g), dim3(256), 0, s, dhead, grad_scale, (bf16_t*)dy, B, H, W, N, Npad);
    else if (dtype == FVA_F32)
        hipLaunchKernelGGL(head_prepare_kernel<float>, dim3((int)g), dim3(256), 0, s, dhead, grad_scale, (float*)dy, B, H, W, N, Npad);
    else
        return fva_fail(FVA_ERR_ARG, "fva_head_bwd_prepare: bad dtype");
    FVA_LAUNCH_CHECK("head_prepare_kernel");
    const int64_t M = (int64_t)B * H * W;
    int nblocks = (int)((M + 63) / 64);
    if (nblocks > 1024) nblocks = 1024;
    const int rows = (int)((M + nblocks - 1) / nblocks);
    nblocks = (int)((M + rows - 1) / rows);
    hipLaunchKernelGGL(head_bias_partial_kernel, dim3(nblocks), dim3(256), 0, s, dhead, (float*)workspace, M, N, rows);
    FVA_LAUNCH_CHECK("head_bias_partial_kernel");
    hipLaunchKernelGGL(head_bias_final_kernel, dim3(N), dim3(256), 0, s, (const float*)workspace, grad_scale, dbias, N,
                       nblocks, accumulate);
    FVA_LAUNCH_CHECK("head_bias_final_kernel");
    return FVA_OK;
}
