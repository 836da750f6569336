// Detection-head backward glue for gfx950: turns the fp32 head gradient [B*H*W][N] (N = A*(5+C) = 255, odd
// pitch) into the halo NHWC operand the MFMA dgrad/wgrad kernels consume (N padded to Npad, zero border), applies
// the upstream scalar gradient, and reduces the bias gradient deterministically.  HBM-bound, small.
//
// Replaces the autograd backward of the biased 1x1 nn.Conv2d heads (reference detection/head/yolov3head.py:50,60).
#include "common.h"

namespace {

// dY (halo, Npad channels, zero beyond N and on the border) = g * dhead (dense fp32 rows of N = 255 floats: not 16-byte aligned).
// One thread per 8 output channels: eight 4-byte loads (a wave covers whole cache lines of the row), one 16- or 32-byte store.
// `partial` (may be null): the bias gradient's first stage rides along -- a thread keeps its 8-channel chunk for the whole grid-stride loop
// (the chunks per pixel divide 256), adds up what it reads, and the block leaves partial[blockIdx][n] = the sum over its pixels: dhead is
// read once instead of twice (the separate pass read 274 MB again at the very start of the backward pass, where nothing runs beside it).
template <typename T>
__global__ __launch_bounds__(256) void head_prepare_kernel(const float* __restrict__ dhead, const float* __restrict__ gscale,
                                                           T* __restrict__ dy, int B, int H, int W, int N, int Npad, float* __restrict__ partial) {
    const uint32_t Hp = H + 2, Wp = W + 2, cpp = Npad / 8;
    const uint32_t total = (uint32_t)B * Hp * Wp * cpp;
    const float g = gscale ? *gscale : 1.f;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t pix = i / cpp, cc = i - pix * cpp;
        const uint32_t row = pix / Wp, xp = pix - row * Wp;
        const uint32_t b = row / Hp, yp = row - b * Hp;
        const int y = (int)yp - 1, x = (int)xp - 1;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const float* src = dhead + (((int64_t)b * H + y) * W + x) * N + cc * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if ((int)(cc * 8 + e) < N) {
                    const float r = src[e];
                    acc[e] += r;
                    v[e] = g * r;
                }
        }
        T* dst = dy + (int64_t)i * 8;
        if constexpr (sizeof(T) == 2) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
            *(bf16x8*)dst = o;
        } else {
            *(f32x4*)dst = f32x4{v[0], v[1], v[2], v[3]};
            *(f32x4*)(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
    }
    if (partial != nullptr) {       // host: 256 % cpp == 0, so thread t owns chunk t % cpp throughout
        __shared__ float red[256 * 8];
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = acc[e];
        __syncthreads();
        const int groups = 256 / (int)cpp;
        for (int n = threadIdx.x; n < N; n += 256) {
            const int cc = n >> 3, e = n & 7;
            float t = 0.f;
            for (int k = 0; k < groups; ++k) t += red[(k * (int)cpp + cc) * 8 + e];      // fixed order
            partial[(int64_t)blockIdx.x * N + n] = t;
        }
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
    const int64_t total = (int64_t)B * (H + 2) * (W + 2) * (Npad / 8);   // items of 8 channels
    if (total >= (1ll << 31)) return fva_fail(FVA_ERR_ARG, "fva_head_bwd_prepare: tensor too large");
    int64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    const int cpp = Npad / 8;
    const bool ride = 256 % cpp == 0;           // the bias sums ride in the prepare pass (workspace: one row of N floats per block, <= 4096 rows)
    float* part = (float*)workspace;
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL(head_prepare_kernel<bf16_t>, dim3((int)g), dim3(256), 0, s, dhead, grad_scale, (bf16_t*)dy, B, H, W, N, Npad, ride ? part : nullptr);
    else if (dtype == FVA_F32)
        hipLaunchKernelGGL(head_prepare_kernel<float>, dim3((int)g), dim3(256), 0, s, dhead, grad_scale, (float*)dy, B, H, W, N, Npad, ride ? part : nullptr);
    else
        return fva_fail(FVA_ERR_ARG, "fva_head_bwd_prepare: bad dtype");
    FVA_LAUNCH_CHECK("head_prepare_kernel");
    int nblocks = (int)g;
    if (!ride) {
        const int64_t M = (int64_t)B * H * W;
        nblocks = (int)((M + 63) / 64);
        if (nblocks > 1024) nblocks = 1024;
        const int rows = (int)((M + nblocks - 1) / nblocks);
        nblocks = (int)((M + rows - 1) / rows);
        hipLaunchKernelGGL(head_bias_partial_kernel, dim3(nblocks), dim3(256), 0, s, dhead, part, M, N, rows);
        FVA_LAUNCH_CHECK("head_bias_partial_kernel");
    }
    hipLaunchKernelGGL(head_bias_final_kernel, dim3(N), dim3(256), 0, s, (const float*)workspace, grad_scale, dbias, N,
                       nblocks, accumulate);
    FVA_LAUNCH_CHECK("head_bias_final_kernel");
    return FVA_OK;
}
