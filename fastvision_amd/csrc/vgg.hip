// Backbone blocks of the two-stage head (scope row f-4): the reference's Faster R-CNN runs on a plain VGG16
// (demos/faster_rcnn/models/vgg.py: Conv2d 3x3 + bias -> ReLU, MaxPool2d(2, 2)), i.e. no BatchNorm.  The convolution itself is
// the implicit-GEMM kernel with the bias + ReLU epilogue (fva_conv_fwd_bias_act, conv_igemm.hip) and the existing dgrad / wgrad
// entries; this file holds what a bias + ReLU block needs around them, all HBM-bound elementwise passes over NHWC tensors:
//   bias_relu_bwd   dY = dZ * (Z > 0) into the halo buffer dgrad and wgrad read, + per-row-block partial sums of dY (= dbias)
//   colsum          fixed-order reduction of those partials (deterministic)
//   maxpool2_fwd    2x2 / stride 2 max into the next block's halo input (floor mode, like nn.MaxPool2d(2, 2))
//   maxpool2_bwd    the gradient goes to the first maximum of each window in scan order (torch's choice)
#include "common.h"

namespace {

// one block per padded row of the output halo buffer; lane-constant channel chunk (cpp divides 256)
template <typename T>
__global__ __launch_bounds__(256) void bias_relu_bwd_kernel(const T* __restrict__ dz, const T* __restrict__ z, int z_pad, T* __restrict__ dy,
                                                            int pad, float* __restrict__ part, int B, int H, int W, int C) {
    constexpr int EPC = Vec16<T>::N;
    extern __shared__ float red[];   // [256 / cpp][C]
    const int cpp = C / EPC, rpi = 256 / cpp;
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int b = blockIdx.x / Hp, yp = blockIdx.x - b * Hp, yy = yp - pad;
    const bool row_in = yy >= 0 && yy < H;
    const int cc = threadIdx.x % cpp, px0 = threadIdx.x / cpp;
    T* orow = dy + (int64_t)blockIdx.x * Wp * C;
    const int zWp = W + 2 * z_pad;
    const T* zrow = z + (((int64_t)b * (H + 2 * z_pad) + yy + z_pad) * zWp + z_pad) * C;
    const T* grow = dz + ((int64_t)b * H + yy) * W * C;
    float s[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = 0.f;
    for (int xp = px0; xp < Wp; xp += rpi) {
        const int xx = xp - pad;
        Vec16<T> out;
        if (row_in && xx >= 0 && xx < W) {
            const Vec16<T> g = *(const Vec16<T>*)(grow + (int64_t)xx * C + cc * EPC);
            const Vec16<T> v = *(const Vec16<T>*)(zrow + (int64_t)xx * C + cc * EPC);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float d = v.get(e) > 0.f ? g.get(e) : 0.f;
                out.set(e, d);
                s[e] += d;
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPC; ++e) out.set(e, 0.f);
        }
        *(Vec16<T>*)(orow + (int64_t)xp * C + cc * EPC) = out;
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[px0 * C + cc * EPC + e] = s[e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float t = 0.f;
        for (int k = 0; k < rpi; ++k) t += red[k * C + c];
        part[(int64_t)blockIdx.x * C + c] = t;
    }
}

// column sums of rows [r0, r1) of part[rows][C], one block per (64 channels, row group); fixed order inside a block
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ part, int rows, int C, float* __restrict__ out, int rows_per_block) {
    __shared__ double red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    double t = 0.0;
    if (c < C)
        for (int r = r0 + g; r < r1; r += 4) t += (double)part[(int64_t)r * C + c];
    red[g][threadIdx.x & 63] = t;
    __syncthreads();
    if (g == 0 && c < C)
        out[(int64_t)blockIdx.y * C + c] = (float)((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// one block per padded output row
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, int x_pad, T* __restrict__ out, int pad, int B, int H, int W,
                                                           int C) {
    constexpr int EPC = Vec16<T>::N;
    const int OH = H / 2, OW = W / 2, cpp = C / EPC;
    const int Hp = OH + 2 * pad, Wp = OW + 2 * pad, xWp = W + 2 * x_pad;
    const int b = blockIdx.x / Hp, yp = blockIdx.x - b * Hp, oy = yp - pad;
    const bool row_in = oy >= 0 && oy < OH;
    T* orow = out + (int64_t)blockIdx.x * Wp * C;
    const T* xr0 = x + (((int64_t)b * (H + 2 * x_pad) + 2 * oy + x_pad) * xWp + x_pad) * C;
    for (int i = threadIdx.x; i < Wp * cpp; i += 256) {
        const int xp = i / cpp, cc = i - xp * cpp, ox = xp - pad;
        Vec16<T> o;
        if (row_in && ox >= 0 && ox < OW) {
            const T* p00 = xr0 + (int64_t)(2 * ox) * C + cc * EPC;
            const Vec16<T> a = *(const Vec16<T>*)p00, bq = *(const Vec16<T>*)(p00 + C), c2 = *(const Vec16<T>*)(p00 + (int64_t)xWp * C),
                           d2 = *(const Vec16<T>*)(p00 + (int64_t)xWp * C + C);
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.set(e, fmaxf(fmaxf(a.get(e), bq.get(e)), fmaxf(c2.get(e), d2.get(e))));
        } else {
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.set(e, 0.f);
        }
        *(Vec16<T>*)(orow + (int64_t)xp * C + cc * EPC) = o;
    }
}

// one block per input row: dx[b][y][x][c] = dz[b][y/2][x/2][c] if (y, x) is the first maximum of its window, else 0
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ dz, const T* __restrict__ x, int x_pad, T* __restrict__ dx, int B,
                                                           int H, int W, int C) {
    constexpr int EPC = Vec16<T>::N;
    const int OH = H / 2, OW = W / 2, cpp = C / EPC, xWp = W + 2 * x_pad;
    const int b = blockIdx.x / H, y = blockIdx.x - b * H, oy = y >> 1;
    T* orow = dx + (int64_t)blockIdx.x * W * C;
    const T* xr0 = x + (((int64_t)b * (H + 2 * x_pad) + 2 * oy + x_pad) * xWp + x_pad) * C;   // first row of the window
    for (int i = threadIdx.x; i < W * cpp; i += 256) {
        const int xx = i / cpp, cc = i - xx * cpp, ox = xx >> 1;
        Vec16<T> o;
        if (oy < OH && ox < OW) {
            const T* p00 = xr0 + (int64_t)(2 * ox) * C + cc * EPC;
            const Vec16<T> a = *(const Vec16<T>*)p00, bq = *(const Vec16<T>*)(p00 + C), c2 = *(const Vec16<T>*)(p00 + (int64_t)xWp * C),
                           d2 = *(const Vec16<T>*)(p00 + (int64_t)xWp * C + C);
            const Vec16<T> g = *(const Vec16<T>*)(dz + (((int64_t)b * OH + oy) * OW + ox) * C + cc * EPC);
            const int me = (y & 1) * 2 + (xx & 1);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float v0 = a.get(e), v1 = bq.get(e), v2 = c2.get(e), v3 = d2.get(e);
                int arg = 0;
                float m = v0;
                if (v1 > m) { m = v1; arg = 1; }
                if (v2 > m) { m = v2; arg = 2; }
                if (v3 > m) { m = v3; arg = 3; }
                o.set(e, arg == me ? g.get(e) : 0.f);
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.set(e, 0.f);
        }
        *(Vec16<T>*)(orow + (int64_t)xx * C + cc * EPC) = o;
    }
}

// Fully connected layers of the Fast head (demos/faster_rcnn/models/vgg.py classifier: Linear -> ReLU): dY = dZ * (Z > 0) over rows
// [R][C] (any C that is a multiple of the 16-byte chunk) and, per block of 32 rows, the column sums of dY (= partial dbias).
template <typename T>
__global__ __launch_bounds__(256) void rows_relu_bwd_kernel(const T* __restrict__ dz, const T* __restrict__ z, T* __restrict__ dy,
                                                            float* __restrict__ partial, int R, int C, int relu, int64_t dy_stride) {
    constexpr int EPC = Vec16<T>::N;
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc * EPC >= C) return;
    const int r0 = blockIdx.y * 32, r1 = min(R, r0 + 32);
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int r = r0; r < r1; ++r) {
        const int64_t off = (int64_t)r * C + cc * EPC;
        const Vec16<T> g = *(const Vec16<T>*)(dz + off);
        Vec16<T> o;
        if (relu) {
            const Vec16<T> zz = *(const Vec16<T>*)(z + off);
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.set(e, zz.get(e) > 0.f ? g.get(e) : 0.f);
        } else {
            o = g;
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += o.get(e);
        *(Vec16<T>*)(dy + (int64_t)r * dy_stride + cc * EPC) = o;
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) partial[(int64_t)blockIdx.y * C + cc * EPC + e] = acc[e];
}

int chan_ok(const char* who, int dtype, int C, bool need_pow2) {
    if (dtype != FVA_F32 && dtype != FVA_BF16) return fva_fail(FVA_ERR_ARG, "%s: bad dtype", who);
    const int epc = dtype == FVA_BF16 ? 8 : 4;
    if (C <= 0 || C % epc) return fva_fail(FVA_ERR_ARG, "%s: C = %d is not a multiple of %d", who, C, epc);
    if (need_pow2 && (C / epc > 256 || 256 % (C / epc))) return fva_fail(FVA_ERR_ARG, "%s: C = %d unsupported (16-byte chunks per pixel must divide 256)", who, C);
    return FVA_OK;
}

}  // namespace

extern "C" {

int32_t fva_bias_relu_bwd_rows(int B, int H, int dy_pad) { return B * (H + 2 * dy_pad); }

int fva_bias_relu_bwd(int dtype, const void* dz, const void* z, int z_pad, void* dy, int dy_pad, float* partial, int B, int H, int W, int C,
                      void* stream) {
    int rc = chan_ok("fva_bias_relu_bwd", dtype, C, true);
    if (rc) return rc;
    if (!dz || !z || !dy || !partial || B <= 0 || H <= 0 || W <= 0 || z_pad < 0 || dy_pad < 0) return fva_fail(FVA_ERR_ARG, "fva_bias_relu_bwd: bad argument");
    const int epc = dtype == FVA_BF16 ? 8 : 4;
    const int smem = (256 / (C / epc)) * C * 4;
    const dim3 grid(B * (H + 2 * dy_pad));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL(bias_relu_bwd_kernel<bf16_t>, grid, dim3(256), smem, s, (const bf16_t*)dz, (const bf16_t*)z, z_pad, (bf16_t*)dy, dy_pad,
                           partial, B, H, W, C);
    else
        hipLaunchKernelGGL(bias_relu_bwd_kernel<float>, grid, dim3(256), smem, s, (const float*)dz, (const float*)z, z_pad, (float*)dy, dy_pad, partial,
                           B, H, W, C);
    FVA_LAUNCH_CHECK("bias_relu_bwd_kernel");
    return FVA_OK;
}

int32_t fva_rows_relu_bwd_rows(int32_t R) { return cdiv(R, 32); }

int fva_rows_relu_bwd(int dtype, const void* dz, const void* z, void* dy, int64_t dy_stride, float* partial, int32_t R, int32_t C, int32_t relu,
                      void* stream) {
    int rc = chan_ok("fva_rows_relu_bwd", dtype, C, false);
    if (rc) return rc;
    if (!dz || (relu && !z) || !dy || !partial || R <= 0 || dy_stride < C) return fva_fail(FVA_ERR_ARG, "fva_rows_relu_bwd: bad argument");
    const int epc = dtype == FVA_BF16 ? 8 : 4;
    const dim3 grid(cdiv(C / epc, 256), cdiv(R, 32));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL(rows_relu_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dz, (const bf16_t*)z, (bf16_t*)dy, partial, R, C, relu, dy_stride);
    else
        hipLaunchKernelGGL(rows_relu_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)dz, (const float*)z, (float*)dy, partial, R, C, relu, dy_stride);
    FVA_LAUNCH_CHECK("rows_relu_bwd_kernel");
    return FVA_OK;
}

int32_t fva_colsum_scratch_rows(int32_t rows) { return rows > 256 ? cdiv(rows, 64) : 0; }

int fva_colsum(const float* partial, int32_t rows, int C, float* out, float* scratch, void* stream) {
    if (!partial || !out || rows <= 0 || C <= 0) return fva_fail(FVA_ERR_ARG, "fva_colsum: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int groups = fva_colsum_scratch_rows(rows);
    if (groups > 0 && scratch) {           // two passes: 64 rows per block, then the group sums (a single block per 64 channels otherwise)
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 64), groups), dim3(256), 0, s, partial, rows, C, scratch, 64);
        FVA_LAUNCH_CHECK("colsum_kernel");
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 64), 1), dim3(256), 0, s, scratch, groups, C, out, groups);
    } else {
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 64), 1), dim3(256), 0, s, partial, rows, C, out, rows);
    }
    FVA_LAUNCH_CHECK("colsum_kernel");
    return FVA_OK;
}

int fva_maxpool2_fwd(int dtype, const void* x, int x_pad, void* out, int out_pad, int B, int H, int W, int C, void* stream) {
    int rc = chan_ok("fva_maxpool2_fwd", dtype, C, false);
    if (rc) return rc;
    if (!x || !out || B <= 0 || H < 2 || W < 2 || x_pad < 0 || out_pad < 0) return fva_fail(FVA_ERR_ARG, "fva_maxpool2_fwd: bad argument");
    const dim3 grid(B * (H / 2 + 2 * out_pad));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL(maxpool2_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, x_pad, (bf16_t*)out, out_pad, B, H, W, C);
    else
        hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, grid, dim3(256), 0, s, (const float*)x, x_pad, (float*)out, out_pad, B, H, W, C);
    FVA_LAUNCH_CHECK("maxpool2_fwd_kernel");
    return FVA_OK;
}

int fva_maxpool2_bwd(int dtype, const void* dz, const void* x, int x_pad, void* dx, int B, int H, int W, int C, void* stream) {
    int rc = chan_ok("fva_maxpool2_bwd", dtype, C, false);
    if (rc) return rc;
    if (!dz || !x || !dx || B <= 0 || H < 2 || W < 2 || x_pad < 0) return fva_fail(FVA_ERR_ARG, "fva_maxpool2_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16_t>, dim3(B * H), dim3(256), 0, s, (const bf16_t*)dz, (const bf16_t*)x, x_pad, (bf16_t*)dx, B, H, W, C);
    else
        hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(B * H), dim3(256), 0, s, (const float*)dz, (const float*)x, x_pad, (float*)dx, B, H, W, C);
    FVA_LAUNCH_CHECK("maxpool2_bwd_kernel");
    return FVA_OK;
}

}  // extern "C"
