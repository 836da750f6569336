// RoIAlign forward / backward for gfx950 (scope row f-4, first operator of the two-stage head).
//
// Replaces torchvision.ops.roi_align as the reference calls it (demos/faster_rcnn/models/fast.py:227-231,258):
// boxes [K][5] = (batch index, x1, y1, x2, y2) in feature cells, output_size (PH, PW), spatial_scale, adaptive sampling grid
// (sampling_ratio <= 0: ceil(roi_size / pooled) samples per bin and axis), aligned = False.  Arithmetic follows torchvision's
// roi_align_kernel (see oracle/roi_align.py), in fp32.
//
// Layout: the feature map is this library's halo NHWC tensor [B][H+2p][W+2p][C] (fp32 or bf16): the four taps of a sample
// are four contiguous channel vectors, so a wave reads whole lines -- one block per (box, 64-channel slice, output row), lane = channel.
// The result leaves in the reference's [K][C][PH][PW] order (what torch.flatten(.., 1) feeds the classifier): the block's
// [64][PW] rows are written from an LDS transpose (PW floats per channel and row of bins).  HBM/L2-bound gather; latency-bound for the
// K <= a few thousand boxes of one step.
// Backward: the same walk scatters grad / count * weight into a dense fp32 NHWC gradient with float atomics (channel-
// contiguous, so a wave's atomics hit consecutive addresses); the sum order is not fixed -- fp32 rounding differences only.
#include "common.h"

namespace {

struct RoiParams {
    const void* feat;
    const float* rois;
    float* out;        // fwd: [K][C][PH][PW]; bwd: grad_out, same layout (read)
    float* dfeat;      // bwd: [B][H][W][C] fp32, zeroed by the caller
    int B, H, W, C, pad;
    int K, PH, PW;
    float scale;
    int sampling;
};

constexpr int RC = 64;   // channels per block

template <typename T>
__device__ __forceinline__ float ldf(const T* p) { return (float)*p; }

// the sample walk of one bin, shared by both directions: f(y_low, x_low, y_high, x_high, w1..w4)
template <typename F>
__device__ __forceinline__ void roi_bin_samples(const RoiParams& p, float x1, float y1, float bin_w, float bin_h, int grid_w, int grid_h,
                                                int ph, int pw, F&& f) {
    for (int iy = 0; iy < grid_h; ++iy) {
        float y = y1 + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)grid_h;
        for (int ix = 0; ix < grid_w; ++ix) {
            float x = x1 + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)grid_w;
            float yy = y;
            if (yy < -1.0f || yy > (float)p.H || x < -1.0f || x > (float)p.W) continue;
            if (yy <= 0.f) yy = 0.f;
            if (x <= 0.f) x = 0.f;
            int y_low = (int)yy, x_low = (int)x, y_high, x_high;
            if (y_low >= p.H - 1) { y_high = y_low = p.H - 1; yy = (float)y_low; } else y_high = y_low + 1;
            if (x_low >= p.W - 1) { x_high = x_low = p.W - 1; x = (float)x_low; } else x_high = x_low + 1;
            const float ly = yy - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
            f(y_low, x_low, y_high, x_high, hy * hx, hy * lx, ly * hx, ly * lx);
        }
    }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void roi_align_kernel(const RoiParams p) {
    __shared__ float tile[RC][50];   // forward: [channel][bin] (+1 pad); PH*PW <= 49 per pass
    const int k = blockIdx.x, c0 = blockIdx.y * RC;
    const int lane_c = threadIdx.x & (RC - 1), sub = threadIdx.x >> 6;   // 4 bins in parallel
    const int c = c0 + lane_c;
    const float* roi = p.rois + (int64_t)k * 5;
    const int b = (int)roi[0];
    if (b < 0 || b >= p.B) return;                                       // uniform per block
    const float x1 = roi[1] * p.scale, y1 = roi[2] * p.scale, x2 = roi[3] * p.scale, y2 = roi[4] * p.scale;
    const float roi_w = fmaxf(x2 - x1, 1.f), roi_h = fmaxf(y2 - y1, 1.f);
    const float bin_h = roi_h / (float)p.PH, bin_w = roi_w / (float)p.PW;
    // adaptive sampling grid (ceil(roi / bins) samples per bin and axis), bounded: a box with a non-finite or absurd extent -- user
    // input, the RPN clamps its own proposals to the map -- must not make the kernel walk 10^9 samples.  64 samples per bin and axis
    // cover boxes of 448 cells at 7 bins; larger boxes are sampled on the 64-grid (a deviation from torchvision only out there).
    constexpr int MAX_GRID = 64;
    const bool finite = isfinite(x1) && isfinite(y1) && isfinite(x2) && isfinite(y2);
    int grid_h = p.sampling > 0 ? p.sampling : (finite ? (int)ceilf(fminf(roi_h, 1e6f) / (float)p.PH) : 1);
    int grid_w = p.sampling > 0 ? p.sampling : (finite ? (int)ceilf(fminf(roi_w, 1e6f) / (float)p.PW) : 1);
    grid_h = grid_h > MAX_GRID ? MAX_GRID : grid_h;
    grid_w = grid_w > MAX_GRID ? MAX_GRID : grid_w;
    const float inv_count = 1.f / (float)max(grid_h * grid_w, 1);
    const int Hp = p.H + 2 * p.pad, Wp = p.W + 2 * p.pad;
    const int nbins = p.PH * p.PW;
    const bool live = c < p.C;
    const int ph = blockIdx.z;                                            // one output row of bins per block
    for (int pw0 = 0; pw0 < p.PW; pw0 += 49) {
        const int nb = min(49, p.PW - pw0);
        for (int bi = sub; bi < nb; bi += 4) {
            const int pw = pw0 + bi, bin = ph * p.PW + pw;
            if constexpr (!BWD) {
                float acc = 0.f;
                if (live) {
                    const T* fm = (const T*)p.feat + ((int64_t)b * Hp * Wp) * p.C + c;
                    roi_bin_samples(p, x1, y1, bin_w, bin_h, grid_w, grid_h, ph, pw,
                                    [&](int yl, int xl, int yh, int xh, float w1, float w2, float w3, float w4) {
                                        const int64_t r0 = (int64_t)(yl + p.pad) * Wp + p.pad, r1 = (int64_t)(yh + p.pad) * Wp + p.pad;
                                        acc += w1 * ldf(fm + (r0 + xl) * p.C) + w2 * ldf(fm + (r0 + xh) * p.C) +
                                               w3 * ldf(fm + (r1 + xl) * p.C) + w4 * ldf(fm + (r1 + xh) * p.C);
                                    });
                }
                tile[lane_c][bi] = acc * inv_count;
            } else {
                if (live) {
                    const float go = p.out[((int64_t)k * p.C + c) * nbins + bin] * inv_count;
                    float* g = p.dfeat + ((int64_t)b * p.H * p.W) * p.C + c;
                    roi_bin_samples(p, x1, y1, bin_w, bin_h, grid_w, grid_h, ph, pw,
                                    [&](int yl, int xl, int yh, int xh, float w1, float w2, float w3, float w4) {
                                        atomicAdd(g + ((int64_t)yl * p.W + xl) * p.C, go * w1);
                                        atomicAdd(g + ((int64_t)yl * p.W + xh) * p.C, go * w2);
                                        atomicAdd(g + ((int64_t)yh * p.W + xl) * p.C, go * w3);
                                        atomicAdd(g + ((int64_t)yh * p.W + xh) * p.C, go * w4);
                                    });
                }
            }
        }
        if constexpr (!BWD) {
            __syncthreads();
            // out[(k*C + c) * nbins + ph*PW + pw]: per channel a run of nb floats
            float* o = p.out + ((int64_t)k * p.C + c0) * nbins + ph * p.PW + pw0;
            const int cmax = min(RC, p.C - c0);
            for (int e = threadIdx.x; e < cmax * nb; e += 256) {
                const int cc = e / nb, bi = e - cc * nb;
                o[(int64_t)cc * nbins + bi] = tile[cc][bi];
            }
            __syncthreads();
        }
    }
}

int check(const char* who, int dtype, const void* feat, const float* rois, const float* io, int B, int H, int W, int C, int pad, int K, int PH, int PW) {
    if (dtype != FVA_F32 && dtype != FVA_BF16) return fva_fail(FVA_ERR_ARG, "%s: bad dtype", who);
    if (K < 0 || B <= 0 || H <= 0 || W <= 0 || C <= 0 || pad < 0 || PH <= 0 || PW <= 0) return fva_fail(FVA_ERR_ARG, "%s: bad shape", who);
    if (K > 0 && (!feat || !rois || !io)) return fva_fail(FVA_ERR_ARG, "%s: null pointer", who);
    if (K > 65535 * 1024 || PH > 65535) return fva_fail(FVA_ERR_ARG, "%s: too many boxes / bins", who);
    return FVA_OK;
}

}  // namespace

extern "C" {

int fva_roi_align_fwd(int dtype, const void* feat, int feat_pad, const float* rois, int K, float* out, int B, int H, int W, int C, int PH,
                      int PW, float spatial_scale, int sampling_ratio, void* stream) {
    int rc = check("fva_roi_align_fwd", dtype, feat, rois, out, B, H, W, C, feat_pad, K, PH, PW);
    if (rc || K == 0) return rc;
    RoiParams p{};
    p.feat = feat; p.rois = rois; p.out = out; p.B = B; p.H = H; p.W = W; p.C = C; p.pad = feat_pad; p.K = K; p.PH = PH; p.PW = PW;
    p.scale = spatial_scale; p.sampling = sampling_ratio;
    const dim3 grid(K, cdiv(C, RC), PH);
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL((roi_align_kernel<bf16_t, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((roi_align_kernel<float, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    FVA_LAUNCH_CHECK("roi_align_kernel");
    return FVA_OK;
}

int fva_roi_align_bwd(const float* grad_out, const float* rois, int K, float* dfeat, int B, int H, int W, int C, int PH, int PW,
                      float spatial_scale, int sampling_ratio, void* stream) {
    int rc = check("fva_roi_align_bwd", FVA_F32, dfeat, rois, grad_out, B, H, W, C, 0, K, PH, PW);
    if (rc || K == 0) return rc;
    RoiParams p{};
    p.rois = rois; p.out = const_cast<float*>(grad_out); p.dfeat = dfeat; p.B = B; p.H = H; p.W = W; p.C = C; p.pad = 0; p.K = K;
    p.PH = PH; p.PW = PW; p.scale = spatial_scale; p.sampling = sampling_ratio;
    hipLaunchKernelGGL((roi_align_kernel<float, true>), dim3(K, cdiv(C, RC), PH), dim3(256), 0, (hipStream_t)stream, p);
    FVA_LAUNCH_CHECK("roi_align_kernel<bwd>");
    return FVA_OK;
}

}  // extern "C"
