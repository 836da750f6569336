// Colour and blur extras of the demo's training image path for gfx950 (scope row f-3, remainder): CLAHE on the luma plane,
// the HSV look-up-table jitter, the 3x3 blur family, channel shuffle and the final "/ 255" -- byte / integer work on uint8
// canvases that are already resident in HBM (the output of fva_paste_resize_u8), HBM- and latency-bound.
//
// Replaces HistEqualize (cv2.cvtColor RGB<->YUV + cv2.createCLAHE(2.0, (8, 8)).apply), HueSaturationValue (cv2.cvtColor RGB<->HSV +
// cv2.LUT) of demos/yolov3_u/data_gen.py:120-146 and the albumentations Compose of :26-33 (OneOf[Blur, MedianBlur, GaussianBlur] at
// 3x3, ChannelShuffle, ToTensorV2) + `image / 255.` (:352-353).  The arithmetic restates OpenCV 4.5's 8-bit code paths; the CPU
// statement of the same arithmetic is oracle/colour.py (parity unpinned: OpenCV is absent from the reference tree and this image).
#include "common.h"

namespace {

__device__ __forceinline__ int descale14(int x) { return (x + (1 << 13)) >> 14; }
__device__ __forceinline__ int sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int luma(int r, int g, int b) { return sat8(descale14(r * 4899 + g * 9617 + b * 1868)); }
__device__ __forceinline__ int reflect101(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// ---- CLAHE, stage 1: one block per (image, tile): histogram of the luma of the tile (image extended by reflection to multiples
// of 8), clipped and redistributed, cumulative -> 256-byte LUT.  lut: [B][64][256]
__global__ __launch_bounds__(256) void clahe_lut_kernel(const uint8_t* __restrict__ img, const fva_colour_job* __restrict__ jobs, int H,
                                                        int W, uint8_t* __restrict__ lut) {
    __shared__ int hist[256];
    __shared__ int scan[256];
    const int b = blockIdx.y, tile = blockIdx.x, ty = tile >> 3, tx = tile & 7;
    const fva_colour_job jb = jobs[b];
    if (!jb.clahe) return;
    const int h = jb.h, w = jb.w;
    const bool fits = (h % 8 == 0) && (w % 8 == 0);      // clahe.cpp: otherwise BOTH axes grow by 8 - size % 8 (a full 8 if it divides)
    const int th = fits ? h / 8 : (h + 8 - h % 8) / 8, tw = fits ? w / 8 : (w + 8 - w % 8) / 8;
    const int area = th * tw;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t* base = img + (int64_t)b * H * W * 3;
    for (int i = threadIdx.x; i < area; i += 256) {
        const int y = reflect101(ty * th + i / tw, h), x = reflect101(tx * tw + i % tw, w);
        const uint8_t* p = base + ((int64_t)y * W + x) * 3;
        atomicAdd(&hist[luma(p[0], p[1], p[2])], 1);
    }
    __syncthreads();
    int clip = (int)(2.0 * (double)area / 256.0);
    clip = clip < 1 ? 1 : clip;
    const int mine = hist[threadIdx.x];
    scan[threadIdx.x] = mine > clip ? mine - clip : 0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {                  // total excess
        if ((int)threadIdx.x < o) scan[threadIdx.x] += scan[threadIdx.x + o];
        __syncthreads();
    }
    const int excess = scan[0];
    __syncthreads();
    const int batch = excess / 256, rest = excess % 256;
    int v = (mine > clip ? clip : mine) + batch;
    if (rest) {
        const int step = 256 / rest > 1 ? 256 / rest : 1;
        if ((int)threadIdx.x % step == 0 && (int)threadIdx.x / step < rest) v += 1;   // bins 0, step, 2*step, ... (rest of them)
    }
    scan[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                  // inclusive prefix sum
        const int add = (int)threadIdx.x >= o ? scan[threadIdx.x - o] : 0;
        __syncthreads();
        scan[threadIdx.x] += add;
        __syncthreads();
    }
    const float scale = 255.0f / (float)area;
    lut[((int64_t)b * 64 + tile) * 256 + threadIdx.x] = (uint8_t)sat8(__float2int_rn((float)scan[threadIdx.x] * scale));
}

__device__ __forceinline__ void rgb2hsv(int r, int g, int b, int& h, int& s, int& v) {
    v = max(max(r, g), b);
    const int diff = v - min(min(r, g), b);
    const int sdiv = v ? __double2int_rn((double)(255 << 12) / (double)v) : 0;
    const int hdiv = diff ? __double2int_rn((double)(180 << 12) / (6.0 * (double)diff)) : 0;
    s = (diff * sdiv + (1 << 11)) >> 12;
    int hh = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    hh = (hh * hdiv + (1 << 11)) >> 12;
    h = sat8(hh < 0 ? hh + 180 : hh);
    s = sat8(s);
}
__device__ __forceinline__ void hsv2rgb(int H, int S, int V, int& r, int& g, int& b) {
    const float h = (float)H * (float)(6.0 / 180.0), s = (float)S * (float)(1.0 / 255.0), v = (float)V * (float)(1.0 / 255.0);
    float fr, fg, fb;
    if (S == 0) {
        fr = fg = fb = v;
    } else {
        int sector = (int)floorf(h);
        float f = h - (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; f = 0.f; }
        const float t0 = v, t1 = v * (1.f - s), t2 = v * (1.f - s * f), t3 = v * (1.f - s * (1.f - f));
        const float tab[4] = {t0, t1, t2, t3};
        const int sb[6] = {1, 1, 3, 0, 0, 2}, sg[6] = {3, 0, 0, 2, 1, 1}, sr[6] = {0, 2, 1, 1, 3, 0};
        fb = tab[sb[sector]]; fg = tab[sg[sector]]; fr = tab[sr[sector]];
    }
    r = sat8(__float2int_rn(fr * 255.f)); g = sat8(__float2int_rn(fg * 255.f)); b = sat8(__float2int_rn(fb * 255.f));
}

// ---- stage 2, per pixel of the valid region, in place: [CLAHE on Y of YUV] then [HSV LUTs]
__global__ __launch_bounds__(256) void clahe_hsv_apply_kernel(uint8_t* __restrict__ img, const fva_colour_job* __restrict__ jobs, int H, int W,
                                                              const uint8_t* __restrict__ clahe_lut, const uint8_t* __restrict__ hsv_lut) {
    const int b = blockIdx.y;
    const fva_colour_job jb = jobs[b];
    if (!jb.clahe && !jb.hsv) return;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= jb.h * jb.w) return;
    const int y = e / jb.w, x = e - y * jb.w;
    uint8_t* p = img + ((int64_t)b * H * W + (int64_t)y * W + x) * 3;
    int r = p[0], g = p[1], bl = p[2];
    if (jb.clahe) {
        const int Y = luma(r, g, bl);
        const int U = sat8(descale14((bl - Y) * 8061 + (128 << 14))), V = sat8(descale14((r - Y) * 14369 + (128 << 14)));
        const bool fits = (jb.h % 8 == 0) && (jb.w % 8 == 0);
        const int th = fits ? jb.h / 8 : (jb.h + 8 - jb.h % 8) / 8, tw = fits ? jb.w / 8 : (jb.w + 8 - jb.w % 8) / 8;
        const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
        const float xf = (float)x * inv_tw - 0.5f, yf = (float)y * inv_th - 0.5f;
        int tx1 = (int)floorf(xf), ty1 = (int)floorf(yf);
        const float xa = xf - (float)tx1, ya = yf - (float)ty1, xa1 = 1.0f - xa, ya1 = 1.0f - ya;
        const int tx2 = tx1 + 1 < 7 ? tx1 + 1 : 7, ty2 = ty1 + 1 < 7 ? ty1 + 1 : 7;
        tx1 = tx1 < 0 ? 0 : tx1;
        ty1 = ty1 < 0 ? 0 : ty1;
        const uint8_t* L = clahe_lut + (int64_t)b * 64 * 256 + Y;
        const float l11 = L[(ty1 * 8 + tx1) * 256], l12 = L[(ty1 * 8 + tx2) * 256], l21 = L[(ty2 * 8 + tx1) * 256], l22 = L[(ty2 * 8 + tx2) * 256];
        // the products and sums are rounded to float32 one by one, as the CPU statement does (no FMA contraction: -ffp-contract=off)
        const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
        const int Y2 = sat8(__float2int_rn(res));
        const int u = U - 128, v = V - 128;
        r = sat8(Y2 + descale14(v * 18678));
        g = sat8(Y2 + descale14(u * -6472 + v * -9519));
        bl = sat8(Y2 + descale14(u * 33292));
    }
    if (jb.hsv) {
        int h, s, v;
        rgb2hsv(r, g, bl, h, s, v);
        const uint8_t* T = hsv_lut + (int64_t)b * 768;
        hsv2rgb(T[h], T[256 + s], T[512 + v], r, g, bl);
    }
    p[0] = (uint8_t)r; p[1] = (uint8_t)g; p[2] = (uint8_t)bl;
}

// ---- blur (0 none / 1 box / 2 median / 3 Gaussian, 3x3) + channel shuffle + value table -> planar fp32
__device__ __forceinline__ void sort2(int& a, int& b) {
    const int lo = min(a, b), hi = max(a, b);
    a = lo; b = hi;
}
__device__ __forceinline__ int median9(int (&p)[9]) {
    sort2(p[1], p[2]); sort2(p[4], p[5]); sort2(p[7], p[8]); sort2(p[0], p[1]); sort2(p[3], p[4]); sort2(p[6], p[7]);
    sort2(p[1], p[2]); sort2(p[4], p[5]); sort2(p[7], p[8]); sort2(p[0], p[3]); sort2(p[5], p[8]); sort2(p[4], p[7]);
    sort2(p[3], p[6]); sort2(p[1], p[4]); sort2(p[2], p[5]); sort2(p[4], p[7]); sort2(p[4], p[2]); sort2(p[6], p[4]);
    sort2(p[4], p[2]);
    return p[4];
}
__global__ __launch_bounds__(256) void blur_shuffle_kernel(const uint8_t* __restrict__ img, const fva_colour_job* __restrict__ jobs, int H, int W,
                                                           const float* __restrict__ lut, float* __restrict__ out) {
    __shared__ float slut[768];
    for (int i = threadIdx.x; i < 768; i += 256) slut[i] = lut[i];
    __syncthreads();
    const int b = blockIdx.y;
    const fva_colour_job jb = jobs[b];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= H * W) return;
    const int y = e / W, x = e - y * W;
    const uint8_t* base = img + (int64_t)b * H * W * 3;
    int v[3];
    if (jb.blur == 0) {
        const uint8_t* p = base + (int64_t)e * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
    } else {
        int ys[3], xs[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (jb.blur == 2) {          // median: replicated border
                ys[d] = min(max(y + d - 1, 0), H - 1);
                xs[d] = min(max(x + d - 1, 0), W - 1);
            } else {
                ys[d] = reflect101(y + d - 1, H);
                xs[d] = reflect101(x + d - 1, W);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int p[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) p[dy * 3 + dx] = base[((int64_t)ys[dy] * W + xs[dx]) * 3 + c];
            if (jb.blur == 1) {
                int s = 0;
#pragma unroll
                for (int i = 0; i < 9; ++i) s += p[i];
                v[c] = sat8(__float2int_rn((float)s * (float)(1.0 / 9.0)));
            } else if (jb.blur == 2) {
                v[c] = median9(p);
            } else {
                v[c] = (p[0] + 2 * p[1] + p[2] + 2 * p[3] + 4 * p[4] + 2 * p[5] + p[6] + 2 * p[7] + p[8] + 8) >> 4;
            }
        }
    }
    const int64_t plane = (int64_t)H * W;
    float* o = out + (int64_t)b * 3 * plane + e;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int src_c = jb.perm[c];
        o[c * plane] = slut[c * 256 + (src_c == 0 ? v[0] : (src_c == 1 ? v[1] : v[2]))];
    }
}

}  // namespace

extern "C" int64_t fva_colour_workspace(int32_t B) { return (int64_t)(B > 0 ? B : 0) * 64 * 256; }

extern "C" int fva_colour_clahe_hsv(uint8_t* canvases, int32_t B, int32_t H, int32_t W, const fva_colour_job* jobs, int32_t max_h,
                                    int32_t max_w, const uint8_t* hsv_luts, uint8_t* workspace, void* stream) {
    if (!canvases || !jobs || !workspace || B < 1 || B > 65535 || H < 1 || W < 1 || max_h < 1 || max_w < 1 || max_h > H || max_w > W)
        return fva_fail(FVA_ERR_ARG, "fva_colour_clahe_hsv: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(clahe_lut_kernel, dim3(64, B), dim3(256), 0, s, (const uint8_t*)canvases, jobs, H, W, workspace);
    FVA_LAUNCH_CHECK("clahe_lut_kernel");
    hipLaunchKernelGGL(clahe_hsv_apply_kernel, dim3(cdiv((int64_t)max_h * max_w, 256), B), dim3(256), 0, s, canvases, jobs, H, W,
                       (const uint8_t*)workspace, hsv_luts);
    FVA_LAUNCH_CHECK("clahe_hsv_apply_kernel");
    return FVA_OK;
}

extern "C" int fva_colour_blur_shuffle_normalize(const uint8_t* canvases, int32_t B, int32_t H, int32_t W, const fva_colour_job* jobs,
                                                 const float* lut, float* out, void* stream) {
    if (!canvases || !jobs || !lut || !out || B < 1 || B > 65535 || H < 2 || W < 2 || (int64_t)H * W >= (1ll << 31))
        return fva_fail(FVA_ERR_ARG, "fva_colour_blur_shuffle_normalize: bad argument");
    hipLaunchKernelGGL(blur_shuffle_kernel, dim3(cdiv((int64_t)H * W, 256), B), dim3(256), 0, (hipStream_t)stream, canvases, jobs, H, W, lut, out);
    FVA_LAUNCH_CHECK("blur_shuffle_kernel");
    return FVA_OK;
}
