// Input side of the detection path for gfx950 (scope row f-3): decoded uint8 RGB images -> the network's input batch in
// ONE launch: bilinear resize with OpenCV's 8-bit fixed-point arithmetic, placement on a constant canvas (letterbox
// padding, or the four tiles of a mosaic), flips, per-channel value map (x/255, (x/255 - mean)/std: a 3x256 table
// computed by the caller exactly as the reference computes it), HWC -> planar CHW.  Byte/integer work, HBM-bound:
// ~3 source bytes read and 12 bytes written per output pixel.
//
// Replaces, per sample, cv2.resize + Padding/cv2.copyMakeBorder + np.fliplr/np.flipud + Normalization + transpose +
// torch.stack of the reference's BaseDataset.__getitem__ / collate_fn (datasets/detection_dataloader.py:44-103,
// datasets/common/padding.py, datasets/common/augmentation.py:298-376) and ResizeByMax / Padding / flips / Mosaic01 /
// `image / 255.` of demos/yolov3_u/data_gen.py:42-131,171-216,352-360.
#include "common.h"

namespace {

struct Tap {
    int s0, s1, w0, w1;
};

// OpenCV INTER_LINEAR, 8-bit: source taps and 11-bit weights of destination index d (resize.cpp, see oracle/pipeline.py)
__device__ __forceinline__ Tap make_tap(int d, double scale, int src) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
    Tap t;
    t.s0 = s;
    t.s1 = s + 1 < src ? s + 1 : src - 1;
    t.w1 = __float2int_rn(f * 2048.f);
    t.w0 = __float2int_rn((1.f - f) * 2048.f);
    return t;
}

// U8 = false: planar fp32 through the value table; U8 = true: interleaved uint8 canvas [B][H][W][3] (an intermediate image
// that a later launch resizes again, e.g. ResizeByMax before Mosaic01)
template <bool U8>
__global__ __launch_bounds__(256) void paste_kernel(const uint8_t* __restrict__ src, const fva_paste_job* __restrict__ jobs,
                                                    const int32_t* __restrict__ job_start, int H, int W, int fill,
                                                    const float* __restrict__ lut, void* __restrict__ out_) {
    __shared__ float slut[3 * 256];
    if constexpr (!U8) {
        for (int i = threadIdx.x; i < 768; i += 256) slut[i] = lut[i];
        __syncthreads();
    }
    const int b = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= H * W) return;
    const int y = e / W, x = e - y * W;
    int v[3] = {fill, fill, fill};
    // the last job that covers the pixel wins (jobs are pasted in order)
    for (int j = job_start[b + 1] - 1; j >= job_start[b]; --j) {
        const fva_paste_job& jb = jobs[j];
        int ry = y - jb.top, rx = x - jb.left;
        if (ry < 0 || ry >= jb.dst_h || rx < 0 || rx >= jb.dst_w) continue;
        if (jb.flip_v) ry = jb.dst_h - 1 - ry;
        if (jb.flip_h) rx = jb.dst_w - 1 - rx;
        const uint8_t* S = src + jb.src_offset;
        const int64_t pitch = jb.src_pitch ? (int64_t)jb.src_pitch : (int64_t)jb.src_w * 3;
        if (jb.src_w == 2 * jb.dst_w && jb.src_h == 2 * jb.dst_h) {
            // OpenCV reroutes an exact 2x decimation to INTER_AREA
            const uint8_t* p0 = S + (int64_t)(2 * ry) * pitch + (2 * rx) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (p0[c] + p0[3 + c] + p0[pitch + c] + p0[pitch + 3 + c] + 2) >> 2;
        } else {
            const Tap tx = make_tap(rx, jb.scale_x, jb.src_w), ty = make_tap(ry, jb.scale_y, jb.src_h);
            const uint8_t* r0 = S + (int64_t)ty.s0 * pitch;
            const uint8_t* r1 = S + (int64_t)ty.s1 * pitch;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int h0 = r0[tx.s0 * 3 + c] * tx.w0 + r0[tx.s1 * 3 + c] * tx.w1;   // horizontal pass, <= 255 * 2048
                const int h1 = r1[tx.s0 * 3 + c] * tx.w0 + r1[tx.s1 * 3 + c] * tx.w1;
                int r = (((ty.w0 * (h0 >> 4)) >> 16) + ((ty.w1 * (h1 >> 4)) >> 16) + 2) >> 2;
                v[c] = r < 0 ? 0 : (r > 255 ? 255 : r);
            }
        }
        break;
    }
    const int64_t plane = (int64_t)H * W;
    if constexpr (U8) {
        uint8_t* o = (uint8_t*)out_ + ((int64_t)b * plane + e) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (uint8_t)v[c];
    } else {
        float* o = (float*)out_ + (int64_t)b * 3 * plane + e;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c * plane] = slut[c * 256 + v[c]];
    }
}

}  // namespace

static int check_paste(const void* src, const void* jobs, const void* job_start, const void* out, int B, int H, int W, int fill,
                       const char* who) {
    if (!src || !jobs || !job_start || !out) return fva_fail(FVA_ERR_ARG, "%s: null pointer", who);
    if (B < 1 || B > 65535 || H < 1 || W < 1 || (int64_t)H * W >= (1ll << 31)) return fva_fail(FVA_ERR_ARG, "%s: bad shape B=%d H=%d W=%d", who, B, H, W);
    if (fill < 0 || fill > 255) return fva_fail(FVA_ERR_ARG, "%s: fill %d not a byte", who, fill);
    return FVA_OK;
}

extern "C" int fva_paste_resize_normalize(const uint8_t* src, const fva_paste_job* jobs, const int32_t* job_start, int32_t B,
                                          int32_t H, int32_t W, int32_t fill, const float* lut, float* out, void* stream) {
    int rc = check_paste(src, jobs, job_start, out, B, H, W, fill, "fva_paste_resize_normalize");
    if (rc) return rc;
    if (!lut) return fva_fail(FVA_ERR_ARG, "fva_paste_resize_normalize: null value table");
    hipLaunchKernelGGL(paste_kernel<false>, dim3(cdiv((int64_t)H * W, 256), B), dim3(256), 0, (hipStream_t)stream, src, jobs, job_start,
                       H, W, fill, lut, (void*)out);
    FVA_LAUNCH_CHECK("paste_kernel");
    return FVA_OK;
}

extern "C" int fva_paste_resize_u8(const uint8_t* src, const fva_paste_job* jobs, const int32_t* job_start, int32_t B, int32_t H,
                                   int32_t W, int32_t fill, uint8_t* out, void* stream) {
    int rc = check_paste(src, jobs, job_start, out, B, H, W, fill, "fva_paste_resize_u8");
    if (rc) return rc;
    hipLaunchKernelGGL(paste_kernel<true>, dim3(cdiv((int64_t)H * W, 256), B), dim3(256), 0, (hipStream_t)stream, src, jobs, job_start,
                       H, W, fill, (const float*)nullptr, (void*)out);
    FVA_LAUNCH_CHECK("paste_kernel");
    return FVA_OK;
}
