// Target assignment and the two YOLOv3 training losses with analytic backward, for gfx950.
// Latency-bound (T ~ 1e2..1e3 targets, B*3*g^2 ~ 1e6 cells): no host syncs, deterministic loss reductions,
// wavefront shuffles for the per-row sums.  Compiled with -ffp-contract=off: the matcher reproduces the
// reference's fp32 operation order bit for bit (integer outputs must be exact).
//
// Library surface: Yolov3Loss.build_target / forward (reference loss/yolov3_loss.py:29-124) with
// BiCrossEntropyLoss (loss/classification_loss.py:42-65), CIOULoss (loss/iou_loss.py:88-107) and the IoU family of
// detection/tools/IOU.py (quirks kept: eps inside the height factor :74-75, DIoU "+" sign :341, alpha no-grad :436).
// Demo surface: ComputeLoss.forward (demos/yolov3_u/utils/lossv3.py:18-119).
#include <math.h>

#include "common.h"

namespace {

constexpr float BCE_EPS = 1e-8f;
constexpr float IOU_EPS = 1e-7f;

// ---------------------------------------------------------------------------------------------------------
// forward-mode dual number over the 4 box parameters (x, y, w, h): exact autograd semantics incl. ties
struct D4 {
    float v, d[4];
};
__device__ __forceinline__ D4 dconst(float v) { return D4{v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 dvar(float v, int i) {
    D4 r = dconst(v);
    r.d[i] = 1.f;
    return r;
}
__device__ __forceinline__ D4 operator+(const D4& a, const D4& b) {
    D4 r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
__device__ __forceinline__ D4 operator-(const D4& a, const D4& b) {
    D4 r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
__device__ __forceinline__ D4 operator*(const D4& a, const D4& b) {
    D4 r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
__device__ __forceinline__ D4 operator/(const D4& a, const D4& b) {
    D4 r; r.v = a.v / b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
    return r;
}
__device__ __forceinline__ D4 operator+(const D4& a, float c) { D4 r = a; r.v += c; return r; }
__device__ __forceinline__ D4 operator*(const D4& a, float c) {
    D4 r; r.v = a.v * c;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * c;
    return r;
}
// torch.maximum / torch.minimum backward: ties split the gradient evenly
__device__ __forceinline__ D4 dmax(const D4& a, const D4& b) {
    if (a.v > b.v) return a;
    if (a.v < b.v) return b;
    D4 r; r.v = a.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = 0.5f * (a.d[i] + b.d[i]);
    return r;
}
__device__ __forceinline__ D4 dmin(const D4& a, const D4& b) {
    if (a.v < b.v) return a;
    if (a.v > b.v) return b;
    D4 r; r.v = a.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = 0.5f * (a.d[i] + b.d[i]);
    return r;
}
__device__ __forceinline__ D4 dclamp0(const D4& a) {  // clamp(min=0): gradient passes where a >= 0
    if (a.v >= 0.f) return a;
    return dconst(0.f);
}
__device__ __forceinline__ D4 datan(const D4& a) {
    D4 r; r.v = atanf(a.v);
    const float g = 1.f / (1.f + a.v * a.v);
#pragma unroll
    for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * g;
    return r;
}

struct BoxD { D4 x1, y1, x2, y2; };
__device__ __forceinline__ BoxD xywh2xyxy_d(const D4& x, const D4& y, const D4& w, const D4& h) {
    const D4 hw = w * 0.5f, hh = h * 0.5f;  // BOX.py:6-9 divides by 2 (exact in fp32 either way)
    return BoxD{x - hw, y - hh, x + hw, y + hh};
}
// xyxy_iou, torch branch (IOU.py:73-87): eps inside the height factor
__device__ __forceinline__ D4 iou_quirk_d(const BoxD& a, const BoxD& b, float eps) {
    const D4 area_a = (a.x2 - a.x1) * ((a.y2 - a.y1) + eps);
    const D4 area_b = (b.x2 - b.x1) * ((b.y2 - b.y1) + eps);
    const D4 iw = dclamp0(dmin(a.x2, b.x2) - dmax(a.x1, b.x1));
    const D4 ih = dclamp0(dmin(a.y2, b.y2) - dmax(a.y1, b.y1));
    const D4 inter = iw * ih;
    const D4 uni = ((area_a + area_b) - inter) + eps;
    return inter / uni;
}
// plain-area IoU used by the *_batch variants and GIoU (IOU.py:143-151, 219-226)
__device__ __forceinline__ D4 iou_plain_d(const BoxD& a, const BoxD& b, float eps, D4* uni_out = nullptr) {
    const D4 area_a = (a.x2 - a.x1) * (a.y2 - a.y1);
    const D4 area_b = (b.x2 - b.x1) * (b.y2 - b.y1);
    const D4 iw = dclamp0(dmin(a.x2, b.x2) - dmax(a.x1, b.x1));
    const D4 ih = dclamp0(dmin(a.y2, b.y2) - dmax(a.y1, b.y1));
    const D4 inter = iw * ih;
    const D4 uni = ((area_a + area_b) - inter) + eps;
    if (uni_out) *uni_out = uni;
    return inter / uni;
}
// DIoU (IOU.py:294-343): iou + rho^2/c^2 (library sign); demo variant (demos/yolov3_u/utils/iou.py:334-341):
// centre sums not halved, minus sign
__device__ __forceinline__ D4 diou_d(const BoxD& a, const BoxD& b, const D4& iou, float eps, int demo) {
    const D4 cw = dmax(a.x2, b.x2) - dmin(a.x1, b.x1);
    const D4 ch = dmax(a.y2, b.y2) - dmin(a.y1, b.y1);
    const D4 c2 = (cw * cw + ch * ch) + eps;
    D4 cxa = a.x1 + a.x2, cya = a.y1 + a.y2, cxb = b.x1 + b.x2, cyb = b.y1 + b.y2;
    if (!demo) { cxa = cxa * 0.5f; cya = cya * 0.5f; cxb = cxb * 0.5f; cyb = cyb * 0.5f; }
    const D4 dx = cxa - cxb, dy = cya - cyb;
    const D4 rho2 = dx * dx + dy * dy;
    const D4 term = rho2 / c2;
    return demo ? iou - term : iou + term;
}
// CIoU (IOU.py:397-440): diou - alpha*v with alpha a constant (no_grad)
__device__ __forceinline__ D4 ciou_d(const BoxD& a, const BoxD& b, float eps, int demo, D4* iou_out) {
    const D4 iou = iou_quirk_d(a, b, eps);
    const D4 diou = diou_d(a, b, iou, eps, demo);
    const D4 wa = a.x2 - a.x1, ha = a.y2 - a.y1, wb = b.x2 - b.x1, hb = b.y2 - b.y1;
    const D4 dt = datan(wb / (hb + eps)) - datan(wa / (ha + eps));
    const D4 v = (dt * dt) * (float)(4.0 / (M_PI * M_PI));
    const float alpha = v.v / ((v.v - iou.v) + (1.f + eps));
    if (iou_out) *iou_out = iou;
    return diou - v * alpha;
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------------------------------------------------
struct Level {  // device copy of fva_head_level
    float* data;
    float* grad;
    int64_t sb, sa, sy, sx, sk;
    int B, A, H, W, K;
    float aw[8], ah[8];
    float stride;
};
Level to_level(const fva_head_level& l) {
    Level r;
    r.data = l.data; r.grad = l.grad;
    r.sb = l.sb; r.sa = l.sa; r.sy = l.sy; r.sx = l.sx; r.sk = l.sk;
    r.B = l.B; r.A = l.A; r.H = l.H; r.W = l.W; r.K = l.K;
    for (int i = 0; i < 8; ++i) { r.aw[i] = l.anchor_w[i]; r.ah[i] = l.anchor_h[i]; }
    r.stride = l.stride;
    return r;
}
struct MatchBuf {
    int32_t* count;
    int64_t *b, *gx, *gy, *a, *cls;
    float *xywh, *anc;
};

// build_target (yolov3_loss.py:87-122) for one level.  One block; ordered compaction keeps the reference's
// row order (target-major, anchor-minor).
__global__ __launch_bounds__(1024) void match_kernel(const float* __restrict__ tg, int T, const Level lv, const MatchBuf o) {
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    const float fw = (float)lv.W, fh = (float)lv.H;
    const int total = T * lv.A;
    for (int start = 0; start < total; start += 1024) {
        const int idx = start + tid;
        bool keep = false;
        int t = 0, a = 0;
        float tx = 0, ty = 0, tw = 0, th = 0, aw = 1, ah = 1;
        if (idx < total) {
            t = idx / lv.A;
            a = idx - t * lv.A;
            aw = lv.aw[a] / lv.stride;  // :88-89
            ah = lv.ah[a] / lv.stride;
            tx = tg[t * 6 + 2] * fw;    // :94-95  [W,H,W,H]
            ty = tg[t * 6 + 3] * fh;
            tw = tg[t * 6 + 4] * fw;
            th = tg[t * 6 + 5] * fh;
            const float rw = tw / aw, rh = th / ah;  // :98
            const float mw = fmaxf(rw, 1.f / rw), mh = fmaxf(rh, 1.f / rh);
            keep = fmaxf(mw, mh) < 4.f;  // :99
        }
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i < wv) woff += wave_cnt[i];
            tot += wave_cnt[i];
        }
        const int base = base_s;
        if (keep) {
            const int r = base + woff + before;
            const float flx = floorf(tx), fly = floorf(ty);  // :113
            long long gx = (long long)flx, gy = (long long)fly;
            o.xywh[r * 4 + 0] = tx - (float)gx;  // :114 offsets are taken BEFORE the clamp
            o.xywh[r * 4 + 1] = ty - (float)gy;
            o.xywh[r * 4 + 2] = tw;
            o.xywh[r * 4 + 3] = th;
            gx = gx < 0 ? 0 : (gx > lv.W - 1 ? lv.W - 1 : gx);  // :116-117
            gy = gy < 0 ? 0 : (gy > lv.H - 1 ? lv.H - 1 : gy);
            o.b[r] = (long long)tg[t * 6 + 0];
            o.cls[r] = (long long)tg[t * 6 + 1];
            o.gx[r] = gx;
            o.gy[r] = gy;
            o.a[r] = a;
            o.anc[r * 2 + 0] = aw;
            o.anc[r * 2 + 1] = ah;
        }
        __syncthreads();
        if (tid == 0) base_s = base + tot;
        __syncthreads();
    }
    if (tid == 0) *o.count = base_s;
}

// cell -> index of the LAST match that lands in it (index_put: last write wins)
__global__ void scatter_kernel(const MatchBuf mb, const Level lv, int32_t* cellmatch) {
    const int n = *mb.count;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < n; m += gridDim.x * blockDim.x) {
        const int64_t b = mb.b[m];
        if (b < 0 || b >= lv.B) continue;  // the reference would raise IndexError
        const int64_t cell = ((b * lv.A + mb.a[m]) * lv.H + mb.gy[m]) * lv.W + mb.gx[m];
        atomicMax(&cellmatch[cell], m);
    }
}

// one wavefront per match: class BCE, CIoU box loss, IoU for the objectness target, and all their gradients
// norm_count / norm_batch (data parallel, see fva_yolov3_loss_dp): the match count of this level and the batch size of the WHOLE job,
// which the per-match means and the "* bs" of yolov3_loss.py:69-71 then refer to; the objectness term keeps the local batch
// (its mean runs over cells, whose number per image is the same on every rank).
__global__ __launch_bounds__(256) void match_loss_kernel(const MatchBuf mb, const Level lv, float* iou_out, float* box_l,
                                                         float* cls_l, float ratio_box, float ratio_conf, float ratio_cls,
                                                         const int32_t* norm_count, int norm_batch) {
    const int n = *mb.count;
    const int nn = norm_count ? *norm_count : n;     // denominator of the per-match means
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int C = lv.K - 5;
    const float fB = (float)lv.B;
    const float fBn = norm_count ? (float)norm_batch : fB;
    const float ncell = (float)lv.B * lv.A * lv.H * lv.W;
    for (int m = blockIdx.x * wpb + (threadIdx.x >> 6); m < n; m += gridDim.x * wpb) {
        const int64_t b = mb.b[m];
        if (b < 0 || b >= lv.B) {
            if (lane == 0) { iou_out[m] = 0.f; box_l[m] = 0.f; cls_l[m] = 0.f; }
            continue;
        }
        const int64_t base = b * lv.sb + mb.a[m] * lv.sa + mb.gy[m] * lv.sy + mb.gx[m] * lv.sx;
        const int cls = (int)mb.cls[m];
        // ---- class term: BCE on probabilities with 1e-8 inside the logs (classification_loss.py:54), mean over M*C
        const float gc = ratio_cls * fBn / ((float)nn * (float)C);
        float lsum = 0.f;
        for (int k = lane; k < C; k += 64) {
            const float z = lv.data[base + (5 + k) * lv.sk];
            const float p = sigm(z);
            const float t = (k == cls) ? 1.f : 0.f;
            lsum += -t * logf(p + BCE_EPS) - (1.f - t) * logf(1.f - p + BCE_EPS);
            if (lv.grad) {
                const float dldp = -t / (p + BCE_EPS) + (1.f - t) / (1.f - p + BCE_EPS);
                atomicAdd(&lv.grad[base + (5 + k) * lv.sk], gc * dldp * p * (1.f - p));
            }
        }
        lsum = wave_sum(lsum);
        // ---- box term (lane 0): pred = (sigmoid xy, exp(wh) * anchor) vs target, CIoU + plain IoU
        if (lane == 0) {
            const float z0 = lv.data[base], z1 = lv.data[base + lv.sk], z2 = lv.data[base + 2 * lv.sk], z3 = lv.data[base + 3 * lv.sk];
            const float zc = lv.data[base + 4 * lv.sk];
            const float px = sigm(z0), py = sigm(z1);
            const float pw = expf(z2) * mb.anc[m * 2], ph = expf(z3) * mb.anc[m * 2 + 1];
            const BoxD pb = xywh2xyxy_d(dvar(px, 0), dvar(py, 1), dvar(pw, 2), dvar(ph, 3));
            const BoxD tb = xywh2xyxy_d(dconst(mb.xywh[m * 4]), dconst(mb.xywh[m * 4 + 1]), dconst(mb.xywh[m * 4 + 2]),
                                        dconst(mb.xywh[m * 4 + 3]));
            D4 iou;
            const D4 ciou = ciou_d(pb, tb, IOU_EPS, 0, &iou);
            iou_out[m] = iou.v;
            box_l[m] = 1.f - ciou.v;
            cls_l[m] = lsum;
            if (lv.grad) {
                // d total / d tau for this cell (yolov3_loss.py:60-64): the IoU target is not detached
                const float pc = sigm(zc);
                const float gtau = (-logf(pc + BCE_EPS) + logf(1.f - pc + BCE_EPS)) * (ratio_conf * fB / ncell);
                const float gb = ratio_box * fBn / (float)nn;
                const float dpdz[4] = {px * (1.f - px), py * (1.f - py), pw, ph};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    atomicAdd(&lv.grad[base + i * lv.sk], (gtau * iou.d[i] - gb * ciou.d[i]) * dpdz[i]);
            }
        }
    }
}

// objectness BCE over every cell (yolov3_loss.py:63-64); one thread per cell, wave + block reduction, fixed order
__global__ __launch_bounds__(256) void conf_kernel(const Level lv, const int32_t* __restrict__ cellmatch,
                                                   const float* __restrict__ iou, float* partial, float ratio_conf) {
    __shared__ float wsum[4];
    const int64_t ncell = (int64_t)lv.B * lv.A * lv.H * lv.W;
    const float g = ratio_conf * (float)lv.B / (float)ncell;
    float acc = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ncell; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % lv.W), y = (int)((i / lv.W) % lv.H), a = (int)((i / ((int64_t)lv.W * lv.H)) % lv.A);
        const int b = (int)(i / ((int64_t)lv.W * lv.H * lv.A));
        const int64_t off = b * lv.sb + a * lv.sa + y * lv.sy + x * lv.sx + 4 * lv.sk;
        const float p = sigm(lv.data[off]);
        const int mi = cellmatch[i];
        const float t = mi >= 0 ? iou[mi] : 0.f;
        acc += -t * logf(p + BCE_EPS) - (1.f - t) * logf(1.f - p + BCE_EPS);
        if (lv.grad) lv.grad[off] = g * (-t / (p + BCE_EPS) + (1.f - t) / (1.f - p + BCE_EPS)) * p * (1.f - p);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

struct FinalArgs {
    const int32_t* count[4];
    const float* box_l[4];
    const float* cls_l[4];
    const float* conf_part[4];
    int conf_blocks[4];
    double ncell[4];
    int C[4];
    int nlevels;
    int B;
    const int32_t* norm_counts;   // data parallel: [nlevels] job-wide match counts, or null
    int norm_batch;
};
__global__ __launch_bounds__(256) void yolo_final_kernel(const FinalArgs fa, float rb, float rc, float rcls, float* out) {
    __shared__ double red[3][256];
    double lbox = 0.0, lconf = 0.0, lcls = 0.0;
    for (int l = 0; l < fa.nlevels; ++l) {
        const int n = *fa.count[l];
        double sb = 0.0, sc = 0.0, sf = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) { sb += fa.box_l[l][i]; sc += fa.cls_l[l][i]; }
        for (int i = threadIdx.x; i < fa.conf_blocks[l]; i += 256) sf += fa.conf_part[l][i];
        red[0][threadIdx.x] = sb; red[1][threadIdx.x] = sc; red[2][threadIdx.x] = sf;
        __syncthreads();
        if (threadIdx.x == 0) {
            sb = sc = sf = 0.0;
            for (int i = 0; i < 256; ++i) { sb += red[0][i]; sc += red[1][i]; sf += red[2][i]; }
            const int nn = fa.norm_counts ? fa.norm_counts[l] : n;
            if (n > 0 && nn > 0) {
                lbox += sb / nn;
                lcls += sc / ((double)nn * fa.C[l]);
            }
            lconf += sf / fa.ncell[l];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[1] = (float)lbox;
        out[2] = (float)lconf;
        out[3] = (float)lcls;
        // data parallel: this rank's SHARE of the job-wide loss (the shares add up to the loss nn.DataParallel computes on the gathered batch)
        out[0] = fa.norm_counts ? (float)((lbox * rb + lcls * rcls) * fa.norm_batch + lconf * rc * fa.B)
                                : (float)((lbox * rb + lconf * rc + lcls * rcls) * fa.B);
    }
}

// ---------------------------------------------------------------------------------------------------------
// demo loss
struct DemoBuf {
    int32_t *img_start, *img_count, *perm;  // [B], [B], [T]
    int8_t* mask;                           // [B*H*W*A] : -1 ignore / 0 / 1 positive
    float *l_xy, *l_wh, *l_cls;             // [T]
    float* conf_part;                       // [blocks][2] = {loss sum, valid count}
    float* nvalid;                          // [1]
    int32_t* cell;                          // [T] flattened (b, gy, gx, a) cell index or -1
};

__global__ void demo_group_kernel(const float* __restrict__ tg, int T, int B, int32_t* start, int32_t* count, int32_t* perm) {
    // single block: per-image target lists (order inside an image does not matter: only max / any are taken)
    extern __shared__ int32_t sh[];  // [B] counts, [B] cursors
    for (int i = threadIdx.x; i < 2 * B; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const int b = (int)tg[t * 6];
        if (b >= 0 && b < B) atomicAdd(&sh[b], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int b = 0; b < B; ++b) { start[b] = run; count[b] = sh[b]; sh[B + b] = run; run += sh[b]; }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const int b = (int)tg[t * 6];
        if (b >= 0 && b < B) perm[atomicAdd(&sh[B + b], 1)] = t;
    }
}

__device__ __forceinline__ float bce_logits(float z, float t) { return fmaxf(z, 0.f) - z * t + log1pf(expf(-fabsf(z))); }

// one wavefront per target: best anchor, cell, xy / wh / class terms and their gradients (lossv3.py:44-84)
__global__ __launch_bounds__(256) void demo_target_kernel(const float* __restrict__ tg, int T, const Level lv, const DemoBuf db) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int C = lv.K - 5;
    for (int t = blockIdx.x * wpb + (threadIdx.x >> 6); t < T; t += gridDim.x * wpb) {
        const float fw = (float)lv.W, fh = (float)lv.H;
        const float tx = tg[t * 6 + 2] * fw, ty = tg[t * 6 + 3] * fh, tw = tg[t * 6 + 4] * fw, th = tg[t * 6 + 5] * fh;
        int best = 0;
        float best_iou = -INFINITY;
        for (int a = 0; a < lv.A; ++a) {  // wh_iou_batch (iou.py:177-189) + torch.max first index
            const float inter = fminf(tw, lv.aw[a]) * fminf(th, lv.ah[a]);
            const float iou = inter / (((tw * th) + (lv.aw[a] * lv.ah[a])) - inter + IOU_EPS);
            if (iou > best_iou) { best_iou = iou; best = a; }
        }
        const float gxf = floorf(tx), gyf = floorf(ty);
        const int b = (int)tg[t * 6], gx = (int)gxf, gy = (int)gyf, cls = (int)tg[t * 6 + 1];
        const bool ok = b >= 0 && b < lv.B && gx >= 0 && gx < lv.W && gy >= 0 && gy < lv.H;  // reference: IndexError otherwise
        if (!ok) {
            if (lane == 0) { db.l_xy[t] = 0.f; db.l_wh[t] = 0.f; db.l_cls[t] = 0.f; db.cell[t] = -1; }
            continue;
        }
        const int64_t base = b * lv.sb + best * lv.sa + gy * lv.sy + gx * lv.sx;
        float lsum = 0.f;
        const float gcls = 1.f / ((float)T * (float)C);
        for (int k = lane; k < C; k += 64) {
            const float z = lv.data[base + (5 + k) * lv.sk];
            const float tt = (k == cls) ? 1.f : 0.f;
            lsum += bce_logits(z, tt);
            if (lv.grad) atomicAdd(&lv.grad[base + (5 + k) * lv.sk], (sigm(z) - tt) * gcls);
        }
        lsum = wave_sum(lsum);
        if (lane == 0) {
            const float z0 = lv.data[base], z1 = lv.data[base + lv.sk], z2 = lv.data[base + 2 * lv.sk], z3 = lv.data[base + 3 * lv.sk];
            const float ox = tx - gxf, oy = ty - gyf;
            const float twh0 = logf(tw / lv.aw[best] + 1e-14f), twh1 = logf(th / lv.ah[best] + 1e-14f);
            db.l_xy[t] = bce_logits(z0, ox) + bce_logits(z1, oy);
            db.l_wh[t] = (z2 - twh0) * (z2 - twh0) + (z3 - twh1) * (z3 - twh1);
            db.l_cls[t] = lsum;
            db.cell[t] = ((b * lv.H + gy) * lv.W + gx) * lv.A + best;
            if (lv.grad) {
                const float g2 = 1.f / (2.f * (float)T);
                atomicAdd(&lv.grad[base], 2.f * (sigm(z0) - ox) * g2);  // loss_xy carries weight 2.0 (:111)
                atomicAdd(&lv.grad[base + lv.sk], 2.f * (sigm(z1) - oy) * g2);
                atomicAdd(&lv.grad[base + 2 * lv.sk], 2.f * (z2 - twh0) * g2);
                atomicAdd(&lv.grad[base + 3 * lv.sk], 2.f * (z3 - twh1) * g2);
            }
        }
    }
}

// ignore mask (lossv3.py:86-100): per predicted box, max IoU against the targets of its image > 0.5 -> -1
__global__ __launch_bounds__(256) void demo_mask_kernel(const float* __restrict__ tg, const Level lv, const DemoBuf db) {
    const int64_t n = (int64_t)lv.B * lv.H * lv.W * lv.A;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int a = (int)(i % lv.A), x = (int)((i / lv.A) % lv.W), y = (int)((i / ((int64_t)lv.A * lv.W)) % lv.H);
        const int b = (int)(i / ((int64_t)lv.A * lv.W * lv.H));
        const int64_t base = b * lv.sb + a * lv.sa + y * lv.sy + x * lv.sx;
        const float px = sigm(lv.data[base]) + (float)x, py = sigm(lv.data[base + lv.sk]) + (float)y;
        const float pw = expf(lv.data[base + 2 * lv.sk]) * lv.aw[a], ph = expf(lv.data[base + 3 * lv.sk]) * lv.ah[a];
        const float ax1 = px - pw / 2, ay1 = py - ph / 2, ax2 = px + pw / 2, ay2 = py + ph / 2;
        const float area_a = (ax2 - ax1) * (ay2 - ay1);
        const float fw = (float)lv.W, fh = (float)lv.H;
        bool ignore = false;
        const int s = db.img_start[b], c = db.img_count[b];
        for (int j = 0; j < c; ++j) {
            const int t = db.perm[s + j];
            const float tx = tg[t * 6 + 2] * fw, ty = tg[t * 6 + 3] * fh, tw = tg[t * 6 + 4] * fw, th = tg[t * 6 + 5] * fh;
            const float bx1 = tx - tw / 2, by1 = ty - th / 2, bx2 = tx + tw / 2, by2 = ty + th / 2;
            const float area_b = (bx2 - bx1) * (by2 - by1);
            const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
            const float inter = iw * ih;
            const float iou = inter / ((area_a + area_b) - inter + IOU_EPS);
            ignore = ignore || (iou > 0.5f);
        }
        db.mask[i] = ignore ? -1 : 0;
    }
}
__global__ void demo_positive_kernel(int T, const DemoBuf db) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x)
        if (db.cell[t] >= 0) db.mask[db.cell[t]] = 1;  // :101 positives override the ignore flag
}
// masked objectness BCE-with-logits: pass 0 sums loss and valid count per block, pass 1 writes gradients
__global__ __launch_bounds__(256) void demo_conf_kernel(const Level lv, const DemoBuf db, int pass) {
    __shared__ float ws[2][4];
    const int64_t n = (int64_t)lv.B * lv.H * lv.W * lv.A;
    float acc = 0.f, cnt = 0.f;
    const float inv = pass ? 1.f / *db.nvalid : 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int a = (int)(i % lv.A), x = (int)((i / lv.A) % lv.W), y = (int)((i / ((int64_t)lv.A * lv.W)) % lv.H);
        const int b = (int)(i / ((int64_t)lv.A * lv.W * lv.H));
        const int64_t off = b * lv.sb + a * lv.sa + y * lv.sy + x * lv.sx + 4 * lv.sk;
        const int mk = db.mask[i];
        if (pass == 0) {
            if (mk != -1) { acc += bce_logits(lv.data[off], (float)mk); cnt += 1.f; }
        } else if (lv.grad) {
            lv.grad[off] = mk != -1 ? (sigm(lv.data[off]) - (float)mk) * inv : 0.f;
        }
    }
    if (pass == 0) {
        acc = wave_sum(acc);
        cnt = wave_sum(cnt);
        if ((threadIdx.x & 63) == 0) { ws[0][threadIdx.x >> 6] = acc; ws[1][threadIdx.x >> 6] = cnt; }
        __syncthreads();
        if (threadIdx.x == 0) {
            db.conf_part[blockIdx.x * 2] = (ws[0][0] + ws[0][1]) + (ws[0][2] + ws[0][3]);
            db.conf_part[blockIdx.x * 2 + 1] = (ws[1][0] + ws[1][1]) + (ws[1][2] + ws[1][3]);
        }
    }
}
// per level: finish N_valid and the four partial losses; acc[5] accumulates across levels
__global__ __launch_bounds__(256) void demo_level_final_kernel(const DemoBuf db, int T, int C, int blocks, double* acc, float* out,
                                                               int last) {
    __shared__ double red[5][256];
    double s[5] = {0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < T; i += 256) { s[0] += db.l_xy[i]; s[1] += db.l_wh[i]; s[2] += db.l_cls[i]; }
    for (int i = threadIdx.x; i < blocks; i += 256) { s[3] += db.conf_part[2 * i]; s[4] += db.conf_part[2 * i + 1]; }
    for (int k = 0; k < 5; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < 5; ++k) { s[k] = 0; for (int i = 0; i < 256; ++i) s[k] += red[k][i]; }
        *db.nvalid = (float)s[4];
        acc[0] += s[0] / (2.0 * T);
        acc[1] += s[1] / (2.0 * T);
        acc[2] += s[2] / ((double)T * C);
        acc[3] += s[3] / s[4];
        if (last) {
            out[1] = (float)acc[0]; out[2] = (float)acc[1]; out[3] = (float)acc[2]; out[4] = (float)acc[3];
            out[0] = (float)(acc[0] * 2.0 + acc[1] + acc[2] + acc[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// IoU family for the tools API (value + gradient w.r.t. the FIRST box in the caller's parametrisation)
__device__ __forceinline__ BoxD load_box(const float* p, int mode, bool var) {
    D4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = var ? dvar(p[i], i) : dconst(p[i]);
    if (mode == 1) return xywh2xyxy_d(v[0], v[1], v[2], v[3]);
    return BoxD{v[0], v[1], v[2], v[3]};
}
__device__ D4 iou_any(int kind, int mode, int variant, const float* pa, const float* pb, float eps, bool batch) {
    if (mode == 2) {  // wh_iou / wh_iou_batch (IOU.py:108-120, 177-189)
        const D4 aw = dvar(pa[0], 0), ah = dvar(pa[1], 1), bw = dconst(pb[0]), bh = dconst(pb[1]);
        const D4 inter = dmin(aw, bw) * dmin(ah, bh);
        return inter / ((((aw * ah) + (bw * bh)) - inter) + eps);
    }
    const BoxD a = load_box(pa, mode, true), b = load_box(pb, mode, false);
    if (kind == 0) return batch ? iou_plain_d(a, b, eps) : iou_quirk_d(a, b, eps);
    if (kind == 1) {  // GIoU (IOU.py:204-239; the batch variant carries a "+" sign, :290)
        D4 uni;
        const D4 iou = iou_plain_d(a, b, eps, &uni);
        const D4 cw = dmax(a.x2, b.x2) - dmin(a.x1, b.x1), ch = dmax(a.y2, b.y2) - dmin(a.y1, b.y1);
        const D4 convex = (cw * ch) + eps;
        const D4 term = (convex - uni) / convex;
        return batch ? iou + term : iou - term;
    }
    const D4 iou = batch ? iou_plain_d(a, b, eps) : iou_quirk_d(a, b, eps);
    const D4 diou = diou_d(a, b, iou, eps, variant);
    if (kind == 2) return diou;
    // CIoU (IOU.py:397-440 / 442-482): the squared atan difference; alpha is a constant
    const D4 wa = a.x2 - a.x1, ha = a.y2 - a.y1, wb = b.x2 - b.x1, hb = b.y2 - b.y1;
    const D4 dt = datan(wb / (hb + eps)) - datan(wa / (ha + eps));
    const D4 v = (dt * dt) * (float)(4.0 / (M_PI * M_PI));
    const float alpha = v.v / ((v.v - iou.v) + (1.f + eps));
    return diou - v * alpha;
}
__global__ void iou_pair_kernel(int kind, int mode, int variant, const float* a, const float* b, float* out, float* grad_a,
                                int64_t N, float eps) {
    const int w = mode == 2 ? 2 : 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const D4 r = iou_any(kind, mode, variant, a + i * w, b + i * w, eps, false);
        out[i] = r.v;
        if (grad_a)
            for (int k = 0; k < w; ++k) grad_a[i * w + k] = r.d[k];
    }
}
__global__ void iou_batch_kernel(int kind, int mode, int variant, const float* a, const float* b, float* out, int64_t N, int64_t M,
                                 float eps) {
    const int w = mode == 2 ? 2 : 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N * M; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = iou_any(kind, mode, variant, a + (i / M) * w, b + (i % M) * w, eps, true).v;
}

// ---------------------------------------------------------------------------------------------------------
// Stand-alone BiCrossEntropyLoss (loss/classification_loss.py:36-65): per element
//   l = -t log(p + 1e-8) - (1 - t) log(1 - p + 1e-8),  p = y or sigmoid(y),  t = one-hot(label) or the dense target,
// times an optional per-element weight; block partial sums in fixed order, and dl/dy for the backward.
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ y, const int64_t* __restrict__ label,
                                                  const float* __restrict__ dense_t, const float* __restrict__ w, int64_t w_numel,
                                                  int64_t numel, int C, int already_sigmoid, float* __restrict__ partial,
                                                  float* __restrict__ grad) {
    __shared__ float wsum[4];
    float acc = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / C;
        const int k = (int)(i - row * C);
        const float t = label ? ((int64_t)k == label[row] ? 1.f : 0.f) : dense_t[i];
        const float v = y[i];
        const float p = already_sigmoid ? v : sigm(v);
        const float wi = w ? (w_numel == 1 ? w[0] : w[i]) : 1.f;
        acc += wi * (-t * logf(p + BCE_EPS) - (1.f - t) * logf(1.f - p + BCE_EPS));
        if (grad) {
            const float dldp = -t / (p + BCE_EPS) + (1.f - t) / (1.f - p + BCE_EPS);
            grad[i] = wi * dldp * (already_sigmoid ? 1.f : p * (1.f - p));
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}
__global__ __launch_bounds__(256) void bce_final_kernel(const float* __restrict__ partial, int nblocks, double denom, float* out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = 0.0;
        for (int i = 0; i < 256; ++i) s += red[i];
        out[0] = (float)(s / denom);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Losses of the two-stage head (demos/faster_rcnn/models/rpn.py:8-64,303-312, fast.py:173-201): over rows of logits with integer
// labels -- mode 0: F.cross_entropy(reduction='mean'); mode 1: the RPN's FocalLoss, -(1 - p_t)^gamma log p_t, mean -- and
// F.smooth_l1_loss(reduction='mean') (beta = 1).  Value and gradient in one launch, a one-block fixed-order finish.
__global__ __launch_bounds__(256) void row_loss_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int R, int C,
                                                       int mode, float gamma, float* __restrict__ row_loss, float* __restrict__ grad) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float* z = logits + (int64_t)r * C;
    const int y = (int)labels[r];
    float m = z[0];
    for (int k = 1; k < C; ++k) m = fmaxf(m, z[k]);
    float den = 0.f;
    for (int k = 0; k < C; ++k) den += expf(z[k] - m);
    const float logp = (z[y] - m) - logf(den), p = expf(logp);
    float loss, dldlogp;          // d loss / d log p_t: grad_k = dldlogp * (delta_ky - softmax_k)
    if (mode == 0) {
        loss = -logp;
        dldlogp = -1.f;
    } else {
        const float q = 1.f - p;
        const float qg = powf(q, gamma);
        loss = -qg * logp;
        // d/dp [-(1-p)^g log p] * p, with d p / d z_k = p (delta - s_k)
        dldlogp = (gamma * powf(q, gamma - 1.f) * logp - qg / p) * p;
    }
    row_loss[r] = loss;
    if (grad) {
        const float inv = 1.f / (float)R;
        for (int k = 0; k < C; ++k) {
            const float sk = expf(z[k] - m) / den;
            grad[(int64_t)r * C + k] = dldlogp * ((k == y ? 1.f : 0.f) - sk) * inv;
        }
    }
}
__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* __restrict__ partial,
                                                        float* __restrict__ grad) {
    __shared__ float wsum[4];
    float acc = 0.f;
    const float inv = 1.f / (float)n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i], ad = fabsf(d);
        acc += ad < 1.f ? 0.5f * d * d : ad - 0.5f;
        if (grad) grad[i] = (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f)) * inv;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

inline int64_t align256(int64_t x) { return (x + 255) & ~255ll; }
struct Carver {
    char* p;
    int64_t off = 0;
    template <typename T>
    T* take(int64_t n) {
        T* r = (T*)(p ? p + off : nullptr);
        off = align256(off + n * (int64_t)sizeof(T));
        return r;
    }
};
constexpr int CONF_BLOCKS = 1024;

MatchBuf carve_match(Carver& c, int cap) {
    MatchBuf m;
    m.count = c.take<int32_t>(1);
    m.b = c.take<int64_t>(cap); m.gx = c.take<int64_t>(cap); m.gy = c.take<int64_t>(cap);
    m.a = c.take<int64_t>(cap); m.cls = c.take<int64_t>(cap);
    m.xywh = c.take<float>((int64_t)cap * 4); m.anc = c.take<float>((int64_t)cap * 2);
    return m;
}
int check_level(const fva_head_level& l, const char* who) {
    if (!l.data) return fva_fail(FVA_ERR_ARG, "%s: null head tensor", who);
    if (l.A < 1 || l.A > 8 || l.K < 6 || l.B < 1 || l.H < 1 || l.W < 1) return fva_fail(FVA_ERR_ARG, "%s: bad level shape", who);
    if ((int64_t)l.B * l.A * l.H * l.W >= (1ll << 31)) return fva_fail(FVA_ERR_ARG, "%s: too many cells", who);
    return FVA_OK;
}

// x *= *scale, in place; every block leaves at once when the scalar is exactly 1 (the upstream gradient of loss.backward()): the loss
// gradients then cost an empty launch instead of a 274 MB read-modify-write at the start of the backward pass
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(float* __restrict__ x, int64_t n, const float* __restrict__ scale) {
    const float s = *scale;
    if (s == 1.f) return;
    const int64_t n4 = n / 4, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        f32x4 v = *(f32x4*)(x + 4 * i);
        v *= s;
        *(f32x4*)(x + 4 * i) = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) x[n4 * 4 + threadIdx.x] *= s;
}

}  // namespace

extern "C" {

int fva_yolov3_match(const float* targets, int32_t T, const fva_head_level* level, const fva_match_out* out, void* stream) {
    if (!level || !out || !out->count || (T > 0 && !targets)) return fva_fail(FVA_ERR_ARG, "fva_yolov3_match: null pointer");
    if (level->A < 1 || level->A > 8) return fva_fail(FVA_ERR_ARG, "fva_yolov3_match: bad anchor count");
    Level lv = to_level(*level);
    MatchBuf mb{out->count, out->b, out->gx, out->gy, out->a, out->cls, out->xywh, out->anc};
    hipLaunchKernelGGL(match_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, targets, T, lv, mb);
    FVA_LAUNCH_CHECK("match_kernel");
    return FVA_OK;
}

int64_t fva_yolov3_loss_workspace(int32_t T, const fva_head_level* levels, int32_t nlevels) {
    Carver c{nullptr};
    for (int l = 0; l < nlevels; ++l) {
        const int cap = T * levels[l].A > 0 ? T * levels[l].A : 1;
        carve_match(c, cap);
        c.take<float>(cap); c.take<float>(cap); c.take<float>(cap);
        c.take<int32_t>((int64_t)levels[l].B * levels[l].A * levels[l].H * levels[l].W);
        c.take<float>(CONF_BLOCKS);
    }
    return c.off;
}

int fva_yolov3_loss(const float* targets, int32_t T, const fva_head_level* levels, int32_t nlevels, float ratio_box,
                    float ratio_conf, float ratio_cls, float* loss_out, void* workspace, int64_t workspace_bytes, void* stream) {
    return fva_yolov3_loss_dp(targets, T, levels, nlevels, ratio_box, ratio_conf, ratio_cls, nullptr, 0, loss_out, workspace, workspace_bytes, stream);
}

int fva_yolov3_loss_dp(const float* targets, int32_t T, const fva_head_level* levels, int32_t nlevels, float ratio_box,
                       float ratio_conf, float ratio_cls, const int32_t* norm_counts, int32_t norm_batch, float* loss_out, void* workspace,
                       int64_t workspace_bytes, void* stream) {
    if (!levels || nlevels < 1 || nlevels > 4 || !loss_out || !workspace || (T > 0 && !targets))
        return fva_fail(FVA_ERR_ARG, "fva_yolov3_loss: bad argument");
    if (norm_counts && norm_batch < levels[0].B) return fva_fail(FVA_ERR_ARG, "fva_yolov3_loss_dp: job batch %d smaller than the local batch %d", norm_batch, levels[0].B);
    if (workspace_bytes < fva_yolov3_loss_workspace(T, levels, nlevels)) return fva_fail(FVA_ERR_WORKSPACE, "fva_yolov3_loss: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Carver c{(char*)workspace};
    FinalArgs fa = FinalArgs();
    fa.nlevels = nlevels;
    fa.B = levels[0].B;
    fa.norm_counts = norm_counts;
    fa.norm_batch = norm_batch;
    for (int l = 0; l < nlevels; ++l) {
        int rc = check_level(levels[l], "fva_yolov3_loss");
        if (rc) return rc;
        const Level lv = to_level(levels[l]);
        const int cap = T * lv.A > 0 ? T * lv.A : 1;
        const MatchBuf mb = carve_match(c, cap);
        float* iou = c.take<float>(cap);
        float* box_l = c.take<float>(cap);
        float* cls_l = c.take<float>(cap);
        const int64_t ncell = (int64_t)lv.B * lv.A * lv.H * lv.W;
        int32_t* cellmatch = c.take<int32_t>(ncell);
        float* conf_part = c.take<float>(CONF_BLOCKS);
        hipLaunchKernelGGL(match_kernel, dim3(1), dim3(1024), 0, s, targets, T, lv, mb);
        FVA_LAUNCH_CHECK("match_kernel");
        if (hipMemsetAsync(cellmatch, 0xFF, ncell * 4, s) != hipSuccess) return fva_fail(FVA_ERR_LAUNCH, "fva_yolov3_loss: memset failed");
        if (T > 0) {
            hipLaunchKernelGGL(scatter_kernel, dim3(cdiv(cap, 256)), dim3(256), 0, s, mb, lv, cellmatch);
            FVA_LAUNCH_CHECK("scatter_kernel");
            hipLaunchKernelGGL(match_loss_kernel, dim3(cdiv(cap, 4) < 2048 ? cdiv(cap, 4) : 2048), dim3(256), 0, s, mb, lv, iou, box_l,
                               cls_l, ratio_box, ratio_conf, ratio_cls, norm_counts ? norm_counts + l : nullptr, norm_batch);
            FVA_LAUNCH_CHECK("match_loss_kernel");
        }
        const int cblocks = (int)((ncell + 255) / 256 < CONF_BLOCKS ? (ncell + 255) / 256 : CONF_BLOCKS);
        hipLaunchKernelGGL(conf_kernel, dim3(cblocks), dim3(256), 0, s, lv, (const int32_t*)cellmatch, (const float*)iou, conf_part,
                           ratio_conf);
        FVA_LAUNCH_CHECK("conf_kernel");
        fa.count[l] = mb.count; fa.box_l[l] = box_l; fa.cls_l[l] = cls_l; fa.conf_part[l] = conf_part;
        fa.conf_blocks[l] = cblocks; fa.ncell[l] = (double)ncell; fa.C[l] = lv.K - 5;
    }
    hipLaunchKernelGGL(yolo_final_kernel, dim3(1), dim3(256), 0, s, fa, ratio_box, ratio_conf, ratio_cls, loss_out);
    FVA_LAUNCH_CHECK("yolo_final_kernel");
    return FVA_OK;
}

static DemoBuf carve_demo(Carver& c, int T, const fva_head_level& l) {
    DemoBuf d;
    const int t1 = T > 0 ? T : 1;
    d.img_start = c.take<int32_t>(l.B); d.img_count = c.take<int32_t>(l.B); d.perm = c.take<int32_t>(t1);
    d.mask = c.take<int8_t>((int64_t)l.B * l.H * l.W * l.A);
    d.l_xy = c.take<float>(t1); d.l_wh = c.take<float>(t1); d.l_cls = c.take<float>(t1);
    d.conf_part = c.take<float>(2 * CONF_BLOCKS);
    d.nvalid = c.take<float>(1);
    d.cell = c.take<int32_t>(t1);
    return d;
}

int64_t fva_demo_loss_workspace(int32_t T, const fva_head_level* levels, int32_t nlevels) {
    Carver c{nullptr};
    c.take<double>(8);
    for (int l = 0; l < nlevels; ++l) carve_demo(c, T, levels[l]);
    return c.off;
}

int fva_demo_loss(const float* targets, int32_t T, const fva_head_level* levels, int32_t nlevels, float* loss_out, void* workspace,
                  int64_t workspace_bytes, void* stream) {
    if (!levels || nlevels < 1 || nlevels > 4 || !loss_out || !workspace || !targets || T < 1)
        return fva_fail(FVA_ERR_ARG, "fva_demo_loss: bad argument (the reference needs >= 1 target)");
    if (workspace_bytes < fva_demo_loss_workspace(T, levels, nlevels)) return fva_fail(FVA_ERR_WORKSPACE, "fva_demo_loss: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Carver c{(char*)workspace};
    double* acc = c.take<double>(8);
    if (hipMemsetAsync(acc, 0, 64, s) != hipSuccess) return fva_fail(FVA_ERR_LAUNCH, "fva_demo_loss: memset failed");
    for (int l = 0; l < nlevels; ++l) {
        int rc = check_level(levels[l], "fva_demo_loss");
        if (rc) return rc;
        const Level lv = to_level(levels[l]);
        const DemoBuf db = carve_demo(c, T, levels[l]);
        const int64_t n = (int64_t)lv.B * lv.H * lv.W * lv.A;
        const int blocks = (int)((n + 255) / 256 < CONF_BLOCKS ? (n + 255) / 256 : CONF_BLOCKS);
        hipLaunchKernelGGL(demo_group_kernel, dim3(1), dim3(256), 2 * lv.B * 4, s, targets, T, lv.B, db.img_start, db.img_count, db.perm);
        FVA_LAUNCH_CHECK("demo_group_kernel");
        hipLaunchKernelGGL(demo_mask_kernel, dim3(blocks), dim3(256), 0, s, targets, lv, db);
        FVA_LAUNCH_CHECK("demo_mask_kernel");
        hipLaunchKernelGGL(demo_target_kernel, dim3(cdiv(T, 4) < 2048 ? cdiv(T, 4) : 2048), dim3(256), 0, s, targets, T, lv, db);
        FVA_LAUNCH_CHECK("demo_target_kernel");
        hipLaunchKernelGGL(demo_positive_kernel, dim3(cdiv(T, 256)), dim3(256), 0, s, T, db);
        FVA_LAUNCH_CHECK("demo_positive_kernel");
        hipLaunchKernelGGL(demo_conf_kernel, dim3(blocks), dim3(256), 0, s, lv, db, 0);
        FVA_LAUNCH_CHECK("demo_conf_kernel");
        hipLaunchKernelGGL(demo_level_final_kernel, dim3(1), dim3(256), 0, s, db, T, lv.K - 5, blocks, acc, loss_out, l == nlevels - 1 ? 1 : 0);
        FVA_LAUNCH_CHECK("demo_level_final_kernel");
        if (lv.grad) {
            hipLaunchKernelGGL(demo_conf_kernel, dim3(blocks), dim3(256), 0, s, lv, db, 1);
            FVA_LAUNCH_CHECK("demo_conf_kernel");
        }
    }
    return FVA_OK;
}

int fva_scale_by_device_scalar(float* x, int64_t n, const float* scale, void* stream) {
    if (!x || !scale || n < 0) return fva_fail(FVA_ERR_ARG, "fva_scale_by_device_scalar: bad argument");
    if (n == 0) return FVA_OK;
    int64_t g = (n / 4 + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(scale_by_scalar_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, n, scale);
    FVA_LAUNCH_CHECK("scale_by_scalar_kernel");
    return FVA_OK;
}

int fva_bce_loss(const float* y, const int64_t* label, const float* dense_target, const float* weights, int64_t weights_numel,
                 int64_t numel, int32_t C, int32_t already_sigmoid, int32_t mean, float* loss_out, float* grad, float* workspace,
                 void* stream) {
    if (!y || (!label && !dense_target) || !loss_out || !workspace || numel < 1 || C < 1 || numel % C)
        return fva_fail(FVA_ERR_ARG, "fva_bce_loss: bad argument");
    if (weights && weights_numel != 1 && weights_numel != numel) return fva_fail(FVA_ERR_ARG, "fva_bce_loss: weights must have 1 or numel entries");
    const int blocks = (int)((numel + 255) / 256 < CONF_BLOCKS ? (numel + 255) / 256 : CONF_BLOCKS);
    hipLaunchKernelGGL(bce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, label, dense_target, weights, weights_numel, numel, C,
                       already_sigmoid, workspace, grad);
    FVA_LAUNCH_CHECK("bce_kernel");
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, blocks, mean ? (double)numel : 1.0,
                       loss_out);
    FVA_LAUNCH_CHECK("bce_final_kernel");
    return FVA_OK;
}

int fva_row_loss(const float* logits, const int64_t* labels, int32_t R, int32_t C, int32_t mode, float gamma, float* loss_out, float* grad,
                 float* workspace, void* stream) {
    if (!logits || !labels || !loss_out || !workspace || R < 1 || C < 1 || mode < 0 || mode > 1) return fva_fail(FVA_ERR_ARG, "fva_row_loss: bad argument");
    hipLaunchKernelGGL(row_loss_kernel, dim3(cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, logits, labels, R, C, mode, gamma, workspace, grad);
    FVA_LAUNCH_CHECK("row_loss_kernel");
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, R, (double)R, loss_out);
    FVA_LAUNCH_CHECK("bce_final_kernel");
    return FVA_OK;
}

int fva_smooth_l1(const float* pred, const float* target, int64_t n, float* loss_out, float* grad, float* workspace, void* stream) {
    if (!pred || !target || !loss_out || !workspace || n < 1) return fva_fail(FVA_ERR_ARG, "fva_smooth_l1: bad argument");
    const int blocks = (int)((n + 255) / 256 < CONF_BLOCKS ? (n + 255) / 256 : CONF_BLOCKS);
    hipLaunchKernelGGL(smooth_l1_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, n, workspace, grad);
    FVA_LAUNCH_CHECK("smooth_l1_kernel");
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, blocks, (double)n, loss_out);
    FVA_LAUNCH_CHECK("bce_final_kernel");
    return FVA_OK;
}

int fva_iou_pairwise(int kind, int mode, int variant, const float* a, const float* b, float* out, float* grad_a, int64_t N, float eps,
                     void* stream) {
    if (!a || !b || !out || kind < 0 || kind > 3 || mode < 0 || mode > 2) return fva_fail(FVA_ERR_ARG, "fva_iou_pairwise: bad argument");
    if (N == 0) return FVA_OK;
    hipLaunchKernelGGL(iou_pair_kernel, dim3((int)((N + 255) / 256 < 2048 ? (N + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream,
                       kind, mode, variant, a, b, out, grad_a, N, eps);
    FVA_LAUNCH_CHECK("iou_pair_kernel");
    return FVA_OK;
}

int fva_iou_batch(int kind, int mode, int variant, const float* a, const float* b, float* out, int64_t N, int64_t M, float eps,
                  void* stream) {
    if (!a || !b || !out || kind < 0 || kind > 3 || mode < 0 || mode > 2) return fva_fail(FVA_ERR_ARG, "fva_iou_batch: bad argument");
    if (N * M == 0) return FVA_OK;
    hipLaunchKernelGGL(iou_batch_kernel, dim3((int)((N * M + 255) / 256 < 2048 ? (N * M + 255) / 256 : 2048)), dim3(256), 0,
                       (hipStream_t)stream, kind, mode, variant, a, b, out, N, M, eps);
    FVA_LAUNCH_CHECK("iou_batch_kernel");
    return FVA_OK;
}

}  // extern "C"
