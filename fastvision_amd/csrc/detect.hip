// Validation side of the detection path for gfx950 (scope row f-2): eval decode of the three head tensors, candidate
// selection, and greedy non-maximum suppression, all on the device.  Byte/compare work, HBM- and latency-bound; built
// with -ffp-contract=off so the IoU test reproduces the CPU arithmetic bit for bit.
//
// Replaces: Yolov3.forward eval branch (reference detection/models/yolov3.py:35-53), postProcess
// (demos/yolov3_u/inference.py:58-106), non_max_suppression (detection/tools/NMS.py:5-23,
// demos/yolov3_u/utils/nms.py:5-98) including the torchvision.ops.nms call underneath them.
#include "common.h"

namespace {

constexpr int MAX_LEVELS = 4;

struct DecodeParams {
    fva_head_level lv[MAX_LEVELS];
    int row0[MAX_LEVELS + 1];
    FastDiv div_k, div_w[MAX_LEVELS], div_a[MAX_LEVELS];
    int nlevels, variant, has_lb;
    fva_letterbox lb;
    int rows, per_image;  // rows per image, rows * K
    int K;
};

__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__global__ __launch_bounds__(256) void decode_kernel(const DecodeParams kp, float* __restrict__ out) {
    // the level table is indexed by a per-thread level and anchor: keep it in LDS (dynamic indexing of kernel
    // arguments would go through scratch memory)
    __shared__ DecodeParams p;
    static_assert(sizeof(DecodeParams) % 4 == 0, "copied as dwords");
    for (int q = threadIdx.x; q < (int)(sizeof(DecodeParams) / 4); q += 256) ((uint32_t*)&p)[q] = ((const uint32_t*)&kp)[q];
    __syncthreads();
    // one image per blockIdx.y; threads walk the elements in INPUT order (level, y, x, a, k): the head kernels write
    // [B][H][W][A*K], so a wave reads whole 128-B lines; 32-bit index arithmetic with precomputed reciprocals
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= p.per_image) return;
    const int b = blockIdx.y;
    const int r = (int)fd_div((uint32_t)e, p.div_k);
    const int k = e - r * p.K;
    int l = 0;
#pragma unroll
    for (int q = 1; q < MAX_LEVELS; ++q)
        if (q < p.nlevels && r >= p.row0[q]) l = q;
    const fva_head_level& lv = p.lv[l];
    const int rr = r - p.row0[l];
    const int c = (int)fd_div((uint32_t)rr, p.div_a[l]);
    const int a = rr - c * lv.A;
    const int y = (int)fd_div((uint32_t)c, p.div_w[l]);
    const int x = c - y * lv.W;
    // output row: (a, y, x) for the library variant, (y, x, a) for the demo
    const int orow = p.row0[l] + (p.variant == 0 ? a * lv.H * lv.W + c : rr);
    const int64_t i = ((int64_t)b * p.rows + orow) * p.K + k;
    const float* src = lv.data + b * lv.sb + a * lv.sa + y * lv.sy + x * lv.sx;
    // centre / size of axis 0 (x, w) or 1 (y, h) in input pixels
    auto centre = [&](int axis) {
        const float cell = axis ? (float)y : (float)x;
        const float s = sigmoid_acc(src[axis * lv.sk]);
        return p.variant == 0 ? (s + cell) * lv.stride : (s * 2.f - 0.5f + cell) * lv.stride;
    };
    auto extent = [&](int axis) {
        const float anc = axis ? lv.anchor_h[a] : lv.anchor_w[a];
        const float t = src[(2 + axis) * lv.sk];
        if (p.variant == 0) return expf(t) * anc;
        const float s2 = sigmoid_acc(t) * 2.f;
        return s2 * s2 * anc * lv.stride;
    };
    // demo postProcess (inference.py:90-106): undo the letterbox, clamp to the original image
    auto extent_ori = [&](int axis) { return clampf(extent(axis) / p.lb.resize_ratio, 0.f, axis ? p.lb.ori_h : p.lb.ori_w); };
    if (k >= 4) {
        float v = sigmoid_acc(src[k * lv.sk]);
        // ... and drop boxes whose w or h is <= min_wh: their objectness is stored as -1, below any threshold
        if (k == 4 && p.has_lb && !(extent_ori(0) > p.lb.min_wh && extent_ori(1) > p.lb.min_wh)) v = -1.f;
        out[i] = v;
        return;
    }
    const int axis = k & 1;
    if (!p.has_lb) {
        out[i] = k < 2 ? centre(axis) : extent(axis);
        return;
    }
    const float lim = axis ? p.lb.ori_h : p.lb.ori_w;
    const float ctr = clampf((centre(axis) - (axis ? p.lb.pad_top : p.lb.pad_left)) / p.lb.resize_ratio, 0.f, lim - 1.f);
    const float half = extent_ori(axis) / 2.f;
    out[i] = clampf(k < 2 ? ctr - half : ctr + half, 0.f, lim - 1.f);
}

// ---- candidates ---------------------------------------------------------------------------------------------------
// pass A: one wave per 64 rows.  flag = obj > conf_thres; score/category of flagged rows by a wave-wide arg-max over
// the classes (first maximum wins, as torch.max does).
__global__ __launch_bounds__(256) void cand_score_kernel(const float* __restrict__ pred, int R, int K, fva_nms_params p,
                                                         uint8_t* __restrict__ flag, float* __restrict__ tscore,
                                                         int32_t* __restrict__ tcat) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int row0 = (blockIdx.x * 4 + w) * 64;
    if (row0 >= R) return;
    const float* P = pred + (int64_t)b * R * K;
    const int r = row0 + lane;
    const float obj = r < R ? P[(int64_t)r * K + 4] : 0.f;
    bool ok = r < R && obj > p.conf_thres;
    uint64_t m = __ballot(ok);
    float my_score = 0.f;
    int my_cat = 0;
    const int C = K - 5;
    while (m) {
        const int src = __builtin_ctzll(m);
        m &= m - 1;
        const int rr = row0 + src;
        const float o = __shfl(obj, src);
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = lane; c < C; c += 64) {
            const float v = P[(int64_t)rr * K + 5 + c] * o;
            if (v > best) { best = v; bi = c; }
        }
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) {
            const float ov = __shfl_xor(best, o2);
            const int oi = __shfl_xor(bi, o2);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == src) {
            my_score = p.score_mode == 0 ? best : o;
            my_cat = bi == 0x7fffffff ? 0 : bi;
            if (p.rethreshold && !(best > p.conf_thres)) ok = false;
        }
    }
    if (r < R) {
        flag[(int64_t)b * R + r] = ok ? 1 : 0;
        tscore[(int64_t)b * R + r] = my_score;
        tcat[(int64_t)b * R + r] = my_cat;
    }
}

// pass B: one block per image, ordered compaction (candidate order = row order, as boolean-mask indexing gives)
__global__ __launch_bounds__(1024) void cand_compact_kernel(const float* __restrict__ pred, int R, int K, int box_mode,
                                                            const uint8_t* __restrict__ flag, const float* __restrict__ tscore,
                                                            const int32_t* __restrict__ tcat, float* __restrict__ cbox,
                                                            float* __restrict__ cscore, int32_t* __restrict__ ccat,
                                                            int32_t* __restrict__ crow, int32_t* __restrict__ counts) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t off = (int64_t)b * R;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int r0 = 0; r0 < R; r0 += 1024) {
        const int r = r0 + tid;
        const bool ok = r < R && flag[off + r];
        const uint64_t m = __ballot(ok);
        const int before = __popcll(m & ((1ull << lane) - 1));
        if (lane == 0) wsum[w] = __popcll(m);
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (q < w) wbase += wsum[q];
            tot += wsum[q];
        }
        const int base = base_s;
        if (ok) {
            const int64_t d = off + base + wbase + before;
            const float* pr = pred + (off + r) * K;
            float x1 = pr[0], y1 = pr[1], x2 = pr[2], y2 = pr[3];
            if (box_mode == 0) {  // xywh -> xyxy (BOX.py: x -/+ w/2)
                const float hw = x2 / 2.f, hh = y2 / 2.f;
                x2 = x1 + hw; y2 = y1 + hh; x1 = x1 - hw; y1 = y1 - hh;
            }
            cbox[d * 4 + 0] = x1; cbox[d * 4 + 1] = y1; cbox[d * 4 + 2] = x2; cbox[d * 4 + 3] = y2;
            cscore[d] = tscore[off + r];
            ccat[d] = tcat[off + r];
            crow[d] = r;
        }
        __syncthreads();
        if (tid == 0) base_s = base + tot;
        __syncthreads();
    }
    if (tid == 0) counts[b] = base_s;
}

// ---- selection ----------------------------------------------------------------------------------------------------
// stable rank by score, highest first (ties: lower candidate index first); writes the NMS boxes (class gap applied) in
// rank order and the rank -> candidate permutation
__global__ __launch_bounds__(256) void rank_kernel(const float* __restrict__ cbox, const float* __restrict__ cscore,
                                                   const int32_t* __restrict__ ccat, const int32_t* __restrict__ counts,
                                                   int R, int nmax, float gap, float* __restrict__ sbox,
                                                   int32_t* __restrict__ perm) {
    __shared__ float tile[256];
    const int b = blockIdx.y, n = counts[b];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= n) return;
    const float* S = cscore + (int64_t)b * R;
    const float si = i < n ? S[i] : 0.f;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + threadIdx.x;
        __syncthreads();
        tile[threadIdx.x] = j < n ? S[j] : -INFINITY;
        __syncthreads();
        const int lim = min(256, n - j0);
        for (int q = 0; q < lim; ++q) {
            const float sj = tile[q];
            rank += (sj > si || (sj == si && j0 + q < i)) ? 1 : 0;
        }
    }
    if (i < n) {
        const float g = (float)ccat[(int64_t)b * R + i] * gap;
        const float* bx = cbox + ((int64_t)b * R + i) * 4;
        float* d = sbox + ((int64_t)b * nmax + rank) * 4;
        d[0] = bx[0] + g; d[1] = bx[1] + g; d[2] = bx[2] + g; d[3] = bx[3] + g;
        perm[(int64_t)b * nmax + rank] = i;
    }
}

// torchvision's IoU test: inter / (area_a + area_b - inter) > thr
__device__ __forceinline__ bool overlaps(const float* a, const float* b, float thr) {
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    const float sa = (a[2] - a[0]) * (a[3] - a[1]);
    const float sb = (b[2] - b[0]) * (b[3] - b[1]);
    return inter / (sa + sb - inter) > thr;
}

// mask[b][i][cb] bit j: sorted box cb*64+j (j-th of column block cb, later in the order than i) overlaps sorted box i
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ sbox, const int32_t* __restrict__ counts,
                                                      int nmax, int max_nms, float thr, uint64_t* __restrict__ mask) {
    const int b = blockIdx.z;
    int n = counts[b];
    if (max_nms > 0 && n > max_nms) n = max_nms;
    const int cb = blockIdx.x, rb = blockIdx.y;
    if (rb * 64 >= n || cb * 64 >= n || cb < rb) return;
    __shared__ float cbx[64 * 4];
    const float* S = sbox + (int64_t)b * nmax * 4;
    const int cn = min(64, n - cb * 64);
    if ((int)threadIdx.x < cn) {
#pragma unroll
        for (int q = 0; q < 4; ++q) cbx[threadIdx.x * 4 + q] = S[(cb * 64 + threadIdx.x) * 4 + q];
    }
    __syncthreads();
    const int i = rb * 64 + threadIdx.x;
    if (i < n) {
        float a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = S[i * 4 + q];
        uint64_t bits = 0;
        const int start = cb == rb ? threadIdx.x + 1 : 0;
        for (int j = start; j < cn; ++j)
            if (overlaps(a, cbx + j * 4, thr)) bits |= 1ull << j;
        const int cbs = (nmax + 63) / 64;
        mask[((int64_t)b * nmax + i) * cbs + cb] = bits;
    }
}

// greedy walk in rank order (one wave per image); stops after max_det kept boxes
__global__ __launch_bounds__(64) void nms_scan_kernel(const uint64_t* __restrict__ mask, const int32_t* __restrict__ counts,
                                                      int R, int nmax, int max_nms, int max_det,
                                                      const int32_t* __restrict__ perm, const float* __restrict__ cbox,
                                                      const float* __restrict__ cscore, const int32_t* __restrict__ ccat,
                                                      const int32_t* __restrict__ crow, float* __restrict__ out,
                                                      int32_t* __restrict__ out_rows, int32_t* __restrict__ keep_counts) {
    extern __shared__ uint64_t remv[];
    const int b = blockIdx.x, lane = threadIdx.x;
    int n = counts[b];
    if (max_nms > 0 && n > max_nms) n = max_nms;
    const int cbs = (nmax + 63) / 64, ncb = (n + 63) / 64;
    for (int q = lane; q < ncb; q += 64) remv[q] = 0;
    __syncthreads();
    int kept = 0;
    for (int i = 0; i < n && kept < max_det; ++i) {
        const uint64_t word = remv[i >> 6];
        if ((word >> (i & 63)) & 1) continue;  // uniform over the wave
        const int c = perm[(int64_t)b * nmax + i];
        const int64_t ci = (int64_t)b * R + c;
        if (lane < 4) out[((int64_t)b * max_det + kept) * 6 + lane] = cbox[ci * 4 + lane];
        if (lane == 4) out[((int64_t)b * max_det + kept) * 6 + 4] = cscore[ci];
        if (lane == 5) out[((int64_t)b * max_det + kept) * 6 + 5] = (float)ccat[ci];
        if (lane == 6) out_rows[(int64_t)b * max_det + kept] = crow[ci];
        ++kept;
        const uint64_t* mrow = mask + ((int64_t)b * nmax + i) * cbs;
        __syncthreads();
        for (int q = (i >> 6) + lane; q < ncb; q += 64) remv[q] |= mrow[q];
        __syncthreads();
    }
    if (lane == 0) keep_counts[b] = kept;
}

struct CandLayout {
    uint8_t* flag;
    float* tscore;
    int32_t* tcat;
    float* cbox;
    float* cscore;
    int32_t* ccat;
    int32_t* crow;
};
inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }
inline int64_t cand_layout(void* ws, int64_t B, int64_t R, CandLayout* L) {
    char* p = (char*)ws;
    int64_t o = 0;
    auto take = [&](int64_t bytes) { char* q = p ? p + o : nullptr; o += align256(bytes); return q; };
    uint8_t* flag = (uint8_t*)take(B * R);
    float* tscore = (float*)take(B * R * 4);
    int32_t* tcat = (int32_t*)take(B * R * 4);
    float* cbox = (float*)take(B * R * 16);
    float* cscore = (float*)take(B * R * 4);
    int32_t* ccat = (int32_t*)take(B * R * 4);
    int32_t* crow = (int32_t*)take(B * R * 4);
    if (L) *L = CandLayout{flag, tscore, tcat, cbox, cscore, ccat, crow};
    return o;
}

}  // namespace

extern "C" int fva_yolo_decode(const fva_head_level* levels, int32_t nlevels, int32_t variant, const fva_letterbox* lb,
                               float* out, int64_t rows_per_image, void* stream) {
    if (!levels || !out) return fva_fail(FVA_ERR_ARG, "fva_yolo_decode: null pointer");
    if (nlevels < 1 || nlevels > MAX_LEVELS) return fva_fail(FVA_ERR_ARG, "fva_yolo_decode: nlevels %d not in 1..%d", nlevels, MAX_LEVELS);
    if (variant != 0 && variant != 1) return fva_fail(FVA_ERR_ARG, "fva_yolo_decode: bad variant %d", variant);
    if (lb && variant != 1) return fva_fail(FVA_ERR_ARG, "fva_yolo_decode: the letterbox mapping belongs to the demo variant");
    DecodeParams p{};
    int64_t rows = 0;
    for (int l = 0; l < nlevels; ++l) {
        const fva_head_level& lv = levels[l];
        if (!lv.data || lv.A < 1 || lv.A > 8 || lv.H < 1 || lv.W < 1 || lv.B != levels[0].B || lv.K != levels[0].K || lv.K < 6)
            return fva_fail(FVA_ERR_ARG, "fva_yolo_decode: bad level %d", l);
        if (rows + (int64_t)lv.A * lv.H * lv.W >= (1ll << 24)) return fva_fail(FVA_ERR_ARG, "fva_yolo_decode: too many rows per image");
        p.lv[l] = lv;
        p.row0[l] = (int)rows;
        p.div_w[l] = make_fastdiv(lv.W);
        p.div_a[l] = make_fastdiv(lv.A);
        rows += (int64_t)lv.A * lv.H * lv.W;
    }
    p.row0[nlevels] = (int)rows;
    if (rows != rows_per_image) return fva_fail(FVA_ERR_ARG, "fva_yolo_decode: rows_per_image %lld, levels hold %lld", (long long)rows_per_image, (long long)rows);
    p.nlevels = nlevels;
    p.variant = variant;
    p.has_lb = lb ? 1 : 0;
    if (lb) {
        if (!(lb->resize_ratio > 0.f)) return fva_fail(FVA_ERR_ARG, "fva_yolo_decode: resize_ratio must be positive");
        p.lb = *lb;
    }
    p.rows = (int)rows;
    p.K = levels[0].K;
    if (rows * p.K >= (1ll << 31) || levels[0].B > 65535) return fva_fail(FVA_ERR_ARG, "fva_yolo_decode: too large");
    p.per_image = (int)(rows * p.K);
    p.div_k = make_fastdiv(p.K);
    hipLaunchKernelGGL(decode_kernel, dim3(cdiv(p.per_image, 256), levels[0].B), dim3(256), 0, (hipStream_t)stream, p, out);
    FVA_LAUNCH_CHECK("decode_kernel");
    return FVA_OK;
}

static int check_nms(const fva_nms_params* p, const char* who) {
    if (!p) return fva_fail(FVA_ERR_ARG, "%s: null params", who);
    if ((p->box_mode | 1) != 1 || (p->score_mode | 1) != 1) return fva_fail(FVA_ERR_ARG, "%s: bad box_mode/score_mode", who);
    if (p->max_det < 1) return fva_fail(FVA_ERR_ARG, "%s: max_det must be >= 1", who);
    return FVA_OK;
}

extern "C" int64_t fva_nms_candidates_workspace(int32_t B, int32_t R) {
    if (B < 1 || R < 1) return 0;
    return cand_layout(nullptr, B, R, nullptr);
}

extern "C" int fva_nms_candidates(const float* pred, int32_t B, int32_t R, int32_t K, const fva_nms_params* p, void* cand,
                                  int64_t cand_bytes, int32_t* counts, void* stream) {
    int rc = check_nms(p, "fva_nms_candidates");
    if (rc) return rc;
    if (!pred || !cand || !counts) return fva_fail(FVA_ERR_ARG, "fva_nms_candidates: null pointer");
    if (B < 1 || R < 1 || K < 6) return fva_fail(FVA_ERR_ARG, "fva_nms_candidates: bad shape B=%d R=%d K=%d", B, R, K);
    if (cand_bytes < cand_layout(nullptr, B, R, nullptr)) return fva_fail(FVA_ERR_ARG, "fva_nms_candidates: workspace too small");
    if (B > 65535) return fva_fail(FVA_ERR_ARG, "fva_nms_candidates: B=%d exceeds 65535", B);
    CandLayout L;
    cand_layout(cand, B, R, &L);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cand_score_kernel, dim3(cdiv(R, 256), B), dim3(256), 0, s, pred, R, K, *p, L.flag, L.tscore, L.tcat);
    FVA_LAUNCH_CHECK("cand_score_kernel");
    hipLaunchKernelGGL(cand_compact_kernel, dim3(B), dim3(1024), 0, s, pred, R, K, p->box_mode, L.flag, L.tscore, L.tcat, L.cbox,
                       L.cscore, L.ccat, L.crow, counts);
    FVA_LAUNCH_CHECK("cand_compact_kernel");
    return FVA_OK;
}

extern "C" int64_t fva_nms_select_workspace(int32_t B, int32_t nmax) {
    if (B < 1 || nmax < 1) return 0;
    const int64_t cbs = (nmax + 63) / 64;
    return align256((int64_t)B * nmax * 16) + align256((int64_t)B * nmax * 4) + align256((int64_t)B * nmax * cbs * 8);
}

extern "C" int fva_nms_select(const void* cand, const int32_t* counts, int32_t B, int32_t R, int32_t nmax, const fva_nms_params* p,
                              void* workspace, int64_t workspace_bytes, float* out, int32_t* out_rows, int32_t* keep_counts,
                              void* stream) {
    int rc = check_nms(p, "fva_nms_select");
    if (rc) return rc;
    if (!cand || !counts || !workspace || !out || !out_rows || !keep_counts) return fva_fail(FVA_ERR_ARG, "fva_nms_select: null pointer");
    if (B < 1 || B > 65535 || R < 1 || nmax < 1 || nmax > R) return fva_fail(FVA_ERR_ARG, "fva_nms_select: bad shape B=%d R=%d nmax=%d", B, R, nmax);
    if (workspace_bytes < fva_nms_select_workspace(B, nmax)) return fva_fail(FVA_ERR_ARG, "fva_nms_select: workspace too small");
    const int64_t cbs = (nmax + 63) / 64;
    if (cbs * 8 > 64 * 1024) return fva_fail(FVA_ERR_ARG, "fva_nms_select: nmax %d too large (suppression bitmap exceeds 64 KiB of LDS)", nmax);
    CandLayout L;
    cand_layout(const_cast<void*>(cand), B, R, &L);
    char* w = (char*)workspace;
    float* sbox = (float*)w;
    int32_t* perm = (int32_t*)(w + align256((int64_t)B * nmax * 16));
    uint64_t* mask = (uint64_t*)(w + align256((int64_t)B * nmax * 16) + align256((int64_t)B * nmax * 4));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(rank_kernel, dim3(cdiv(nmax, 256), B), dim3(256), 0, s, L.cbox, L.cscore, L.ccat, counts, R, nmax, p->class_gap,
                       sbox, perm);
    FVA_LAUNCH_CHECK("rank_kernel");
    hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)cbs, (unsigned)cbs, B), dim3(64), 0, s, sbox, counts, nmax, p->max_nms, p->iou_thres, mask);
    FVA_LAUNCH_CHECK("nms_mask_kernel");
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), (size_t)cbs * 8, s, mask, counts, R, nmax, p->max_nms, p->max_det, perm, L.cbox,
                       L.cscore, L.ccat, L.crow, out, out_rows, keep_counts);
    FVA_LAUNCH_CHECK("nms_scan_kernel");
    return FVA_OK;
}

// ---- RPN proposal rows (demos/faster_rcnn/models/rpn.py:110-186) ---------------------------------------------
// filter_proposals up to the per-image top-k: decode the regression against the anchor grid, objectness = softmax over the two
// logits, xywh -> xyxy, clamp to the feature map.  One row per (y, x, a) in the reference's view(bs, -1, 5) order, laid out as
// the NMS stage expects them: x1, y1, x2, y2, score, 1.0.  Quirk kept: BOTH extents use exp(d[2]) (rpn.py:118-119).
namespace {
__global__ __launch_bounds__(256) void rpn_decode_kernel(const float* __restrict__ cls, const float* __restrict__ d,
                                                         const float* __restrict__ anchors, float* __restrict__ out, int B, int H, int W,
                                                         int A) {
    const int64_t n = (int64_t)B * H * W * A;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int a = (int)(i % A);
        const int64_t cell = i / A;
        const int x = (int)(cell % W), y = (int)((cell / W) % H);
        const float aw = anchors[2 * a], ah = anchors[2 * a + 1];
        const float4 t = *(const float4*)(d + i * 4);
        const float2 c = *(const float2*)(cls + i * 2);
        const float xc = t.x * aw + (float)x, yc = t.y * ah + (float)y;
        const float e = expf(t.z);
        const float w = e * aw, h = e * ah;
        const float m = fmaxf(c.x, c.y), e0 = expf(c.x - m), e1 = expf(c.y - m);
        float* o = out + i * 6;
        o[0] = clampf(xc - w / 2.f, 0.f, (float)(W - 1));
        o[1] = clampf(yc - h / 2.f, 0.f, (float)(H - 1));
        o[2] = clampf(xc + w / 2.f, 0.f, (float)(W - 1));
        o[3] = clampf(yc + h / 2.f, 0.f, (float)(H - 1));
        o[4] = e1 / (e0 + e1);
        o[5] = 1.f;
    }
}
}  // namespace

extern "C" int fva_rpn_decode(const float* cls, const float* deltas, const float* anchors_wh, float* out, int32_t B, int32_t H, int32_t W,
                              int32_t A, void* stream) {
    if (!cls || !deltas || !anchors_wh || !out || B <= 0 || H <= 0 || W <= 0 || A <= 0) return fva_fail(FVA_ERR_ARG, "fva_rpn_decode: bad argument");
    const int64_t n = (int64_t)B * H * W * A;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(rpn_decode_kernel, dim3((unsigned)(blocks < 65535 ? blocks : 65535)), dim3(256), 0, (hipStream_t)stream, cls, deltas,
                       anchors_wh, out, B, H, W, A);
    FVA_LAUNCH_CHECK("rpn_decode_kernel");
    return FVA_OK;
}

// ---- RPN anchor / ground-truth matcher (demos/faster_rcnn/models/rpn.py:209-277) ------------------------------
// Per image: IoU of every anchor (xywh, feature cells) with every ground-truth box of that image (normalised xywh scaled by the
// map size), then the reference's labelling: label = index of the best box if its IoU > pos_thr, -1 (negative) if the best IoU <
// neg_thr, -2 (ignored) otherwise; finally every box claims the anchor it overlaps most, in box order (a later box overrides
// an earlier one on the same anchor).  Index work: bit-exact with the reference -- same fp32 expression order (this file is
// built with -ffp-contract=off), first maximum wins (torch.max).
namespace {
__device__ __forceinline__ float rpn_iou(float ax, float ay, float aw, float ah, float bx, float by, float bw, float bh) {
    const float ax1 = ax - aw / 2.f, ay1 = ay - ah / 2.f, ax2 = ax + aw / 2.f, ay2 = ay + ah / 2.f;
    const float bx1 = bx - bw / 2.f, by1 = by - bh / 2.f, bx2 = bx + bw / 2.f, by2 = by + bh / 2.f;
    const float area1 = (ax2 - ax1) * (ay2 - ay1), area2 = (bx2 - bx1) * (by2 - by1);
    const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
    const float inter = iw * ih;
    const float uni = area1 + area2 - inter + 1e-7f;
    return inter / uni;
}

// labels of phase 1: one thread per (image, anchor); the image's boxes are the rows of targets with column 0 == image
// (the Fast head's variant, fast.py:117-127: positive if best >= pos_thr, negative if neg_floor <= best < neg_thr, no claims)
__global__ __launch_bounds__(256) void rpn_label_kernel(const float* __restrict__ anchors, int Na, const float* __restrict__ targets, int T,
                                                        float fw, float fh, float pos_thr, float neg_thr, int32_t* __restrict__ labels,
                                                        int image_base, int pos_inclusive, float neg_floor) {
    const int b = image_base + blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= Na) return;
    const float4 an = *(const float4*)(anchors + (int64_t)a * 4);
    float best = -INFINITY;
    int best_t = 0, local = 0;
    for (int t = 0; t < T; ++t) {
        const float* tg = targets + (int64_t)t * 6;
        if (tg[0] != (float)b) continue;
        const float v = rpn_iou(an.x, an.y, an.z, an.w, tg[2] * fw, tg[3] * fh, tg[4] * fw, tg[5] * fh);
        if (v > best) { best = v; best_t = local; }
        ++local;
    }
    int lab = -2;
    if (local > 0) {
        if (pos_inclusive ? best >= pos_thr : best > pos_thr) lab = best_t;
        if (best < neg_thr && best >= neg_floor) lab = -1;          // the reference applies this test second: it wins when both hold
    }
    labels[(int64_t)blockIdx.y * Na + a] = lab;
}

// phase 2: one block per (box row): arg-max of its IoU over the anchors (first maximum), wave shuffles + LDS
__global__ __launch_bounds__(256) void rpn_best_anchor_kernel(const float* __restrict__ anchors, int Na, const float* __restrict__ targets,
                                                              float fw, float fh, int32_t* __restrict__ best_anchor) {
    const int t = blockIdx.x;
    const float* tg = targets + (int64_t)t * 6;
    const float bx = tg[2] * fw, by = tg[3] * fh, bw = tg[4] * fw, bh = tg[5] * fh;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int a = threadIdx.x; a < Na; a += 256) {
        const float4 an = *(const float4*)(anchors + (int64_t)a * 4);
        const float v = rpn_iou(an.x, an.y, an.z, an.w, bx, by, bw, bh);
        if (v > best) { best = v; idx = a; }     // ascending a per thread: keeps the first maximum
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(idx, o);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    __shared__ float sv[4];
    __shared__ int si[4];
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < idx)) { best = sv[w]; idx = si[w]; }
        best_anchor[t] = idx;
    }
}

// phase 3: per image, in box order: labels[best_anchor[box]] = local index of the box
__global__ void rpn_claim_kernel(const float* __restrict__ targets, int T, const int32_t* __restrict__ best_anchor, int Na,
                                 int32_t* __restrict__ labels) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    int local = 0;
    for (int t = 0; t < T; ++t) {
        if (targets[(int64_t)t * 6] != (float)b) continue;
        labels[(int64_t)b * Na + best_anchor[t]] = local;
        ++local;
    }
}
}  // namespace

extern "C" int fva_rpn_match(const float* anchors_xywh, int32_t Na, const float* targets, int32_t T, int32_t B, int32_t feature_h,
                             int32_t feature_w, float pos_thr, float neg_thr, int32_t* labels, int32_t* workspace, void* stream) {
    if (!anchors_xywh || !labels || Na <= 0 || B <= 0 || T < 0 || (T > 0 && (!targets || !workspace)))
        return fva_fail(FVA_ERR_ARG, "fva_rpn_match: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(rpn_label_kernel, dim3(cdiv(Na, 256), B), dim3(256), 0, s, anchors_xywh, Na, targets, T, (float)feature_w, (float)feature_h,
                       pos_thr, neg_thr, labels, 0, 0, -INFINITY);
    FVA_LAUNCH_CHECK("rpn_label_kernel");
    if (T > 0) {
        hipLaunchKernelGGL(rpn_best_anchor_kernel, dim3(T), dim3(256), 0, s, anchors_xywh, Na, targets, (float)feature_w, (float)feature_h, workspace);
        FVA_LAUNCH_CHECK("rpn_best_anchor_kernel");
        hipLaunchKernelGGL(rpn_claim_kernel, dim3(B), dim3(64), 0, s, targets, T, workspace, Na, labels);
        FVA_LAUNCH_CHECK("rpn_claim_kernel");
    }
    return FVA_OK;
}

// Fast head (demos/faster_rcnn/models/fast.py:100-127): the proposals of ONE image against that image's boxes (targets rows with
// column 0 == image; their xywh already in feature cells, fast.py:217).  labels [N]: >= 0 matched box (best IoU >= pos_thr),
// -1 negative (neg_floor <= best IoU < neg_thr), -2 ignored.
extern "C" int fva_fast_match(const float* proposals_xywh, int32_t N, const float* targets, int32_t T, int32_t image, float pos_thr,
                              float neg_thr, float neg_floor, int32_t* labels, void* stream) {
    if (N < 0 || T < 0 || (N > 0 && (!proposals_xywh || !labels)) || (T > 0 && !targets)) return fva_fail(FVA_ERR_ARG, "fva_fast_match: bad argument");
    if (N == 0) return FVA_OK;
    hipLaunchKernelGGL(rpn_label_kernel, dim3(cdiv(N, 256), 1), dim3(256), 0, (hipStream_t)stream, proposals_xywh, N, targets, T, 1.f, 1.f, pos_thr,
                       neg_thr, labels, image, 1, neg_floor);
    FVA_LAUNCH_CHECK("rpn_label_kernel<fast>");
    return FVA_OK;
}
