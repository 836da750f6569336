// BatchNorm2d (train/eval) + SiLU (+ residual add) forward and backward, and the FPN upsample/concat and
// layout helpers, for gfx950.  All of these are HBM-bound streaming kernels: 16 bytes per lane, channels
// contiguous (NHWC), no LDS except for the per-block channel reductions.
//
// Algorithmic bytes per element (bf16): apply = 2 (y) + 2 (z) [+2 residual]; backward pass 1 = 4 (dz, y);
// backward pass 2 = 4 + 2 (dy).
//
// Replaces nn.BatchNorm2d / nn.SiLU / `identity + conv2` (reference classfication/models/darknet53.py:11-17,
// 28-31, 58-62) and nn.Upsample + torch.cat (detection/neck/yolov3neck.py:71,105,110).
#include "common.h"

namespace {


// ---------------------------------------------------------------------------------------------------------
// forward statistics -> coefficients.  Block = 16 channels x 64 partial-row groups (1024 threads): the partial
// table [nblocks][2][C] is summed with 4 independent loads in flight per thread, then in double across groups.
constexpr int FIN_G = 64;
__device__ __forceinline__ void reduce_partials(const float* __restrict__ part, int nblocks, int C, int c, int ry, double& s1,
                                                double& s2) {
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
    int b = ry;
    for (; b + FIN_G < nblocks; b += 2 * FIN_G) {
        const float p0 = part[((int64_t)b * 2 + 0) * C + c], q0 = part[((int64_t)b * 2 + 1) * C + c];
        const float p1 = part[((int64_t)(b + FIN_G) * 2 + 0) * C + c], q1 = part[((int64_t)(b + FIN_G) * 2 + 1) * C + c];
        a0 += p0; b0 += q0; a1 += p1; b1 += q1;
    }
    if (b < nblocks) { a0 += part[((int64_t)b * 2 + 0) * C + c]; b0 += part[((int64_t)b * 2 + 1) * C + c]; }
    s1 = (double)a0 + (double)a1;
    s2 = (double)b0 + (double)b1;
}
// Layers with many row blocks (the 640^2 / 320^2 maps: up to 51200 partial rows for 2-4 channel groups) first fold
// chunks of PRE_ROWS rows in parallel into doubles kept in the tail of the same table (rows nblocks .. of
// fva_bn_partial_rows()): tail[(r * 2 + which) * C + c].  Fixed order everywhere: deterministic.
constexpr int PRE_ROWS = 256, PRE_MIN = 1024;
__global__ __launch_bounds__(1024) void bn_prereduce_kernel(const float* __restrict__ part, int nblocks, int C, double* __restrict__ tail) {
    __shared__ double red[2][FIN_G][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    const int r0 = blockIdx.y * PRE_ROWS;
    const int n = nblocks - r0 < PRE_ROWS ? nblocks - r0 : PRE_ROWS;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) reduce_partials(part + (int64_t)r0 * 2 * C, n, C, c, ry, s1, s2);
    red[0][ry][cx] = s1;
    red[1][ry][cx] = s2;
    __syncthreads();
    if (ry < 2 && c < C) {
        double s = 0.0;
        for (int k = 0; k < FIN_G; ++k) s += red[ry][k][cx];
        tail[((int64_t)blockIdx.y * 2 + ry) * C + c] = s;
    }
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int nblocks, double count, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* running_mean, float* running_var, int64_t* nbt,
                                                           float momentum, float eps, float* save_mean, float* save_rstd,
                                                           float* scale, float* shift, const double* __restrict__ pre, int pre_rows) {
    __shared__ double red[2][FIN_G][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        if (pre_rows > 0) {  // pre-reduced doubles
            for (int r = ry; r < pre_rows; r += FIN_G) {
                s1 += pre[((int64_t)r * 2 + 0) * C + c];
                s2 += pre[((int64_t)r * 2 + 1) * C + c];
            }
        } else {
            reduce_partials(part, nblocks, C, c, ry, s1, s2);
        }
    }
    red[0][ry][cx] = s1;
    red[1][ry][cx] = s2;
    __syncthreads();
    if (ry == 0 && c < C) {
        s1 = s2 = 0.0;
        for (int k = 0; k < FIN_G; ++k) {
            s1 += red[0][k][cx];
            s2 += red[1][k][cx];
        }
        const BnN n = bn_n(count);
        const BnFwdCoef k = bn_fwd_coef(s1, s2, n, gamma[c], beta[c], eps);
        save_mean[c] = k.mean;
        save_rstd[c] = k.rstd;
        scale[c] = k.scale;
        shift[c] = k.shift;
        if (running_mean) bn_running_update(running_mean, running_var, c, k, n, momentum);
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

__global__ void bn_eval_coeffs_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sc = gamma[c] / sqrtf(rv[c] + eps);
        scale[c] = sc;
        shift[c] = beta[c] - rm[c] * sc;
    }
}

struct HaloIdx {
    int B, H, W, C, pad, Hp, Wp, cpp;  // cpp = 16-B chunks per pixel
    FastDiv div_cpp, div_wp, div_img;
    int64_t total;  // chunks over the padded buffer
};
HaloIdx make_halo(int B, int H, int W, int C, int pad, int epc) {
    HaloIdx h;
    h.B = B; h.H = H; h.W = W; h.C = C; h.pad = pad;
    h.Hp = H + 2 * pad; h.Wp = W + 2 * pad; h.cpp = C / epc;
    h.div_cpp = make_fastdiv(h.cpp);
    h.div_wp = make_fastdiv(h.Wp);
    h.div_img = make_fastdiv(h.Hp * h.Wp);
    h.total = (int64_t)B * h.Hp * h.Wp * h.cpp;
    return h;
}
// decode a chunk index of the padded buffer; returns false on the zero border
__device__ __forceinline__ bool halo_decode(const HaloIdx& h, uint32_t idx, int& b, int& y, int& x, int& cc) {
    const uint32_t pix = fd_div(idx, h.div_cpp);
    cc = idx - pix * h.cpp;
    b = fd_div(pix, h.div_img);
    const uint32_t rem = pix - (uint32_t)b * (h.Hp * h.Wp);
    const uint32_t yp = fd_div(rem, h.div_wp);
    const uint32_t xp = rem - yp * h.Wp;
    y = (int)yp - h.pad;
    x = (int)xp - h.pad;
    return y >= 0 && y < h.H && x >= 0 && x < h.W;
}

// The apply passes read their dense inputs (y; dz and y) for the last time before those tensors go cold; read non-temporal they leave
// the caches to what the pass WRITES, which the next convolution launches read.  Same-box A/B (bench.py, two runs each): 1053.6 / 1057.3
// against 1048.4 / 1046.4 img/s.  -DFVA_NT_BN=0 switches back; -DFVA_NT_RES=1 also reads the residual that way (measured neutral, as were
// non-temporal slab reads in the split-K reduce).
#ifndef FVA_NT_BN
#define FVA_NT_BN 1
#endif
#ifndef FVA_NT_RES
#define FVA_NT_RES 0
#endif
template <typename T>
__device__ __forceinline__ Vec16<T> ld_last(const T* p) {
#if FVA_NT_BN
    Vec16<T> r;
    const u32x4 raw = __builtin_nontemporal_load((const u32x4*)p);
    __builtin_memcpy(&r, &raw, 16);
    return r;
#else
    return *(const Vec16<T>*)p;
#endif
}

// One block per row of the padded output buffer (blockIdx.x = b * Hp + yp): no index division per chunk.  A thread keeps U chunks
// in flight per step (all loads of a step are issued before the first is used; masked chunks read a clamped address and store
// zeros -- no branch around a load): at one chunk per step the passes ran 4.3 (backward) - 5.6 TB/s, latency- not bandwidth-bound
// (PMC: 78 % of the wave cycles waiting, 10 % issuing).
#ifndef FVA_BN_UNROLL
#define FVA_BN_UNROLL 2      // measured 1 / 2 / 4 (tools/bench_bn.py, sum over the layer shapes): fwd 348 / 337 / 340 us, bwd 514 / 506 / 504 us
#endif
// FIN: the batch statistics arrive as fixed-point accumulators that the producing convolution's tiles added to (common.h fx_*); every
// block finalises ALL channels once in its prologue, through LDS (the arithmetic of bn_finalize_kernel, so all blocks agree to the bit),
// block 0 also writes mean / rstd / scale / shift for the backward pass, updates the running statistics and returns the layer's OTHER
// accumulator to zero (`zero`: the one this launch reads is still being read by its other blocks).  No finalize launch.
struct BnFwdAcc {
    long long* acc;          // [replicas][FX_WORDS][C]: the sums the producers added
    long long* zero;         // may be null: another accumulator of the same size that this launch returns to zero (the layer's other direction)
    int replicas;
    const float *gamma, *beta;
    float *running_mean, *running_var;
    long long* nbt;
    float momentum, eps;
    BnN n;
    float *save_mean, *save_rstd, *scale, *shift;
};
struct BnBwdAcc {
    long long* acc;          // sums of dU and dU * xhat
    long long* zero;
    int replicas;
    const float *gamma, *rstd, *mean, *shift;
    float *dgamma, *dbeta;
    int accumulate;
    BnN n;
};
// the accumulator finalised by a launch of its own: for consumers that cannot do it in their prologue (the thin fused 1x1 tile, foreign code)
__global__ __launch_bounds__(256) void bn_acc_finalize_kernel(const BnFwdAcc fin, int C) {     // ONE block; dynamic LDS: FX_WORDS * C int64
    extern __shared__ long long fx_words[];
    fx_gather_lds(fin.acc, C, fin.replicas, fx_words, threadIdx.x, 256);
    for (int c = threadIdx.x; c < C; c += 256) {
        double s1, s2;
        fx_value2(fx_from_lds(fx_words, C, c), s1, s2);
        const BnFwdCoef k = bn_fwd_coef(s1, s2, fin.n, fin.gamma[c], fin.beta[c], fin.eps);
        fin.save_mean[c] = k.mean; fin.save_rstd[c] = k.rstd; fin.scale[c] = k.scale; fin.shift[c] = k.shift;
        if (fin.running_mean) bn_running_update(fin.running_mean, fin.running_var, c, k, fin.n, fin.momentum);
    }
    fx_zero(fin.acc, C, fin.replicas, threadIdx.x, 256);          // nobody else reads it: this launch is the only consumer
    fx_zero(fin.zero, C, fin.replicas, threadIdx.x, 256);
    if (threadIdx.x == 0 && fin.nbt) *fin.nbt += 1;
}

// One padded row of z per block and step (grid = rows for the plain form; the accumulator form runs at most 2048 blocks that walk the rows,
// so that its prologue is paid once per block).  FIN: scale / shift come from the layer's accumulator -- all C channels once per block,
// C / 256 per thread, through LDS (dynamic: 2 C floats), while the first row's loads are already in flight; block 0 also writes them out
// for the backward pass, updates the running statistics and returns the OTHER direction's accumulator to zero (its own one is still being
// read by the other blocks: the layer's backward pass zeroes that).
template <typename T, bool FIN>
__global__ __launch_bounds__(256) void bn_silu_apply_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const T* __restrict__ res,
                                                            int res_pad, T* __restrict__ z, const HaloIdx h, const BnFwdAcc fin, int nrows) {
    constexpr int EPC = Vec16<T>::N, U = FVA_BN_UNROLL;
    extern __shared__ float fx_tab[];          // FIN: [2][C]
    const int row_chunks = h.Wp * h.cpp;
    const int cmask = h.cpp - 1, cshift = 31 - __builtin_clz(h.cpp);  // cpp is a power of two (BN channel counts)
    // cpp divides 256, so a lane keeps the same channel chunk for the whole row: its coefficients live in registers
    const int cc = threadIdx.x & cmask;
    const int rW = h.W + 2 * res_pad;
    const bool has_res = res != nullptr;
    Vec16<T> v[U], r[U];
    bool ok[U];
    const T *yrow = nullptr, *rrow = nullptr;
    auto set_row = [&](int row) -> bool {       // false: a border row
        const int b = row / h.Hp, yy = row - b * h.Hp - h.pad;
        if (yy < 0 || yy >= h.H) return false;
        yrow = y + ((int64_t)b * h.H + yy) * h.W * h.C;
        rrow = has_res ? res + (((int64_t)b * (h.H + 2 * res_pad) + yy + res_pad) * rW + res_pad) * h.C : nullptr;
        return true;
    };
    auto load = [&](int i0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 256;
            const int xx = (i >> cshift) - h.pad;
            ok[u] = i < row_chunks && xx >= 0 && xx < h.W;
            const int64_t off = (int64_t)(ok[u] ? xx : 0) * h.C + cc * EPC;
            v[u] = ld_last<T>(yrow + off);
            if (has_res) {
#if FVA_NT_RES
                r[u] = ld_last<T>(rrow + off);
#else
                r[u] = *(const Vec16<T>*)(rrow + off);
#endif
            }
        }
    };
    int row = blockIdx.x;
    bool row_in = row < nrows && set_row(row);
    if (FIN && row_in && (int)threadIdx.x < row_chunks) load(threadIdx.x);
    float sc[EPC], sh[EPC];
    if constexpr (FIN) {
        const bool writer = blockIdx.x == 0;
        long long* fx_words = (long long*)(fx_tab + 2 * h.C);      // replicas > 1: [FX_WORDS][C] behind the table
        if (fin.replicas > 1) fx_gather_lds(fin.acc, h.C, fin.replicas, fx_words, threadIdx.x, 256);
        for (int c = threadIdx.x; c < h.C; c += 256) {
            double s1, s2;
            if (fin.replicas > 1) fx_value2(fx_from_lds(fx_words, h.C, c), s1, s2);
            else fx_load2(fin.acc, h.C, 1, c, s1, s2);
            const BnFwdCoef k = bn_fwd_coef(s1, s2, fin.n, fin.gamma[c], fin.beta[c], fin.eps);
            fx_tab[c] = k.scale;
            fx_tab[h.C + c] = k.shift;
            if (writer) {
                fin.save_mean[c] = k.mean; fin.save_rstd[c] = k.rstd; fin.scale[c] = k.scale; fin.shift[c] = k.shift;
                if (fin.running_mean) bn_running_update(fin.running_mean, fin.running_var, c, k, fin.n, fin.momentum);
            }
        }
        if (writer) {
            if (threadIdx.x == 0 && fin.nbt) *fin.nbt += 1;
            fx_zero(fin.zero, h.C, fin.replicas, threadIdx.x, 256);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            sc[e] = fx_tab[cc * EPC + e];
            sh[e] = fx_tab[h.C + cc * EPC + e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            sc[e] = scale[cc * EPC + e];
            sh[e] = shift[cc * EPC + e];
        }
    }
    bool first = FIN;
    for (; row < nrows; row += gridDim.x, row_in = row < nrows && set_row(row)) {
        T* zrow = z + (int64_t)row * row_chunks * EPC;
        if (!row_in) {   // a border row: zeros (block-uniform branch)
            Vec16<T> zero;
#pragma unroll
            for (int e = 0; e < EPC; ++e) zero.set(e, 0.f);
            for (int i = threadIdx.x; i < row_chunks; i += 256) *(Vec16<T>*)(zrow + (int64_t)i * EPC) = zero;
            first = false;
            continue;
        }
        for (int i0 = threadIdx.x; i0 < row_chunks; i0 += 256 * U) {
            if (!first) load(i0);
            first = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * 256;
                Vec16<T> out;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    float o = bn_silu_fwd_elem(v[u].get(e), sc[e], sh[e]);
                    if (has_res) o += r[u].get(e);
                    out.set(e, ok[u] ? o : 0.f);
                }
                if (i < row_chunks) *(Vec16<T>*)(zrow + (int64_t)i * EPC) = out;
            }
        }
    }
}

// backward pass 1: per-channel partial sums of dU and dU*xhat.  threads = (channel chunk, pixel row group)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dz, const T* __restrict__ y,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ part, long long* acc, int replicas, int64_t M, int C, int rows_per_block) {
    constexpr int EPC = Vec16<T>::N;
    extern __shared__ float red[];  // [2][rpi][C]
    const int cpp = C / EPC, rpi = 256 / cpp;
    const int cx = threadIdx.x % cpp, py = threadIdx.x / cpp;
    float sc[EPC], sh[EPC], mu[EPC], rs[EPC], s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int c = cx * EPC + e;
        sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; rs[e] = rstd[c];
        s1[e] = s2[e] = 0.f;
    }
    const int64_t m0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t m1 = m0 + rows_per_block;
    if (m1 > M) m1 = M;
    for (int64_t m = m0 + py; m < m1; m += rpi) {
        const Vec16<T> g = *(const Vec16<T>*)(dz + m * C + cx * EPC);
        const Vec16<T> v = *(const Vec16<T>*)(y + m * C + cx * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float yv = v.get(e);
            const float du = g.get(e) * silu_grad(yv * sc[e] + sh[e]);
            s1[e] += du;
            s2[e] += du * (yv - mu[e]) * rs[e];
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        red[(0 * rpi + py) * C + cx * EPC + e] = s1[e];
        red[(1 * rpi + py) * C + cx * EPC + e] = s2[e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i - which * C;
        float s = 0.f;
        for (int k = 0; k < rpi; ++k) s += red[(which * rpi + k) * C + c];
        if (acc != nullptr) fx_atomic_add(fx_replica(acc, C, replicas), C, which, c, s);
        else part[((int64_t)blockIdx.x * 2 + which) * C + c] = s;
    }
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblocks, double count, int C,
                                                               const float* __restrict__ gamma, const float* __restrict__ rstd,
                                                               float* dgamma, float* dbeta, int accumulate, float* coef,
                                                               const double* __restrict__ pre, int pre_rows) {
    __shared__ double red[2][FIN_G][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        if (pre_rows > 0) {
            for (int r = ry; r < pre_rows; r += FIN_G) {
                s1 += pre[((int64_t)r * 2 + 0) * C + c];
                s2 += pre[((int64_t)r * 2 + 1) * C + c];
            }
        } else {
            reduce_partials(part, nblocks, C, c, ry, s1, s2);
        }
    }
    red[0][ry][cx] = s1;
    red[1][ry][cx] = s2;
    __syncthreads();
    if (ry == 0 && c < C) {
        s1 = s2 = 0.0;
        for (int k = 0; k < FIN_G; ++k) {
            s1 += red[0][k][cx];
            s2 += red[1][k][cx];
        }
        const BnBwdCoef k = bn_bwd_coef(s1, s2, bn_n(count), gamma[c], rstd[c]);
        dbeta[c] = accumulate ? dbeta[c] + k.dbeta : k.dbeta;
        dgamma[c] = accumulate ? dgamma[c] + k.dgamma : k.dgamma;
        coef[c] = k.a;
        coef[C + c] = k.cb;
        coef[2 * C + c] = k.cc;
    }
}

// FIN: dgamma, dbeta and the three coefficients come from the layer's backward accumulator (the sums the dgrad epilogues / the reduce
// pass added), all C channels once per block through LDS (dynamic: 5 C floats); block 0 writes dgamma / dbeta and returns the layer's
// FORWARD accumulator to zero (fin.zero) -- nobody reads that one any more; its own is zeroed by the layer's next forward pass.
template <typename T, bool FIN>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ y,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ coef, T* __restrict__ dy, const HaloIdx h, int nrows,
                                                           const BnBwdAcc fin) {
    constexpr int EPC = Vec16<T>::N, U = FVA_BN_UNROLL;
    extern __shared__ float fx_tab[];          // FIN: [5][C]
    const int row_chunks = h.Wp * h.cpp;
    const int cmask = h.cpp - 1, cshift = 31 - __builtin_clz(h.cpp);
    // lane-constant channel chunk (cpp divides 256): dY = a*dU + k1*y + k2 with k1 = coefB*rstd, k2 = coefC - k1*mean
    const int cc = threadIdx.x & cmask;
    float sc[EPC], sh[EPC], ka[EPC], k1[EPC], k2[EPC];
    Vec16<T> g[U], v[U];
    bool ok[U];
    int64_t m0 = 0;
    auto set_row = [&](int row) -> bool {       // false: a border row
        const int b = row / h.Hp, yy = row - b * h.Hp - h.pad;
        m0 = ((int64_t)b * h.H + yy) * h.W;
        return yy >= 0 && yy < h.H;
    };
    auto load = [&](int i0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 256;
            const int xx = (i >> cshift) - h.pad;
            ok[u] = i < row_chunks && xx >= 0 && xx < h.W;
            const int64_t off = (m0 + (ok[u] ? xx : 0)) * h.C + cc * EPC;
            g[u] = ld_last<T>(dz + off);
            v[u] = ld_last<T>(y + off);
        }
    };
    // FIN: the first row's operands are requested before the prologue and stay in flight across its (cold) accumulator reads
    bool first = FIN && (int)blockIdx.x < nrows && set_row(blockIdx.x) && (int)threadIdx.x < row_chunks;
    if (first) load(threadIdx.x);
    if constexpr (FIN) {
        const bool writer = blockIdx.x == 0;
        long long* fx_words = (long long*)(fx_tab + 5 * h.C + (h.C & 1));     // replicas > 1: [FX_WORDS][C] behind the table, 8-byte aligned
        if (fin.replicas > 1) fx_gather_lds(fin.acc, h.C, fin.replicas, fx_words, threadIdx.x, 256);
        for (int c = threadIdx.x; c < h.C; c += 256) {
            double s1, s2;
            if (fin.replicas > 1) fx_value2(fx_from_lds(fx_words, h.C, c), s1, s2);
            else fx_load2(fin.acc, h.C, 1, c, s1, s2);
            const BnBwdCoef q = bn_bwd_coef(s1, s2, fin.n, fin.gamma[c], rstd[c]);
            const BnBwdK k = bn_bwd_pack_coef(q.a, shift[c], mean[c], rstd[c], q.cb, q.cc);
            fx_tab[c] = scale[c];
            fx_tab[h.C + c] = k.sh;
            fx_tab[2 * h.C + c] = k.a;
            fx_tab[3 * h.C + c] = k.k1;
            fx_tab[4 * h.C + c] = k.k2;
            if (writer) {
                fin.dbeta[c] = fin.accumulate ? fin.dbeta[c] + q.dbeta : q.dbeta;
                fin.dgamma[c] = fin.accumulate ? fin.dgamma[c] + q.dgamma : q.dgamma;
            }
        }
        if (writer) fx_zero(fin.zero, h.C, fin.replicas, threadIdx.x, 256);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int c = cc * EPC + e;
            sc[e] = fx_tab[c]; sh[e] = fx_tab[h.C + c]; ka[e] = fx_tab[2 * h.C + c]; k1[e] = fx_tab[3 * h.C + c]; k2[e] = fx_tab[4 * h.C + c];
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int c = cc * EPC + e;
            const BnBwdK k = bn_bwd_pack_coef(coef[c], shift[c], mean[c], rstd[c], coef[h.C + c], coef[2 * h.C + c]);
            sc[e] = scale[c];
            sh[e] = k.sh;
            ka[e] = k.a;
            k1[e] = k.k1;
            k2[e] = k.k2;
        }
    }
    // one padded row per block and step
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const bool row_in = set_row(row);
        T* orow = dy + (int64_t)row * row_chunks * EPC;
        if (!row_in) {
            Vec16<T> zero;
#pragma unroll
            for (int e = 0; e < EPC; ++e) zero.set(e, 0.f);
            for (int i = threadIdx.x; i < row_chunks; i += 256) *(Vec16<T>*)(orow + (int64_t)i * EPC) = zero;
            continue;
        }
        for (int i0 = threadIdx.x; i0 < row_chunks; i0 += 256 * U) {
            if (!first) load(i0);
            first = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * 256;
                Vec16<T> out;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    out.set(e, ok[u] ? bn_bwd_apply_elem(g[u].get(e), v[u].get(e), sc[e], sh[e], ka[e], k1[e], k2[e]) : 0.f);
                }
                if (i < row_chunks) *(Vec16<T>*)(orow + (int64_t)i * EPC) = out;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// nearest x2 upsample + channel concat into a halo buffer (pad 1)
template <typename T>
__global__ __launch_bounds__(256) void upcat_fwd_kernel(const T* __restrict__ up, int up_pad, const T* __restrict__ skip,
                                                        int skip_pad, T* __restrict__ out, const HaloIdx h, int hh, int ww,
                                                        int Cup, int Cskip, int up_first) {
    constexpr int EPC = Vec16<T>::N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < h.total; i += (int64_t)gridDim.x * blockDim.x) {
        int b, yy, xx, cc;
        Vec16<T> v;
        if (halo_decode(h, (uint32_t)i, b, yy, xx, cc)) {
            int c = cc * EPC;
            const bool from_up = up_first ? (c < Cup) : (c >= Cskip);
            if (from_up) {
                if (!up_first) c -= Cskip;
                const int64_t pix = ((int64_t)b * (hh + 2 * up_pad) + (yy >> 1) + up_pad) * (ww + 2 * up_pad) + (xx >> 1) + up_pad;
                v = *(const Vec16<T>*)(up + pix * Cup + c);
            } else {
                if (up_first) c -= Cup;
                const int64_t pix = ((int64_t)b * (2 * hh + 2 * skip_pad) + yy + skip_pad) * (2 * ww + 2 * skip_pad) + xx + skip_pad;
                v = *(const Vec16<T>*)(skip + pix * Cskip + c);
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPC; ++e) v.set(e, 0.f);
        }
        *(Vec16<T>*)(out + i * EPC) = v;
    }
}

// dcat dense [B][2h][2w][Ct] -> dup dense [B][h][w][Cup] (sum of the 2x2 block) and dskip dense [B][2h][2w][Cskip]
template <typename T>
__global__ __launch_bounds__(256) void upcat_bwd_kernel(const T* __restrict__ dcat, T* __restrict__ dup, T* __restrict__ dskip,
                                                        int B, int hh, int ww, int Cup, int Cskip, int up_first) {
    constexpr int EPC = Vec16<T>::N;
    const int Ct = Cup + Cskip, cu = Cup / EPC, cs = Cskip / EPC;
    const int64_t n_up = (int64_t)B * hh * ww * cu, n_sk = (int64_t)B * 4 * hh * ww * cs;
    const int uoff = up_first ? 0 : Cskip, soff = up_first ? Cup : 0;
    // 32-bit item arithmetic (the host checks the item count): 64-bit divisions cost more than the 16-byte copy they address
    const uint32_t total = (uint32_t)(n_up + n_sk), nu = (uint32_t)n_up;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < nu) {
            const uint32_t pix = i / (uint32_t)cu, cc = i - pix * (uint32_t)cu;
            const uint32_t row = pix / (uint32_t)ww, x = pix - row * (uint32_t)ww;        // row = b * hh + y
            const uint32_t b = row / (uint32_t)hh, y = row - b * (uint32_t)hh;
            float acc[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
            Vec16<T> v[4];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int64_t sp = ((int64_t)b * 2 * hh + 2 * y + dy) * (2 * ww) + 2 * x + dx;
                    v[dy * 2 + dx] = *(const Vec16<T>*)(dcat + sp * Ct + uoff + cc * EPC);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc[e] += v[q].get(e);
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.set(e, acc[e]);
            *(Vec16<T>*)(dup + (int64_t)pix * Cup + cc * EPC) = o;
        } else {
            const uint32_t j = i - nu;
            const uint32_t pix = j / (uint32_t)cs, cc = j - pix * (uint32_t)cs;
            *(Vec16<T>*)(dskip + (int64_t)pix * Cskip + cc * EPC) = *(const Vec16<T>*)(dcat + (int64_t)pix * Ct + soff + cc * EPC);
        }
    }
}

// arbitrary-stride [B,C,H,W] (fp32 or bf16) -> halo NHWC of T.  One thread per output element (small tensors only).
template <typename T, typename S>
__global__ void pack_nchw_kernel(const S* __restrict__ src, int64_t sb, int64_t sc, int64_t sh, int64_t sw, T* __restrict__ dst,
                                 int pad, int B, int C, int H, int W) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int64_t total = (int64_t)B * Hp * Wp * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int xp = (int)(pix % Wp), yp = (int)((pix / Wp) % Hp), b = (int)(pix / ((int64_t)Wp * Hp));
        const int y = yp - pad, x = xp - pad;
        float v = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) v = to_f(src[b * sb + c * sc + y * sh + x * sw]);
        dst[i] = from_f<T>(v);
    }
}

template <typename TD, typename TS>
__global__ void cast_nhwc_kernel(const TS* __restrict__ src, int spad, TD* __restrict__ dst, int dpad, int B, int H, int W, int C) {
    const int Hd = H + 2 * dpad, Wd = W + 2 * dpad, Hs = H + 2 * spad, Ws = W + 2 * spad;
    const int64_t total = (int64_t)B * Hd * Wd * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int xp = (int)(pix % Wd), yp = (int)((pix / Wd) % Hd), b = (int)(pix / ((int64_t)Wd * Hd));
        const int y = yp - dpad, x = xp - dpad;
        float v = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) v = to_f(src[(((int64_t)b * Hs + y + spad) * Ws + x + spad) * C + c]);
        dst[i] = from_f<TD>(v);
    }
}

constexpr int ACC_GRID = 2048;    // blocks of the accumulator forms of the apply kernels: each walks rows, its prologue paid once (1280 ... one block per row measured alike)

inline int stream_grid(int64_t items) {
    int64_t g = (items + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" {

int fva_bn_finalize(float* part, int32_t nblocks, int32_t partial_rows, int64_t count, int32_t C, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                    float* save_mean, float* save_rstd, float* scale, float* shift, void* stream) {
    if (!part || !gamma || !beta || !save_mean || !save_rstd || !scale || !shift || nblocks <= 0 || count <= 0 || C <= 0)
        return fva_fail(FVA_ERR_ARG, "fva_bn_finalize: bad argument");
    if (partial_rows < fva_bn_partial_rows(nblocks))
        return fva_fail(FVA_ERR_WORKSPACE, "fva_bn_finalize: the table holds %d rows, %d rows of producers need fva_bn_partial_rows() = %d", partial_rows, nblocks,
                        fva_bn_partial_rows(nblocks));
    const double* pre = nullptr;
    int pre_rows = 0;
    if (nblocks >= PRE_MIN) {
        pre_rows = cdiv(nblocks, PRE_ROWS);
        // the table was allocated with fva_bn_partial_rows(nblocks) rows: doubles live behind the nblocks float rows
        double* tail = (double*)(part + ((int64_t)nblocks * 2 * C + 1) / 2 * 2);
        pre = tail;
        hipLaunchKernelGGL(bn_prereduce_kernel, dim3(cdiv(C, 16), pre_rows), dim3(1024), 0, (hipStream_t)stream, (const float*)part, nblocks,
                           C, tail);
        FVA_LAUNCH_CHECK("bn_prereduce_kernel");
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 16)), dim3(1024), 0, (hipStream_t)stream, (const float*)part, nblocks, (double)count,
                       C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, save_mean, save_rstd,
                       scale, shift, pre, pre_rows);
    FVA_LAUNCH_CHECK("bn_finalize_kernel");
    return FVA_OK;
}

int32_t fva_bn_partial_rows(int32_t nblocks) {
    if (nblocks < PRE_MIN) return nblocks;
    return nblocks + 2 * cdiv(nblocks, PRE_ROWS) + 1;   // pre-reduced doubles (2 floats each) + alignment slack
}

int fva_bn_eval_coeffs(int32_t C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                       float* scale, float* shift, void* stream) {
    if (!gamma || !beta || !rm || !rv || !scale || !shift || C <= 0) return fva_fail(FVA_ERR_ARG, "fva_bn_eval_coeffs: bad argument");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, C, gamma, beta, rm, rv, eps,
                       scale, shift);
    FVA_LAUNCH_CHECK("bn_eval_coeffs_kernel");
    return FVA_OK;
}

static int check_chan(int dtype, int C, const char* who) {
    if (dtype != FVA_F32 && dtype != FVA_BF16) return fva_fail(FVA_ERR_ARG, "%s: bad dtype", who);
    const int epc = dtype == FVA_BF16 ? 8 : 4;
    if (C <= 0 || C % epc) return fva_fail(FVA_ERR_ARG, "%s: C=%d not a multiple of %d", who, C, epc);
    return FVA_OK;
}

int fva_bn_silu_apply(int dtype, const void* y, const float* scale, const float* shift, const void* residual, int res_pad,
                      void* z, int z_pad, int B, int H, int W, int C, void* stream) {
    int rc = check_chan(dtype, C, "fva_bn_silu_apply");
    if (rc) return rc;
    if (!y || !scale || !shift || !z) return fva_fail(FVA_ERR_ARG, "fva_bn_silu_apply: null pointer");
    const HaloIdx h = make_halo(B, H, W, C, z_pad, dtype == FVA_BF16 ? 8 : 4);
    if (h.total >= (1ll << 31)) return fva_fail(FVA_ERR_ARG, "fva_bn_silu_apply: tensor too large");
    if ((h.cpp & (h.cpp - 1)) || h.cpp > 256) return fva_fail(FVA_ERR_ARG, "fva_bn_silu_apply: C=%d must be a power of two (<= 256 chunks)", C);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL((bn_silu_apply_kernel<bf16_t, false>), dim3(B * h.Hp), dim3(256), 0, s, (const bf16_t*)y, scale, shift,
                           (const bf16_t*)residual, res_pad, (bf16_t*)z, h, BnFwdAcc(), B * h.Hp);
    else
        hipLaunchKernelGGL((bn_silu_apply_kernel<float, false>), dim3(B * h.Hp), dim3(256), 0, s, (const float*)y, scale, shift,
                           (const float*)residual, res_pad, (float*)z, h, BnFwdAcc(), B * h.Hp);
    FVA_LAUNCH_CHECK("bn_silu_apply_kernel");
    return FVA_OK;
}

static int fill_fwd_acc(const fva_bn_fwd_acc* a, int64_t M, BnFwdAcc& f, const char* who) {
    if (!a || !a->acc || !a->gamma || !a->beta || !a->save_mean || !a->save_rstd || !a->scale || !a->shift)
        return fva_fail(FVA_ERR_ARG, "%s: null pointer in the accumulator descriptor", who);
    if (a->zero == a->acc) return fva_fail(FVA_ERR_ARG, "%s: `zero` must be another accumulator (this one is still being read)", who);
    if (!fva_replicas_ok(a->replicas)) return fva_fail(FVA_ERR_ARG, "%s: replicas = %d is not a power of two in 1..%d", who, a->replicas, FVA_BN_ACC_MAX_REPLICAS);
    f.acc = (long long*)a->acc; f.zero = (long long*)a->zero; f.replicas = a->replicas; f.gamma = a->gamma; f.beta = a->beta;
    f.running_mean = a->running_mean; f.running_var = a->running_var; f.nbt = (long long*)a->num_batches_tracked;
    f.momentum = a->momentum; f.eps = a->eps; f.n = bn_n((double)M);
    f.save_mean = a->save_mean; f.save_rstd = a->save_rstd; f.scale = a->scale; f.shift = a->shift;
    return FVA_OK;
}

int fva_bn_acc_finalize(const fva_bn_fwd_acc* acc, int64_t M, int C, void* stream) {
    BnFwdAcc f = BnFwdAcc();
    const int rc = fill_fwd_acc(acc, M, f, "fva_bn_acc_finalize");
    if (rc) return rc;
    if (C < 1 || M < 1) return fva_fail(FVA_ERR_ARG, "fva_bn_acc_finalize: bad size");
    if (C > 2048) return fva_fail(FVA_ERR_ARG, "fva_bn_acc_finalize: C = %d > 2048", C);
    hipLaunchKernelGGL(bn_acc_finalize_kernel, dim3(1), dim3(256), FX_WORDS * C * 8, (hipStream_t)stream, f, C);
    FVA_LAUNCH_CHECK("bn_acc_finalize_kernel");
    return FVA_OK;
}

int fva_bn_silu_apply_acc(int dtype, const void* y, const fva_bn_fwd_acc* acc, const void* residual, int res_pad, void* z, int z_pad, int B,
                          int H, int W, int C, void* stream) {
    int rc = check_chan(dtype, C, "fva_bn_silu_apply_acc");
    if (rc) return rc;
    if (!y || !z) return fva_fail(FVA_ERR_ARG, "fva_bn_silu_apply_acc: null pointer");
    BnFwdAcc f = BnFwdAcc();
    rc = fill_fwd_acc(acc, (int64_t)B * H * W, f, "fva_bn_silu_apply_acc");
    if (rc) return rc;
    const HaloIdx h = make_halo(B, H, W, C, z_pad, dtype == FVA_BF16 ? 8 : 4);
    if (h.total >= (1ll << 31)) return fva_fail(FVA_ERR_ARG, "fva_bn_silu_apply_acc: tensor too large");
    if ((h.cpp & (h.cpp - 1)) || h.cpp > 256) return fva_fail(FVA_ERR_ARG, "fva_bn_silu_apply_acc: C=%d must be a power of two (<= 256 chunks)", C);
    hipStream_t s = (hipStream_t)stream;
    const int nrows = B * h.Hp, grid = nrows < ACC_GRID ? nrows : ACC_GRID, smem = 2 * C * 4 + (f.replicas > 1 ? FX_WORDS * C * 8 : 0);
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL((bn_silu_apply_kernel<bf16_t, true>), dim3(grid), dim3(256), smem, s, (const bf16_t*)y, nullptr, nullptr,
                           (const bf16_t*)residual, res_pad, (bf16_t*)z, h, f, nrows);
    else
        hipLaunchKernelGGL((bn_silu_apply_kernel<float, true>), dim3(grid), dim3(256), smem, s, (const float*)y, nullptr, nullptr,
                           (const float*)residual, res_pad, (float*)z, h, f, nrows);
    FVA_LAUNCH_CHECK("bn_silu_apply_kernel<acc>");
    return FVA_OK;
}

static int bwd_rows_per_block(int64_t M, int C, int epc) {
    const int rpi = 256 / (C / epc) > 0 ? 256 / (C / epc) : 1;
    int64_t rows = (M + 2047) / 2048;           // at most 2048 blocks
    if (rows < (int64_t)rpi * 8) rows = (int64_t)rpi * 8;
    rows = (rows + rpi - 1) / rpi * rpi;
    return (int)rows;
}

int32_t fva_bn_bwd_blocks(int dtype, int64_t M, int C) {
    const int epc = dtype == FVA_BF16 ? 8 : 4;
    if (C % epc || C / epc > 256 || 256 % (C / epc)) return 0;
    return cdiv(M, bwd_rows_per_block(M, C, epc));
}

static int bwd_reduce_impl(const char* who, int dtype, const void* dz, const void* y, const float* scale, const float* shift,
                           const float* save_mean, const float* save_rstd, float* partial, int64_t* acc, int replicas, int32_t nblocks, int64_t M, int C,
                           void* stream) {
    int rc = check_chan(dtype, C, who);
    if (acc && !fva_replicas_ok(replicas)) return fva_fail(FVA_ERR_ARG, "%s: replicas = %d is not a power of two in 1..%d", who, replicas, FVA_BN_ACC_MAX_REPLICAS);
    if (rc) return rc;
    const int epc = dtype == FVA_BF16 ? 8 : 4;
    const int cpp = C / epc;
    if (cpp > 256 || 256 % cpp) return fva_fail(FVA_ERR_ARG, "%s: C=%d unsupported", who, C);
    if (!dz || !y || !scale || !shift || !save_mean || !save_rstd || (!partial && !acc)) return fva_fail(FVA_ERR_ARG, "%s: null pointer", who);
    const int rows = bwd_rows_per_block(M, C, epc);
    if (acc) nblocks = cdiv(M, rows);
    if (nblocks != cdiv(M, rows)) return fva_fail(FVA_ERR_ARG, "%s: nblocks %d != %d", who, nblocks, cdiv(M, rows));
    const int smem = 2 * (256 / cpp) * C * 4;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(nblocks), dim3(256), smem, s, (const bf16_t*)dz, (const bf16_t*)y, scale,
                           shift, save_mean, save_rstd, partial, (long long*)acc, replicas, M, C, rows);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(nblocks), dim3(256), smem, s, (const float*)dz, (const float*)y, scale,
                           shift, save_mean, save_rstd, partial, (long long*)acc, replicas, M, C, rows);
    FVA_LAUNCH_CHECK("bn_bwd_reduce_kernel");
    return FVA_OK;
}

int fva_bn_silu_bwd_reduce(int dtype, const void* dz, const void* y, const float* scale, const float* shift,
                           const float* save_mean, const float* save_rstd, float* partial, int32_t nblocks, int64_t M, int C,
                           void* stream) {
    if (!partial) return fva_fail(FVA_ERR_ARG, "fva_bn_silu_bwd_reduce: null pointer");
    return bwd_reduce_impl("fva_bn_silu_bwd_reduce", dtype, dz, y, scale, shift, save_mean, save_rstd, partial, nullptr, 1, nblocks, M, C, stream);
}

int fva_bn_silu_bwd_reduce_acc(int dtype, const void* dz, const void* y, const float* scale, const float* shift,
                               const float* save_mean, const float* save_rstd, int64_t* acc, int32_t replicas, int64_t M, int C, void* stream) {
    if (!acc) return fva_fail(FVA_ERR_ARG, "fva_bn_silu_bwd_reduce_acc: null pointer");
    return bwd_reduce_impl("fva_bn_silu_bwd_reduce_acc", dtype, dz, y, scale, shift, save_mean, save_rstd, nullptr, acc, replicas, 0, M, C, stream);
}

int fva_bn_bwd_finalize(float* partial, int32_t nblocks, int32_t partial_rows, int64_t M, int C, const float* gamma, const float* save_rstd,
                        float* dgamma, float* dbeta, int accumulate, float* coef, void* stream) {
    if (!partial || !gamma || !save_rstd || !dgamma || !dbeta || !coef || nblocks <= 0 || M <= 0 || C <= 0)
        return fva_fail(FVA_ERR_ARG, "fva_bn_bwd_finalize: bad argument");
    if (partial_rows < fva_bn_partial_rows(nblocks))
        return fva_fail(FVA_ERR_WORKSPACE, "fva_bn_bwd_finalize: the table holds %d rows, %d rows of producers need fva_bn_partial_rows() = %d", partial_rows,
                        nblocks, fva_bn_partial_rows(nblocks));
    // long tables (the fused dgrad epilogues write one row per 128- or 256-pixel block) are first folded in parallel, as in the
    // forward pass: C / 16 blocks walking 1600 rows of a C = 128 layer took 72 us, the two-level form takes 5 + 5 (the caller
    // allocated fva_bn_partial_rows(nblocks) rows: the doubles live behind the table)
    const double* pre = nullptr;
    int pre_rows = 0;
    if (nblocks >= PRE_MIN) {
        pre_rows = cdiv(nblocks, PRE_ROWS);
        double* tail = (double*)(partial + ((int64_t)nblocks * 2 * C + 1) / 2 * 2);
        pre = tail;
        hipLaunchKernelGGL(bn_prereduce_kernel, dim3(cdiv(C, 16), pre_rows), dim3(1024), 0, (hipStream_t)stream, partial, nblocks, C, tail);
        FVA_LAUNCH_CHECK("bn_prereduce_kernel");
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 16)), dim3(1024), 0, (hipStream_t)stream, partial, nblocks, (double)M, C,
                       gamma, save_rstd, dgamma, dbeta, accumulate, coef, pre, pre_rows);
    FVA_LAUNCH_CHECK("bn_bwd_finalize_kernel");
    return FVA_OK;
}

static int bwd_apply_impl(const char* who, int dtype, const void* dz, const void* y, const float* scale, const float* shift, const float* save_mean,
                          const float* save_rstd, const float* coef, const fva_bn_bwd_acc* a, void* dy, int dy_pad, int B, int H, int W, int C,
                          void* stream) {
    int rc = check_chan(dtype, C, who);
    if (rc) return rc;
    if (!dz || !y || !scale || !shift || !save_mean || !save_rstd || (!coef && !a) || !dy) return fva_fail(FVA_ERR_ARG, "%s: null pointer", who);
    const HaloIdx h = make_halo(B, H, W, C, dy_pad, dtype == FVA_BF16 ? 8 : 4);
    if (h.total >= (1ll << 31)) return fva_fail(FVA_ERR_ARG, "%s: tensor too large", who);
    if ((h.cpp & (h.cpp - 1)) || h.cpp > 256) return fva_fail(FVA_ERR_ARG, "%s: C=%d must be a power of two (<= 256 chunks)", who, C);
    hipStream_t s = (hipStream_t)stream;
    const int nrows = B * h.Hp;
    if (a) {
        if (!a->acc || !a->gamma || !a->dgamma || !a->dbeta) return fva_fail(FVA_ERR_ARG, "%s: null pointer in the accumulator descriptor", who);
        if (a->zero == a->acc) return fva_fail(FVA_ERR_ARG, "%s: `zero` must be another accumulator (this one is still being read)", who);
        BnBwdAcc f = BnBwdAcc();
        if (!fva_replicas_ok(a->replicas)) return fva_fail(FVA_ERR_ARG, "%s: replicas = %d is not a power of two in 1..%d", who, a->replicas, FVA_BN_ACC_MAX_REPLICAS);
        f.acc = (long long*)a->acc; f.zero = (long long*)a->zero; f.replicas = a->replicas; f.gamma = a->gamma; f.dgamma = a->dgamma; f.dbeta = a->dbeta;
        f.accumulate = a->accumulate; f.n = bn_n((double)B * H * W);
        const int grid = nrows < ACC_GRID ? nrows : ACC_GRID, smem = (5 * C + (C & 1)) * 4 + (f.replicas > 1 ? FX_WORDS * C * 8 : 0);
        if (dtype == FVA_BF16)
            hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, true>), dim3(grid), dim3(256), smem, s, (const bf16_t*)dz, (const bf16_t*)y,
                               scale, shift, save_mean, save_rstd, nullptr, (bf16_t*)dy, h, nrows, f);
        else
            hipLaunchKernelGGL((bn_bwd_apply_kernel<float, true>), dim3(grid), dim3(256), smem, s, (const float*)dz, (const float*)y,
                               scale, shift, save_mean, save_rstd, nullptr, (float*)dy, h, nrows, f);
        FVA_LAUNCH_CHECK("bn_bwd_apply_kernel<acc>");
        return FVA_OK;
    }
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, false>), dim3(nrows), dim3(256), 0, s, (const bf16_t*)dz, (const bf16_t*)y,
                           scale, shift, save_mean, save_rstd, coef, (bf16_t*)dy, h, nrows, BnBwdAcc());
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<float, false>), dim3(nrows), dim3(256), 0, s, (const float*)dz, (const float*)y,
                           scale, shift, save_mean, save_rstd, coef, (float*)dy, h, nrows, BnBwdAcc());
    FVA_LAUNCH_CHECK("bn_bwd_apply_kernel");
    return FVA_OK;
}

int fva_bn_silu_bwd_apply(int dtype, const void* dz, const void* y, const float* scale, const float* shift, const float* save_mean,
                          const float* save_rstd, const float* coef, void* dy, int dy_pad, int B, int H, int W, int C, void* stream) {
    if (!coef) return fva_fail(FVA_ERR_ARG, "fva_bn_silu_bwd_apply: null pointer");
    return bwd_apply_impl("fva_bn_silu_bwd_apply", dtype, dz, y, scale, shift, save_mean, save_rstd, coef, nullptr, dy, dy_pad, B, H, W, C, stream);
}

int fva_bn_silu_bwd_apply_acc(int dtype, const void* dz, const void* y, const float* scale, const float* shift, const float* save_mean,
                              const float* save_rstd, const fva_bn_bwd_acc* acc, void* dy, int dy_pad, int B, int H, int W, int C, void* stream) {
    if (!acc) return fva_fail(FVA_ERR_ARG, "fva_bn_silu_bwd_apply_acc: null pointer");
    return bwd_apply_impl("fva_bn_silu_bwd_apply_acc", dtype, dz, y, scale, shift, save_mean, save_rstd, nullptr, acc, dy, dy_pad, B, H, W, C, stream);
}

int fva_upsample2_concat_fwd(int dtype, const void* up, int up_pad, const void* skip, int skip_pad, void* out, int B, int h, int w,
                             int Cup, int Cskip, int up_first, void* stream) {
    int rc = check_chan(dtype, Cup, "fva_upsample2_concat_fwd");
    if (!rc) rc = check_chan(dtype, Cskip, "fva_upsample2_concat_fwd");
    if (rc) return rc;
    if (!up || !skip || !out) return fva_fail(FVA_ERR_ARG, "fva_upsample2_concat_fwd: null pointer");
    const HaloIdx hi = make_halo(B, 2 * h, 2 * w, Cup + Cskip, 1, dtype == FVA_BF16 ? 8 : 4);
    if (hi.total >= (1ll << 31)) return fva_fail(FVA_ERR_ARG, "fva_upsample2_concat_fwd: tensor too large");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL(upcat_fwd_kernel<bf16_t>, dim3(stream_grid(hi.total)), dim3(256), 0, s, (const bf16_t*)up, up_pad,
                           (const bf16_t*)skip, skip_pad, (bf16_t*)out, hi, h, w, Cup, Cskip, up_first);
    else
        hipLaunchKernelGGL(upcat_fwd_kernel<float>, dim3(stream_grid(hi.total)), dim3(256), 0, s, (const float*)up, up_pad,
                           (const float*)skip, skip_pad, (float*)out, hi, h, w, Cup, Cskip, up_first);
    FVA_LAUNCH_CHECK("upcat_fwd_kernel");
    return FVA_OK;
}

int fva_upsample2_concat_bwd(int dtype, const void* dcat, void* dup, void* dskip, int B, int h, int w, int Cup, int Cskip,
                             int up_first, void* stream) {
    int rc = check_chan(dtype, Cup, "fva_upsample2_concat_bwd");
    if (!rc) rc = check_chan(dtype, Cskip, "fva_upsample2_concat_bwd");
    if (rc) return rc;
    if (!dcat || !dup || !dskip) return fva_fail(FVA_ERR_ARG, "fva_upsample2_concat_bwd: null pointer");
    const int epc = dtype == FVA_BF16 ? 8 : 4;
    const int64_t items = (int64_t)B * h * w * (Cup / epc) + (int64_t)B * 4 * h * w * (Cskip / epc);
    if (items >= (1ll << 31)) return fva_fail(FVA_ERR_ARG, "fva_upsample2_concat_bwd: tensor too large");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL(upcat_bwd_kernel<bf16_t>, dim3(stream_grid(items)), dim3(256), 0, s, (const bf16_t*)dcat, (bf16_t*)dup,
                           (bf16_t*)dskip, B, h, w, Cup, Cskip, up_first);
    else
        hipLaunchKernelGGL(upcat_bwd_kernel<float>, dim3(stream_grid(items)), dim3(256), 0, s, (const float*)dcat, (float*)dup,
                           (float*)dskip, B, h, w, Cup, Cskip, up_first);
    FVA_LAUNCH_CHECK("upcat_bwd_kernel");
    return FVA_OK;
}

int fva_pack_nchw(int dtype, const void* src, int src_is_bf16, int64_t sb, int64_t sc, int64_t sh, int64_t sw, void* dst,
                  int dst_pad, int B, int C, int H, int W, void* stream) {
    if (!src || !dst || (dtype != FVA_F32 && dtype != FVA_BF16)) return fva_fail(FVA_ERR_ARG, "fva_pack_nchw: bad argument");
    const int64_t total = (int64_t)B * (H + 2 * dst_pad) * (W + 2 * dst_pad) * C;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(stream_grid(total)), b(256);
    if (dtype == FVA_BF16 && src_is_bf16)
        hipLaunchKernelGGL((pack_nchw_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)src, sb, sc, sh, sw, (bf16_t*)dst, dst_pad, B, C, H, W);
    else if (dtype == FVA_BF16)
        hipLaunchKernelGGL((pack_nchw_kernel<bf16_t, float>), g, b, 0, s, (const float*)src, sb, sc, sh, sw, (bf16_t*)dst, dst_pad, B, C, H, W);
    else if (src_is_bf16)
        hipLaunchKernelGGL((pack_nchw_kernel<float, bf16_t>), g, b, 0, s, (const bf16_t*)src, sb, sc, sh, sw, (float*)dst, dst_pad, B, C, H, W);
    else
        hipLaunchKernelGGL((pack_nchw_kernel<float, float>), g, b, 0, s, (const float*)src, sb, sc, sh, sw, (float*)dst, dst_pad, B, C, H, W);
    FVA_LAUNCH_CHECK("pack_nchw_kernel");
    return FVA_OK;
}

int fva_cast_nhwc(const void* src, int src_dtype, int src_pad, void* dst, int dst_dtype, int dst_pad, int B, int H, int W, int C,
                  void* stream) {
    if (!src || !dst) return fva_fail(FVA_ERR_ARG, "fva_cast_nhwc: null pointer");
    const int64_t total = (int64_t)B * (H + 2 * dst_pad) * (W + 2 * dst_pad) * C;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(stream_grid(total)), b(256);
    if (dst_dtype == FVA_BF16 && src_dtype == FVA_BF16)
        hipLaunchKernelGGL((cast_nhwc_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)src, src_pad, (bf16_t*)dst, dst_pad, B, H, W, C);
    else if (dst_dtype == FVA_BF16)
        hipLaunchKernelGGL((cast_nhwc_kernel<bf16_t, float>), g, b, 0, s, (const float*)src, src_pad, (bf16_t*)dst, dst_pad, B, H, W, C);
    else if (src_dtype == FVA_BF16)
        hipLaunchKernelGGL((cast_nhwc_kernel<float, bf16_t>), g, b, 0, s, (const bf16_t*)src, src_pad, (float*)dst, dst_pad, B, H, W, C);
    else
        hipLaunchKernelGGL((cast_nhwc_kernel<float, float>), g, b, 0, s, (const float*)src, src_pad, (float*)dst, dst_pad, B, H, W, C);
    FVA_LAUNCH_CHECK("cast_nhwc_kernel");
    return FVA_OK;
}

}  // extern "C"
