// Stem convolution (3x3, stride 1, pad 1, Cin <= 3, Cout == 32) read straight from the caller's fp32 NCHW
// images: 27-deep dot products are too shallow for MFMA, so this is a direct VALU kernel, one output pixel
// (all 32 channels) per lane.  HBM-bound: reads 12 B and writes 64 B (bf16) per pixel.
//
// Replaces `conv0` of Darknet-53 (reference classfication/models/darknet53.py:73) forward and weight gradient
// (the input image needs no gradient).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int CO = 32;      // output channels handled
constexpr int MAXJ = 36;    // Cin*9 <= 36

template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                       T* __restrict__ y, float* __restrict__ stats, int B, int Cin, int H,
                                                       int W, int64_t M) {
    __shared__ float sw[MAXJ * CO];
    __shared__ float tile[256 * 33];
    const int tid = threadIdx.x;
    const int J = Cin * 9;
    for (int i = tid; i < J * CO; i += 256) {
        const int co = i % CO, j = i / CO;  // sw[j][co] = w[co][ci][kh][kw], j = ci*9 + kh*3 + kw
        sw[j * CO + co] = w[co * J + j];
    }
    __syncthreads();
    const int64_t m = (int64_t)blockIdx.x * 256 + tid;
    const bool live = m < M;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    if (live) {
        const int x = (int)(m % W), yy = (int)((m / W) % H), b = (int)(m / ((int64_t)W * H));
#pragma unroll 1
        for (int ci = 0; ci < Cin; ++ci) {
            const float* plane = img + ((int64_t)b * Cin + ci) * H * W;
#pragma unroll 1
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int iy = yy + kh - 1, ix = x + kw - 1;
                    const float v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? plane[(int64_t)iy * W + ix] : 0.f;
                    const float4* wr = (const float4*)&sw[(ci * 9 + kh * 3 + kw) * CO];
#pragma unroll
                    for (int q = 0; q < CO / 4; ++q) {
                        const float4 ww = wr[q];
                        acc[4 * q + 0] += v * ww.x;
                        acc[4 * q + 1] += v * ww.y;
                        acc[4 * q + 2] += v * ww.z;
                        acc[4 * q + 3] += v * ww.w;
                    }
                }
        }
        constexpr int EPC = Vec16<T>::N;
#pragma unroll
        for (int q = 0; q < CO / EPC; ++q) {
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.set(e, acc[q * EPC + e]);
            *(Vec16<T>*)(y + m * CO + q * EPC) = o;
        }
    }
    if (stats == nullptr) return;
    // BatchNorm partials of the STORED values: [blk][2][32]
#pragma unroll
    for (int c = 0; c < CO; ++c) tile[tid * 33 + c] = live ? to_f(from_f<T>(acc[c])) : 0.f;
    __syncthreads();
    const int c = tid & 31, grp = tid >> 5;
    float s1 = 0.f, s2 = 0.f;
    for (int p = grp; p < 256; p += 8) {
        const float v = tile[p * 33 + c];
        s1 += v;
        s2 += v * v;
    }
    __syncthreads();
    tile[grp * 64 + c] = s1;
    tile[grp * 64 + 32 + c] = s2;
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) s += tile[g * 64 + tid];
        stats[((int64_t)blockIdx.x * 2 + (tid >> 5)) * CO + (tid & 31)] = s;
    }
}

// dW partial per block over a strided set of 128-pixel tiles: slab[blk][co][j].  Per tile, lanes 0-127 stage the dY
// rows and lanes 128-255 the 3x3 input patches in LDS (fp32); then each of 4 pixel groups x 56 lanes accumulates a
// 4 (co) x 4 (j) register block over its 32 pixels with two 16-byte LDS reads per 16 FMAs.
template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ img, const T* __restrict__ dy,
                                                         float* __restrict__ slab, int B, int Cin, int H, int W, int64_t M,
                                                         int ntiles) {
    constexpr int DS = 36, PS = 40;  // LDS row strides in floats (16-byte aligned, padded against bank conflicts)
    __shared__ __attribute__((aligned(16))) float sdy[128 * DS];
    __shared__ __attribute__((aligned(16))) float spt[128 * PS];
    const int tid = threadIdx.x;
    const int J = Cin * 9;
    const int grp = tid >> 6, t = tid & 63;
    const int co4 = t & 7, j4 = t >> 3;       // 8 x 9 register blocks cover 32 co x 36 j; lanes with j4*4 >= J idle
    const bool worker = j4 * 4 < J;
    const int px = tid & 127, role = tid >> 7;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = 0.f;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int64_t m = (int64_t)tl * 128 + px;
        const bool live = m < M;
        constexpr int EPC = Vec16<T>::N;
        if (role == 0) {
#pragma unroll
            for (int q = 0; q < CO / EPC; ++q) {
                Vec16<T> v;
                if (live) v = *(const Vec16<T>*)(dy + m * CO + q * EPC);
#pragma unroll
                for (int e = 0; e < EPC; ++e) sdy[px * DS + q * EPC + e] = live ? v.get(e) : 0.f;
            }
        } else {
            const int x = (int)(m % W), yy = (int)((m / W) % H), b = (int)(m / ((int64_t)W * H));
            for (int ci = 0; ci < 4; ++ci) {
                const float* plane = img + ((int64_t)b * Cin + ci) * H * W;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int iy = yy + kh - 1, ix = x + kw - 1;
                        spt[px * PS + ci * 9 + kh * 3 + kw] =
                            (live && ci < Cin && iy >= 0 && iy < H && ix >= 0 && ix < W) ? plane[(int64_t)iy * W + ix] : 0.f;
                    }
            }
        }
        __syncthreads();
        if (worker) {
            const float* pd = sdy + grp * 32 * DS + co4 * 4;
            const float* pp = spt + grp * 32 * PS + j4 * 4;
#pragma unroll 4
            for (int p = 0; p < 32; ++p) {
                const float4 g = *(const float4*)(pd + p * DS);
                const float4 v = *(const float4*)(pp + p * PS);
                const float gv[4] = {g.x, g.y, g.z, g.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[a][c] += gv[a] * vv[c];
            }
        }
        __syncthreads();
    }
    // combine the 4 pixel groups in fixed order, then one partial per block
    float* red = sdy;  // 4 x 32 x 36 floats = 18 KiB (fits the dY staging area)
    if (worker) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) red[(grp * CO + co4 * 4 + a) * 36 + j4 * 4 + c] = acc[a][c];
    }
    __syncthreads();
    for (int e = tid; e < CO * J; e += 256) {
        const int co = e / J, j = e - co * J;
        const float v = (red[(0 * CO + co) * 36 + j] + red[(1 * CO + co) * 36 + j]) + (red[(2 * CO + co) * 36 + j] + red[(3 * CO + co) * 36 + j]);
        slab[(int64_t)blockIdx.x * CO * J + e] = v;
    }
}

// one block per weight element: 256 lanes share the per-block partials, fixed-order tree in LDS (deterministic)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int n,
                                                                int nblocks, int accumulate) {
    __shared__ float red[256];
    const int i = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += slab[(int64_t)b * n + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) dw[i] = accumulate ? dw[i] + red[0] : red[0];
}

// ---- MFMA forward (bf16) ---------------------------------------------------------------------------------------------
// conv0 as D[co][pix] = sum_k W[co][k] * P[k][pix] on v_mfma_f32_16x16x32_bf16, K = 36 (3 rows x 3 columns x 4 channels,
// the 4th channel zero) padded to 64.  The images are first packed to a zero-bordered bf16 NHWC4 buffer (8 bytes per pixel:
// stem_pack_kernel), so that lane (pixel r, k-group g) gets its eight consecutive k -- two horizontally adjacent taps of four
// channels -- by two 8-byte loads straight from global memory / L1 (no LDS: every patch byte is used by one wave only).
// A wave walks TILES_PER_WAVE tiles of 16 consecutive pixels of one image row.  Per tile: 3 loads, 4 MFMAs (two 16-channel
// halves x two k-steps), two 8-byte stores per lane; BatchNorm partial sums stay in registers until the end of the block.
constexpr int STEM_TPW = 16;                       // tiles per wave
constexpr int STEM_BLOCK_PIX = 4 * STEM_TPW * 16;  // 1024 pixels per block

__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ img, uint2* __restrict__ out, int B, int Cin, int H,
                                                        int W) {
    const int Wp = W + 2, Hp = H + 2;
    const int64_t total = (int64_t)B * Hp * Wp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xp = (int)(i % Wp), yp = (int)((i / Wp) % Hp), b = (int)(i / ((int64_t)Wp * Hp));
        const int x = xp - 1, y = yp - 1;
        bf16x4 v = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        if (x >= 0 && x < W && y >= 0 && y < H) {
            const float* p = img + ((int64_t)b * Cin * H + y) * W + x;
            for (int c = 0; c < Cin; ++c) v[c] = (bf16_t)p[(int64_t)c * H * W];
        }
        out[i] = __builtin_bit_cast(uint2, v);
    }
}

// A-operand fragments of the 32 x 27 filter bank for the 16x16x32 MFMA: A[row][k = 32s + 8g + j], k = kh*12 + kw*4 + ci (K padded
// 36 -> 64 with zeros).  The bank is staged through LDS by the whole block first: 64 dependent scalar loads per wave cost more
// than the tile loop they precede.  CONTIG: MFMA row r of half c is channel 8(r/4) + 4c + r%4 (a lane's eight accumulators are
// then eight contiguous channels) instead of 16c + r.
template <bool CONTIG>
__device__ inline void stem_weight_fragments(const float* __restrict__ w, int Cin, float* wsh, bf16x8 (&wa)[2][2]) {
    const int tid = threadIdx.x, r = tid & 15, g = (tid & 63) >> 4;
    for (int i = tid; i < CO * Cin * 9; i += 256) wsh[i] = w[i];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 32 * s + 8 * g + j;
                const int kh = k / 12, kw = (k % 12) / 4, ci = k & 3;
                const int ch = CONTIG ? 8 * (r >> 2) + 4 * c + (r & 3) : 16 * c + r;
                const bool live = k < 36 && ci < Cin;
                const float v = wsh[live ? ((ch * Cin + ci) * 3 + kh) * 3 + kw : 0];
                wa[c][s][j] = (bf16_t)(live ? v : 0.f);
            }
}

__global__ __launch_bounds__(256) void stem_fwd_mfma_kernel(const uint2* __restrict__ img4, const float* __restrict__ w,
                                                            bf16_t* __restrict__ y, float* __restrict__ stats, int Cin, int H, int W,
                                                            int64_t M) {
    __shared__ float red[2][4][32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int Wp = W + 2;
    __shared__ float wsh[CO * 27];
    bf16x8 wa[2][2];
    stem_weight_fragments<false>(w, Cin, wsh, wa);
    // patch fragment: taps t = 2g, 2g + 1 (k-step 0) and t = 8 for g == 0 (k-step 1); tap t sits at (t / 3, t % 3)
    const int ta = 2 * g, tb = 2 * g + 1;
    const int offa = (ta / 3) * Wp + ta % 3, offb = (tb / 3) * Wp + tb % 3, offc = 2 * Wp + 2;
    float s1[2][4], s2[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) s1[c][j] = s2[c][j] = 0.f;

    const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wv) * STEM_TPW;
    const int tiles_per_row = W / 16;
#pragma unroll 2
    for (int t = 0; t < STEM_TPW; ++t) {
        const int64_t tile = tile0 + t;
        const int64_t m0 = tile * 16;
        if (m0 >= M) break;
        const int64_t row = tile / tiles_per_row;            // b * H + y
        const int x0 = (int)(tile - row * tiles_per_row) * 16;
        const int b = (int)(row / H), yy = (int)(row - (int64_t)b * H);
        const uint2* base = img4 + ((int64_t)b * (H + 2) + yy) * Wp + x0 + r;   // halo pixel of tap (0, 0)
        const uint2 la = base[offa], lb = base[offb], lc = base[offc];
        const bf16x8 p0 = __builtin_bit_cast(bf16x8, make_uint4(la.x, la.y, lb.x, lb.y));
        const bf16x8 p1 = __builtin_bit_cast(bf16x8, g == 0 ? make_uint4(lc.x, lc.y, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u));
        f32x4 acc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c][0], p0, acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c][1], p1, acc[c], 0, 0, 0);
        }
        // D rows = channels 16c + 4g + j, column = pixel r: four consecutive channels per lane -> one 8-byte store
        bf16_t* yo = y + (m0 + r) * 32 + 4 * g;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (bf16_t)acc[c][j];
                s1[c][j] += acc[c][j];
                s2[c][j] += acc[c][j] * acc[c][j];
            }
            *(bf16x4*)(yo + 16 * c) = o;
        }
    }
    if (stats == nullptr) return;
    // per-channel sums over the wave's pixels (lanes that share g), then over the four waves, one row per block
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s1[c][j] += __shfl_xor(s1[c][j], o);
                s2[c][j] += __shfl_xor(s2[c][j], o);
            }
            if (r == 0) {
                red[0][wv][16 * c + 4 * g + j] = s1[c][j];
                red[1][wv][16 * c + 4 * g + j] = s2[c][j];
            }
        }
    __syncthreads();
    if (tid < 64) {
        const int which = tid >> 5, ch = tid & 31;
        stats[((int64_t)blockIdx.x * 2 + which) * 32 + ch] = (red[which][0][ch] + red[which][1][ch]) + (red[which][2][ch] + red[which][3][ch]);
    }
}

// ---- conv0 recomputed inside its own BatchNorm / SiLU passes (bf16 training) -----------------------------------------------
// conv0 costs ~0.05 ms of MFMA time, its 32-channel 640x640 output 839 MB per trip through HBM.  So the pre-BN output is never
// stored: each of the four passes that need it recomputes it from the 105 MB NHWC4 image copy, tile by tile, in registers.
//   STATS   forward pass 1: per-block partial sums of y, y^2                        (then fva_bn_finalize)
//   APPLY   forward pass 2: z = SiLU(y * scale + shift) into the halo buffer the next conv reads
//   REDUCE  backward pass 1: partial sums of dU, dU * xhat (dU = dz * SiLU'(u))     (then fva_bn_bwd_finalize)
//   DGRAD   backward pass 2: dY = a * dU + k1 * y + k2 into the halo buffer the weight gradient reads
// HBM per pixel: 8 B (image) [+ 64 B dz] [+ 64 B z / dY] instead of an extra 64 B for y in every pass.
enum { STEM_STATS = 0, STEM_APPLY = 1, STEM_REDUCE = 2, STEM_DGRAD = 3 };

struct StemFusedParams {
    const uint2* img4;
    const float* w;
    const bf16_t* dz;      // dense [M][32]                               (REDUCE, DGRAD)
    const float *scale, *shift, *mean, *rstd, *coef;
    bf16_t* out;           // halo [B][H+2][W+2][32], interior written    (APPLY, DGRAD)
    float* part;           // [blocks][2][32]                             (STATS, REDUCE)
    int Cin, H, W;
    int64_t M;
};

template <int MODE>
__global__ __launch_bounds__(256) void stem_fused_kernel(const StemFusedParams p) {
    __shared__ float red[2][4][32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int H = p.H, W = p.W, Wp = W + 2;
    __shared__ float wsh[CO * 27];
    bf16x8 wa[2][2];
    stem_weight_fragments<true>(p.w, p.Cin, wsh, wa);
    const int ta = 2 * g, tb = 2 * g + 1;
    const int offa = (ta / 3) * Wp + ta % 3, offb = (tb / 3) * Wp + tb % 3, offc = 2 * Wp + 2;
    // this lane's eight accumulators are the contiguous channels 8g + 4c + j: one 16-byte access per pixel
    float sc[2][4], sh[2][4], mu[2][4], rs[2][4], ka[2][4], k1[2][4], k2[2][4], s1[2][4], s2[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = 8 * g + 4 * c + j;
            s1[c][j] = s2[c][j] = 0.f;
            if constexpr (MODE != STEM_STATS) {
                sc[c][j] = p.scale[ch];
                sh[c][j] = p.shift[ch];
            }
            if constexpr (MODE == STEM_REDUCE || MODE == STEM_DGRAD) {
                mu[c][j] = p.mean[ch];
                rs[c][j] = p.rstd[ch];
            }
            if constexpr (MODE == STEM_DGRAD) {
                ka[c][j] = p.coef[ch];
                k1[c][j] = p.coef[32 + ch] * rs[c][j];
                k2[c][j] = p.coef[64 + ch] - k1[c][j] * mu[c][j];
            }
        }

    // persistent waves: chunks of STEM_TPW consecutive tiles, strided over the grid.  A chunk's first tile is decoded once, the
    // rest step along the row.  The pass is bound by load latency, not by bytes or MFMA time, so a wave issues the operands of
    // STEM_NB tiles back to back before it computes any of them.
    constexpr int STEM_NB = MODE == STEM_REDUCE ? 2 : 4;                         // REDUCE is VALU-bound as well: it keeps the registers for a fourth wave
    const int tiles_per_row = W / 16;
    const int nchunks = (int)((p.M / 16 + STEM_TPW - 1) / STEM_TPW);
    const int wvu = __builtin_amdgcn_readfirstlane(wv);                         // wave-uniform: tile positions live in SGPRs
    for (int chunk = blockIdx.x * 4 + wvu; chunk < nchunks; chunk += gridDim.x * 4) {
    const int tile0 = chunk * STEM_TPW;
    int64_t m0 = (int64_t)tile0 * 16;
    const int row0 = tile0 / tiles_per_row;
    int x0 = (tile0 - row0 * tiles_per_row) * 16;
    const int b0 = row0 / H;
    int yy = row0 - b0 * H;
    int64_t hpix = ((int64_t)b0 * (H + 2) + yy) * Wp + x0;                      // halo pixel of tap (0, 0) of the tile's first pixel
    for (int bt = 0; bt < STEM_TPW / STEM_NB; ++bt) {
    uint2 la[STEM_NB], lb[STEM_NB], lc[STEM_NB];
    bf16x8 gz[STEM_NB];
    int64_t hp[STEM_NB];
    bool ok[STEM_NB];
#pragma unroll
    for (int i = 0; i < STEM_NB; ++i) {
        ok[i] = m0 < p.M;
        hp[i] = hpix + r;
        if (ok[i]) {
            const uint2* base = p.img4 + hp[i];
            la[i] = base[offa];
            lb[i] = base[offb];
            lc[i] = base[offc];
            if constexpr (MODE == STEM_REDUCE || MODE == STEM_DGRAD) gz[i] = *(const bf16x8*)(p.dz + (m0 + r) * 32 + 8 * g);
        }
        m0 += 16;
        x0 += 16;
        hpix += 16;
        if (x0 == W) {                                                          // next image row: skip the two halo columns
            x0 = 0;
            hpix += 2;
            if (++yy == H) {                                                    // next image: skip its two halo rows
                yy = 0;
                hpix += 2 * Wp;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < STEM_NB; ++i) {
        if (!ok[i]) break;
        const bf16x8 p0 = __builtin_bit_cast(bf16x8, make_uint4(la[i].x, la[i].y, lb[i].x, lb[i].y));
        const bf16x8 p1 = __builtin_bit_cast(bf16x8, g == 0 ? make_uint4(lc[i].x, lc[i].y, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u));
        const bf16x8 gzc = gz[i];
        const int64_t hcur = hp[i];
        f32x4 acc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c][0], p0, acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c][1], p1, acc[c], 0, 0, 0);
        }
        bf16_t* oo = nullptr;
        if constexpr (MODE == STEM_APPLY || MODE == STEM_DGRAD) oo = p.out + (hcur + Wp + 1) * 32 + 8 * g;   // interior pixel (y, x)
        bf16x8 o;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float yv = acc[c][j];
                if constexpr (MODE == STEM_STATS) {
                    s1[c][j] += yv;
                    s2[c][j] += yv * yv;
                } else if constexpr (MODE == STEM_APPLY) {
                    o[4 * c + j] = (bf16_t)silu_f(yv * sc[c][j] + sh[c][j]);
                } else {
                    const float du = (float)gzc[4 * c + j] * silu_grad(yv * sc[c][j] + sh[c][j]);
                    if constexpr (MODE == STEM_REDUCE) {
                        s1[c][j] += du;
                        s2[c][j] += du * (yv - mu[c][j]) * rs[c][j];
                    } else {
                        o[4 * c + j] = (bf16_t)(ka[c][j] * du + k1[c][j] * yv + k2[c][j]);
                    }
                }
            }
        }
        if constexpr (MODE == STEM_APPLY || MODE == STEM_DGRAD) *(bf16x8*)oo = o;
    }
    }
    }
    if constexpr (MODE == STEM_STATS || MODE == STEM_REDUCE) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    s1[c][j] += __shfl_xor(s1[c][j], o);
                    s2[c][j] += __shfl_xor(s2[c][j], o);
                }
                if (r == 0) {
                    red[0][wv][8 * g + 4 * c + j] = s1[c][j];
                    red[1][wv][8 * g + 4 * c + j] = s2[c][j];
                }
            }
        __syncthreads();
        if (tid < 64) {
            const int which = tid >> 5, ch = tid & 31;
            p.part[((int64_t)blockIdx.x * 2 + which) * 32 + ch] = (red[which][0][ch] + red[which][1][ch]) + (red[which][2][ch] + red[which][3][ch]);
        }
    }
}

constexpr int WGRAD_BLOCKS = 1024;
constexpr int STEM_FUSED_GRID = 2048;   // persistent blocks of the fused stem passes: 8 per CU

}  // namespace

extern "C" {

// the MFMA forward serves the bf16 path when an image row is a whole number of 16-pixel tiles
static bool stem_mfma(int dtype, int W) {
    static const bool on = [] { const char* e = getenv("FVA_STEM_MFMA"); return !e || atoi(e) != 0; }();
    return on && dtype == FVA_BF16 && W % 16 == 0;
}

int32_t fva_stem_stat_blocks(int dtype, int B, int H, int W) {
    const int64_t M = (int64_t)B * H * W;
    return stem_mfma(dtype, W) ? cdiv(M, STEM_BLOCK_PIX) : cdiv(M, 256);
}

int64_t fva_stem_fwd_workspace(int dtype, int B, int H, int W) {
    return stem_mfma(dtype, W) ? (int64_t)B * (H + 2) * (W + 2) * 8 + 64 : 0;   // + slack: the wgrad's 4-pixel windows read one pixel past a row
}

int fva_stem_fwd(int dtype, const float* img, const float* w, void* y, float* stats, void* workspace, int64_t workspace_bytes, int B,
                 int Cin, int H, int W, int Cout, void* stream) {
    if (!img || !w || !y) return fva_fail(FVA_ERR_ARG, "fva_stem_fwd: null pointer");
    if (Cout != CO || Cin < 1 || Cin > 3) return fva_fail(FVA_ERR_ARG, "fva_stem_fwd: needs Cout==32, Cin<=3 (got %d, %d)", Cout, Cin);
    const int64_t M = (int64_t)B * H * W;
    const int grid = cdiv(M, 256);
    hipStream_t s = (hipStream_t)stream;
    if (stem_mfma(dtype, W)) {
        const int64_t need = fva_stem_fwd_workspace(dtype, B, H, W);
        if (!workspace || workspace_bytes < need) return fva_fail(FVA_ERR_WORKSPACE, "fva_stem_fwd: workspace %lld < %lld", (long long)workspace_bytes, (long long)need);
        const int64_t hp = (need - 64) / 8;
        hipLaunchKernelGGL(stem_pack_kernel, dim3((int)((hp + 255) / 256 < 65536 ? (hp + 255) / 256 : 65536)), dim3(256), 0, s, img, (uint2*)workspace, B, Cin, H, W);
        FVA_LAUNCH_CHECK("stem_pack_kernel");
        hipLaunchKernelGGL(stem_fwd_mfma_kernel, dim3(cdiv(M, STEM_BLOCK_PIX)), dim3(256), 0, s, (const uint2*)workspace, w, (bf16_t*)y, stats, Cin, H,
                           W, M);
        FVA_LAUNCH_CHECK("stem_fwd_mfma_kernel");
        return FVA_OK;
    }
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL(stem_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, img, w, (bf16_t*)y, stats, B, Cin, H, W, M);
    else if (dtype == FVA_F32)
        hipLaunchKernelGGL(stem_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, img, w, (float*)y, stats, B, Cin, H, W, M);
    else
        return fva_fail(FVA_ERR_ARG, "fva_stem_fwd: bad dtype");
    FVA_LAUNCH_CHECK("stem_fwd_kernel");
    return FVA_OK;
}

int64_t fva_stem_wgrad_workspace(int B, int Cin, int H, int W, int Cout) {
    (void)B; (void)H; (void)W;
    return (int64_t)WGRAD_BLOCKS * Cout * Cin * 9 * 4;
}

int fva_stem_wgrad(int dtype, const float* img, const void* dy, float* dw, int accumulate, void* workspace, int64_t workspace_bytes,
                   int B, int Cin, int H, int W, int Cout, void* stream) {
    if (!img || !dy || !dw || !workspace) return fva_fail(FVA_ERR_ARG, "fva_stem_wgrad: null pointer");
    if (Cout != CO || Cin < 1 || Cin > 3) return fva_fail(FVA_ERR_ARG, "fva_stem_wgrad: needs Cout==32, Cin<=3");
    if (workspace_bytes < fva_stem_wgrad_workspace(B, Cin, H, W, Cout)) return fva_fail(FVA_ERR_WORKSPACE, "fva_stem_wgrad: workspace too small");
    const int64_t M = (int64_t)B * H * W;
    const int ntiles = cdiv(M, 128);
    const int grid = ntiles < WGRAD_BLOCKS ? ntiles : WGRAD_BLOCKS;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FVA_BF16)
        hipLaunchKernelGGL(stem_wgrad_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, img, (const bf16_t*)dy, (float*)workspace, B, Cin, H, W, M, ntiles);
    else if (dtype == FVA_F32)
        hipLaunchKernelGGL(stem_wgrad_kernel<float>, dim3(grid), dim3(256), 0, s, img, (const float*)dy, (float*)workspace, B, Cin, H, W, M, ntiles);
    else
        return fva_fail(FVA_ERR_ARG, "fva_stem_wgrad: bad dtype");
    FVA_LAUNCH_CHECK("stem_wgrad_kernel");
    const int n = Cout * Cin * 9;
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(n), dim3(256), 0, s, (const float*)workspace, dw, n, grid, accumulate);
    FVA_LAUNCH_CHECK("stem_wgrad_reduce_kernel");
    return FVA_OK;
}

/* conv0 fused with its BatchNorm / SiLU passes (bf16, W % 16 == 0): see stem_fused_kernel.  mode 0 STATS -> part; 1 APPLY ->
 * out (halo z, border zeroed here); 2 REDUCE -> part; 3 DGRAD -> out (halo dY, border zeroed here).  img4 = the NHWC4 image
 * copy (fva_stem_pack); part has fva_stem_fused_blocks() rows of [2][32]; unused pointers may be NULL. */
int32_t fva_stem_fused_blocks(int B, int H, int W) {
    const int64_t blocks = cdiv((int64_t)B * H * W, STEM_BLOCK_PIX);
    return (int32_t)(blocks < STEM_FUSED_GRID ? blocks : STEM_FUSED_GRID);
}

int fva_stem_pack(const float* img, void* img4, int64_t img4_bytes, int B, int Cin, int H, int W, void* stream) {
    if (!img || !img4 || Cin < 1 || Cin > 3) return fva_fail(FVA_ERR_ARG, "fva_stem_pack: bad argument");
    const int64_t need = fva_stem_fwd_workspace(FVA_BF16, B, H, W);
    if (need == 0) return fva_fail(FVA_ERR_ARG, "fva_stem_pack: W = %d is not a multiple of 16", W);
    if (img4_bytes < need) return fva_fail(FVA_ERR_WORKSPACE, "fva_stem_pack: buffer %lld < %lld", (long long)img4_bytes, (long long)need);
    const int64_t hp = (need - 64) / 8;
    hipLaunchKernelGGL(stem_pack_kernel, dim3((int)((hp + 255) / 256 < 65536 ? (hp + 255) / 256 : 65536)), dim3(256), 0, (hipStream_t)stream, img,
                       (uint2*)img4, B, Cin, H, W);
    FVA_LAUNCH_CHECK("stem_pack_kernel");
    return FVA_OK;
}

int fva_stem_fused(int mode, const void* img4, const float* w, const void* dz, const float* scale, const float* shift, const float* mean,
                   const float* rstd, const float* coef, void* out, float* part, int B, int Cin, int H, int W, void* stream) {
    if (!img4 || !w || Cin < 1 || Cin > 3 || W % 16) return fva_fail(FVA_ERR_ARG, "fva_stem_fused: bad argument");
    const bool need_affine = mode != STEM_STATS, need_dz = mode == STEM_REDUCE || mode == STEM_DGRAD;
    if ((need_affine && (!scale || !shift)) || (need_dz && (!dz || !mean || !rstd)) || (mode == STEM_DGRAD && !coef) ||
        ((mode == STEM_APPLY || mode == STEM_DGRAD) && !out) || ((mode == STEM_STATS || mode == STEM_REDUCE) && !part))
        return fva_fail(FVA_ERR_ARG, "fva_stem_fused: missing operand for mode %d", mode);
    StemFusedParams p{};
    p.img4 = (const uint2*)img4; p.w = w; p.dz = (const bf16_t*)dz;
    p.scale = scale; p.shift = shift; p.mean = mean; p.rstd = rstd; p.coef = coef;
    p.out = (bf16_t*)out; p.part = part;
    p.Cin = Cin; p.H = H; p.W = W; p.M = (int64_t)B * H * W;
    if (p.M / 16 > INT32_MAX) return fva_fail(FVA_ERR_ARG, "fva_stem_fused: batch of %lld pixels too large", (long long)p.M);
    const dim3 grid(fva_stem_fused_blocks(B, H, W));
    hipStream_t s = (hipStream_t)stream;
    switch (mode) {
        case STEM_STATS: hipLaunchKernelGGL(stem_fused_kernel<STEM_STATS>, grid, dim3(256), 0, s, p); break;
        case STEM_APPLY: hipLaunchKernelGGL(stem_fused_kernel<STEM_APPLY>, grid, dim3(256), 0, s, p); break;
        case STEM_REDUCE: hipLaunchKernelGGL(stem_fused_kernel<STEM_REDUCE>, grid, dim3(256), 0, s, p); break;
        case STEM_DGRAD: hipLaunchKernelGGL(stem_fused_kernel<STEM_DGRAD>, grid, dim3(256), 0, s, p); break;
        default: return fva_fail(FVA_ERR_ARG, "fva_stem_fused: bad mode %d", mode);
    }
    FVA_LAUNCH_CHECK("stem_fused_kernel");
    if (mode == STEM_APPLY || mode == STEM_DGRAD) return fva_zero_halo_border(out, B, H, W, 4, 1, s);
    return FVA_OK;
}

}  // extern "C"
