// Implicit-GEMM convolution on MFMA for gfx950: forward, data-gradient and the biased detection head.
//
//   out[m][n] = sum_t sum_c in[pix(m) + tap_pix[t]][c] * wt[tap_w[t]][n][c]
//
// Activations are halo NHWC (channels contiguous, zero border), so a 3x3 tap is a plain pixel offset and
// needs no bounds check.  One k-tile is a 128-byte slice of channels of one tap (64 bf16 / 32 f32);
// A (pixels) and B (output channels) tiles are staged by LDS-DMA (global_load_lds, 16 B/lane) into a
// double-buffered, XOR-swizzled LDS image and consumed with ds_read_b128 + v_mfma_f32_16x16x32_bf16
// (bf16) or v_mfma_f32_32x32x2_f32 (exact fp32).  Roofline: MFMA-bound (2*M*N*K flop per launch).
//
// Replaces nn.Conv2d forward/backward-data as issued by ConvBlock3x3/ConvBlock1x1
// (reference classfication/models/darknet53.py:5-9, 22-44) and the head conv (detection/head/yolov3head.py:50).
#include <stddef.h>
#include <stdlib.h>

#include "common.h"

namespace {

enum { EPI_STATS = 0, EPI_PLAIN = 1, EPI_HEAD = 2, EPI_BNACT = 3, EPI_BNB = 4 };   // BNACT: out = SiLU(acc * scale[n] + shift[n]) (+ addend); BNB: PLAIN + fused BatchNorm-backward statistics (IgemmParams::bnb_*)
constexpr int MAX_TAPS = 10;

struct IgemmParams {
    const void* in;
    const void* wt;
    void* out;
    float* stats;
    int acc_rep, bnb_rep, ax_rep;   // replicas of stats_acc / bnb_acc / ax_acc (+ ax_zero): powers of two, see common.h
    long long* stats_acc;   // instead of the table `stats`: per-channel fixed-point accumulators acc[2][2][N] (common.h fx_atomic_add) that every tile ADDS to
    const float* bias;
    const float* scale;  // EPI_BNACT: per-output-channel affine (eval-mode BatchNorm folded in) before SiLU; NULL = 1 (bias only)
    int act;             // EPI_BNACT: 0 SiLU, 1 ReLU (VGG blocks of the two-stage head), 2 none
    const float* shift;
    int M, N, C;
    int OW, OHW;
    FastDiv div_ow, div_ohw;
    int in_img, in_row;  // pixels
    int sy, sx, y0, x0;
    int ktiles, kt_per_tap, halfrow;
    int ktaps;   // > 0: k-tiles run channel-slice-major, ktaps taps per slice (see finish_taps); 0: tap-major
    int tap_pix[MAX_TAPS];
    int tap_w[MAX_TAPS];
    int out_dense;  // out pixel index == m
    int out_img, out_row, osy, osx, ooy, oox, out_pitch;
    const void* addend;  // optional tensor added in the epilogue (same addressing as out)
    int nblocks;         // column blocks
    // EPI_PLAIN, optional: BatchNorm-backward statistics of the layer that PRODUCED this convolution's input, taken in the dgrad
    // epilogue that writes dz (= this launch's output incl. the addend, as stored): per block and channel sum(dU) and
    // sum(dU * xhat), dU = dz * SiLU'(y * scale + shift), xhat = (y - mean) * rstd, with y that layer's pre-BN output (same
    // [pixel][channel] addressing as the output).  One plain store per block, channel and sum: deterministic.
    // Table row of block (mblk, column n) = (bnb_row0 + mblk) * (N / bnb_C) + n / bnb_C, column n % bnb_C.
    const void* bnb_y;
    const float *bnb_scale, *bnb_shift, *bnb_mean, *bnb_rstd;
    float* bnb_part;
    long long* bnb_acc;     // instead of bnb_part: the producer layer's backward accumulator (common.h fx_atomic_add)
    int bnb_row0, bnb_C;
    // AX (fva_conv1x1_fwd_apply): the A operand of a 1x1 convolution is PRODUCED by the kernel from the raw output of the block before
    // it -- z = SiLU(ax_y * scale[c] + shift[c]) (+ ax_res), rounded to bf16 -- written once to the halo buffer ax_z (zero border
    // included) for everybody else who reads z, and straight into the LDS operand tile for this convolution.
    const void* ax_y;      // dense [M][C]
    const void* ax_res;    // halo buffer [B][H + 2 ax_res_pad][W + 2 ax_res_pad][C], or NULL
    void* ax_z;            // halo buffer [B][H + 2 ax_pad][W + 2 ax_pad][C]
    const float *ax_scale, *ax_shift;
    int ax_pad, ax_res_pad, ax_H, ax_W;
    // ... with the statistics of the block before in a fixed-point accumulator (common.h): finalised in this launch's prologue
    long long* ax_acc;     // NULL: ax_scale / ax_shift are given
    long long* ax_zero;
    const float *ax_gamma, *ax_beta;
    float *ax_rm, *ax_rv;
    long long* ax_nbt;
    float ax_momentum, ax_eps;
    BnN ax_n;
    float *ax_save_mean, *ax_save_rstd, *ax_scale_out, *ax_shift_out;
    long long* stamps;   // diagnostic (fva_conv_debug_stamps): [stamp_rows][8] wall-clock stamps of the block's phases (igemm_kernel)
    int stamp_rows;
    int pt_tx, pt_ty, pt_H, pt_W;   // pconv_kernel: tiles of 8 x 32 output pixels per image (x, y), output image size
    int pt_tiles, pt_tpb;           //   tiles of the launch, consecutive tiles per block (the resident weights are staged once per block)
    FastDiv pt_div_tx, pt_div_img;  //   divisions by pt_tx and pt_tx * pt_ty
};

// coefficients of eight (bf16 chunk) consecutive channels for the fused BatchNorm-backward statistics
struct BnbCoef {
    float sc[8], sh[8], mu[8], rs[8];
};
__device__ __forceinline__ void bnb_load(const IgemmParams& p, int n, BnbCoef& k) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int c = n + e;
        c = c < p.N ? c : p.N - 1;
        c = c >= p.bnb_C ? c - p.bnb_C : c;      // paired stride-2 dgrad: the columns are two pixels' channels
        k.sc[e] = p.bnb_scale[c]; k.sh[e] = p.bnb_shift[c]; k.mu[e] = p.bnb_mean[c]; k.rs[e] = p.bnb_rstd[c];
    }
}
// The same coefficients staged in LDS once per tile, [4][BN] floats indexed by the tile column: read from global memory in the
// epilogue, the 32 dependent loads per thread sat in front of the statistics arithmetic (1-2 us per tile, tools/tile_timing.py pw).
template <int BN, int NT>
__device__ __forceinline__ float bnb_coef_at(const IgemmParams& p, int i, int n0) {   // table entry i of [4][BN]
    const int which = i / BN, col = i - which * BN;
    int c = n0 + col;
    c = c < p.N ? c : p.N - 1;
    c = c >= p.bnb_C ? c - p.bnb_C : c;
    const float* src = which == 0 ? p.bnb_scale : which == 1 ? p.bnb_shift : which == 2 ? p.bnb_mean : p.bnb_rstd;
    return src[c];
}
template <int BN, int NT>
__device__ __forceinline__ void bnb_fill_lds(const IgemmParams& p, float* tab, int tid, int n0) {
    for (int i = tid; i < 4 * BN; i += NT) tab[i] = bnb_coef_at<BN, NT>(p, i, n0);
}
template <int BN>
__device__ __forceinline__ void bnb_load_lds(const float* tab, int col, BnbCoef& k) {
    const f32x4* t = (const f32x4*)(tab + col);
    const f32x4 a0 = t[0], a1 = t[1], b0 = t[BN / 4], b1 = t[BN / 4 + 1], c0 = t[2 * BN / 4], c1 = t[2 * BN / 4 + 1], d0 = t[3 * BN / 4],
                d1 = t[3 * BN / 4 + 1];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        k.sc[e] = a0[e]; k.sc[4 + e] = a1[e];
        k.sh[e] = b0[e]; k.sh[4 + e] = b1[e];
        k.mu[e] = c0[e]; k.mu[4 + e] = c1[e];
        k.rs[e] = d0[e]; k.rs[4 + e] = d1[e];
    }
}
// the block's partial sums, gathered per 16-byte column chunk by the threads of the store loop: red[which][group][BN] in LDS ->
// one store per column and sum
template <int BN, int NT, int CPR>
__device__ __forceinline__ void bnb_finish(const IgemmParams& p, float* red, const float (&s1)[8], const float (&s2)[8], int tid, int mblk,
                                           int n0) {
    constexpr int G = NT / CPR;
    static_assert(NT / 64 >= BN / 32, "one wave per 32-channel slice of the block's columns");
    const int cc = tid % CPR, grp = tid / CPR;
    __syncthreads();   // every thread is done reading the transposed tile
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[(0 * G + grp) * BN + cc * 8 + e] = s1[e];
        red[(1 * G + grp) * BN + cc * 8 + e] = s2[e];
    }
    __syncthreads();
    // wave w owns the 32 columns n0 + 32 w ..: lanes 0-31 the first sum, lanes 32-63 the second (one 128-byte line each), stored
    // write-through so that the launch can fold the table itself 
    const int w = tid >> 6, lane = tid & 63;
    if (w < BN / 32) {
        const int which = lane >> 5, col = w * 32 + (lane & 31);
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) t += red[(which * G + g) * BN + col];
        const int n = n0 + col;
        const int sub = n >= p.bnb_C ? 1 : 0, per = p.N > p.bnb_C ? 2 : 1;
        if (n < p.N) {
            if (p.bnb_acc != nullptr) fx_atomic_add(fx_replica(p.bnb_acc, p.bnb_C, p.bnb_rep), p.bnb_C, which, n - sub * p.bnb_C, t);
            else *(p.bnb_part + (((int64_t)(p.bnb_row0 + mblk) * per + sub) * 2 + which) * p.bnb_C + (n - sub * p.bnb_C)) = t;
        }
    }
}

__device__ __forceinline__ float bnact_f(const IgemmParams& p, float v, int n) {
    const float u = (p.scale ? v * p.scale[n] : v) + p.shift[n];
    return p.act == 0 ? silu_f(u) : p.act == 1 ? fmaxf(u, 0.f) : u;
}


// The epilogue's output tile is written once and not read again by this launch: stored non-temporal it does not evict the input
// rows that the nine taps re-read through L2 (an L2 hit arrives at 34 TB/s, a miss at 6-7: profiles/r02_dma_probe.md).  Same-box A/B
// (bench.py, two runs each): 1041.4 / 1041.5 against 1033.3 / 1036.8 img/s, graph replay 31.8 against 32.1 ms; isolated launches of the
// thin and 1x1 layers 4-20 % faster.  Non-temporal LOADS of the residual addend / the producer's y measured neutral (round 2; removed).
#ifndef FVA_NT_STORES
#define FVA_NT_STORES 1
#endif
__device__ __forceinline__ bf16x8 ld_stream(const bf16_t* p) { return *(const bf16x8*)p; }
__device__ __forceinline__ void st_stream_keep(bf16_t* p, bf16x8 v) { *(bf16x8*)p = v; }   // an output the NEXT launches read: keep it in cache
__device__ __forceinline__ void st_stream(bf16_t* p, bf16x8 v) {
#if FVA_NT_STORES
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), (u32x4*)p);
#else
    *(bf16x8*)p = v;
#endif
}

// The block's transposed bf16 tile (LDS, row pitch PITCH bytes) -> global memory, 16 bytes per thread and step.  A thread keeps
// its 16-byte column chunk over all steps (NT % CPR == 0).  The global operands of a GROUP of steps -- the residual addend, and
// for EPI_BNB the producer's y -- are fetched first, all in flight together, and consumed afterwards: issued one per step behind
// the previous step's store, each load pays a full memory round trip (measured: +60 us on a 100 us dgrad launch).
// row_pixel(r) = output pixel index of tile row r, or -1 when that row is not stored (beyond M, or outside the image for the patch tiles).
template <int EPI, int BM, int BN, int NT, int PITCH, typename RowPixel>
__device__ __forceinline__ void store_tile_bf16(const IgemmParams& p, char* smem, int tid, int n0, int mblk, RowPixel&& row_pixel,
                                                const bf16x8* y_pre = nullptr, const bf16x8* a_pre = nullptr, const float* coef_lds = nullptr) {
    constexpr int CPR = BN / 8, ITERS = BM * CPR / NT, RSTEP = NT / CPR;
    constexpr int GROUP = ITERS % 8 == 0 ? 8 : ITERS % 7 == 0 ? 7 : ITERS;   // 7: the 224-row tile (14 steps); 4: the 256 x 32 patch tile
    static_assert(NT % CPR == 0 && ITERS % GROUP == 0 && GROUP <= 8, "a thread keeps its column chunk; whole groups");
    bf16_t* out = (bf16_t*)p.out;
    const int cc = tid % CPR, r0 = tid / CPR;
    const int n = n0 + cc * 8;
    const bool col_ok = n < p.N;
    const bool has_add = p.addend != nullptr;
    BnbCoef kc;
    float b1[8], b2[8];
    if constexpr (EPI == EPI_BNB) {
        if (coef_lds != nullptr) bnb_load_lds<BN>(coef_lds, cc * 8, kc);
        else bnb_load(p, n, kc);
#pragma unroll
        for (int e = 0; e < 8; ++e) b1[e] = b2[e] = 0.f;
    }
    for (int g = 0; g < ITERS / GROUP; ++g) {
        int64_t oi[GROUP];
        bool ok[GROUP];
        bf16x8 ad[GROUP], yv[GROUP];
#pragma unroll
        for (int j = 0; j < GROUP; ++j) {
            const int64_t pixel = row_pixel(r0 + (g * GROUP + j) * RSTEP);
            ok[j] = col_ok && pixel >= 0;
            oi[j] = ok[j] ? pixel * p.out_pitch + n : 0;            // masked steps read element 0: no branch around a load
        }
        if (has_add) {
            if (a_pre != nullptr) {
#pragma unroll
                for (int j = 0; j < GROUP; ++j) ad[j] = a_pre[j];
            } else {
#pragma unroll
                for (int j = 0; j < GROUP; ++j) ad[j] = ld_stream((const bf16_t*)p.addend + oi[j]);
            }
        }
        if constexpr (EPI == EPI_BNB) {
            if (y_pre != nullptr) {       // fetched before the k loop (ITERS == GROUP): the launch's rounds of tiles run in lockstep,
#pragma unroll                             // so a read issued in the epilogue is exposed, not hidden behind another block's MFMAs
                for (int j = 0; j < GROUP; ++j) yv[j] = y_pre[j];
            } else {
#pragma unroll
                for (int j = 0; j < GROUP; ++j) yv[j] = ld_stream((const bf16_t*)p.bnb_y + oi[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < GROUP; ++j) {
            const int row = r0 + (g * GROUP + j) * RSTEP;
            bf16x8 v = *(const bf16x8*)(smem + row * PITCH + cc * 16);
            if (has_add) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)ad[j][e]);
            }
            if (ok[j]) {
                st_stream(out + oi[j], v);
                if constexpr (EPI == EPI_BNB) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float yf = (float)yv[j][e];
                        const float du = (float)v[e] * silu_grad(yf * kc.sc[e] + kc.sh[e]);
                        b1[e] += du;
                        b2[e] += du * (yf - kc.mu[e]);           // * rstd once, below
                    }
                }
            }
        }
    }
    if constexpr (EPI == EPI_BNB) {
#pragma unroll
        for (int e = 0; e < 8; ++e) b2[e] *= kc.rs[e];
        bnb_finish<BN, NT, CPR>(p, (float*)smem, b1, b2, tid, mblk, n0);
    }
}

template <typename T>
struct Acc;
template <>
struct Acc<bf16_t> {  // per wave 64x64 = 4x4 tiles of 16x16
    f32x4 a[4][4];
};
template <>
struct Acc<float> {  // per wave 64x64 = 2x2 tiles of 32x32
    f32x16 a[2][2];
};

// visit every accumulator element with its (row, col) inside the wave's 64x64 tile.  bf16: the kernel issues its MFMAs with the
// operands swapped (D^T = B^T A^T), so that lane (r, g) holds of tile (mt, nt) the 1 x 4 piece  row mt*16 + r, columns nt*16 + 4g .. +3
// -- four adjacent output channels of one pixel, one 8-byte LDS write in the epilogue instead of four 2-byte ones.
template <typename F>
__device__ __forceinline__ void foreach_acc(Acc<bf16_t>& acc, int lane, F&& f) {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) f(mt * 16 + r, nt * 16 + g * 4 + j, nt, acc.a[mt][nt][j]);
}
template <typename F>
__device__ __forceinline__ void foreach_acc(Acc<float>& acc, int lane, F&& f) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) f(mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, nt * 32 + r, nt, acc.a[mt][nt][e]);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BM, int BN, int EPI, int NSTAGE, bool AX = false>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64) void igemm_kernel(const IgemmParams p) {
    static_assert(!AX || (sizeof(T) == 2 && EPI == EPI_STATS), "the produced-operand form is the bf16 training forward of a 1x1 layer");
    constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
    constexpr int BK = 8 * EPC;               // 128-byte rows
    constexpr int WN = BN / 64;               // waves along n
    constexpr int NW = (BM / 64) * WN;        // waves per block (4 or 8); each owns a 64x64 sub-tile
    constexpr int NT = NW * 64;
    static_assert(NW % 2 == 0, "the source swizzle assumes an even wave count");
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int A_ITERS = BM / (8 * NW), B_ITERS = BN / (8 * NW);
    constexpr bool IS_BF16 = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w / WN, wc = w % WN;
    auto stamp = [&](int i) {   // 0 entry, 6 source rows / k table ready, 7 epilogue operands requested, 1 first DMA issued, 2 first k-tile landed, 3 k loop done, 4 tile staged in LDS, 5 exit
        if (p.stamps && tid == 0 && (int)blockIdx.x < p.stamp_rows) p.stamps[(int64_t)blockIdx.x * 8 + i] = wall_clock64();
    };
    stamp(0);

    // XCD-aware block order: blocks b and b+8 share an XCD (L2); give each XCD a contiguous run of tiles.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, xq = nwg >> 3, xr = nwg & 7;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int mblk = logical / p.nblocks, nblk = logical - mblk * p.nblocks;
    const int m0 = mblk * BM, n0 = nblk * BN;

    // ---- per-lane source rows for the LDS-DMA (fixed for the whole k loop) -----------------------------
    const T* a_ptr[A_ITERS];
    const T* b_ptr[B_ITERS];
    const int lrow = lane >> 3;
    const int sw = ((w & 1) << 2) + (lane >> 4);  // ((tile_row >> 1) & 7) of this lane's rows
    const int chunk = (lane & 7) ^ sw;            // source 16-B chunk that lands at LDS position (lane & 7)
    const int tapsel = chunk >> 2;                // halfrow: which of the tile's two taps
    const int sub = p.halfrow ? (chunk & 3) : chunk;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        int m = m0 + (i * NW + w) * 8 + lrow;
        m = m < p.M ? m : p.M - 1;
        const uint32_t b = fd_div((uint32_t)m, p.div_ohw);
        const uint32_t rem = (uint32_t)m - b * (uint32_t)p.OHW;
        const uint32_t oy = fd_div(rem, p.div_ow);
        const uint32_t ox = rem - oy * (uint32_t)p.OW;
        const int64_t pix = (int64_t)b * p.in_img + (int64_t)(oy * p.sy + p.y0) * p.in_row + (ox * p.sx + p.x0);
        a_ptr[i] = (const T*)p.in + pix * p.C + sub * EPC;
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
        int n = n0 + (i * NW + w) * 8 + lrow;
        n = n < p.N ? n : p.N - 1;
        b_ptr[i] = (const T*)p.wt + (int64_t)n * p.C + sub * EPC;
    }

    // per-k-tile element offsets (A: tap pixel offset * C + channel slice; B: tap matrix + channel slice).  Wide
    // tiles tabulate them once in LDS (no division / dynamic kernarg indexing in the loop); the thin 256x64 tile
    // keeps its LDS at exactly 2 x 40 KiB so that two blocks fit a CU, and selects the tap with a short unrolled scan.
    constexpr bool USE_KTAB = BN >= 128;
    constexpr int COEF_BYTES = (EPI == EPI_BNB && IS_BF16 && BN >= 128) ? 4 * BN * 4 : 0;   // bnb_fill_lds (the 256x64 tile has no LDS to spare)
    float* coef_tab = (float*)(smem + NSTAGE * STAGE);   // EPI_BNB: [4][BN] floats, see bnb_fill_lds
    int* ktab = (int*)(smem + NSTAGE * STAGE + COEF_BYTES);
    // the table's values are fetched now and written to LDS only after the first DMA has been issued: a store right here would
    // put a global round trip in front of everything else the block has to set up
    constexpr int COEF_PER_THREAD = COEF_BYTES > 0 ? 4 * BN / NT : 0;
    float coef_v[COEF_PER_THREAD > 0 ? COEF_PER_THREAD : 1];
    if constexpr (COEF_BYTES > 0) {
#pragma unroll
        for (int j = 0; j < COEF_PER_THREAD; ++j) coef_v[j] = bnb_coef_at<BN, NT>(p, tid + j * NT, n0);
    }
    if constexpr (USE_KTAB) {
        const int nent = p.halfrow ? 2 * p.ktiles : p.ktiles;
        for (int e = tid; e < nent; e += NT) {
            int t, kk;
            if (p.halfrow) { t = e; kk = 0; }
            else if (p.ktaps) { kk = e / p.ktaps; t = e - kk * p.ktaps; }
            else { t = e / p.kt_per_tap; kk = e - t * p.kt_per_tap; }
            int tp = 0, tw = 0;
#pragma unroll
            for (int i = 0; i < MAX_TAPS; ++i)
                if (i == t) { tp = p.tap_pix[i]; tw = p.tap_w[i]; }
            ktab[2 * e] = tp * p.C + kk * BK;
            ktab[2 * e + 1] = tw * p.N * p.C + kk * BK;
        }
        __syncthreads();
    }
    stamp(6);
    int ld_tap = 0, ld_kk = 0;  // running (tap, k-slice) of the next tile to fetch (thin tile only)
    auto load_tile = [&](int kt, int stage) {
        char* sA = smem + stage * STAGE;
        char* sB = sA + A_BYTES;
        int aoff, boff;
        if constexpr (USE_KTAB) {
            const int e = p.halfrow ? 2 * kt + tapsel : kt;
            aoff = ktab[2 * e];
            boff = ktab[2 * e + 1];
        } else {
            const int t = p.halfrow ? ld_tap + tapsel : ld_tap;
            int tp = 0, tw = 0;
#pragma unroll
            for (int i = 0; i < MAX_TAPS; ++i)
                if (i == t) { tp = p.tap_pix[i]; tw = p.tap_w[i]; }
            aoff = tp * p.C + ld_kk * BK;
            boff = tw * p.N * p.C + ld_kk * BK;
            if (p.halfrow) ld_tap += 2;
            else if (++ld_kk == p.kt_per_tap) { ld_kk = 0; ++ld_tap; }
        }
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(a_ptr[i] + aoff), LDS_PTR(sA + (i * NW + w) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(b_ptr[i] + boff), LDS_PTR(sB + (i * NW + w) * 1024), 16, 0, 0);
    };

    // dgrad epilogues (bf16): this thread's eight 16-byte pieces of the residual addend and (EPI_BNB) of the producer's y tile, in
    // the store loop's (row, chunk) order, fetched BEFORE the k loop: the 1x1 dgrads of the residual blocks reduce over two or
    // four k-tiles only and are all epilogue -- reads issued there are exposed (the rounds of tiles run in lockstep)
    bf16x8 y_pre[8], a_pre[8];
    constexpr bool PREFETCH = (EPI == EPI_BNB || EPI == EPI_PLAIN) && sizeof(T) == 2;
    if constexpr (PREFETCH) {
        constexpr int CPR = BN / 8, RSTEP = NT / CPR;
        static_assert(BM * CPR / NT == 8, "the prefetch covers the whole store loop");
        const int cc = tid % CPR, r0 = tid / CPR;
        const int n = n0 + cc * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = m0 + r0 + j * RSTEP;
            int64_t oi = 0;
            if (n < p.N && m < p.M) {
                int64_t pixel = m;
                if (!p.out_dense) {
                    const uint32_t b = fd_div((uint32_t)m, p.div_ohw);
                    const uint32_t rem = (uint32_t)m - b * (uint32_t)p.OHW;
                    const uint32_t oy = fd_div(rem, p.div_ow);
                    const uint32_t ox = rem - oy * (uint32_t)p.OW;
                    pixel = (int64_t)b * p.out_img + (int64_t)(oy * p.osy + p.ooy) * p.out_row + (ox * p.osx + p.oox);
                }
                oi = pixel * p.out_pitch + n;
            }
            if constexpr (EPI == EPI_BNB) y_pre[j] = ld_stream((const bf16_t*)p.bnb_y + oi);
            if (p.addend != nullptr) a_pre[j] = ld_stream((const bf16_t*)p.addend + oi);
        }
    }

    Acc<T> acc;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc.a[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc.a[i][j][e] = 0.f;
    }

    // ---- main loop: two LDS stages, one raw barrier per k-tile (deeper rings leave one block per CU and measured slower).  Both
    // stages are free at entry, so k-tiles 0 and 1 are issued together (the 1x1 layers reduce over 1-16 k-tiles: every exposed DMA
    // round trip counts); from then on k-tile kt+1 is issued right after the barrier of kt, into the stage every wave has just
    // finished reading (counted vmcnt: LDS-DMA retires in issue order).
    static_assert(NSTAGE == 2, "the loop below is written for two stages");
    constexpr int LPT = A_ITERS + B_ITERS;  // DMA instructions per wave and tile
#ifndef FVA_COEF_LATE
#define FVA_COEF_LATE 1
#endif
    if constexpr (COEF_BYTES > 0 && !FVA_COEF_LATE) {
        // before the first LDS-DMA (hipcc drains vmcnt in front of an LDS access once one is in flight); the values were requested
        // ahead of the epilogue operands, so the wait covers loads that have long returned.  Read in the epilogue, behind the barriers.
#pragma unroll
        for (int j = 0; j < COEF_PER_THREAD; ++j) coef_tab[tid + j * NT] = coef_v[j];
    }
    stamp(7);
    int st_cur = 0;
    // ---- AX: this thread's rows of the A tile (the DMA mapping: row (i * NW + w) * 8 + lrow, source chunk `chunk` lands at LDS slot
    // lane & 7), as 32-bit element offsets into y / z / the residual (the host keeps all three below 2^31 elements).  Two register sets
    // of raw operands where they fit (the 128 x 128 tile): k-tile kt + 1 is in flight while k-tile kt is transformed -------------
    constexpr int AXS = (AX && BM == 128) ? 2 : 1;
    uint32_t ax_yo[AX ? A_ITERS : 1], ax_zo[AX ? A_ITERS : 1], ax_ro[AX ? A_ITERS : 1], ax_fl[AX ? A_ITERS : 1];
    bf16x8 ax_ry[AXS][AX ? A_ITERS : 1], ax_rr[AXS][AX ? A_ITERS : 1];
    (void)ax_yo; (void)ax_zo; (void)ax_ro; (void)ax_fl; (void)ax_ry; (void)ax_rr;
    // the eight scales / shifts of a k-tile: from an LDS table [2][C] (the 128 x 128 tile: the k-offset table's LDS, unused in this
    // form; the host keeps C <= 512) or fetched with the raw operands (the thin tile has no LDS to spare and a single register set)
    constexpr bool AXT = AX && USE_KTAB;
    float* ax_tab = (float*)ktab;
    f32x4 ax_csc[2], ax_csh[2];
    (void)ax_csc; (void)ax_csh;
    // scale / shift of channel c: given, or finalised here from the accumulator the previous block's tiles added to (every thread that
    // asks for a channel gets the same bits: same inputs, same arithmetic).  `store`: also written out for the backward pass.
    auto ax_coef = [&](int c, bool store, float& sc_o, float& sh_o) {
        if (!AXT || p.ax_acc == nullptr) {          // (the thin tile, which has neither LDS nor registers to spare, always gets them given)
            sc_o = p.ax_scale[c];
            sh_o = p.ax_shift[c];
            return;
        }
        double s1, s2;
        fx_load2(p.ax_acc, p.C, p.ax_rep, c, s1, s2);
        const BnFwdCoef k = bn_fwd_coef(s1, s2, p.ax_n, p.ax_gamma[c], p.ax_beta[c], p.ax_eps);
        sc_o = k.scale;
        sh_o = k.shift;
        if (store) {
            p.ax_save_mean[c] = k.mean; p.ax_save_rstd[c] = k.rstd; p.ax_scale_out[c] = k.scale; p.ax_shift_out[c] = k.shift;
            if (p.ax_rm) bn_running_update(p.ax_rm, p.ax_rv, c, k, p.ax_n, p.ax_momentum);
        }
    };
    const bool ax_has_res = AX && p.ax_res != nullptr;
    auto ax_issue_b = [&](int kt, int stage) {      // the weight tile of k-tile kt (1x1: one tap, channel slice kt) by LDS-DMA
        char* sB = smem + stage * STAGE + A_BYTES;
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(b_ptr[i] + kt * BK), LDS_PTR(sB + (i * NW + w) * 1024), 16, 0, 0);
    };
    auto ax_load_raw = [&](int kt, auto set_tag) {  // raw y (+ residual) chunks and the eight channels' coefficients of k-tile kt
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            ax_ry[S][i] = *(const bf16x8*)((const bf16_t*)p.ax_y + ax_yo[i] + kt * BK);
            if (ax_has_res) ax_rr[S][i] = *(const bf16x8*)((const bf16_t*)p.ax_res + ax_ro[i] + kt * BK);
        }
        if constexpr (!AXT) {
            const int c0 = kt * BK + chunk * 8;
            ax_csc[0] = *(const f32x4*)(p.ax_scale + c0); ax_csc[1] = *(const f32x4*)(p.ax_scale + c0 + 4);
            ax_csh[0] = *(const f32x4*)(p.ax_shift + c0); ax_csh[1] = *(const f32x4*)(p.ax_shift + c0 + 4);
        }
    };
    // one k-tile of the produced operand: transform set S into the stage that MFMA(kt - 2) read last, send z on its way, barrier
    auto ax_step = [&](int kt, auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        if constexpr (AXS == 2) {
            // the next k-tile's raw operands are requested FIRST and stay in flight across this step; the counted wait below lets
            // exactly them stand and retires everything older: this k-tile's weights (LDS-DMA) and raw operands, the previous z stores
            if (kt + 1 < p.ktiles) {
                ax_load_raw(kt + 1, std::integral_constant<int, S ^ 1>{});
                if (ax_has_res) wait_vmcnt<2 * A_ITERS>();
                else wait_vmcnt<A_ITERS>();
            } else {
                wait_vmcnt<0>();
            }
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t sAw = (uint32_t)(size_t)LDS_PTR(smem + st_cur * STAGE);
        // this k-tile's eight scales and shifts from the LDS table (inline asm: see the store below)
        f32x4 ax_sc[2], ax_sh[2];
        if constexpr (!AXT) {
            ax_sc[0] = ax_csc[0]; ax_sc[1] = ax_csc[1]; ax_sh[0] = ax_csh[0]; ax_sh[1] = ax_csh[1];
        } else {
            const uint32_t ta = (uint32_t)(size_t)LDS_PTR(ax_tab) + (uint32_t)(kt * BK + chunk * 8) * 4u;
            const uint32_t tb = ta + (uint32_t)p.C * 4u;
            asm volatile("ds_read_b128 %0, %1" : "=v"(ax_sc[0]) : "v"(ta));
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(ax_sc[1]) : "v"(ta));
            asm volatile("ds_read_b128 %0, %1" : "=v"(ax_sh[0]) : "v"(tb));
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(ax_sh[1]) : "v"(tb));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // z leaves once, for every other reader of this activation; a row on the edge of its image also zeroes the halo pixels next
        // to it (every border pixel has exactly one such neighbour row; the corners are taken by the corner rows)
        bf16_t* zb = (bf16_t*)p.ax_z;
        const bf16x8 zero = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        const int zrow = (p.ax_W + 2 * p.ax_pad) * p.C;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float o = bn_silu_fwd_elem((float)ax_ry[S][i][e], ax_sc[e >> 2][e & 3], ax_sh[e >> 2][e & 3]);
                if (ax_has_res) o += (float)ax_rr[S][i][e];
                t[e] = (bf16_t)o;
            }
            // inline asm: a store hipcc can see draws `s_waitcnt vmcnt(0)` in front of it while an LDS-DMA is in flight, which would
            // also drain the raw operands just requested
            asm volatile("ds_write_b128 %0, %1" ::"v"(sAw + (uint32_t)((i * NW + w) * 1024 + lane * 16)), "v"(t) : "memory");
            const uint32_t f = ax_fl[i];
            if (f & 1u) {
                bf16_t* q = zb + ax_zo[i] + kt * BK;
                st_stream_keep(q, t);
                if (f & 30u) {
                    if (f & 2u) *(bf16x8*)(q - p.C) = zero;
                    if (f & 4u) *(bf16x8*)(q + p.C) = zero;
                    if (f & 8u) {
                        *(bf16x8*)(q - zrow) = zero;
                        if (f & 2u) *(bf16x8*)(q - zrow - p.C) = zero;
                        if (f & 4u) *(bf16x8*)(q - zrow + p.C) = zero;
                    }
                    if (f & 16u) {
                        *(bf16x8*)(q + zrow) = zero;
                        if (f & 2u) *(bf16x8*)(q + zrow - p.C) = zero;
                        if (f & 4u) *(bf16x8*)(q + zrow + p.C) = zero;
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < p.ktiles) {
            ax_issue_b(kt + 1, st_cur ^ 1);
            if constexpr (AXS == 1) ax_load_raw(kt + 1, std::integral_constant<int, 0>{});
        }
    };
    if constexpr (AX) {
        const int zp = p.ax_pad, rp = p.ax_res_pad;
        const uint32_t zW = (uint32_t)(p.ax_W + 2 * zp), zHW = (uint32_t)(p.ax_H + 2 * zp) * zW;
        const uint32_t rW = (uint32_t)(p.ax_W + 2 * rp), rHW = (uint32_t)(p.ax_H + 2 * rp) * rW;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            int m = m0 + (i * NW + w) * 8 + lrow;
            const bool live = m < p.M;
            m = live ? m : p.M - 1;
            const uint32_t b = fd_div((uint32_t)m, p.div_ohw);
            const uint32_t rem = (uint32_t)m - b * (uint32_t)p.OHW;
            const uint32_t oy = fd_div(rem, p.div_ow);
            const uint32_t ox = rem - oy * (uint32_t)p.OW;
            ax_yo[i] = (uint32_t)m * (uint32_t)p.C + chunk * 8;
            ax_zo[i] = (b * zHW + (oy + zp) * zW + ox + zp) * (uint32_t)p.C + chunk * 8;
            ax_ro[i] = (b * rHW + (oy + rp) * rW + ox + rp) * (uint32_t)p.C + chunk * 8;
            ax_fl[i] = (live ? 1u : 0u) | (zp > 0 && live ? ((ox == 0 ? 2u : 0u) | (ox == (uint32_t)p.ax_W - 1 ? 4u : 0u) | (oy == 0 ? 8u : 0u) |
                                                           (oy == (uint32_t)p.ax_H - 1 ? 16u : 0u)) : 0u);
        }
        // the coefficient table, written before the first LDS-DMA is in flight (hipcc drains vmcnt in front of an LDS store it can
        // see once one is); read behind the first barrier at the earliest... by the threads that wrote other entries: hence the sync
        ax_load_raw(0, std::integral_constant<int, 0>{});      // in flight across the table's (cold) accumulator reads
        if constexpr (AXT) {
            for (int c = tid; c < p.C; c += NT) ax_coef(c, logical == 0, ax_tab[c], ax_tab[p.C + c]);
            __syncthreads();
        }
        if (AXT && p.ax_acc != nullptr && logical == 0) {     // the other direction's accumulator back to zero (this one is still being read)
            if (tid == 0 && p.ax_nbt) *p.ax_nbt += 1;
            fx_zero(p.ax_zero, p.C, p.ax_rep, tid, NT);
        }
        ax_issue_b(0, 0);
    } else {
        load_tile(0, 0);
        if (p.ktiles > 1) load_tile(1, 1);
    }
    stamp(1);
    for (int kt = 0; kt < p.ktiles; ++kt) {
        if constexpr (AX) {
            if (AXS == 1 || (kt & 1) == 0) ax_step(kt, std::integral_constant<int, 0>{});
            else ax_step(kt, std::integral_constant<int, AXS - 1>{});
            if (kt == 0) stamp(2);
        } else {
        if (kt == 0 && p.ktiles > 1) wait_vmcnt<LPT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt == 0) stamp(2);
        if (kt >= 1 && kt + 1 < p.ktiles) load_tile(kt + 1, st_cur ^ 1);
        }
        const char* sA = smem + st_cur * STAGE;
        const char* sB = sA + A_BYTES;
        st_cur ^= 1;
        if constexpr (IS_BF16) {
            const int r = lane & 15, g = lane >> 4, sr = (r >> 1) & 7;
            const char* pa = sA + (wr * 64 + r) * 128;
            const char* pb = sB + (wc * 64 + r) * 128;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int co = ((ks * 4 + g) ^ sr) << 4;
                bf16x8 af[4], bfr[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    af[t] = *(const bf16x8*)(pa + t * 16 * 128 + co);
                    bfr[t] = *(const bf16x8*)(pb + t * 16 * 128 + co);
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc.a[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc.a[mt][nt], 0, 0, 0);   // swapped: see foreach_acc
            }
        } else {
            const int r = lane & 31, h = lane >> 5, sr = (r >> 1) & 7;
            const char* pa = sA + (wr * 64 + r) * 128;
            const char* pb = sB + (wc * 64 + r) * 128;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = ((2 * q + h) ^ sr) << 4;
                f32x4 af[2], bfr[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    af[t] = *(const f32x4*)(pa + t * 32 * 128 + co);
                    bfr[t] = *(const f32x4*)(pb + t * 32 * 128 + co);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc.a[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][j], bfr[nt][j], acc.a[mt][nt], 0, 0, 0);
            }
        }
    }
    __syncthreads();  // LDS is free for the epilogue
    stamp(3);
    if constexpr (COEF_BYTES > 0 && FVA_COEF_LATE) {
        // round 3: the table is written HERE, behind the k loop -- its values were requested at block entry and returned long ago, no
        // LDS-DMA is in flight, and the first DMA of the block no longer waits for them (the write used to sit in front of it: a
        // global round trip of the 4-5 us "setup" phase of every fused 1x1 dgrad tile, profiles/r02_tile_phases.md).  Read in the
        // store loop, behind the barrier that follows the transposition.
#pragma unroll
        for (int j = 0; j < COEF_PER_THREAD; ++j) coef_tab[tid + j * NT] = coef_v[j];
    }

    const int wrow0 = wr * 64, wcol0 = wc * 64;

    // ---- BatchNorm partial statistics: per-column sum and sum of squares over this block's valid rows ----
    if constexpr (EPI == EPI_STATS) {
        if (p.stats != nullptr || p.stats_acc != nullptr) {
            // statistics of the fp32 accumulators (the bf16 rounding of the stored tile is zero-mean, 2^-9 relative:
            // far below the batch-statistics noise); full tiles take the mask-free path
            constexpr int WMc = BM / 64;
            if constexpr (IS_BF16) {
                // a lane's accumulators are 16 rows (mt, r fixed) x 16 columns (nt, j): column sums over mt in registers, then the
                // 16 lane rows r and the BM / 64 wave rows through LDS, added by the column's thread in a fixed order
                const int r = lane & 15, g = lane >> 4;
                f32x4 s1[4], s2[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) s1[nt] = s2[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                const bool whole = m0 + BM <= p.M;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const bool live = whole || m0 + wrow0 + mt * 16 + r < p.M;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const f32x4 v = live ? acc.a[mt][nt] : f32x4{0.f, 0.f, 0.f, 0.f};
                        s1[nt] += v;
                        s2[nt] += v * v;
                    }
                }
                float* red = (float*)smem;  // [2][BM/64][16][RP]
                constexpr int RP = BN + 4;  // rows 16 bytes apart in the banks
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    *(f32x4*)(red + ((0 * WMc + wr) * 16 + r) * RP + wcol0 + nt * 16 + 4 * g) = s1[nt];
                    *(f32x4*)(red + ((1 * WMc + wr) * 16 + r) * RP + wcol0 + nt * 16 + 4 * g) = s2[nt];
                }
                __syncthreads();
                if (w < BN / 32) {   // wave w: columns n0 + 32 w .., lanes 0-31 the sum, lanes 32-63 the sum of squares 
                    const int which = lane >> 5, col = w * 32 + (lane & 31);
                    float s = 0.f;
#pragma unroll
                    for (int k = 0; k < WMc * 16; ++k) s += red[(which * WMc * 16 + k) * RP + col];
                    if (n0 + col < p.N) {
                        if (p.stats_acc != nullptr) fx_atomic_add(fx_replica(p.stats_acc, p.N, p.acc_rep), p.N, which, n0 + col, s);
                        else *(p.stats + ((int64_t)mblk * 2 + which) * p.N + n0 + col) = s;
                    }
                }
                __syncthreads();
            } else {
                float s1[2], s2[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) s1[i] = s2[i] = 0.f;
                if (m0 + BM <= p.M) {
                    foreach_acc(acc, lane, [&](int, int, int nt, float v) {
                        s1[nt] += v;
                        s2[nt] += v * v;
                    });
                } else {
                    foreach_acc(acc, lane, [&](int row, int, int nt, float v) {
                        const float x = (m0 + wrow0 + row < p.M) ? v : 0.f;
                        s1[nt] += x;
                        s2[nt] += x * x;
                    });
                }
                float* red = (float*)smem;  // [2][BM/64][BN]
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    s1[i] += __shfl_xor(s1[i], 32);
                    s2[i] += __shfl_xor(s2[i], 32);
                    if (lane < 32) {
                        red[(0 * WMc + wr) * BN + wcol0 + i * 32 + lane] = s1[i];
                        red[(1 * WMc + wr) * BN + wcol0 + i * 32 + lane] = s2[i];
                    }
                }
                __syncthreads();
                if (w < BN / 32) {
                    const int which = lane >> 5, col = w * 32 + (lane & 31);
                    float s = 0.f;
#pragma unroll
                    for (int k = 0; k < WMc; ++k) s += red[(which * WMc + k) * BN + col];
                    if (n0 + col < p.N) {
                        if (p.stats_acc != nullptr) fx_atomic_add(fx_replica(p.stats_acc, p.N, p.acc_rep), p.N, which, n0 + col, s);
                        else *(p.stats + ((int64_t)mblk * 2 + which) * p.N + n0 + col) = s;
                    }
                }
                __syncthreads();
            }
        }
    }

    auto out_pixel = [&](int m) -> int64_t {
        if (p.out_dense) return m;
        const uint32_t b = fd_div((uint32_t)m, p.div_ohw);
        const uint32_t rem = (uint32_t)m - b * (uint32_t)p.OHW;
        const uint32_t oy = fd_div(rem, p.div_ow);
        const uint32_t ox = rem - oy * (uint32_t)p.OW;
        return (int64_t)b * p.out_img + (int64_t)(oy * p.osy + p.ooy) * p.out_row + (ox * p.osx + p.oox);
    };

    if constexpr (EPI == EPI_HEAD) {
        // fp32 output with bias, row pitch out_pitch (= N, odd): direct 4-byte stores
        float* out = (float*)p.out;
        foreach_acc(acc, lane, [&](int row, int col, int, float v) {
            const int m = m0 + wrow0 + row, n = n0 + wcol0 + col;
            if (m < p.M && n < p.N) out[(int64_t)m * p.out_pitch + n] = v + p.bias[n];
        });
    } else if constexpr (!IS_BF16) {
        // fp32: a wave row is 32 lanes x 4 B = one 128-B line already
        float* out = (float*)p.out;
        int64_t pix_cache = -1;
        int m_cache = -1;
        constexpr bool bnb = EPI == EPI_BNB;
        float b1[2] = {0.f, 0.f}, b2[2] = {0.f, 0.f}, ksc[2], ksh[2], kmu[2], krs[2];
        if constexpr (bnb) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                int c = n0 + wcol0 + nt * 32 + (lane & 31);
                c = c < p.N ? c : p.N - 1;
                c = c >= p.bnb_C ? c - p.bnb_C : c;
                ksc[nt] = p.bnb_scale[c]; ksh[nt] = p.bnb_shift[c]; kmu[nt] = p.bnb_mean[c]; krs[nt] = p.bnb_rstd[c];
            }
        }
        foreach_acc(acc, lane, [&](int row, int col, int nt, float v) {
            const int m = m0 + wrow0 + row, n = n0 + wcol0 + col;
            if (m < p.M && n < p.N) {
                if (m != m_cache) {
                    m_cache = m;
                    pix_cache = out_pixel(m);
                }
                const int64_t oi = pix_cache * p.out_pitch + n;
                if constexpr (EPI == EPI_BNACT) v = bnact_f(p, v, n);
                const float o = p.addend ? ((const float*)p.addend)[oi] + v : v;
                out[oi] = o;
                if constexpr (EPI == EPI_BNB) {
                    {
                        const float yv = ((const float*)p.bnb_y)[oi];
                        const float du = o * silu_grad(yv * ksc[nt] + ksh[nt]);
                        b1[nt] += du;
                        b2[nt] += du * (yv - kmu[nt]) * krs[nt];
                    }
                }
            }
        });
        if constexpr (EPI == EPI_BNB) {
            {   // same wave-shuffle -> LDS -> plain-store scheme as the forward statistics
                float* red = (float*)smem;  // [2][BM/64][BN]
                constexpr int WMc = BM / 64;
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    b1[i] += __shfl_xor(b1[i], 32);
                    b2[i] += __shfl_xor(b2[i], 32);
                    if (lane < 32) {
                        red[(0 * WMc + wr) * BN + wcol0 + i * 32 + lane] = b1[i];
                        red[(1 * WMc + wr) * BN + wcol0 + i * 32 + lane] = b2[i];
                    }
                }
                __syncthreads();
                if (w < BN / 32) {
                    const int which = lane >> 5, col = w * 32 + (lane & 31);
                    float t = 0.f;
#pragma unroll
                    for (int k = 0; k < WMc; ++k) t += red[(which * WMc + k) * BN + col];
                    const int n = n0 + col;
                    const int sub = n >= p.bnb_C ? 1 : 0, per = p.N > p.bnb_C ? 2 : 1;
                    if (n < p.N) {
                        if (p.bnb_acc != nullptr) fx_atomic_add(fx_replica(p.bnb_acc, p.bnb_C, p.bnb_rep), p.bnb_C, which, n - sub * p.bnb_C, t);
                        else *(p.bnb_part + (((int64_t)(p.bnb_row0 + mblk) * per + sub) * 2 + which) * p.bnb_C + (n - sub * p.bnb_C)) = t;
                    }
                }
            }
        }
    } else {
        // bf16: transpose through LDS so that every row leaves as 16-byte pieces
        constexpr int PITCH = BN * 2 + 16;
        if constexpr (EPI == EPI_BNB && COEF_BYTES == 0) {
            // thin tile (its two stages are exactly half a CU's LDS): the coefficient table goes behind the staged tile now that the
            // ring is free -- one global load per thread under the transposition instead of 32 dependent ones in the store loop
            coef_tab = (float*)(smem + BM * PITCH);
            static_assert(4 * BN <= NT || EPI != EPI_BNB || COEF_BYTES > 0, "one table entry per thread");
            if (tid < 4 * BN) coef_v[0] = bnb_coef_at<BN, NT>(p, tid, n0);
        }
        {
            const int r = lane & 15, g = lane >> 4;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    f32x4 v = acc.a[mt][nt];
                    const int col = wcol0 + nt * 16 + 4 * g;
                    if constexpr (EPI == EPI_BNACT) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = bnact_f(p, v[j], n0 + col + j < p.N ? n0 + col + j : p.N - 1);
                    }
                    *(u32x2*)(smem + (wrow0 + mt * 16 + r) * PITCH + col * 2) = u32x2{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])};
                }
        }
        if constexpr (EPI == EPI_BNB && COEF_BYTES == 0) {
            if (tid < 4 * BN) coef_tab[tid] = coef_v[0];
        }
        __syncthreads();
        stamp(4);
        store_tile_bf16<EPI, BM, BN, NT, PITCH>(p, smem, tid, n0, mblk, [&](int row) { const int m = m0 + row; return m < p.M ? out_pixel(m) : (int64_t)-1; },
                                                EPI == EPI_BNB ? y_pre : nullptr, PREFETCH ? a_pre : nullptr, coef_tab);
    }
    if constexpr (AX) {
        // the blocks of this launch were the consumers of the previous block's accumulator: the last one to get here zeroes it

    }
    stamp(5);
}

template <typename T, int BM, int BN, int EPI, int NSTAGE, bool AX = false>
int launch_one(const IgemmParams& p, hipStream_t s) {
    const int mblocks = cdiv(p.M, BM);
    IgemmParams q = p;
    q.nblocks = cdiv(p.N, BN);
    static const bool stamp_tiles = [] { const char* e = getenv("FVA_STAMP_IGEMM"); return e && atoi(e) != 0; }();   // tools/tile_timing.py pw
    q.stamps = stamp_tiles ? fva_debug_stamps_ptr() : nullptr;
    q.stamp_rows = stamp_tiles ? fva_debug_stamps_rows() : 0;
    // stages (+ k-tile offset table, wide tiles: 8 bytes per k-tile -- or per half k-tile in half-row mode --, at least 4 KiB; a fully
    // connected layer seen as a 1x1 convolution reduces over 25088 channels = 784 fp32 k-tiles)
    constexpr int KTAB_MAX = 16384;
    const int nent = (p.halfrow ? 2 : 1) * p.ktiles;
    if (BN >= 128 && nent * 8 > KTAB_MAX) return fva_fail(FVA_ERR_ARG, "igemm: %d k-tiles exceed the offset table (%d)", p.ktiles, KTAB_MAX / 8);
    const int ktab_bytes = BN >= 128 ? (nent * 8 > 4096 ? (nent * 8 + 255) & ~255 : 4096) : 0;
    constexpr int coef_bytes = (EPI == EPI_BNB && sizeof(T) == 2 && BN >= 128) ? 4 * BN * 4 : 0;
    const int smem = NSTAGE * (BM + BN) * 128 + ktab_bytes + coef_bytes;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<T, BM, BN, EPI, NSTAGE, AX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  NSTAGE * (BM + BN) * 128 + (BN >= 128 ? KTAB_MAX : 0) + coef_bytes);
        attr_done = true;
    }
    hipLaunchKernelGGL((igemm_kernel<T, BM, BN, EPI, NSTAGE, AX>), dim3(mblocks * q.nblocks), dim3((BM / 64) * (BN / 64) * 64), smem, s, q);
    FVA_LAUNCH_CHECK("igemm_kernel");
    fva_note_kernel(AX ? (BM == 128 ? "igemm128ax" : "igemm256x64ax") : (BM == 128 ? "igemm128" : "igemm256x64"));
    return FVA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Patch convolution: the thin 3x3 stride-1 layers (32 -> 64 at 320^2, 64 -> 128 at 160^2 and their data gradients, 64 -> 32 and
// 128 -> 64: reduction channels C <= 128, outputs N <= 128, bf16).  The implicit-GEMM kernels above fetch every tap's A tile
// separately: 9 x (256 pixels x 128 B) through L2 -> LDS for 256 output pixels, and with K this short nothing amortises it -- the
// block-phase stamps (profiles/r02_tile_phases.md) put 43-75 % of a block's life into a k loop that runs at the L2 -> LDS rate
// (17-20 TB/s chip-wide), four times slower than its MFMAs.  Here a block owns a PATCH of 8 x 32 output pixels of one image: the
// (8+2) x (32+2) input pixels it needs are staged ONCE (LDS-DMA, pixel-major rows of one 64- or 32-channel slice, XOR-swizzled on
// the source side), and the nine taps are nine shifted fragment reads of that image: 1.33 instead of 9 input bytes per output
// pixel and channel.  The weights of one (tap, channel slice) are a [N][slice] matrix: all nine resident when they fit 36 KiB
// (C x N <= 2048), else streamed through two LDS slots one step ahead of the MFMAs.  4 waves, each 2 patch rows (64 pixels =
// four 16-pixel A tiles) x all N columns; epilogues, statistics and the store loop are the implicit-GEMM kernel's
// (EPI_STATS / EPI_PLAIN / EPI_BNB, tile row r <-> patch pixel (r / 32, r % 32)).  Same products in the same order per output as
// igemm_kernel with tap-major k order -- but slice-major there (ktaps), so results agree to fp32 summation order, not bit for bit.
template <int CS, int BN, int NSL, int EPI>
__global__ __launch_bounds__(256, 2) void pconv_kernel(const IgemmParams p) {
    constexpr int TH = 8, TW = 32, PW = TW + 2, NPIX = (TH + 2) * PW;             // 340 patch pixels
    constexpr int PIXB = CS * 2;                                                   // bytes of a pixel row in LDS: 128 / 64
    constexpr int CPP = PIXB / 16;                                                 // 16-byte chunks per pixel: 8 / 4
    constexpr int PPI = 1024 / PIXB;                                               // pixels per LDS-DMA instruction: 8 / 16
    constexpr int PATCH_INSTR = (NPIX + PPI - 1) / PPI, PATCH_BYTES = PATCH_INSTR * 1024;
    constexpr int WSTEP_BYTES = BN * PIXB, WPW = WSTEP_BYTES / 1024 / 4;           // one (tap, slice) matrix; its DMA instructions per wave
    static_assert(WSTEP_BYTES % 4096 == 0, "whole DMA instructions per wave");
    constexpr int STEPS = 9 * NSL;
    constexpr bool RESIDENT = NSL == 1 && 9 * WSTEP_BYTES <= 36 * 1024;
    constexpr int WBUF_BYTES = (RESIDENT ? 9 : 2) * WSTEP_BYTES;
    constexpr int BM = 256, NT = 256, NTILE = BN / 16, KS = CS / 32;
    constexpr int PITCH = BN * 2 + 16, TILE_BYTES = BM * PITCH;
    // resident weights live across the block's tiles: the epilogue's tile may alias the patch, not them; a ring is re-streamed per tile
    constexpr int WOFF = RESIDENT && TILE_BYTES > PATCH_BYTES ? TILE_BYTES : PATCH_BYTES;
    constexpr int COEF0 = TILE_BYTES > WOFF + WBUF_BYTES ? TILE_BYTES : WOFF + WBUF_BYTES;   // BNB coefficient table: clear of both uses
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const patch = smem;
    char* const wbuf = smem + WOFF;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, xq = nwg >> 3, xr = nwg & 7;
    const int tile0 = ((xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3)) * (RESIDENT ? p.pt_tpb : 1);
    const int Hp = p.in_img / p.in_row, Wp = p.in_row;
    float* coef_tab = (float*)(smem + COEF0);
    if constexpr (EPI == EPI_BNB) bnb_fill_lds<BN, NT>(p, coef_tab, tid, 0);   // before the first LDS-DMA; read in the epilogue
    int b = 0, ty = 0, tx = 0, yorg = 0, xorg = 0;                     // the current tile (set at the top of the tile loop)

    // swizzle of a row (pixel or weight row) of the LDS images: chunk c of row r sits in slot c ^ swz(r)
    auto swz = [](int r) { return CS == 64 ? (r >> 1) & 7 : (-(r >> 2)) & 3; };
    auto stage_patch = [&](int slice) {
        const bf16_t* src0 = (const bf16_t*)p.in + (int64_t)b * p.in_img * p.C + slice * CS;
#pragma unroll
        for (int i = 0; i < (PATCH_INSTR + 3) / 4; ++i) {
            const int j = i * 4 + w;                                   // wave-uniform
            if (j < PATCH_INSTR) {
                int pix = j * PPI + lane / CPP;
                pix = pix < NPIX ? pix : NPIX - 1;
                const int py = (pix * 1928) >> 16, px = pix - py * PW;             // / 34, exact below 400
                int iy = yorg + py, ix = xorg + px;
                iy = iy < Hp ? iy : Hp - 1;                                        // tiles that overhang the image: rows / columns that are
                ix = ix < Wp ? ix : Wp - 1;                                        // never stored read a valid pixel
                const int chunk = (lane % CPP) ^ swz(j * PPI + lane / CPP);
                __builtin_amdgcn_global_load_lds(GLB_PTR(src0 + ((int64_t)iy * Wp + ix) * p.C + chunk * 8), LDS_PTR(patch + j * 1024), 16, 0, 0);
            }
        }
    };
    auto stage_w = [&](int step, char* dst) {                          // step = slice * 9 + tap
        const int slice = step / 9, tap = step - slice * 9;
        int tw = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i)
            if (i == tap) tw = p.tap_w[i];
        const bf16_t* src0 = (const bf16_t*)p.wt + (int64_t)tw * p.N * p.C + slice * CS;
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int j = i * 4 + w;
            const int n = j * PPI + lane / CPP;                        // < BN (whole instructions)
            const int nn = n < p.N ? n : p.N - 1;
            const int chunk = (lane % CPP) ^ swz(n);
            __builtin_amdgcn_global_load_lds(GLB_PTR(src0 + (int64_t)nn * p.C + chunk * 8), LDS_PTR(dst + j * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][NTILE];
    // this lane's A rows: wave w owns patch rows 2w, 2w + 1; tile mt = row (mt >> 1), x half (mt & 1); tap (dy, dx) adds dy * PW + dx
    int pix0[4];
    int opq = 0;   // re-made opaque in every tile: the 72 + fragment addresses of a tile are then computed where they are used, not
                   // hoisted out of the tile loop (as loop invariants they cost 150-280 spilled registers)
    auto compute = [&](int tap, const char* wb) {
        const int tapoff = (tap / 3) * PW + (tap % 3);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 af[4], bfr[NTILE];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int pix = pix0[mt] + tapoff;
                af[mt] = *(const bf16x8*)(patch + pix * PIXB + (((ks * 4 + g) ^ swz(pix)) << 4));
            }
#pragma unroll
            for (int nt = 0; nt < NTILE; ++nt) {
                const int n = nt * 16 + r16 + opq;
                bfr[nt] = *(const bf16x8*)(wb + n * PIXB + (((ks * 4 + g) ^ swz(n)) << 4));
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTILE; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[mt], acc[mt][nt], 0, 0, 0);   // swapped: foreach_acc
        }
    };

    if constexpr (RESIDENT) {
#pragma unroll
        for (int t = 0; t < 9; ++t) stage_w(t, wbuf + t * WSTEP_BYTES);
    }
    // only the resident-weight forms loop over tiles (a ring re-streams its weights per tile anyway, and inside a loop its per-step
    // staging addresses are hoisted as loop invariants: 120-240 spilled registers)
#pragma unroll 1
    for (int it = 0; it < (RESIDENT ? p.pt_tpb : 1); ++it) {
    const int tile = tile0 + it;
    if (tile >= p.pt_tiles) break;                                     // block-uniform
    b = (int)fd_div((uint32_t)tile, p.pt_div_img);
    {
        const int trem = tile - b * (p.pt_tx * p.pt_ty);
        ty = (int)fd_div((uint32_t)trem, p.pt_div_tx);
        tx = trem - ty * p.pt_tx;
    }
    yorg = p.y0 + ty * TH;                                             // patch origin in the padded input
    xorg = p.x0 + tx * TW;
    asm volatile("" : "+v"(opq));
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) pix0[mt] = (2 * w + (mt >> 1)) * PW + (mt & 1) * 16 + r16 + opq;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTILE; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (RESIDENT) {
        stage_patch(0);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int t = 0; t < 9; ++t) compute(t, wbuf + t * WSTEP_BYTES);
    } else {
        // two weight slots: the matrix of step s + 1 is issued right after the barrier of step s, into the slot every wave has just
        // finished reading; a new channel slice re-stages the patch at its first step (exposed once per tile: LDS holds one patch)
        stage_patch(0);
        stage_w(0, wbuf);
#pragma unroll 1
        for (int sl = 0; sl < NSL; ++sl) {                             // a runtime loop: unrolled, the 18 steps' addresses spill
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int s = sl * 9 + t;
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                if (NSL > 1 && t == 0 && sl > 0) {
                    stage_patch(sl);
                    stage_w(s + 1, wbuf + ((s + 1) & 1) * WSTEP_BYTES);
                    wait_vmcnt<WPW>();                                 // the patch has landed, the next step's weights may still fly
                    __builtin_amdgcn_s_barrier();
                } else if (s + 1 < STEPS) {
                    stage_w(s + 1, wbuf + ((s + 1) & 1) * WSTEP_BYTES);
                }
                compute(t, wbuf + (s & 1) * WSTEP_BYTES);
            }
        }
    }
    __syncthreads();   // LDS is free for the epilogue

    // accumulator tile (mt, nt) of lane (r16, g) (MFMA operands swapped, see foreach_acc): tile row = patch pixel
    // (2w + (mt >> 1), (mt & 1) * 16 + r16), columns nt * 16 + 4 g .. + 3
    const bool full = (ty + 1) * TH <= p.pt_H && (tx + 1) * TW <= p.pt_W;
    auto row_ok = [&](int rr) { return ty * TH + (rr >> 5) < p.pt_H && tx * TW + (rr & 31) < p.pt_W; };
    if constexpr (EPI == EPI_STATS) {
        if (p.stats != nullptr || p.stats_acc != nullptr) {
            f32x4 s1[NTILE], s2[NTILE];
#pragma unroll
            for (int i = 0; i < NTILE; ++i) s1[i] = s2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bool ok = full || row_ok((2 * w + (mt >> 1)) * 32 + (mt & 1) * 16 + r16);
#pragma unroll
                for (int nt = 0; nt < NTILE; ++nt) {
                    const f32x4 v = ok ? acc[mt][nt] : f32x4{0.f, 0.f, 0.f, 0.f};
                    s1[nt] += v;
                    s2[nt] += v * v;
                }
            }
            // the 16 lane rows and the four waves through LDS, added by the column's thread in a fixed order (igemm_kernel's scheme)
            float* red = (float*)smem;  // [2][4 waves][16][RP]: below the tile's end, clear of resident weights
            constexpr int RP = BN + 4;
            static_assert(2 * 4 * 16 * RP * 4 <= TILE_BYTES, "the statistics scratch stays inside the epilogue tile's bytes");
#pragma unroll
            for (int nt = 0; nt < NTILE; ++nt) {
                *(f32x4*)(red + ((0 * 4 + w) * 16 + r16) * RP + nt * 16 + 4 * g) = s1[nt];
                *(f32x4*)(red + ((1 * 4 + w) * 16 + r16) * RP + nt * 16 + 4 * g) = s2[nt];
            }
            __syncthreads();
            if (w < BN / 32) {
                const int which = lane >> 5, col = w * 32 + (lane & 31);
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 64; ++k) t += red[(which * 64 + k) * RP + col];
                if (col < p.N) {
                    if (p.stats_acc != nullptr) fx_atomic_add(fx_replica(p.stats_acc, p.N, p.acc_rep), p.N, which, col, t);
                    else *(p.stats + ((int64_t)tile * 2 + which) * p.N + col) = t;
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTILE; ++nt) {
            const f32x4 v = acc[mt][nt];
            *(u32x2*)(smem + ((2 * w + (mt >> 1)) * 32 + (mt & 1) * 16 + r16) * PITCH + (nt * 16 + 4 * g) * 2) = u32x2{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])};
        }
    __syncthreads();
    store_tile_bf16<EPI, BM, BN, NT, PITCH>(p, smem, tid, 0, tile, [&](int rr) -> int64_t {
        const int oy = ty * TH + (rr >> 5), ox = tx * TW + (rr & 31);
        return oy < p.pt_H && ox < p.pt_W ? ((int64_t)b * p.pt_H + oy) * p.pt_W + ox : (int64_t)-1;
    }, nullptr, nullptr, EPI == EPI_BNB ? coef_tab : nullptr);
    __syncthreads();   // the next tile's patch lands where this tile's epilogue read
    }
}

// Layers the patch kernel serves (bf16, 3x3, stride 1): reduction channels C in {32, 64, 128}, outputs N in {32, 64, 128}, at most
// C x N = 8192, on maps of at least 64 x 64 (below that the 8 x 32 patches overhang too much).  FVA_PCONV=0 switches it off.
int g_pconv = -1;   // -1: read FVA_PCONV on first use; fva_conv_patch_kernel() sets it
inline bool pconv_enabled() {
    if (g_pconv < 0) {
        const char* e = getenv("FVA_PCONV");
        g_pconv = (!e || atoi(e) != 0) ? 1 : 0;
    }
    return g_pconv != 0;
}
inline bool use_pconv(int dtype, int ksize, int stride, int C, int N, int H, int W) {
    if (!pconv_enabled() || dtype != FVA_BF16 || ksize != 3 || stride != 1 || H < 64 || W < 64) return false;
    return (C == 32 && N == 64) || (C == 64 && N == 128) || (C == 64 && N == 32) || (C == 128 && N == 64) || (C == 64 && N == 64);
}
inline int pconv_tiles(int B, int H, int W) { return B * cdiv(H, 8) * cdiv(W, 32); }

template <int CS, int BN, int NSL, int EPI>
int launch_pconv_one(const IgemmParams& p0, int tiles, hipStream_t s) {
    constexpr int PIXB = CS * 2, PPI = 1024 / PIXB, PATCH_BYTES = ((340 + PPI - 1) / PPI) * 1024, WSTEP = BN * PIXB;
    constexpr bool RESIDENT = NSL == 1 && 9 * WSTEP <= 36 * 1024;
    constexpr int WBUF = (RESIDENT ? 9 : 2) * WSTEP, TILE = 256 * (BN * 2 + 16);
    constexpr int WOFF = RESIDENT && TILE > PATCH_BYTES ? TILE : PATCH_BYTES;
    constexpr int smem = (TILE > WOFF + WBUF ? TILE : WOFF + WBUF) + (EPI == EPI_BNB ? 4 * BN * 4 : 0);
    static_assert(smem <= 80 * 1024, "two blocks per CU");
    // consecutive tiles per block: resident weights (36 KiB) cost more L2 -> LDS bytes than the patch itself when re-staged per tile
    static const int tpb_env = [] { const char* e = getenv("FVA_PCONV_TPB"); return e ? atoi(e) : 0; }();
    IgemmParams p = p0;
    p.pt_tiles = tiles;
    p.pt_tpb = RESIDENT ? (tpb_env > 0 ? tpb_env : 2) : 1;     // measured 1 / 2 / 4 / 8 on 32 -> 64 at 320^2: fwd 253 / 235 / 260 / 261 us, dgrad 218 / 209 / 214 / 224
    const int blocks = cdiv(tiles, p.pt_tpb);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)pconv_kernel<CS, BN, NSL, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    hipLaunchKernelGGL((pconv_kernel<CS, BN, NSL, EPI>), dim3(blocks), dim3(256), smem, s, p);
    FVA_LAUNCH_CHECK("pconv_kernel");
    fva_note_kernel("pconv");
    return FVA_OK;
}

// p is set up as for igemm (in / wt / out / N / C / in_row / in_img / y0 / x0 / epilogue fields); H, W = output image; flip: data gradient
template <int EPI>
int launch_pconv(const IgemmParams& p0, int B, int H, int W, bool flip, hipStream_t s) {
    IgemmParams p = p0;
    p.pt_H = H;
    p.pt_W = W;
    p.pt_tx = cdiv(W, 32);
    p.pt_ty = cdiv(H, 8);
    p.pt_div_tx = make_fastdiv(p.pt_tx);
    p.pt_div_img = make_fastdiv(p.pt_tx * p.pt_ty);
    for (int t = 0; t < 9; ++t) p.tap_w[t] = flip ? 8 - t : t;     // patch tap (dy, dx) <-> filter tap: the data gradient flips both axes
    const int tiles = pconv_tiles(B, H, W);
    if (p.C == 32 && p.N == 64) return launch_pconv_one<32, 64, 1, EPI>(p, tiles, s);
    if (p.C == 64 && p.N == 32) return launch_pconv_one<64, 32, 1, EPI>(p, tiles, s);
    if (p.C == 64 && p.N == 64) return launch_pconv_one<64, 64, 1, EPI>(p, tiles, s);
    if (p.C == 64 && p.N == 128) return launch_pconv_one<64, 128, 1, EPI>(p, tiles, s);
    if (p.C == 128 && p.N == 64) return launch_pconv_one<64, 64, 2, EPI>(p, tiles, s);
    return fva_fail(FVA_ERR_ARG, "pconv: unsupported channels %d -> %d", p.C, p.N);
}

// ---------------------------------------------------------------------------------------------------------------------
// Patch form of the STRIDE-2 data gradient of the thin 3x3 layers (64 -> 32 channels at 640^2, 128 -> 64 at 320^2; bf16).
//   dx[2i + py][2j + px] = sum over the taps (kh, kw) with kh = 1 (py = 0) or kh in {0, 2} (py = 1), likewise kw / px, of
//                          dy[i + (kh == 0)][j + (kw == 0)] . W[kh][kw]
// Until round 3 these ran as two "paired" implicit-GEMM launches (both x parities as N' = 2 Cin columns: 4/3 of the MACs, every
// tap's dy tile fetched again) -- 505 / 330 us for 1.26 / 1.05 GB of HBM traffic.  Here a block owns 4 x 32 cells = 8 x 64 dx
// pixels: the (4+1) x (32+1) dy pixels they need are staged once (21 KiB per 64-channel slice), every tap is ONE shifted fragment
// read feeding exactly the parity class it belongs to (the exact 9/4 MACs per dx pixel), wave w owns cell row w with eight
// accumulator groups (4 parity classes x 2 halves of the row).  Weights: the plain [9][Cin][Cout] dgrad layout, resident (36 KiB)
// or streamed through two slots like pconv_kernel; epilogue / statistics / store loop are the shared ones (512-row tile).
template <int CS, int BN, int NSL, int EPI>
__global__ __launch_bounds__(256, 2) void pdgrad2_kernel(const IgemmParams p) {
    constexpr int CH = 4, CW = 32, PW = CW + 1, NPIX = (CH + 1) * PW;             // 165 dy pixels
    constexpr int PIXB = CS * 2, CPP = PIXB / 16, PPI = 1024 / PIXB;
    static_assert(CS == 64, "64-channel slices (128-byte pixel rows)");
    constexpr int PATCH_INSTR = (NPIX + PPI - 1) / PPI, PATCH_BYTES = PATCH_INSTR * 1024;
    constexpr int WSTEP_BYTES = BN * PIXB, WPW = WSTEP_BYTES / 1024 / 4;
    static_assert(WSTEP_BYTES % 4096 == 0, "whole DMA instructions per wave");
    constexpr int STEPS = 9 * NSL;
    constexpr bool RESIDENT = NSL == 1 && 9 * WSTEP_BYTES <= 36 * 1024;
    constexpr int WBUF_BYTES = (RESIDENT ? 9 : 2) * WSTEP_BYTES;
    constexpr bool LOOPED = RESIDENT && EPI != EPI_BNB;   // the fused-statistics epilogue inside the tile loop spills (35 registers): one tile per block there
    constexpr int BM = 512, NT = 256, NTILE = BN / 16, KS = CS / 32;
    constexpr int PITCH = BN * 2 + 16, TILE_BYTES = BM * PITCH;
    constexpr int WOFF = RESIDENT && TILE_BYTES > PATCH_BYTES ? TILE_BYTES : PATCH_BYTES;
    constexpr int COEF0 = TILE_BYTES > WOFF + WBUF_BYTES ? TILE_BYTES : WOFF + WBUF_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const patch = smem;
    char* const wbuf = smem + WOFF;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, xq = nwg >> 3, xr = nwg & 7;
    const int tile0 = ((xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3)) * (LOOPED ? p.pt_tpb : 1);
    const int Hp = p.in_img / p.in_row, Wp = p.in_row;
    float* coef_tab = (float*)(smem + COEF0);
    if constexpr (EPI == EPI_BNB) bnb_fill_lds<BN, NT>(p, coef_tab, tid, 0);
    int b = 0, ty = 0, tx = 0, yorg = 0, xorg = 0;

    auto swz = [](int r) { return (r >> 1) & 7; };
    auto stage_patch = [&](int slice) {
        const bf16_t* src0 = (const bf16_t*)p.in + (int64_t)b * p.in_img * p.C + slice * CS;
#pragma unroll
        for (int i = 0; i < (PATCH_INSTR + 3) / 4; ++i) {
            const int j = i * 4 + w;
            if (j < PATCH_INSTR) {
                int pix = j * PPI + lane / CPP;
                pix = pix < NPIX ? pix : NPIX - 1;
                const int py = (pix * 1986) >> 16, px = pix - py * PW;             // / 33, exact below 400
                int iy = yorg + py, ix = xorg + px;
                iy = iy < Hp ? iy : Hp - 1;
                ix = ix < Wp ? ix : Wp - 1;
                const int chunk = (lane % CPP) ^ swz(j * PPI + lane / CPP);
                __builtin_amdgcn_global_load_lds(GLB_PTR(src0 + ((int64_t)iy * Wp + ix) * p.C + chunk * 8), LDS_PTR(patch + j * 1024), 16, 0, 0);
            }
        }
    };
    auto stage_w = [&](int step, char* dst) {                          // step = slice * 9 + tap (tap = kh * 3 + kw)
        const int slice = step / 9, tap = step - slice * 9;
        const bf16_t* src0 = (const bf16_t*)p.wt + (int64_t)tap * p.N * p.C + slice * CS;
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int j = i * 4 + w;
            const int n = j * PPI + lane / CPP;
            const int nn = n < p.N ? n : p.N - 1;
            const int chunk = (lane % CPP) ^ swz(n);
            __builtin_amdgcn_global_load_lds(GLB_PTR(src0 + (int64_t)nn * p.C + chunk * 8), LDS_PTR(dst + j * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][2][NTILE];            // [parity class py * 2 + px][half of the cell row][n tile]
    int pix0[2];
    int opq = 0;
    auto compute = [&](int tap, const char* wb) {
        const int kh = tap / 3, kw = tap % 3;
        const int cls = (kh == 1 ? 0 : 2) + (kw == 1 ? 0 : 1);
        const int tapoff = (kh == 0 ? PW : 0) + (kw == 0 ? 1 : 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 af[2], bfr[NTILE];
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) {
                const int pix = pix0[xt] + tapoff;
                af[xt] = *(const bf16x8*)(patch + pix * PIXB + (((ks * 4 + g) ^ swz(pix)) << 4));
            }
#pragma unroll
            for (int nt = 0; nt < NTILE; ++nt) {
                const int n = nt * 16 + r16 + opq;
                bfr[nt] = *(const bf16x8*)(wb + n * PIXB + (((ks * 4 + g) ^ swz(n)) << 4));
            }
#pragma unroll
            for (int xt = 0; xt < 2; ++xt)
#pragma unroll
                for (int nt = 0; nt < NTILE; ++nt) acc[cls][xt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt], af[xt], acc[cls][xt][nt], 0, 0, 0);   // swapped: foreach_acc
        }
    };

    if constexpr (RESIDENT) {
#pragma unroll
        for (int t = 0; t < 9; ++t) stage_w(t, wbuf + t * WSTEP_BYTES);
    }
#pragma unroll 1
    for (int it = 0; it < (LOOPED ? p.pt_tpb : 1); ++it) {
    const int tile = tile0 + it;
    if (tile >= p.pt_tiles) break;
    b = (int)fd_div((uint32_t)tile, p.pt_div_img);
    {
        const int trem = tile - b * (p.pt_tx * p.pt_ty);
        ty = (int)fd_div((uint32_t)trem, p.pt_div_tx);
        tx = trem - ty * p.pt_tx;
    }
    yorg = p.y0 + ty * CH;                                             // first dy pixel of the patch in the padded dy buffer
    xorg = p.x0 + tx * CW;
    asm volatile("" : "+v"(opq));
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) pix0[xt] = w * PW + xt * 16 + r16 + opq;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NTILE; ++j) acc[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (RESIDENT) {
        stage_patch(0);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int t = 0; t < 9; ++t) compute(t, wbuf + t * WSTEP_BYTES);
    } else {
        stage_patch(0);
        stage_w(0, wbuf);
#pragma unroll 1
        for (int sl = 0; sl < NSL; ++sl) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int s = sl * 9 + t;
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                if (NSL > 1 && t == 0 && sl > 0) {
                    stage_patch(sl);
                    stage_w(s + 1, wbuf + ((s + 1) & 1) * WSTEP_BYTES);
                    wait_vmcnt<WPW>();
                    __builtin_amdgcn_s_barrier();
                } else if (s + 1 < STEPS) {
                    stage_w(s + 1, wbuf + ((s + 1) & 1) * WSTEP_BYTES);
                }
                compute(t, wbuf + (s & 1) * WSTEP_BYTES);
            }
        }
    }
    __syncthreads();   // LDS is free for the epilogue
    // accumulator element (class (py, px), half xt, j) = dx pixel (2 w + py, 2 (16 xt + 4 g + j) + px) of the 8 x 64 tile
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int xt = 0; xt < 2; ++xt)
#pragma unroll
            for (int nt = 0; nt < NTILE; ++nt) {   // MFMA operands swapped (foreach_acc): lane (r16, g) holds cell xt * 16 + r16, channels nt * 16 + 4 g .. + 3
                const f32x4 v = acc[c][xt][nt];
                *(u32x2*)(smem + ((2 * w + (c >> 1)) * 64 + 2 * (xt * 16 + r16) + (c & 1)) * PITCH + (nt * 16 + 4 * g) * 2) = u32x2{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])};
            }
    __syncthreads();
    store_tile_bf16<EPI, BM, BN, NT, PITCH>(p, smem, tid, 0, tile, [&](int rr) -> int64_t {
        const int oy = ty * 2 * CH + (rr >> 6), ox = tx * 2 * CW + (rr & 63);
        return oy < p.pt_H && ox < p.pt_W ? ((int64_t)b * p.pt_H + oy) * p.pt_W + ox : (int64_t)-1;
    }, nullptr, nullptr, EPI == EPI_BNB ? coef_tab : nullptr);
    __syncthreads();
    }
}

// stride-2 3x3 bf16 data gradients served by pdgrad2_kernel: reduction over C = Cout channels, N = Cin outputs.  Decided by the
// channel counts alone (the weight packer must know the layout without seeing a feature-map size: these layers take the plain
// [9][Cin][Cout] dgrad weights, the other thin stride-2 layers the paired layout).  FVA_PDGRAD2=0 switches it off.
inline bool use_pdgrad2(int dtype, int ksize, int stride, int C, int N) {
    static const bool on = [] { const char* e = getenv("FVA_PDGRAD2"); return !e || atoi(e) != 0; }();
    return on && dtype == FVA_BF16 && ksize == 3 && stride == 2 && ((C == 64 && N == 32) || (C == 128 && N == 64));
}
inline int pdgrad2_tiles(int B, int H, int W) { return B * cdiv(H, 8) * cdiv(W, 64); }   // H, W: the dx image

template <int CS, int BN, int NSL, int EPI>
int launch_pdgrad2_one(const IgemmParams& p0, int tiles, hipStream_t s) {
    constexpr int PIXB = CS * 2, PPI = 1024 / PIXB, PATCH_BYTES = ((165 + PPI - 1) / PPI) * 1024, WSTEP = BN * PIXB;
    constexpr bool RESIDENT = NSL == 1 && 9 * WSTEP <= 36 * 1024;
    constexpr int WBUF = (RESIDENT ? 9 : 2) * WSTEP, TILE = 512 * (BN * 2 + 16);
    constexpr int WOFF = RESIDENT && TILE > PATCH_BYTES ? TILE : PATCH_BYTES;
    constexpr int smem = (TILE > WOFF + WBUF ? TILE : WOFF + WBUF) + (EPI == EPI_BNB ? 4 * BN * 4 : 0);
    static_assert(smem <= 80 * 1024, "two blocks per CU");
    static const int tpb_env = [] { const char* e = getenv("FVA_PCONV_TPB"); return e ? atoi(e) : 0; }();
    IgemmParams p = p0;
    p.pt_tiles = tiles;
    p.pt_tpb = RESIDENT && EPI != EPI_BNB ? (tpb_env > 0 ? tpb_env : 2) : 1;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)pdgrad2_kernel<CS, BN, NSL, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    hipLaunchKernelGGL((pdgrad2_kernel<CS, BN, NSL, EPI>), dim3(cdiv(tiles, p.pt_tpb)), dim3(256), smem, s, p);
    FVA_LAUNCH_CHECK("pdgrad2_kernel");
    fva_note_kernel("pdgrad2");
    return FVA_OK;
}

// p: in = dy halo buffer (in_row / in_img / y0 = x0 = dy_pad), wt = [9][Cin][Cout], out = dx dense, N = Cin, C = Cout; H, W = dx image
template <int EPI>
int launch_pdgrad2(const IgemmParams& p0, int B, int H, int W, hipStream_t s) {
    IgemmParams p = p0;
    p.pt_H = H;
    p.pt_W = W;
    p.pt_tx = cdiv(W, 64);
    p.pt_ty = cdiv(H, 8);
    p.pt_div_tx = make_fastdiv(p.pt_tx);
    p.pt_div_img = make_fastdiv(p.pt_tx * p.pt_ty);
    const int tiles = pdgrad2_tiles(B, H, W);
    if (p.C == 64 && p.N == 32) return launch_pdgrad2_one<64, 32, 1, EPI>(p, tiles, s);
    if (p.C == 128 && p.N == 64) return launch_pdgrad2_one<64, 64, 2, EPI>(p, tiles, s);
    return fva_fail(FVA_ERR_ARG, "pdgrad2: unsupported channels %d -> %d", p.C, p.N);
}

// ---------------------------------------------------------------------------------------------------------------------
// 256x256 tile, 8 waves (2 along m x 4 along n, 128x64 each), the "8-phase" schedule of the CDNA GEMM playbook
// (cdna_hip_programming.md section 5): one block per CU, two LDS buffers of four 16-KiB half-tiles (A rows / B rows of the
// first and second half of every wave's sub-tile), a k-tile consumed in four phases of 16 MFMAs (one 64x32 quadrant of
// the wave's accumulator each), one half-tile of LDS-DMA issued per phase, counted vmcnt once per k-tile, and the two
// wave groups (wr = 0 / 1) running half a phase apart so that one group's MFMAs cover the other group's LDS reads.
//
// Hazards (phase p of k-tile t = q_p(t); "stage X(t)" = LDS-DMA of half-tile X of k-tile t into buffer t & 1):
//   q0: read B-h0, A-h0 | stage A-h1(t+1)      q1: read B-h1 | stage B-h0(t+2)
//   q2: read A-h1       | stage A-h0(t+2)      q3: no reads  | stage B-h1(t+2), then vmcnt(6) (everything up to
//                                                              A-h1(t+1) has landed => k-tile t+1 complete)
//   RAW: the wait precedes q3's first barrier, the first read of k-tile t+1 is in q0(t+1), behind q3's second barrier,
//        which the other group only reaches after its own wait.
//   WAR: every phase retires its LDS reads (lgkmcnt(0)) before its first barrier; a half-tile is re-staged at the earliest
//        one phase after its last read, so the re-staging wave has passed a barrier that every reader reached after
//        retiring (B-h0: read q0, staged q1; A-h0: q0 -> q2; B-h1: q1 -> q3; A-h1: q2 -> q0 of the next k-tile).
struct Igemm8Diag {
    long long* stamps;   // diagnostic (fva_conv_debug_stamps): [stamp_rows][8], see the kernel's stamp()
    int stamp_rows;      // blocks beyond the caller's buffer do not stamp
};
// (Round 1-3 experiments on this kernel that lost and were removed in round 4, numbers in profiles/r03_experiments.md and
// profiles/HISTORY.md: a stream-K form with fp32 slab hand-off, a 224-row tile, a start skew between the halves of an XCD.)

#ifndef FVA_EPI_STAMPS
#define FVA_EPI_STAMPS 0     // 1: the four diagnostic stamps bracket the epilogue's phases instead of the tile's (tools/tile_timing.py)
#endif
template <int EPI>
__global__ __launch_bounds__(512) void igemm8_kernel(const IgemmParams p, const Igemm8Diag sk) {
    constexpr int MT = 8;                             // 16-row accumulator tiles per wave along m
    constexpr int WM = 16 * MT;                       // rows of a wave row
    constexpr int BM = 2 * WM, BN = 256, NT = 512, BK = 64;
    constexpr int HALF = 128 * 128, BUF = 4 * HALF;   // bytes
    constexpr int EPI_BYTES = BM * (BN * 2 + 16);     // the epilogue's transposed bf16 tile
    constexpr int TAB0 = EPI_BYTES > 2 * BUF ? EPI_BYTES : 2 * BUF;   // tables that live across tiles: behind the staging buffers AND the epilogue tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 2, wc = w & 3;
    auto stamp = [&](int i) {   // per block: 4 wall-clock stamps, then the shader-clock cycle counter at the same points (ratio = the clock under load)
        if (sk.stamps && tid == 0 && (int)blockIdx.x < sk.stamp_rows) {
            sk.stamps[(int64_t)blockIdx.x * 8 + i] = wall_clock64();
            sk.stamps[(int64_t)blockIdx.x * 8 + 4 + i] = clock64();
        }
    };
#if !FVA_EPI_STAMPS
    stamp(0);
#endif

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, xq = nwg >> 3, xr = nwg & 7;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int KT = p.ktiles;

    int* ktab = (int*)(smem + TAB0);
    for (int e = tid; e < KT; e += NT) {
        int t, kk;
        if (p.ktaps) { kk = e / p.ktaps; t = e - kk * p.ktaps; }
        else { t = e / p.kt_per_tap; kk = e - t * p.kt_per_tap; }
        int tp = 0, tw = 0;
#pragma unroll
        for (int i = 0; i < MAX_TAPS; ++i)
            if (i == t) { tp = p.tap_pix[i]; tw = p.tap_w[i]; }
        ktab[2 * e] = tp * p.C + kk * BK;
        ktab[2 * e + 1] = tw * p.N * p.C + kk * BK;
    }

    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ (((w & 1) << 2) + (lane >> 4));   // source chunk landing at LDS slot (lane & 7)
    const int r = lane & 15, g = lane >> 4, sr = (r >> 1) & 7;
    const int co0 = (g ^ sr) << 4, co1 = ((4 + g) ^ sr) << 4;        // byte offsets of the two k-steps' 16-B chunks
    const int a_row = (wr * 64 + r) * 128, b_row = (wc * 32 + r) * 128;

    {
        const int tile = logical, kt0 = 0, kt1 = KT;
        const int mblk = tile / p.nblocks, nblk = tile - mblk * p.nblocks;
        const int m0 = mblk * BM, n0 = nblk * BN;
        __syncthreads();   // ktab is written / the previous tile's epilogue is done with LDS
        float* coef_tab = (float*)(smem + TAB0 + 4096);
        if constexpr (EPI == EPI_BNB) bnb_fill_lds<BN, NT>(p, coef_tab, tid, n0);   // read in the epilogue, many barriers later

        // ---- per-lane source rows of the LDS-DMA pieces: piece (i, w) of a half-tile = its rows (i*8 + w)*8 .. +8 ---
        // (32-bit byte offsets from the tensor bases: the host checks that both operands are smaller than 2 GiB)
        uint32_t a_off[2][2], b_off[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rh = (i * 8 + w) * 8 + lrow;                   // row inside the half-tile, 0..127
                int m = m0 + (rh >> 6) * WM + h * 64 + (rh & 63);
                m = m < p.M ? m : p.M - 1;
                const uint32_t b = fd_div((uint32_t)m, p.div_ohw);
                const uint32_t rem = (uint32_t)m - b * (uint32_t)p.OHW;
                const uint32_t oy = fd_div(rem, p.div_ow);
                const uint32_t ox = rem - oy * (uint32_t)p.OW;
                const uint32_t pix = b * (uint32_t)p.in_img + (oy * p.sy + p.y0) * (uint32_t)p.in_row + (ox * p.sx + p.x0);
                a_off[h][i] = (pix * (uint32_t)p.C + chunk * 8) * 2;
                int n = n0 + (rh >> 5) * 64 + h * 32 + (rh & 31);
                n = n < p.N ? n : p.N - 1;
                b_off[h][i] = ((uint32_t)n * (uint32_t)p.C + chunk * 8) * 2;
            }

        // kind: 0 A-h0, 1 A-h1, 2 B-h0, 3 B-h1; the LDS buffer alternates with the k-tile
        auto stage = [&](int kind, int kt, int koff) {
            char* dst = smem + ((kt - kt0) & 1) * BUF + kind * HALF + w * 1024;
            const uint32_t off = (uint32_t)koff * 2;   // tap offsets may be "negative": wraps in 32 bits
            const int h = kind & 1;
            if (kind < 2) {
                __builtin_amdgcn_global_load_lds(GLB_PTR((const char*)p.in + (uint32_t)(a_off[h][0] + off)), LDS_PTR(dst), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLB_PTR((const char*)p.in + (uint32_t)(a_off[h][1] + off)), LDS_PTR(dst + 8192), 16, 0, 0);
            } else {
                __builtin_amdgcn_global_load_lds(GLB_PTR((const char*)p.wt + (uint32_t)(b_off[h][0] + off)), LDS_PTR(dst), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLB_PTR((const char*)p.wt + (uint32_t)(b_off[h][1] + off)), LDS_PTR(dst + 8192), 16, 0, 0);
            }
        };

        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);   // keep the clear ahead of the prologue's LDS-DMA (no register wait after it)

        bf16x8 af[4][2], b0[2][2], b1[2][2];
        auto read_a = [&](const char* buf, int h) {
            const char* q = buf + h * HALF + a_row;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                af[mt][0] = *(const bf16x8*)(q + mt * 2048 + co0);
                af[mt][1] = *(const bf16x8*)(q + mt * 2048 + co1);
            }
        };
        auto read_b = [&](const char* buf, int h, bf16x8 (&bf)[2][2]) {
            const char* q = buf + (2 + h) * HALF + b_row;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                bf[nt][0] = *(const bf16x8*)(q + nt * 2048 + co0);
                bf[nt][1] = *(const bf16x8*)(q + nt * 2048 + co1);
            }
        };
        auto mma = [&](int ha, int hb, bf16x8 (&bf)[2][2]) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[ha * 4 + mt][hb * 2 + nt] =      // operands swapped: D = (B^T A^T), a lane's four accumulators are four adjacent COLUMNS
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[nt][ks], af[mt][ks], acc[ha * 4 + mt][hb * 2 + nt], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        auto retire_reads_then_barrier = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };

        // ---- prologue: first k-tile complete, three half-tiles of the second in flight ----------------------------
        // the offset table is read one phase ahead of its use (kA1: A of k-tile t+1; kA2 / kB2: A / B of k-tile t+2), so that a
        // phase's LDS-DMA does not queue behind a fresh LDS read
        auto kofs = [&](int kt, int ab) { return ktab[2 * (kt < KT ? kt : KT - 1) + ab]; };
        {
            const int a0 = kofs(kt0, 0), b0k = kofs(kt0, 1);
            stage(2, kt0, b0k); stage(0, kt0, a0); stage(3, kt0, b0k); stage(1, kt0, a0);
        }
        int kA1 = kofs(kt0 + 1, 0), kA2 = kofs(kt0 + 2, 0), kB2 = kofs(kt0 + 2, 1);
        if (kt0 + 1 < kt1) {
            const int b1k = kofs(kt0 + 1, 1);
            stage(2, kt0 + 1, b1k); stage(0, kt0 + 1, kA1); stage(3, kt0 + 1, b1k);
            wait_vmcnt<6>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
#if !FVA_EPI_STAMPS
        stamp(1);
#endif
        if (wr == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0

        for (int t = kt0; t < kt1; ++t) {
            const char* buf = smem + ((t - kt0) & 1) * BUF;
            // q0
            read_b(buf, 0, b0);
            read_a(buf, 0);
            if (t + 1 < kt1) stage(1, t + 1, kA1);
            retire_reads_then_barrier();
            mma(0, 0, b0);
            __builtin_amdgcn_s_barrier();
            // q1
            read_b(buf, 1, b1);
            if (t + 2 < kt1) stage(2, t + 2, kB2);
            retire_reads_then_barrier();
            mma(0, 1, b1);
            __builtin_amdgcn_s_barrier();
            // q2
            read_a(buf, 1);
            if (t + 2 < kt1) stage(0, t + 2, kA2);
            retire_reads_then_barrier();
            mma(1, 1, b1);
            __builtin_amdgcn_s_barrier();
            // q3
            if (t + 2 < kt1) {
                stage(3, t + 2, kB2);
                wait_vmcnt<6>();
            } else {
                wait_vmcnt<0>();
            }
            kA1 = kA2;                       // next iteration: A of k-tile (t+1)+1
            kA2 = kofs(t + 3, 0);            // read under this phase's MFMAs
            kB2 = kofs(t + 3, 1);
            __builtin_amdgcn_s_barrier();
            mma(1, 0, b0);
            __builtin_amdgcn_s_barrier();
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();   // re-align the two groups
        __syncthreads();                             // LDS is free for the epilogue
#if FVA_EPI_STAMPS
        stamp(0);
#else
        stamp(2);
#endif
        // everything below derives its addresses from this copy: computed after the loop, not carried (spilled) through it
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        const int lane_e = tid_e & 63, r_e = lane_e & 15, g_e = lane_e >> 4, wr_e = tid_e >> 8, wc_e = (tid_e >> 6) & 3;

        // The wave's 128x64 accumulator.  With the MFMA operands swapped, acc[mi][ni] of lane (r, g) is the 1 x 4 piece
        //   row  wr * WM + mi * 16 + r,   columns  wc * 64 + ni * 16 + 4 g .. + 3
        // of the block tile: four adjacent output channels of one pixel -- one packed 8-byte LDS write instead of four 2-byte ones
        // (the unswapped layout, four ROWS of one column per lane, cost 128 ds_write_b16 per lane: 2.0 of the epilogue's 7.5 us).
        const int row_e = wr_e * WM + r_e, col_e = wc_e * 64 + 4 * g_e;

        if constexpr (EPI == EPI_STATS) {
            if (p.stats != nullptr || p.stats_acc != nullptr) {
                f32x4 s1[4], s2[4];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) s1[ni] = s2[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                const bool whole = m0 + BM <= p.M;
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const bool live = whole || m0 + row_e + mi * 16 < p.M;
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        const f32x4 v = live ? acc[mi][ni] : f32x4{0.f, 0.f, 0.f, 0.f};
                        s1[ni] += v;
                        s2[ni] += v * v;
                    }
                }
                // the sum over the 16 lane rows r goes through LDS: every lane leaves its 2 x 16 partial sums (eight 16-byte writes) and
                // the column's thread adds the 2 (wr) x 16 (r) pieces in a fixed order -- 32 reads and adds per thread instead of 256
                // cross-lane operations (four DPP steps on 32 values, which the compiler does not fuse into the adds)
                float* red = (float*)smem;  // [2 (sum, sum of squares)][2 (wr)][16 (r)][RP]: 65 KiB of the dead staging buffers
                constexpr int RP = BN + 4;  // rows 16 bytes apart in the banks: the 16 lanes r of a 16-byte write hit 64 distinct banks
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    *(f32x4*)(red + ((0 * 2 + wr_e) * 16 + r_e) * RP + col_e + ni * 16) = s1[ni];
                    *(f32x4*)(red + ((1 * 2 + wr_e) * 16 + r_e) * RP + col_e + ni * 16) = s2[ni];
                }
                __syncthreads();
                {   // wave w: columns n0 + 32 w .., lanes 0-31 the sum, lanes 32-63 the sum of squares 
                    const int which = lane_e >> 5, col = (tid_e >> 6) * 32 + (lane_e & 31);
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < 32; ++i) s += red[(which * 32 + i) * RP + col];
                    if (n0 + col < p.N) {
                        if (p.stats_acc != nullptr) fx_atomic_add(fx_replica(p.stats_acc, p.N, p.acc_rep), p.N, which, n0 + col, s);
                        else *(p.stats + ((int64_t)mblk * 2 + which) * p.N + n0 + col) = s;
                    }
                }
                __syncthreads();
            }
        }

        auto out_pixel = [&](int m) -> int64_t {
            if (p.out_dense) return m;
            const uint32_t b = fd_div((uint32_t)m, p.div_ohw);
            const uint32_t rem = (uint32_t)m - b * (uint32_t)p.OHW;
            const uint32_t oy = fd_div(rem, p.div_ow);
            const uint32_t ox = rem - oy * (uint32_t)p.OW;
            return (int64_t)b * p.out_img + (int64_t)(oy * p.osy + p.ooy) * p.out_row + (ox * p.osx + p.oox);
        };
        // bf16 tile through LDS so that every row leaves as 16-byte pieces
        constexpr int PITCH = BN * 2 + 16;
#if FVA_EPI_STAMPS
        stamp(1);
#endif
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                f32x4 v = acc[mi][ni];
                if constexpr (EPI == EPI_BNACT) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = bnact_f(p, v[j], n0 + col_e + ni * 16 + j);
                }
                *(u32x2*)(smem + (row_e + mi * 16) * PITCH + (col_e + ni * 16) * 2) = u32x2{cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])};
            }
        __syncthreads();
#if FVA_EPI_STAMPS
        stamp(2);
#endif
        store_tile_bf16<EPI, BM, BN, NT, PITCH>(p, smem, tid_e, n0, mblk, [&](int row) { const int m = m0 + row; return m < p.M ? out_pixel(m) : (int64_t)-1; },
                                                nullptr, nullptr, EPI == EPI_BNB ? coef_tab : nullptr);
    }
    stamp(3);
}

constexpr int IGEMM8_SMEM = 256 * (256 * 2 + 16) + 4096 + 4096;   // the epilogue's transposed tile (> 2 staging buffers) + k table + bnb_fill_lds table

// Eligible bf16 layers go to the 8-phase kernel: N a multiple of 256, full 64-channel k-tiles, at least 128 tiles of 256x256 and
// either a long reduction (>= 16 k-tiles, FVA_IGEMM8_MINKT: the per-tile prologue / epilogue is not overlapped by a second block as
// in the 128x128 kernel) or at most one round of tiles.  Measured (tools/check_igemm8.py, B = 32): 256->512 @40^2 161 -> 129 us,
// 512->1024 @20^2 154 -> 118 us, 128->256 @80^2 (18 k-tiles, 800 tiles) 158 -> 157 us.  FVA_IGEMM8=0 turns the kernel off.
inline bool igemm8_enabled() {
    static bool v = [] {
        const char* e = getenv("FVA_IGEMM8");
        return !e || atoi(e) != 0;
    }();
    return v;
}
// in_pixels: pixels of the (halo) input tensor -- the kernel addresses both operands with 32-bit byte offsets
inline bool use_igemm8(int dtype, int64_t M, int N, int C, int ntaps, int64_t in_pixels) {
    if (!igemm8_enabled() || dtype != FVA_BF16 || N % 256 || C % 64) return false;
    if (in_pixels * C * 2 >= (1ll << 31) || (int64_t)(ntaps + 1) * N * C * 2 >= (1ll << 31)) return false;
    const int64_t ktiles = (int64_t)ntaps * (C / 64), tiles = (int64_t)cdiv(M, 256) * (N / 256);
    if (ktiles > 496 || ktiles < 8) return false;
    if (tiles < 128) return false;
    static const int min_kt = [] { const char* e = getenv("FVA_IGEMM8_MINKT"); return e ? atoi(e) : 16; }();
    return ktiles >= min_kt || tiles <= 256;
}

long long* g_stamps = nullptr;   // fva_conv_debug_stamps
int g_stamp_rows = 0;

template <int EPI>
int launch_igemm8(const IgemmParams& p, hipStream_t s) {
    IgemmParams q = p;
    q.nblocks = p.N / 256;
    const int tiles = cdiv(p.M, 256) * q.nblocks;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)igemm8_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, IGEMM8_SMEM);
        attr_done = true;
    }
    Igemm8Diag sk{};
    sk.stamps = g_stamps;
    sk.stamp_rows = g_stamp_rows;
    hipLaunchKernelGGL((igemm8_kernel<EPI>), dim3(tiles), dim3(512), IGEMM8_SMEM, s, q, sk);
    fva_note_kernel("igemm8");
    FVA_LAUNCH_CHECK("igemm8_kernel");
    return FVA_OK;
}

inline bool wide_tile(int N) { return N >= 128; }

// 128x128 tiles for N >= 128, 256x64 for thinner outputs (the rows of BatchNorm partial statistics follow the tile)
inline int tile_bm(int /*dtype*/, int /*M*/, int N) { return wide_tile(N) ? 128 : 256; }

template <int EPI>
int launch_igemm(int dtype, const IgemmParams& p, hipStream_t s) {
    if (dtype == FVA_BF16) {
        if constexpr (EPI != EPI_HEAD) {
            if (!p.halfrow && use_igemm8(dtype, p.M, p.N, p.C, p.ktiles / p.kt_per_tap, (int64_t)(p.M / p.OHW + 1) * p.in_img))
                return launch_igemm8<EPI>(p, s);
        }
        return wide_tile(p.N) ? launch_one<bf16_t, 128, 128, EPI, 2>(p, s) : launch_one<bf16_t, 256, 64, EPI, 2>(p, s);
    }
    return wide_tile(p.N) ? launch_one<float, 128, 128, EPI, 2>(p, s) : launch_one<float, 256, 64, EPI, 2>(p, s);
}

int check_desc(const fva_conv_desc* d, const char* who) {
    if (!d) return fva_fail(FVA_ERR_ARG, "%s: null descriptor", who);
    if (d->dtype != FVA_F32 && d->dtype != FVA_BF16) return fva_fail(FVA_ERR_ARG, "%s: bad dtype %d", who, d->dtype);
    if (!((d->ksize == 1 && d->stride == 1) || (d->ksize == 3 && (d->stride == 1 || d->stride == 2))))
        return fva_fail(FVA_ERR_ARG, "%s: unsupported ksize/stride %d/%d", who, d->ksize, d->stride);
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0)
        return fva_fail(FVA_ERR_ARG, "%s: non-positive size", who);
    if (d->stride == 2 && ((d->H | d->W) & 1)) return fva_fail(FVA_ERR_ARG, "%s: stride 2 needs even H, W", who);
    if (d->in_pad < d->ksize / 2) return fva_fail(FVA_ERR_ARG, "%s: in_pad %d < %d", who, d->in_pad, d->ksize / 2);
    if ((int64_t)d->B * (d->H + 2) * (d->W + 2) >= (1ll << 31)) return fva_fail(FVA_ERR_ARG, "%s: too many pixels", who);
    return FVA_OK;
}

// reduction channels must fill 128-byte k-tiles (or be exactly half a bf16 tile)
int check_red_channels(int dtype, int C, const char* who) {
    const int bk = dtype == FVA_BF16 ? 64 : 32;
    if (C % bk == 0) return FVA_OK;
    if (dtype == FVA_BF16 && C == 32) return FVA_OK;
    return fva_fail(FVA_ERR_ARG, "%s: reduction channels %d not a multiple of %d", who, C, bk);
}

inline bool halfrow_mode(int dtype, int C) { return dtype == FVA_BF16 && C == 32; }

void finish_taps(IgemmParams& p, int ntaps, int dtype) {
    const int bk = dtype == FVA_BF16 ? 64 : 32;
    p.halfrow = halfrow_mode(dtype, p.C) ? 1 : 0;
    if (p.halfrow) {
        if (ntaps & 1) {  // pad with a zero-weight tap (packed weights carry the zero tap at index k*k)
            p.tap_pix[ntaps] = p.tap_pix[0];
            p.tap_w[ntaps] = p.tap_w[ntaps];  // set by caller
        }
        p.ktiles = (ntaps + 1) / 2;
        p.kt_per_tap = 1;
    } else {
        p.kt_per_tap = p.C / bk;
        p.ktiles = ntaps * p.kt_per_tap;
    }
    // Order of the k-tiles.  Tap-major (all channel slices of tap 0, then tap 1, ...) puts C/64 k-tiles -- C/64 x 64 KiB x
    // the 32 CUs of an XCD -- between two reads of (nearly) the same input bytes by neighbouring taps: beyond the 4 MiB L2
    // from C = 256 on (measured: 50 % L2 hits for the C = 512 dgrad against 84 % for the C <= 256 forward).  Slice-major
    // (all taps of channel slice 0, then slice 1, ...) re-reads them one to three k-tiles later.
    static const bool slice_major = [] { const char* e = getenv("FVA_KORDER"); return !e || atoi(e) != 0; }();
    p.ktaps = (slice_major && !p.halfrow && p.kt_per_tap > 1) ? ntaps : 0;
}

// destination of tap (ky, kx), input channel ci in the paired stride-2 dgrad layout [v][2*Cin][Cout] (see dgrad_paired)
__device__ __forceinline__ int64_t paired_row(int t, int ci, int Cin) {
    const int ky = t / 3, kx = t - ky * 3;
    const int v = ky * 2 + (kx == 0 ? 1 : 0);
    return (int64_t)v * 2 * Cin + (kx == 1 ? ci : Cin + ci);
}

__global__ void pack_weights_kernel(const float* __restrict__ w, void* fwd, void* dgr, int Cout, int Cin, int kk,
                                    int taps_f, int taps_d, int bf16, int paired) {
    // one thread per (tap, co, ci) of the padded space max(taps_f, taps_d) x Cout x Cin
    const int64_t total = (int64_t)(taps_f > taps_d ? taps_f : taps_d) * Cout * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const int co = (int)((i / Cin) % Cout);
        const int t = (int)(i / ((int64_t)Cin * Cout));
        const float v = t < kk ? w[((int64_t)co * Cin + ci) * kk + t] : 0.f;
        if (fwd && t < taps_f) {
            const int64_t o = ((int64_t)t * Cout + co) * Cin + ci;
            if (bf16) ((bf16_t*)fwd)[o] = (bf16_t)v; else ((float*)fwd)[o] = v;
        }
        if (dgr && paired) {
            if (t < kk) {
                const int64_t o = paired_row(t, ci, Cin) * Cout + co;
                if (bf16) ((bf16_t*)dgr)[o] = (bf16_t)v; else ((float*)dgr)[o] = v;
                if (t % 3 == 0) {  // the x-even half of a dxo=1 tap is zero
                    const int64_t z = (paired_row(t, ci, Cin) - Cin) * Cout + co;
                    if (bf16) ((bf16_t*)dgr)[z] = (bf16_t)0.f; else ((float*)dgr)[z] = 0.f;
                }
            }
        } else if (dgr && t < taps_d) {
            const int64_t o = ((int64_t)t * Cin + ci) * Cout + co;
            if (bf16) ((bf16_t*)dgr)[o] = (bf16_t)v; else ((float*)dgr)[o] = v;
        }
    }
}

// All conv layers of a model in ONE launch: blockIdx.y selects the table entry (one per layer); each block walks
// 32 (Cout) x 32 (Cin) tiles of that layer.  A tile's k*k taps are read as contiguous runs of the OIHW source,
// staged in LDS and written out as 64-byte (bf16) rows of both operand layouts.
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const fva_pack_entry* __restrict__ table) {
    __shared__ float tile[32][32 * 9 + 1];
    const fva_pack_entry e = table[blockIdx.y];
    const int kk = e.ksize * e.ksize;
    const int tco = (e.Cout + 31) / 32, tci = (e.Cin + 31) / 32;
    const bool bf = e.dtype == FVA_BF16;
    for (int tl = blockIdx.x; tl < tco * tci; tl += gridDim.x) {
        const int co0 = (tl / tci) * 32, ci0 = (tl % tci) * 32;
        const int ncol = (e.Cin - ci0 < 32 ? e.Cin - ci0 : 32) * kk;   // contiguous floats per co row
        for (int i = threadIdx.x; i < 32 * 32 * kk; i += 256) {
            const int r = i / (32 * kk), c = i - r * (32 * kk);
            tile[r][c] = (co0 + r < e.Cout && c < ncol) ? e.w[((int64_t)(co0 + r) * e.Cin + ci0) * kk + c] : 0.f;
        }
        __syncthreads();
        const int a = threadIdx.x >> 5, l = threadIdx.x & 31;        // 8 rows at a time, 32 lanes along the row
        for (int t = 0; t < kk; ++t) {
#pragma unroll
            for (int rr = 0; rr < 32; rr += 8) {
                const int r = rr + a;
                // forward layout [t][co][ci]: row = co, lanes along ci
                if (e.w_fwd && co0 + r < e.Cout && ci0 + l < e.Cin) {
                    const int64_t o = ((int64_t)t * e.Cout + co0 + r) * e.Cin + ci0 + l;
                    const float v = tile[r][l * kk + t];
                    if (bf) ((bf16_t*)e.w_fwd)[o] = (bf16_t)v; else ((float*)e.w_fwd)[o] = v;
                }
                // dgrad layout [t][ci][co]: row = ci, lanes along co
                if (e.w_dgrad && ci0 + r < e.Cin && co0 + l < e.Cout) {
                    const float v = tile[l][r * kk + t];
                    if (e.dgrad_paired) {
                        const int64_t row = paired_row(t, ci0 + r, e.Cin);
                        const int64_t o = row * e.Cout + co0 + l;
                        if (bf) ((bf16_t*)e.w_dgrad)[o] = (bf16_t)v; else ((float*)e.w_dgrad)[o] = v;
                        if (t % 3 == 0) {
                            const int64_t z = (row - e.Cin) * e.Cout + co0 + l;
                            if (bf) ((bf16_t*)e.w_dgrad)[z] = (bf16_t)0.f; else ((float*)e.w_dgrad)[z] = 0.f;
                        }
                    } else {
                        const int64_t o = ((int64_t)t * e.Cin + ci0 + r) * e.Cout + co0 + l;
                        if (bf) ((bf16_t*)e.w_dgrad)[o] = (bf16_t)v; else ((float*)e.w_dgrad)[o] = v;
                    }
                }
            }
        }
        __syncthreads();
    }
    // zero pad tap (bf16 half-row k-tiles): taps_* may exceed k*k by one
    const int64_t cc = (int64_t)e.Cout * e.Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cc; i += (int64_t)gridDim.x * blockDim.x) {
        if (e.w_fwd && e.taps_fwd > kk) { if (bf) ((bf16_t*)e.w_fwd)[kk * cc + i] = (bf16_t)0.f; else ((float*)e.w_fwd)[kk * cc + i] = 0.f; }
        if (e.w_dgrad && !e.dgrad_paired && e.taps_dgrad > kk) { if (bf) ((bf16_t*)e.w_dgrad)[kk * cc + i] = (bf16_t)0.f; else ((float*)e.w_dgrad)[kk * cc + i] = 0.f; }
    }
}

// The same re-pack with ONE block per 32 (Cout) x 32 (Cin) tile of ANY layer (round 3): the 2-D grid above starts 512 blocks per
// layer whatever its size (38 400 blocks for YOLOv3's 75 layers, most of which find nothing to do), its tile load divides every
// index by 32 k^2 and its stores are 2 bytes per lane -- 287 us per step for 0.5 GB (1.7 TB/s).  Here the caller numbers the tiles
// of all layers consecutively (entry.tile_start = tiles of the entries before it), a block finds its layer by bisection, aligned
// tiles (k = 3 or 1, whole 32 x 32) load 16 bytes per lane and store four bf16 per lane (whole 64-byte rows); ragged or fp32
// tiles and the paired layout take the element-wise path of the kernel above.
__global__ __launch_bounds__(256) void pack_weights_tiled_kernel(const fva_pack_entry* __restrict__ table, int n) {
    __shared__ float tile[32][32 * 9 + 1];
    int lo = 0, hi = n - 1;                       // last entry whose tile_start <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].tile_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const fva_pack_entry e = table[lo];
    const int kk = e.ksize * e.ksize;
    const int tci = (e.Cin + 31) / 32;
    const int tl = (int)blockIdx.x - e.tile_start;
    const int co0 = (tl / tci) * 32, ci0 = (tl % tci) * 32;
    const bool bf = e.dtype == FVA_BF16;
    const int tid = threadIdx.x;
    const bool whole = co0 + 32 <= e.Cout && ci0 + 32 <= e.Cin;
    if (whole && bf && (kk == 9 || kk == 1) && !e.dgrad_paired && e.Cin % 4 == 0 && e.Cout % 4 == 0 && ((uintptr_t)e.w & 15) == 0) {   // 16-byte loads, 8-byte stores
        const int rowf4 = 8 * kk;                 // float4 per tile row (32 ci x kk floats, 16-byte aligned: ci0 * kk * 4 bytes)
        for (int i = tid; i < 32 * rowf4; i += 256) {
            const int r = i / rowf4, c4 = i - r * rowf4;
            const f32x4 v = *(const f32x4*)(e.w + ((int64_t)(co0 + r) * e.Cin + ci0) * kk + c4 * 4);
            tile[r][c4 * 4 + 0] = v[0]; tile[r][c4 * 4 + 1] = v[1]; tile[r][c4 * 4 + 2] = v[2]; tile[r][c4 * 4 + 3] = v[3];
        }
        __syncthreads();
        const int row = tid >> 3, q = (tid & 7) * 4;   // 32 rows x 8 lanes x 4 elements
        for (int t = 0; t < kk; ++t) {
            if (e.w_fwd) {                             // [t][co][ci]: row = co, four consecutive ci
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16_t)tile[row][(q + j) * kk + t];
                *(bf16x4*)((bf16_t*)e.w_fwd + ((int64_t)t * e.Cout + co0 + row) * e.Cin + ci0 + q) = o;
            }
            if (e.w_dgrad) {                           // [t][ci][co]: row = ci, four consecutive co
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16_t)tile[q + j][row * kk + t];
                *(bf16x4*)((bf16_t*)e.w_dgrad + ((int64_t)t * e.Cin + ci0 + row) * e.Cout + co0 + q) = o;
            }
        }
    } else {
        const int ncol = (e.Cin - ci0 < 32 ? e.Cin - ci0 : 32) * kk;
        for (int i = tid; i < 32 * 32 * kk; i += 256) {
            const int r = i / (32 * kk), c = i - r * (32 * kk);
            tile[r][c] = (co0 + r < e.Cout && c < ncol) ? e.w[((int64_t)(co0 + r) * e.Cin + ci0) * kk + c] : 0.f;
        }
        __syncthreads();
        const int a = tid >> 5, l = tid & 31;
        for (int t = 0; t < kk; ++t) {
#pragma unroll
            for (int rr = 0; rr < 32; rr += 8) {
                const int r = rr + a;
                if (e.w_fwd && co0 + r < e.Cout && ci0 + l < e.Cin) {
                    const int64_t o = ((int64_t)t * e.Cout + co0 + r) * e.Cin + ci0 + l;
                    const float v = tile[r][l * kk + t];
                    if (bf) ((bf16_t*)e.w_fwd)[o] = (bf16_t)v; else ((float*)e.w_fwd)[o] = v;
                }
                if (e.w_dgrad && ci0 + r < e.Cin && co0 + l < e.Cout) {
                    const float v = tile[l][r * kk + t];
                    if (e.dgrad_paired) {
                        const int64_t row = paired_row(t, ci0 + r, e.Cin);
                        const int64_t o = row * e.Cout + co0 + l;
                        if (bf) ((bf16_t*)e.w_dgrad)[o] = (bf16_t)v; else ((float*)e.w_dgrad)[o] = v;
                        if (t % 3 == 0) {
                            const int64_t z = (row - e.Cin) * e.Cout + co0 + l;
                            if (bf) ((bf16_t*)e.w_dgrad)[z] = (bf16_t)0.f; else ((float*)e.w_dgrad)[z] = 0.f;
                        }
                    } else {
                        const int64_t o = ((int64_t)t * e.Cin + ci0 + r) * e.Cout + co0 + l;
                        if (bf) ((bf16_t*)e.w_dgrad)[o] = (bf16_t)v; else ((float*)e.w_dgrad)[o] = v;
                    }
                }
            }
        }
    }
    // zero pad tap (bf16 half-row k-tiles): taps_* may exceed k*k by one -- this tile's share of it
    if (e.taps_fwd > kk || (!e.dgrad_paired && e.taps_dgrad > kk)) {
        const int64_t cc = (int64_t)e.Cout * e.Cin;
        for (int i = tid; i < 32 * 32; i += 256) {
            const int r = i >> 5, c = i & 31;
            if (co0 + r < e.Cout && ci0 + c < e.Cin) {
                if (e.w_fwd && e.taps_fwd > kk) {
                    const int64_t o = kk * cc + (int64_t)(co0 + r) * e.Cin + ci0 + c;
                    if (bf) ((bf16_t*)e.w_fwd)[o] = (bf16_t)0.f; else ((float*)e.w_fwd)[o] = 0.f;
                }
                if (e.w_dgrad && !e.dgrad_paired && e.taps_dgrad > kk) {
                    const int64_t o = kk * cc + (int64_t)(ci0 + c) * e.Cout + co0 + r;
                    if (bf) ((bf16_t*)e.w_dgrad)[o] = (bf16_t)0.f; else ((float*)e.w_dgrad)[o] = 0.f;
                }
            }
        }
    }
}

// Stride-2 dgrad of thin layers: the two output x-parities of a row are produced together as N' = 2*Cin columns (the
// pixel pair (2j, 2j+1) is contiguous in NHWC), from "virtual taps" v = ky*2 + dxo whose [2*Cin][Cout] matrices hold
// kx=1 | kx=2 for dxo=0 and zeros | kx=0 for dxo=1.  2 launches instead of 4, full-line stores, 4/3 of the MACs.
inline int pair_max_cin() {
    static int v = [] {
        const char* e = getenv("FVA_DGRAD_PAIR_MAX");
        return e ? atoi(e) : 64;
    }();
    return v;
}
inline bool dgrad_paired(int dtype, int ksize, int stride, int Cin, int Cout) {
    return ksize == 3 && stride == 2 && Cin <= pair_max_cin() && !use_pdgrad2(dtype, ksize, stride, Cout, Cin);
}

int packed_taps(const fva_conv_desc* d, int for_dgrad) {
    if (for_dgrad && dgrad_paired(d->dtype, d->ksize, d->stride, d->Cin, d->Cout)) return 12;   // 6 virtual taps x 2 parities

    const int kk = d->ksize * d->ksize;
    const int C = for_dgrad ? d->Cout : d->Cin;
    return (halfrow_mode(d->dtype, C) && (kk & 1)) ? kk + 1 : kk;
}

}  // namespace

long long* fva_debug_stamps_ptr() { return g_stamps; }
int fva_debug_stamps_rows() { return g_stamp_rows; }
extern "C" {

int fva_conv_patch_kernel(int on) {
    const int prev = pconv_enabled() ? 1 : 0;
    g_pconv = on ? 1 : 0;
    return prev;
}

int fva_conv_debug_stamps(void* stamps, int32_t rows) {
    if (stamps && rows < 1) return fva_fail(FVA_ERR_ARG, "fva_conv_debug_stamps: rows must be >= 1");
    g_stamps = (long long*)stamps;
    g_stamp_rows = stamps ? rows : 0;
    return FVA_OK;
}

int64_t fva_conv_packed_elems(const fva_conv_desc* d, int for_dgrad) {
    if (!d) return 0;
    return (int64_t)packed_taps(d, for_dgrad) * d->Cout * d->Cin;
}

int fva_conv_pack_weights(const fva_conv_desc* d, const float* w, void* w_fwd, void* w_dgrad, void* stream) {
    int rc = check_desc(d, "fva_conv_pack_weights");
    if (rc) return rc;
    if (!w) return fva_fail(FVA_ERR_ARG, "fva_conv_pack_weights: null weights");
    const int paired = dgrad_paired(d->dtype, d->ksize, d->stride, d->Cin, d->Cout) ? 1 : 0;
    const int tf = packed_taps(d, 0), td = paired ? d->ksize * d->ksize : packed_taps(d, 1);
    const int64_t total = (int64_t)(tf > td ? tf : td) * d->Cout * d->Cin;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, w_fwd, w_dgrad, d->Cout,
                       d->Cin, d->ksize * d->ksize, tf, td, d->dtype == FVA_BF16 ? 1 : 0, paired);
    FVA_LAUNCH_CHECK("pack_weights_kernel");
    return FVA_OK;
}

int fva_conv_pack_weights_multi(const fva_pack_entry* table_dev, int32_t n, int64_t max_elems, void* stream) {
    if (!table_dev || n < 1 || max_elems < 1) return fva_fail(FVA_ERR_ARG, "fva_conv_pack_weights_multi: bad argument");
    int64_t gx = (max_elems / 9 + 1023) / 1024;   // one block per 32x32 tile of the largest layer, at most 512
    if (gx > 512) gx = 512;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(pack_weights_multi_kernel, dim3((int)gx, n), dim3(256), 0, (hipStream_t)stream, table_dev);
    FVA_LAUNCH_CHECK("pack_weights_multi_kernel");
    return FVA_OK;
}

int fva_conv_pack_weights_tiled(const fva_pack_entry* table_dev, int32_t n, int32_t total_tiles, void* stream) {
    if (!table_dev || n < 1 || total_tiles < 1) return fva_fail(FVA_ERR_ARG, "fva_conv_pack_weights_tiled: bad argument");
    hipLaunchKernelGGL(pack_weights_tiled_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, table_dev, n);
    FVA_LAUNCH_CHECK("pack_weights_tiled_kernel");
    return FVA_OK;
}

int32_t fva_conv_stat_blocks(const fva_conv_desc* d) {
    if (!d) return 0;
    if (use_pconv(d->dtype, d->ksize, d->stride, d->Cin, d->Cout, d->H, d->W)) return pconv_tiles(d->B, d->H, d->W);
    const int OH = (d->H - 1) / d->stride + 1, OW = (d->W - 1) / d->stride + 1;
    const int64_t M = (int64_t)d->B * OH * OW;
    const int64_t in_img = (int64_t)(d->H + 2 * d->in_pad) * (d->W + 2 * d->in_pad);
    if (!halfrow_mode(d->dtype, d->Cin) && use_igemm8(d->dtype, M, d->Cout, d->Cin, d->ksize * d->ksize, (M / ((int64_t)OH * OW) + 1) * in_img))
        return cdiv(M, 256);
    return cdiv(M, tile_bm(d->dtype, (int)M, d->Cout));
}

static int setup_fwd(const fva_conv_desc* d, IgemmParams& p, const char* who) {
    int rc = check_desc(d, who);
    if (rc) return rc;
    rc = check_red_channels(d->dtype, d->Cin, who);
    if (rc) return rc;
    const int k = d->ksize, s = d->stride;
    const int OH = (d->H - 1) / s + 1, OW = (d->W - 1) / s + 1;
    p = IgemmParams();
    p.acc_rep = p.bnb_rep = p.ax_rep = 1;
    p.M = d->B * OH * OW;
    p.N = d->Cout;
    p.C = d->Cin;
    p.OW = OW;
    p.OHW = OH * OW;
    p.div_ow = make_fastdiv(OW);
    p.div_ohw = make_fastdiv(OH * OW);
    p.in_row = d->W + 2 * d->in_pad;
    p.in_img = (d->H + 2 * d->in_pad) * p.in_row;
    p.sy = p.sx = s;
    p.y0 = p.x0 = d->in_pad - k / 2;
    int nt = 0;
    for (int kh = 0; kh < k; ++kh)
        for (int kw = 0; kw < k; ++kw) {
            p.tap_pix[nt] = kh * p.in_row + kw;
            p.tap_w[nt] = kh * k + kw;
            ++nt;
        }
    p.tap_w[nt] = k * k;  // zero tap (present in packed weights when needed)
    finish_taps(p, nt, d->dtype);
    p.out_dense = 1;
    p.out_pitch = d->Cout;
    return FVA_OK;
}

int fva_conv_fwd(const fva_conv_desc* d, const void* x, const void* w_fwd, void* y, float* stats_partial, void* stream) {
    IgemmParams p;
    int rc = setup_fwd(d, p, "fva_conv_fwd");
    if (rc) return rc;
    if (!x || !w_fwd || !y) return fva_fail(FVA_ERR_ARG, "fva_conv_fwd: null pointer");
    if (d->Cout % 8) return fva_fail(FVA_ERR_ARG, "fva_conv_fwd: Cout %d not a multiple of 8", d->Cout);
    p.in = x;
    p.wt = w_fwd;
    p.out = y;
    p.stats = stats_partial;
    FvaProfileSpan span(0 | (d->ksize << 8), 2.0 * p.M * (double)d->Cout * d->Cin * d->ksize * d->ksize, (hipStream_t)stream);
    if (use_pconv(d->dtype, d->ksize, d->stride, d->Cin, d->Cout, d->H, d->W)) return launch_pconv<EPI_STATS>(p, d->B, d->H, d->W, false, (hipStream_t)stream);
    return launch_igemm<EPI_STATS>(d->dtype, p, (hipStream_t)stream);
}

/* The forward apply pass of the block BEFORE a 1x1 convolution, fused into that convolution (igemm_kernel<..., AX>): see the header. */
static int conv1x1_fwd_apply_impl(const fva_conv_desc* d, const void* y_prev, const float* scale, const float* shift, const fva_bn_fwd_acc* prev,
                                  const void* residual, int32_t res_pad, void* z, const void* w_fwd, void* y, float* stats_partial,
                                  int64_t* acc_out, int32_t replicas_out, void* stream);

int fva_conv1x1_fwd_apply(const fva_conv_desc* d, const void* y_prev, const float* scale, const float* shift, const void* residual,
                          int32_t res_pad, void* z, const void* w_fwd, void* y, float* stats_partial, void* stream) {
    if (!scale || !shift) return fva_fail(FVA_ERR_ARG, "fva_conv1x1_fwd_apply: null pointer");
    return conv1x1_fwd_apply_impl(d, y_prev, scale, shift, nullptr, residual, res_pad, z, w_fwd, y, stats_partial, nullptr, 1, stream);
}

static int conv1x1_fwd_apply_impl(const fva_conv_desc* d, const void* y_prev, const float* scale, const float* shift, const fva_bn_fwd_acc* prev,
                                  const void* residual, int32_t res_pad, void* z, const void* w_fwd, void* y, float* stats_partial,
                                  int64_t* acc_out, int32_t replicas_out, void* stream) {
    IgemmParams p;
    int rc = setup_fwd(d, p, "fva_conv1x1_fwd_apply");
    if (rc) return rc;
    if (!y_prev || !z || !w_fwd || !y) return fva_fail(FVA_ERR_ARG, "fva_conv1x1_fwd_apply: null pointer");
    if (d->dtype != FVA_BF16 || d->ksize != 1 || d->stride != 1) return fva_fail(FVA_ERR_ARG, "fva_conv1x1_fwd_apply: a bf16 1x1 stride-1 layer only");
    if (d->Cin % 64 || d->Cin > 512 || d->Cout % 8 || d->Cout > 128)
        return fva_fail(FVA_ERR_ARG, "fva_conv1x1_fwd_apply: needs Cin %% 64 == 0, Cin <= 512 and Cout <= 128 (one column block: every element is transformed once); got %d -> %d", d->Cin, d->Cout);
    if (d->in_pad > 1 || res_pad < 0 || res_pad > 1) return fva_fail(FVA_ERR_ARG, "fva_conv1x1_fwd_apply: borders of 0 or 1 pixel only");
    if ((int64_t)d->B * (d->H + 2) * (d->W + 2) * d->Cin >= (1ll << 31)) return fva_fail(FVA_ERR_ARG, "fva_conv1x1_fwd_apply: activation larger than 2^31 elements");
    p.in = z;                       // not read by the kernel's operand path; kept for the address checks of the common code
    p.wt = w_fwd;
    p.out = y;
    p.stats = stats_partial;
    p.stats_acc = (long long*)acc_out;
    p.acc_rep = replicas_out;
    p.ax_y = y_prev; p.ax_res = residual; p.ax_z = z;
    p.ax_scale = scale; p.ax_shift = shift;
    if (prev) {
        p.ax_acc = (long long*)prev->acc; p.ax_zero = (long long*)prev->zero; p.ax_rep = prev->replicas;
        p.ax_gamma = prev->gamma; p.ax_beta = prev->beta; p.ax_rm = prev->running_mean; p.ax_rv = prev->running_var;
        p.ax_nbt = (long long*)prev->num_batches_tracked; p.ax_momentum = prev->momentum; p.ax_eps = prev->eps;
        p.ax_n = bn_n((double)p.M);
        p.ax_save_mean = prev->save_mean; p.ax_save_rstd = prev->save_rstd; p.ax_scale_out = prev->scale; p.ax_shift_out = prev->shift;
    }
    p.ax_pad = d->in_pad; p.ax_res_pad = res_pad; p.ax_H = d->H; p.ax_W = d->W;
    FvaProfileSpan span(3 | (1 << 8), 2.0 * p.M * (double)d->Cout * d->Cin, (hipStream_t)stream);   // class 3: a forward launch that also carries an apply pass
    return wide_tile(p.N) ? launch_one<bf16_t, 128, 128, EPI_STATS, 2, true>(p, (hipStream_t)stream)
                          : launch_one<bf16_t, 256, 64, EPI_STATS, 2, true>(p, (hipStream_t)stream);
}

int fva_conv_fwd_acc(const fva_conv_desc* d, const void* x, const void* w_fwd, void* y, int64_t* acc, int32_t replicas, void* stream) {
    IgemmParams p;
    int rc = setup_fwd(d, p, "fva_conv_fwd_acc");
    if (rc) return rc;
    if (!x || !w_fwd || !y || !acc) return fva_fail(FVA_ERR_ARG, "fva_conv_fwd_acc: null pointer");
    if (!fva_replicas_ok(replicas)) return fva_fail(FVA_ERR_ARG, "fva_conv_fwd_acc: replicas = %d is not a power of two in 1..%d", replicas, FVA_BN_ACC_MAX_REPLICAS);
    if (d->Cout % 8) return fva_fail(FVA_ERR_ARG, "fva_conv_fwd_acc: Cout %d not a multiple of 8", d->Cout);
    p.in = x;
    p.wt = w_fwd;
    p.out = y;
    p.stats_acc = (long long*)acc;
    p.acc_rep = replicas;
    FvaProfileSpan span(0 | (d->ksize << 8), 2.0 * p.M * (double)d->Cout * d->Cin * d->ksize * d->ksize, (hipStream_t)stream);
    if (use_pconv(d->dtype, d->ksize, d->stride, d->Cin, d->Cout, d->H, d->W)) return launch_pconv<EPI_STATS>(p, d->B, d->H, d->W, false, (hipStream_t)stream);
    return launch_igemm<EPI_STATS>(d->dtype, p, (hipStream_t)stream);
}

int fva_conv1x1_fwd_apply_acc(const fva_conv_desc* d, const void* y_prev, const fva_bn_fwd_acc* prev, const void* residual, int32_t res_pad,
                              void* z, const void* w_fwd, void* y, int64_t* acc_out, int32_t replicas_out, void* stream) {
    if (!prev || !prev->scale || !prev->shift || !acc_out) return fva_fail(FVA_ERR_ARG, "fva_conv1x1_fwd_apply_acc: null pointer");
    if (!fva_replicas_ok(replicas_out)) return fva_fail(FVA_ERR_ARG, "fva_conv1x1_fwd_apply_acc: replicas_out = %d is not a power of two in 1..%d", replicas_out, FVA_BN_ACC_MAX_REPLICAS);
    if (!prev->acc)          // the previous block's coefficients are given (finalised already: fva_bn_acc_finalize); only this layer's statistics accumulate
        return conv1x1_fwd_apply_impl(d, y_prev, prev->scale, prev->shift, nullptr, residual, res_pad, z, w_fwd, y, nullptr, acc_out, replicas_out, stream);
    if (!prev->gamma || !prev->beta || !prev->save_mean || !prev->save_rstd || prev->zero == prev->acc)
        return fva_fail(FVA_ERR_ARG, "fva_conv1x1_fwd_apply_acc: null pointer in the accumulator descriptor");
    if (!fva_replicas_ok(prev->replicas)) return fva_fail(FVA_ERR_ARG, "fva_conv1x1_fwd_apply_acc: prev->replicas = %d is not a power of two in 1..%d", prev->replicas, FVA_BN_ACC_MAX_REPLICAS);
    if (!d) return fva_fail(FVA_ERR_ARG, "fva_conv1x1_fwd_apply_acc: null descriptor");
    if (!wide_tile(d->Cout)) {   // the thin tile has no room for the prologue (264 registers with it): a small launch finalises, the fused one takes the arrays
        const int rc = fva_bn_acc_finalize(prev, (int64_t)d->B * d->H * d->W, d->Cin, stream);
        if (rc) return rc;
        return conv1x1_fwd_apply_impl(d, y_prev, prev->scale, prev->shift, nullptr, residual, res_pad, z, w_fwd, y, nullptr, acc_out, replicas_out, stream);
    }
    return conv1x1_fwd_apply_impl(d, y_prev, nullptr, nullptr, prev, residual, res_pad, z, w_fwd, y, nullptr, acc_out, replicas_out, stream);
}

int fva_conv_fwd_bnact(const fva_conv_desc* d, const void* x, const void* w_fwd, const float* scale, const float* shift,
                       const void* residual, void* z, int32_t z_pad, void* stream) {
    IgemmParams p;
    int rc = setup_fwd(d, p, "fva_conv_fwd_bnact");
    if (rc) return rc;
    if (!x || !w_fwd || !scale || !shift || !z) return fva_fail(FVA_ERR_ARG, "fva_conv_fwd_bnact: null pointer");
    if (d->Cout % 8 || z_pad < 0) return fva_fail(FVA_ERR_ARG, "fva_conv_fwd_bnact: Cout %d not a multiple of 8 or bad z_pad", d->Cout);
    const int OH = (d->H - 1) / d->stride + 1, OW = (d->W - 1) / d->stride + 1;
    p.in = x;
    p.wt = w_fwd;
    p.out = z;
    p.scale = scale;
    p.shift = shift;
    p.addend = residual;          // same halo geometry as z
    p.out_dense = 0;
    p.out_row = OW + 2 * z_pad;
    p.out_img = (OH + 2 * z_pad) * p.out_row;
    p.osy = p.osx = 1;
    p.ooy = p.oox = z_pad;
    FvaProfileSpan span(0 | (d->ksize << 8), 2.0 * p.M * (double)d->Cout * d->Cin * d->ksize * d->ksize, (hipStream_t)stream);
    rc = launch_igemm<EPI_BNACT>(d->dtype, p, (hipStream_t)stream);
    if (rc || z_pad == 0) return rc;
    return fva_zero_halo_border(z, d->B, OH, OW, d->Cout * (d->dtype == FVA_BF16 ? 2 : 4) / 16, z_pad, (hipStream_t)stream);
}

int fva_conv_fwd_bias_act(const fva_conv_desc* d, const void* x, const void* w_fwd, const float* bias, int32_t act, void* z, int32_t z_pad,
                          void* stream) {
    IgemmParams p;
    int rc = setup_fwd(d, p, "fva_conv_fwd_bias_act");
    if (rc) return rc;
    if (!x || !w_fwd || !bias || !z) return fva_fail(FVA_ERR_ARG, "fva_conv_fwd_bias_act: null pointer");
    if (d->Cout % 8 || z_pad < 0 || act < 1 || act > 2) return fva_fail(FVA_ERR_ARG, "fva_conv_fwd_bias_act: Cout %d not a multiple of 8, bad z_pad or act", d->Cout);
    const int OH = (d->H - 1) / d->stride + 1, OW = (d->W - 1) / d->stride + 1;
    p.in = x;
    p.wt = w_fwd;
    p.out = z;
    p.scale = nullptr;
    p.shift = bias;
    p.act = act;
    p.addend = nullptr;
    p.out_dense = 0;
    p.out_row = OW + 2 * z_pad;
    p.out_img = (OH + 2 * z_pad) * p.out_row;
    p.osy = p.osx = 1;
    p.ooy = p.oox = z_pad;
    FvaProfileSpan span(0 | (d->ksize << 8), 2.0 * p.M * (double)d->Cout * d->Cin * d->ksize * d->ksize, (hipStream_t)stream);
    rc = launch_igemm<EPI_BNACT>(d->dtype, p, (hipStream_t)stream);
    if (rc || z_pad == 0) return rc;
    return fva_zero_halo_border(z, d->B, OH, OW, d->Cout * (d->dtype == FVA_BF16 ? 2 : 4) / 16, z_pad, (hipStream_t)stream);
}

int fva_head_fwd(const fva_conv_desc* d, const void* x, const void* w_fwd, const float* bias, float* out, void* stream) {
    IgemmParams p;
    int rc = setup_fwd(d, p, "fva_head_fwd");
    if (rc) return rc;
    if (d->ksize != 1) return fva_fail(FVA_ERR_ARG, "fva_head_fwd: ksize must be 1");
    if (!x || !w_fwd || !bias || !out) return fva_fail(FVA_ERR_ARG, "fva_head_fwd: null pointer");
    p.in = x;
    p.wt = w_fwd;
    p.out = out;
    p.bias = bias;
    FvaProfileSpan span(0 | (d->ksize << 8), 2.0 * p.M * (double)d->Cout * d->Cin * d->ksize * d->ksize, (hipStream_t)stream);
    return launch_igemm<EPI_HEAD>(d->dtype, p, (hipStream_t)stream);
}

// row blocks (BM of the tile the dispatcher picks) of ONE dgrad launch with m rows, n columns, reduction over c channels x ntaps
static int dgrad_launch_rows(const fva_conv_desc* d, int64_t m, int n, int c, int ntaps, int64_t in_pixels) {
    if (!halfrow_mode(d->dtype, c) && use_igemm8(d->dtype, m, n, c, ntaps, in_pixels)) return cdiv(m, 256);
    return cdiv(m, tile_bm(d->dtype, (int)m, n));
}

int32_t fva_conv_dgrad_stat_rows(const fva_conv_desc* d) {
    if (!d || check_desc(d, "fva_conv_dgrad_stat_rows")) return 0;
    const int k = d->ksize, s = d->stride;
    const int OH = (d->H - 1) / s + 1, OW = (d->W - 1) / s + 1;
    const int64_t in_pixels = (int64_t)(d->B + 1) * (OH + 2 * d->dy_pad) * (OW + 2 * d->dy_pad);
    if (use_pconv(d->dtype, k, s, d->Cout, d->Cin, d->H, d->W)) return pconv_tiles(d->B, d->H, d->W);
    if (s == 1) return dgrad_launch_rows(d, (int64_t)d->B * d->H * d->W, d->Cin, d->Cout, k * k, in_pixels);
    const int64_t mq = (int64_t)d->B * (d->H / 2) * (d->W / 2);
    if (use_pdgrad2(d->dtype, k, s, d->Cout, d->Cin)) return pdgrad2_tiles(d->B, d->H, d->W);
    if (dgrad_paired(d->dtype, k, s, d->Cin, d->Cout)) {   // two launches (row parity) of N' = 2 * Cin columns: two table rows per block
        int rows = 0;
        for (int py = 0; py < 2; ++py) rows += 2 * dgrad_launch_rows(d, mq, 2 * d->Cin, d->Cout, py == 0 ? 2 : 4, in_pixels);
        return rows;
    }
    int rows = 0;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) rows += dgrad_launch_rows(d, mq, d->Cin, d->Cout, (py ? 2 : 1) * (px ? 2 : 1), in_pixels);
    return rows;
}

int fva_conv_dgrad(const fva_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const void* addend, void* stream) {
    return fva_conv_dgrad_bnstats(d, dy, w_dgrad, dx, addend, nullptr, stream);
}

int fva_conv_dgrad_bnstats(const fva_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const void* addend,
                           const fva_bn_bwd_fuse* f, void* stream) {
    int rc = check_desc(d, "fva_conv_dgrad");
    if (rc) return rc;
    rc = check_red_channels(d->dtype, d->Cout, "fva_conv_dgrad");
    if (rc) return rc;
    if (!dy || !w_dgrad || !dx) return fva_fail(FVA_ERR_ARG, "fva_conv_dgrad: null pointer");
    if (d->Cin % 8) return fva_fail(FVA_ERR_ARG, "fva_conv_dgrad: Cin %d not a multiple of 8", d->Cin);
    const int k = d->ksize, s = d->stride, pd = k / 2;
    if (d->dy_pad < pd) return fva_fail(FVA_ERR_ARG, "fva_conv_dgrad: dy_pad %d < %d", d->dy_pad, pd);
    const int OH = (d->H - 1) / s + 1, OW = (d->W - 1) / s + 1;
    FvaProfileSpan span(1 | (k << 8), 2.0 * d->B * OH * OW * (double)d->Cout * d->Cin * k * k, (hipStream_t)stream);
    IgemmParams p = IgemmParams();
    p.acc_rep = p.bnb_rep = p.ax_rep = 1;
    p.in = dy;
    p.wt = w_dgrad;
    p.out = dx;
    p.N = d->Cin;
    p.C = d->Cout;
    p.in_row = OW + 2 * d->dy_pad;
    p.in_img = (OH + 2 * d->dy_pad) * p.in_row;
    p.sy = p.sx = 1;
    p.addend = addend;
    p.out_pitch = d->Cin;
    if (f) {
        if (!f->y || !f->scale || !f->shift || !f->mean || !f->rstd || (!f->partial && !f->acc)) return fva_fail(FVA_ERR_ARG, "fva_conv_dgrad_bnstats: null pointer");
        p.bnb_y = f->y; p.bnb_scale = f->scale; p.bnb_shift = f->shift; p.bnb_mean = f->mean; p.bnb_rstd = f->rstd;
        p.bnb_part = f->partial;
        p.bnb_acc = (long long*)f->acc;
        p.bnb_rep = f->acc ? f->acc_replicas : 1;
        if (f->acc && !fva_replicas_ok(f->acc_replicas)) return fva_fail(FVA_ERR_ARG, "fva_conv_dgrad_bnstats: acc_replicas = %d is not a power of two in 1..%d", f->acc_replicas, FVA_BN_ACC_MAX_REPLICAS);
        p.bnb_row0 = 0;
        p.bnb_C = d->Cin;
    }
    const int64_t in_pixels = (int64_t)(d->B + 1) * (OH + 2 * d->dy_pad) * (OW + 2 * d->dy_pad);
    if (s == 1) {
        p.M = d->B * d->H * d->W;
        p.OW = d->W;
        p.OHW = d->H * d->W;
        p.div_ow = make_fastdiv(p.OW);
        p.div_ohw = make_fastdiv(p.OHW);
        p.y0 = p.x0 = d->dy_pad - pd;
        int nt = 0;
        for (int kh = 0; kh < k; ++kh)
            for (int kw = 0; kw < k; ++kw) {
                p.tap_pix[nt] = (2 * pd - kh) * p.in_row + (2 * pd - kw);
                p.tap_w[nt] = kh * k + kw;
                ++nt;
            }
        p.tap_w[nt] = k * k;
        finish_taps(p, nt, d->dtype);
        p.out_dense = 1;
        if (use_pconv(d->dtype, k, s, d->Cout, d->Cin, d->H, d->W))
            return f ? launch_pconv<EPI_BNB>(p, d->B, d->H, d->W, true, (hipStream_t)stream) : launch_pconv<EPI_PLAIN>(p, d->B, d->H, d->W, true, (hipStream_t)stream);
        return f ? launch_igemm<EPI_BNB>(d->dtype, p, (hipStream_t)stream) : launch_igemm<EPI_PLAIN>(d->dtype, p, (hipStream_t)stream);
    }
    const int JH = d->H / 2, JW = d->W / 2;
    if (use_pdgrad2(d->dtype, k, s, d->Cout, d->Cin)) {
        p.y0 = p.x0 = d->dy_pad;          // dy pixel (i, j) sits at padded (i + pad, j + pad); rows / columns up to OH / OW are the zero halo
        p.out_dense = 1;
        return f ? launch_pdgrad2<EPI_BNB>(p, d->B, d->H, d->W, (hipStream_t)stream) : launch_pdgrad2<EPI_PLAIN>(p, d->B, d->H, d->W, (hipStream_t)stream);
    }
    if (dgrad_paired(d->dtype, k, s, d->Cin, d->Cout)) {
        // thin layers: both x-parities of an output row per launch (see dgrad_paired) -- 2 launches, N' = 2*Cin
        p.M = d->B * JH * JW;
        p.N = 2 * d->Cin;
        p.OW = JW;
        p.OHW = JH * JW;
        p.div_ow = make_fastdiv(p.OW);
        p.div_ohw = make_fastdiv(p.OHW);
        p.y0 = p.x0 = 0;
        p.out_dense = 0;
        p.out_img = d->H * d->W;
        p.out_row = d->W;
        p.osy = p.osx = 2;
        p.oox = 0;
        for (int py = 0; py < 2; ++py) {
            int nt = 0;
            const int nky = py == 0 ? 1 : 2;
            const int kys[2] = {py == 0 ? 1 : 0, 2};
            const int yos[2] = {py == 0 ? d->dy_pad : d->dy_pad + 1, d->dy_pad};
            for (int a = 0; a < nky; ++a)
                for (int dxo = 0; dxo < 2; ++dxo) {
                    p.tap_pix[nt] = yos[a] * p.in_row + d->dy_pad + dxo;
                    p.tap_w[nt] = kys[a] * 2 + dxo;
                    ++nt;
                }
            p.tap_w[nt] = 0;
            finish_taps(p, nt, d->dtype);
            if (p.halfrow) return fva_fail(FVA_ERR_ARG, "fva_conv_dgrad: paired stride-2 path needs Cout >= 64");
            p.ooy = py;
            rc = f ? launch_igemm<EPI_BNB>(d->dtype, p, (hipStream_t)stream) : launch_igemm<EPI_PLAIN>(d->dtype, p, (hipStream_t)stream);
            if (rc) return rc;
            p.bnb_row0 += dgrad_launch_rows(d, p.M, p.N, p.C, nt, in_pixels);
        }
        return FVA_OK;
    }
    // stride 2, 3x3: four output-parity classes, each with its own tap subset (no wasted MACs)
    p.M = d->B * JH * JW;
    p.OW = JW;
    p.OHW = JH * JW;
    p.div_ow = make_fastdiv(p.OW);
    p.div_ohw = make_fastdiv(p.OHW);
    p.y0 = p.x0 = 0;
    p.out_dense = 0;
    p.out_img = d->H * d->W;
    p.out_row = d->W;
    p.osy = p.osx = 2;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            int khs[2], yo[2], nky, kws[2], xo[2], nkx;
            if (py == 0) { nky = 1; khs[0] = 1; yo[0] = d->dy_pad; }
            else { nky = 2; khs[0] = 0; yo[0] = d->dy_pad + 1; khs[1] = 2; yo[1] = d->dy_pad; }
            if (px == 0) { nkx = 1; kws[0] = 1; xo[0] = d->dy_pad; }
            else { nkx = 2; kws[0] = 0; xo[0] = d->dy_pad + 1; kws[1] = 2; xo[1] = d->dy_pad; }
            int nt = 0;
            for (int a = 0; a < nky; ++a)
                for (int b = 0; b < nkx; ++b) {
                    p.tap_pix[nt] = yo[a] * p.in_row + xo[b];
                    p.tap_w[nt] = khs[a] * 3 + kws[b];
                    ++nt;
                }
            p.tap_w[nt] = 9;
            finish_taps(p, nt, d->dtype);
            p.ooy = py;
            p.oox = px;
            rc = f ? launch_igemm<EPI_BNB>(d->dtype, p, (hipStream_t)stream) : launch_igemm<EPI_PLAIN>(d->dtype, p, (hipStream_t)stream);
            if (rc) return rc;
            p.bnb_row0 += dgrad_launch_rows(d, p.M, p.N, p.C, nt, in_pixels);
        }
    return FVA_OK;
}

}  // extern "C"
