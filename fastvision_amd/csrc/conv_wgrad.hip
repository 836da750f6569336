// Weight gradient of the convolution on MFMA for gfx950 (split-K over pixels, deterministic reduce).
//
//   dW[t][n][c] = sum_m dY[m][n] * X[pix(m) + tap_pix[t]][c]          m = (b, oy, ox)
//
// Both operands are pixel-major NHWC (channels contiguous), i.e. the contraction index m is the SLOW axis of
// both.  Tiles of 64 pixels x 256 bytes of channels are staged by LDS-DMA; the bf16 MFMA fragments (8
// consecutive k per lane) are read with ds_read_b64_tr_b16, the gfx950 transposing LDS read, from an
// XOR-swizzled image (conflict-free for the 16x16x32 operand).  fp32 uses v_mfma_f32_32x32x2_f32 whose
// operands are one float per lane, read straight out of the pixel-major image.
// Each block owns one [n-tile x c-tile] of one tap over one pixel range and writes an fp32 partial tile to
// the workspace slab [ksplit][tap][N][C]; a second kernel sums the partials in fixed order into the OIHW
// fp32 gradient.  Roofline: MFMA-bound (2*M*N*C*taps flop).
//
// Replaces the backward-weight of nn.Conv2d (reference classfication/models/darknet53.py:5-9 via autograd).
#include <type_traits>

#include "common.h"

namespace {

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_vmcnt_n() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The partial tiles are written with ordinary stores: non-temporal ones LOSE (measured in round 2: wgrad class 8.5 -> 9.15 ms), because
// the reduce kernel that follows finds the slabs in cache only when they were written the ordinary way.
__device__ __forceinline__ void slab_store(float* p, float v) { *p = v; }

struct WgradParams {
    const void* x;
    const void* dy;
    float* slab;
    int M, N, C;
    int OW, OHW;
    FastDiv div_ow, div_ohw;
    int x_img, x_row, sy, sx, x_y0, x_x0;
    int dy_img, dy_row, dy_pad, dy_pitch;
    uint32_t dy_zero_off;  // byte offset of a pixel that is guaranteed zero (halo corner)
    int ntaps;
    int tap_pix[9];
    int ntn, ntc, ksplit, mchunk;
    int tpt, ngroups;  // taps packed side by side in one column tile (thin layers: C < tile), tap groups
    int x_pix_bytes;   // bytes between consecutive pixels of x; 0 = C * sizeof(T) (the stem reads overlapping 4-pixel windows)
    long long* stamps; // diagnostic (fva_conv_debug_stamps): per block 4 wall-clock + 4 cycle-counter values, 8-phase kernel only
    int stamp_rows;    // capacity of the caller's buffer in blocks
};

// ds_read_b64_tr_b16 as inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of the builtin form whenever an LDS-DMA
// is in flight, which would serialise the next tile's DMA with this tile's MFMAs.  The asm form is invisible to
// that bookkeeping; its completion is awaited by hand (`s_waitcnt lgkmcnt(0)` + sched_barrier) before the MFMAs.
template <int OFF>
__device__ __forceinline__ s16x4 tr_read(uint32_t addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
using s16x8 = __attribute__((ext_vector_type(8))) short;
__device__ __forceinline__ bf16x8 cat8(s16x4 lo, s16x4 hi) {
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// THIN (bf16, Cout <= 64): the four waves split the tile's 128 COLUMNS (32 each) and all take rows 0..63, instead of 2 x 2 waves of
// 64 x 64 -- with 64 output channels the two lower waves of the square arrangement multiplied rows that do not exist (the
// 32 -> 64 layers at 320^2 executed 2.7 x their useful MFMAs and ran at 276 TFLOP/s).
template <typename T, bool THIN = false>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
    static_assert(!THIN || sizeof(T) == 2, "the thin arrangement is a bf16 variant");
    constexpr bool IS_BF16 = sizeof(T) == 2;
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int TILE = 16 * EPC;  // channels per tile side: 128 bf16 / 64 f32 (256-byte rows)
    constexpr int BKP = 64;         // pixels per k-step
    constexpr int OP_BYTES = BKP * 256, STAGE = 2 * OP_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = THIN ? 0 : w >> 1, wc = THIN ? w : w & 1;
    constexpr int NTW = 4, CTW = THIN ? 2 : 4;     // 16 x 16 accumulator tiles per wave along n / along the columns
    constexpr int WCOLS = CTW * 16;                // columns per wave

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, xq = nwg >> 3, xr = nwg & 7;
    int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int per_ks = p.ngroups * p.ntn * p.ntc;
    const int ks = logical / per_ks;
    logical -= ks * per_ks;
    const int tg = logical / (p.ntn * p.ntc);  // tap group (a single tap when tpt == 1)
    logical -= tg * (p.ntn * p.ntc);
    const int tn = logical / p.ntc, tc = logical - tn * p.ntc;
    const int n0 = tn * TILE, c0 = tc * TILE;
    const int mbeg = ks * p.mchunk;
    const int mend = (mbeg + p.mchunk < p.M) ? mbeg + p.mchunk : p.M;
    const int steps = p.mchunk / BKP;

    // ---- LDS-DMA source mapping: one instruction = 4 pixel rows x 256 B; lane -> (row, 16-B chunk) ---------
    const int lrow = lane >> 4;
    const int f = IS_BF16 ? ((lrow << 2) | w) : 0;  // swizzle of this lane's rows: ((R&3)<<2)|((R>>2)&3)
    const int chunk = (lane & 15) ^ f;
    const uint32_t dy_pixb = (uint32_t)p.dy_pitch * (uint32_t)sizeof(T);
    const uint32_t x_pixb = p.x_pix_bytes ? (uint32_t)p.x_pix_bytes : (uint32_t)p.C * (uint32_t)sizeof(T);
    const bool a_ok = n0 + chunk * EPC < p.N;  // columns beyond N / C feed only unused outputs: point them
    const uint32_t a_colb = (uint32_t)(a_ok ? n0 + chunk * EPC : 0) * (uint32_t)sizeof(T);  // at column 0 (in bounds)
    // B column -> (tap, channel): with tpt > 1 the tile holds tpt taps of all C channels side by side
    const int jcol = c0 + chunk * EPC;
    const int tsub = p.tpt > 1 ? jcol / p.C : 0;
    const int ccol = p.tpt > 1 ? jcol - tsub * p.C : jcol;
    const int my_tap = tg * p.tpt + tsub;
    const bool b_ok = my_tap < p.ntaps && ccol < p.C;
    int tpix = 0;  // tap pixel offset without dynamic kernarg indexing
#pragma unroll
    for (int i = 0; i < 9; ++i)
        if (i == (b_ok ? my_tap : tg * p.tpt)) tpix = p.tap_pix[i];
    const uint32_t b_colb = (uint32_t)tpix * x_pixb + (uint32_t)(b_ok ? ccol : 0) * (uint32_t)sizeof(T);

    // Address generation: each lane decodes ONE pixel row per k-step (the row owned by lane & 15) and the four
    // rows a lane needs for its DMA instructions are fetched from their owner lanes by wave shuffles.
    const int own_row = ((((lane & 15) >> 2) * 4 + w) * 4) + (lane & 3);
    const char* dy_base = (const char*)p.dy;
    const char* x_base = (const char*)p.x;
    uint32_t nx_dy[4], nx_x[4];  // DMA byte offsets of the NEXT tile to issue (decoded one iteration ahead)
    auto decode_step = [&](int step) {
        int m = mbeg + step * BKP + own_row;
        const bool live = m < mend;
        m = m < p.M ? m : p.M - 1;
        const uint32_t b = fd_div((uint32_t)m, p.div_ohw);
        const uint32_t rem = (uint32_t)m - b * (uint32_t)p.OHW;
        const uint32_t oy = fd_div(rem, p.div_ow);
        const uint32_t ox = rem - oy * (uint32_t)p.OW;
        const uint32_t dpix = b * (uint32_t)p.dy_img + (oy + p.dy_pad) * (uint32_t)p.dy_row + (ox + p.dy_pad);
        const uint32_t xpix = b * (uint32_t)p.x_img + (oy * p.sy + p.x_y0) * (uint32_t)p.x_row + (ox * p.sx + p.x_x0);
        const uint32_t own_dy = live ? dpix * dy_pixb : p.dy_zero_off;
        const uint32_t own_x = xpix * x_pixb;
        // lanes whose columns do not exist (N or C*taps smaller than the tile) all read ONE 16-byte piece at the start of
        // the buffer instead of a real row: their products land in outputs that are never stored, and the dead half of a
        // thin layer's tile then costs no L2 / HBM traffic (32->64 layers: half of the dY rows, a quarter of the X slots)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t rd = (uint32_t)__shfl((int)own_dy, i * 4 + lrow), rx = (uint32_t)__shfl((int)own_x, i * 4 + lrow);
            nx_dy[i] = a_ok ? rd + a_colb : 0u;
            nx_x[i] = b_ok ? rx + b_colb : 0u;
        }
    };
    auto issue_step = [&](int stage) {
        char* sA = smem + stage * STAGE;
        char* sB = sA + OP_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(dy_base + nx_dy[i]), LDS_PTR(sA + (i * 4 + w) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(x_base + nx_x[i]), LDS_PTR(sB + (i * 4 + w) * 1024), 16, 0, 0);
        }
    };

    f32x4 acc16[4][4];
    f32x16 acc32;
    if constexpr (IS_BF16) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[e] = 0.f;
    }

    decode_step(0);
    issue_step(0);
    if (steps > 1) decode_step(1);
    if constexpr (IS_BF16) {
        // both stages are free at entry: the second k-step's DMA does not wait for the first barrier (a block of a 1x1 layer
        // runs 7-13 k-steps, every exposed round trip counts)
        if (steps > 1) {
            issue_step(1);
            if (steps > 2) decode_step(2);
        }
        // lane-constant LDS addresses of the transposed reads (see the mapping note below); stage, operand and the
        // 32-pixel half of the tile go into the instruction's immediate offset, so the loop does no address math
        const int i16 = lane & 15, g = lane >> 4, q = i16 >> 2, pp = i16 & 3;
        const uint32_t lds0 = (uint32_t)(size_t)LDS_PTR(smem);
        uint32_t ra[2][NTW], rb[2][CTW];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int row = 8 * g + 4 * hh + q;                 // + 32 * kk
            const int fr = (q << 2) | ((2 * g + hh) & 3);       // swizzle of that row (unchanged by + 32)
            const int rbase = row * 256 + 8 * (pp & 1);
#pragma unroll
            for (int t = 0; t < NTW; ++t) ra[hh][t] = lds0 + rbase + (((wr * 8 + t * 2 + (pp >> 1)) ^ fr) << 4);
#pragma unroll
            for (int t = 0; t < CTW; ++t) rb[hh][t] = lds0 + rbase + (((wc * (WCOLS / 8) + t * 2 + (pp >> 1)) ^ fr) << 4);
        }
        // transposed read: lane 4q+pp of each 16-lane group addresses row q, columns 4pp..4pp+3 of a 4-row x 16-column
        // block and receives column (lane&15) of the 4 rows; two reads (k = 8g+0..3, 8g+4..7) make one MFMA fragment
        auto iteration = [&](int st, auto stage_tag) {
            constexpr int SO = decltype(stage_tag)::value * STAGE;
            if (st == 0 && steps > 1) wait_vmcnt_n<8>();   // k-step 1 (8 DMA instructions) stays in flight
            else wait_vm0();
            __builtin_amdgcn_s_barrier();
            if (st >= 1 && st + 1 < steps) issue_step(decltype(stage_tag)::value ^ 1);
            bf16x8 af[2][NTW], bfr[2][CTW];
#define FVA_TR_A(KK, TT) af[KK][TT] = cat8(tr_read<SO + KK * 8192>(ra[0][TT]), tr_read<SO + KK * 8192>(ra[1][TT]));
#define FVA_TR_B(KK, TT) bfr[KK][TT] = cat8(tr_read<SO + OP_BYTES + KK * 8192>(rb[0][TT]), tr_read<SO + OP_BYTES + KK * 8192>(rb[1][TT]));
            FVA_TR_A(0, 0) FVA_TR_B(0, 0) FVA_TR_A(0, 1) FVA_TR_B(0, 1) FVA_TR_A(0, 2) FVA_TR_A(0, 3)
            if constexpr (!THIN) { FVA_TR_B(0, CTW - 2) FVA_TR_B(0, CTW - 1) }
            FVA_TR_A(1, 0) FVA_TR_B(1, 0) FVA_TR_A(1, 1) FVA_TR_B(1, 1) FVA_TR_A(1, 2) FVA_TR_A(1, 3)
            if constexpr (!THIN) { FVA_TR_B(1, CTW - 2) FVA_TR_B(1, CTW - 1) }
#undef FVA_TR_A
#undef FVA_TR_B
            if (st >= 1 && st + 2 < steps) decode_step(st + 2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int ct = 0; ct < CTW; ++ct)
                        acc16[nt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][nt], bfr[kk][ct], acc16[nt][ct], 0, 0, 0);
        };
        for (int st = 0; st < steps; st += 2) {
            iteration(st, std::integral_constant<int, 0>{});
            if (st + 1 < steps) iteration(st + 1, std::integral_constant<int, 1>{});
        }
    } else {
        for (int st = 0; st < steps; ++st) {
            __syncthreads();
            if (st + 1 < steps) {
                issue_step((st + 1) & 1);
                if (st + 2 < steps) decode_step(st + 2);
            }
            const char* sA = smem + (st & 1) * STAGE;
            const char* sB = sA + OP_BYTES;
            const int r = lane & 31, h = lane >> 5;
            const float* fa = (const float*)sA + wr * 32 + r;
            const float* fb = (const float*)sB + wc * 32 + r;
#pragma unroll 8
            for (int k2 = 0; k2 < 32; ++k2) {
                const float a = fa[(2 * k2 + h) * 64];
                const float b = fb[(2 * k2 + h) * 64];
                acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32, 0, 0, 0);
            }
        }
    }

    // ---- partial tile -> slab[ks][tap][N][C] ------------------------------------------------------------
    float* out = p.slab + (int64_t)ks * p.ntaps * p.N * p.C;
    auto store = [&](int n, int j, float v) {  // j = column inside the tile
        int tap, c;
        if (p.tpt > 1) {
            const int ts = j / p.C;
            tap = tg * p.tpt + ts;
            c = j - ts * p.C;
        } else {
            tap = tg;
            c = c0 + j;
        }
        if (n < p.N && c < p.C && tap < p.ntaps) slab_store(out + ((int64_t)tap * p.N + n) * p.C + c, v);
    };
    if constexpr (IS_BF16) {
        const int r = lane & 15, g = lane >> 4;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j) store(n0 + wr * 64 + nt * 16 + g * 4 + j, wc * WCOLS + ct * 16 + r, acc16[nt][ct][j]);
    } else {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int e = 0; e < 16; ++e) store(n0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, wc * 32 + r, acc32[e]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 256x256 tile, 8 waves, the 8-phase schedule of conv_igemm.hip's igemm8_kernel applied to the weight gradient (bf16, 3x3
// layers with Cout % 256 == 0 and Cin = 128 or a multiple of 256): one block per CU, two LDS buffers of four 16-KiB
// half-tile images [64 pixels][128 channels] (the operand image of wgrad_kernel, so the ds_read_b64_tr_b16 fragment
// addressing is the same), a 64-pixel k-step consumed in four phases of 16 MFMAs, one half-tile of LDS-DMA per phase,
// vmcnt(6) once per k-step, the two wave groups (wr = 0 / 1) half a phase apart.
//   A half h = dY channels n0 + wr*128 + h*64 + [0, 64) of both wave rows; B half h = X columns c0 + wc*64 + h*32 + [0, 32)
//   of the four wave columns (with Cin = 128 a column tile holds two taps side by side).
//   q0: read B-h0, A-h0 | stage A-h1(s+1)   q1: read B-h1 | stage B-h0(s+2)   q2: read A-h1 | stage A-h0(s+2)
//   q3: decode the pixels of step s+3 | stage B-h1(s+2), vmcnt(6).  RAW / WAR as in igemm8_kernel.
__global__ __launch_bounds__(512) void wgrad8_kernel(const WgradParams p) {
    constexpr int BKP = 64, OP = BKP * 256, BUF = 4 * OP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 2, wc = w & 3;
    auto stamp = [&](int i) {
        if (p.stamps && tid == 0 && (int)blockIdx.x < p.stamp_rows) {
            p.stamps[(int64_t)blockIdx.x * 8 + i] = wall_clock64();
            p.stamps[(int64_t)blockIdx.x * 8 + 4 + i] = clock64();
        }
    };
    stamp(0);

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, xq = nwg >> 3, xr = nwg & 7;
    int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int per_ks = p.ngroups * p.ntn * p.ntc;
    const int ks = logical / per_ks;
    logical -= ks * per_ks;
    const int tg = logical / (p.ntn * p.ntc);
    logical -= tg * (p.ntn * p.ntc);
    const int tn = logical / p.ntc, tc = logical - tn * p.ntc;
    const int n0 = tn * 256, c0 = tc * 256;
    const int mbeg = ks * p.mchunk;
    const int mend = (mbeg + p.mchunk < p.M) ? mbeg + p.mchunk : p.M;
    const int steps = p.mchunk / BKP;

    // ---- LDS-DMA source mapping: one instruction = 4 pixel rows x 256 B of a half-tile image ------------------------
    const int lrow = lane >> 4;
    const int f = (lrow << 2) | (w & 3);          // swizzle of this lane's rows R = (i*8 + w)*4 + lrow
    const int chunk = (lane & 15) ^ f;            // image chunk (8 channels) whose data lands at LDS slot (lane & 15)
    const uint32_t dy_pixb = (uint32_t)p.dy_pitch * 2u, x_pixb = (uint32_t)p.C * 2u;
    uint32_t a_colb[2], b_colb[2];
    bool b_live[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        a_colb[h] = (uint32_t)(n0 + (chunk >> 3) * 128 + h * 64 + (chunk & 7) * 8) * 2u;
        const int jcol = c0 + (chunk >> 2) * 64 + h * 32 + (chunk & 3) * 8;
        const int tsub = p.tpt > 1 ? jcol / p.C : 0;
        const int ccol = p.tpt > 1 ? jcol - tsub * p.C : jcol;
        const int my_tap = tg * p.tpt + tsub;
        const bool b_ok = my_tap < p.ntaps && ccol < p.C;   // columns of a tap that does not exist feed unused outputs
        int tpix = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i)
            if (i == (b_ok ? my_tap : tg * p.tpt)) tpix = p.tap_pix[i];
        b_colb[h] = (uint32_t)tpix * x_pixb + (uint32_t)(b_ok ? ccol : 0) * 2u;
        b_live[h] = b_ok;   // dead columns (the 10th tap slot of a two-taps-per-tile layer) read one shared 16-byte piece
    }

    // each wave's DMA touches 8 pixel rows per step; lanes 0..7 decode one each, the others fetch by shuffle
    const int own_row = (((lane >> 2) & 1) * 8 + w) * 4 + (lane & 3);
    const char* dy_base = (const char*)p.dy;
    const char* x_base = (const char*)p.x;
    auto decode = [&](int step, uint32_t (&dyo)[2], uint32_t (&xo)[2]) {
        int m = mbeg + step * BKP + own_row;
        const bool live = m < mend;
        m = m < p.M ? m : p.M - 1;
        const uint32_t b = fd_div((uint32_t)m, p.div_ohw);
        const uint32_t rem = (uint32_t)m - b * (uint32_t)p.OHW;
        const uint32_t oy = fd_div(rem, p.div_ow);
        const uint32_t ox = rem - oy * (uint32_t)p.OW;
        const uint32_t dpix = b * (uint32_t)p.dy_img + (oy + p.dy_pad) * (uint32_t)p.dy_row + (ox + p.dy_pad);
        const uint32_t xpix = b * (uint32_t)p.x_img + (oy * p.sy + p.x_y0) * (uint32_t)p.x_row + (ox * p.sx + p.x_x0);
        const uint32_t own_dy = live ? dpix * dy_pixb : p.dy_zero_off;
        const uint32_t own_x = xpix * x_pixb;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            dyo[i] = (uint32_t)__shfl((int)own_dy, i * 4 + lrow);
            xo[i] = (uint32_t)__shfl((int)own_x, i * 4 + lrow);
        }
    };
    // kind: 0 A-h0, 1 A-h1, 2 B-h0, 3 B-h1
    auto stage = [&](int kind, int buf, const uint32_t (&dyo)[2], const uint32_t (&xo)[2]) {
        char* dst = smem + buf * BUF + kind * OP + w * 1024;
        const int h = kind & 1;
        if (kind < 2) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(dy_base + (dyo[0] + a_colb[h])), LDS_PTR(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(dy_base + (dyo[1] + a_colb[h])), LDS_PTR(dst + 8192), 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds(GLB_PTR(x_base + (b_live[h] ? xo[0] + b_colb[h] : 0u)), LDS_PTR(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLB_PTR(x_base + (b_live[h] ? xo[1] + b_colb[h] : 0u)), LDS_PTR(dst + 8192), 16, 0, 0);
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);

    // lane-constant addresses of the transposed reads inside a half-tile image (see wgrad_kernel)
    const int i16 = lane & 15, g = lane >> 4, q = i16 >> 2, pp = i16 & 3;
    const uint32_t lds0 = (uint32_t)(size_t)LDS_PTR(smem);
    uint32_t ra[2][4], rb[2][2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int row = 8 * g + 4 * hh + q;
        const int fr = (q << 2) | ((2 * g + hh) & 3);
        const int rbase = row * 256 + 8 * (pp & 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) ra[hh][t] = lds0 + rbase + (((wr * 8 + t * 2 + (pp >> 1)) ^ fr) << 4);
#pragma unroll
        for (int t = 0; t < 2; ++t) rb[hh][t] = lds0 + rbase + (((wc * 4 + t * 2 + (pp >> 1)) ^ fr) << 4);
    }
    bf16x8 af[4][2], b0[2][2], b1[2][2];
    auto read_a = [&](uint32_t boff, auto kind_tag) {
        constexpr int KO = decltype(kind_tag)::value * OP;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            af[mt][0] = cat8(tr_read<KO>(ra[0][mt] + boff), tr_read<KO>(ra[1][mt] + boff));
            af[mt][1] = cat8(tr_read<KO + 8192>(ra[0][mt] + boff), tr_read<KO + 8192>(ra[1][mt] + boff));
        }
    };
    auto read_b = [&](uint32_t boff, auto kind_tag, bf16x8 (&bf)[2][2]) {
        constexpr int KO = decltype(kind_tag)::value * OP;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            bf[nt][0] = cat8(tr_read<KO>(rb[0][nt] + boff), tr_read<KO>(rb[1][nt] + boff));
            bf[nt][1] = cat8(tr_read<KO + 8192>(rb[0][nt] + boff), tr_read<KO + 8192>(rb[1][nt] + boff));
        }
    };
    auto mma = [&](int ha, int hb, bf16x8 (&bf)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[ha * 4 + mt][hb * 2 + nt] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt][kk], bf[nt][kk], acc[ha * 4 + mt][hb * 2 + nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // the transposed reads are inline asm (invisible to the compiler's counters): retire them by hand before the barrier
    // that precedes the MFMAs, and keep the MFMAs behind it
    auto retire_reads_then_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;

    // ---- prologue: step 0 complete, three half-tiles of step 1 in flight; offsets of steps 1 and 2 decoded ------------
    uint32_t d1[2], x1[2], d2[2], x2[2];
    {
        uint32_t d0[2], x0[2];
        decode(0, d0, x0);
        stage(2, 0, d0, x0); stage(0, 0, d0, x0); stage(3, 0, d0, x0); stage(1, 0, d0, x0);
    }
    decode(1, d1, x1);
    decode(2, d2, x2);
    if (steps > 1) {
        stage(2, 1, d1, x1); stage(0, 1, d1, x1); stage(3, 1, d1, x1);
        wait_vmcnt_n<6>();
    } else {
        wait_vmcnt_n<0>();
    }
    __builtin_amdgcn_s_barrier();
    stamp(1);
    if (wr == 1) __builtin_amdgcn_s_barrier();

    for (int s = 0; s < steps; ++s) {
        const int cur = s & 1;
        const uint32_t boff = (uint32_t)cur * BUF;
        // q0
        read_b(boff, K2{}, b0);
        read_a(boff, K0{});
        if (s + 1 < steps) stage(1, cur ^ 1, d1, x1);
        retire_reads_then_barrier();
        mma(0, 0, b0);
        __builtin_amdgcn_s_barrier();
        // q1
        read_b(boff, K3{}, b1);
        if (s + 2 < steps) stage(2, cur, d2, x2);
        retire_reads_then_barrier();
        mma(0, 1, b1);
        __builtin_amdgcn_s_barrier();
        // q2
        read_a(boff, K1{});
        if (s + 2 < steps) stage(0, cur, d2, x2);
        retire_reads_then_barrier();
        mma(1, 1, b1);
        __builtin_amdgcn_s_barrier();
        // q3
        if (s + 2 < steps) {
            stage(3, cur, d2, x2);
            wait_vmcnt_n<6>();
        } else {
            wait_vmcnt_n<0>();
        }
        d1[0] = d2[0]; d1[1] = d2[1]; x1[0] = x2[0]; x1[1] = x2[1];
        decode(s + 3, d2, x2);                       // under this phase's MFMAs
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 0, b0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    stamp(2);

    // ---- partial tile -> slab[ks][tap][N][C] --------------------------------------------------------------------------
    float* out = p.slab + (int64_t)ks * p.ntaps * p.N * p.C;
    const int r = lane & 15;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int j = wc * 64 + ni * 16 + r;     // column inside the tile
            int tap, c;
            if (p.tpt > 1) {
                const int ts = j / p.C;
                tap = tg * p.tpt + ts;
                c = j - ts * p.C;
            } else {
                tap = tg;
                c = c0 + j;
            }
            if (c < p.C && tap < p.ntaps) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int n = n0 + wr * 128 + mi * 16 + g * 4 + jj;
                    if (n < p.N) slab_store(out + ((int64_t)tap * p.N + n) * p.C + c, acc[mi][ni][jj]);
                }
            }
        }
    stamp(3);
}

// ---------------------------------------------------------------------------------------------------------------------
// All-taps weight gradient of the thin 3x3 layers (32 -> 64 and 64 -> 128 channels, stride 1 and 2, bf16; round 3).
// wgrad_kernel gives these layers a 128 x 128 tile per tap group: with Cin = 32 four taps share a column tile (three groups for nine
// taps, the last a quarter full), every group re-reads the dY tile, and each tap's X rows are fetched separately -- 360-380 us for
// 0.12 TFLOP.  Here a block of four waves owns a [64][9 Cin] slice of the gradient in its accumulators (all of it for Cout = 64, one
// of two halves for 128) and walks a run of output patches (TH x 32 pixels): per patch the dY tile and the input pixels under it are
// staged once (pixel-major rows, LDS-DMA), and a 32-pixel k-step of patch row r takes its A fragments (dY^T) and, for every tap, its B
// fragments (X at the tap's pixel offset) by transposing LDS reads (ds_read_b64_tr_b16) from those two images.  No dead MFMAs, one
// staging of X for nine taps, one of dY for all of them.
//  * Every fragment address is  (a per-lane register set up once per kernel) + (an immediate that depends on r only): the swizzle of
//    the 32-byte column blocks is a function of the pixel's COLUMN in the patch (never of its row), so a row step is a constant byte
//    offset.  The first version recomputed tap, block and swizzle per read: ~300 VALU instructions per k-step against 18 MFMAs, and
//    ran slower than the tap groups (profiles/r03_experiments.md).
//  * Stride 2: the patch's input columns are stored de-interleaved (even columns, then odd ones), so that the 32 input pixels of a
//    k-step of one tap are consecutive LDS rows as for stride 1, and the same conflict-free swizzle holds.
//  * Bank argument (64 banks x 4 B; a b64 read is served half a wave at a time): lanes 0-31 read 8 pixel rows x 32 bytes, rows
//    o + {0..3, 8..11}.  Row pitch 64 B (Cin = 32): four consecutive rows fill a 256-byte window once, rows + 8 fall on the same
//    slots and take the other 32-byte half -- swizzle bit = bit 3 of the column.  Pitch 128 B (Cin = 64, and dY's 64 channels): rows
//    {o, o+2, o+8, o+10} share a slot of four blocks -- swizzle = (bit 1, bit 3) of the column, distinct for the four (checked
//    exhaustively for every offset, tools/check_pwgrad_banks.py).
// Split-K = the blocks' runs of patches; slab [run][tap][Cout][Cin] fp32 and the fixed-order reduce are wgrad_kernel's.
struct PwgradParams {
    const void* x;
    const void* dy;
    float* slab;
    int B, OH, OW;                 // output (dY) image
    int x_img, x_row, x_y0, x_x0;  // padded input: pixels per image / row, padded coordinates of input pixel (-1, -1) relative to output (0, 0)
    int dy_img, dy_row, dy_pad;
    int tx, ty, tiles, per_block;  // patches per image (x, y), total, per run
    int runs;
    FastDiv div_tx, div_img;
};

template <int V>
struct IntC {
    static constexpr int value = V;
};
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IntC<I>{});
        static_for<I + 1, N>(f);
    }
}

template <int C, int NTOT, int STRIDE, int TH>
__global__ __launch_bounds__(256, (C == 32 && TH * STRIDE <= 4) ? 3 : 2) void pwgrad_kernel(const PwgradParams p) {
    constexpr int TW = 32, NB = 64, NH = NTOT / NB;
    constexpr int XW = STRIDE * TW + 2, XH = STRIDE * (TH - 1) + 3, XPIX = XW * XH;
    constexpr int XROWB = C * 2, YROWB = NB * 2;
    constexpr int XPPI = 1024 / XROWB, YPPI = 1024 / YROWB;
    constexpr int X_INSTR = (XPIX + XPPI - 1) / XPPI, Y_INSTR = TH * TW / YPPI;
    constexpr int X_BYTES = X_INSTR * 1024;
    constexpr int WC = C / 16, WN = 4 / WC;                // waves along the (tap, c) columns / along n
    constexpr int NT = NB / 16 / WN, CT = 9;               // 16 x 16 accumulator tiles per wave: 2 x 9 (Cin = 32) or 4 x 9 (Cin = 64)
    static_assert(9 * (C / 16) == WC * CT && WN * WC == 4, "nine column tiles per wave");
    static_assert((TH - 1) * STRIDE * XW * XROWB + 4 * XROWB < 65536 && (TH - 1) * 32 * YROWB + 4 * YROWB < 65536, "ds_read immediates");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xs = smem;
    char* const ys = smem + X_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wn = w / WC, wc = w % WC;
    const int i16 = lane & 15, g = lane >> 4, q = i16 >> 2, pp = i16 & 3;

    int run, nh;
    if (NH == 1) {
        run = blockIdx.x;
        nh = 0;
    } else {                                               // both halves of a run on one XCD (they read the same input pixels), back to back
        const int t = blockIdx.x >> 3;
        nh = t & 1;
        run = (t >> 1) * 8 + (blockIdx.x & 7);
        if (run >= p.runs) return;
    }

    auto xswz = [](int lcol) { return C == 32 ? (lcol >> 3) & 1 : ((lcol >> 1) & 1) | (((lcol >> 3) & 1) << 1); };
    auto yswz = [](int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); };
    const uint32_t lds_x = (uint32_t)(size_t)LDS_PTR(xs), lds_y = (uint32_t)(size_t)LDS_PTR(ys);

    // fragment addresses of patch row 0 (lo: pixels 8g + q, hi: + 4); row r adds an immediate
    uint32_t ya[NT], xa[CT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int row0 = 8 * g + q;
        ya[nt] = lds_y + row0 * YROWB + (((wn * NT + nt) ^ yswz(row0)) << 5) + pp * 8;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int gct = wc * CT + ct;                      // global column tile = (tap, 16-channel block)
        const int tap = gct / (C / 16), blk = gct % (C / 16);
        const int tyy = tap / 3, txx = tap - 3 * tyy;
        const int off = STRIDE == 1 ? txx : (txx & 1) * (XW / 2) + (txx >> 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int lcol = off + 8 * g + q + 4 * h;
            xa[ct][h] = lds_x + (tyy * XW + lcol) * XROWB + ((blk ^ xswz(lcol)) << 5) + pp * 8;
        }
    }

    f32x4 acc[NT][CT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int t_begin = run * p.per_block;
    const int t_end = t_begin + p.per_block < p.tiles ? t_begin + p.per_block : p.tiles;
    const int Hx = p.x_img / p.x_row;
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int b = (int)fd_div((uint32_t)tile, p.div_img);
        const int trem = tile - b * (p.tx * p.ty);
        const int tyi = (int)fd_div((uint32_t)trem, p.div_tx), txi = trem - tyi * p.tx;
        const int oy0 = tyi * TH, ox0 = txi * TW;
        __syncthreads();                                   // every wave is done reading the previous patch
        {
        // ---- stage X: LDS row py * XW + lcol <- padded input pixel (STRIDE * oy0 + x_y0 + py, STRIDE * ox0 + x_x0 + px) ----
        {
            const bf16_t* src0 = (const bf16_t*)p.x + (int64_t)b * p.x_img * C;
#pragma unroll 1
            for (int i = 0; i < (X_INSTR + 3) / 4; ++i) {
                const int j = i * 4 + w;
                if (j < X_INSTR) {
                    const int row = j * XPPI + lane / (XROWB / 16);
                    const int pix = row < XPIX ? row : XPIX - 1;
                    const int py = STRIDE == 1 ? (pix * 1928) >> 16 : (pix * 993) >> 16;       // / 34, / 66
                    const int lcol = pix - py * XW;
                    const int px = STRIDE == 1 ? lcol : (lcol < XW / 2 ? 2 * lcol : 2 * (lcol - XW / 2) + 1);
                    int iy = STRIDE * oy0 + p.x_y0 + py, ix = STRIDE * ox0 + p.x_x0 + px;
                    iy = iy < Hx ? iy : Hx - 1;
                    ix = ix < p.x_row ? ix : p.x_row - 1;
                    const int chunk = (lane % (XROWB / 16)) ^ (xswz(lcol) << 1);
                    __builtin_amdgcn_global_load_lds(GLB_PTR(src0 + ((int64_t)iy * p.x_row + ix) * C + chunk * 8), LDS_PTR(xs + j * 1024), 16, 0, 0);
                }
            }
        }
        // ---- stage dY: output pixels (oy0 + r, ox0 + k), this block's 64 channels; pixels outside the image read the zero halo ----
        {
            const bf16_t* src0 = (const bf16_t*)p.dy + (int64_t)b * p.dy_img * NTOT + nh * NB;
#pragma unroll 1
            for (int i = 0; i < (Y_INSTR + 3) / 4; ++i) {
                const int j = i * 4 + w;
                if (j < Y_INSTR) {
                    const int row = j * YPPI + lane / (YROWB / 16);                  // r * 32 + k
                    const int oy = oy0 + (row >> 5), ox = ox0 + (row & 31);
                    const bool in = oy < p.OH && ox < p.OW;
                    const int64_t pix = in ? (int64_t)(oy + p.dy_pad) * p.dy_row + ox + p.dy_pad : 0;   // (0, 0) of the padded image: zero
                    const int chunk = (lane % (YROWB / 16)) ^ (yswz(row) << 1);
                    __builtin_amdgcn_global_load_lds(GLB_PTR(src0 + pix * NTOT + chunk * 8), LDS_PTR(ys + j * 1024), 16, 0, 0);
                }
            }
        }
        }
        wait_vm0();
        __syncthreads();
        // ---- TH k-steps of 32 pixels (one patch row each) ----
        static_for<0, TH>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            constexpr int XO = r * STRIDE * XW * XROWB, YO = r * 32 * YROWB;
            // column tiles in groups of CG: the 64-channel form holds 144 accumulators, and nine B fragments on top of them spill
            constexpr int CG = C == 32 ? 9 : 5;
            bf16x8 af[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) af[nt] = cat8(tr_read<YO>(ya[nt]), tr_read<YO + 4 * YROWB>(ya[nt]));
#pragma unroll
            for (int c0 = 0; c0 < CT; c0 += CG) {
                bf16x8 bfr[CG];
#pragma unroll
                for (int ct = c0; ct < (c0 + CG < CT ? c0 + CG : CT); ++ct) bfr[ct - c0] = cat8(tr_read<XO>(xa[ct][0]), tr_read<XO>(xa[ct][1]));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int ct = c0; ct < (c0 + CG < CT ? c0 + CG : CT); ++ct)
                        acc[nt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bfr[ct - c0], acc[nt][ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    }
    // ---- partial gradient -> slab[run][tap][Cout][Cin] ----
    float* out = p.slab + (int64_t)run * 9 * NTOT * C;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int gct = wc * CT + ct;
            const int tap = gct / (C / 16), c = (gct % (C / 16)) * 16 + i16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = nh * NB + (wn * NT + nt) * 16 + g * 4 + j;
                slab_store(out + ((int64_t)tap * NTOT + n) * C + c, acc[nt][ct][j]);
            }
        }
}

// dw[n][c][t] (+)= sum_ks slab[ks][t][n][c].  Block = 64 consecutive (n,c) pairs x 4 split-K groups: every load is
// a coalesced 256-B row of the slab, partial sums are combined in a fixed order (deterministic), and the k*k taps
// of the 64 pairs leave as one contiguous run of the OIHW gradient.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int N, int C,
                                                           int ntaps, int ksplit, int accumulate) {
    __shared__ float part[4][9][64];
    __shared__ float outs[64 * 9];
    const int64_t nc = (int64_t)N * C;
    const int64_t i0 = (int64_t)blockIdx.x * 64;
    const int li = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = i0 + li;
    for (int t = 0; t < ntaps; ++t) {
        float s = 0.f;
        if (i < nc) {
            const float* src = slab + (int64_t)t * nc + i;
            int k = grp;
            for (; k + 12 < ksplit; k += 16) {
                const float v0 = src[(int64_t)k * ntaps * nc], v1 = src[(int64_t)(k + 4) * ntaps * nc];
                const float v2 = src[(int64_t)(k + 8) * ntaps * nc], v3 = src[(int64_t)(k + 12) * ntaps * nc];
                s += (v0 + v1) + (v2 + v3);
            }
            for (; k < ksplit; k += 4) s += src[(int64_t)k * ntaps * nc];
        }
        part[grp][t][li] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * ntaps; e += 256) {
        const int t = e / 64, l = e - t * 64;
        outs[l * ntaps + t] = (part[0][t][l] + part[1][t][l]) + (part[2][t][l] + part[3][t][l]);
    }
    __syncthreads();
    const int64_t valid = (nc - i0 < 64 ? nc - i0 : 64) * ntaps;
    for (int e = threadIdx.x; e < valid; e += 256) {
        float* o = dw + i0 * ntaps + e;
        *o = accumulate ? *o + outs[e] : outs[e];
    }
}

// The same reduction with 16-byte loads, built for bandwidth: a block owns 256 consecutive (n,c) pairs (64 lanes x float4) x all
// taps x KG split-K groups (KG waves: 4, 8 or 16, chosen so that every thread sums a handful of slabs).  Round 2's version gave a
// thread ksplit / 4 slabs of ONE tap at a time with four loads in flight -- 16-32 dependent round trips for a 1x1 layer with 256
// splits: 25 us per launch on average, 1.9 ms per step, latency not bytes.  Here a thread issues the loads of ALL taps of up to
// four slabs together (up to 36 in flight), so a launch is one to four round trips.  The sum order is fixed: slabs of a group in
// index order, groups in index order (deterministic; the grouping differs from round 2's, so results differ in the last bits).
template <int TAPS>
__global__ __launch_bounds__(1024) void wgrad_reduce4_kernel(const float* __restrict__ slab, float* __restrict__ dw, int N, int C,
                                                             int ksplit, int accumulate, int kg) {
    extern __shared__ float rsm[];                       // part[kg][TAPS][256] then outs[256 * TAPS]
    float* part = rsm;
    float* outs = rsm + kg * TAPS * 256;
    const int64_t nc = (int64_t)N * C;
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    const int li = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = i0 + li * 4;
    const int64_t kstride = (int64_t)TAPS * nc;
    f32x4 s[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < nc) {
        const float* src = slab + i;
        constexpr int KU = TAPS == 1 ? 8 : 4;            // slabs in flight per thread
        int k = grp;
        for (; k + (KU - 1) * kg < ksplit; k += KU * kg) {
            f32x4 v[KU][TAPS];
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int t = 0; t < TAPS; ++t) v[u][t] = *(const f32x4*)(src + (int64_t)(k + u * kg) * kstride + (int64_t)t * nc);
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int t = 0; t < TAPS; ++t) s[t] += v[u][t];
        }
        for (; k < ksplit; k += kg) {
            f32x4 v[TAPS];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) v[t] = *(const f32x4*)(src + (int64_t)k * kstride + (int64_t)t * nc);
#pragma unroll
            for (int t = 0; t < TAPS; ++t) s[t] += v[t];
        }
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) *(f32x4*)(part + ((grp * TAPS + t) * 256 + li * 4)) = s[t];
    __syncthreads();
    for (int e = threadIdx.x; e < 256 * TAPS; e += blockDim.x) {
        const int t = e / 256, l = e - t * 256;
        float a = 0.f;
        for (int g = 0; g < kg; ++g) a += part[(g * TAPS + t) * 256 + l];
        outs[l * TAPS + t] = a;
    }
    __syncthreads();
    const int64_t valid = (nc - i0 < 256 ? nc - i0 : 256) * TAPS;
    for (int e = threadIdx.x; e < valid; e += blockDim.x) {
        float* o = dw + i0 * TAPS + e;
        *o = accumulate ? *o + outs[e] : outs[e];
    }
}

// The reduction for the all-taps kernel's slabs: few (n,c) pairs (2048 or 8192) but 256-512 split-K runs, so wgrad_reduce4_kernel's
// grid -- one block per 256 pairs, all taps -- is 8 or 32 blocks that each pull 1-5 MiB through one CU: ~70 us for the 32 -> 64
// layers, a third of the weight gradient's time (tools/thin_pwgrad_abl.sh).  Here a block owns 256 pairs of ONE tap (grid.y = 9)
// and its 16 waves take every 16th run, eight 1-KiB rows in flight each; partial sums are combined in wave order (deterministic).
__global__ __launch_bounds__(1024) void wgrad_reduce_tap_kernel(const float* __restrict__ slab, float* __restrict__ dw, int64_t nc, int ksplit,
                                                                int accumulate) {
    __shared__ f32x4 part[16][64];
    const int t = blockIdx.y;
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    const int li = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = i0 + li * 4;
    const int64_t kstride = 9 * nc;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < nc) {
        const float* src = slab + (int64_t)t * nc + i;
        int k = w;
        for (; k + 7 * 16 < ksplit; k += 8 * 16) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(src + (int64_t)(k + u * 16) * kstride);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < ksplit; k += 16) s += *(const f32x4*)(src + (int64_t)k * kstride);
    }
    part[w][li] = s;
    __syncthreads();
    if (threadIdx.x < 256) {
        const int l = threadIdx.x;
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) a += part[g][l >> 2][l & 3];
        if (i0 + l < nc) {
            float* o = dw + (i0 + l) * 9 + t;
            *o = accumulate ? *o + a : a;
        }
    }
}

struct WgradPlan {
    int tile, ntn, ntc, ntaps, ksplit, mchunk, M, OH, OW, tpt, ngroups;
};

// FVA_WGRAD_THIN=0: layers with Cout <= 64 keep the square 2 x 2 wave arrangement (A/B aid)
inline bool wgrad_thin_enabled() {
    static bool v = [] {
        const char* e = getenv("FVA_WGRAD_THIN");
        return !e || atoi(e) != 0;
    }();
    return v;
}
// FVA_WGRAD8=0 keeps every layer on the 128x128 kernel (A/B aid)
inline bool wgrad8_enabled() {
    static bool v = [] {
        const char* e = getenv("FVA_WGRAD8");
        return !e || atoi(e) != 0;
    }();
    return v;
}
// 3x3 bf16 layers whose tiles fill 256 x 256 (Cin = 128 packs two taps per column tile) and that give every CU a run of
// at least 24 k-steps of 64 pixels (below that the 512-thread block's prologue / slab epilogue dominate: 7x33^2 pixels of a
// 256->512 layer measured 52 vs 44 us).  Measured at B = 32 (tools/check_wgrad8.py): 256->512 @40^2 177 -> 148 us,
// 512->1024 @20^2 209 -> 169 us, 128->256 @80^2 176 -> 169 us.
inline bool use_wgrad8(const fva_conv_desc* d) {
    if (!wgrad8_enabled() || d->dtype != FVA_BF16 || d->ksize != 3 || d->Cout % 256 || !(d->Cin == 128 || d->Cin % 256 == 0)) return false;
    const int64_t OH = (d->H - 1) / d->stride + 1, OW = (d->W - 1) / d->stride + 1, M = d->B * OH * OW;
    const int64_t units = (int64_t)(d->Cout / 256) * cdiv(d->Cin, 256) * (d->Cin == 128 ? 5 : 9);
    return M * units >= 24ll * 64 * 256;
}

// beside: the weight gradients run on the library's low-priority side stream and share the chip with the rest of the backward pass:
// planned for HALF the resident block slots, their fewer, longer blocks leave CUs to the launch stream and write half the slab bytes.
// Measured in the whole step (round 3, same box): 29.94 ms with 128 / 256 slots against 30.2 with 256 / 512; alone on a stream the same
// plan is slower (33.8 against 31.0 ms single-stream).  The split factor fixes the fp32 summation order of dW, so WHICH plan is used is
// a process-wide setting (fva_conv_wgrad_plan; ops sets it together with the side stream), never a property of the stream a launch
// happens to be given: a layer that falls back to the launch stream, or a step captured on one stream, sums in the same order.
int g_wgrad_plan_beside = 0;
int plan_wgrad(const fva_conv_desc* d, WgradPlan& pl, bool beside = false) {
    pl.tile = d->dtype == FVA_BF16 ? (use_wgrad8(d) ? 256 : 128) : 64;
    pl.OH = (d->H - 1) / d->stride + 1;
    pl.OW = (d->W - 1) / d->stride + 1;
    pl.M = d->B * pl.OH * pl.OW;
    pl.ntn = cdiv(d->Cout, pl.tile);
    pl.ntc = cdiv(d->Cin, pl.tile);
    pl.ntaps = d->ksize * d->ksize;
    pl.tpt = (d->Cin < pl.tile && pl.tile % d->Cin == 0 && pl.ntaps > 1) ? pl.tile / d->Cin : 1;
    pl.ngroups = cdiv(pl.ntaps, pl.tpt);
    const int tiles = pl.ntn * pl.ntc * pl.ngroups;
    // Split-K factor: 256 CUs x 2 resident blocks (64 KiB LDS each) = 512 slots.  Model: the launch takes
    // ceil(blocks / 512) rounds of blocks that each do 1/ks of a tile's pixels (t_full: one block doing a whole tile at
    // ~1.6 TFLOP/s per resident block), plus ks partial tiles to write and re-read at ~3 TB/s.  E.g. 18 tiles of a
    // 128->256 3x3 layer -> ks = 28 (504 blocks, one full round); a 1-tile 1x1 layer -> ks = 512.
    const int max_by_m = cdiv(pl.M, 256) > 0 ? cdiv(pl.M, 256) : 1;
    const int64_t per = (int64_t)pl.ntaps * d->Cout * d->Cin * 4;
    int64_t max_by_ws = (512ll << 20) / per;
    int kmax = max_by_m < 1024 ? max_by_m : 1024;
    if (kmax > max_by_ws) kmax = (int)max_by_ws;
    if (kmax < 1) kmax = 1;
    // the 256x256 8-phase kernel runs one block per CU (256 slots) at ~4.3 TFLOP/s per block
    // FVA_WGRAD_SLOTS8 / FVA_WGRAD_SLOTS (experiment): plan for fewer resident blocks than the chip holds -- fewer, longer blocks and
    // less slab traffic, leaving CUs to the launch stream when the weight gradients run beside it
    static const int slots8 = [] { const char* e = getenv("FVA_WGRAD_SLOTS8"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
    static const int slots4 = [] { const char* e = getenv("FVA_WGRAD_SLOTS"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
    const int slots = pl.tile == 256 ? (slots8 ? slots8 : beside ? 128 : 256) : (slots4 ? slots4 : beside ? 256 : 512);
    const double t_full = 2.0 * pl.M * pl.tile * pl.tile / (pl.tile == 256 ? 4.3e12 : 1.6e12);
    const double t_slab = (double)per * 2.0 / 3.0e12 + 0.05e-6;
    int ks = 1;
    double best = 1e30;
    for (int k = 1; k <= kmax; ++k) {
        const int rounds = cdiv((int64_t)tiles * k, slots);
        const double cost = rounds * t_full / k + k * t_slab;
        if (cost < best * (1.0 - 1e-9)) { best = cost; ks = k; }
    }
    pl.mchunk = cdiv(cdiv(pl.M, ks), 64) * 64;
    pl.ksplit = cdiv(pl.M, pl.mchunk);
    return FVA_OK;
}

// the all-taps kernel: thin 3x3 bf16 layers on maps at least one patch wide.  FVA_PWGRAD=0 switches it off.
inline bool use_pwgrad(const fva_conv_desc* d) {
    static const bool on = [] { const char* e = getenv("FVA_PWGRAD"); return !e || atoi(e) != 0; }();
    if (!on || d->dtype != FVA_BF16 || d->ksize != 3 || !((d->Cin == 32 && d->Cout == 64) || (d->Cin == 64 && d->Cout == 128))) return false;
    const int OW = (d->W - 1) / d->stride + 1, OH = (d->H - 1) / d->stride + 1;
    return OW >= 32 && OH >= 8 && d->in_pad >= 1 && d->dy_pad >= 1;
}
struct PwgradPlan {
    int th, tx, ty, tiles, runs, per_block, grid;
};
// patch rows: 8 (stride 1), 4 (stride 2, Cin = 32), 2 (stride 2, Cin = 64: the 4-row patch is 92 KiB of LDS, one block per CU -- 272 against
// 211 us).  FVA_PWGRAD_TH=<stride-1 rows>,<stride-2 rows> forces the other instantiations (8 or 4, 4 or 2; tools/thin_pwgrad.sh).
inline int pwgrad_th(const fva_conv_desc* d) {
    static int th1 = 0, th2 = 0;
    static const bool init = [] {
        const char* e = getenv("FVA_PWGRAD_TH");
        if (e) {
            int a = 0, b = 0;
            const int n = sscanf(e, "%d,%d", &a, &b);
            if (n >= 1 && (a == 8 || a == 4)) th1 = a;
            if (n >= 2 && (b == 4 || b == 2)) th2 = b;
        }
        return true;
    }();
    (void)init;
    if (d->stride == 1) return th1 ? th1 : 8;
    return th2 ? th2 : (d->Cin == 64 ? 2 : 4);
}
inline void plan_pwgrad(const fva_conv_desc* d, PwgradPlan& pl, bool beside) {
    const int OH = (d->H - 1) / d->stride + 1, OW = (d->W - 1) / d->stride + 1;
    pl.th = pwgrad_th(d);
    pl.tx = cdiv(OW, 32);
    pl.ty = cdiv(OH, pl.th);
    pl.tiles = d->B * pl.tx * pl.ty;
    const int nh = d->Cout / 64;
    static const int per_cu = [] { const char* e = getenv("FVA_PWGRAD_SLOTS"); return e ? atoi(e) : 2; }();
    int slots = 256 * per_cu / nh;                  // resident runs: two 4-wave blocks per CU, a run of a 128-channel layer is two blocks
    if (beside) slots /= 2;                         // beside the launch stream: plan_wgrad
    const int runs = pl.tiles < slots ? pl.tiles : slots;
    pl.per_block = cdiv(pl.tiles, runs);
    pl.runs = cdiv(pl.tiles, pl.per_block);
    pl.grid = nh == 1 ? pl.runs : cdiv(pl.runs, 8) * 16;
}

template <int C, int N, int STRIDE, int TH>
int launch_pwgrad(const PwgradParams& p, int grid, hipStream_t s) {
    constexpr int XPIX = (STRIDE * 32 + 2) * (STRIDE * (TH - 1) + 3);
    constexpr int XPPI = 1024 / (C * 2);
    constexpr int smem = ((XPIX + XPPI - 1) / XPPI) * 1024 + (TH * 32 / 8) * 1024;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)pwgrad_kernel<C, N, STRIDE, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    hipLaunchKernelGGL((pwgrad_kernel<C, N, STRIDE, TH>), dim3(grid), dim3(256), smem, s, p);
    FVA_LAUNCH_CHECK("pwgrad_kernel");
    fva_note_kernel("pwgrad");
    return FVA_OK;
}
template <int C, int N>
int launch_pwgrad_any(const PwgradParams& p, int stride, int th, int grid, hipStream_t s) {
    if (stride == 1) return th == 8 ? launch_pwgrad<C, N, 1, 8>(p, grid, s) : launch_pwgrad<C, N, 1, 4>(p, grid, s);
    return th == 4 ? launch_pwgrad<C, N, 2, 4>(p, grid, s) : launch_pwgrad<C, N, 2, 2>(p, grid, s);
}

}  // namespace

extern "C" {

int64_t fva_conv_wgrad_workspace(const fva_conv_desc* d) {
    if (!d) return 0;
    WgradPlan pl, pb;
    plan_wgrad(d, pl, false);
    plan_wgrad(d, pb, true);          // whichever stream the launch will be given
    int ks = pl.ksplit > pb.ksplit ? pl.ksplit : pb.ksplit;
    if (use_pwgrad(d)) {
        PwgradPlan pp;
        plan_pwgrad(d, pp, false);
        ks = ks > pp.runs ? ks : pp.runs;
    }
    return (int64_t)ks * pl.ntaps * d->Cout * d->Cin * 4;
}

int fva_conv_wgrad_plan(int beside) {
    const int prev = g_wgrad_plan_beside;
    if (beside == 0 || beside == 1) g_wgrad_plan_beside = beside;
    return prev;
}

int fva_conv_wgrad(const fva_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate, void* workspace,
                   int64_t workspace_bytes, void* stream) {
    if (!d || !x || !dy || !dw || !workspace) return fva_fail(FVA_ERR_ARG, "fva_conv_wgrad: null pointer");
    if (d->dtype != FVA_F32 && d->dtype != FVA_BF16) return fva_fail(FVA_ERR_ARG, "fva_conv_wgrad: bad dtype");
    if (!((d->ksize == 1 && d->stride == 1) || (d->ksize == 3 && (d->stride == 1 || d->stride == 2))))
        return fva_fail(FVA_ERR_ARG, "fva_conv_wgrad: unsupported ksize/stride");
    const int epc = d->dtype == FVA_BF16 ? 8 : 4;
    if (d->Cin % epc || d->Cout % epc) return fva_fail(FVA_ERR_ARG, "fva_conv_wgrad: channels must be multiples of %d", epc);
    if (d->in_pad < d->ksize / 2) return fva_fail(FVA_ERR_ARG, "fva_conv_wgrad: in_pad too small");
    if (d->dy_pad < 1) return fva_fail(FVA_ERR_ARG, "fva_conv_wgrad: dy buffer needs a zero border (dy_pad >= 1)");
    const int64_t esz = d->dtype == FVA_BF16 ? 2 : 4;
    if ((int64_t)d->B * (d->H + 2 * d->in_pad) * (d->W + 2 * d->in_pad) * d->Cin * esz >= (1ll << 32) ||
        (int64_t)d->B * (d->H + 2) * (d->W + 2) * d->Cout * esz >= (1ll << 32))
        return fva_fail(FVA_ERR_ARG, "fva_conv_wgrad: operand larger than 4 GiB (32-bit byte offsets)");
    WgradPlan pl;
    const bool beside = g_wgrad_plan_beside != 0;
    plan_wgrad(d, pl, beside);
    FvaProfileSpan span(2 | (d->ksize << 8), 2.0 * pl.M * (double)d->Cout * d->Cin * d->ksize * d->ksize, (hipStream_t)stream);
    if (use_pwgrad(d)) {
        PwgradPlan pp;
        plan_pwgrad(d, pp, beside);
        const int64_t need = (int64_t)pp.runs * 9 * d->Cout * d->Cin * 4;
        if (workspace_bytes < need) return fva_fail(FVA_ERR_WORKSPACE, "fva_conv_wgrad: workspace %lld < %lld", (long long)workspace_bytes, (long long)need);
        PwgradParams q = PwgradParams();
        q.x = x; q.dy = dy; q.slab = (float*)workspace;
        q.B = d->B; q.OH = pl.OH; q.OW = pl.OW;
        q.x_row = d->W + 2 * d->in_pad;
        q.x_img = (d->H + 2 * d->in_pad) * q.x_row;
        q.x_y0 = q.x_x0 = d->in_pad - 1;
        q.dy_row = pl.OW + 2 * d->dy_pad;
        q.dy_img = (pl.OH + 2 * d->dy_pad) * q.dy_row;
        q.dy_pad = d->dy_pad;
        q.tx = pp.tx; q.ty = pp.ty; q.tiles = pp.tiles; q.per_block = pp.per_block; q.runs = pp.runs;
        q.div_tx = make_fastdiv(pp.tx);
        q.div_img = make_fastdiv(pp.tx * pp.ty);
        hipStream_t s = (hipStream_t)stream;
        const int rc = d->Cin == 32 ? launch_pwgrad_any<32, 64>(q, d->stride, pp.th, pp.grid, s) : launch_pwgrad_any<64, 128>(q, d->stride, pp.th, pp.grid, s);
        if (rc) return rc;
        const int64_t nc = (int64_t)d->Cout * d->Cin;
        hipLaunchKernelGGL(wgrad_reduce_tap_kernel, dim3((int)((nc + 255) / 256), 9), dim3(1024), 0, s, (const float*)workspace, dw, nc, pp.runs, accumulate);
        FVA_LAUNCH_CHECK("wgrad_reduce_tap_kernel");
        return FVA_OK;
    }
    const int64_t need = (int64_t)pl.ksplit * pl.ntaps * d->Cout * d->Cin * 4;
    if (workspace_bytes < need) return fva_fail(FVA_ERR_WORKSPACE, "fva_conv_wgrad: workspace %lld < %lld", (long long)workspace_bytes, (long long)need);
    WgradParams p = WgradParams();
    p.x = x;
    p.dy = dy;
    p.slab = (float*)workspace;
    p.M = pl.M;
    p.N = d->Cout;
    p.C = d->Cin;
    p.OW = pl.OW;
    p.OHW = pl.OH * pl.OW;
    p.div_ow = make_fastdiv(p.OW);
    p.div_ohw = make_fastdiv(p.OHW);
    p.x_row = d->W + 2 * d->in_pad;
    p.x_img = (d->H + 2 * d->in_pad) * p.x_row;
    p.sy = p.sx = d->stride;
    p.x_y0 = p.x_x0 = d->in_pad - d->ksize / 2;
    p.dy_row = pl.OW + 2 * d->dy_pad;
    p.dy_img = (pl.OH + 2 * d->dy_pad) * p.dy_row;
    p.dy_pad = d->dy_pad;
    p.dy_pitch = d->Cout;
    p.dy_zero_off = 0;  // top-left border pixel of image 0 is always zero
    p.ntaps = pl.ntaps;
    for (int kh = 0; kh < d->ksize; ++kh)
        for (int kw = 0; kw < d->ksize; ++kw) p.tap_pix[kh * d->ksize + kw] = kh * p.x_row + kw;
    p.ntn = pl.ntn;
    p.ntc = pl.ntc;
    p.ksplit = pl.ksplit;
    p.mchunk = pl.mchunk;
    p.tpt = pl.tpt;
    p.ngroups = pl.ngroups;
    const int grid = pl.ksplit * pl.ngroups * pl.ntn * pl.ntc;
    const int smem = 2 * 2 * 64 * 256;
    hipStream_t s = (hipStream_t)stream;
    if (pl.tile == 256) {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)wgrad8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 64 * 256);
            attr_done = true;
        }
        p.stamps = fva_debug_stamps_ptr();
        p.stamp_rows = fva_debug_stamps_rows();
        hipLaunchKernelGGL(wgrad8_kernel, dim3(grid), dim3(512), 2 * 4 * 64 * 256, s, p);
        fva_note_kernel("wgrad8");
    } else if (d->dtype == FVA_BF16 && d->Cout <= 64 && wgrad_thin_enabled()) {
        hipLaunchKernelGGL((wgrad_kernel<bf16_t, true>), dim3(grid), dim3(256), smem, s, p);
        fva_note_kernel("wgrad128thin");
    } else if (d->dtype == FVA_BF16) {
        hipLaunchKernelGGL(wgrad_kernel<bf16_t>, dim3(grid), dim3(256), smem, s, p);
        fva_note_kernel("wgrad128");
    } else {
        hipLaunchKernelGGL(wgrad_kernel<float>, dim3(grid), dim3(256), smem, s, p);
        fva_note_kernel("wgrad64f32");
    }
    FVA_LAUNCH_CHECK("wgrad_kernel");
    const int64_t nc = (int64_t)d->Cout * d->Cin;
    static const bool reduce4 = [] { const char* e = getenv("FVA_WGRAD_REDUCE4"); return !e || atoi(e) != 0; }();   // =0: A/B aid
    if (reduce4 && nc % 4 == 0 && ((uintptr_t)workspace & 15) == 0 && (pl.ntaps == 1 || pl.ntaps == 9)) {
        // split-K groups per block: enough that a thread sums at most ~8 (1x1) / ~4 (3x3) slabs, and that small layers still put
        // a few hundred waves on the chip
        int kg = 4;                                     // 3x3: four groups (nine taps x four slabs = 36 loads in flight per thread; 46 KiB of LDS)
        if (pl.ntaps == 1)
            while (kg < 16 && kg * 8 < pl.ksplit) kg *= 2;
        const int smem_r = (kg + 1) * pl.ntaps * 256 * 4;
        const dim3 grid((int)((nc + 255) / 256)), block(64 * kg);
        if (pl.ntaps == 1) {
            hipLaunchKernelGGL(wgrad_reduce4_kernel<1>, grid, block, smem_r, s, (const float*)workspace, dw, d->Cout, d->Cin, pl.ksplit, accumulate, kg);
        } else {
            static bool attr9 = false;
            if (!attr9) {
                (void)hipFuncSetAttribute((const void*)wgrad_reduce4_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 9 * 256 * 4);
                attr9 = true;
            }
            hipLaunchKernelGGL(wgrad_reduce4_kernel<9>, grid, block, smem_r, s, (const float*)workspace, dw, d->Cout, d->Cin, pl.ksplit, accumulate, kg);
        }
    } else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)((nc + 63) / 64)), dim3(256), 0, s, (const float*)workspace, dw, d->Cout,
                           d->Cin, pl.ntaps, pl.ksplit, accumulate);
    FVA_LAUNCH_CHECK("wgrad_reduce_kernel");
    return FVA_OK;
}

/* Stem weight gradient on MFMA: conv0 (3 x 3, Cin <= 3) seen as a 3 x 1 convolution over "pixels" of 16 channels -- the four
 * horizontally adjacent pixels x-1 .. x+2 of the bf16 NHWC4 image fva_stem_fwd packs (8 bytes per pixel, overlapping windows:
 * x_pix_bytes = 8) -- so that the generic kernel applies: N = 32, C = 16, three vertical taps packed into one column tile.
 * dw_raw[32][16][3] = [co][kw*4 + ci][kh]; the caller keeps kw < 3, ci < Cin. */
int fva_stem_wgrad_mfma(const void* img4, const void* dy_halo, float* dw_raw, void* workspace, int64_t workspace_bytes, int B, int H,
                        int W, void* stream) {
    if (!img4 || !dy_halo || !dw_raw || !workspace) return fva_fail(FVA_ERR_ARG, "fva_stem_wgrad_mfma: null pointer");
    const int64_t M = (int64_t)B * H * W;
    if (M >= (1ll << 31) || (int64_t)B * (H + 2) * (W + 2) * 64 >= (1ll << 32)) return fva_fail(FVA_ERR_ARG, "fva_stem_wgrad_mfma: too large");
    const int ksplit_max = 512;
    int mchunk = cdiv(cdiv(M, ksplit_max), 64) * 64;
    const int ksplit = cdiv(M, mchunk);
    const int64_t need = (int64_t)ksplit * 3 * 32 * 16 * 4;
    if (workspace_bytes < need) return fva_fail(FVA_ERR_WORKSPACE, "fva_stem_wgrad_mfma: workspace %lld < %lld", (long long)workspace_bytes, (long long)need);
    WgradParams p = WgradParams();
    p.x = img4;
    p.dy = dy_halo;
    p.slab = (float*)workspace;
    p.M = (int)M;
    p.N = 32;
    p.C = 16;
    p.x_pix_bytes = 8;
    p.OW = W;
    p.OHW = H * W;
    p.div_ow = make_fastdiv(p.OW);
    p.div_ohw = make_fastdiv(p.OHW);
    p.x_row = W + 2;
    p.x_img = (H + 2) * p.x_row;
    p.sy = p.sx = 1;
    p.x_y0 = p.x_x0 = 0;
    p.dy_row = W + 2;
    p.dy_img = (H + 2) * p.dy_row;
    p.dy_pad = 1;
    p.dy_pitch = 32;
    p.dy_zero_off = 0;
    p.ntaps = 3;
    for (int kh = 0; kh < 3; ++kh) p.tap_pix[kh] = kh * p.x_row;
    p.ntn = p.ntc = 1;
    p.ksplit = ksplit;
    p.mchunk = mchunk;
    p.tpt = 8;
    p.ngroups = 1;
    FvaProfileSpan span(2 | (3 << 8), 2.0 * M * 32.0 * 27.0, (hipStream_t)stream);
    hipStream_t s = (hipStream_t)stream;
    if (wgrad_thin_enabled())
        hipLaunchKernelGGL((wgrad_kernel<bf16_t, true>), dim3(ksplit), dim3(256), 2 * 2 * 64 * 256, s, p);
    else
        hipLaunchKernelGGL(wgrad_kernel<bf16_t>, dim3(ksplit), dim3(256), 2 * 2 * 64 * 256, s, p);
    FVA_LAUNCH_CHECK("wgrad_kernel");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((32 * 16 + 63) / 64), dim3(256), 0, s, (const float*)workspace, dw_raw, 32, 16, 3, ksplit, 0);
    FVA_LAUNCH_CHECK("wgrad_reduce_kernel");
    return FVA_OK;
}

int64_t fva_stem_wgrad_mfma_workspace(void) { return 512ll * 3 * 32 * 16 * 4; }

}  // extern "C"
