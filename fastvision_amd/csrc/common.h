// Shared device/host helpers for libfastvision_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/fastvision_amd.h"

typedef __bf16 bf16_t;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- error reporting -------------------------------------------------------------------------------
extern thread_local char fva_err_buf[512];
static inline int fva_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(fva_err_buf, sizeof(fva_err_buf), fmt, ap);
    va_end(ap);
    return code;
}
#define FVA_LAUNCH_CHECK(name)                                                                     \
    do {                                                                                           \
        hipError_t e_ = hipGetLastError();                                                         \
        if (e_ != hipSuccess) return fva_fail(FVA_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ---- division by a runtime constant ------------------------------------------------------------------
struct FastDiv {
    uint32_t d, mul, sh;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t sh = 0;
    while ((1ull << sh) < d) ++sh;
    f.sh = sh;
    f.mul = (uint32_t)(((1ull << 32) * ((1ull << sh) - d)) / d + 1);
    return f;
}
// valid for n < 2^31
__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) { return (__umulhi(n, f.mul) + n) >> f.sh; }

// ---- element conversion ------------------------------------------------------------------------------
__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f(float v);
template <>
__device__ __forceinline__ float from_f<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return (bf16_t)v; }

// 16-byte vector of T: 4 floats or 8 bf16
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
    static constexpr int N = 4;
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <>
struct Vec16<bf16_t> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16_t)x; }
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// sigmoid with the hardware reciprocal (1 ulp) instead of an IEEE divide: the divide sequence (~10 VALU ops per element)
// competes with the memory pipeline in the HBM-bound BatchNorm / SiLU kernels and in fused conv epilogues
__device__ __forceinline__ float sigm_fast(float u) { return __builtin_amdgcn_rcpf(1.f + __expf(-u)); }
__device__ __forceinline__ float silu_f(float u) { return u * sigm_fast(u); }
// forward apply, per element: z = SiLU(y * scale + shift) (bn_act.hip's apply pass and the fused A-operand path of conv_igemm.hip share it,
// so that z has the same bits wherever it is produced)
__device__ __forceinline__ float bn_silu_fwd_elem(float y, float sc, float sh) { return silu_f(__builtin_fmaf(y, sc, sh)); }
__device__ __forceinline__ float silu_grad(float u) {
    const float s = sigm_fast(u);
    return s * (1.f + u * (1.f - s));
}

// ---- BatchNorm statistics summed by integer atomics (round 4) ------------------------------------------------------------------------
// A convolution epilogue leaves, per tile and channel, two fp32 partial sums.  Until round 4 they went to a table that a separate launch
// folded (bn_finalize / bn_prereduce + bn_bwd_finalize: 197 latency-bound launches per YOLOv3 step on the critical chain, 2.1 ms of it).
// Now every tile ADDS its partials into one accumulator per channel and the pass that consumes the statistics (the apply kernel, or the
// fused 1x1 convolution) finalises them in its own prologue.  Adding in floating point would make the result depend on the order in which
// the tiles arrive; the accumulator is therefore a two-word FIXED-POINT number -- hi in units of 2^-8, lo in units of 2^-56, both int64
// -- into which a partial is split exactly: integer addition is associative, so the sum is the same bits whatever the order
// (deterministic run to run), and it carries every bit of a partial down to 2^-56 (an fp32 partial of magnitude >= 2^-32 exactly).
// A non-finite partial is counted in a word of its own instead (the finalised sums of that channel then read NaN).
struct FxSplit {
    long long hi, lo;
    bool bad;
};
__device__ __forceinline__ FxSplit fx_split(float p) {
    FxSplit r;
    r.bad = !(__builtin_fabsf(p) < 1.0e15f);        // NaN, Inf, or beyond any plausible sum
    const double d = r.bad ? 0.0 : (double)p;
    r.hi = __double2ll_rn(d * 256.0);
    const double rest = d - (double)r.hi * (1.0 / 256.0);         // exact: |rest| <= 2^-9 and no finer than ulp(p)
    r.lo = __double2ll_rn(rest * 72057594037927936.0);            // 2^56
    return r;
}
// Accumulator of one BatchNorm layer and direction: int64 acc[replicas][FX_WORDS][C], words = {hi of sum 1, hi of sum 2, lo of sum 1,
// lo of sum 2, count of non-finite partials} -- a wave that adds 32 channels of both sums touches two contiguous 256-byte runs per word,
// the shape the memory-side atomic units take at full rate.  The atomics on ONE address are served one after the other (~10 ns each,
// measured: a 12,800-tile launch ran 140 us longer than its table form), so layers with many tiles spread their adds over `replicas`
// copies (a power of two; block b adds to copy b mod replicas) and the consumer adds the copies up -- still integers, still exact.
// A layer has one accumulator per direction (forward: sums of y, y^2; backward: sums of dU, dU * xhat); the consumer of one direction
// returns the OTHER one to zero (its own is still being read by its other blocks), so both are zero again when their producers next
// run -- graph replays included.
constexpr int FX_WORDS = FVA_BN_ACC_WORDS;
inline bool fva_replicas_ok(int r) { return r >= 1 && r <= FVA_BN_ACC_MAX_REPLICAS && !(r & (r - 1)); }
__device__ __forceinline__ long long* fx_replica(long long* acc, int C, int replicas) {
    return acc + (size_t)(blockIdx.x & (unsigned)(replicas - 1)) * FX_WORDS * C;
}
__device__ __forceinline__ void fx_atomic_add(long long* acc, int C, int which, int c, float p) {
    const FxSplit s = fx_split(p);
    if (s.bad) {                                       // poisons the channel: its finalised sums read NaN, as a floating-point sum would
        atomicAdd((unsigned long long*)acc + (size_t)4 * C + c, 1ull);
        return;
    }
    atomicAdd((unsigned long long*)acc + (size_t)which * C + c, (unsigned long long)s.hi);
    atomicAdd((unsigned long long*)acc + (size_t)(2 + which) * C + c, (unsigned long long)s.lo);
}
__device__ __forceinline__ void fx_load2(const long long* acc, int C, int replicas, int c, double& s1, double& s2) {
    long long h1 = 0, h2 = 0, l1 = 0, l2 = 0, bad = 0;
    for (int r = 0; r < replicas; ++r) {
        const long long* a = acc + (size_t)r * FX_WORDS * C;
        h1 += a[c]; h2 += a[(size_t)C + c]; l1 += a[(size_t)2 * C + c]; l2 += a[(size_t)3 * C + c]; bad += a[(size_t)4 * C + c];
    }
    s1 = (double)h1 * (1.0 / 256.0) + (double)l1 * (1.0 / 72057594037927936.0);
    s2 = (double)h2 * (1.0 / 256.0) + (double)l2 * (1.0 / 72057594037927936.0);
    if (bad != 0) s1 = s2 = __builtin_nan("");
}
// The consumer's view of one channel: the four words and the poison count, summed over the replicas.
struct FxSums {
    long long h1, h2, l1, l2, bad;
};
__device__ __forceinline__ void fx_value2(const FxSums& w, double& s1, double& s2) {
    s1 = (double)w.h1 * (1.0 / 256.0) + (double)w.l1 * (1.0 / 72057594037927936.0);
    s2 = (double)w.h2 * (1.0 / 256.0) + (double)w.l2 * (1.0 / 72057594037927936.0);
    if (w.bad != 0) s1 = s2 = __builtin_nan("");
}
// All channels of a replicated accumulator added up by the whole block: lds = long long [FX_WORDS][C]; every thread fetches its share of
// the replicas * FX_WORDS * C words (independent loads, all in flight together -- a thread walking the replicas of its own channel pays
// one memory round trip per replica) and adds it with an LDS integer atomic.  Ends with a barrier.
__device__ __forceinline__ void fx_gather_lds(const long long* acc, int C, int replicas, long long* lds, int tid, int nthreads) {
    const int n = FX_WORDS * C;
    for (int i = tid; i < n; i += nthreads) lds[i] = 0;
    __syncthreads();
    const int total = n * replicas;
    for (int j0 = tid; j0 < total; j0 += nthreads * 8) {      // eight loads in flight per thread, then eight adds: no branch between a load and the next
        long long v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * nthreads;
            v[u] = j < total ? acc[j] : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * nthreads;
            if (j < total) atomicAdd((unsigned long long*)(lds + (j - (j / n) * n)), (unsigned long long)v[u]);
        }
    }
    __syncthreads();
}
__device__ __forceinline__ FxSums fx_from_lds(const long long* lds, int C, int c) {
    FxSums w;
    w.h1 = lds[c]; w.h2 = lds[C + c]; w.l1 = lds[2 * C + c]; w.l2 = lds[3 * C + c]; w.bad = lds[4 * C + c];
    return w;
}
__device__ __forceinline__ void fx_zero(long long* acc, int C, int replicas, int tid, int nthreads) {
    if (acc != nullptr)
        for (int i = tid; i < FX_WORDS * C * replicas; i += nthreads) acc[i] = 0;
}
// forward: batch statistics -> the coefficients of the apply pass.  Written to be cheap per channel (consumer kernels run it in their
// prologue, every block): the pixel count comes as its reciprocal, and 1 / sqrt is v_rsq_f64 plus one Newton step in double (then rounded
// to fp32 once, like the division it replaces) instead of an IEEE double division and square root (~100 instructions each on this chip).
struct BnN {
    double inv, unbias;      // 1 / count;  count / (count - 1): the factor of the unbiased variance for the running estimate
};
__host__ __device__ inline BnN bn_n(double count) {
    BnN n;
    n.inv = 1.0 / count;
    n.unbias = count > 1.0 ? count / (count - 1.0) : 1.0;
    return n;
}
struct BnFwdCoef {
    float mean, rstd, scale, shift;
    double var;
};
__device__ __forceinline__ BnFwdCoef bn_fwd_coef(double s1, double s2, const BnN& n, float gamma, float beta, float eps) {
    BnFwdCoef k;
    const double mean = s1 * n.inv;
    double var = __builtin_fma(s2, n.inv, -mean * mean);
    if (var < 0.0) var = 0.0;
    k.var = var;
    const double x = var + (double)eps;
    double r = __builtin_amdgcn_rsq(x);                      // v_rsq_f64: ~2^-26 relative
    r = __builtin_fma(0.5 * r, __builtin_fma(-x * r, r, 1.0), r);   // one Newton step: to the last bits of the double
    k.rstd = (float)r;                                       // (fp32 training holds 1e-3 on the reference's loss curve only with rstd rounded once)
    k.mean = (float)mean;
    k.scale = gamma * k.rstd;
    k.shift = __builtin_fmaf(-k.mean, k.scale, beta);
    return k;
}
// unbiased variance for the running estimate (nn.BatchNorm2d, training mode)
__device__ __forceinline__ void bn_running_update(float* running_mean, float* running_var, int c, const BnFwdCoef& k, const BnN& n, float momentum) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * k.mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(k.var * n.unbias);
}
// backward: sums of dU and dU * xhat -> dgamma, dbeta and the three coefficients of the second pass (the arithmetic of bn_bwd_finalize_kernel)
struct BnBwdCoef {
    float dbeta, dgamma, a, cb, cc;
};
__device__ __forceinline__ BnBwdCoef bn_bwd_coef(double s1, double s2, const BnN& n, float gamma, float rstd) {
    BnBwdCoef k;
    k.dbeta = (float)s1;
    k.dgamma = (float)s2;
    k.a = gamma * rstd;
    k.cb = (float)(-(double)k.a * s2 * n.inv);
    k.cc = (float)(-(double)k.a * s1 * n.inv);
    return k;
}

// ---- second pass of the BatchNorm + SiLU backward, per element (bn_act.hip's apply kernels and the apply "rider" of the weight-gradient
// kernels share these, so that a pass produces the same bits wherever it runs):  dY = a * dU + k1 * y + k2,  dU = dz * SiLU'(y * a + shift),
// a = gamma * rstd (= the forward scale), k1 = coefB * rstd, k2 = coefC - k1 * mean  (coef = fva_bn_bwd_finalize's [3][C] table).
struct BnBwdK {
    float a, sh, k1, k2;
};
__device__ __forceinline__ BnBwdK bn_bwd_pack_coef(float a, float shift, float mean, float rstd, float coef_b, float coef_c) {
    BnBwdK k;
    k.a = a;
    k.sh = shift;
    k.k1 = coef_b * rstd;
    k.k2 = __builtin_fmaf(-k.k1, mean, coef_c);
    return k;
}
// sc: the forward scale (scale[c]); a: coef[c].  Both are gamma * rstd, computed by fva_bn_finalize and fva_bn_bwd_finalize from the same
// two floats, hence the same bits; the riders keep one copy (BnBwdK::a) and pass it for both.
__device__ __forceinline__ float bn_bwd_apply_elem(float dz, float y, float sc, float sh, float a, float k1, float k2) {
    const float du = dz * silu_grad(__builtin_fmaf(y, sc, sh));
    return __builtin_fmaf(a, du, __builtin_fmaf(k1, y, k2));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15): four rotations, every lane of the row ends with the total
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0x128>(v);   // row_ror:8
    v += dpp_mov<0x124>(v);   // row_ror:4
    v += dpp_mov<0x122>(v);   // row_ror:2
    v += dpp_mov<0x121>(v);   // row_ror:1
    return v;
}

// two floats -> packed bf16 pair (round to nearest even): one v_cvt_pk_bf16_f32.  Through the compiler's own vector conversion, NOT
// inline asm: an asm statement that reads an MFMA result is invisible to the hazard recognizer (no wait states are inserted between
// the MFMA and the read) -- the patch kernels' plain epilogue, where only a barrier separates the two, produced NaNs that way.
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// zero border of a halo NHWC buffer [B][H+2p][W+2p][C] given in 16-byte pieces per pixel (errors.hip)
int fva_zero_halo_border(void* z, int B, int H, int W, int chunks_per_pixel, int pad, hipStream_t stream);

// ---- in-library timing of the MFMA convolution entry points (fva_profile_start / fva_profile_stop) ---------------------
// A span records one HIP event on the launch stream when it is made and one when it goes out of scope; disabled = no-op.
struct FvaProfileSpan {
    int slot;
    hipStream_t stream;
    FvaProfileSpan(int cls, double flop, hipStream_t s);
    ~FvaProfileSpan();
};

// name of the convolution kernel the calling thread launched last (fva_conv_last_kernel: the parity tests assert which kernel they compared)
void fva_note_kernel(const char* name);


// diagnostic stamp buffer shared by the 8-phase kernels (set by fva_conv_debug_stamps)
long long* fva_debug_stamps_ptr();
int fva_debug_stamps_rows();
