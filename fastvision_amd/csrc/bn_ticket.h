// BatchNorm finalisation INSIDE the launch that produces the partial statistics ("ticket" form), gfx950.
//
// A convolution epilogue leaves one row of partial sums per row block: part[row][2][C] (forward: sum y, sum y^2; backward:
// sum dU, sum dU * xhat).  Until round 3 a second launch (bn_finalize / bn_bwd_finalize, plus bn_prereduce for long tables)
// folded the table: 197 latency-bound launches per YOLOv3 step, 2.0 ms.  Here the table is folded by the waves that wrote it:
//
//   * the table is cut into channel slices of 32 (one wave: lanes 0-31 carry sum 0, lanes 32-63 sum 1 of channel
//     slice * 32 + (lane & 31)) and, when long, into groups of G consecutive rows;
//   * every (row, slice) pair is written by exactly ONE wave, with write-through (agent-scope) stores, 128 bytes = one cache
//     line per sum; the wave then drains its stores and draws a ticket from counter[group][slice];
//   * the wave that draws the LAST ticket of a (group, slice) folds that group's rows in row order in double, and -- when there
//     are several groups -- stores the two doubles per channel write-through, drains, and draws a ticket from counter2[slice];
//     the last of those folds the groups in group order and finalises the slice's 32 channels (mean / rstd / scale / shift and
//     the running statistics, or dgamma / dbeta / the pass-2 coefficients).
//
// Fixed order everywhere => the result does not depend on which wave happens to be last: bit-identical from launch to launch.
// Hand-off (cdna_hip_programming.md G16, MI355X_MICROARCH.md "Valid forms"): payload stored sc1, storing wave waits vmcnt(0),
// then its agent-scope atomic add; the consumer is the wave whose add returned last, it loads only after the add has returned,
// every load of handed-off bytes is an sc1 load (which bypasses the reader's L1: the agent-scope acquire it replaces would only
// invalidate that L1).  Every line is written by one wave and read by one wave within a launch, and launch boundaries invalidate
// the caches, so no reader can hold a stale copy.  The last arriver puts the counters back to
// zero: the caller zero-fills them once (hipMemset / torch.zeros) and they stay usable launch after launch, graph replays
// included.  Several launches may fill ONE table (the parity launches of a stride-2 dgrad): tickets simply span them.
#pragma once
#include "common.h"

struct BnTicket {
    int mode;             // 0 off, 1 forward statistics, 2 backward statistics
    int rows, G, ngroups; // rows of the whole table, rows per group, groups
    int C;                // channels (multiple of 32)
    int32_t* counters;    // [ngroups][C/32] then [C/32]
    double* gsum;         // [ngroups][2][C], used when ngroups > 1
    const float* part;    // the table
    double count;         // elements per channel
    const float *gamma, *beta;
    float *running_mean, *running_var;
    long long* nbt;
    float momentum, eps;
    float *save_mean, *save_rstd, *scale, *shift;   // forward results
    const float* rstd;                              // backward input
    float *dgamma, *dbeta, *coef;                   // backward results (coef: [3][C])
    int accumulate;
};

// The descriptor is read where it is used, through the kernel-argument segment: taken from the by-value parameter the compiler
// hoists its 40 dwords above the convolution's main loop and spills them (igemm8_kernel<EPI_BNB>: 82 SGPR spills, 20 bytes of
// scratch).  `offset` = offsetof(first kernel parameter, ticket member).
typedef const __attribute__((address_space(4))) BnTicket* BnTicketPtr;
__device__ __forceinline__ BnTicketPtr bn_ticket_kernarg(int offset) {
    BnTicketPtr t = (BnTicketPtr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offset);
    asm volatile("" : "+s"(t));
    return t;
}

using gf32 = __attribute__((address_space(1))) float;
using gf64 = __attribute__((address_space(1))) double;
using gi32 = __attribute__((address_space(1))) int32_t;

__device__ __forceinline__ void st_agent(float* p, float v) {
    __hip_atomic_store((gf32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
    return __hip_atomic_load((gf32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double* p, double v) {
    __hip_atomic_store((gf64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
    return __hip_atomic_load((gf64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// rows per group for a table of `rows` rows: one level up to 96 rows (three chunks of loads), else about sqrt(rows) (multiple of 8)
static inline int bn_ticket_group_rows(int rows) {
    if (rows <= 96) return rows > 0 ? rows : 1;
    int g = 8;
    while ((int64_t)g * g < rows) g += 8;
    return g;
}

// Called by a WHOLE wave (all 64 lanes, wave-uniform arguments) after it has stored its 64 partial sums of table row `trow`,
// channel slice `cslice`, with st_agent().  Returns after the ticket unless this wave is the last of its group.
__device__ __forceinline__ void bn_ticket_arrive(BnTicketPtr tp, int trow, int cslice, int lane) {
    const auto& t = *tp;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's partial sums have been written through
    const int nsl = t.C >> 5;
    const int g = trow / t.G;
    const int g_rows = t.rows - g * t.G < t.G ? t.rows - g * t.G : t.G;
    int32_t* cnt1 = t.counters + g * nsl + cslice;
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add((gi32*)cnt1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != g_rows - 1) return;
    // no agent-scope acquire here: it would only invalidate this CU's L1, which the sc1 loads below bypass, and cost 1.7 us of
    // the launch's serial tail (MI355X_MICROARCH.md, fence table); the compiler barrier keeps the loads behind the ticket
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int which = lane >> 5, c = (cslice << 5) + (lane & 31);
    double acc = 0.0;
    {
        // 32 rows in flight per lane: every chunk is one memory round trip (1.5-2 us under load) of the launch's tail.  Rows
        // beyond the group re-read its last row and add 0.0 (no branch around a load: hipcc would wait for each one).
        const float* src = t.part + ((int64_t)g * t.G * 2 + which) * t.C + c;
        const int64_t pitch = 2 * (int64_t)t.C;
        for (int r = 0; r < g_rows; r += 32) {
            float v[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) v[j] = ld_agent(src + (r + j < g_rows ? r + j : g_rows - 1) * pitch);
#pragma unroll
            for (int j = 0; j < 32; ++j) acc += r + j < g_rows ? (double)v[j] : 0.0;
        }
    }
    if (lane == 0) __hip_atomic_store((gi32*)cnt1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t.ngroups > 1) {
        st_agent(t.gsum + ((int64_t)g * 2 + which) * t.C + c, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int32_t* cnt2 = t.counters + t.ngroups * nsl + cslice;
        int old2 = 0;
        if (lane == 0) old2 = __hip_atomic_fetch_add((gi32*)cnt2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old2 = __builtin_amdgcn_readfirstlane(old2);
        if (old2 != t.ngroups - 1) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        acc = 0.0;
        const double* gs = t.gsum + (int64_t)which * t.C + c;
        const int64_t gp = 2 * (int64_t)t.C;
        for (int gg = 0; gg < t.ngroups; gg += 16) {
            double v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ld_agent(gs + (gg + j < t.ngroups ? gg + j : t.ngroups - 1) * gp);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += gg + j < t.ngroups ? v[j] : 0.0;
        }
        if (lane == 0) __hip_atomic_store((gi32*)cnt2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the slice's 32 channels: lanes 0-31 hold sum 0, lanes 32-63 sum 1 ----
    const double s1 = __shfl(acc, lane & 31), s2 = __shfl(acc, (lane & 31) + 32);
    if (lane >= 32) return;
    if (t.mode == 1) {
        const double mean = s1 / t.count;
        double var = s2 / t.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)t.eps));
        const float gm = t.gamma[c], bt = t.beta[c];
        t.save_mean[c] = (float)mean;
        t.save_rstd[c] = rstd;
        const float sc = gm * rstd;
        t.scale[c] = sc;
        t.shift[c] = bt - (float)mean * sc;
        if (t.running_mean) {
            const double unbiased = t.count > 1.0 ? var * t.count / (t.count - 1.0) : var;
            t.running_mean[c] = (1.f - t.momentum) * t.running_mean[c] + t.momentum * (float)mean;
            t.running_var[c] = (1.f - t.momentum) * t.running_var[c] + t.momentum * (float)unbiased;
        }
        if (t.nbt && c == 0) *t.nbt += 1;
    } else {
        const float db = (float)s1, dg = (float)s2;
        t.dbeta[c] = t.accumulate ? t.dbeta[c] + db : db;
        t.dgamma[c] = t.accumulate ? t.dgamma[c] + dg : dg;
        const float a = t.gamma[c] * t.rstd[c];
        t.coef[c] = a;
        t.coef[t.C + c] = (float)(-(double)a * s2 / t.count);
        t.coef[2 * t.C + c] = (float)(-(double)a * s1 / t.count);
    }
}
