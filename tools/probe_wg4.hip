// Diagnostic: the inner loop of a weight-gradient block as FOUR waves with 128 x 128 wave tiles (256 fp32 accumulators per lane, one wave per
// SIMD) against EIGHT waves with 128 x 64 tiles (the 8-phase kernel's arrangement), fragments by ds_read_b64_tr_b16 from a pre-filled LDS
// image, no global traffic: what the matrix pipe reaches when only LDS reads and barriers stand beside the MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_wg4.hip -o tools/probe_wg4 && tools/probe_wg4
#include <hip/hip_runtime.h>
#include <stdio.h>
#ifndef PF_ATOMIC
#define PF_ATOMIC 0
#endif
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int OFF>
__device__ __forceinline__ s16x4 tr_read(unsigned addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ bf16x8 cat8(s16x4 lo, s16x4 hi) {
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// LDS image per buffer: A [64 px][256 ch] and B [64 px][256 ch] as four half images [64 px][128 ch] (256-byte rows), the kernel's layout.
// MT x NT = 16 x 16 tiles per wave along rows / columns; WAVES = 256 * 256 / (MT * NT * 256).
// DM: LDS-DMA of the NEXT step's 64 KB image (issued at the start of a step, awaited at its end; 1-KiB instructions of ROWS rows x 1024 / ROWS bytes)
//   0 none | 1 the same 64 KB of the block every step (cache hits) | 2 a stream of the block's own (no reuse) | 3 a stream shared by 5 blocks
template <int MT, int NT, int WAVES, int DM, int ROWS>
__global__ __launch_bounds__(WAVES * 64) void probe(float* sink, int steps, const char* src, long long stride_step, int* progress) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 2 * 65536 / 4; i += WAVES * 64) ((unsigned*)smem)[i] = 0x3c003c00u + (i & 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int WCOLS = 256 / (NT * 16);                 // waves along the columns
    const int wr = w / WCOLS, wc = w % WCOLS;
    const int i16 = lane & 15, g = lane >> 4, q = i16 >> 2, pp = i16 & 3;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // fragment addresses: rows 8g + 4hh + q of a 256-byte-row half image, 32-byte blocks swizzled as in the kernel
    unsigned ra[2][MT], rb[2][NT];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int row = 8 * g + 4 * hh + q;
        const int fr = (q << 2) | ((2 * g + hh) & 3);
        const int rbase = row * 256 + 8 * (pp & 1);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int tile = wr * MT + t;                  // 16-channel tile of A: half image tile / 8, block (tile % 8) * 2 + ..
            ra[hh][t] = lds0 + (tile >> 3) * 16384 + rbase + ((((tile & 7) * 2 + (pp >> 1)) ^ fr) << 4);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tile = wc * NT + t;
            rb[hh][t] = lds0 + 32768 + (tile >> 3) * 16384 + rbase + ((((tile & 7) * 2 + (pp >> 1)) ^ fr) << 4);
        }
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // DMA source of this lane: instruction j (of 64 / WAVES per wave) covers ROWS rows of 1024 / ROWS bytes, rows 512 bytes apart
    // DM 3: the five sharers sit on ONE XCD (workgroups go round-robin over the 8 XCDs), as the kernels' block remap arranges it
    const int blk = DM == 3 ? ((int)blockIdx.x & 7) * 7 + ((int)blockIdx.x >> 3) / 5 : (int)blockIdx.x;
    const char* base = src + (DM == 1 ? (long long)blockIdx.x * 65536 : (long long)blk * 65536);
    constexpr int LPR = 64 / ROWS;                          // lanes per row
    const int drow = lane / LPR, dcol = (lane % LPR) * 16;
    for (int s = 0; s < steps; ++s) {
        const unsigned boff = (s & 1) * 65536;
        // progress report: ONE block per XCD, a plain agent-scope store into the XCD's own cache line (256 returning atomics on one line
        // per step cost 3 us per step: the step's vmcnt(0) waits for them)
        if (DM == 3 && progress && steps < 0 && threadIdx.x == 0 && (blockIdx.x >> 3) == 0) __hip_atomic_store(progress + (blockIdx.x & 7) * 32, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (DM != 0) {
            const char* sb = base + (DM == 1 ? 0ll : (long long)(s + 1) * stride_step);
#pragma unroll
            for (int i = 0; i < 64 / WAVES; ++i) {
                const int j = i * WAVES + w;
                // the 64 KB image as 64 instructions of ROWS rows x 1024 / ROWS contiguous bytes (the region itself is contiguous)
                const char* g = sb + (long long)(j * ROWS + drow) * (1024 / ROWS) + dcol;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(smem + ((s + 1) & 1) * 65536 + j * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {                   // two 32-pixel halves of the 64-pixel k-step
            bf16x8 af[MT], bf[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t) af[t] = kk == 0 ? cat8(tr_read<0>(ra[0][t] + boff), tr_read<0>(ra[1][t] + boff))
                                                         : cat8(tr_read<8192>(ra[0][t] + boff), tr_read<8192>(ra[1][t] + boff));
#pragma unroll
            for (int t = 0; t < NT; ++t) bf[t] = kk == 0 ? cat8(tr_read<0>(rb[0][t] + boff), tr_read<0>(rb[1][t] + boff))
                                                         : cat8(tr_read<8192>(rb[0][t] + boff), tr_read<8192>(rb[1][t] + boff));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DM != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) t += acc[i][j][0] + acc[i][j][3];
    if (t == 12345.f) sink[0] = t;
}

// The same work as probe<8, 4, 8, 3> with 32-pixel k-steps and a ring of five 32-KB LDS buffers (160 KB): the DMA of step s + 4 is issued
// at step s, so that four steps (two 64-pixel steps) lie between issue and use instead of one.  vmcnt(12) at the end of a step leaves the
// three youngest steps' DMA (4 instructions per wave each) in flight.
__global__ __launch_bounds__(512) void probe_ring(float* sink, int steps32, const char* src, long long stride_step) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 5 * 32768 / 4; i += 512) ((unsigned*)smem)[i] = 0x3c003c00u + (i & 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 2, wc = w & 3;
    const int i16 = lane & 15, g = lane >> 4, q = i16 >> 2, pp = i16 & 3;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // buffer = A [32 px][256 ch] as two half images [32 px][128 ch] (8 KB each), then B likewise
    unsigned ra[2][8], rb[2][4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int row = 8 * g + 4 * hh + q;
        const int fr = (q << 2) | ((2 * g + hh) & 3);
        const int rbase = row * 256 + 8 * (pp & 1);
#pragma unroll
        for (int t = 0; t < 8; ++t) { const int tile = wr * 8 + t; ra[hh][t] = lds0 + (tile >> 3) * 8192 + rbase + ((((tile & 7) * 2 + (pp >> 1)) ^ fr) << 4); }
#pragma unroll
        for (int t = 0; t < 4; ++t) { const int tile = wc * 4 + t; rb[hh][t] = lds0 + 16384 + (tile >> 3) * 8192 + rbase + ((((tile & 7) * 2 + (pp >> 1)) ^ fr) << 4); }
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int blk = ((int)blockIdx.x & 7) * 7 + ((int)blockIdx.x >> 3) / 5;
    const char* base = src + (long long)blk * 65536 + lane * 16;
    auto issue = [&](int step) {                          // 32 KB = 32 instructions, 4 per wave; two 32-pixel steps share a 64-KB region
        const char* sb = base + (long long)(step >> 1) * stride_step + (step & 1) * 32768;
        char* dst = smem + (step % 5) * 32768;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = i * 8 + w;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + j * 1024), (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
        }
    };
    for (int s = 1; s <= 4; ++s) issue(s + 1);            // steps 2..5 in flight (step 0 / 1 use the initial LDS contents)
    for (int s = 0; s < steps32; ++s) {
        const unsigned boff = (s % 5) * 32768;
        bf16x8 af[8], bf[4];
#pragma unroll
        for (int t = 0; t < 8; ++t) af[t] = cat8(tr_read<0>(ra[0][t] + boff), tr_read<0>(ra[1][t] + boff));
#pragma unroll
        for (int t = 0; t < 4; ++t) bf[t] = cat8(tr_read<0>(rb[0][t] + boff), tr_read<0>(rb[1][t] + boff));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // step s + 1 has landed (steps s + 2 .. s + 4 may fly)
        __builtin_amdgcn_s_barrier();
        if (s + 5 < steps32 + 5) issue(s + 5);             // into the buffer every wave has just finished reading
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][3];
    if (t == 12345.f) sink[0] = t;
}

// L2 prefetcher for DM 3: one wave per XCD walks the XCD's seven shared streams LEAD steps ahead of the slowest progress report with
// plain loads whose results are dropped (at most 56 in flight), paced by the compute blocks' progress counter; gives up after a while.
__global__ __launch_bounds__(256) void prefetcher(const char* src, long long stride_step, int steps, int lead, int* progress, unsigned* sink) {
    const int xcd = blockIdx.x & 7, lane = threadIdx.x & 63, pw = threadIdx.x >> 6;   // four waves per XCD: 114 GB/s per wave is what one wave of line-per-lane loads delivers
    unsigned acc = 0;
    const unsigned zero = 0;
    const long long t0 = wall_clock64();                  // 100 MHz
    for (int p = 1; p <= steps; ++p) {
        int spins = 0;
        if (lead >= 100) {                                 // open loop: step p not before t0 + (p - 3) * lead * 10 ns
            while (wall_clock64() - t0 < (long long)(p - 3) * lead && spins < 200000) { __builtin_amdgcn_s_sleep(2); ++spins; }
        } else {
            while (__hip_atomic_load(progress + xcd * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + lead < p && spins < 200000) {
                __builtin_amdgcn_s_sleep(8);
                ++spins;
            }
        }
        if (spins >= 200000) break;
        for (int grp = 0; grp < 7; ++grp) {
            const char* sb = src + (long long)(xcd * 7 + grp) * 65536 + (long long)p * stride_step + lane * 128;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {                // fire and forget: the result register is never read before the final wait
                const char* g = sb + (pw * 2 + jj) * 8192;
#if PF_ATOMIC
                asm volatile("global_atomic_or %0, %1, off" :: "v"(g), "v"(zero) : "memory");   // executes at L2, returns nothing to the CU
#else
                asm volatile("global_load_dword %0, %1, off" : "+v"(acc) : "v"(g));
#endif
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345u) sink[0] = acc;
}

template <int MT, int NT, int WAVES, int DM = 0, int ROWS = 4>
void run(const char* name, float* sink, const char* src = nullptr, int lead = -1) {
    hipFuncSetAttribute((const void*)probe<MT, NT, WAVES, DM, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int steps = DM >= 2 ? 60 : 2000;                 // a stream: 60 steps x 256 blocks x 64 KB = 1 GB (one weight-gradient launch)
    const long long stride = DM == 2 ? 256ll * 65536 : DM == 3 ? 56ll * 65536 : 0;
    static int* progress = nullptr;
    static hipStream_t s1 = nullptr, s2 = nullptr;
    if (!progress) { hipMalloc(&progress, 1024); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); }
    probe<MT, NT, WAVES, DM, ROWS><<<256, WAVES * 64, 131072>>>(sink, 10, src, stride, nullptr);
    hipDeviceSynchronize();
    float ms = 0.f;
    for (int r = 0; r < (DM >= 2 ? 10 : 1); ++r) {
        if (lead >= 0) {
            hipMemsetAsync(progress, 0, 1024, 0);
            hipStreamSynchronize(0);
            prefetcher<<<8, 256, 0, s2>>>(src, stride, steps, lead, progress, (unsigned*)sink);
        }
        hipEventRecord(e0, s1);
        probe<MT, NT, WAVES, DM, ROWS><<<256, WAVES * 64, 131072, s1>>>(sink, steps, src, stride, lead >= 0 ? progress : nullptr);
        hipEventRecord(e1, s1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        ms += t;
        if (lead >= 0) hipStreamSynchronize(s2);
    }
    const int reps = DM >= 2 ? 10 : 1;
    const double flop = 256.0 * steps * reps * 2.0 * 256 * 256 * 64;
    printf("%-64s %8.1f us  %7.1f TFLOP/s  (%.0f ns per 64-pixel k-step)\n", name, ms * 1e3 / reps, flop / ms / 1e9, ms * 1e6 / steps / reps);
}

int main() {
    float* sink;
    hipMalloc(&sink, 4);
    run<8, 4, 8>("8 waves, 128 x 64 wave tiles, serial halves", sink);
    run<8, 8, 4>("4 waves, 128 x 128 wave tiles, serial halves", sink);
    run<4, 8, 8>("8 waves, 64 x 128 wave tiles, serial halves", sink);
    char* src;
    hipMalloc(&src, (size_t)(62 * 256 + 8) * 65536);        // ~1 GB
    hipMemset(src, 0x3c, (size_t)(62 * 256 + 8) * 65536);
    run<8, 4, 8, 1, 4>("8 waves 128 x 64 + DMA, the same 64 KB every step (4 x 256 B)", sink, src);
    run<8, 4, 8, 1, 8>("8 waves 128 x 64 + DMA, the same 64 KB every step (8 x 128 B)", sink, src);
    run<8, 4, 8, 2, 4>("8 waves 128 x 64 + DMA, own stream, no reuse (4 x 256 B)", sink, src);
    run<8, 4, 8, 3, 4>("8 waves 128 x 64 + DMA, stream shared by 5 blocks (4 x 256 B)", sink, src);
    run<8, 4, 8, 3, 8>("8 waves 128 x 64 + DMA, stream shared by 5 blocks (8 x 128 B)", sink, src);
    run<8, 8, 4, 3, 4>("4 waves 128 x 128 + DMA, stream shared by 5 blocks (4 x 256 B)", sink, src);
    run<8, 4, 8, 3, 4>("... + L2 prefetch wave per XCD, open loop 1.6 us per step", sink, src, 160);
    {
        hipFuncSetAttribute((const void*)probe_ring, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        probe_ring<<<256, 512, 163840>>>(sink, 20, src, 56ll * 65536);
        hipDeviceSynchronize();
        printf("ring launch: %s\n", hipGetErrorString(hipGetLastError()));
        float ms = 0.f;
        for (int r = 0; r < 10; ++r) {
            hipEventRecord(e0);
            probe_ring<<<256, 512, 163840>>>(sink, 120, src, 56ll * 65536);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); ms += t;
        }
        printf("%-64s %8.1f us  %7.1f TFLOP/s  (%.0f ns per 64 pixels)\n", "8 waves 128 x 64, 32-pixel steps, ring of five 32-KB buffers, shared by 5", ms * 100,
               256.0 * 60 * 10 * 2.0 * 256 * 256 * 64 / ms / 1e9, ms * 1e6 / 60 / 10);
    }
    if (0) {   // do the two kernels overlap at all?  wall-clock of each alone and of both (compute launched FIRST here)
        hipStream_t a, b;
        hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
        hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
        int* prog;
        hipMalloc(&prog, 1024);
        hipMemset(prog, 0, 1024);
        auto wall = [&](int which) {
            hipDeviceSynchronize();
            hipEvent_t e0, e1, e2, e3;
            hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
            const long long stride = 56ll * 65536;
            hipEventRecord(e0, a);
            if (which & 1) probe<8, 4, 8, 3, 4><<<256, 512, 131072, a>>>(sink, 60, src, stride, nullptr);
            hipEventRecord(e1, a);
            hipEventRecord(e2, b);
            if (which & 2) prefetcher<<<8, 256, 0, b>>>(src, stride, 60, 160, prog, (unsigned*)sink);
            hipEventRecord(e3, b);
            hipDeviceSynchronize();
            float ta, tb, tab;
            hipEventElapsedTime(&ta, e0, e1); hipEventElapsedTime(&tb, e2, e3); hipEventElapsedTime(&tab, e0, e3);
            printf("which %d: compute %.1f us, prefetcher %.1f us, first start to last end %.1f us\n", which, ta * 1e3, tb * 1e3, tab * 1e3);
        };
        wall(1); wall(2); wall(3); wall(3);
    }
    return 0;
}
