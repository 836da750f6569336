// Diagnostic: how fast can a CU stream an activation tensor into LDS by LDS-DMA (global_load_lds, 16 B per lane), as a function of
// the number of 16-KiB pieces each wave keeps in flight (counted vmcnt) and of the blocks per CU?  No arithmetic: this is the
// memory phase alone of a persistent, software-pipelined convolution kernel, i.e. its upper bound.
// Also: the same traffic as plain 16-byte global loads into registers (what the BatchNorm kernels do).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_dma tools/probe_dma.hip && tools/probe_dma [MiB [reps]]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Each block of 256 threads streams its contiguous share of the buffer in pieces of 16 KiB (4 waves x 4 instructions x 1 KiB),
// DEPTH pieces in flight per wave; a piece is "consumed" by one ds_read per lane (so that the data really has to land).
template <int DEPTH>
__global__ __launch_bounds__(256) void dma_stream(const char* __restrict__ src, long long bytes, float* __restrict__ sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // DEPTH x 16 KiB
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long long per = (bytes / gridDim.x) & ~16383ll;
    const char* p = src + (long long)blockIdx.x * per + (w * 4) * 1024 + lane * 16;
    const int span = (int)(per >> 14);          // pieces of this block's share; the share is walked `reps` times (reps > 1: L2-resident)
    const int pieces = span * reps;
    float acc = 0.f;
    auto issue = [&](int piece, int slot) {
        const char* q = p + (long long)(piece % span) * 16384;
        char* d = smem + slot * 16384 + w * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds(GLB_PTR(q + i * 1024), LDS_PTR(d + i * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i)
        if (i < pieces) issue(i, i);
    for (int k = 0; k < pieces; ++k) {
        if (k + DEPTH - 1 < pieces) issue(k + DEPTH - 1, (k + DEPTH - 1) % DEPTH);
        // everything but the DEPTH-1 newest pieces (4 instructions each) has landed
        if (k + DEPTH - 1 < pieces) wait_vmcnt<4 * (DEPTH - 1)>();
        else wait_vmcnt<0>();
        acc += *(const float*)(smem + (k % DEPTH) * 16384 + w * 4096 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void reg_stream(const f32x4* __restrict__ src, long long n16, float* __restrict__ sink) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const long long stride = (long long)gridDim.x * 256;
    long long i = blockIdx.x * 256ll + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const f32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        s += (a + b) + (c + d);
    }
    for (; i < n16; i += stride) s += src[i];
    if (s[0] == 123.456f) sink[blockIdx.x] = s[1];
}

template <int DEPTH>
static float run_dma(const char* buf, long long bytes, float* sink, int blocks, int iters, int reps) {
    hipFuncSetAttribute((const void*)dma_stream<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 16384);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) dma_stream<DEPTH><<<blocks, 256, DEPTH * 16384>>>(buf, bytes, sink, reps);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) dma_stream<DEPTH><<<blocks, 256, DEPTH * 16384>>>(buf, bytes, sink, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main(int argc, char** argv) {
    const long long bytes = (argc > 1 ? atoll(argv[1]) : 2048ll) << 20;   // MiB; default 2 GiB: far beyond L2 + Infinity Cache
    char* buf; float* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 1 << 20) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("%d CUs, buffer %lld MiB\n", cus, bytes >> 20);
    const int iters = 5;
    const int reps = argc > 2 ? atoi(argv[2]) : 1;   // > 1: every block walks its share that many times per launch (cache-resident rates)
    const double moved = (double)bytes * reps;
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        const int blocks = cus * bpc;
        const float t1 = run_dma<1>(buf, bytes, sink, blocks, iters, reps), t2 = run_dma<2>(buf, bytes, sink, blocks, iters, reps);
        const float t3 = run_dma<3>(buf, bytes, sink, blocks, iters, reps), t4 = bpc <= 2 ? run_dma<4>(buf, bytes, sink, blocks, iters, reps) : 0.f;
        printf("LDS-DMA x%d, %d block(s) of 4 waves per CU: 16-KiB pieces in flight 1: %.2f TB/s  2: %.2f  3: %.2f  4: %.2f\n", reps, bpc,
               moved / t1 / 1e9, moved / t2 / 1e9, moved / t3 / 1e9, t4 > 0 ? moved / t4 / 1e9 : 0.0);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bpc = 2; bpc <= 16; bpc *= 2) {
        reg_stream<<<cus * bpc, 256>>>((const f32x4*)buf, bytes / 16, sink);
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) reg_stream<<<cus * bpc, 256>>>((const f32x4*)buf, bytes / 16, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("register loads (4 x 16 B in flight per lane), %d blocks per CU: %.2f TB/s\n", bpc, bytes / (ms / iters) / 1e9);
    }
    return hipDeviceSynchronize() == hipSuccess ? 0 : 2;
}
