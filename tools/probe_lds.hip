// Diagnostic: LDS read throughput per CU of the read forms the kernels use (bytes per clock, all CUs busy).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_lds.hip -o tools/probe_lds && tools/probe_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// MODE 0: ds_read_b64_tr_b16, rows of 128 B with the (bit 1, bit 3) block swizzle (conflict-free by half-wave)
// MODE 1: ds_read_b64_tr_b16, rows of 64 B with the bit-3 swizzle
// MODE 2: ds_read_b64 lane-linear (8 B per lane, 512 B contiguous)
// MODE 3: ds_read_b128 lane-linear (16 B per lane, 1 KiB contiguous)
// MODE 4: ds_read_b64_tr_b16 without a swizzle, rows of 128 B (conflicting)
template <int MODE>
__global__ __launch_bounds__(1024) void probe(unsigned* sink, int iters, int off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i16 = lane & 15, g = lane >> 4, q = i16 >> 2, pp = i16 & 3;
    const int row = off + 8 * g + q;
    unsigned addr;
    if (MODE == 0) addr = row * 128 + (((w & 3) ^ (((row >> 1) & 1) | (((row >> 3) & 1) << 1))) << 5) + pp * 8;
    else if (MODE == 1) addr = row * 64 + (((w & 1) ^ ((row >> 3) & 1)) << 5) + pp * 8;
    else if (MODE == 2) addr = lane * 8;
    else if (MODE == 3) addr = lane * 16;
    else addr = row * 128 + ((w & 3) << 5) + pp * 8;
    addr += (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (w >> 2) * 4096;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
            u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:4096\n ds_read_b128 %2, %8 offset:8192\n ds_read_b128 %3, %8 offset:12288\n"
                         "ds_read_b128 %4, %8 offset:16384\n ds_read_b128 %5, %8 offset:20480\n ds_read_b128 %6, %8 offset:24576\n ds_read_b128 %7, %8 offset:28672\n"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(addr));
            acc ^= v0[0] ^ v1[1] ^ v2[2] ^ v3[3] ^ v4[0] ^ v5[1] ^ v6[2] ^ v7[3];
        } else if (MODE == 2) {
            u32x2 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:4096\n ds_read_b64 %2, %8 offset:8192\n ds_read_b64 %3, %8 offset:12288\n"
                         "ds_read_b64 %4, %8 offset:16384\n ds_read_b64 %5, %8 offset:20480\n ds_read_b64 %6, %8 offset:24576\n ds_read_b64 %7, %8 offset:28672\n"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(addr));
            acc ^= v0[0] ^ v1[1] ^ v2[0] ^ v3[1] ^ v4[0] ^ v5[1] ^ v6[0] ^ v7[1];
        } else {
            u32x2 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b64_tr_b16 %0, %8\n ds_read_b64_tr_b16 %1, %8 offset:4096\n ds_read_b64_tr_b16 %2, %8 offset:8192\n ds_read_b64_tr_b16 %3, %8 offset:12288\n"
                         "ds_read_b64_tr_b16 %4, %8 offset:16384\n ds_read_b64_tr_b16 %5, %8 offset:20480\n ds_read_b64_tr_b16 %6, %8 offset:24576\n ds_read_b64_tr_b16 %7, %8 offset:28672\n"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(addr));
            acc ^= v0[0] ^ v1[1] ^ v2[0] ^ v3[1] ^ v4[0] ^ v5[1] ^ v6[0] ^ v7[1];
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int MODE>
void run(const char* name, int bytes_per_read, unsigned* sink) {
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int waves_list[3] = {4, 8, 16};
    for (int off = 0; off < (MODE <= 1 ? 8 : 1); ++off)
    for (int wi = 1; wi < 2; ++wi) {
        const int waves = waves_list[wi], iters = 20000, blocks = 256;
        probe<MODE><<<blocks, waves * 64, 65536>>>(sink, 100, off);
        hipEventRecord(e0);
        probe<MODE><<<blocks, waves * 64, 65536>>>(sink, iters, off);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)waves * iters * 8 * 64 * bytes_per_read;   // per CU
        printf("%-44s row offset %d %2d waves/CU: %7.1f GB/s per CU = %5.1f B/clk at 2.1 GHz\n", name, off, waves, bytes / ms / 1e6, bytes / ms / 1e6 / 2.1);
    }
}

int main() {
    unsigned* sink;
    hipMalloc(&sink, 4);
    run<0>("ds_read_b64_tr_b16, 128 B rows, swizzled", 8, sink);
    run<1>("ds_read_b64_tr_b16, 64 B rows, swizzled", 8, sink);
    run<4>("ds_read_b64_tr_b16, 128 B rows, no swizzle", 8, sink);
    run<2>("ds_read_b64 lane-linear", 8, sink);
    run<3>("ds_read_b128 lane-linear", 16, sink);
    return 0;
}
