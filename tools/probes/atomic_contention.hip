// Probe (round 4): rate of 8-byte integer atomic adds when G workgroups add C channels x 2 sums into one of R replicas of a small accumulator.
// Build: hipcc -O3 --offload-arch=gfx950 -o atomic_contention atomic_contention.hip ; run: ./atomic_contention
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int WORDS>
__global__ __launch_bounds__(256) void adders(unsigned long long* acc, int C, int R, size_t stride_words, int spin) {
    // each block: 256 threads; thread t adds to sum (t / C') ... here: C channels x 2 sums = 2C adds per word, spread over the threads
    const int r = blockIdx.x % R;
    unsigned long long* a = acc + (size_t)r * stride_words;
    // a little "work" first so that blocks arrive spread out like tiles of a convolution
    float v = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    const unsigned long long inc = (unsigned long long)(long long)v | 1ull;
    for (int i = threadIdx.x; i < 2 * C * WORDS; i += 256) atomicAdd(a + i, inc);
}

int main() {
    const size_t bytes = 64ull << 20;
    unsigned long long* acc;
    hipMalloc(&acc, bytes);
    hipMemset(acc, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int Cs[] = {64, 128, 256};
    const int Gs[] = {12800, 25600};
    const int Rs[] = {1, 2, 4, 8, 16, 32, 64, 256};
    printf("words C G R stride_B us GB/s_added\n");
    for (int words = 1; words <= 2; ++words)
        for (int C : Cs)
            for (int G : Gs)
                for (int R : Rs)
                    for (int sm = 0; sm < 3; ++sm) {
                        const size_t fp = (size_t)2 * C * words;   // words per replica
                        size_t stride = sm == 0 ? fp : (sm == 1 ? (fp < 512 ? 512 : fp) + 32 : 8192 + 32);   // packed; 4 KB + 256 B; 64 KB + 256 B
                        if (R == 1 && sm) continue;
                        if (stride * R * 8 > bytes) continue;
                        float best = 1e9f;
                        for (int rep = 0; rep < 3; ++rep) {
                            hipEventRecord(e0);
                            if (words == 1) hipLaunchKernelGGL(adders<1>, dim3(G), dim3(256), 0, 0, acc, C, R, stride, 2000);
                            else hipLaunchKernelGGL(adders<2>, dim3(G), dim3(256), 0, 0, acc, C, R, stride, 2000);
                            hipEventRecord(e1);
                            hipEventSynchronize(e1);
                            float ms; hipEventElapsedTime(&ms, e0, e1);
                            if (ms < best) best = ms;
                        }
                        printf("%d %d %d %d %zu %.1f %.1f\n", words, C, G, R, stride * 8, best * 1e3, (double)G * fp * 8 / (best * 1e-3) / 1e9);
                    }
    // baseline: the spin alone
    for (int G : Gs) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(adders<1>, dim3(G), dim3(256), 0, 0, acc, 0, 1, (size_t)0, 2000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("spin-only G=%d %.1f us\n", G, ms * 1e3);
    }
    return 0;
}
