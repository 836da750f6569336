// Diagnostic: dump the lane mapping of ds_read_b64_tr_b16 and the C layout of the two MFMA shapes used.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void tr_probe(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 128];
    for (int i = threadIdx.x; i < 64 * 128; i += 64) lds[i] = (short)i;   // value = row*128 + col  (row stride 256 B)
    __syncthreads();
    const int lane = threadIdx.x, i16 = lane & 15, g = lane >> 4, q = i16 >> 2, pp = i16 & 3;
    const int row = 8 * g + q, col = 4 * pp;     // block: rows 8g..8g+3, cols 0..15
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + row * 128 + col));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
__global__ void mfma_probe(float* out16, float* out32) {
    const int lane = threadIdx.x;
    // A[i][k] = (i == k), B[k][j] = k*100 + j  -> C[i][j] = i*100 + j (i < 16/32)
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * (lane >> 4) + j;
        a[j] = (__bf16)((lane & 15) == k ? 1.f : 0.f);
        b[j] = (__bf16)(float)(k * 16 + (lane & 15));
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out16[lane * 4 + r] = c[r];
    f32x16 d = {};
    for (int kk = 0; kk < 16; ++kk) {   // 32x32x2 f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; k block kk -> global k = 2kk + (l>>5)
        const int k = 2 * kk + (lane >> 5);
        const float av = ((lane & 31) == k) ? 1.f : 0.f;
        const float bv = (float)(k * 32 + (lane & 31));
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, d, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) out32[lane * 16 + r] = d[r];
}
int main() {
    short* o; float *o16, *o32;
    hipMalloc(&o, 64 * 4 * 2); hipMalloc(&o16, 64 * 4 * 4); hipMalloc(&o32, 64 * 16 * 4);
    tr_probe<<<1, 64>>>(o);
    mfma_probe<<<1, 64>>>(o16, o32);
    short h[256]; float h16[256], h32[1024];
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(h16, o16, sizeof(h16), hipMemcpyDeviceToHost);
    hipMemcpy(h32, o32, sizeof(h32), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const int want = (8 * (l >> 4) + e) * 128 + (l & 15);   // expected: row 8g+e, column lane&15
            if (h[l * 4 + e] != want) { if (bad < 8) printf("tr lane %d e %d got (row %d col %d) want (row %d col %d)\n", l, e, h[l*4+e] / 128, h[l*4+e] % 128, want / 128, want % 128); ++bad; }
        }
    printf("tr_read mapping mismatches: %d\n", bad);
    bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int row = (l >> 4) * 4 + r, col = l & 15;
            if (h16[l * 4 + r] != (float)(row * 16 + col)) { if (bad < 8) printf("mfma16 lane %d r %d got %f want %d\n", l, r, h16[l*4+r], row*16+col); ++bad; }
        }
    printf("mfma 16x16x32 C-layout mismatches: %d\n", bad);
    bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
            if (h32[l * 16 + r] != (float)(row * 32 + col)) { if (bad < 8) printf("mfma32 lane %d r %d got %f want %d\n", l, r, h32[l*16+r], row*32+col); ++bad; }
        }
    printf("mfma 32x32x2 f32 C-layout mismatches: %d\n", bad);
    return 0;
}
