// cycles per MFMA instruction on one SIMD: legacy v_mfma_f32_16x16x16_f16 vs gfx950's v_mfma_f32_16x16x32_f16 (one wave per SIMD and two)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ void k(long long* out, float* sink, int n) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    f16x4 x4 = {(_Float16)threadIdx.x, 1, 2, 3}, y4 = {1, (_Float16)(threadIdx.x & 7), 1, 1};
    f16x8 x8 = {1, 2, 3, 4, 5, 6, 7, (_Float16)threadIdx.x}, y8 = {1, 1, 1, 1, 1, 1, 1, (_Float16)(threadIdx.x & 3)};
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x16f16(x4, y4, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x16f16(x4, y4, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x16f16(x4, y4, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x16f16(x4, y4, a3, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x8, y8, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x8, y8, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x8, y8, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x8, y8, a3, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
int main() {
    long long* d; float* s; hipMalloc(&d, 8); hipMalloc(&s, 4 * 512 * 256);
    const int n = 4096;
    for (int thr : {256, 512}) {
        long long h0, h1;
        k<0><<<256, thr>>>(d, s, n); hipMemcpy(&h0, d, 8, hipMemcpyDeviceToHost);
        k<1><<<256, thr>>>(d, s, n); hipMemcpy(&h1, d, 8, hipMemcpyDeviceToHost);
        printf("%d threads/WG (%d wave(s) per SIMD): 16x16x16f16 %.1f cyc per MFMA per wave, 16x16x32_f16 %.1f\n", thr, thr / 256, (double)h0 / (4.0 * n), (double)h1 / (4.0 * n));
    }
    return 0;
}
