// Shared device helpers for the gfx950 kernels (wave = 64 lanes, hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "m3t_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define M3T_WAVE 64

#define M3T_LAUNCH_CHECK()                          \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
static __device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Block-wide sum for blockDim.x <= 1024 (deterministic order). `red` >= 16 floats of LDS.
static __device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- in-kernel dropout masks (reference models/tcn.py:17,23,29: nn.Dropout(p) after each ReLU) --------------------------------
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11), counter-based: the mask of element
// (row, col) of a [rows, C] activation is word (row & 3) of philox(counter = {col, row >> 2, 0, 0}, key = seed): keep (x 1/(1-p))
// iff word < thr = (1-p) * 2^32.  Forward epilogues and the backward pass regenerate the same words from (seed, row, col): the mask
// never exists in HBM.  oracle/m3t_oracle.py restates the generator in numpy so that train-mode outputs have a parity target.
struct M3TDrop { uint32_t k0, k1, thr; float scale; int on; };
static inline M3TDrop m3t_make_drop(float p, unsigned long long seed) {
    M3TDrop d;
    d.on = p > 0.f ? 1 : 0;
    const double keep = 1.0 - (double)p;
    const double t = keep * 4294967296.0;
    d.thr = t >= 4294967295.0 ? 0xffffffffu : (t <= 0.0 ? 0u : (uint32_t)t);
    d.scale = keep > 0.0 ? (float)(1.0 / keep) : 0.f;
    d.k0 = (uint32_t)seed; d.k1 = (uint32_t)(seed >> 32);
    return d;
}
static __device__ __forceinline__ void m3t_drop_mask4(const M3TDrop& d, uint32_t g, uint32_t col, float (&m)[4]) {
    uint32_t c0 = col, c1 = g, c2 = 0u, c3 = 0u, k0 = d.k0, k1 = d.k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    m[0] = c0 < d.thr ? d.scale : 0.f; m[1] = c1 < d.thr ? d.scale : 0.f;
    m[2] = c2 < d.thr ? d.scale : 0.f; m[3] = c3 < d.thr ? d.scale : 0.f;
}
