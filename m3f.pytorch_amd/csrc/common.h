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

// ---- fp16x3 GEMM operands (gemm_x6.hip / gemm_x6d.hip, NS = 4) ----------------------------------------------------------------
// An operand is staged as two fp16 terms of s x, s the power of two that puts max |x| into [2^14, 2^15) (fp16 overflows at 65504).
// bits = the fp32 bit pattern of max |x| over (a superset of) the operand, measured on the device right before the GEMM
// (gemm.hip, the backward scans).  Componentwise this is NOT fp32 (error model; tests/test_gpu_parity.py::
// test_sgemm_fp16x3_stated_limit_and_the_x6_escape_hatch): two fp16 terms carry 22 significant bits; an element more than 2^17 below
// the operand's maximum has a subnormal low term and keeps an ABSOLUTE error of at most 2^-39 of the maximum (18 bits at 2^-20 of it,
// flushed to zero below ~2^-39); normwise the product is as accurate as an fp32 GEMM, and M3T_GEMM_F16X3 off (ops.precision("x6"))
// has no such limit.  The scale's exponent is clamped to 2^126 so that its inverse stays a normal number: an operand whose maximum
// is below 2^-112 is scaled by 2^126 (not into [2^14, 2^15), but its values keep their bits) instead of producing exact zeros; an
// all-zero operand gives zeros either way.
// inf / NaN do not count towards the maximum (m3t_fin_abs): the finite values keep their scale and the non-finite ones become fp16
// inf / NaN, which poison exactly the outputs they would poison in an fp32 GEMM.
static __device__ __forceinline__ float m3t_fin_abs(float x) {          // |x|, or 0 for inf / NaN
    const unsigned b = __float_as_uint(x) & 0x7fffffffu;
    return b < 0x7f800000u ? __uint_as_float(b) : 0.f;
}
static __device__ __forceinline__ void m3t_f16_scale(unsigned bits, float& s, float& inv) {
    const int e = (int)((bits >> 23) & 0xffu);
    const int es = min(max(268 - e, 1), 253);            // biased exponent of s = 2^(14 - (e - 127)); <= 253: 1 / s = 2^(127 - es) stays normal
    s = __uint_as_float((unsigned)es << 23);
    inv = __uint_as_float((unsigned)(254 - es) << 23);
}
// Magnitude slots: 8 bytes holding (epoch << 32 | bits), only ever raised with a 64-bit atomic max -- a measurement with a larger
// epoch overrides what the slot held without a memset in between; caller-provided slots (m3t_absmax, m3t_gru_bwd_desc.amax) are
// zero-initialised by their owner and written with epoch 0.  m3t_f16x3_measure: the library's own slot pair per (device, stream) for
// operands that arrive without a slot -- stream order keeps a call's GEMM between its own measurement and the next one.  It
// measures the regions whose `have` pointer is null and returns the two pointers to use.  Regions are [rows x cols] fp32 with leading
// dimension ld (cols % 4 == 0, 16-B aligned); any SUPERSET of the operand is a valid bound.
struct M3TRegion { const float* p; unsigned long long rows; unsigned long long ld; int c4; unsigned long long* slot; };
int m3t_f16x3_measure(const M3TRegion& a, const unsigned long long* have_a, const M3TRegion& b, const unsigned long long* have_b,
                      const unsigned long long** use_a, const unsigned long long** use_b, hipStream_t s);
int m3t_absmax_regions(const M3TRegion* regs, int n, hipStream_t s);      // any n; slots raised with epoch 0 (caller-owned)
// m3t_amax_out(slot): the NEXT m3t_conv1d_fwd(_scaled) / m3t_mask_pos / m3t_mask_pos_drop / m3t_weight_norm_fwd / m3t_bct_to_btc call of the calling thread
// raises `slot` (a caller-owned, zero-initialised magnitude slot) to the bits of max |its output| -- the producer measures what the next
// fp16x3 contraction will scale by, instead of a measuring launch in front of that contraction.  take: returns and clears it.
unsigned long long* m3t_take_amax_out();
// block-wide: one 64-bit atomic max per block (callers: every thread of a 256-thread block reaches this)
static __device__ __forceinline__ void m3t_block_raise_slot(unsigned long long* slot, float mx, float* red4) {
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(red4[0], red4[1]), fmaxf(red4[2], red4[3]));
        const unsigned long long bits = (unsigned long long)__float_as_uint(m);
        // thousands of blocks on ONE address: only a block whose value the slot does not cover yet pays for the atomic (~12 ns each, serial)
        if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
    }
}
bool m3t_f16x3_enabled();                                                      // env M3T_GEMM_F16X3 != 0
