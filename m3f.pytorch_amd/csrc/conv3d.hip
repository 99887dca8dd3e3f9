// Patch matrix (im2col) of a Conv3d input for the weight-gradient GEMM dW = dy^T P (m3t.ops.conv3d; reference
// models/backbone.py:73-103,179-271,327-332: the 3-D conv stems).  Round 3 built P with torch (unfold x 3, permute, reshape): the
// strided copy of an 8-D view ran as ~50 `direct_copy` launches per convolution (344 per C5 step, 8.2 % of it) and the fp16x3 GEMM then
// measured the matrix's magnitude with one more pass (112 `f16x3_absmax` launches, 3.4 %).  ONE launch here writes P -- rows (n, t', h',
// w'), columns (c_in, kt, kh, kw), zero columns up to Kp and zero rows up to rows_pad for the GEMM's tiles, zero padding of the
// convolution itself -- as 16-byte stores, and raises the operand's magnitude slot on the way (block maximum, one 64-bit atomic max per
// block that exceeds what the slot holds).
#include "common.h"

namespace {

struct Im2colArgs {
    const float* x;
    float* out;
    unsigned long long* slot;
    int Ci, T, H, W, kt, kh, kw, st, sh, sw, pt, ph, pw, To, Ho, Wo, Kc, Kp;
    long long rows, rows_pad;
};

__global__ __launch_bounds__(256) void im2col3d_kernel(Im2colArgs a) {
    const long long q4 = (long long)(a.Kp >> 2);                 // float4 groups per row
    const long long total = a.rows_pad * q4;
    const int khw = a.kh * a.kw, kvol = a.kt * khw;
    float mx = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / q4;
        const int c0 = (int)(i - row * q4) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (row < a.rows) {
            long long r = row;
            const int wo = (int)(r % a.Wo); r /= a.Wo;
            const int ho = (int)(r % a.Ho); r /= a.Ho;
            const int to = (int)(r % a.To);
            const int n = (int)(r / a.To);
            const float* xn = a.x + (size_t)n * a.Ci * a.T * a.H * a.W;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = c0 + e;
                if (c < a.Kc) {
                    const int ci = c / kvol, rem = c - ci * kvol;
                    const int dt = rem / khw, rem2 = rem - dt * khw;
                    const int dh = rem2 / a.kw, dw = rem2 - dh * a.kw;
                    const int t = to * a.st + dt - a.pt, h = ho * a.sh + dh - a.ph, w = wo * a.sw + dw - a.pw;
                    if (t >= 0 && t < a.T && h >= 0 && h < a.H && w >= 0 && w < a.W)
                        v[e] = xn[(((size_t)ci * a.T + t) * a.H + h) * a.W + w];
                }
            }
        }
        *reinterpret_cast<float4*>(a.out + (size_t)row * a.Kp + c0) = make_float4(v[0], v[1], v[2], v[3]);
        mx = fmaxf(fmaxf(mx, m3t_fin_abs(v[0])), fmaxf(m3t_fin_abs(v[1]), fmaxf(m3t_fin_abs(v[2]), m3t_fin_abs(v[3]))));
    }
    if (a.slot) {
        __shared__ float red[4];
        mx = wave_max(mx);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            if (m > 0.f) atomicMax(a.slot, (unsigned long long)__float_as_uint(m));      // (caller-owned slot: epoch 0; <= 8192 blocks)
        }
    }
}

}  // namespace

extern "C" int m3t_im2col3d(const float* x, int N, int Ci, int T, int H, int W, int kt, int kh, int kw, int st, int sh, int sw,
                            int pt, int ph, int pw, float* out, long long rows_pad, int Kp, unsigned long long* amax_slot, void* stream) {
    if (!x || !out || N <= 0 || Ci <= 0 || kt <= 0 || kh <= 0 || kw <= 0 || st <= 0 || sh <= 0 || sw <= 0 || Kp % 4 != 0 ||
        ((uintptr_t)out % 16) != 0)
        return M3T_EINVAL;
    Im2colArgs a;
    a.x = x; a.out = out; a.slot = amax_slot;
    a.Ci = Ci; a.T = T; a.H = H; a.W = W; a.kt = kt; a.kh = kh; a.kw = kw; a.st = st; a.sh = sh; a.sw = sw; a.pt = pt; a.ph = ph; a.pw = pw;
    a.To = (T + 2 * pt - kt) / st + 1; a.Ho = (H + 2 * ph - kh) / sh + 1; a.Wo = (W + 2 * pw - kw) / sw + 1;
    if (a.To <= 0 || a.Ho <= 0 || a.Wo <= 0) return M3T_EINVAL;
    a.Kc = Ci * kt * kh * kw; a.Kp = Kp;
    a.rows = (long long)N * a.To * a.Ho * a.Wo; a.rows_pad = rows_pad;
    if (Kp < a.Kc || rows_pad < a.rows) return M3T_EINVAL;
    const long long total = rows_pad * (Kp / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    im2col3d_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(a);
    M3T_LAUNCH_CHECK();
    return 0;
}
