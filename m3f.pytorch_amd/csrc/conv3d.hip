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

// A block takes RPB consecutive output positions per pass.  Column table (per block, LDS): offset of (ci, dt, dh, dw) inside a sample and the
// packed (dt, dh, dw); row table (per pass, LDS): offset of the patch origin and the packed origin (t0, h0, w0) -- so that an element costs
// two LDS reads, three range checks and one (cached) load instead of a dozen integer divisions; 16-byte stores, consecutive threads on
// consecutive columns of a row.
constexpr int IM_MAXROWS = 64;
__global__ __launch_bounds__(256) void im2col3d_kernel(Im2colArgs a, int rpb) {
    extern __shared__ int cm[];                                  // [Kp] offsets, [Kp] packed (dt, dh, dw) (or -1: padding column)
    int* coff = cm;
    int* cpos = cm + a.Kp;
    __shared__ long long rbase[IM_MAXROWS];
    __shared__ int rt0[IM_MAXROWS], rh0[IM_MAXROWS], rw0[IM_MAXROWS];
    const int tid = threadIdx.x;
    const int khw = a.kh * a.kw, kvol = a.kt * khw;
    for (int c = tid; c < a.Kp; c += 256) {
        if (c < a.Kc) {
            const int ci = c / kvol, rem = c - ci * kvol;
            const int dt = rem / khw, rem2 = rem - dt * khw;
            const int dh = rem2 / a.kw, dw = rem2 - dh * a.kw;
            coff[c] = ((ci * a.T + dt) * a.H + dh) * a.W + dw;
            cpos[c] = (dt << 20) | (dh << 10) | dw;
        } else { coff[c] = 0; cpos[c] = -1; }
    }
    const int q4 = a.Kp >> 2, items = rpb * q4;
    const long long sample = (long long)a.Ci * a.T * a.H * a.W;
    float mx = 0.f;
    for (long long row0 = (long long)blockIdx.x * rpb; row0 < a.rows_pad; row0 += (long long)gridDim.x * rpb) {
        __syncthreads();                                         // (column table ready; previous pass done with the row table)
        if (tid < rpb) {
            const long long row = row0 + tid;
            if (row < a.rows) {
                const int hw = a.Ho * a.Wo;
                const long long nt = row / hw;
                const int r2 = (int)(row - nt * hw);
                const int ho = r2 / a.Wo, wo = r2 - ho * a.Wo;
                const int n = (int)(nt / a.To), to = (int)(nt - (long long)n * a.To);
                const int t0 = to * a.st - a.pt, h0 = ho * a.sh - a.ph, w0 = wo * a.sw - a.pw;
                rt0[tid] = t0; rh0[tid] = h0; rw0[tid] = w0;
                rbase[tid] = (long long)n * sample + ((long long)t0 * a.H + h0) * a.W + w0;
            } else { rt0[tid] = -(1 << 28); rh0[tid] = 0; rw0[tid] = 0; rbase[tid] = 0; }      // padding row: every range check fails
        }
        __syncthreads();
        for (int it = tid; it < items; it += 256) {
            const int rs = it / q4, c0 = (it - rs * q4) * 4;
            const long long row = row0 + rs;
            if (row >= a.rows_pad) break;
            const int t0 = rt0[rs], h0 = rh0[rs], w0 = rw0[rs];
            const float* xb = a.x + rbase[rs];
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int p = cpos[c0 + e];
                const int t = t0 + (p >> 20), h = h0 + ((p >> 10) & 1023), w = w0 + (p & 1023);
                const bool ok = p >= 0 && (unsigned)t < (unsigned)a.T && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
                v[e] = ok ? xb[coff[c0 + e]] : 0.f;
            }
            *reinterpret_cast<float4*>(a.out + (size_t)row * a.Kp + c0) = make_float4(v[0], v[1], v[2], v[3]);
            mx = fmaxf(fmaxf(mx, m3t_fin_abs(v[0])), fmaxf(m3t_fin_abs(v[1]), fmaxf(m3t_fin_abs(v[2]), m3t_fin_abs(v[3]))));
        }
    }
    if (a.slot) {
        __shared__ float red[4];
        mx = wave_max(mx);
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) {
            const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            const unsigned long long bits = (unsigned long long)__float_as_uint(m);      // (caller-owned slot: epoch 0)
            if (bits > __hip_atomic_load(a.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.slot, bits);
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
    if (kt >= 1024 || kh >= 1024 || kw >= 1024 || (size_t)Kp * 8 > 150 * 1024) return M3T_EINVAL;      // (packed kernel offsets; column tables in LDS: C_in k^3 <= 19 200)
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(im2col3d_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return M3T_EINVAL;
        }
        attr_set = true;
    }
    int rpb = 2048 / (Kp / 4);
    rpb = rpb < 1 ? 1 : (rpb > IM_MAXROWS ? IM_MAXROWS : rpb);
    long long blocks = (rows_pad + rpb - 1) / rpb;
    if (blocks > 256 * 16) blocks = 256 * 16;
    im2col3d_kernel<<<dim3((unsigned)blocks), 256, (size_t)Kp * 8, (hipStream_t)stream>>>(a, rpb);
    M3T_LAUNCH_CHECK();
    return 0;
}
