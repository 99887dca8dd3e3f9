// BatchNorm1d (+ReLU) over channel-last rows, for the `tcn_simple` back-end
// (reference models/backbone.py:107-111, 214-231: Conv1d -> BatchNorm1d(512) -> ReLU).
// x is [M = B*T, C] row-major: a channel is a column, so every wavefront reads 64 consecutive
// channels of one frame row (256 B, coalesced) and walks down the rows.  HBM-bound: the
// forward reads x twice (statistics, apply) and writes y once; the backward reads dy, x, y
// twice and writes dx once.  Column statistics go through fp64 per-chunk partials reduced in
// a fixed order (deterministic, no atomics).
#include "common.h"

namespace {

constexpr int BN_COLS = 64;     // channels per workgroup (one wavefront wide)
constexpr int BN_RPH = 4;       // row phases per workgroup (256 threads)

static int bn_chunks(int M) {
    int c = cdiv(M, 128);
    return c < 1 ? 1 : (c > 128 ? 128 : c);
}

// partial[chunk][2][C]: sum x, sum x^2 over the chunk's rows
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, int M, int C, int rows_per_chunk,
                                                               double* __restrict__ partial) {
    __shared__ double red[2][BN_RPH][BN_COLS];
    const int cl = threadIdx.x & 63, rp = threadIdx.x >> 6;
    const int c = blockIdx.x * BN_COLS + cl;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(M, r0 + rows_per_chunk);
    double s = 0.0, ss = 0.0;
    if (c < C)
        for (int r = r0 + rp; r < r1; r += BN_RPH) {
            const double v = (double)x[(size_t)r * C + c];
            s += v; ss += v * v;
        }
    red[0][rp][cl] = s; red[1][rp][cl] = ss;
    __syncthreads();
    if (rp == 0 && c < C) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int i = 0; i < BN_RPH; ++i) { a += red[0][i][cl]; b += red[1][i][cl]; }
        partial[((size_t)blockIdx.y * 2 + 0) * C + c] = a;
        partial[((size_t)blockIdx.y * 2 + 1) * C + c] = b;
    }
}

// mean / biased variance -> invstd; running statistics updated as torch does (unbiased variance)
__global__ void bn_stats_final_kernel(const double* __restrict__ partial, int nchunks, int M, int C, float eps, float momentum,
                                      float* __restrict__ run_mean, float* __restrict__ run_var, float* __restrict__ mean,
                                      float* __restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < nchunks; ++k) {
        s += partial[((size_t)k * 2 + 0) * C + c];
        ss += partial[((size_t)k * 2 + 1) * C + c];
    }
    const double mu = s / M;
    double var = ss / M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mu;
    if (run_var) {
        const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var, int C, float eps,
                                     float* __restrict__ mean, float* __restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = run_mean[c];
    invstd[c] = 1.0f / sqrtf(run_var[c] + eps);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, float* __restrict__ y, size_t total,
                                                       int C, int relu) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % (size_t)C);
        float v = (x[i] - mean[c]) * invstd[c] * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
        if (relu) v = fmaxf(v, 0.f);
        y[i] = v;
    }
}

// partial[chunk][2][C]: sum g, sum g*xhat with g = dy * (y > 0 if relu)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, int M, int C, int rows_per_chunk,
                                                             int relu, double* __restrict__ partial) {
    __shared__ double red[2][BN_RPH][BN_COLS];
    const int cl = threadIdx.x & 63, rp = threadIdx.x >> 6;
    const int c = blockIdx.x * BN_COLS + cl;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(M, r0 + rows_per_chunk);
    double s = 0.0, sx = 0.0;
    if (c < C) {
        const float mu = mean[c], is = invstd[c];
        for (int r = r0 + rp; r < r1; r += BN_RPH) {
            const size_t o = (size_t)r * C + c;
            float g = dy[o];
            if (relu && !(y[o] > 0.f)) g = 0.f;
            const float xh = (x[o] - mu) * is;
            s += (double)g; sx += (double)g * (double)xh;
        }
    }
    red[0][rp][cl] = s; red[1][rp][cl] = sx;
    __syncthreads();
    if (rp == 0 && c < C) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int i = 0; i < BN_RPH; ++i) { a += red[0][i][cl]; b += red[1][i][cl]; }
        partial[((size_t)blockIdx.y * 2 + 0) * C + c] = a;
        partial[((size_t)blockIdx.y * 2 + 1) * C + c] = b;
    }
}

// sums[0][C] = dbeta, sums[1][C] = dgamma (also written to the caller's buffers)
__global__ void bn_bwd_final_kernel(const double* __restrict__ partial, int nchunks, int C, float* __restrict__ sums,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, sx = 0.0;
    for (int k = 0; k < nchunks; ++k) {
        s += partial[((size_t)k * 2 + 0) * C + c];
        sx += partial[((size_t)k * 2 + 1) * C + c];
    }
    sums[c] = (float)s; sums[C + c] = (float)sx;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)sx;
}

__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ y, const float* __restrict__ gamma,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ sums, float* __restrict__ dx, size_t total,
                                                        int M, int C, int training, int relu) {
    const float invM = 1.0f / (float)M;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % (size_t)C);
        float g = dy[i];
        if (relu && !(y[i] > 0.f)) g = 0.f;
        const float is = invstd[c], w = gamma ? gamma[c] : 1.f;
        float v;
        if (training) {
            const float xh = (x[i] - mean[c]) * is;
            v = w * is * (g - sums[c] * invM - xh * sums[C + c] * invM);
        } else {
            v = g * w * is;
        }
        dx[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------- channel PLANES
// BatchNorm3d (+ReLU) of the 3-D conv stems (reference models/backbone.py:73-103,179-191: Conv3d -> BatchNorm3d -> ReLU), x [N][C][S]
// with S = T H W contiguous (or [N][C][H W]: BatchNorm2d): a channel is N planes of S floats.  A workgroup takes one chunk (<= 8192 floats) of one plane: float4
// sweeps, the plane's scalars (mean, invstd, gamma, beta) loaded once.  Forward: statistics sweep (per-chunk fp64 partials, reduced per
// channel in a fixed order) | apply + ReLU; backward: sums sweep (d beta, d gamma with the ReLU mask from y) | dx.  MIOpen ran the
// normalisation and torch the ReLU as separate passes: one read + one write of the activation less in each direction.
constexpr int PL_CHUNK = 8192;

__device__ __forceinline__ void pl_block_sum2(double& a, double& b) {
    __shared__ double red[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
}

// partial[(plane * nch + chunk) * 2 + {0, 1}]: sum x, sum x^2
template <bool VEC>
__global__ __launch_bounds__(256) void bnp_stats_partial_kernel(const float* __restrict__ x, int S, int nch, double* __restrict__ partial) {
    const size_t plane = blockIdx.x;
    const int s0 = blockIdx.y * PL_CHUNK, s1 = min(S, s0 + PL_CHUNK);
    const float* p = x + plane * (size_t)S;
    float s = 0.f, ss = 0.f;
    if (VEC) {
        for (int i = s0 + 4 * threadIdx.x; i < s1; i += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(p + i);
            s += (v.x + v.y) + (v.z + v.w);
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    } else {
        for (int i = s0 + threadIdx.x; i < s1; i += 256) { const float v = p[i]; s += v; ss += v * v; }
    }
    double a = (double)s, b = (double)ss;
    pl_block_sum2(a, b);
    if (threadIdx.x == 0) {
        partial[(plane * nch + blockIdx.y) * 2 + 0] = a;
        partial[(plane * nch + blockIdx.y) * 2 + 1] = b;
    }
}

// per channel: the N x nch partial pairs in (n, chunk) order.  mode 0: forward statistics (mean, invstd, running statistics);
// mode 1: backward sums (out0 = d beta, out1 = d gamma)
__global__ __launch_bounds__(64) void bnp_final_kernel(const double* __restrict__ partial, int N, int C, int nch, double count, int mode, float eps,
                                                       float momentum, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                       float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta) {
    const int c = blockIdx.x;
    double s = 0.0, ss = 0.0;
    const int per = N * nch;
    for (int k = threadIdx.x; k < per; k += 64) {
        const int n = k / nch, ch = k - n * nch;
        const size_t o = (((size_t)n * C + c) * nch + ch) * 2;
        s += partial[o]; ss += partial[o + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
    if (threadIdx.x != 0) return;
    if (mode == 0) {
        const double mu = s / count;
        double var = ss / count - mu * mu;
        if (var < 0.0) var = 0.0;
        out0[c] = (float)mu;
        out1[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mu;
        if (run_var) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
        }
    } else {
        out0[c] = (float)s; out1[c] = (float)ss;
        if (dbeta) dbeta[c] = (float)s;
        if (dgamma) dgamma[c] = (float)ss;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bnp_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ y,
                                                        int C, int S, int relu, unsigned long long* __restrict__ amax) {
    const size_t plane = blockIdx.x;
    const int c = (int)(plane % (size_t)C);
    const float a = invstd[c] * (gamma ? gamma[c] : 1.f), mu = mean[c], b = beta ? beta[c] : 0.f;
    const int s0 = blockIdx.y * PL_CHUNK, s1 = min(S, s0 + PL_CHUNK);
    const float* p = x + plane * (size_t)S;
    float* q = y + plane * (size_t)S;
    float mx = 0.f;                                  // amax (m3t_amax_out): the slot is raised to max |y| -- the next convolution's operand scale
    auto f = [&](float v) { float r = (v - mu) * a + b; r = relu ? fmaxf(r, 0.f) : r; mx = fmaxf(mx, m3t_fin_abs(r)); return r; };
    if (VEC) {
        for (int i = s0 + 4 * threadIdx.x; i < s1; i += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(p + i);
            *reinterpret_cast<float4*>(q + i) = make_float4(f(v.x), f(v.y), f(v.z), f(v.w));
        }
    } else {
        for (int i = s0 + threadIdx.x; i < s1; i += 256) q[i] = f(p[i]);
    }
    __shared__ float red4[4];
    if (amax) m3t_block_raise_slot(amax, mx, red4);
}

// partial pairs: sum g, sum g * xhat with g = dy * (y > 0 if relu)
template <bool VEC>
__global__ __launch_bounds__(256) void bnp_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd, int C, int S, int nch,
                                                              int relu, double* __restrict__ partial) {
    const size_t plane = blockIdx.x;
    const int c = (int)(plane % (size_t)C);
    const float mu = mean[c], is = invstd[c];
    const int s0 = blockIdx.y * PL_CHUNK, s1 = min(S, s0 + PL_CHUNK);
    const size_t base = plane * (size_t)S;
    float s = 0.f, sx = 0.f;
    auto acc = [&](float g, float xv, float yv) {
        if (relu && !(yv > 0.f)) g = 0.f;
        s += g; sx += g * ((xv - mu) * is);
    };
    if (VEC) {
        for (int i = s0 + 4 * threadIdx.x; i < s1; i += 1024) {
            const float4 g = *reinterpret_cast<const float4*>(dy + base + i), xv = *reinterpret_cast<const float4*>(x + base + i);
            float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (relu) yv = *reinterpret_cast<const float4*>(y + base + i);
            acc(g.x, xv.x, yv.x); acc(g.y, xv.y, yv.y); acc(g.z, xv.z, yv.z); acc(g.w, xv.w, yv.w);
        }
    } else {
        for (int i = s0 + threadIdx.x; i < s1; i += 256) acc(dy[base + i], x[base + i], relu ? y[base + i] : 1.f);
    }
    double a = (double)s, b = (double)sx;
    pl_block_sum2(a, b);
    if (threadIdx.x == 0) {
        partial[(plane * nch + blockIdx.y) * 2 + 0] = a;
        partial[(plane * nch + blockIdx.y) * 2 + 1] = b;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bnp_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ sums, float* __restrict__ dx, int C, int S, float inv_count,
                                                         int training, int relu, unsigned long long* __restrict__ amax) {
    const size_t plane = blockIdx.x;
    const int c = (int)(plane % (size_t)C);
    const float mu = mean[c], is = invstd[c], w = (gamma ? gamma[c] : 1.f) * is;
    const float k1 = training ? sums[c] * inv_count : 0.f, k2 = training ? sums[C + c] * inv_count : 0.f;
    const int s0 = blockIdx.y * PL_CHUNK, s1 = min(S, s0 + PL_CHUNK);
    const size_t base = plane * (size_t)S;
    float mx = 0.f;
    auto f = [&](float g, float xv, float yv) {
        if (relu && !(yv > 0.f)) g = 0.f;
        const float r = training ? w * (g - k1 - ((xv - mu) * is) * k2) : g * w;
        mx = fmaxf(mx, m3t_fin_abs(r));
        return r;
    };
    if (VEC) {
        for (int i = s0 + 4 * threadIdx.x; i < s1; i += 1024) {
            const float4 g = *reinterpret_cast<const float4*>(dy + base + i), xv = *reinterpret_cast<const float4*>(x + base + i);
            float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (relu) yv = *reinterpret_cast<const float4*>(y + base + i);
            *reinterpret_cast<float4*>(dx + base + i) = make_float4(f(g.x, xv.x, yv.x), f(g.y, xv.y, yv.y), f(g.z, xv.z, yv.z), f(g.w, xv.w, yv.w));
        }
    } else {
        for (int i = s0 + threadIdx.x; i < s1; i += 256) dx[base + i] = f(dy[base + i], x[base + i], relu ? y[base + i] : 1.f);
    }
    __shared__ float red4[4];
    if (amax) m3t_block_raise_slot(amax, mx, red4);
}

// ---- small planes (S <= PL_SMALL: BatchNorm2d on the per-frame ResNet's 28 x 28 ... 4 x 4 maps, N C = 32 768 ... 262 144 planes): a
// workgroup per plane would be a workgroup per 64 bytes.  Here LPP = the power of two >= the plane's units (float4 when S % 4 == 0,
// else floats) lanes take one plane, a wave 64 / LPP planes, a workgroup four waves' worth; no divisions, every load a full row.
constexpr int PL_SMALL = 1024;
struct SmallMap { int lpp, lg, ppw; };      // lanes per plane, log2 of it, planes per wave
static SmallMap small_map(int S, bool vec) {
    const int units = vec ? S / 4 : S;
    SmallMap m;
    m.lpp = 1; m.lg = 0;
    while (m.lpp < units && m.lpp < 64) { m.lpp <<= 1; ++m.lg; }
    m.ppw = 64 / m.lpp;
    return m;
}
#define M3T_SMALL_PROLOGUE                                                                                      \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                                 \
    const int sub = lane & (lpp - 1);                                                                           \
    const size_t plane = ((size_t)blockIdx.x * 4 + wave) * (64 >> lg) + (lane >> lg);                           \
    const bool live = plane < P;                                                                                \
    const size_t base = plane * (size_t)S;                                                                      \
    constexpr int E = VEC ? 4 : 1;

__device__ __forceinline__ void seg_sum2(float& a, float& b, int lpp) {
    for (int o = 1; o < lpp; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
}

// MODE 0: sum x, sum x^2;  MODE 1: sum g, sum g xhat (g = dy masked by y > 0 when relu).  partial[plane * 2 + {0, 1}]
template <bool VEC, int MODE>
__global__ __launch_bounds__(256) void bnp_small_reduce_kernel(const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ y,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd, size_t P, int C, int S,
                                                               int lpp, int lg, int relu, double* __restrict__ partial) {
    M3T_SMALL_PROLOGUE
    float s = 0.f, t = 0.f;
    if (live) {
        float mu = 0.f, is = 1.f;
        if (MODE == 1) { const int c = (int)(plane % (size_t)C); mu = mean[c]; is = invstd[c]; }
        for (int i = sub * E; i < S; i += lpp * E) {
            float v[4], xv[4], yv[4];
            if (VEC) {
                const float4 q = *reinterpret_cast<const float4*>(a + base + i);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                if (MODE == 1) {
                    const float4 r = *reinterpret_cast<const float4*>(x + base + i);
                    xv[0] = r.x; xv[1] = r.y; xv[2] = r.z; xv[3] = r.w;
                    if (relu) { const float4 w = *reinterpret_cast<const float4*>(y + base + i); yv[0] = w.x; yv[1] = w.y; yv[2] = w.z; yv[3] = w.w; }
                }
            } else {
                v[0] = a[base + i];
                if (MODE == 1) { xv[0] = x[base + i]; if (relu) yv[0] = y[base + i]; }
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (MODE == 0) { s += v[e]; t += v[e] * v[e]; }
                else {
                    float g = v[e];
                    if (relu && !(yv[e] > 0.f)) g = 0.f;
                    s += g; t += g * ((xv[e] - mu) * is);
                }
            }
        }
    }
    seg_sum2(s, t, lpp);
    if (live && sub == 0) { partial[plane * 2] = (double)s; partial[plane * 2 + 1] = (double)t; }
}

// MODE 0: y = [relu]((x - mean) invstd gamma + beta);  MODE 1: dx (training: gamma invstd (g - k1 - xhat k2), eval: g gamma invstd)
template <bool VEC, int MODE>
__global__ __launch_bounds__(256) void bnp_small_map_kernel(const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ sums, float* __restrict__ out,
                                                            size_t P, int C, int S, int lpp, int lg, float inv_count, int training, int relu,
                                                            unsigned long long* __restrict__ amax) {
    M3T_SMALL_PROLOGUE
    const int c = live ? (int)(plane % (size_t)C) : 0;
    const float mu = mean[c], is = invstd[c], w = (gamma ? gamma[c] : 1.f) * is, b = (MODE == 0 && beta) ? beta[c] : 0.f;
    const float k1 = (MODE == 1 && training) ? sums[c] * inv_count : 0.f, k2 = (MODE == 1 && training) ? sums[C + c] * inv_count : 0.f;
    float mx = 0.f;                                  // amax (m3t_amax_out): max |out| of the launch into the slot -- every thread reaches the end
    for (int i = sub * E; live && i < S; i += lpp * E) {
        float v[4], xv[4], yv[4], o[4];
        if (VEC) {
            const float4 q = *reinterpret_cast<const float4*>(a + base + i);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            if (MODE == 1) {
                const float4 r = *reinterpret_cast<const float4*>(x + base + i);
                xv[0] = r.x; xv[1] = r.y; xv[2] = r.z; xv[3] = r.w;
                if (relu) { const float4 u = *reinterpret_cast<const float4*>(y + base + i); yv[0] = u.x; yv[1] = u.y; yv[2] = u.z; yv[3] = u.w; }
            }
        } else {
            v[0] = a[base + i];
            if (MODE == 1) { xv[0] = x[base + i]; if (relu) yv[0] = y[base + i]; }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (MODE == 0) {
                const float r = (v[e] - mu) * w + b;
                o[e] = relu ? fmaxf(r, 0.f) : r;
            } else {
                float g = v[e];
                if (relu && !(yv[e] > 0.f)) g = 0.f;
                o[e] = training ? w * (g - k1 - ((xv[e] - mu) * is) * k2) : g * w;
            }
            mx = fmaxf(mx, m3t_fin_abs(o[e]));
        }
        if (VEC) *reinterpret_cast<float4*>(out + base + i) = make_float4(o[0], o[1], o[2], o[3]);
        else out[base + i] = o[0];
    }
    __shared__ float red4[4];
    if (amax) m3t_block_raise_slot(amax, mx, red4);
}

static int pl_chunks(int S) { return S <= PL_SMALL ? 1 : cdiv(S, PL_CHUNK); }

}  // namespace

extern "C" size_t m3t_bn_planes_ws_bytes(int N, int C, int S) {
    return (size_t)N * C * pl_chunks(S) * 2 * sizeof(double) + 2 * (size_t)C * sizeof(float) + 256;
}

// x, y [N][C][S] (S = T H W contiguous).  training: batch statistics over N * S values per channel, running statistics updated with
// `momentum` (torch's unbiased variance); else the running ones.  save_mean / save_invstd [C] for the backward pass.  relu: fused ReLU.
extern "C" int m3t_bn_planes_fwd(const float* x, int N, int C, int S, const float* gamma, const float* beta, float* run_mean, float* run_var,
                                 float momentum, float eps, int training, int relu, float* y, float* save_mean, float* save_invstd, float* ws,
                                 size_t ws_bytes, void* stream) {
    unsigned long long* amax = m3t_take_amax_out();      // (m3t_amax_out: raised to max |y| by the apply kernel)
    if (N <= 0 || C <= 0 || S <= 0 || !x || !y || !save_mean || !save_invstd) return M3T_EINVAL;
    if (!training && (!run_mean || !run_var)) return M3T_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nch = pl_chunks(S);
    const bool vec = S % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0;
    const dim3 grid((unsigned)((size_t)N * C), nch);
    const bool small = S <= PL_SMALL;
    const size_t P = (size_t)N * C;
    const SmallMap sm = small_map(S, vec);
    const unsigned sgrid = (unsigned)((P + (size_t)4 * sm.ppw - 1) / ((size_t)4 * sm.ppw));
    if (training) {
        if ((size_t)N * S < 2) return M3T_EINVAL;      // torch: "Expected more than 1 value per channel when training"
        if (!ws || ws_bytes < m3t_bn_planes_ws_bytes(N, C, S) || ((uintptr_t)ws % 8) != 0) return M3T_EINVAL;
        double* partial = reinterpret_cast<double*>(ws);
        if (small) {
            if (vec) bnp_small_reduce_kernel<true, 0><<<sgrid, 256, 0, s>>>(x, nullptr, nullptr, nullptr, nullptr, P, C, S, sm.lpp, sm.lg, 0, partial);
            else bnp_small_reduce_kernel<false, 0><<<sgrid, 256, 0, s>>>(x, nullptr, nullptr, nullptr, nullptr, P, C, S, sm.lpp, sm.lg, 0, partial);
        }
        else if (vec) bnp_stats_partial_kernel<true><<<grid, 256, 0, s>>>(x, S, nch, partial);
        else bnp_stats_partial_kernel<false><<<grid, 256, 0, s>>>(x, S, nch, partial);
        M3T_LAUNCH_CHECK();
        bnp_final_kernel<<<C, 64, 0, s>>>(partial, N, C, nch, (double)N * S, 0, eps, momentum, run_mean, run_var, save_mean, save_invstd, nullptr, nullptr);
        M3T_LAUNCH_CHECK();
    } else {
        bn_eval_stats_kernel<<<cdiv(C, 256), 256, 0, s>>>(run_mean, run_var, C, eps, save_mean, save_invstd);
        M3T_LAUNCH_CHECK();
    }
    if (small) {
        if (vec) bnp_small_map_kernel<true, 0><<<sgrid, 256, 0, s>>>(x, nullptr, nullptr, gamma, beta, save_mean, save_invstd, nullptr, y, P, C, S, sm.lpp, sm.lg, 0.f, training, relu, amax);
        else bnp_small_map_kernel<false, 0><<<sgrid, 256, 0, s>>>(x, nullptr, nullptr, gamma, beta, save_mean, save_invstd, nullptr, y, P, C, S, sm.lpp, sm.lg, 0.f, training, relu, amax);
    }
    else if (vec) bnp_apply_kernel<true><<<grid, 256, 0, s>>>(x, gamma, beta, save_mean, save_invstd, y, C, S, relu, amax);
    else bnp_apply_kernel<false><<<grid, 256, 0, s>>>(x, gamma, beta, save_mean, save_invstd, y, C, S, relu, amax);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_bn_planes_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* save_mean, const float* save_invstd,
                                 int N, int C, int S, int training, int relu, float* dx, float* dgamma, float* dbeta, float* ws, size_t ws_bytes,
                                 void* stream) {
    unsigned long long* amax = m3t_take_amax_out();      // (m3t_amax_out: raised to max |dx| by the dx kernel -- the operand scale of the convolution in front)
    if (N <= 0 || C <= 0 || S <= 0 || !dy || !x || !dx || !save_mean || !save_invstd) return M3T_EINVAL;
    if (relu && !y) return M3T_EINVAL;
    if (!ws || ws_bytes < m3t_bn_planes_ws_bytes(N, C, S) || ((uintptr_t)ws % 8) != 0) return M3T_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nch = pl_chunks(S);
    const bool vec = S % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dx % 16) == 0 && (!relu || ((uintptr_t)y % 16) == 0);
    const dim3 grid((unsigned)((size_t)N * C), nch);
    const bool small = S <= PL_SMALL;
    const size_t P = (size_t)N * C;
    const SmallMap sm = small_map(S, vec);
    const unsigned sgrid = (unsigned)((P + (size_t)4 * sm.ppw - 1) / ((size_t)4 * sm.ppw));
    double* partial = reinterpret_cast<double*>(ws);
    float* sums = reinterpret_cast<float*>(partial + (size_t)N * C * nch * 2);
    if (small) {
        if (vec) bnp_small_reduce_kernel<true, 1><<<sgrid, 256, 0, s>>>(dy, x, y, save_mean, save_invstd, P, C, S, sm.lpp, sm.lg, relu, partial);
        else bnp_small_reduce_kernel<false, 1><<<sgrid, 256, 0, s>>>(dy, x, y, save_mean, save_invstd, P, C, S, sm.lpp, sm.lg, relu, partial);
    }
    else if (vec) bnp_bwd_partial_kernel<true><<<grid, 256, 0, s>>>(dy, x, y, save_mean, save_invstd, C, S, nch, relu, partial);
    else bnp_bwd_partial_kernel<false><<<grid, 256, 0, s>>>(dy, x, y, save_mean, save_invstd, C, S, nch, relu, partial);
    M3T_LAUNCH_CHECK();
    bnp_final_kernel<<<C, 64, 0, s>>>(partial, N, C, nch, (double)N * S, 1, 0.f, 0.f, nullptr, nullptr, sums, sums + C, dgamma, dbeta);
    M3T_LAUNCH_CHECK();
    const float inv_count = (float)(1.0 / ((double)N * S));
    if (small) {
        if (vec) bnp_small_map_kernel<true, 1><<<sgrid, 256, 0, s>>>(dy, x, y, gamma, nullptr, save_mean, save_invstd, sums, dx, P, C, S, sm.lpp, sm.lg, inv_count, training, relu, amax);
        else bnp_small_map_kernel<false, 1><<<sgrid, 256, 0, s>>>(dy, x, y, gamma, nullptr, save_mean, save_invstd, sums, dx, P, C, S, sm.lpp, sm.lg, inv_count, training, relu, amax);
    }
    else if (vec) bnp_bwd_dx_kernel<true><<<grid, 256, 0, s>>>(dy, x, y, gamma, save_mean, save_invstd, sums, dx, C, S, inv_count, training, relu, amax);
    else bnp_bwd_dx_kernel<false><<<grid, 256, 0, s>>>(dy, x, y, gamma, save_mean, save_invstd, sums, dx, C, S, inv_count, training, relu, amax);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t m3t_bn_rows_ws_bytes(int M, int C) {
    return (size_t)bn_chunks(M) * 2 * (size_t)C * sizeof(double) + 2 * (size_t)C * sizeof(float) + 256;
}

extern "C" int m3t_bn_rows_fwd(const float* x, int M, int C, const float* gamma, const float* beta, float* run_mean,
                               float* run_var, float momentum, float eps, int training, int relu, float* y, float* save_mean,
                               float* save_invstd, float* ws, size_t ws_bytes, void* stream) {
    if (C <= 0 || M < 0 || !x || !y || !save_mean || !save_invstd) return M3T_EINVAL;
    if (!training && (!run_mean || !run_var)) return M3T_EINVAL;
    if (M == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (training) {
        if (M < 2) return M3T_EINVAL;      // torch: "Expected more than 1 value per channel when training"
        if (!ws || ws_bytes < m3t_bn_rows_ws_bytes(M, C) || ((uintptr_t)ws % 8) != 0) return M3T_EINVAL;
        const int nch = bn_chunks(M), rpc = cdiv(M, nch);
        double* partial = reinterpret_cast<double*>(ws);
        bn_stats_partial_kernel<<<dim3(cdiv(C, BN_COLS), nch), 256, 0, s>>>(x, M, C, rpc, partial);
        M3T_LAUNCH_CHECK();
        bn_stats_final_kernel<<<cdiv(C, 256), 256, 0, s>>>(partial, nch, M, C, eps, momentum, run_mean, run_var, save_mean,
                                                          save_invstd);
        M3T_LAUNCH_CHECK();
    } else {
        bn_eval_stats_kernel<<<cdiv(C, 256), 256, 0, s>>>(run_mean, run_var, C, eps, save_mean, save_invstd);
        M3T_LAUNCH_CHECK();
    }
    const size_t total = (size_t)M * C;
    int blocks = (int)((total + 1023) / 1024);
    if (blocks > 4096) blocks = 4096;
    bn_apply_kernel<<<blocks, 256, 0, s>>>(x, gamma, beta, save_mean, save_invstd, y, total, C, relu);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_bn_rows_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* save_mean,
                               const float* save_invstd, int M, int C, int training, int relu, float* dx, float* dgamma,
                               float* dbeta, float* ws, size_t ws_bytes, void* stream) {
    if (C <= 0 || M < 0 || !dy || !x || !dx || !save_mean || !save_invstd) return M3T_EINVAL;
    if (relu && !y) return M3T_EINVAL;
    if (M == 0) return 0;
    if (!ws || ws_bytes < m3t_bn_rows_ws_bytes(M, C) || ((uintptr_t)ws % 8) != 0) return M3T_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nch = bn_chunks(M), rpc = cdiv(M, nch);
    double* partial = reinterpret_cast<double*>(ws);
    float* sums = reinterpret_cast<float*>(partial + (size_t)nch * 2 * C);
    bn_bwd_partial_kernel<<<dim3(cdiv(C, BN_COLS), nch), 256, 0, s>>>(dy, x, y, save_mean, save_invstd, M, C, rpc, relu, partial);
    M3T_LAUNCH_CHECK();
    bn_bwd_final_kernel<<<cdiv(C, 256), 256, 0, s>>>(partial, nch, C, sums, dgamma, dbeta);
    M3T_LAUNCH_CHECK();
    const size_t total = (size_t)M * C;
    int blocks = (int)((total + 1023) / 1024);
    if (blocks > 4096) blocks = 4096;
    bn_bwd_dx_kernel<<<blocks, 256, 0, s>>>(dy, x, y, gamma, save_mean, save_invstd, sums, dx, total, M, C, training, relu);
    M3T_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------- spatial max pooling
// nn.MaxPool3d with a (1, k, k) window (every pooling layer of the 3-D stems, reference models/backbone.py:80,86,92,182) = a 2-D pooling
// of P = N C T planes [H][W].  Forward keeps the winner's position INSIDE its window as one byte (dh * kw + dw: torch keeps an int64
// flat index per output); backward is a gather -- every input position asks the <= ceil(k/s)^2 windows that cover it whether it won --
// so overlapping windows (3 x 3, stride 2) need no atomics and the result does not depend on scheduling.  Ties and NaN as torch's
// kernel: the first maximum in row-major window order, NaN wins.  One workgroup per plane.
namespace {

struct PoolGeo { int H, W, Ho, Wo, kh, kw, sh, sw, ph, pw; };

// K > 0: the window / stride / padding are the compile-time squares (K, S_, P_) -- the ResNet stem's 3 x 3 / 2 / 1 and the VGG stems'
// 2 x 2 / 2 / 0: divisions by the stride become shifts and the window loops unroll; K = 0: the geometry in `gr` as given
template <int K, int S_, int P_>
__device__ __forceinline__ PoolGeo pool_geo(const PoolGeo& gr) {
    if (K == 0) return gr;
    PoolGeo g = gr;
    g.kh = g.kw = K; g.sh = g.sw = S_; g.ph = g.pw = P_;
    return g;
}
template <int K, int S_, int P_>
__global__ __launch_bounds__(256) void pool_planes_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ win, PoolGeo gr) {
    const PoolGeo g = pool_geo<K, S_, P_>(gr);
    const size_t plane = blockIdx.x;
    const float* xp = x + plane * (size_t)g.H * g.W;
    const size_t ob = plane * (size_t)g.Ho * g.Wo;
    for (int o = threadIdx.x; o < g.Ho * g.Wo; o += 256) {
        const int ho = o / g.Wo, wo = o - ho * g.Wo;
        const int h0 = ho * g.sh - g.ph, w0 = wo * g.sw - g.pw;
        // torch's rule (max_pool_forward_nchw): start at -inf with the window's first valid element as winner; an element takes over
        // if it is greater or NaN
        float best = -INFINITY;
        int bi = max(0, -h0) * g.kw + max(0, -w0);
        if (K > 0) {
#pragma unroll
            for (int dh = 0; dh < K; ++dh)
#pragma unroll
                for (int dw = 0; dw < K; ++dw) {
                    const int h = h0 + dh, w = w0 + dw;
                    if (h >= 0 && h < g.H && w >= 0 && w < g.W) {
                        const float v = xp[h * g.W + w];
                        if (v > best || v != v) { best = v; bi = dh * K + dw; }
                    }
                }
        } else
        for (int dh = max(0, -h0); dh < g.kh && h0 + dh < g.H; ++dh)
            for (int dw = max(0, -w0); dw < g.kw && w0 + dw < g.W; ++dw) {
                const float v = xp[(h0 + dh) * g.W + w0 + dw];
                if (v > best || v != v) { best = v; bi = dh * g.kw + dw; }
            }
        y[ob + o] = best;
        win[ob + o] = (unsigned char)bi;
    }
}

template <int K, int S_, int P_>
__global__ __launch_bounds__(256) void pool_planes_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ win, float* __restrict__ dx,
                                                              PoolGeo gr) {
    const PoolGeo g = pool_geo<K, S_, P_>(gr);
    const size_t plane = blockIdx.x;
    const size_t ob = plane * (size_t)g.Ho * g.Wo;
    float* dp = dx + plane * (size_t)g.H * g.W;
    for (int i = threadIdx.x; i < g.H * g.W; i += 256) {
        const int h = i / g.W, w = i - h * g.W;
        // windows ho with ho * sh - ph <= h <= ho * sh - ph + kh - 1
        const int hp = h + g.ph, wp = w + g.pw;
        const int ho_hi = min(g.Ho - 1, hp / g.sh), wo_hi = min(g.Wo - 1, wp / g.sw);
        const int ho_lo = max(0, (hp - g.kh + g.sh) / g.sh), wo_lo = max(0, (wp - g.kw + g.sw) / g.sw);      // ceil((hp - kh + 1) / sh), hp - kh + 1 may be < 0
        float acc = 0.f;
        for (int ho = (hp - g.kh + 1 <= 0 ? 0 : ho_lo); ho <= ho_hi; ++ho)
            for (int wo = (wp - g.kw + 1 <= 0 ? 0 : wo_lo); wo <= wo_hi; ++wo) {
                const int b = win[ob + ho * g.Wo + wo];
                const int dh = b / g.kw, dw = b - dh * g.kw;
                if (ho * g.sh - g.ph + dh == h && wo * g.sw - g.pw + dw == w) acc += dy[ob + ho * g.Wo + wo];
            }
        dp[i] = acc;
    }
}

// overlapping windows (the ResNet stem's 3 x 3 / 2 / 1 on 56 x 56 planes): every input cell is wanted by up to four windows, in columns two
// apart -- read straight from memory the forward ran at 1.4 TB/s and the gather backward at 0.9.  Here the plane (forward) or the
// plane's dy + winner bytes (backward) go through LDS first: one coalesced sweep of global memory each way.  Planes up to 48 KB.
template <int K, int S_, int P_>
__global__ __launch_bounds__(256) void pool_planes_fwd_lds_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ win,
                                                                  PoolGeo gr) {
    extern __shared__ __attribute__((aligned(16))) float sp[];
    const PoolGeo g = pool_geo<K, S_, P_>(gr);
    const size_t plane = blockIdx.x;
    const int HW = g.H * g.W;
    const float* xp = x + plane * (size_t)HW;
    if ((HW & 3) == 0 && ((uintptr_t)x & 15) == 0) {
        for (int i = threadIdx.x; i < (HW >> 2); i += 256) reinterpret_cast<float4*>(sp)[i] = reinterpret_cast<const float4*>(xp)[i];
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) sp[i] = xp[i];
    }
    __syncthreads();
    const size_t ob = plane * (size_t)g.Ho * g.Wo;
    for (int o = threadIdx.x; o < g.Ho * g.Wo; o += 256) {
        const int ho = o / g.Wo, wo = o - ho * g.Wo;
        const int h0 = ho * g.sh - g.ph, w0 = wo * g.sw - g.pw;
        float best = -INFINITY;
        int bi = max(0, -h0) * g.kw + max(0, -w0);
        for (int dh = max(0, -h0); dh < g.kh && h0 + dh < g.H; ++dh)
            for (int dw = max(0, -w0); dw < g.kw && w0 + dw < g.W; ++dw) {
                const float v = sp[(h0 + dh) * g.W + w0 + dw];
                if (v > best || v != v) { best = v; bi = dh * g.kw + dw; }
            }
        y[ob + o] = best;
        win[ob + o] = (unsigned char)bi;
    }
}

template <int K, int S_, int P_>
__global__ __launch_bounds__(256) void pool_planes_bwd_lds_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ win,
                                                                  float* __restrict__ dx, PoolGeo gr) {
    extern __shared__ __attribute__((aligned(16))) float sp[];
    const PoolGeo g = pool_geo<K, S_, P_>(gr);
    const size_t plane = blockIdx.x;
    const int no = g.Ho * g.Wo;
    unsigned char* sw_ = reinterpret_cast<unsigned char*>(sp + ((no + 3) & ~3));
    const size_t ob = plane * (size_t)no;
    for (int i = threadIdx.x; i < no; i += 256) { sp[i] = dy[ob + i]; sw_[i] = win[ob + i]; }
    __syncthreads();
    float* dp = dx + plane * (size_t)g.H * g.W;
    for (int i = threadIdx.x; i < g.H * g.W; i += 256) {
        const int h = i / g.W, w = i - h * g.W;
        const int hp = h + g.ph, wp = w + g.pw;
        const int ho_hi = min(g.Ho - 1, hp / g.sh), wo_hi = min(g.Wo - 1, wp / g.sw);
        const int ho_lo = hp - g.kh + 1 <= 0 ? 0 : (hp - g.kh + g.sh) / g.sh, wo_lo = wp - g.kw + 1 <= 0 ? 0 : (wp - g.kw + g.sw) / g.sw;
        float acc = 0.f;
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const int b = sw_[ho * g.Wo + wo];
                const int dh = b / g.kw, dw = b - dh * g.kw;
                if (ho * g.sh - g.ph + dh == h && wo * g.sw - g.pw + dw == w) acc += sp[ho * g.Wo + wo];
            }
        dp[i] = acc;
    }
}

// windows that do not overlap (k <= s, no padding: the VGG stem's 2 x 2 / 2 poolings): a thread per OUTPUT writes its window's cells
// (dy to the winner, zero to the rest) -- one division per window instead of three per cell -- and the cells no window covers (the
// last row / column of an odd map) are zeroed separately
__global__ __launch_bounds__(256) void pool_planes_bwd_disjoint_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ win,
                                                                       float* __restrict__ dx, PoolGeo g) {
    const size_t plane = blockIdx.x;
    const size_t ob = plane * (size_t)g.Ho * g.Wo;
    float* dp = dx + plane * (size_t)g.H * g.W;
    for (int o = threadIdx.x; o < g.Ho * g.Wo; o += 256) {
        const int ho = o / g.Wo, wo = o - ho * g.Wo;
        const float gv = dy[ob + o];
        const int b = win[ob + o];
        float* cell = dp + (ho * g.sh) * g.W + wo * g.sw;
        int idx = 0;
        for (int dh = 0; dh < g.sh; ++dh)
            for (int dw = 0; dw < g.sw; ++dw, ++idx) {
                // cells of the stride box beyond the window (k < s) belong to no window: zero
                const bool inwin = dh < g.kh && dw < g.kw;
                if ((ho * g.sh + dh) < g.H && (wo * g.sw + dw) < g.W) cell[dh * g.W + dw] = (inwin && b == dh * g.kw + dw) ? gv : 0.f;
            }
    }
    // covered by stride boxes: [0, hc) x [0, wc), clipped to the plane (k < s: the last box may reach past the map -- ADVICE r4: the tail
    // loops below then wrote rows h >= H, i.e. the next plane)
    const int hc = min(g.Ho * g.sh, g.H), wc = min(g.Wo * g.sw, g.W);
    for (int i = threadIdx.x; i < (g.H - hc) * g.W; i += 256) dp[hc * g.W + i] = 0.f;
    if (wc < g.W)
        for (int i = threadIdx.x; i < hc * (g.W - wc); i += 256) {
            const int h = i / (g.W - wc), w = wc + i - h * (g.W - wc);
            dp[h * g.W + w] = 0.f;
        }
}

}  // namespace

extern "C" int m3t_pool_planes_fwd(const float* x, long long P, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw, float* y,
                                   unsigned char* win, void* stream) {
    if (P <= 0) return 0;
    if (!x || !y || !win || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0 || kh * kw > 255 || 2 * ph > kh || 2 * pw > kw ||
        P > 0x7fffffffll)
        return M3T_EINVAL;
    PoolGeo g{H, W, (H + 2 * ph - kh) / sh + 1, (W + 2 * pw - kw) / sw + 1, kh, kw, sh, sw, ph, pw};
    if (g.Ho <= 0 || g.Wo <= 0) return M3T_EINVAL;
    const bool sq = kh == kw && sh == sw && ph == pw;
    const bool overlap = kh > sh || kw > sw;
    const size_t plane_b = (size_t)H * W * sizeof(float);
    if (overlap && plane_b <= 48 * 1024) {
        if (sq && kh == 3 && sh == 2 && ph == 1) pool_planes_fwd_lds_kernel<3, 2, 1><<<(unsigned)P, 256, plane_b, (hipStream_t)stream>>>(x, y, win, g);
        else pool_planes_fwd_lds_kernel<0, 0, 0><<<(unsigned)P, 256, plane_b, (hipStream_t)stream>>>(x, y, win, g);
    }
    else if (sq && kh == 3 && sh == 2 && ph == 1) pool_planes_fwd_kernel<3, 2, 1><<<(unsigned)P, 256, 0, (hipStream_t)stream>>>(x, y, win, g);
    else if (sq && kh == 2 && sh == 2 && ph == 0) pool_planes_fwd_kernel<2, 2, 0><<<(unsigned)P, 256, 0, (hipStream_t)stream>>>(x, y, win, g);
    else pool_planes_fwd_kernel<0, 0, 0><<<(unsigned)P, 256, 0, (hipStream_t)stream>>>(x, y, win, g);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_pool_planes_bwd(const float* dy, const unsigned char* win, long long P, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw,
                                   float* dx, void* stream) {
    if (P <= 0) return 0;
    if (!dy || !win || !dx || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0 || kh * kw > 255 || P > 0x7fffffffll) return M3T_EINVAL;
    PoolGeo g{H, W, (H + 2 * ph - kh) / sh + 1, (W + 2 * pw - kw) / sw + 1, kh, kw, sh, sw, ph, pw};
    if (g.Ho <= 0 || g.Wo <= 0) return M3T_EINVAL;
    if (kh <= sh && kw <= sw && ph == 0 && pw == 0) pool_planes_bwd_disjoint_kernel<<<(unsigned)P, 256, 0, (hipStream_t)stream>>>(dy, win, dx, g);
    else if ((size_t)g.Ho * g.Wo * 5 + 16 <= 48 * 1024) {
        const size_t lds = (size_t)((g.Ho * g.Wo + 3) & ~3) * sizeof(float) + (size_t)g.Ho * g.Wo;
        if (kh == kw && sh == sw && ph == pw && kh == 3 && sh == 2 && ph == 1) pool_planes_bwd_lds_kernel<3, 2, 1><<<(unsigned)P, 256, lds, (hipStream_t)stream>>>(dy, win, dx, g);
        else pool_planes_bwd_lds_kernel<0, 0, 0><<<(unsigned)P, 256, lds, (hipStream_t)stream>>>(dy, win, dx, g);
    }
    else if (kh == kw && sh == sw && ph == pw && kh == 3 && sh == 2 && ph == 1) pool_planes_bwd_kernel<3, 2, 1><<<(unsigned)P, 256, 0, (hipStream_t)stream>>>(dy, win, dx, g);
    else pool_planes_bwd_kernel<0, 0, 0><<<(unsigned)P, 256, 0, (hipStream_t)stream>>>(dy, win, dx, g);
    M3T_LAUNCH_CHECK();
    return 0;
}
