// Channels-last operators of the 3-D VGG-M stems (round 6).  Reference: models/backbone.py:73-103,179-271 -- Conv3d -> BatchNorm3d -> ReLU
// (-> MaxPool3d((1, 2, 2))) five times over.  Until round 5 BatchNorm / pooling worked on channel PLANES [N][C][T H W] and every convolution
// was bracketed by tiled transposes (x -> channels-last for the tap walk, dy -> channels-last for the gradients: 25 launches and ~0.85 ms per
// C5 step).  The tap walks read AND write channels-last rows natively; with these operators the whole stem stays
// [rows = N T H W][C]: no transpose between the video and the GRU input.
//   * BatchNorm (+ReLU) over rows at a million rows (csrc/bn.hip's rows kernels serve tcn_simple's 9 600 x 512): float4 sweeps, a thread's four
//     channels fixed for the whole sweep, fp64 per-chunk partials reduced in a fixed order (deterministic), apply / dx raise the
//     magnitude slot of what they write (m3t_amax_out: the next tap walk scales by it -- no measuring pass)
//   * max pooling of frames [P][H][W][C] with a (kh, kw) window: a thread = one output position x four channels; the winner's place inside its
//     window is kept as a byte per element, backward is a gather over the windows that cover an input position (no atomics), as the plane
//     kernels; ties / NaN as torch (first maximum in window order, NaN wins)
#include "common.h"

namespace {

constexpr int CL_TH = 256;

// ---- statistics: partial[chunk][2][C] doubles.  MODE 0: sum x, sum x^2;  MODE 1 (backward): sum g, sum g xhat with g = dy (y > 0 if relu)
template <int MODE>
__global__ __launch_bounds__(CL_TH) void bncl_partial_kernel(const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd, size_t M, int C,
                                                             size_t rows_per_chunk, int relu, double* __restrict__ partial) {
    extern __shared__ double red[];                         // [2][groups][C]
    const int c4n = C >> 2, groups = CL_TH / c4n;            // row groups that sweep the chunk side by side (C / 4 divides 256)
    const int q = threadIdx.x % c4n, g = threadIdx.x / c4n;
    const size_t r0 = (size_t)blockIdx.x * rows_per_chunk;
    const size_t r1 = r0 + rows_per_chunk < M ? r0 + rows_per_chunk : M;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, t[4] = {0.0, 0.0, 0.0, 0.0};
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { mu[e] = mean[4 * q + e]; is[e] = invstd[4 * q + e]; }
    }
    if (g < groups) {
        // four rows in flight per thread (the loads of all four issued before the first is used); their sum in fp32, the running sums in fp64
        for (size_t rb = r0 + g; rb < r1; rb += (size_t)4 * groups) {
            float4 av[4], xv4[4], yv4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t r = rb + (size_t)u * groups;
                const bool in = r < r1;
                const size_t o = (in ? r : rb) * C + 4 * q;
                av[u] = *reinterpret_cast<const float4*>(a + o);
                if (!in) av[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MODE == 1) {
                    xv4[u] = *reinterpret_cast<const float4*>(x + o);
                    if (relu) yv4[u] = *reinterpret_cast<const float4*>(y + o);
                }
            }
            float fs[4] = {0.f, 0.f, 0.f, 0.f}, ft[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float ve[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
                if (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { fs[e] += ve[e]; ft[e] = fmaf(ve[e], ve[e], ft[e]); }
                } else {
                    const float xe[4] = {xv4[u].x, xv4[u].y, xv4[u].z, xv4[u].w};
                    if (relu) {
                        const float ye[4] = {yv4[u].x, yv4[u].y, yv4[u].z, yv4[u].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (!(ye[e] > 0.f)) ve[e] = 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { fs[e] += ve[e]; ft[e] = fmaf(ve[e], (xe[e] - mu[e]) * is[e], ft[e]); }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += (double)fs[e]; t[e] += (double)ft[e]; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[(size_t)g * C + 4 * q + e] = s[e]; red[(size_t)(groups + g) * C + 4 * q + e] = t[e]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += CL_TH) {
        double a0 = 0.0, b0 = 0.0;
        for (int k = 0; k < groups; ++k) { a0 += red[(size_t)k * C + c]; b0 += red[(size_t)(groups + k) * C + c]; }      // fixed order
        partial[((size_t)blockIdx.x * 2 + 0) * C + c] = a0;
        partial[((size_t)blockIdx.x * 2 + 1) * C + c] = b0;
    }
}

// mean / biased variance -> invstd; running statistics as torch (unbiased variance)
// (both final kernels: 64 channels per block, the chunks in four interleaved phases -- thread (channel, phase) adds every fourth chunk, the four
// phase sums are combined in phase order: a fixed order, four times shorter a dependent chain than one thread per channel)
__device__ __forceinline__ void bncl_sum_chunks(const double* __restrict__ partial, int nchunks, int C, int c, double& s, double& t,
                                                double (&red)[2][4][64]) {
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    double a = 0.0, b = 0.0;
    if (c < C) {
        double a4[4] = {0.0, 0.0, 0.0, 0.0}, b4[4] = {0.0, 0.0, 0.0, 0.0};      // four loads in flight per statistic; combined in index order
        int k = ph;
        for (; k + 12 < nchunks; k += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a4[u] += partial[((size_t)(k + 4 * u) * 2 + 0) * C + c];
                b4[u] += partial[((size_t)(k + 4 * u) * 2 + 1) * C + c];
            }
        }
        for (; k < nchunks; k += 4) { a4[0] += partial[((size_t)k * 2 + 0) * C + c]; b4[0] += partial[((size_t)k * 2 + 1) * C + c]; }
        a = (a4[0] + a4[1]) + (a4[2] + a4[3]); b = (b4[0] + b4[1]) + (b4[2] + b4[3]);
    }
    red[0][ph][cl] = a; red[1][ph][cl] = b;
    __syncthreads();
    s = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    t = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
}

__global__ __launch_bounds__(256) void bncl_stats_final_kernel(const double* __restrict__ partial, int nchunks, double M, int C, float eps,
                                                               float momentum, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                               float* __restrict__ mean, float* __restrict__ invstd) {
    __shared__ double red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double s, ss;
    bncl_sum_chunks(partial, nchunks, C, c, s, ss, red);
    if (c >= C || threadIdx.x >= 64) return;
    const double mu = s / M;
    double var = ss / M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mu;
    if (run_var) {
        const double unb = M > 1.0 ? var * M / (M - 1.0) : var;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
}
__global__ void bncl_eval_stats_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var, int C, float eps,
                                       float* __restrict__ mean, float* __restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = run_mean[c];
    invstd[c] = 1.0f / sqrtf(run_var[c] + eps);
}
__global__ __launch_bounds__(256) void bncl_bwd_final_kernel(const double* __restrict__ partial, int nchunks, int C, float* __restrict__ sums,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double s, sx;
    bncl_sum_chunks(partial, nchunks, C, c, s, sx, red);
    if (c >= C || threadIdx.x >= 64) return;
    sums[c] = (float)s; sums[C + c] = (float)sx;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)sx;
}

// apply (+ReLU) / dx.  total4 = M C / 4; the grid's stride is a multiple of C / 4: a thread's four channels never change
template <int MODE>
__global__ __launch_bounds__(CL_TH) void bncl_map_kernel(const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ sums, float* __restrict__ out, size_t total4, int C,
                                                         float inv_count, int training, int relu, unsigned long long* __restrict__ slot,
                                                         float* __restrict__ colpart) {
    __shared__ float red4[4];
    __shared__ float csum[CL_TH * 4];
    const int c4n = C >> 2;
    const size_t i0 = (size_t)blockIdx.x * CL_TH + threadIdx.x;
    const int q = (int)(i0 % (size_t)c4n);
    float w[4], b[4], mu[4], is[4], k1[4], k2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * q + e;
        mu[e] = mean[c]; is[e] = invstd[c];
        w[e] = (gamma ? gamma[c] : 1.f) * is[e];
        b[e] = (MODE == 0 && beta) ? beta[c] : 0.f;
        k1[e] = (MODE == 1 && training) ? sums[c] * inv_count : 0.f;
        k2[e] = (MODE == 1 && training) ? sums[C + c] * inv_count : 0.f;
    }
    float mx = 0.f;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};                      // MODE 1 + colpart: this thread's sums of dx over its rows (its four channels are fixed)
    for (size_t i = i0; i < total4; i += (size_t)gridDim.x * CL_TH) {
        const float4 v = reinterpret_cast<const float4*>(a)[i];
        const float ve[4] = {v.x, v.y, v.z, v.w};
        float o[4];
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float r = (ve[e] - mu[e]) * w[e] + b[e];
                o[e] = relu ? fmaxf(r, 0.f) : r;
            }
        } else {
            const float4 xv = reinterpret_cast<const float4*>(x)[i];
            const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
            float ye[4] = {1.f, 1.f, 1.f, 1.f};
            if (relu) { const float4 yv = reinterpret_cast<const float4*>(y)[i]; ye[0] = yv.x; ye[1] = yv.y; ye[2] = yv.z; ye[3] = yv.w; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = (relu && !(ye[e] > 0.f)) ? 0.f : ve[e];
                o[e] = training ? w[e] * (g - k1[e] - ((xe[e] - mu[e]) * is[e]) * k2[e]) : g * w[e];
            }
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
        if (slot) mx = fmaxf(fmaxf(mx, fmaxf(m3t_fin_abs(o[0]), m3t_fin_abs(o[1]))), fmaxf(m3t_fin_abs(o[2]), m3t_fin_abs(o[3])));
        if (MODE == 1 && colpart) { cs[0] += o[0]; cs[1] += o[1]; cs[2] += o[2]; cs[3] += o[3]; }
    }
    if (MODE == 1 && colpart) {
        // block partial [C]: the 256 / (C / 4) threads that share a channel quad, in thread order (deterministic); colpart [blocks][C]
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[threadIdx.x * 4 + e] = cs[e];
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += CL_TH) {
            float t = 0.f;
            const int base = (int)((size_t)blockIdx.x * CL_TH % (size_t)c4n);      // quad of this block's thread 0
            const int first = ((c >> 2) - base + c4n) % c4n;                        // first thread of the block that holds quad c / 4
            for (int th = first; th < CL_TH; th += c4n) t += csum[th * 4 + (c & 3)];
            colpart[(size_t)blockIdx.x * C + c] = t;
        }
        __syncthreads();
    }
    if (slot) m3t_block_raise_slot(slot, mx, red4);         // (uniform: every thread of the block gets here)
}

// out[c] = sum over the blocks' partials, in block order
__global__ __launch_bounds__(256) void bncl_colsum_final_kernel(const float* __restrict__ colpart, int nblocks, int C, float* __restrict__ out) {
    __shared__ double red[4][64];
    const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    double t = 0.0;
    if (c < C) {
        double t8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};                  // eight loads in flight; combined in index order
        int k = ph;
        for (; k + 28 < nblocks; k += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) t8[u] += (double)colpart[(size_t)(k + 4 * u) * C + c];
        }
        for (; k < nblocks; k += 4) t8[0] += (double)colpart[(size_t)k * C + c];
        t = ((t8[0] + t8[1]) + (t8[2] + t8[3])) + ((t8[4] + t8[5]) + (t8[6] + t8[7]));
    }
    red[ph][cl] = t;
    __syncthreads();
    if (ph == 0 && c < C) out[c] = (float)(red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

static int cl_chunks(size_t M) {
    size_t c = (M + 255) / 256;
    return (int)(c < 1 ? 1 : (c > 1024 ? 1024 : c));
}

// ---- pooling of channels-last frames [P][H][W][C]: thread = (output position, channel quad)
struct PoolCL { int H, W, Ho, Wo, kh, kw, sh, sw, ph, pw, C; };

__global__ __launch_bounds__(CL_TH) void poolcl_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ win,
                                                           size_t total4, PoolCL g, unsigned long long* __restrict__ slot) {
    __shared__ float red4[4];
    const int c4n = g.C >> 2;
    float mx = 0.f;
    for (size_t i = (size_t)blockIdx.x * CL_TH + threadIdx.x; i < total4; i += (size_t)gridDim.x * CL_TH) {
        const int q = (int)(i % (size_t)c4n);
        size_t r = i / (size_t)c4n;
        const int wo = (int)(r % (size_t)g.Wo); r /= (size_t)g.Wo;
        const int ho = (int)(r % (size_t)g.Ho);
        const size_t p = r / (size_t)g.Ho;
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        bool first = true;
        for (int dh = 0; dh < g.kh; ++dh) {
            const int h = ho * g.sh - g.ph + dh;
            if ((unsigned)h >= (unsigned)g.H) continue;
            for (int dw = 0; dw < g.kw; ++dw) {
                const int w = wo * g.sw - g.pw + dw;
                if ((unsigned)w >= (unsigned)g.W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + ((p * g.H + h) * g.W + w) * (size_t)g.C + 4 * q);
                const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (first || ve[e] > best[e] || (ve[e] != ve[e] && best[e] == best[e])) { best[e] = ve[e]; bi[e] = dh * g.kw + dw; }      // first maximum; NaN wins
                first = false;
            }
        }
        reinterpret_cast<float4*>(y)[i] = make_float4(best[0], best[1], best[2], best[3]);
        reinterpret_cast<uchar4*>(win)[i] = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
        if (slot) mx = fmaxf(fmaxf(mx, fmaxf(m3t_fin_abs(best[0]), m3t_fin_abs(best[1]))), fmaxf(m3t_fin_abs(best[2]), m3t_fin_abs(best[3])));
    }
    if (slot) m3t_block_raise_slot(slot, mx, red4);
}

// dx[p][h][w][c] = sum over the windows (ho, wo) that cover (h, w) of dy[p][ho][wo][c] if that window's winner is (h, w)
__global__ __launch_bounds__(CL_TH) void poolcl_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ win, float* __restrict__ dx,
                                                           size_t total4, PoolCL g) {
    const int c4n = g.C >> 2;
    for (size_t i = (size_t)blockIdx.x * CL_TH + threadIdx.x; i < total4; i += (size_t)gridDim.x * CL_TH) {
        const int q = (int)(i % (size_t)c4n);
        size_t r = i / (size_t)c4n;
        const int w = (int)(r % (size_t)g.W); r /= (size_t)g.W;
        const int h = (int)(r % (size_t)g.H);
        const size_t p = r / (size_t)g.H;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        // windows with ho sh - ph <= h < ho sh - ph + kh
        const int ho_hi = min((h + g.ph) / g.sh, g.Ho - 1), wo_hi = min((w + g.pw) / g.sw, g.Wo - 1);
        for (int ho = ho_hi; ho >= 0; --ho) {
            const int dh = h + g.ph - ho * g.sh;
            if (dh >= g.kh) break;
            for (int wo = wo_hi; wo >= 0; --wo) {
                const int dw = w + g.pw - wo * g.sw;
                if (dw >= g.kw) break;
                const size_t o = ((p * g.Ho + ho) * g.Wo + wo) * (size_t)c4n + q;
                const uchar4 b = reinterpret_cast<const uchar4*>(win)[o];
                const float4 gq = reinterpret_cast<const float4*>(dy)[o];
                const int me = dh * g.kw + dw;
                if (b.x == me) acc[0] += gq.x;
                if (b.y == me) acc[1] += gq.y;
                if (b.z == me) acc[2] += gq.z;
                if (b.w == me) acc[3] += gq.w;
            }
        }
        reinterpret_cast<float4*>(dx)[i] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// ---- BatchNorm + ReLU + max pooling as ONE operator when the windows tile the frame (k x k, stride k, no padding: every pooling layer of the VGG-M
// stems).  y = relu(bn(x)) at full resolution is never written and never read: forward applies BatchNorm + ReLU inside the window loop and keeps
// the pooled value and the winner byte; in backward an input position's gradient through the pooling is dP of its window if it won, else 0 --
// computed in the BatchNorm backward's two sweeps, no scatter pass, and the ReLU mask is P > 0 of the window (the winner's y IS P).  Per pooled
// group: 2.3 instead of 4.9 passes over the convolution's output.
struct PoolF { int H, W, Ho, Wo, He, We, k, C; };           // He x We = ceil(H / k) x ceil(W / k): edge positions no window covers (odd H, W)

template <int K>
__global__ __launch_bounds__(CL_TH) void bnpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ yp,
                                                           unsigned char* __restrict__ win, size_t total4, PoolF g,
                                                           unsigned long long* __restrict__ slot) {
    __shared__ float red4[4];
    const int c4n = g.C >> 2;
    const size_t i0 = (size_t)blockIdx.x * CL_TH + threadIdx.x;
    const int q = (int)(i0 % (size_t)c4n);
    float w[4], b[4], mu[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * q + e;
        mu[e] = mean[c];
        w[e] = (gamma ? gamma[c] : 1.f) * invstd[c];
        b[e] = beta ? beta[c] : 0.f;
    }
    float mx = 0.f;
    for (size_t i = i0; i < total4; i += (size_t)gridDim.x * CL_TH) {
        size_t r = i / (size_t)c4n;
        const int wo = (int)(r % (size_t)g.Wo); r /= (size_t)g.Wo;
        const int ho = (int)(r % (size_t)g.Ho);
        const size_t p = r / (size_t)g.Ho;
        float4 v[K * K];
#pragma unroll
        for (int dh = 0; dh < K; ++dh)
#pragma unroll
            for (int dw = 0; dw < K; ++dw)
                v[dh * K + dw] = *reinterpret_cast<const float4*>(x + ((p * g.H + (size_t)(ho * K + dh)) * g.W + (size_t)(wo * K + dw)) * (size_t)g.C + 4 * q);
        float best[4];
        int bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < K * K; ++t) {
            const float ve[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = fmaxf((ve[e] - mu[e]) * w[e] + b[e], 0.f);
                const float yy = (ve[e] != ve[e]) ? ve[e] : y;          // (fmaxf would drop a NaN; torch.relu keeps it and the pooling lets it win)
                if (t == 0 || yy > best[e] || (yy != yy && best[e] == best[e])) { best[e] = yy; bi[e] = t; }
            }
        }
        reinterpret_cast<float4*>(yp)[i] = make_float4(best[0], best[1], best[2], best[3]);
        reinterpret_cast<uchar4*>(win)[i] = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
        if (slot) mx = fmaxf(fmaxf(mx, fmaxf(m3t_fin_abs(best[0]), m3t_fin_abs(best[1]))), fmaxf(m3t_fin_abs(best[2]), m3t_fin_abs(best[3])));
    }
    if (slot) m3t_block_raise_slot(slot, mx, red4);
}

// backward sums over the pooled windows: partial[chunk][2][C] = sum g, sum g xhat with g = dP where P > 0, taken at the window's winner
template <int K>
__global__ __launch_bounds__(CL_TH) void bnpool_bwd_partial_kernel(const float* __restrict__ dyp, const float* __restrict__ x, const float* __restrict__ yp,
                                                                   const unsigned char* __restrict__ win, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, size_t nwin, PoolF g, size_t win_per_chunk,
                                                                   double* __restrict__ partial) {
    extern __shared__ double red[];
    const int C = g.C, c4n = C >> 2, groups = CL_TH / c4n;
    const int q = threadIdx.x % c4n, gi = threadIdx.x / c4n;
    const size_t r0 = (size_t)blockIdx.x * win_per_chunk;
    const size_t r1 = r0 + win_per_chunk < nwin ? r0 + win_per_chunk : nwin;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, t[4] = {0.0, 0.0, 0.0, 0.0};
    float mu[4], is[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { mu[e] = mean[4 * q + e]; is[e] = invstd[4 * q + e]; }
    if (gi < groups) {
        for (size_t r = r0 + gi; r < r1; r += groups) {
            size_t rr = r;
            const int wo = (int)(rr % (size_t)g.Wo); rr /= (size_t)g.Wo;
            const int ho = (int)(rr % (size_t)g.Ho);
            const size_t p = rr / (size_t)g.Ho;
            const size_t o = r * (size_t)c4n + q;
            const float4 d = reinterpret_cast<const float4*>(dyp)[o];
            const float4 pv = reinterpret_cast<const float4*>(yp)[o];
            const uchar4 wb = reinterpret_cast<const uchar4*>(win)[o];
            const float de[4] = {d.x, d.y, d.z, d.w}, pe[4] = {pv.x, pv.y, pv.z, pv.w};
            const int we[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gq = pe[e] > 0.f ? de[e] : 0.f;
                const int dh = we[e] / K, dw = we[e] - dh * K;
                const float xv = x[((p * g.H + (size_t)(ho * K + dh)) * g.W + (size_t)(wo * K + dw)) * (size_t)C + 4 * q + e];
                s[e] += (double)gq;
                t[e] += (double)(gq * ((xv - mu[e]) * is[e]));
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[(size_t)gi * C + 4 * q + e] = s[e]; red[(size_t)(groups + gi) * C + 4 * q + e] = t[e]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += CL_TH) {
        double a0 = 0.0, b0 = 0.0;
        for (int k = 0; k < groups; ++k) { a0 += red[(size_t)k * C + c]; b0 += red[(size_t)(groups + k) * C + c]; }
        partial[((size_t)blockIdx.x * 2 + 0) * C + c] = a0;
        partial[((size_t)blockIdx.x * 2 + 1) * C + c] = b0;
    }
}

// dx at full resolution: a thread = one (extended) window x four channels; positions outside the pooled grid (odd H / W) have g = 0
template <int K>
__global__ __launch_bounds__(CL_TH) void bnpool_bwd_dx_kernel(const float* __restrict__ dyp, const float* __restrict__ x, const float* __restrict__ yp,
                                                              const unsigned char* __restrict__ win, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ sums, float* __restrict__ dx, size_t total4, PoolF g,
                                                              float inv_count, int training, unsigned long long* __restrict__ slot,
                                                              float* __restrict__ colpart) {
    __shared__ float red4[4];
    __shared__ float csum[CL_TH * 4];
    const int C = g.C, c4n = C >> 2;
    const size_t i0 = (size_t)blockIdx.x * CL_TH + threadIdx.x;
    const int q = (int)(i0 % (size_t)c4n);
    float w[4], mu[4], is[4], k1[4], k2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * q + e;
        mu[e] = mean[c]; is[e] = invstd[c];
        w[e] = (gamma ? gamma[c] : 1.f) * is[e];
        k1[e] = training ? sums[c] * inv_count : 0.f;
        k2[e] = training ? sums[C + c] * inv_count : 0.f;
    }
    float mx = 0.f, cs[4] = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = i0; i < total4; i += (size_t)gridDim.x * CL_TH) {
        size_t r = i / (size_t)c4n;
        const int we_ = (int)(r % (size_t)g.We); r /= (size_t)g.We;
        const int he = (int)(r % (size_t)g.He);
        const size_t p = r / (size_t)g.He;
        const bool pooled = he < g.Ho && we_ < g.Wo;
        float de[4] = {0.f, 0.f, 0.f, 0.f};
        int wi[4] = {-1, -1, -1, -1};
        if (pooled) {
            const size_t o = ((p * g.Ho + he) * g.Wo + we_) * (size_t)c4n + q;
            const float4 d = reinterpret_cast<const float4*>(dyp)[o];
            const float4 pv = reinterpret_cast<const float4*>(yp)[o];
            const uchar4 wb = reinterpret_cast<const uchar4*>(win)[o];
            de[0] = pv.x > 0.f ? d.x : 0.f; de[1] = pv.y > 0.f ? d.y : 0.f; de[2] = pv.z > 0.f ? d.z : 0.f; de[3] = pv.w > 0.f ? d.w : 0.f;
            wi[0] = wb.x; wi[1] = wb.y; wi[2] = wb.z; wi[3] = wb.w;
        }
#pragma unroll
        for (int dh = 0; dh < K; ++dh) {
            const int h = he * K + dh;
            if (h >= g.H) continue;
#pragma unroll
            for (int dw = 0; dw < K; ++dw) {
                const int ww = we_ * K + dw;
                if (ww >= g.W) continue;
                const size_t xo = ((p * g.H + h) * g.W + ww) * (size_t)C + 4 * q;
                const float4 xv = *reinterpret_cast<const float4*>(x + xo);
                const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gq = (wi[e] == dh * K + dw) ? de[e] : 0.f;
                    o[e] = training ? w[e] * (gq - k1[e] - ((xe[e] - mu[e]) * is[e]) * k2[e]) : gq * w[e];
                }
                *reinterpret_cast<float4*>(dx + xo) = make_float4(o[0], o[1], o[2], o[3]);
                if (slot) mx = fmaxf(fmaxf(mx, fmaxf(m3t_fin_abs(o[0]), m3t_fin_abs(o[1]))), fmaxf(m3t_fin_abs(o[2]), m3t_fin_abs(o[3])));
                if (colpart) { cs[0] += o[0]; cs[1] += o[1]; cs[2] += o[2]; cs[3] += o[3]; }
            }
        }
    }
    if (colpart) {
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[threadIdx.x * 4 + e] = cs[e];
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += CL_TH) {
            float t = 0.f;
            for (int th = c >> 2; th < CL_TH; th += c4n) t += csum[th * 4 + (c & 3)];
            colpart[(size_t)blockIdx.x * C + c] = t;
        }
        __syncthreads();
    }
    if (slot) m3t_block_raise_slot(slot, mx, red4);
}

static int grid_for(size_t total4, int c4n) {
    // a grid whose stride (blocks x 256) is a multiple of C / 4 (it divides 256), >= ~16 float4 per thread, at most 2048 blocks
    size_t b = (total4 + (size_t)CL_TH * 16 - 1) / ((size_t)CL_TH * 16);
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    (void)c4n;
    return (int)b;
}

}  // namespace

// BatchNorm (+ReLU) over channels-last rows x [M][C] at any M (the 3-D stems: M = N T H W up to millions); C % 4 == 0, C / 4 divides 256, 16-B
// aligned tensors.  Semantics of m3t_bn_rows_fwd / _bwd.  m3t_amax_out arms the magnitude slot of y (forward) / dx (backward).
extern "C" size_t m3t_bn_cl_ws_bytes(size_t M, int C) {
    // fp64 chunk partials | sums [2 C] | the dx pass's block partials [<= 2048 blocks][C] (m3t_bn_cl_bwd with dx_colsum)
    return (size_t)cl_chunks(M) * 2 * (size_t)C * sizeof(double) + 2 * (size_t)C * sizeof(float) + 512 +
           (size_t)grid_for(M * (size_t)(C / 4), C / 4) * C * sizeof(float);
}

static bool bncl_shape_ok(size_t M, int C) { return C > 0 && C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && M > 0; }

extern "C" int m3t_bn_cl_fwd(const float* x, size_t M, int C, const float* gamma, const float* beta, float* run_mean, float* run_var,
                             float momentum, float eps, int training, int relu, float* y, float* save_mean, float* save_invstd, float* ws,
                             size_t ws_bytes, void* stream) {
    unsigned long long* slot = m3t_take_amax_out();
    if (!bncl_shape_ok(M, C) || !x || !y || !save_mean || !save_invstd || ((uintptr_t)x % 16) != 0 || ((uintptr_t)y % 16) != 0) return M3T_EINVAL;
    if (!training && (!run_mean || !run_var)) return M3T_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (training) {
        if (M < 2) return M3T_EINVAL;
        if (!ws || ws_bytes < m3t_bn_cl_ws_bytes(M, C) || ((uintptr_t)ws % 8) != 0) return M3T_EINVAL;
        const int nch = cl_chunks(M);
        const size_t rpc = (M + nch - 1) / nch;
        double* partial = reinterpret_cast<double*>(ws);
        const int groups = CL_TH / (C / 4);
        bncl_partial_kernel<0><<<nch, CL_TH, (size_t)2 * groups * C * sizeof(double), s>>>(x, nullptr, nullptr, nullptr, nullptr, M, C, rpc, 0, partial);
        M3T_LAUNCH_CHECK();
        bncl_stats_final_kernel<<<cdiv(C, 64), 256, 0, s>>>(partial, nch, (double)M, C, eps, momentum, run_mean, run_var, save_mean, save_invstd);
        M3T_LAUNCH_CHECK();
    } else {
        bncl_eval_stats_kernel<<<cdiv(C, 256), 256, 0, s>>>(run_mean, run_var, C, eps, save_mean, save_invstd);
        M3T_LAUNCH_CHECK();
    }
    const size_t total4 = M * (size_t)(C / 4);
    bncl_map_kernel<0><<<grid_for(total4, C / 4), CL_TH, 0, s>>>(x, nullptr, nullptr, gamma, beta, save_mean, save_invstd, nullptr, y, total4, C, 0.f,
                                                                training, relu, slot, nullptr);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_bn_cl_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* save_mean, const float* save_invstd,
                             size_t M, int C, int training, int relu, float* dx, float* dgamma, float* dbeta, float* dx_colsum, float* ws,
                             size_t ws_bytes, void* stream) {
    unsigned long long* slot = m3t_take_amax_out();
    if (!bncl_shape_ok(M, C) || !dy || !x || !dx || !save_mean || !save_invstd || ((uintptr_t)dy % 16) != 0 || ((uintptr_t)x % 16) != 0 ||
        ((uintptr_t)dx % 16) != 0)
        return M3T_EINVAL;
    if (relu && (!y || ((uintptr_t)y % 16) != 0)) return M3T_EINVAL;
    if (!ws || ws_bytes < m3t_bn_cl_ws_bytes(M, C) || ((uintptr_t)ws % 8) != 0) return M3T_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nch = cl_chunks(M);
    const size_t rpc = (M + nch - 1) / nch;
    double* partial = reinterpret_cast<double*>(ws);
    float* sums = reinterpret_cast<float*>(partial + (size_t)nch * 2 * C);
    const int groups = CL_TH / (C / 4);
    bncl_partial_kernel<1><<<nch, CL_TH, (size_t)2 * groups * C * sizeof(double), s>>>(dy, x, y, save_mean, save_invstd, M, C, rpc, relu, partial);
    M3T_LAUNCH_CHECK();
    bncl_bwd_final_kernel<<<cdiv(C, 64), 256, 0, s>>>(partial, nch, C, sums, dgamma, dbeta);
    M3T_LAUNCH_CHECK();
    const size_t total4 = M * (size_t)(C / 4);
    // dx_colsum [C] (optional): the column sums of dx = the bias gradient of the convolution in front of this BatchNorm, from the same pass
    // (block partials behind the fp64 partials in ws, summed in block order)
    const int nblk = grid_for(total4, C / 4);
    float* colpart = dx_colsum ? sums + 2 * (size_t)C + 64 : nullptr;
    bncl_map_kernel<1><<<nblk, CL_TH, 0, s>>>(dy, x, y, gamma, nullptr, save_mean, save_invstd, sums, dx, total4, C, (float)(1.0 / (double)M),
                                             training, relu, slot, colpart);
    M3T_LAUNCH_CHECK();
    if (dx_colsum) {
        bncl_colsum_final_kernel<<<cdiv(C, 64), 256, 0, s>>>(colpart, nblk, C, dx_colsum);
        M3T_LAUNCH_CHECK();
    }
    return 0;
}

// Max pooling of P channels-last frames x [P][H][W][C] -> y [P][Ho][Wo][C] (nn.MaxPool3d((1, kh, kw)) of the stems on channels-last rows);
// win: one byte per output element (the winner's place in its window).  C % 4 == 0, kh kw <= 255.  m3t_amax_out arms y's magnitude slot.
extern "C" int m3t_pool_cl_fwd(const float* x, size_t P, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, float* y,
                               unsigned char* win, void* stream) {
    unsigned long long* slot = m3t_take_amax_out();
    if (P == 0) return 0;
    if (!x || !y || !win || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0 || kh <= 0 || kw <= 0 || kh * kw > 255 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0 ||
        ph >= kh || pw >= kw || ((uintptr_t)x % 16) != 0 || ((uintptr_t)y % 16) != 0 || ((uintptr_t)win % 4) != 0)
        return M3T_EINVAL;
    PoolCL g;
    g.H = H; g.W = W; g.kh = kh; g.kw = kw; g.sh = sh; g.sw = sw; g.ph = ph; g.pw = pw; g.C = C;
    g.Ho = (H + 2 * ph - kh) / sh + 1; g.Wo = (W + 2 * pw - kw) / sw + 1;
    if (g.Ho < 1 || g.Wo < 1) return M3T_EINVAL;
    const size_t total4 = P * (size_t)g.Ho * g.Wo * (size_t)(C / 4);
    poolcl_fwd_kernel<<<grid_for(total4, C / 4), CL_TH, 0, (hipStream_t)stream>>>(x, y, win, total4, g, slot);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_pool_cl_bwd(const float* dy, const unsigned char* win, size_t P, int H, int W, int C, int kh, int kw, int sh, int sw, int ph,
                               int pw, float* dx, void* stream) {
    if (P == 0) return 0;
    if (!dy || !dx || !win || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0 || kh <= 0 || kw <= 0 || kh * kw > 255 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0 ||
        ph >= kh || pw >= kw || ((uintptr_t)dy % 16) != 0 || ((uintptr_t)dx % 16) != 0 || ((uintptr_t)win % 4) != 0)
        return M3T_EINVAL;
    PoolCL g;
    g.H = H; g.W = W; g.kh = kh; g.kw = kw; g.sh = sh; g.sw = sw; g.ph = ph; g.pw = pw; g.C = C;
    g.Ho = (H + 2 * ph - kh) / sh + 1; g.Wo = (W + 2 * pw - kw) / sw + 1;
    if (g.Ho < 1 || g.Wo < 1) return M3T_EINVAL;
    const size_t total4 = P * (size_t)H * W * (size_t)(C / 4);
    poolcl_bwd_kernel<<<grid_for(total4, C / 4), CL_TH, 0, (hipStream_t)stream>>>(dy, win, dx, total4, g);
    M3T_LAUNCH_CHECK();
    return 0;
}

// BatchNorm (+ReLU) + max pooling with a k x k window, stride k, no padding (k = 2 or 3) as ONE operator over P channels-last frames x [P][H][W][C]:
// forward writes only the pooled frames yp [P][Ho][Wo][C] and the winner bytes; backward takes d(yp) and writes dx at full resolution.  Statistics,
// running-statistics update, ws, slots and dx_colsum as m3t_bn_cl_fwd / _bwd (M = P H W values per channel).
extern "C" int m3t_bn_pool_cl_fwd(const float* x, size_t P, int H, int W, int C, int k, const float* gamma, const float* beta, float* run_mean,
                                  float* run_var, float momentum, float eps, int training, float* yp, unsigned char* win, float* save_mean,
                                  float* save_invstd, float* ws, size_t ws_bytes, void* stream) {
    unsigned long long* slot = m3t_take_amax_out();
    const size_t M = P * (size_t)H * W;
    if (!bncl_shape_ok(M, C) || (k != 2 && k != 3) || H < k || W < k || !x || !yp || !win || !save_mean || !save_invstd || ((uintptr_t)x % 16) != 0 ||
        ((uintptr_t)yp % 16) != 0 || ((uintptr_t)win % 4) != 0)
        return M3T_EINVAL;
    if (!training && (!run_mean || !run_var)) return M3T_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (training) {
        if (M < 2) return M3T_EINVAL;
        if (!ws || ws_bytes < m3t_bn_cl_ws_bytes(M, C) || ((uintptr_t)ws % 8) != 0) return M3T_EINVAL;
        const int nch = cl_chunks(M);
        const size_t rpc = (M + nch - 1) / nch;
        double* partial = reinterpret_cast<double*>(ws);
        const int groups = CL_TH / (C / 4);
        bncl_partial_kernel<0><<<nch, CL_TH, (size_t)2 * groups * C * sizeof(double), s>>>(x, nullptr, nullptr, nullptr, nullptr, M, C, rpc, 0, partial);
        M3T_LAUNCH_CHECK();
        bncl_stats_final_kernel<<<cdiv(C, 64), 256, 0, s>>>(partial, nch, (double)M, C, eps, momentum, run_mean, run_var, save_mean, save_invstd);
        M3T_LAUNCH_CHECK();
    } else {
        bncl_eval_stats_kernel<<<cdiv(C, 256), 256, 0, s>>>(run_mean, run_var, C, eps, save_mean, save_invstd);
        M3T_LAUNCH_CHECK();
    }
    PoolF g;
    g.H = H; g.W = W; g.k = k; g.C = C; g.Ho = H / k; g.Wo = W / k; g.He = (H + k - 1) / k; g.We = (W + k - 1) / k;
    const size_t total4 = P * (size_t)g.Ho * g.Wo * (size_t)(C / 4);
    if (k == 2) bnpool_fwd_kernel<2><<<grid_for(total4, C / 4), CL_TH, 0, s>>>(x, gamma, beta, save_mean, save_invstd, yp, win, total4, g, slot);
    else bnpool_fwd_kernel<3><<<grid_for(total4, C / 4), CL_TH, 0, s>>>(x, gamma, beta, save_mean, save_invstd, yp, win, total4, g, slot);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_bn_pool_cl_bwd(const float* dyp, const float* x, const float* yp, const unsigned char* win, const float* gamma,
                                  const float* save_mean, const float* save_invstd, size_t P, int H, int W, int C, int k, int training, float* dx,
                                  float* dgamma, float* dbeta, float* dx_colsum, float* ws, size_t ws_bytes, void* stream) {
    unsigned long long* slot = m3t_take_amax_out();
    const size_t M = P * (size_t)H * W;
    if (!bncl_shape_ok(M, C) || (k != 2 && k != 3) || H < k || W < k || !dyp || !x || !yp || !win || !dx || !save_mean || !save_invstd ||
        ((uintptr_t)dyp % 16) != 0 || ((uintptr_t)x % 16) != 0 || ((uintptr_t)yp % 16) != 0 || ((uintptr_t)dx % 16) != 0 || ((uintptr_t)win % 4) != 0)
        return M3T_EINVAL;
    if (!ws || ws_bytes < m3t_bn_cl_ws_bytes(M, C) || ((uintptr_t)ws % 8) != 0) return M3T_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    PoolF g;
    g.H = H; g.W = W; g.k = k; g.C = C; g.Ho = H / k; g.Wo = W / k; g.He = (H + k - 1) / k; g.We = (W + k - 1) / k;
    const size_t nwin = P * (size_t)g.Ho * g.Wo;
    const int nch = cl_chunks(nwin);
    const size_t wpc = (nwin + nch - 1) / nch;
    double* partial = reinterpret_cast<double*>(ws);
    float* sums = reinterpret_cast<float*>(partial + (size_t)cl_chunks(M) * 2 * C);
    const int groups = CL_TH / (C / 4);
    if (k == 2) bnpool_bwd_partial_kernel<2><<<nch, CL_TH, (size_t)2 * groups * C * sizeof(double), s>>>(dyp, x, yp, win, save_mean, save_invstd, nwin, g, wpc, partial);
    else bnpool_bwd_partial_kernel<3><<<nch, CL_TH, (size_t)2 * groups * C * sizeof(double), s>>>(dyp, x, yp, win, save_mean, save_invstd, nwin, g, wpc, partial);
    M3T_LAUNCH_CHECK();
    bncl_bwd_final_kernel<<<cdiv(C, 64), 256, 0, s>>>(partial, nch, C, sums, dgamma, dbeta);
    M3T_LAUNCH_CHECK();
    const size_t total4 = P * (size_t)g.He * g.We * (size_t)(C / 4);
    const int nblk = grid_for(total4, C / 4);
    float* colpart = dx_colsum ? sums + 2 * (size_t)C + 64 : nullptr;
    if (k == 2) bnpool_bwd_dx_kernel<2><<<nblk, CL_TH, 0, s>>>(dyp, x, yp, win, gamma, save_mean, save_invstd, sums, dx, total4, g, (float)(1.0 / (double)M), training, slot, colpart);
    else bnpool_bwd_dx_kernel<3><<<nblk, CL_TH, 0, s>>>(dyp, x, yp, win, gamma, save_mean, save_invstd, sums, dx, total4, g, (float)(1.0 / (double)M), training, slot, colpart);
    M3T_LAUNCH_CHECK();
    if (dx_colsum) {
        bncl_colsum_final_kernel<<<cdiv(C, 64), 256, 0, s>>>(colpart, nblk, C, dx_colsum);
        M3T_LAUNCH_CHECK();
    }
    return 0;
}
