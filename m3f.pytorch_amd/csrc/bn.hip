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

}  // namespace

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
