// Post-processing of valence/arousal tracks: the Wiener / median smoothing the reference does with
// scipy (models/utils.py:29-33, called with window 35 in get_smoothed_ccc.py:16-17 and window 13 in
// create_submission.py:34-35) and its numpy CCC report (models/utils.py:20-22).  Tiny, HBM/latency-bound work: one
// workgroup per track, all tracks of a call (every video x {valence, arousal}) in ONE launch; fp64 arithmetic like
// scipy on fp32 predictions; fixed-order reductions (deterministic).
#include "common.h"

namespace {

constexpr int MAXW = 129;      // largest smoothing window (odd)

__device__ __forceinline__ double block_sum_f64(double v, double* red) {
    const int tid = threadIdx.x;
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    return red[0];
}

// local mean / variance over the centred, zero-padded window; x*x is rounded to fp32 first (scipy squares in the input dtype)
__device__ __forceinline__ void local_stats(const float* __restrict__ x, long n, long i, int window, double& lmean, double& lvar) {
    const int h = window >> 1;
    double s = 0.0, ss = 0.0;
    for (int k = -h; k <= h; ++k) {
        const long j = i + k;
        if (j >= 0 && j < n) {
            const float v = x[j];
            const float sq = v * v;
            s += (double)v;
            ss += (double)sq;
        }
    }
    lmean = s / window;
    lvar = ss / window - lmean * lmean;
}

// mode 0: scipy.signal.wiener(x, window); mode 1: scipy.signal.medfilt(x, window)
__global__ __launch_bounds__(256) void smooth_tracks_kernel(const float* __restrict__ x, const long long* __restrict__ offsets,
                                                            int window, int mode, double* __restrict__ y) {
    __shared__ double red[256];
    const long b = offsets[blockIdx.x], n = offsets[blockIdx.x + 1] - b;
    const float* xt = x + b;
    double* yt = y + b;
    if (n <= 0) return;
    if (mode == 1) {
        const int h = window >> 1;
        for (long i = threadIdx.x; i < n; i += blockDim.x) {
            float w[MAXW];
            for (int k = 0; k < window; ++k) {
                const long j = i + k - h;
                w[k] = (j >= 0 && j < n) ? xt[j] : 0.f;
            }
            float med = 0.f;                                     // the element of rank h (ties broken by position)
            for (int a = 0; a < window; ++a) {
                int rank = 0;
                for (int c = 0; c < window; ++c) rank += (w[c] < w[a]) || (w[c] == w[a] && c < a);
                if (rank == h) med = w[a];
            }
            yt[i] = (double)med;
        }
        return;
    }
    double acc = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        double lm, lv;
        local_stats(xt, n, i, window, lm, lv);
        acc += lv;
    }
    const double noise = block_sum_f64(acc, red) / (double)n;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        double lm, lv;
        local_stats(xt, n, i, window, lm, lv);
        yt[i] = lv < noise ? lm : ((double)xt[i] - lm) * (1.0 - noise / lv) + lm;
    }
}

// out[0] = CCC of p vs g over the frames with g >= -1 (and g2 >= -1 when g2 != NULL), numpy semantics
// (models/utils.py:20-22): biased variances, except var(p) when p_unbiased (the torch-tensor quirk of
// get_smoothed_ccc.py).  out[1] = number of valid frames.  One workgroup, two passes.
__global__ __launch_bounds__(256) void ccc_masked_kernel(const double* __restrict__ p, const float* __restrict__ g,
                                                         const float* __restrict__ g2, long n, int p_unbiased,
                                                         double* __restrict__ out) {
    __shared__ double red[256];
    double sp = 0.0, sg = 0.0, cnt = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const bool ok = g[i] >= -1.f && (!g2 || g2[i] >= -1.f);
        if (ok) { sp += p[i]; sg += (double)g[i]; cnt += 1.0; }
    }
    sp = block_sum_f64(sp, red);
    sg = block_sum_f64(sg, red);
    cnt = block_sum_f64(cnt, red);
    const double mp = sp / cnt, mg = sg / cnt;
    double vp = 0.0, vg = 0.0, cv = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const bool ok = g[i] >= -1.f && (!g2 || g2[i] >= -1.f);
        if (ok) {
            const double a = p[i] - mp, c = (double)g[i] - mg;
            vp += a * a; vg += c * c; cv += a * c;
        }
    }
    vp = block_sum_f64(vp, red);
    vg = block_sum_f64(vg, red);
    cv = block_sum_f64(cv, red);
    if (threadIdx.x == 0) {
        const double varp = vp / (p_unbiased ? cnt - 1.0 : cnt), varg = vg / cnt, cov = cv / cnt;
        out[0] = 2.0 * cov / (varp + varg + (mp - mg) * (mp - mg));
        out[1] = cnt;
    }
}

}  // namespace

extern "C" int m3t_smooth_tracks(const float* x, const long long* offsets, int n_tracks, int window, int mode, double* y,
                                 void* stream) {
    if (n_tracks <= 0) return 0;
    if (!x || !offsets || !y || window < 1 || window > MAXW || (window & 1) == 0 || (mode != 0 && mode != 1)) return M3T_EINVAL;
    smooth_tracks_kernel<<<n_tracks, 256, 0, (hipStream_t)stream>>>(x, offsets, window, mode, y);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_ccc_masked(const double* p, const float* g, const float* g2, long long n, int p_unbiased, double* out2,
                              void* stream) {
    if (!p || !g || !out2 || n <= 0) return M3T_EINVAL;
    ccc_masked_kernel<<<1, 256, 0, (hipStream_t)stream>>>(p, g, g2, (long)n, p_unbiased, out2);
    M3T_LAUNCH_CHECK();
    return 0;
}
