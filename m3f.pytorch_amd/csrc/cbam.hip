// Per-frame SE/CBAM gating (models/cbam.py) for gfx950.  x is [N,C,H,W] with N = B*T frames.
// Channel gate: one workgroup per frame; every (n,c) plane is squeezed (avg + max + argmax)
// by ONE wavefront with shuffle reductions, the shared MLP C -> C/r -> C runs out of LDS, and
// the same workgroup rescales the frame while it is still L2-hot: x is read from HBM once.
// Spatial gate: thread-per-pixel kernels, coalesced along H*W for every channel; BatchNorm2d(1)
// batch statistics go through fp64 per-block partials and a fixed-order final sum
// (deterministic, no atomics).  Parameter gradients that reduce over frames are left as
// [N, .] slabs for m3t_sgemm / m3t_colsum.
#include "common.h"

namespace {

__device__ __forceinline__ void wave_argmax(float& v, int& idx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double r = 0.0;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}

// ------------------------------------------------------------------ channel gate
__global__ __launch_bounds__(256) void channel_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, float* __restrict__ y,
                                                          float* __restrict__ pooled, int32_t* __restrict__ argmax,
                                                          float* __restrict__ hidden, float* __restrict__ scale, int C,
                                                          int Cr, int HW) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_avg = sm;            // [C]
    float* s_max = sm + C;        // [C]
    float* s_sc = sm + 2 * C;     // [C]
    float* s_h = sm + 3 * C;      // [2*Cr]
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xb = x + (size_t)n * C * HW;
    if (HW <= 32) {
        // small maps (e.g. 4x4 at the last ResNet stage): a wavefront squeezes 64/g planes at once, g = pow2 >= H*W
        int g = 4;
        while (g < HW) g <<= 1;
        const int per = 64 / g, sub = lane % g, pi = lane / g;
        for (int c0 = wave * per; c0 < C; c0 += 4 * per) {
            const int c = c0 + pi;
            const bool ok = c < C && sub < HW;
            const float v = ok ? xb[(size_t)c * HW + sub] : -INFINITY;
            float sum = ok ? v : 0.f, mx = v;
            int am = ok ? sub : 0x7fffffff;
            for (int o = g >> 1; o > 0; o >>= 1) {
                sum += __shfl_xor(sum, o, 64);
                const float ov = __shfl_xor(mx, o, 64);
                const int oi = __shfl_xor(am, o, 64);
                if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
            }
            if (sub == 0 && c < C) {
                const float avg = sum / (float)HW;
                s_avg[c] = avg; s_max[c] = mx;
                pooled[((size_t)n * 2 + 0) * C + c] = avg;
                pooled[((size_t)n * 2 + 1) * C + c] = mx;
                argmax[(size_t)n * C + c] = am;
            }
        }
    } else
    for (int c = wave; c < C; c += 4) {
        const float* pl = xb + (size_t)c * HW;
        float sum = 0.f, mx = -INFINITY;
        int am = 0x7fffffff;
        if ((HW & 3) == 0 && ((uintptr_t)pl & 15) == 0) {
            const float4* p4 = reinterpret_cast<const float4*>(pl);
            for (int i = lane; i < (HW >> 2); i += 64) {
                const float4 v = p4[i];
                sum += (v.x + v.y) + (v.z + v.w);
                if (v.x > mx) { mx = v.x; am = 4 * i; }
                if (v.y > mx) { mx = v.y; am = 4 * i + 1; }
                if (v.z > mx) { mx = v.z; am = 4 * i + 2; }
                if (v.w > mx) { mx = v.w; am = 4 * i + 3; }
            }
        } else
        for (int i = lane; i < HW; i += 64) {
            const float v = pl[i];
            sum += v;
            if (v > mx) { mx = v; am = i; }
        }
        sum = wave_sum(sum);
        wave_argmax(mx, am);
        if (lane == 0) {
            const float avg = sum / (float)HW;
            s_avg[c] = avg; s_max[c] = mx;
            pooled[((size_t)n * 2 + 0) * C + c] = avg;
            pooled[((size_t)n * 2 + 1) * C + c] = mx;
            argmax[(size_t)n * C + c] = am;
        }
    }
    __syncthreads();
    // hidden = W1 [Cr,C] x pooled: one WAVEFRONT per output (coalesced reads of the W1 row, shuffle reduction)
    for (int j = wave; j < 2 * Cr; j += 4) {
        const int which = j / Cr, r = j % Cr;
        const float* src = which ? s_max : s_avg;
        const float* wr = w1 + (size_t)r * C;
        float h = 0.f;
        for (int c = lane; c < C; c += 64) h += wr[c] * src[c];
        h = wave_sum(h) + b1[r];
        if (lane == 0) {
            hidden[((size_t)n * 2 + which) * Cr + r] = h;
            s_h[j] = fmaxf(h, 0.f);
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const float* wr = w2 + (size_t)c * Cr;
        float a0 = b2[c], a1 = b2[c];
        for (int r = 0; r < Cr; ++r) { a0 += wr[r] * s_h[r]; a1 += wr[r] * s_h[Cr + r]; }
        const float sc = 1.f / (1.f + expf(-(a0 + a1)));
        s_sc[c] = sc;
        scale[(size_t)n * C + c] = sc;
    }
    __syncthreads();
    float* yb = y + (size_t)n * C * HW;
    const int total = C * HW;
    if ((HW & 3) == 0 && (((uintptr_t)xb | (uintptr_t)yb) & 15) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(xb);
        float4* y4 = reinterpret_cast<float4*>(yb);
        for (int i = tid; i < (total >> 2); i += 256) {
            const float sc = s_sc[(i << 2) / HW];
            float4 v = x4[i];
            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
            y4[i] = v;
        }
    } else {
        for (int i = tid; i < total; i += 256) yb[i] = xb[i] * s_sc[i / HW];
    }
}

__global__ __launch_bounds__(256) void channel_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ w1, const float* __restrict__ w2,
                                                          const int32_t* __restrict__ argmax, const float* __restrict__ hidden,
                                                          const float* __restrict__ scale, float* __restrict__ dx,
                                                          float* __restrict__ g_datt, float* __restrict__ g_dh,
                                                          float* __restrict__ g_r, int C, int Cr, int HW) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_datt = sm;            // [C]
    float* s_davg = sm + C;        // [C]
    float* s_dmax = sm + 2 * C;    // [C]
    float* s_dh = sm + 3 * C;      // [2*Cr]
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xb = x + (size_t)n * C * HW;
    const float* gb = dy + (size_t)n * C * HW;
    if (HW <= 32) {
        int g = 4;
        while (g < HW) g <<= 1;
        const int per = 64 / g, sub = lane % g, pi = lane / g;
        for (int c0 = wave * per; c0 < C; c0 += 4 * per) {
            const int c = c0 + pi;
            const bool ok = c < C && sub < HW;
            float ds = ok ? gb[(size_t)c * HW + sub] * xb[(size_t)c * HW + sub] : 0.f;
            for (int o = g >> 1; o > 0; o >>= 1) ds += __shfl_xor(ds, o, 64);
            if (sub == 0 && c < C) {
                const float sc = scale[(size_t)n * C + c];
                const float da = ds * sc * (1.f - sc);
                s_datt[c] = da;
                g_datt[(size_t)n * C + c] = da;
            }
        }
    } else
    for (int c = wave; c < C; c += 4) {
        const float* pl = xb + (size_t)c * HW;
        const float* gl = gb + (size_t)c * HW;
        float ds = 0.f;
        if ((HW & 3) == 0 && (((uintptr_t)pl | (uintptr_t)gl) & 15) == 0) {
            const float4* p4 = reinterpret_cast<const float4*>(pl);
            const float4* g4 = reinterpret_cast<const float4*>(gl);
            for (int i = lane; i < (HW >> 2); i += 64) {
                const float4 a = g4[i], b = p4[i];
                ds += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
            }
        } else
        for (int i = lane; i < HW; i += 64) ds += gl[i] * pl[i];
        ds = wave_sum(ds);
        if (lane == 0) {
            const float s = scale[(size_t)n * C + c];
            const float da = ds * s * (1.f - s);
            s_datt[c] = da;
            g_datt[(size_t)n * C + c] = da;
        }
    }
    __syncthreads();
    // g[r] = sum_c datt[c] * W2[c][r]: the C range is split over the 4 waves (coalesced W2 reads), partials in LDS
    float* s_part = s_davg;                     // [4][Cr] scratch (s_davg/s_dmax are written later)
    if (lane < Cr) {
        float g = 0.f;
        for (int c = wave; c < C; c += 4) g += s_datt[c] * w2[(size_t)c * Cr + lane];
        s_part[wave * Cr + lane] = g;
    }
    __syncthreads();
    for (int j = tid; j < 2 * Cr; j += 256) {
        const int which = j / Cr, r = j % Cr;
        const float g = (s_part[r] + s_part[Cr + r]) + (s_part[2 * Cr + r] + s_part[3 * Cr + r]);
        const float h = hidden[((size_t)n * 2 + which) * Cr + r];
        const float dh = h > 0.f ? g : 0.f;
        s_dh[j] = dh;
        g_dh[((size_t)n * 2 + which) * Cr + r] = dh;
        if (which == 0) {
            const float hm = hidden[((size_t)n * 2 + 1) * Cr + r];
            g_r[(size_t)n * Cr + r] = fmaxf(h, 0.f) + fmaxf(hm, 0.f);
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float da = 0.f, dm = 0.f;
        for (int r = 0; r < Cr; ++r) {
            const float w = w1[(size_t)r * C + c];
            da += s_dh[r] * w;
            dm += s_dh[Cr + r] * w;
        }
        s_davg[c] = da / (float)HW;
        s_dmax[c] = dm;
    }
    __syncthreads();
    float* db = dx + (size_t)n * C * HW;
    const int total = C * HW;
    if ((HW & 3) == 0 && (((uintptr_t)gb | (uintptr_t)db) & 15) == 0) {
        const float4* g4 = reinterpret_cast<const float4*>(gb);
        float4* d4 = reinterpret_cast<float4*>(db);
        const int HW4 = HW >> 2;
        for (int c = wave; c < C; c += 4) {                     // one wavefront per plane: no per-element division
            const float sc = scale[(size_t)n * C + c], da = s_davg[c], dm = s_dmax[c];
            const int am = argmax[(size_t)n * C + c];
            for (int i = lane; i < HW4; i += 64) {
                float4 v = g4[(size_t)c * HW4 + i];
                v.x = v.x * sc + da; v.y = v.y * sc + da; v.z = v.z * sc + da; v.w = v.w * sc + da;
                if ((am >> 2) == i) {
                    const int e = am & 3;
                    if (e == 0) v.x += dm; else if (e == 1) v.y += dm; else if (e == 2) v.z += dm; else v.w += dm;
                }
                d4[(size_t)c * HW4 + i] = v;
            }
        }
        return;
    }
    for (int i = tid; i < total; i += 256) {
        const int c = i / HW, p = i - c * HW;
        float v = gb[i] * scale[(size_t)n * C + c] + s_davg[c];
        if (p == argmax[(size_t)n * C + c]) v += s_dmax[c];
        db[i] = v;
    }
}

// ------------------------------------------------------------------ spatial gate
__global__ __launch_bounds__(256) void spatial_compress_kernel(const float* __restrict__ x, float* __restrict__ comp,
                                                               int32_t* __restrict__ cargmax, int N, int C, int HW) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;      // flattened (frame, pixel): full waves for any H*W
    if (gid >= (size_t)N * HW) return;
    const int n = (int)(gid / HW), p = (int)(gid % HW);
    const float* xb = x + (size_t)n * C * HW + p;
    float mx = -INFINITY, sum = 0.f;
    int am = 0;
    for (int c = 0; c < C; ++c) {
        const float v = xb[(size_t)c * HW];
        sum += v;
        if (v > mx) { mx = v; am = c; }
    }
    comp[((size_t)n * 2 + 0) * HW + p] = mx;
    comp[((size_t)n * 2 + 1) * HW + p] = sum / (float)C;
    cargmax[(size_t)n * HW + p] = am;
}

__global__ __launch_bounds__(256) void spatial_conv_kernel(const float* __restrict__ comp, const float* __restrict__ w,
                                                           float* __restrict__ conv, double* __restrict__ part, int N,
                                                           int H, int W) {
    __shared__ double red[8];
    const int HW = H * W;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int n = (int)(gid / HW), p = (int)(gid % HW);
    double s1 = 0.0, s2 = 0.0;
    if (gid < (size_t)N * HW) {
        const int h = p / W, ww = p % W;
        float acc = 0.f;
        for (int ch = 0; ch < 2; ++ch) {
            const float* cp = comp + ((size_t)n * 2 + ch) * HW;
            for (int i = 0; i < 5; ++i) {
                const int hh = h + i - 2;
                if (hh < 0 || hh >= H) continue;
                for (int j = 0; j < 5; ++j) {
                    const int wj = ww + j - 2;
                    if (wj < 0 || wj >= W) continue;
                    acc += w[(ch * 5 + i) * 5 + j] * cp[hh * W + wj];
                }
            }
        }
        conv[(size_t)n * HW + p] = acc;
        s1 = acc; s2 = (double)acc * acc;
    }
    s1 = block_sum_d(s1, red);
    s2 = block_sum_d(s2, red);
    if (threadIdx.x == 0) {
        part[2 * (size_t)blockIdx.x] = s1; part[2 * (size_t)blockIdx.x + 1] = s2;
    }
}

__global__ __launch_bounds__(256) void spatial_stats_kernel(const double* __restrict__ part, int nparts, double cnt,
                                                            float* __restrict__ running, float* __restrict__ stats,
                                                            int training, float momentum, float eps) {
    __shared__ double red[8];
    double s1 = 0.0, s2 = 0.0;
    if (training)
        for (int i = threadIdx.x; i < nparts; i += 256) { s1 += part[2 * i]; s2 += part[2 * i + 1]; }
    s1 = block_sum_d(s1, red);
    s2 = block_sum_d(s2, red);
    if (threadIdx.x == 0) {
        if (training) {
            const double mean = s1 / cnt;
            double var = s2 / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            stats[0] = (float)mean;
            stats[1] = (float)(1.0 / sqrt(var + (double)eps));
            const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            running[0] = (float)((1.0 - momentum) * running[0] + momentum * mean);
            running[1] = (float)((1.0 - momentum) * running[1] + momentum * unb);
        } else {
            stats[0] = running[0];
            stats[1] = 1.f / sqrtf(running[1] + eps);
        }
    }
}

__global__ __launch_bounds__(256) void spatial_apply_kernel(const float* __restrict__ x, const float* __restrict__ bn,
                                                            const float* __restrict__ stats, float* __restrict__ xhat,
                                                            float* __restrict__ scale, float* __restrict__ y, int N,
                                                            int C, int HW) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)N * HW) return;
    const int n = (int)(gid / HW), p = (int)(gid % HW);
    const size_t o = gid;
    const float xh = (xhat[o] - stats[0]) * stats[1];     // xhat holds the raw conv output on entry
    xhat[o] = xh;
    const float s = 1.f / (1.f + expf(-(xh * bn[0] + bn[1])));
    scale[o] = s;
    const float* xb = x + (size_t)n * C * HW + p;
    float* yb = y + (size_t)n * C * HW + p;
    for (int c = 0; c < C; ++c) yb[(size_t)c * HW] = xb[(size_t)c * HW] * s;
}

__global__ __launch_bounds__(256) void spatial_bwd_ds_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ scale, const float* __restrict__ xhat,
                                                             float* __restrict__ dbn, double* __restrict__ part, int N,
                                                             int C, int HW) {
    __shared__ double red[8];
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int n = (int)(gid / HW), p = (int)(gid % HW);
    double s1 = 0.0, s2 = 0.0;
    if (gid < (size_t)N * HW) {
        const float* xb = x + (size_t)n * C * HW + p;
        const float* gb = dy + (size_t)n * C * HW + p;
        float ds = 0.f;
        for (int c = 0; c < C; ++c) ds += gb[(size_t)c * HW] * xb[(size_t)c * HW];
        const size_t o = (size_t)n * HW + p;
        const float s = scale[o];
        const float d = ds * s * (1.f - s);
        dbn[o] = d;
        s1 = d; s2 = (double)d * xhat[o];
    }
    s1 = block_sum_d(s1, red);
    s2 = block_sum_d(s2, red);
    if (threadIdx.x == 0) {
        part[2 * (size_t)blockIdx.x] = s1; part[2 * (size_t)blockIdx.x + 1] = s2;
    }
}

// out2[0] = sum of part[2i] (d beta), out2[1] = sum of part[2i+1] (d gamma)
__global__ __launch_bounds__(256) void sum_pairs_kernel(const double* __restrict__ part, int nparts, float* __restrict__ out_gamma_beta) {
    __shared__ double red[8];
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) { s1 += part[2 * i]; s2 += part[2 * i + 1]; }
    s1 = block_sum_d(s1, red);
    s2 = block_sum_d(s2, red);
    if (threadIdx.x == 0) { out_gamma_beta[0] = (float)s2; out_gamma_beta[1] = (float)s1; }
}

__global__ __launch_bounds__(256) void spatial_bwd_dc_kernel(const float* __restrict__ dbn, const float* __restrict__ xhat,
                                                             const float* __restrict__ bn, const float* __restrict__ stats,
                                                             const float* __restrict__ dgb, float* __restrict__ dc,
                                                             size_t total, int training) {
    const float gamma = bn[0], invstd = stats[1];
    const float m1 = gamma * dgb[1] / (float)total;     // mean(dxhat)
    const float m2 = gamma * dgb[0] / (float)total;     // mean(dxhat * xhat)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float dxh = dbn[i] * gamma;
        dc[i] = training ? invstd * (dxh - m1 - xhat[i] * m2) : dxh * invstd;
    }
}

// per-tap weight gradient partials: grid (50, chunks)
__global__ __launch_bounds__(256) void spatial_bwd_dw_kernel(const float* __restrict__ dc, const float* __restrict__ comp,
                                                             double* __restrict__ part, int N, int H, int W, int per_chunk) {
    __shared__ double red[8];
    const int tap = blockIdx.x, ch = tap / 25, i = (tap % 25) / 5, j = tap % 5;
    const int HW = H * W;
    const size_t total = (size_t)N * HW;
    const size_t beg = (size_t)blockIdx.y * per_chunk;
    size_t end = beg + per_chunk;
    if (end > total) end = total;
    double s = 0.0;
    for (size_t q = beg + threadIdx.x; q < end; q += 256) {
        const int n = (int)(q / HW), p = (int)(q % HW);
        const int h = p / W + i - 2, w = p % W + j - 2;
        if (h < 0 || h >= H || w < 0 || w >= W) continue;
        s += (double)dc[q] * comp[((size_t)n * 2 + ch) * HW + h * W + w];
    }
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) part[(size_t)tap * gridDim.y + blockIdx.y] = s;
}

__global__ __launch_bounds__(64) void spatial_bwd_dw_final_kernel(const double* __restrict__ part, int chunks, float* __restrict__ dw) {
    const int tap = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < chunks; i += 64) s += part[(size_t)tap * chunks + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) dw[tap] = (float)s;
}

__global__ __launch_bounds__(256) void spatial_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ dc,
                                                             const float* __restrict__ w, const float* __restrict__ scale,
                                                             const int32_t* __restrict__ cargmax, float* __restrict__ dx,
                                                             int N, int C, int H, int W) {
    const int HW = H * W;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)N * HW) return;
    const int n = (int)(gid / HW), p = (int)(gid % HW);
    const int h = p / W, ww = p % W;
    const float* dcn = dc + (size_t)n * HW;
    float dmax = 0.f, dmean = 0.f;
    for (int i = 0; i < 5; ++i) {
        const int hh = h - i + 2;
        if (hh < 0 || hh >= H) continue;
        for (int j = 0; j < 5; ++j) {
            const int wj = ww - j + 2;
            if (wj < 0 || wj >= W) continue;
            const float d = dcn[hh * W + wj];
            dmax += w[(0 * 5 + i) * 5 + j] * d;
            dmean += w[(1 * 5 + i) * 5 + j] * d;
        }
    }
    dmean /= (float)C;
    const size_t o = (size_t)n * HW + p;
    const float s = scale[o];
    const int am = cargmax[o];
    const float* gb = dy + (size_t)n * C * HW + p;
    float* db = dx + (size_t)n * C * HW + p;
    for (int c = 0; c < C; ++c) db[(size_t)c * HW] = gb[(size_t)c * HW] * s + dmean + (c == am ? dmax : 0.f);
}

// ---- float4 forms of the four kernels that sweep x / dy (H*W a multiple of 4, 16-B aligned tensors): one thread = 4 consecutive
// pixels of a frame, 16 B per lane and channel, four channels in flight per loop trip.  The thread-per-pixel kernels above moved
// 4 B per lane with one load in flight and reached ~1.5 TB/s effective; they keep the ragged maps (7 x 7).
__global__ __launch_bounds__(256) void spatial_compress4_kernel(const float4* __restrict__ x, float4* __restrict__ comp,
                                                                int4* __restrict__ cargmax, int N, int C, int HW4) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)N * HW4) return;
    const int n = (int)(gid / HW4), q = (int)(gid % HW4);
    const float4* xb = x + (size_t)n * C * HW4 + q;
    float4 mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), sum = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 am = make_int4(0, 0, 0, 0);
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        const float4 v = xb[(size_t)c * HW4];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        if (v.x > mx.x) { mx.x = v.x; am.x = c; }
        if (v.y > mx.y) { mx.y = v.y; am.y = c; }
        if (v.z > mx.z) { mx.z = v.z; am.z = c; }
        if (v.w > mx.w) { mx.w = v.w; am.w = c; }
    }
    const float ic = 1.f / (float)C;
    comp[((size_t)n * 2 + 0) * HW4 + q] = mx;
    comp[((size_t)n * 2 + 1) * HW4 + q] = make_float4(sum.x / (float)C, sum.y / (float)C, sum.z / (float)C, sum.w / (float)C);
    (void)ic;
    cargmax[(size_t)n * HW4 + q] = am;
}

__global__ __launch_bounds__(256) void spatial_apply4_kernel(const float4* __restrict__ x, const float* __restrict__ bn,
                                                             const float* __restrict__ stats, float4* __restrict__ xhat,
                                                             float4* __restrict__ scale, float4* __restrict__ y, int N, int C,
                                                             int HW4) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)N * HW4) return;
    const int n = (int)(gid / HW4), q = (int)(gid % HW4);
    const float mean = stats[0], inv = stats[1], ga = bn[0], be = bn[1];
    float4 xh = xhat[gid];                                   // raw conv output on entry
    xh.x = (xh.x - mean) * inv; xh.y = (xh.y - mean) * inv; xh.z = (xh.z - mean) * inv; xh.w = (xh.w - mean) * inv;
    xhat[gid] = xh;
    float4 s;
    s.x = 1.f / (1.f + expf(-(xh.x * ga + be))); s.y = 1.f / (1.f + expf(-(xh.y * ga + be)));
    s.z = 1.f / (1.f + expf(-(xh.z * ga + be))); s.w = 1.f / (1.f + expf(-(xh.w * ga + be)));
    scale[gid] = s;
    const float4* xb = x + (size_t)n * C * HW4 + q;
    float4* yb = y + (size_t)n * C * HW4 + q;
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        float4 v = xb[(size_t)c * HW4];
        v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
        yb[(size_t)c * HW4] = v;
    }
}

__global__ __launch_bounds__(256) void spatial_bwd_ds4_kernel(const float4* __restrict__ dy, const float4* __restrict__ x,
                                                              const float4* __restrict__ scale, const float4* __restrict__ xhat,
                                                              float4* __restrict__ dbn, double* __restrict__ part, int N, int C,
                                                              int HW4) {
    __shared__ double red[8];
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    if (gid < (size_t)N * HW4) {
        const int n = (int)(gid / HW4), q = (int)(gid % HW4);
        const float4* xb = x + (size_t)n * C * HW4 + q;
        const float4* gb = dy + (size_t)n * C * HW4 + q;
        float4 ds = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int c = 0; c < C; ++c) {
            const float4 g = gb[(size_t)c * HW4], v = xb[(size_t)c * HW4];
            ds.x += g.x * v.x; ds.y += g.y * v.y; ds.z += g.z * v.z; ds.w += g.w * v.w;
        }
        const float4 s = scale[gid], xh = xhat[gid];
        float4 d;
        d.x = ds.x * s.x * (1.f - s.x); d.y = ds.y * s.y * (1.f - s.y); d.z = ds.z * s.z * (1.f - s.z); d.w = ds.w * s.w * (1.f - s.w);
        dbn[gid] = d;
        s1 = ((double)d.x + d.y) + ((double)d.z + d.w);
        s2 = ((double)d.x * xh.x + (double)d.y * xh.y) + ((double)d.z * xh.z + (double)d.w * xh.w);
    }
    s1 = block_sum_d(s1, red);
    s2 = block_sum_d(s2, red);
    if (threadIdx.x == 0) {
        part[2 * (size_t)blockIdx.x] = s1; part[2 * (size_t)blockIdx.x + 1] = s2;
    }
}

__global__ __launch_bounds__(256) void spatial_bwd_dx4_kernel(const float4* __restrict__ dy, const float* __restrict__ dc,
                                                              const float* __restrict__ w, const float4* __restrict__ scale,
                                                              const int4* __restrict__ cargmax, float4* __restrict__ dx, int N,
                                                              int C, int H, int W) {
    const int HW = H * W, HW4 = HW >> 2;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)N * HW4) return;
    const int n = (int)(gid / HW4), q = (int)(gid % HW4);
    const float* dcn = dc + (size_t)n * HW;
    float dmax[4], dmean[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int p = 4 * q + e, h = p / W, ww = p % W;
        float a = 0.f, b = 0.f;
        for (int i = 0; i < 5; ++i) {
            const int hh = h - i + 2;
            if (hh < 0 || hh >= H) continue;
            for (int j = 0; j < 5; ++j) {
                const int wj = ww - j + 2;
                if (wj < 0 || wj >= W) continue;
                const float d = dcn[hh * W + wj];
                a += w[(0 * 5 + i) * 5 + j] * d;
                b += w[(1 * 5 + i) * 5 + j] * d;
            }
        }
        dmax[e] = a; dmean[e] = b / (float)C;
    }
    const float4 s = scale[gid];
    const int4 am = cargmax[gid];
    const float4* gb = dy + (size_t)n * C * HW4 + q;
    float4* db = dx + (size_t)n * C * HW4 + q;
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
        const float4 g = gb[(size_t)c * HW4];
        float4 o;
        o.x = g.x * s.x + dmean[0] + (c == am.x ? dmax[0] : 0.f);
        o.y = g.y * s.y + dmean[1] + (c == am.y ? dmax[1] : 0.f);
        o.z = g.z * s.z + dmean[2] + (c == am.z ? dmax[2] : 0.f);
        o.w = g.w * s.w + dmean[3] + (c == am.w ? dmax[3] : 0.f);
        db[(size_t)c * HW4] = o;
    }
}

}  // namespace

extern "C" int m3t_sgemm(int, int, int, int, int, const float*, int, const float*, int, float*, int, const float*, int, int,
                         int, int, int, int, float*, size_t, int, void*);
extern "C" int m3t_colsum(const float*, int, int, int, float*, int, float*, size_t, void*);

extern "C" int m3t_cbam_channel_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                    float* y, float* pooled, int32_t* argmax, float* hidden, float* scale, int N, int C,
                                    int Cr, int HW, void* stream) {
    if (N <= 0) return 0;
    if (C <= 0 || Cr <= 0 || HW <= 0 || !x || !w1 || !b1 || !w2 || !b2 || !y || !pooled || !argmax || !hidden || !scale)
        return M3T_EINVAL;
    const size_t lds = (size_t)(3 * C + 2 * Cr) * sizeof(float);
    if (lds > 64 * 1024) return M3T_EINVAL;
    channel_fwd_kernel<<<N, 256, lds, (hipStream_t)stream>>>(x, w1, b1, w2, b2, y, pooled, argmax, hidden, scale, C, Cr, HW);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_cbam_channel_bwd(const float* dy, const float* x, const float* w1, const float* w2, const float* pooled,
                                    const int32_t* argmax, const float* hidden, const float* scale, float* dx, float* dw1,
                                    float* db1, float* dw2, float* db2, int N, int C, int Cr, int HW, float* ws,
                                    size_t ws_bytes, void* stream) {
    if (N <= 0) return 0;
    if (C <= 0 || Cr <= 0 || HW <= 0 || !dy || !x || !w1 || !w2 || !pooled || !argmax || !hidden || !scale || !dx || !dw1 ||
        !db1 || !dw2 || !db2 || !ws)
        return M3T_EINVAL;
    const size_t need = (size_t)N * (C + 3 * Cr);
    if (ws_bytes < need * sizeof(float)) return M3T_EINVAL;
    float* g_datt = ws;                         // [N,C]
    float* g_dh = g_datt + (size_t)N * C;       // [N,2,Cr]
    float* g_r = g_dh + (size_t)N * 2 * Cr;     // [N,Cr]
    float* rest = g_r + (size_t)N * Cr;
    const size_t rest_bytes = ws_bytes - need * sizeof(float);
    const size_t lds = (size_t)(3 * C + 2 * Cr) * sizeof(float);
    if (lds > 64 * 1024) return M3T_EINVAL;
    channel_bwd_kernel<<<N, 256, lds, (hipStream_t)stream>>>(dy, x, w1, w2, argmax, hidden, scale, dx, g_datt, g_dh, g_r, C,
                                                             Cr, HW);
    M3T_LAUNCH_CHECK();
    int rc;
    // dW2[C,Cr] = datt^T R ; db2 = 2 * colsum(datt)
    if ((rc = m3t_sgemm(1, 0, C, Cr, N, g_datt, C, g_r, Cr, dw2, Cr, nullptr, 0, 0, 0, 0, 0, 0, rest, rest_bytes, 0, stream))) return rc;
    if ((rc = m3t_colsum(g_datt, N, C, C, db2, 0, rest, rest_bytes, stream))) return rc;
    if ((rc = m3t_colsum(g_datt, N, C, C, db2, 1, rest, rest_bytes, stream))) return rc;
    // dW1[Cr,C] = dha^T avg + dhm^T max ; db1 = colsum(dha) + colsum(dhm)
    if ((rc = m3t_sgemm(1, 0, Cr, C, N, g_dh, 2 * Cr, pooled, 2 * C, dw1, C, nullptr, 0, 0, 0, 0, 0, 0, rest, rest_bytes, 0, stream))) return rc;
    if ((rc = m3t_sgemm(1, 0, Cr, C, N, g_dh + Cr, 2 * Cr, pooled + C, 2 * C, dw1, C, nullptr, 0, 1, 0, 0, 0, 0, rest, rest_bytes, 0, stream))) return rc;
    if ((rc = m3t_colsum(g_dh, N, Cr, 2 * Cr, db1, 0, rest, rest_bytes, stream))) return rc;
    if ((rc = m3t_colsum(g_dh + Cr, N, Cr, 2 * Cr, db1, 1, rest, rest_bytes, stream))) return rc;
    return 0;
}

extern "C" int m3t_cbam_spatial_fwd(const float* x, const float* conv_w, const float* bn, float* running, float* y,
                                    float* comp, int32_t* cargmax, float* xhat, float* stats, float* scale, int N, int C,
                                    int H, int W, int training, float momentum, float eps, float* ws, size_t ws_bytes,
                                    void* stream) {
    if (N <= 0) return 0;
    if (C <= 0 || H <= 0 || W <= 0 || !x || !conv_w || !bn || !running || !y || !comp || !cargmax || !xhat || !stats ||
        !scale || !ws)
        return M3T_EINVAL;
    const int HW = H * W;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (int)(((size_t)N * HW + 255) / 256);
    double* part = reinterpret_cast<double*>(ws);      // fp64 (sum, sumsq) per block
    if (ws_bytes < (size_t)nblk * 2 * sizeof(double) || ((uintptr_t)ws & 7) != 0) return M3T_EINVAL;
    const bool vec4 = (HW % 4 == 0) && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)comp | (uintptr_t)cargmax | (uintptr_t)xhat |
                                          (uintptr_t)scale) & 15) == 0);
    const int nblk4 = (int)(((size_t)N * (HW / 4) + 255) / 256);
    if (vec4) spatial_compress4_kernel<<<nblk4, 256, 0, s>>>(reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(comp),
                                                            reinterpret_cast<int4*>(cargmax), N, C, HW / 4);
    else spatial_compress_kernel<<<nblk, 256, 0, s>>>(x, comp, cargmax, N, C, HW);
    M3T_LAUNCH_CHECK();
    spatial_conv_kernel<<<nblk, 256, 0, s>>>(comp, conv_w, xhat, part, N, H, W);
    M3T_LAUNCH_CHECK();
    spatial_stats_kernel<<<1, 256, 0, s>>>(part, nblk, (double)N * HW, running, stats, training, momentum, eps);
    M3T_LAUNCH_CHECK();
    if (vec4) spatial_apply4_kernel<<<nblk4, 256, 0, s>>>(reinterpret_cast<const float4*>(x), bn, stats, reinterpret_cast<float4*>(xhat),
                                                         reinterpret_cast<float4*>(scale), reinterpret_cast<float4*>(y), N, C, HW / 4);
    else spatial_apply_kernel<<<nblk, 256, 0, s>>>(x, bn, stats, xhat, scale, y, N, C, HW);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_cbam_spatial_bwd(const float* dy, const float* x, const float* conv_w, const float* bn, const float* comp,
                                    const int32_t* cargmax, const float* xhat, const float* stats, const float* scale,
                                    float* dx, float* dconv_w, float* dbn, int N, int C, int H, int W, int training,
                                    float* ws, size_t ws_bytes, void* stream) {
    if (N <= 0) return 0;
    if (C <= 0 || H <= 0 || W <= 0 || !dy || !x || !conv_w || !bn || !comp || !cargmax || !xhat || !stats || !scale || !dx ||
        !dconv_w || !dbn || !ws)
        return M3T_EINVAL;
    const int HW = H * W;
    const size_t total = (size_t)N * HW;
    hipStream_t s = (hipStream_t)stream;
    const size_t nblk = (total + 255) / 256;
    int chunks = (int)((total + 16383) / 16384);
    if (chunks > 256) chunks = 256;
    const int per_chunk = (int)((total + chunks - 1) / chunks);
    // ws: [dbn total][dc total] floats, then doubles: part[2*nblk], dwpart[50*chunks]
    size_t off = 2 * total;
    off = (off + 1) & ~(size_t)1;
    const size_t need = off * sizeof(float) + (2 * nblk + (size_t)50 * chunks) * sizeof(double);
    if (ws_bytes < need || ((uintptr_t)ws & 7) != 0) return M3T_EINVAL;
    float* g_dbn = ws;
    float* g_dc = ws + total;
    double* part = reinterpret_cast<double*>(ws + off);
    double* dwpart = part + 2 * nblk;
    const bool vec4 = (HW % 4 == 0) && ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)cargmax | (uintptr_t)xhat |
                                          (uintptr_t)scale | (uintptr_t)g_dbn) & 15) == 0);
    const size_t nblk4 = (total / 4 + 255) / 256;
    if (vec4) spatial_bwd_ds4_kernel<<<(int)nblk4, 256, 0, s>>>(reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(x),
                                                                reinterpret_cast<const float4*>(scale), reinterpret_cast<const float4*>(xhat),
                                                                reinterpret_cast<float4*>(g_dbn), part, N, C, HW / 4);
    else spatial_bwd_ds_kernel<<<(int)nblk, 256, 0, s>>>(dy, x, scale, xhat, g_dbn, part, N, C, HW);
    M3T_LAUNCH_CHECK();
    sum_pairs_kernel<<<1, 256, 0, s>>>(part, (int)(vec4 ? nblk4 : nblk), dbn);
    M3T_LAUNCH_CHECK();
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    spatial_bwd_dc_kernel<<<blocks, 256, 0, s>>>(g_dbn, xhat, bn, stats, dbn, g_dc, total, training);
    M3T_LAUNCH_CHECK();
    spatial_bwd_dw_kernel<<<dim3(50, chunks), 256, 0, s>>>(g_dc, comp, dwpart, N, H, W, per_chunk);
    M3T_LAUNCH_CHECK();
    spatial_bwd_dw_final_kernel<<<50, 64, 0, s>>>(dwpart, chunks, dconv_w);
    M3T_LAUNCH_CHECK();
    if (vec4) spatial_bwd_dx4_kernel<<<(int)nblk4, 256, 0, s>>>(reinterpret_cast<const float4*>(dy), g_dc, conv_w,
                                                                reinterpret_cast<const float4*>(scale), reinterpret_cast<const int4*>(cargmax),
                                                                reinterpret_cast<float4*>(dx), N, C, H, W);
    else spatial_bwd_dx_kernel<<<(int)nblk, 256, 0, s>>>(dy, g_dc, conv_w, scale, cargmax, dx, N, C, H, W);
    M3T_LAUNCH_CHECK();
    return 0;
}
