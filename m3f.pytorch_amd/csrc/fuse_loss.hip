// HBM-bound kernels of the path: the attention-fusion reduction (models/att_fusion.py:21-25),
// the VA training loss with its gradient (models/model.py:132-182, models/utils.py:6-17) and
// the data-parallel gradient post-processing (train.py:35).  One wavefront per [B,T,C] frame
// row, float4 (16 B/lane) accesses, wavefront-shuffle reductions, no atomics.
#include "gru_common.h"
#include <cmath>
#include <cstdlib>
#include <mutex>

namespace {

__device__ __forceinline__ void fuse_weights(float sv, float sa, float& hv, float& ha, float& w0, float& w1) {
    hv = 1.f / (1.f + expf(-sv));
    ha = 1.f / (1.f + expf(-sa));
    const float m = fmaxf(hv, ha);
    const float ev = expf(hv - m), ea = expf(ha - m);
    const float inv = 1.f / (ev + ea);
    w0 = ev * inv;
    w1 = ea * inv;
}

__global__ __launch_bounds__(256) void att_fuse_fwd_kernel(const float* __restrict__ s_v, const float* __restrict__ s_a,
                                                           const float* __restrict__ x_v, const float* __restrict__ x_a,
                                                           float* __restrict__ f, int rows, int D, int vec) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float hv, ha, w0, w1;
    fuse_weights(s_v[row], s_a[row], hv, ha, w0, w1);
    const size_t base = (size_t)row * D;
    if (vec) {
        const float4* pv = reinterpret_cast<const float4*>(x_v + base);
        const float4* pa = reinterpret_cast<const float4*>(x_a + base);
        float4* pf = reinterpret_cast<float4*>(f + base);
        for (int i = lane; i < (D >> 2); i += 64) {
            const float4 v = pv[i], a = pa[i];
            pf[i] = make_float4(w0 * v.x + w1 * a.x, w0 * v.y + w1 * a.y, w0 * v.z + w1 * a.z, w0 * v.w + w1 * a.w);
        }
    } else {
        for (int i = lane; i < D; i += 64) f[base + i] = w0 * x_v[base + i] + w1 * x_a[base + i];
    }
}

__global__ __launch_bounds__(256) void att_fuse_bwd_kernel(const float* __restrict__ df, const float* __restrict__ s_v,
                                                           const float* __restrict__ s_a, const float* __restrict__ x_v,
                                                           const float* __restrict__ x_a, float* __restrict__ ds_v,
                                                           float* __restrict__ ds_a, float* __restrict__ dx_v,
                                                           float* __restrict__ dx_a, int rows, int D, int vec) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float hv, ha, w0, w1;
    fuse_weights(s_v[row], s_a[row], hv, ha, w0, w1);
    const size_t base = (size_t)row * D;
    float d0 = 0.f, d1 = 0.f;
    if (vec) {
        const float4* pg = reinterpret_cast<const float4*>(df + base);
        const float4* pv = reinterpret_cast<const float4*>(x_v + base);
        const float4* pa = reinterpret_cast<const float4*>(x_a + base);
        float4* qv = reinterpret_cast<float4*>(dx_v + base);
        float4* qa = reinterpret_cast<float4*>(dx_a + base);
        for (int i = lane; i < (D >> 2); i += 64) {
            const float4 g = pg[i], v = pv[i], a = pa[i];
            d0 += g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w;
            d1 += g.x * a.x + g.y * a.y + g.z * a.z + g.w * a.w;
            qv[i] = make_float4(w0 * g.x, w0 * g.y, w0 * g.z, w0 * g.w);
            qa[i] = make_float4(w1 * g.x, w1 * g.y, w1 * g.z, w1 * g.w);
        }
    } else {
        for (int i = lane; i < D; i += 64) {
            const float g = df[base + i];
            d0 += g * x_v[base + i];
            d1 += g * x_a[base + i];
            dx_v[base + i] = w0 * g;
            dx_a[base + i] = w1 * g;
        }
    }
    d0 = wave_sum(d0);
    d1 = wave_sum(d1);
    if (lane == 0) {
        const float dot = w0 * d0 + w1 * d1;
        ds_v[row] = w0 * (d0 - dot) * hv * (1.f - hv);
        ds_a[row] = w1 * (d1 - dot) * ha * (1.f - ha);
    }
}

// ------------------------------------------------------------------------------ VA loss
struct CccStats { float mx, mt, cov, vx, vt, den, ccc; };

__global__ __launch_bounds__(1024) void va_loss_kernel(const float* __restrict__ y, int rows, int C, int iv, int ia,
                                                       const float* __restrict__ val, const float* __restrict__ aro,
                                                       const int64_t* __restrict__ cls, const uint8_t* __restrict__ valid,
                                                       int n_expr, float wv, float wa, float expr_w, int use_mse,
                                                       float* __restrict__ out, float* __restrict__ dy) {
    __shared__ float red[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    const float invn = 1.f / (float)rows;
    // pass 1: means
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int i = tid; i < rows; i += nt) {
        s0 += y[(size_t)i * C + iv]; s1 += val[i];
        s2 += y[(size_t)i * C + ia]; s3 += aro[i];
    }
    const float mxv = block_sum(s0, red) * invn, mtv = block_sum(s1, red) * invn;
    const float mxa = block_sum(s2, red) * invn, mta = block_sum(s3, red) * invn;
    // pass 2: centred second moments (or squared errors)
    float cv = 0.f, xv = 0.f, tv = 0.f, ca = 0.f, xa = 0.f, ta = 0.f, ev = 0.f, ea = 0.f;
    for (int i = tid; i < rows; i += nt) {
        const float a = y[(size_t)i * C + iv], b = val[i];
        const float c = y[(size_t)i * C + ia], e = aro[i];
        cv += (a - mxv) * (b - mtv); xv += (a - mxv) * (a - mxv); tv += (b - mtv) * (b - mtv);
        ca += (c - mxa) * (e - mta); xa += (c - mxa) * (c - mxa); ta += (e - mta) * (e - mta);
        ev += (a - b) * (a - b); ea += (c - e) * (c - e);
    }
    cv = block_sum(cv, red); xv = block_sum(xv, red); tv = block_sum(tv, red);
    ca = block_sum(ca, red); xa = block_sum(xa, red); ta = block_sum(ta, red);
    ev = block_sum(ev, red); ea = block_sum(ea, red);
    const float nm1 = 1.f / (float)(rows > 1 ? rows - 1 : 1);
    const float covv = cv * invn, covA = ca * invn;
    const float denv = xv * nm1 + tv * nm1 + (mxv - mtv) * (mxv - mtv);
    const float dena = xa * nm1 + ta * nm1 + (mxa - mta) * (mxa - mta);
    const float cccv = 2.f * covv / denv, ccca = 2.f * covA / dena;
    // pass 3: masked cross entropy on the first n_expr logits
    float ce = 0.f, nvalid = 0.f, ncorrect = 0.f;
    if (n_expr > 0) {
        for (int i = tid; i < rows; i += nt) {
            if (!valid[i]) continue;
            const float* l = y + (size_t)i * C;
            float m = l[0];
            int am = 0;
            for (int k = 1; k < n_expr; ++k)
                if (l[k] > m) { m = l[k]; am = k; }
            float se = 0.f;
            for (int k = 0; k < n_expr; ++k) se += expf(l[k] - m);
            const int lab = (int)cls[i];
            ce += m + logf(se) - l[lab];
            nvalid += 1.f;
            ncorrect += (am == lab) ? 1.f : 0.f;
        }
        ce = block_sum(ce, red); nvalid = block_sum(nvalid, red); ncorrect = block_sum(ncorrect, red);
    }
    const float loss_e = ce * invn;
    const bool use_e = n_expr > 0 && nvalid > 0.f;
    const float lv = wv == 0.f ? 0.f : (use_mse ? ev * invn : 1.f - cccv);
    const float la = wa == 0.f ? 0.f : (use_mse ? ea * invn : 1.f - ccca);
    if (tid == 0) {
        out[0] = wv * lv + wa * la + (use_e ? expr_w * loss_e : 0.f);
        out[1] = lv; out[2] = la; out[3] = loss_e; out[4] = nvalid; out[5] = ncorrect; out[6] = cccv; out[7] = ccca;
    }
    // gradient
    for (int i = tid; i < rows; i += nt) {
        float* g = dy + (size_t)i * C;
        const float* l = y + (size_t)i * C;
        for (int k = 0; k < C; ++k) g[k] = 0.f;
        if (use_e && valid[i]) {
            float m = l[0];
            for (int k = 1; k < n_expr; ++k) m = fmaxf(m, l[k]);
            float se = 0.f;
            for (int k = 0; k < n_expr; ++k) se += expf(l[k] - m);
            const float sc = expr_w * invn;
            const int lab = (int)cls[i];
            for (int k = 0; k < n_expr; ++k) g[k] = sc * (expf(l[k] - m) / se - (k == lab ? 1.f : 0.f));
        }
        const float a = l[iv], b = val[i], c = l[ia], e = aro[i];
        float gv, ga;
        if (use_mse) {
            gv = 2.f * (a - b) * invn;
            ga = 2.f * (c - e) * invn;
        } else {
            const float dcv = 2.f * (b - mtv) * invn / denv -
                              (2.f * covv / (denv * denv)) * (2.f * (a - mxv) * nm1 + 2.f * (mxv - mtv) * invn);
            const float dca = 2.f * (e - mta) * invn / dena -
                              (2.f * covA / (dena * dena)) * (2.f * (c - mxa) * nm1 + 2.f * (mxa - mta) * invn);
            gv = -dcv;
            ga = -dca;
        }
        if (wv != 0.f) g[iv] += wv * gv;
        if (wa != 0.f) g[ia] += wa * ga;
    }
}

// The same loss as three short grid-wide launches for B*T in the thousands (one workgroup took 141 us on 9600 rows,
// un-overlappable between forward and backward).  256 rows per block, one row per thread; per-block partial sums go to
// ws and are reduced in a fixed order by EVERY block of the next launch (deterministic, no atomics).
//   part A [nb][8]: sum y_v, sum val, sum y_a, sum aro, CE sum, n_valid, n_correct, -
//   part B [nb][8]: centred co-moments cv xv tv ca xa ta and squared errors ev ea
constexpr int LB = 256;

__device__ __forceinline__ void reduce_parts(const float* __restrict__ part, int nb, float* tot /* LDS [8] */) {
    // threads 0..7 each add one column over the blocks in block order
    if (threadIdx.x < 8) {
        float s = 0.f;
        for (int b = 0; b < nb; ++b) s += part[(size_t)b * 8 + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(LB) void va_loss_sums_kernel(const float* __restrict__ y, int rows, int C, int iv, int ia,
                                                          const float* __restrict__ val, const float* __restrict__ aro,
                                                          const int64_t* __restrict__ cls, const uint8_t* __restrict__ valid,
                                                          int n_expr, float* __restrict__ partA) {
    __shared__ float red[16];
    const int i = blockIdx.x * LB + threadIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, ce = 0.f, nvalid = 0.f, ncorrect = 0.f;
    if (i < rows) {
        const float* l = y + (size_t)i * C;
        s0 = l[iv]; s1 = val[i]; s2 = l[ia]; s3 = aro[i];
        if (n_expr > 0 && valid[i]) {
            float m = l[0];
            int am = 0;
            for (int k = 1; k < n_expr; ++k)
                if (l[k] > m) { m = l[k]; am = k; }
            float se = 0.f;
            for (int k = 0; k < n_expr; ++k) se += expf(l[k] - m);
            const int lab = (int)cls[i];
            ce = m + logf(se) - l[lab];
            nvalid = 1.f;
            ncorrect = (am == lab) ? 1.f : 0.f;
        }
    }
    s0 = block_sum(s0, red); s1 = block_sum(s1, red); s2 = block_sum(s2, red); s3 = block_sum(s3, red);
    ce = block_sum(ce, red); nvalid = block_sum(nvalid, red); ncorrect = block_sum(ncorrect, red);
    if (threadIdx.x == 0) {
        float* q = partA + (size_t)blockIdx.x * 8;
        q[0] = s0; q[1] = s1; q[2] = s2; q[3] = s3; q[4] = ce; q[5] = nvalid; q[6] = ncorrect; q[7] = 0.f;
    }
}

__global__ __launch_bounds__(LB) void va_loss_moments_kernel(const float* __restrict__ y, int rows, int C, int iv, int ia,
                                                             const float* __restrict__ val, const float* __restrict__ aro,
                                                             const float* __restrict__ partA, float* __restrict__ partB) {
    __shared__ float red[16];
    __shared__ float tot[8];
    reduce_parts(partA, gridDim.x, tot);
    const float invn = 1.f / (float)rows;
    const float mxv = tot[0] * invn, mtv = tot[1] * invn, mxa = tot[2] * invn, mta = tot[3] * invn;
    const int i = blockIdx.x * LB + threadIdx.x;
    float cv = 0.f, xv = 0.f, tv = 0.f, ca = 0.f, xa = 0.f, ta = 0.f, ev = 0.f, ea = 0.f;
    if (i < rows) {
        const float a = y[(size_t)i * C + iv], b = val[i];
        const float c = y[(size_t)i * C + ia], e = aro[i];
        cv = (a - mxv) * (b - mtv); xv = (a - mxv) * (a - mxv); tv = (b - mtv) * (b - mtv);
        ca = (c - mxa) * (e - mta); xa = (c - mxa) * (c - mxa); ta = (e - mta) * (e - mta);
        ev = (a - b) * (a - b); ea = (c - e) * (c - e);
    }
    cv = block_sum(cv, red); xv = block_sum(xv, red); tv = block_sum(tv, red);
    ca = block_sum(ca, red); xa = block_sum(xa, red); ta = block_sum(ta, red);
    ev = block_sum(ev, red); ea = block_sum(ea, red);
    if (threadIdx.x == 0) {
        float* q = partB + (size_t)blockIdx.x * 8;
        q[0] = cv; q[1] = xv; q[2] = tv; q[3] = ca; q[4] = xa; q[5] = ta; q[6] = ev; q[7] = ea;
    }
}

__global__ __launch_bounds__(LB) void va_loss_grad_kernel(const float* __restrict__ y, int rows, int C, int iv, int ia,
                                                          const float* __restrict__ val, const float* __restrict__ aro,
                                                          const int64_t* __restrict__ cls, const uint8_t* __restrict__ valid,
                                                          int n_expr, float wv, float wa, float expr_w, int use_mse,
                                                          const float* __restrict__ partA, const float* __restrict__ partB,
                                                          float* __restrict__ out, float* __restrict__ dy) {
    __shared__ float ta_[8], tb_[8];
    reduce_parts(partA, gridDim.x, ta_);
    reduce_parts(partB, gridDim.x, tb_);
    const float invn = 1.f / (float)rows;
    const float mxv = ta_[0] * invn, mtv = ta_[1] * invn, mxa = ta_[2] * invn, mta = ta_[3] * invn;
    const float nm1 = 1.f / (float)(rows > 1 ? rows - 1 : 1);
    const float covv = tb_[0] * invn, covA = tb_[3] * invn;
    const float denv = tb_[1] * nm1 + tb_[2] * nm1 + (mxv - mtv) * (mxv - mtv);
    const float dena = tb_[4] * nm1 + tb_[5] * nm1 + (mxa - mta) * (mxa - mta);
    const float cccv = 2.f * covv / denv, ccca = 2.f * covA / dena;
    const float loss_e = ta_[4] * invn, nvalid = ta_[5];
    const bool use_e = n_expr > 0 && nvalid > 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float lv = wv == 0.f ? 0.f : (use_mse ? tb_[6] * invn : 1.f - cccv);
        const float la = wa == 0.f ? 0.f : (use_mse ? tb_[7] * invn : 1.f - ccca);
        out[0] = wv * lv + wa * la + (use_e ? expr_w * loss_e : 0.f);
        out[1] = lv; out[2] = la; out[3] = loss_e; out[4] = nvalid; out[5] = ta_[6]; out[6] = cccv; out[7] = ccca;
    }
    const int i = blockIdx.x * LB + threadIdx.x;
    if (i >= rows) return;
    float* g = dy + (size_t)i * C;
    const float* l = y + (size_t)i * C;
    for (int k = 0; k < C; ++k) g[k] = 0.f;
    if (use_e && valid[i]) {
        float m = l[0];
        for (int k = 1; k < n_expr; ++k) m = fmaxf(m, l[k]);
        float se = 0.f;
        for (int k = 0; k < n_expr; ++k) se += expf(l[k] - m);
        const float sc = expr_w * invn;
        const int lab = (int)cls[i];
        for (int k = 0; k < n_expr; ++k) g[k] = sc * (expf(l[k] - m) / se - (k == lab ? 1.f : 0.f));
    }
    const float a = l[iv], b = val[i], c = l[ia], e = aro[i];
    float gv, ga;
    if (use_mse) {
        gv = 2.f * (a - b) * invn;
        ga = 2.f * (c - e) * invn;
    } else {
        const float dcv = 2.f * (b - mtv) * invn / denv -
                          (2.f * covv / (denv * denv)) * (2.f * (a - mxv) * nm1 + 2.f * (mxv - mtv) * invn);
        const float dca = 2.f * (e - mta) * invn / dena -
                          (2.f * covA / (dena * dena)) * (2.f * (c - mxa) * nm1 + 2.f * (mxa - mta) * invn);
        gv = -dcv;
        ga = -dca;
    }
    if (wv != 0.f) g[iv] += wv * gv;
    if (wa != 0.f) g[ia] += wa * ga;
}

// Round 6: the same loss as ONE launch (three launches for 1.9 MB are three launch latencies on the chain between forward
// and backward).  One sweep gathers RAW moments in fp64 (sum y, sum t, sum y^2, sum t^2, sum y t, sum (y - t)^2 per output, the masked CE terms):
// every centred statistic of the two-pass form follows from them without cancellation trouble at fp64 (9 600 values of magnitude ~1), so the
// grid needs ONE meeting point instead of two: each block publishes its partials (agent-scope release + ticket), waits -- bounded -- until
// all have, sums the partials of ALL blocks in block order (deterministic, identical in every block) and writes its rows of dL/dy.  The two
// ticket words live in a library-owned allocation per device; the last block to leave resets them.  Blocks must be co-resident: the host
// takes this path for at most 128 blocks (32 768 rows), else the three launches above.
struct VaSync { unsigned arrive, depart; };
constexpr int VP = 16;                                // doubles per block: 0-5 valence (y, t, yy, tt, yt, (y-t)^2), 6-11 arousal, 12 CE, 13 n_valid, 14 n_correct

__global__ __launch_bounds__(LB) void va_loss_fused_kernel(const float* __restrict__ y, int rows, int C, int iv, int ia,
                                                           const float* __restrict__ val, const float* __restrict__ aro,
                                                           const int64_t* __restrict__ cls, const uint8_t* __restrict__ valid,
                                                           int n_expr, float wv, float wa, float expr_w, int use_mse,
                                                           double* __restrict__ part, VaSync* __restrict__ sync, int spin_limit,
                                                           float* __restrict__ out, float* __restrict__ dy) {
    __shared__ double wred[LB / 64][VP];
    __shared__ double tot[VP];
    const int i = blockIdx.x * LB + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nb = gridDim.x;
    double v[VP];
#pragma unroll
    for (int k = 0; k < VP; ++k) v[k] = 0.0;
    float a = 0.f, b = 0.f, c = 0.f, e = 0.f;
    if (i < rows) {
        const float* l = y + (size_t)i * C;
        a = l[iv]; b = val[i]; c = l[ia]; e = aro[i];
        v[0] = a; v[1] = b; v[2] = (double)a * a; v[3] = (double)b * b; v[4] = (double)a * b; v[5] = ((double)a - b) * ((double)a - b);
        v[6] = c; v[7] = e; v[8] = (double)c * c; v[9] = (double)e * e; v[10] = (double)c * e; v[11] = ((double)c - e) * ((double)c - e);
        if (n_expr > 0 && valid[i]) {
            float m = l[0];
            int am = 0;
            for (int k = 1; k < n_expr; ++k)
                if (l[k] > m) { m = l[k]; am = k; }
            float se = 0.f;
            for (int k = 0; k < n_expr; ++k) se += expf(l[k] - m);
            const int lab = (int)cls[i];
            v[12] = m + logf(se) - l[lab];
            v[13] = 1.0;
            v[14] = (am == lab) ? 1.0 : 0.0;
        }
    }
#pragma unroll
    for (int k = 0; k < 15; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
        if (lane == 0) wred[wave][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < VP) {
        double s = 0.0;
        if (threadIdx.x < 15)
            for (int w = 0; w < LB / 64; ++w) s += wred[w][threadIdx.x];
        part[(size_t)blockIdx.x * VP + threadIdx.x] = s;
    }
    // publish: every storing thread's stores done, the block's barrier, then ONE agent-scope release and the ticket (guide, Guideline 16)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&sync->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&sync->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nb) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > spin_limit) break;              // (bounded like every wait of the library: the loss comes out NaN below)
        }
        tot[VP - 1] = spins > spin_limit ? 1.0 : 0.0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const bool gave_up = tot[VP - 1] != 0.0;
    __syncthreads();
    if (threadIdx.x < 15) {
        double s = 0.0;
        for (int bk = 0; bk < nb; ++bk) s += part[(size_t)bk * VP + threadIdx.x];       // block order: the same sum in every block
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    const double n = (double)rows, invn_d = 1.0 / n, nm1_d = 1.0 / (double)(rows > 1 ? rows - 1 : 1);
    const double mxv_d = tot[0] * invn_d, mtv_d = tot[1] * invn_d, mxa_d = tot[6] * invn_d, mta_d = tot[7] * invn_d;
    const double cv = tot[4] - n * mxv_d * mtv_d, xv = tot[2] - n * mxv_d * mxv_d, tv = tot[3] - n * mtv_d * mtv_d;
    const double ca = tot[10] - n * mxa_d * mta_d, xa = tot[8] - n * mxa_d * mxa_d, ta = tot[9] - n * mta_d * mta_d;
    const float invn = (float)invn_d, nm1 = (float)nm1_d;
    const float mxv = (float)mxv_d, mtv = (float)mtv_d, mxa = (float)mxa_d, mta = (float)mta_d;
    const float covv = (float)(cv * invn_d), covA = (float)(ca * invn_d);
    const float denv = (float)(xv * nm1_d + tv * nm1_d + (mxv_d - mtv_d) * (mxv_d - mtv_d));
    const float dena = (float)(xa * nm1_d + ta * nm1_d + (mxa_d - mta_d) * (mxa_d - mta_d));
    const float cccv = 2.f * covv / denv, ccca = 2.f * covA / dena;
    const float loss_e = (float)(tot[12] * invn_d), nvalid = (float)tot[13];
    const bool use_e = n_expr > 0 && nvalid > 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float lv = wv == 0.f ? 0.f : (use_mse ? (float)(tot[5] * invn_d) : 1.f - cccv);
        const float la = wa == 0.f ? 0.f : (use_mse ? (float)(tot[11] * invn_d) : 1.f - ccca);
        out[0] = gave_up ? __builtin_nanf("") : wv * lv + wa * la + (use_e ? expr_w * loss_e : 0.f);
        out[1] = lv; out[2] = la; out[3] = loss_e; out[4] = nvalid; out[5] = (float)tot[14]; out[6] = cccv; out[7] = ccca;
    }
    if (i < rows) {
        float* g = dy + (size_t)i * C;
        const float* l = y + (size_t)i * C;
        for (int k = 0; k < C; ++k) g[k] = 0.f;
        if (use_e && valid[i]) {
            float m = l[0];
            for (int k = 1; k < n_expr; ++k) m = fmaxf(m, l[k]);
            float se = 0.f;
            for (int k = 0; k < n_expr; ++k) se += expf(l[k] - m);
            const float sc = expr_w * invn;
            const int lab = (int)cls[i];
            for (int k = 0; k < n_expr; ++k) g[k] = sc * (expf(l[k] - m) / se - (k == lab ? 1.f : 0.f));
        }
        float gv, ga;
        if (use_mse) {
            gv = 2.f * (a - b) * invn;
            ga = 2.f * (c - e) * invn;
        } else {
            const float dcv = 2.f * (b - mtv) * invn / denv -
                              (2.f * covv / (denv * denv)) * (2.f * (a - mxv) * nm1 + 2.f * (mxv - mtv) * invn);
            const float dca = 2.f * (e - mta) * invn / dena -
                              (2.f * covA / (dena * dena)) * (2.f * (c - mxa) * nm1 + 2.f * (mxa - mta) * invn);
            gv = -dcv;
            ga = -dca;
        }
        if (wv != 0.f) g[iv] += wv * gv;
        if (wa != 0.f) g[ia] += wa * ga;
    }
    // leave: the last block out resets both words for the next launch (stream order: a kernel boundary lies in between)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned d = __hip_atomic_fetch_add(&sync->depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == (unsigned)nb - 1u) {
            __hip_atomic_store(&sync->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sync->depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static VaSync* va_sync_words() {                      // two zeroed words per device, allocated on first use and never freed
    static VaSync* words[32] = {};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!words[dev]) {
        VaSync* p = nullptr;
        if (hipMalloc(&p, sizeof(VaSync)) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, sizeof(VaSync)) != hipSuccess) { (void)hipFree(p); return nullptr; }
        words[dev] = p;
    }
    return words[dev];
}

// ------------------------------------------------------------------------------ DDP helpers
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ part,
                                                            const unsigned* __restrict__ scan_err) {
    __shared__ float red[16];
    // ONE read of the host-mapped scan error word per call (a PCIe round trip: 1024 blocks reading it took 115 us), copied into
    // device memory behind the partials for norm_scale_kernel
    if (blockIdx.x == 0 && threadIdx.x == 0)
        reinterpret_cast<unsigned*>(part)[gridDim.x] = scan_err ? __hip_atomic_load(scan_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
    float s = 0.f;
    const size_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void norm_scale_kernel(float* __restrict__ g, size_t n, const float* __restrict__ part,
                                                         int nparts, float inv_world, float max_norm,
                                                         float* __restrict__ norm_out) {
    __shared__ float red[16];
    __shared__ unsigned bad_s;
    // a persistent GRU scan that gave up (gru_persist.hip) left garbage in these gradients: the host-mapped error word was read
    // by sumsq_partial_kernel, in stream order behind that scan -- no host synchronisation -- and turns the step into a no-op:
    // gradients zeroed, norm = NaN (m3t_adam_step / m3t_sgd_step skip on a non-finite guard)
    if (threadIdx.x == 0) bad_s = reinterpret_cast<const unsigned*>(part)[nparts];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += part[i];
    s = block_sum(s, red);                               // (its barriers also publish bad_s)
    const bool bad = bad_s != 0u;
    const float norm = bad ? __builtin_nanf("") : sqrtf(s) * inv_world;            // norm of the averaged gradient
    float coef = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.f;   // max_norm <= 0: no clipping
    coef = coef < 1.f ? coef : 1.f;
    const float sc = bad ? 0.f : inv_world * coef;
    if (blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = norm;
    if (bad) {
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) g[i] = 0.f;
        return;
    }
    if (sc == 1.f) return;
    const size_t n4 = n >> 2;
    float4* g4 = reinterpret_cast<float4*>(g);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = g4[i];
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        g4[i] = v;
    }
    if (blockIdx.x == 0)
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) g[i] *= sc;
}

// ------------------------------------------------------------------------------ optimizers (flat buffers)
// torch.optim.Adam semantics (reference models/model.py:388-390: Adam(lr, weight_decay=1e-4), L2 added to the gradient)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float inv_bc1, float inv_sqrt_bc2, const float* __restrict__ guard) {
    if (guard && !isfinite(guard[0])) return;          // skipped step: gradients came from a failed scan or overflowed
    const size_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg += wd * pp;
        mm = b1 * mm + (1.f - b1) * gg;
        vv = b2 * vv + (1.f - b2) * gg * gg;
        pp -= lr * inv_bc1 * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
    };
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0)
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) upd(p[i], g[i], m[i], v[i]);
}

// torch.optim.SGD(momentum, weight_decay) semantics (reference models/model.py:392-394), dampening 0, no nesterov
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  size_t n, float lr, float momentum, float wd, int first,
                                                  const float* __restrict__ guard) {
    if (guard && !isfinite(guard[0])) return;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gg = g[i] + wd * p[i];
        const float b = first ? gg : momentum * buf[i] + gg;
        buf[i] = b;
        p[i] -= lr * b;
    }
}

}  // namespace

extern "C" int m3t_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, const float* guard, void* stream) {
    if (n == 0) return 0;
    if (!p || !g || !m || !v || step < 1 || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16) != 0)
        return M3T_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    adam_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1),
                                                         (float)(1.0 / sqrt(bc2)), guard);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_sgd_step(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float weight_decay,
                            int step, const float* guard, void* stream) {
    if (n == 0) return 0;
    if (!p || !g || !buf || step < 1) return M3T_EINVAL;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    sgd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(p, g, buf, n, lr, momentum, weight_decay, step == 1, guard);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_att_fuse_fwd(const float* s_v, const float* s_a, const float* x_v, const float* x_a, float* f,
                                int rows, int D, void* stream) {
    if (rows <= 0 || D <= 0) return 0;
    if (!s_v || !s_a || !x_v || !x_a || !f) return M3T_EINVAL;
    const int vec = (D % 4 == 0) && (((uintptr_t)x_v | (uintptr_t)x_a | (uintptr_t)f) % 16 == 0);
    att_fuse_fwd_kernel<<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>(s_v, s_a, x_v, x_a, f, rows, D, vec);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_att_fuse_bwd(const float* df, const float* s_v, const float* s_a, const float* x_v, const float* x_a,
                                float* ds_v, float* ds_a, float* dx_v, float* dx_a, int rows, int D, void* stream) {
    if (rows <= 0 || D <= 0) return 0;
    if (!df || !s_v || !s_a || !x_v || !x_a || !ds_v || !ds_a || !dx_v || !dx_a) return M3T_EINVAL;
    const int vec = (D % 4 == 0) &&
                    (((uintptr_t)x_v | (uintptr_t)x_a | (uintptr_t)df | (uintptr_t)dx_v | (uintptr_t)dx_a) % 16 == 0);
    att_fuse_bwd_kernel<<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>(df, s_v, s_a, x_v, x_a, ds_v, ds_a, dx_v, dx_a,
                                                                         rows, D, vec);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t m3t_va_loss_ws_bytes(int rows) { return rows > 0 ? (size_t)cdiv(rows, LB) * 16 * sizeof(float) : 0; }

extern "C" int m3t_va_loss(const float* y_hat, int rows, int C, int iv, int ia, const float* valence,
                           const float* arousal, const int64_t* class_expr, const uint8_t* expr_valid, int n_expr,
                           float w_v, float w_a, float expr_w, int use_mse, float* out_scalars, float* dy,
                           float* ws, size_t ws_bytes, void* stream) {
    if (rows <= 0 || C <= 0 || iv < 0 || ia < 0 || iv >= C || ia >= C || n_expr > C) return M3T_EINVAL;
    if (!y_hat || !valence || !arousal || !out_scalars || !dy) return M3T_EINVAL;
    if (n_expr > 0 && (!class_expr || !expr_valid)) return M3T_EINVAL;
    static const bool fused_on = !(getenv("M3T_VA_LOSS_FUSED") && getenv("M3T_VA_LOSS_FUSED")[0] == '0');
    if (rows > 1024 && ws && ws_bytes >= m3t_va_loss_ws_bytes(rows) && cdiv(rows, LB) <= 128 && fused_on && ((uintptr_t)ws % 8) == 0) {
        // one launch (round 6): the blocks meet once inside the kernel; <= 128 blocks so that all are resident whatever else runs
        VaSync* sync = va_sync_words();
        if (sync) {
            const int nb = cdiv(rows, LB);
            va_loss_fused_kernel<<<nb, LB, 0, (hipStream_t)stream>>>(y_hat, rows, C, iv, ia, valence, arousal, class_expr, expr_valid, n_expr, w_v,
                                                                     w_a, expr_w, use_mse, reinterpret_cast<double*>(ws), sync, 1 << 22,
                                                                     out_scalars, dy);
            M3T_LAUNCH_CHECK();
            return 0;
        }
    }
    if (rows > 1024 && ws && ws_bytes >= m3t_va_loss_ws_bytes(rows)) {
        const int nb = cdiv(rows, LB);
        float* partA = ws;
        float* partB = ws + (size_t)nb * 8;
        hipStream_t s = (hipStream_t)stream;
        va_loss_sums_kernel<<<nb, LB, 0, s>>>(y_hat, rows, C, iv, ia, valence, arousal, class_expr, expr_valid, n_expr, partA);
        va_loss_moments_kernel<<<nb, LB, 0, s>>>(y_hat, rows, C, iv, ia, valence, arousal, partA, partB);
        va_loss_grad_kernel<<<nb, LB, 0, s>>>(y_hat, rows, C, iv, ia, valence, arousal, class_expr, expr_valid, n_expr, w_v, w_a,
                                             expr_w, use_mse, partA, partB, out_scalars, dy);
        M3T_LAUNCH_CHECK();
        return 0;
    }
    va_loss_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(y_hat, rows, C, iv, ia, valence, arousal, class_expr, expr_valid,
                                                        n_expr, w_v, w_a, expr_w, use_mse, out_scalars, dy);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_grad_norm_scale(float* flat, size_t n, float inv_world, float max_norm, float* norm_out, float* ws,
                                   size_t ws_bytes, void* stream) {
    if (n == 0) return 0;
    if (!flat || !norm_out || !ws || ((uintptr_t)flat % 16) != 0) return M3T_EINVAL;
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    if (ws_bytes < (size_t)(blocks + 1) * sizeof(float)) return M3T_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    sumsq_partial_kernel<<<blocks, 256, 0, s>>>(flat, n, ws, m3t_gru::persist_error_word_dev());
    M3T_LAUNCH_CHECK();
    norm_scale_kernel<<<blocks, 256, 0, s>>>(flat, n, ws, blocks, inv_world, max_norm, norm_out);
    M3T_LAUNCH_CHECK();
    return 0;
}

// ---- a dead scan on ONE rank must stop EVERY rank (the all-reduce spreads its garbage) --------------------------------
// m3t_grad_poison runs BEFORE the gradient all-reduce: if this process's sticky scan-error flag is set it writes NaN into
// flat[0] (SUM carries it to every rank: every rank's clip norm is NaN, every fused optimizer step skips itself) and 1 into
// the `dead` slot -- one float of padding that rides in the same all-reduce.  m3t_grad_dead_check runs AFTER the
// all-reduce: a non-zero slot (some rank died) raises THIS process's sticky flag too, so every rank's next host poll raises
// instead of the healthy ranks blocking in the next collective until its timeout.  One thread each, no synchronisation.
namespace {
__global__ void grad_poison_kernel(float* flat, float* dead, const unsigned* sticky) {
    const bool bad = sticky && __hip_atomic_load(sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
    if (bad) flat[0] = __builtin_nanf("");
    if (dead) dead[0] = bad ? 1.f : 0.f;
}
__global__ void grad_dead_check_kernel(const float* dead, unsigned* sticky) {
    if (dead[0] != 0.f) {                    // (NaN compares unequal to zero as well)
        __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
}  // namespace

extern "C" int m3t_grad_poison(float* flat, float* dead, void* stream) {
    if (!flat) return M3T_EINVAL;
    grad_poison_kernel<<<1, 1, 0, (hipStream_t)stream>>>(flat, dead, m3t_gru::persist_error_word_dev());
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_grad_dead_check(const float* dead, void* stream) {
    if (!dead) return M3T_EINVAL;
    unsigned* sticky = m3t_gru::persist_error_word_dev(true);
    if (!sticky) return M3T_EINVAL;
    grad_dead_check_kernel<<<1, 1, 0, (hipStream_t)stream>>>(dead, sticky);
    M3T_LAUNCH_CHECK();
    return 0;
}
