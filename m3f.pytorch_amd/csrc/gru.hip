// BiGRU recurrence for gfx950.  The T-step dependency chain is cut at every time step
// (one launch per step: an RNN step is an all-to-all seam -- every hidden unit of h_t needs
// all of h_{t-1} -- and on MI355X a kernel boundary (~1.5 us) is cheaper than an in-launch
// cross-CU exchange, MI355X_MICROARCH.md price list).  What one launch does:
//   * it advances ALL independent direction-scans of the level at once (both directions,
//     every independent stack), so the chip sees sum_s (H_s/16) x ceil(B/32) workgroups;
//   * a workgroup owns 16 hidden units of one scan: the three gate columns r,z,n of those
//     units (a [32 x 48] tile of h_{t-1} W_hh^T, K = H) on fp32 MFMA 16x16x4 (exact fp32),
//     K split over its 4 waves, operands loaded straight to registers as float4 along K
//     (W_hh stays L2/MALL-resident across the T launches), fixed-order LDS reduction, then
//     the gate math for its own units -- so gates never travel through HBM un-fused.
// Backward runs the same structure in reverse time: dh_{t} = dout_t + z_{t+1} dh_{t+1}
// + dgh_{t+1} W_hh, with the matmul of step t+1 and the gate derivative of step t fused
// in one launch; dW_hh / dW_ih / dx are left to big GEMMs after the scan.
#include "common.h"

namespace {

constexpr int UB = 16;   // hidden units per workgroup
constexpr int RB = 32;   // batch rows per workgroup

struct FwdGroup {
    m3t_gru_fwd_desc d[M3T_MAX_SCANS];
    int blk_start[M3T_MAX_SCANS + 1];
    int n;
};
struct BwdGroup {
    m3t_gru_bwd_desc d[M3T_MAX_SCANS];
    int blk_start[M3T_MAX_SCANS + 1];
    int n;
};

__device__ __forceinline__ float4 ld4(const float* __restrict__ p, int nvalid, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid >= 4 && vec) return *reinterpret_cast<const float4*>(p);
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
    return v;
}

// acc[rt][ct] += A[rows r0+16rt.., K] * Bt[cols.., K]^T on MFMA 16x16x4, this wave taking
// the 16-wide k chunks c = wave, wave+4, ...  A row stride lda, Bt row stride ldb (both K-contiguous).
template <int CT>
__device__ __forceinline__ void wave_mma(const float* __restrict__ A, size_t lda, const bool (&arow_ok)[2],
                                         const size_t (&arow_off)[2], const float* __restrict__ Bt, size_t ldb,
                                         const bool (&brow_ok)[CT], const size_t (&brow_off)[CT], int K, bool vecA,
                                         bool vecB, int wave, int lane, f32x4 (&acc)[2][CT]) {
    const int kq = (lane >> 4) * 4;
    const int nchunks = (K + 15) >> 4;
    for (int c = wave; c < nchunks; c += 4) {
        const int kk = c * 16 + kq;
        const int nv = K - kk;
        float4 a[2], b[CT];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
            a[rt] = (arow_ok[rt] && nv > 0) ? ld4(A + arow_off[rt] + kk, nv, vecA) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            b[ct] = (brow_ok[ct] && nv > 0) ? ld4(Bt + brow_off[ct] + kk, nv, vecB) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].x, b[ct].x, acc[rt][ct], 0, 0, 0);
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].y, b[ct].y, acc[rt][ct], 0, 0, 0);
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].z, b[ct].z, acc[rt][ct], 0, 0, 0);
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].w, b[ct].w, acc[rt][ct], 0, 0, 0);
            }
    }
}

__global__ __launch_bounds__(256) void gru_step_fwd_kernel(FwdGroup g, int B, int T, int step) {
    __shared__ float red[4][3][RB][UB];   // [wave][gate][row][unit], 24 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int s = 0;
    while (s + 1 < g.n && (int)blockIdx.x >= g.blk_start[s + 1]) ++s;
    const m3t_gru_fwd_desc d = g.d[s];
    const int H = d.H;
    const int j0 = ((int)blockIdx.x - g.blk_start[s]) * UB;
    const int r0 = blockIdx.y * RB;
    const int t = d.reverse ? T - 1 - step : step;
    const int tp = d.reverse ? t + 1 : t - 1;
    const bool has_prev = step > 0;

    if (has_prev) {
        f32x4 acc[2][3];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bool aok[2], bok[3];
        size_t aoff[2], boff[3];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = r0 + rt * 16 + (lane & 15);
            aok[rt] = row < B;
            aoff[rt] = ((size_t)row * T + tp) * d.ldo + d.ooff;
        }
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            const int j = j0 + (lane & 15);
            bok[ct] = j < H;
            boff[ct] = ((size_t)ct * H + j) * H;
        }
        const bool vecA = ((d.ldo | d.ooff) & 3) == 0 && ((uintptr_t)d.out & 15) == 0;
        const bool vecB = (H & 3) == 0 && ((uintptr_t)d.w_hh & 15) == 0;
        wave_mma<3>(d.out, 0, aok, aoff, d.w_hh, 0, bok, boff, H, vecA, vecB, wave, lane, acc);
        // C/D map 16x16: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][ct][rt * 16 + (lane >> 4) * 4 + r][lane & 15] = acc[rt][ct][r];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int p = tid + e * 256;
        const int row = p >> 4, u = p & 15;
        const int b = r0 + row, j = j0 + u;
        if (b >= B || j >= H) continue;
        float hr = 0.f, hz = 0.f, hn = 0.f, hprev = 0.f;
        if (has_prev) {
            hr = (red[0][0][row][u] + red[1][0][row][u]) + (red[2][0][row][u] + red[3][0][row][u]);
            hz = (red[0][1][row][u] + red[1][1][row][u]) + (red[2][1][row][u] + red[3][1][row][u]);
            hn = (red[0][2][row][u] + red[1][2][row][u]) + (red[2][2][row][u] + red[3][2][row][u]);
            hprev = d.out[((size_t)b * T + tp) * d.ldo + d.ooff + j];
        }
        hr += d.b_hh[j];
        hz += d.b_hh[H + j];
        hn += d.b_hh[2 * H + j];
        const float* xp = d.xproj + ((size_t)b * T + t) * d.ldx + d.xoff;
        const float r = 1.f / (1.f + expf(-(xp[j] + hr)));
        const float z = 1.f / (1.f + expf(-(xp[H + j] + hz)));
        const float n = tanhf(xp[2 * H + j] + r * hn);
        const float h = n + z * (hprev - n);
        d.out[((size_t)b * T + t) * d.ldo + d.ooff + j] = h;
        if (d.gates) {
            float* gp = d.gates + ((size_t)b * T + t) * 4 * H;
            gp[j] = r; gp[H + j] = z; gp[2 * H + j] = n; gp[3 * H + j] = hn;
        }
        if (d.h_n && step == T - 1) d.h_n[(size_t)b * H + j] = h;
    }
}

__global__ __launch_bounds__(256) void gru_step_bwd_kernel(BwdGroup g, int B, int T, int step) {
    __shared__ float red[4][RB][UB];   // 8 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int s = 0;
    while (s + 1 < g.n && (int)blockIdx.x >= g.blk_start[s + 1]) ++s;
    const m3t_gru_bwd_desc d = g.d[s];
    const int H = d.H, H3 = 3 * d.H;
    const int j0 = ((int)blockIdx.x - g.blk_start[s]) * UB;
    const int r0 = blockIdx.y * RB;
    const int t = d.reverse ? step : T - 1 - step;           // backward visits the forward order reversed
    const int tn = d.reverse ? t - 1 : t + 1;                // the step handled by the previous launch
    const int tp = d.reverse ? t + 1 : t - 1;                // forward predecessor (h_{t-1})
    const bool has_next = step > 0, has_prev = step < T - 1;

    if (has_next) {
        f32x4 acc[2][1];
        acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[1][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bool aok[2], bok[1];
        size_t aoff[2], boff[1];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = r0 + rt * 16 + (lane & 15);
            aok[rt] = row < B;
            aoff[rt] = ((size_t)row * T + tn) * H3;
        }
        const int j = j0 + (lane & 15);
        bok[0] = j < H;
        boff[0] = (size_t)j * H3;
        const bool vecA = (H3 & 3) == 0 && ((uintptr_t)d.dgh & 15) == 0;
        const bool vecB = (H3 & 3) == 0 && ((uintptr_t)d.w_hh_t & 15) == 0;
        wave_mma<1>(d.dgh, 0, aok, aoff, d.w_hh_t, 0, bok, boff, H3, vecA, vecB, wave, lane, acc);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][rt * 16 + (lane >> 4) * 4 + r][lane & 15] = acc[rt][0][r];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int p = tid + e * 256;
        const int row = p >> 4, u = p & 15;
        const int b = r0 + row, j = j0 + u;
        if (b >= B || j >= H) continue;
        float carry;
        if (has_next) {
            const float mm = (red[0][row][u] + red[1][row][u]) + (red[2][row][u] + red[3][row][u]);
            const float zn = d.gates[((size_t)b * T + tn) * 4 * H + H + j];
            carry = d.dh[(size_t)b * H + j] * zn + mm;
        } else {
            carry = d.dh_n ? d.dh_n[(size_t)b * H + j] : 0.f;
        }
        const float dht = d.dout[((size_t)b * T + t) * d.ldo + d.ooff + j] + carry;
        const float* gp = d.gates + ((size_t)b * T + t) * 4 * H;
        const float r = gp[j], z = gp[H + j], n = gp[2 * H + j], hn = gp[3 * H + j];
        const float hprev = has_prev ? d.out[((size_t)b * T + tp) * d.ldo + d.ooff + j] : 0.f;
        const float dn = dht * (1.f - z) * (1.f - n * n);
        const float dz = dht * (hprev - n) * z * (1.f - z);
        const float dr = dn * hn * r * (1.f - r);
        float* gx = d.dgx + ((size_t)b * T + t) * d.ldg + d.goff;
        gx[j] = dr; gx[H + j] = dz; gx[2 * H + j] = dn;
        float* gh = d.dgh + ((size_t)b * T + t) * H3;
        gh[j] = dr; gh[H + j] = dz; gh[2 * H + j] = dn * r;
        d.dh[(size_t)b * H + j] = dht;
    }
}

}  // namespace

extern "C" int m3t_gru_scan_fwd(const m3t_gru_fwd_desc* scans, int n_scans, int B, int T, void* stream) {
    if (n_scans <= 0 || B <= 0 || T <= 0) return 0;
    if (n_scans > M3T_MAX_SCANS || !scans) return M3T_EINVAL;
    FwdGroup g;
    g.n = n_scans;
    int blocks = 0;
    for (int i = 0; i < n_scans; ++i) {
        if (scans[i].H <= 0 || !scans[i].xproj || !scans[i].w_hh || !scans[i].b_hh || !scans[i].out) return M3T_EINVAL;
        g.d[i] = scans[i];
        g.blk_start[i] = blocks;
        blocks += cdiv(scans[i].H, UB);
    }
    for (int i = n_scans; i <= M3T_MAX_SCANS; ++i) g.blk_start[i] = blocks;
    dim3 grid(blocks, cdiv(B, RB));
    hipStream_t s = (hipStream_t)stream;
    for (int step = 0; step < T; ++step) gru_step_fwd_kernel<<<grid, 256, 0, s>>>(g, B, T, step);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_gru_scan_bwd(const m3t_gru_bwd_desc* scans, int n_scans, int B, int T, void* stream) {
    if (n_scans <= 0 || B <= 0 || T <= 0) return 0;
    if (n_scans > M3T_MAX_SCANS || !scans) return M3T_EINVAL;
    BwdGroup g;
    g.n = n_scans;
    int blocks = 0;
    for (int i = 0; i < n_scans; ++i) {
        const m3t_gru_bwd_desc& d = scans[i];
        if (d.H <= 0 || !d.dout || !d.out || !d.gates || !d.w_hh_t || !d.dgx || !d.dgh || !d.dh) return M3T_EINVAL;
        g.d[i] = d;
        g.blk_start[i] = blocks;
        blocks += cdiv(d.H, UB);
    }
    for (int i = n_scans; i <= M3T_MAX_SCANS; ++i) g.blk_start[i] = blocks;
    dim3 grid(blocks, cdiv(B, RB));
    hipStream_t s = (hipStream_t)stream;
    for (int step = 0; step < T; ++step) gru_step_bwd_kernel<<<grid, 256, 0, s>>>(g, B, T, step);
    M3T_LAUNCH_CHECK();
    return 0;
}
