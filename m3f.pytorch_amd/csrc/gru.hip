// BiGRU recurrence for gfx950: entry points, launch planning, and the LAUNCH-PER-STEP kernels.
// Default path since the persistent scans exist (gru_persist.hip: one launch per level, W_hh in registers, h_t exchanged
// between CUs as tagged granules): this file's step kernels now serve the levels that path does not take -- H not a
// multiple of 128 or above 512, levels that do not fit the chip at one workgroup per CU, scans issued on a side
// stream without fencing, M3T_SCAN_PERSIST=0 -- and are the reference the persistent kernels are tested against bit
// for bit.  What one step launch does:
//   * it advances ALL independent direction-scans of the level at once (both directions,
//     every independent stack), so the chip sees sum_s (H_s/16) x ceil(B/32) workgroups;
//   * a workgroup owns 16 hidden units of one scan: the three gate columns r,z,n of those
//     units (a [32 x 48] tile of h_{t-1} W_hh^T, K = H) on fp32 MFMA 16x16x4 (exact fp32),
//     K split over its 8 waves, operands loaded straight to registers as float4 along K with
//     EVERY load of the pass in flight before the first MFMA (one L2 round trip per step, not
//     one per k-chunk), fixed-order LDS reduction, then the cell math for its own units --
//     gates never travel un-fused.  The T launches of a call are replayed as one hipGraph.
// Backward runs the same structure in reverse time: dh_{t} = dout_t + z_{t+1} dh_{t+1}
// + dgh_{t+1} W_hh, with the matmul of step t+1 and the gate derivative of step t fused
// in one launch; dW_hh / dW_ih / dx are left to big GEMMs after the scan.
#include "gru_common.h"

using namespace m3t_gru;

namespace {

constexpr int CF = 4;         // k-chunks (16 wide) a wave keeps in flight, forward  (covers H  <= 512 in one pass)
constexpr int CB = 12;        // k-chunks a wave keeps in flight, backward            (covers 3H <= 1536 in one pass)

__device__ __forceinline__ float4 ld4(const float* __restrict__ p, int nvalid, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid >= 4 && vec) return *reinterpret_cast<const float4*>(p);
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
    return v;
}

// acc[rt][ct] += A[2 row tiles, K] * Bt[CT col tiles, K]^T on fp32 MFMA 16x16x4.  Wave w owns the 16-wide
// k chunks c = w, w+NW, ...  All of a pass's loads (NC chunks x (2+CT) float4 per lane) are issued before the
// first MFMA so ONE L2 round trip covers the pass instead of one per chunk.
// FAST: K % 16 == 0 and 16-B aligned rows -- no predicates (out-of-range rows/cols are clamped by the caller
// and their results discarded).
template <int CT, int NC, bool FAST>
__device__ __forceinline__ void wave_mma(const float* __restrict__ A, const size_t (&arow)[2], const bool (&aok)[2],
                                         const float* __restrict__ Bt, const size_t (&brow)[CT], const bool (&bok)[CT],
                                         int K, bool vecA, bool vecB, int wave, int lane, f32x4 (&acc)[2][CT], bool bf = false) {
    const int kq = (lane >> 4) * 4;
    const int nchunks = (K + 15) >> 4;
    for (int c0 = wave; c0 < nchunks; c0 += NW * NC) {
        float4 a[NC][2], b[NC][CT];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = c0 + i * NW;
            const int kk = c * 16 + kq;
            if (FAST) {
                if (c < nchunks) {
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) a[i][rt] = *reinterpret_cast<const float4*>(A + arow[rt] + kk);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) b[i][ct] = *reinterpret_cast<const float4*>(Bt + brow[ct] + kk);
                }
            } else {
                const int nv = c < nchunks ? K - kk : 0;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    a[i][rt] = (aok[rt] && nv > 0) ? ld4(A + arow[rt] + kk, nv, vecA) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    b[i][ct] = (bok[ct] && nv > 0) ? ld4(Bt + brow[ct] + kk, nv, vecB) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (bf) {
#pragma unroll
            for (int i = 0; i < NC; ++i) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) a[i][rt] = rbf4(a[i][rt]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) b[i][ct] = rbf4(b[i][ct]);
            }
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (FAST && c0 + i * NW >= nchunks) break;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][rt].x, b[i][ct].x, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][rt].y, b[i][ct].y, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][rt].z, b[i][ct].z, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][rt].w, b[i][ct].w, acc[rt][ct], 0, 0, 0);
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fragment-ordered fast path.  A wave's MFMA 16x16x4 operand for a 16-wide k chunk is, per lane,
// (row|col = lane&15, k = 4*(lane>>4) .. +3): read from row-major [rows][K] storage that is 16 rows x 64 B per
// wave instruction, which the vector memory path serves at about HALF the rate of a contiguous 1 KiB request
// (measured: 36 vs 73 GB/s per CU, tools/patprobe.hip).  So both operands are kept in the exact order the lanes
// consume them -- value(lane, e) at [chunk][tile][lane][e] -- and every operand load is one coalesced 1 KiB
// wave instruction:
//   wfrag : W_hh re-laid once per call by a prep kernel            [unit-block][chunk][gate tile][lane][4]
//   hfrag : h_t, written by the gate-math epilogue of step t in    [row-block][chunk][row tile][lane][4]
//           addition to out[b,t,:] (ping-pong by step parity): each workgroup owns exactly chunk == its unit-block,
//           i.e. one contiguous 2 KiB piece.
// Backward mirrors it with dgh_t (3H wide) and W_hh^T.
__global__ void wfrag_fwd_prep_kernel(const float* __restrict__ w_hh, float* __restrict__ wfrag, int H, int bf16) {
    const int nch = H >> 4;
    const size_t total = (size_t)3 * H * H;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = i & 3, l = (i >> 2) & 63;
        size_t r = i >> 8;
        const int ct = r % 3; r /= 3;
        const int c = r % nch, ub = r / nch;
        const float v = w_hh[((size_t)ct * H + ub * 16 + (l & 15)) * H + c * 16 + (l >> 4) * 4 + e];
        wfrag[i] = bf16 ? rbf(v) : v;
    }
}
// wtfrag[ub][c][lane][e] = W_hh[k = 16c + 4*(lane>>4) + e][ub*16 + (lane&15)] = w_hh_t[ub*16 + (lane&15)][k]
__global__ void wfrag_bwd_prep_kernel(const float* __restrict__ w_hh_t, float* __restrict__ wtfrag, int H, int bf16) {
    const int nch = (3 * H) >> 4;
    const size_t total = (size_t)3 * H * H;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = i & 3, l = (i >> 2) & 63;
        const size_t r = i >> 8;
        const int c = r % nch, ub = r / nch;
        const float v = w_hh_t[((size_t)ub * 16 + (l & 15)) * 3 * H + c * 16 + (l >> 4) * 4 + e];
        wtfrag[i] = bf16 ? rbf(v) : v;
    }
}

// the same fragments straight from the untransposed parameter w_hh [3H][H] (M3T_SCAN_WHH): W_hh[k][j] = w_hh[k*H + j]
__global__ void wfrag_bwd_direct_kernel(const float* __restrict__ w_hh, float* __restrict__ wtfrag, int H, int bf16) {
    const int nch = (3 * H) >> 4;
    const size_t total = (size_t)3 * H * H;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = i & 3, l = (i >> 2) & 63;
        const size_t r = i >> 8;
        const int c = r % nch, ub = r / nch;
        const float v = w_hh[((size_t)c * 16 + (l >> 4) * 4 + e) * H + ub * 16 + (l & 15)];
        wtfrag[i] = bf16 ? rbf(v) : v;
    }
}

// NC = chunks per wave kept in flight; CT = column tiles; RT = row tiles.
// A frag: [chunk][RT row tiles][64][4]; B frag: [chunk][CT][64][4]
template <int CT, int NC, int RT>
__device__ __forceinline__ void wave_mma_frag(const float4* __restrict__ Af, const float4* __restrict__ Bf, int nchunks,
                                              int wave, int lane, f32x4 (&acc)[RT][CT]) {
    for (int c0 = wave; c0 < nchunks; c0 += NW * NC) {
        float4 a[NC][RT], b[NC][CT];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = c0 + i * NW;
            if (c < nchunks) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[i][rt] = Af[((size_t)c * RT + rt) * 64 + lane];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) b[i][ct] = Bf[((size_t)c * CT + ct) * 64 + lane];
            }
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (c0 + i * NW >= nchunks) break;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][rt].x, b[i][ct].x, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][rt].y, b[i][ct].y, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][rt].z, b[i][ct].z, acc[rt][ct], 0, 0, 0);
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][rt].w, b[i][ct].w, acc[rt][ct], 0, 0, 0);
                }
        }
    }
}

// RT row tiles per workgroup: RT = 2 covers 32 batch rows, RT = 1 covers 16 (twice the workgroups, half the MFMA
// chain each -- used when the level would otherwise leave most CUs idle).  Always 8 waves: K is split 8 ways.
template <int RT>
__global__ __launch_bounds__(NT) void gru_step_fwd_frag_kernel(FwdGroup g, FragPtrs fp, int B, int T, int step) {
    constexpr int ROWS = 16 * RT;
    __shared__ float red[NW][3][ROWS][UB];   // 48 / 24 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int s = 0;
    while (s + 1 < g.n && (int)blockIdx.x >= g.blk_start[s + 1]) ++s;
    const m3t_gru_fwd_desc d = g.d[s];
    const int H = d.H, nch = H >> 4;
    const int local = (int)blockIdx.x - g.blk_start[s];
    const int ub = local % nch, rb = local / nch;     // unit block fastest: workgroups sharing a W slice sit nch apart
    const int j0 = ub * UB, r0 = rb * ROWS;
    const int t = d.reverse ? T - 1 - step : step;
    const int tp = d.reverse ? t + 1 : t - 1;
    const bool has_prev = step > 0;
    const float* hin = fp.xfrag[s] + (size_t)((step + 1) & 1) * fp.xstride[s] + (size_t)rb * ROWS * H;
    float* hout = fp.xfrag[s] + (size_t)(step & 1) * fp.xstride[s] + (size_t)rb * ROWS * H;

    const bool pw = tid < ROWS * UB;             // gate-math threads
    const int prow = tid >> 4, pu = tid & 15;
    const int pb = r0 + prow, pj = j0 + pu;
    const bool pok = pw && pb < B;
    float xr = 0.f, xz = 0.f, xn = 0.f, hprev = 0.f, br = 0.f, bz = 0.f, bn = 0.f;
    if (pw) { br = d.b_hh[pj]; bz = d.b_hh[H + pj]; bn = d.b_hh[2 * H + pj]; }
    if (pok) {
        const float* xp = d.xproj + ((size_t)pb * T + t) * d.ldx + d.xoff;
        xr = xp[pj]; xz = xp[H + pj]; xn = xp[2 * H + pj];
        if (has_prev) hprev = d.out[((size_t)pb * T + tp) * d.ldo + d.ooff + pj];
    }
    if (has_prev) {
        f32x4 acc[RT][3];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        wave_mma_frag<3, CF, RT>(reinterpret_cast<const float4*>(hin),
                                 reinterpret_cast<const float4*>(fp.wfrag[s] + (size_t)ub * nch * 3 * 256), nch, wave, lane,
                                 acc);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][ct][rt * 16 + (lane >> 4) * 4 + r][lane & 15] = acc[rt][ct][r];
    }
    __syncthreads();
    if (!pw) return;
    float hr = br, hz = bz, hn = bn;
    if (has_prev) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            hr += red[w][0][prow][pu];
            hz += red[w][1][prow][pu];
            hn += red[w][2][prow][pu];
        }
    }
    const GateFwd c = gru_cell_fwd(xr, xz, xn, hr, hz, hn, hprev);
    const float r = c.r, z = c.z, n = c.n, h = c.h;
    // fragment-ordered copy for the next step: chunk == ub, tile = prow>>4, lane = (pu>>2)*16 + (prow&15), e = pu&3
    hout[(((size_t)ub * RT + (prow >> 4)) * 64 + (pu >> 2) * 16 + (prow & 15)) * 4 + (pu & 3)] = pok ? (g.bf16 ? rbf(h) : h) : 0.f;
    if (!pok) return;
    d.out[((size_t)pb * T + t) * d.ldo + d.ooff + pj] = h;
    if (d.gates) {
        float* gp = d.gates + ((size_t)pb * T + t) * 4 * H;
        *reinterpret_cast<float4*>(gp + 4 * (size_t)pj) = make_float4(r, z, n, hn);
    }
    if (d.h_n && step == T - 1) d.h_n[(size_t)pb * H + pj] = h;
}

template <int RT>
__global__ __launch_bounds__(NT) void gru_step_bwd_frag_kernel(BwdGroup g, FragPtrs fp, int B, int T, int step) {
    constexpr int ROWS = 16 * RT;
    __shared__ float red[NW][ROWS][UB];   // 16 / 8 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int s = 0;
    while (s + 1 < g.n && (int)blockIdx.x >= g.blk_start[s + 1]) ++s;
    const m3t_gru_bwd_desc d = g.d[s];
    const int H = d.H, H3 = 3 * d.H, nchh = H >> 4, nch = H3 >> 4;
    const int local = (int)blockIdx.x - g.blk_start[s];
    const int ub = local % nchh, rb = local / nchh;
    const int j0 = ub * UB, r0 = rb * ROWS;
    const int t = d.reverse ? step : T - 1 - step;
    const int tn = d.reverse ? t - 1 : t + 1;
    const int tp = d.reverse ? t + 1 : t - 1;
    const bool has_next = step > 0, has_prev = step < T - 1;
    const float* gin = fp.xfrag[s] + (size_t)((step + 1) & 1) * fp.xstride[s] + (size_t)rb * ROWS * H3;
    float* gout = fp.xfrag[s] + (size_t)(step & 1) * fp.xstride[s] + (size_t)rb * ROWS * H3;

    const bool pw = tid < ROWS * UB;
    const int prow = tid >> 4, pu = tid & 15;
    const int pb = r0 + prow, pj = j0 + pu;
    const bool pok = pw && pb < B;
    float dout = 0.f, gr = 0.f, gz = 0.f, gn = 0.f, ghn = 0.f, hprev = 0.f, zn = 0.f, dhn = 0.f;
    if (pok) {
        dout = d.dout[((size_t)pb * T + t) * d.ldo + d.ooff + pj];
        const float* gp = d.gates + ((size_t)pb * T + t) * 4 * H;
        { const float4 g4 = *reinterpret_cast<const float4*>(gp + 4 * (size_t)pj); gr = g4.x; gz = g4.y; gn = g4.z; ghn = g4.w; }
        if (has_prev) hprev = d.out[((size_t)pb * T + tp) * d.ldo + d.ooff + pj];
        if (has_next) {
            zn = d.gates[((size_t)pb * T + tn) * 4 * H + 4 * (size_t)pj + 1];
            dhn = d.dh[(size_t)pb * H + pj];
        } else if (d.dh_n) {
            dhn = d.dh_n[(size_t)pb * H + pj];
        }
    }
    if (has_next) {
        f32x4 acc[RT][1];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        wave_mma_frag<1, CB, RT>(reinterpret_cast<const float4*>(gin),
                                 reinterpret_cast<const float4*>(fp.wfrag[s] + (size_t)ub * nch * 256), nch, wave, lane, acc);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][rt * 16 + (lane >> 4) * 4 + r][lane & 15] = acc[rt][0][r];
    }
    __syncthreads();
    if (!pw) return;
    float mm = 0.f;
    if (has_next) {
#pragma unroll
        for (int w = 0; w < NW; ++w) mm += red[w][prow][pu];
    }
    const GateBwd c = gru_cell_bwd(dout, dhn, zn, mm, has_next, gr, gz, gn, ghn, hprev);
    const float dht = c.dht, dn = c.dn, dz = c.dz, dr = c.dr;
    // fragment-ordered dgh for the next launch: gate g lives in chunk g*(H/16) + ub
    const size_t fo = ((size_t)(prow >> 4) * 64 + (pu >> 2) * 16 + (prow & 15)) * 4 + (pu & 3);
    gout[((size_t)(0 * nchh + ub) * RT) * 256 + fo] = pok ? (g.bf16 ? rbf(dr) : dr) : 0.f;
    gout[((size_t)(1 * nchh + ub) * RT) * 256 + fo] = pok ? (g.bf16 ? rbf(dz) : dz) : 0.f;
    gout[((size_t)(2 * nchh + ub) * RT) * 256 + fo] = pok ? (g.bf16 ? rbf(c.dnr) : c.dnr) : 0.f;
    if (!pok) return;
    float* gx = d.dgx + ((size_t)pb * T + t) * d.ldg + d.goff;
    gx[pj] = dr; gx[H + pj] = dz; gx[2 * H + pj] = dn;
    float* gh = d.dgh + ((size_t)pb * T + t) * H3;
    gh[pj] = dr; gh[H + pj] = dz; gh[2 * H + pj] = c.dnr;
    d.dh[(size_t)pb * H + pj] = dht;
    if (d.db_part) {                       // bias-gradient partials of this clip, accumulated across the step launches
        float* q = d.db_part + (size_t)pb * 4 * H + pj;
        if (step == 0) { q[0] = dr; q[H] = dz; q[2 * H] = dn; q[3 * H] = c.dnr; }
        else { q[0] += dr; q[H] += dz; q[2 * H] += dn; q[3 * H] += c.dnr; }
    }
}

template <bool FAST>
__global__ __launch_bounds__(NT) void gru_step_fwd_kernel(FwdGroup g, int B, int T, int step) {
    __shared__ float red[NW][3][RB][UB];   // [wave][gate][row][unit], 48 KiB
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

    // this thread's gate-math element; its inputs are requested before the matmul so they ride under it
    const int prow = tid >> 4, pu = tid & 15;
    const int pb = r0 + prow, pj = j0 + pu;
    const bool pok = pb < B && pj < H;
    float xr = 0.f, xz = 0.f, xn = 0.f, br = 0.f, bz = 0.f, bn = 0.f, hprev = 0.f;
    if (pok) {
        const float* xp = d.xproj + ((size_t)pb * T + t) * d.ldx + d.xoff;
        xr = xp[pj]; xz = xp[H + pj]; xn = xp[2 * H + pj];
        br = d.b_hh[pj]; bz = d.b_hh[H + pj]; bn = d.b_hh[2 * H + pj];
        if (has_prev) hprev = d.out[((size_t)pb * T + tp) * d.ldo + d.ooff + pj];
    }

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
            aoff[rt] = ((size_t)(FAST ? min(row, B - 1) : row) * T + tp) * d.ldo + d.ooff;
        }
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            const int j = j0 + (lane & 15);
            bok[ct] = j < H;
            boff[ct] = ((size_t)ct * H + (FAST ? min(j, H - 1) : j)) * H;
        }
        const bool vecA = ((d.ldo | d.ooff) & 3) == 0 && ((uintptr_t)d.out & 15) == 0;
        const bool vecB = (H & 3) == 0 && ((uintptr_t)d.w_hh & 15) == 0;
        wave_mma<3, CF, FAST>(d.out, aoff, aok, d.w_hh, boff, bok, H, vecA, vecB, wave, lane, acc, g.bf16 != 0);
        // C/D map 16x16: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][ct][rt * 16 + (lane >> 4) * 4 + r][lane & 15] = acc[rt][ct][r];
    }
    __syncthreads();
    if (!pok) return;
    float hr = 0.f, hz = 0.f, hn = 0.f;
    if (has_prev) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {      // fixed order: deterministic
            hr += red[w][0][prow][pu];
            hz += red[w][1][prow][pu];
            hn += red[w][2][prow][pu];
        }
    }
    hr += br; hz += bz; hn += bn;
    const GateFwd c = gru_cell_fwd(xr, xz, xn, hr, hz, hn, hprev);
    const float r = c.r, z = c.z, n = c.n, h = c.h;
    d.out[((size_t)pb * T + t) * d.ldo + d.ooff + pj] = h;
    if (d.gates) {
        float* gp = d.gates + ((size_t)pb * T + t) * 4 * H;
        *reinterpret_cast<float4*>(gp + 4 * (size_t)pj) = make_float4(r, z, n, hn);
    }
    if (d.h_n && step == T - 1) d.h_n[(size_t)pb * H + pj] = h;
}

template <bool FAST>
__global__ __launch_bounds__(NT) void gru_step_bwd_kernel(BwdGroup g, int B, int T, int step) {
    __shared__ float red[NW][RB][UB];   // 16 KiB
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

    const int prow = tid >> 4, pu = tid & 15;
    const int pb = r0 + prow, pj = j0 + pu;
    const bool pok = pb < B && pj < H;
    float dout = 0.f, gr = 0.f, gz = 0.f, gn = 0.f, ghn = 0.f, hprev = 0.f, zn = 0.f, dhn = 0.f;
    if (pok) {
        dout = d.dout[((size_t)pb * T + t) * d.ldo + d.ooff + pj];
        const float* gp = d.gates + ((size_t)pb * T + t) * 4 * H;
        { const float4 g4 = *reinterpret_cast<const float4*>(gp + 4 * (size_t)pj); gr = g4.x; gz = g4.y; gn = g4.z; ghn = g4.w; }
        if (has_prev) hprev = d.out[((size_t)pb * T + tp) * d.ldo + d.ooff + pj];
        if (has_next) {
            zn = d.gates[((size_t)pb * T + tn) * 4 * H + 4 * (size_t)pj + 1];
            dhn = d.dh[(size_t)pb * H + pj];
        } else if (d.dh_n) {
            dhn = d.dh_n[(size_t)pb * H + pj];
        }
    }

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
            aoff[rt] = ((size_t)(FAST ? min(row, B - 1) : row) * T + tn) * H3;
        }
        const int j = j0 + (lane & 15);
        bok[0] = j < H;
        boff[0] = (size_t)(FAST ? min(j, H - 1) : j) * H3;
        const bool vecA = (H3 & 3) == 0 && ((uintptr_t)d.dgh & 15) == 0;
        const bool vecB = (H3 & 3) == 0 && ((uintptr_t)d.w_hh_t & 15) == 0;
        wave_mma<1, CB, FAST>(d.dgh, aoff, aok, d.w_hh_t, boff, bok, H3, vecA, vecB, wave, lane, acc, g.bf16 != 0);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][rt * 16 + (lane >> 4) * 4 + r][lane & 15] = acc[rt][0][r];
    }
    __syncthreads();
    if (!pok) return;
    float mm = 0.f;                          // has_next: dhn is dL/dh_{tn} total; else grad wrt the final hidden state
    if (has_next) {
#pragma unroll
        for (int w = 0; w < NW; ++w) mm += red[w][prow][pu];
    }
    const GateBwd c = gru_cell_bwd(dout, dhn, zn, mm, has_next, gr, gz, gn, ghn, hprev);
    const float dht = c.dht, dn = c.dn, dz = c.dz, dr = c.dr;
    float* gx = d.dgx + ((size_t)pb * T + t) * d.ldg + d.goff;
    gx[pj] = dr; gx[H + pj] = dz; gx[2 * H + pj] = dn;
    float* gh = d.dgh + ((size_t)pb * T + t) * H3;
    gh[pj] = dr; gh[H + pj] = dz; gh[2 * H + pj] = c.dnr;
    d.dh[(size_t)pb * H + pj] = dht;
    if (d.db_part) {                       // bias-gradient partials of this clip, accumulated across the step launches
        float* q = d.db_part + (size_t)pb * 4 * H + pj;
        if (step == 0) { q[0] = dr; q[H] = dz; q[2 * H] = dn; q[3 * H] = c.dnr; }
        else { q[0] += dr; q[H] += dz; q[2 * H] += dn; q[3 * H] += c.dnr; }
    }
}

// db_ih[g*H + j] = sum_b part[b][g][j] (g = r,z,n); db_hh = the same with the n block taken from part[b][3] (dn*r).
// One thread per (scan, unit); B terms in a fixed order.
struct BiasFinish { const float* part[M3T_MAX_SCANS]; float* db_ih[M3T_MAX_SCANS]; float* db_hh[M3T_MAX_SCANS]; int H[M3T_MAX_SCANS]; int n; };
__global__ void gru_bias_finish_kernel(BiasFinish f, int B) {
    const int s = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= f.n || !f.part[s] || j >= f.H[s]) return;
    const int H = f.H[s];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = 0;
    for (; b + 8 <= B; b += 8) {                       // 32 loads in flight, the sums in the same (b) order as before
        float v[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* q = f.part[s] + (size_t)(b + u) * 4 * H + j;
            v[u][0] = q[0]; v[u][1] = q[H]; v[u][2] = q[2 * H]; v[u][3] = q[3 * H];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { a0 += v[u][0]; a1 += v[u][1]; a2 += v[u][2]; a3 += v[u][3]; }
    }
    for (; b < B; ++b) {
        const float* q = f.part[s] + (size_t)b * 4 * H + j;
        a0 += q[0]; a1 += q[H]; a2 += q[2 * H]; a3 += q[3 * H];
    }
    if (f.db_ih[s]) { f.db_ih[s][j] = a0; f.db_ih[s][H + j] = a1; f.db_ih[s][2 * H + j] = a2; }
    if (f.db_hh[s]) { f.db_hh[s][j] = a0; f.db_hh[s][H + j] = a1; f.db_hh[s][2 * H + j] = a3; }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// hipGraph replay of a whole T-launch scan.  The T step kernels of a call differ only in `step`, and in a
// training loop the same (pointers, shapes) recur every iteration (caching allocator), so a call is captured
// once into a graph (stream capture, thread-local mode), instantiated, and replayed afterwards: one
// hipGraphLaunch instead of T launches.  Keyed by the exact argument bytes; small LRU; M3T_GRAPH=0 disables.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

struct GraphEntry {
    std::vector<unsigned char> key;
    hipGraphExec_t exec;
    unsigned long long stamp;
};
std::mutex g_graph_mu;
std::vector<GraphEntry> g_graphs;
unsigned long long g_graph_clock = 0;
constexpr size_t kMaxGraphs = 96;

bool graphs_enabled() { return true; }

template <typename Launch>
int replay_or_capture(const std::vector<unsigned char>& key, hipStream_t s, Launch&& launch_all) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone || !graphs_enabled()) {
        launch_all();            // already inside somebody else's capture (or disabled): plain launches
        return (int)hipGetLastError();
    }
    std::lock_guard<std::mutex> lk(g_graph_mu);
    for (auto& e : g_graphs)
        if (e.key == key) {
            e.stamp = ++g_graph_clock;
            return (int)hipGraphLaunch(e.exec, s);
        }
    hipGraph_t graph = nullptr;
    hipError_t err = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (err != hipSuccess) {     // capture unavailable on this stream: fall back to plain launches
        (void)hipGetLastError();
        launch_all();
        return (int)hipGetLastError();
    }
    launch_all();
    err = hipStreamEndCapture(s, &graph);
    if (err != hipSuccess || !graph) return (int)(err != hipSuccess ? err : hipErrorUnknown);
    hipGraphExec_t exec = nullptr;
    err = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (err != hipSuccess) return (int)err;
    if (g_graphs.size() >= kMaxGraphs) {
        size_t victim = 0;
        for (size_t i = 1; i < g_graphs.size(); ++i)
            if (g_graphs[i].stamp < g_graphs[victim].stamp) victim = i;
        (void)hipGraphExecDestroy(g_graphs[victim].exec);
        g_graphs.erase(g_graphs.begin() + victim);
    }
    g_graphs.push_back({key, exec, ++g_graph_clock});
    return (int)hipGraphLaunch(exec, s);
}

// Launch plan of a level.  Workgroups are issued heaviest scan first (largest H), so when a level has more
// workgroups than CUs the ones that double up on a CU are the light ones.  Row tiles per workgroup (RT = 2: 32
// rows, RT = 1: 16 rows, twice the workgroups at half the MFMA chain) are chosen by simulating the dispatcher's
// round-robin over 256 CUs with cost(workgroup) ~ fixed + RT * H/512.  M3T_SCAN_RT=1|2 overrides (tuning).
struct ScanPlan { int order[M3T_MAX_SCANS]; int rt; };

ScanPlan plan_level(const int* Hs, int n, int B) {
    ScanPlan p;
    for (int i = 0; i < n; ++i) p.order[i] = i;
    for (int i = 1; i < n; ++i)          // stable insertion sort by H, descending
        for (int j = i; j > 0 && Hs[p.order[j]] > Hs[p.order[j - 1]]; --j) {
            const int tmp = p.order[j]; p.order[j] = p.order[j - 1]; p.order[j - 1] = tmp;
        }
    if (B <= 16) { p.rt = 1; return p; }
    double best = 1e30;
    p.rt = 2;
    for (int rt = 2; rt >= 1; --rt) {
        double load[256] = {0};
        int cu = 0;
        for (int i = 0; i < n; ++i) {
            const int H = Hs[p.order[i]];
            const int wgs = cdiv(H, 16) * cdiv(B, 16 * rt);
            const double cost = 0.35 + rt * (double)H / 512.0;
            for (int w = 0; w < wgs; ++w) { load[cu] += cost; cu = (cu + 1) & 255; }
        }
        double mx = 0;
        for (int c = 0; c < 256; ++c) mx = load[c] > mx ? load[c] : mx;
        if (mx < best - 1e-9) { best = mx; p.rt = rt; }
    }
    return p;
}

// floats reserved per scan for the per-step ping-pong state fragments; large enough for the persistent exchange
// granules too (8 B per h value forward, 16 B per (row, unit) backward), so either path can use the region
size_t xfrag_floats(int H, int B, bool backward) {
    const size_t bpad = (size_t)cdiv(B, RB) * RB;
    const size_t step = 2 * bpad * (size_t)H * (backward ? 3 : 1);
    const size_t ex = (persist_exchange_bytes(H, B, backward) + 3) / 4;
    return ((step > ex ? step : ex) + 3) & ~(size_t)3;
}

template <typename G>
std::vector<unsigned char> make_key(int kind, const G& g, const FragPtrs* fp, int B, int T, int variant) {
    std::vector<unsigned char> k(sizeof(int) * 4 + sizeof(G) + (fp ? sizeof(FragPtrs) : 0));
    unsigned char* p = k.data();
    const int hdr[4] = {kind, B, T, variant};
    std::memcpy(p, hdr, sizeof(hdr)); p += sizeof(hdr);
    std::memcpy(p, &g, sizeof(G)); p += sizeof(G);
    if (fp) std::memcpy(p, fp, sizeof(FragPtrs));
    return k;
}

}  // namespace

extern "C" int m3t_gru_scan_fwd(const m3t_gru_fwd_desc* scans, int n_scans, int B, int T, float* ws, size_t ws_bytes,
                                int flags, void* stream) {
    AfterGuard after_guard;      // a pending m3t_gru_scan_after event never outlives this call
    if (n_scans <= 0 || B <= 0 || T <= 0) return 0;
    if (!persist_deferred() && persist_poll_error()) return M3T_ESPIN;
    if (n_scans > M3T_MAX_SCANS || !scans) return M3T_EINVAL;
    FwdGroup g;
    std::memset(&g, 0, sizeof(g));
    g.n = n_scans;
    g.bf16 = (flags & M3T_BF16) ? 1 : 0;
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
    if (solo_fwd_ok(g, B, T, flags)) {                                        // H = 128: one workgroup per (scan, clip), no exchange
        if (persist_progress_armed()) return M3T_EINVAL;                      // (progress marks: only the persistent fp16x3 kernels carry them)
        return solo_fwd_launch(g, B, T, s);
    }
    bool fast = true;     // every scan: H % 16 == 0 and float4-aligned operands
    for (int i = 0; i < n_scans; ++i) {
        const m3t_gru_fwd_desc& d = scans[i];
        fast = fast && (d.H % 16 == 0) && ((d.ldo | d.ooff) % 4 == 0) && ((uintptr_t)d.out % 16 == 0) &&
               ((uintptr_t)d.w_hh % 16 == 0);
    }
    // fragment-ordered path: needs H % 16 == 0 and workspace for re-laid weights + ping-pong h fragments
    bool frag = ws != nullptr && ((uintptr_t)ws % 16 == 0);
    size_t need = 0;
    const size_t bpad = (size_t)cdiv(B, RB) * RB;
    int Hs[M3T_MAX_SCANS];
    for (int i = 0; i < n_scans; ++i) {
        Hs[i] = scans[i].H;
        frag = frag && scans[i].H % 16 == 0;
        need += (size_t)5 * scans[i].H * scans[i].H + xfrag_floats(scans[i].H, B, false);   // 5 H^2: fp32 or bf16x3 fragments
    }
    if (frag && need * sizeof(float) <= ws_bytes) {
        const ScanPlan plan = plan_level(Hs, n_scans, B);
        const int rt = plan.rt, nrb = cdiv(B, 16 * rt);
        FwdGroup fg;
        FragPtrs fp;
        std::memset(&fg, 0, sizeof(fg));
        std::memset(&fp, 0, sizeof(fp));
        fg.n = n_scans;
        fg.bf16 = g.bf16;
        float* p = ws;
        int nblk = 0;
        for (int i = 0; i < n_scans; ++i) {
            const m3t_gru_fwd_desc& d = scans[plan.order[i]];
            fg.d[i] = d;
            fg.blk_start[i] = nblk;
            nblk += (d.H / 16) * nrb;
            fp.wfrag[i] = p; p += (size_t)5 * d.H * d.H;
            fp.xfrag[i] = p; fp.xstride[i] = bpad * d.H; p += xfrag_floats(d.H, B, false);
        }
        for (int i = n_scans; i <= M3T_MAX_SCANS; ++i) fg.blk_start[i] = nblk;
        if (!(flags & M3T_SCAN_NO_PERSIST) && persist_fwd_check(fg, B, T)) {
            // one launch for all T steps: W_hh in registers, h_t exchanged through tagged granules (gru_persist.hip)
            if (!persist_fwd_uses_x6(fg, B, T, flags)) {
                for (int i = 0; i < n_scans; ++i) {
                    const int H = fg.d[i].H;
                    int blk = (3 * H * H + 255) / 256;
                    if (blk > 1024) blk = 1024;
                    wfrag_fwd_prep_kernel<<<blk, 256, 0, s>>>(fg.d[i].w_hh, fp.wfrag[i], H, fg.bf16);
                }
                M3T_LAUNCH_CHECK();
            }
            return persist_fwd_launch(fg, fp, B, T, flags, s);
        }
        if (persist_progress_armed()) return M3T_EINVAL;
        { const int e = persist_take_after(s); if (e) return e; }
        persist_record_start(s);
        struct EndRec { hipStream_t s; ~EndRec() { persist_record_end(s); } } end_rec{s};
        return replay_or_capture(make_key(1, fg, &fp, B, T, rt), s, [&]() {
            for (int i = 0; i < n_scans; ++i) {
                const int H = fg.d[i].H;
                int blk = (3 * H * H + 255) / 256;
                if (blk > 1024) blk = 1024;
                wfrag_fwd_prep_kernel<<<blk, 256, 0, s>>>(fg.d[i].w_hh, fp.wfrag[i], H, fg.bf16);
            }
            if (rt == 2)
                for (int step = 0; step < T; ++step) gru_step_fwd_frag_kernel<2><<<nblk, NT, 0, s>>>(fg, fp, B, T, step);
            else
                for (int step = 0; step < T; ++step) gru_step_fwd_frag_kernel<1><<<nblk, NT, 0, s>>>(fg, fp, B, T, step);
        });
    }
    if (persist_progress_armed()) return M3T_EINVAL;
    { const int e = persist_take_after(s); if (e) return e; }
    persist_record_start(s);
    struct EndRec2 { hipStream_t s; ~EndRec2() { persist_record_end(s); } } end_rec2{s};
    if (fast)
        for (int step = 0; step < T; ++step) gru_step_fwd_kernel<true><<<grid, NT, 0, s>>>(g, B, T, step);
    else
        for (int step = 0; step < T; ++step) gru_step_fwd_kernel<false><<<grid, NT, 0, s>>>(g, B, T, step);
    M3T_LAUNCH_CHECK();
    return 0;
}

static int scan_bwd_impl(const m3t_gru_bwd_desc* scans, int n_scans, int B, int T, float* ws, size_t ws_bytes, int flags,
                         void* stream, bool* amax_done) {
    AfterGuard after_guard;
    if (n_scans <= 0 || B <= 0 || T <= 0) return 0;
    if (!persist_deferred() && persist_poll_error()) return M3T_ESPIN;
    if (n_scans > M3T_MAX_SCANS || !scans) return M3T_EINVAL;
    BwdGroup g;
    std::memset(&g, 0, sizeof(g));
    g.n = n_scans;
    g.bf16 = (flags & M3T_BF16) ? 1 : 0;
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
    if (solo_bwd_ok(g, B, T, flags)) {               // H = 128: one workgroup per (scan, clip); raises the magnitude slots itself
        if (persist_progress_armed()) return M3T_EINVAL;                      // (progress marks: only the wide producer-split kernel carries them)
        *amax_done = true;
        return solo_bwd_launch(g, B, T, flags, s);
    }
    bool fast = true;
    for (int i = 0; i < n_scans; ++i) {
        const m3t_gru_bwd_desc& d = scans[i];
        fast = fast && (d.H % 16 == 0) && ((uintptr_t)d.dgh % 16 == 0) && ((uintptr_t)d.w_hh_t % 16 == 0);
    }
    bool frag = ws != nullptr && ((uintptr_t)ws % 16 == 0);
    size_t need = 0;
    const size_t bpad = (size_t)cdiv(B, RB) * RB;
    int Hs[M3T_MAX_SCANS];
    for (int i = 0; i < n_scans; ++i) {
        Hs[i] = scans[i].H;
        frag = frag && scans[i].H % 16 == 0;
        need += (size_t)5 * scans[i].H * scans[i].H + xfrag_floats(scans[i].H, B, true);   // 5 H^2: fp32 (3 H^2) or bf16x3 (4.5 H^2) fragments
    }
    const bool whh = (flags & M3T_SCAN_WHH) != 0;      // desc.w_hh_t is the untransposed parameter w_hh [3H][H]
    if (whh && !(frag && need * sizeof(float) <= ws_bytes)) return M3T_EINVAL;
    if (frag && need * sizeof(float) <= ws_bytes) {
        const ScanPlan plan = plan_level(Hs, n_scans, B);
        const int rt = plan.rt, nrb = cdiv(B, 16 * rt);
        BwdGroup bg;
        FragPtrs fp;
        std::memset(&bg, 0, sizeof(bg));
        std::memset(&fp, 0, sizeof(fp));
        bg.n = n_scans;
        bg.bf16 = g.bf16;
        float* p = ws;
        int nblk = 0;
        for (int i = 0; i < n_scans; ++i) {
            const m3t_gru_bwd_desc& d = scans[plan.order[i]];
            bg.d[i] = d;
            bg.blk_start[i] = nblk;
            nblk += (d.H / 16) * nrb;
            fp.wfrag[i] = p; p += (size_t)5 * d.H * d.H;
            fp.xfrag[i] = p; fp.xstride[i] = bpad * 3 * d.H; p += xfrag_floats(d.H, B, true);
        }
        for (int i = n_scans; i <= M3T_MAX_SCANS; ++i) bg.blk_start[i] = nblk;
        if (!(flags & M3T_SCAN_NO_PERSIST) && persist_bwd_check(bg, B, T)) {
            for (int i = 0; i < n_scans && !persist_bwd_uses_16(bg, B, T, flags) && !persist_bwd_uses_x6(bg, B, T, flags); ++i) {
                const int H = bg.d[i].H;
                int blk = (3 * H * H + 255) / 256;
                if (blk > 1024) blk = 1024;
                if (whh) wfrag_bwd_direct_kernel<<<blk, 256, 0, s>>>(bg.d[i].w_hh_t, fp.wfrag[i], H, bg.bf16);
                else wfrag_bwd_prep_kernel<<<blk, 256, 0, s>>>(bg.d[i].w_hh_t, fp.wfrag[i], H, bg.bf16);
            }
            M3T_LAUNCH_CHECK();
            *amax_done = persist_bwd_uses_x6(bg, B, T, flags);          // that kernel raises the descs' magnitude slots itself
            return persist_bwd_launch(bg, fp, B, T, flags, s);
        }
        if (persist_progress_armed()) return M3T_EINVAL;
        { const int e = persist_take_after(s); if (e) return e; }
        persist_record_start(s);
        struct EndRec { hipStream_t s; ~EndRec() { persist_record_end(s); } } end_rec{s};
        return replay_or_capture(make_key(2, bg, &fp, B, T, rt), s, [&]() {
            for (int i = 0; i < n_scans; ++i) {
                const int H = bg.d[i].H;
                int blk = (3 * H * H + 255) / 256;
                if (blk > 1024) blk = 1024;
                if (whh) wfrag_bwd_direct_kernel<<<blk, 256, 0, s>>>(bg.d[i].w_hh_t, fp.wfrag[i], H, bg.bf16);
                else wfrag_bwd_prep_kernel<<<blk, 256, 0, s>>>(bg.d[i].w_hh_t, fp.wfrag[i], H, bg.bf16);
            }
            if (rt == 2)
                for (int step = 0; step < T; ++step) gru_step_bwd_frag_kernel<2><<<nblk, NT, 0, s>>>(bg, fp, B, T, step);
            else
                for (int step = 0; step < T; ++step) gru_step_bwd_frag_kernel<1><<<nblk, NT, 0, s>>>(bg, fp, B, T, step);
        });
    }
    if (persist_progress_armed()) return M3T_EINVAL;
    { const int e = persist_take_after(s); if (e) return e; }
    persist_record_start(s);
    struct EndRec2 { hipStream_t s; ~EndRec2() { persist_record_end(s); } } end_rec2{s};
    if (fast)
        for (int step = 0; step < T; ++step) gru_step_bwd_kernel<true><<<grid, NT, 0, s>>>(g, B, T, step);
    else
        for (int step = 0; step < T; ++step) gru_step_bwd_kernel<false><<<grid, NT, 0, s>>>(g, B, T, step);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_gru_scan_bwd(const m3t_gru_bwd_desc* scans, int n_scans, int B, int T, float* ws, size_t ws_bytes,
                                int flags, void* stream) {
    bool amax_done = false;
    const int rc = scan_bwd_impl(scans, n_scans, B, T, ws, ws_bytes, flags, stream, &amax_done);
    if (rc || n_scans <= 0 || B <= 0 || T <= 0) return rc;
    if (!amax_done) {
        // magnitude slots (desc.amax) on the paths whose kernels do not track them: one pass over dgx / dgh behind the scan
        M3TRegion regs[2 * M3T_MAX_SCANS];
        int nr = 0;
        for (int i = 0; i < n_scans; ++i) {
            const m3t_gru_bwd_desc& d = scans[i];
            if (!d.amax) continue;
            if (d.H % 4 != 0 || d.ldg % 4 != 0 || d.goff % 4 != 0 || (uintptr_t)d.dgx % 16 != 0 || (uintptr_t)d.dgh % 16 != 0) return M3T_EINVAL;
            regs[nr++] = M3TRegion{d.dgx + d.goff, (unsigned long long)B * T, (unsigned long long)d.ldg, 3 * d.H / 4, d.amax};
            regs[nr++] = M3TRegion{d.dgh, (unsigned long long)B * T, (unsigned long long)3 * d.H, 3 * d.H / 4, d.amax};
        }
        if (nr) {
            const int ra = m3t_absmax_regions(regs, nr, (hipStream_t)stream);
            if (ra) return ra;
        }
    }
    BiasFinish f;
    std::memset(&f, 0, sizeof(f));
    f.n = n_scans;
    int maxh = 0;
    bool any = false;
    for (int i = 0; i < n_scans; ++i) {
        const m3t_gru_bwd_desc& d = scans[i];
        if ((d.db_ih || d.db_hh) && !d.db_part) return M3T_EINVAL;
        f.part[i] = d.db_part; f.db_ih[i] = d.db_ih; f.db_hh[i] = d.db_hh; f.H[i] = d.H;
        if (d.db_part && (d.db_ih || d.db_hh)) { any = true; maxh = d.H > maxh ? d.H : maxh; }
    }
    if (any) {
        gru_bias_finish_kernel<<<dim3(cdiv(maxh, 128), n_scans), 128, 0, (hipStream_t)stream>>>(f, B);
        M3T_LAUNCH_CHECK();
    }
    return 0;
}
