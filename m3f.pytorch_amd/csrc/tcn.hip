// TCN kernels (models/tcn.py) for gfx950, channel-last [B,T,C] activations.
//   * weight-norm reparametrisation and its gradient (per-output-channel norms by one
//     wavefront per channel);
//   * the dilated causal convolution as an implicit GEMM on fp32 MFMA 32x32x2: the
//     (b,t) rows are flattened to M = B*T; a workgroup stages rows [m0-halo, m0+128) of a
//     16-channel slab ONCE in LDS (the dilated temporal receptive field) and every tap j
//     reads its shifted window from that image -- no im2col, no chomp copy, no padded
//     copy; taps that would cross t<0 (or a clip boundary) are masked per row;
//     bias, ReLU and the residual add + ReLU of the block are fused in the epilogue;
//   * the same kernel run anti-causally with the [co][ci] weight read as [K][N] is the
//     data gradient; the weight gradient is K segment-mapped TN GEMMs (m3t_sgemm).
#include "common.h"
#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDB = 132;

// M3T_BF16: operands rounded to bf16 while they are staged; products of bf16 values are exact in fp32, so the fp32
// MFMA then computes exactly a bf16 MFMA with fp32 accumulation
__device__ __forceinline__ float rbf(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ float4 rbf4(float4 v) { return make_float4(rbf(v.x), rbf(v.y), rbf(v.z), rbf(v.w)); }

struct ConvParams {
    const float* x; const float* w_t; const float* bias; const float* res; const float* mask; float* y; float* pre;
    int B, T, Ci, Co, K, dil, act, anti, halo, lead, lda, w_kn, vecx, vecw, bf16;
    M3TDrop drop;
};

// dynamic LDS: As[BK][lda] (lda = BM + halo + pad), Bs[K][BK][LDB]
template <int WKN>
__global__ __launch_bounds__(256) void causal_conv_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + BK * p.lda;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int M = p.B * p.T;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int halo = p.halo;
    // tap j of output row t reads source time  t + off_j,  off_j = lead - sft_j  (forward; lead = 0: causal)
    //                                                      off_j = sft_j - lead  (time-flipped: the data gradient)
    // LDS image row r (0 .. BM+halo-1) holds source row  m0 + minoff + r,  minoff = min_j off_j
    const int minoff = p.anti ? -p.lead : p.lead - halo;
    const int src0 = m0 + minoff;
    const int nrows = BM + halo;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-lane time index of its two A-fragment rows
    int trow[2];
    bool mrow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + l31;
        mrow[i] = m < M;
        trow[i] = m % p.T;
    }

    for (int c0 = 0; c0 < p.Ci; c0 += BK) {
        // ---- stage x rows (transposed to k-major) ----
        for (int idx = tid; idx < nrows * 4; idx += 256) {
            const int r = idx >> 2, kc = (idx & 3) * 4;
            const int m = src0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m >= 0 && m < M) {
                const float* q = p.x + (size_t)m * p.Ci + c0 + kc;
                const int nv = p.Ci - (c0 + kc);
                if (p.vecx && nv >= 4) v = *reinterpret_cast<const float4*>(q);
                else {
                    if (nv > 0) v.x = q[0];
                    if (nv > 1) v.y = q[1];
                    if (nv > 2) v.z = q[2];
                    if (nv > 3) v.w = q[3];
                }
            }
            if (p.bf16) v = rbf4(v);
            As[(kc + 0) * p.lda + r] = v.x;
            As[(kc + 1) * p.lda + r] = v.y;
            As[(kc + 2) * p.lda + r] = v.z;
            As[(kc + 3) * p.lda + r] = v.w;
        }
        // ---- stage the K weight taps of this channel slab ----
        for (int j = 0; j < p.K; ++j) {
            float* Bj = Bs + j * BK * LDB;
            if (WKN == 0) {
                // w_t[j][n][ci]: K(=ci)-contiguous rows -> transpose into k-major
                const float* wj = p.w_t + (size_t)j * p.Co * p.Ci;
                for (int idx = tid; idx < BN * 4; idx += 256) {
                    const int r = idx >> 2, kc = (idx & 3) * 4;
                    const int n = n0 + r;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (n < p.Co) {
                        const float* q = wj + (size_t)n * p.Ci + c0 + kc;
                        const int nv = p.Ci - (c0 + kc);
                        if (p.vecw && nv >= 4) v = *reinterpret_cast<const float4*>(q);
                        else {
                            if (nv > 0) v.x = q[0];
                            if (nv > 1) v.y = q[1];
                            if (nv > 2) v.z = q[2];
                            if (nv > 3) v.w = q[3];
                        }
                    }
                    if (p.bf16) v = rbf4(v);
                    Bj[(kc + 0) * LDB + r] = v.x;
                    Bj[(kc + 1) * LDB + r] = v.y;
                    Bj[(kc + 2) * LDB + r] = v.z;
                    Bj[(kc + 3) * LDB + r] = v.w;
                }
            } else {
                // data gradient: reduce over the weight's [co] rows, output its [ci] columns:
                // w_t[j][k=co][n=ci] is N-contiguous.  Here p.Ci = #rows(co) and p.Co = #cols(ci).
                const float* wj = p.w_t + (size_t)j * p.Ci * p.Co;
                for (int idx = tid; idx < BK * 32; idx += 256) {
                    const int kk = idx >> 5, c = (idx & 31) * 4;
                    const int k = c0 + kk, n = n0 + c;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k < p.Ci) {
                        const float* q = wj + (size_t)k * p.Co + n;
                        const int nv = p.Co - n;
                        if (p.vecw && nv >= 4) v = *reinterpret_cast<const float4*>(q);
                        else {
                            if (nv > 0) v.x = q[0];
                            if (nv > 1) v.y = q[1];
                            if (nv > 2) v.z = q[2];
                            if (nv > 3) v.w = q[3];
                        }
                    }
                    if (p.bf16) v = rbf4(v);
                    *reinterpret_cast<float4*>(&Bj[kk * LDB + c]) = v;
                }
            }
        }
        __syncthreads();
        for (int j = 0; j < p.K; ++j) {
            const int sft = (p.K - 1 - j) * p.dil;
            // image row of A-fragment row i for this tap
            const int off = p.anti ? sft - p.lead : p.lead - sft;
            const int roff = off - minoff;
            bool ok[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) ok[i] = mrow[i] && (unsigned)(trow[i] + off) < (unsigned)p.T;
            const float* a_s = As + wm * 64 + l31 + roff;
            const float* b_s = Bs + j * BK * LDB + wn * 64 + l31;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const float a0 = ok[0] ? a_s[(kk + hi) * p.lda] : 0.f;
                const float a1 = ok[1] ? a_s[(kk + hi) * p.lda + 32] : 0.f;
                const float b0 = b_s[(kk + hi) * LDB], b1 = b_s[(kk + hi) * LDB + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l31;
            if (col >= p.Co) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (p.drop.on && (r & 3) == 0) m3t_drop_mask4(p.drop, (uint32_t)row >> 2, (uint32_t)col, dm);     // rows row .. row+3
                if (row >= M) continue;
                const size_t o = (size_t)row * p.Co + col;
                float v = acc[i][j][r] + bv;
                if (p.pre) p.pre[o] = v;
                const float mk = p.drop.on ? dm[r & 3] : (p.mask ? p.mask[o] : 1.f);
                if (p.act == 1) v = fmaxf(v, 0.f) * mk;
                else if (p.act == 2) v = fmaxf(fmaxf(v, 0.f) * mk + p.res[o], 0.f);
                else if (p.res) v += p.res[o];
                p.y[o] = v;
            }
        }
}

// one wavefront per output channel: norm over (Ci,K); writes tap-major w_t[j][co][ci]
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                              float* __restrict__ w_t, float* __restrict__ norm, int Co,
                                                              int Ci, int K, unsigned long long* amax) {
    const int lane = threadIdx.x & 63;
    const int co = blockIdx.x * 4 + (threadIdx.x >> 6);
    float mx = 0.f;
    if (co < Co) {
        const float* vp = v + (size_t)co * Ci * K;
        float s = 0.f;
        for (int i = lane; i < Ci * K; i += 64) s += vp[i] * vp[i];
        s = wave_sum(s);
        const float nrm = sqrtf(s);
        const float sc = g[co] / nrm;
        if (lane == 0) norm[co] = nrm;
        for (int i = lane; i < Ci * K; i += 64) {
            const int ci = i / K, j = i % K;
            const float w = vp[i] * sc;
            w_t[((size_t)j * Co + co) * Ci + ci] = w;
            mx = fmaxf(mx, m3t_fin_abs(w));
        }
    }
    __shared__ float red4[4];
    if (amax) m3t_block_raise_slot(amax, mx, red4);          // (magnitude slot of the weight-normed kernel: m3t_amax_out)
}

__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dw_t, const float* __restrict__ v,
                                                              const float* __restrict__ g, const float* __restrict__ norm,
                                                              float* __restrict__ dv, float* __restrict__ dg, int Co, int Ci,
                                                              int K) {
    const int lane = threadIdx.x & 63;
    const int co = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (co >= Co) return;
    const float* vp = v + (size_t)co * Ci * K;
    float dot = 0.f;
    for (int i = lane; i < Ci * K; i += 64) {
        const int ci = i / K, j = i % K;
        dot += dw_t[((size_t)j * Co + co) * Ci + ci] * vp[i];
    }
    dot = wave_sum(dot);
    const float nrm = norm[co], gg = g[co];
    if (lane == 0) dg[co] = dot / nrm;
    const float a = gg / nrm, b = dot / (nrm * nrm);
    for (int i = lane; i < Ci * K; i += 64) {
        const int ci = i / K, j = i % K;
        dv[(size_t)co * Ci * K + i] = a * (dw_t[((size_t)j * Co + co) * Ci + ci] - vp[i] * b);
    }
}

// batched [R][C] -> [C][R] through a padded LDS tile (both sides coalesced)
// amax (optional): a magnitude slot raised to max |src| on the way (m3t_amax_out: the conv3d weight gradient's dy goes channels-last
// through here and straight into an fp16x3 GEMM)
// rowsum (optional): [gridDim.z * gridDim.x][R] partial sums of the SOURCE rows over this block's 32 columns (the conv3d bias gradient:
// a row of dy's [C_out][positions] plane is a channel) -- summed by m3t_colsum afterwards, fixed order
__global__ __launch_bounds__(256) void batched_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R,
                                                                int C, unsigned long long* __restrict__ amax, float* __restrict__ rowsum) {
    __shared__ float tile[32][33];
    __shared__ float red4[4];
    const size_t boff = (size_t)blockIdx.z * R * C;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float mx = 0.f;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        const float v = (r < R && c < C) ? src[boff + (size_t)r * C + c] : 0.f;
        tile[i][tx] = v;
        mx = fmaxf(mx, m3t_fin_abs(v));
        if (rowsum) {
            float sv = v;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) sv += __shfl_xor(sv, o, 64);      // (a wave holds two tile rows of 32 columns)
            if (tx == 0 && r < R) rowsum[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * R + r] = sv;
        }
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) dst[boff + (size_t)c * R + r] = tile[tx][i];
    }
    if (amax) m3t_block_raise_slot(amax, mx, red4);
}

// [R][C] -> the m3t_f16x3_split IMAGE of [C][R] (round 6): the transpose above with the split of its result folded in -- for a source whose
// magnitude slot its PRODUCER raised (BatchNorm's apply / dx kernels), so the scale is known before the first store.  R % 4 == 0 (R = the
// channels: four consecutive ones share a 16-byte record {hi 0|1, hi 2|3, lo 0|1, lo 2|3}); a thread packs the record of (column, channel quad).
typedef _Float16 tr_f16x2 __attribute__((ext_vector_type(2)));
typedef float tr_f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void batched_transpose_img_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C,
                                                                    const unsigned long long* __restrict__ slot, float* __restrict__ rowsum) {
    __shared__ float tile[32][33];
    const size_t boff = (size_t)blockIdx.z * R * C;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float sc, inv;
    m3t_f16_scale((unsigned)*slot, sc, inv);
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        const float v = (r < R && c < C) ? src[boff + (size_t)r * C + c] : 0.f;
        tile[i][tx] = v;
        if (rowsum) {
            float sv = v;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) sv += __shfl_xor(sv, o, 64);
            if (tx == 0 && r < R) rowsum[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * R + r] = sv;
        }
    }
    __syncthreads();
    const int q = threadIdx.x & 7, ci = threadIdx.x >> 3;       // channel quad (rows r0 + 4 q ..+3 of the source), column c0 + ci
    const int c = c0 + ci, r = r0 + 4 * q;
    if (c < C && r < R) {
        const tr_f32x2 a = (tr_f32x2){tile[4 * q][ci], tile[4 * q + 1][ci]} * sc, b = (tr_f32x2){tile[4 * q + 2][ci], tile[4 * q + 3][ci]} * sc;
        const tr_f16x2 ha = __builtin_convertvector(a, tr_f16x2), hb = __builtin_convertvector(b, tr_f16x2);
        const tr_f16x2 la = __builtin_convertvector(a - __builtin_convertvector(ha, tr_f32x2), tr_f16x2);
        const tr_f16x2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, tr_f32x2), tr_f16x2);
        float4 o;
        o.x = __uint_as_float(__builtin_bit_cast(unsigned, ha)); o.y = __uint_as_float(__builtin_bit_cast(unsigned, hb));
        o.z = __uint_as_float(__builtin_bit_cast(unsigned, la)); o.w = __uint_as_float(__builtin_bit_cast(unsigned, lb));
        *reinterpret_cast<float4*>(dst + boff + (size_t)c * R + r) = o;
    }
}

}  // namespace

extern "C" int m3t_bct_to_btc_img(const float* src, float* dst_img, int B, int C, int T, const unsigned long long* slot, float* part, void* stream) {
    if (B <= 0 || C <= 0 || T <= 0) return 0;
    if (!src || !dst_img || !slot || C % 4 != 0 || (uintptr_t)dst_img % 16 != 0) return M3T_EINVAL;
    batched_transpose_img_kernel<<<dim3(cdiv(T, 32), cdiv(C, 32), B), 256, 0, (hipStream_t)stream>>>(src, dst_img, C, T, slot, part);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_weight_norm_fwd(const float* v, const float* g, float* w_t, float* norm, int Co, int Ci, int K,
                                   void* stream) {
    unsigned long long* amax = m3t_take_amax_out();
    if (Co <= 0 || Ci <= 0 || K <= 0 || !v || !g || !w_t || !norm) return M3T_EINVAL;
    weight_norm_fwd_kernel<<<cdiv(Co, 4), 256, 0, (hipStream_t)stream>>>(v, g, w_t, norm, Co, Ci, K, amax);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_weight_norm_bwd(const float* dw_t, const float* v, const float* g, const float* norm, float* dv,
                                   float* dg, int Co, int Ci, int K, void* stream) {
    if (Co <= 0 || Ci <= 0 || K <= 0 || !dw_t || !v || !g || !norm || !dv || !dg) return M3T_EINVAL;
    weight_norm_bwd_kernel<<<cdiv(Co, 4), 256, 0, (hipStream_t)stream>>>(dw_t, v, g, norm, dv, dg, Co, Ci, K);
    M3T_LAUNCH_CHECK();
    return 0;
}

int m3t_conv_x6d_launch(const float* x, const float* w_t, const float* bias, const float* res, const float* mask, float* y,
                        float* pre, int B, int T, int Ci, int Co, int K, int dil, int lead, int act, int anti, int bf16_operands,
                        M3TDrop drop, hipStream_t s);
int m3t_conv_x6_launch(const float* x, const float* w_t, const float* bias, const float* res, const float* mask, float* y,
                       float* pre, int B, int T, int Ci, int Co, int K, int dil, int lead, int act, int anti, int bf16_operands,
                       M3TDrop drop, const unsigned long long* amax_a, const unsigned long long* amax_b, unsigned long long* amax_y,
                       hipStream_t s);

static bool conv_x6_enabled() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("M3T_CONV_X6");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    return on == 1;
}

extern "C" int m3t_conv1d_fwd(const float* x, const float* w_t, const float* bias, const float* res,
                              const float* drop_mask, float* y, float* pre, int B, int T, int Ci, int Co, int K,
                              int dilation, int lead, int act, int anticausal, float drop_p, unsigned long long drop_seed,
                              int flags, void* stream) {
    return m3t_conv1d_fwd_scaled(x, w_t, bias, res, drop_mask, y, pre, B, T, Ci, Co, K, dilation, lead, act, anticausal, drop_p, drop_seed,
                                 flags, nullptr, nullptr, stream);
}

extern "C" int m3t_conv1d_fwd_scaled(const float* x, const float* w_t, const float* bias, const float* res,
                                     const float* drop_mask, float* y, float* pre, int B, int T, int Ci, int Co, int K,
                                     int dilation, int lead, int act, int anticausal, float drop_p, unsigned long long drop_seed,
                                     int flags, const unsigned long long* amax_x, const unsigned long long* amax_w, void* stream) {
    unsigned long long* amax_y = m3t_take_amax_out();      // m3t_amax_out: raise this slot to max |y| (epilogue of the implicit GEMM)
    if (B <= 0 || T <= 0) return 0;
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && drop_mask)) return M3T_EINVAL;
    if (Ci <= 0 || Co <= 0 || K <= 0 || dilation <= 0 || !x || !w_t || !y) return M3T_EINVAL;
    if (lead < 0 || lead > (K - 1) * dilation) return M3T_EINVAL;
    if (act == 2 && !res) return M3T_EINVAL;
    ConvParams p;
    p.x = x; p.w_t = w_t; p.bias = bias; p.res = res; p.mask = drop_mask; p.y = y; p.pre = pre;
    p.B = B; p.T = T; p.Ci = Ci; p.Co = Co; p.K = K; p.dil = dilation; p.act = act; p.anti = anticausal; p.lead = lead; p.bf16 = (flags & M3T_BF16) ? 1 : 0;
    p.drop = m3t_make_drop(drop_p, drop_seed);
    // interior shapes: the implicit GEMM on the bf16 matrix pipe (gemm_x6.hip, CONV): fp32-accurate bf16x6 products (one bf16
    // product in the M3T_BF16 mode), 128 x 128 x 32 tiles with the next tile's loads in flight during the MFMAs -- 2-3x this
    // file's single-stage fp32-MFMA kernel, which keeps the edge shapes (channels not multiples of 32 / 128, ragged B*T)
    {
        auto al = [](const void* q) { return q == nullptr || ((uintptr_t)q % 16) == 0; };
        if (conv_x6_enabled() && ((size_t)B * T) % 128 == 0 && Co % 128 == 0 && Ci % 32 == 0 && al(x) && al(w_t) && al(y) && al(res) &&
            al(drop_mask) && al(pre)) {
            // (the software-pipelined CONV form of gemm_x6d.hip was no faster on these 300-tile grids than the 128 x 64 tiles of
            // gemm_x6.hip -- C1 1.96 vs 2.03 ms, C2 7.44 vs 7.35 ms, round 2 -- and is no longer dispatched)
            const int mode = p.bf16 ? 1 : ((flags & M3T_GEMM_HIGH) ? 2 : (((flags & M3T_GEMM_F16X3) && m3t_f16x3_enabled()) ? 3 : 0));
            const unsigned long long* ua = nullptr; const unsigned long long* ub = nullptr;
            if (mode == 3) {                       // fp16x3: max |x|, max |w| (one launch, this stream)
                const M3TRegion rx{x, (unsigned long long)B * T, (unsigned long long)Ci, Ci / 4, nullptr};
                const M3TRegion rw{w_t, (unsigned long long)K * (anticausal ? Ci : Co), (unsigned long long)(anticausal ? Co : Ci),
                                   (anticausal ? Co : Ci) / 4, nullptr};
                const int rm = m3t_f16x3_measure(rx, amax_x, rw, amax_w, &ua, &ub, (hipStream_t)stream);
                if (rm) return rm;
            }
            return m3t_conv_x6_launch(x, w_t, bias, res, drop_mask, y, pre, B, T, Ci, Co, K, dilation, lead, act, anticausal,
                                      mode, p.drop, ua, ub, amax_y, (hipStream_t)stream);
        }
    }
    const int halo = (K - 1) * dilation;
    p.halo = halo;
    p.lda = ((BM + halo - 4 + 31) / 32) * 32 + 4;   // == 4 (mod 32): 2-way (free) transposing LDS writes
    p.vecx = (Ci % 4 == 0) && ((uintptr_t)x % 16 == 0);
    p.w_kn = anticausal ? 1 : 0;
    p.vecw = anticausal ? ((Co % 4 == 0) && ((uintptr_t)w_t % 16 == 0)) : ((Ci % 4 == 0) && ((uintptr_t)w_t % 16 == 0));
    const size_t lds = ((size_t)BK * p.lda + (size_t)K * BK * LDB) * sizeof(float);
    if (lds > 160 * 1024) return M3T_EINVAL;
    dim3 grid(cdiv(Co, BN), cdiv(B * T, BM));
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (anticausal) {
        e = hipFuncSetAttribute((const void*)causal_conv_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        causal_conv_kernel<1><<<grid, 256, lds, s>>>(p);
    } else {
        e = hipFuncSetAttribute((const void*)causal_conv_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        causal_conv_kernel<0><<<grid, 256, lds, s>>>(p);
    }
    M3T_LAUNCH_CHECK();
    if (amax_y) {                                    // edge shapes: the slot the caller asked for (m3t_amax_out), by one pass over y
        if (Co % 4 != 0 || ((uintptr_t)y % 16) != 0) return M3T_EINVAL;
        const M3TRegion ry{y, (unsigned long long)B * T, (unsigned long long)Co, Co / 4, amax_y};
        return m3t_absmax_regions(&ry, 1, s);
    }
    return 0;
}

extern "C" int m3t_causal_conv_fwd(const float* x, const float* w_t, const float* bias, const float* res,
                                   const float* drop_mask, float* y, float* pre, int B, int T, int Ci, int Co, int K,
                                   int dilation, int act, int anticausal, void* stream) {
    return m3t_conv1d_fwd(x, w_t, bias, res, drop_mask, y, pre, B, T, Ci, Co, K, dilation, 0, act, anticausal, 0.f, 0ull, 0, stream);
}

extern "C" int m3t_conv1d_wgrad(const float* dy, const float* x, float* dw_t, int B, int T, int Ci, int Co, int K,
                                int dilation, int lead, float* ws, size_t ws_bytes, int flags, void* stream) {
    return m3t_conv1d_wgrad_scaled(dy, x, dw_t, B, T, Ci, Co, K, dilation, lead, ws, ws_bytes, flags, nullptr, nullptr, stream);
}

extern "C" int m3t_conv1d_wgrad_scaled(const float* dy, const float* x, float* dw_t, int B, int T, int Ci, int Co, int K,
                                       int dilation, int lead, float* ws, size_t ws_bytes, int flags, const unsigned long long* amax_dy,
                                       const unsigned long long* amax_x, void* stream) {
    if (Ci <= 0 || Co <= 0 || K <= 0 || !dy || !x || !dw_t) return M3T_EINVAL;
    if (lead < 0 || lead > (K - 1) * dilation) return M3T_EINVAL;
    const int gflags = (flags & M3T_BF16) ? M3T_GEMM_BF16 : (flags & (M3T_GEMM_HIGH | M3T_GEMM_F16X3));
    const unsigned long long* ua = nullptr; const unsigned long long* ub = nullptr;
    if ((gflags & M3T_GEMM_F16X3) && m3t_f16x3_enabled() && !(gflags & M3T_GEMM_HIGH) && B > 0 && Co % 4 == 0 && Ci % 4 == 0 && (uintptr_t)dy % 16 == 0 &&
        (uintptr_t)x % 16 == 0) {
        // fp16x3: the K taps multiply the same two tensors -- measure them once
        const M3TRegion rd{dy, (unsigned long long)B * T, (unsigned long long)Co, Co / 4, nullptr};
        const M3TRegion rx{x, (unsigned long long)B * T, (unsigned long long)Ci, Ci / 4, nullptr};
        const int rm = m3t_f16x3_measure(rd, amax_dy, rx, amax_x, &ua, &ub, (hipStream_t)stream);
        if (rm) return rm;
    }
    for (int j = 0; j < K; ++j) {
        const int off = lead - (K - 1 - j) * dilation;     // x row = dy row + off
        const int aoff = off < 0 ? -off : 0, boff = off > 0 ? off : 0, span = aoff + boff;
        float* out = dw_t + (size_t)j * Co * Ci;
        if (span >= T || B <= 0) {
            hipError_t e = hipMemsetAsync(out, 0, (size_t)Co * Ci * sizeof(float), (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
            continue;
        }
        // dw_t[j][co][ci] = sum_b sum_t dy[b,t,co] * x[b,t+off,ci] over the t with both rows inside the clip
        const int rc = m3t_sgemm_scaled(1, 0, Co, Ci, B * (T - span), dy, Co, x, Ci, out, Ci, nullptr, 0, 0, T - span, T, aoff, boff,
                                        ws, ws_bytes, gflags, ua, ub, stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int m3t_causal_conv_wgrad(const float* dy, const float* x, float* dw_t, int B, int T, int Ci, int Co, int K,
                                     int dilation, float* ws, size_t ws_bytes, void* stream) {
    return m3t_conv1d_wgrad(dy, x, dw_t, B, T, Ci, Co, K, dilation, 0, ws, ws_bytes, 0, stream);
}

extern "C" int m3t_bct_to_btc(const float* src, float* dst, int B, int C, int T, void* stream) {
    unsigned long long* amax = m3t_take_amax_out();
    if (B <= 0 || C <= 0 || T <= 0) return 0;
    batched_transpose_kernel<<<dim3(cdiv(T, 32), cdiv(C, 32), B), 256, 0, (hipStream_t)stream>>>(src, dst, C, T, amax, nullptr);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_bct_to_btc_sums(const float* src, float* dst, int B, int C, int T, float* part, void* stream) {
    unsigned long long* amax = m3t_take_amax_out();
    if (B <= 0 || C <= 0 || T <= 0) return 0;
    if (!src || !dst || !part) return M3T_EINVAL;
    batched_transpose_kernel<<<dim3(cdiv(T, 32), cdiv(C, 32), B), 256, 0, (hipStream_t)stream>>>(src, dst, C, T, amax, part);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_btc_to_bct(const float* src, float* dst, int B, int T, int C, void* stream) {
    if (B <= 0 || C <= 0 || T <= 0) return 0;
    batched_transpose_kernel<<<dim3(cdiv(C, 32), cdiv(T, 32), B), 256, 0, (hipStream_t)stream>>>(src, dst, T, C, nullptr, nullptr);
    M3T_LAUNCH_CHECK();
    return 0;
}
