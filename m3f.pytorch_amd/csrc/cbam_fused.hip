// CBAM (channel gate -> spatial gate, reference models/cbam.py:95-111) as ONE fused operator for gfx950.
//
// cbam.hip runs the two gates as two operators (what the reference's module tree says): x -> y1 = x * cs -> y2 = y1 * ss,
// with y1 written and read back three times and ~14 launches per forward + backward: 10 HBM passes over a tensor of x's
// size, 0.7-1.7 TB/s effective.  The gate as a whole needs much less.  y1 is never materialised here:
//
//   forward   F1  per frame (one workgroup): read x -> per-channel avg / max / argmax (G lanes per plane, shuffles) -> shared MLP ->
//                 cs[c] -> re-read the frame (L2 / Infinity Cache: the workgroup touched it microseconds ago) -> per-pixel
//                 max_c / mean_c / argmax_c of x * cs -> 5x5 conv out of LDS -> conv[n,p] + fp64 BatchNorm partials
//             ST  one workgroup: batch statistics (BatchNorm2d(1) in train mode is a GLOBAL reduction over N*H*W: the one
//                 place where the gate must be cut), running statistics
//             F2  per frame: ss[p] = sigmoid(gamma * xhat + beta); y = x * cs[c] * ss[p]
//   backward  B1  per frame: dss[p] = sum_c dy * x * cs[c] -> dpre = dss * ss (1 - ss) + fp64 partials of the two BatchNorm sums
//             SP  one workgroup: (d gamma, d beta)  -- the second global reduction
//             B2  per frame: BatchNorm backward -> conv backward (weight-gradient partials per frame, data gradient to the two
//                 compressed maps) -> dy1 = dy * ss + dmean / C + [c == argmax_c] dmax, recomputed on the fly -> dcs[c] = sum_p
//                 dy1 * x -> MLP backward -> dx = dy1 * cs + davg / HW + [p == argmax_p] dmaxc (dy re-read while cache-hot)
//             + the parameter-gradient finish (GEMMs / column sums over the [N, .] slabs, as cbam.hip)
//
// HBM passes over a tensor of x's size: F1 1, F2 2, B1 2, B2 3 = 8 (+2 re-reads served by L2 / Infinity Cache), 7 launches
// + the small parameter-gradient kernels.  Two global BatchNorm reductions make four sweeps the minimum in train mode.
// Thread mappings: a "unit" is 4 consecutive pixels (float4, H*W % 4 == 0) or 1 pixel; Q units per plane.
//   * plane sweeps (squeeze, apply, dcs, dx): G = min(64, pow2 >= Q) lanes per plane, 64 / G planes per wave pass, up to four
//     16-B loads in flight per lane -- a 4 x 4 map (Q = 4) keeps all 64 lanes busy on 16 planes, a 28 x 28 map one plane;
//   * pixel reductions over channels (compress, dss): thread = (unit q, channel slice k), KS = 512 / pow2(Q) slices, partials
//     combined by shuffles (inside a wave) and LDS, fixed order, ties to the lower channel index (torch's first maximum).
// Deterministic: no atomics; every reduction has a fixed order.  fp64 only for the per-workgroup BatchNorm partials.
#include "common.h"

namespace {

constexpr int FT = 512;       // threads per workgroup (8 waves)

// phase stamps of F1 / B2 for tools/cbam_phase_probe.hip (which includes this file with M3T_CBAM_STAMPS): compiled out of the library
#ifdef M3T_CBAM_STAMPS
__device__ unsigned long long g_cb_stamps[8192 * 8];
#define M3T_CB_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_cb_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define M3T_CB_STAMP(i) do { } while (0)
#endif
// planes in flight per lane in the plane sweeps of SMALL maps (one unit per lane per plane).  More does not pay: at 8 / 16 / 32 hipcc
// keeps every plane's bookkeeping live (86 / 151 / 181 VGPRs in F1), one workgroup per CU instead of 2.6, and the sweep is not bound
// by loads in flight but by the per-plane cross-lane reductions (tools/cbam_phase_probe.hip: 20 us per 256 x 7 x 7 frame even as one
// batch).  Frames that fit LDS take the frame-resident kernels below (F1L / B2L), which have no cross-lane reductions at all.
constexpr int PB = 4;

// Reductions over aligned groups of G adjacent lanes (G a power of two, wave-uniform) WITHOUT the LDS pipeline.  __shfl_xor is a
// ds_bpermute: with three of them per butterfly step and six steps per plane, the squeeze of a 256 x 7 x 7 frame queued 4 600 wave-wide
// LDS operations, more time than its loads.  Inside a row of 16 lanes the steps are DPP operands of the add / compare itself
// (quad_perm, then the two mirrors: after steps 1 and 2 a quad holds one value, so the mirror image is "the other quad"); across
// rows gfx950 has v_permlane16_swap / v_permlane32_swap, one VALU move that hands every lane both halves.
// All lanes of the wave must be active.  Order: 1, 2, 4, 8, 16, 32 (fixed, the same in every lane: deterministic).
template <int CTRL> __device__ __forceinline__ float dppf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
constexpr int DPP_X1 = 0xB1, DPP_X2 = 0x4E, DPP_HM = 0x141, DPP_RM = 0x140;      // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror

__device__ __forceinline__ float group_sum(float v, int G) {
    if (G >= 2) v += dppf<DPP_X1>(v);
    if (G >= 4) v += dppf<DPP_X2>(v);
    if (G >= 8) v += dppf<DPP_HM>(v);
    if (G >= 16) v += dppf<DPP_RM>(v);
    if (G >= 32) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    if (G >= 64) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return v;
}

// (max, first index of the max) with torch's tie rule (the lower index wins): symmetric, so both partners end with the same pair
__device__ __forceinline__ void amax_take(float& mx, int& am, float ov, int oi) {
    if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
}
__device__ __forceinline__ void group_sum_argmax(float& sum, float& mx, int& am, int G) {
#define M3T_CB_STEP(CTRL)                                              \
    do {                                                               \
        sum += dppf<CTRL>(sum);                                        \
        const float ov = dppf<CTRL>(mx);                               \
        const int oi = dppi<CTRL>(am);                                 \
        amax_take(mx, am, ov, oi);                                     \
    } while (0)
    if (G >= 2) M3T_CB_STEP(DPP_X1);
    if (G >= 4) M3T_CB_STEP(DPP_X2);
    if (G >= 8) M3T_CB_STEP(DPP_HM);
    if (G >= 16) M3T_CB_STEP(DPP_RM);
#undef M3T_CB_STEP
#define M3T_CB_SWAP(B)                                                                                              \
    do {                                                                                                            \
        const auto rs = B(__float_as_uint(sum), __float_as_uint(sum), false, false);                                \
        const auto rm = B(__float_as_uint(mx), __float_as_uint(mx), false, false);                                  \
        const auto ri = B((unsigned)am, (unsigned)am, false, false);                                                \
        sum = __uint_as_float(rs[0]) + __uint_as_float(rs[1]);                                                      \
        mx = __uint_as_float(rm[0]); am = (int)ri[0];                                                               \
        amax_take(mx, am, __uint_as_float(rm[1]), (int)ri[1]);                                                      \
    } while (0)
    if (G >= 32) M3T_CB_SWAP(__builtin_amdgcn_permlane16_swap);
    if (G >= 64) M3T_CB_SWAP(__builtin_amdgcn_permlane32_swap);
#undef M3T_CB_SWAP
}

// both BatchNorm partial sums of a frame in one pass (two barriers instead of four): red holds 2 * FT / 64 doubles
__device__ __forceinline__ void block_sum2_d8(double& a, double& b, double* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    __syncthreads();
    if (lane == 0) { red[w] = a; red[FT / 64 + w] = b; }
    __syncthreads();
    double ra = 0.0, rb = 0.0;
#pragma unroll
    for (int i = 0; i < FT / 64; ++i) { ra += red[i]; rb += red[FT / 64 + i]; }
    a = ra; b = rb;
}

// zero-padded copies of H x W maps for the 5 x 5 convolutions: (H + 4) x (W + 4), pixel (y, x) at (y + 2) * (W + 4) + x + 2
__host__ __device__ inline int pad_len(int H, int W) { return (H + 4) * (W + 4); }

template <int E> struct Unit;
template <> struct Unit<4> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    static __device__ __forceinline__ void ldi(const int32_t* p, int (&v)[4]) {
        const int4 t = *reinterpret_cast<const int4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void sti(int32_t* p, const int (&v)[4]) {
        *reinterpret_cast<int4*>(p) = make_int4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Unit<1> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void st(float* p, const float (&v)[1]) { *p = v[0]; }
    static __device__ __forceinline__ void ldi(const int32_t* p, int (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void sti(int32_t* p, const int (&v)[1]) { *p = v[0]; }
};

__host__ __device__ inline int al4(int n) { return (n + 3) & ~3; }

// ---- the gate's shared MLP for one frame (phase b of F1 / F1L).  In: s_avg, s_max [C] in LDS (16-B aligned).  Out: s_sc [C] (sigmoid),
// the slabs hidden [2][Cr] (pre-ReLU) and cs [C]; s_h [2 Cr] is scratch.  Ends WITHOUT a barrier: the caller syncs before reading s_sc.
__device__ __forceinline__ void gate_mlp_fwd(const float* s_avg, const float* s_max, float* s_h, float* s_sc, const float* __restrict__ w1,
                                             const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                             float* __restrict__ hidden, float* __restrict__ cs, int n, int C, int Cr) {
    const int tid = threadIdx.x;
    // ---- b. shared MLP C -> Cr -> C on both pooled vectors, sigmoid of the sum (reference cbam.py:51-58).  Both layers read their
    // weight matrix as float4 streams with every load of the frame in flight before the first use and every byte of a fetched line
    // used by the instruction that fetched it.  (Round 3 read W2 a row per LANE: 128-B stride, Cr load instructions of 64 lines each,
    // ~2 000 cycles of the CU's one address path per wave -- 55 of the 73 us of F1 at the 512 x 4 x 4 stage; and W1 / W2 in loops of
    // unknown trip count, which hipcc leaves as one load per L2 round trip.)
    // First layer: 16 lanes per hidden unit, one W1 row feeds both pooled vectors.
    const bool vec1 = (C & 3) == 0 && ((uintptr_t)w1 & 15) == 0;
    for (int r0 = 0; r0 < Cr; r0 += FT / 16) {
        const int r = r0 + (tid >> 4), part = tid & 15;
        float ha = 0.f, hm = 0.f;
        const float bias1 = r < Cr ? b1[r] : 0.f;            // in flight with the W1 loads, not a round trip of its own after them
        if (r < Cr) {
            const float* wr = w1 + (size_t)r * C;
            if (vec1) {
                const float4* w4 = reinterpret_cast<const float4*>(wr);
                const float4* a4 = reinterpret_cast<const float4*>(s_avg);
                const float4* m4 = reinterpret_cast<const float4*>(s_max);
                const int n4 = C >> 2;
                for (int c0 = part; c0 < n4; c0 += 4 * 16) {
                    float4 w[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c4 = c0 + 16 * k;
                        w[k] = c4 < n4 ? w4[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c4 = c0 + 16 * k;
                        if (c4 < n4) {
                            const float4 a = a4[c4], m = m4[c4];
                            ha += (w[k].x * a.x + w[k].y * a.y) + (w[k].z * a.z + w[k].w * a.w);
                            hm += (w[k].x * m.x + w[k].y * m.y) + (w[k].z * m.z + w[k].w * m.w);
                        }
                    }
                }
            } else {
                for (int c = part; c < C; c += 16) { ha += wr[c] * s_avg[c]; hm += wr[c] * s_max[c]; }
            }
        }
        ha = group_sum(ha, 16);
        hm = group_sum(hm, 16);
        if (r < Cr && part == 0) {
            ha += bias1; hm += bias1;
            hidden[((size_t)n * 2 + 0) * Cr + r] = ha;
            hidden[((size_t)n * 2 + 1) * Cr + r] = hm;
            s_h[r] = fmaxf(ha, 0.f);
            s_h[Cr + r] = fmaxf(hm, 0.f);
        }
    }
    __syncthreads();
    M3T_CB_STAMP(2);
    // Second layer: W2 [C][Cr] as ONE flat float4 stream, L = Cr / 4 adjacent lanes per channel
    if ((Cr & 3) == 0 && (Cr & (Cr - 1)) == 0 && ((uintptr_t)w2 & 15) == 0) {
        const int L = Cr >> 2, lg = __ffs(L) - 1, sub2 = tid & (L - 1), n4 = C * L;
        const float4* w4 = reinterpret_cast<const float4*>(w2);
        const float4 h0 = reinterpret_cast<const float4*>(s_h)[sub2], h1 = reinterpret_cast<const float4*>(s_h + Cr)[sub2];
        for (int i0 = 0; i0 < n4; i0 += 4 * FT) {
            float4 w[4];
            float bb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * FT + tid;
                w[k] = i < n4 ? w4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                bb[k] = i < n4 ? b2[i >> lg] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (i0 + k * FT >= n4) break;           // workgroup-uniform
                const int i = i0 + k * FT + tid;
                float a0 = (w[k].x * h0.x + w[k].y * h0.y) + (w[k].z * h0.z + w[k].w * h0.w);
                float a1 = (w[k].x * h1.x + w[k].y * h1.y) + (w[k].z * h1.z + w[k].w * h1.w);
                a0 = group_sum(a0, L);
                a1 = group_sum(a1, L);
                if (i < n4 && sub2 == 0) {
                    const int c = i >> lg;
                    const float sc = 1.f / (1.f + expf(-((bb[k] + a0) + (bb[k] + a1))));
                    s_sc[c] = sc;
                    cs[(size_t)n * C + c] = sc;
                }
            }
        }
    } else {
        for (int c = tid; c < C; c += FT) {
            const float* wr = w2 + (size_t)c * Cr;
            float a0 = b2[c], a1 = b2[c];
            for (int r = 0; r < Cr; ++r) { a0 += wr[r] * s_h[r]; a1 += wr[r] * s_h[Cr + r]; }
            const float sc = 1.f / (1.f + expf(-(a0 + a1)));
            s_sc[c] = sc;
            cs[(size_t)n * C + c] = sc;
        }
    }
}

// ---- the spatial gate's 5 x 5 convolution (Conv2d(2, 1, 5, pad 2, bias=False), reference cbam.py:70-77) of the two compressed maps and
// the frame's BatchNorm partial sums (phase d).  s_pad: the maps zero-padded in LDS ([2][pad_len]), so a pixel is 50 unconditional
// multiply-adds; the weights are read with constant indices from the kernel argument: scalar loads, SGPR operands.  (Round 3 tested
// the bounds of every tap and read the weights from LDS: 3 us per frame on the one wave that has pixels at the small stages.)
__device__ __forceinline__ void spatial_conv_fwd(const float* s_pad, const float* __restrict__ convw, double* red, float* __restrict__ conv_out,
                                                 double* __restrict__ part, int n, int H, int W) {
    const int tid = threadIdx.x, HW = H * W, Wp = W + 4, PP = pad_len(H, W);
    double s1 = 0.0, s2 = 0.0;
    for (int p = tid; p < HW; p += FT) {
        const int h = p / W, ww = p - h * W;
        const float* c0 = s_pad + h * Wp + ww;
        float acc = 0.f;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) acc += convw[(ch * 5 + i) * 5 + j] * c0[ch * PP + i * Wp + j];
        conv_out[(size_t)n * HW + p] = acc;
        s1 += acc; s2 += (double)acc * acc;
    }
    M3T_CB_STAMP(6);
    block_sum2_d8(s1, s2, red);
    if (tid == 0) { part[2 * (size_t)n] = s1; part[2 * (size_t)n + 1] = s2; }
}

// ------------------------------------------------------------------------------------------------ F1
template <int E, bool SMALL>
__global__ __launch_bounds__(FT) void cbam_f1_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2,
                                                     const float* __restrict__ b2, const float* __restrict__ convw,
                                                     float* __restrict__ cs, int32_t* __restrict__ argmax_p,
                                                     float* __restrict__ pooled, float* __restrict__ hidden,
                                                     float* __restrict__ comp, int32_t* __restrict__ cargmax,
                                                     float* __restrict__ conv_out, double* __restrict__ part, int C, int Cr,
                                                     int H, int W, int G, int Qp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ double red[2 * FT / 64];
    const int HW = H * W, Q = HW / E, PP = pad_len(H, W);
    float* s_avg = sm;
    float* s_max = s_avg + al4(C);
    float* s_sc = s_max + al4(C);
    float* s_h = s_sc + al4(C);               // [2 Cr]
    float* s_pad = s_h + al4(2 * Cr);         // [2][PP] the compressed maps, zero-padded
    float* p_mx = s_pad + al4(2 * PP);        // [nk][Qp][E]
    const int nk = Qp < 64 ? FT / 64 : FT / Qp;
    float* p_sum = p_mx + nk * Qp * E;
    int* p_am = reinterpret_cast<int*>(p_sum + nk * Qp * E);
    int* s_amp = p_am + nk * Qp * E;          // [C] argmax_p
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xb = x + (size_t)n * C * HW;
    for (int i = tid; i < 2 * PP; i += FT) s_pad[i] = 0.f;
    M3T_CB_STAMP(0);

    // ---- a. channel squeeze: avg, max, argmax per plane, G lanes per plane.  SMALL (Q <= G: a plane is ONE unit per lane): four
    // plane groups in flight per wave pass instead of four units of one plane
    {
        const int per = 64 / G, sub = lane % G, pi = lane / G;
        if (SMALL) {
            const int step = (FT / 64) * per;
            for (int c0 = wave * per; c0 < C; c0 += PB * step) {
                float v[PB][E];
                bool ok[PB];
#pragma unroll
                for (int k = 0; k < PB; ++k) {
                    const int c = c0 + k * step + pi;
                    ok[k] = c < C && sub < Q;
                    if (ok[k]) Unit<E>::ld(xb + (size_t)c * HW + (size_t)sub * E, v[k]);
                }
#pragma unroll
                for (int k = 0; k < PB; ++k) {
                    const int c = c0 + k * step + pi;
                    float sum = 0.f, mx = -INFINITY;
                    int am = 0x7fffffff;
                    if (ok[k]) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            sum += v[k][e];
                            if (v[k][e] > mx) { mx = v[k][e]; am = sub * E + e; }
                        }
                    }
                    group_sum_argmax(sum, mx, am, G);
                    if (sub == 0 && c < C) {
                        const float avg = sum / (float)HW;
                        s_avg[c] = avg; s_max[c] = mx; s_amp[c] = am;
                    }
                }
            }
        } else
        for (int c0 = wave * per; c0 < C; c0 += (FT / 64) * per) {
            const int c = c0 + pi;
            const bool okc = c < C;
            const float* pl = xb + (size_t)(okc ? c : 0) * HW;
            float sum = 0.f, mx = -INFINITY;
            int am = 0x7fffffff;
            for (int u0 = sub; u0 < Q; u0 += 4 * G) {
                float v[4][E];
                bool ok[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int u = u0 + k * G;
                    ok[k] = okc && u < Q;
                    if (ok[k]) Unit<E>::ld(pl + (size_t)u * E, v[k]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ok[k]) {
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            sum += v[k][e];
                            if (v[k][e] > mx) { mx = v[k][e]; am = (u0 + k * G) * E + e; }
                        }
                    }
            }
            group_sum_argmax(sum, mx, am, G);
            if (sub == 0 && okc) {
                const float avg = sum / (float)HW;
                s_avg[c] = avg; s_max[c] = mx; s_amp[c] = am;
            }
        }
    }
    __syncthreads();
    M3T_CB_STAMP(1);
    for (int c = tid; c < C; c += FT) {        // the frame's slab rows, coalesced (not three single-lane stores per plane)
        pooled[((size_t)n * 2 + 0) * C + c] = s_avg[c];
        pooled[((size_t)n * 2 + 1) * C + c] = s_max[c];
        argmax_p[(size_t)n * C + c] = s_amp[c];
    }
    gate_mlp_fwd(s_avg, s_max, s_h, s_sc, w1, b1, w2, b2, hidden, cs, n, C, Cr);
    __syncthreads();
    M3T_CB_STAMP(3);
    // ---- c. compress x * cs over channels: (max, mean, argmax) per pixel; thread = (unit q, channel slice k)
    {
        const int q = tid % Qp, k = tid / Qp, KS = FT / Qp;
        float mx[E], sum[E];
        int am[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { mx[e] = -INFINITY; sum[e] = 0.f; am[e] = 0x7fffffff; }
        if (q < Q) {
            constexpr int CB = 8;        // channels in flight per thread: this sweep is the frame's second read (L2 / Infinity Cache), a
                                         // chain of C / KS / CB round trips -- at 4 in flight 12.6 of F1's 53 us per 64 x 28 x 28 frame
            for (int c0 = k; c0 < C; c0 += CB * KS) {
                float v[CB][E];
                bool ok[CB];
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    const int c = c0 + j * KS;
                    ok[j] = c < C;
                    if (ok[j]) Unit<E>::ld(xb + ((size_t)c * Q + q) * E, v[j]);
                }
#pragma unroll
                for (int j = 0; j < CB; ++j)
                    if (ok[j]) {
                        const int c = c0 + j * KS;
                        const float sc = s_sc[c];
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const float t = v[j][e] * sc;
                            sum[e] += t;
                            if (t > mx[e]) { mx[e] = t; am[e] = c; }
                        }
                    }
            }
        }
        if (Qp < 64) {        // the slices of a unit sit Qp lanes apart inside the wave
            for (int o = Qp; o < 64; o <<= 1) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    sum[e] += __shfl_xor(sum[e], o, 64);
                    const float ov = __shfl_xor(mx[e], o, 64);
                    const int oi = __shfl_xor(am[e], o, 64);
                    if (ov > mx[e] || (ov == mx[e] && oi < am[e])) { mx[e] = ov; am[e] = oi; }
                }
            }
            if (lane < Qp) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int idx = (wave * Qp + q) * E + e;
                    p_mx[idx] = mx[e]; p_sum[idx] = sum[e]; p_am[idx] = am[e];
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int idx = (k * Qp + q) * E + e;
                p_mx[idx] = mx[e]; p_sum[idx] = sum[e]; p_am[idx] = am[e];
            }
        }
        __syncthreads();
        M3T_CB_STAMP(4);
        if (tid < Q) {
            float fm[E], fs[E];
            int fa[E];
#pragma unroll
            for (int e = 0; e < E; ++e) { fm[e] = -INFINITY; fs[e] = 0.f; fa[e] = 0x7fffffff; }
            for (int j = 0; j < nk; ++j) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int idx = (j * Qp + tid) * E + e;
                    fs[e] += p_sum[idx];
                    const float ov = p_mx[idx];
                    const int oi = p_am[idx];
                    if (ov > fm[e] || (ov == fm[e] && oi < fa[e])) { fm[e] = ov; fa[e] = oi; }
                }
            }
            float mean[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                mean[e] = fs[e] / (float)C;
                const int pp = tid * E + e, ph = pp / W, ip = (ph + 2) * (W + 4) + (pp - ph * W) + 2;
                s_pad[ip] = fm[e];
                s_pad[PP + ip] = mean[e];
            }
            Unit<E>::st(comp + ((size_t)n * 2 + 0) * HW + (size_t)tid * E, fm);
            Unit<E>::st(comp + ((size_t)n * 2 + 1) * HW + (size_t)tid * E, mean);
            Unit<E>::sti(cargmax + (size_t)n * HW + (size_t)tid * E, fa);
        }
    }
    __syncthreads();
    M3T_CB_STAMP(5);
    spatial_conv_fwd(s_pad, convw, red, conv_out, part, n, H, W);
    M3T_CB_STAMP(7);
}

// batch statistics of the conv output (train) or the running ones (eval) -> stats = (mean, 1 / sqrt(var + eps))
__global__ __launch_bounds__(256) void cbam_stats_kernel(const double* __restrict__ part, int nparts, double cnt,
                                                         float* __restrict__ run_mean, float* __restrict__ run_var,
                                                         float* __restrict__ stats, int training, float momentum, float eps) {
    __shared__ double red[4];
    double s1 = 0.0, s2 = 0.0;
    if (training)
        for (int i = threadIdx.x; i < nparts; i += 256) { s1 += part[2 * i]; s2 += part[2 * i + 1]; }
    for (int pass = 0; pass < 2; ++pass) {
        double v = pass ? s2 : s1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        v = (red[0] + red[1]) + (red[2] + red[3]);
        if (pass) s2 = v; else s1 = v;
    }
    if (threadIdx.x == 0) {
        if (training) {
            const double mean = s1 / cnt;
            double var = s2 / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            stats[0] = (float)mean;
            stats[1] = (float)(1.0 / sqrt(var + (double)eps));
            const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            run_mean[0] = (float)((1.0 - momentum) * run_mean[0] + momentum * mean);
            run_var[0] = (float)((1.0 - momentum) * run_var[0] + momentum * unb);
        } else {
            stats[0] = run_mean[0];
            stats[1] = 1.f / sqrtf(run_var[0] + eps);
        }
    }
}

// ------------------------------------------------------------------------------------------------ F2
template <int E, bool SMALL>
__global__ __launch_bounds__(FT) void cbam_f2_kernel(const float* __restrict__ x, const float* __restrict__ cs,
                                                     const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                     const float* __restrict__ stats,
                                                     float* __restrict__ xhat, float* __restrict__ ss, float* __restrict__ y,
                                                     int C, int HW, int G) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_sc = sm;
    float* s_ss = s_sc + al4(C);
    const int Q = HW / E;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float mean = stats[0], inv = stats[1], ga = bn_w[0], be = bn_b[0];
    for (int p = tid; p < HW; p += FT) {
        const size_t o = (size_t)n * HW + p;
        const float xh = (xhat[o] - mean) * inv;           // xhat holds the raw conv output on entry
        xhat[o] = xh;
        const float s = 1.f / (1.f + expf(-(xh * ga + be)));
        ss[o] = s;
        s_ss[p] = s;
    }
    for (int c = tid; c < C; c += FT) s_sc[c] = cs[(size_t)n * C + c];
    __syncthreads();
    const float* xb = x + (size_t)n * C * HW;
    float* yb = y + (size_t)n * C * HW;
    const int per = 64 / G, sub = lane % G, pi = lane / G;
    if (SMALL) {
        const int step = (FT / 64) * per;
        for (int c0 = wave * per; c0 < C; c0 += PB * step) {
            float v[PB][E];
            bool ok[PB];
#pragma unroll
            for (int k = 0; k < PB; ++k) {
                const int c = c0 + k * step + pi;
                ok[k] = c < C && sub < Q;
                if (ok[k]) Unit<E>::ld(xb + (size_t)c * HW + (size_t)sub * E, v[k]);
            }
#pragma unroll
            for (int k = 0; k < PB; ++k)
                if (ok[k]) {
                    const int c = c0 + k * step + pi;
                    const float sc = s_sc[c];
#pragma unroll
                    for (int e = 0; e < E; ++e) v[k][e] *= sc * s_ss[sub * E + e];
                    Unit<E>::st(yb + (size_t)c * HW + (size_t)sub * E, v[k]);
                }
        }
        return;
    }
    for (int c0 = wave * per; c0 < C; c0 += (FT / 64) * per) {
        const int c = c0 + pi;
        if (c >= C) continue;
        const float sc = s_sc[c];
        const float* pl = xb + (size_t)c * HW;
        float* yl = yb + (size_t)c * HW;
        for (int u0 = sub; u0 < Q; u0 += 4 * G) {
            float v[4][E];
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int u = u0 + k * G;
                ok[k] = u < Q;
                if (ok[k]) Unit<E>::ld(pl + (size_t)u * E, v[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ok[k]) {
                    const int u = u0 + k * G;
#pragma unroll
                    for (int e = 0; e < E; ++e) v[k][e] *= sc * s_ss[u * E + e];
                    Unit<E>::st(yl + (size_t)u * E, v[k]);
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------ B1
template <int E>
__global__ __launch_bounds__(FT) void cbam_b1_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ cs, const float* __restrict__ ss,
                                                     const float* __restrict__ xhat, float* __restrict__ dpre,
                                                     double* __restrict__ part, int C, int HW, int Qp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ double red[2 * FT / 64];
    float* s_sc = sm;
    float* p_sum = s_sc + al4(C);
    const int Q = HW / E;
    const int nk = Qp < 64 ? FT / 64 : FT / Qp;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += FT) s_sc[c] = cs[(size_t)n * C + c];
    __syncthreads();
    const float* xb = x + (size_t)n * C * HW;
    const float* gb = dy + (size_t)n * C * HW;
    const int q = tid % Qp, k = tid / Qp, KS = FT / Qp;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    if (q < Q) {
        constexpr int BB = 4;        // channels in flight per thread, two loads each (6 / 8: 183 -> 181 / 179 us at 64 x 28 x 28, 98 -> 100 / 98 at
                                     // 128 x 14 x 14: this sweep runs at 4.5 TB/s, it is not waiting for latency)
        for (int c0 = k; c0 < C; c0 += BB * KS) {
            float a[BB][E], b[BB][E];
            bool ok[BB];
#pragma unroll
            for (int j = 0; j < BB; ++j) {
                const int c = c0 + j * KS;
                ok[j] = c < C;
                if (ok[j]) {
                    Unit<E>::ld(gb + ((size_t)c * Q + q) * E, a[j]);
                    Unit<E>::ld(xb + ((size_t)c * Q + q) * E, b[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < BB; ++j)
                if (ok[j]) {
                    const float sc = s_sc[c0 + j * KS];
#pragma unroll
                    for (int e = 0; e < E; ++e) acc[e] += a[j][e] * (b[j][e] * sc);
                }
        }
    }
    if (Qp < 64) {
        for (int o = Qp; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
        if (lane < Qp)
#pragma unroll
            for (int e = 0; e < E; ++e) p_sum[(wave * Qp + q) * E + e] = acc[e];
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) p_sum[(k * Qp + q) * E + e] = acc[e];
    }
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    if (tid < Q) {
        float ds[E], s[E], xh[E], d[E];
#pragma unroll
        for (int e = 0; e < E; ++e) ds[e] = 0.f;
        for (int j = 0; j < nk; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) ds[e] += p_sum[(j * Qp + tid) * E + e];
        const size_t o = (size_t)n * HW + (size_t)tid * E;
        Unit<E>::ld(ss + o, s);
        Unit<E>::ld(xhat + o, xh);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            d[e] = ds[e] * s[e] * (1.f - s[e]);
            s1 += d[e];
            s2 += (double)d[e] * xh[e];
        }
        Unit<E>::st(dpre + o, d);
    }
    block_sum2_d8(s1, s2, red);
    if (tid == 0) { part[2 * (size_t)n] = s1; part[2 * (size_t)n + 1] = s2; }
}

// out2[0] = sum of part[2i+1] (d gamma), out2[1] = sum of part[2i] (d beta)
__global__ __launch_bounds__(256) void cbam_sum_pairs_kernel(const double* __restrict__ part, int nparts, float* __restrict__ out_gamma_beta,
                                                             float* __restrict__ dbn_w, float* __restrict__ dbn_b) {
    __shared__ double red[4];
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) { s1 += part[2 * i]; s2 += part[2 * i + 1]; }
    for (int pass = 0; pass < 2; ++pass) {
        double v = pass ? s2 : s1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        v = (red[0] + red[1]) + (red[2] + red[3]);
        if (pass) s2 = v; else s1 = v;
    }
    if (threadIdx.x == 0) {
        out_gamma_beta[0] = (float)s2; out_gamma_beta[1] = (float)s1;
        dbn_w[0] = (float)s2; dbn_b[0] = (float)s1;
    }
}

// ---- B2's per-pixel LDS maps
struct B2Lds {
    float* dc;        // [HW] gradient wrt the conv output
    float* ss;        // [HW] spatial scale
    float* dmx;       // [HW] gradient wrt the max map
    float* dmn;       // [HW] gradient wrt the mean map, already / C
    int* cam;         // [HW] argmax_c
    int* pb;          // [HW] a pixel's 5 x 5 window origin in the padded maps: h * (W + 4) + w
    float* dcp;       // [PP] dc, zero-padded
    float* compp;     // [2][PP] the two compressed maps, zero-padded
    float* rest;      // what the kernel places behind the maps
};
__device__ __forceinline__ B2Lds b2_lds_map(float* sm, int H, int W) {
    const int HWa = al4(H * W), PPa = al4(pad_len(H, W));
    B2Lds L;
    L.dc = sm; L.ss = L.dc + HWa; L.dmx = L.ss + HWa; L.dmn = L.dmx + HWa;
    L.cam = reinterpret_cast<int*>(L.dmn + HWa); L.pb = L.cam + HWa;
    L.dcp = reinterpret_cast<float*>(L.pb + HWa); L.compp = L.dcp + PPa; L.rest = L.compp + 2 * PPa;
    return L;
}
__host__ __device__ inline int b2_map_floats(int H, int W) { return 6 * al4(H * W) + 3 * al4(pad_len(H, W)); }

// ---- the spatial gate's backward for one frame up to the gradients of the two compressed maps (phases a-c of B2 / B2L):
//   a. BatchNorm2d(1) backward per pixel (reference cbam.py:78), the frame's per-pixel maps into LDS: one sweep over the PADDED index
//      space fills the padded and the plain copies, so no location has two writers
//   b. this frame's share of the conv weight gradient: 50 taps, one wave per tap, no bounds tests (padded maps), no divisions (pb)
//   c. conv backward to the two compressed maps: 25 unconditional taps per pixel, weights as scalar operands
// Ends WITHOUT a barrier after c.
__device__ __forceinline__ void spatial_bwd(const B2Lds& L, const float* __restrict__ convw, const float* __restrict__ bn_w,
                                            const float* __restrict__ stats, const float* __restrict__ dgb, const float* __restrict__ comp,
                                            const int32_t* __restrict__ cargmax, const float* __restrict__ xhat, const float* __restrict__ ss,
                                            const float* __restrict__ dpre, float* __restrict__ dwpart, int n, int H, int W, int C,
                                            float inv_total, int training) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HW = H * W, Wp = W + 4, PP = pad_len(H, W), PPa = al4(PP);
    {
        const float gamma = bn_w[0], invstd = stats[1];
        const float m1 = gamma * dgb[1] * inv_total;       // mean(dxhat)
        const float m2 = gamma * dgb[0] * inv_total;       // mean(dxhat * xhat)
        for (int i = tid; i < PP; i += FT) {
            const int yy = i / Wp, y = yy - 2, xq = i - yy * Wp - 2;
            const bool in = y >= 0 && y < H && xq >= 0 && xq < W;
            float dc = 0.f, c0 = 0.f, c1 = 0.f;
            if (in) {
                const int pp = y * W + xq;
                const size_t o = (size_t)n * HW + pp;
                const float dxh = dpre[o] * gamma;
                dc = training ? invstd * (dxh - m1 - xhat[o] * m2) : dxh * invstd;
                c0 = comp[((size_t)n * 2 + 0) * HW + pp];
                c1 = comp[((size_t)n * 2 + 1) * HW + pp];
                L.dc[pp] = dc;
                L.ss[pp] = ss[o];
                L.cam[pp] = cargmax[o];
                L.pb[pp] = y * Wp + xq;
            }
            L.dcp[i] = dc;
            L.compp[i] = c0;
            L.compp[PPa + i] = c1;
        }
    }
    __syncthreads();
    M3T_CB_STAMP(1);
    {
        // a wave's taps (wave, wave + 8, ...: 7 at most) share one walk over the pixels: dc and the window origin are read once per
        // pixel, the seven map reads are independent (a tap at a time, each pixel was a chain of two dependent LDS reads: 10 of B2's
        // 100 us per 64 x 28 x 28 frame)
        constexpr int TW = (50 + FT / 64 - 1) / (FT / 64);
        int off[TW];
        float sw[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const int tap = wave + (FT / 64) * t, tc = tap < 50 ? tap : 0;
            off[t] = (tc / 25) * PPa + ((tc % 25) / 5) * Wp + tc % 5;
            sw[t] = 0.f;
        }
        for (int pp = lane; pp < HW; pp += 64) {
            const float d = L.dc[pp];
            const float* cp = L.compp + L.pb[pp];
#pragma unroll
            for (int t = 0; t < TW; ++t) sw[t] += d * cp[off[t]];
        }
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            const int tap = wave + (FT / 64) * t;
            const float sv = group_sum(sw[t], 64);
            if (lane == 0 && tap < 50) dwpart[(size_t)n * 50 + tap] = sv;
        }
    }
    for (int pp = tid; pp < HW; pp += FT) {
        const float* d0 = L.dcp + L.pb[pp] + 4 * Wp + 4;      // dc at (h + 2 - i, w + 2 - j) = d0[-(i * Wp + j)]
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float d = d0[-(i * Wp + j)];
                a += convw[(0 * 5 + i) * 5 + j] * d;
                b += convw[(1 * 5 + i) * 5 + j] * d;
            }
        L.dmx[pp] = a;
        L.dmn[pp] = b / (float)C;
    }
}

// ---- the gate's shared MLP backward for one frame (phase e of B2 / B2L, reference cbam.py:51-58).  In: s_datt [C] (gradient at the
// sigmoid's input).  Out: s_davg [C] (already / HW), s_dmaxc [C]; slabs g_dh, g_r for the parameter gradients.  Ends WITHOUT a barrier.
__device__ __forceinline__ void gate_mlp_bwd(const float* s_datt, float* s_part, float* s_dh, float* s_davg, float* s_dmaxc,
                                             const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ hidden,
                                             float* __restrict__ g_dh, float* __restrict__ g_r, int n, int C, int Cr, int HW) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- e. shared MLP backward (as cbam.hip): g[r] = sum_c datt[c] W2[c][r] -> ReLU masks -> davg, dmax per channel
    // W2 as one flat float4 stream (as F1's second layer): a thread keeps its four columns r over every pass (FT is a multiple of
    // L = Cr / 4), all loads of the frame in flight at once; then the lanes of a wave that share columns (stride L) are summed
    if ((Cr & 3) == 0 && (Cr & (Cr - 1)) == 0 && ((uintptr_t)w2 & 15) == 0) {
        const int L = Cr >> 2, lg = __ffs(L) - 1, n4 = C * L;
        const float4* w4 = reinterpret_cast<const float4*>(w2);
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i0 = tid; i0 < n4; i0 += 4 * FT) {
            float4 w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * FT;
                w[k] = i < n4 ? w4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * FT;
                if (i < n4) {
                    const float d = s_datt[i >> lg];
                    g[0] += d * w[k].x; g[1] += d * w[k].y; g[2] += d * w[k].z; g[3] += d * w[k].w;
                }
            }
        }
        for (int o = L; o < 64; o <<= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] += __shfl_xor(g[e], o, 64);
        }
        if (lane < L) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s_part[wave * Cr + 4 * lane + e] = g[e];
        }
    } else if (lane < Cr) {
        float g = 0.f;
        for (int c = wave; c < C; c += FT / 64) g += s_datt[c] * w2[(size_t)c * Cr + lane];
        s_part[wave * Cr + lane] = g;
    }
    __syncthreads();
    M3T_CB_STAMP(4);
    for (int j = tid; j < 2 * Cr; j += FT) {
        const int which = j / Cr, r = j % Cr;
        float g = 0.f;
#pragma unroll
        for (int wv = 0; wv < FT / 64; ++wv) g += s_part[wv * Cr + r];
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
    M3T_CB_STAMP(5);
    for (int c = tid; c < C; c += FT) {
        float da = 0.f, dm = 0.f;
        for (int r0 = 0; r0 < Cr; r0 += 16) {              // sixteen rows of W1 in flight (coalesced over c)
            float w[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) w[k] = r0 + k < Cr ? w1[(size_t)(r0 + k) * C + c] : 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (r0 + k < Cr) {
                    da += s_dh[r0 + k] * w[k];
                    dm += s_dh[Cr + r0 + k] * w[k];
                }
        }
        s_davg[c] = da / (float)HW;
        s_dmaxc[c] = dm;
    }
}

// ------------------------------------------------------------------------------------------------ B2
#ifndef M3T_CBAM_B2_UB
#define M3T_CBAM_B2_UB 2
#endif
constexpr int UB = M3T_CBAM_B2_UB;      // units in flight per lane in B2's plane sweeps of large maps: 2 -> 77 VGPRs, three workgroups
                                        // per CU (4 -> 118, two): 0.996 vs 1.037 ms for the whole gate at 2048 x 64 x 28 x 28

template <int E, bool SMALL>
__global__ __launch_bounds__(FT, 6) void cbam_b2_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ w1, const float* __restrict__ w2,
                                                     const float* __restrict__ convw, const float* __restrict__ bn_w,
                                                     const float* __restrict__ stats, const float* __restrict__ dgb,
                                                     const float* __restrict__ cs, const int32_t* __restrict__ argmax_p,
                                                     const float* __restrict__ hidden, const float* __restrict__ comp,
                                                     const int32_t* __restrict__ cargmax, const float* __restrict__ xhat,
                                                     const float* __restrict__ ss, const float* __restrict__ dpre,
                                                     float* __restrict__ dx, float* __restrict__ g_datt, float* __restrict__ g_dh,
                                                     float* __restrict__ g_r, float* __restrict__ dwpart, int C, int Cr, int H,
                                                     int W, int G, float inv_total, int training) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HW = H * W, Q = HW / E;
    const B2Lds L = b2_lds_map(sm, H, W);
    const float* s_ss = L.ss;
    const float* s_dmx = L.dmx;
    const float* s_dmn = L.dmn;
    const int* s_cam = L.cam;
    float* s_sc = L.rest;                      // [C]
    float* s_datt = s_sc + al4(C);
    float* s_davg = s_datt + al4(C);
    float* s_dmaxc = s_davg + al4(C);
    float* s_dh = s_dmaxc + al4(C);            // [2 Cr]
    float* s_part = s_dh + al4(2 * Cr);        // [8][Cr]
    int* s_amp = reinterpret_cast<int*>(s_part + 8 * Cr);      // [C] argmax_p
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xb = x + (size_t)n * C * HW;
    const float* gb = dy + (size_t)n * C * HW;
    M3T_CB_STAMP(0);
    spatial_bwd(L, convw, bn_w, stats, dgb, comp, cargmax, xhat, ss, dpre, dwpart, n, H, W, C, inv_total, training);
    for (int c = tid; c < C; c += FT) { s_sc[c] = cs[(size_t)n * C + c]; s_amp[c] = argmax_p[(size_t)n * C + c]; }
    __syncthreads();
    M3T_CB_STAMP(2);
    // ---- d. dcs[c] = sum_p dy1 * x with dy1 = dy * ss + dmean / C + [c == argmax_c] dmax (the spatial gate's input gradient)
    const int per = 64 / G, sub = lane % G, pi = lane / G;
    if (SMALL) {
        const int step = (FT / 64) * per;
        for (int c0 = wave * per; c0 < C; c0 += PB * step) {
            float a[PB][E], b[PB][E];
            bool ok[PB];
#pragma unroll
            for (int k = 0; k < PB; ++k) {
                const int c = c0 + k * step + pi;
                ok[k] = c < C && sub < Q;
                if (ok[k]) {
                    Unit<E>::ld(gb + (size_t)c * HW + (size_t)sub * E, a[k]);
                    Unit<E>::ld(xb + (size_t)c * HW + (size_t)sub * E, b[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < PB; ++k) {
                const int c = c0 + k * step + pi;
                float ds = 0.f;
                if (ok[k]) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const int p = sub * E + e;
                        const float dy1 = a[k][e] * s_ss[p] + s_dmn[p] + (c == s_cam[p] ? s_dmx[p] : 0.f);
                        ds += dy1 * b[k][e];
                    }
                }
                ds = group_sum(ds, G);
                if (sub == 0 && c < C) {
                    const float sc = s_sc[c];
                    const float da = ds * sc * (1.f - sc);
                    s_datt[c] = da;
                    g_datt[(size_t)n * C + c] = da;
                }
            }
        }
    } else
    for (int c0 = wave * per; c0 < C; c0 += (FT / 64) * per) {
        const int c = c0 + pi;
        const bool okc = c < C;
        const float* pl = xb + (size_t)(okc ? c : 0) * HW;
        const float* gl = gb + (size_t)(okc ? c : 0) * HW;
        float ds = 0.f;
        for (int u0 = sub; u0 < Q; u0 += UB * G) {
            float a[UB][E], b[UB][E];
            bool ok[UB];
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int u = u0 + k * G;
                ok[k] = okc && u < Q;
                if (ok[k]) { Unit<E>::ld(gl + (size_t)u * E, a[k]); Unit<E>::ld(pl + (size_t)u * E, b[k]); }
            }
#pragma unroll
            for (int k = 0; k < UB; ++k)
                if (ok[k]) {
                    const int u = u0 + k * G;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const int p = u * E + e;
                        const float dy1 = a[k][e] * s_ss[p] + s_dmn[p] + (c == s_cam[p] ? s_dmx[p] : 0.f);
                        ds += dy1 * b[k][e];
                    }
                }
        }
        ds = group_sum(ds, G);
        if (sub == 0 && okc) {
            const float sc = s_sc[c];
            const float da = ds * sc * (1.f - sc);
            s_datt[c] = da;
            g_datt[(size_t)n * C + c] = da;
        }
    }
    __syncthreads();
    M3T_CB_STAMP(3);
    gate_mlp_bwd(s_datt, s_part, s_dh, s_davg, s_dmaxc, w1, w2, hidden, g_dh, g_r, n, C, Cr, HW);
    __syncthreads();
    M3T_CB_STAMP(6);
    // ---- f. dx = dy1 * cs + davg / HW + [p == argmax_p] dmaxc; dy is re-read while the frame is still cache-hot
    float* db = dx + (size_t)n * C * HW;
    if (SMALL) {
        const int step = (FT / 64) * per;
        for (int c0 = wave * per; c0 < C; c0 += PB * step) {
            float a[PB][E];
            bool ok[PB];
#pragma unroll
            for (int k = 0; k < PB; ++k) {
                const int c = c0 + k * step + pi;
                ok[k] = c < C && sub < Q;
                if (ok[k]) Unit<E>::ld(gb + (size_t)c * HW + (size_t)sub * E, a[k]);
            }
#pragma unroll
            for (int k = 0; k < PB; ++k)
                if (ok[k]) {
                    const int c = c0 + k * step + pi;
                    const float sc = s_sc[c], da = s_davg[c], dm = s_dmaxc[c];
                    const int amp = s_amp[c];
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const int p = sub * E + e;
                        const float dy1 = a[k][e] * s_ss[p] + s_dmn[p] + (c == s_cam[p] ? s_dmx[p] : 0.f);
                        a[k][e] = dy1 * sc + da + (p == amp ? dm : 0.f);
                    }
                    Unit<E>::st(db + (size_t)c * HW + (size_t)sub * E, a[k]);
                }
        }
        M3T_CB_STAMP(7);
        return;
    }
    for (int c0 = wave * per; c0 < C; c0 += (FT / 64) * per) {
        const int c = c0 + pi;
        if (c >= C) continue;
        const float sc = s_sc[c], da = s_davg[c], dm = s_dmaxc[c];
        const int amp = s_amp[c];
        const float* gl = gb + (size_t)c * HW;
        float* dl = db + (size_t)c * HW;
        for (int u0 = sub; u0 < Q; u0 += UB * G) {
            float a[UB][E];
            bool ok[UB];
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int u = u0 + k * G;
                ok[k] = u < Q;
                if (ok[k]) Unit<E>::ld(gl + (size_t)u * E, a[k]);
            }
#pragma unroll
            for (int k = 0; k < UB; ++k)
                if (ok[k]) {
                    const int u = u0 + k * G;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const int p = u * E + e;
                        const float dy1 = a[k][e] * s_ss[p] + s_dmn[p] + (c == s_cam[p] ? s_dmx[p] : 0.f);
                        a[k][e] = dy1 * sc + da + (p == amp ? dm : 0.f);
                    }
                    Unit<E>::st(dl + (size_t)u * E, a[k]);
                }
        }
    }
    M3T_CB_STAMP(7);
}

// ---- parameter gradients of the shared MLP (and the 5 x 5 conv: each frame left its 50-tap share in dwpart) from the per-frame slabs, two launches instead of 3 GEMMs + 3 split-K reduces + 8
// column-sum kernels (the slabs are small: N x (2 C + 4 Cr) floats; at the 4 x 4 stage those 14 launches were a quarter of the gate)
//   dW2[c,r] = sum_n datt[n,c] R[n,r]        db2[c] = 2 sum_n datt[n,c]          (reference cbam.py:44-49: mlp.3)
//   dW1[r,c] = sum_n dha[n,r] avg[n,c] + dhm[n,r] max[n,c]      db1[r] = sum_n dha[n,r] + dhm[n,r]          (mlp.1)
// stage A: workgroup (channel block of 64, slice of 32 frames) -> partials; stage B: sums the slices in slice order.  Deterministic.
constexpr int PG_CHUNK = 32;          // frames per workgroup of stage A: N / 32 slices (64 at N = 2048) x C / 64 channel blocks
// PER contiguous floats out of LDS as the widest reads their alignment allows (the caller's offset is a multiple of PER)
template <int PER> __device__ __forceinline__ void lds_row(const float* p, float (&o)[PER]) {
    if constexpr (PER % 4 == 0) {
#pragma unroll
        for (int i = 0; i < PER / 4; ++i) {
            const float4 t = reinterpret_cast<const float4*>(p)[i];
            o[4 * i] = t.x; o[4 * i + 1] = t.y; o[4 * i + 2] = t.z; o[4 * i + 3] = t.w;
        }
    } else if constexpr (PER == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        o[0] = t.x; o[1] = t.y;
    } else {
#pragma unroll
        for (int i = 0; i < PER; ++i) o[i] = p[i];
    }
}

// PER = Cr / 4 hidden units per wave (Cr = 4, 8, 16, 32: ResNet-18's four stages), contiguous: r = rg * PER + i, so a frame's R / dha / dhm
// values come out of LDS as float4 broadcasts (round 3: r = rg + 4 i, 24 scalar reads per frame and thread at Cr = 32 -- one per
// multiply-add).  PER = 0: any Cr <= 64, the scalar mapping.
template <int PER>
__global__ __launch_bounds__(256) void cbam_pgrad_partial_kernel(const float* __restrict__ g_datt, const float* __restrict__ g_r,
                                                                 const float* __restrict__ g_dh, const float* __restrict__ pooled,
                                                                 const float* __restrict__ dwpart, float* __restrict__ part, int N, int C,
                                                                 int Cr) {
    __shared__ __attribute__((aligned(16))) float s_small[PG_CHUNK * 3 * 64];          // per frame of the slice: R[Cr] | dha[Cr] | dhm[Cr]
    __shared__ double s_dw[4 * 50];
    static_assert(PG_CHUNK == 32, "the conv-tap share below is laid out as 4 groups of 8 frames");
    constexpr int NA = PER > 0 ? PER : 16;
    const int cb = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
    const int cl = tid & 63, rg = tid >> 6, c = cb * 64 + cl;
    const bool okc = c < C;
    const int n0 = sl * PG_CHUNK, cnt = min(PG_CHUNK, N - n0);
    // the slice's share of the conv weight gradient (each frame left its 50 taps in dwpart): 4 groups of 8 frames x 50 taps, one
    // batch of loads, fp64 sums; the channel-block-0 workgroup of the slice does it while its slab rows are in flight
    if (cb == 0 && tid < 200) {
        const int tap = tid % 50, fg = tid / 50;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int f = fg * 8 + k; v[k] = f < cnt ? dwpart[(size_t)(n0 + f) * 50 + tap] : 0.f; }
        double sd = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) sd += (double)v[k];
        s_dw[fg * 50 + tap] = sd;
    }
    // g_r rows are [Cr], g_dh rows [2 Cr] (dha | dhm): both contiguous over the slice's frames
    for (int i = tid; i < cnt * Cr; i += 256) {
        const int f = i / Cr, j = i - f * Cr;
        s_small[f * 3 * Cr + j] = g_r[(size_t)n0 * Cr + i];
    }
    for (int i = tid; i < cnt * 2 * Cr; i += 256) {
        const int f = i / (2 * Cr), j = i - f * 2 * Cr;
        s_small[f * 3 * Cr + Cr + j] = g_dh[(size_t)n0 * 2 * Cr + i];
    }
    float a2[NA], a1[NA], sb2 = 0.f, sb1 = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) { a2[i] = 0.f; a1[i] = 0.f; }
    const int nr = (Cr + 3) / 4;
    __syncthreads();
    for (int f0 = 0; f0 < cnt; f0 += 8) {                  // eight frames' loads in flight
        float da[8], av[8], mx[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int n = n0 + f0 + k;
            const bool ok = okc && f0 + k < cnt;
            da[k] = ok ? g_datt[(size_t)n * C + c] : 0.f;
            av[k] = ok ? pooled[((size_t)n * 2 + 0) * C + c] : 0.f;
            mx[k] = ok ? pooled[((size_t)n * 2 + 1) * C + c] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (f0 + k >= cnt) break;
            const float* sm3 = s_small + (f0 + k) * 3 * Cr;
            sb2 += da[k];
            if constexpr (PER > 0) {
                float rr[PER], ha[PER], hm[PER];
                lds_row<PER>(sm3 + rg * PER, rr);
                lds_row<PER>(sm3 + Cr + rg * PER, ha);
                lds_row<PER>(sm3 + 2 * Cr + rg * PER, hm);
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    a2[i] += da[k] * rr[i];
                    a1[i] += ha[i] * av[k] + hm[i] * mx[k];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = rg + 4 * i;
                    if (i < nr && r < Cr) {
                        a2[i] += da[k] * sm3[r];
                        a1[i] += sm3[Cr + r] * av[k] + sm3[2 * Cr + r] * mx[k];
                    }
                }
            }
            if (cb == 0 && tid < Cr) sb1 += sm3[Cr + tid] + sm3[2 * Cr + tid];
        }
    }
    // partial layout per slice: dW2 [C][Cr] | dW1 [Cr][C] | db2 [C] | db1 [Cr] | dconv [50]
    float* ps = part + (size_t)sl * (2 * (size_t)C * Cr + C + Cr + 50);
    if (cb == 0 && tid < 50) ps[2 * (size_t)C * Cr + C + Cr + tid] = (float)((s_dw[tid] + s_dw[50 + tid]) + (s_dw[100 + tid] + s_dw[150 + tid]));
    if (okc) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = PER > 0 ? rg * PER + i : rg + 4 * i;
            if (PER > 0 || (i < nr && r < Cr)) {
                ps[(size_t)c * Cr + r] = a2[i];
                ps[(size_t)C * Cr + (size_t)r * C + c] = a1[i];
            }
        }
        if (rg == 0) ps[2 * (size_t)C * Cr + c] = 2.f * sb2;
    }
    if (cb == 0 && tid < Cr) ps[2 * (size_t)C * Cr + C + tid] = sb1;
}

__global__ __launch_bounds__(256) void cbam_pgrad_final_kernel(const float* __restrict__ part, int slices, int C, int Cr,
                                                               float* __restrict__ dw2, float* __restrict__ dw1,
                                                               float* __restrict__ db2, float* __restrict__ db1,
                                                               float* __restrict__ dconv) {
    const size_t per = 2 * (size_t)C * Cr + C + Cr + 50;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) {
        float s = 0.f;
        for (int k0 = 0; k0 < slices; k0 += 8) {               // eight slices in flight, summed in slice order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = k0 + k < slices ? part[(size_t)(k0 + k) * per + i] : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        const size_t cc = (size_t)C * Cr;
        if (i < cc) dw2[i] = s;
        else if (i < 2 * cc) dw1[i - cc] = s;
        else if (i < 2 * cc + C) db2[i - 2 * cc] = s;
        else if (i < 2 * cc + C + Cr) db1[i - 2 * cc - C] = s;
        else dconv[i - 2 * cc - C - Cr] = s;
    }
}

// ================================================================================ frame-resident kernels for small frames
// A frame of the late ResNet stages is small (256 x 7 x 7: 50 KB, 512 x 4 x 4: 32 KB) and F1 / B2 above spend their time there on
// what the frame's SHAPE costs them, not on its bytes (tools/cbam_phase_probe.hip, round 4): a plane is a fraction of a wave, so a
// wave-load carries 196 B, every plane needs its own cross-lane reduction (30 of F1's 42 us per 256 x 7 x 7 frame, 17 + 11 of B2's
// 37), and the sweeps are chains of 4-plane batches.  F1L / B2L read the frame as ONE flat float4 stream -- every load of the frame
// in flight before anything else happens, full 1 KB wave-loads whatever H x W is -- and park it in LDS as [C][S] with S odd, where
//   * a per-plane reduction is a THREAD walking its plane (stride S: conflict-free), no cross-lane step at all,
//   * a per-pixel reduction over channels is lane = pixel walking down the planes (the second read of x comes from LDS, not L2),
// and B2L keeps dy in registers from the first instruction to the dx store (read once, never re-read).  Arithmetic per element is
// F1 / B2's; only the order of the sums differs (fixed, deterministic).
// Eligible: C H W a multiple of 4 and at most 16 384 elements (8 float4 per thread), H W <= 64, 16-B aligned tensors.
// (Tried on F1L, round 4: persistent workgroups -- grid = what the chip holds, a frame every gridDim.x -- with the next frame's eight
// float4 loads issued behind the MLP so that they land during compress / conv / sums.  It needs the weight pointers and the thread
// index laundered once per trip (asm volatile "+s" / "+v"), or hipcc hoists every invariant out of the frame loop: 128 VGPRs + 73
// spilled.  Without spills: 54.5 vs 51 us at 256 x 7 x 7, 63 vs 53 us at 512 x 4 x 4.  With two or three workgroups per CU the
// frame's load wait was already covered by the neighbours' arithmetic; what is left per frame is instruction issue, not memory.)

// element e of a frame -> (plane c, pixel p).  Exact: (e + 0.5) / HW is at least 0.5 / HW >= 1/128 away from an integer, the float
// product is off by < 2^-22 * 2^14 / 4; the fix-up is belt and braces
__device__ __forceinline__ void plane_pixel(int e, int HW, float inv_hw, int& c, int& p) {
    c = (int)(((float)e + 0.5f) * inv_hw);
    p = e - c * HW;
    if (p < 0) { --c; p += HW; } else if (p >= HW) { ++c; p -= HW; }
}

__host__ __device__ inline int odd_stride(int HW) { return HW | 1; }

template <int NV>
__global__ __launch_bounds__(FT, NV <= 4 ? 6 : 4) void cbam_f1l_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2,
                                                     const float* __restrict__ b2, const float* __restrict__ convw,
                                                     float* __restrict__ cs, int32_t* __restrict__ argmax_p,
                                                     float* __restrict__ pooled, float* __restrict__ hidden,
                                                     float* __restrict__ comp, int32_t* __restrict__ cargmax,
                                                     float* __restrict__ conv_out, double* __restrict__ part, int C, int Cr,
                                                     int H, int W, int Qp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ double red[2 * FT / 64];
    const int HW = H * W, PP = pad_len(H, W), S = odd_stride(HW);
    float* s_f = sm;                          // [C][S] the frame
    float* s_avg = s_f + al4(C * S);
    float* s_max = s_avg + al4(C);
    float* s_sc = s_max + al4(C);
    float* s_h = s_sc + al4(C);               // [2 Cr]
    float* s_pad = s_h + al4(2 * Cr);         // [2][PP]
    float* p_mx = s_pad + al4(2 * PP);        // [8][Qp]
    float* p_sum = p_mx + (FT / 64) * Qp;
    int* p_am = reinterpret_cast<int*>(p_sum + (FT / 64) * Qp);
    int* s_amp = p_am + (FT / 64) * Qp;       // [C]
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n4 = (C * HW) >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)n * C * HW);
    M3T_CB_STAMP(0);
    // ---- a. the whole frame in flight, then into LDS
    float4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = tid + k * FT;
        v[k] = i < n4 ? x4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = tid; i < 2 * PP; i += FT) s_pad[i] = 0.f;
    const float inv_hw = 1.f / (float)HW;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = tid + k * FT;
        if (i < n4) {
            if (S == HW) {
                reinterpret_cast<float4*>(s_f)[i] = v[k];
            } else {
                int c, p;
                plane_pixel(4 * i, HW, inv_hw, c, p);
                const float e4[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s_f[c * S + p] = e4[j];
                    if (++p == HW) { p = 0; ++c; }
                }
            }
        }
    }
    __syncthreads();
    // ---- channel squeeze: a thread per plane
    for (int c = tid; c < C; c += FT) {
        const float* pl = s_f + c * S;
        float sum = 0.f, mx = -INFINITY;
        int am = 0x7fffffff;
#pragma unroll 4
        for (int p = 0; p < HW; ++p) {
            const float t = pl[p];
            sum += t;
            if (t > mx) { mx = t; am = p; }
        }
        s_avg[c] = sum / (float)HW; s_max[c] = mx; s_amp[c] = am;
    }
    __syncthreads();
    M3T_CB_STAMP(1);
    for (int c = tid; c < C; c += FT) {
        pooled[((size_t)n * 2 + 0) * C + c] = s_avg[c];
        pooled[((size_t)n * 2 + 1) * C + c] = s_max[c];
        argmax_p[(size_t)n * C + c] = s_amp[c];
    }
    // (Tried: W1 fetched before the squeeze and W2 before layer 1's arithmetic, 8 float4 each held in registers.  512 x 4 x 4: 80 VGPRs
    // + 19 spilled, F1L 53 -> 73 us; 256 x 7 x 7: 52 vs 53 us.  The MLP's cost at the 4 x 4 stage is streaming 128 KB of weights per
    // 32 KB frame through the CU's L1, not the latency of the fetch.)
    gate_mlp_fwd(s_avg, s_max, s_h, s_sc, w1, b1, w2, b2, hidden, cs, n, C, Cr);
    __syncthreads();
    M3T_CB_STAMP(3);
    // ---- c. compress x * cs over channels out of LDS: lane = pixel q, FT / Qp channel slices
    {
        const int q = tid & (Qp - 1), kq = tid / Qp, KS = FT / Qp;
        float mx = -INFINITY, sum = 0.f;
        int am = 0x7fffffff;
        if (q < HW) {
#pragma unroll 4
            for (int c = kq; c < C; c += KS) {
                const float t = s_f[c * S + q] * s_sc[c];
                sum += t;
                if (t > mx) { mx = t; am = c; }
            }
        }
        for (int o = Qp; o < 64; o <<= 1) {       // the slices of a pixel inside the wave sit Qp lanes apart
            if (o == 16) {
                const auto rs = __builtin_amdgcn_permlane16_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
                const auto rm = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                const auto ri = __builtin_amdgcn_permlane16_swap((unsigned)am, (unsigned)am, false, false);
                sum = __uint_as_float(rs[0]) + __uint_as_float(rs[1]);
                mx = __uint_as_float(rm[0]); am = (int)ri[0];
                amax_take(mx, am, __uint_as_float(rm[1]), (int)ri[1]);
            } else if (o == 32) {
                const auto rs = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
                const auto rm = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                const auto ri = __builtin_amdgcn_permlane32_swap((unsigned)am, (unsigned)am, false, false);
                sum = __uint_as_float(rs[0]) + __uint_as_float(rs[1]);
                mx = __uint_as_float(rm[0]); am = (int)ri[0];
                amax_take(mx, am, __uint_as_float(rm[1]), (int)ri[1]);
            } else {
                sum += __shfl_xor(sum, o, 64);
                const float ov = __shfl_xor(mx, o, 64);
                const int oi = __shfl_xor(am, o, 64);
                amax_take(mx, am, ov, oi);
            }
        }
        if (lane < Qp) { p_mx[wave * Qp + q] = mx; p_sum[wave * Qp + q] = sum; p_am[wave * Qp + q] = am; }
        __syncthreads();
        M3T_CB_STAMP(4);
        if (tid < HW) {
            float fm = -INFINITY, fs = 0.f;
            int fa = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < FT / 64; ++j) {
                fs += p_sum[j * Qp + tid];
                amax_take(fm, fa, p_mx[j * Qp + tid], p_am[j * Qp + tid]);
            }
            const float mean = fs / (float)C;
            const int ph = tid / W, ip = (ph + 2) * (W + 4) + (tid - ph * W) + 2;
            s_pad[ip] = fm;
            s_pad[PP + ip] = mean;
            comp[((size_t)n * 2 + 0) * HW + tid] = fm;
            comp[((size_t)n * 2 + 1) * HW + tid] = mean;
            cargmax[(size_t)n * HW + tid] = fa;
        }
    }
    __syncthreads();
    M3T_CB_STAMP(5);
    spatial_conv_fwd(s_pad, convw, red, conv_out, part, n, H, W);
    M3T_CB_STAMP(7);
}

template <int NV>
__global__ __launch_bounds__(FT, NV <= 4 ? 6 : 4) void cbam_b2l_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ w1, const float* __restrict__ w2,
                                                     const float* __restrict__ convw, const float* __restrict__ bn_w,
                                                     const float* __restrict__ stats, const float* __restrict__ dgb,
                                                     const float* __restrict__ cs, const int32_t* __restrict__ argmax_p,
                                                     const float* __restrict__ hidden, const float* __restrict__ comp,
                                                     const int32_t* __restrict__ cargmax, const float* __restrict__ xhat,
                                                     const float* __restrict__ ss, const float* __restrict__ dpre,
                                                     float* __restrict__ dx, float* __restrict__ g_datt, float* __restrict__ g_dh,
                                                     float* __restrict__ g_r, float* __restrict__ dwpart, int C, int Cr, int H,
                                                     int W, float inv_total, int training) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HW = H * W, S = odd_stride(HW);
    const B2Lds L = b2_lds_map(sm, H, W);
    float4* s_ch = reinterpret_cast<float4*>(L.rest);          // [C] {cs, davg / HW, dmaxc, argmax_p}
    float* s_datt = L.rest + 4 * al4(C);
    float* s_davg = s_datt + al4(C);
    float* s_dmaxc = s_davg + al4(C);
    float* s_dh = s_dmaxc + al4(C);            // [2 Cr]
    float* s_part = s_dh + al4(2 * Cr);        // [8][Cr]
    float4* s_pix = reinterpret_cast<float4*>(s_part + 8 * Cr);      // [HW] {ss, dmean / C, dmax, argmax_c}
    float* s_f = reinterpret_cast<float*>(s_pix + al4(HW));    // [C][S] dy1 * x
    const int n = blockIdx.x, tid = threadIdx.x;
    const int n4 = (C * HW) >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(dy + (size_t)n * C * HW);
    const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)n * C * HW);
    M3T_CB_STAMP(0);
    // ---- both frames in flight before anything else; they land while the spatial gate's backward runs
    float4 gy[NV], xv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = tid + k * FT;
        gy[k] = i < n4 ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        xv[k] = i < n4 ? x4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    spatial_bwd(L, convw, bn_w, stats, dgb, comp, cargmax, xhat, ss, dpre, dwpart, n, H, W, C, inv_total, training);
    for (int p = tid; p < HW; p += FT)         // (dmx, dmn of pixel p were written by this thread)
        s_pix[p] = make_float4(L.ss[p], L.dmn[p], L.dmx[p], __int_as_float(L.cam[p]));
    for (int c = tid; c < C; c += FT) s_ch[c] = make_float4(cs[(size_t)n * C + c], 0.f, 0.f, __int_as_float(argmax_p[(size_t)n * C + c]));
    __syncthreads();
    M3T_CB_STAMP(2);
    // ---- d. dcs[c] = sum_p dy1 * x: the products elementwise into LDS, then a thread per plane
    const float inv_hw = 1.f / (float)HW;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = tid + k * FT;
        if (i < n4) {
            int c, p;
            plane_pixel(4 * i, HW, inv_hw, c, p);
            const float g[4] = {gy[k].x, gy[k].y, gy[k].z, gy[k].w}, xe[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 px = s_pix[p];
                const float dy1 = g[j] * px.x + px.y + (c == __float_as_int(px.w) ? px.z : 0.f);
                s_f[c * S + p] = dy1 * xe[j];
                if (++p == HW) { p = 0; ++c; }
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += FT) {
        const float* pl = s_f + c * S;
        float ds = 0.f;
#pragma unroll 4
        for (int p = 0; p < HW; ++p) ds += pl[p];
        const float sc = s_ch[c].x;
        const float da = ds * sc * (1.f - sc);
        s_datt[c] = da;
        g_datt[(size_t)n * C + c] = da;
    }
    __syncthreads();
    M3T_CB_STAMP(3);
    gate_mlp_bwd(s_datt, s_part, s_dh, s_davg, s_dmaxc, w1, w2, hidden, g_dh, g_r, n, C, Cr, HW);
    __syncthreads();
    for (int c = tid; c < C; c += FT) { s_ch[c].y = s_davg[c]; s_ch[c].z = s_dmaxc[c]; }
    __syncthreads();
    M3T_CB_STAMP(6);
    // ---- f. dx = dy1 * cs + davg / HW + [p == argmax_p] dmaxc from the dy registers, one float4 store per load
    float4* d4 = reinterpret_cast<float4*>(dx + (size_t)n * C * HW);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = tid + k * FT;
        if (i < n4) {
            int c, p;
            plane_pixel(4 * i, HW, inv_hw, c, p);
            const float g[4] = {gy[k].x, gy[k].y, gy[k].z, gy[k].w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 px = s_pix[p];
                const float4 ch = s_ch[c];
                const float dy1 = g[j] * px.x + px.y + (c == __float_as_int(px.w) ? px.z : 0.f);
                o[j] = dy1 * ch.x + ch.y + (p == __float_as_int(ch.w) ? ch.z : 0.f);
                if (++p == HW) { p = 0; ++c; }
            }
            d4[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    M3T_CB_STAMP(7);
}

struct Geo { int E, Q, G, Qp, nk; };

// units of E pixels: Q per plane (at most FT: one thread per unit in the pixel reductions)
bool geo_for(int HW, int E, Geo& g) {
    g.E = E;
    g.Q = HW / E;
    if (g.Q > FT || g.Q * E != HW) return false;
    int p2 = 1;
    while (p2 < g.Q) p2 <<= 1;
    g.Qp = p2;
    g.G = p2 < 64 ? p2 : 64;
    g.nk = g.Qp < 64 ? FT / 64 : FT / g.Qp;
    return true;
}

// float4 units when H*W is a multiple of 4 and every swept tensor is 16-B aligned, else single pixels
bool geometry(int HW, const void* const* ptrs, int nptr, Geo& g) {
    bool al = true;
    for (int i = 0; i < nptr; ++i) al = al && (((uintptr_t)ptrs[i] & 15) == 0);
    return geo_for(HW, (HW % 4 == 0 && al) ? 4 : 1, g);
}

size_t f1_lds(int C, int Cr, int PP, const Geo& g) {
    return (size_t)(3 * al4(C) + al4(2 * Cr) + al4(2 * PP) + 3 * g.nk * g.Qp * g.E + al4(C)) * sizeof(float);
}
size_t b2_lds(int C, int Cr, int H, int W) {
    return (size_t)(b2_map_floats(H, W) + 4 * al4(C) + al4(2 * Cr) + 8 * Cr + al4(C)) * sizeof(float);
}

// frame-resident path: LDS bytes of F1L / B2L, eligibility, pixel lanes
int resident_qp(int HW) { int q = 1; while (q < HW) q <<= 1; return q; }
size_t f1l_lds(int C, int Cr, int H, int W) {
    const int HW = H * W;
    return (size_t)(al4(C * odd_stride(HW)) + 3 * al4(C) + al4(2 * Cr) + al4(2 * pad_len(H, W)) + 3 * (FT / 64) * resident_qp(HW) + al4(C)) * sizeof(float);
}
size_t b2l_lds(int C, int Cr, int H, int W) {
    const int HW = H * W;
    return (size_t)(b2_map_floats(H, W) + 4 * al4(C) + 3 * al4(C) + al4(2 * Cr) + 8 * Cr + 4 * al4(HW) + al4(C * odd_stride(HW))) * sizeof(float);
}
bool resident_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("M3T_CBAM_RESIDENT"); on = (e && e[0] == '0') ? 0 : 1; }
    return on == 1;
}
// NV (float4 per thread) of the frame-resident kernels, 0 = not eligible
int resident_nv(int C, int Cr, int H, int W, const void* const* ptrs, int nptr) {
    const int HW = H * W;
    if (!resident_enabled() || HW > 64 || ((C * HW) & 3) != 0 || C * HW > 8 * 4 * FT) return 0;
    for (int i = 0; i < nptr; ++i)
        if (((uintptr_t)ptrs[i] & 15) != 0) return 0;
    if (f1l_lds(C, Cr, H, W) > 64 * 1024 || b2l_lds(C, Cr, H, W) > 64 * 1024) return 0;
    return C * HW <= 4 * 4 * FT ? 4 : 8;
}

}  // namespace

extern "C" int m3t_cbam_fused_ok(int C, int Cr, int H, int W) {
    if (C <= 0 || Cr <= 0 || Cr > 64 || H <= 0 || W <= 0) return 0;
    const int HW = H * W;
    Geo g;
    if (!geo_for(HW, HW % 4 == 0 ? 4 : 1, g)) return 0;
    return (f1_lds(C, Cr, pad_len(H, W), g) <= 60 * 1024 && b2_lds(C, Cr, H, W) <= 100 * 1024) ? 1 : 0;
}

extern "C" size_t m3t_cbam_fused_ws_bytes(int N, int C, int Cr, int H, int W) {
    // doubles: part[2 N]; floats: dgb[2 -> 4], dpre[N HW], dwpart[N 50], g_datt[N C], g_dh[N 2 Cr], g_r[N Cr] + GEMM scratch
    const size_t HW = (size_t)H * W;
    return (size_t)N * 2 * sizeof(double) + ((size_t)4 + N * HW + (size_t)N * 50 + (size_t)N * (C + 3 * Cr) +
                                             (size_t)((N + PG_CHUNK - 1) / PG_CHUNK) * (2 * (size_t)C * Cr + C + Cr + 50)) * sizeof(float) + 256;
}

extern "C" int m3t_cbam_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                            const float* conv_w, const float* bn_w, const float* bn_b, float* running_mean, float* running_var,
                            float* y, float* cs, int32_t* argmax_p,
                            float* pooled, float* hidden, float* comp, int32_t* cargmax, float* xhat, float* ss, float* stats,
                            int N, int C, int Cr, int H, int W, int training, float momentum, float eps, float* ws,
                            size_t ws_bytes, void* stream) {
    if (N <= 0) return 0;
    if (!x || !w1 || !b1 || !w2 || !b2 || !conv_w || !bn_w || !bn_b || !running_mean || !running_var || !y || !cs || !argmax_p || !pooled || !hidden || !comp ||
        !cargmax || !xhat || !ss || !stats || !ws || ((uintptr_t)ws & 7) != 0)
        return M3T_EINVAL;
    if (!m3t_cbam_fused_ok(C, Cr, H, W)) return M3T_EINVAL;
    if (ws_bytes < (size_t)N * 2 * sizeof(double)) return M3T_EINVAL;
    const int HW = H * W;
    const void* ptrs[] = {x, y, comp, cargmax, xhat, ss};
    Geo g;
    if (!geometry(HW, ptrs, 6, g)) return M3T_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(ws);
    const size_t lds1 = f1_lds(C, Cr, pad_len(H, W), g), lds2 = (size_t)(al4(C) + al4(HW)) * sizeof(float);
    const bool small = g.Q <= g.G;       // one unit per lane per plane: batch planes instead of units
#define M3T_CBAM_PICK(K, ...)                                                 \
    do {                                                                     \
        if (g.E == 4) { if (small) K<4, true> __VA_ARGS__; else K<4, false> __VA_ARGS__; } \
        else { if (small) K<1, true> __VA_ARGS__; else K<1, false> __VA_ARGS__; }         \
    } while (0)
    const void* rptr[] = {x};
    const int nv = resident_nv(C, Cr, H, W, rptr, 1);
    if (nv == 4)
        cbam_f1l_kernel<4><<<N, FT, f1l_lds(C, Cr, H, W), s>>>(x, w1, b1, w2, b2, conv_w, cs, argmax_p, pooled, hidden, comp, cargmax, xhat, part, C, Cr, H, W, resident_qp(HW));
    else if (nv == 8)
        cbam_f1l_kernel<8><<<N, FT, f1l_lds(C, Cr, H, W), s>>>(x, w1, b1, w2, b2, conv_w, cs, argmax_p, pooled, hidden, comp, cargmax, xhat, part, C, Cr, H, W, resident_qp(HW));
    else
        M3T_CBAM_PICK(cbam_f1_kernel, <<<N, FT, lds1, s>>>(x, w1, b1, w2, b2, conv_w, cs, argmax_p, pooled, hidden, comp, cargmax, xhat, part, C, Cr, H, W, g.G, g.Qp));
    M3T_LAUNCH_CHECK();
    cbam_stats_kernel<<<1, 256, 0, s>>>(part, N, (double)N * HW, running_mean, running_var, stats, training, momentum, eps);
    M3T_LAUNCH_CHECK();
    M3T_CBAM_PICK(cbam_f2_kernel, <<<N, FT, lds2, s>>>(x, cs, bn_w, bn_b, stats, xhat, ss, y, C, HW, g.G));
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_cbam_bwd(const float* dy, const float* x, const float* w1, const float* w2, const float* conv_w,
                            const float* bn_w, const float* cs, const int32_t* argmax_p, const float* pooled,
                            const float* hidden, const float* comp, const int32_t* cargmax, const float* xhat, const float* ss,
                            const float* stats, float* dx, float* dw1, float* db1, float* dw2, float* db2, float* dconv_w,
                            float* dbn_w, float* dbn_b, int N, int C, int Cr, int H, int W, int training, float* ws,
                            size_t ws_bytes, void* stream) {
    if (N <= 0) return 0;
    if (!dy || !x || !w1 || !w2 || !conv_w || !bn_w || !cs || !argmax_p || !pooled || !hidden || !comp || !cargmax || !xhat || !ss ||
        !stats || !dx || !dw1 || !db1 || !dw2 || !db2 || !dconv_w || !dbn_w || !dbn_b || !ws || ((uintptr_t)ws & 15) != 0)
        return M3T_EINVAL;
    if (!m3t_cbam_fused_ok(C, Cr, H, W)) return M3T_EINVAL;
    const int HW = H * W;
    const size_t total = (size_t)N * HW;
    if (ws_bytes < m3t_cbam_fused_ws_bytes(N, C, Cr, H, W) || Cr > 64) return M3T_EINVAL;
    double* part = reinterpret_cast<double*>(ws);
    float* f = reinterpret_cast<float*>(part + 2 * (size_t)N);
    float* dgb = f;                              // (d gamma, d beta), 16 B
    float* dpre = f + 4;                         // [N, HW]
    float* dwpart = dpre + ((total + 3) & ~(size_t)3);     // [N, 50]
    float* g_datt = dwpart + (((size_t)N * 50 + 3) & ~(size_t)3);     // [N, C]
    float* g_dh = g_datt + (size_t)N * C;        // [N, 2, Cr]
    float* g_r = g_dh + (size_t)N * 2 * Cr;      // [N, Cr]
    float* pgpart = g_r + (((size_t)N * Cr + 3) & ~(size_t)3);      // [N / 32 slices][2 C Cr + C + Cr]
    const void* ptrs[] = {x, dy, dx, ss, xhat, dpre};
    Geo g;
    if (!geometry(HW, ptrs, 6, g)) return M3T_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds1 = (size_t)(al4(C) + g.nk * g.Qp * g.E) * sizeof(float), lds2 = b2_lds(C, Cr, H, W);
    if (g.E == 4) cbam_b1_kernel<4><<<N, FT, lds1, s>>>(dy, x, cs, ss, xhat, dpre, part, C, HW, g.Qp);
    else cbam_b1_kernel<1><<<N, FT, lds1, s>>>(dy, x, cs, ss, xhat, dpre, part, C, HW, g.Qp);
    M3T_LAUNCH_CHECK();
    cbam_sum_pairs_kernel<<<1, 256, 0, s>>>(part, N, dgb, dbn_w, dbn_b);
    M3T_LAUNCH_CHECK();
    const bool small = g.Q <= g.G;
    const void* rptr[] = {x, dy, dx};
    const int nv = resident_nv(C, Cr, H, W, rptr, 3);
    if (nv == 4)
        cbam_b2l_kernel<4><<<N, FT, b2l_lds(C, Cr, H, W), s>>>(dy, x, w1, w2, conv_w, bn_w, stats, dgb, cs, argmax_p, hidden, comp, cargmax, xhat, ss, dpre, dx, g_datt, g_dh, g_r, dwpart, C, Cr, H, W, (float)(1.0 / (double)total), training);
    else if (nv == 8)
        cbam_b2l_kernel<8><<<N, FT, b2l_lds(C, Cr, H, W), s>>>(dy, x, w1, w2, conv_w, bn_w, stats, dgb, cs, argmax_p, hidden, comp, cargmax, xhat, ss, dpre, dx, g_datt, g_dh, g_r, dwpart, C, Cr, H, W, (float)(1.0 / (double)total), training);
    else {
    if (lds2 > 64 * 1024) {       // big maps: B2's padded per-pixel maps pass the 64 KB a launch gets without asking
        const void* kf = g.E == 4 ? (small ? (const void*)cbam_b2_kernel<4, true> : (const void*)cbam_b2_kernel<4, false>)
                                  : (small ? (const void*)cbam_b2_kernel<1, true> : (const void*)cbam_b2_kernel<1, false>);
        if (hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return M3T_EINVAL;
        }
    }
    M3T_CBAM_PICK(cbam_b2_kernel, <<<N, FT, lds2, s>>>(dy, x, w1, w2, conv_w, bn_w, stats, dgb, cs, argmax_p, hidden, comp, cargmax, xhat, ss, dpre, dx, g_datt, g_dh, g_r, dwpart, C, Cr, H, W, g.G, (float)(1.0 / (double)total), training));
    }
    M3T_LAUNCH_CHECK();
    const int slices = (N + PG_CHUNK - 1) / PG_CHUNK;
    {
        const dim3 pg((C + 63) / 64, slices);
#define M3T_PG(P) cbam_pgrad_partial_kernel<P><<<pg, 256, 0, s>>>(g_datt, g_r, g_dh, pooled, dwpart, pgpart, N, C, Cr)
        if (Cr == 32) M3T_PG(8); else if (Cr == 16) M3T_PG(4); else if (Cr == 8) M3T_PG(2); else if (Cr == 4) M3T_PG(1); else M3T_PG(0);
#undef M3T_PG
    }
    M3T_LAUNCH_CHECK();
    int fb = (int)((2 * (size_t)C * Cr + C + Cr + 50 + 255) / 256);
    if (fb > 256) fb = 256;
    cbam_pgrad_final_kernel<<<fb, 256, 0, s>>>(pgpart, slices, C, Cr, dw2, dw1, db2, db1, dconv_w);
    M3T_LAUNCH_CHECK();
    return 0;
}
