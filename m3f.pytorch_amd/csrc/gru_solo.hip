// "Solo" BiGRU scans for H = 128 (the AttFusion scorers, reference models/att_fusion.py:14-15: GRU(512, 128, 1, 1, 1)).
// One launch runs all T steps, ONE WORKGROUP PER (scan, clip): nothing is exchanged between workgroups -- no tags, no polls,
// no residency requirement (any number of these launches may share the GPU with anything).
//
// Why not the persistent kernels of gru_persist.hip at this size: they split the 128 hidden units over 8 workgroups that
// exchange h_t through L2 every step -- at H = 128 the exchange is the whole step (1.66 us forward, 2.37 us backward per
// step).  The recurrent product of ONE clip is a 128 x 384 matrix-vector product: 98 kFLOP, 1/8 of what a padded 16-row MFMA
// tile spends on it, and small enough for the vector ALUs of one CU with W_hh in registers:
//   * 512 threads; wave w owns hidden units 16w .. 16w+15 (lane & 15), the four 16-lane rows of a wave split K four ways
//     (forward: k = 32 row .. +31 of h_{t-1}; backward: c = 96 row .. +95 of the 384 gate gradients);
//   * a thread keeps its 96 weights in VGPRs for the whole scan;
//   * the vector operand lives in 2 (forward) / 6 (backward) VGPRs, lane l holding element 16 m + (l & 15) of its row's range,
//     and is broadcast inside the 16-lane row by the DPP row_newbcast modifier of v_fmac_f32: one instruction per (weight, k),
//     no LDS traffic, no v_readlane;
//   * the four row partials are summed with two cross-row shuffles; every row then runs the cell arithmetic of its unit
//     redundantly (same instruction count as one row doing it), row 0 stores;
//   * the new vector goes through a 0.5 / 1.5 KiB double-buffered LDS array: one barrier per step.
// Per step: 96 v_fmac + ~60 other VALU instructions per wave, two waves per SIMD: ~0.55 us (forward) instead of 1.66 us.
// Arithmetic: plain fp32 FMA chains (32- or 96-term partial sums, then a 4-leaf tree) -- as accurate as the fp32 MFMA
// kernels, not bit-identical to them (different summation order); M3T_SCAN_FP32 / M3T_SCAN_NO_PERSIST keep the other paths.
// The bf16 mode (M3T_BF16) rounds W_hh and the vector operand to bf16 first, like every other scan kernel.
#include "gru_common.h"
#include <cstdlib>

namespace m3t_gru {
namespace {

constexpr int SH = 128;        // hidden size served
constexpr int ST = 512;        // threads: 8 waves x (16 units x 4 k-rows)

// One asm statement = "s_nop 1" + a run of v_fmac_f32_dpp that all read the SAME broadcast source register(s).  A VALU write of a
// VGPR followed by a DPP read of it needs two wait states, and LLVM's hazard recognizer does not look inside inline asm: with the
// s_nop in a statement of its own (as before) nothing stopped the register allocator from placing a v_mov copy of the vector
// between it and the first DPP instruction.  Inside one statement there is no such gap: whatever the compiler does
// in front of the statement -- copies included -- is followed by the nop.  (Inline asm takes at most 30 operands: hence the
// blocks of 8 broadcast positions forward, 2 backward.)
#define M3T_DPPL(acc, w, n) "v_fmac_f32_dpp " acc ", %3, " w " row_newbcast:" #n " row_mask:0xf bank_mask:0xf\n\t"
#define M3T_DPP_ROW3(i, n) M3T_DPPL("%0", "%" #i, n)
// forward: three accumulators x 8 broadcast positions [n0, n0 + 8) of ONE vector register against w0/w1/w2[base + n]
#define M3T_DPP3x8_ASM(N0, N1, N2, N3, N4, N5, N6, N7)                                                                       \
    "s_nop 1\n\t"                                                                                                         \
    M3T_DPPL("%0", "%4", N0) M3T_DPPL("%1", "%5", N0) M3T_DPPL("%2", "%6", N0)                                              \
    M3T_DPPL("%0", "%7", N1) M3T_DPPL("%1", "%8", N1) M3T_DPPL("%2", "%9", N1)                                              \
    M3T_DPPL("%0", "%10", N2) M3T_DPPL("%1", "%11", N2) M3T_DPPL("%2", "%12", N2)                                           \
    M3T_DPPL("%0", "%13", N3) M3T_DPPL("%1", "%14", N3) M3T_DPPL("%2", "%15", N3)                                           \
    M3T_DPPL("%0", "%16", N4) M3T_DPPL("%1", "%17", N4) M3T_DPPL("%2", "%18", N4)                                           \
    M3T_DPPL("%0", "%19", N5) M3T_DPPL("%1", "%20", N5) M3T_DPPL("%2", "%21", N5)                                           \
    M3T_DPPL("%0", "%22", N6) M3T_DPPL("%1", "%23", N6) M3T_DPPL("%2", "%24", N6)                                           \
    M3T_DPPL("%0", "%25", N7) M3T_DPPL("%1", "%26", N7) M3T_DPPL("%2", "%27", N7)
#define M3T_DPP3x8_OPS(a0, a1, a2, vec, w0, w1, w2, b)                                                                       \
    : "+v"(a0), "+v"(a1), "+v"(a2)                                                                                           \
    : "v"(vec), "v"(w0[(b) + 0]), "v"(w1[(b) + 0]), "v"(w2[(b) + 0]), "v"(w0[(b) + 1]), "v"(w1[(b) + 1]), "v"(w2[(b) + 1]),      \
      "v"(w0[(b) + 2]), "v"(w1[(b) + 2]), "v"(w2[(b) + 2]), "v"(w0[(b) + 3]), "v"(w1[(b) + 3]), "v"(w2[(b) + 3]),                \
      "v"(w0[(b) + 4]), "v"(w1[(b) + 4]), "v"(w2[(b) + 4]), "v"(w0[(b) + 5]), "v"(w1[(b) + 5]), "v"(w2[(b) + 5]),                \
      "v"(w0[(b) + 6]), "v"(w1[(b) + 6]), "v"(w2[(b) + 6]), "v"(w0[(b) + 7]), "v"(w1[(b) + 7]), "v"(w2[(b) + 7])
// 16 k of one vector register against three weight sets (same instruction order as before: n ascending, gates r, z, n inside)
#define M3T_DPP3x16(a0, a1, a2, vec, w0, w1, w2, base)                                                                       \
    do {                                                                                                                    \
        asm(M3T_DPP3x8_ASM(0, 1, 2, 3, 4, 5, 6, 7) M3T_DPP3x8_OPS(a0, a1, a2, vec, w0, w1, w2, (base)));                      \
        asm(M3T_DPP3x8_ASM(8, 9, 10, 11, 12, 13, 14, 15) M3T_DPP3x8_OPS(a0, a1, a2, vec, w0, w1, w2, (base) + 8));            \
    } while (0)
// backward: broadcast positions n, n + 1 of SIX vector registers (dv[0..5]) -- operands: %0-%2 accumulators, %3-%8 vectors,
// %9.. weights in the order (n: w0[n], w1[n], w2[n], w0[16+n], w1[16+n], w2[16+n]), then the same for n + 1
#define M3T_DPPLB(acc, vec, w, n) "v_fmac_f32_dpp " acc ", " vec ", " w " row_newbcast:" #n " row_mask:0xf bank_mask:0xf\n\t"
#define M3T_DPPB2_ASM(NA, NB)                                                                                               \
    "s_nop 1\n\t"                                                                                                         \
    M3T_DPPLB("%0", "%3", "%9", NA) M3T_DPPLB("%1", "%5", "%10", NA) M3T_DPPLB("%2", "%7", "%11", NA)                        \
    M3T_DPPLB("%0", "%4", "%12", NA) M3T_DPPLB("%1", "%6", "%13", NA) M3T_DPPLB("%2", "%8", "%14", NA)                       \
    M3T_DPPLB("%0", "%3", "%15", NB) M3T_DPPLB("%1", "%5", "%16", NB) M3T_DPPLB("%2", "%7", "%17", NB)                       \
    M3T_DPPLB("%0", "%4", "%18", NB) M3T_DPPLB("%1", "%6", "%19", NB) M3T_DPPLB("%2", "%8", "%20", NB)
#define M3T_DPPB2(a0, a1, a2, dv, w0, w1, w2, NA, NB)                                                                        \
    asm(M3T_DPPB2_ASM(NA, NB)                                                                                               \
        : "+v"(a0), "+v"(a1), "+v"(a2)                                                                                       \
        : "v"(dv[0]), "v"(dv[1]), "v"(dv[2]), "v"(dv[3]), "v"(dv[4]), "v"(dv[5]),                                            \
          "v"(w0[NA]), "v"(w1[NA]), "v"(w2[NA]), "v"(w0[16 + NA]), "v"(w1[16 + NA]), "v"(w2[16 + NA]),                       \
          "v"(w0[NB]), "v"(w1[NB]), "v"(w2[NB]), "v"(w0[16 + NB]), "v"(w1[16 + NB]), "v"(w2[16 + NB]))

// sum over the four 16-lane rows of a wave; every lane ends with the total
__device__ __forceinline__ float row_sum4(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

__global__ __launch_bounds__(ST) void gru_solo_fwd_kernel(FwdGroup g, int B, int T) {
    const int s = (int)blockIdx.x / B, b = (int)blockIdx.x % B;
    const m3t_gru_fwd_desc d = g.d[s];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane >> 4, l16 = lane & 15;
    const int j = wave * 16 + l16;                     // this thread's hidden unit
    const bool bf = g.bf16 != 0;
    __shared__ float hbuf[2][SH];

    float wr[32], wz[32], wn[32];                      // W_hh[gate * 128 + j][32 row .. +31]
    {
        const float* base = d.w_hh + (size_t)j * SH + 32 * row;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(base + 4 * q);
            const float4 c = *reinterpret_cast<const float4*>(base + (size_t)SH * SH + 4 * q);
            const float4 e = *reinterpret_cast<const float4*>(base + (size_t)2 * SH * SH + 4 * q);
            wr[4 * q] = a.x; wr[4 * q + 1] = a.y; wr[4 * q + 2] = a.z; wr[4 * q + 3] = a.w;
            wz[4 * q] = c.x; wz[4 * q + 1] = c.y; wz[4 * q + 2] = c.z; wz[4 * q + 3] = c.w;
            wn[4 * q] = e.x; wn[4 * q + 1] = e.y; wn[4 * q + 2] = e.z; wn[4 * q + 3] = e.w;
        }
        if (bf) {
#pragma unroll
            for (int k = 0; k < 32; ++k) { wr[k] = rbf(wr[k]); wz[k] = rbf(wz[k]); wn[k] = rbf(wn[k]); }
        }
    }
    const float br = d.b_hh[j], bz = d.b_hh[SH + j], bn = d.b_hh[2 * SH + j];
    float hj = 0.f, hv0 = 0.f, hv1 = 0.f;             // h_{t-1}[j]; h_{t-1}[32 row + l16], h_{t-1}[32 row + 16 + l16]

    const float* px = d.xproj + (size_t)b * T * d.ldx + d.xoff + j;
    const int dt = d.reverse ? -1 : 1;
    int t = d.reverse ? T - 1 : 0;
    float xr = px[(size_t)t * d.ldx], xz = px[(size_t)t * d.ldx + SH], xn = px[(size_t)t * d.ldx + 2 * SH];
    for (int step = 0; step < T; ++step) {
        const int tn = step + 1 < T ? t + dt : t;      // next step's x-projection: a whole step to arrive
        const float nxr = px[(size_t)tn * d.ldx], nxz = px[(size_t)tn * d.ldx + SH], nxn = px[(size_t)tn * d.ldx + 2 * SH];
        float ar = 0.f, az = 0.f, an = 0.f;
        M3T_DPP3x16(ar, az, an, hv0, wr, wz, wn, 0);              // (each asm block opens with the s_nop the DPP reads need)
        M3T_DPP3x16(ar, az, an, hv1, wr, wz, wn, 16);
        const float hr = row_sum4(ar) + br, hz = row_sum4(az) + bz, hn = row_sum4(an) + bn;
        const GateFwd c = gru_cell_fwd(xr, xz, xn, hr, hz, hn, hj);
        hj = c.h;
        if (row == 0) {
            hbuf[step & 1][j] = bf ? rbf(c.h) : c.h;
            d.out[((size_t)b * T + t) * d.ldo + d.ooff + j] = c.h;
            if (d.gates) *reinterpret_cast<float4*>(d.gates + (((size_t)b * T + t) * SH + j) * 4) = make_float4(c.r, c.z, c.n, hn);
            if (d.h_n && step == T - 1) d.h_n[(size_t)b * SH + j] = c.h;
        }
        __syncthreads();       // double-buffered: a wave writes hbuf[p] again only after everyone has passed the NEXT barrier
        hv0 = hbuf[step & 1][32 * row + l16];
        hv1 = hbuf[step & 1][32 * row + 16 + l16];
        xr = nxr; xz = nxz; xn = nxn; t = tn;
    }
}

// `direct`: desc.w_hh_t is the untransposed parameter w_hh [3H][H] (M3T_SCAN_WHH), else W_hh^T [H][3H]
__global__ __launch_bounds__(ST) void gru_solo_bwd_kernel(BwdGroup g, int B, int T, int direct) {
    const int s = (int)blockIdx.x / B, b = (int)blockIdx.x % B;
    const m3t_gru_bwd_desc d = g.d[s];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane >> 4, l16 = lane & 15;
    const int j = wave * 16 + l16;
    const bool bf = g.bf16 != 0;
    constexpr int H3 = 3 * SH;
    __shared__ float dbuf[2][H3];

    float w0[32], w1[32], w2[32];                      // W_hh[96 row + 32 i + k][j], i = 0..2: the rows this 16-lane row reduces over
    if (direct) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            w0[k] = d.w_hh_t[(size_t)(96 * row + k) * SH + j];
            w1[k] = d.w_hh_t[(size_t)(96 * row + 32 + k) * SH + j];
            w2[k] = d.w_hh_t[(size_t)(96 * row + 64 + k) * SH + j];
        }
    } else {
        const float* base = d.w_hh_t + (size_t)j * H3 + 96 * row;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(base + 4 * q);
            const float4 c = *reinterpret_cast<const float4*>(base + 32 + 4 * q);
            const float4 e = *reinterpret_cast<const float4*>(base + 64 + 4 * q);
            w0[4 * q] = a.x; w0[4 * q + 1] = a.y; w0[4 * q + 2] = a.z; w0[4 * q + 3] = a.w;
            w1[4 * q] = c.x; w1[4 * q + 1] = c.y; w1[4 * q + 2] = c.z; w1[4 * q + 3] = c.w;
            w2[4 * q] = e.x; w2[4 * q + 1] = e.y; w2[4 * q + 2] = e.z; w2[4 * q + 3] = e.w;
        }
    }
    if (bf) {
#pragma unroll
        for (int k = 0; k < 32; ++k) { w0[k] = rbf(w0[k]); w1[k] = rbf(w1[k]); w2[k] = rbf(w2[k]); }
    }
    float dh_carry = d.dh_n ? d.dh_n[(size_t)b * SH + j] : 0.f, z_next = 0.f;
    float sb_r = 0.f, sb_z = 0.f, sb_n = 0.f, sb_nr = 0.f;
    float amx = 0.f;                                   // max |dr~|, |dz~|, |dn~|: the scan's magnitude slot (fp16x3 GEMMs)
    float dv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // dgh_{t+1}[96 row + 16 m + l16]

    const float* pd = d.dout + (size_t)b * T * d.ldo + d.ooff + j;
    const float* pg = d.gates + ((size_t)b * T * SH + j) * 4;
    const float* ph = d.out + (size_t)b * T * d.ldo + d.ooff + j;
    // step -> time index processed (a forward-direction scan is walked back from T-1) and the time index of its h_{t-1}
    auto t_of = [&](int st) { return d.reverse ? st : T - 1 - st; };
    auto tp_of = [&](int st) { const int lt = t_of(st); return st < T - 1 ? (d.reverse ? lt + 1 : lt - 1) : lt; };
    float dout = pd[(size_t)t_of(0) * d.ldo], hprev = ph[(size_t)tp_of(0) * d.ldo];
    float4 g4 = *reinterpret_cast<const float4*>(pg + (size_t)t_of(0) * SH * 4);
    for (int step = 0; step < T; ++step) {
        const int sn = step + 1 < T ? step + 1 : step;
        const float ndout = pd[(size_t)t_of(sn) * d.ldo], nhprev = ph[(size_t)tp_of(sn) * d.ldo];
        const float4 ng4 = *reinterpret_cast<const float4*>(pg + (size_t)t_of(sn) * SH * 4);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        // three accumulators, each over 32 of this row's 96 gate gradients: (dv[0], dv[1]) x w0, (dv[2], dv[3]) x w1, (dv[4], dv[5]) x w2;
        // same instruction order as before (n ascending; per n: a0/a1/a2 against dv[0]/dv[2]/dv[4], then against dv[1]/dv[3]/dv[5])
        M3T_DPPB2(a0, a1, a2, dv, w0, w1, w2, 0, 1); M3T_DPPB2(a0, a1, a2, dv, w0, w1, w2, 2, 3);
        M3T_DPPB2(a0, a1, a2, dv, w0, w1, w2, 4, 5); M3T_DPPB2(a0, a1, a2, dv, w0, w1, w2, 6, 7);
        M3T_DPPB2(a0, a1, a2, dv, w0, w1, w2, 8, 9); M3T_DPPB2(a0, a1, a2, dv, w0, w1, w2, 10, 11);
        M3T_DPPB2(a0, a1, a2, dv, w0, w1, w2, 12, 13); M3T_DPPB2(a0, a1, a2, dv, w0, w1, w2, 14, 15);
        const float mm = row_sum4((a0 + a1) + a2);
        const int t = t_of(step);
        const GateBwd c = gru_cell_bwd(dout, dh_carry, z_next, mm, step > 0, g4.x, g4.y, g4.z, g4.w, step < T - 1 ? hprev : 0.f);
        dh_carry = c.dht; z_next = g4.y;
        sb_r += c.dr; sb_z += c.dz; sb_n += c.dn; sb_nr += c.dnr;
        amx = fmaxf(fmaxf(amx, m3t_fin_abs(c.dr)), fmaxf(m3t_fin_abs(c.dz), m3t_fin_abs(c.dn)));      // (|dn r| <= |dn|: one bound for dgx and dgh)
        if (row == 0) {
            dbuf[step & 1][j] = bf ? rbf(c.dr) : c.dr;
            dbuf[step & 1][SH + j] = bf ? rbf(c.dz) : c.dz;
            dbuf[step & 1][2 * SH + j] = bf ? rbf(c.dnr) : c.dnr;
            float* gx = d.dgx + ((size_t)b * T + t) * d.ldg + d.goff;
            gx[j] = c.dr; gx[SH + j] = c.dz; gx[2 * SH + j] = c.dn;
            float* gh = d.dgh + ((size_t)b * T + t) * H3;
            gh[j] = c.dr; gh[SH + j] = c.dz; gh[2 * SH + j] = c.dnr;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 6; ++m) dv[m] = dbuf[step & 1][96 * row + 16 * m + l16];
        dout = ndout; hprev = nhprev; g4 = ng4;
    }
    if (row == 0) {
        d.dh[(size_t)b * SH + j] = dh_carry;
        if (d.db_part) {
            float* q = d.db_part + (size_t)b * 4 * SH + j;
            q[0] = sb_r; q[SH] = sb_z; q[2 * SH] = sb_n; q[3 * SH] = sb_nr;
        }
    }
    if (d.amax) {                                      // (every row computes the same cell values: any of them serves)
        const float m = wave_max(amx);
        if ((threadIdx.x & 63) == 0) atomicMax(d.amax, (unsigned long long)__float_as_uint(m));
    }
}

bool solo_enabled() {
    static int on = -1;
    if (on < 0) {
        const char* e = std::getenv("M3T_SCAN_SOLO");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    return on == 1;
}

}  // namespace

// every scan of the level has H = 128, float4-aligned operands, and the caller has not asked for a specific other path
bool solo_fwd_ok(const FwdGroup& g, int B, int T, int flags) {
    if (!solo_enabled() || (flags & (M3T_SCAN_NO_PERSIST | M3T_SCAN_FP32 | M3T_SCAN_FAULT)) || T < 2 || g.n < 1) return false;
    for (int i = 0; i < g.n; ++i) {
        const m3t_gru_fwd_desc& d = g.d[i];
        if (d.H != SH || (uintptr_t)d.w_hh % 16 != 0 || (d.gates && (uintptr_t)d.gates % 16 != 0)) return false;
    }
    return true;
}

bool solo_bwd_ok(const BwdGroup& g, int B, int T, int flags) {
    if (!solo_enabled() || (flags & (M3T_SCAN_NO_PERSIST | M3T_SCAN_FP32 | M3T_SCAN_FAULT)) || T < 2 || g.n < 1) return false;
    for (int i = 0; i < g.n; ++i) {
        const m3t_gru_bwd_desc& d = g.d[i];
        if (d.H != SH || (uintptr_t)d.w_hh_t % 16 != 0 || (uintptr_t)d.gates % 16 != 0) return false;
    }
    return true;
}

int solo_fwd_launch(const FwdGroup& g, int B, int T, hipStream_t s) {
    { const int e = persist_take_after(s); if (e) return e; }
    persist_count_launch();
    persist_record_start(s);
    gru_solo_fwd_kernel<<<g.n * B, ST, 0, s>>>(g, B, T);
    persist_record_end(s);
    return (int)hipGetLastError();
}

int solo_bwd_launch(const BwdGroup& g, int B, int T, int flags, hipStream_t s) {
    { const int e = persist_take_after(s); if (e) return e; }
    persist_count_launch();
    persist_record_start(s);
    gru_solo_bwd_kernel<<<g.n * B, ST, 0, s>>>(g, B, T, (flags & M3T_SCAN_WHH) ? 1 : 0);
    persist_record_end(s);
    return (int)hipGetLastError();
}

}  // namespace m3t_gru
