// bf16x6 GEMM, 128 x 128 tile, 16-k stages through a double-buffered LDS image, software-pipelined INSIDE each wave ("x6d").
// gemm_x6.hip (128 x 128 x 32, single LDS buffer, two barriers per K-step) reaches 55 % of the MFMA pipe in steady state: a workgroup
// multiplies, waits, splits + stores, waits, and those phases only overlap across co-resident workgroups -- measured time per tile
// is close to the SUM of its MFMA, LDS and L2-port times.  Here the loop body is straight-line code: the fragments of stage t are
// read, and the split + LDS store of stage t+1 (raw values fetched one iteration earlier) and the global loads of stage t+2 are left
// to the scheduler to place in the shadow of the 24 MFMAs of stage t (a dependent MFMA leaves ~7 issue slots; the split needs ~4 VALU /
// LDS instructions per MFMA); ONE barrier per stage, no vmcnt(0) in the loop.  Same exact 3-term split, same six products smallest
// first, same XCD-contiguous tile order, same deterministic split-K slabs, same (row, k-octet) 16-B LDS records (one ds_read_b128
// per fragment) as gemm_x6.hip, whose loaders this file shares in 256-thread form: results are bit-identical to
// gemm_x6.hip's.  130 VGPRs, 51 KiB of LDS: three workgroups per CU, and two fit beside a persistent scan workgroup.
// Measured (tools/x6d_bench.py, tools/gemm_bench.py): 9600 x 1536 x 1024 NT 226 -> 177 us (171 TFLOP/s algorithmic), 2048^3 118 -> 95 us.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BKS = 16, NTH = 256;
constexpr int PLANEB = BM * 16 + 128;              // one k-octet plane: 128 rows x 16 B; +128 B: 2 planes x 8 rows hit 64 distinct banks
constexpr int SPLITB = 2 * PLANEB;
constexpr int OPERB = 3 * SPLITB;
constexpr int STAGEB = 2 * OPERB;                  // 26 112 B

struct X6DParams {
    const float* A; const float* B; float* C; const float* bias; float* ws;
    int M, N, K, lda, ldb, ldc;
    int act, accumulate, splits, kchunk;
    int seg_len, seg_stride, a_off, b_off;
    // implicit-GEMM 1-D convolution (CONV kernels, m3t_conv_x6d_launch; semantics of the CONV form of gemm_x6.hip)
    int cv_T, cv_C, cv_K, cv_dil, cv_lead, cv_anti;
    size_t cv_btap;
    const float* cv_mask; const float* cv_res; float* cv_pre;
    M3TDrop cv_drop;
    const unsigned long long* amax_a;   // NS = 4 (fp16x3): magnitude slots of A and B (common.h)
    const unsigned long long* amax_b;
};

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int planes_of(int NS) { return NS == 4 ? 2 : NS; }      // NS = 4: fp16x3, two fp16 terms of the scaled operand (gemm_x6.hip)
template <int NS>
__device__ __forceinline__ void split3_pair(f32x2 v, unsigned (&o)[3], float scale = 1.f) {
    if (NS == 4) {
        const f32x2 vs = v * scale;
        const f16x2 h = __builtin_convertvector(vs, f16x2);
        o[0] = __builtin_bit_cast(unsigned, h);
        const f32x2 r1 = vs - __builtin_convertvector(h, f32x2);
        const f16x2 l = __builtin_convertvector(r1, f16x2);
        o[1] = __builtin_bit_cast(unsigned, l);
        return;
    }
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    o[0] = __builtin_bit_cast(unsigned, h);
    if (NS == 1) return;
    f32x2 hf;
    hf.x = __uint_as_float(o[0] << 16); hf.y = __uint_as_float(o[0] & 0xffff0000u);
    const f32x2 r1 = v - hf;
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    o[1] = __builtin_bit_cast(unsigned, m);
    if (NS == 2) return;
    f32x2 mf;
    mf.x = __uint_as_float(o[1] << 16); mf.y = __uint_as_float(o[1] & 0xffff0000u);
    const f32x2 r2 = r1 - mf;
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    o[2] = __builtin_bit_cast(unsigned, l);
}

// K-contiguous operand: thread = (k-quad tid & 3, rows (tid >> 2) + 64 i); r[i] = 4 k of row i
template <int NS>
__device__ __forceinline__ void kc_store_c(unsigned char* __restrict__ S, const f32x4 (&r)[2], float scale) {
    const int tid = threadIdx.x;
    unsigned char* q = S + ((tid >> 1) & 1) * PLANEB + (tid >> 2) * 16 + (tid & 1) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        unsigned lo[3], hi2[3];
        split3_pair<NS>((f32x2){r[i].x, r[i].y}, lo, scale);
        split3_pair<NS>((f32x2){r[i].z, r[i].w}, hi2, scale);
#pragma unroll
        for (int s = 0; s < planes_of(NS); ++s) *reinterpret_cast<u32x2*>(q + s * SPLITB + i * 1024) = (u32x2){lo[s], hi2[s]};
    }
}
// row-contiguous operand: wave w holds k-octet w >> 1 of rows 64 (w & 1) ..+63; lane = (k-pair g = lane >> 4, rows
// 4 (lane & 15) ..+3); r[e] = those 4 rows at k = 8 (w >> 1) + 2 g + e.  The four lanes 16 apart hold the four k-pairs of
// the same 4 rows: a 4 x 4 transpose across them (v_permlane16_swap, then v_permlane32_swap: 4 per split) leaves lane
// g with the complete 16-B record of row g -- a wave stores 1 KiB contiguous per split.
template <int NS>
__device__ __forceinline__ void mc_store_c(unsigned char* __restrict__ S, const f32x4 (&r)[2], float scale) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned o[4][3];
    split3_pair<NS>((f32x2){r[0].x, r[1].x}, o[0], scale);
    split3_pair<NS>((f32x2){r[0].y, r[1].y}, o[1], scale);
    split3_pair<NS>((f32x2){r[0].z, r[1].z}, o[2], scale);
    split3_pair<NS>((f32x2){r[0].w, r[1].w}, o[3], scale);
    unsigned char* q = S + (w >> 1) * PLANEB + ((w & 1) * 64 + (lane & 15) * 4 + (lane >> 4)) * 16;
#pragma unroll
    for (int s = 0; s < planes_of(NS); ++s) {
        // X[g][i] = o[i][s] in lane group g.  permlane16_swap(a, b): a.group1 <-> b.group0, a.group3 <-> b.group2;
        // permlane32_swap(a, b): a.groups{2,3} <-> b.groups{0,1}.  After both: (c0, c1, c2, c3) in group g = X[0..3][g].
        const u32x2 p01 = __builtin_amdgcn_permlane16_swap(o[0][s], o[1][s], false, false);
        const u32x2 p23 = __builtin_amdgcn_permlane16_swap(o[2][s], o[3][s], false, false);
        const u32x2 c02 = __builtin_amdgcn_permlane32_swap(p01.x, p23.x, false, false);
        const u32x2 c13 = __builtin_amdgcn_permlane32_swap(p01.y, p23.y, false, false);
        *reinterpret_cast<u32x4*>(q + s * SPLITB) = (u32x4){c02.x, c13.x, c02.y, c13.y};
    }
}

// CONV (TA == 0): k = (tap j, channel c), A[m][k] = x[m + off_j][c] with rows whose source frame falls outside the clip read as zero
// (a 16-deep stage lies inside one tap: C % 32 == 0); TB == 1: B = one [Co][Ci] plane per tap; TB == 0 (data gradient): the
// row-major [K*Ci][Co] matrix.  Epilogue: bias, pre-activation copy, ReLU x dropout mask, residual add + ReLU, as gemm_x6.hip's.
template <int TA, int TB, bool SEG, int NS, bool CONV = false>
__global__ __launch_bounds__(NTH, 3) void sgemm_x6d_kernel(X6DParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];     // stage 0 | stage 1, each A | B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                  // 2 x 2 waves: 64 x 64 of the tile each
    const int l31 = lane & 31, hi = lane >> 5;

    const int tn_ = gridDim.x, nt_ = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * tn_ + blockIdx.x;
    const int xq = nt_ >> 3, xr = nt_ & 7, xcd = lin & 7, slot = lin >> 3;
    const int til = xcd * xq + min(xcd, xr) + slot;           // XCD-contiguous tile order (see gemm.hip)
    const int bm = (til / tn_) * BM, bn = (til % tn_) * BN;
    const int k_begin = blockIdx.z * p.kchunk;
    const int k_end = min(p.K, k_begin + p.kchunk);
    const int nst = (k_end - k_begin) / BKS;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float sc_a = 1.f, sc_b = 1.f, sc_ia = 1.f, sc_ib = 1.f;      // NS = 4: operand scales and their inverses (powers of two)
    if (NS == 4) {
        m3t_f16_scale((unsigned)*p.amax_a, sc_a, sc_ia);
        m3t_f16_scale((unsigned)*p.amax_b, sc_b, sc_ib);
    }

    // per-thread source pointers (two float4 per operand per stage); M, N are multiples of 128: no edges (the clamps are no-ops)
    const float* pa[2]; const float* pb[2];
    size_t a_step, b_step;
    const int mck = 8 * (wave >> 1) + 2 * (lane >> 4);       // row-contiguous operands: this lane's first k of a stage
    const int mcr = (wave & 1) * 64 + (lane & 15) * 4;        // ... and its first row / column of the tile
    if (TA == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) pa[i] = p.A + (size_t)min(bm + (tid >> 2) + 64 * i, p.M - 1) * p.lda + k_begin + (tid & 3) * 4;
        a_step = BKS;
    } else {
        const int c0 = min(bm + mcr, p.M - 4);
#pragma unroll
        for (int e = 0; e < 2; ++e) pa[e] = p.A + (size_t)(k_begin + mck + e) * p.lda + c0;
        a_step = (size_t)BKS * p.lda;
    }
    if (TB == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) pb[i] = p.B + (size_t)min(bn + (tid >> 2) + 64 * i, p.N - 1) * p.ldb + k_begin + (tid & 3) * 4;
        b_step = BKS;
    } else {
        const int c0 = min(bn + mcr, p.N - 4);
#pragma unroll
        for (int e = 0; e < 2; ++e) pb[e] = p.B + (size_t)(k_begin + mck + e) * p.ldb + c0;
        b_step = (size_t)BKS * p.ldb;
    }
    int sq[2] = {0, 0}, sr[2] = {0, 0};                       // segmented K: (segment, offset) of this lane's two k
    if (SEG) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = k_begin + mck + e;
            sq[e] = k / p.seg_len; sr[e] = k % p.seg_len;
        }
        const int ca = min(bm + mcr, p.M - 4), cb = min(bn + mcr, p.N - 4);
        pa[0] = p.A + (size_t)p.a_off * p.lda + ca;
        pb[0] = p.B + (size_t)p.b_off * p.ldb + cb;
    }

    int cv_t[2] = {0, 0};                                     // CONV: frame index of this thread's two A rows
    if (CONV) {
#pragma unroll
        for (int i = 0; i < 2; ++i) cv_t[i] = (bm + (tid >> 2) + 64 * i) % p.cv_T;
    }
    f32x4 ra0[2], rb0[2], ra1[2], rb1[2];                    // raw fp32 values of two stages in flight
    int loaded = 0;                                           // stages fetched so far
    auto gload = [&](f32x4 (&ra)[2], f32x4 (&rb)[2]) {        // fetch the next stage; past the end: re-fetch the last one (unused)
        const bool adv = loaded + 1 < nst;
        if (CONV) {
            const int k0 = k_begin + BKS * min(loaded, nst - 1);
            const int j = k0 / p.cv_C, kc = k0 - j * p.cv_C;
            const int sft = (p.cv_K - 1 - j) * p.cv_dil;
            const int off = p.cv_anti ? sft - p.cv_lead : p.cv_lead - sft;
            const float* qa = p.A + ((ptrdiff_t)(bm + (tid >> 2)) + off) * (ptrdiff_t)p.lda + kc + (tid & 3) * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok = (unsigned)(cv_t[i] + off) < (unsigned)p.cv_T;
                ra[i] = ok ? *reinterpret_cast<const f32x4*>(qa + (ptrdiff_t)i * 64 * p.lda) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (TB == 1) {
                const float* qb = p.B + (size_t)j * p.cv_btap + (size_t)(bn + (tid >> 2)) * p.ldb + kc + (tid & 3) * 4;
#pragma unroll
                for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const f32x4*>(qb + (size_t)i * 64 * p.ldb);
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    rb[e] = *reinterpret_cast<const f32x4*>(pb[e]);
                    pb[e] += adv ? b_step : 0;
                }
            }
        } else if (SEG) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const size_t row = (size_t)sq[e] * p.seg_stride + sr[e];
                ra[e] = *reinterpret_cast<const f32x4*>(pa[0] + row * p.lda);
                rb[e] = *reinterpret_cast<const f32x4*>(pb[0] + row * p.ldb);
                int r = sr[e] + (adv ? BKS : 0), q = sq[e];
                if (r >= p.seg_len) { r -= p.seg_len; ++q; }   // seg_len >= 32 > BKS: at most one wrap
                sr[e] = r; sq[e] = q;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                ra[e] = *reinterpret_cast<const f32x4*>(pa[e]);
                rb[e] = *reinterpret_cast<const f32x4*>(pb[e]);
                pa[e] += adv ? a_step : 0;
                pb[e] += adv ? b_step : 0;
            }
        }
        ++loaded;
    };
    auto sstore = [&](unsigned char* st, const f32x4 (&ra)[2], const f32x4 (&rb)[2]) {
        if (TA == 0) kc_store_c<NS>(st, ra, sc_a); else mc_store_c<NS>(st, ra, sc_a);
        if (TB == 1) kc_store_c<NS>(st + OPERB, rb, sc_b); else mc_store_c<NS>(st + OPERB, rb, sc_b);
    };

    // the products, smallest first.  NS = 3: (a3,b1) (a2,b2) (a1,b3) (a2,b1) (a1,b2) (a1,b1); NS = 2 ("high" mode): the last four
    // entries read (a2,b2) (a2,b1) (a1,b2) (a1,b1); NS = 1: (a1,b1)
    constexpr int NSB = NS == 4 ? 2 : NS;             // (the bf16 product table; unused when NS = 4)
    constexpr int PA[6] = {NSB - 1, NSB / 2, NSB == 2 ? 1 : 0, NSB == 2 ? 1 : NSB / 2, 0, 0};
    constexpr int PB[6] = {0, NSB / 2, NSB == 2 ? 1 : NSB - 1, 0, NSB == 2 ? 1 : NSB / 2, 0};
    const int fro_a = hi * PLANEB + (wm * 64 + l31) * 16;
    const int fro_b = OPERB + hi * PLANEB + (wn * 64 + l31) * 16;
    // one stage, straight-line: the fragments of `cur`, the split + store of the held stage into `nxt`, the 24 MFMAs -- no
    // scheduling fence between them: the split's VALU / LDS instructions go into the shadow of the dependent MFMA chains
    auto stage = [&](const unsigned char* cur, unsigned char* nxt, const f32x4 (&ua)[2], const f32x4 (&ub)[2]) {
        constexpr int NP = planes_of(NS);
        bf16x8 fa[NP][2], fb[NP][2];
#pragma unroll
        for (int s = 0; s < NP; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[s][i] = *reinterpret_cast<const bf16x8*>(cur + fro_a + s * SPLITB + i * 512);
                fb[s][i] = *reinterpret_cast<const bf16x8*>(cur + fro_b + s * SPLITB + i * 512);
            }
        sstore(nxt, ua, ub);                                  // (after the last stage: a stage nobody reads)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 c = acc[i][j];
                if (NS == 4) {                 // fp16x3: lo hi, hi lo, hi hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[1][i]), __builtin_bit_cast(f16x8, fb[0][j]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[0][i]), __builtin_bit_cast(f16x8, fb[1][j]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[0][i]), __builtin_bit_cast(f16x8, fb[0][j]), c, 0, 0, 0);
                } else {
#pragma unroll
                    for (int q = (NS == 3 ? 0 : (NS == 2 ? 2 : 5)); q < 6; ++q)
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]][i], fb[PB[q]][j], c, 0, 0, 0);
                }
                acc[i][j] = c;
            }
    };

    unsigned char* buf0 = ldsb;
    unsigned char* buf1 = ldsb + STAGEB;
    if (nst > 0) {
        gload(ra0, rb0);
        sstore(buf0, ra0, rb0);
        gload(ra1, rb1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int t = 0; t < nst; t += 2) {                        // nst is even (kchunk % 32 == 0)
        gload(ra0, rb0);                                      // stage t+2 (past the end: the last stage again, unused)
        stage(buf0, buf1, ra1, rb1);                          // multiply stage t, split + store stage t+1
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (the global loads stay in flight across it)
        gload(ra1, rb1);                                      // stage t+3
        stage(buf1, buf0, ra0, rb0);                          // multiply stage t+1, split + store stage t+2
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    const bool direct = p.splits == 1;
    float* dst = direct ? p.C : p.ws + (size_t)blockIdx.z * p.M * p.N;
    const int ldd = direct ? p.ldc : p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = bn + wn * 64 + j * 32 + l31;
            const float bv = (direct && p.bias) ? p.bias[col] : 0.f;
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = bm + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (CONV && p.cv_drop.on && (r & 3) == 0) m3t_drop_mask4(p.cv_drop, (uint32_t)row >> 2, (uint32_t)col, dm);
                float v = acc[i][j][r];
                if (NS == 4) v = v * sc_ia * sc_ib;              // (exact: powers of two)
                float* q = dst + (size_t)row * ldd + col;
                if (CONV) {
                    const size_t o = (size_t)row * ldd + col;
                    v += bv;
                    if (p.cv_pre) p.cv_pre[o] = v;
                    const float mk = p.cv_drop.on ? dm[r & 3] : (p.cv_mask ? p.cv_mask[o] : 1.f);
                    if (p.act == 1) v = fmaxf(v, 0.f) * mk;
                    else if (p.act == 2) v = fmaxf(fmaxf(v, 0.f) * mk + p.cv_res[o], 0.f);
                    else if (p.cv_res) v += p.cv_res[o];
                } else if (direct) {
                    v += bv;
                    if (p.act == 1) v = fmaxf(v, 0.f);
                    if (p.accumulate) v += *q;
                }
                *q = v;
            }
        }
}

}  // namespace

// Same contract as m3t_sgemm_x6_launch (M, N multiples of 128, K and kchunk multiples of 32, 16-B aligned operands,
// ld % 4 == 0, seg_len >= 32 when segmented); bf16_operands: 0 fp32-accurate (six products), 1 bf16 mode, 2 "high" (four products).
int m3t_sgemm_x6d_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                         float* C, int ldc, const float* bias, int act, int accumulate, int seg_len, int seg_stride,
                         int a_off, int b_off, float* ws, int splits, int kchunk, int bf16_operands,
                         const unsigned long long* amax_a, const unsigned long long* amax_b, hipStream_t s) {
    X6DParams p;
    p.amax_a = amax_a; p.amax_b = amax_b;
    if (bf16_operands == 3 && (!amax_a || !amax_b)) return M3T_EINVAL;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.ws = ws;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.act = act; p.accumulate = accumulate; p.splits = splits; p.kchunk = kchunk;
    p.seg_len = seg_len; p.seg_stride = seg_stride; p.a_off = a_off; p.b_off = b_off;
    p.cv_T = p.cv_C = p.cv_K = p.cv_dil = p.cv_lead = p.cv_anti = 0; p.cv_btap = 0; p.cv_mask = p.cv_res = nullptr; p.cv_pre = nullptr;
    p.cv_drop = m3t_make_drop(0.f, 0ull);
    dim3 grid(N / BN, M / BM, splits), block(NTH);
    const size_t lds = 2 * (size_t)STAGEB;
#define M3T_X6D_GO(TA_, TB_, SEG_, NS_)                                                                               \
    do {                                                                                                               \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            hipError_t ea = hipFuncSetAttribute((const void*)sgemm_x6d_kernel<TA_, TB_, SEG_, NS_>,                    \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
            if (ea != hipSuccess) return (int)ea;                                                                      \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        sgemm_x6d_kernel<TA_, TB_, SEG_, NS_><<<grid, block, lds, s>>>(p);                                             \
    } while (0)
#define M3T_X6D_DISPATCH(NS_)                                                                                        \
    do {                                                                                                               \
        if (seg_len > 0) M3T_X6D_GO(1, 0, true, NS_);                                                                  \
        else if (transA == 0 && transB == 1) M3T_X6D_GO(0, 1, false, NS_);                                             \
        else if (transA == 0 && transB == 0) M3T_X6D_GO(0, 0, false, NS_);                                             \
        else if (transA == 1 && transB == 0) M3T_X6D_GO(1, 0, false, NS_);                                             \
        else M3T_X6D_GO(1, 1, false, NS_);                                                                             \
    } while (0)
    if (bf16_operands == 1) M3T_X6D_DISPATCH(1);
    else if (bf16_operands == 2) M3T_X6D_DISPATCH(2);
    else if (bf16_operands == 3) M3T_X6D_DISPATCH(4);       // fp16x3
    else M3T_X6D_DISPATCH(3);
#undef M3T_X6D_DISPATCH
#undef M3T_X6D_GO
    return (int)hipGetLastError();
}


// Dilated 1-D convolution on the software-pipelined kernel (contract of m3t_conv_x6_launch: (B*T) % 128 == 0, Co % 128 == 0,
// Ci % 32 == 0, 16-B aligned operands; anti = 0: w_t is [K][Co][Ci], anti = 1 (data gradient): [K][Ci][Co] read as [K*Ci][Co]).
int m3t_conv_x6d_launch(const float* x, const float* w_t, const float* bias, const float* res, const float* mask, float* y,
                        float* pre, int B, int T, int Ci, int Co, int K, int dil, int lead, int act, int anti, int bf16_operands,
                        M3TDrop drop, hipStream_t s) {
    X6DParams p;
    p.amax_a = p.amax_b = nullptr;
    p.A = x; p.B = w_t; p.C = y; p.bias = bias; p.ws = nullptr;
    p.M = B * T; p.N = Co; p.K = K * Ci; p.lda = Ci; p.ldb = anti ? Co : Ci; p.ldc = Co;
    p.act = act; p.accumulate = 0; p.splits = 1; p.kchunk = K * Ci;
    p.seg_len = p.seg_stride = p.a_off = p.b_off = 0;
    p.cv_T = T; p.cv_C = Ci; p.cv_K = K; p.cv_dil = dil; p.cv_lead = lead; p.cv_anti = anti;
    p.cv_btap = (size_t)Co * Ci; p.cv_mask = mask; p.cv_res = res; p.cv_pre = pre; p.cv_drop = drop;
    dim3 grid(Co / BN, p.M / BM, 1), block(NTH);
    const size_t lds = 2 * (size_t)STAGEB;
#define M3T_CONVD_GO(TB_, NS_)                                                                                            \
    do {                                                                                                                   \
        static bool attr_set = false;                                                                                      \
        if (!attr_set) {                                                                                                   \
            hipError_t ea = hipFuncSetAttribute((const void*)sgemm_x6d_kernel<0, TB_, false, NS_, true>,                   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
            if (ea != hipSuccess) return (int)ea;                                                                          \
            attr_set = true;                                                                                               \
        }                                                                                                                  \
        sgemm_x6d_kernel<0, TB_, false, NS_, true><<<grid, block, lds, s>>>(p);                                            \
    } while (0)
    if (anti) {
        if (bf16_operands == 1) M3T_CONVD_GO(0, 1); else if (bf16_operands == 2) M3T_CONVD_GO(0, 2); else M3T_CONVD_GO(0, 3);
    } else {
        if (bf16_operands == 1) M3T_CONVD_GO(1, 1); else if (bf16_operands == 2) M3T_CONVD_GO(1, 2); else M3T_CONVD_GO(1, 3);
    }
#undef M3T_CONVD_GO
    return (int)hipGetLastError();
}
