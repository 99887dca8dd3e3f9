// 16-bit-term GEMM, 128 x 256 tile ("x6w"): gemm_x6d.hip's software-pipelined kernel with a B tile twice as wide.
//
// Why (round 5, tools/gemm_pmc.sh on 9600 x 1536 x 1024 NT, fp16x3): the 128 x 128 kernel keeps the matrix pipe 42 % busy WHILE its
// workgroups run (SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CYCLES x SIMDs), but the launch as a whole reaches 30 % (247 of 833 TFLOP/s
// fp32-equivalent): 75 x 12 = 900 workgroups on 768 resident slots (three per CU) are one full round and a second one at 17 %
// occupancy.  Every input projection of the model (N = 3H = 1536, 768) has that shape.  With 128 x 256 tiles the same product is
// 450 workgroups for 512 slots (two per CU, 78 KiB of LDS each): ONE round, every CU busy to the end.  On the way the tile halves the
// A-operand staging per flop -- LDS traffic per MFMA falls by a quarter ((128 + 256) instead of 2 x (128 + 128) rows per 128 x 256 x 16 step:
// the 128-tile kernel needs the LDS pipe as long as the matrix pipe), the in-kernel operand split by the same quarter.
//
// Same arithmetic as gemm_x6d.hip (same exact split, same products smallest first, same k order, same split-K slabs): results are
// bit-identical to the 128-tile kernels'.  4 waves as 2 (M) x 2 (N), each 64 x 128 = 2 x 4 v_mfma_f32_32x32x16 tiles; 16-k stages,
// double-buffered LDS, one barrier per stage.  The planner (gemm.hip, plan_gemm) takes it where it fills the chip better.
#include "common.h"
#include <atomic>
#include <cstdlib>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int WM = 128, WN = 256, WKS = 16, WTH = 256;
constexpr int PLA = WM * 16 + 128;                 // one k-octet plane of A: 128 rows x 16 B (+128 B: the stage's two planes hit distinct banks)
constexpr int PLB = WN * 16 + 128;                 // ... of B: 256 rows
constexpr int planes_of(int NS) { return NS == 4 ? 2 : NS; }
constexpr int opera(int NS) { return planes_of(NS) * 2 * PLA; }
constexpr int operb(int NS) { return planes_of(NS) * 2 * PLB; }
constexpr int stageb(int NS) { return opera(NS) + operb(NS); }

struct X6WParams {
    const float* A; const float* B; float* C; const float* bias; float* ws;
    int M, N, K, lda, ldb, ldc;
    int act, accumulate, splits, kchunk;
    int seg_len, seg_stride, a_off, b_off;
    const unsigned long long* amax_a;   // NS = 4 (fp16x3): magnitude slots of A and B (common.h)
    const unsigned long long* amax_b;
};

// (the exact splits of gemm_x6d.hip)
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
    f32x2 hf;
    hf.x = __uint_as_float(o[0] << 16); hf.y = __uint_as_float(o[0] & 0xffff0000u);
    const f32x2 r1 = v - hf;
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    o[1] = __builtin_bit_cast(unsigned, m);
    f32x2 mf;
    mf.x = __uint_as_float(o[1] << 16); mf.y = __uint_as_float(o[1] & 0xffff0000u);
    const f32x2 r2 = r1 - mf;
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    o[2] = __builtin_bit_cast(unsigned, l);
}

// K-contiguous operand of NR x 64 rows: thread = (k-quad tid & 3, rows (tid >> 2) + 64 i); r[i] = 4 k of row i; PL = the plane's bytes
template <int NS, int NR, int PL>
__device__ __forceinline__ void kc_store_w(unsigned char* __restrict__ S, const f32x4 (&r)[NR], float scale) {
    const int tid = threadIdx.x;
    unsigned char* q = S + ((tid >> 1) & 1) * PL + (tid >> 2) * 16 + (tid & 1) * 8;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        unsigned lo[3], hi2[3];
        split3_pair<NS>((f32x2){r[i].x, r[i].y}, lo, scale);
        split3_pair<NS>((f32x2){r[i].z, r[i].w}, hi2, scale);
#pragma unroll
        for (int s = 0; s < planes_of(NS); ++s) *reinterpret_cast<u32x2*>(q + s * 2 * PL + i * 1024) = (u32x2){lo[s], hi2[s]};
    }
}
// row-contiguous operand, one 128-row half: wave w holds k-octet w >> 1 of rows 64 (w & 1) ..+63 of the half; lane = (k-pair g = lane >> 4,
// rows 4 (lane & 15) ..+3); r[e] = those 4 rows at k = 8 (w >> 1) + 2 g + e.  4 x 4 transpose across the four lanes 16 apart (gemm_x6d.hip)
template <int NS, int PL>
__device__ __forceinline__ void mc_store_w(unsigned char* __restrict__ S, const f32x4& r0, const f32x4& r1, float scale, int half) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned o[4][3];
    split3_pair<NS>((f32x2){r0.x, r1.x}, o[0], scale);
    split3_pair<NS>((f32x2){r0.y, r1.y}, o[1], scale);
    split3_pair<NS>((f32x2){r0.z, r1.z}, o[2], scale);
    split3_pair<NS>((f32x2){r0.w, r1.w}, o[3], scale);
    unsigned char* q = S + (w >> 1) * PL + (half * 128 + (w & 1) * 64 + (lane & 15) * 4 + (lane >> 4)) * 16;
#pragma unroll
    for (int s = 0; s < planes_of(NS); ++s) {
        const u32x2 p01 = __builtin_amdgcn_permlane16_swap(o[0][s], o[1][s], false, false);
        const u32x2 p23 = __builtin_amdgcn_permlane16_swap(o[2][s], o[3][s], false, false);
        const u32x2 c02 = __builtin_amdgcn_permlane32_swap(p01.x, p23.x, false, false);
        const u32x2 c13 = __builtin_amdgcn_permlane32_swap(p01.y, p23.y, false, false);
        *reinterpret_cast<u32x4*>(q + s * 2 * PL) = (u32x4){c02.x, c13.x, c02.y, c13.y};
    }
}

// BD (TA = 0, TB = 1, NS = 4): the B operand is a STAGED IMAGE of the weights (m3t_f16x3_image_b: [N / 64][K / 8][term][64 rows][16 B] -- one
// 1-KiB piece per (64-row block, k-octet, term)) and reaches LDS by LDS-DMA (global_load_lds_dwordx4: one piece per wave instruction, no
// registers, no conversion, no ds_write): wave w fetches row block w of the 256-row tile, four pieces per 16-k stage.  Why (NOTEBOOK R5.4b):
// of the 134 us of a 128-tile launch 28 are the register -> LDS store path of the split operands; the 256-row B tile is two thirds of them.
template <int TA, int TB, bool SEG, int NS, int BD = 0>
__global__ __launch_bounds__(WTH, 2) void sgemm_x6w_kernel(X6WParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];     // stage 0 | stage 1, each A | B
    constexpr int NP = planes_of(NS);
    constexpr int OPA = opera(NS), STG = stageb(NS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                  // 2 x 2 waves: 64 x 128 of the tile each
    const int l31 = lane & 31, hi = lane >> 5;

    const int tn_ = gridDim.x, nt_ = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * tn_ + blockIdx.x;
    const int xq = nt_ >> 3, xr = nt_ & 7, xcd = lin & 7, slot = lin >> 3;
    const int til = xcd * xq + min(xcd, xr) + slot;           // XCD-contiguous tile order (see gemm.hip)
    const int bm = (til / tn_) * WM, bn = (til % tn_) * WN;
    const int k_begin = blockIdx.z * p.kchunk;
    const int k_end = min(p.K, k_begin + p.kchunk);
    const int nst = (k_end - k_begin) / WKS;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float sc_a = 1.f, sc_b = 1.f, sc_ia = 1.f, sc_ib = 1.f;      // NS = 4: operand scales and their inverses (powers of two)
    if (NS == 4) {
        m3t_f16_scale((unsigned)*p.amax_a, sc_a, sc_ia);
        m3t_f16_scale((unsigned)*p.amax_b, sc_b, sc_ib);
    }

    // per-thread source pointers: A two float4 per stage, B four
    const float* pa[2]; const float* pb[4];
    size_t a_step, b_step;
    const int mck = 8 * (wave >> 1) + 2 * (lane >> 4);       // row-contiguous operands: this lane's first k of a stage
    const int mcr = (wave & 1) * 64 + (lane & 15) * 4;        // ... and its first row / column of a 128-wide half
    if (TA == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) pa[i] = p.A + (size_t)(bm + (tid >> 2) + 64 * i) * p.lda + k_begin + (tid & 3) * 4;
        a_step = WKS;
    } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) pa[e] = p.A + (size_t)(k_begin + mck + e) * p.lda + bm + mcr;
        a_step = (size_t)WKS * p.lda;
    }
    // BD: this wave's piece stream -- pieces ((bn / 64 + w) (K / 8) + k / 8 + o) * 2 + s, 1 KiB each, lane l takes bytes 16 l ..
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned char* bd_src = nullptr;
    if (BD) {
        bd_src = reinterpret_cast<const unsigned char*>(p.B) + ((size_t)(bn / 64 + wv) * (p.K / 8) + k_begin / 8) * 2048 + lane * 16;
        b_step = 0;
    } else if (TB == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pb[i] = p.B + (size_t)(bn + (tid >> 2) + 64 * i) * p.ldb + k_begin + (tid & 3) * 4;
        b_step = WKS;
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e) pb[2 * h + e] = p.B + (size_t)(k_begin + mck + e) * p.ldb + bn + 128 * h + mcr;
        b_step = (size_t)WKS * p.ldb;
    }
    int sq[2] = {0, 0}, sr[2] = {0, 0};                       // segmented K: (segment, offset) of this lane's two k
    if (SEG) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = k_begin + mck + e;
            sq[e] = k / p.seg_len; sr[e] = k % p.seg_len;
        }
        pa[0] = p.A + (size_t)p.a_off * p.lda + bm + mcr;
        pb[0] = p.B + (size_t)p.b_off * p.ldb + bn + mcr;
    }

    f32x4 ra0[2], rb0[4], ra1[2], rb1[4];                    // raw fp32 values of two stages in flight
    int loaded = 0;
    auto gload = [&](f32x4 (&ra)[2], f32x4 (&rb)[4]) {        // fetch the next stage; past the end: re-fetch the last one (unused)
        const bool adv = loaded + 1 < nst;
        if (SEG) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const size_t row = (size_t)sq[e] * p.seg_stride + sr[e];
                ra[e] = *reinterpret_cast<const f32x4*>(pa[0] + row * p.lda);
                rb[e] = *reinterpret_cast<const f32x4*>(pb[0] + row * p.ldb);
                rb[2 + e] = *reinterpret_cast<const f32x4*>(pb[0] + row * p.ldb + 128);
                int r = sr[e] + (adv ? WKS : 0), q = sq[e];
                if (r >= p.seg_len) { r -= p.seg_len; ++q; }   // seg_len >= 32 > WKS: at most one wrap
                sr[e] = r; sq[e] = q;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                ra[e] = *reinterpret_cast<const f32x4*>(pa[e]);
                pa[e] += adv ? a_step : 0;
            }
            if (!BD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rb[e] = *reinterpret_cast<const f32x4*>(pb[e]);
                    pb[e] += adv ? b_step : 0;
                }
            }
        }
        ++loaded;
    };
    // BD: the four pieces of the next stage of this wave's row block into `st` (planes term * 2 + octet, rows 64 w ..+63)
    int bd_left = nst;
    auto bdma = [&](unsigned char* st) {
        if (bd_left > 0) {
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(bd_src + (o * 2 + t2) * 1024),
                                                     (__attribute__((address_space(3))) void*)(st + OPA + (t2 * 2 + o) * PLB + wv * 1024), 16, 0, 0);
            bd_src += 4096;
            --bd_left;
        }
    };
    auto sstore = [&](unsigned char* st, const f32x4 (&ra)[2], const f32x4 (&rb)[4]) {
        if (TA == 0) kc_store_w<NS, 2, PLA>(st, ra, sc_a); else mc_store_w<NS, PLA>(st, ra[0], ra[1], sc_a, 0);
        if (BD) {}
        else if (TB == 1) kc_store_w<NS, 4, PLB>(st + OPA, rb, sc_b);
        else { mc_store_w<NS, PLB>(st + OPA, rb[0], rb[1], sc_b, 0); mc_store_w<NS, PLB>(st + OPA, rb[2], rb[3], sc_b, 1); }
    };

    // the products, smallest first (NS = 3: (a3,b1) (a2,b2) (a1,b3) (a2,b1) (a1,b2) (a1,b1); NS = 4: lo hi, hi lo, hi hi)
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    const int fro_a = hi * PLA + (wm * 64 + l31) * 16;
    const int fro_b = OPA + hi * PLB + (wn * 128 + l31) * 16;
    auto stage = [&](const unsigned char* cur, unsigned char* nxt, const f32x4 (&ua)[2], const f32x4 (&ub)[4]) {
        bf16x8 fa[NP][2], fb[NP][4];
#pragma unroll
        for (int s = 0; s < NP; ++s) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[s][i] = *reinterpret_cast<const bf16x8*>(cur + fro_a + s * 2 * PLA + i * 512);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[s][j] = *reinterpret_cast<const bf16x8*>(cur + fro_b + s * 2 * PLB + j * 512);
        }
        sstore(nxt, ua, ub);                                  // (after the last stage: a stage nobody reads)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x16 c = acc[i][j];
                if (NS == 4) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[1][i]), __builtin_bit_cast(f16x8, fb[0][j]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[0][i]), __builtin_bit_cast(f16x8, fb[1][j]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[0][i]), __builtin_bit_cast(f16x8, fb[0][j]), c, 0, 0, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < 6; ++q)
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q] < NP ? PA[q] : 0][i], fb[PB[q] < NP ? PB[q] : 0][j], c, 0, 0, 0);
                }
                acc[i][j] = c;
            }
    };

    unsigned char* buf0 = ldsb;
    unsigned char* buf1 = ldsb + STG;
    if (BD) {
        // order of the vector-memory queue per stage: the DMA pieces of stage t+1 FIRST, then A's two loads of stage t+2 -- the wait in front of the
        // barrier is vmcnt(2): the pieces have landed in LDS, A's prefetch stays in flight across the barrier
        if (nst > 0) {
            bdma(buf0);
            gload(ra0, rb0);
            sstore(buf0, ra0, rb0);
            __builtin_amdgcn_sched_barrier(0);
            bdma(buf1);
            __builtin_amdgcn_sched_barrier(0);
            gload(ra1, rb1);
        }
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int t = 0; t < nst; t += 2) {
            // stage t multiplies buf0; A(t+1) is in ra1, B(t+1) already in buf1 (pieces issued one stage ahead)
            __builtin_amdgcn_sched_barrier(0);
            gload(ra0, rb0);                                  // A of stage t+2
            stage(buf0, buf1, ra1, rb1);                      // multiply stage t, split + store A of stage t+1
            asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // pieces of t+1 landed; everyone is done reading buf0
            __builtin_amdgcn_sched_barrier(0);
            bdma(buf0);                                       // B of stage t+2 (needs buf0 free: after the barrier)
            __builtin_amdgcn_sched_barrier(0);
            gload(ra1, rb1);                                  // A of stage t+3
            stage(buf1, buf0, ra0, rb0);                      // multiply stage t+1, split + store A of stage t+2
            asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // pieces of t+2 landed; buf1 free
            __builtin_amdgcn_sched_barrier(0);
            bdma(buf1);                                       // B of stage t+3
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
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
    }

    const bool direct = p.splits == 1;
    float* dst = direct ? p.C : p.ws + (size_t)blockIdx.z * p.M * p.N;
    const int ldd = direct ? p.ldc : p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = bn + wn * 128 + j * 32 + l31;
            const float bv = (direct && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = bm + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                float v = acc[i][j][r];
                if (NS == 4) v = v * sc_ia * sc_ib;              // (exact: powers of two)
                float* q = dst + (size_t)row * ldd + col;
                if (direct) {
                    v += bv;
                    if (p.act == 1) v = fmaxf(v, 0.f);
                    if (p.accumulate) v += *q;
                }
                *q = v;
            }
        }
}

}  // namespace

// Contract of m3t_sgemm_x6d_launch, with N % 256 == 0 (M % 128 == 0, K and kchunk % 32 == 0, 16-B aligned operands, ld % 4 == 0,
// seg_len >= 32 when segmented); bf16_operands: 0 the six-product bf16 form, 3 fp16x3 (the two forms this tile is built for).
int m3t_sgemm_x6w_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                         float* C, int ldc, const float* bias, int act, int accumulate, int seg_len, int seg_stride,
                         int a_off, int b_off, float* ws, int splits, int kchunk, int bf16_operands,
                         const unsigned long long* amax_a, const unsigned long long* amax_b, hipStream_t s) {
    if (bf16_operands != 0 && bf16_operands != 3) return M3T_EINVAL;
    if (bf16_operands == 3 && (!amax_a || !amax_b)) return M3T_EINVAL;
    X6WParams p;
    p.amax_a = amax_a; p.amax_b = amax_b;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.ws = ws;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.act = act; p.accumulate = accumulate; p.splits = splits; p.kchunk = kchunk;
    p.seg_len = seg_len; p.seg_stride = seg_stride; p.a_off = a_off; p.b_off = b_off;
    dim3 grid(N / WN, M / WM, splits), block(WTH);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return M3T_EINVAL;
#define M3T_X6W_GO(TA_, TB_, SEG_, NS_)                                                                               \
    do {                                                                                                               \
        const size_t lds = 2 * (size_t)stageb(NS_);                                                                    \
        static std::atomic<unsigned> attr_set{0};           /* one bit per device (was one flag per process) */           \
        if (!(attr_set.load(std::memory_order_relaxed) & (1u << dev))) {                                               \
            hipError_t ea = hipFuncSetAttribute((const void*)sgemm_x6w_kernel<TA_, TB_, SEG_, NS_>,                    \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
            if (ea != hipSuccess) return (int)ea;                                                                      \
            attr_set.fetch_or(1u << dev, std::memory_order_relaxed);                                                   \
        }                                                                                                              \
        sgemm_x6w_kernel<TA_, TB_, SEG_, NS_><<<grid, block, lds, s>>>(p);                                             \
    } while (0)
#define M3T_X6W_DISPATCH(NS_)                                                                                        \
    do {                                                                                                               \
        if (seg_len > 0) M3T_X6W_GO(1, 0, true, NS_);                                                                  \
        else if (transA == 0 && transB == 1) M3T_X6W_GO(0, 1, false, NS_);                                             \
        else if (transA == 0 && transB == 0) M3T_X6W_GO(0, 0, false, NS_);                                             \
        else if (transA == 1 && transB == 0) M3T_X6W_GO(1, 0, false, NS_);                                             \
        else M3T_X6W_GO(1, 1, false, NS_);                                                                             \
    } while (0)
    if (bf16_operands == 3) M3T_X6W_DISPATCH(4);
    else M3T_X6W_DISPATCH(3);
#undef M3T_X6W_DISPATCH
#undef M3T_X6W_GO
    return (int)hipGetLastError();
}

// NT product with the B operand as a staged image (BD kernels; m3t_sgemm_bimg): fp16x3, M % 128 == 0, N % 256 == 0, K % 32 == 0, one K pass
// or split-K slabs as above; B_img from m3t_f16x3_image_b under amax_b.
int m3t_sgemm_x6w_bimg_launch(int M, int N, int K, const float* A, int lda, const float* B_img, float* C, int ldc, const float* bias, int act,
                              int accumulate, float* ws, int splits, int kchunk, const unsigned long long* amax_a,
                              const unsigned long long* amax_b, hipStream_t s) {
    if (!amax_a || !amax_b) return M3T_EINVAL;
    X6WParams p;
    p.amax_a = amax_a; p.amax_b = amax_b;
    p.A = A; p.B = B_img; p.C = C; p.bias = bias; p.ws = ws;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = 0; p.ldc = ldc;
    p.act = act; p.accumulate = accumulate; p.splits = splits; p.kchunk = kchunk;
    p.seg_len = p.seg_stride = p.a_off = p.b_off = 0;
    dim3 grid(N / WN, M / WM, splits), block(WTH);
    const size_t lds = 2 * (size_t)stageb(4);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return M3T_EINVAL;
    static std::atomic<unsigned> attr_set{0};
    if (!(attr_set.load(std::memory_order_relaxed) & (1u << dev))) {
        hipError_t ea = hipFuncSetAttribute((const void*)sgemm_x6w_kernel<0, 1, false, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return (int)ea;
        attr_set.fetch_or(1u << dev, std::memory_order_relaxed);
    }
    sgemm_x6w_kernel<0, 1, false, 4, 1><<<grid, block, lds, s>>>(p);
    return (int)hipGetLastError();
}
