// fp32-accurate GEMM on the bf16 matrix pipe ("bf16x6").
// gfx950 has no reduced-precision fast path for fp32 inputs (no xf32/TF32): v_mfma_f32_*_f32 runs at the vector
// rate, 1/16 of the bf16 MFMA rate.  Each fp32 operand is therefore split EXACTLY into three bf16 terms
// (a = a1 + a2 + a3, 24 significant bits) while its tile is written to LDS, and a product is the six bf16 MFMAs
// whose weight is >= 2^-16 (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1; the dropped terms are <= 2^-24 relative), all
// accumulated in fp32, smallest first.  Error vs an fp64 reference: ~1e-6 absolute on O(1) outputs at K = 1024 --
// the same as a plain fp32 fmaf chain (tools/x6_accuracy.py); cost: 6 x 16-deep bf16 MFMAs instead of 8 x 2-deep
// fp32 ones per 16 k = 2.67x the fp32 MFMA rate (~420 TFLOP/s ceiling), deterministic.
//
// Tile: 128 x 128 x 32, 4 waves (2 x 2), each wave 64 x 64 = 2 x 2 v_mfma_f32_32x32x16_bf16 tiles.
// LDS (single buffer, 49.5 KiB -> 3 workgroups/CU): per operand and split, [k-octet 4][128 rows][8 k] bf16: one
// 16-B record per (row, k-octet), rows contiguous inside a plane, planes padded by 64 B.  A fragment (row = lane&31,
// k = 8*(lane>>5)..+7) is ONE conflict-free ds_read_b128 straight into the MFMA operand registers (two 8-B reads per
// fragment made hipcc pair the reads ACROSS fragments into ds_read2_b64 and reassemble them with 72 v_mov per K-step).
// K-contiguous operands store 8 B per thread (4 rows x 8 k-quads per half-wave = all 64 banks once); row-contiguous
// operands hold (4 rows, one k-quad) per thread, and the lane 32 away holds the other k-quad of the same octet: two
// v_permlane32_swap per split and dword pair leave every lane with two complete 16-B records (32 contiguous bytes).
// Global loads of tile t+1 stay in flight in registers while tile t is multiplied.
#include "common.h"
#include <cstdlib>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int XM = 128, XN = 128, XK = 32;
constexpr int PLANE = XM * 16 + 64;                // one k-octet plane: 128 rows x 16 B (+64 B: conflict-free K-contiguous stores)
constexpr int SPLIT_BYTES = 4 * PLANE;             // one split of one operand: 4 k-octets
constexpr int OPER_BYTES = 3 * SPLIT_BYTES;        // 24.75 KiB

struct X6Params {
    const float* A; const float* B; float* C; const float* bias; float* ws;
    int M, N, K, lda, ldb, ldc;
    int act, accumulate, splits, kchunk;
    int seg_len, seg_stride, a_off, b_off;
    // implicit-GEMM 1-D convolution (CONV kernels, m3t_conv_x6_launch): A = x [B*T][C] read through the taps' row shifts
    int cv_T, cv_C, cv_K, cv_dil, cv_lead, cv_anti;
    size_t cv_btap;                      // TB == 1: elements between the [N][C] weight planes of consecutive taps
    const float* cv_mask; const float* cv_res; float* cv_pre;
    M3TDrop cv_drop;                     // in-kernel dropout mask instead of cv_mask
    unsigned long long* cv_amax;         // CONV: magnitude slot raised to max |y| by the epilogue (m3t_amax_out), or nullptr
    const unsigned long long* amax_a;    // NS = 4 (fp16x3): low words = the bits of max |A|, max |B| (magnitude slots, common.h)
    const unsigned long long* amax_b;
    // MW kernels (m3t_sgemm_window): row m of A (TA == 0) and of C lives in storage row (m / mw_len) * mw_stride + m % mw_len + mw_off --
    // a TIME WINDOW [mw_off, mw_off + mw_len) of every clip of a [B, T, C] tensor (mw_stride = T)
    int mw_len, mw_stride, mw_off;
    // C3 kernels (m3t_conv3d_taps): A[m][(tap, c)] = src[(n, t + bt + sg kt, h + bh + sg kh, w + bw + sg kw)][c] over channels-last grids --
    // row m = (n, t, h, w) of the DESTINATION grid c3_T x c3_H x c3_W, source rows on the grid c3_To x c3_Ho x c3_Wo (zero outside it)
    int c3_T, c3_H, c3_W, c3_To, c3_Ho, c3_Wo, c3_kt, c3_kh, c3_kw, c3_bt, c3_bh, c3_bw, c3_sg, c3_C;
    int c3_st, c3_sh, c3_sw;             // source coordinate = destination coordinate * stride + base + sign * tap (1 for the data gradient's walk)
    int tr_S;                            // C3 = 1, 3, one K pass: > 0 -- C is written as channel planes [M / tr_S][N][tr_S] (row m = sample m / tr_S, position m % tr_S)
};
// MW kernels run up to M3T_WINDOW_BATCH problems of one shape in one launch (blockIdx.z = problem): the pieces of one progress mark
// (m3t_sgemm_window_batch) -- every (stack, direction) pair that reads the same window -- fill the CUs a scan leaves free as ONE grid
struct X6Batch {
    const float* A[M3T_WINDOW_BATCH]; const float* B[M3T_WINDOW_BATCH]; float* C[M3T_WINDOW_BATCH]; const float* bias[M3T_WINDOW_BATCH];
    const unsigned long long* amax_a[M3T_WINDOW_BATCH]; const unsigned long long* amax_b[M3T_WINDOW_BATCH];
    int accumulate[M3T_WINDOW_BATCH];
};

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// The same exact split on a PAIR of values with packed instructions: one v_cvt_pk_bf16_f32 rounds both (nearest even),
// the two bf16 are widened back with a shift / a mask, one v_pk_add_f32 forms both remainders.  9 VALU instructions per
// pair for all three terms, against ~17 per ELEMENT when hipcc scalarises split3 -- the splitting, not the MFMAs, was
// what bounded this kernel.  o[s] = the pair's term s, packed (first value in the low half).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// NS = 4, "fp16x3": the operand, scaled by a power of two s that puts its largest magnitude into [2^14, 2^15), is the sum of TWO fp16
// terms (hi = rn(s x), lo = rn(s x - hi): 22 significant bits wherever lo is a normal number, absolute error <= 2^-25 / s below that);
// a product is the three fp16 MFMAs hi hi + hi lo + lo hi (exact in fp32; the dropped lo lo is <= 2^-22 relative), the scales leave
// in the epilogue.  6 VALU instructions per pair (v_pk_mul, v_cvt_pk_f16_f32, 2 v_cvt_f32_f16, v_pk_fma, v_cvt_pk_f16_f32).
constexpr int planes_of(int NS) { return NS == 4 ? 2 : NS; }
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

// ---- K-contiguous operand ([rows][K]): 8 lanes cover one row's 32 k (a full 128-B line); thread = (k-quad c =
// tid&7, rows (tid>>3) + 32 i): every wave load instruction fetches 8 whole lines
template <int NR = 4>
__device__ __forceinline__ void kc_load(const float* __restrict__ p, size_t ld, float4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = *reinterpret_cast<const float4*>(p + (size_t)i * 32 * ld);
}
// PRE: the operand arrives PRE-SPLIT (m3t_f16x3_split, "P4" image: the 16 bytes of four consecutive k hold {hi k0|k1, hi k2|k3, lo k0|k1, lo k2|k3},
// two fp16 terms of the scaled values): the store is a copy -- no conversion, no arithmetic in the loop
template <int NS, int NR = 4, bool PRE = false>
__device__ __forceinline__ void kc_store(unsigned char* __restrict__ S, const float4 (&r)[4], float scale = 1.f) {
    const int tid = threadIdx.x & 255;
    const int c = tid & 7, r0 = tid >> 3;            // k-quad c of the 32-k tile: octet c >> 1, half c & 1
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        unsigned lo[3], hi2[3];                      // k pairs (0,1) and (2,3) of this row
        if (PRE) {
            lo[0] = __float_as_uint(r[i].x); hi2[0] = __float_as_uint(r[i].y); lo[1] = __float_as_uint(r[i].z); hi2[1] = __float_as_uint(r[i].w);
        } else {
            split3_pair<NS>((f32x2){r[i].x, r[i].y}, lo, scale);
            split3_pair<NS>((f32x2){r[i].z, r[i].w}, hi2, scale);
        }
#pragma unroll
        for (int s = 0; s < planes_of(NS); ++s)
            *reinterpret_cast<u32x2*>(S + s * SPLIT_BYTES + (c >> 1) * PLANE + (r0 + 32 * i) * 16 + (c & 1) * 8) = (u32x2){lo[s], hi2[s]};
    }
}
// ---- row-contiguous operand ([K][rows]): thread = (k-quad = tid>>5 (4 k), 4 rows at (tid&31)*4): a 4x4 block.
// Lanes l and l+32 of a wave hold the two k-quads of ONE octet (octet = wave) for the same rows; after the swaps the
// lower lane owns the complete records of rows 0, 1 and the upper lane those of rows 2, 3.
// PRE (NS = 4): the operand arrives PRE-SPLIT along its contiguous axis (m3t_f16x3_split: the 16 bytes of rows row0 .. row0 + 3 at one k hold
// {hi r0|r1, hi r2|r3, lo r0|r1, lo r2|r3}); the records pair consecutive k of ONE row, so the halves of two loads are re-paired with one
// v_perm_b32 per (row, k pair, term) -- 16 per thread and tile instead of 8 conversions of 6 VALU instructions each
template <int NS, bool PRE = false>
__device__ __forceinline__ void mc_store(unsigned char* __restrict__ S, const float4 (&r)[4], float scale = 1.f) {
    const int tid = threadIdx.x & 255;
    const int wave = tid >> 6, h = (tid >> 5) & 1, row0 = (tid & 31) * 4;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned lo[4][3], hi2[4][3];                     // [row][split]: k pairs (0,1) and (2,3) of this thread's k-quad
    if (PRE && NS == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned sel = (i & 1) ? 0x07060302u : 0x05040100u;      // the row's half of each word: (first k in the low half)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned c0 = __float_as_uint(s == 0 ? (i < 2 ? r[0].x : r[0].y) : (i < 2 ? r[0].z : r[0].w));
                const unsigned c1 = __float_as_uint(s == 0 ? (i < 2 ? r[1].x : r[1].y) : (i < 2 ? r[1].z : r[1].w));
                const unsigned c2 = __float_as_uint(s == 0 ? (i < 2 ? r[2].x : r[2].y) : (i < 2 ? r[2].z : r[2].w));
                const unsigned c3 = __float_as_uint(s == 0 ? (i < 2 ? r[3].x : r[3].y) : (i < 2 ? r[3].z : r[3].w));
                lo[i][s] = __builtin_amdgcn_perm(c1, c0, sel);
                hi2[i][s] = __builtin_amdgcn_perm(c3, c2, sel);
            }
        }
    } else
#pragma unroll
    for (int i = 0; i < 4; ++i) {                     // row row0 + i: its 4 k values are component i of r[0..3]
        const float v0 = i == 0 ? r[0].x : i == 1 ? r[0].y : i == 2 ? r[0].z : r[0].w;
        const float v1 = i == 0 ? r[1].x : i == 1 ? r[1].y : i == 2 ? r[1].z : r[1].w;
        const float v2 = i == 0 ? r[2].x : i == 1 ? r[2].y : i == 2 ? r[2].z : r[2].w;
        const float v3 = i == 0 ? r[3].x : i == 1 ? r[3].y : i == 2 ? r[3].z : r[3].w;
        split3_pair<NS>((f32x2){v0, v1}, lo[i], scale);
        split3_pair<NS>((f32x2){v2, v3}, hi2[i], scale);
    }
    unsigned char* q = S + wave * PLANE + (row0 + 2 * h) * 16;
#pragma unroll
    for (int s = 0; s < planes_of(NS); ++s) {
        // permlane32_swap(a, b): a's upper half <-> b's lower half.  lower lanes: (own row 0 | partner's row 0);
        // upper lanes: (partner's row 2 | own row 2) -- in both cases (.x, .y) = (even k-quad, odd k-quad)
        const u32x2 e0 = __builtin_amdgcn_permlane32_swap(lo[0][s], lo[2][s], false, false);
        const u32x2 e1 = __builtin_amdgcn_permlane32_swap(hi2[0][s], hi2[2][s], false, false);
        const u32x2 f0 = __builtin_amdgcn_permlane32_swap(lo[1][s], lo[3][s], false, false);
        const u32x2 f1 = __builtin_amdgcn_permlane32_swap(hi2[1][s], hi2[3][s], false, false);
        *reinterpret_cast<u32x4*>(q + s * SPLIT_BYTES) = (u32x4){e0.x, e1.x, e0.y, e1.y};
        *reinterpret_cast<u32x4*>(q + s * SPLIT_BYTES + 16) = (u32x4){f0.x, f1.x, f0.y, f1.y};
    }
}

// NS = 3: fp32-accurate product from 3 bf16 terms per operand (6 MFMAs per tile step);
// NS = 1: plain bf16 operands (round to nearest even), fp32 accumulate -- the mixed-precision mode (M3T_GEMM_BF16)
// CONV (TA == 0): the product is a dilated 1-D convolution over channel-last rows (models/tcn.py:16-46, the tcn_simple stages of
// models/backbone.py:214-222) as an implicit GEMM: k = (tap j, channel c), A[m][k] = x[m + off_j][c] where off_j is the tap's time
// offset (causal / look-ahead `lead` / time-flipped for the data gradient) and rows whose source frame falls outside the clip
// read as zero -- no im2col, no padded copy, no chomp copy.  A 32-deep k tile lies inside one tap (C % 32 == 0), so a tile's
// loads are ordinary full-line row loads from shifted rows.  Epilogue: bias, pre-activation copy, ReLU x dropout mask,
// residual add + ReLU (the TemporalBlock's tail) fused.
// XNT = 64: a 128 x 64 output tile (each wave 64 x 32) for grids that would leave most CUs with a single 128 x 128 workgroup
// (N = 512 at M = 9600: 300 tiles for 768 slots) -- twice the workgroups, the B operand staged for 64 rows only.
// MW (TA == 0, no CONV / SEG): A's and C's rows go through the window map of X6Params (a 128-entry table in LDS: one division per thread)
// C3 (TA == 0, TB == 0): the 3-D form of CONV for the visual stems -- k = (tap (kt, kh, kw), channel c) over CHANNELS-LAST rows, a 32-deep k tile
// inside one tap (C % 32 == 0): A rows are whole-line loads from rows shifted by the tap on three axes, zero outside the source grid; B is
// the plain [taps * C][N] matrix.  Used for the convolutions' data gradient (source = dy channels-last, which backward has anyway for the
// weight gradient; reference models/backbone.py:73-103,179-271): no patch matrix, no col2im.
// C3 = 2 (TA == 1, TB == 0): the convolution's WEIGHT gradient as the same walk turned round (m3t_conv3d_wgrad_taps) -- dW^T[(tap, ci)][co] =
// sum over rows r = (n, t', h', w') of the dy grid of x[(n, t' st + kt - pt, ...)][ci] dy[r][co]: the REDUCTION runs over the rows, A's column
// m = (tap, ci) picks the tap once per thread, every k row is decoded on the dy grid (a mixed-radix counter advanced by 32 per tile: no
// division in the loop) and shifted by that tap, zero outside x's grid; columns past taps * C (M padded to the tile) read as zero.  Neither
// the forward pass nor this one needs the patch matrix (9 - 27 x the activations: 0.9 GB per 3 x 3 layer at 512 frames of 28 x 28).
// C3 = 3 (TA == 0, TB == 1, images): the stems' FIRST layers (3 input channels: reference models/backbone.py:73-78,179-184).  The input is
// channels-last PADDED TO FOUR channels, the kernel's width to eight taps (zero weights): a 32-deep k tile is one (kt, kh) pair -- its eight
// k-quads are the eight pixels w .. w + 7 of one source row, a thread's quad one pixel's four channels -- so the A tile is again whole
// 16-byte loads from shifted rows, 21 of 32 k useful (the patch matrix of this layer was 4.9 GB at 512 frames of 112 x 112).
// PRE (NS == 4, K-contiguous operands): bit 0 -- A is a pre-split image, bit 1 -- B is (see kc_store)
template <int TA, int TB, bool SEG, int NS, bool CONV = false, int XNT = 128, bool MW = false, int C3 = 0, int PRE = 0>
__global__ __launch_bounds__(256, 3) void sgemm_x6_kernel(X6Params p, X6Batch bt) {
    if (MW) {                                            // this workgroup's problem of the batch
        const int z = blockIdx.z;
        p.A = bt.A[z]; p.B = bt.B[z]; p.C = bt.C[z]; p.bias = bt.bias[z];
        p.amax_a = bt.amax_a[z]; p.amax_b = bt.amax_b[z]; p.accumulate = bt.accumulate[z];
    }
    constexpr int NJ = XNT / 64;                     // 32-column MFMA tiles per wave along N
    constexpr int BR = XNT / 32;                     // rows per thread of a K-contiguous B tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * OPER_BYTES];   // A | B, 48 KiB
    unsigned char* As = lds;
    unsigned char* Bs = lds + OPER_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    const int tn_ = gridDim.x, nt_ = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * tn_ + blockIdx.x;
    const int xq = nt_ >> 3, xr = nt_ & 7, xcd = lin & 7, slot = lin >> 3;
    const int til = xcd * xq + min(xcd, xr) + slot;           // XCD-contiguous tile order (see gemm.hip)
    const int bm = (til / tn_) * XM, bn = (til % tn_) * XNT;
    const bool bcol = XNT == 128 || (tid & 31) < 16;      // row-contiguous B: this thread's 4 columns lie inside the tile
    const int k_begin = MW ? 0 : blockIdx.z * p.kchunk;
    const int k_end = min(p.K, k_begin + p.kchunk);
    // (C3 = 2 and the segmented reduction: a ragged last tile reads zeros past K)
    const int ntiles = (C3 == 2 || SEG) ? (k_end - k_begin + XK - 1) / XK : (k_end - k_begin) / XK;
    __shared__ int rowmap[MW ? XM : 1];              // MW: storage row of the tile's row i
    if (MW) {
        if (tid < XM) {
            const int m = bm + tid, q = m / p.mw_len;
            rowmap[tid] = q * p.mw_stride + (m - q * p.mw_len) + p.mw_off;
        }
        __syncthreads();
    }

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float sc_a = 1.f, sc_b = 1.f, sc_ia = 1.f, sc_ib = 1.f;      // NS = 4: operand scales and their inverses (powers of two)
    if (NS == 4) {
        m3t_f16_scale((unsigned)*p.amax_a, sc_a, sc_ia);
        m3t_f16_scale((unsigned)*p.amax_b, sc_b, sc_ib);
    }

    // per-thread operand pointers
    const float* pa; const float* pb;
    size_t a_step, b_step, a_krow = 0, b_krow = 0;
    const float* paw[MW ? 4 : 1];                    // MW: one pointer per row of this thread (the rows are not equidistant)
    if (TA == 0) { pa = p.A + (size_t)(bm + (tid >> 3)) * p.lda + k_begin + (tid & 7) * 4; a_step = XK; }
    else { pa = p.A + (size_t)(k_begin + (tid >> 5) * 4) * p.lda + bm + (tid & 31) * 4; a_step = (size_t)XK * p.lda; a_krow = p.lda; }
    if (MW) {
#pragma unroll
        for (int i = 0; i < 4; ++i) paw[i] = p.A + (size_t)rowmap[(tid >> 3) + 32 * i] * p.lda + k_begin + (tid & 7) * 4;
    }
    if (TB == 1) { pb = p.B + (size_t)(bn + (tid >> 3)) * p.ldb + k_begin + (tid & 7) * 4; b_step = XK; }
    else { pb = p.B + (size_t)(k_begin + (tid >> 5) * 4) * p.ldb + bn + (tid & 31) * 4; b_step = (size_t)XK * p.ldb; b_krow = p.ldb; }
    // segmented reduction rows (dW_hh): (segment, offset) of this thread's first k row, advanced per tile
    int sq = 0, sr = 0;
    if (SEG) {
        const int k = k_begin + (tid >> 5) * 4;
        sq = k / p.seg_len; sr = k % p.seg_len;
    }

    constexpr bool C3ROWS = C3 == 1 || C3 == 3;
    int c3_tt[C3ROWS ? 4 : 1], c3_hh[C3ROWS ? 4 : 1], c3_ww[C3ROWS ? 4 : 1], c3_nb[C3ROWS ? 4 : 1];      // C3 = 1, 3: (t st + bt, h sh + bh, w sw + bw, n * To) of this thread's four rows
    if (C3ROWS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = bm + (tid >> 3) + 32 * i;
            const bool past = m >= p.M;                        // round 6: M need not fill the last tile -- its rows past M read zeros and are not stored
            const int w_ = m % p.c3_W; m /= p.c3_W;
            const int h_ = m % p.c3_H; m /= p.c3_H;
            const int t_ = m % p.c3_T; m /= p.c3_T;
            c3_tt[i] = past ? -(1 << 28) : t_ * p.c3_st + p.c3_bt; c3_hh[i] = h_ * p.c3_sh + p.c3_bh; c3_ww[i] = w_ * p.c3_sw + p.c3_bw; c3_nb[i] = past ? 0 : m * p.c3_To;
        }
    }
    // C3 = 2: this thread's tap (from its four A columns) and the dy-grid coordinates of its first k row, advanced by XK per tile
    int g_w = 0, g_h = 0, g_t = 0, g_n = 0, adv_w = 0, adv_h = 0, adv_t = 0, adv_n = 0, g_dt = 0, g_dh = 0, g_dw = 0, g_c = 0;
    int g_k = 0;                                     // C3 = 2, SEG: the reduction row of this thread's first load of the next tile (rows past K: zeros)
    bool g_ok = false;
    if (SEG) g_k = k_begin + (tid >> 5) * 4;
    if (C3 == 2) {
        g_k = k_begin + (tid >> 5) * 4;
        const int m0 = bm + (tid & 31) * 4;
        const int j = m0 / p.c3_C;
        g_c = m0 - j * p.c3_C;
        g_ok = j < p.c3_kt * p.c3_kh * p.c3_kw;
        const int khw = p.c3_kh * p.c3_kw;
        const int jt = j / khw, jr = j - jt * khw, jh = jr / p.c3_kw, jw = jr - jh * p.c3_kw;
        g_dt = p.c3_bt + jt; g_dh = p.c3_bh + jh; g_dw = p.c3_bw + jw;
        int k = k_begin + (tid >> 5) * 4;
        g_w = k % p.c3_W; k /= p.c3_W; g_h = k % p.c3_H; k /= p.c3_H; g_t = k % p.c3_T; g_n = k / p.c3_T;
        int a = XK;
        adv_w = a % p.c3_W; a /= p.c3_W; adv_h = a % p.c3_H; a /= p.c3_H; adv_t = a % p.c3_T; adv_n = a / p.c3_T;
    }
    int cv_t[4] = {0, 0, 0, 0}, cv_k = k_begin;       // CONV: time index of this thread's four A rows; k of the next tile
    if (CONV) {
#pragma unroll
        for (int i = 0; i < 4; ++i) cv_t[i] = (bm + (tid >> 3) + 32 * i) % p.cv_T;
    }
    float4 ra[4], rb[4];
    auto gload = [&]() {
        if (C3 == 2) {
            int w_ = g_w, h_ = g_h, t_ = g_t, n_ = g_n;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ts = t_ * p.c3_st + g_dt, hs = h_ * p.c3_sh + g_dh, ws = w_ * p.c3_sw + g_dw;
                const bool kin = g_k + e < k_end;
                const bool ok = kin && g_ok && (unsigned)ts < (unsigned)p.c3_To && (unsigned)hs < (unsigned)p.c3_Ho && (unsigned)ws < (unsigned)p.c3_Wo;
                const size_t row = ((size_t)(n_ * p.c3_To + ts) * p.c3_Ho + hs) * p.c3_Wo + ws;
                ra[e] = ok ? *reinterpret_cast<const float4*>(p.A + row * p.lda + g_c) : make_float4(0.f, 0.f, 0.f, 0.f);
                rb[e] = (bcol && kin) ? *reinterpret_cast<const float4*>(pb + e * b_krow) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (++w_ == p.c3_W) { w_ = 0; if (++h_ == p.c3_H) { h_ = 0; if (++t_ == p.c3_T) { t_ = 0; ++n_; } } }
            }
            pb += b_step;
            g_k += XK;
            int c;                                               // the counter + 32 rows: one carry per digit
            g_w += adv_w; c = g_w >= p.c3_W; g_w -= c ? p.c3_W : 0;
            g_h += adv_h + c; c = g_h >= p.c3_H; g_h -= c ? p.c3_H : 0;
            g_t += adv_t + c; c = g_t >= p.c3_T; g_t -= c ? p.c3_T : 0;
            g_n += adv_n + c;
        } else if (C3 == 3) {
            const int j = cv_k >> 5;                                        // the tile's (kt, kh) pair: scalar
            const int jt = j / p.c3_kh, jh = j - jt * p.c3_kh;
            const int q = tid & 7;                                          // this thread's pixel of the eight
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ts = c3_tt[i] + jt, hs = c3_hh[i] + jh, ws = c3_ww[i] + q;
                const bool ok = q < p.c3_kw && (unsigned)ts < (unsigned)p.c3_To && (unsigned)hs < (unsigned)p.c3_Ho && (unsigned)ws < (unsigned)p.c3_Wo;
                const size_t row = ((size_t)(c3_nb[i] + ts) * p.c3_Ho + hs) * p.c3_Wo + ws;
                ra[i] = ok ? *reinterpret_cast<const float4*>(p.A + row * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            kc_load<BR>(pb, p.ldb, rb);
            pb += b_step;
            cv_k += XK;
        } else if (C3) {
            const int j = cv_k / p.c3_C, kc = cv_k - j * p.c3_C;            // tap index (kt, kh, kw), channel offset: scalar
            const int khw = p.c3_kh * p.c3_kw;
            const int jt = j / khw, jr = j - jt * khw, jh = jr / p.c3_kw, jw = jr - jh * p.c3_kw;
            const int dt = p.c3_sg * jt, dh = p.c3_sg * jh, dw = p.c3_sg * jw;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ts = c3_tt[i] + dt, hs = c3_hh[i] + dh, ws = c3_ww[i] + dw;
                const bool ok = (unsigned)ts < (unsigned)p.c3_To && (unsigned)hs < (unsigned)p.c3_Ho && (unsigned)ws < (unsigned)p.c3_Wo;
                const size_t row = ((size_t)(c3_nb[i] + ts) * p.c3_Ho + hs) * p.c3_Wo + ws;
                ra[i] = ok ? *reinterpret_cast<const float4*>(p.A + row * p.lda + kc + (tid & 7) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (TB == 1) kc_load<BR>(pb, p.ldb, rb);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[e] = bcol ? *reinterpret_cast<const float4*>(pb + e * b_krow) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            pb += b_step;
            cv_k += XK;
        } else if (CONV) {
            const int j = cv_k / p.cv_C, kc = cv_k - j * p.cv_C;
            const int sft = (p.cv_K - 1 - j) * p.cv_dil;
            const int off = p.cv_anti ? sft - p.cv_lead : p.cv_lead - sft;
            const float* qa = p.A + ((ptrdiff_t)(bm + (tid >> 3)) + off) * (ptrdiff_t)p.lda + kc + (tid & 7) * 4;     // (signed row shift)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = (unsigned)(cv_t[i] + off) < (unsigned)p.cv_T;
                ra[i] = ok ? *reinterpret_cast<const float4*>(qa + (ptrdiff_t)i * 32 * p.lda) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (TB == 1) kc_load<BR>(p.B + (size_t)j * p.cv_btap + (size_t)(bn + (tid >> 3)) * p.ldb + kc + (tid & 7) * 4, p.ldb, rb);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[e] = bcol ? *reinterpret_cast<const float4*>(pb + e * b_krow) : make_float4(0.f, 0.f, 0.f, 0.f);
                pb += b_step;
            }
            cv_k += XK;
        } else if (SEG) {   // TA == 1 && TB == 0: both operands row-contiguous, k rows through the segment map
            int q = sq, r = sr;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const size_t row = (size_t)q * p.seg_stride + r;
                const bool kin = g_k + e < k_end;        // round 6: K = segments x seg_len need not be a multiple of 32 (8 clips x 63 steps = 504)
                ra[e] = kin ? *reinterpret_cast<const float4*>(p.A + (row + p.a_off) * p.lda + bm + (tid & 31) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                rb[e] = (bcol && kin) ? *reinterpret_cast<const float4*>(p.B + (row + p.b_off) * p.ldb + bn + (tid & 31) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (++r >= p.seg_len) { r = 0; ++q; }
            }
            g_k += XK;
            sr += XK;
            while (sr >= p.seg_len) { sr -= p.seg_len; ++sq; }
        } else {
            if (MW) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { ra[i] = *reinterpret_cast<const float4*>(paw[i]); paw[i] += XK; }
            } else if (TA == 0) kc_load(pa, p.lda, ra);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[e] = *reinterpret_cast<const float4*>(pa + e * a_krow);
            }
            if (TB == 1) kc_load<BR>(pb, p.ldb, rb);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[e] = bcol ? *reinterpret_cast<const float4*>(pb + e * b_krow) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            pa += a_step; pb += b_step;
        }
    };
    auto sstore = [&]() {
        if (TA == 0) kc_store<NS, 4, (PRE & 1) != 0>(As, ra, sc_a); else mc_store<NS, (PRE & 1) != 0>(As, ra, sc_a);
        if (TB == 1) kc_store<NS, BR, (PRE & 2) != 0>(Bs, rb, sc_b); else mc_store<NS, (PRE & 2) != 0>(Bs, rb, sc_b);
    };

    if (ntiles > 0) { gload(); sstore(); }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) gload();
        __builtin_amdgcn_sched_barrier(0);     // the prefetch stays in flight: nothing that consumes it may be hoisted here
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            constexpr int NP = planes_of(NS);
            bf16x8 fa[NP][2], fb[NP][NJ];
#pragma unroll
            for (int s = 0; s < NP; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[s][i] = *reinterpret_cast<const bf16x8*>(As + s * SPLIT_BYTES + (kh * 2 + hi) * PLANE + (wm * 64 + i * 32 + l31) * 16);
                    if (i < NJ) fb[s][i] = *reinterpret_cast<const bf16x8*>(Bs + s * SPLIT_BYTES + (kh * 2 + hi) * PLANE + (wn * 32 * NJ + i * 32 + l31) * 16);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x16 c = acc[i][j];      // smallest terms first
                    if (NS == 4) {                 // fp16x3: lo hi, hi lo, hi hi
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[1][i]), __builtin_bit_cast(f16x8, fb[0][j]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[0][i]), __builtin_bit_cast(f16x8, fb[1][j]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[0][i]), __builtin_bit_cast(f16x8, fb[0][j]), c, 0, 0, 0);
                        acc[i][j] = c;
                        continue;
                    }
                    if (NS == 2) {                 // "high" mode: two bf16 terms per operand (16 mantissa bits), four products
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], c, 0, 0, 0);
                    }
                    if (NS == 3) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[NS - 1][i], fb[0][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[NS / 2][i], fb[NS / 2][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[NS - 1][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[NS / 2][i], fb[0][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[NS / 2][j], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
        __builtin_amdgcn_sched_barrier(0);     // bf16 splitting of the next tile happens after this tile's MFMAs
        __syncthreads();                       // everyone is done reading this tile
        if (t + 1 < ntiles) {
            sstore();
            __syncthreads();
        }
    }

    const bool direct = p.splits == 1;
    float* dst = direct ? p.C : p.ws + (size_t)blockIdx.z * p.M * p.N;
    const int ldd = direct ? p.ldc : p.N;
    if constexpr (C3 == 1 || C3 == 3) {
        // the walk's result straight into the planes layout the next operator reads (BatchNorm / pooling / CBAM work on [N, C, T, H, W]): a
        // lane holds 4 consecutive positions of one channel -- one 16-byte store per (channel, 4 positions), the 8 stores of a 32 x 32
        // accumulator tile complete one 128-byte line per channel.  Was: channels-last rows + a tiled transpose (one more pass each way)
        if (direct && p.tr_S > 0) {
            const int S = p.tr_S;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int col = bn + wn * 32 * NJ + j * 32 + l31;
                    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int m0 = bm + wm * 64 + i * 32 + 8 * q4 + 4 * hi;
                        if (m0 >= p.M) continue;
                        int n = m0 / S, sp = m0 - n * S;
                        float4 v;
                        v.x = acc[i][j][4 * q4 + 0] * sc_ia * sc_ib + bv; v.y = acc[i][j][4 * q4 + 1] * sc_ia * sc_ib + bv;
                        v.z = acc[i][j][4 * q4 + 2] * sc_ia * sc_ib + bv; v.w = acc[i][j][4 * q4 + 3] * sc_ia * sc_ib + bv;
                        if ((S & 3) == 0) *reinterpret_cast<float4*>(p.C + ((size_t)n * p.N + col) * S + sp) = v;      // (S % 4 == 0: so is M)
                        else {
                            const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (m0 + e >= p.M) break;
                                p.C[((size_t)n * p.N + col) * S + sp] = ve[e];
                                if (++sp == S) { sp = 0; ++n; }
                            }
                        }
                    }
                }
            return;
        }
    }
    float vmax = 0.f;                                // CONV: max |y| of this thread's outputs (p.cv_amax)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = bn + wn * 32 * NJ + j * 32 + l31;
            const float bv = (direct && p.bias) ? p.bias[col] : 0.f;
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = bm + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (C3ROWS && row >= p.M) continue;            // (a ragged last tile of a tap walk)
                if (CONV && p.cv_drop.on && (r & 3) == 0) m3t_drop_mask4(p.cv_drop, (uint32_t)row >> 2, (uint32_t)col, dm);
                float v = acc[i][j][r];
                if (NS == 4) v = v * sc_ia * sc_ib;              // (exact: powers of two)
                float* q = dst + (size_t)(MW ? rowmap[row - bm] : row) * ldd + col;
                if (CONV) {
                    const size_t o = (size_t)row * ldd + col;
                    v += bv;
                    if (p.cv_pre) p.cv_pre[o] = v;
                    const float mk = p.cv_drop.on ? dm[r & 3] : (p.cv_mask ? p.cv_mask[o] : 1.f);
                    if (p.act == 1) v = fmaxf(v, 0.f) * mk;
                    else if (p.act == 2) v = fmaxf(fmaxf(v, 0.f) * mk + p.cv_res[o], 0.f);
                    else if (p.cv_res) v += p.cv_res[o];
                    vmax = fmaxf(vmax, m3t_fin_abs(v));
                } else if (direct) {
                    v += bv;
                    if (p.act == 1) v = fmaxf(v, 0.f);
                    if (p.accumulate) v += *q;
                }
                *q = v;
            }
        }
    if (CONV && p.cv_amax) {                         // (uniform: every thread of the block gets here)
        __shared__ float red4[4];
        m3t_block_raise_slot(p.cv_amax, vmax, red4);
    }
}

const X6Batch g_no_batch = {};

}  // namespace

// Launches the bf16x6 kernel; the caller (m3t_sgemm) has verified: M % 128 == 0, N % 128 == 0, K % 32 == 0,
// kchunk % 32 == 0, 16-B aligned operands with ld % 4 == 0, and seg_len >= 32 when segmented.
int m3t_sgemm_x6_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                        float* C, int ldc, const float* bias, int act, int accumulate, int seg_len, int seg_stride,
                        int a_off, int b_off, float* ws, int splits, int kchunk, size_t dyn_lds, int bf16_operands, int narrow,
                        const unsigned long long* amax_a, const unsigned long long* amax_b, hipStream_t s) {
    X6Params p;
    p.amax_a = amax_a; p.amax_b = amax_b;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.ws = ws;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.act = act; p.accumulate = accumulate; p.splits = splits; p.kchunk = kchunk;
    p.seg_len = seg_len; p.seg_stride = seg_stride; p.a_off = a_off; p.b_off = b_off;
    p.cv_T = p.cv_C = p.cv_K = p.cv_dil = p.cv_lead = p.cv_anti = 0; p.cv_btap = 0; p.cv_mask = p.cv_res = nullptr; p.cv_pre = nullptr;
    p.cv_drop = m3t_make_drop(0.f, 0ull);
    p.cv_amax = nullptr;
    p.mw_len = p.mw_stride = p.mw_off = 0;
    p.c3_T = p.c3_H = p.c3_W = p.c3_To = p.c3_Ho = p.c3_Wo = p.c3_kt = p.c3_kh = p.c3_kw = p.c3_bt = p.c3_bh = p.c3_bw = p.c3_sg = p.c3_C = 0; p.c3_st = p.c3_sh = p.c3_sw = 1; p.tr_S = 0;
    dim3 grid(N / (narrow ? 64 : XN), M / XM, splits), block(256);
#define M3T_X6_DISPATCH(NS_, XNT_)                                                                                                  \
    do {                                                                                                                           \
        if (seg_len > 0) sgemm_x6_kernel<1, 0, true, NS_, false, XNT_><<<grid, block, dyn_lds, s>>>(p, g_no_batch);                            \
        else if (transA == 0 && transB == 1) sgemm_x6_kernel<0, 1, false, NS_, false, XNT_><<<grid, block, dyn_lds, s>>>(p, g_no_batch);      \
        else if (transA == 0 && transB == 0) sgemm_x6_kernel<0, 0, false, NS_, false, XNT_><<<grid, block, dyn_lds, s>>>(p, g_no_batch);      \
        else if (transA == 1 && transB == 0) sgemm_x6_kernel<1, 0, false, NS_, false, XNT_><<<grid, block, dyn_lds, s>>>(p, g_no_batch);      \
        else sgemm_x6_kernel<1, 1, false, NS_, false, XNT_><<<grid, block, dyn_lds, s>>>(p, g_no_batch);                                       \
    } while (0)
    // bf16_operands: 1 = bf16 mode (one product), 2 = "high" mode (two bf16 terms per operand, four products), 0 = fp32-accurate (bf16x6)
    // 3 = fp16x3 (two fp16 terms per scaled operand, three products; amax = the operands' magnitude slots)
    if (bf16_operands == 3 && (!amax_a || !amax_b)) return M3T_EINVAL;
    if (bf16_operands == 1) { if (narrow) M3T_X6_DISPATCH(1, 64); else M3T_X6_DISPATCH(1, 128); }
    else if (bf16_operands == 2) { if (narrow) M3T_X6_DISPATCH(2, 64); else M3T_X6_DISPATCH(2, 128); }
    else if (bf16_operands == 3) { if (narrow) M3T_X6_DISPATCH(4, 64); else M3T_X6_DISPATCH(4, 128); }
    else { if (narrow) M3T_X6_DISPATCH(3, 64); else M3T_X6_DISPATCH(3, 128); }
#undef M3T_X6_DISPATCH
    hipError_t e = hipGetLastError();
    return (int)e;
}

// C's and A's rows through a time window (MW kernels; m3t_sgemm_window / _batch): transA = 0, one K pass (no split-K slabs), fp16x3 or the
// six-product form, n problems of one shape per launch.  The caller has verified M % 128 == 0, N % 64 == 0, K % 32 == 0, alignment, mw_len >= 1.
int m3t_sgemm_x6_window_launch(int n, const m3t_window_problem* pr, int transB, int M, int N, int K, int lda, int ldb, int ldc,
                               int act, int mw_len, int mw_stride, int mw_off, int f16x3, int narrow, hipStream_t s) {
    X6Params p;
    X6Batch bt = {};
    for (int i = 0; i < n; ++i) {
        if (f16x3 && (!pr[i].amax_a || !pr[i].amax_b)) return M3T_EINVAL;
        bt.A[i] = pr[i].A; bt.B[i] = pr[i].B; bt.C[i] = pr[i].C; bt.bias[i] = pr[i].bias;
        bt.amax_a[i] = pr[i].amax_a; bt.amax_b[i] = pr[i].amax_b; bt.accumulate[i] = pr[i].accumulate;
    }
    p.amax_a = p.amax_b = nullptr;
    p.A = p.B = nullptr; p.C = nullptr; p.bias = nullptr; p.ws = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.act = act; p.accumulate = 0; p.splits = 1; p.kchunk = K;
    p.seg_len = p.seg_stride = p.a_off = p.b_off = 0;
    p.cv_T = p.cv_C = p.cv_K = p.cv_dil = p.cv_lead = p.cv_anti = 0; p.cv_btap = 0; p.cv_mask = p.cv_res = nullptr; p.cv_pre = nullptr;
    p.cv_drop = m3t_make_drop(0.f, 0ull);
    p.cv_amax = nullptr;
    p.mw_len = mw_len; p.mw_stride = mw_stride; p.mw_off = mw_off;
    dim3 grid(N / (narrow ? 64 : XN), M / XM, n), block(256);
#define M3T_X6W_GO(NS_, XNT_)                                                                             \
    do {                                                                                                  \
        if (transB) sgemm_x6_kernel<0, 1, false, NS_, false, XNT_, true><<<grid, block, 0, s>>>(p, bt);   \
        else sgemm_x6_kernel<0, 0, false, NS_, false, XNT_, true><<<grid, block, 0, s>>>(p, bt);          \
    } while (0)
    if (f16x3) { if (narrow) M3T_X6W_GO(4, 64); else M3T_X6W_GO(4, 128); }
    else { if (narrow) M3T_X6W_GO(3, 64); else M3T_X6W_GO(3, 128); }
#undef M3T_X6W_GO
    return (int)hipGetLastError();
}


// NT product on operands split once (m3t_sgemm_pre): A [M][K], B [N][K] as "P4" images.  M % 128 == 0, N % 128 == 0 (or % 64: narrow), K % 32 == 0.
int m3t_sgemm_x6_pre_launch(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, const float* bias,
                            int act, int accumulate, int narrow, const unsigned long long* amax_a, const unsigned long long* amax_b, hipStream_t s) {
    if (!amax_a || !amax_b) return M3T_EINVAL;
    X6Params p;
    p.amax_a = amax_a; p.amax_b = amax_b; p.cv_amax = nullptr;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.ws = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.act = act; p.accumulate = accumulate; p.splits = 1; p.kchunk = K;
    p.seg_len = p.seg_stride = p.a_off = p.b_off = 0;
    p.cv_T = p.cv_C = p.cv_K = p.cv_dil = p.cv_lead = p.cv_anti = 0; p.cv_btap = 0; p.cv_mask = p.cv_res = nullptr; p.cv_pre = nullptr;
    p.cv_drop = m3t_make_drop(0.f, 0ull);
    p.mw_len = p.mw_stride = p.mw_off = 0;
    p.c3_T = p.c3_H = p.c3_W = p.c3_To = p.c3_Ho = p.c3_Wo = p.c3_kt = p.c3_kh = p.c3_kw = p.c3_bt = p.c3_bh = p.c3_bw = p.c3_sg = p.c3_C = 0; p.c3_st = p.c3_sh = p.c3_sw = 1; p.tr_S = 0;
    dim3 grid(N / (narrow ? 64 : XN), M / XM, 1), block(256);
    if (narrow) sgemm_x6_kernel<0, 1, false, 4, false, 64, false, false, 3><<<grid, block, 0, s>>>(p, g_no_batch);
    else sgemm_x6_kernel<0, 1, false, 4, false, 128, false, false, 3><<<grid, block, 0, s>>>(p, g_no_batch);
    return (int)hipGetLastError();
}

// The 3-D tap walk (C3 kernels; m3t_conv3d_taps).  The caller has verified: Cd % 64 == 0 (any number of rows since round 6), Cs % 32 == 0, 16-B aligned operands.
// pre: both operands are pre-split images (m3t_f16x3_split) and w_taps is [Cd][taps * Cs] (K-contiguous)
int m3t_conv3d_taps_launch(const float* src, const float* w_taps, float* dst, int N, int Cs, int Cd, int T, int H, int W, int To, int Ho, int Wo,
                           int kt, int kh, int kw, int bt_, int bh, int bw, int sg, int f16x3, int pre, const unsigned long long* amax_a,
                           const unsigned long long* amax_b, float* ws, int splits, int kchunk, hipStream_t s, const int* stride3,
                           const float* bias, int planes) {
    X6Params p;
    if (pre && !f16x3) return M3T_EINVAL;
    if (f16x3 && (!amax_a || !amax_b)) return M3T_EINVAL;
    p.amax_a = amax_a; p.amax_b = amax_b; p.cv_amax = nullptr;
    p.A = src; p.B = w_taps; p.C = dst; p.bias = bias; p.ws = ws;            // (bias: added by the direct epilogue; the caller's slab reduction otherwise)
    p.M = N * T * H * W; p.N = Cd; p.K = kt * kh * kw * Cs; p.lda = Cs; p.ldb = pre ? kt * kh * kw * Cs : Cd; p.ldc = Cd;
    p.act = 0; p.accumulate = 0; p.splits = splits; p.kchunk = kchunk;      // (deterministic split-K slabs: the deep layers are 144-200 tiles with K = 13 824)
    p.seg_len = p.seg_stride = p.a_off = p.b_off = 0;
    p.cv_T = p.cv_C = p.cv_K = p.cv_dil = p.cv_lead = p.cv_anti = 0; p.cv_btap = 0; p.cv_mask = p.cv_res = nullptr; p.cv_pre = nullptr;
    p.cv_drop = m3t_make_drop(0.f, 0ull);
    p.mw_len = p.mw_stride = p.mw_off = 0;
    p.c3_T = T; p.c3_H = H; p.c3_W = W; p.c3_To = To; p.c3_Ho = Ho; p.c3_Wo = Wo; p.c3_kt = kt; p.c3_kh = kh; p.c3_kw = kw;
    p.c3_bt = bt_; p.c3_bh = bh; p.c3_bw = bw; p.c3_sg = sg; p.c3_C = Cs;
    p.c3_st = stride3 ? stride3[0] : 1; p.c3_sh = stride3 ? stride3[1] : 1; p.c3_sw = stride3 ? stride3[2] : 1;
    p.tr_S = (planes && splits == 1) ? T * H * W : 0;      // (dst is then [N][Cd][T H W])
    const int tm = (p.M + XM - 1) / XM;                        // (round 6: a ragged last row tile)
    const bool narrow = (Cd % 128 != 0) || (Cd / XN) * tm * splits <= 384;
    dim3 grid(Cd / (narrow ? 64 : XN), tm, splits), block(256);
#define M3T_C3_GO(NS_)                                                                                                  \
    do {                                                                                                               \
        if (narrow) sgemm_x6_kernel<0, 0, false, NS_, false, 64, false, 1><<<grid, block, 0, s>>>(p, g_no_batch);   \
        else sgemm_x6_kernel<0, 0, false, NS_, false, 128, false, 1><<<grid, block, 0, s>>>(p, g_no_batch);         \
    } while (0)
    if (pre) {
        if (narrow) sgemm_x6_kernel<0, 1, false, 4, false, 64, false, 1, 3><<<grid, block, 0, s>>>(p, g_no_batch);
        else sgemm_x6_kernel<0, 1, false, 4, false, 128, false, 1, 3><<<grid, block, 0, s>>>(p, g_no_batch);
    }
    else if (f16x3) M3T_C3_GO(4); else M3T_C3_GO(3);
#undef M3T_C3_GO
    return (int)hipGetLastError();
}

// The first layers' walk (C3 = 3 kernels; m3t_conv3d_fwd_taps4): x_img4 = image of x channels-last padded to 4 channels, w_img = image of
// [Co][kt][kh][8][4].  The caller has verified: Co % 64 == 0, kw <= 8 (any number of rows), 16-B aligned operands.
int m3t_conv3d_taps4_launch(const float* x_img4, const float* w_img, const float* bias, float* y_cl, int N, int Co, int T, int H, int W, int To,
                            int Ho, int Wo, int kt, int kh, int kw, const int* stride3, int pt, int ph, int pw, const unsigned long long* amax_x,
                            const unsigned long long* amax_w, float* ws, int splits, int kchunk, hipStream_t s, int planes) {
    X6Params p;
    if (!amax_x || !amax_w) return M3T_EINVAL;
    p.amax_a = amax_x; p.amax_b = amax_w; p.cv_amax = nullptr;
    p.A = x_img4; p.B = w_img; p.C = y_cl; p.bias = bias; p.ws = ws;
    p.M = N * To * Ho * Wo; p.N = Co; p.K = kt * kh * 32; p.lda = 4; p.ldb = kt * kh * 32; p.ldc = Co;
    p.act = 0; p.accumulate = 0; p.splits = splits; p.kchunk = kchunk;
    p.seg_len = p.seg_stride = p.a_off = p.b_off = 0;
    p.cv_T = p.cv_C = p.cv_K = p.cv_dil = p.cv_lead = p.cv_anti = 0; p.cv_btap = 0; p.cv_mask = p.cv_res = nullptr; p.cv_pre = nullptr;
    p.cv_drop = m3t_make_drop(0.f, 0ull);
    p.mw_len = p.mw_stride = p.mw_off = 0;
    p.c3_T = To; p.c3_H = Ho; p.c3_W = Wo; p.c3_To = T; p.c3_Ho = H; p.c3_Wo = W; p.c3_kt = kt; p.c3_kh = kh; p.c3_kw = kw;
    p.c3_bt = -pt; p.c3_bh = -ph; p.c3_bw = -pw; p.c3_sg = 1; p.c3_C = 4;
    p.c3_st = stride3[0]; p.c3_sh = stride3[1]; p.c3_sw = stride3[2];
    p.tr_S = (planes && splits == 1) ? To * Ho * Wo : 0;
    const int tm = (p.M + XM - 1) / XM;
    const bool narrow = (Co % 128 != 0) || (Co / XN) * tm * splits <= 384;
    dim3 grid(Co / (narrow ? 64 : XN), tm, splits), block(256);
    if (narrow) sgemm_x6_kernel<0, 1, false, 4, false, 64, false, 3, 3><<<grid, block, 0, s>>>(p, g_no_batch);
    else sgemm_x6_kernel<0, 1, false, 4, false, 128, false, 3, 3><<<grid, block, 0, s>>>(p, g_no_batch);
    return (int)hipGetLastError();
}

// The weight gradient's walk (C3 = 2 kernels; m3t_conv3d_wgrad_taps): dwt [Mp][Co] = sum over the dy grid's rows, Mp = taps * Ci rounded up
// to 128.  The caller has verified: Co % 64 == 0, Ci % 4 == 0, kchunk % 32 == 0 (any number of rows: a ragged last k tile reads zeros), 16-B aligned operands.
int m3t_conv3d_wgrad_launch(const float* x_cl, const float* dy_cl, float* dwt, int N, int Ci, int Co, int T, int H, int W, int To, int Ho, int Wo,
                            int kt, int kh, int kw, const int* stride3, int pt, int ph, int pw, int f16x3, const unsigned long long* amax_x,
                            const unsigned long long* amax_dy, float* ws, int splits, int kchunk, hipStream_t s, int pre) {
    X6Params p;
    if (f16x3 && (!amax_x || !amax_dy)) return M3T_EINVAL;
    if (pre && !f16x3) return M3T_EINVAL;             // (pre: x_cl and dy_cl are m3t_f16x3_split images under those slots)
    p.amax_a = amax_x; p.amax_b = amax_dy; p.cv_amax = nullptr;
    p.A = x_cl; p.B = dy_cl; p.C = dwt; p.bias = nullptr; p.ws = ws;
    const int Mp = (kt * kh * kw * Ci + XM - 1) / XM * XM;
    p.M = Mp; p.N = Co; p.K = N * To * Ho * Wo; p.lda = Ci; p.ldb = Co; p.ldc = Co;
    p.act = 0; p.accumulate = 0; p.splits = splits; p.kchunk = kchunk;
    p.seg_len = p.seg_stride = p.a_off = p.b_off = 0;
    p.cv_T = p.cv_C = p.cv_K = p.cv_dil = p.cv_lead = p.cv_anti = 0; p.cv_btap = 0; p.cv_mask = p.cv_res = nullptr; p.cv_pre = nullptr;
    p.cv_drop = m3t_make_drop(0.f, 0ull);
    p.mw_len = p.mw_stride = p.mw_off = 0;
    // the counter runs over the dy grid (c3_T/H/W), the source is x's grid (c3_To/Ho/Wo)
    p.c3_T = To; p.c3_H = Ho; p.c3_W = Wo; p.c3_To = T; p.c3_Ho = H; p.c3_Wo = W; p.c3_kt = kt; p.c3_kh = kh; p.c3_kw = kw;
    p.c3_bt = -pt; p.c3_bh = -ph; p.c3_bw = -pw; p.c3_sg = 1; p.c3_C = Ci;
    p.c3_st = stride3[0]; p.c3_sh = stride3[1]; p.c3_sw = stride3[2]; p.tr_S = 0;
    const bool narrow = (Co % 128 != 0) || (Co / XN) * (Mp / XM) * splits <= 384;
    dim3 grid(Co / (narrow ? 64 : XN), Mp / XM, splits), block(256);
    if (f16x3 && pre) {
        if (narrow) sgemm_x6_kernel<1, 0, false, 4, false, 64, false, 2, 3><<<grid, block, 0, s>>>(p, g_no_batch);
        else sgemm_x6_kernel<1, 0, false, 4, false, 128, false, 2, 3><<<grid, block, 0, s>>>(p, g_no_batch);
    } else if (f16x3) {
        if (narrow) sgemm_x6_kernel<1, 0, false, 4, false, 64, false, 2><<<grid, block, 0, s>>>(p, g_no_batch);
        else sgemm_x6_kernel<1, 0, false, 4, false, 128, false, 2><<<grid, block, 0, s>>>(p, g_no_batch);
    } else {
        if (narrow) sgemm_x6_kernel<1, 0, false, 3, false, 64, false, 2><<<grid, block, 0, s>>>(p, g_no_batch);
        else sgemm_x6_kernel<1, 0, false, 3, false, 128, false, 2><<<grid, block, 0, s>>>(p, g_no_batch);
    }
    return (int)hipGetLastError();
}


// Dilated 1-D convolution on the bf16x6 kernel (see CONV above).  The caller (m3t_conv1d_fwd) has verified: (B*T) % 128 == 0,
// Co % 128 == 0, Ci % 32 == 0, 16-B aligned operands.  anti = 0: w_t is [K][Co][Ci] (rows of Ci contiguous: the "TB = 1" form,
// one [Co][Ci] plane per tap); anti = 1 (data gradient): w_t is [K][Ci][Co] read as the row-major [K*Ci][Co] matrix.
int m3t_conv_x6_launch(const float* x, const float* w_t, const float* bias, const float* res, const float* mask, float* y,
                       float* pre, int B, int T, int Ci, int Co, int K, int dil, int lead, int act, int anti, int bf16_operands,
                       M3TDrop drop, const unsigned long long* amax_a, const unsigned long long* amax_b, unsigned long long* amax_y,
                       hipStream_t s) {
    X6Params p;
    p.amax_a = amax_a; p.amax_b = amax_b; p.cv_amax = amax_y;
    if (bf16_operands == 3 && (!amax_a || !amax_b)) return M3T_EINVAL;
    p.A = x; p.B = w_t; p.C = y; p.bias = bias; p.ws = nullptr;
    p.M = B * T; p.N = Co; p.K = K * Ci; p.lda = Ci; p.ldb = anti ? Co : Ci; p.ldc = Co;
    p.act = act; p.accumulate = 0; p.splits = 1; p.kchunk = K * Ci;
    p.seg_len = p.seg_stride = p.a_off = p.b_off = 0;
    p.cv_T = T; p.cv_C = Ci; p.cv_K = K; p.cv_dil = dil; p.cv_lead = lead; p.cv_anti = anti;
    p.cv_btap = (size_t)Co * Ci; p.cv_mask = mask; p.cv_res = res; p.cv_pre = pre; p.cv_drop = drop;
    p.mw_len = p.mw_stride = p.mw_off = 0;
    // 128 x 64 tiles when 128 x 128 ones would leave most CUs with one workgroup (Co = 512 at B*T = 9600: 300 tiles)
    const bool narrow = (Co / XN) * (p.M / XM) <= 384;
    dim3 grid(Co / (narrow ? 64 : XN), p.M / XM, 1), block(256);
#define M3T_CONV_GO(TB_, NS_)                                                                          \
    do {                                                                                               \
        if (narrow) sgemm_x6_kernel<0, TB_, false, NS_, true, 64><<<grid, block, 0, s>>>(p, g_no_batch); \
        else sgemm_x6_kernel<0, TB_, false, NS_, true, 128><<<grid, block, 0, s>>>(p, g_no_batch);     \
    } while (0)
    if (anti) {
        if (bf16_operands == 1) M3T_CONV_GO(0, 1); else if (bf16_operands == 2) M3T_CONV_GO(0, 2); else if (bf16_operands == 3) M3T_CONV_GO(0, 4); else M3T_CONV_GO(0, 3);
    } else {
        if (bf16_operands == 1) M3T_CONV_GO(1, 1); else if (bf16_operands == 2) M3T_CONV_GO(1, 2); else if (bf16_operands == 3) M3T_CONV_GO(1, 4); else M3T_CONV_GO(1, 3);
    }
#undef M3T_CONV_GO
    return (int)hipGetLastError();
}
