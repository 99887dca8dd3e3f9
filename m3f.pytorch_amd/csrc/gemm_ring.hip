// fp16x3 GEMM, 256 x 256 tile, operands staged by LDS-DMA into a ring of 16-k stages ("ring").  Round 6.
//
// Why (NOTEBOOK R5.4b): in the 128-tile kernels (gemm_x6d.hip, gemm_x6w.hip) 48 of 134 us are the stage's dependency chain
// global load -> wait -> split -> ds_write -> barrier; neither pre-split operands nor an LDS-DMA'd weight image moved it, because a third of
// the bytes still carried the whole chain.  Here NOTHING passes through registers on its way to LDS: both operands arrive as RAW fp32 by
// global_load_lds_dwordx4 (64-B row segments of a 16-k stage, four instructions per wave and stage), a ring of FOUR stages (128 KiB), the wait in
// front of the stage's single barrier is a counted vmcnt (two stages stay in flight across it), and the exact two-term fp16 split happens when a
// FRAGMENT is read (two ds_read_b128 = the 8 k of a lane -> hi, lo): per wave and stage 6 fragments are split for 24 MFMAs.  A wave splits its
// own fragments again for every stage it multiplies (a value is split by 2 (A) or 4 (B) waves instead of once per tile) -- VALU work bought for a
// pipeline without a store path.  8 waves as 4 (M) x 2 (N), each 64 x 128 = 2 x 4 v_mfma_f32_32x32x16_f16 tiles (128 accumulator registers).
//
// Same arithmetic as m3t_sgemm_scaled's fp16x3 kernels, bit for bit: the same power-of-two scales from the operands' magnitude slots, hi =
// fp16(s x), lo = fp16(s x - hi), the lane <-> (row, k-octet) placement of v_mfma_f32_32x32x16_f16, per 16-k stage lo hi, hi lo, hi hi, stages in
// k order, deterministic split-K slabs (m3t_sgemm's reduce).
//
// LDS image of one operand's stage: 16 pieces of 1 KiB (one wave instruction each) = 16 rows x 64 B; inside a piece the 16-B quad q of row r sits at
// slot q ^ (r >> 2): the swizzle is applied to the SOURCE address (lane (r, c) fetches quad c ^ (r >> 2)) and again on the fragment read, the LDS
// destination stays lane-linear (guide, rule 21).  A 16-lane group of a ds_read_b128 then covers 16 distinct 16-B bank groups.
#include "common.h"
#include <cstdlib>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int RM = 256, RN = 256, RKS = 16, RTH = 512, RING = 4;
constexpr int OPB = 256 * 64;                      // one operand's stage: 256 rows x 16 k x 4 B
constexpr int STG = 2 * OPB;                       // A | B

struct RingParams {
    const float* A; const float* B; float* C; const float* bias; float* ws;
    int M, N, K, lda, ldb, ldc;
    int act, accumulate, splits, kchunk;
    int seg_len, seg_stride, a_off, b_off;             // segmented reduction rows (m3t_sgemm: dW_hh), row-contiguous operands only
    const unsigned long long* amax_a;
    const unsigned long long* amax_b;
};

// the exact split of gemm_x6d.hip (NS = 4), eight values at once.  MIX: the low term by v_fma_mix{lo,hi}_f16 -- fp16(fma(x, s, -hi)) in ONE
// instruction per value (s x is exact, so is the difference: the same rounding as fp16(s x - float(hi))) instead of cvt_f32_f16 + fma + cvt:
// 16 instead of 24 VALU instructions per fragment.
template <bool MIX, bool NOSPLIT = false>
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, float sc, f16x8& h, f16x8& l) {
    if (NOSPLIT) {                                            // ablation: the raw bits as operands (timing only)
        h = __builtin_bit_cast(f16x8, v0); l = __builtin_bit_cast(f16x8, v1);
        return;
    }
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned hw[4], lw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const f32x2 vs = (f32x2){x[2 * e], x[2 * e + 1]} * sc;
        const f16x2 hh = __builtin_convertvector(vs, f16x2);
        hw[e] = __builtin_bit_cast(unsigned, hh);
        if (MIX) {
            unsigned L;
            asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(L) : "v"(x[2 * e]), "s"(sc), "v"(hw[e]));
            asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(L) : "v"(x[2 * e + 1]), "s"(sc), "v"(hw[e]));
            lw[e] = L;
        } else {
            const f32x2 r1 = vs - __builtin_convertvector(hh, f32x2);
            const f16x2 ll = __builtin_convertvector(r1, f16x2);
            lw[e] = __builtin_bit_cast(unsigned, ll);
        }
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    h = __builtin_bit_cast(f16x8, (u32x4){hw[0], hw[1], hw[2], hw[3]});
    l = __builtin_bit_cast(f16x8, (u32x4){lw[0], lw[1], lw[2], lw[3]});
}

template <int VAR>
__global__ __launch_bounds__(RTH, 1) void sgemm_ring_kernel(RingParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];     // RING stages, each A | B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                  // 4 x 2 waves: 64 x 128 of the tile each
    const int l31 = lane & 31, hi = lane >> 5;

    const int tn_ = gridDim.x, nt_ = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * tn_ + blockIdx.x;
    const int xq = nt_ >> 3, xr = nt_ & 7, xcd = lin & 7, slot = lin >> 3;
    const int til = xcd * xq + min(xcd, xr) + slot;           // XCD-contiguous tile order (see gemm.hip)
    const int bm = (til / tn_) * RM, bn = (til % tn_) * RN;
    const int k_begin = blockIdx.z * p.kchunk;
    const int k_end = min(p.K, k_begin + p.kchunk);
    const int nst = (k_end - k_begin) / RKS;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float sc_a, sc_b, sc_ia, sc_ib;
    m3t_f16_scale((unsigned)*p.amax_a, sc_a, sc_ia);
    m3t_f16_scale((unsigned)*p.amax_b, sc_b, sc_ib);

    // LDS-DMA sources: wave w fetches rows 32 w ..+31 of both operands' tiles, two pieces each; lane = (row lane >> 2, slot lane & 3) of a piece
    // and reads the quad (slot ^ (row >> 2)) of its row.  Rows past the operand's end re-read its last row (never stored).
    const int pr = lane >> 2, pq = (lane & 3) ^ ((lane >> 4) & 3);
    const float* ga[2]; const float* gb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ra = min(bm + 32 * wave + 16 * j + pr, p.M - 1);
        const int rb = min(bn + 32 * wave + 16 * j + pr, p.N - 1);
        ga[j] = p.A + (size_t)ra * p.lda + k_begin + pq * 4;
        gb[j] = p.B + (size_t)rb * p.ldb + k_begin + pq * 4;
    }
    int issued = 0;
    auto issue = [&]() {                                      // the next stage into its ring slot; past the end: the last stage again (unread)
        unsigned char* st = ldsb + (issued & (RING - 1)) * STG + wave * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(ga[j]), (__attribute__((address_space(3))) void*)(st + j * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(gb[j]), (__attribute__((address_space(3))) void*)(st + OPB + j * 1024), 16, 0,
                                             0);
        }
        const int adv = issued + 1 < nst ? RKS : 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) { ga[j] += adv; gb[j] += adv; }
        ++issued;
    };

    // fragment addresses: row r of a tile -> piece r >> 4, row r & 15, quads 2 hi and 2 hi + 1 at slots q ^ ((r & 15) >> 2)
    const int sw = (l31 & 15) >> 2;
    const int fro = (l31 >> 4) * 1024 + (l31 & 15) * 64;
    const int q0 = ((2 * hi) ^ sw) * 16, q1 = ((2 * hi + 1) ^ sw) * 16;
    const int fa_off = wm * 4096 + fro;
    const int fb_off = OPB + wn * 8192 + fro;

    // The fragment reads are inline asm: hipcc cannot tell a ds_read of stage t from the LDS-DMA of stage t+3 into the same array and would wait
    // vmcnt(0) in front of the first read of every stage (the pipeline this kernel exists for).  The counts are therefore kept by hand: LDS reads
    // return in order, so after the 12 reads of a stage lgkmcnt(8) means A's four have arrived, lgkmcnt(6 - 2 j) B's fragment j; every wait is
    // followed by a sched_barrier (hipcc moves register-only consumers above an asm wait: guide 5.4 rule 18).
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ldsb;
    const unsigned ra0 = lds0 + fa_off + q0, ra1 = lds0 + fa_off + q1, rb0 = lds0 + fb_off + q0, rb1 = lds0 + fb_off + q1;
#define RING_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define RING_WAIT_LGKM(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define RING_MFMA6(j)                                                                                                                          \
    do {                                                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i][j], 0, 0, 0);     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i][j], 0, 0, 0);     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i][j], 0, 0, 0);     \
    } while (0)
    constexpr bool PIPE = (VAR & 1) != 0, MIX = (VAR & 2) != 0, PAIR = (VAR & 4) != 0;
    constexpr bool NODMA = (VAR & 8) != 0, NOMM = (VAR & 16) != 0, NOSP = (VAR & 32) != 0;     // ablations (timing only: wrong results)
    f32x4 va[2][2], vb[4][2];
    f16x8 ah[2], al[2];
    if (!PIPE) {
        // PAIR: stages are issued two at a time (t+2, t+3 at every even t) so that the two 64-B halves of a 128-B line are requested back to back
        if (nst > 0) { issue(); issue(); if (!PAIR) issue(); }
        for (int t = 0; t < nst; ++t) {
            if (PAIR) {
                if (t & 1) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
                if (!(t & 1) && !NODMA) { issue(); issue(); }
            } else {
                asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");  // this wave's pieces of stage t have landed (t+1, t+2 in flight); then
                                                                               // everyone's have, and everyone is done reading stage t-1
                if (!NODMA) issue();                                           // stage t+3 into the slot of stage t-1
            }
            const unsigned so = (unsigned)(t & (RING - 1)) * STG;
            const unsigned a0 = ra0 + so, a1 = ra1 + so, b0 = rb0 + so, b1 = rb1 + so;
            RING_RD(va[0][0], a0, 0); RING_RD(va[0][1], a1, 0); RING_RD(va[1][0], a0, 2048); RING_RD(va[1][1], a1, 2048);
            RING_RD(vb[0][0], b0, 0); RING_RD(vb[0][1], b1, 0); RING_RD(vb[1][0], b0, 2048); RING_RD(vb[1][1], b1, 2048);
            RING_RD(vb[2][0], b0, 4096); RING_RD(vb[2][1], b1, 4096); RING_RD(vb[3][0], b0, 6144); RING_RD(vb[3][1], b1, 6144);
            RING_WAIT_LGKM(8);
#pragma unroll
            for (int i = 0; i < 2; ++i) split8<MIX, NOSP>(va[i][0], va[i][1], sc_a, ah[i], al[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j == 0) RING_WAIT_LGKM(6);
                if (j == 1) RING_WAIT_LGKM(4);
                if (j == 2) RING_WAIT_LGKM(2);
                if (j == 3) RING_WAIT_LGKM(0);
                f16x8 bh, bl;
                split8<MIX, NOSP>(vb[j][0], vb[j][1], sc_b, bh, bl);
                RING_MFMA6(j);
            }
        }
    } else {
        // PIPE: a stage's fragments are read one stage AHEAD, each into the registers its predecessor's split has just freed -- the LDS latency of
        // stage t+1 runs under the MFMAs of stage t, and A's split of stage t+1 under the last MFMAs of stage t.  Stage t+1 must then be in LDS
        // at barrier t: vmcnt(4) (only stage t+2 in flight across it); stage t+3 goes into the slot of stage t-1, whose reads were issued in
        // iteration t-2 and all waited for in iteration t-1.  Outstanding reads at the wait for B_j(t): B_j+1..3 (t), A (t+1), B_0..j-1 (t+1) = 10.
        if (nst > 0) {
            issue(); issue(); issue();
            asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            const unsigned a0 = ra0, a1 = ra1, b0 = rb0, b1 = rb1;
            RING_RD(va[0][0], a0, 0); RING_RD(va[0][1], a1, 0); RING_RD(va[1][0], a0, 2048); RING_RD(va[1][1], a1, 2048);
            RING_RD(vb[0][0], b0, 0); RING_RD(vb[0][1], b1, 0); RING_RD(vb[1][0], b0, 2048); RING_RD(vb[1][1], b1, 2048);
            RING_RD(vb[2][0], b0, 4096); RING_RD(vb[2][1], b1, 4096); RING_RD(vb[3][0], b0, 6144); RING_RD(vb[3][1], b1, 6144);
            RING_WAIT_LGKM(8);
#pragma unroll
            for (int i = 0; i < 2; ++i) split8<MIX, NOSP>(va[i][0], va[i][1], sc_a, ah[i], al[i]);
        }
        for (int t = 0; t < nst; ++t) {
            if (NODMA) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");  // stage t+1 has landed for everyone; everyone is done with stage t-1's slot
            if (!NODMA) issue();                                           // stage t+3
            const unsigned so = (unsigned)((t + 1) & (RING - 1)) * STG;
            const unsigned a0 = ra0 + so, a1 = ra1 + so, b0 = rb0 + so, b1 = rb1 + so;
            RING_RD(va[0][0], a0, 0); RING_RD(va[0][1], a1, 0); RING_RD(va[1][0], a0, 2048); RING_RD(va[1][1], a1, 2048);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                RING_WAIT_LGKM(10);
                f16x8 bh, bl;
                if (NOMM) {
                    asm volatile("" :: "v"(vb[j][0]), "v"(vb[j][1]));
                } else
                split8<MIX, NOSP>(vb[j][0], vb[j][1], sc_b, bh, bl);
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0) { RING_RD(vb[0][0], b0, 0); RING_RD(vb[0][1], b1, 0); }
                if (j == 1) { RING_RD(vb[1][0], b0, 2048); RING_RD(vb[1][1], b1, 2048); }
                if (j == 2) { RING_RD(vb[2][0], b0, 4096); RING_RD(vb[2][1], b1, 4096); }
                if (j == 3) { RING_RD(vb[3][0], b0, 6144); RING_RD(vb[3][1], b1, 6144); }
                if (!NOMM) RING_MFMA6(j);
            }
            RING_WAIT_LGKM(8);                                             // A (t+1); B (t+1) stays in flight
            if (NOMM) asm volatile("" :: "v"(va[0][0]), "v"(va[0][1]), "v"(va[1][0]), "v"(va[1][1]));
            else
#pragma unroll
            for (int i = 0; i < 2; ++i) split8<MIX, NOSP>(va[i][0], va[i][1], sc_a, ah[i], al[i]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#undef RING_MFMA6
#undef RING_WAIT_LGKM
#undef RING_RD
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the unread tail stages: no LDS-DMA may outlive the workgroup

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
                if (row >= p.M) continue;
                float v = acc[i][j][r] * sc_ia * sc_ib;      // (exact: powers of two)
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

// ---- the same tile for operands stored ROW-contiguous in their M / N index (transA = 1: A [K][M]; transB = 0: B [K][N]) ---------------------------
// The weight gradients dW = dG^T X (reference autograd of models/rnn.py:17,22-55,75: K = B T = 9 600 rows deep, both operands activations stored
// [rows][features]) and the data gradients dX = dG W are the larger part of the step's GEMM time, and the 128-tile kernels pay for them with a
// 4 x 4 cross-lane transpose per staged float4 on top of the split.  Here a k ROW of the tile is one LDS-DMA instruction (256 m x 4 B = 1 KiB,
// whole 128-B lines), the stage is [16 k][256 m], and a lane picks its 8 k with four ds_read2st64_b32 (two rows 1 KiB apart each): no transpose
// anywhere.  AM / BM: that operand is row-contiguous (else K-contiguous, the image of the NT kernel above).  SEG: the reduction index walks
// segments of rows (m3t_sgemm's dW_hh form).  One fragment is read ahead of the one being split (at most 12 LDS reads outstanding: lgkmcnt has
// four bits); the v_fma_mix split; stages and barrier as the NT kernel's plain loop.
template <bool AM, bool BM, bool SEG>
__global__ __launch_bounds__(RTH, 1) void sgemm_ringt_kernel(RingParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];     // RING stages, each A | B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int tn_ = gridDim.x, nt_ = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * tn_ + blockIdx.x;
    const int xq = nt_ >> 3, xr = nt_ & 7, xcd = lin & 7, slot = lin >> 3;
    const int til = xcd * xq + min(xcd, xr) + slot;
    const int bm = (til / tn_) * RM, bn = (til % tn_) * RN;
    const int k_begin = blockIdx.z * p.kchunk;
    const int k_end = min(p.K, k_begin + p.kchunk);
    const int nst = (k_end - k_begin) / RKS;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float sc_a, sc_b, sc_ia, sc_ib;
    m3t_f16_scale((unsigned)*p.amax_a, sc_a, sc_ia);
    m3t_f16_scale((unsigned)*p.amax_b, sc_b, sc_ib);

    // ---- LDS-DMA sources.  K-contiguous operand: as the NT kernel (rows 32 w ..+31, two 16-row pieces, quad swizzle on the source).
    //      Row-contiguous operand: wave w fetches the k rows 2 w and 2 w + 1 of the stage, lane = columns 4 lane ..+3 (clamped inside the operand).
    const int pr = lane >> 2, pq = (lane & 3) ^ ((lane >> 4) & 3);
    const float* ga[2]; const float* gb[2];
    int sqa[2] = {0, 0}, sra[2] = {0, 0};                       // SEG: (segment, offset) of this wave's two k rows
    const int mca = min(bm + lane * 4, max(p.M - 4, 0)), mcb = min(bn + lane * 4, max(p.N - 4, 0));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k_begin + 2 * wave + j;
        if (SEG) { sqa[j] = k / p.seg_len; sra[j] = k % p.seg_len; }
        if (AM) ga[j] = SEG ? p.A : p.A + (size_t)k * p.lda + mca;
        else ga[j] = p.A + (size_t)min(bm + 32 * wave + 16 * j + pr, p.M - 1) * p.lda + k_begin + pq * 4;
        if (BM) gb[j] = SEG ? p.B : p.B + (size_t)k * p.ldb + mcb;
        else gb[j] = p.B + (size_t)min(bn + 32 * wave + 16 * j + pr, p.N - 1) * p.ldb + k_begin + pq * 4;
    }
    int issued = 0;
    auto issue = [&]() {
        unsigned char* st = ldsb + (issued & (RING - 1)) * STG + wave * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* sa = ga[j]; const float* sb = gb[j];
            if (SEG) {
                const size_t row = (size_t)sqa[j] * p.seg_stride + sra[j];
                sa = p.A + (row + p.a_off) * p.lda + mca;
                sb = p.B + (row + p.b_off) * p.ldb + mcb;
            }
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(sa), (__attribute__((address_space(3))) void*)(st + j * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(sb), (__attribute__((address_space(3))) void*)(st + OPB + j * 1024), 16, 0, 0);
        }
        const bool adv = issued + 1 < nst;                    // past the end: the last stage again (unread)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (SEG) {
                if (adv) { sra[j] += RKS; if (sra[j] >= p.seg_len) { sra[j] -= p.seg_len; ++sqa[j]; } }      // (seg_len >= 32 > RKS: one wrap at most)
            } else {
                ga[j] += adv ? (AM ? (size_t)RKS * p.lda : (size_t)RKS) : 0;
                gb[j] += adv ? (BM ? (size_t)RKS * p.ldb : (size_t)RKS) : 0;
            }
        }
        ++issued;
    };

    // ---- fragment addresses
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ldsb;
    const int sw = (l31 & 15) >> 2;
    const int fro = (l31 >> 4) * 1024 + (l31 & 15) * 64;
    const int q0 = ((2 * hi) ^ sw) * 16, q1 = ((2 * hi + 1) ^ sw) * 16;
    // K-contiguous: two ds_read_b128 at ra0 / ra1 (+ 2048 per 32 rows); row-contiguous: rows 8 hi + e at 1 KiB each, column (tile row) x 4 B
    const unsigned ra0 = lds0 + (AM ? hi * 8192 + (wm * 64 + l31) * 4 : wm * 4096 + fro + q0), ra1 = lds0 + wm * 4096 + fro + q1;
    const unsigned rb0 = lds0 + OPB + (BM ? hi * 8192 + (wn * 128 + l31) * 4 : wn * 8192 + fro + q0), rb1 = lds0 + OPB + wn * 8192 + fro + q1;
    typedef unsigned ru32x2 __attribute__((ext_vector_type(2)));
#define RT_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define RT_RD2(dst, addr, o0, o1) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(o0), "n"(o1))
#define RT_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    struct Raw { f32x4 q[2]; ru32x2 d[4]; };
    // one fragment's reads: `idx` = 32-row block of the operand's wave tile
    auto rd_a = [&](Raw& r, unsigned so, int idx) {
        if (AM) {
            const unsigned a = ra0 + so + idx * 128;
            RT_RD2(r.d[0], a, 0, 4); RT_RD2(r.d[1], a, 8, 12); RT_RD2(r.d[2], a, 16, 20); RT_RD2(r.d[3], a, 24, 28);
        } else {
            const unsigned a = ra0 + so + idx * 2048, b = ra1 + so + idx * 2048;
            RT_RD128(r.q[0], a, 0); RT_RD128(r.q[1], b, 0);
        }
    };
    auto rd_b = [&](Raw& r, unsigned so, int idx) {
        if (BM) {
            const unsigned a = rb0 + so + idx * 128;
            RT_RD2(r.d[0], a, 0, 4); RT_RD2(r.d[1], a, 8, 12); RT_RD2(r.d[2], a, 16, 20); RT_RD2(r.d[3], a, 24, 28);
        } else {
            const unsigned a = rb0 + so + idx * 2048, b = rb1 + so + idx * 2048;
            RT_RD128(r.q[0], a, 0); RT_RD128(r.q[1], b, 0);
        }
    };
    auto split_raw = [&](const Raw& r, bool mc, float sc, f16x8& h, f16x8& l) {
        if (mc) {
            const f32x4 v0 = {__uint_as_float(r.d[0].x), __uint_as_float(r.d[0].y), __uint_as_float(r.d[1].x), __uint_as_float(r.d[1].y)};
            const f32x4 v1 = {__uint_as_float(r.d[2].x), __uint_as_float(r.d[2].y), __uint_as_float(r.d[3].x), __uint_as_float(r.d[3].y)};
            split8<true>(v0, v1, sc, h, l);
        } else split8<true>(r.q[0], r.q[1], sc, h, l);
    };
    constexpr int NB = BM ? 4 : 2;                            // LDS reads per B fragment

    if (nst > 0) { issue(); issue(); issue(); }
    for (int t = 0; t < nst; ++t) {
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");      // stage t has landed for everyone (t+1, t+2 in flight); stage t-1's slot is free
        issue();                                                           // stage t+3
        const unsigned so = (unsigned)(t & (RING - 1)) * STG;
        Raw xa[2], xb[2];
        f16x8 ah[2], al[2];
        rd_a(xa[0], so, 0); rd_a(xa[1], so, 1); rd_b(xb[0], so, 0);
        if (NB == 4) RT_WAIT(4); else RT_WAIT(2);                          // A's fragments are in, B_0 may still be in flight
        split_raw(xa[0], AM, sc_a, ah[0], al[0]);
        split_raw(xa[1], AM, sc_a, ah[1], al[1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            if (j < 3) {
                rd_b(xb[(j + 1) & 1], so, j + 1);                          // one fragment ahead
                if (NB == 4) RT_WAIT(4); else RT_WAIT(2);
            } else RT_WAIT(0);
            f16x8 bh, bl;
            split_raw(xb[j & 1], BM, sc_b, bh, bl);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i][j], 0, 0, 0);
        }
    }
#undef RT_WAIT
#undef RT_RD2
#undef RT_RD128
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

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
                if (row >= p.M) continue;
                float v = acc[i][j][r] * sc_ia * sc_ib;
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

// The fp16x3 product of m3t_sgemm_scaled on the ring kernels: N % 256 == 0, K and kchunk % 16 == 0, 16-B aligned operands with ld % 4 == 0; any M
// (M % 4 == 0 when A is row-contiguous).  transA = 0, transB = 1: the NT kernel (variant = build of its main loop); else the row-contiguous
// loaders (sgemm_ringt_kernel), seg_len > 0 with transA = 1, transB = 0 only.  splits > 1: slabs into ws (the caller runs m3t_sgemm's reduce).
int m3t_sgemm_ring_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                          const float* bias, int act, int accumulate, int seg_len, int seg_stride, int a_off, int b_off, float* ws, int splits,
                          int kchunk, const unsigned long long* amax_a, const unsigned long long* amax_b, int variant, hipStream_t s) {
    if (!amax_a || !amax_b || N % RN != 0 || K % RKS != 0 || kchunk % RKS != 0) return M3T_EINVAL;
    if (transA && M % 4 != 0) return M3T_EINVAL;
    if (seg_len > 0 && !(transA == 1 && transB == 0 && seg_len >= 32)) return M3T_EINVAL;
    if (transA == 1 && transB == 1) return M3T_EINVAL;          // (no caller: the reference's products are NT, NN and TN)
    RingParams p;
    p.amax_a = amax_a; p.amax_b = amax_b;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.ws = ws;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.act = act; p.accumulate = accumulate; p.splits = splits; p.kchunk = kchunk;
    p.seg_len = seg_len; p.seg_stride = seg_stride; p.a_off = a_off; p.b_off = b_off;
    dim3 grid(N / RN, cdiv(M, RM), splits), block(RTH);
    const size_t lds = (size_t)RING * STG;
    static bool attr_set[16][72] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return M3T_EINVAL;
#define M3T_RING_LAUNCH(SLOT_, KERNEL_)                                                                                                 \
    do {                                                                                                                                 \
        if (!attr_set[dev][SLOT_]) {                                                                                                     \
            hipError_t ea = hipFuncSetAttribute((const void*)KERNEL_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
            if (ea != hipSuccess) return (int)ea;                                                                                        \
            attr_set[dev][SLOT_] = true;                                                                                                 \
        }                                                                                                                                \
        KERNEL_<<<grid, block, lds, s>>>(p);                                                                                             \
    } while (0)
#define M3T_RING_GO(V_) M3T_RING_LAUNCH(V_, sgemm_ring_kernel<V_>)
    if (transA == 0 && transB == 1) {
        switch (variant) {
            case 0: M3T_RING_GO(0); break;
            case 1: M3T_RING_GO(1); break;
            case 2: M3T_RING_GO(2); break;
            case 3: M3T_RING_GO(3); break;
            case 6: M3T_RING_GO(6); break;
            case 11: M3T_RING_GO(11); break;
            case 19: M3T_RING_GO(19); break;
            case 35: M3T_RING_GO(35); break;
            case 43: M3T_RING_GO(43); break;
            default: return M3T_EINVAL;
        }
    } else if (transA == 0) M3T_RING_LAUNCH(64, (sgemm_ringt_kernel<false, true, false>));
    else if (seg_len > 0) M3T_RING_LAUNCH(65, (sgemm_ringt_kernel<true, true, true>));
    else M3T_RING_LAUNCH(66, (sgemm_ringt_kernel<true, true, false>));
#undef M3T_RING_GO
#undef M3T_RING_LAUNCH
    return (int)hipGetLastError();
}
