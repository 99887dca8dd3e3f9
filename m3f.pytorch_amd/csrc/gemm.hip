// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, bit-equal to
// an fmaf chain, 256 FLOP/clk/CU).  128x128x16 block tile, 4 waves in 2x2, each wave a
// 64x64 sub-tile = 2x2 MFMA tiles; operands staged k-major in LDS (row stride 132 floats:
// conflict-free 32-lane fragment reads, 2-way (free) transposing writes); global loads
// for tile k+1 are issued before the MFMAs of tile k (register prefetch, double LDS buffer).
// Split-K goes through fp32 slabs + a fixed-order reduce: deterministic, no atomics.
#include "common.h"
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132;

struct GemmParams {
    const float* A; const float* B; float* C; const float* bias; float* ws;
    int M, N, K, lda, ldb, ldc;
    int act, accumulate, splits, kchunk;
    int seg_len, seg_stride, a_off, b_off;
    int bf16;          // round both operands to bf16 (nearest even) before the product: the mixed-precision mode
    int vecA, vecB;
};

__device__ __forceinline__ int seg_row(int k, int seg_len, int seg_stride, int off) {
    return seg_len > 0 ? (k / seg_len) * seg_stride + (k % seg_len) + off : k;
}

// Storage [rows][K] (K contiguous): thread loads 2 x float4 along K, LDS image is k-major.
__device__ __forceinline__ void load_kc(const float* __restrict__ src, int ld, int rows, int row0,
                                        int k0, int k_end, int vec, float4 (&r)[2]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = row0 + (tid >> 2) + 64 * i;
        const int k = k0 + (tid & 3) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows) {
            const float* p = src + (size_t)row * ld + k;
            if (vec && k + 3 < k_end) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                if (k + 0 < k_end) v.x = p[0];
                if (k + 1 < k_end) v.y = p[1];
                if (k + 2 < k_end) v.z = p[2];
                if (k + 3 < k_end) v.w = p[3];
            }
        }
        r[i] = v;
    }
}
__device__ __forceinline__ void store_kc(float* __restrict__ S, const float4 (&r)[2]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 2) + 64 * i;
        const int kc = (tid & 3) * 4;
        S[(kc + 0) * LDT + row] = r[i].x;
        S[(kc + 1) * LDT + row] = r[i].y;
        S[(kc + 2) * LDT + row] = r[i].z;
        S[(kc + 3) * LDT + row] = r[i].w;
    }
}
// Storage [K][cols] (cols contiguous): thread loads 2 x float4 along cols.
__device__ __forceinline__ void load_mc(const float* __restrict__ src, int ld, int cols, int col0,
                                        int k0, int k_end, int vec, int seg_len, int seg_stride, int off,
                                        float4 (&r)[2]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = k0 + (tid >> 5) + 8 * i;
        const int c = col0 + (tid & 31) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < k_end) {
            const float* p = src + (size_t)seg_row(k, seg_len, seg_stride, off) * ld + c;
            if (vec && c + 3 < cols) {
                v = *reinterpret_cast<const float4*>(p);
            } else {
                if (c + 0 < cols) v.x = p[0];
                if (c + 1 < cols) v.y = p[1];
                if (c + 2 < cols) v.z = p[2];
                if (c + 3 < cols) v.w = p[3];
            }
        }
        r[i] = v;
    }
}
__device__ __forceinline__ void store_mc(float* __restrict__ S, const float4 (&r)[2]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int kk = (tid >> 5) + 8 * i;
        const int c = (tid & 31) * 4;
        *reinterpret_cast<float4*>(&S[kk * LDT + c]) = r[i];
    }
}

// mixed-precision mode on this kernel: bf16 x bf16 products are exact in fp32, so rounding the operands and running
// the fp32 MFMA gives exactly what a bf16 MFMA with fp32 accumulation gives
__device__ __forceinline__ float rbf(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ void round_bf16(float4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { r[i].x = rbf(r[i].x); r[i].y = rbf(r[i].y); r[i].z = rbf(r[i].z); r[i].w = rbf(r[i].w); }
}

// unguarded variants for the interior fast path (M,N % 128 == 0, K-range % 16 == 0, 16-B aligned rows)
__device__ __forceinline__ void load_kc_fast(const float* __restrict__ src, int ld, int row0, int k0, float4 (&r)[2]) {
    const int tid = threadIdx.x;
    const float* p = src + (size_t)(row0 + (tid >> 2)) * ld + k0 + (tid & 3) * 4;
    r[0] = *reinterpret_cast<const float4*>(p);
    r[1] = *reinterpret_cast<const float4*>(p + (size_t)64 * ld);
}
__device__ __forceinline__ void load_mc_fast(const float* __restrict__ src, int ld, int col0, int k0, int seg_len,
                                             int seg_stride, int off, float4 (&r)[2]) {
    const int tid = threadIdx.x;
    const int k = k0 + (tid >> 5), c = col0 + (tid & 31) * 4;
    r[0] = *reinterpret_cast<const float4*>(src + (size_t)seg_row(k, seg_len, seg_stride, off) * ld + c);
    r[1] = *reinterpret_cast<const float4*>(src + (size_t)seg_row(k + 8, seg_len, seg_stride, off) * ld + c);
}

template <int TA, int TB, bool FAST, bool SEG = false>
__global__ __launch_bounds__(256) void sgemm_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2s), so give each XCD a
    // CONTIGUOUS run of tiles (column tile fastest): neighbours then share their A row-panel in that XCD's L2.
    // Bijective for any tile count; placement only affects speed.
    const int tn_ = gridDim.x, nt_ = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * tn_ + blockIdx.x;
    const int xq = nt_ >> 3, xr = nt_ & 7, xcd = lin & 7, slot = lin >> 3;
    const int til = xcd * xq + min(xcd, xr) + slot;
    const int bm = (til / tn_) * BM, bn = (til % tn_) * BN;
    const int k_begin = blockIdx.z * p.kchunk;
    const int k_end = min(p.K, k_begin + p.kchunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[2], rb[2];
    // fast path: per-thread operand pointers, advanced by one k-tile per iteration (no per-tile index math)
    const float* pa0 = nullptr; const float* pb0 = nullptr;
    size_t a_step = 0, b_step = 0, a_second = 0, b_second = 0;
    if (FAST && !SEG) {
        if (TA == 0) { pa0 = p.A + (size_t)(bm + (tid >> 2)) * p.lda + k_begin + (tid & 3) * 4; a_step = BK; a_second = (size_t)64 * p.lda; }
        else { pa0 = p.A + (size_t)(k_begin + (tid >> 5)) * p.lda + bm + (tid & 31) * 4; a_step = (size_t)BK * p.lda; a_second = (size_t)8 * p.lda; }
        if (TB == 1) { pb0 = p.B + (size_t)(bn + (tid >> 2)) * p.ldb + k_begin + (tid & 3) * 4; b_step = BK; b_second = (size_t)64 * p.ldb; }
        else { pb0 = p.B + (size_t)(k_begin + (tid >> 5)) * p.ldb + bn + (tid & 31) * 4; b_step = (size_t)BK * p.ldb; b_second = (size_t)8 * p.ldb; }
    }
    // segmented reduction rows (dW_hh): (segment, offset) of this thread's two k rows, advanced incrementally
    int sq0 = 0, sr0 = 0, sq1 = 0, sr1 = 0;
    if (FAST && SEG) {
        const int ka = k_begin + (tid >> 5), kb = ka + 8;
        sq0 = ka / p.seg_len; sr0 = ka % p.seg_len;
        sq1 = kb / p.seg_len; sr1 = kb % p.seg_len;
    }
    auto gload = [&](int k0) {
        if (FAST && SEG && TA == 1 && TB == 0) {
            const int c = (tid & 31) * 4;
            const size_t r0a = (size_t)sq0 * p.seg_stride + sr0, r1a = (size_t)sq1 * p.seg_stride + sr1;
            ra[0] = *reinterpret_cast<const float4*>(p.A + (r0a + p.a_off) * p.lda + bm + c);
            ra[1] = *reinterpret_cast<const float4*>(p.A + (r1a + p.a_off) * p.lda + bm + c);
            rb[0] = *reinterpret_cast<const float4*>(p.B + (r0a + p.b_off) * p.ldb + bn + c);
            rb[1] = *reinterpret_cast<const float4*>(p.B + (r1a + p.b_off) * p.ldb + bn + c);
            sr0 += BK; if (sr0 >= p.seg_len) { sr0 -= p.seg_len; ++sq0; }
            sr1 += BK; if (sr1 >= p.seg_len) { sr1 -= p.seg_len; ++sq1; }
        } else if (FAST && !SEG) {
            ra[0] = *reinterpret_cast<const float4*>(pa0);
            ra[1] = *reinterpret_cast<const float4*>(pa0 + a_second);
            rb[0] = *reinterpret_cast<const float4*>(pb0);
            rb[1] = *reinterpret_cast<const float4*>(pb0 + b_second);
            pa0 += a_step; pb0 += b_step;
        } else if (FAST) {
            if (TA == 0) load_kc_fast(p.A, p.lda, bm, k0, ra);
            else load_mc_fast(p.A, p.lda, bm, k0, p.seg_len, p.seg_stride, p.a_off, ra);
            if (TB == 1) load_kc_fast(p.B, p.ldb, bn, k0, rb);
            else load_mc_fast(p.B, p.ldb, bn, k0, p.seg_len, p.seg_stride, p.b_off, rb);
        } else {
            if (TA == 0) load_kc(p.A, p.lda, p.M, bm, k0, k_end, p.vecA, ra);
            else load_mc(p.A, p.lda, p.M, bm, k0, k_end, p.vecA, p.seg_len, p.seg_stride, p.a_off, ra);
            if (TB == 1) load_kc(p.B, p.ldb, p.N, bn, k0, k_end, p.vecB, rb);
            else load_mc(p.B, p.ldb, p.N, bn, k0, k_end, p.vecB, p.seg_len, p.seg_stride, p.b_off, rb);
        }
    };
    auto sstore = [&](int buf) {
        if (p.bf16) { round_bf16(ra); round_bf16(rb); }
        if (TA == 0) store_kc(As[buf], ra); else store_mc(As[buf], ra);
        if (TB == 1) store_kc(Bs[buf], rb); else store_mc(Bs[buf], rb);
    };

    const int ntiles = (k_end - k_begin + BK - 1) / BK;
    if (ntiles > 0) {
        gload(k_begin);
        sstore(0);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) gload(k_begin + (t + 1) * BK);
        const float* a_s = As[cur] + wm * 64 + l31 + hi * LDT;
        const float* b_s = Bs[cur] + wn * 64 + l31 + hi * LDT;
        // all 32 fragment reads of the tile are issued up front; the MFMAs then drain them in order
        float fa[BK / 2][2], fb[BK / 2][2];
#pragma unroll
        for (int q = 0; q < BK / 2; ++q) {
            fa[q][0] = a_s[2 * q * LDT]; fa[q][1] = a_s[2 * q * LDT + 32];
            fb[q][0] = b_s[2 * q * LDT]; fb[q][1] = b_s[2 * q * LDT + 32];
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the reads ahead of the MFMAs (hipcc would sink each next to its use)
#pragma unroll
        for (int q = 0; q < BK / 2; ++q) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][0], fb[q][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][0], fb[q][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1], fb[q][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1], fb[q][1], acc[1][1], 0, 0, 0);
        }
        if (t + 1 < ntiles) sstore(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool direct = p.splits == 1;
    float* dst = direct ? p.C : p.ws + (size_t)blockIdx.z * p.M * p.N;
    const int ldd = direct ? p.ldc : p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = bn + wn * 64 + j * 32 + l31;
            if (!FAST && col >= p.N) continue;
            const float bv = (direct && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = bm + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (!FAST && row >= p.M) continue;
                float v = acc[i][j][r];
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

__global__ void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, const float* __restrict__ bias,
                                     int M, int N, int ldc, int splits, int act, int accumulate) {
    const size_t total = (size_t)M * N;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i % N);
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += ws[(size_t)k * total + i];
        if (bias) s += bias[n];
        if (act == 1) s = fmaxf(s, 0.f);
        float* q = C + (size_t)m * ldc + n;
        if (accumulate) s += *q;
        *q = s;
    }
}

// the same sums, four columns per thread and eight slabs in flight (N % 4 == 0, 16-B aligned C / ldc % 4 == 0): the
// scalar kernel ran at 0.75 TB/s -- one dependent load per slab -- and the reduces are a quarter of the weight-gradient
// tail that ends the backward pass.  Same addition order per element (slab 0, 1, 2, ...): bit-identical results.
// ... and for MANY slabs of a SMALL output (the first layers' weight gradient: 1 513 slabs of 128 x 64 -- eight workgroups walked 1 513 slabs
// each, 94 us on the tail of the C5 step): groups of FOLD consecutive slabs are summed in place into the group's first slab by grid.y =
// groups workgroup rows, then splitk_reduce4_kernel sums the group sums (kstride = FOLD).  Fixed order, deterministic; not the order of the
// one-pass sum (the launcher takes this path by shape only, never by timing).
constexpr int SK_FOLD = 32;
__global__ __launch_bounds__(256) void splitk_fold4_kernel(float* __restrict__ ws, size_t total4, int splits) {
    float4* w4 = reinterpret_cast<float4*>(ws);
    const int k0 = blockIdx.y * SK_FOLD, k1 = min(splits, k0 + SK_FOLD);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = w4[(size_t)(k + u) * total4 + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < k1; ++k) {
            const float4 v = w4[(size_t)k * total4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        w4[(size_t)k0 * total4 + i] = s;
    }
}

__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const float* __restrict__ ws, float* __restrict__ C,
                                                             const float* __restrict__ bias, int M, int N, int ldc, int splits,
                                                             int act, int accumulate, int kstride) {
    const size_t total4 = (size_t)M * N / 4, slab4 = total4 * (size_t)kstride;
    const float4* w4 = reinterpret_cast<const float4*>(ws);
    const int n4 = N / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 8 <= splits; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = w4[(size_t)(k + u) * slab4 + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < splits; ++k) {
            const float4 v = w4[(size_t)k * slab4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int m = (int)(i / n4), n = (int)(i % n4) * 4;
        if (bias) { s.x += bias[n]; s.y += bias[n + 1]; s.z += bias[n + 2]; s.w += bias[n + 3]; }
        if (act == 1) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
        float4* q = reinterpret_cast<float4*>(C + (size_t)m * ldc + n);
        if (accumulate) { const float4 c = *q; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
        *q = s;
    }
}


// ---- fp16x3: operand magnitudes (see m3t_f16_scale and the magnitude slots in common.h) ------------------------------------------
// grid (blocks, regions): blockIdx.y = region; every block reduces a grid-stride share of its region's float4s to one value and raises
// the region's slot with ONE 64-bit atomic max of (epoch << 32 | bits of max |x| over the FINITE x): |x| as an unsigned integer
// orders like the number.
constexpr int ABSMAX_REGIONS = 16;
struct AbsmaxArgs { M3TRegion r[ABSMAX_REGIONS]; };
__global__ __launch_bounds__(256) void f16x3_absmax_kernel(AbsmaxArgs a, unsigned epoch) {
    const M3TRegion r = a.r[blockIdx.y];
    const unsigned long long total = r.rows * (unsigned long long)r.c4;
    if ((unsigned long long)blockIdx.x * 256 >= total && blockIdx.x > 0) return;
    unsigned m = 0u;
    const bool dense = (unsigned long long)r.c4 * 4ull == r.ld;
    auto at = [&](unsigned long long i) -> const uint4* {
        if (dense) return reinterpret_cast<const uint4*>(r.p + (size_t)i * 4);
        const unsigned long long row = i / (unsigned)r.c4;
        return reinterpret_cast<const uint4*>(r.p + (size_t)(row * r.ld + (i - row * (unsigned)r.c4) * 4ull));
    };
    auto fin = [](unsigned b) { b &= 0x7fffffffu; return b < 0x7f800000u ? b : 0u; };      // inf / NaN do not count: see m3t_f16_scale
    auto fold = [&](const uint4& v) { m = max(max(m, fin(v.x)), max(fin(v.y), max(fin(v.z), fin(v.w)))); };
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < total; i += 4 * stride) {            // four independent 16-B loads in flight per thread
        const uint4 v0 = *at(i), v1 = *at(i + stride), v2 = *at(i + 2 * stride), v3 = *at(i + 3 * stride);
        fold(v0); fold(v1); fold(v2); fold(v3);
    }
    for (; i < total; i += stride) fold(*at(i));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    __shared__ unsigned red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        // (a thousand 64-bit atomics on one address cost more than the pass itself: a block whose value the slot already
        // covers -- the slot only ever grows, so a stale read errs on the safe side -- skips its atomic)
        const unsigned long long val = ((unsigned long long)epoch << 32) | m;
        if (__hip_atomic_load(r.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < val) atomicMax(r.slot, val);
    }
}

static int launch_absmax(const M3TRegion* regs, int n, unsigned epoch, hipStream_t s) {
    AbsmaxArgs a;
    unsigned long long tm = 0;
    for (int i = 0; i < n; ++i) {
        a.r[i] = regs[i];
        const unsigned long long t = regs[i].rows * (unsigned long long)regs[i].c4;
        if (t > tm) tm = t;
    }
    for (int i = n; i < ABSMAX_REGIONS; ++i) a.r[i] = regs[0];
    unsigned long long blocks = (tm + 256ull * 8 - 1) / (256ull * 8);                  // >= 8 float4 per thread
    // ... and about `total` workgroups per launch: the weights' pass (16 regions per launch, the largest 393 216 float4s) ran 192 x 16 short
    // workgroups in 42 us per launch -- dispatch, not bandwidth; 24 x 16 take 14 us (round 6: 125 -> 42 us per C3 step)
    static int total = 0;
    if (!total) { const char* e = getenv("M3T_ABSMAX_TOTAL"); total = (e && atoi(e) > 0) ? atoi(e) : 384; }
    const unsigned long long cap = (unsigned long long)max(16, total / max(1, n));
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    f16x3_absmax_kernel<<<dim3((unsigned)blocks, n), 256, 0, s>>>(a, epoch);
    return (int)hipGetLastError();
}

static void launch_splitk_reduce(const float* ws, float* C, const float* bias, int M, int N, int ldc, int splits, int act,
                                 int accumulate, hipStream_t s) {
    const size_t total = (size_t)M * N;
    if (N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C % 16) == 0 && ((uintptr_t)ws % 16) == 0) {
        int blocks = (int)((total / 4 + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        if (splits >= 4 * SK_FOLD && blocks <= 64) {       // many slabs, a handful of workgroups: fold groups of slabs first
            const int groups = (splits + SK_FOLD - 1) / SK_FOLD;
            splitk_fold4_kernel<<<dim3(blocks, groups), 256, 0, s>>>(const_cast<float*>(ws), total / 4, splits);
            splitk_reduce4_kernel<<<blocks, 256, 0, s>>>(ws, C, bias, M, N, ldc, groups, act, accumulate, SK_FOLD);
            return;
        }
        splitk_reduce4_kernel<<<blocks, 256, 0, s>>>(ws, C, bias, M, N, ldc, splits, act, accumulate, 1);
        return;
    }
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    splitk_reduce_kernel<<<blocks, 256, 0, s>>>(ws, C, bias, M, N, ldc, splits, act, accumulate);
}

__global__ void colsum_kernel(const float* __restrict__ X, int M, int N, int ld, float* __restrict__ out, int accumulate) {
    // block = 256 threads = 32 columns x 8 row-lanes; fixed-order LDS tree over the 8 partials
    __shared__ float red[8][33];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    const int rl = threadIdx.x >> 5;
    float s = 0.f;
    if (c < N)
        for (int m = rl; m < M; m += 8) s += X[(size_t)m * ld + c];
    red[rl][threadIdx.x & 31] = s;
    __syncthreads();
    if (rl == 0 && c < N) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][threadIdx.x & 31];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// Two-stage column sum for tall matrices: stage 1 partial sums per row-chunk, stage 2 = colsum over partials.
__global__ void colsum_partial_kernel(const float* __restrict__ X, int M, int N, int ld, float* __restrict__ part, int rows_per) {
    __shared__ float red[8][33];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    const int rl = threadIdx.x >> 5;
    const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
    float s = 0.f;
    if (c < N)
        for (int m = m0 + rl; m < m1; m += 8) s += X[(size_t)m * ld + c];
    red[rl][threadIdx.x & 31] = s;
    __syncthreads();
    if (rl == 0 && c < N) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][threadIdx.x & 31];
        part[(size_t)blockIdx.y * N + c] = t;
    }
}

__global__ void transpose_kernel(const float* __restrict__ src, int R, int C, int lds_, float* __restrict__ dst, int ldd) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(size_t)r * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) dst[(size_t)c * ldd + r] = tile[tx][i];
    }
}

__global__ void relu_bwd_kernel(const float* __restrict__ a, float* __restrict__ dy, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (!(a[i] > 0.f)) dy[i] = 0.f;
}

__global__ __launch_bounds__(256) void mask_pos_kernel(const float* __restrict__ s, const float* __restrict__ dy, const float* __restrict__ mul,
                                                       float* __restrict__ out, size_t n, unsigned long long* amax) {
    float mx = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = s[i] > 0.f ? (mul ? dy[i] * mul[i] : dy[i]) : 0.f;
        out[i] = v;
        mx = fmaxf(mx, m3t_fin_abs(v));
    }
    __shared__ float red4[4];
    if (amax) m3t_block_raise_slot(amax, mx, red4);
}

// out = max(a + b, 0): the residual add + ReLU that ends a ResNet block (reference models/resnet.py:52-54) in one pass -- torch's `relu_(a + b)`
// is two kernels, five passes over the map instead of three.  float4 where the three pointers are 16-B aligned.
__global__ __launch_bounds__(256) void add_relu_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n, int vec,
                                                       unsigned long long* __restrict__ amax) {
    const size_t n4 = vec ? n / 4 : 0;
    const float4* a4 = reinterpret_cast<const float4*>(a); const float4* b4 = reinterpret_cast<const float4*>(b);
    float4* o4 = reinterpret_cast<float4*>(out);
    float mx = 0.f;                                  // (m3t_amax_out: the slot is raised to max |out| -- the next convolutions' operand scale)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = a4[i], y = b4[i];
        const float4 o = make_float4(fmaxf(x.x + y.x, 0.f), fmaxf(x.y + y.y, 0.f), fmaxf(x.z + y.z, 0.f), fmaxf(x.w + y.w, 0.f));
        o4[i] = o;
        mx = fmaxf(fmaxf(mx, m3t_fin_abs(o.x)), fmaxf(m3t_fin_abs(o.y), fmaxf(m3t_fin_abs(o.z), m3t_fin_abs(o.w))));
    }
    for (size_t i = n4 * 4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float o = fmaxf(a[i] + b[i], 0.f);
        out[i] = o;
        mx = fmaxf(mx, m3t_fin_abs(o));
    }
    __shared__ float red4[4];
    if (amax) m3t_block_raise_slot(amax, mx, red4);
}

// out = (s > 0) ? dy * dropout-mask(row, col) : 0 with the mask REGENERATED from (seed, row, col) (common.h): the backward of
// ReLU -> Dropout without a mask tensor.  One thread: 4 consecutive rows x 1 column (one Philox call), coalesced across columns.
__global__ __launch_bounds__(256) void mask_pos_drop_kernel(const float* __restrict__ s, const float* __restrict__ dy,
                                                            float* __restrict__ out, int rows, int C, M3TDrop drop,
                                                            unsigned long long* amax) {
    float mx = 0.f;
    const size_t total = (size_t)((rows + 3) >> 2) * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t g = (uint32_t)(i / C), col = (uint32_t)(i % C);
        float m[4];
        m3t_drop_mask4(drop, g, col, m);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t row = (size_t)g * 4 + r;
            if (row < (size_t)rows) {
                const size_t o = row * C + col;
                const float v = s[o] > 0.f ? dy[o] * m[r] : 0.f;
                out[o] = v;
                mx = fmaxf(mx, m3t_fin_abs(v));
            }
        }
    }
    __shared__ float red4[4];
    if (amax) m3t_block_raise_slot(amax, mx, red4);
}

}  // namespace


#include <mutex>
#include <map>
int m3t_f16x3_measure(const M3TRegion& ra, const unsigned long long* have_a, const M3TRegion& rb, const unsigned long long* have_b,
                      const unsigned long long** use_a, const unsigned long long** use_b, hipStream_t s) {
    *use_a = have_a; *use_b = have_b;
    if (have_a && have_b) return 0;
    struct Slot { unsigned long long* dev; unsigned epoch; };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Slot> slots;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    Slot sl;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = slots.find({dev, s});
        if (it == slots.end()) {
            Slot n; n.dev = nullptr; n.epoch = 0;
            if ((e = hipMalloc(reinterpret_cast<void**>(&n.dev), 256)) != hipSuccess) return (int)e;
            if ((e = hipMemset(n.dev, 0, 256)) != hipSuccess) return (int)e;
            it = slots.emplace(std::make_pair(dev, s), n).first;
        }
        sl.epoch = ++it->second.epoch;
        sl.dev = it->second.dev;
    }
    M3TRegion regs[2];
    int n = 0;
    if (!have_a) { regs[n] = ra; regs[n].slot = sl.dev; *use_a = sl.dev; ++n; }
    if (!have_b) { regs[n] = rb; regs[n].slot = sl.dev + 1; *use_b = sl.dev + 1; ++n; }
    return launch_absmax(regs, n, sl.epoch, s);
}

int m3t_absmax_regions(const M3TRegion* regs, int n, hipStream_t s) {          // (epoch 0: caller-owned slots)
    for (int i = 0; i < n; i += ABSMAX_REGIONS) {
        const int rc = launch_absmax(regs + i, n - i < ABSMAX_REGIONS ? n - i : ABSMAX_REGIONS, 0u, s);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int m3t_absmax(int n, const float* const* x, const size_t* rows, const int* cols, const size_t* ld,
                          unsigned long long* const* slots, void* stream) {
    if (n <= 0) return 0;
    if (n > ABSMAX_REGIONS || !x || !rows || !cols || !ld || !slots) return M3T_EINVAL;
    M3TRegion regs[ABSMAX_REGIONS];
    for (int i = 0; i < n; ++i) {
        if (!x[i] || !slots[i] || cols[i] <= 0 || cols[i] % 4 != 0 || ld[i] % 4 != 0 || (uintptr_t)x[i] % 16 != 0 || (uintptr_t)slots[i] % 8 != 0)
            return M3T_EINVAL;
        regs[i] = M3TRegion{x[i], (unsigned long long)rows[i], (unsigned long long)ld[i], cols[i] / 4, slots[i]};
    }
    return launch_absmax(regs, n, 0u, (hipStream_t)stream);
}

int m3t_sgemm_x6_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                        float* C, int ldc, const float* bias, int act, int accumulate, int seg_len, int seg_stride,
                        int a_off, int b_off, float* ws, int splits, int kchunk, size_t dyn_lds, int bf16_operands, int narrow,
                        const unsigned long long* amax_a, const unsigned long long* amax_b, hipStream_t s);

int m3t_sgemm_x6d_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                         float* C, int ldc, const float* bias, int act, int accumulate, int seg_len, int seg_stride,
                         int a_off, int b_off, float* ws, int splits, int kchunk, int bf16_operands,
                         const unsigned long long* amax_a, const unsigned long long* amax_b, hipStream_t s);

int m3t_sgemm_x6w_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                         float* C, int ldc, const float* bias, int act, int accumulate, int seg_len, int seg_stride,
                         int a_off, int b_off, float* ws, int splits, int kchunk, int bf16_operands,
                         const unsigned long long* amax_a, const unsigned long long* amax_b, hipStream_t s);

// Kernel choice among the bf16x6 GEMMs (round 3: the A/B switches M3T_GEMM_X6D / _X6C / _NARROW / _SPLITS are retired, their
// outcomes are the rules below): the 128-tile GEMMs run on the software-pipelined gemm_x6d.hip, EXCEPT those issued beside
// another stream's persistent scan (M3T_GEMM_BESIDE_SCAN: a kernel with a higher request rate takes from the scans' exchange what
// it gains, NOTEBOOK.md section 5c), which keep gemm_x6.hip; the 128 x 64 tile for N % 64 == 0 and under-filled grids.  (The 256 x 256-tile
// gemm_x6c.hip of rounds 2-3 is gone: it had no fp16x3 form, so the default mode never ran it.)
static int x6d_mode() { return 2; }

// M3T_GEMM_F16X3=0: M3T_GEMM_F16X3 is ignored (the six-product bf16 form runs instead) -- A/B runs and the switch test
bool m3t_f16x3_enabled() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("M3T_GEMM_F16X3");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    return on == 1;
}

// fp16x3 GEMMs that run alone take the software-pipelined kernel too (measured, slots given: 9600 x 1024 x 1536 NN 140 -> 129 us,
// 1536 x 1024 x 9600 TN 145 -> 138, 9600 x 1536 x 1024 NT 126 -> 122; C3 step 14.66 -> 14.57 ms)
static bool f16x3_on_x6d() { return true; }

static bool x6_enabled() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("M3T_GEMM_X6");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    return on == 1;
}

// Kernel and split-K choice of one m3t_sgemm call.  kernel: 0 fp32-MFMA (gemm.hip), 1 the 16-bit-term 128-tile kernels
// (gemm_x6.hip / gemm_x6d.hip: fp16x3, bf16x6, "high", bf16).
// (internal flag of plan_gemm: the call is not an NT product -- measured, tools/gemm_one.py: the wide tile wins on the NT forms (+9..+21 %:
// 9600 x 1536 x 1024 122 -> 115 us, 9600 x 512 x 2048 112 -> 92), loses on NN 9600 x 2048 x 512 (86 -> 104: its row-contiguous B path spills
// five registers) and is level on the TN weight gradients)
constexpr int GEMM_NO_WIDE = 1 << 20;
struct GemmPlan { int kernel, splits, kchunk, narrow, wide; };      // wide: the 128 x 256 tile of gemm_x6w.hip

static int narrow_mode() { return 1; }

// M3T_GEMM_X6W=0: never the 128 x 256 tile (A/B runs, the switch test)
static bool x6w_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("M3T_GEMM_X6W"); on = (e && e[0] == '0') ? 0 : 1; }
    return on == 1;
}

static GemmPlan plan_gemm(int transA, int M, int N, int K, int seg_len, bool vec, size_t ws_bytes, int flags) {
    GemmPlan g;
    g.narrow = 0;
    const int bf16 = (flags & M3T_GEMM_BF16) ? 1 : 0;
    const int high = (!bf16 && (flags & M3T_GEMM_HIGH)) ? 1 : 0;      // two bf16 terms per operand, four products (bf16x6 kernel, NS = 2)
    const int f16x3 = (!bf16 && !high && (flags & M3T_GEMM_F16X3) && m3t_f16x3_enabled()) ? 1 : 0;   // two fp16 terms per scaled operand, three products (NS = 4)
    const int tiles = cdiv(M, BM) * cdiv(N, BN);
    // interior shapes go to the bf16x6 kernels (fp32-accurate, 2.67x the fp32 MFMA rate)
    // (N % 64 == 0 is enough with the 128 x 64 tile: e.g. the conv3d weight gradient with N = C_in k^3 = 1728)
    const bool n64 = (N % 128 != 0) && (N % 64 == 0) && narrow_mode();
    // (round 6: the segmented reduction takes any K -- a ragged last k tile reads zeros: dW_hh at 8 clips x 63 steps, K = 504)
    const bool x6 = x6_enabled() && (M % 128 == 0) && (N % 128 == 0 || n64) && (K % 32 == 0 || seg_len >= 32) && K > 0 && vec &&
                    (seg_len == 0 || seg_len >= 32);
    const double ns_per_k = x6 ? (bf16 ? 14.0 : (high ? 27.0 : (f16x3 ? 22.0 : 36.0))) : 84.0;
    const int kq = x6 ? 32 : BK;
    // split-K choice by a small cost model (ns): a CU works through its co-resident blocks at ~0.39 TFLOP/s
    // (84 ns per k per 128x128 block; 1.3x slower when it holds a single block), slabs cost their HBM traffic.
    int splits = 1;
    const size_t cap = ws_bytes / ((size_t)M * N * sizeof(float));
    double best = 1e30;
    auto model = [&](int ntiles, double nsk, int sp) {
        const int rounds = cdiv(ntiles * sp, 256);
        double t = (double)rounds * ((double)K / sp) * nsk * (rounds == 1 ? 1.3 : 1.0);
        if (sp > 1) t += (double)(sp + 2) * M * N * 4.0 / 3000.0 + 3000.0;
        return t;
    };
    if (ws_bytes && K >= 512) {
        for (int sp = 1; sp <= 96 && sp <= K / 96 && (sp == 1 || (size_t)sp <= cap); ++sp) {
            const double t = model(tiles, ns_per_k, sp);
            if (t < best) { best = t; splits = sp; }
        }
    } else best = model(tiles, ns_per_k, 1);
    // round 5: the 128 x 256 tile (gemm_x6w.hip: fp16x3 and the six-product form) where the same model says it fills the chip better -- a
    // block of twice the work at 0.8 x the time per flop (a quarter less LDS traffic and operand splitting per MFMA), two per CU.  The
    // input projections (N = 1536: 900 tiles of 128 x 128 = one round of 768 resident workgroups and a second at 17 %) are the case.
    g.wide = 0;
    if (x6 && !bf16 && !high && N % 256 == 0 && x6w_enabled() && !(flags & (M3T_GEMM_BACKGROUND | GEMM_NO_WIDE))) {
        const int tiles_w = cdiv(M, BM) * (N / 256);
        double best_w = 1e30; int splits_w = 1;
        if (ws_bytes && K >= 512) {
            for (int sp = 1; sp <= 96 && sp <= K / 96 && (sp == 1 || (size_t)sp <= cap); ++sp) {
                const double t = model(tiles_w, 2.0 * 0.8 * ns_per_k, sp);
                if (t < best_w) { best_w = t; splits_w = sp; }
            }
        } else best_w = model(tiles_w, 2.0 * 0.8 * ns_per_k, 1);
        if (best_w < best) { g.wide = 1; splits = splits_w; }
    }
    int kchunk = cdiv(cdiv(K, splits), kq) * kq;
    if (kchunk < kq) kchunk = kq;
    g.kernel = x6 ? 1 : 0; g.kchunk = kchunk; g.splits = K > 0 ? cdiv(K, kchunk) : 1;
    // a grid that leaves most CUs with a single 128 x 128 workgroup (<= 1.5 per CU) takes 128 x 64 tiles: twice the workgroups
    // (fc0 forward, 9600 x 512 x 1024: 96 -> 82 us; not with split-K: those small problems got 10-15 % slower)
    g.narrow = (!g.wide && x6 && narrow_mode() && (n64 || narrow_mode() == 2 || (g.splits == 1 && tiles <= 384))) ? 1 : 0;      // (2: every bf16x6 GEMM, experiments)
    return g;
}

extern "C" int m3t_sgemm_plan(int transA, int M, int N, int K, int seg_len, size_t ws_bytes, int flags, int* kernel, int* splits) {
    if (M <= 0 || N <= 0 || K < 0 || !kernel || !splits) return M3T_EINVAL;
    // (the 128 x 256 tile exists for NT products only: a transA = 1 call never takes it; with transA = 0 this reports the NT form's plan)
    const GemmPlan g = plan_gemm(transA, M, N, K, seg_len, true, ws_bytes, flags | (transA ? GEMM_NO_WIDE : 0));
    *kernel = g.kernel; *splits = g.splits;
    return 0;
}

extern "C" int m3t_sgemm(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                         float* C, int ldc, const float* bias, int act, int accumulate, int seg_len, int seg_stride,
                         int a_off, int b_off, float* ws, size_t ws_bytes, int flags, void* stream) {
    return m3t_sgemm_scaled(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, seg_len, seg_stride, a_off, b_off,
                            ws, ws_bytes, flags, nullptr, nullptr, stream);
}

extern "C" int m3t_sgemm_scaled(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                                float* C, int ldc, const float* bias, int act, int accumulate, int seg_len, int seg_stride,
                                int a_off, int b_off, float* ws, size_t ws_bytes, int flags, const unsigned long long* amax_a,
                                const unsigned long long* amax_b, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K < 0 || !A || !B || !C) return M3T_EINVAL;
    if (seg_len > 0 && !(transA == 1 && transB == 0)) return M3T_EINVAL;
    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.ws = ws;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.act = act; p.accumulate = accumulate;
    p.seg_len = seg_len; p.seg_stride = seg_stride; p.a_off = a_off; p.b_off = b_off;
    p.bf16 = (flags & M3T_GEMM_BF16) ? 1 : 0;
    p.vecA = (lda % 4 == 0) && ((uintptr_t)A % 16 == 0);
    p.vecB = (ldb % 4 == 0) && ((uintptr_t)B % 16 == 0);
    const int tm = cdiv(M, BM), tn = cdiv(N, BN);
    const GemmPlan g = plan_gemm(transA, M, N, K, seg_len, p.vecA && p.vecB, ws ? ws_bytes : 0,
                                 flags | ((transA == 0 && transB == 1) ? 0 : GEMM_NO_WIDE));
    int splits = g.splits, kchunk = g.kchunk;
    // round 6 (NOTEBOOK R6.8): a GEMM issued beside persistent scans (M3T_GEMM_BESIDE_SCAN) finds ~96 free CUs, not 256: its split-K is sized for
    // 384 resident workgroups instead of the whole chip's 768 -- fewer slabs (5 instead of 10 for the 1536 x 1024 weight gradients), ~2 GB less
    // slab traffic per C3 step, a shorter reduce; measured 12.04 -> 12.00 ms (288: 12.15, 336 / 448: 12.03, 576: 12.02).  M3T_GEMM_BESIDE_WGS=0:
    // the whole-chip plan for every call; another positive value: that many workgroups
    {
        static int beside_wgs = -1;
        if (beside_wgs < 0) { const char* e = getenv("M3T_GEMM_BESIDE_WGS"); beside_wgs = e ? atoi(e) : 384; }
        if (beside_wgs > 0 && (flags & M3T_GEMM_BESIDE_SCAN) && g.kernel != 0 && !g.wide && g.splits > 1) {
            const int tiles = cdiv(M, BM) * cdiv(N, g.narrow ? 64 : BN);
            int sp = (beside_wgs + tiles / 2) / tiles;
            if (sp < 1) sp = 1;
            if (sp < g.splits) {
                kchunk = cdiv(cdiv(K, sp), 32) * 32;
                splits = cdiv(K, kchunk);
            }
        }
    }
    p.splits = splits; p.kchunk = kchunk;
    hipStream_t s = (hipStream_t)stream;
    if (g.kernel != 0) {
        int rc;
        const int f16x3 = (!p.bf16 && !(flags & M3T_GEMM_HIGH) && (flags & M3T_GEMM_F16X3) && m3t_f16x3_enabled()) ? 1 : 0;
        const unsigned long long* use_a = nullptr; const unsigned long long* use_b = nullptr;
        if (f16x3) {
            // operands without a caller's magnitude slot are measured here, over the rows x columns each one spans in memory
            // (segmented K: the rows from the first segment's first to the last segment's last, a superset)
            const size_t kr = seg_len > 0 ? (size_t)(K / seg_len - 1) * seg_stride + seg_len : (size_t)K;
            const float* A0 = seg_len > 0 ? A + (size_t)a_off * lda : A;
            const float* B0 = seg_len > 0 ? B + (size_t)b_off * ldb : B;
            const M3TRegion ra{A0, (unsigned long long)(transA ? kr : (size_t)M), (unsigned long long)lda, (transA ? M : K) / 4, nullptr};
            const M3TRegion rb{B0, (unsigned long long)(transB ? (size_t)N : kr), (unsigned long long)ldb, (transB ? K : N) / 4, nullptr};
            const int rm = m3t_f16x3_measure(ra, amax_a, rb, amax_b, &use_a, &use_b, s);
            if (rm) return rm;
        }
        if (g.wide)
            rc = m3t_sgemm_x6w_launch(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, seg_len, seg_stride,
                                      a_off, b_off, ws, splits, kchunk, f16x3 ? 3 : 0, use_a, use_b, s);
        else if (!g.narrow && K % 32 == 0 && (!f16x3 || f16x3_on_x6d()) && !(flags & (M3T_GEMM_BACKGROUND | M3T_GEMM_HIGH)) && (x6d_mode() == 1 || (x6d_mode() == 2 && !(flags & M3T_GEMM_BESIDE_SCAN))))      // ("high": measured better on gemm_x6.hip)
            // the same product, software-pipelined inside each wave (gemm_x6d.hip): bit-identical results, 8-28 % faster
            rc = m3t_sgemm_x6d_launch(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, seg_len, seg_stride,
                                      a_off, b_off, ws, splits, kchunk, p.bf16 ? 1 : ((flags & M3T_GEMM_HIGH) ? 2 : (f16x3 ? 3 : 0)), use_a, use_b, s);
        else
            rc = m3t_sgemm_x6_launch(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, seg_len, seg_stride,
                                     a_off, b_off, ws, splits, kchunk, (flags & M3T_GEMM_BACKGROUND) ? (size_t)40 * 1024 : 0,
                                     p.bf16 ? 1 : ((flags & M3T_GEMM_HIGH) ? 2 : (f16x3 ? 3 : 0)), g.narrow, use_a, use_b, s);
        if (rc) return rc;
        if (splits > 1) {
            launch_splitk_reduce(ws, C, bias, M, N, ldc, splits, act, accumulate, s);
            M3T_LAUNCH_CHECK();
        }
        return 0;
    }
    dim3 grid(tn, tm, splits), block(256);
    const bool fast = (M % BM == 0) && (N % BN == 0) && (K % BK == 0) && K > 0 && p.vecA && p.vecB;
    // M3T_GEMM_BACKGROUND: an (unused) dynamic-LDS request of 56 KiB caps residency at ONE workgroup per CU, so a
    // latency-critical kernel on another stream (the GRU scans) still finds room on every CU.
    const size_t dyn = (flags & M3T_GEMM_BACKGROUND) ? (size_t)56 * 1024 + 0 : 0;
#define M3T_GEMM_LAUNCH(TA_, TB_)                                                 \
    do {                                                                          \
        if (fast && seg_len >= BK) sgemm_kernel<TA_, TB_, true, true><<<grid, block, dyn, s>>>(p); \
        else if (seg_len > 0) sgemm_kernel<TA_, TB_, false><<<grid, block, dyn, s>>>(p);           \
        else if (fast) sgemm_kernel<TA_, TB_, true><<<grid, block, dyn, s>>>(p);    \
        else sgemm_kernel<TA_, TB_, false><<<grid, block, dyn, s>>>(p);             \
    } while (0)
    if (transA == 0 && transB == 1) M3T_GEMM_LAUNCH(0, 1);
    else if (transA == 0 && transB == 0) M3T_GEMM_LAUNCH(0, 0);
    else if (transA == 1 && transB == 0) M3T_GEMM_LAUNCH(1, 0);
    else M3T_GEMM_LAUNCH(1, 1);
#undef M3T_GEMM_LAUNCH
    M3T_LAUNCH_CHECK();
    if (splits > 1) {
        launch_splitk_reduce(ws, C, bias, M, N, ldc, splits, act, accumulate, s);
        M3T_LAUNCH_CHECK();
    }
    return 0;
}

int m3t_sgemm_x6_window_launch(int n, const m3t_window_problem* pr, int transB, int M, int N, int K, int lda, int ldb, int ldc,
                               int act, int mw_len, int mw_stride, int mw_off, int f16x3, int narrow, hipStream_t s);

// include/m3t_hip.h: the contraction over a TIME WINDOW of batch-major sequence tensors (A rows, C rows), n problems of one shape per launch
extern "C" int m3t_sgemm_window_batch(int n, const m3t_window_problem* pr, int transB, int n_seg, int win_len, int win_stride, int win_off,
                                      int N, int K, int lda, int ldb, int ldc, int act, int flags, void* stream) {
    if (n <= 0 || n_seg <= 0 || win_len <= 0 || N <= 0) return 0;
    if (n > M3T_WINDOW_BATCH || !pr || K <= 0 || win_stride < win_len || win_off < 0 || win_off + win_len > win_stride) return M3T_EINVAL;
    const long long Ml = (long long)n_seg * win_len;
    if (Ml > 0x7fffffffll || (long long)n_seg * win_stride > 0x7fffffffll) return M3T_EINVAL;
    const int M = (int)Ml;
    // only the 16-bit-term tile kernel has the windowed form: whole 128-row tiles, N in 64-column tiles, 32-deep k tiles
    if (!x6_enabled() || M % 128 != 0 || N % 64 != 0 || K % 32 != 0 || lda % 4 != 0 || ldb % 4 != 0 || (flags & (M3T_GEMM_BF16 | M3T_GEMM_HIGH)))
        return M3T_EINVAL;
    const int f16x3 = ((flags & M3T_GEMM_F16X3) && m3t_f16x3_enabled()) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    m3t_window_problem use[M3T_WINDOW_BATCH];
    for (int i = 0; i < n; ++i) {
        use[i] = pr[i];
        if (!use[i].A || !use[i].B || !use[i].C || (uintptr_t)use[i].A % 16 != 0 || (uintptr_t)use[i].B % 16 != 0) return M3T_EINVAL;
        if (f16x3 && (!use[i].amax_a || !use[i].amax_b)) {
            // an operand without a slot is measured over the rows the window's first to last storage row span (a superset): one
            // measuring launch per such problem (the hot path always brings its slots)
            if (n > 1) return M3T_EINVAL;
            const size_t span = (size_t)(n_seg - 1) * win_stride + win_len;
            const M3TRegion ra{use[i].A + (size_t)win_off * lda, (unsigned long long)span, (unsigned long long)lda, K / 4, nullptr};
            const M3TRegion rb{use[i].B, (unsigned long long)(transB ? (size_t)N : (size_t)K), (unsigned long long)ldb, (transB ? K : N) / 4, nullptr};
            const int rm = m3t_f16x3_measure(ra, use[i].amax_a, rb, use[i].amax_b, &use[i].amax_a, &use[i].amax_b, s);
            if (rm) return rm;
        }
    }
    // 128 x 64 tiles when 128 x 128 ones would not give every CU a workgroup or two
    const int narrow = (N % 128 != 0 || (long long)(M / 128) * (N / 128) * n <= 384) ? 1 : 0;
    return m3t_sgemm_x6_window_launch(n, use, transB ? 1 : 0, M, N, K, lda, ldb, ldc, act, win_len, win_stride, win_off, f16x3, narrow, s);
}

extern "C" int m3t_sgemm_window(int transB, int n_seg, int win_len, int win_stride, int win_off, int N, int K,
                                const float* A, int lda, const float* B, int ldb, float* C, int ldc, const float* bias,
                                int act, int accumulate, int flags, const unsigned long long* amax_a, const unsigned long long* amax_b,
                                void* stream) {
    const m3t_window_problem pr{A, B, C, bias, amax_a, amax_b, accumulate};
    return m3t_sgemm_window_batch(1, &pr, transB, n_seg, win_len, win_stride, win_off, N, K, lda, ldb, ldc, act, flags, stream);
}

int m3t_conv3d_taps_launch(const float* src, const float* w_taps, float* dst, int N, int Cs, int Cd, int T, int H, int W, int To, int Ho, int Wo,
                           int kt, int kh, int kw, int bt_, int bh, int bw, int sg, int f16x3, int pre, const unsigned long long* amax_a,
                           const unsigned long long* amax_b, float* ws, int splits, int kchunk, hipStream_t s, const int* stride3 = nullptr,
                           const float* bias = nullptr, int planes = 0);
extern "C" int m3t_btc_to_bct(const float* src, float* dst, int B, int T, int C, void* stream);
int m3t_conv3d_wgrad_launch(const float* x_cl, const float* dy_cl, float* dwt, int N, int Ci, int Co, int T, int H, int W, int To, int Ho, int Wo,
                            int kt, int kh, int kw, const int* stride3, int pt, int ph, int pw, int f16x3, const unsigned long long* amax_x,
                            const unsigned long long* amax_dy, float* ws, int splits, int kchunk, hipStream_t s, int pre);

int m3t_conv3d_taps4_launch(const float* x_img4, const float* w_img, const float* bias, float* y_cl, int N, int Co, int T, int H, int W, int To,
                            int Ho, int Wo, int kt, int kh, int kw, const int* stride3, int pt, int ph, int pw, const unsigned long long* amax_x,
                            const unsigned long long* amax_w, float* ws, int splits, int kchunk, hipStream_t s, int planes);

// split-K of a tap walk: enough workgroups for the chip (the deep layers are a few hundred tiles with K = 13 824), slabs in `ws`
static void taps_split(long long rows, int Cd, int K, float* ws, size_t ws_bytes, int& splits, int& kchunk) {
    splits = 1; kchunk = K;
    const long long tiles = ((rows + 127) / 128) * ((Cd + 127) / 128);
    if (!ws || tiles >= 512) return;
    int sp = (int)((640 + tiles - 1) / tiles);
    const size_t cap = ws_bytes / ((size_t)rows * Cd * sizeof(float));
    if ((size_t)sp > cap) sp = (int)cap;
    if (sp > K / 256) sp = K / 256;
    if (sp < 2) return;
    kchunk = cdiv(cdiv(K, sp), 32) * 32;
    splits = cdiv(K, kchunk);
}

namespace {
typedef _Float16 sp_f16x2 __attribute__((ext_vector_type(2)));
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));
// the "P4" image of a K-contiguous fp16x3 operand (gemm_x6.hip, kc_store<PRE>): per four consecutive values {hi 0|1, hi 2|3, lo 0|1, lo 2|3}
__global__ __launch_bounds__(256) void f16x3_split_kernel(const float* __restrict__ x, float* __restrict__ out, size_t rows, int c4, size_t ld,
                                                          size_t ldo, const unsigned long long* __restrict__ slot) {
    float sc, inv;
    m3t_f16_scale((unsigned)*slot, sc, inv);
    const size_t total = rows * (size_t)c4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / c4; const int c = (int)(i - r * c4);
        const float4 v = *reinterpret_cast<const float4*>(x + r * ld + 4 * (size_t)c);
        const sp_f32x2 a = (sp_f32x2){v.x, v.y} * sc, b = (sp_f32x2){v.z, v.w} * sc;
        const sp_f16x2 ha = __builtin_convertvector(a, sp_f16x2), hb = __builtin_convertvector(b, sp_f16x2);
        const sp_f16x2 la = __builtin_convertvector(a - __builtin_convertvector(ha, sp_f32x2), sp_f16x2);
        const sp_f16x2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, sp_f32x2), sp_f16x2);
        float4 o;
        o.x = __uint_as_float(__builtin_bit_cast(unsigned, ha)); o.y = __uint_as_float(__builtin_bit_cast(unsigned, hb));
        o.z = __uint_as_float(__builtin_bit_cast(unsigned, la)); o.w = __uint_as_float(__builtin_bit_cast(unsigned, lb));
        *reinterpret_cast<float4*>(out + r * ldo + 4 * (size_t)c) = o;
    }
}
// ... of a PERMUTED view of a small tensor (round 6: a convolution's weights [Co][Ci][taps] as the K-contiguous matrices the walks read --
// [co][(tap, ci)] forward, [ci][(tap, co)] data gradient -- without a permute copy in front of the split): out[r][t * Cc + c] from
// w[r * sR + t * sT + c * sC], four consecutive c per thread (strided reads: the tensors are a few MB)
__global__ __launch_bounds__(256) void f16x3_split_perm_kernel(const float* __restrict__ w, float* __restrict__ out, size_t R, int T, int Cc, size_t sR,
                                                               size_t sT, size_t sC, const unsigned long long* __restrict__ slot) {
    float sc, inv;
    m3t_f16_scale((unsigned)*slot, sc, inv);
    const int c4 = Cc >> 2;
    const size_t total = R * (size_t)T * c4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % (size_t)c4);
        const size_t rt = i / (size_t)c4;
        const int t = (int)(rt % (size_t)T);
        const size_t r = rt / (size_t)T;
        const float* q = w + r * sR + (size_t)t * sT + (size_t)(4 * c) * sC;
        const sp_f32x2 a = (sp_f32x2){q[0], q[sC]} * sc, b = (sp_f32x2){q[2 * sC], q[3 * sC]} * sc;
        const sp_f16x2 ha = __builtin_convertvector(a, sp_f16x2), hb = __builtin_convertvector(b, sp_f16x2);
        const sp_f16x2 la = __builtin_convertvector(a - __builtin_convertvector(ha, sp_f32x2), sp_f16x2);
        const sp_f16x2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, sp_f32x2), sp_f16x2);
        float4 o;
        o.x = __uint_as_float(__builtin_bit_cast(unsigned, ha)); o.y = __uint_as_float(__builtin_bit_cast(unsigned, hb));
        o.z = __uint_as_float(__builtin_bit_cast(unsigned, la)); o.w = __uint_as_float(__builtin_bit_cast(unsigned, lb));
        *reinterpret_cast<float4*>(out + (r * T + t) * (size_t)Cc + 4 * (size_t)c) = o;
    }
}
}  // namespace

int m3t_sgemm_x6_pre_launch(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, const float* bias,
                            int act, int accumulate, int narrow, const unsigned long long* amax_a, const unsigned long long* amax_b, hipStream_t s);

// include/m3t_hip.h: C = act(A B^T + bias) (+ C) on operands split once
extern "C" int m3t_sgemm_pre(int M, int N, int K, const float* A_img, int lda, const float* B_img, int ldb, float* C, int ldc,
                             const float* bias, int act, int accumulate, const unsigned long long* amax_a, const unsigned long long* amax_b,
                             void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (!A_img || !B_img || !C || !amax_a || !amax_b || K <= 0 || !x6_enabled() || !m3t_f16x3_enabled() || M % 128 != 0 || N % 64 != 0 ||
        K % 32 != 0 || lda % 4 != 0 || ldb % 4 != 0 || (uintptr_t)A_img % 16 != 0 || (uintptr_t)B_img % 16 != 0)
        return M3T_EINVAL;
    const int narrow = (N % 128 != 0 || (long long)(M / 128) * (N / 128) <= 384) ? 1 : 0;
    return m3t_sgemm_x6_pre_launch(M, N, K, A_img, lda, B_img, ldb, C, ldc, bias, act, accumulate, narrow, amax_a, amax_b, (hipStream_t)stream);
}

namespace {
// the staged image of a K-contiguous fp16x3 operand w [N][K] (gemm_x6w.hip, BD kernels): [N / 64][K / 8][term][64 rows][8 halves] -- every
// (64-row block, k-octet, term) one contiguous KiB, the unit of an LDS-DMA wave instruction.  One thread per (row, octet).
__global__ __launch_bounds__(256) void f16x3_image_b_kernel(const float* __restrict__ w, unsigned char* __restrict__ img, int N, int K8, size_t ld,
                                                            const unsigned long long* __restrict__ slot) {
    float sc, inv;
    m3t_f16_scale((unsigned)*slot, sc, inv);
    const size_t total = (size_t)N * K8;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i % 64); const size_t q = i / 64;            // consecutive threads: consecutive rows of a block (1 KiB stores)
        const int o = (int)(q % K8); const int rb = (int)(q / K8);
        const float* src = w + ((size_t)rb * 64 + r) * ld + 8 * (size_t)o;
        const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
        const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        unsigned hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const sp_f32x2 a = (sp_f32x2){x[2 * e], x[2 * e + 1]} * sc;
            const sp_f16x2 h = __builtin_convertvector(a, sp_f16x2);
            const sp_f16x2 l = __builtin_convertvector(a - __builtin_convertvector(h, sp_f32x2), sp_f16x2);
            hi[e] = __builtin_bit_cast(unsigned, h); lo[e] = __builtin_bit_cast(unsigned, l);
        }
        unsigned char* dst = img + (((size_t)rb * K8 + o) * 2) * 1024 + (size_t)r * 16;
        *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}
}  // namespace

int m3t_sgemm_x6w_bimg_launch(int M, int N, int K, const float* A, int lda, const float* B_img, float* C, int ldc, const float* bias, int act,
                              int accumulate, float* ws, int splits, int kchunk, const unsigned long long* amax_a,
                              const unsigned long long* amax_b, hipStream_t s);

// include/m3t_hip.h
extern "C" int m3t_f16x3_image_b(const float* w, int N, int K, size_t ld, float* img, const unsigned long long* slot, void* stream) {
    if (N <= 0 || K <= 0) return 0;
    if (!w || !img || !slot || N % 64 != 0 || K % 8 != 0 || ld % 4 != 0 || (uintptr_t)w % 16 != 0 || (uintptr_t)img % 16 != 0) return M3T_EINVAL;
    const size_t total = (size_t)N * (K / 8);
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    f16x3_image_b_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(w, reinterpret_cast<unsigned char*>(img), N, K / 8, ld, slot);
    M3T_LAUNCH_CHECK();
    return 0;
}

// include/m3t_hip.h: C = act(A B^T + bias) (+ C), B given as its staged image
extern "C" int m3t_sgemm_bimg(int M, int N, int K, const float* A, int lda, const float* B_img, float* C, int ldc, const float* bias, int act,
                              int accumulate, float* ws, size_t ws_bytes, const unsigned long long* amax_a, const unsigned long long* amax_b,
                              void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (!A || !B_img || !C || !amax_b || K <= 0 || !x6_enabled() || !m3t_f16x3_enabled() || M % 128 != 0 || N % 256 != 0 || K % 32 != 0 ||
        lda % 4 != 0 || (uintptr_t)A % 16 != 0 || (uintptr_t)B_img % 16 != 0)
        return M3T_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const unsigned long long* use_a = amax_a; const unsigned long long* use_b = amax_b;
    if (!amax_a) {
        const M3TRegion ra{A, (unsigned long long)M, (unsigned long long)lda, K / 4, nullptr};
        const M3TRegion rb{A, 0ull, 0ull, 0, nullptr};
        const int rm = m3t_f16x3_measure(ra, nullptr, rb, amax_b, &use_a, &use_b, s);
        if (rm) return rm;
    }
    const GemmPlan g = plan_gemm(0, M, N, K, 0, true, ws ? ws_bytes : 0, M3T_GEMM_F16X3);
    const int splits = g.wide ? g.splits : 1, kchunk = g.wide ? g.kchunk : K;
    const int rc = m3t_sgemm_x6w_bimg_launch(M, N, K, A, lda, B_img, C, ldc, bias, act, accumulate, ws, splits, kchunk, use_a, use_b, s);
    if (rc) return rc;
    if (splits > 1) { launch_splitk_reduce(ws, C, bias, M, N, ldc, splits, act, accumulate, s); M3T_LAUNCH_CHECK(); }
    return 0;
}

int m3t_sgemm_ring_launch(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                          const float* bias, int act, int accumulate, int seg_len, int seg_stride, int a_off, int b_off, float* ws, int splits,
                          int kchunk, const unsigned long long* amax_a, const unsigned long long* amax_b, int variant, hipStream_t s);

// include/m3t_hip.h: the fp16x3 product of m3t_sgemm_scaled on the 256 x 256 LDS-DMA ring kernels (gemm_ring.hip)
extern "C" int m3t_sgemm_ring(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                              const float* bias, int act, int accumulate, int seg_len, int seg_stride, int a_off, int b_off, float* ws,
                              size_t ws_bytes, int splits, const unsigned long long* amax_a, const unsigned long long* amax_b, int variant,
                              void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (!A || !B || !C || K <= 0 || !m3t_f16x3_enabled() || N % 256 != 0 || K % 16 != 0 || lda % 4 != 0 || ldb % 4 != 0 ||
        (uintptr_t)A % 16 != 0 || (uintptr_t)B % 16 != 0 || splits < 1)
        return M3T_EINVAL;
    if ((transA == 1 && transB == 1) || (transA == 1 && M % 4 != 0)) return M3T_EINVAL;
    if (seg_len > 0 && !(transA == 1 && transB == 0 && seg_len >= 32 && K % seg_len == 0)) return M3T_EINVAL;
    // variants 0-3 compute the product; 6, 11, 19, 35, 43 are the timing-only builds of profiles/r06_ring_gemm_ablation.txt (parts of the loop
    // removed: WRONG results) and need M3T_RING_ABLATIONS=1 in the environment
    if ((variant < 0 || variant > 3) && !getenv("M3T_RING_ABLATIONS")) return M3T_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const unsigned long long* use_a = amax_a; const unsigned long long* use_b = amax_b;
    if (!amax_a || !amax_b) {
        const size_t kr = seg_len > 0 ? (size_t)(K / seg_len - 1) * seg_stride + seg_len : (size_t)K;
        const float* A0 = seg_len > 0 ? A + (size_t)a_off * lda : A;
        const float* B0 = seg_len > 0 ? B + (size_t)b_off * ldb : B;
        const M3TRegion ra{A0, (unsigned long long)(transA ? kr : (size_t)M), (unsigned long long)lda, (transA ? M : K) / 4, nullptr};
        const M3TRegion rb{B0, (unsigned long long)(transB ? (size_t)N : kr), (unsigned long long)ldb, (transB ? K : N) / 4, nullptr};
        const int rm = m3t_f16x3_measure(ra, amax_a, rb, amax_b, &use_a, &use_b, s);
        if (rm) return rm;
    }
    int kchunk = cdiv(cdiv(K, splits), 32) * 32;                 // (the slabs of plan_gemm)
    splits = cdiv(K, kchunk);
    if (splits > 1 && (!ws || ws_bytes < (size_t)splits * M * N * sizeof(float))) return M3T_EINVAL;
    const int rc = m3t_sgemm_ring_launch(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, seg_len, seg_stride, a_off, b_off,
                                         ws, splits, kchunk, use_a, use_b, variant, s);
    if (rc) return rc;
    if (splits > 1) { launch_splitk_reduce(ws, C, bias, M, N, ldc, splits, act, accumulate, s); M3T_LAUNCH_CHECK(); }
    return 0;
}

// include/m3t_hip.h
extern "C" int m3t_f16x3_split_perm(const float* w, size_t R, int T, int Cc, size_t sR, size_t sT, size_t sC, float* out,
                                    const unsigned long long* slot, void* stream) {
    if (R == 0 || T <= 0 || Cc <= 0) return 0;
    if (!w || !out || !slot || Cc % 4 != 0 || (uintptr_t)out % 16 != 0) return M3T_EINVAL;
    const size_t total = R * (size_t)T * (Cc / 4);
    int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    f16x3_split_perm_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(w, out, R, T, Cc, sR, sT, sC, slot);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_f16x3_split(const float* x, size_t rows, int cols, size_t ld, float* out, size_t ldo, const unsigned long long* slot,
                               void* stream) {
    if (rows == 0 || cols <= 0) return 0;
    if (!x || !out || !slot || cols % 4 != 0 || ld % 4 != 0 || ldo % 4 != 0 || (uintptr_t)x % 16 != 0 || (uintptr_t)out % 16 != 0) return M3T_EINVAL;
    const size_t total = rows * (size_t)(cols / 4);
    int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    f16x3_split_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(x, out, rows, cols / 4, ld, ldo, slot);
    M3T_LAUNCH_CHECK();
    return 0;
}

// include/m3t_hip.h: the tap-walk contraction over channels-last grids (the convolutions' data gradient without a patch matrix)
extern "C" int m3t_conv3d_taps(const float* src, const float* w_taps, float* dst, int N, int C_src, int C_dst, int T, int H, int W,
                               int To, int Ho, int Wo, int kt, int kh, int kw, int base_t, int base_h, int base_w, int sign, int flags,
                               const unsigned long long* amax_src, const unsigned long long* amax_w, float* ws, size_t ws_bytes, float* dst_planes,
                               void* stream) {
    if (N <= 0 || C_dst <= 0) return 0;
    if (!src || !w_taps || !dst || C_src <= 0 || T <= 0 || H <= 0 || W <= 0 || To <= 0 || Ho <= 0 || Wo <= 0 || kt <= 0 || kh <= 0 || kw <= 0 ||
        (sign != 1 && sign != -1))
        return M3T_EINVAL;
    const long long rows = (long long)N * T * H * W, srows = (long long)N * To * Ho * Wo;
    if (rows > 0x7fffffffll || srows > 0x7fffffffll || (long long)kt * kh * kw * C_src > 0x7fffffffll) return M3T_EINVAL;
    if (!x6_enabled() || C_dst % 64 != 0 || C_src % 32 != 0 || (uintptr_t)src % 16 != 0 || (uintptr_t)w_taps % 16 != 0 ||
        (flags & (M3T_GEMM_BF16 | M3T_GEMM_HIGH)))
        return M3T_EINVAL;
    const int f16x3 = ((flags & M3T_GEMM_F16X3) && m3t_f16x3_enabled()) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    const unsigned long long* use_a = amax_src; const unsigned long long* use_b = amax_w;
    if (f16x3) {
        const M3TRegion ra{src, (unsigned long long)srows, (unsigned long long)C_src, C_src / 4, nullptr};
        const M3TRegion rb{w_taps, (unsigned long long)kt * kh * kw * C_src, (unsigned long long)C_dst, C_dst / 4, nullptr};
        const int rm = m3t_f16x3_measure(ra, amax_src, rb, amax_w, &use_a, &use_b, s);
        if (rm) return rm;
    }
    int splits, kchunk;
    taps_split(rows, C_dst, kt * kh * kw * C_src, ws, ws_bytes, splits, kchunk);
    const bool direct_planes = dst_planes && splits == 1;
    const int rc = m3t_conv3d_taps_launch(src, w_taps, direct_planes ? dst_planes : dst, N, C_src, C_dst, T, H, W, To, Ho, Wo, kt, kh, kw, base_t,
                                          base_h, base_w, sign, f16x3, 0, use_a, use_b, ws, splits, kchunk, s, nullptr, nullptr, direct_planes);
    if (rc) return rc;
    if (splits > 1) { launch_splitk_reduce(ws, dst, nullptr, (int)rows, C_dst, C_dst, splits, 0, 0, s); M3T_LAUNCH_CHECK(); }
    if (dst_planes && !direct_planes) return m3t_btc_to_bct(dst, dst_planes, N, T * H * W, C_dst, stream);
    return 0;
}

// ... with both operands PRE-SPLIT (m3t_f16x3_split under the SAME slots) and the weights K-contiguous: w_img[cd][(tap, cs)]
extern "C" int m3t_conv3d_taps_pre(const float* src_img, const float* w_img, float* dst, int N, int C_src, int C_dst, int T, int H, int W,
                                   int To, int Ho, int Wo, int kt, int kh, int kw, int base_t, int base_h, int base_w, int sign,
                                   const unsigned long long* amax_src, const unsigned long long* amax_w, float* ws, size_t ws_bytes,
                                   float* dst_planes, void* stream) {
    if (N <= 0 || C_dst <= 0) return 0;
    if (!src_img || !w_img || !dst || !amax_src || !amax_w || C_src <= 0 || T <= 0 || H <= 0 || W <= 0 || To <= 0 || Ho <= 0 || Wo <= 0 ||
        kt <= 0 || kh <= 0 || kw <= 0 || (sign != 1 && sign != -1))
        return M3T_EINVAL;
    const long long rows = (long long)N * T * H * W, srows = (long long)N * To * Ho * Wo;
    if (rows > 0x7fffffffll || srows > 0x7fffffffll || (long long)kt * kh * kw * C_src > 0x7fffffffll) return M3T_EINVAL;
    if (!x6_enabled() || !m3t_f16x3_enabled() || C_dst % 64 != 0 || C_src % 32 != 0 || (uintptr_t)src_img % 16 != 0 ||
        (uintptr_t)w_img % 16 != 0)
        return M3T_EINVAL;
    int splits, kchunk;
    taps_split(rows, C_dst, kt * kh * kw * C_src, ws, ws_bytes, splits, kchunk);
    const bool direct_planes = dst_planes && splits == 1;
    const int rc = m3t_conv3d_taps_launch(src_img, w_img, direct_planes ? dst_planes : dst, N, C_src, C_dst, T, H, W, To, Ho, Wo, kt, kh, kw, base_t,
                                          base_h, base_w, sign, 1, 1, amax_src, amax_w, ws, splits, kchunk, (hipStream_t)stream, nullptr, nullptr,
                                          direct_planes);
    if (rc) return rc;
    if (splits > 1) { launch_splitk_reduce(ws, dst, nullptr, (int)rows, C_dst, C_dst, splits, 0, 0, (hipStream_t)stream); M3T_LAUNCH_CHECK(); }
    if (dst_planes && !direct_planes) return m3t_btc_to_bct(dst, dst_planes, N, T * H * W, C_dst, stream);
    return 0;
}

// include/m3t_hip.h: the forward convolution as the tap walk on operands split once
extern "C" int m3t_conv3d_fwd_taps(const float* x_img, const float* w_img, const float* bias, float* y_cl, int N, int Ci, int Co, int T, int H,
                                   int W, int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw,
                                   const unsigned long long* amax_x, const unsigned long long* amax_w, float* ws, size_t ws_bytes, float* y_planes,
                                   void* stream) {
    if (N <= 0 || Co <= 0) return 0;
    if (!x_img || !w_img || !y_cl || !amax_x || !amax_w || Ci <= 0 || T <= 0 || H <= 0 || W <= 0 || kt <= 0 || kh <= 0 || kw <= 0 || st <= 0 ||
        sh <= 0 || sw <= 0 || pt < 0 || ph < 0 || pw < 0)
        return M3T_EINVAL;
    const int To = (T + 2 * pt - kt) / st + 1, Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    if (T + 2 * pt < kt || H + 2 * ph < kh || W + 2 * pw < kw) return M3T_EINVAL;
    const long long rows = (long long)N * To * Ho * Wo, srows = (long long)N * T * H * W;
    if (rows > 0x7fffffffll || srows > 0x7fffffffll || (long long)kt * kh * kw * Ci > 0x7fffffffll) return M3T_EINVAL;
    if (!x6_enabled() || !m3t_f16x3_enabled() || Co % 64 != 0 || Ci % 32 != 0 || (uintptr_t)x_img % 16 != 0 ||
        (uintptr_t)w_img % 16 != 0)
        return M3T_EINVAL;
    int splits, kchunk;
    taps_split(rows, Co, kt * kh * kw * Ci, ws, ws_bytes, splits, kchunk);
    const int stride3[3] = {st, sh, sw};
    hipStream_t s = (hipStream_t)stream;
    const bool direct_planes = y_planes && splits == 1;
    const int rc = m3t_conv3d_taps_launch(x_img, w_img, direct_planes ? y_planes : y_cl, N, Ci, Co, To, Ho, Wo, T, H, W, kt, kh, kw, -pt, -ph, -pw, 1,
                                          1, 1, amax_x, amax_w, ws, splits, kchunk, s, stride3, bias, direct_planes);
    if (rc) return rc;
    if (splits > 1) { launch_splitk_reduce(ws, y_cl, bias, (int)rows, Co, Co, splits, 0, 0, s); M3T_LAUNCH_CHECK(); }
    if (y_planes && !direct_planes) return m3t_btc_to_bct(y_cl, y_planes, N, To * Ho * Wo, Co, stream);
    return 0;
}

// include/m3t_hip.h: the first layers (C_in <= 4) on the four-channel image
extern "C" int m3t_conv3d_fwd_taps4(const float* x_img4, const float* w_img, const float* bias, float* y_cl, int N, int Co, int T, int H, int W,
                                    int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, const unsigned long long* amax_x,
                                    const unsigned long long* amax_w, float* ws, size_t ws_bytes, float* y_planes, void* stream) {
    if (N <= 0 || Co <= 0) return 0;
    if (!x_img4 || !w_img || !y_cl || !amax_x || !amax_w || T <= 0 || H <= 0 || W <= 0 || kt <= 0 || kh <= 0 || kw <= 0 || kw > 8 || st <= 0 ||
        sh <= 0 || sw <= 0 || pt < 0 || ph < 0 || pw < 0 || T + 2 * pt < kt || H + 2 * ph < kh || W + 2 * pw < kw)
        return M3T_EINVAL;
    const int To = (T + 2 * pt - kt) / st + 1, Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    const long long rows = (long long)N * To * Ho * Wo, srows = (long long)N * T * H * W;
    if (rows > 0x7fffffffll || srows > 0x7fffffffll || (long long)kt * kh * 32 > 0x7fffffffll) return M3T_EINVAL;
    if (!x6_enabled() || !m3t_f16x3_enabled() || Co % 64 != 0 || (uintptr_t)x_img4 % 16 != 0 || (uintptr_t)w_img % 16 != 0)
        return M3T_EINVAL;
    int splits, kchunk;
    taps_split(rows, Co, kt * kh * 32, ws, ws_bytes, splits, kchunk);
    const int stride3[3] = {st, sh, sw};
    hipStream_t s = (hipStream_t)stream;
    const bool direct_planes = y_planes && splits == 1;
    const int rc = m3t_conv3d_taps4_launch(x_img4, w_img, bias, direct_planes ? y_planes : y_cl, N, Co, T, H, W, To, Ho, Wo, kt, kh, kw, stride3, pt,
                                           ph, pw, amax_x, amax_w, ws, splits, kchunk, s, direct_planes);
    if (rc) return rc;
    if (splits > 1) { launch_splitk_reduce(ws, y_cl, bias, (int)rows, Co, Co, splits, 0, 0, s); M3T_LAUNCH_CHECK(); }
    if (y_planes && !direct_planes) return m3t_btc_to_bct(y_cl, y_planes, N, To * Ho * Wo, Co, stream);
    return 0;
}

// include/m3t_hip.h: the weight gradient as the walk turned round: dwt[(tap, ci)][co], taps * Ci rounded up to 128 rows
extern "C" int m3t_conv3d_wgrad_taps(const float* x_cl, const float* dy_cl, float* dwt, int N, int Ci, int Co, int T, int H, int W, int kt, int kh,
                                     int kw, int st, int sh, int sw, int pt, int ph, int pw, int flags, const unsigned long long* amax_x,
                                     const unsigned long long* amax_dy, float* ws, size_t ws_bytes, void* stream) {
    if (Ci <= 0 || Co <= 0) return 0;
    if (!x_cl || !dy_cl || !dwt || N <= 0 || T <= 0 || H <= 0 || W <= 0 || kt <= 0 || kh <= 0 || kw <= 0 || st <= 0 || sh <= 0 || sw <= 0 ||
        pt < 0 || ph < 0 || pw < 0 || T + 2 * pt < kt || H + 2 * ph < kh || W + 2 * pw < kw)
        return M3T_EINVAL;
    const int To = (T + 2 * pt - kt) / st + 1, Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    const long long rows = (long long)N * To * Ho * Wo, srows = (long long)N * T * H * W, Kc = (long long)kt * kh * kw * Ci;
    if (rows > 0x7fffffffll || srows > 0x7fffffffll || Kc > 0x7fffff00ll) return M3T_EINVAL;
    if (!x6_enabled() || Co % 64 != 0 || Ci % 4 != 0 || (uintptr_t)x_cl % 16 != 0 || (uintptr_t)dy_cl % 16 != 0 ||
        (uintptr_t)dwt % 16 != 0 || (flags & (M3T_GEMM_BF16 | M3T_GEMM_HIGH)))
        return M3T_EINVAL;
    const int f16x3 = ((flags & M3T_GEMM_F16X3) && m3t_f16x3_enabled()) ? 1 : 0;
    // M3T_CONV_IMAGES: x_cl and dy_cl are the m3t_f16x3_split images of the operands under amax_x / amax_dy (the forward pass and the data
    // gradient made them already): the loop re-pairs halves instead of converting -- each x value is staged 27 x C_out / 128 times
    const int pre = (flags & M3T_CONV_IMAGES) ? 1 : 0;
    if (pre && (!f16x3 || !amax_x || !amax_dy)) return M3T_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const unsigned long long* use_a = amax_x; const unsigned long long* use_b = amax_dy;
    if (f16x3 && !pre) {
        const M3TRegion ra{x_cl, (unsigned long long)srows, (unsigned long long)Ci, Ci / 4, nullptr};
        const M3TRegion rb{dy_cl, (unsigned long long)rows, (unsigned long long)Co, Co / 4, nullptr};
        const int rm = m3t_f16x3_measure(ra, amax_x, rb, amax_dy, &use_a, &use_b, s);
        if (rm) return rm;
    }
    // split-K over the rows: the output is a few tiles, the reduction hundreds of thousands of rows deep
    const int Mp = (int)((Kc + 127) / 128 * 128);
    const long long tiles = (long long)(Mp / 128) * ((Co + 127) / 128);
    int splits = 1, kchunk = (int)rows;
    if (ws) {
        long long sp = (1536 + tiles - 1) / tiles;
        const long long cap = (long long)(ws_bytes / ((size_t)Mp * Co * sizeof(float)));
        if (sp > cap) sp = cap;
        if (sp > rows / 256) sp = rows / 256;
        if (sp >= 2) {
            kchunk = cdiv(cdiv((int)rows, (int)sp), 32) * 32;
            splits = cdiv((int)rows, kchunk);
        }
    }
    const int stride3[3] = {st, sh, sw};
    const int rc = m3t_conv3d_wgrad_launch(x_cl, dy_cl, dwt, N, Ci, Co, T, H, W, To, Ho, Wo, kt, kh, kw, stride3, pt, ph, pw, f16x3, use_a, use_b, ws,
                                           splits, kchunk, s, pre);
    if (rc) return rc;
    if (splits > 1) { launch_splitk_reduce(ws, dwt, nullptr, Mp, Co, Co, splits, 0, 0, s); M3T_LAUNCH_CHECK(); }
    return 0;
}

extern "C" int m3t_colsum(const float* X, int M, int N, int ld, float* out, int accumulate, float* ws, size_t ws_bytes,
                          void* stream) {
    if (N <= 0) return 0;
    if (!X || !out) return M3T_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int chunks = cdiv(M, 128);
    if (chunks > 128) chunks = 128;
    if (!ws || ws_bytes < (size_t)chunks * N * sizeof(float) || chunks <= 1) {
        colsum_kernel<<<cdiv(N, 32), 256, 0, s>>>(X, M, N, ld, out, accumulate);
        M3T_LAUNCH_CHECK();
        return 0;
    }
    // tall matrix: per-row-chunk partials, then a fixed-order sum of the partials
    const int rows_per = cdiv(M, chunks);
    colsum_partial_kernel<<<dim3(cdiv(N, 32), chunks), 256, 0, s>>>(X, M, N, ld, ws, rows_per);
    M3T_LAUNCH_CHECK();
    colsum_kernel<<<cdiv(N, 32), 256, 0, s>>>(ws, chunks, N, N, out, accumulate);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_transpose(const float* src, int R, int C, int lds_, float* dst, int ldd, void* stream) {
    if (R <= 0 || C <= 0) return 0;
    transpose_kernel<<<dim3(cdiv(C, 32), cdiv(R, 32)), 256, 0, (hipStream_t)stream>>>(src, R, C, lds_, dst, ldd);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_relu_bwd(const float* a, float* dy, size_t n, void* stream) {
    if (n == 0) return 0;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    relu_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(a, dy, n);
    M3T_LAUNCH_CHECK();
    return 0;
}

static thread_local unsigned long long* g_amax_out = nullptr;
unsigned long long* m3t_take_amax_out() { unsigned long long* p = g_amax_out; g_amax_out = nullptr; return p; }
extern "C" int m3t_amax_out(unsigned long long* slot) { g_amax_out = slot; return 0; }

extern "C" int m3t_mask_pos(const float* s, const float* dy, const float* mul, float* out, size_t n, void* stream) {
    unsigned long long* amax = m3t_take_amax_out();
    if (n == 0) return 0;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    mask_pos_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(s, dy, mul, out, n, amax);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_add_relu(const float* a, const float* b, float* out, size_t n, void* stream) {
    unsigned long long* amax = m3t_take_amax_out();
    if (n == 0) return 0;
    if (!a || !b || !out) return M3T_EINVAL;
    const int vec = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) % 16 == 0) ? 1 : 0;
    size_t blocks = ((vec ? n / 4 : n) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    add_relu_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(a, b, out, n, vec, amax);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_mask_pos_drop(const float* s, const float* dy, float* out, int rows, int C, float drop_p,
                                 unsigned long long drop_seed, void* stream) {
    unsigned long long* amax = m3t_take_amax_out();
    if (rows <= 0 || C <= 0) return 0;
    if (!s || !dy || !out || drop_p < 0.f || drop_p >= 1.f) return M3T_EINVAL;
    const size_t total = (size_t)((rows + 3) / 4) * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    mask_pos_drop_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(s, dy, out, rows, C, m3t_make_drop(drop_p, drop_seed), amax);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_version(void) { return 1; }

extern "C" int m3t_device_arch(char* arch, int cap) {
    hipDeviceProp_t prop;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return (int)e;
    if (arch && cap > 0) {
        int i = 0;
        for (; i < cap - 1 && prop.gcnArchName[i]; ++i) arch[i] = prop.gcnArchName[i];
        arch[i] = 0;
    }
    return 0;
}
