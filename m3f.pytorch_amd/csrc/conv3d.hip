// Patch matrix (im2col) of a Conv3d input for the weight-gradient GEMM dW = dy^T P (m3t.ops.conv3d; reference
// models/backbone.py:73-103,179-271,327-332: the 3-D conv stems).  Round 3 built P with torch (unfold x 3, permute, reshape): the
// strided copy of an 8-D view ran as ~50 `direct_copy` launches per convolution (344 per C5 step, 8.2 % of it) and the fp16x3 GEMM then
// measured the matrix's magnitude with one more pass (112 `f16x3_absmax` launches, 3.4 %).  ONE launch here writes P -- rows (n, t', h',
// w'), columns (c_in, kt, kh, kw), zero columns up to Kp and zero rows up to rows_pad for the GEMM's tiles, zero padding of the
// convolution itself -- as 16-byte stores, and raises the operand's magnitude slot on the way.
#include "common.h"

namespace {

struct Im2colArgs {
    const float* x;
    float* out;
    unsigned long long* slot;
    int Ci, T, H, W, kt, kh, kw, st, sh, sw, pt, ph, pw, To, Ho, Wo, Kc, Kp;
    long long rows, rows_pad;
};

// A block takes 64 consecutive output positions (rows of P) per pass and walks the columns 64 at a time through an LDS tile:
//   load   lane = ROW, wave w takes columns c0 + w + 4 j: for one column the 64 lanes read 64 consecutive output positions, i.e.
//          (stride-1 convolutions) 64 consecutive floats of x -- two or three cache lines per wave instruction.  Round 4's first
//          version put consecutive lanes on consecutive COLUMNS of one row: every lane in another kw-run of x, ~64 lines per load
//          instruction, four of them per 16-byte store: the kernel ran at the address path's rate (2.9 TB/s of stores), not HBM's;
//   store  out of the tile, transposed: 16-byte stores, a wave writes four rows x 64 columns (256 B contiguous per row).
// Column table (per block, LDS): offset of (ci, dt, dh, dw) inside a sample and the packed (dt, dh, dw) -- broadcast reads in the load
// phase (all lanes of a wave share the column); a row's origin lives in its lane's registers.  The operand's magnitude slot is raised
// on the way (block maximum, one 64-bit atomic max per block that exceeds what the slot holds).
// The grid is 2-D: blockIdx.y owns `cw` consecutive columns (a multiple of 64; its share of the column table is all the LDS it needs),
// blockIdx.x strides over the row blocks -- the deep layers have few output positions and many columns (512 -> 512 on a 3 x 3 map: 8
// row blocks x 216 column tiles), the first layers the opposite.
constexpr int IM_TILE = 64;
__global__ __launch_bounds__(256) void im2col3d_kernel(Im2colArgs a, int cw) {
    extern __shared__ int cm[];                                  // [cw] offsets, [cw] packed (dt, dh, dw) (or -1: padding column), the tile
    int* coff = cm;
    int* cpos = cm + cw;
    float (*tile)[IM_TILE + 1] = reinterpret_cast<float (*)[IM_TILE + 1]>(cm + 2 * cw);      // [column][row]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int khw = a.kh * a.kw, kvol = a.kt * khw;
    const int cbeg = blockIdx.y * cw, cend = min(a.Kp, cbeg + cw);
    for (int cl = tid; cbeg + cl < cend; cl += 256) {
        const int c = cbeg + cl;
        if (c < a.Kc) {
            const int ci = c / kvol, rem = c - ci * kvol;
            const int dt = rem / khw, rem2 = rem - dt * khw;
            const int dh = rem2 / a.kw, dw = rem2 - dh * a.kw;
            coff[cl] = ((ci * a.T + dt) * a.H + dh) * a.W + dw;
            cpos[cl] = (dt << 20) | (dh << 10) | dw;
        } else { coff[cl] = 0; cpos[cl] = -1; }
    }
    const long long sample = (long long)a.Ci * a.T * a.H * a.W;
    const int hw = a.Ho * a.Wo;
    float mx = 0.f;
    for (long long row0 = (long long)blockIdx.x * IM_TILE; row0 < a.rows_pad; row0 += (long long)gridDim.x * IM_TILE) {
        // this lane's row: origin of its patch (a padding row fails every range check)
        const long long row = row0 + lane;
        int t0 = -(1 << 28), h0 = 0, w0 = 0;
        long long rbase = 0;
        if (row < a.rows) {
            const long long nt = row / hw;
            const int r2 = (int)(row - nt * hw);
            const int ho = r2 / a.Wo, wo = r2 - ho * a.Wo;
            const int n = (int)(nt / a.To), to = (int)(nt - (long long)n * a.To);
            t0 = to * a.st - a.pt; h0 = ho * a.sh - a.ph; w0 = wo * a.sw - a.pw;
            rbase = (long long)n * sample + ((long long)t0 * a.H + h0) * a.W + w0;
        }
        const float* xb = a.x + rbase;
        for (int c0 = cbeg; c0 < cend; c0 += IM_TILE) {
            __syncthreads();                                     // (column table ready; the previous tile has been stored)
            float v[IM_TILE / 4];                                // all sixteen loads of the chunk in flight
#pragma unroll
            for (int j = 0; j < IM_TILE / 4; ++j) {
                const int c = c0 + wave + 4 * j;
                v[j] = 0.f;
                if (c < cend) {
                    const int p = cpos[c - cbeg];
                    const int t = t0 + (p >> 20), h = h0 + ((p >> 10) & 1023), w = w0 + (p & 1023);
                    if (p >= 0 && (unsigned)t < (unsigned)a.T && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W) v[j] = xb[coff[c - cbeg]];
                }
            }
#pragma unroll
            for (int j = 0; j < IM_TILE / 4; ++j) tile[wave + 4 * j][lane] = v[j];
            __syncthreads();
            const int q = tid & 15, rr = tid >> 4;               // this thread's four columns 4 q .. 4 q + 3 of rows rr + 16 k
            if (c0 + 4 * q < a.Kp) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = rr + 16 * k;
                    if (row0 + r < a.rows_pad) {
                        const float4 v = make_float4(tile[4 * q][r], tile[4 * q + 1][r], tile[4 * q + 2][r], tile[4 * q + 3][r]);
                        *reinterpret_cast<float4*>(a.out + (size_t)(row0 + r) * a.Kp + c0 + 4 * q) = v;
                        mx = fmaxf(fmaxf(mx, m3t_fin_abs(v.x)), fmaxf(m3t_fin_abs(v.y), fmaxf(m3t_fin_abs(v.z), m3t_fin_abs(v.w))));
                    }
                }
            }
        }
    }
    if (a.slot) {
        __shared__ float red[4];
        mx = wave_max(mx);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) {
            const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            const unsigned long long bits = (unsigned long long)__float_as_uint(m);      // (caller-owned slot: epoch 0)
            if (bits > __hip_atomic_load(a.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.slot, bits);
        }
    }
}

// x [N][C][S] (C <= 4 channel planes) -> out [N][S][4], missing channels zero: the channels-last image the first layers' tap walk reads
// (gemm_x6.hip, C3 = 3).  One position per thread: C coalesced plane reads, one 16-byte store.  Raises the pending magnitude slot.
__global__ __launch_bounds__(256) void planes_to_cl4_kernel(const float* __restrict__ x, float* __restrict__ out, int C, long long S, long long total,
                                                            unsigned long long* slot) {
    float mx = 0.f;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / S, sp = i - n * S;
        const float* q = x + (n * C) * S + sp;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = q[0];
        if (C > 1) v.y = q[S];
        if (C > 2) v.z = q[2 * S];
        if (C > 3) v.w = q[3 * S];
        mx = fmaxf(mx, fmaxf(fmaxf(m3t_fin_abs(v.x), m3t_fin_abs(v.y)), fmaxf(m3t_fin_abs(v.z), m3t_fin_abs(v.w))));
        *reinterpret_cast<float4*>(out + 4 * i) = v;
    }
    if (slot) {
        __shared__ float red4[4];
        m3t_block_raise_slot(slot, mx, red4);
    }
}

}  // namespace

extern "C" int m3t_planes_to_cl4(const float* x, float* out, int N, int C, long long S, void* stream) {
    unsigned long long* amax = m3t_take_amax_out();
    if (N <= 0 || S <= 0) return 0;
    if (!x || !out || C < 1 || C > 4 || ((uintptr_t)out % 16) != 0) return M3T_EINVAL;
    const long long total = (long long)N * S;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    planes_to_cl4_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(x, out, C, S, total, amax);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_im2col3d(const float* x, int N, int Ci, int T, int H, int W, int kt, int kh, int kw, int st, int sh, int sw,
                            int pt, int ph, int pw, float* out, long long rows_pad, int Kp, unsigned long long* amax_slot, void* stream) {
    if (!x || !out || N <= 0 || Ci <= 0 || kt <= 0 || kh <= 0 || kw <= 0 || st <= 0 || sh <= 0 || sw <= 0 || Kp % 4 != 0 ||
        ((uintptr_t)out % 16) != 0)
        return M3T_EINVAL;
    Im2colArgs a;
    a.x = x; a.out = out; a.slot = amax_slot;
    a.Ci = Ci; a.T = T; a.H = H; a.W = W; a.kt = kt; a.kh = kh; a.kw = kw; a.st = st; a.sh = sh; a.sw = sw; a.pt = pt; a.ph = ph; a.pw = pw;
    a.To = (T + 2 * pt - kt) / st + 1; a.Ho = (H + 2 * ph - kh) / sh + 1; a.Wo = (W + 2 * pw - kw) / sw + 1;
    if (a.To <= 0 || a.Ho <= 0 || a.Wo <= 0) return M3T_EINVAL;
    a.Kc = Ci * kt * kh * kw; a.Kp = Kp;
    a.rows = (long long)N * a.To * a.Ho * a.Wo; a.rows_pad = rows_pad;
    if (Kp < a.Kc || rows_pad < a.rows) return M3T_EINVAL;
    if (kt >= 1024 || kh >= 1024 || kw >= 1024) return M3T_EINVAL;      // (packed kernel offsets)
    // about 8 x 256 workgroups: row blocks first, then column tiles per workgroup down to one
    const long long rblocks = (rows_pad + IM_TILE - 1) / IM_TILE;
    const int ctiles = (Kp + IM_TILE - 1) / IM_TILE;
    int per = (int)((rblocks * ctiles + 2047) / 2048);          // column tiles per workgroup
    per = per < 1 ? 1 : (per > ctiles ? ctiles : per);
    if (per > 32) per = 32;                                      // (<= 16 KB of column table)
    const int cw = per * IM_TILE, ygrid = (ctiles + per - 1) / per;
    long long xgrid = rblocks;
    if (xgrid * ygrid > 4096) xgrid = (4096 + ygrid - 1) / ygrid;
    const size_t dyn = (size_t)cw * 8 + (size_t)IM_TILE * (IM_TILE + 1) * sizeof(float);
    im2col3d_kernel<<<dim3((unsigned)xgrid, (unsigned)ygrid), 256, dyn, (hipStream_t)stream>>>(a, cw);
    M3T_LAUNCH_CHECK();
    return 0;
}
