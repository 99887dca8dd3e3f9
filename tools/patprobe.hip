// Load-pattern probe: a 512-thread workgroup pulls a [R rows x 512 floats] fp32 panel (row stride LD floats)
// into registers with float4 loads under three lane->address maps, REP panels per launch.
//  P0: fully coalesced (lane i -> 16 B at offset 16*i)
//  P1: MFMA 16x16x4 operand map: row = lane&15, k = 16*chunk + 4*(lane>>4); wave w takes chunks w, w+8, ...
//  P2: same map, but a wave takes chunk PAIRS (2w, 2w+1), (2w+16, ...): both halves of a 128-B line from one wave
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int P, int REP>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ buf, float* __restrict__ out, int LD, size_t panel_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int rep = 0; rep < REP; ++rep) {
        const float* base = buf + ((size_t)blockIdx.x * REP + rep) * panel_stride;
        float4 v[20];
#pragma unroll
        for (int n = 0; n < 20; ++n) {     // 80 rows x 512 floats = 160 KB = 5 row-tiles x 32 chunks; wave handles 4 chunks x 5 tiles
            const int tile = n / 4, i = n % 4;
            if (P == 0) {
                v[n] = *reinterpret_cast<const float4*>(base + ((size_t)(n * 512 + threadIdx.x)) * 4);
            } else {
                const int c = P == 1 ? wave + 8 * i : (i >> 1) * 16 + 2 * wave + (i & 1);
                const int row = tile * 16 + (lane & 15);
                v[n] = *reinterpret_cast<const float4*>(base + (size_t)row * LD + c * 16 + (lane >> 4) * 4);
            }
        }
#pragma unroll
        for (int n = 0; n < 20; ++n) { acc.x += v[n].x; acc.y += v[n].y; acc.z += v[n].z; acc.w += v[n].w; }
    }
    if (acc.x == 1234.5f) out[blockIdx.x] = acc.y + acc.z + acc.w;
}

template <int P, int REP>
float run(const float* buf, float* out, int blocks, int LD, size_t ps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int T = 200;
    for (int r = 0; r < 2; ++r) {
        hipEventRecord(e0);
        for (int t = 0; t < T; ++t) probe<P, REP><<<blocks, 512>>>(buf, out, LD, ps);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / T * 1e3;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    float* buf; float* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4096)); CK(hipMemset(buf, 0, bytes));
    for (int blocks : {64, 256}) {
        const size_t ps = 80 * 1024;   // floats per panel (>= 80 rows * 1024 LD)
        printf("blocks %d: REP=1  P0 %.2f  P1(LD512) %.2f  P2(LD512) %.2f  P1(LD1024) %.2f  P2(LD1024) %.2f us\n", blocks,
               run<0, 1>(buf, out, blocks, 512, ps), run<1, 1>(buf, out, blocks, 512, ps), run<2, 1>(buf, out, blocks, 512, ps),
               run<1, 1>(buf, out, blocks, 1024, ps), run<2, 1>(buf, out, blocks, 1024, ps));
        printf("blocks %d: REP=4  P0 %.2f  P1(LD512) %.2f  P2(LD512) %.2f  P1(LD1024) %.2f  P2(LD1024) %.2f us\n", blocks,
               run<0, 4>(buf, out, blocks, 512, ps), run<1, 4>(buf, out, blocks, 512, ps), run<2, 4>(buf, out, blocks, 512, ps),
               run<1, 4>(buf, out, blocks, 1024, ps), run<2, 4>(buf, out, blocks, 1024, ps));
    }
    return 0;
}
