// Micro-probe: per-launch cost of a workgroup streaming N KB (a) from its own stable slice, (b) from one
// slice shared by all workgroups, (c) from a buffer other workgroups rewrote in the previous launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NL>
__global__ __launch_bounds__(512) void probe(const float4* __restrict__ buf, float* __restrict__ out, int f4_per_block, int mode,
                                              float4* __restrict__ wbuf, int wf4) {
    const float4* p = buf + (mode == 1 ? 0 : (size_t)blockIdx.x * f4_per_block);
    float4 acc = make_float4(0, 0, 0, 0);
    float4 v[NL + 1];
#pragma unroll
    for (int n = 0; n < NL; ++n) v[n] = p[threadIdx.x + n * 512];
#pragma unroll
    for (int i = 0; i < NL; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    if (wbuf) {   // each block rewrites its piece of the shared buffer
        for (int i = threadIdx.x; i < wf4; i += 512) wbuf[(size_t)blockIdx.x * wf4 + i] = acc;
    }
    if (acc.x == 1234.5f) out[blockIdx.x] = acc.y + acc.z + acc.w;
}

int main() {
    const int NB = 256, MAXKB = 192;
    float4 *buf, *wb; float* out;
    CK(hipMalloc(&buf, (size_t)NB * MAXKB * 1024));
    CK(hipMalloc(&wb, 1 << 20));
    CK(hipMalloc(&out, NB * 4));
    CK(hipMemset(buf, 0, (size_t)NB * MAXKB * 1024));
    CK(hipMemset(wb, 0, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int T = 300;
    for (int blocks : {64, 160, 256}) {
        for (int kb : {0, 16, 40, 80, 160}) {
            for (int mode : {0, 1, 2}) {
                if (kb == 0 && mode) continue;
                int f4 = kb * 64;
                // mode 2: all blocks read the same 64 KB-ish buffer that the previous launch rewrote piecewise
                const float4* src = mode == 2 ? wb : buf;
                int m = mode == 2 ? 1 : mode;
                float4* w = mode == 2 ? wb : nullptr;
                int wf4 = mode == 2 ? (f4 + blocks - 1) / blocks : 0;
                if (mode == 2 && kb > 64) continue;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    for (int t = 0; t < T; ++t) {
                        switch (kb) {
                            case 0: probe<0><<<blocks, 512>>>(src, out, f4, m, w, wf4); break;
                            case 16: probe<2><<<blocks, 512>>>(src, out, f4, m, w, wf4); break;
                            case 40: probe<5><<<blocks, 512>>>(src, out, f4, m, w, wf4); break;
                            case 80: probe<10><<<blocks, 512>>>(src, out, f4, m, w, wf4); break;
                            default: probe<20><<<blocks, 512>>>(src, out, f4, m, w, wf4); break;
                        }
                    }
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                }
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("blocks %3d  %3d KB/block  mode %d : %6.2f us/launch  (%.1f GB/s per CU)\n", blocks, kb, mode, ms / T * 1e3,
                       kb ? kb * 1024.0 / (ms / T * 1e-3) / 1e9 : 0.0);
            }
        }
    }
    return 0;
}
