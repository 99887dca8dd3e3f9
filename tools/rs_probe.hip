// Exchange pattern probe for the persistent scans: per step, a group of 32 workgroups (8 groups = 256 workgroups, memory-side
// sc1 stores / loads, tags inside the granules, two slots) exchanges either
//   P0  ALL-GATHER, 8-B granules  {value, tag}:       publish 2 KB,  gather 64 KB  (the forward scans)
//   P1  ALL-GATHER, 16-B granules {v, v, v, tag}:     publish 4 KB,  gather 128 KB (the backward scans today)
//   P2  REDUCE-SCATTER, 8-B granules:                 publish 64 KB (2 KB to each of 32 peers), gather 64 KB (2 KB from each)
//   P3  REDUCE-SCATTER, 16-B granules of 3 values:    publish 48 KB, gather 48 KB
// P2 / P3 are what a K-split ("row-parallel") backward product would exchange: every workgroup multiplies its own 48 gate
// gradients by its 48 x 512 slice of W_hh and sends each peer the partial sums of that peer's 16 units.
//   hipcc -O3 --offload-arch=gfx950 tools/rs_probe.hip -o tools/bin/rs_probe && tools/bin/rs_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int SPIN_LIMIT = 1 << 16;

// bytes one workgroup gathers per step / publishes per step
template <int P> struct Pat;
template <> struct Pat<0> { static constexpr int NL = 8, NS = 0, GS = 32; };      // NL dwordx4 loads per lane; NS dwordx4 stores per lane (0: 256 lanes x 8 B)
template <> struct Pat<1> { static constexpr int NL = 16, NS = 0, GS = 32; };
template <> struct Pat<2> { static constexpr int NL = 8, NS = 8, GS = 32; };
template <> struct Pat<3> { static constexpr int NL = 6, NS = 6, GS = 32; };
// P4: ALL-GATHER, 16-B granules, groups of 16 workgroups (8 rows x 32 units per workgroup): publish 4 KB, gather 64 KB
template <> struct Pat<4> { static constexpr int NL = 8, NS = 0, GS = 16; };
// P5: the same for the forward scans: 8-B granules, groups of 16: publish 2 KB, gather 32 KB
template <> struct Pat<5> { static constexpr int NL = 4, NS = 0, GS = 16; };

template <int P>
__global__ __launch_bounds__(512) void probe(u32x4* gran, int T, int delay, int* err, float* out) {
    constexpr int NL = Pat<P>::NL, NS = Pat<P>::NS, GS = Pat<P>::GS;
    __shared__ float red[8][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int group = blockIdx.x / GS, member = blockIdx.x % GS;
    const bool g16 = (P == 1 || P == 3 || P == 4);                 // 16-B granules: tag in .w; else two 8-B granules per dwordx4: tags in .y and .w
    // region of (slot, group, destination): NL * 512 dwordx4.  all-gather: one destination ("everyone"): producer m owns dwordx4s
    // [m * NL * 16, +NL * 16); reduce-scatter: destination d's region holds, from source m, dwordx4s [m * NL * 16, +NL * 16)
    const size_t per_dst = (size_t)NL * 512;
    const size_t ndst = NS ? GS : 1;
    constexpr int NGRP = 256 / GS, PUB = NL * 512 / GS;      // groups in the launch; dwordx4s a producer publishes (all-gather)
    float keep = 0.f;
    bool dead = false;
    for (int t = 0; t < T && !dead; ++t) {
        const unsigned tag = (unsigned)t + 1u;
        const u32x4* src = gran + (((size_t)(t & 1) * NGRP + group) * ndst + (NS ? member : 0)) * per_dst;
        u32x4 w[NL];
        int spins = 0;
        for (int z = 0; z < delay; ++z) __builtin_amdgcn_s_sleep(1);
        for (;;) {
#pragma unroll
            for (int c = 0; c < NL; ++c) {
                const u32x4* q = src + (size_t)c * 512 + tid;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w[c]) : "v"(q) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bool ok = true;
#pragma unroll
            for (int c = 0; c < NL; ++c) { asm volatile("" : "+v"(w[c])); ok = ok && w[c].w == tag && (g16 || w[c].y == tag); }
            if (__all(ok)) break;
            if (++spins > SPIN_LIMIT) { dead = true; if (lane == 0) atomicExch(err, t + 1); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NL; ++c) s += __uint_as_float(w[c].x) + __uint_as_float(w[c].z);
        red[wave][lane] = s;
        __syncthreads();
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) r += red[k][lane];
        r = r * 1e-4f + 1.0f;
        keep += r;
        const unsigned nt = tag + 1u;
        u32x4 g;
        g.x = __float_as_uint(r); g.y = g16 ? __float_as_uint(r) : nt; g.z = __float_as_uint(r); g.w = nt;
        if (NS == 0) {
            // all-gather: this workgroup's NL * 16 dwordx4s (8-B: 256 granules = 128 dwordx4; 16-B: 256 granules = 256 dwordx4)
            u32x4* dst = gran + (((size_t)((t + 1) & 1) * NGRP + group)) * per_dst + (size_t)member * PUB;
            if (tid < PUB) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + tid), "v"(g) : "memory");
        } else {
            // reduce-scatter: NS dwordx4 per lane = NS * 512 in all = 32 destinations x (NS * 16) each
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                const int i = c * 512 + tid, d = i / (NS * 16), o = i % (NS * 16);
                u32x4* dst = gran + (((size_t)((t + 1) & 1) * NGRP + group) * ndst + d) * per_dst + (size_t)member * NS * 16 + o;
                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(g) : "memory");
            }
        }
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = keep;
}

template <int P>
int run(int T, int delay, int nwg = 256) {
    constexpr int NL = Pat<P>::NL, NS = Pat<P>::NS, GS = Pat<P>::GS;
    const size_t ndst = NS ? GS : 1;
    const size_t n = (size_t)2 * (256 / GS) * ndst * NL * 512;       // dwordx4s
    u32x4* gran; int* err; float* out;
    CK(hipMalloc(&gran, n * 16)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&out, 256 * 4));
    float best = 1e9f; int e = 0;
    for (int rep = 0; rep < 4; ++rep) {
        unsigned* h = (unsigned*)malloc(n * 16);
        for (size_t i = 0; i < n * 4; ++i) h[i] = (i < n * 2) ? 1u : 0u;     // slot 0: every word 1 (tag 1, value bits 1)
        CK(hipMemcpy(gran, h, n * 16, hipMemcpyHostToDevice)); free(h);
        CK(hipMemset(err, 0, 4));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        probe<P><<<nwg, 512>>>(gran, T, delay, err, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        if (e) break;
        if (ms < best) best = ms;
    }
    printf("pattern %d (gather %3d KB, publish %3d KB) %3d workgroups delay %2d: %s  %.3f us/step\n", P, NL * 8, NS ? NS * 8 : NL * 8 / GS,
           nwg, delay, e ? "SPIN LIMIT" : "ok", best * 1e3f / T);
    hipFree(gran); hipFree(err); hipFree(out);
    return 0;
}

int main() {
    const int T = 2000;
    for (int d : {0, 6, 12}) {
        if (run<0>(T, d)) return 1;
        if (run<1>(T, d)) return 1;
        if (run<2>(T, d)) return 1;
        if (run<3>(T, d)) return 1;
    }
    // does the all-gather time depend on how many workgroups gather at once (aggregate bandwidth) or only on the bytes per workgroup?
    for (int nwg : {32, 64, 128, 256}) {
        if (run<0>(T, 12, nwg)) return 1;
        if (run<1>(T, 12, nwg)) return 1;
    }
    for (int d : {0, 6, 12}) {
        if (run<4>(T, d)) return 1;
        if (run<5>(T, d)) return 1;
    }
    if (run<4>(T, 6, 128)) return 1;
    return 0;
}
