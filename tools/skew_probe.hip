// Probe for VERDICT r2 item 1(b): "two row tiles per workgroup, phase-skewed by half a step" in the persistent BACKWARD scan.
// Exchange + compute only (no GRU arithmetic), same transport as the scans: memory-side sc1 granules of 16 B {v, v, v, tag}, two
// slots, groups of 32 workgroups that all-gather 128 KB per member and step (the backward pattern of rs_probe.hip, P1).
// Between gather and publish every wave runs a compute phase shaped like the real kernel's (split + MFMA: `nmfma` bf16 16x16x32
// MFMAs, each followed by `nvalu` dependent-free VALU operations on the gathered registers; two waves share a SIMD, as in the
// 512-thread scan workgroups), then a workgroup barrier + LDS reduction, then the publish.
//   mode 1 (today):   256 workgroups, ONE tile each:  gather -> compute -> barrier -> publish
//   mode 2 (skewed):  128 workgroups, TWO tiles each, tile B half a step behind tile A:
//                     issue A's gather | compute + publish B | wait A | issue B's gather | compute + publish A | wait B
//                     (a tile's exchange flight is covered by the other tile's compute phase)
// Both advance the same 8 + 8 exchange groups (16 rows x 512 units x 3 values per group and step): us per step is comparable.
// Also printed: the compute phase alone (mode 0: no exchange), so the compute load can be calibrated against the real kernel's
// stamps (split + MFMA 0.70 us per wave, NOTEBOOK.md section 5a).
//   hipcc -O3 --offload-arch=gfx950 tools/skew_probe.hip -o tools/bin/skew_probe && tools/bin/skew_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int SPIN_LIMIT = 1 << 16;
constexpr int NL = 16, GS = 32;                  // dwordx4 loads per lane and step (128 KB per workgroup); members per group
constexpr int PUB = NL * 512 / GS;               // dwordx4s a member publishes per step (256 = 4 KB)
constexpr size_t PER_GROUP = (size_t)NL * 512;   // dwordx4s of one (slot, group) region

__device__ __forceinline__ void issue_gather(const u32x4* src, int tid, u32x4 (&w)[NL]) {
#pragma unroll
    for (int c = 0; c < NL; ++c) {
        const u32x4* q = src + (size_t)c * 512 + tid;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w[c]) : "v"(q) : "memory");
    }
}

// returns true when every granule of this wave carries `tag`
__device__ __forceinline__ bool tags_ok(u32x4 (&w)[NL], unsigned tag) {
    bool ok = true;
#pragma unroll
    for (int c = 0; c < NL; ++c) { asm volatile("" : "+v"(w[c])); ok = ok && w[c].w == tag; }
    return __all(ok);
}

// the compute phase of one tile in one wave: nmfma MFMAs on operands taken from the gathered registers, nvalu VALU ops per MFMA
constexpr int NMFMA = 36;                         // the real backward step: 36 v_mfma_f32_16x16x32_bf16 per wave
// one quarter (9 MFMAs into three independent accumulators, as the three gates of the real product) of a tile's compute phase
template <int Q>
__device__ __forceinline__ void compute_q(u32x4 (&w)[NL], int nvalu, f32x4 (&acc)[3]) {
    const unsigned m = 0xffff0000u;
#pragma unroll
    for (int i = Q * (NMFMA / 4); i < (Q + 1) * (NMFMA / 4); ++i) {        // (fully unrolled: the register array is never indexed at run time)
        u32x4 a = w[i & (NL - 1)], b = w[(i + 5) & (NL - 1)];
        for (int v = 0; v < nvalu; ++v) {            // "split" work: and / shift / sub on the gathered values
            a.x = (a.x & m) - (b.y >> 1); a.y = (a.y & m) + (b.z >> 1); b.x ^= a.z; b.w += a.w;
            asm volatile("" : "+v"(a), "+v"(b));
        }
        acc[i % 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i % 3], 0, 0, 0);
    }
}
__device__ __forceinline__ float acc_sum(const f32x4 (&acc)[3]) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    return s;
}
__device__ __forceinline__ float compute(u32x4 (&w)[NL], int nmfma, int nvalu) {
    f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    compute_q<0>(w, nvalu, acc); compute_q<1>(w, nvalu, acc); compute_q<2>(w, nvalu, acc); compute_q<3>(w, nvalu, acc);
    return acc_sum(acc);
}
// a tile's compute phase with the OTHER tile's next gather issued after quarter `at` of it (0: before it)
__device__ __forceinline__ float compute_issue(u32x4 (&w)[NL], int nvalu, int at, const u32x4* other_src, int tid, u32x4 (&other)[NL], bool issue) {
    f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (issue && at == 0) issue_gather(other_src, tid, other);
    compute_q<0>(w, nvalu, acc);
    if (issue && at == 1) issue_gather(other_src, tid, other);
    compute_q<1>(w, nvalu, acc);
    if (issue && at == 2) issue_gather(other_src, tid, other);
    compute_q<2>(w, nvalu, acc);
    if (issue && at == 3) issue_gather(other_src, tid, other);
    compute_q<3>(w, nvalu, acc);
    return acc_sum(acc);
}

__device__ __forceinline__ float reduce_and_publish(float s, float (*red)[64], int tid, int lane, int wave, u32x4* dst, unsigned next_tag) {
    red[wave][lane] = s;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += red[k][lane];
    r = r * 1e-30f + 1.0f;
    u32x4 g;
    g.x = __float_as_uint(r); g.y = __float_as_uint(r); g.z = __float_as_uint(r); g.w = next_tag;
    if (tid < PUB) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + tid), "v"(g) : "memory");
    __syncthreads();
    return r;
}

// region of (slot, exchange group): exchange group = tile * ngroups + group
__device__ __forceinline__ u32x4* region(u32x4* gran, int slot, int xg, int nxg) { return gran + ((size_t)slot * nxg + xg) * PER_GROUP; }

template <int MODE>
__global__ __launch_bounds__(512) void probe(u32x4* gran, int T, int delay, int nmfma, int nvalu, int* err, float* out) {
    __shared__ float red[8][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int group = blockIdx.x / GS, member = blockIdx.x % GS;
    const int ngroups = gridDim.x / GS;
    constexpr int TILES = MODE == 2 ? 2 : 1;
    const int nxg = ngroups * TILES;
    float keep = 0.f;
    if (MODE == 0) {                                   // compute phase alone
        u32x4 w[NL];
#pragma unroll
        for (int c = 0; c < NL; ++c) w[c] = u32x4{(unsigned)tid, 1u, 2u, 3u};
        for (int t = 0; t < T; ++t) {
            keep += compute(w, nmfma, nvalu);
            red[wave][lane] = keep;
            __syncthreads();
            keep += red[(wave + 1) & 7][lane] * 1e-30f;
            __syncthreads();
        }
        if (tid == 0) out[blockIdx.x] = keep;
        return;
    }
    if (MODE == 1) {
        u32x4 w[NL];
        bool dead = false;
        for (int t = 0; t < T && !dead; ++t) {
            const unsigned tag = (unsigned)t + 1u;
            const u32x4* src = region(gran, t & 1, group, nxg);
            int spins = 0;
            for (int z = 0; z < delay; ++z) __builtin_amdgcn_s_sleep(1);
            for (;;) {
                issue_gather(src, tid, w);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tags_ok(w, tag)) break;
                if (++spins > SPIN_LIMIT) { dead = true; if (lane == 0) atomicExch(err, t + 1); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            const float s = compute(w, nmfma, nvalu);
            keep += reduce_and_publish(s, red, tid, lane, wave, region(gran, (t + 1) & 1, group, nxg) + (size_t)member * PUB, tag + 1u);
        }
        if (tid == 0) out[blockIdx.x] = keep;
        return;
    }
    // MODE 2: two tiles, skewed.  Invariant at the top of an iteration: A(t) is in wa; B(t)'s gather is in flight or done.
    // `delay` = the quarter of the other tile's compute phase after which a tile's next gather is issued (0..3): issued right
    // after the workgroup's own publish, an attempt is always too early (the peers publish at the same moment)
    u32x4 wa[NL], wb[NL];
    bool dead = false;
    const int xa = group, xb = ngroups + group, at = delay;
    auto wait_for = [&](u32x4 (&w)[NL], const u32x4* src, unsigned tag, int t, bool store_behind) {
        int spins = 0;
        if (store_behind && tid < PUB) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        while (!tags_ok(w, tag)) {
            if (++spins > SPIN_LIMIT) { dead = true; if (lane == 0) atomicExch(err, t + 1); break; }
            __builtin_amdgcn_s_sleep(1);
            issue_gather(src, tid, w);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    issue_gather(region(gran, 0, xa, nxg), tid, wa);
    wait_for(wa, region(gran, 0, xa, nxg), 1u, 0, false);
    for (int t = 0; t < T && !dead; ++t) {
        const unsigned tag = (unsigned)t + 1u;
        // ---- tile A, step t; B(t)'s gather goes out after quarter `at` of it (the peers published B(t) a quarter-and-a-publish ago)
        {
            const float s = compute_issue(wa, nvalu, at, region(gran, t & 1, xb, nxg), tid, wb, true);
            keep += reduce_and_publish(s, red, tid, lane, wave, region(gran, (t + 1) & 1, xa, nxg) + (size_t)member * PUB, tag + 1u);
        }
        wait_for(wb, region(gran, t & 1, xb, nxg), tag, t, true);
        if (dead) break;
        // ---- tile B, step t; A(t+1)'s gather goes out after quarter `at` of it
        {
            const float s = compute_issue(wb, nvalu, at, region(gran, (t + 1) & 1, xa, nxg), tid, wa, true);
            keep += reduce_and_publish(s, red, tid, lane, wave, region(gran, (t + 1) & 1, xb, nxg) + (size_t)member * PUB, tag + 1u);
        }
        wait_for(wa, region(gran, (t + 1) & 1, xa, nxg), tag + 1u, t, true);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) out[blockIdx.x] = keep;
}

template <int MODE>
int run(int T, int delay, int nmfma, int nvalu, int nwg, float* us_out) {
    const int ngroups = nwg / GS, nxg = ngroups * (MODE == 2 ? 2 : 1);
    const size_t n = (size_t)2 * (nxg ? nxg : 1) * PER_GROUP;
    u32x4* gran; int* err; float* out;
    CK(hipMalloc(&gran, n * 16)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&out, 256 * 4));
    float best = 1e9f; int e = 0;
    for (int rep = 0; rep < 4; ++rep) {
        unsigned* h = (unsigned*)malloc(n * 16);
        for (size_t i = 0; i < n * 4; ++i) h[i] = (i < n * 2) ? 1u : 0u;     // slot 0: every word 1 (tag 1); slot 1: zeros
        CK(hipMemcpy(gran, h, n * 16, hipMemcpyHostToDevice)); free(h);
        CK(hipMemset(err, 0, 4));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        probe<MODE><<<nwg, 512>>>(gran, T, delay, nmfma, nvalu, err, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        if (e) break;
        if (ms < best) best = ms;
    }
    *us_out = e ? -1.f : best * 1e3f / T;
    hipFree(gran); hipFree(err); hipFree(out);
    return 0;
}

int main() {
    const int T = 2000;
    printf("compute phase per tile and wave: 36 MFMAs (16x16x32 bf16, three accumulators) + nvalu x 8 VALU operations per MFMA; two waves per SIMD\n");
    for (int nvalu : {0, 1}) {
        float c, a1, a1d;
        if (run<0>(T, 0, 36, nvalu, 256, &c)) return 1;
        if (run<1>(T, 0, 36, nvalu, 256, &a1)) return 1;
        if (run<1>(T, 12, 36, nvalu, 256, &a1d)) return 1;
        printf("nvalu %d: compute phase alone %.2f us per step | today: one tile x 256 workgroups %.2f us per step (%.2f with a 12-unit sleep before the gather)\n", nvalu, c, a1, a1d);
        for (int at : {0, 1, 2, 3}) {
            float a2;
            if (run<2>(T, at, 36, nvalu, 128, &a2)) return 1;
            printf("         two skewed tiles x 128 workgroups, next gather issued after quarter %d of the other tile's compute phase: %.2f us per step (both tiles)\n", at, a2);
        }
    }
    return 0;
}
