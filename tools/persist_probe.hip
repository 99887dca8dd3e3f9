// Exchange-cost probe for a persistent GRU scan: NG independent groups of GS workgroups run T dependent steps inside
// ONE launch.  Each step every workgroup (512 threads) gathers the 8-byte {value, tag} granules that the GS members
// of its group published in the previous step (GS x 256 granules = 64 KB at GS=32), reduces across its 8 waves
// through LDS, and publishes its own 256 granules.  No barrier, no flags: a granule is one naturally aligned
// 8-byte agent-scope store, the tag is the step number; two slots ping-pong.  Spins are bounded.
// (Tried and dropped: an XCD-local variant -- groups on one XCD via blockIdx % 8, which tools/xcc_probe.hip shows is a
// stable map, workgroup-scope (sc0) stores/loads through the shared L2 -- ran 1.5 us per step when it worked and hit the
// spin limit on other runs -- also with an explicit `buffer_inv sc0` before every poll and plain loads/stores: not a
// reliable publish, as the guide's "XCD-local ending" warning says.)
//   hipcc -O3 --offload-arch=gfx950 tools/persist_probe.hip -o tools/bin/persist_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int GRAN = 256;        // granules a workgroup publishes per step
constexpr int SPIN_LIMIT = 1 << 18;

template <int GS, int PLACE, int WORK>
__global__ __launch_bounds__(512) void persist(unsigned long long* gran, int NG, int T, int* err, float* out) {
    __shared__ float red[8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bid = blockIdx.x;
    const int group = PLACE == 0 ? bid % NG : bid / GS;
    const int member = PLACE == 0 ? bid / NG : bid % GS;
    constexpr int PER_WAVE = GS / 8;                 // producers a wave reads
    constexpr int NL = PER_WAVE * 4;                 // 8-byte loads per lane per step
    float keep = 0.f;
    bool dead = false;
    for (int t = 0; t < T && !dead; ++t) {
        const unsigned tag = (unsigned)t + 1u;
        const unsigned long long* src = gran + ((size_t)(t & 1) * NG + group) * GS * GRAN;
        unsigned long long v[NL];
        int spins = 0;
        for (;;) {
            bool ok = true;
            if (WORK == -4) {
                // same 8-byte granules, gathered by HALF as many instructions: each lane takes two adjacent granules with one
                // dwordx4 load (is the exchange priced per instruction or per byte?)
                typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                u32x4_ w[NL / 2];
#pragma unroll
                for (int c = 0; c < NL / 2; ++c) {
                    const int u = wave * PER_WAVE + c / 2;
                    const u32x4_* q = reinterpret_cast<const u32x4_*>(src + (size_t)u * GRAN + (c % 2) * 128) + lane;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w[c]) : "v"(q) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int c = 0; c < NL / 2; ++c) {
                    asm volatile("" : "+v"(w[c]));
                    v[2 * c] = ((unsigned long long)w[c].y << 32) | w[c].x;
                    v[2 * c + 1] = ((unsigned long long)w[c].w << 32) | w[c].z;
                }
            } else {
#pragma unroll
            for (int c = 0; c < NL; ++c) {
                const int u = wave * PER_WAVE + c / 4;
                v[c] = __hip_atomic_load(src + (size_t)u * GRAN + (c % 4) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            }
#pragma unroll
            for (int c = 0; c < NL; ++c) ok = ok && ((unsigned)(v[c] >> 32) == tag);
            if (__all(ok)) break;
            if (++spins > SPIN_LIMIT) { dead = true; if (lane == 0) atomicExch(err, t + 1); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NL; ++c) s += __uint_as_float((unsigned)v[c]);
        if (WORK > 0) {                               // stand-in for the MFMA chain: dependent FMAs
#pragma unroll 1
            for (int i = 0; i < WORK; ++i) s = fmaf(s, 0.999f, 0.001f);
        }
        red[wave][lane] = s;
        __syncthreads();
        if (threadIdx.x < GRAN) {
            float r = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) r += red[w][lane];
            r = r * 0.001f + 1.0f;
            keep += r;
            unsigned long long* dst = gran + ((size_t)((t + 1) & 1) * NG + group) * GS * GRAN + (size_t)member * GRAN;
            const unsigned long long g = ((unsigned long long)(tag + 1u) << 32) | __float_as_uint(r);
            // WORK == -1: the scattered publish map of a (row, unit) thread layout (16 x 32-byte pieces per wave)
            const int pos = WORK == -1 ? ((threadIdx.x & 3) * 64 + ((threadIdx.x & 15) >> 2) * 16 + (threadIdx.x >> 4)) : threadIdx.x;
            __hip_atomic_store(dst + pos, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[bid] = keep;
}

// 16-byte granules {a, b, c, tag} (the backward exchange: dr, dz, dn*r per (row, unit)): one global_store_dwordx4 sc1 per
// granule, gathered with global_load_dwordx4 sc1.  `torn` counts granules whose tag matched but whose payload was not the
// producer's triple (x, x+1, x+2): a torn 16-byte access.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 ld16_sc1(const u32x4* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int GS, int DUP>
__global__ __launch_bounds__(512) void persist16(u32x4* gran, int NG, int T, int* err, int* torn, float* out) {
    __shared__ float red[8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bid = blockIdx.x;
    const int group = bid % NG, member = bid / NG;
    constexpr int PER_WAVE = GS / 8;
    constexpr int NL = PER_WAVE * 4;
    float keep = 0.f;
    bool dead = false;
    int ntorn = 0;
    for (int t = 0; t < T && !dead; ++t) {
        const unsigned tag = (unsigned)t + 1u;
        const size_t plane = (size_t)2 * NG * GS * GRAN;          // DUP planes, each laid out like the single-granule case
        const u32x4* src = gran + ((size_t)(t & 1) * NG + group) * GS * GRAN;
        u32x4 v[NL * DUP];
        int spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int c = 0; c < NL * DUP; ++c) {
                const int u = wave * PER_WAVE + (c / DUP) / 4;
                const u32x4* q = src + (size_t)(c % DUP) * plane + (size_t)u * GRAN + ((c / DUP) % 4) * 64 + lane;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[c]) : "v"(q) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int c = 0; c < NL * DUP; ++c) {
                asm volatile("" : "+v"(v[c]));       // values are defined only after the wait above
                ok = ok && (v[c].w == tag);
            }
            if (__all(ok)) break;
            if (++spins > SPIN_LIMIT) { dead = true; if (lane == 0) atomicExch(err, t + 1); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NL * DUP; ++c) {
            const float a = __uint_as_float(v[c].x), b = __uint_as_float(v[c].y), cc = __uint_as_float(v[c].z);
            if (!dead && (b != a + 1.0f || cc != a + 2.0f)) ++ntorn;
            s += a;
        }
        red[wave][lane] = s;
        __syncthreads();
        if (threadIdx.x < GRAN) {
            float r = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) r += red[w][lane];
            r = r * 0.001f + 1.0f + (float)(t & 7);
            keep += r;
            u32x4* dst = gran + ((size_t)((t + 1) & 1) * NG + group) * GS * GRAN + (size_t)member * GRAN + threadIdx.x;
            u32x4 g;
            g.x = __float_as_uint(r); g.y = __float_as_uint(r + 1.0f); g.z = __float_as_uint(r + 2.0f); g.w = tag + 1u;
#pragma unroll
            for (int dd = 0; dd < DUP; ++dd) { u32x4* q = dst + (size_t)dd * plane; asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(g) : "memory"); }
        }
        __syncthreads();
    }
    if (ntorn) atomicAdd(torn, ntorn);
    if (threadIdx.x == 0) out[bid] = keep;
}

template <int GS, int DUP = 1>
int run16(int NG, int T) {
    const size_t n = (size_t)2 * NG * GS * GRAN * DUP;
    u32x4* gran; int* err; float* out;
    CK(hipMalloc(&gran, n * 16)); CK(hipMalloc(&err, 8)); CK(hipMalloc(&out, 4 * NG * GS));
    std::vector<unsigned> init(n * 4, 0u);
    for (size_t i = 0; i < n; ++i) {
        if ((i % ((size_t)2 * NG * GS * GRAN)) >= (size_t)NG * GS * GRAN) continue;      // slot 0 of every plane
        init[4 * i] = 0x3f800000u; init[4 * i + 1] = 0x40000000u; init[4 * i + 2] = 0x40400000u; init[4 * i + 3] = 1u;   // 1,2,3,tag 1
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f; int h_err[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemcpy(gran, init.data(), n * 16, hipMemcpyHostToDevice));
        CK(hipMemset(err, 0, 8));
        CK(hipDeviceSynchronize());
        hipEventRecord(e0);
        persist16<GS, DUP><<<NG * GS, 512>>>(gran, NG, T, err, err + 1, out);
        hipEventRecord(e1);
        CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        CK(hipMemcpy(h_err, err, 8, hipMemcpyDeviceToHost));
        if (h_err[0] || h_err[1]) break;
    }
    printf("%dx16B granules GS=%d NG=%d T=%d: %.2f us/step  torn=%d%s\n", DUP, GS, NG, T, best * 1e3f / T, h_err[1],
           h_err[0] ? "  ** SPIN LIMIT HIT **" : "");
    hipFree(gran); hipFree(err); hipFree(out);
    return 0;
}

template <int GS, int PLACE, int WORK>
int run(int NG, int T) {
    const size_t n = (size_t)2 * NG * GS * GRAN;
    unsigned long long* gran; int* err; float* out;
    CK(hipMalloc(&gran, n * 8)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&out, 4 * NG * GS));
    std::vector<unsigned long long> init(n, 0ull);
    for (size_t i = 0; i < (size_t)NG * GS * GRAN; ++i) init[i] = (1ull << 32) | 0x3f800000ull;   // h_0: tag 1, value 1.0f
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f; int h_err = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemcpy(gran, init.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemset(err, 0, 4));
        CK(hipDeviceSynchronize());
        hipEventRecord(e0);
        persist<GS, PLACE, WORK><<<NG * GS, 512>>>(gran, NG, T, err, out);
        hipEventRecord(e1);
        CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
        if (h_err) break;
    }
    printf("GS=%d NG=%d place=%s work=%d: %.2f us/step%s\n", GS, NG, PLACE == 0 ? "xcd" : "linear", WORK, best * 1e3f / T,
           h_err ? "  ** SPIN LIMIT HIT **" : "");
    hipFree(gran); hipFree(err); hipFree(out);
    return 0;
}

int main() {
    const int T = 300;
    run<32, 0, 0>(8, T);
    run<32, 1, 0>(8, T);
    run<32, 0, 0>(4, T);
    run<32, 1, 0>(4, T);
    run<32, 0, 0>(2, T);
    run<16, 0, 0>(8, T);
    run<16, 0, 0>(16, T);
    run<32, 0, 600>(8, T);
    run<8, 0, 0>(2, T);
    run<32, 0, -1>(8, T);
    run<32, 0, -4>(8, T);
    run<32, 0, -4>(4, T);
    run<16, 0, -4>(8, T);
    run<32, 0, -1>(4, T);
    run16<32>(8, T);
    run16<32>(8, 20000);
    run16<32>(4, T);
    run16<16>(8, T);
    run16<8>(2, T);
    run16<32, 2>(8, T);
    run16<32, 2>(4, T);
    return 0;
}
