// Can the h_t exchange of a persistent scan stay inside ONE XCD's L2?  256 workgroups; a workgroup joins the group of the
// XCD it runs on (HW_REG_XCC_ID; its member index is a ticket from a per-XCD counter), so a group's 32 members share an
// L2.  Per step every workgroup gathers the 32 x 256 8-byte {value, tag} granules of its group (8 dwordx4 loads per
// lane), reduces through LDS, publishes its own 256.  MODE 0: agent-scope (sc1) stores and loads -- what gru_persist.hip
// does, served from memory.  MODE 1: plain stores (write-through to L2), `buffer_inv sc1` before every gather attempt,
// plain loads -- TCP invalidated, L2 hits.  MODE 2: sc0 stores / sc0 loads, no invalidate.  MODE 3: plain stores followed
// by `buffer_wbl2 sc0`, buffer_inv sc0 + plain loads.  delay = 64-cycle sleeps before the first attempt.
//   hipcc -O3 --offload-arch=gfx950 tools/l2x_probe.hip -o tools/bin/l2x_probe && tools/bin/l2x_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int GS = 32, GRAN = 256, SPIN_LIMIT = 1 << 16;

template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned long long* gran, int* tickets, int T, int delay, int* err, float* out, int* place) {
    __shared__ float red[8][64];
    __shared__ int s_member;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;
    if (threadIdx.x == 0) s_member = atomicAdd(&tickets[xcc], 1);
    __syncthreads();
    const int group = xcc, member = s_member;
    if (threadIdx.x == 0) place[blockIdx.x] = xcc * 100 + member;
    if (member >= GS) { if (threadIdx.x == 0) atomicExch(err, -1); return; }
    float keep = 0.f;
    bool dead = false;
    for (int t = 0; t < T && !dead; ++t) {
        const unsigned tag = (unsigned)t + 1u;
        const unsigned long long* src = gran + ((size_t)(t & 1) * 8 + group) * GS * GRAN;
        u32x4 w[8];
        int spins = 0;
        for (int z = 0; z < delay; ++z) __builtin_amdgcn_s_sleep(1);
        for (;;) {
            if (MODE == 1) asm volatile("buffer_inv sc1" ::: "memory");
            if (MODE == 3) asm volatile("buffer_inv sc0" ::: "memory");
#pragma unroll
            for (int c = 0; c < 8; ++c) {                 // wave reads producers 4*wave .. 4*wave+3, two dwordx4 per producer per lane
                const u32x4* q = reinterpret_cast<const u32x4*>(src + (size_t)(wave * 4 + c / 2) * GRAN + (c % 2) * 128) + lane;
                if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w[c]) : "v"(q) : "memory");
                else if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(w[c]) : "v"(q) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w[c]) : "v"(q) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bool ok = true;
#pragma unroll
            for (int c = 0; c < 8; ++c) { asm volatile("" : "+v"(w[c])); ok = ok && w[c].y == tag && w[c].w == tag; }
            if (__all(ok)) break;
            if (++spins > SPIN_LIMIT) { dead = true; if (lane == 0) atomicExch(err, t + 1); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) s += __uint_as_float(w[c].x) + __uint_as_float(w[c].z);
        red[wave][lane] = s;
        __syncthreads();
        if (threadIdx.x < GRAN) {
            float r = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) r += red[k][lane];
            r = r * 0.001f + 1.0f;
            keep += r;
            unsigned long long* dst = gran + ((size_t)((t + 1) & 1) * 8 + group) * GS * GRAN + (size_t)member * GRAN + threadIdx.x;
            const unsigned long long g = ((unsigned long long)(tag + 1u) << 32) | __float_as_uint(r);
            if (MODE == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(dst), "v"(g) : "memory");
            else if (MODE == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(dst), "v"(g) : "memory");
            else if (MODE == 3) asm volatile("global_store_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)\n\tbuffer_wbl2 sc0" :: "v"(dst), "v"(g) : "memory");
            else asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(dst), "v"(g) : "memory");
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = keep;
}

template <int MODE>
int run(int T, int delay) {
    unsigned long long* gran; int *tickets, *err, *place; float* out;
    const size_t n = (size_t)2 * 8 * GS * GRAN;
    CK(hipMalloc(&gran, n * 8)); CK(hipMalloc(&tickets, 64)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&out, 256 * 4)); CK(hipMalloc(&place, 256 * 4));
    float best = 1e9f; int e = 0;
    for (int rep = 0; rep < 4; ++rep) {
        // slot 0 must read tag 1 at step 0: seed every granule of slot 0 with tag 1
        unsigned long long* h = (unsigned long long*)malloc(n * 8);
        for (size_t i = 0; i < n; ++i) h[i] = i < n / 2 ? (1ull << 32) : 0ull;
        CK(hipMemcpy(gran, h, n * 8, hipMemcpyHostToDevice)); free(h);
        CK(hipMemset(tickets, 0, 64)); CK(hipMemset(err, 0, 4));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        probe<MODE><<<256, 512>>>(gran, tickets, T, delay, err, out, place);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        if (e) break;
        if (ms < best) best = ms;
    }
    printf("mode %d delay %2d: %s  %.3f us/step\n", MODE, delay, e ? (e < 0 ? "BAD PLACEMENT" : "SPIN LIMIT") : "ok", best * 1e3f / T);
    if (e > 0) printf("   (gave up at step %d)\n", e);
    hipFree(gran); hipFree(tickets); hipFree(err); hipFree(out); hipFree(place);
    return 0;
}

int main() {
    const int T = 2000;
    for (int d : {0, 6, 10, 14, 18}) {
        if (run<0>(T, d)) return 1;
        if (run<1>(T, d)) return 1;
    }
    for (int d : {0, 8}) { if (run<2>(T, d)) return 1; if (run<3>(T, d)) return 1; }
    return 0;
}
