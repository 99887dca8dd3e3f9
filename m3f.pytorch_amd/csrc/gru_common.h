// Types shared by the per-step (gru.hip) and persistent (gru_persist.hip) BiGRU scan kernels.
#pragma once
#include "common.h"

namespace m3t_gru {

constexpr int UB = 16;        // hidden units per workgroup
constexpr int RB = 32;        // batch rows per workgroup (per-step fallback kernels)
constexpr int NW = 8;         // waves per workgroup (512 threads): K is split 8 ways
constexpr int NT = NW * 64;

struct FwdGroup {
    m3t_gru_fwd_desc d[M3T_MAX_SCANS];
    int blk_start[M3T_MAX_SCANS + 1];
    int n;
    int bf16;     // M3T_BF16: the recurrent product takes bf16-rounded h_{t-1} and W_hh (fp32 accumulate, fp32 state)
};
struct BwdGroup {
    m3t_gru_bwd_desc d[M3T_MAX_SCANS];
    int blk_start[M3T_MAX_SCANS + 1];
    int n;
    int bf16;     // M3T_BF16: dgh_{t+1} and W_hh rounded to bf16 in the recurrent product
};

// bf16 x bf16 products are exact in fp32: rounding the operands and using the fp32 MFMA is exactly a bf16 MFMA with
// fp32 accumulation
__device__ __forceinline__ float rbf(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ float4 rbf4(float4 v) { return make_float4(rbf(v.x), rbf(v.y), rbf(v.z), rbf(v.w)); }
struct FragPtrs {
    float* wfrag[M3T_MAX_SCANS];    // fragment-ordered weights
    float* xfrag[M3T_MAX_SCANS];    // 2 ping-pong buffers of fragment-ordered h_t (fwd) / dgh_t (bwd)
    size_t xstride[M3T_MAX_SCANS];  // floats per ping-pong buffer
};

// The cell arithmetic, shared by every scan kernel.  Contraction is pinned (explicit fmaf, fp contract off) so that the
// per-step, fragment-ordered and persistent kernels give bit-identical results whatever hipcc fuses around them.
struct GateFwd { float r, z, n, h; };
// sigmoid and tanh on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: 1 ulp each) instead of the libm routines:
// the cell math sits on the T-step chain, where the precise expf / tanhf / division cost ~0.2 us per step.
// sigmoid(x) = 1 / (1 + 2^(-x log2 e)) (relative error ~3e-7); tanh(x) = 1 - 2 / (2^(2x log2 e) + 1) (absolute error
// ~2e-7); both saturate correctly through exp2 -> 0 / inf.
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.88539008177792681f * x) + 1.0f);
}
__device__ __forceinline__ GateFwd gru_cell_fwd(float xr, float xz, float xn, float hr, float hz, float hn, float hprev) {
#pragma clang fp contract(off)
    GateFwd o;
    o.r = fast_sigmoid(xr + hr);
    o.z = fast_sigmoid(xz + hz);
    o.n = fast_tanh(fmaf(o.r, hn, xn));
    o.h = fmaf(o.z, hprev - o.n, o.n);          // (1-z)*n + z*h
    return o;
}
// dh_next/z_next: dL/dh and update gate of the step processed before (time t+1 of a forward-direction scan);
// mm = (dgh_{t+1} W_hh)[unit].  Returns dL/dh_t and the gate gradients (dnr = dn * r feeds W_hn).
struct GateBwd { float dht, dr, dz, dn, dnr; };
__device__ __forceinline__ GateBwd gru_cell_bwd(float dout, float dh_next, float z_next, float mm, bool has_next, float gr,
                                                float gz, float gn, float ghn, float hprev) {
#pragma clang fp contract(off)
    GateBwd o;
    const float carry = has_next ? fmaf(dh_next, z_next, mm) : dh_next;
    o.dht = dout + carry;
    o.dn = o.dht * (1.f - gz) * fmaf(-gn, gn, 1.f);
    o.dz = o.dht * (hprev - gn) * gz * (1.f - gz);
    o.dr = o.dn * ghn * gr * (1.f - gr);
    o.dnr = o.dn * gr;
    return o;
}

// Persistent scans (gru_persist.hip): one launch runs all T steps of a level.  *_check says whether the level is
// eligible (H in {128,256,384,512}, the grid fits the chip at one workgroup per CU, M3T_SCAN_PERSIST != 0); *_launch
// needs fp.wfrag already filled by the prep kernels (stream order) and uses fp.xfrag[i] as the exchange buffer
// (persist_exchange_bytes() bytes each, 16-B aligned).
size_t persist_exchange_bytes(int H, int B, bool backward);
bool persist_enabled();
// m3t_gru_scan_after: an event the NEXT scan call of this thread waits for right before its scan kernels (its preparation
// kernels and memsets are not held back).  take: make stream s wait for it (if set) and clear it; drop: clear it.
void persist_set_after(hipEvent_t ev);
int persist_take_after(hipStream_t s);
void persist_drop_after();
// m3t_gru_scan_events: a (start, end) hipEvent pair the NEXT scan call records right around its scan kernel(s) -- not around
// its preparation kernels, memsets and fences -- so that a caller's timing is the kernel's, as rocprofv3 reports it
void persist_set_events(hipEvent_t start, hipEvent_t end);
void persist_record_start(hipStream_t s);
void persist_record_end(hipStream_t s);
void persist_drop_events();
// m3t_gru_scan_arena: the exchange arena the NEXT scan call of this thread uses (launch-unique tags instead of a memset per
// launch, see gru_persist.hip); forget: drop what is known about an arena (its memory was reallocated or written by others)
void persist_set_arena(void* arena, size_t bytes);
void persist_drop_arena();
void persist_forget_arena(void* arena);
// m3t_gru_scan_progress: progress marks for the NEXT scan call of this thread (gru_persist.hip, ExPtrs.prog)
void persist_set_progress(unsigned* ctr, int n, const int* tb, unsigned* need);
void persist_drop_progress();
bool persist_progress_armed();
void persist_forget_progress(unsigned* ctr);
int persist_progress_ok(int n, int H, int B, int T, int flags, bool backward);
int persist_wait_progress(const unsigned* ctr, unsigned need, hipStream_t s);
int persist_bwd_prepare(const float* const* w, int n, int H, int direct, float* const* out, hipStream_t s);
struct AfterGuard { ~AfterGuard() { persist_drop_after(); persist_drop_events(); persist_drop_arena(); persist_drop_progress(); } };
int persist_poll_error();       // step+1 of a scan that hit its spin limit since the last persist_reset_error(), else 0 (sticky)
void persist_reset_error();     // clears the error words; only after the device has been synchronised
void persist_set_defer(int on); // m3t_gru_error_defer: scan calls do not return M3T_ESPIN behind a dead scan (several ranks: raise at an agreed point)
bool persist_deferred();
int persist_inject_error(hipStream_t s);      // fault injection: raises the error words from a kernel, in stream order
unsigned* persist_error_word_dev(bool create = false);   // device address of the STICKY flag word (host-mapped), or nullptr before the first persistent scan
int persist_owner_state();      // 0 not decided yet, 1 this process owns the device's persistent scans, 2 another process does
bool persist_fwd_check(const FwdGroup& g, int B, int T);
bool persist_bwd_check(const BwdGroup& g, int B, int T);
bool persist_fwd_uses_x6(const FwdGroup& g, int B, int T, int flags);   // then wfrag holds bf16x3 fragments (18 H^2 bytes), filled by the launch itself
int persist_fwd_launch(const FwdGroup& g, const FragPtrs& fp, int B, int T, int flags, hipStream_t s);
bool persist_bwd_uses_x6(const BwdGroup& g, int B, int T, int flags);   // fp32-accurate bf16x6 backward: wfrag holds bf16x3 fragments (18 H^2 bytes), filled by the launch itself
bool persist_bwd_uses_16(const BwdGroup& g, int B, int T, int flags);   // bf16 mode: wfrag holds bf16 fragments, filled by the launch itself
int persist_bwd_launch(const BwdGroup& g, const FragPtrs& fp, int B, int T, int flags, hipStream_t s);
int persist_launch_count();
void persist_count_launch();     // one more one-launch scan (the solo kernels count as such)
// H = 128 levels (gru_solo.hip): one launch, one workgroup per (scan, clip), no exchange between workgroups
bool solo_fwd_ok(const FwdGroup& g, int B, int T, int flags);
bool solo_bwd_ok(const BwdGroup& g, int B, int T, int flags);
int solo_fwd_launch(const FwdGroup& g, int B, int T, hipStream_t s);
int solo_bwd_launch(const BwdGroup& g, int B, int T, int flags, hipStream_t s);
int persist_profile(unsigned long long* out6);

}  // namespace m3t_gru
