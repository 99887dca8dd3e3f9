// Where does a frame's time go inside CBAM's F1 and B2?  Includes csrc/cbam_fused.hip with M3T_CBAM_STAMPS (thread 0 of every
// workgroup writes the 100 MHz wall clock at each phase boundary), runs the fused operator on 2048 random frames of one stage
// shape and prints the mean phase durations, the mean workgroup lifetime and the kernel span.
//   hipcc -O3 --offload-arch=gfx950 -DM3T_CBAM_STAMPS -Iinclude -Im3f.pytorch_amd/csrc tools/cbam_phase_probe.hip -o tools/bin/cbam_phase_probe
//   tools/bin/cbam_phase_probe 256 7      (C, H = W)
#include "cbam_fused.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static float* dev_rand(size_t n, float scale, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((int)(s >> 8) % 20001 - 10000) / 10000.f; }
    float* d = nullptr;
    if (hipMalloc(&d, n * sizeof(float)) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    return d;
}
template <class T> static T* dev_zero(size_t n) {
    T* d = nullptr;
    if (hipMalloc(&d, n * sizeof(T)) != hipSuccess) return nullptr;
    (void)hipMemset(d, 0, n * sizeof(T));
    return d;
}

static void report(const char* name, const char* const* phase, int N) {
    std::vector<unsigned long long> st((size_t)8192 * 8);
    (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_cb_stamps), st.size() * sizeof(unsigned long long));
    double dur[7] = {0, 0, 0, 0, 0, 0, 0}, life = 0;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int n = 0; n < N; ++n) {
        const unsigned long long* s = &st[(size_t)n * 8];
        for (int i = 0; i < 7; ++i) dur[i] += (double)(s[i + 1] - s[i]);
        life += (double)(s[7] - s[0]);
        if (s[0] < t0) t0 = s[0];
        if (s[7] > t1) t1 = s[7];
    }
    printf("%s: span %.1f us, mean workgroup lifetime %.2f us (%d frames; %.1f lifetimes per span => %.1f frames resident per CU)\n", name, (t1 - t0) * 0.01, life / N * 0.01, N,
           (t1 - t0) / (life / N), (double)N / 256.0 / ((t1 - t0) / (life / N)));
    for (int i = 0; i < 7; ++i) printf("   %-34s %7.2f us\n", phase[i], dur[i] / N * 0.01);
}

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 7, W = H, N = 2048, Cr = C / 16, HW = H * W;
    const size_t nx = (size_t)N * C * HW;
    float *x = dev_rand(nx, 1.f, 1), *dy = dev_rand(nx, 1.f, 2), *w1 = dev_rand((size_t)Cr * C, 0.1f, 3), *b1 = dev_rand(Cr, 0.1f, 4);
    float *w2 = dev_rand((size_t)C * Cr, 0.1f, 5), *b2 = dev_rand(C, 0.1f, 6), *cw = dev_rand(50, 0.2f, 7), *bnw = dev_rand(1, 1.f, 8), *bnb = dev_rand(1, 1.f, 9);
    float *rm = dev_zero<float>(1), *rv = dev_rand(1, 1.f, 10), *y = dev_zero<float>(nx), *dx = dev_zero<float>(nx), *cs = dev_zero<float>((size_t)N * C);
    int32_t *amp = dev_zero<int32_t>((size_t)N * C), *cam = dev_zero<int32_t>((size_t)N * HW);
    float *pooled = dev_zero<float>((size_t)N * 2 * C), *hidden = dev_zero<float>((size_t)N * 2 * Cr), *comp = dev_zero<float>((size_t)N * 2 * HW);
    float *xhat = dev_zero<float>((size_t)N * HW), *ss = dev_zero<float>((size_t)N * HW), *stats = dev_zero<float>(4);
    float *dw1 = dev_zero<float>((size_t)Cr * C), *db1 = dev_zero<float>(Cr), *dw2 = dev_zero<float>((size_t)C * Cr), *db2 = dev_zero<float>(C), *dcw = dev_zero<float>(50);
    float *dbnw = dev_zero<float>(1), *dbnb = dev_zero<float>(1);
    const size_t wsb = m3t_cbam_fused_ws_bytes(N, C, Cr, H, W);
    float* ws = reinterpret_cast<float*>(dev_zero<char>(wsb));
    if (!x || !dy || !y || !dx || !ws) { printf("alloc failed\n"); return 1; }
    if (!m3t_cbam_fused_ok(C, Cr, H, W)) { printf("shape not covered by the fused operator\n"); return 1; }
    static const char* f1p[7] = {"squeeze (HBM read of the frame)", "MLP layer 1", "MLP layer 2 + sigmoid", "compress loop (L2 re-read)", "compress combine", "5x5 conv", "BatchNorm partial sums"};
    static const char* b2p[7] = {"BatchNorm backward, frame copies", "conv weight-grad share + conv bwd", "dcs sweep (HBM read dy, x)", "MLP bwd: W2^T product", "MLP bwd: ReLU masks", "MLP bwd: W1^T product", "dx sweep (dy re-read, write)"};
    for (int rep = 0; rep < 3; ++rep) {
        int rc = m3t_cbam_fwd(x, w1, b1, w2, b2, cw, bnw, bnb, rm, rv, y, cs, amp, pooled, hidden, comp, cam, xhat, ss, stats, N, C, Cr, H, W, 1, 0.1f, 1e-5f, ws, wsb, nullptr);
        if (rc) { printf("fwd rc %d\n", rc); return 1; }
        CK(hipDeviceSynchronize());
        if (rep == 2) report("F1", f1p, N);
        rc = m3t_cbam_bwd(dy, x, w1, w2, cw, bnw, cs, amp, pooled, hidden, comp, cam, xhat, ss, stats, dx, dw1, db1, dw2, db2, dcw, dbnw, dbnb, N, C, Cr, H, W, 1, ws, wsb, nullptr);
        if (rc) { printf("bwd rc %d\n", rc); return 1; }
        CK(hipDeviceSynchronize());
        if (rep == 2) report("B2", b2p, N);
    }
    return 0;
}
