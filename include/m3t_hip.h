/*
 * m3t_hip.h -- C ABI of libm3t_hip.so: the MI355X (gfx950) kernels behind the M3T
 * (sailordiary/m3f.pytorch) forward/backward hot path.
 *
 * The reference has no native code and no operator registry: every function below
 * replaces a stock torch op that a reference module calls (cited per entry, paths
 * relative to the reference root).  The host side (the models package under m3f.pytorch_amd) keeps the
 * reference's Python module API and binds these entry points with ctypes
 * (INTEGRATION.md shows the stub a maintainer would add to the reference itself).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated; tensors are row-major;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); all work is
 *     enqueued on it, nothing synchronises, no allocation happens inside a call;
 *   - return value: 0 on success, otherwise a hipError_t (launch/config error) or
 *     M3T_EINVAL for bad arguments.  The caller raises;
 *   - kernels are deterministic: fixed reduction order, no floating-point atomics.
 */
#ifndef M3T_HIP_H
#define M3T_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M3T_EINVAL 10001
#define M3T_MAX_SCANS 8
#define M3T_ESPIN 10002          /* a persistent scan gave up waiting for a peer workgroup (see m3t_gru_scan_fwd) */
#define M3T_SCAN_NO_PERSIST 1    /* m3t_gru_scan_* flags: take the launch-per-step path */
#define M3T_SCAN_WHH 8           /* m3t_gru_scan_bwd flags: every desc.w_hh_t points at the UNtransposed parameter w_hh [3H][H] (saves the caller
                                  * a transpose per direction in front of the scan); needs H % 16 == 0 and the workspace, else M3T_EINVAL */
#define M3T_SCAN_FP32 4          /* m3t_gru_scan_fwd flags: keep the recurrent product on fp32 MFMAs (bit-identical to the
                                  * launch-per-step kernels) instead of the fp32-accurate bf16x6 form */
#define M3T_SCAN_FAULT 16        /* m3t_gru_scan_* flags, FAULT INJECTION for tests: workgroup 0 of a persistent launch stays silent at step T/2,
                                  * so its peers run into their spin limit (env M3T_SCAN_SPIN_LIMIT lowers it) and the error path below runs */
#define M3T_SCAN_WIDE 32         /* m3t_gru_scan_* flags: run the level with WIDE workgroups where that form exists (H = 512 / 256 in the M3T_GEMM_F16X3 mode;
                                  * other levels ignore the flag): 16 rows x 32 units per workgroup instead of x 16 -- the same gather per workgroup,
                                  * half the workgroups (gru_v | gru_a, 4 scans x 32 clips: 128 instead of 256), each owning its CU, every group on one
                                  * XCD.  m3t.ops asks for it where it makes room for a SECOND persistent scan (the audio stack's scans run at the same
                                  * time as the gru_v | gru_a level's) and for every backward level (the wide backward kernel is no slower than the narrow
                                  * one on twice the CUs).  m3t_gru_scan_workgroups() says what a call would launch.  Same results as without the flag up to
                                  * fp32 rounding (the backward kernel scales per producer workgroup instead of per 16-unit tile). */
#define M3T_BF16 2               /* precision flag shared by m3t_sgemm (= M3T_GEMM_BF16), m3t_conv1d_* and m3t_gru_scan_*:
                                  * matmul operands rounded to bf16 (nearest even), fp32 accumulate, fp32 state/epilogue */
/* m3t_sgemm flags: BACKGROUND caps residency at one workgroup per CU (for GEMMs that run on a side stream
 * beside the latency-critical recurrence, e.g. weight gradients) */
#define M3T_GEMM_BACKGROUND 1
/* mixed precision (config C2 of BASELINE.json: "bf16, fp32 accumulate, fp32 master weights"): both operands are rounded
 * to bf16 (nearest even) before the product, accumulation and epilogue stay fp32.  Interior shapes then run ONE bf16
 * MFMA per tile step instead of the six of the fp32-accurate path. */
#define M3T_GEMM_BF16 2
/* "high" precision in the sense of torch.set_float32_matmul_precision('high'): each fp32 operand is treated as the sum of TWO
 * bfloat16 numbers (16 mantissa bits) and a product is four bf16 MFMAs with fp32 accumulation -- relative error ~2^-16 per term
 * instead of the 2^-23 of the default six-product form ('highest'), 1.3-1.5x the GEMM rate.  Opt-in (m3t.ops.precision("high"));
 * interior shapes only (others run exact fp32); M3T_GEMM_BF16 wins if both are set.  The recurrent scans ignore it. */
#define M3T_GEMM_HIGH 256
/* fp32-accurate products from TWO fp16 terms per operand and three MFMAs (a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulate) instead
 * of three bf16 terms and six: fp16 holds 11 significant bits, so two terms carry 22 (+ the sign of the remainder) and the dropped
 * a_lo b_lo is <= 2^-22 relative; fp16's narrow exponent range is met by scaling each operand by the power of two that puts its
 * largest magnitude into [2^14, 2^15) -- the library measures max |A|, max |B| with one extra launch in front of the GEMM, on the
 * same stream, no host round trip -- and unscaling in the epilogue (exact).  Elements more than 2^17 below their operand's maximum
 * keep an absolute error of 2^-40 of that maximum instead of a relative one.  Measured against fp64 the mode is as accurate as the
 * six-product form (fewer fp32 accumulate roundings per k) -- DESIGN.md section 7.  Interior shapes only; the BF16 and HIGH flags
 * win if set.  m3t_gru_scan_fwd / m3t_gru_scan_bwd take the flag too: the forward persistent scans then form their recurrent product from
 * two fp16 terms (h is bounded by 1; W_hh is scaled per workgroup slice by the scan's own prep launch), the H = 512 backward
 * persistent scans run the producer-split kernel (two fp16 terms per exchanged value, scaled per producer tile); both fp32-accurate,
 * no slots needed.  Without the flag (or with env M3T_GEMM_F16X3=0) the scans keep the three-bf16-term products. */
#define M3T_GEMM_F16X3 1024
/* scheduling hint: other streams run persistent scans while this GEMM runs (the interleaved encoder level of m3t.ops._MultiBiGRU):
 * take gemm_x6.hip, whose phases only overlap across workgroups, instead of the software-pipelined gemm_x6d.hip -- the faster
 * kernel's higher request rate slows the scans' exchange by as much as it gains (measured: same step time, backward scans
 * 6.8 -> 7.7 ms by events).  Results are bit-identical either way. */
#define M3T_GEMM_BESIDE_SCAN 512
/* the caller promises that nothing else shares the chip while this GEMM runs.  Rounds 2-3 gave such calls a 256 x 256-tile kernel in the
 * six-product mode; it had no fp16x3 form and was removed in round 4 -- the flag is accepted and currently changes nothing. */
#define M3T_GEMM_EXCLUSIVE 8
/* m3t_conv3d_wgrad_taps only: x_cl and dy_cl are the m3t_f16x3_split images of the two operands under amax_x / amax_dy (both required, with
 * M3T_GEMM_F16X3) -- the forward walk and the data gradient's walk have made them already; the same sums, bit for bit (round 6). */
#define M3T_CONV_IMAGES 4096

/* Patch matrix (im2col) of a Conv3d input x [N, Ci, T, H, W] for the weight-gradient GEMM dW = dy^T P of the 3-D conv stems (reference
 * models/backbone.py:73-103,179-271,327-332; forward and data gradient stay on MIOpen): out [rows_pad, Kp] row-major, row = (n, t', h', w'),
 * column = (ci, kt, kh, kw) -- the order of weight.view(Co, -1) -- zero for the convolution's padding, for columns >= Ci kt kh kw and for
 * rows >= N T' H' W' (tile padding of the GEMM).  Kp % 4 == 0, out 16-B aligned.  amax_slot (optional): raised to the bits of max |P|
 * (a magnitude slot of m3t_sgemm_scaled, zero-initialised by the caller).  One launch. */
int m3t_im2col3d(const float* x, int N, int Ci, int T, int H, int W, int kt, int kh, int kw, int st, int sh, int sw,
                 int pt, int ph, int pw, float* out, long long rows_pad, int Kp, unsigned long long* amax_slot, void* stream);

/* The convolutions' data gradient WITHOUT a patch matrix (round 5; reference models/backbone.py:73-103,179-271, models/resnet.py:40-45): a tap-walk
 * contraction over CHANNELS-LAST grids,
 *   dst[(n, t, h, w)][cd] = sum over taps (jt, jh, jw) and channels cs of
 *                           src[(n, t + base_t + sign jt, h + base_h + sign jh, w + base_w + sign jw)][cs] * w_taps[((jt kh + jh) kw + jw) C_src + cs][cd]
 * with src rows on the grid To x Ho x Wo (terms outside it are zero) and dst rows on T x H x W: an implicit GEMM whose A tiles are whole-line loads
 * of shifted source rows (a 32-deep k tile lies inside one tap).  Data gradient of a stride-1 convolution with padding p: src = dy channels-last
 * [N T' H' W'][C_out] (backward has it for the weight gradient anyway), w_taps[(tap, co)][ci] = W[co][ci][tap], base = +p, sign = -1, dst = dx
 * channels-last.  (sign = +1, base = -p, w_taps[(tap, ci)][co]: the forward convolution over a channels-last input.)
 * C_dst % 64 == 0, C_src % 32 == 0, 16-B aligned (any number of rows N T H W since round 6: a ragged last tile reads zeros and is not stored); flags / amax as m3t_sgemm_scaled (NULL slots are measured); ws (optional):
 * split-K slabs for the layers whose tile count does not fill the chip (deterministic reduction, as m3t_sgemm). */
int m3t_conv3d_taps(const float* src, const float* w_taps, float* dst, int N, int C_src, int C_dst, int T, int H, int W,
                    int To, int Ho, int Wo, int kt, int kh, int kw, int base_t, int base_h, int base_w, int sign, int flags,
                    const unsigned long long* amax_src, const unsigned long long* amax_w, float* ws, size_t ws_bytes, float* dst_planes,
                    void* stream);

/* Operands split ONCE (round 5).  In the fp16x3 kernels every workgroup splits the operand tiles it stages -- each element as often as tiles read
 * it, five to six vector instructions per pair beside every MFMA.  m3t_f16x3_split writes the "P4" image of a K-contiguous operand x
 * [rows][cols] (ld): the 16 bytes of four consecutive values become {hi 0|1, hi 2|3, lo 0|1, lo 2|3}, the two fp16 terms of x * s, s the
 * power of two of `slot` (a raised magnitude slot of x) -- same size, same addressing as x.  A kernel that takes images copies them to LDS.
 * m3t_conv3d_taps_pre: m3t_conv3d_taps on the image of src and on w_img[cd][(tap, cs)] (the image of the K-contiguous weight matrix), under
 * the slots the images were made with.  Bit-identical to the in-kernel split (the same roundings).  cols % 4, ld % 4, 16-B aligned. */
int m3t_f16x3_split(const float* x, size_t rows, int cols, size_t ld, float* out, size_t ldo, const unsigned long long* slot, void* stream);
/* ... of a permuted view (round 6): out [R][T * Cc] <- the image of w[r * sR + t * sT + c * sC] -- a convolution's weights [Co][Ci][taps] as the
 * K-contiguous matrix a walk reads ([co][(tap, ci)]: R = Co, Cc = Ci, sR = Ci taps, sT = 1, sC = taps; [ci][(tap, co)]: R = Ci, Cc = Co, sR =
 * taps, sT = 1, sC = Ci taps) without a permute copy in front of the split.  Cc % 4 == 0, out 16-B aligned. */
int m3t_f16x3_split_perm(const float* w, size_t R, int T, int Cc, size_t sR, size_t sT, size_t sC, float* out, const unsigned long long* slot,
                         void* stream);
/* C[M,N] = act(A B^T + bias) (+ C) with A [M][K] and B [N][K] given as images (the NT product of m3t_sgemm_scaled, fp16x3 mode; M % 128 == 0,
 * N % 64 == 0, K % 32 == 0, both slots required) */
int m3t_sgemm_pre(int M, int N, int K, const float* A_img, int lda, const float* B_img, int ldb, float* C, int ldc,
                  const float* bias, int act, int accumulate, const unsigned long long* amax_a, const unsigned long long* amax_b, void* stream);
/* The WEIGHT operand as a staged image (round 5): m3t_f16x3_image_b writes w [N][K] (ld) as [N / 64][K / 8][term][64 rows][8 halves] -- the two
 * fp16 terms of w * s (s from `slot`), every (64-row block, k-octet, term) one contiguous KiB = one LDS-DMA wave instruction; same size as w.
 * m3t_sgemm_bimg: C[M,N] = act(A B^T + bias) (+ C) with B given as that image: the 128 x 256 tile kernel fetches B straight into LDS
 * (global_load_lds_dwordx4: no registers, no conversion, no ds_write for two thirds of the staged bytes), A as m3t_sgemm_scaled.  Bit-identical
 * to m3t_sgemm_scaled with the same slots.  N % 64 (image) / % 256 (product), K % 32, M % 128; amax_a NULL: measured; amax_b required. */
int m3t_f16x3_image_b(const float* w, int N, int K, size_t ld, float* img, const unsigned long long* slot, void* stream);
int m3t_sgemm_bimg(int M, int N, int K, const float* A, int lda, const float* B_img, float* C, int ldc, const float* bias, int act,
                   int accumulate, float* ws, size_t ws_bytes, const unsigned long long* amax_a, const unsigned long long* amax_b, void* stream);
/* Round 6: the fp16x3 product of m3t_sgemm_scaled on the 256 x 256 "ring" kernels (csrc/gemm_ring.hip): both operands reach LDS as raw fp32 by
 * LDS-DMA (global_load_lds_dwordx4) into a ring of four 16-k stages, counted vmcnt in front of the stage's single barrier, the two-term split on
 * the fragment read.  Same arithmetic as m3t_sgemm_scaled under the same slots (same split, same MFMA operand placement, same product and k
 * order, same split-K slabs: bit-identical where both take the same K passes).  transA / transB / seg_* as m3t_sgemm (nn.Linear, the GRU input
 * projections and their gradients: reference models/rnn.py:17,22-55,75; transA = transB = 1 has no caller and is refused); row-contiguous
 * operands (transA = 1, transB = 0) are fetched a k row per LDS-DMA instruction and read back with ds_read2st64_b32 -- no cross-lane
 * transpose.  N % 256 == 0, K % 16 == 0 (K % seg_len == 0, seg_len >= 32 when segmented), any M (M % 4 == 0 with transA = 1), 16-B aligned
 * operands with ld % 4 == 0; splits >= 1 slabs (ws of splits * M * N floats when > 1); NULL slots are measured; `variant` selects a build of
 * the NT main loop (0 = plain, 1 = reads a stage ahead, 2 = v_fma_mix split, 3 = both; the timing-only ablation builds behind
 * profiles/r06_ring_gemm_ablation.txt are refused unless M3T_RING_ABLATIONS is set). */
int m3t_sgemm_ring(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   const float* bias, int act, int accumulate, int seg_len, int seg_stride, int a_off, int b_off, float* ws, size_t ws_bytes,
                   int splits, const unsigned long long* amax_a, const unsigned long long* amax_b, int variant, void* stream);
int m3t_conv3d_taps_pre(const float* src_img, const float* w_img, float* dst, int N, int C_src, int C_dst, int T, int H, int W,
                        int To, int Ho, int Wo, int kt, int kh, int kw, int base_t, int base_h, int base_w, int sign,
                        const unsigned long long* amax_src, const unsigned long long* amax_w, float* ws, size_t ws_bytes, float* dst_planes,
                    void* stream);

/* The whole convolution without a patch matrix (round 5, second half; reference models/backbone.py:73-103,179-271, models/resnet.py:40-45 --
 * nn.Conv3d / nn.Conv2d forward and backward of the visual stems and the per-frame ResNet).
 * m3t_conv3d_fwd_taps: y_cl[(n, t', h', w')][co] = bias[co] + sum over (kt, kh, kw, ci) of x[(n, t' st + kt - pt, h' sh + kh - ph, w' sw + kw - pw)][ci]
 *   W[co][ci][kt][kh][kw] over the CHANNELS-LAST input, any stride: x_img = the m3t_f16x3_split image of x_cl [N T H W][Ci], w_img = the image
 *   of the K-contiguous weights [Co][(tap, ci)], both under the slots given; bias may be NULL.  Co % 64 == 0, Ci % 32 == 0, any number of rows.
 * m3t_conv3d_wgrad_taps: dwt[(tap, ci)][co] = sum over the rows r = (n, t', h', w') of the dy grid of x_cl[source row of (r, tap)][ci] dy_cl[r][co]
 *   -- the walk turned round: the reduction runs over the rows (deterministic split-K slabs in ws), the tap is picked once per thread from
 *   its output row, every reduction row is decoded on the dy grid by a counter (no division in the loop).  dwt has ceil128(taps Ci) rows
 *   (rows past taps Ci are written as zeros); fp32 operands, split in the kernel -- or, with M3T_CONV_IMAGES, their m3t_f16x3_split images;
 *   flags / amax as m3t_sgemm_scaled (NULL slots are measured).
 *   Co % 64 == 0, Ci % 4 == 0, 16-B aligned; any number of rows (round 6: a ragged last reduction tile reads zeros).
 * With m3t_conv3d_taps(_pre) for the data gradient of the stride-1 layers, no convolution of the path materialises its patches.
 * dst_planes / y_planes (optional, every walk but the weight gradient's): the result is left as channel planes [N][C][T H W] there -- written by
 * the walk's own epilogue when it runs in one K pass (16-byte stores of 4 positions per channel), else reduced into dst / y_cl (then
 * scratch of the same size) and transposed by m3t_btc_to_bct. */
/* The stems' first layers (C_in <= 4; reference models/backbone.py:73-78,179-184): m3t_planes_to_cl4 writes x [N][C][S] channels-last with
 * FOUR channels (missing ones zero; raises the slot armed by m3t_amax_out); m3t_conv3d_fwd_taps4 walks the m3t_f16x3_split image of that
 * against w_img = the image of [Co][kt][kh][8][4] (kernel width padded to eight taps, channels to four, zeros): one 32-deep k tile per
 * (kt, kh) pair.  kw <= 8, Co % 64 == 0, any number of rows.  The weight gradient is m3t_conv3d_wgrad_taps with Ci = 4. */
int m3t_planes_to_cl4(const float* x, float* out, int N, int C, long long S, void* stream);
int m3t_conv3d_fwd_taps4(const float* x_img4, const float* w_img, const float* bias, float* y_cl, int N, int Co, int T, int H, int W,
                         int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, const unsigned long long* amax_x,
                         const unsigned long long* amax_w, float* ws, size_t ws_bytes, float* y_planes, void* stream);
int m3t_conv3d_fwd_taps(const float* x_img, const float* w_img, const float* bias, float* y_cl, int N, int Ci, int Co, int T, int H, int W,
                        int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, const unsigned long long* amax_x,
                        const unsigned long long* amax_w, float* ws, size_t ws_bytes, float* y_planes, void* stream);
int m3t_conv3d_wgrad_taps(const float* x_cl, const float* dy_cl, float* dwt, int N, int Ci, int Co, int T, int H, int W, int kt, int kh, int kw,
                          int st, int sh, int sw, int pt, int ph, int pw, int flags, const unsigned long long* amax_x,
                          const unsigned long long* amax_dy, float* ws, size_t ws_bytes, void* stream);

/* Magnitude slots from the PRODUCER (fp16x3 mode, see m3t_sgemm_scaled): the NEXT m3t_conv1d_fwd(_scaled) / m3t_mask_pos / m3t_mask_pos_drop /
 * m3t_weight_norm_fwd / m3t_bct_to_btc call of the calling thread raises `slot` (8 bytes, zero-initialised by the caller, epoch 0) to the bits of max |x| over
 * its output (y / out / w_t) -- in the same kernel, one 64-bit atomic max per workgroup -- so that the contraction that consumes the output
 * needs no measuring launch.  Consumed by that call (also when it fails); slot == NULL clears a pending one.  Returns 0. */
int m3t_amax_out(unsigned long long* slot);

/* library / device info: returns the ABI version; arch string copied to `arch` if non-null */
int m3t_version(void);
int m3t_device_arch(char* arch, int cap);

/* ---------------------------------------------------------------------------------
 * Dense contraction on fp32 MFMA (v_mfma_f32_32x32x2_f32), exact fp32:
 *   C[M,N] (ldc) = act( alpha-free  op(A)[M,K] * op(B)[K,N] + bias[N] ) (+ C if accumulate)
 * transA=0: A stored [M][K] (lda)   transA=1: A stored [K][M] (lda)
 * transB=0: B stored [K][N] (ldb)   transB=1: B stored [N][K] (ldb)  <- nn.Linear weight
 * act: 0 none, 1 ReLU.  bias may be NULL.
 * seg_len>0 (only with transA=1, transB=0): the reduction index k walks SEGMENTS:
 *   storage row of A = (k / seg_len) * seg_stride + k % seg_len + a_off, same for B
 *   with b_off (used for dW_hh = sum_t dgh_t^T h_{t-1}: per-clip shifted rows).
 * ws/ws_bytes: optional split-K workspace (deterministic slab reduction); may be NULL.
 * flags: 0 or any of M3T_GEMM_BACKGROUND, M3T_GEMM_BF16, M3T_GEMM_EXCLUSIVE.
 * Replaces: nn.Linear (models/rnn.py:22-55, models/model.py:88, models/att_fusion.py:13),
 * the input projections W_ih x inside nn.GRU (models/rnn.py:17,75) and their autograd. */
int m3t_sgemm(int transA, int transB, int M, int N, int K,
              const float* A, int lda, const float* B, int ldb,
              float* C, int ldc, const float* bias, int act, int accumulate,
              int seg_len, int seg_stride, int a_off, int b_off,
              float* ws, size_t ws_bytes, int flags, void* stream);

/* m3t_sgemm with the callers' MAGNITUDE SLOTS for the M3T_GEMM_F16X3 mode.  A slot is 8 bytes, 8-B aligned, whose low word holds the
 * fp32 bit pattern of an upper bound of max |x| over the operand (any bound is valid; a loose one only costs precision at the small
 * end).  amax_a / amax_b may each be NULL: that operand is then measured by the library (one extra launch on `stream`).  Producers
 * that know their magnitudes for free provide the slots: m3t_absmax (one launch for up to 16 tensors), m3t_gru_scan_bwd
 * (m3t_gru_bwd_desc.amax), a constant (|h| <= 1 for GRU outputs).  Ignored unless the fp16x3 kernel runs. */
int m3t_sgemm_scaled(int transA, int transB, int M, int N, int K,
                     const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, int act, int accumulate,
                     int seg_len, int seg_stride, int a_off, int b_off,
                     float* ws, size_t ws_bytes, int flags,
                     const unsigned long long* amax_a, const unsigned long long* amax_b, void* stream);

/* The same contraction (transA = 0) over a TIME WINDOW of batch-major sequence tensors: A is a [n_seg, win_stride, *] tensor of which only
 * the frames [win_off, win_off + win_len) of every clip take part -- row m of the product reads storage row
 * (m / win_len) * win_stride + m % win_len + win_off of A and writes the same storage row of C:
 *   C[rows of the window, 0:N] (ldc) = act( A[rows of the window, 0:K] (lda) * op(B)[K,N] + bias ) (+ C if accumulate)
 * Used by the direction-split, time-chunked hand-offs between a BiGRU scan and the projections / data gradients on either side of it
 * (reference models/rnn.py:17,75: layer l+1's input projection is out_fwd W_ih[:, :H]^T + out_rev W_ih[:, H:]^T, and each half is final for the
 * frames its direction's scan has passed: m3t_gru_scan_progress).  Only the 16-bit-term tile kernel has this form: n_seg * win_len % 128 == 0,
 * N % 64 == 0, K % 32 == 0, 16-B aligned operands with ld % 4 == 0, flags without M3T_GEMM_BF16 / _HIGH; else M3T_EINVAL.  One launch, no
 * split-K.  amax_a / amax_b as m3t_sgemm_scaled. */
int m3t_sgemm_window(int transB, int n_seg, int win_len, int win_stride, int win_off, int N, int K,
                     const float* A, int lda, const float* B, int ldb, float* C, int ldc, const float* bias,
                     int act, int accumulate, int flags, const unsigned long long* amax_a, const unsigned long long* amax_b, void* stream);
/* ... and up to M3T_WINDOW_BATCH such problems of ONE shape (same window, N, K, leading dimensions, act) as one launch: the pieces that become
 * ready at one progress mark -- every (stack, direction) pair reading that window -- fill the CUs a scan leaves free as one grid instead of
 * a queue of under-filled launches.  With n > 1 every problem must bring its magnitude slots in the fp16x3 mode. */
#define M3T_WINDOW_BATCH 8
typedef struct {
    const float* A; const float* B; float* C; const float* bias;          /* bias may be NULL */
    const unsigned long long* amax_a; const unsigned long long* amax_b;
    int accumulate;
} m3t_window_problem;
int m3t_sgemm_window_batch(int n, const m3t_window_problem* problems, int transB, int n_seg, int win_len, int win_stride, int win_off,
                           int N, int K, int lda, int ldb, int ldc, int act, int flags, void* stream);

/* slots[i] = max(slots[i], bits of max |x| over x[i] = [rows[i] x cols[i]] fp32 with leading dimension ld[i]) for n <= 16 tensors in
 * ONE launch (cols % 4 == 0, ld % 4 == 0, 16-B aligned).  The caller zero-initialises a slot before its first use. */
int m3t_absmax(int n, const float* const* x, const size_t* rows, const int* cols, const size_t* ld,
               unsigned long long* const* slots, void* stream);

/* Which kernel m3t_sgemm would run for a call of this shape (all operands 16-B aligned, ld % 4 == 0) and how many
 * split-K slabs: *kernel = 0 fp32-MFMA tile kernel, 1 the 16-bit-term 128 x 128 / 128 x 64 tile kernels (formerly also 2: the 256 x 256 tile, removed; only with
 * M3T_GEMM_EXCLUSIVE).  For tools and tests; no device work. */
int m3t_sgemm_plan(int transA, int M, int N, int K, int seg_len, size_t ws_bytes, int flags, int* kernel, int* splits);

/* out[n] (+)= sum_m X[m*ld + n], m<M, n<N  (bias gradients); ws optional (tall inputs) */
int m3t_colsum(const float* X, int M, int N, int ld, float* out, int accumulate,
               float* ws, size_t ws_bytes, void* stream);

/* dst[c][r] = src[r][c]  (src [R][C] with lds, dst [C][R] with ldd) */
int m3t_transpose(const float* src, int R, int C, int lds, float* dst, int ldd, void* stream);

/* y = relu'(a) * dy elementwise: dy[i] = a[i] > 0 ? dy[i] : 0 over n elements (in place on dy) */
int m3t_relu_bwd(const float* a, float* dy, size_t n, void* stream);

/* ---------------------------------------------------------------------------------
 * BiGRU recurrence (the T-step scan inside nn.GRU, models/rnn.py:17,75).
 * One call advances up to M3T_MAX_SCANS INDEPENDENT direction-scans (both directions
 * of a layer; the same layer of independent stacks) in lock-step: one launch per
 * time step covers all of them.  Gate order [r;z;n] (torch):
 *   r = sig(xr + Whr h + bhr); z = sig(xz + Whz h + bhz); n = tanh(xn + r*(Whn h + bhn))
 *   h' = (1-z)*n + z*h ; h0 = 0 ; reverse scans run t = T-1 .. 0.
 * xproj already contains W_ih x + b_ih (m3t_sgemm). */
typedef struct {
    const float* xproj;  /* [B,T,ldx]; this scan reads columns [xoff, xoff+3H)            */
    const float* w_hh;   /* [3H,H]                                                         */
    const float* b_hh;   /* [3H]                                                           */
    float* out;          /* [B,T,ldo]; h_t written to columns [ooff, ooff+H)               */
    float* gates;        /* [B,T,H,4] saved (r,z,n,Whn h + bhn) per unit for backward (16-B aligned); NULL = skip */
    float* h_n;          /* [B,H] final hidden state (NULL = skip)                         */
    int H, reverse, ldx, xoff, ldo, ooff;
} m3t_gru_fwd_desc;

/* ws (optional, 16-B aligned): scratch for the fragment-ordered fast paths -- per scan 3*H*H floats of re-laid
 * weights + 4 * ceil32(B) * H floats of ping-pong state / exchange granules; without it (or when H % 16 != 0) a
 * slower kernel that reads the row-major operands directly is used.  Results are identical on every path, except that the persistent FORWARD scan
 * at H = 256 / 512 runs its recurrent product as bf16x6 (fp32-accurate, ~1e-6 from the fp32-MFMA kernels; flag
 * M3T_SCAN_FP32 or env M3T_SCAN_X6=0 keeps fp32 MFMAs and bit-identity).
 * Execution: a level whose scans all have H = 128 runs ONE launch with one workgroup per (scan, clip) and nothing exchanged
 * between workgroups (gru_solo.hip: fp32 FMA chains, equal to the other paths to fp32 rounding; no residency requirement;
 * flags M3T_SCAN_NO_PERSIST / M3T_SCAN_FP32 or env M3T_SCAN_SOLO=0 keep the paths below).  Otherwise,
 * when every H of the level is a multiple of 128 (<= 512) and the level fits the chip at one workgroup
 * per CU, ONE persistent launch runs all T steps (W_hh held in registers, h_t exchanged between CUs through tagged
 * granules); otherwise one launch per time step.  A persistent launch needs all its workgroups resident: never run
 * two of them concurrently on one device (flags = M3T_SCAN_NO_PERSIST for scans issued on a side stream; env
 * M3T_SCAN_PERSIST=0 disables globally).  One process per device: the first process that launches a persistent scan on a
 * GPU takes an advisory lock ($XDG_RUNTIME_DIR or /tmp/m3t-<uid>/ m3t_persist_<pci-bus-id>.lock, held until it exits); any
 * other process on that GPU gets the launch-per-step path, says so on stderr, and m3t_gru_persist_owner() returns 2 there
 * (env M3T_SCAN_LOCK=0 disables the guard).
 * Error model.  Every wait of a persistent scan is bounded (M3T_SCAN_SPIN_LIMIT gather attempts, default 2^21 ~ 2 s).  A
 * workgroup whose wait expires raises a STICKY host-mapped flag and the scan finishes with INVALID results.  The flag stays
 * set until m3t_gru_error_reset(), which the host may call only after it has synchronised the device.  While it is set:
 * (a) m3t_gru_poll_error() returns non-zero; (b) every m3t_gru_scan_* call returns M3T_ESPIN; (c) m3t_grad_norm_scale, which
 * reads the flag ON THE DEVICE in stream order, zeroes the gradient buffer and returns norm = NaN, and m3t_adam_step /
 * m3t_sgd_step skip an update whose `guard` scalar is not finite.  Because the host never clears the flag on a read, every
 * step queued behind the dead scan is skipped however far ahead of the GPU the host runs: a dead scan can never reach the
 * parameters, with no host synchronisation.  Recovery = synchronise, m3t_gru_error_reset(), redo the step.
 * Several ranks: m3t_grad_poison / m3t_grad_dead_check (below) carry the failure of one rank through the gradient
 * all-reduce to every rank. */
int m3t_gru_scan_fwd(const m3t_gru_fwd_desc* scans, int n_scans, int B, int T,
                     float* ws, size_t ws_bytes, int flags, void* stream);

/* BPTT of the scans above (autograd of nn.GRU).  Per scan:
 *   dgx[B,T,ldg][goff..goff+3H) = grad wrt xproj  = (dr~, dz~, dn~)
 *   dgh[B,T,3H]                 = grad wrt W_hh h + b_hh = (dr~, dz~, dn~ * r)
 * The caller finishes with m3t_sgemm / m3t_colsum: dW_hh = sum dgh_t^T h_{t-1},
 * db_hh = colsum(dgh), dW_ih = dgx^T x, db_ih = colsum(dgx), dx = dgx W_ih. */
typedef struct {
    const float* dout;    /* [B,T,ldo]: grad wrt out, columns [ooff, ooff+H)               */
    const float* out;     /* forward h_t, same layout                                      */
    const float* gates;   /* [B,T,H,4] from forward                                        */
    const float* w_hh_t;  /* [H,3H] = W_hh transposed (m3t_transpose)                      */
    const float* dh_n;    /* [B,H] grad wrt final hidden state, or NULL                    */
    float* dgx;           /* [B,T,ldg], columns [goff, goff+3H)                            */
    float* dgh;           /* [B,T,3H]                                                      */
    float* dh;            /* [B,H] scratch: running dL/dh_t                                */
    float* db_part;       /* optional [B,4,H] scratch: per-clip sums over t of (dr~, dz~, dn~, dn~*r); with it the  */
    float* db_ih;         /* scan also delivers db_ih [3H] = sum_b (dr~,dz~,dn~) and db_hh [3H] = sum_b (dr~,dz~,   */
    float* db_hh;         /* dn~*r) -- the bias gradients, without a pass over dgx / dgh (all three NULL = skip)    */
    int H, reverse, ldo, ooff, ldg, goff;
    unsigned long long* amax;  /* optional magnitude slot (m3t_sgemm_scaled): raised to the bits of max |dgx|, |dgh| of this scan */
    const float* wfrag;        /* optional: W_hh already in the backward scan's fragment order (m3t_gru_bwd_prepare) -- the launch then   */
                               /* runs no preparation kernel in front of the scan; used only when EVERY scan of the call brings one       */
} m3t_gru_bwd_desc;

/* W_hh does not change between the forward and the backward pass of a step: the backward scans' weight re-layout (one launch in front of
 * every backward scan, on the critical chain) can be done any time after the weights are final -- e.g. on an idle stream during the forward
 * pass.  m3t_gru_bwd_prepare writes, for n <= M3T_MAX_SCANS scans of one H, the fragments the wide producer-split backward kernel reads
 * (two fp16 terms of the per-slice scaled W_hh^T + the slices' inverse scales) into out[i] (m3t_gru_bwd_prepare_floats(H) floats each, 16-B
 * aligned); whh_direct = 1: w[i] is W_hh [3H, H] as stored (M3T_SCAN_WHH), 0: its transpose [H, 3H].  A later m3t_gru_scan_bwd whose descs
 * all carry desc.wfrag = out[i] and whose launch takes that kernel (m3t_gru_scan_progress_ok(..., backward = 1) says so) skips its own
 * preparation; any other launch ignores the field.  H % 256 == 0. */
size_t m3t_gru_bwd_prepare_floats(int H);
int m3t_gru_bwd_prepare(const float* const* w, int n, int H, int whh_direct, float* const* out, void* stream);

/* Workgroups (= CUs: a scan workgroup owns its CU) that m3t_gru_scan_fwd (backward = 0) / m3t_gru_scan_bwd (backward = 1) would hold
 * resident for ONE persistent launch over n_scans scans of hidden size H at batch B with `flags`; 0 when that level does not run as a
 * persistent launch that needs residency (launch-per-step path, solo kernels, M3T_SCAN_NO_PERSIST).  Two persistent launches may run
 * at the same time iff the sum of their answers fits the device's CUs, per XCD (ceil(answer / 8) each, 32 CUs per XCD on MI355X):
 * then both grids become resident whatever else is draining from the CUs, and neither can wait for the other forever.  No device work. */
int m3t_gru_scan_workgroups(int n_scans, int H, int B, int T, int flags, int backward);
/* Number of one-launch scans (persistent or solo) this process has issued so far (tests use it to assert which path ran). */
int m3t_gru_persist_count(void);
/* 0, or (step + 1) of a persistent scan that gave up waiting since the last m3t_gru_error_reset() (sticky: reading does
 * NOT clear it).  The words are host-mapped: no synchronisation happens here, so synchronise the scan's stream first if the
 * answer must cover it. */
int m3t_gru_poll_error(void);
/* Clears the scan error state.  ONLY after the device has been synchronised (nothing queued may still read the flag). */
int m3t_gru_error_reset(void);
/* Several ranks (m3t.ddp): on = 1 makes m3t_gru_scan_fwd / _bwd launch even while the sticky flag is set instead of returning
 * M3T_ESPIN.  A rank that raised in the middle of a step -- the moment ITS host happened to see the flag -- would leave its peers
 * waiting in that step's collective; the device-side guards (m3t_grad_norm_scale, the optimizer steps) skip every step queued behind
 * the dead scan anyway, m3t_grad_poison / m3t_grad_dead_check carry the failure to every rank through the all-reduce, and the host
 * raises on ALL ranks at the same step from the all-reduced dead slot.  m3t_gru_poll_error() is unaffected.  Returns 0. */
int m3t_gru_error_defer(int on);
/* Fault injection (tests): a one-thread kernel on `stream` raises the scan error flag exactly as a dying scan would. */
int m3t_gru_inject_error(void* stream);
/* Who owns the persistent scans of the current device: 0 not decided yet (no persistent scan attempted), 1 this process,
 * 2 another process (this one runs the launch-per-step kernels). */
int m3t_gru_persist_owner(void);
/* Exchange arena (optional, speed only).  A persistent scan exchanges h_t / dgh_t between workgroups through tagged granules
 * and must never meet a stale granule whose tag matches; without an arena every launch therefore zeroes its exchange buffers
 * (3-4 fill kernels in front of every scan).  m3t_gru_scan_arena(arena, bytes): the NEXT m3t_gru_scan_fwd / _bwd call of the
 * calling thread keeps its exchange buffers in `arena` (device memory, 16-B aligned, >= 8 MiB + 64 KiB) and draws launch-unique
 * tags from a per-arena counter instead -- no fill kernel, except when the arena is first seen and when a 16-bit tag counter wraps
 * (every ~200 launches).  With an arena the launch also runs a placement handshake: workgroups publish the XCD they sit on, and a
 * group (one scan x one row block) whose members all share an XCD exchanges through that XCD's L2 (plain stores + sc1 loads)
 * instead of through the memory side -- 30-40 % less HBM / fabric traffic per launch, same results; used by the
 * launches whose groups are XCD-aligned by construction (a multiple of 8 groups).  CONTRACT: nothing but scan launches may ever write the arena, and launches that share an arena
 * must be ordered (one stream).  m3t_gru_scan_arena_reset(arena): forget what is known about `arena` (call it when the
 * memory was reallocated or written by anything else; NULL = every arena).  Both return 0. */
int m3t_gru_scan_arena(void* arena, size_t bytes);
int m3t_gru_scan_arena_reset(void* arena);
/* PROGRESS MARKS (round 5): consumers of a persistent scan's results need not wait for the launch to end.  Layer l+1's input projection
 * (reference models/rnn.py:17,75) is out_fwd W_ih[:, :H]^T + out_rev W_ih[:, H:]^T: each half needs ONE direction's scan, and the frames
 * that scan has passed are final.  m3t_gru_scan_progress(counters, n_marks, time_bounds, need): the NEXT m3t_gru_scan_fwd / _bwd call of the
 * calling thread publishes its progress through `counters` (two 32-bit device words, 8-B aligned, zero-initialised ONCE by the caller and
 * from then on written by scan launches only): counters[0] counts for the scans of the call that walk time upwards (forward pass:
 * reverse = 0; backward pass: reverse = 1), counters[1] for those that walk it downwards.  time_bounds[0 .. n_marks) (n_marks <= 3,
 * ascending, 4 <= tb, tb <= T - 4, at least 4 apart) cut [0, T) into n_marks + 1 windows.  Each workgroup adds 1 to its counter once its
 * results -- out / gates (forward), dgx / dgh and the per-window magnitude slots (backward) -- of every frame on the finished side of a bound
 * are in memory and visible device-wide.  On return need[k] (k < n_marks) is the value counters[0] reaches when the upward scans have
 * finished frames [0, time_bounds[k]); need[n_marks + k] the value counters[1] reaches when the downward scans have finished frames
 * [time_bounds[n_marks - 1 - k], T).  A consumer stream waits with m3t_stream_wait_progress (a one-lane gate kernel that polls the word;
 * what follows it starts behind a kernel boundary and reads the written-back data); the last window of each direction is final when the
 * scan launch ends (ordinary stream order / events).  Backward calls with marks armed write their magnitude slots as an ARRAY:
 * desc.amax[0] the whole scan (final at the end of the launch), desc.amax[1 + w] the w-th window IN THE ORDER THE SCAN WALKS THEM.
 * Only the launches whose kernels carry the marks accept an armed call (m3t_gru_scan_progress_ok: the fp16x3 persistent forward kernels,
 * the wide producer-split backward kernel); any other path returns M3T_EINVAL rather than leave consumers waiting.  Consumed by that call
 * (also when it fails); counters == NULL disarms.  m3t_gru_scan_progress_reset(counters): forget the library's count of what it has asked
 * of `counters` (after the caller re-zeroed or freed them; NULL = all).  Every wait is bounded (M3T_SCAN_SPIN_LIMIT): a gate that gives
 * up raises the sticky scan error. */
int m3t_gru_scan_progress(unsigned* counters, int n_marks, const int* time_bounds, unsigned* need);
int m3t_gru_scan_progress_reset(unsigned* counters);
int m3t_gru_scan_progress_ok(int n_scans, int H, int B, int T, int flags, int backward);
int m3t_stream_wait_progress(const unsigned* counter, unsigned need, void* stream);
/* Ordering between scans on different streams without holding back their preparation: the NEXT m3t_gru_scan_fwd /
 * m3t_gru_scan_bwd call of the calling thread makes its stream wait for `event` (a hipEvent_t) right before it launches
 * its scan kernel(s); the weight re-layout kernels and memsets it issues first run as soon as the stream allows.  Used
 * by the host schedule that alternates the persistent scans of two streams (they must never run at the same time).
 * The event is consumed by that call (also when it fails).  Returns 0. */
int m3t_gru_scan_after(void* event);
/* Timing hook: the NEXT m3t_gru_scan_fwd / m3t_gru_scan_bwd call of the calling thread records `start` right before and
 * `end` right after its scan kernel(s) on its stream (two hipEvent_t created with timing enabled) -- not around its weight
 * re-layout kernels, memsets or the m3t_gru_scan_after fence.  bench.py's roofline.achieved uses it.  Returns 0. */
int m3t_gru_scan_events(void* start, void* end);
/* With env M3T_SCAN_PROF=1: s_memtime cycles that workgroup 0 / lane 0 of the LAST persistent launch spent per phase,
 * summed over its T steps: [0] step top, [1] gather (wait for peers), [2] MFMA + LDS partials, [3] barrier,
 * [4] reduce + gate math + publish, [5] stores.  Synchronises the device.  M3T_EINVAL when profiling is off. */
int m3t_gru_persist_profile(unsigned long long* out6);

/* ws as above: per scan 3*H*H + 8 * ceil32(B) * H floats; flags as above. */
int m3t_gru_scan_bwd(const m3t_gru_bwd_desc* scans, int n_scans, int B, int T,
                     float* ws, size_t ws_bytes, int flags, void* stream);

/* ---------------------------------------------------------------------------------
 * Attention-fusion reduction (models/att_fusion.py:21-25) on [B*T] frames of D floats:
 *   w = softmax([sigmoid(s_v), sigmoid(s_a)]);  f = w0 * x_v + w1 * x_a   (index 0 = VIDEO)
 * s_v, s_a: [rows] raw scorer outputs. */
int m3t_att_fuse_fwd(const float* s_v, const float* s_a, const float* x_v, const float* x_a,
                     float* f, int rows, int D, void* stream);
int m3t_att_fuse_bwd(const float* df, const float* s_v, const float* s_a,
                     const float* x_v, const float* x_a,
                     float* ds_v, float* ds_a, float* dx_v, float* dx_a,
                     int rows, int D, void* stream);

/* ---------------------------------------------------------------------------------
 * Loss of AffWild2VA.training_step (models/model.py:132-141,146-182), forward and
 * gradient in one pass over y_hat [rows, C]:
 *   L = w_v * (1 - ccc(y[:,iv], valence)) + w_a * (1 - ccc(y[:,ia], arousal))
 *       + expr_w * mean_rows( CE(y[:, :n_expr], class_expr) * expr_valid )     (n_expr>0)
 * (training_step uses w_v = loss_lambda, w_a = 1 - loss_lambda, expr_w = 0.8; a zero weight
 * skips its term entirely)
 * ccc per models/utils.py:6-17 (unbiased variance, biased covariance).  With
 * use_mse!=0 the two ccc terms become mean squared errors (models/model.py:143-144).
 * The CE term is dropped when no row is valid (model.py:173-174) -- decided on device.
 * out_scalars[8] = {loss, loss_v, loss_a, loss_expr, n_valid, n_correct, ccc_v, ccc_a};
 * dy [rows, C] receives dL/dy_hat (fully written).  class_expr int64, expr_valid uint8. */
int m3t_va_loss(const float* y_hat, int rows, int C, int iv, int ia,
                const float* valence, const float* arousal,
                const int64_t* class_expr, const uint8_t* expr_valid, int n_expr,
                float w_v, float w_a, float expr_w, int use_mse,
                float* out_scalars, float* dy, float* ws, size_t ws_bytes, void* stream);
/* ws: 32 floats per 256 rows (m3t_va_loss_ws_bytes), 8-B aligned.  With it, 1024 < rows <= 32768 run as ONE grid-wide launch (round 6): raw
 * moments in fp64 in one sweep, the blocks meet once inside the kernel (agent-scope release + ticket, bounded wait: a wait that expires gives
 * loss = NaN), every block sums all partials in block order (deterministic) and writes its rows of dL/dy; M3T_VA_LOSS_FUSED=0 or more
 * rows: three short launches (sums -> centred moments -> closed form + gradient); without ws, or for rows <= 1024, one workgroup does all
 * passes. */
size_t m3t_va_loss_ws_bytes(int rows);

/* Round 6: channels-last operators of the 3-D VGG-M stems (csrc/stem_cl.hip; reference models/backbone.py:73-103,179-271: Conv3d -> BatchNorm3d ->
 * ReLU (-> MaxPool3d((1, 2, 2)))).  The tap walks read and write channels-last rows [N T H W][C]; with these two the whole stem stays in that
 * layout -- no planes <-> channels-last transpose between the video and the GRU input.
 * m3t_bn_cl_fwd / _bwd: BatchNorm (+ fused ReLU) over rows x [M][C] at any M (semantics of m3t_bn_rows_fwd / _bwd: batch statistics with
 *   torch's running-statistics update when training, fp64 per-chunk partials reduced in a fixed order); C % 4 == 0, C / 4 divides 256
 *   (C = 64 ... 1024 in powers of two), 16-B aligned tensors; ws: m3t_bn_cl_ws_bytes(M, C) bytes, 8-B aligned.  m3t_amax_out arms the
 *   magnitude slot of y (forward) / dx (backward): the next tap walk scales by it.  dx_colsum [C] (optional): the column sums of dx -- the bias
 *   gradient of the convolution in front of this BatchNorm -- from the dx pass itself (block partials summed in block order).
 * m3t_pool_cl_fwd / _bwd: max pooling of P frames x [P][H][W][C] with a (kh, kw) window, stride (sh, sw), padding (ph, pw) -- nn.MaxPool3d((1,
 *   kh, kw)) on channels-last rows; win [P][Ho][Wo][C] bytes: the winner's place in its window (ties: the first maximum in window order, NaN
 *   wins, as torch); backward is a gather (no atomics, deterministic).  C % 4 == 0, kh kw <= 255, padding < window.  m3t_amax_out arms y's slot.
 * m3t_bn_pool_cl_fwd / _bwd: BatchNorm + ReLU + max pooling with a k x k window, stride k, no padding (k = 2, 3: every pooling layer of the
 *   VGG-M stems) as ONE operator: relu(bn(x)) at full resolution is neither written nor read -- forward keeps the pooled frames yp [P][H / k][W /
 *   k][C] and the winner bytes, backward takes d(yp) and writes dx [P][H][W][C] (an input position's gradient through the pooling is d(yp) of
 *   its window if it won and yp > 0, else 0; positions no window covers get the BatchNorm terms only).  ws, slots, dx_colsum as m3t_bn_cl_*
 *   with M = P H W. */
size_t m3t_bn_cl_ws_bytes(size_t M, int C);
int m3t_bn_pool_cl_fwd(const float* x, size_t P, int H, int W, int C, int k, const float* gamma, const float* beta, float* run_mean, float* run_var,
                       float momentum, float eps, int training, float* yp, unsigned char* win, float* save_mean, float* save_invstd, float* ws,
                       size_t ws_bytes, void* stream);
int m3t_bn_pool_cl_bwd(const float* dyp, const float* x, const float* yp, const unsigned char* win, const float* gamma, const float* save_mean,
                       const float* save_invstd, size_t P, int H, int W, int C, int k, int training, float* dx, float* dgamma, float* dbeta,
                       float* dx_colsum, float* ws, size_t ws_bytes, void* stream);
int m3t_bn_cl_fwd(const float* x, size_t M, int C, const float* gamma, const float* beta, float* run_mean, float* run_var, float momentum,
                  float eps, int training, int relu, float* y, float* save_mean, float* save_invstd, float* ws, size_t ws_bytes, void* stream);
int m3t_bn_cl_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* save_mean, const float* save_invstd, size_t M,
                  int C, int training, int relu, float* dx, float* dgamma, float* dbeta, float* dx_colsum, float* ws, size_t ws_bytes, void* stream);
int m3t_pool_cl_fwd(const float* x, size_t P, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, float* y, unsigned char* win,
                    void* stream);
int m3t_pool_cl_bwd(const float* dy, const unsigned char* win, size_t P, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                    float* dx, void* stream);

/* ---------------------------------------------------------------------------------
 * TCN (models/tcn.py).  Activations are channel-last [B,T,C] inside the library.
 * weight-norm reparametrisation (torch.nn.utils.weight_norm at models/tcn.py:19-20):
 *   w[co][ci][j] = g[co] * v[co][ci][j] / ||v[co]||;  output layout w_t[j][co][ci]
 *   (tap-major, so each tap is a K-contiguous [C_out,C_in] matrix); norm[co] saved. */
int m3t_weight_norm_fwd(const float* v, const float* g, float* w_t, float* norm,
                        int Co, int Ci, int K, void* stream);
/* dw_t [K][Co][Ci] -> dv [Co][Ci][K], dg [Co] */
int m3t_weight_norm_bwd(const float* dw_t, const float* v, const float* g, const float* norm,
                        float* dv, float* dg, int Co, int Ci, int K, void* stream);

/* Dilated causal conv (Conv1d(pad=(k-1)d, dil=d) + Chomp1d, models/tcn.py:7-13,19-21),
 * channel-last, left zero padding, halo staged in LDS:
 *   y[b,t,co] = act( bias[co] + sum_{j,ci} w_t[j][co][ci] * x[b, t-(K-1-j)*d, ci] (+ res[b,t,co]) )
 * act: 0 none, 1 ReLU, 2 ReLU(ReLU(conv) + res) (the block output, models/tcn.py:46);
 * pre (optional, [B,T,Co]) receives the pre-activation conv output (+bias) for backward.
 * drop_mask (optional, [B,T,Co], already scaled by 1/(1-p)) multiplies the ReLU output
 * (nn.Dropout after relu1/relu2 in train mode, models/tcn.py:23,29); NULL in eval mode.
 * anticausal!=0 flips the time direction (x[b, t+(K-1-j)*d]) -- the data-gradient conv. */
int m3t_causal_conv_fwd(const float* x, const float* w_t, const float* bias, const float* res,
                        const float* drop_mask, float* y, float* pre,
                        int B, int T, int Ci, int Co, int K, int dilation,
                        int act, int anticausal, void* stream);
/* dw_t[j][co][ci] = sum_{b,t} dy[b,t,co] * x[b, t-(K-1-j)*d, ci] */
int m3t_causal_conv_wgrad(const float* dy, const float* x, float* dw_t,
                          int B, int T, int Ci, int Co, int K, int dilation,
                          float* ws, size_t ws_bytes, void* stream);
/* General 1-D conv over channel-last rows with `lead` frames of look-ahead (0 <= lead <= (K-1)*d):
 *   y[b,t,co] = act( bias[co] + sum_{j,ci} w_t[j][co][ci] * x[b, t + lead - (K-1-j)*d, ci] ... )
 * lead = 0 is m3t_causal_conv_fwd; lead = pad is nn.Conv1d(k, stride 1, padding=pad) with 2*pad = (K-1)*d --
 * the `tcn_simple` back-end's Conv1d(.,.,5,1,2) / Conv1d(.,.,3,1,1) (reference models/backbone.py:107-111,
 * 214-231).  anticausal != 0 flips time (x[b, t - lead + (K-1-j)*d]): the data gradient.
 * flags: 0 or M3T_BF16 (x and w rounded to bf16 while staged; fp32 accumulate and epilogue).
 * Dropout (nn.Dropout(p) behind each ReLU of the TemporalBlock, reference models/tcn.py:17,23,29): either an explicit
 * pre-scaled mask tensor `drop_mask` [B*T, Co], or drop_p in (0, 1) with drop_mask = NULL: the mask is then generated IN the
 * epilogue -- element (row, col) keeps its value x 1/(1-p) iff word (row & 3) of Philox4x32-10(counter {col, row >> 2, 0, 0},
 * key drop_seed) < (1-p) * 2^32 -- and never exists in memory; m3t_mask_pos_drop regenerates it for the backward pass.
 * Interior shapes ((B*T) % 128 == 0, Co % 128 == 0, Ci % 32 == 0, 16-B aligned operands) run as an implicit GEMM on the
 * bf16 matrix pipe (fp32-accurate bf16x6 products; one bf16 product with M3T_BF16); env M3T_CONV_X6=0 keeps the fp32-MFMA kernel. */
int m3t_conv1d_fwd(const float* x, const float* w_t, const float* bias, const float* res,
                   const float* drop_mask, float* y, float* pre,
                   int B, int T, int Ci, int Co, int K, int dilation, int lead, int act, int anticausal,
                   float drop_p, unsigned long long drop_seed, int flags, void* stream);
/* dw_t[j][co][ci] = sum_{b,t} dy[b,t,co] * x[b, t + lead - (K-1-j)*d, ci] */
int m3t_conv1d_wgrad(const float* dy, const float* x, float* dw_t,
                     int B, int T, int Ci, int Co, int K, int dilation, int lead,
                     float* ws, size_t ws_bytes, int flags, void* stream);
/* m3t_conv1d_fwd / m3t_conv1d_wgrad with the callers' magnitude slots for M3T_GEMM_F16X3 (see m3t_sgemm_scaled; NULL = measured by the
 * library): amax_x over x [B*T][Ci], amax_w over w_t, amax_dy over dy [B*T][Co].  The K taps of a weight gradient share both. */
int m3t_conv1d_fwd_scaled(const float* x, const float* w_t, const float* bias, const float* res,
                          const float* drop_mask, float* y, float* pre,
                          int B, int T, int Ci, int Co, int K, int dilation, int lead, int act, int anticausal,
                          float drop_p, unsigned long long drop_seed, int flags,
                          const unsigned long long* amax_x, const unsigned long long* amax_w, void* stream);
int m3t_conv1d_wgrad_scaled(const float* dy, const float* x, float* dw_t,
                            int B, int T, int Ci, int Co, int K, int dilation, int lead,
                            float* ws, size_t ws_bytes, int flags,
                            const unsigned long long* amax_dy, const unsigned long long* amax_x, void* stream);

/* BatchNorm1d (+ optional fused ReLU) over channel-last rows x [M = B*T, C]
 * (nn.BatchNorm1d(512) + nn.ReLU(True) of `tcn_simple`, reference models/backbone.py:108-110, 217-222).
 * training != 0: batch statistics (biased variance for normalisation; run_mean/run_var, when given, are
 * updated in place with `momentum` and the unbiased variance, as torch); training == 0: running statistics.
 * save_mean / save_invstd [C] receive the statistics used (needed by the backward).
 * ws: m3t_bn_rows_ws_bytes(M, C) bytes, 8-byte aligned (fp64 per-chunk partial sums, fixed-order reduce). */
size_t m3t_bn_rows_ws_bytes(int M, int C);
int m3t_bn_rows_fwd(const float* x, int M, int C, const float* gamma, const float* beta,
                    float* run_mean, float* run_var, float momentum, float eps, int training, int relu,
                    float* y, float* save_mean, float* save_invstd, float* ws, size_t ws_bytes, void* stream);
/* g = dy * (y > 0) when relu; dbeta = sum g, dgamma = sum g*xhat,
 * training: dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat));  eval: dx = g*gamma*invstd */
int m3t_bn_rows_bwd(const float* dy, const float* x, const float* y, const float* gamma,
                    const float* save_mean, const float* save_invstd, int M, int C, int training, int relu,
                    float* dx, float* dgamma, float* dbeta, float* ws, size_t ws_bytes, void* stream);
/* BatchNorm3d (+ optional fused ReLU) of the 3-D conv stems, channels-first x [N][C][S] with S = T*H*W contiguous (nn.BatchNorm3d(C) + nn.ReLU(True),
 * reference models/backbone.py:73-103,179-191): same statistics / running-statistics / gradient formulas as the rows form above with M = N*S values per
 * channel.  ws: m3t_bn_planes_ws_bytes(N, C, S) bytes, 8-byte aligned.  (One workgroup per plane chunk of <= 8192 floats: made for the stems'
 * large planes; planes of a few dozen floats work but waste the workgroup.) */
size_t m3t_bn_planes_ws_bytes(int N, int C, int S);
int m3t_bn_planes_fwd(const float* x, int N, int C, int S, const float* gamma, const float* beta, float* run_mean, float* run_var,
                      float momentum, float eps, int training, int relu, float* y, float* save_mean, float* save_invstd,
                      float* ws, size_t ws_bytes, void* stream);
int m3t_bn_planes_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* save_mean, const float* save_invstd,
                      int N, int C, int S, int training, int relu, float* dx, float* dgamma, float* dbeta, float* ws, size_t ws_bytes,
                      void* stream);
/* nn.MaxPool3d with a (1, kh, kw) window (stride (1, sh, sw), padding (0, ph, pw), floor mode: the 3-D stems' pooling layers, reference
 * models/backbone.py:80,86,92,182) as the 2-D pooling of P = N*C*T planes x [P][H][W] -> y [P][Ho][Wo].  win [P][Ho][Wo] (one byte per
 * output): the winner's position inside its window, dh * kw + dw (kh * kw <= 255), for the backward pass, which is a gather (no atomics:
 * overlapping windows are deterministic).  Ties: the first maximum in row-major window order; NaN wins (torch's kernel). */
int m3t_pool_planes_fwd(const float* x, long long P, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw, float* y,
                        unsigned char* win, void* stream);
int m3t_pool_planes_bwd(const float* dy, const unsigned char* win, long long P, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw,
                        float* dx, void* stream);
/* [B,C,T] <-> [B,T,C] */
int m3t_bct_to_btc(const float* src, float* dst, int B, int C, int T, void* stream);
int m3t_btc_to_bct(const float* src, float* dst, int B, int T, int C, void* stream);
/* m3t_bct_to_btc that also leaves, in part [B * ceil(T / 32)][C], the sums of every channel row over each tile of 32 positions: the column
 * sums of `part` (m3t_colsum) are the per-channel sums of src over (B, T) -- the conv3d bias gradient without another pass over dy */
int m3t_bct_to_btc_sums(const float* src, float* dst, int B, int C, int T, float* part, void* stream);
/* ... straight to the m3t_f16x3_split IMAGE of the channels-last rows (round 6): for a source whose magnitude slot its producer raised
 * (m3t_bn_planes_fwd / _bwd take m3t_amax_out for their y / dx), so that transpose and split are one pass; part as m3t_bct_to_btc_sums
 * or NULL.  C % 4 == 0, 16-B aligned destination. */
int m3t_bct_to_btc_img(const float* src, float* dst_img, int B, int C, int T, const unsigned long long* slot, float* part, void* stream);
/* out[i] = s[i] > 0 ? dy[i] * (mul ? mul[i] : 1) : 0   (ReLU / dropout gradient masks) */
int m3t_mask_pos(const float* s, const float* dy, const float* mul, float* out, size_t n, void* stream);
/* out[i] = max(a[i] + b[i], 0): the residual add + ReLU at the end of a ResNet block (reference models/resnet.py:52-54, 84-86) in one pass;
 * its gradient to both inputs is m3t_mask_pos(out, dout).  Takes m3t_amax_out (max |out|). */
int m3t_add_relu(const float* a, const float* b, float* out, size_t n, void* stream);
/* out[row, col] = s > 0 ? dy * mask(row, col) : 0 over [rows, C] with m3t_conv1d_fwd's in-kernel dropout mask regenerated from
 * (drop_p, drop_seed): the gradient through ReLU -> Dropout without a mask tensor. */
int m3t_mask_pos_drop(const float* s, const float* dy, float* out, int rows, int C, float drop_p,
                      unsigned long long drop_seed, void* stream);

/* ---------------------------------------------------------------------------------
 * CBAM (models/cbam.py), x [N,C,H,W] contiguous.
 * Channel gate (cbam.py:51-58): avg+max over H*W per (n,c) by wavefront shuffles,
 * shared MLP C -> C/r -> C, sigmoid, scale.  pooled [N,2,C] (avg,max), argmax [N,C] int32,
 * hidden [N,2,Cr] pre-ReLU, scale [N,C] are saved for backward. */
int m3t_cbam_channel_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                         float* y, float* pooled, int32_t* argmax, float* hidden, float* scale,
                         int N, int C, int Cr, int HW, void* stream);
int m3t_cbam_channel_bwd(const float* dy, const float* x, const float* w1, const float* w2,
                         const float* pooled, const int32_t* argmax, const float* hidden, const float* scale,
                         float* dx, float* dw1, float* db1, float* dw2, float* db2,
                         int N, int C, int Cr, int HW, float* ws, size_t ws_bytes, void* stream);
/* Spatial gate (cbam.py:61-92): per-pixel (max,mean) over C -> 5x5 conv 2->1 (pad 2) ->
 * BatchNorm2d(1) (training: batch statistics, running stats updated with `momentum`,
 * unbiased running variance) -> sigmoid -> scale.
 * bn = {gamma, beta}; running = {mean, var} (updated in place when training).
 * saved: comp [N,2,HW], cargmax [N,HW] int32, xhat [N,HW], stats[2] = {mean, invstd}, scale [N,HW]. */
int m3t_cbam_spatial_fwd(const float* x, const float* conv_w, const float* bn, float* running,
                         float* y, float* comp, int32_t* cargmax, float* xhat, float* stats, float* scale,
                         int N, int C, int H, int W, int training, float momentum, float eps,
                         float* ws, size_t ws_bytes, void* stream);
int m3t_cbam_spatial_bwd(const float* dy, const float* x, const float* conv_w, const float* bn,
                         const float* comp, const int32_t* cargmax, const float* xhat, const float* stats,
                         const float* scale, float* dx, float* dconv_w, float* dbn,
                         int N, int C, int H, int W, int training, float* ws, size_t ws_bytes, void* stream);

/* CBAM as ONE fused operator (reference models/cbam.py:95-111: SpatialGate(ChannelGate(x))), csrc/cbam_fused.hip: the
 * intermediate x * channel_scale is never written; forward = 2 sweeps (per-frame squeeze + MLP + compress + 5x5 conv | apply),
 * cut only by BatchNorm2d(1)'s global batch statistics, backward = 2 sweeps likewise: 8 HBM passes over a tensor of x's size
 * for forward + backward against 10 + for the two separate gates above, 6 launches + the two-launch parameter-gradient finish.
 * Frames of at most 64 KB with H*W <= 64 (the 7 x 7 and 4 x 4 ResNet-18 stages; C*H*W % 4 == 0, x / dy / dx 16-B aligned) take
 * frame-resident kernels (the frame parked in LDS, read once; environment M3T_CBAM_RESIDENT=0 keeps them on the general kernels):
 * same arithmetic per element, another fixed summation order.
 * Saved for backward: cs [N,C] (channel scale), argmax_p [N,C], pooled [N,2,C], hidden [N,2,Cr], comp [N,2,HW] (max, mean of
 * x * cs over channels), cargmax [N,HW], xhat [N,HW], ss [N,HW] (spatial scale), stats [2] (mean, 1/sqrt(var+eps)).
 * bn_w / bn_b = BatchNorm2d(1)'s gamma / beta (one float each), running_mean / running_var updated in place when training.  m3t_cbam_fused_ok: 1 when the
 * shape is covered (H*W/4 <= 512 float4 units, or H*W <= 512 pixels when H*W % 4 != 0); otherwise use the two gates above.
 * ws: forward 16 N bytes (8-B aligned), backward m3t_cbam_fused_ws_bytes (16-B aligned).  Results equal the two-operator
 * path to fp32 rounding (other summation order); deterministic. */
int m3t_cbam_fused_ok(int C, int Cr, int H, int W);
size_t m3t_cbam_fused_ws_bytes(int N, int C, int Cr, int H, int W);
int m3t_cbam_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                 const float* conv_w, const float* bn_w, const float* bn_b, float* running_mean, float* running_var,
                 float* y, float* cs, int32_t* argmax_p,
                 float* pooled, float* hidden, float* comp, int32_t* cargmax, float* xhat, float* ss, float* stats,
                 int N, int C, int Cr, int H, int W, int training, float momentum, float eps, float* ws,
                 size_t ws_bytes, void* stream);
int m3t_cbam_bwd(const float* dy, const float* x, const float* w1, const float* w2, const float* conv_w,
                 const float* bn_w, const float* cs, const int32_t* argmax_p, const float* pooled,
                 const float* hidden, const float* comp, const int32_t* cargmax, const float* xhat, const float* ss,
                 const float* stats, float* dx, float* dw1, float* db1, float* dw2, float* db2, float* dconv_w,
                 float* dbn_w, float* dbn_b, int N, int C, int Cr, int H, int W, int training, float* ws,
                 size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------
 * Data-parallel helpers (train.py:32-41: DDP mean of gradients + clip_grad_norm_(1.0)).
 * After the RCCL all-reduce(sum) of the flat gradient buffer: one pass computes
 * sum(g^2) of g/world (partials[blocks] + final), a second scales by
 * (1/world) * min(1, max_norm/(norm+1e-6)).  norm_out[0] = total norm (pre-clip). */
int m3t_grad_norm_scale(float* flat, size_t n, float inv_world, float max_norm,
                        float* norm_out, float* ws, size_t ws_bytes, void* stream);
/* One rank's dead scan must stop every rank: the all-reduce would otherwise spread its garbage while only the failing
 * rank's guard fires.  m3t_grad_poison, BEFORE the all-reduce: if this process's scan error flag is set, flat[0] = NaN (SUM
 * carries it to every rank: every clip norm is NaN, every fused optimizer step skips) and dead[0] = 1, else dead[0] = 0
 * (`dead`: one float of padding that rides in the same all-reduce; NULL = skip).  m3t_grad_dead_check, AFTER the
 * all-reduce: dead[0] != 0 raises this process's scan error flag as well, so every rank's next poll reports the failure
 * (a healthy rank would otherwise wait in its next collective for the rank that raised).  One thread each, stream-ordered. */
int m3t_grad_poison(float* flat, float* dead, void* stream);
int m3t_grad_dead_check(const float* dead, void* stream);

/* ---------------------------------------------------------------------------------
 * Optimizer steps over the flat parameter / gradient buffers (reference
 * models/model.py:388-394).  torch.optim semantics: Adam(lr, betas, eps, weight_decay as L2 on the gradient,
 * bias-corrected, `step` counts from 1); SGD(momentum, weight_decay), dampening 0, no Nesterov. */
int m3t_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, const float* guard, void* stream);
int m3t_sgd_step(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float weight_decay,
                 int step, const float* guard, void* stream);
/* guard: optional device scalar (the norm m3t_grad_norm_scale returned).  When it is not finite the kernel changes
 * nothing (parameters and optimizer state keep their values): a step whose gradients came from a failed scan, or
 * overflowed, is skipped on the device. */

/* ---------------------------------------------------------------------------------
 * Post-processing of prediction tracks (reference models/utils.py:20-33,
 * get_smoothed_ccc.py:13-28, create_submission.py:30-38).
 * n_tracks tracks are concatenated in x; track i = x[offsets[i] .. offsets[i+1]) (offsets: n_tracks+1 int64 on the
 * device).  mode 0 = scipy.signal.wiener(track, window), mode 1 = scipy.signal.medfilt(track, window): centred odd
 * window (<= 129), zero padding, fp64 arithmetic on the fp32 input (squares rounded to fp32 first, as scipy does);
 * y (fp64) has the layout of x. */
int m3t_smooth_tracks(const float* x, const long long* offsets, int n_tracks, int window, int mode, double* y,
                      void* stream);
/* out2[0] = numpy-style CCC (biased variances) of p against g over the frames with g >= -1 (and g2 >= -1 when g2 is
 * given): models/utils.py:20-22 as driven by get_smoothed_ccc.py:19-21; p_unbiased != 0 uses the unbiased variance
 * for p (what the reference computes when the smoothed track is a torch tensor).  out2[1] = number of valid frames. */
int m3t_ccc_masked(const double* p, const float* g, const float* g2, long long n, int p_unbiased, double* out2,
                   void* stream);

/* ---------------------------------------------------------------------------------
 * Audio front-end: the glue kernels of the log-Mel pipeline that the reference runs offline with
 * librosa (process/extract_melspec.py:13-20) and the context stacking of models/dataset.py:83-95.  The DFT and the
 * mel projection themselves are m3t_sgemm calls (see m3t/audio.py).
 * frames[f][k] = window[k] * ypad[f*hop + k], ypad = y with n_fft/2 samples of padding per side
 * (pad_mode 0 zeros, 1 reflect); n_frames = 1 + n / hop. */
int m3t_frame_window(const float* y, long long n, int n_fft, int hop, int pad_mode, const float* window,
                     float* frames, long long n_frames, void* stream);
/* spec [n_frames][2*bins] = (re | im) -> power [n_frames][bins] = re^2 + im^2 */
int m3t_power_spectrum(const float* spec, long long n_frames, int bins, float* power, void* stream);
/* librosa.power_to_db(S, ref=1.0, amin, top_db): 10 log10(max(amin, S)), floored at (max - top_db) when top_db >= 0.
 * ws: >= 2 KiB of scratch. */
int m3t_power_to_db(const float* s, long long n, float amin, float top_db, float* out, float* ws, size_t ws_bytes,
                    void* stream);
/* out [w_len][width*n_mels]: row i = mel rows (start+i)*step .. +width concatenated, zero rows past the end */
int m3t_stack_context(const float* mel, long long n_rows, int n_mels, long long start, int w_len, int step, int width,
                      float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* M3T_HIP_H */
