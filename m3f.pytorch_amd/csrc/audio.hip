// Audio front-end on the GPU: log-Mel spectrogram as the reference extracts it offline with librosa
// (process/extract_melspec.py:13-20: n_fft 512, win 400, hop int(16000 / (3 fps)), 40 mel bands, power_to_db) and the
// 5-frame context stacking of models/dataset.py:83-95.  The two contractions (DFT as a [frames x 512] x [512 x 514]
// product, mel projection) run on the fp32-accurate GEMM (m3t_sgemm); the kernels here are the HBM-bound glue:
// framing + window, |X|^2, 10 log10 with the top_db floor (global max by a fixed-order two-pass reduction), and the
// context gather.
#include "common.h"

namespace {

// frames[f][k] = window[k] * y_padded[f*hop + k], y_padded = y with n_fft/2 samples of padding on both sides
// (pad_mode 0: zeros = librosa >= 0.10 default "constant"; 1: reflect = the older default)
__global__ void frame_window_kernel(const float* __restrict__ y, long n, int n_fft, int hop, int pad_mode,
                                    const float* __restrict__ window, float* __restrict__ frames, long total) {
    const int half = n_fft >> 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long f = i / n_fft;
        const int k = (int)(i - f * n_fft);
        long j = f * hop + k - half;
        float v = 0.f;
        if (j >= 0 && j < n) v = y[j];
        else if (pad_mode == 1 && n > 1) {
            const long period = 2 * (n - 1);                 // numpy 'reflect': ... 2 1 | 0 1 2 ... n-1 | n-2 n-3 ...
            long m = j % period;
            if (m < 0) m += period;
            v = y[m < n ? m : period - m];
        }
        frames[i] = v * window[k];
    }
}

// spec [frames][2*bins] = (re | im) halves from the DFT product -> power [frames][bins]
__global__ void power_kernel(const float* __restrict__ spec, int bins, float* __restrict__ power, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long f = i / bins;
        const int b = (int)(i - f * bins);
        const float re = spec[f * 2 * bins + b], im = spec[f * 2 * bins + bins + b];
        power[i] = re * re + im * im;
    }
}

__global__ __launch_bounds__(256) void db_partial_max_kernel(const float* __restrict__ s, long n, float amin, float* __restrict__ part) {
    __shared__ float red[256];
    float m = -3.0e38f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        m = fmaxf(m, 10.0f * log10f(fmaxf(amin, s[i])));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + k]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// librosa.power_to_db(S, ref=1.0, amin, top_db): 10 log10(max(amin, S)) - 10 log10(max(amin, 1)), floored at max - top_db
__global__ void db_apply_kernel(const float* __restrict__ s, long n, float amin, float top_db, const float* __restrict__ part,
                                int nparts, float* __restrict__ out) {
    float mx = -3.0e38f;
    for (int k = 0; k < nparts; ++k) mx = fmaxf(mx, part[k]);
    const float ref = 10.0f * log10f(fmaxf(amin, 1.0f));
    const float floor_db = (mx - ref) - top_db;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = 10.0f * log10f(fmaxf(amin, s[i])) - ref;
        out[i] = top_db >= 0.f ? fmaxf(v, floor_db) : v;
    }
}

// models/dataset.py:83-95: out[i] = concat(mel[(start+i)*step .. +width)) with zero rows past the end of the track
__global__ void stack_context_kernel(const float* __restrict__ mel, long n_rows, int n_mels, long start, int w_len, int step,
                                     int width, float* __restrict__ out) {
    const long total = (long)w_len * width * n_mels;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % n_mels);
        const long r = i / n_mels;
        const int k = (int)(r % width);
        const long f = r / width;
        const long row = (start + f) * step + k;
        out[i] = (row >= 0 && row < n_rows) ? mel[row * n_mels + c] : 0.f;
    }
}

int grid_for(long total) {
    long b = (total + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int m3t_frame_window(const float* y, long long n, int n_fft, int hop, int pad_mode, const float* window,
                                float* frames, long long n_frames, void* stream) {
    if (n_frames <= 0) return 0;
    if (!y || !window || !frames || n <= 0 || n_fft <= 0 || (n_fft & 1) || hop <= 0 || (pad_mode != 0 && pad_mode != 1))
        return M3T_EINVAL;
    const long total = (long)n_frames * n_fft;
    frame_window_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(y, (long)n, n_fft, hop, pad_mode, window, frames, total);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_power_spectrum(const float* spec, long long n_frames, int bins, float* power, void* stream) {
    if (n_frames <= 0) return 0;
    if (!spec || !power || bins <= 0) return M3T_EINVAL;
    const long total = (long)n_frames * bins;
    power_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(spec, bins, power, total);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_power_to_db(const float* s, long long n, float amin, float top_db, float* out, float* ws, size_t ws_bytes,
                               void* stream) {
    if (n <= 0) return 0;
    if (!s || !out || !ws || amin <= 0.f) return M3T_EINVAL;
    int parts = grid_for((long)n);
    if (parts > 512) parts = 512;
    if (ws_bytes < (size_t)parts * sizeof(float)) return M3T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    db_partial_max_kernel<<<parts, 256, 0, st>>>(s, (long)n, amin, ws);
    M3T_LAUNCH_CHECK();
    db_apply_kernel<<<grid_for((long)n), 256, 0, st>>>(s, (long)n, amin, top_db, ws, parts, out);
    M3T_LAUNCH_CHECK();
    return 0;
}

extern "C" int m3t_stack_context(const float* mel, long long n_rows, int n_mels, long long start, int w_len, int step, int width,
                                 float* out, void* stream) {
    if (w_len <= 0) return 0;
    if (!mel || !out || n_rows < 0 || n_mels <= 0 || step <= 0 || width <= 0) return M3T_EINVAL;
    const long total = (long)w_len * width * n_mels;
    stack_context_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(mel, (long)n_rows, n_mels, (long)start, w_len, step, width, out);
    M3T_LAUNCH_CHECK();
    return 0;
}
