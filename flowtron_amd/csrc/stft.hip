// STFT magnitude + mel filterbank + log compression (reference audio_processing.py:207-235,
// :117-134).  The reference evaluates the DFT as a dense conv1d against a 1026x1024 basis
// (2.1 MFLOP/frame); here each workgroup stages the audio span of 8 consecutive frames in LDS
// once (one coalesced HBM read of 11 KB, reflect padding resolved on the fly), runs a radix-2
// FFT per frame entirely in LDS, applies the [n_mel, n_fft/2+1] filterbank with one wave per
// band group, and writes the 8-frame x n_mel tile so that consecutive frames of a band are
// adjacent in memory.  HBM traffic per frame = 256 new samples in + n_mel floats out.
#include "common.h"

namespace {

constexpr int FPB = 8;   // frames per workgroup

__global__ __launch_bounds__(256) void stft_mel_k(const float* __restrict__ y, const float* __restrict__ window,
                                                  const float* __restrict__ fb, float* __restrict__ mel,
                                                  int N, int n_fft, int log2n, int hop, int n_mel, int n_frames) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int half = n_fft >> 1, nb = half + 1;
    const int span = (FPB - 1) * hop + n_fft;
    float* xs = sm;                    // [span] audio samples (reflect-padded coordinates)
    float* re = xs + span;             // [n_fft]
    float* im = re + n_fft;            // [n_fft]
    float* twc = im + n_fft;           // [half] cos
    float* tws = twc + half;           // [half] -sin
    float* mag = tws + half;           // [nb (+pad)]
    float* mo = mag + ((nb + 3) & ~3); // [n_mel][FPB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, f0 = blockIdx.x * FPB;
    const float* yb = y + (size_t)b * N;

    for (int j = tid; j < span; j += 256) {
        int n = f0 * hop + j - half;               // position in the unpadded signal
        if (n < 0) n = -n;
        if (n >= N) n = 2 * (N - 1) - n;
        xs[j] = (n >= 0 && n < N) ? yb[n] : 0.f;
    }
    for (int k = tid; k < half; k += 256) {
        float s, c;
        sincospif(2.0f * (float)k / (float)n_fft, &s, &c);
        twc[k] = c; tws[k] = -s;
    }
    __syncthreads();

    for (int f = 0; f < FPB; ++f) {
        const int t = f0 + f;
        if (t >= n_frames) break;
        for (int j = tid; j < n_fft; j += 256) {
            const int r = (int)(__brev((unsigned)j) >> (32 - log2n));
            re[r] = xs[f * hop + j] * window[j];
            im[r] = 0.f;
        }
        __syncthreads();
        for (int s = 1; s <= log2n; ++s) {
            const int m = 1 << s, hm = m >> 1, tstep = n_fft >> s;
            for (int k = tid; k < half; k += 256) {
                const int grp = k / hm, pos = k - grp * hm;
                const int i0 = grp * m + pos, i1 = i0 + hm;
                const float wr = twc[pos * tstep], wi = tws[pos * tstep];
                const float xr = re[i1], xi = im[i1];
                const float tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
                const float ur = re[i0], ui = im[i0];
                re[i0] = ur + tr; im[i0] = ui + ti;
                re[i1] = ur - tr; im[i1] = ui - ti;
            }
            __syncthreads();
        }
        for (int k = tid; k < nb; k += 256) mag[k] = sqrtf(re[k] * re[k] + im[k] * im[k]);
        __syncthreads();
        for (int mbin = wave; mbin < n_mel; mbin += 4) {
            const float* fr = fb + (size_t)mbin * nb;
            float s = 0.f;
            for (int k = lane; k < nb; k += 64) s += fr[k] * mag[k];
            s = wave_sum(s);
            if (lane == 0) mo[mbin * FPB + f] = logf(fmaxf(s, 1e-5f));
        }
        __syncthreads();
    }
    const int nf = min(FPB, n_frames - f0);
    for (int idx = tid; idx < n_mel * FPB; idx += 256) {
        const int mbin = idx / FPB, f = idx - mbin * FPB;
        if (f < nf) mel[((size_t)b * n_mel + mbin) * n_frames + f0 + f] = mo[idx];
    }
}

}  // namespace

extern "C" int ft_stft_mel(const float* y, const float* window, const float* fb, float* mel,
                           int B, int N, int n_fft, int hop, int n_mel, void* stream) {
    FT_CHECK_ARG(y && window && fb && mel);
    FT_CHECK_ARG(B >= 1 && B <= 65535 && n_fft >= 64 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0);
    FT_CHECK_ARG(hop >= 1 && hop <= n_fft && n_mel >= 1 && N > n_fft / 2);
    int log2n = 0;
    while ((1 << log2n) < n_fft) ++log2n;
    const int n_frames = N / hop + 1;
    const int half = n_fft / 2, nb = half + 1;
    const size_t lds = sizeof(float) * ((size_t)(FPB - 1) * hop + n_fft + 2 * (size_t)n_fft + 2 * (size_t)half + ((nb + 3) & ~3) + (size_t)n_mel * FPB);
    if (lds > 160 * 1024) return ft_fail(FT_EUNSUPPORTED, "ft_stft_mel: n_fft=%d hop=%d needs %zu B of LDS", n_fft, hop, lds);
    FT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stft_mel_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(stft_mel_k, dim3(cdiv(n_frames, FPB), B), dim3(256), lds, reinterpret_cast<hipStream_t>(stream),
                       y, window, fb, mel, N, n_fft, log2n, hop, n_mel, n_frames);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
