// STFT magnitude / phase / mel for the reference's analysis setting n_fft = 1024 (audio_processing.py:207-235, :117-134;
// config.json:32-34) as the north star words it: a real FFT + sparse triangular filterbank, HBM-bound.
//
//   * rFFT-1024 = ONE 512-point complex FFT of z[m] = x[2m] + i x[2m+1] plus the split step
//       X[k] = (Z[k] + conj Z[512-k]) / 2  -  i e^{-2 pi i k / 1024} (Z[k] - conj Z[512-k]) / 2,   k = 0 .. 512
//     -- half the butterflies of the complex radix-2 transform in stft.hip.
//   * one WAVE per frame, 4 frames of a workgroup in flight at once: the 512-point FFT is three radix-8 passes with the 8 points
//     of a lane in REGISTERS (n = 64 j + 8 a + b, k = k1 + 8 k2 + 64 k3: DFT_8 over j, twiddle W512^{l k1}, transpose through
//     LDS, DFT_8 over a, twiddle W64^{b k2}, transpose, DFT_8 over b) -- 2 LDS transposes instead of 9 radix-2 LDS stages,
//     no workgroup barrier inside a frame (LDS operations of one wave execute in order).
//   * filterbank: the Slaney triangles overlap at most pairwise, so the dense [80][513] matrix has ~1 000 non-zeros; the host
//     hands them over as CSR (band -> first bin, weights) and a lane accumulates its band over 2 .. 60 consecutive bins.
//   * the audio span of the workgroup's 16 frames (4 864 samples) is staged in LDS once: 256 new samples in + 80 floats out
//     per frame = 1 344 B of HBM traffic.
#include "common.h"

namespace {

constexpr int NFFT = 1024, NH = 512, NB = 513;
constexpr int FPW = 4, FPG = 4 * FPW;                  // frames per wave / per workgroup
constexpr int SPAN = (FPG - 1) * 256 + NFFT;           // hop is a runtime argument <= 256 in the reference; sized for 256
constexpr int BWMAX = 2048;                            // CSR values staged in LDS when they fit (Slaney, 80 bands: ~1 000)

struct cpx { float re, im; };
__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cpx mul_mi(cpx a) { return {a.im, -a.re}; }                       // a * (-i)

// in-place forward DFT of 8 points (e^{-2 pi i jk/8}), natural order in and out
__device__ __forceinline__ void dft8(cpx (&v)[8]) {
    const float r = 0.70710678118654752f;
    cpx a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]), a2 = cadd(v[2], v[6]), a3 = mul_mi(csub(v[2], v[6]));
    cpx a4 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]), a6 = cadd(v[3], v[7]), a7 = mul_mi(csub(v[3], v[7]));
    cpx b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = csub(a1, a3);
    cpx b4 = cadd(a4, a6), b6 = mul_mi(csub(a4, a6)), b5 = cadd(a5, a7), b7 = csub(a5, a7);
    b5 = (cpx){r * (b5.re + b5.im), r * (b5.im - b5.re)};                                     // * e^{-i pi/4}
    b7 = (cpx){r * (b7.im - b7.re), -r * (b7.re + b7.im)};                                    // * e^{-3 i pi/4}
    v[0] = cadd(b0, b4); v[4] = csub(b0, b4);
    v[1] = cadd(b1, b5); v[5] = csub(b1, b5);
    v[2] = cadd(b2, b6); v[6] = csub(b2, b6);
    v[3] = cadd(b3, b7); v[7] = csub(b3, b7);
}

struct StftP {
    const float* y; const float* window;
    const int* band_bin0; const int* band_ptr; const float* band_w;      // CSR of the filterbank: band b covers bins
    float* mel; float* mag; float* phase;                                // [bin0[b], bin0[b] + ptr[b+1] - ptr[b])
    int N, hop, n_mel, n_frames;
    const int* n_samples;            // ragged batch (ft_stft_r8_ragged): utterance b holds n_samples[b] <= N samples and
    int ldt;                         // n_samples[b] / hop + 1 frames; frames beyond that are written as zeros; ldt = output row stride
};

__global__ __launch_bounds__(256) void stft_r8_k(StftP p) {
    __shared__ __attribute__((aligned(16))) float xs[SPAN];
    __shared__ __attribute__((aligned(16))) cpx tr[4][NH];               // per-wave transpose buffer
    __shared__ float mg[4][NB + 3];                                      // per-wave magnitudes
    __shared__ float mo[128][FPG + 1];                                   // mel tile of the workgroup's frames [band][frame]
    __shared__ float bw[BWMAX];                                          // the filterbank's non-zero weights (CSR values)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, f0 = blockIdx.x * FPG;
    const float* yb = p.y + (size_t)b * p.N;
    const int Nb = p.n_samples ? min(max(p.n_samples[b], 1), p.N) : p.N;   // this utterance's own length: the reflection is about ITS end
    const int nfb = p.n_samples ? Nb / p.hop + 1 : p.n_frames;
    const int span = (FPG - 1) * p.hop + NFFT;
    {                                                                     // all of a thread's loads in flight before the first LDS write
        constexpr int NL = (SPAN + 255) / 256;
        float xv[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int j = tid + 256 * u;
            int n = f0 * p.hop + j - NH;                                  // reflect padding (audio_processing.py:210-214)
            if (n < 0) n = -n;
            if (n >= Nb) n = 2 * (Nb - 1) - n;
            xv[u] = (j < span && n >= 0 && n < Nb) ? yb[n] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NL; ++u) if (tid + 256 * u < span) xs[tid + 256 * u] = xv[u];
    }
    // filterbank: this lane's bands (lane, lane + 64) and, when they fit, the CSR values in LDS -- the band loop below then reads
    // weight and magnitude from LDS eight bins at a time (one global load per bin, un-unrolled, was ~60 dependent L1 round trips
    // on the lanes that hold the widest bands: most of a frame's time)
    int bk0[2] = {0, 0}, bw0[2] = {0, 0}, bn[2] = {0, 0};
    bool w_lds = false;
    if (p.mel) {
        const int nnz = p.band_ptr[p.n_mel];
        w_lds = nnz <= BWMAX;
        if (w_lds) for (int j = tid; j < nnz; j += 256) bw[j] = p.band_w[j];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int mb = lane + 64 * h;
            if (mb < p.n_mel) { bk0[h] = p.band_bin0[mb]; bw0[h] = p.band_ptr[mb]; bn[h] = p.band_ptr[mb + 1] - bw0[h]; }
        }
    }
    // per-lane constants for all frames: twiddles W512^{l k1} (l = lane), W64^{b2 k2} (b2 = lane & 7), the split twiddles
    // W1024^{k} of this lane's bins k = lane + 64 r, and the window taps of its 8 complex input points
    // The three twiddle tables depend on (lane, k) only: the workgroup evaluates each entry ONCE into LDS (1 089 sincospif over 256
    // threads instead of 25 per lane of every wave) and a lane then picks its 25 constants up; the table aliases the per-wave
    // transpose buffers, which are not in use yet.
    cpx w1[8], w2[8], w3[9];
    float2 win[8];
    {
        cpx* tw = &tr[0][0];                                              // [0, 512): W512^{l k}; [512, 576): W64^{b k}; [576, 1152): W1024^{k}
        for (int i = tid; i < 512 + 64 + 576; i += 256) {
            float s, c, a;
            if (i < 512) a = (float)((i >> 3) * (i & 7)) / 512.0f;
            else if (i < 576) a = (float)(((i - 512) >> 3) * ((i - 512) & 7)) / 64.0f;
            else a = (float)(i - 576) / 1024.0f;
            sincospif(-2.0f * a, &s, &c);
            tw[i] = (cpx){c, s};
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            w1[k] = tw[lane * 8 + k];
            w2[k] = tw[512 + (lane & 7) * 8 + k];
            win[k] = *reinterpret_cast<const float2*>(p.window + 2 * (lane + 64 * k));
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) w3[r] = tw[576 + lane + 64 * r];
    }
    __syncthreads();
    cpx* T = tr[wave];
    float* M = mg[wave];
    for (int fi = 0; fi < FPW; ++fi) {
        const int t = f0 + wave * FPW + fi;
        if (t >= nfb) break;                                              // wave-uniform
        const float* xf = xs + (wave * FPW + fi) * p.hop;
        // ---- pass 1: lane l holds z[l + 64 j]; DFT over j; twiddle W512^{l k1}
        cpx v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = lane + 64 * j;
            const float2 x2 = *reinterpret_cast<const float2*>(xf + 2 * m);
            v[j] = (cpx){x2.x * win[j].x, x2.y * win[j].y};
        }
        dft8(v);
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], w1[k]);
        // transpose 1: element (l = 8 a + b2, k1) -> lane (k1, b2), register a
        {
            const int a = lane >> 3, b2 = lane & 7;
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) T[(k1 * 8 + b2) * 8 + a] = v[k1];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int a = 0; a < 8; ++a) v[a] = T[lane * 8 + a];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- pass 2: DFT over a; twiddle W64^{b2 k2}
        dft8(v);
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], w2[k]);
        // transpose 2: element (k1, b2, k2) -> lane (k1, k2), register b2
        {
            const int k1 = lane >> 3, b2 = lane & 7;
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) T[(k1 * 8 + k2) * 8 + b2] = v[k2];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int b2 = 0; b2 < 8; ++b2) v[b2] = T[lane * 8 + b2];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- pass 3: DFT over b2 -> Z[k], k = k1 + 8 k2 + 64 k3 with (k1, k2) = (lane >> 3, lane & 7)
        dft8(v);
        {
            const int q = (lane >> 3) + 8 * (lane & 7);
#pragma unroll
            for (int k3 = 0; k3 < 8; ++k3) T[q + 64 * k3] = v[k3];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- split: X[k], k = lane + 64 r (r = 0 .. 7), and k = 512 on lane 0
#pragma unroll
        for (int r = 0; r <= 8; ++r) {
            const int k = lane + 64 * r;
            if (k <= NH) {
                const cpx zk = T[k & (NH - 1)], zc = T[(NH - k) & (NH - 1)];
                const cpx e = {0.5f * (zk.re + zc.re), 0.5f * (zk.im - zc.im)};    // (Z[k] + conj Z[512-k]) / 2
                const cpx o = {0.5f * (zk.re - zc.re), 0.5f * (zk.im + zc.im)};    // (Z[k] - conj Z[512-k]) / 2
                const cpx tw = cmul(w3[r], mul_mi(o));                              // -i e^{-2 pi i k / 1024} o
                const float re = e.re + tw.re, im = e.im + tw.im;
                const float m = sqrtf(re * re + im * im);
                M[k] = m;
                if (p.mag) {
                    p.mag[((size_t)b * NB + k) * p.ldt + t] = m;
                    p.phase[((size_t)b * NB + k) * p.ldt + t] = atan2f(im, re);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- sparse triangular filterbank + log compression (audio_processing.py:132-133, :81-82)
        if (p.mel) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int mb = lane + 64 * h;
                if (mb >= p.n_mel) continue;
                const int k0 = bk0[h], w0 = bw0[h], n = bn[h];
                float s = 0.f;                                             // one accumulator, bins in ascending order (as before)
                if (w_lds) {
                    for (int i = 0; i < n; i += 8) {
                        float wv[8], mv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int j = i + u < n ? i + u : n - 1;
                            wv[u] = bw[w0 + j];
                            mv[u] = M[k0 + j];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) if (i + u < n) s += wv[u] * mv[u];
                    }
                } else {
                    for (int i = 0; i < n; ++i) s += p.band_w[w0 + i] * M[k0 + i];
                }
                mo[mb][wave * FPW + fi] = logf(fmaxf(s, 1e-5f));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (p.mel) {                                                          // [band][16 consecutive frames]: 64-byte row pieces
        __syncthreads();
        const int nf = min(FPG, p.n_frames - f0);
        for (int idx = tid; idx < p.n_mel * FPG; idx += 256) {
            const int mb = idx / FPG, f = idx - mb * FPG;
            if (f < nf) p.mel[((size_t)b * p.n_mel + mb) * p.ldt + f0 + f] = (f0 + f < nfb) ? mo[mb][f] : 0.f;   // zero padding of DataCollate
        }
    }
}

}  // namespace

// y [B,N] -> any of mel [B,n_mel,T] (needs the CSR filterbank), mag [B,513,T], phase [B,513,T] (both or neither); T = N / hop + 1.
// n_fft = 1024 (hann window [1024] passed in, any win_length zero-padded by the caller), hop <= 256.
extern "C" int ft_stft_r8(const float* y, const float* window, const int32_t* band_bin0, const int32_t* band_ptr,
                          const float* band_w, float* mel, float* mag, float* phase, int B, int N, int hop, int n_mel,
                          void* stream) {
    FT_CHECK_ARG(y && window && (mel || mag));
    FT_CHECK_ARG((mag == nullptr) == (phase == nullptr));
    FT_CHECK_ARG(!mel || (band_bin0 && band_ptr && band_w && n_mel >= 1 && n_mel <= 128));
    FT_CHECK_ARG(B >= 1 && B <= 65535 && hop >= 1 && hop <= 256 && N > NH);
    const int n_frames = N / hop + 1;
    StftP p{y, window, band_bin0, band_ptr, band_w, mel, mag, phase, N, hop, n_mel, n_frames, nullptr, n_frames};
    hipLaunchKernelGGL(stft_r8_k, dim3(cdiv(n_frames, FPG), B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    FT_CHECK_LAUNCH();
    return FT_OK;
}

// The collated batch of the data path (data.py:207-229 pads every mel with zeros to the longest one): y [B,N] zero-padded audio,
// utterance b holds n_samples[b] samples (device int32) -> mel [B,n_mel,T_out]: frames < n_samples[b] / hop + 1 as ft_stft_r8
// computes them for that utterance alone (reflection about ITS last sample), zeros beyond.  ONE launch for the batch instead of
// one per utterance (each ~35 us of latency for <= 862 frames).  T_out >= max_b (n_samples[b] / hop + 1).
extern "C" int ft_stft_r8_ragged(const float* y, const int32_t* n_samples, const float* window, const int32_t* band_bin0,
                                 const int32_t* band_ptr, const float* band_w, float* mel, int B, int N, int hop, int n_mel,
                                 int T_out, void* stream) {
    FT_CHECK_ARG(y && n_samples && window && mel && band_bin0 && band_ptr && band_w && n_mel >= 1 && n_mel <= 128);
    FT_CHECK_ARG(B >= 1 && B <= 65535 && hop >= 1 && hop <= 256 && N > NH && T_out >= 1);
    StftP p{y, window, band_bin0, band_ptr, band_w, mel, nullptr, nullptr, N, hop, n_mel, T_out, n_samples, T_out};
    hipLaunchKernelGGL(stft_r8_k, dim3(cdiv(T_out, FPG), B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
