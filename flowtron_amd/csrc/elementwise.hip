// HBM-bound elementwise / gather / reduction kernels of the Flowtron hot path (gfx950).
// Every kernel is a grid-stride or one-row-per-wave streaming pass: coalesced along the
// channel dimension, fp32, no LDS beyond block reductions.
#include "common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float r = 0.f;
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}

inline int grid_for(int64_t n, int per_block = NT, int cap = 256 * 16) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ---------------- embedding ----------------
__global__ void embedding_fwd_k(const int64_t* __restrict__ ids, const float* __restrict__ W, float* __restrict__ out,
                                int n, int dim, long ld) {
    const long total = (long)n * dim;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / dim;
        const int c = (int)(i - r * dim);
        out[r * ld + c] = W[ids[r] * (long)dim + c];
    }
}
__global__ void embedding_bwd_k(const int64_t* __restrict__ ids, const float* __restrict__ dout, float* __restrict__ dW,
                                int n, int dim, long ld) {
    const long total = (long)n * dim;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / dim;
        const int c = (int)(i - r * dim);
        atomicAdd(dW + ids[r] * (long)dim + c, dout[r * ld + c]);
    }
}

// Rows walked with a stride (row r = i * stride + j, j < stride): one thread per (j, row chunk, column) sums its rows in a register
// while the id stays the same and sends ONE atomic per run.  The speaker embedding is gathered for every text position ([L,B] rows,
// id = speaker of utterance j = r % B: flowtron.py:886-887 expand + cat), so a chunk of RB rows of one utterance is one run -- the
// plain kernel above sent L * B * dim atomics to a handful of rows (all to ONE row for a single-speaker corpus: 122 us for 2.6 M
// same-address atomics at B 32, L 157).
__global__ void embedding_bwd_runs_k(const int64_t* __restrict__ ids, const float* __restrict__ dout, float* __restrict__ dW,
                                     int n, int stride, int dim, long ld, int RB) {
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= dim) return;
    const int j = blockIdx.x % stride, chunk = blockIdx.x / stride;
    const long i0 = (long)chunk * RB;
    long cur = -1;
    float acc = 0.f;
    for (int u = 0; u < RB; ++u) {
        const long r = (i0 + u) * stride + j;
        if (r >= n) break;
        const long id = ids[r];
        const float v = dout[r * ld + c];
        if (id != cur) {
            if (cur >= 0) atomicAdd(dW + cur * (long)dim + c, acc);
            cur = id;
            acc = v;
        } else {
            acc += v;
        }
    }
    if (cur >= 0) atomicAdd(dW + cur * (long)dim + c, acc);
}

// ---------------- im2col / col2im (encoder conv, time-major) ----------------
// col[(l*B+b)][c*KW+k] = x[l+k-KW/2][b][c] if 0 <= l+k-KW/2 < lens[b] else 0
__global__ void im2col_k(const float* __restrict__ x, float* __restrict__ col, const int* __restrict__ lens,
                         int L, int B, int C, int KW) {
    const long CK = (long)C * KW;
    const long total = (long)L * B * CK;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / CK;
        const int j = (int)(i - row * CK);
        const int c = j / KW, k = j - c * KW;
        const int l = (int)(row / B), b = (int)(row - (long)l * B);
        const int ls = l + k - KW / 2;
        float v = 0.f;
        if (ls >= 0 && ls < lens[b]) v = x[((long)ls * B + b) * C + c];
        col[i] = v;
    }
}
// dx[l][b][c] = sum_k dcol[(l-k+KW/2)*B+b][c*KW+k]  for rows inside [0,L); 0 at l >= lens[b]
__global__ void col2im_k(const float* __restrict__ dcol, float* __restrict__ dx, const int* __restrict__ lens,
                         int L, int B, int C, int KW) {
    const long total = (long)L * B * C;
    const long CK = (long)C * KW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / C;
        const int c = (int)(i - row * C);
        const int l = (int)(row / B), b = (int)(row - (long)l * B);
        float v = 0.f;
        if (l < lens[b]) {
            for (int k = 0; k < KW; ++k) {
                const int lo = l - k + KW / 2;
                if (lo >= 0 && lo < L) v += dcol[((long)lo * B + b) * CK + (long)c * KW + k];
            }
        }
        dx[i] = v;
    }
}

// ---------------- reverse by length ----------------
__global__ void reverse_k(const float* __restrict__ x, float* __restrict__ y, const int* __restrict__ lens,
                          int T, int B, int C, int time_major) {
    const long total = (long)T * B * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / C;
        const int c = (int)(i - row * C);
        int t, b;
        if (time_major) { t = (int)(row / B); b = (int)(row - (long)t * B); }
        else { b = (int)(row / T); t = (int)(row - (long)b * T); }
        const int len = lens[b];
        const int src = (t < len) ? (len - 1 - t) : (T - 1 + len - t);
        const long srow = time_major ? ((long)src * B + b) : ((long)b * T + src);
        y[i] = x[srow * C + c];
    }
}

// ---------------- affine coupling ----------------
__global__ void affine_fwd_k(const float* __restrict__ out, const float* __restrict__ x, float* __restrict__ z,
                             long n_rows, int M) {
    const long total = n_rows * M;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / M;
        const int c = (int)(i - r * M);
        const float ls = out[r * 2 * M + c], bb = out[r * 2 * M + M + c];
        z[i] = expf(ls) * x[i] + bb;
    }
}
__global__ void affine_bwd_k(const float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ dz,
                             const float* __restrict__ dls_ext, float* __restrict__ dout, float* __restrict__ dx,
                             long n_rows, int M) {
    const long total = n_rows * M;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / M;
        const int c = (int)(i - r * M);
        const float s = expf(out[r * 2 * M + c]);
        const float g = dz[i];
        float dls = g * x[i] * s;
        if (dls_ext) dls += dls_ext[i];
        dout[r * 2 * M + c] = dls;
        dout[r * 2 * M + M + c] = g;
        if (dx) dx[i] = g * s;
    }
}
__global__ void affine_inv_k(const float* __restrict__ out, const float* __restrict__ z, float* __restrict__ x,
                             long n_rows, int M) {
    const long total = n_rows * M;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / M;
        const int c = (int)(i - r * M);
        x[i] = (z[i] - out[r * 2 * M + M + c]) / expf(out[r * 2 * M + c]);
    }
}

// ---------------- masked sums (NLL) ----------------
__global__ void masked_sum_k(const float* __restrict__ x, long ld, const int* __restrict__ lens, float* __restrict__ acc,
                             int square, int T, int B, int M) {
    __shared__ float red[NT / 64];
    const long total = (long)T * B * M;
    float s = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / M;
        const int c = (int)(i - row * M);
        const int t = (int)(row / B), b = (int)(row - (long)t * B);
        if (t < lens[b]) {
            const float v = x[row * ld + c];
            s += square ? v * v : v;
        }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(acc, s);
}
__global__ void masked_sum_bwd_k(const float* __restrict__ x, long ld, const int* __restrict__ lens,
                                 const float* __restrict__ scale_dev, float coef, int square,
                                 float* __restrict__ dx, long ld_dx, int T, int B, int M) {
    const long total = (long)T * B * M;
    const float sc = scale_dev[0] * coef;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / M;
        const int c = (int)(i - row * M);
        const int t = (int)(row / B), b = (int)(row - (long)t * B);
        float v = 0.f;
        if (t < lens[b]) v = square ? sc * x[row * ld + c] : sc;
        dx[row * ld_dx + c] = v;
    }
}

// ---------------- gate BCE with logits ----------------
// loss(g,y) = max(g,0) - g*y + log1p(exp(-|g|))
__global__ void gate_bce_fwd_k(const float* __restrict__ gate, const float* __restrict__ target, const int* __restrict__ lens,
                               float* __restrict__ acc, int T, int B) {
    __shared__ float red[NT / 64];
    const long total = (long)T * B;
    float s = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i / B), b = (int)(i - (long)t * B);
        if (t < lens[b]) {
            const float g = gate[i], y = target[(long)b * T + t];
            s += fmaxf(g, 0.f) - g * y + log1pf(expf(-fabsf(g)));
        }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(acc, s);
}
__global__ void gate_bce_bwd_k(const float* __restrict__ gate, const float* __restrict__ target, const int* __restrict__ lens,
                               const float* __restrict__ scale_dev, float coef, float* __restrict__ dgate, int T, int B) {
    const long total = (long)T * B;
    const float sc = scale_dev[0] * coef;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i / B), b = (int)(i - (long)t * B);
        float v = 0.f;
        if (t < lens[b]) {
            const float g = gate[i], y = target[(long)b * T + t];
            v = sc * (1.f / (1.f + expf(-g)) - y);
        }
        dgate[i] = v;
    }
}

// ---------------- FlowtronLoss in four launches (flowtron.py:200-243) ----------------
// The per-term kernels above leave the scalar arithmetic (the frame count, the normalisers, g / n in backward) to the caller: ~40
// four-byte torch kernels per training step, right where the host waits for the losses (train.py:300-303 .item()).  Here the sums of
// z and of every flow's log_s come from ONE pass, a one-workgroup kernel forms n = sum(lens) and the two losses on the device and
// leaves 1 / (n M) and 1 / n for the backward kernels, which read the incoming gradients through pointers.
// acc: [0] sum z^2, [1] sum log_s, [2] sum BCE, [4] 1 / (n M), [5] 1 / n, [6] n
struct LsPtrs { const float* p[8]; };

__global__ void nll_sums_k(const float* __restrict__ z, LsPtrs ls, int n_ls, long ld_ls, const int* __restrict__ lens,
                           float* __restrict__ acc, int T, int B, int M) {
    __shared__ float red[NT / 64];
    const long total = (long)T * B * M;
    float s2 = 0.f, sl = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / M;
        const int c = (int)(i - row * M);
        const int t = (int)(row / B), b = (int)(row - (long)t * B);
        if (t < lens[b]) {
            const float v = z[i];
            s2 += v * v;
            for (int f = 0; f < n_ls; ++f) sl += ls.p[f][row * ld_ls + c];
        }
    }
    s2 = block_sum(s2, red);
    sl = block_sum(sl, red);
    if (threadIdx.x == 0) { atomicAdd(acc, s2); atomicAdd(acc + 1, sl); }
}

__global__ void loss_finalize_k(float* __restrict__ acc, const int* __restrict__ lens, int B, int M, float two_sigma2,
                                float* __restrict__ nll_out, float* __restrict__ gate_out) {
    __shared__ float red[NT / 64];
    float n = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) n += (float)lens[b];
    n = block_sum(n, red);
    if (threadIdx.x == 0) {
        const float nm = n * (float)M;
        nll_out[0] = (acc[0] / two_sigma2 - acc[1]) / nm;
        if (gate_out) gate_out[0] = acc[2] / n;
        acc[4] = 1.f / nm;
        acc[5] = 1.f / n;
        acc[6] = n;
    }
}

// dz = g inv_nm z / sigma^2, dls = -g inv_nm at valid frames, 0 at padded ones (dls: ONE tensor, the gradient of every flow's log_s)
__global__ void nll_bwd_k(const float* __restrict__ z, const int* __restrict__ lens, const float* __restrict__ g,
                          const float* __restrict__ inv_nm, float inv_sigma2, float* __restrict__ dz, float* __restrict__ dls,
                          int T, int B, int M) {
    const long total = (long)T * B * M;
    const float sc = g[0] * inv_nm[0];
    const float scz = sc * inv_sigma2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / M;
        const int t = (int)(row / B), b = (int)(row - (long)t * B);
        const bool valid = t < lens[b];
        dz[i] = valid ? scz * z[i] : 0.f;
        if (dls) dls[i] = valid ? -sc : 0.f;
    }
}

__global__ void gate_bce_bwd2_k(const float* __restrict__ gate, const float* __restrict__ target, const int* __restrict__ lens,
                                const float* __restrict__ g, const float* __restrict__ inv_n, float* __restrict__ dgate, int T, int B) {
    const long total = (long)T * B;
    const float sc = g[0] * inv_n[0];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i / B), b = (int)(i - (long)t * B);
        float v = 0.f;
        if (t < lens[b]) {
            const float gl = gate[i], y = target[(long)b * T + t];
            v = sc * (1.f / (1.f + expf(-gl)) - y);
        }
        dgate[i] = v;
    }
}

// ---------------- column sums ----------------
// grid.x = column blocks of 64, grid.y = row slabs; 256 threads = 4 row-lanes x 64 columns.
__global__ void colsum_k(const float* __restrict__ x, float* __restrict__ out, long rows, int N, long ld, long rows_per_slab) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const long r0 = (long)blockIdx.y * rows_per_slab;
    long r1 = r0 + rows_per_slab;
    if (r1 > rows) r1 = rows;
    float s = 0.f;
    if (col < N)
        for (long r = r0 + rl; r < r1; r += 4) s += x[r * ld + col];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && col < N) atomicAdd(out + col, red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

// dpre = dy * act'(pre) expressed through the saved OUTPUT y = act(pre)
__global__ void act_bwd_k(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dpre, long n, int act) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float yy = y[i], g = dy[i];
        float d;
        if (act == FT_ACT_TANH) d = g * (1.f - yy * yy);
        else if (act == FT_ACT_RELU) d = yy > 0.f ? g : 0.f;
        else if (act == FT_ACT_SIGMOID) d = g * yy * (1.f - yy);
        else d = g;
        dpre[i] = d;
    }
}

// ---------------- beta-binomial attention prior (data.py:31-41) ----------------
// prior[b][t][k] = BetaBinom(k; n = P_b-1, alpha = s*(t+1), beta = s*(M_b-t))   t < M_b = out_lens[b], k < P_b = in_lens[b]
// evaluated in float64 through lgamma like scipy.stats.betabinom.pmf, rounded to fp32; 0 in the padding.
__global__ void beta_binomial_prior_k(const int* __restrict__ in_lens, const int* __restrict__ out_lens,
                                      float* __restrict__ prior, int B, int T, int L, double scaling) {
    const long total = (long)B * T * L;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % L);
        const long r = i / L;
        const int t = (int)(r % T), b = (int)(r / T);
        const int P = in_lens[b], M = out_lens[b];
        float v = 0.f;
        if (t < M && k < P) {
            const double n = (double)(P - 1), kk = (double)k;
            const double a = scaling * (double)(t + 1), bb = scaling * (double)(M - t);
            const double lp = lgamma(n + 1.0) - lgamma(kk + 1.0) - lgamma(n - kk + 1.0) + lgamma(kk + a) + lgamma(n - kk + bb) -
                              lgamma(n + a + bb) - (lgamma(a) + lgamma(bb) - lgamma(a + bb));
            v = (float)exp(lp);
        }
        prior[i] = v;
    }
}

// out = a (+|*) b  -- the key modulation text*cond and the running attention sum of the cumulative-attention branch
__global__ void eltwise_k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n, int op) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = op ? a[i] * b[i] : a[i] + b[i];
}

__global__ void zero_k(float* __restrict__ p, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0.f;
}

// ---------------- optimizer (flat arena) ----------------
// Two launches and NO float atomics: the square norm feeds the clip factor of every rank's optimizer step, and data-parallel
// replicas only stay bit-identical if every rank computes bit-identical norms from its bit-identical reduced gradients (an
// atomicAdd per block sums in arrival order: two ranks on one MI355X ended a step with weights one ulp apart).  Block b writes
// its partial sum to part[b]; one workgroup then adds the partials in index order.
__global__ void sumsq_part_k(const float* __restrict__ x, float* __restrict__ part, long n) {
    __shared__ float red[NT / 64];
    float s = 0.f;
    const long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = x4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        s += x[i] * x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void sumsq_fin_k(const float* __restrict__ part, int nb, float* __restrict__ acc) {
    __shared__ float red[NT / 64];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) acc[0] += s;
}

// radam.py:44-122 restated per element (fp32 state):
//   g' = g * min(1, clip / (||g|| + 1e-6))            (torch.nn.utils.clip_grad_norm_, train.py:328)
//   v = b2 v + (1-b2) g'^2 ; m = b1 m + (1-b1) g'
//   p -= wd*lr*p (if wd != 0) ; N_sma >= 5: p -= step_size * m/(sqrt(v)+eps) ; else p -= step_size*m
//   (step_size is the host-computed radam.py:95-105 value and already contains lr)
// one block per utterance: offset = sum of (len + 1) of the utterances before it (B is small), then the block writes its span
__global__ void rowmap_k(const int* __restrict__ lens, int* __restrict__ rowmap, int* __restrict__ rows_dev, int T, int B) {
    const int b = blockIdx.x;
    int off = 0;
    for (int i = 0; i < b; ++i) { int l = lens[i]; l = l < 0 ? 0 : (l > T ? T : l); off += l + 1; }
    int len = lens[b];
    len = len < 0 ? 0 : (len > T ? T : len);
    for (int t = threadIdx.x; t < len; t += blockDim.x) rowmap[off + t] = t * B + b;
    if (threadIdx.x == 0) {
        rowmap[off + len] = len < T ? len * B + b : -1;
        if (b == B - 1) rows_dev[0] = off + len + 1;
    }
}

// grid (T*B rows, cols/1024 chunks): padded rows t > lens[b] only
__global__ void pad_fill_k(float* __restrict__ y, long ld, int cols, const int* __restrict__ lens, int T, int B, int mode) {
    const int row = blockIdx.x, t = row / B, b = row % B;
    int len = lens[b];
    len = len < 0 ? 0 : len;
    if (t <= len) return;
    const float* src = y + (size_t)(len * B + b) * ld;
    float* dst = y + (size_t)row * ld;
    for (int c = blockIdx.y * blockDim.x + threadIdx.x; c < cols; c += gridDim.y * blockDim.x) dst[c] = mode ? src[c] : 0.f;
}

__global__ void poison_k(const int* __restrict__ status, float* __restrict__ dst) {
    if (status[0] != 0) dst[0] = __builtin_nanf("");
}

__global__ void radam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                        long n, const float* __restrict__ gnorm_sq, float clip, float wd_lr, float b1, float b2, float omb1,
                        float omb2, float eps, float step_size, int rectified, int* __restrict__ skipped) {
    float cs = 1.f;
    if (gnorm_sq) {
        const float sq = gnorm_sq[0];
        if (!(sq <= 3.0e38f)) {                     // NaN or Inf: drop the step (include/flowtron_hip.h, guard)
            if (skipped && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skipped, 1);
            return;
        }
        if (clip > 0.f) cs = fminf(1.f, clip / (sqrtf(sq) + 1e-6f));
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * cs;
        // radam.py:76-77: exp_avg_sq.mul_(beta2).addcmul_(1 - beta2, grad, grad); exp_avg.mul_(beta1).add_(1 - beta1, grad)
        // with (1 - beta) evaluated in DOUBLE by the caller (1 - 0.999f differs from float(1 - 0.999) by 4.7e-5 relative)
        const float vi = __fmaf_rn(omb2, __fmul_rn(gi, gi), __fmul_rn(b2, v[i]));
        const float mi = __fmaf_rn(omb1, gi, __fmul_rn(b1, m[i]));
        float pi = p[i];
        if (wd_lr != 0.f) pi = __fmaf_rn(-wd_lr, pi, pi);
        if (rectified) pi += -step_size * (mi / (sqrtf(vi) + eps));
        else pi += -step_size * mi;
        v[i] = vi; m[i] = mi; p[i] = pi;
    }
}

// The same update with the step count formed ON THE DEVICE: step = calls - *skipped (calls = optimizer.step() invocations so far,
// this one included; *skipped = updates the guard has dropped so far), so bias correction and N_sma (radam.py:82-106) follow the
// updates that were APPLIED -- what radam.py's per-parameter state['step'] counts under GradScaler, which simply does not call
// step() after an overflow (train.py:330) -- without the host ever reading the drop decision.  One thread per workgroup evaluates
// the closed form in double (the arithmetic of ops/optim.py: RAdam.step_size_for) and broadcasts it through LDS.  No race on
// *skipped: it is written only when this launch is dropped, and then nobody uses the coefficients.
__global__ void radam_dev_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, const float* __restrict__ gnorm_sq, float clip, float wd_lr, float b1, float b2, float omb1,
                            float omb2, float eps, double lr, double beta1, double beta2, int calls, int* __restrict__ skipped) {
    __shared__ float coef[2];
    float cs = 1.f;
    {
        const float sq = gnorm_sq[0];
        if (!(sq <= 3.0e38f)) {                     // NaN or Inf: drop the step
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skipped, 1);
            return;
        }
        if (clip > 0.f) cs = fminf(1.f, clip / (sqrtf(sq) + 1e-6f));
    }
    if (threadIdx.x == 0) {
        int step = calls - skipped[0];
        if (step < 1) step = 1;
        const double beta2_t = pow(beta2, (double)step);
        const double n_sma_max = 2.0 / (1.0 - beta2) - 1.0;
        const double n_sma = n_sma_max - 2.0 * step * beta2_t / (1.0 - beta2_t);
        const double bc1 = 1.0 - pow(beta1, (double)step);
        double ss;
        float rect;
        if (n_sma >= 5.0) {
            ss = lr * sqrt((1.0 - beta2_t) * (n_sma - 4.0) / (n_sma_max - 4.0) * (n_sma - 2.0) / n_sma * n_sma_max / (n_sma_max - 2.0)) / bc1;
            rect = 1.f;
        } else {
            ss = lr / bc1;
            rect = 0.f;
        }
        coef[0] = (float)ss;
        coef[1] = rect;
    }
    __syncthreads();
    const float step_size = coef[0];
    const bool rectified = coef[1] != 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * cs;
        const float vi = __fmaf_rn(omb2, __fmul_rn(gi, gi), __fmul_rn(b2, v[i]));
        const float mi = __fmaf_rn(omb1, gi, __fmul_rn(b1, m[i]));
        float pi = p[i];
        if (wd_lr != 0.f) pi = __fmaf_rn(-wd_lr, pi, pi);
        if (rectified) pi += -step_size * (mi / (sqrtf(vi) + eps));
        else pi += -step_size * mi;
        v[i] = vi; m[i] = mi; p[i] = pi;
    }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int ft_embedding_fwd(const int64_t* ids, const float* W, float* out, int n, int dim, int64_t ld_out, void* stream) {
    FT_CHECK_ARG(ids && W && out && n >= 0 && dim > 0 && ld_out >= dim);
    if (n == 0) return FT_OK;
    hipLaunchKernelGGL(embedding_fwd_k, dim3(grid_for((int64_t)n * dim)), dim3(NT), 0, ST(stream), ids, W, out, n, dim, (long)ld_out);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_embedding_bwd(const int64_t* ids, const float* dout, float* dW, int n, int dim, int64_t ld_dout, void* stream) {
    FT_CHECK_ARG(ids && dout && dW && n >= 0 && dim > 0 && ld_dout >= dim);
    if (n == 0) return FT_OK;
    hipLaunchKernelGGL(embedding_bwd_k, dim3(grid_for((int64_t)n * dim)), dim3(NT), 0, ST(stream), ids, dout, dW, n, dim, (long)ld_dout);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_embedding_bwd_runs(const int64_t* ids, const float* dout, float* dW, int n, int dim, int64_t ld_dout, int stride,
                                     void* stream) {
    FT_CHECK_ARG(ids && dout && dW && n >= 0 && dim > 0 && ld_dout >= dim && stride >= 1);
    if (n == 0) return FT_OK;
    const int RB = 32, per = cdiv(n, stride);                      // rows per (j) sequence, chunks of RB of them
    const int tx = dim >= 128 ? 128 : 64;
    const int64_t gx = (int64_t)cdiv(per, RB) * stride;
    FT_CHECK_ARG(gx <= 0x7fffffffLL);
    hipLaunchKernelGGL(embedding_bwd_runs_k, dim3((unsigned)gx, cdiv(dim, tx)), dim3(tx), 0, ST(stream), ids, dout, dW, n, stride, dim,
                       (long)ld_dout, RB);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_im2col(const float* x, float* col, const int32_t* lens, int L, int B, int C, int KW, void* stream) {
    FT_CHECK_ARG(x && col && lens && L >= 0 && B >= 1 && C >= 1 && KW >= 1 && (KW & 1));
    if (L == 0) return FT_OK;
    hipLaunchKernelGGL(im2col_k, dim3(grid_for((int64_t)L * B * C * KW)), dim3(NT), 0, ST(stream), x, col, lens, L, B, C, KW);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_col2im(const float* dcol, float* dx, const int32_t* lens, int L, int B, int C, int KW, void* stream) {
    FT_CHECK_ARG(dcol && dx && lens && L >= 0 && B >= 1 && C >= 1 && KW >= 1 && (KW & 1));
    if (L == 0) return FT_OK;
    hipLaunchKernelGGL(col2im_k, dim3(grid_for((int64_t)L * B * C)), dim3(NT), 0, ST(stream), dcol, dx, lens, L, B, C, KW);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_reverse_by_length(const float* x, float* y, const int32_t* lens, int T, int B, int C, int time_major, void* stream) {
    FT_CHECK_ARG(x && y && lens && x != y && T >= 0 && B >= 1 && C >= 1);
    if (T == 0) return FT_OK;
    hipLaunchKernelGGL(reverse_k, dim3(grid_for((int64_t)T * B * C)), dim3(NT), 0, ST(stream), x, y, lens, T, B, C, time_major);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_affine_fwd(const float* out, const float* x, float* z, int64_t n_rows, int M, void* stream) {
    FT_CHECK_ARG(out && x && z && n_rows >= 0 && M >= 1);
    if (n_rows == 0) return FT_OK;
    hipLaunchKernelGGL(affine_fwd_k, dim3(grid_for(n_rows * M)), dim3(NT), 0, ST(stream), out, x, z, (long)n_rows, M);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_affine_bwd(const float* out, const float* x, const float* dz, const float* dlog_s_ext,
                             float* dout, float* dx, int64_t n_rows, int M, void* stream) {
    FT_CHECK_ARG(out && x && dz && dout && n_rows >= 0 && M >= 1);
    if (n_rows == 0) return FT_OK;
    hipLaunchKernelGGL(affine_bwd_k, dim3(grid_for(n_rows * M)), dim3(NT), 0, ST(stream), out, x, dz, dlog_s_ext, dout, dx, (long)n_rows, M);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_affine_inv(const float* out, const float* z, float* x, int64_t n_rows, int M, void* stream) {
    FT_CHECK_ARG(out && z && x && n_rows >= 0 && M >= 1);
    if (n_rows == 0) return FT_OK;
    hipLaunchKernelGGL(affine_inv_k, dim3(grid_for(n_rows * M)), dim3(NT), 0, ST(stream), out, z, x, (long)n_rows, M);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_masked_sum(const float* x, int64_t ld, const int32_t* lens, float* acc, int square,
                             int T, int B, int M, void* stream) {
    FT_CHECK_ARG(x && lens && acc && ld >= M && T >= 0 && B >= 1 && M >= 1);
    if (T == 0) return FT_OK;
    hipLaunchKernelGGL(masked_sum_k, dim3(grid_for((int64_t)T * B * M, NT, 1024)), dim3(NT), 0, ST(stream), x, (long)ld, lens, acc, square, T, B, M);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_masked_sum_bwd(const float* x, int64_t ld, const int32_t* lens, const float* scale_dev, float coef, int square,
                                 float* dx, int64_t ld_dx, int T, int B, int M, void* stream) {
    FT_CHECK_ARG(lens && scale_dev && dx && (x || !square) && ld_dx >= M && T >= 0 && B >= 1 && M >= 1);
    if (T == 0) return FT_OK;
    hipLaunchKernelGGL(masked_sum_bwd_k, dim3(grid_for((int64_t)T * B * M)), dim3(NT), 0, ST(stream), x, (long)ld, lens, scale_dev, coef, square, dx, (long)ld_dx, T, B, M);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_gate_bce_fwd(const float* gate, const float* target, const int32_t* lens, float* acc, int T, int B, void* stream) {
    FT_CHECK_ARG(gate && target && lens && acc && T >= 0 && B >= 1);
    if (T == 0) return FT_OK;
    hipLaunchKernelGGL(gate_bce_fwd_k, dim3(grid_for((int64_t)T * B, NT, 256)), dim3(NT), 0, ST(stream), gate, target, lens, acc, T, B);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_gate_bce_bwd(const float* gate, const float* target, const int32_t* lens, const float* scale_dev, float coef,
                               float* dgate, int T, int B, void* stream) {
    FT_CHECK_ARG(gate && target && lens && scale_dev && dgate && T >= 0 && B >= 1);
    if (T == 0) return FT_OK;
    hipLaunchKernelGGL(gate_bce_bwd_k, dim3(grid_for((int64_t)T * B)), dim3(NT), 0, ST(stream), gate, target, lens, scale_dev, coef, dgate, T, B);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_flowtron_loss_fwd(const float* z, const float* const* log_s, int n_ls, int64_t ld_ls, const float* gate,
                                    const float* gate_target, const int32_t* out_lens, float sigma, float* acc, float* nll_out,
                                    float* gate_out, int T, int B, int M, void* stream) {
    FT_CHECK_ARG(z && out_lens && acc && nll_out && T >= 1 && B >= 1 && M >= 1 && sigma > 0.f);
    FT_CHECK_ARG(n_ls >= 0 && n_ls <= 8 && (n_ls == 0 || (log_s && ld_ls >= M)));
    FT_CHECK_ARG((gate == nullptr) == (gate_out == nullptr) && (gate == nullptr || gate_target != nullptr));
    LsPtrs ls{};
    for (int f = 0; f < n_ls; ++f) { FT_CHECK_ARG(log_s[f] != nullptr); ls.p[f] = log_s[f]; }
    FT_CHECK_HIP(hipMemsetAsync(acc, 0, 8 * sizeof(float), ST(stream)));
    hipLaunchKernelGGL(nll_sums_k, dim3(grid_for((int64_t)T * B * M, NT, 1024)), dim3(NT), 0, ST(stream), z, ls, n_ls, (long)ld_ls, out_lens,
                       acc, T, B, M);
    if (gate)
        hipLaunchKernelGGL(gate_bce_fwd_k, dim3(grid_for((int64_t)T * B, NT, 256)), dim3(NT), 0, ST(stream), gate, gate_target, out_lens, acc + 2,
                           T, B);
    hipLaunchKernelGGL(loss_finalize_k, dim3(1), dim3(NT), 0, ST(stream), acc, out_lens, B, M, 2.0f * sigma * sigma, nll_out, gate_out);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_flowtron_loss_bwd(const float* z, const float* gate, const float* gate_target, const int32_t* out_lens, float sigma,
                                    const float* acc, const float* g_nll, const float* g_gate, float* dz, float* dls, float* dgate,
                                    int T, int B, int M, void* stream) {
    FT_CHECK_ARG(out_lens && acc && T >= 1 && B >= 1 && M >= 1 && sigma > 0.f);
    FT_CHECK_ARG((g_nll == nullptr) == (dz == nullptr) && (dz == nullptr || z != nullptr) && (dls == nullptr || dz != nullptr));
    FT_CHECK_ARG((g_gate == nullptr) == (dgate == nullptr) && (dgate == nullptr || (gate && gate_target)));
    if (dz)
        hipLaunchKernelGGL(nll_bwd_k, dim3(grid_for((int64_t)T * B * M)), dim3(NT), 0, ST(stream), z, out_lens, g_nll, acc + 4,
                           1.0f / (sigma * sigma), dz, dls, T, B, M);
    if (dgate)
        hipLaunchKernelGGL(gate_bce_bwd2_k, dim3(grid_for((int64_t)T * B)), dim3(NT), 0, ST(stream), gate, gate_target, out_lens, g_gate,
                           acc + 5, dgate, T, B);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_colsum(const float* x, float* out, int64_t rows, int N, int64_t ld, void* stream) {
    FT_CHECK_ARG(x && out && rows >= 0 && N >= 1 && ld >= N);
    hipLaunchKernelGGL(zero_k, dim3(grid_for(N)), dim3(NT), 0, ST(stream), out, (long)N);
    if (rows > 0) {
        const int cb = cdiv(N, 64);
        int slabs = 2048 / cb;
        if (slabs < 1) slabs = 1;
        if (slabs > cdiv(rows, 32)) slabs = cdiv(rows, 32);
        const long rps = (rows + slabs - 1) / slabs;
        hipLaunchKernelGGL(colsum_k, dim3(cb, cdiv(rows, rps)), dim3(NT), 0, ST(stream), x, out, (long)rows, N, (long)ld, rps);
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_act_bwd(const float* y, const float* dy, float* dpre, int64_t n, int act, void* stream) {
    FT_CHECK_ARG(y && dy && dpre && n >= 0 && act >= FT_ACT_NONE && act <= FT_ACT_SIGMOID);
    if (n == 0) return FT_OK;
    hipLaunchKernelGGL(act_bwd_k, dim3(grid_for(n)), dim3(NT), 0, ST(stream), y, dy, dpre, (long)n, act);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_beta_binomial_prior(const int32_t* in_lens, const int32_t* out_lens, float* prior,
                                      int B, int T, int L, float scaling, void* stream) {
    FT_CHECK_ARG(in_lens && out_lens && prior && B >= 1 && T >= 1 && L >= 1 && scaling > 0.f);
    hipLaunchKernelGGL(beta_binomial_prior_k, dim3(grid_for((int64_t)B * T * L)), dim3(NT), 0, ST(stream), in_lens, out_lens, prior,
                       B, T, L, (double)scaling);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_eltwise(const float* a, const float* b, float* out, int64_t n, int op, void* stream) {
    FT_CHECK_ARG(a && b && out && n >= 0 && (op == 0 || op == 1));
    if (n == 0) return FT_OK;
    hipLaunchKernelGGL(eltwise_k, dim3(grid_for(n)), dim3(NT), 0, ST(stream), a, b, out, (long)n, op);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_sumsq(const float* x, float* acc, int64_t n, float* partials, void* stream) {
    FT_CHECK_ARG(x && acc && partials && n >= 0);
    FT_CHECK_ARG(reinterpret_cast<uintptr_t>(x) % 16 == 0);
    if (n == 0) return FT_OK;
    const int nb = grid_for(n / 4 + 1, NT, FT_SUMSQ_PARTIALS);       // a function of n alone: the same order on every rank
    hipLaunchKernelGGL(sumsq_part_k, dim3(nb), dim3(NT), 0, ST(stream), x, partials, (long)n);
    hipLaunchKernelGGL(sumsq_fin_k, dim3(1), dim3(NT), 0, ST(stream), partials, nb, acc);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_radam_step(float* p, const float* g, float* m, float* v, int64_t n,
                             const float* gnorm_sq_dev, double clip, double lr, double beta1, double beta2, double eps,
                             double weight_decay, double step_size, int rectified, int32_t* skipped_dev, void* stream) {
    FT_CHECK_ARG(p && g && m && v && n >= 0);
    if (n == 0) return FT_OK;
    // hyper-parameters arrive in double, as the python optimizer holds them: the derived coefficients are formed in double
    // and rounded once, like the scalars radam.py hands to mul_/add_/addcmul_
    hipLaunchKernelGGL(radam_k, dim3(grid_for(n, NT, 4096)), dim3(NT), 0, ST(stream), p, g, m, v, (long)n, gnorm_sq_dev, (float)clip,
                       (float)(weight_decay * lr), (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
                       (float)step_size, rectified, skipped_dev);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_radam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* gnorm_sq_dev, double clip,
                                 double lr, double beta1, double beta2, double eps, double weight_decay, int calls,
                                 int32_t* skipped_dev, void* stream) {
    FT_CHECK_ARG(p && g && m && v && gnorm_sq_dev && skipped_dev && n >= 0 && calls >= 1);
    if (n == 0) return FT_OK;
    hipLaunchKernelGGL(radam_dev_k, dim3(grid_for(n, NT, 4096)), dim3(NT), 0, ST(stream), p, g, m, v, (long)n, gnorm_sq_dev, (float)clip,
                       (float)(weight_decay * lr), (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
                       lr, beta1, beta2, calls, skipped_dev);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_rowmap_build(const int32_t* lens, int32_t* rowmap, int32_t* rows_dev, int T, int B, void* stream) {
    FT_CHECK_ARG(lens && rowmap && rows_dev && T >= 1 && B >= 1 && (int64_t)T * B + B < (1ll << 31));
    hipLaunchKernelGGL(rowmap_k, dim3(B), dim3(256), 0, ST(stream), lens, rowmap, rows_dev, T, B);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_pad_rows_fill(float* y, int64_t ld, int cols, const int32_t* lens, int T, int B, int mode, void* stream) {
    FT_CHECK_ARG(y && lens && T >= 1 && B >= 1 && cols >= 1 && ld >= cols && (mode == 0 || mode == 1));
    hipLaunchKernelGGL(pad_fill_k, dim3(T * B, cols > 1024 ? 4 : 1), dim3(256), 0, ST(stream), y, (long)ld, cols, lens, T, B, mode);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
extern "C" int ft_poison_if_nonzero(const int32_t* status_dev, float* dst, void* stream) {
    FT_CHECK_ARG(status_dev && dst);
    hipLaunchKernelGGL(poison_k, dim3(1), dim3(1), 0, ST(stream), status_dev, dst);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
