// Attention-CTC loss of Flowtron (reference flowtron.py:155-182, 245-274) as a banded dynamic programme.
//
// The reference loops over samples: pad a blank column (logit `blank_logprob`), slice [:T_b, :K_b+1], log_softmax over the
// K_b+1 classes, nn.CTCLoss(blank=0, reduction='mean', zero_infinity=True) against the trivial target 1..K_b, then
// averages over the batch.  Because the target labels are all distinct, the extended label sequence
// (blank,1,blank,2,...,K,blank) has 2K+1 states with the plain CTC transitions (stay, +1, and +2 into a label state), and
// every label class owns exactly ONE state -- so the gradient needs no per-class reduction.
//
// One workgroup per sample, one thread per state, the time recursion is the only sequential dimension:
//   forward : lse[t] (log-softmax normaliser, one wave per frame), alpha[t][s] in LDS (double buffer) + saved to HBM,
//             nll_b = -logsumexp(alpha[T-1][2K], alpha[T-1][2K-1]);  loss += nll_b / K_b / B.
//             When a gradient will be wanted the beta recursion runs IN THE SAME LAUNCH as a second workgroup per sample
//             (grid (B, 2): the two recursions are independent and each is latency-bound in T), saving beta to HBM.
//   backward: d loss / d logit[t][k] = (softmax[t][k] - exp(alpha+beta+nll-logp)[state 2k-1]) * g/(K_b B), an elementwise
//             kernel over (b, t, k) -- no recursion left in the backward pass.
// The next frame's logits are prefetched into registers before the barrier of the current frame, so the per-frame cost is
// one LDS round trip + three exp/log, not a global-memory round trip.
#include "common.h"

namespace {

// workgroup barrier for the DP loops: LDS traffic drained, VMEM left alone.  `__syncthreads()` makes hipcc wait for vmcnt(0) too, i.e.
// for the alpha / beta row just stored AND for the next step's log-probabilities just requested: a memory round trip per DP step
// (0.64 us per step, measured) for a recurrence whose own work is ~0.15 us.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}


// log-sum-exp of the DP recursions through the hardware transcendentals (v_exp_f32 / v_log_f32, ~1 ulp): the arguments of exp are
// <= 0 and the sum lies in [1, 3], so the libm range handling bought nothing -- and its ~60 instructions per step were most of the
// recursion's 0.55 us per frame (round 6: 479 -> 385 us per call, profiles/r06_*; the parity tests hold the loss to the same 1e-5)
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == -INFINITY) return -INFINITY;
    return m + fast_log(fast_exp(a - m) + fast_exp(b - m) + fast_exp(c - m));
}
__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    if (m == -INFINITY) return -INFINITY;
    return m + fast_log(fast_exp(a - m) + fast_exp(b - m));
}

// The samples of one call: F stacked groups of B samples (one group per flow, FlowtronLoss: flowtron.py:245-274), sample
// bs = f * B + b reads lp[f][b] with the lengths of utterance b.  rev bit f: that flow's log-probabilities are stored in REVERSED
// time (an AR_Back_Step: frame t of the utterance is row T_b - 1 - t of the tensor) -- the reference flips and rolls the tensor
// there and back (:250-271); here the row index is mirrored where it is formed, in the loss AND in the gradient, so neither a
// reversed copy nor a concatenation of the flows exists.  F = 1, rev = 0: the plain [B,T,L] call.
struct CtcSrc {
    const float* lp[8];
    float* dlp[8];
    unsigned rev;
    int B;                  // samples per group
};
struct CtcSample { int f, b; bool rev; };
__device__ __forceinline__ CtcSample ctc_sample(const CtcSrc& src, int bs) {
    const int f = bs / src.B;
    return CtcSample{f, bs - f * src.B, ((src.rev >> f) & 1u) != 0};
}

// lse[b][t] = log(exp(blank) + sum_{k<K_b} exp(lp[b][t][k])) for t < T_b  (grid: (ceil(T/4), B), one wave per frame)
__global__ __launch_bounds__(256) void ctc_lse_k(CtcSrc src, const int* __restrict__ in_lens,
                                                 const int* __restrict__ out_lens, float blank, float* __restrict__ lse,
                                                 int T, int L) {
    const int bs = blockIdx.y, t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const CtcSample sm = ctc_sample(src, bs);
    const int K = min(in_lens[sm.b], L), Tb = min(out_lens[sm.b], T);
    if (t >= Tb) return;
    const float* row = src.lp[sm.f] + ((size_t)sm.b * T + (sm.rev ? Tb - 1 - t : t)) * L;
    float m = blank;
    for (int k = lane; k < K; k += 64) m = fmaxf(m, row[k]);
    m = wave_max(m);
    float s = (lane == 0) ? expf(blank - m) : 0.f;
    for (int k = lane; k < K; k += 64) s += expf(row[k] - m);
    s = wave_sum(s);
    if (lane == 0) lse[(size_t)bs * T + t] = m + logf(s);
}

// one workgroup per sample, thread s = extended state
__device__ __forceinline__ void ctc_alpha_body(const CtcSrc& src, const int* __restrict__ in_lens,
                                               const int* __restrict__ out_lens, float blank, const float* __restrict__ lse,
                                               float* __restrict__ alpha, float* __restrict__ nll, float* __restrict__ loss,
                                               int B, int T, int L, float* sm) {
    // sm: 2 x (S + 2), two -inf guard cells in front of each.  B = ALL samples of the call (the batch mean's divisor)
    const int b = blockIdx.x, s = threadIdx.x;
    const CtcSample smp = ctc_sample(src, b);
    const int K = min(in_lens[smp.b], L), Tb = min(out_lens[smp.b], T);
    const bool rev = smp.rev;
#define CTC_ROW(tt) ((size_t)(rev ? Tb - 1 - (tt) : (tt)) * L)
    const int S = 2 * K + 1, SP = 2 * L + 1 + 2;
    float* buf[2] = {sm + 2, sm + SP + 2};
    if (s < 2) { sm[s] = -INFINITY; sm[SP + s] = -INFINITY; }
    const bool on = s < S;
    const bool lab = (s & 1) != 0;
    const int k = (s - 1) >> 1;
    const float* lpb = src.lp[smp.f] + (size_t)smp.b * T * L;
    const float* lseb = lse + (size_t)b * T;
    float* ab = alpha + (size_t)b * T * (2 * L + 1);
    // t = 0
    float logit = (on && lab && Tb > 0) ? lpb[CTC_ROW(0) + k] : blank;
    float norm = (Tb > 0) ? lseb[0] : 0.f;
    float a = -INFINITY;
    if (on && s < 2) a = logit - norm;
    if (on) { buf[0][s] = a; if (Tb > 0) ab[s] = a; }
    // The emission terms of four steps at a time, requested a whole group (four DP steps) before they are used: with a one-step
    // prefetch the recurrence paid a memory round trip per step (the load issued at the top of a step is needed at its bottom).
    float lg[4], nm[4], lgn[4], nmn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int tt = 1 + u;
        lg[u] = (tt < Tb && on && lab) ? lpb[CTC_ROW(tt) + k] : blank;
        nm[u] = tt < Tb ? lseb[tt] : 0.f;
    }
    __syncthreads();
    int cur = 0;
    for (int tb = 1; tb < Tb; tb += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tt = tb + 4 + u;
            lgn[u] = (tt < Tb && on && lab) ? lpb[CTC_ROW(tt) + k] : blank;
            nmn[u] = tt < Tb ? lseb[tt] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tb + u;
            if (t < Tb) {                                             // (uniform over the workgroup)
                if (on) {
                    const float* pv = buf[cur];
                    const float x2 = (lab && s >= 3) ? pv[s - 2] : -INFINITY;
                    a = lse3(pv[s], pv[s - 1], x2) + (lab ? lg[u] : blank) - nm[u];
                    buf[cur ^ 1][s] = a;
                    ab[(size_t)t * (2 * L + 1) + s] = a;
                }
                lds_barrier();
                cur ^= 1;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { lg[u] = lgn[u]; nm[u] = nmn[u]; }
    }
    if (s == 0) {
        float v = INFINITY;
        if (Tb > 0 && K > 0) v = -lse2(buf[cur][S - 1], buf[cur][S - 2]);
        nll[b] = v;
        if (v != INFINITY && K > 0) atomicAdd(loss, v / (float)K / (float)B);      // zero_infinity=True, reduction='mean', batch mean
    }
}

// beta[t][s] (emission of frame t included, like alpha), saved to HBM; same thread <-> state mapping as alpha
__device__ __forceinline__ void ctc_beta_body(const CtcSrc& src, const int* __restrict__ in_lens,
                                              const int* __restrict__ out_lens, float blank, const float* __restrict__ lse,
                                              float* __restrict__ beta, int B, int T, int L, float* sm) {
    // sm: 2 x (S + 2), two -inf guard cells BEHIND each
    const int b = blockIdx.x, s = threadIdx.x;
    const CtcSample smp = ctc_sample(src, b);
    const int K = min(in_lens[smp.b], L), Tb = min(out_lens[smp.b], T);
    const bool rev = smp.rev;
    if (Tb <= 0 || K <= 0) return;
    const int S = 2 * K + 1, SP = 2 * L + 1 + 2;
    float* buf[2] = {sm, sm + SP};
    if (s < 2) { sm[S + s] = -INFINITY; sm[SP + S + s] = -INFINITY; }
    const bool on = s < S;
    const bool lab = (s & 1) != 0;
    const int k = (s - 1) >> 1;
    const float* lpb = src.lp[smp.f] + (size_t)smp.b * T * L;
    const float* lseb = lse + (size_t)b * T;
    float* bb = beta + (size_t)b * T * (2 * L + 1);
    int t = Tb - 1;
    float logit = (on && lab) ? lpb[CTC_ROW(t) + k] : blank;
    float norm = lseb[t];
    float be = -INFINITY;
    if (on && s >= S - 2) be = logit - norm;
    if (on) { buf[0][s] = be; bb[(size_t)t * (2 * L + 1) + s] = be; }
    float lg[4], nm[4], lgn[4], nmn[4];                              // emission terms of steps t - 1 .. t - 4, a group ahead (see alpha)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int tt = Tb - 2 - u;
        lg[u] = (tt >= 0 && on && lab) ? lpb[CTC_ROW(tt) + k] : blank;
        nm[u] = tt >= 0 ? lseb[tt] : 0.f;
    }
    __syncthreads();
    int cur = 0;
    for (int tb = Tb - 2; tb >= 0; tb -= 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tt = tb - 4 - u;
            lgn[u] = (tt >= 0 && on && lab) ? lpb[CTC_ROW(tt) + k] : blank;
            nmn[u] = tt >= 0 ? lseb[tt] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            t = tb - u;
            if (t >= 0) {
                if (on) {
                    const float* nx = buf[cur];
                    const float x2 = (lab && s + 2 < S) ? nx[s + 2] : -INFINITY;
                    be = lse3(nx[s], nx[s + 1], x2) + (lab ? lg[u] : blank) - nm[u];
                    buf[cur ^ 1][s] = be;
                    bb[(size_t)t * (2 * L + 1) + s] = be;
                }
                lds_barrier();
                cur ^= 1;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { lg[u] = lgn[u]; nm[u] = nmn[u]; }
    }
}

// grid (B, 1 or 2): y = 0 alpha recursion (+ loss), y = 1 beta recursion -- independent, both latency-bound in T
#undef CTC_ROW
__global__ __launch_bounds__(1024) void ctc_alpha_beta_k(CtcSrc src, const int* __restrict__ in_lens,
                                                         const int* __restrict__ out_lens, float blank, const float* __restrict__ lse,
                                                         float* __restrict__ alpha, float* __restrict__ beta, float* __restrict__ nll,
                                                         float* __restrict__ loss, int B, int T, int L) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    if (blockIdx.y == 0) ctc_alpha_body(src, in_lens, out_lens, blank, lse, alpha, nll, loss, B, T, L, sm);
    else ctc_beta_body(src, in_lens, out_lens, blank, lse, beta, B, T, L, sm);
}
__global__ __launch_bounds__(1024) void ctc_beta_k(CtcSrc src, const int* __restrict__ in_lens,
                                                   const int* __restrict__ out_lens, float blank, const float* __restrict__ lse,
                                                   float* __restrict__ beta, int B, int T, int L) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    ctc_beta_body(src, in_lens, out_lens, blank, lse, beta, B, T, L, sm);
}

// dlp[b][t][k] for every (b, t, k): grid (ceil(T/4), B), one wave per frame, lanes over labels
__global__ __launch_bounds__(256) void ctc_grad_k(CtcSrc src, const int* __restrict__ in_lens,
                                                  const int* __restrict__ out_lens, const float* __restrict__ lse,
                                                  const float* __restrict__ alpha, const float* __restrict__ beta,
                                                  const float* __restrict__ nll, const float* __restrict__ gout,
                                                  int B, int T, int L) {
    // B = ALL samples of the call; t = the frame in natural time, ts = its row in the flow's own (possibly reversed) tensor
    const int b = blockIdx.y, t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    const CtcSample smp = ctc_sample(src, b);
    const int K = min(in_lens[smp.b], L), Tb = min(out_lens[smp.b], T);
    const int ts = (smp.rev && t < Tb) ? Tb - 1 - t : t;
    float* drow = src.dlp[smp.f] + ((size_t)smp.b * T + ts) * L;
    const float nl = nll[b];
    if (t >= Tb || K <= 0 || nl == INFINITY) {                    // padding frame / zero_infinity: zero gradient
        for (int k = lane; k < L; k += 64) drow[k] = 0.f;
        return;
    }
    const float* row = src.lp[smp.f] + ((size_t)smp.b * T + ts) * L;
    const float* ar = alpha + ((size_t)b * T + t) * (2 * L + 1);
    const float* br = beta + ((size_t)b * T + t) * (2 * L + 1);
    const float norm = lse[(size_t)b * T + t];
    const float scale = gout[0] / (float)K / (float)B;
    for (int k = lane; k < L; k += 64) {
        float g = 0.f;
        if (k < K) {
            const float logp = row[k] - norm;
            g = (expf(logp) - expf(ar[2 * k + 1] + br[2 * k + 1] + nl - logp)) * scale;
        }
        drow[k] = g;
    }
}

}  // namespace

extern "C" size_t ft_attn_ctc_workspace_floats(int B, int T, int L) {
    return 2 * (size_t)B * T * (2 * (size_t)L + 1) + (size_t)B * T + (size_t)B;        // alpha | beta | lse | nll
}

namespace {
int ctc_fwd(const CtcSrc& src, int F, const int32_t* in_lens, const int32_t* out_lens, float blank_logprob, float* work, float* loss,
            int T, int L, int with_beta, hipStream_t st) {
    const int BT = F * src.B;                                       // samples of the call
    const size_t na = (size_t)BT * T * (2 * (size_t)L + 1);
    float* alpha = work;
    float* beta = alpha + na;
    float* lse = beta + na;
    float* nll = lse + (size_t)BT * T;
    const int threads = cdiv(2 * L + 1, 64) * 64;
    FT_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(float), st));
    hipLaunchKernelGGL(ctc_lse_k, dim3(cdiv(T, 4), BT), dim3(256), 0, st, src, in_lens, out_lens, blank_logprob, lse, T, L);
    hipLaunchKernelGGL(ctc_alpha_beta_k, dim3(BT, with_beta ? 2 : 1), dim3(threads), sizeof(float) * 2 * (2 * L + 3), st, src, in_lens,
                       out_lens, blank_logprob, lse, alpha, beta, nll, loss, BT, T, L);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
int ctc_bwd(const CtcSrc& src, int F, const int32_t* in_lens, const int32_t* out_lens, float blank_logprob, float* work,
            const float* gout_dev, int T, int L, int beta_ready, hipStream_t st) {
    const int BT = F * src.B;
    const size_t na = (size_t)BT * T * (2 * (size_t)L + 1);
    const float* alpha = work;
    float* beta = work + na;
    const float* lse = beta + na;
    const float* nll = lse + (size_t)BT * T;
    if (!beta_ready) {
        const int threads = cdiv(2 * L + 1, 64) * 64;
        hipLaunchKernelGGL(ctc_beta_k, dim3(BT), dim3(threads), sizeof(float) * 2 * (2 * L + 3), st, src, in_lens, out_lens, blank_logprob,
                           lse, beta, BT, T, L);
    }
    hipLaunchKernelGGL(ctc_grad_k, dim3(cdiv(T, 4), BT), dim3(256), 0, st, src, in_lens, out_lens, lse, alpha, beta, nll, gout_dev, BT, T, L);
    FT_CHECK_LAUNCH();
    return FT_OK;
}
}  // namespace

extern "C" int ft_attn_ctc_fwd(const float* lp, const int32_t* in_lens, const int32_t* out_lens, float blank_logprob,
                               float* work, float* loss, int B, int T, int L, int with_beta, void* stream) {
    FT_CHECK_ARG(lp && in_lens && out_lens && work && loss && B >= 1 && T >= 1 && L >= 1);
    if (2 * L + 1 > 1024) return ft_fail(FT_EUNSUPPORTED, "ft_attn_ctc_fwd: L=%d needs more than 1024 states", L);
    CtcSrc src{};
    src.lp[0] = lp; src.rev = 0; src.B = B;
    return ctc_fwd(src, 1, in_lens, out_lens, blank_logprob, work, loss, T, L, with_beta, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int ft_attn_ctc_bwd(const float* lp, const int32_t* in_lens, const int32_t* out_lens, float blank_logprob,
                               float* work, const float* gout_dev, float* dlp, int B, int T, int L, int beta_ready, void* stream) {
    FT_CHECK_ARG(lp && in_lens && out_lens && work && gout_dev && dlp && B >= 1 && T >= 1 && L >= 1);
    if (2 * L + 1 > 1024) return ft_fail(FT_EUNSUPPORTED, "ft_attn_ctc_bwd: L=%d needs more than 1024 states", L);
    CtcSrc src{};
    src.lp[0] = lp; src.dlp[0] = dlp; src.rev = 0; src.B = B;
    return ctc_bwd(src, 1, in_lens, out_lens, blank_logprob, work, gout_dev, T, L, beta_ready, reinterpret_cast<hipStream_t>(stream));
}

// F flows' log-probabilities [B,T,L] each, as one stacked call of F * B samples (see CtcSrc): lp / dlp = host arrays of F device
// pointers, reversed[f] != 0: flow f's tensor is in reversed time.  work: ft_attn_ctc_workspace_floats(F * B, T, L) floats.
extern "C" int ft_attn_ctc_fwd_multi(const float* const* lp, const int32_t* reversed, int F, const int32_t* in_lens,
                                     const int32_t* out_lens, float blank_logprob, float* work, float* loss, int B, int T, int L,
                                     int with_beta, void* stream) {
    FT_CHECK_ARG(lp && reversed && F >= 1 && F <= 8 && in_lens && out_lens && work && loss && B >= 1 && T >= 1 && L >= 1);
    if (2 * L + 1 > 1024) return ft_fail(FT_EUNSUPPORTED, "ft_attn_ctc_fwd_multi: L=%d needs more than 1024 states", L);
    CtcSrc src{};
    src.B = B;
    for (int f = 0; f < F; ++f) { FT_CHECK_ARG(lp[f] != nullptr); src.lp[f] = lp[f]; if (reversed[f]) src.rev |= 1u << f; }
    return ctc_fwd(src, F, in_lens, out_lens, blank_logprob, work, loss, T, L, with_beta, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int ft_attn_ctc_bwd_multi(const float* const* lp, const int32_t* reversed, int F, const int32_t* in_lens,
                                     const int32_t* out_lens, float blank_logprob, float* work, const float* gout_dev, float* const* dlp,
                                     int B, int T, int L, int beta_ready, void* stream) {
    FT_CHECK_ARG(lp && reversed && dlp && F >= 1 && F <= 8 && in_lens && out_lens && work && gout_dev && B >= 1 && T >= 1 && L >= 1);
    if (2 * L + 1 > 1024) return ft_fail(FT_EUNSUPPORTED, "ft_attn_ctc_bwd_multi: L=%d needs more than 1024 states", L);
    CtcSrc src{};
    src.B = B;
    for (int f = 0; f < F; ++f) {
        FT_CHECK_ARG(lp[f] != nullptr && dlp[f] != nullptr);
        src.lp[f] = lp[f]; src.dlp[f] = dlp[f];
        if (reversed[f]) src.rev |= 1u << f;
    }
    return ctc_bwd(src, F, in_lens, out_lens, blank_logprob, work, gout_dev, T, L, beta_ready, reinterpret_cast<hipStream_t>(stream));
}
