// The three recurrences of a flow as ONE forward launch chain over a time chunk.
//
// The attention LSTM of a flow (flowtron.py:647-651, :689-694) has no dependence on the decoder LSTM: it consumes the
// shifted mel input only.  Its output feeds the attention, whose context feeds the two-layer decoder LSTM -- a dependence
// at CHUNK granularity, not per step.  So the sequence is cut into time chunks and launch i of a phase runs
//     attention LSTM, step  a0 + i      (chunk c+1)           [workgroup row y = 0: lstm.hip's single-layer step body]
//     decoder layers 0/1,  launch b0 + i (chunk c, wavefront)  [row y = 1: lstm2.hip's two-group body]
// as co-resident workgroups of one launch (12 waves per CU, 156 VGPRs), with the batched per-chunk work (query projection,
// attention, context, decoder input projection) between the phases.  A recurrence launch is priced at a fixed ~3 us + the
// weight stream, so the attention-LSTM steps ride almost free.  Forward only: the backward step bodies are
// 1024-thread workgroups that fill a CU on their own, a fused backward launch would just queue them.
// Requires the fragment path with 8 k-chunks per wave (H % 1024 == 0) and B <= 32.
#include <mutex>
#include <unordered_map>

#include "common.h"

#define FT_LSTM_NO_ENTRY
namespace lone {
#include "lstm.hip"
}
namespace ltwo {
#include "lstm2.hip"
}
#undef FT_LSTM_NO_ENTRY

namespace {

template <int MT>
__global__ __launch_bounds__(256, 2) void lstm3_fwd_step(lone::FwdP pa, ltwo::L2FwdP pb) {
    if (blockIdx.y == 0) {
        if ((int)blockIdx.x < (pa.H >> 2)) lone::lstm_fwd_body<1, MT, 8, false>(pa);
    } else {
        ltwo::lstm2_fwd_body<MT, 8, 4>(pb);
    }
}

}  // namespace

extern "C" int ft_lstm3_supported(int B, int H) { return (B >= 1 && B <= 32 && H >= 1024 && H % 1024 == 0) ? 1 : 0; }

extern "C" int ft_lstm3_chunk_fwd(const float* gx_a, const float* w_hh_a, float* y_a, float* gates_a, float* cell_a, void* work_a,
                                  int a0, int a1,
                                  const float* gx0, const float* w_hh0, const float* w_ih1, const float* bias1, const float* w_hh1,
                                  float* y0, float* gates0, float* cell0, float* y1, float* gates1, float* cell1, void* work_2,
                                  int b0, int b1, const int32_t* lens, int T, int B, int H, void* stream) {
    FT_CHECK_ARG(lens && T >= 1 && 0 <= a0 && a0 <= a1 && a1 <= T && 0 <= b0 && b0 <= b1 && b1 <= T + 1);
    if (!ft_lstm3_supported(B, H)) return ft_fail(FT_EUNSUPPORTED, "ft_lstm3_chunk_fwd: needs H %% 1024 == 0 and B <= 32 (H=%d B=%d)", H, B);
    const bool has_a = a1 > a0, has_b = b1 > b0;
    if (has_a) FT_CHECK_ARG(gx_a && w_hh_a && y_a && gates_a && cell_a && work_a && reinterpret_cast<uintptr_t>(work_a) % 256 == 0);
    if (has_b) FT_CHECK_ARG(gx0 && w_hh0 && w_ih1 && bias1 && w_hh1 && y0 && gates0 && cell0 && y1 && gates1 && cell1 && work_2 &&
                            reinterpret_cast<uintptr_t>(work_2) % 256 == 0);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int mt = B <= 16 ? 1 : 2;
    lone::FwdCarve ca{};
    ltwo::Fwd2Setup u{};
    if (has_a) {
        ca = lone::carve_fwd(work_a, B, H, mt);
        if (a0 == 0) {                                   // first chunk of the sequence: zero state, build the W_hh image
            FT_CHECK_HIP(hipMemsetAsync(work_a, 0, ca.state_bytes, st));
            hipLaunchKernelGGL(lone::make_wfrag_fwd, dim3(2048), dim3(256), 0, st, w_hh_a, ca.wfrag, H);
        }
    }
    if (has_b) {
        u = ltwo::setup_fwd2(work_2, B, H);
        u.p.gx0 = gx0; u.p.bias1 = bias1; u.p.lens = lens;
        u.p.y0 = y0; u.p.gates0 = gates0; u.p.cell0 = cell0; u.p.y1 = y1; u.p.gates1 = gates1; u.p.cell1 = cell1;
        u.p.T = T; u.p.B = B; u.p.H = H;
        if (b0 == 0) {
            FT_CHECK_HIP(hipMemsetAsync(work_2, 0, u.state_bytes, st));
            hipLaunchKernelGGL(ltwo::make_wfrag_fwd_cat, dim3(2048), dim3(256), 0, st, (const float*)nullptr, 0, w_hh0, H, u.w0, H);
            hipLaunchKernelGGL(ltwo::make_wfrag_fwd_cat, dim3(2048), dim3(256), 0, st, w_ih1, H, w_hh1, H, u.w1, H);
        }
    }
    const int na = a1 - a0, nb = b1 - b0, n = na > nb ? na : nb;
    for (int i = 0; i < n; ++i) {
        const bool da = i < na, db = i < nb;
        lone::FwdP pa{};
        if (da) {
            const int s = a0 + i;
            pa = lone::FwdP{gx_a, w_hh_a, lens, ca.hbuf[s & 1], ca.hbuf[(s + 1) & 1], ca.cstate, y_a, (long)H, gates_a, cell_a,
                            ca.wfrag, ca.hfrag[s & 1], ca.hfrag[(s + 1) & 1], s, T, B, H, 0};
        }
        if (db) u.p.s = b0 + i;
        if (da && db) {
            const dim3 grid(2 * (H >> 2), 2);
            if (mt == 1) hipLaunchKernelGGL(lstm3_fwd_step<1>, grid, dim3(256), 0, st, pa, u.p);
            else hipLaunchKernelGGL(lstm3_fwd_step<2>, grid, dim3(256), 0, st, pa, u.p);
        } else if (da) {
            if (mt == 1) hipLaunchKernelGGL((lone::lstm_fwd_step<1, 1, 8, false>), dim3(H >> 2), dim3(256), 0, st, pa);
            else hipLaunchKernelGGL((lone::lstm_fwd_step<1, 2, 8, false>), dim3(H >> 2), dim3(256), 0, st, pa);
        } else {
            if (mt == 1) hipLaunchKernelGGL((ltwo::lstm2_fwd_step<1, 8, 4>), dim3(2 * (H >> 2)), dim3(256), 0, st, u.p);
            else hipLaunchKernelGGL((ltwo::lstm2_fwd_step<2, 8, 4>), dim3(2 * (H >> 2)), dim3(256), 0, st, u.p);
        }
    }
    FT_CHECK_LAUNCH();
    return FT_OK;
}
