/*
 * flowtron_hip.h -- C ABI of libflowtron_hip.so (MI355X / gfx950 only).
 *
 * The reference (NVIDIA/flowtron) has no FFI: its only boundary is the Python
 * module API (flowtron.py Flowtron.forward/infer, audio_processing.py
 * TacotronSTFT, distributed.py).  This header is the boundary a maintainer
 * binds instead of the PyTorch/cuDNN/cuBLAS op call sites listed below; the
 * ctypes stub that does it is flowtron_amd/_lib.py.
 *
 * Conventions
 *   - every entry point returns 0 on success, a negative FT_E* code on error;
 *     ft_last_error() returns a thread-local message for the last failure.
 *   - all buffers are caller-owned DEVICE pointers (fp32 unless stated);
 *     the library never allocates, frees or retains them.  Lengths are device
 *     int32 arrays.  No entry point synchronises: work is enqueued on `stream`
 *     (a hipStream_t passed as void*), in order, and is hipGraph-capturable.
 *   - `mode` selects the MFMA operand type of matmul-shaped work:
 *     FT_F32 = fp32 operands (v_mfma_f32_16x16x4_f32, exact fp32, parity mode),
 *     FT_BF16 = operands rounded to bf16 on the way into the matrix core
 *     (v_mfma_f32_16x16x32_bf16), fp32 accumulate.  Storage stays fp32.
 *     FT_F16 = the same code path with fp16 operands (v_mfma_f32_16x16x32_f16):
 *     the reference's fp16 AMP configuration (train.py:254,292).  Values beyond
 *     65504 round to +-inf, which is what torch's GradScaler inf check expects.
 *     Entry points with a `mode` argument accept FT_F16 directly; the ones that
 *     consume or produce 16-bit images / fragments without a mode argument have
 *     a twin of identical signature and the suffix _f16 (end of this header).
 *   - time-major activations: [T,B,C] row-major, like the reference's
 *     internal layout after flowtron.py:884.
 *
 * Where this header departs from the conventions SURVEY.md 8(b) sketched (the
 * reference itself has no FFI to be compatible with; these are choices):
 *   - no `ft_ctx` handle with create / destroy entry points: the library keeps NO
 *     per-device state.  Every entry point works on the device of the stream it
 *     is handed (the caller -- torch -- has made it current); scratch, status
 *     words and weight images are caller-owned buffers passed per call, so the
 *     library is re-entrant across devices and processes by construction.
 *   - positional arguments instead of one args struct per op, except where an op
 *     has more than ~12 operands (ft_gemm, ft_gemm_img, ft_decode_flow,
 *     ft_cumm_attn_*: structs).
 *   - no `ft_allreduce_flat`: the gradient exchange is torch.distributed's RCCL
 *     all-reduce of the flat arena (north_star: "a single RCCL all-reduce over
 *     xGMI per step"; flowtron_amd/dist.py); the library only supplies the
 *     device-side poison / non-finite-norm guard around it.
 *   - `ft_dense_conv_affine` (dense -> dense -> 1x1 conv -> coupling as ONE
 *     kernel) is not built: a 1024-wide layer pair per row tile needs both weight
 *     matrices (4 MB of 16-bit operands) per tile from the L2 -- 2.6 GB per
 *     call at 32-row tiles, the largest that keep two activation tiles in LDS --
 *     which is slower than the three image GEMMs it would replace.  What IS fused:
 *     bias + tanh in the GEMM epilogue, the activation backward + bias gradient
 *     in the gradient's image pass (ft_bf16_image_rows_act_bwd), the coupling
 *     with its own backward (ft_affine_*), the masked NLL sums (ft_masked_sum).
 *   - `ft_decode_init` / `ft_decode_steps` became ft_decode_flow (one call per
 *     flow: a persistent launch, or a hipGraph of staged frames).
 *   - `ft_lstm_seq` "multi-layer": one call per layer (ft_lstm_persist_* /
 *     ft_lstm_seq_*), with the inter-layer projection as a batched GEMM between
 *     them (faster than the fused two-layer wavefront, which is kept as
 *     ft_lstm2_seq_* for shapes the persistent kernels do not take).
 */
#ifndef FLOWTRON_HIP_H
#define FLOWTRON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FT_ABI_VERSION 14

enum { FT_OK = 0, FT_EINVAL = -1, FT_EHIP = -2, FT_EUNSUPPORTED = -3 };
enum { FT_F32 = 0, FT_BF16 = 1, FT_F16 = 2 };
enum { FT_ACT_NONE = 0, FT_ACT_TANH = 1, FT_ACT_RELU = 2, FT_ACT_SIGMOID = 3 };
enum { FT_GEMM_SPLITK = 1, FT_GEMM_SPLITK_DET = 2, FT_GEMM_C16 = 4 };   /* FT_GEMM_C16 (ABI 13, ft_gemm_img only): C is a 16-bit matrix of the
 * call's operand format (bf16; fp16 for the _f16 twin), ldc in 16-bit elements, beta = 0, no split-K: what the persistent forward recurrence reads as gx */

int ft_abi_version(void);
const char* ft_last_error(void);
/* test hook: n_wg workgroups that each hold a whole CU for `ticks` of the 100 MHz wall clock (clamped to 5 s) on `stream` -- a
 * stand-in for a foreign kernel (an in-flight collective) beside a whole-chip persistent launch (tests/test_gpu_dist.py) */
int ft_debug_hold_cus(int n_wg, int64_t ticks, void* stream);

/* ---- GEMM ---------------------------------------------------------------
 * C[b][m][n] = act( alpha * sum_k A[b](m,k) * B[b](k,n) + beta * C[b][m][n] + bias[n] )
 * A(m,k) = A[b*bsA + m*sAm + k*sAk],  B(k,n) = B[b*bsB + k*sBk + n*sBn],
 * C row-major with leading dimension ldc.  Replaces every nn.Linear / Conv1d /
 * torch.bmm call site of the hot path: flowtron.py:568-571 (Q/K/V), :590 (context),
 * :758 (gate), :767-768 (dense, 1x1 conv), the LSTM input projections inside
 * nn.LSTM (:654-655, :488), the encoder Conv1d (:479-483, as im2col GEMM) and
 * their autograd-derived transposes. */
typedef struct {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, batch;
    int64_t sAm, sAk, sBk, sBn, ldc;
    int64_t bsA, bsB, bsC;
    float alpha, beta;
    int act, mode;
    int flags;   /* FT_GEMM_SPLITK: allow split-K with fp32 atomics when the output has few tiles and K is long (weight
                  * gradients over T*B rows).  Summation order is then not reproducible bit-for-bit, so the forward path
                  * never sets it. */
    void* work;  /* optional scratch (256-byte aligned) of ft_gemm_workspace_bytes() bytes.  When given (FT_BF16, batch 1,
                  * problem large enough for the query to be non-zero) the operands are first rewritten once as zero-padded
                  * bf16 images with the reduction dimension innermost and the GEMM runs from those through direct
                  * global->LDS DMA; same rounding (RNE to bf16, fp32 accumulate) as the staging path taken when it is NULL. */
    size_t work_bytes;
} ft_gemm_args;
size_t ft_gemm_workspace_bytes(const ft_gemm_args* a);
int ft_gemm(const ft_gemm_args* a, void* stream);

/* bf16 operand images (FT_BF16 training path): a row-major fp32 matrix [rows][cols] (row stride ld) is rounded ONCE to a
 * zero-padded bf16 image [ceil256(rows + 32)][ceil256(cols)] (ft_bf16_image_bytes bytes, 256-byte aligned) and then feeds
 * every GEMM that reads it, in either role:
 *   k-contiguous operand (a_kmajor / b_kmajor = 0): image rows are the operand's m (or n) index, columns the reduction;
 *   k-major operand (= 1): image rows are the reduction index, columns the m (or n) index -- read through the LDS
 *   transpose-read, so the weight-gradient GEMMs dW = dY^T X (both operands k-major) need no transposed copies.
 * A / B may point INSIDE an image (row offset * ld for a time shift, column offset % 8 == 0 for a column block of a
 * weight); whatever lies beyond the logical extent must be finite and is multiplied by the partner's zero padding (k) or
 * dropped by the epilogue (m, n).  lda / ldb = image row stride in elements = ceil256(cols).  Epilogue as ft_gemm. */
typedef struct {
    const void* A; const void* B; float* C; const float* bias;
    int M, N, K;
    int64_t lda, ldb, ldc;
    int a_kmajor, b_kmajor;
    float alpha, beta;
    int act, flags;
    /* compact row space (pack-by-length, the reference's pack_padded_sequence at flowtron.py:689-694): the images were made by
     * ft_bf16_image_rows from the VALID rows only.  compact = 0: plain.  1: the M rows are compact rows -- workgroup tiles at or
     * beyond *rows_dev exit at once, and compact row m is written to C row rowmap[m] (negative: dropped); M = the capacity the
     * grid is sized for; no split-K.  2: the reduction runs over compact rows -- K = capacity, k-steps at or beyond
     * *rows_dev - k_shift are not visited (k_shift = the row offset already applied to A for a one-step time shift). */
    const int32_t* rowmap; const int32_t* rows_dev;
    int compact, k_shift;
    /* optional rank-1 epilogue term (ABI 10): C[row][col] += r1_row[row] * r1_col[col] in fp32, row = the OUTPUT row (after the
     * row map), before the activation; both NULL = none; not with split-K.  The gate layer's input gradient dgate (x) w_gate rides
     * on the decoder input projection's dX GEMM this way (flowtron.py:758-761: both read [h_att ; ctx]). */
    const float* r1_row; const float* r1_col;
    /* deterministic split-K (ABI 12; flags & FT_GEMM_SPLITK_DET, beta = 0, no activation, compact = 0, N % 4 == 0): the k-slices
     * write their partial products side by side into split_work (ft_gemm_img_split_work_bytes(M, N, K) bytes, 16-byte aligned) with
     * plain stores and a second kernel adds them in ascending slice order (+ bias): the result is a function of the operands, which
     * the fp32 atomics of FT_GEMM_SPLITK are not (their order varies from run to run).  For FORWARD GEMMs with few output tiles and
     * a long reduction (the encoder convolutions' split-image product, flowtron.py:499-502); NULL / 0: never splits. */
    void* split_work; size_t split_work_bytes;
} ft_gemm_img_args;
size_t ft_gemm_img_split_work_bytes(int M, int N, int K);
size_t ft_bf16_image_bytes(int64_t rows, int64_t cols);
int ft_bf16_image(const float* src, int64_t ld, int64_t rows, int64_t cols, void* dst, void* stream);
/* the same image plus colsum[c] = sum_r src[r][c] in fp32 (bias gradients ride on the conversion pass of the output
 * gradient; colsum [cols] is overwritten; row slabs combine with fp32 atomics like ft_colsum) */
int ft_bf16_image_colsum(const float* src, int64_t ld, int64_t rows, int64_t cols, void* dst, float* colsum, void* stream);
int ft_gemm_img(const ft_gemm_img_args* a, void* stream);
/* _acc forms (ABI 12) of the three image passes that also produce column sums (ft_bf16_image_colsum, ft_bf16_image_rows with a colsum,
 * ft_bf16_image_rows_act_bwd): the sums are ADDED to colsum, which the caller has zeroed (a slice of one zeroed slab per backward
 * pass) -- the plain forms clear it with a memset dispatch of their own, 11 per training step. */
int ft_bf16_image_colsum_acc(const float* src, int64_t ld, int64_t rows, int64_t cols, void* dst, float* colsum, void* stream);
int ft_bf16_image_rows_acc(const float* src, int64_t ld, int64_t cap_rows, int64_t cols, void* dst, float* colsum,
                           const int32_t* rowmap, const int32_t* rows_dev, void* stream);
int ft_bf16_image_rows_act_bwd_acc(const float* dy, int64_t ld, const float* y, int64_t ldy, int act, int64_t cap_rows, int64_t cols,
                                   void* dst, float* colsum, const int32_t* rowmap, const int32_t* rows_dev, void* stream);
int ft_bf16_image_colsum_acc_f16(const float* src, int64_t ld, int64_t rows, int64_t cols, void* dst, float* colsum, void* stream);
int ft_bf16_image_rows_acc_f16(const float* src, int64_t ld, int64_t cap_rows, int64_t cols, void* dst, float* colsum,
                               const int32_t* rowmap, const int32_t* rows_dev, void* stream);
int ft_bf16_image_rows_act_bwd_acc_f16(const float* dy, int64_t ld, const float* y, int64_t ldy, int act, int64_t cap_rows, int64_t cols,
                                       void* dst, float* colsum, const int32_t* rowmap, const int32_t* rows_dev, void* stream);
/* Split images (ABI 11): x = hi + lo, hi = op16(x), lo = op16(x - hi).  dst = [rows][3 cols] 16-bit (ft_bf16_image_bytes(rows, 3 cols),
 * row stride ceil256(3 cols)): an activation (weight = 0) as [hi | lo | hi], a weight matrix (weight = 1) as [hi | hi | lo], so that ONE
 * ft_gemm_img with K = 3 cols computes x_hi w_hi + x_lo w_hi + x_hi w_lo = x . w to ~2^-17 -- fp32-grade products at three times a
 * 16-bit GEMM's cost.  Used for the FORWARD of the encoder's convolutions (flowtron.py:499-502): their 16-bit rounding, renormalised
 * by the instance norm behind them, was the source of the 0.11 relative deviation of the text-embedding gradient in bf16 training
 * (measured round 5: fp32 forward products alone bring it to 0.008).  cols % 8 == 0. */
int ft_bf16_image_split3(const float* src, int64_t ld, int64_t rows, int64_t cols, void* dst, int weight, void* stream);
int ft_bf16_image_split3_f16(const float* src, int64_t ld, int64_t rows, int64_t cols, void* dst, int weight, void* stream);
/* ABI 14: the split ACTIVATION image of the im2col matrix of ft_im2col(x, lens, L, B, C, KW) -- [L * B rows][3 * C * KW] as [hi | lo | hi] --
 * made straight from x [L][B][C] (flowtron.py:499-502: the encoder convolution as a GEMM): the fp32 column matrix is never written.
 * dst: ft_bf16_image_bytes(L * B, 3 * C * KW) bytes; (C * KW) % 8 == 0, KW odd. */
int ft_bf16_image_split3_im2col(const float* x, const int32_t* lens, int L, int B, int C, int KW, void* dst, void* stream);
int ft_bf16_image_split3_im2col_f16(const float* x, const int32_t* lens, int L, int B, int C, int KW, void* dst, void* stream);
/* The weight images of one forward pass in ONE launch (ABI 12): descs = HOST array of n <= 32 descriptors, kind 0 = what
 * ft_bf16_image(src, ld, rows, cols, dst) writes, kind 1 = ft_bf16_image_split3(src, ld, rows, cols, dst, weight = 1).  A training step
 * rounds its 23 weight matrices afresh every iteration; one dispatch instead of 23 of 5-15 us each. */
typedef struct { const float* src; int64_t ld; int64_t rows; int64_t cols; void* dst; int kind; } ft_img_desc;
int ft_bf16_image_table(const ft_img_desc* descs, int n, void* stream);
int ft_bf16_image_table_f16(const ft_img_desc* descs, int n, void* stream);
/* Pack-by-length row map of a time-major [T][B][*] activation (flowtron.py:689-694 packs, here without a host sync): compact
 * rows are batch-major -- utterance b contributes rows (t, b), t < lens[b], then ONE separator: row (lens[b], b) when
 * lens[b] < T (the utterance's first padded frame, which stands for all of them: every padded frame of b holds the same
 * values), else -1 (a zero row).  rowmap needs T*B + B entries; rows_dev[0] = sum_b (lens[b] + 1).  The separator rows make the
 * one-step shift of the recurrent weight gradient (dW_hh = sum_t dgates_t^T h_{t-1}) a one-ROW shift in compact space. */
int ft_rowmap_build(const int32_t* lens, int32_t* rowmap, int32_t* rows_dev, int T, int B, void* stream);
/* compact image: image row i = bf16(src row rowmap[i]) for i < *rows_dev (negative: zeros), zeros up to
 * ceil256(*rows_dev + 32); dst holds ft_bf16_image_bytes(cap_rows, cols).  colsum (optional) as ft_bf16_image_colsum. */
int ft_bf16_image_rows(const float* src, int64_t ld, int64_t cap_rows, int64_t cols, void* dst, float* colsum,
                       const int32_t* rowmap, const int32_t* rows_dev, void* stream);
/* The output gradient of an ACTIVATED dense layer (tanh / relu / sigmoid fused in ft_gemm_img's epilogue, flowtron.py:453-464)
 * straight into its operand image: dst = image(dy * act'(pre)) over the compact rows of `rowmap`, act' through the saved output
 * y = act(pre) like ft_act_bwd; colsum [cols] = the bias gradient.  Replaces ft_act_bwd + ft_bf16_image_rows. */
int ft_bf16_image_rows_act_bwd(const float* dy, int64_t ld, const float* y, int64_t ldy, int act, int64_t cap_rows, int64_t cols,
                               void* dst, float* colsum, const int32_t* rowmap, const int32_t* rows_dev, void* stream);
int ft_bf16_image_rows_act_bwd_f16(const float* dy, int64_t ld, const float* y, int64_t ldy, int act, int64_t cap_rows, int64_t cols,
                                   void* dst, float* colsum, const int32_t* rowmap, const int32_t* rows_dev, void* stream);
/* One column block of a compact image: src [*, cols] -> columns [col_off, col_off + cols) of dst (row stride dst_ld elements; the
 * image is sized by ft_bf16_image_bytes for its TOTAL width), columns up to col_off + fill_cols zeroed.  A Linear over two
 * inputs ([h_att ; ctx] W^T, flowtron.py:758-765) then runs as ONE GEMM over one image instead of two K pieces. */
int ft_bf16_image_rows_into(const float* src, int64_t ld, int64_t cap_rows, int64_t cols, void* dst, int64_t dst_ld, int64_t col_off,
                            int64_t fill_cols, const int32_t* rowmap, const int32_t* rows_dev, void* stream);
int ft_bf16_image_rows_into_f16(const float* src, int64_t ld, int64_t cap_rows, int64_t cols, void* dst, int64_t dst_ld, int64_t col_off,
                                int64_t fill_cols, const int32_t* rowmap, const int32_t* rows_dev, void* stream);
/* N = 1 projection over a compact image (the gate layer LinearNorm(n_hidden + n_attn, 1), flowtron.py:668-669 / :760-761, reads the
 * rows the decoder LSTM's input projection has just multiplied): y[rowmap[c]] = sum_k img[c][k] * op16(w[k]) + bias[0] for the compact
 * rows c < *rows_dev; y = [T*B] rows of stride ldy; every padded frame of an utterance receives its separator's value.  The image
 * must be zero from column K up to the next multiple of 8 (ft_bf16_image_rows / _into pad with zeros).
 * _bwd: dw[k] += sum_c dy[rowmap[c]] img[c][k], db[0] += sum_c dy[rowmap[c]] (db may be NULL); dw / db ACCUMULATE: the caller zeroes. */
int ft_img_gemv_rows(const void* img, int64_t ld, int K, const float* w, const float* bias, float* y, int64_t ldy,
                     const int32_t* rowmap, const int32_t* rows_dev, const int32_t* lens, int T, int B, void* stream);
int ft_img_gemv_rows_f16(const void* img, int64_t ld, int K, const float* w, const float* bias, float* y, int64_t ldy,
                         const int32_t* rowmap, const int32_t* rows_dev, const int32_t* lens, int T, int B, void* stream);
int ft_img_gemv_rows_bwd(const void* img, int64_t ld, int K, const float* dy, int64_t lddy, float* dw, float* db,
                         const int32_t* rowmap, const int32_t* rows_dev, int64_t cap_rows, void* stream);
int ft_img_gemv_rows_bwd_f16(const void* img, int64_t ld, int K, const float* dy, int64_t lddy, float* dw, float* db,
                             const int32_t* rowmap, const int32_t* rows_dev, int64_t cap_rows, void* stream);
/* rows (t, b) with t > lens[b] of the time-major matrix y [T*B][cols] (row stride ld): mode 0 = zero them, 1 = copy row
 * (lens[b], b) into them (a compact GEMM wrote only valid rows and the separator; consumers that walk every frame need the rest) */
int ft_pad_rows_fill(float* y, int64_t ld, int cols, const int32_t* lens, int T, int B, int mode, void* stream);

/* ---- embedding gather (flowtron.py:873-874) ------------------------------
 * out[r][0:dim] = W[ids[r]] for r < n (out row stride ld_out).  bwd: dW[ids[r]] += dout[r]. */
int ft_embedding_fwd(const int64_t* ids, const float* W, float* out, int n, int dim, int64_t ld_out, void* stream);
int ft_embedding_bwd(const int64_t* ids, const float* dout, float* dW, int n, int dim, int64_t ld_dout, void* stream);
/* The same sums with the rows walked at a stride (row r = i * stride + j): a thread adds its rows in a register while the id stays
 * the same and sends one atomic per run.  For ids that repeat with period `stride` -- the speaker embedding gathered for every text
 * position of a [L,B] batch (flowtron.py:886-887): stride = B -- that is 1 / 32 of the atomics; correct for ANY ids.  (ABI 12) */
int ft_embedding_bwd_runs(const int64_t* ids, const float* dout, float* dW, int n, int dim, int64_t ld_dout, int stride, void* stream);

/* ---- encoder conv as im2col (flowtron.py:499-502) --------------------------
 * x [L,B,C] time-major (must already be zero at l >= lens[b]);
 * col[l][b][c*KW + k] = x[l + k - KW/2][b][c] (0 outside [0,lens[b])).
 * col2im is the adjoint (gather form, deterministic). */
int ft_im2col(const float* x, float* col, const int32_t* lens, int L, int B, int C, int KW, void* stream);
int ft_col2im(const float* dcol, float* dx, const int32_t* lens, int L, int B, int C, int KW, void* stream);

/* ---- masked instance norm + ReLU (+dropout keep-mask) (flowtron.py:53-92, :502)
 * x,y [L,B,C]; statistics over l < lens[b] (biased variance, eps); y = relu(xhat*gamma+beta)*keep,
 * y = 0 at l >= lens[b].  keep may be NULL.  Saves mean/rstd [B,C] for backward.
 * bwd: dx (0 at pads), dgamma/dbeta [C] (overwritten). */
int ft_instnorm_relu_fwd(const float* x, const float* gamma, const float* beta, const float* keep,
                         const int32_t* lens, float* y, float* mean, float* rstd,
                         int L, int B, int C, float eps, void* stream);
int ft_instnorm_relu_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* keep,
                         const int32_t* lens, const float* mean, const float* rstd,
                         float* dx, float* dgamma, float* dbeta, int L, int B, int C, void* stream);

/* ---- length-masked LSTM over a whole sequence (nn.LSTM on a packed sequence:
 * flowtron.py:689-694 attention_lstm/lstm, :505-512 encoder BiLSTM) -----------------
 * gx [T,B,4H] = x W_ih^T + b_ih + b_hh (from ft_gemm); w_hh [4H,H], gate rows i|f|g|o.
 * Step s of sample b touches time t = s (forward) or lens[b]-1-s (reverse) while s < lens[b].
 * y [T,B,ldy] gets h_t at valid t and 0 at t >= lens[b].  gates [T,B,4H] (post-activation
 * i,f,g,o) and cell [T,B,H] are saved for backward.  work: ft_lstm_workspace_bytes() bytes.
 * bwd: dy [T,B,ldy] -> dgx [T,B,4H] (0 at pads); dW_hh/dW_ih/dx/db follow as ft_gemm /
 * ft_colsum calls over all T*B rows. */
size_t ft_lstm_workspace_bytes(int B, int H);
int ft_lstm_seq_fwd(const float* gx, const float* w_hh, const int32_t* lens,
                    float* y, int64_t ldy, float* gates, float* cell, void* work,
                    int T, int B, int H, int reverse, int mode, void* stream);
int ft_lstm_seq_bwd(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens,
                    const float* gates, const float* cell, float* dgx, void* work,
                    int T, int B, int H, int reverse, int mode, void* stream);

/* Persistent form of ft_lstm_seq_bwd (16-bit MFMA operands, H == 1024, B <= 32, 256-CU device): ONE launch for the whole sequence
 * (csrc/lstm_persist.hip, lstm_persist_bwd_rs_k).  The chip is split into 8 independent batch groups = the 8 XCDs, formed at run
 * time from the workgroups' XCC ids; every CU keeps the 16-bit MFMA fragments of its W_hh rows in registers for all T steps,
 * multiplies its OWN dgates with them and the fp32 partials are REDUCE-SCATTERED through the XCD's own L2, tagged in the mantissa LSB:
 * the same products as ft_lstm_seq_bwd(FT_BF16) in another, fixed association -- equal to fp32 rounding, deterministic.
 * `ng` = 21 (1 is accepted as "the default"); round 6 removed the all-gather transports 1 | 9 | 11 | 19 and the round-2..5 forward
 * kernel ft_lstm_persist_fwd / _fwd_rows / _bwd_rows: the forward recurrence and every batch wider than 32 run on ft_lstm_roles_* below.
 * `status` (device int32, zeroed by the caller once) is raised to 1 if a hand-off wait times out (grid not co-resident);
 * the caller must check it before trusting dgx.  work: ft_lstm_persist_workspace_bytes(), 256-byte aligned. */
int ft_lstm_persist_supported(int B, int H);
size_t ft_lstm_persist_workspace_bytes(int B, int H);
/* debug: device buffer [1024][4][5] int64 that subsequent ft_lstm_persist_bwd launches fill with per-step phase stamps
 * (100 MHz wall clock) of one workgroup; NULL switches it off. */
int ft_lstm_persist_debug_prof(void* dev_buf);
int ft_lstm_persist_bwd(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens, const float* gates,
                        const float* cell, float* dgx, void* work, int32_t* status, int T, int B, int H, int ng, void* stream);
/* The same launch, additionally leaving the 16-bit operand image of dgates in pack-by-length row order in `dimg` -- exactly what
 * ft_bf16_image_rows(dgx, ..., rowmap of `lens`) would produce afterwards (utterance b = its len_b frames + one zero separator
 * row, batch-major; zero rows up to ceil256(R + 32); dimg: ft_bf16_image_bytes(T*B + B, 4H), row stride dimg_ld elements,
 * dimg_rows rows allocated) -- and ADDING the column sums of dgates (the bias gradient of the layer) to dbias[4H] (zeroed by the
 * caller; fp32 atomics, one per workgroup row and column).  The weight- and input-gradient GEMMs (flowtron.py:689-694's
 * autograd transposes) then start without a conversion pass over the 450 MB dgx.  dgx == NULL: the image only. */
int ft_lstm_persist_bwd_img(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens, const float* gates,
                            const float* cell, float* dgx, void* work, int32_t* status, int T, int B, int H, int ng,
                            void* dimg, int64_t dimg_ld, int64_t dimg_rows, float* dbias, void* stream);

/* Round 6 (ABI 13): the persistent recurrences with the geometry as PARAMETERS (csrc/lstm_roles.hip) -- rows per XCD group R = 4 | 8 | 16
 * (an MFMA tile has 16 rows; the kernels above use 4), a time WINDOW [t0, t1) per launch with the recurrent state carried through
 * small fp32 buffers, and up to two ROLES (independent recurrences) per launch, role i on XCD groups i * 8 / n_roles .. (batch rows
 * <= (8 / n_roles) * R each).  What they replace is still cuDNN's nn.LSTM over packed sequences (flowtron.py:654-655, 689-694, 760-765):
 *   - a batch of 64 / 128 as ONE launch per sequence (n_roles 1, R 8 / 16) instead of 2 / 4 sliced launches;
 *   - the two decoder layers of a flow CONCURRENTLY, layer 1 one time chunk behind layer 0 (backward: the other way round), the chunk's
 *     input projection GEMM between the launches (flowtron_amd/ops.py DecoderPairFn).
 * wimg: MFMA fragment image of W_hh (ft_lstm_roles_wimg_bytes, 256-byte aligned), made once per pass by ft_lstm_roles_prepare_fwd / _bwd
 * (the backward image is the reduce-scatter layout).  ctx: launch context of ft_lstm_roles_ctx_bytes() bytes, 256-byte aligned,
 * initialised ONCE by ft_lstm_roles_ctx_init and then owned by the launches: `phase` = number of earlier launches of the same kind
 * (forward / backward) on this ctx -- a launch works in hand-off set phase & 1 and presets set (phase + 1) & 1 for `reset_rows` (>= the R
 * of the next launch of its kind) while it runs, so no preset dispatch precedes a launch.  Launches on one ctx must be serialised on
 * one stream.  `status` as for ft_lstm_persist_*.  Forward results are bit-identical to ft_lstm_seq_fwd for every R / windowing / role
 * placement; backward to fp32 rounding (R = 4: bit-identical to ft_lstm_persist_bwd's reduce-scatter transport). */
typedef struct {
    const void* gx; const int32_t* lens; float* y; int64_t ldy; float* gates; float* cell;   /* as ft_lstm_seq_fwd; time steps ldb rows apart */
    const void* wimg;
    float* state_h; float* state_c;      /* [B][H] fp32: read when t0 > 0, written at the end of the window; NULL (both) for t0 == 0 without successor */
    int32_t B, ldb, t0, t1;
    int32_t gx16;                        /* 0: gx = fp32 rows [T][ldb][4H]; 1: 16-bit rows of the call's operand format (what ft_gemm_img writes with
                                          * FT_GEMM_C16: half the bytes; widened exactly before it is added to the fp32 recurrent sums) -- one format per launch */
} ft_lstm_fwd_role;
typedef struct {
    const float* dy; int64_t ldy; const int32_t* lens; const float* gates; const float* cell;
    float* dgx;                          /* fp32 dgates rows [T][ldb][4H], or NULL when only the image is wanted */
    const void* wimg;
    void* dimg; int64_t dimg_ld; int64_t dimg_rows; float* dbias;    /* optional: as ft_lstm_persist_bwd_img (needs ldb == B) */
    float* state_da; float* state_dc;    /* [B][4H], [B][H] fp32: read when carry_in, written at the end of the window */
    int32_t B, ldb, t0, t1, carry_in;    /* carry_in: the window [t1, ..) has run before and left its state */
} ft_lstm_bwd_role;
size_t ft_lstm_roles_ctx_bytes(void);
size_t ft_lstm_roles_wimg_bytes(int H);
int ft_lstm_roles_ctx_init(void* ctx, void* stream);
int ft_lstm_roles_debug_prof(void* dev_buf);        /* as ft_lstm_persist_debug_prof, for the kernels below */
int ft_lstm_roles_prepare_fwd(const float* w_hh, void* wimg, int H, void* stream);
int ft_lstm_roles_prepare_bwd(const float* w_hh, void* wimg, int H, void* stream);
int ft_lstm_roles_fwd(const ft_lstm_fwd_role* roles, int n_roles, int rows_per_group, int reset_rows, void* ctx, int phase,
                      int32_t* status, int H, void* stream);
int ft_lstm_roles_bwd(const ft_lstm_bwd_role* roles, int n_roles, int rows_per_group, int reset_rows, void* ctx, int phase,
                      int32_t* status, int H, void* stream);

/* Two stacked layers (the decoder nn.LSTM(.., num_layers=2), flowtron.py:654, :760-765) as ONE launch chain: layer 1 at
 * time t-1 and layer 0 at time t are two workgroup groups of the same launch, and layer 1's input projection
 * (h0 W_ih1^T) rides along as a second bf16 fragment stream, so T+1 launches replace 2T + a batched GEMM.
 * gx0 [T,B,4H] = x W_ih0^T + b0 (from ft_gemm), bias1 [4H] = b_ih1 + b_hh1; forward direction, bf16 MFMA operands.
 * Saves y/gates/cell of BOTH layers; backward returns dgx0 and dgx1 [T,B,4H] (weight/bias gradients follow as GEMMs /
 * column sums over all T*B rows: dW_hh0 = dgx0[1:]^T y0[:-1], dW_ih1 = dgx1^T y0, dW_hh1 = dgx1[1:]^T y1[:-1]).
 * Only for ft_lstm2_supported(B,H) (H % 128 == 0, B <= 64); work: ft_lstm2_workspace_bytes(B,H), 256-byte aligned. */
int ft_lstm2_supported(int B, int H);
size_t ft_lstm2_workspace_bytes(int B, int H);
int ft_lstm2_seq_fwd(const float* gx0, const float* w_hh0, const float* w_ih1, const float* bias1, const float* w_hh1,
                     const int32_t* lens, float* y0, float* gates0, float* cell0, float* y1, float* gates1, float* cell1,
                     void* work, int T, int B, int H, void* stream);
int ft_lstm2_seq_bwd(const float* dy1, const float* w_hh0, const float* w_ih1, const float* w_hh1, const int32_t* lens,
                     const float* gates0, const float* cell0, const float* gates1, const float* cell1,
                     float* dgx0, float* dgx1, void* work, int T, int B, int H, void* stream);

/* Both directions of a bidirectional layer (the encoder nn.LSTM(.., bidirectional=True), flowtron.py:488, :505-512) as ONE
 * launch chain: the forward-in-time and the reverse recurrence are independent and each step is latency-bound, so they run
 * as the two z-slices of the same launch (T launches instead of 2T).  bf16 MFMA operands; ft_lstm_bidir_supported(B,H):
 * H % 128 == 0, B <= 64.  y [T,B,ldy] with ldy >= 2H receives [h_fwd | h_rev]; dy has the same layout.  gx_*, gates_*,
 * cell_*, dgx_* as in ft_lstm_seq_*; work_f / work_r: ft_lstm_workspace_bytes(B,H) each, 256-byte aligned. */
int ft_lstm_bidir_supported(int B, int H);
int ft_lstm_bidir_seq_fwd(const float* gx_f, const float* gx_r, const float* w_hh_f, const float* w_hh_r,
                          const int32_t* lens, float* y, int64_t ldy, float* gates_f, float* gates_r,
                          float* cell_f, float* cell_r, void* work_f, void* work_r, int T, int B, int H, void* stream);
int ft_lstm_bidir_seq_bwd(const float* dy, int64_t ldy, const float* w_hh_f, const float* w_hh_r, const int32_t* lens,
                          const float* gates_f, const float* gates_r, const float* cell_f, const float* cell_r,
                          float* dgx_f, float* dgx_r, void* work_f, void* work_r, int T, int B, int H, void* stream);

/* Persistent form of ft_lstm_bidir_seq_* for the text encoder's shape (H == 256, B <= 32, 256-CU device, 16-bit operands): ONE launch
 * per pass for both directions and all time steps (csrc/bilstm_persist.hip: every wave an independent agent, no workgroup-level
 * step, XCD-local hand-off).  Same tensors as ft_lstm_bidir_seq_*; one workspace (ft_bilstm_persist_workspace_bytes, 256-byte
 * aligned); `status` as for ft_lstm_persist_*.  Same operand rounding as the launch-per-step chain, different fp32 summation order
 * (K is not split over waves): results agree to rounding, not bit for bit. */
int ft_bilstm_persist_supported(int B, int H);
size_t ft_bilstm_persist_workspace_bytes(int B, int H);
int ft_bilstm_persist_fwd(const float* gx_f, const float* gx_r, const float* w_hh_f, const float* w_hh_r, const int32_t* lens,
                          float* y, int64_t ldy, float* gates_f, float* gates_r, float* cell_f, float* cell_r,
                          void* work, int32_t* status, int T, int B, int H, void* stream);
int ft_bilstm_persist_bwd(const float* dy, int64_t ldy, const float* w_hh_f, const float* w_hh_r, const int32_t* lens,
                          const float* gates_f, const float* gates_r, const float* cell_f, const float* cell_r,
                          float* dgx_f, float* dgx_r, void* work, int32_t* status, int T, int B, int H, void* stream);

/* ---- additive attention scores + softmax + prior posterior (flowtron.py:544-583)
 * Q [T,B,A] (time-major), K [L,B,A], v [A], in_lens [B], prior [B,T,L] or NULL.
 * e[b,t,l] = sum_a v[a] tanh(Q[t,b,a]+K[l,b,a]) / temperature, -inf at l >= in_lens[b];
 * p = softmax_l(e); with prior: logprob = log(p+1e-20)+log(prior+1e-20), attn = softmax_l(masked logprob);
 * without: attn = p, logprob = log(p+1e-8).  The B*T*L*A tanh tensor is never materialised.
 * Outputs attn, logprob [B,T,L]; p [B,T,L] is saved for backward when prior != NULL (may alias attn otherwise).
 * bwd: dattn, dlogprob (may be NULL) -> dQ [T,B,A], dK [L,B,A] (atomically accumulated: zero it first),
 * dv [A] (accumulated: zero it first). tanh is recomputed. */
int ft_attention_fwd(const float* Q, const float* K, const float* v, const int32_t* in_lens, const float* prior,
                     float* attn, float* logprob, float* p_save,
                     int T, int B, int L, int A, float temperature, void* stream);
int ft_attention_bwd(const float* Q, const float* K, const float* v, const int32_t* in_lens, const float* prior,
                     const float* attn, const float* p_save, const float* dattn, const float* dlogprob,
                     float* de_work, float* dQ, float* dK, float* dv,
                     int T, int B, int L, int A, float temperature, void* stream);

/* ---- affine coupling (flowtron.py:770-772) --------------------------------
 * out [T,B,2M] = [log_s | b];  z = exp(log_s)*x + b.
 * bwd: dout = [dz*x*exp(log_s) + dlog_s_ext | dz], dx = dz*exp(log_s)  (dlog_s_ext may be NULL). */
int ft_affine_fwd(const float* out, const float* x, float* z, int64_t n_rows, int M, void* stream);
int ft_affine_bwd(const float* out, const float* x, const float* dz, const float* dlog_s_ext,
                  float* dout, float* dx, int64_t n_rows, int M, void* stream);
/* inverse used by inference (flowtron.py:821): x = (z - b) / exp(log_s) */
int ft_affine_inv(const float* out, const float* z, float* x, int64_t n_rows, int M, void* stream);

/* ---- masked NLL partial sums (flowtron.py:206-235) -------------------------
 * acc[0] += sum_{t<len_b} z^2 ; acc[1] += sum_{t<len_b} log_s  (acc zeroed by caller).
 * x [T,B,M] with row stride ld (log_s is a strided view of the coupling output). */
int ft_masked_sum(const float* x, int64_t ld, const int32_t* lens, float* acc, int square,
                  int T, int B, int M, void* stream);
/* dx = scale_dev[0] * coef * (square ? x : 1) at valid positions, 0 at pads */
int ft_masked_sum_bwd(const float* x, int64_t ld, const int32_t* lens, const float* scale_dev, float coef, int square,
                      float* dx, int64_t ld_dx, int T, int B, int M, void* stream);

/* ---- gate BCE-with-logits (flowtron.py:237-243) ----------------------------
 * gate [T,B], target [B,T]; acc[0] += sum_valid BCE(gate, target).  bwd: dgate = scale*(sigmoid(g)-y) valid, 0 pads */
int ft_gate_bce_fwd(const float* gate, const float* target, const int32_t* lens, float* acc, int T, int B, void* stream);
int ft_gate_bce_bwd(const float* gate, const float* target, const int32_t* lens, const float* scale_dev, float coef,
                    float* dgate, int T, int B, void* stream);

/* ---- FlowtronLoss.forward's NLL + gate terms in four launches, their backward in two (flowtron.py:200-243; ABI 12) ------------
 * The per-term entry points above leave the scalar arithmetic (frame count, normalisers, g / n) to the caller -- ~40 four-byte
 * kernels per training step right where train.py:300-303 waits for the losses.  fwd: acc (8 floats, zeroed here) receives
 * [0] sum_valid z^2, [1] sum_f sum_valid log_s[f], [2] sum_valid BCE, and from a one-workgroup kernel [4] 1 / (n M), [5] 1 / n,
 * [6] n = sum(out_lens); nll_out[0] = (acc[0] / (2 sigma^2) - acc[1]) / (n M), gate_out[0] = acc[2] / n (gate, gate_target,
 * gate_out all NULL: no gate term).  z [T,B,M] contiguous; log_s = HOST array of n_ls (<= 8) device pointers to [T,B,M] views
 * with row stride ld_ls; gate [T,B], gate_target [B,T].
 * bwd (reads acc as fwd left it; g_nll / g_gate = device scalars, the gradients of the two outputs): dz = g_nll z / (sigma^2 n M),
 * dls = -g_nll / (n M) (ONE tensor: the gradient of every flow's log_s; may be NULL), dgate = g_gate (sigmoid(gate) - target) / n,
 * all zero at padded frames.  (g_nll, dz) and (g_gate, dgate) are each both NULL or both set. */
int ft_flowtron_loss_fwd(const float* z, const float* const* log_s, int n_ls, int64_t ld_ls, const float* gate,
                         const float* gate_target, const int32_t* out_lens, float sigma, float* acc, float* nll_out,
                         float* gate_out, int T, int B, int M, void* stream);
int ft_flowtron_loss_bwd(const float* z, const float* gate, const float* gate_target, const int32_t* out_lens, float sigma,
                         const float* acc, const float* g_nll, const float* g_gate, float* dz, float* dls, float* dgate,
                         int T, int B, int M, void* stream);

/* ---- reverse-by-length (flowtron.py:606-622, the flip+roll involution) --------
 * time_major=1: x,y [T,B,C]; time_major=0: x,y [B,T,C].
 * y[t] = x[len-1-t] (t < len), x[T-1+len-t] (t >= len). */
int ft_reverse_by_length(const float* x, float* y, const int32_t* lens, int T, int B, int C, int time_major, void* stream);

/* ---- activation backward (autograd of torch.tanh after nn.Linear, flowtron.py:461-464):
 * dpre = dy * act'(pre), written through the saved output y = act(pre).  dpre may alias dy. */
int ft_act_bwd(const float* y, const float* dy, float* dpre, int64_t n, int act, void* stream);

/* ---- elementwise out = a + b (op 0) or a * b (op 1): key modulation text*cond (flowtron.py:712) and the running
 * attention sum (:719) of the cumulative-attention branch.  out may alias a or b. */
int ft_eltwise(const float* a, const float* b, float* out, int64_t n, int op, void* stream);

/* ---- column sums (bias gradients): out[n] = sum_r x[r*ld + n] -------------- */
int ft_colsum(const float* x, float* out, int64_t rows, int N, int64_t ld, void* stream);

/* ---- autoregressive decode, batch 1 (flowtron.py:775-828) ------------------
 * One flow, N frames, fully enqueued without host synchronisation.  See ft_decode_args.
 * n_done_dev[0] receives the number of frames produced (gate stop, flowtron.py:823-826). */
typedef struct {
    /* attention_lstm */  const float *att_w_ih, *att_w_hh, *att_b_ih, *att_b_hh;
    /* attention     */   const float *w_query, *v, *K, *V;   /* K,V [L,A] precomputed once per utterance */
    /* lstm l0, l1   */   const float *l0_w_ih, *l0_w_hh, *l0_b_ih, *l0_b_hh, *l1_w_ih, *l1_w_hh, *l1_b_ih, *l1_b_hh;
    /* dense, conv   */   const float *d0_w, *d0_b, *d1_w, *d1_b, *conv_w, *conv_b;
    /* gate (or NULL)*/   const float *gate_w, *gate_b;
    const float* residual;   /* [N,M] */
    float* mel_out;          /* [N,M] */
    float* attn_out;         /* [N,L] */
    int32_t* n_done_dev;     /* [1] */
    void* work; size_t work_bytes;
    int N, L, H, A, M;
    float temperature, gate_threshold;
    int use_graph;
    /* cumulative / location-sensitive attention (flowtron.py:129-152, :793-806); all NULL when use_cumm_attention is off:
     * Conv1d(2->32,k5)+ReLU, Conv1d(32->E,k3)+Sigmoid over [cumulative attn ; previous attn]; the result scales the
     * encoder outputs enc [L,E] before the key projection w_key [A,E], every frame.  K is then ignored. */
    const float *cond_w1, *cond_b1, *cond_w2, *cond_b2, *w_key, *enc;
    int E;
    /* optional [N,L] rows: attention prior (attn = softmax(log(p+1e-20)+log(prior+1e-20)), flowtron.py:544-557, :799) and
     * forced alignment (the given row replaces the computed attention, :585-588, :798); NULL = off */
    const float *prior, *forced;
    /* optional scratch of ft_decode_wimg_bytes() bytes (256-byte aligned): when given, the ten weight matrices are rounded once
     * per call to bf16 images and the per-frame GEMVs stream those (bf16 operand mode: half the bytes per frame; fp32
     * activations and accumulation).  NULL = stream the fp32 weights (parity mode). */
    void* wimg; size_t wimg_bytes;
    /* optional: hand-off buffer of ft_decode_persist_gran_bytes() bytes + a device status word (zeroed by the caller once).  When
     * given together with wimg, and the flow has the default geometry (H 1024, A 640, M 80, L <= 1024, no cumulative attention /
     * prior / forced alignment) on a 256-CU device, the WHOLE flow runs as one persistent launch (256 workgroups hand each
     * stage's output vector to one another through tag-checked granules) instead of ten launches per frame.  *persist_status
     * becomes non-zero if a hand-off wait times out; the caller must check it before trusting the output. */
    void* persist_gran; int32_t* persist_status;
    /* decoder LSTM depth (ABI 11; nn.LSTM(n_hidden + n_attn, n_hidden, n_lstm_layers), flowtron.py:654-655): 0 or 2 = the two layers
     * above; 1 = layer 0 only (l1_* NULL); n > 2 = layers 2 .. n-1 from `extra_layers`, a HOST array of 4 (n - 2) device pointers
     * {w_ih [4H,H], w_hh [4H,H], b_ih, b_hh} per layer.  Depths other than 2 run on the staged chain (one launch per stage and
     * frame, hipGraph of 8 frames); layers beyond the second stream their fp32 weights in every operand mode. */
    int n_layers; const float* const* extra_layers;
} ft_decode_args;
size_t ft_decode_workspace_bytes(int L, int H, int A, int M, int E, int n_layers);
size_t ft_decode_wimg_bytes(int H, int A, int M);
size_t ft_decode_persist_gran_bytes(void);
int ft_decode_debug_prof(void* dev_buf);   /* debug: [512][12] int64 stage stamps of the persistent decode, NULL = off */
int ft_decode_flow(const ft_decode_args* a, void* stream);

/* ---- STFT magnitude + mel + log (audio_processing.py:117-134, 207-235) -------
 * y [B,N] in [-1,1] -> mel [B,n_mel,N/hop+1]; window [n_fft] (periodic hann),
 * fb [n_mel, n_fft/2+1].  Radix-2 real FFT per frame in LDS (n_fft = 1024). */
int ft_stft_mel(const float* y, const float* window, const float* fb, float* mel,
                int B, int N, int n_fft, int hop, int n_mel, void* stream);
/* The same front end for the reference's analysis setting n_fft = 1024, hop <= 256 (config.json:32-34) as a REAL FFT (one
 * 512-point complex FFT + split step, one wave per frame, radix-8 passes in registers) with the triangular filterbank in
 * sparse form: band b = weights band_w[band_ptr[b] .. band_ptr[b+1]) over the consecutive bins starting at band_bin0[b].
 * Any of mel [B,n_mel,T], (mag, phase) [B,513,T] may be requested (NULL = skip; mag and phase together = STFT.transform,
 * audio_processing.py:207-235).  window: hann [1024] (win_length zero-padded by the caller). */
int ft_stft_r8(const float* y, const float* window, const int32_t* band_bin0, const int32_t* band_ptr, const float* band_w,
               float* mel, float* mag, float* phase, int B, int N, int hop, int n_mel, void* stream);
/* The collated batch of the data path (data.py:149-155 per item, :207-229 zero padding) in ONE launch:
 * y [B,N] zero-padded audio, utterance b holds n_samples[b] samples (device int32) -> mel [B,n_mel,T_out]: the frames
 * t < n_samples[b] / hop + 1 exactly as ft_stft_r8 computes them for that utterance alone (reflect padding about ITS last
 * sample), zeros beyond -- the tensor DataCollate builds on the host.  T_out >= max_b (n_samples[b] / hop + 1). */
int ft_stft_r8_ragged(const float* y, const int32_t* n_samples, const float* window, const int32_t* band_bin0,
                      const int32_t* band_ptr, const float* band_w, float* mel, int B, int N, int hop, int n_mel, int T_out,
                      void* stream);

/* ---- attention-CTC loss (flowtron.py:155-182, 245-274) ------------------------------
 * lp [B,T,L] = attn_logprob in natural time order.  Per sample: classes {blank (logit blank_logprob), 1..K_b} with
 * K_b = in_lens[b], frames t < out_lens[b]; log_softmax over the classes, CTC against the target 1..K_b (blank 0),
 * reduction 'mean' (divide by K_b), zero_infinity, then the batch mean.  loss[0] receives the scalar.  work holds
 * alpha and beta [B,T,2L+1], the log-softmax normalisers [B,T] and nll [B] (ft_attn_ctc_workspace_floats floats).
 * with_beta != 0: the beta recursion runs beside alpha in the same launch (a second workgroup per sample; both are
 * latency-bound in T), so the backward pass is a single elementwise kernel -- pass beta_ready = with_beta to bwd.
 * bwd: dlp [B,T,L] = gout_dev[0] * d loss / d lp (overwritten; zero outside the valid [T_b, K_b] window). */
size_t ft_attn_ctc_workspace_floats(int B, int T, int L);
int ft_attn_ctc_fwd(const float* lp, const int32_t* in_lens, const int32_t* out_lens, float blank_logprob,
                    float* work, float* loss, int B, int T, int L, int with_beta, void* stream);
int ft_attn_ctc_bwd(const float* lp, const int32_t* in_lens, const int32_t* out_lens, float blank_logprob,
                    float* work, const float* gout_dev, float* dlp, int B, int T, int L, int beta_ready, void* stream);
/* The same loss over F flows at once (FlowtronLoss, flowtron.py:245-274: the per-flow losses averaged): F * B stacked samples,
 * sample f * B + b = flow f's log-probabilities of utterance b; loss[0] = mean over flows of the per-flow batch means.  lp / dlp:
 * HOST arrays of F (<= 8) device pointers to [B,T,L] tensors; reversed[f] != 0 (host array): flow f's tensor is in REVERSED time
 * (an AR_Back_Step: frame t of utterance b is row out_lens[b] - 1 - t) -- the reference flips + rolls the tensor there and back
 * (:250-271), here the row index is mirrored in the kernels, for the loss and for dlp[f] (which comes out in the flow's own order).
 * No concatenated or re-reversed copy exists.  work: ft_attn_ctc_workspace_floats(F * B, T, L) floats.  (ABI 12) */
int ft_attn_ctc_fwd_multi(const float* const* lp, const int32_t* reversed, int F, const int32_t* in_lens, const int32_t* out_lens,
                          float blank_logprob, float* work, float* loss, int B, int T, int L, int with_beta, void* stream);
int ft_attn_ctc_bwd_multi(const float* const* lp, const int32_t* reversed, int F, const int32_t* in_lens, const int32_t* out_lens,
                          float blank_logprob, float* work, const float* gout_dev, float* const* dlp, int B, int T, int L,
                          int beta_ready, void* stream);

/* ---- cumulative ("location-sensitive") attention of a teacher-forced flow, one call per sequence ----------
 * flowtron.py:697-723 (run_cumm_attn_sequence), :129-152 (AttentionConditioningLayer), :544-592 (Attention.forward).  Per frame
 * i: cond = sigmoid(conv_K2(relu(conv_K1([cumm_i ; prev_i])))) over the text axis (2 -> NF -> E channels, zero padding at the ends),
 * keys = (text . cond) w_key^T, scores v . tanh(Q_i + keys) / temperature, softmax over l < in_lens[b], logprob = log(attn + 1e-8),
 * ctx_i = attn_i V, cumm_{i+1} = cumm_i + attn_i, prev_{i+1} = attn_i.  The LIBRARY walks the T dependent frames, enqueued back to
 * back on `stream`, no host synchronisation:
 *   - 16-bit operand modes at the config.json width (E = A = 640, NF 32, K1 5, K2 3; ft_cumm_attn_fused() != 0, ABI 11): ONE fused
 *     launch per frame and direction (csrc/cumm_fused.hip); context, dV, dctx . V and every weight gradient leave the frame loop
 *     and run as a few large GEMMs over all frames (the backward keeps 16-bit streams of dK / km / dpre2 / col2 for a chunk of
 *     frames in `work`).  kproj_all then holds tanh(Q_i + K_i) of the rows l < in_lens[b] instead of K_i.
 *   - otherwise (fp32 parity mode, other widths, FT_CUMM_FUSED=0): a chain of 8 launches per frame forward, 22 backward.
 * Layouts: text [L][B][E] (encoder outputs), Q [T][B][A] and V [L][B][A] (already projected), w_key [A][E], v [A], w1 [NF][2][K1],
 * w2 [E][NF][K2]; outputs ctx [T][B][A], attn / logprob [B][T][L].  Saved for backward (caller-owned, filled by fwd):
 * cumm_all [T][B][L] and kproj_all [T][L*B][A].  work: ft_cumm_attn_workspace_bytes() bytes, 256-byte aligned.
 * bwd: dctx [T][B][A]; dattn / dlogprob [B][T][L] or NULL; every gradient output is overwritten (accumulated over the frames
 * inside): dQ [T][B][A], dV [L][B][A], dtext [L][B][E], dw_key, dv, dw1, db1, dw2, db2. */
typedef struct {
    const float *text, *Q, *V, *w_key, *v, *w1, *b1, *w2, *b2;
    const int32_t* in_lens;
    float *ctx, *attn, *logprob, *cumm_all, *kproj_all;
    void* work; size_t work_bytes;
    int T, B, L, E, A, NF, K1, K2;
    float temperature;
    int mode;
    /* optional (ABI 11): a device status word (zeroed by the caller once; the one the persistent recurrences use will do).  When given,
     * and the fused path's work list is co-resident by construction (B x ceil(L / 32) <= CUs), the forward frames run as ONE
     * persistent launch (per-utterance granule hand-offs instead of a launch boundary per frame); *persist_status becomes non-zero if
     * a hand-off wait gives up (grid not co-resident: a foreign kernel holds CUs) -- the caller must then not trust the outputs. */
    int32_t* persist_status;
} ft_cumm_attn_args;
size_t ft_cumm_attn_workspace_bytes(int T, int L, int B, int E, int A, int NF, int K1, int K2, int mode, int backward);
int ft_cumm_attn_fused(const ft_cumm_attn_args* a);     /* 1: fwd / bwd of these arguments take the fused one-launch-per-frame path */
/* debug: device buffer [2][4096][16] int64 that later fused launches fill with stage stamps of workgroup (0, 0) (100 MHz clock;
 * [0] = forward, [1] = backward, second index = frame); NULL switches it off (scripts/exp/cumm_prof.py) */
int ft_cumm_attn_debug_prof(void* dev_buf);
int ft_cumm_attn_fwd(const ft_cumm_attn_args* a, void* stream);
int ft_cumm_attn_bwd(const ft_cumm_attn_args* a, const float* dctx, const float* dattn, const float* dlogprob,
                     float* dQ, float* dV, float* dtext, float* dw_key, float* dv, float* dw1, float* db1, float* dw2, float* db2,
                     void* stream);

/* ---- beta-binomial attention prior (data.py:31-41, 111-141) --------
 * prior[b,t,k] = BetaBinom.pmf(k; n = in_lens[b]-1, a = scaling*(t+1), b = scaling*(out_lens[b]-t)) for
 * t < out_lens[b], k < in_lens[b]; 0 in the padding.  [B,T,L] fp32, float64 lgamma inside. */
int ft_beta_binomial_prior(const int32_t* in_lens, const int32_t* out_lens, float* prior,
                           int B, int T, int L, float scaling, void* stream);

/* ---- fused RAdam over a flat arena (radam.py:44-122) + grad-norm clip ----------
 * One pass: g *= min(1, clip / (sqrt(*gnorm_sq_dev) + 1e-6)) (torch clip_grad_norm_, train.py:328; skipped when clip == 0
 * or gnorm_sq_dev == NULL), v = beta2 v + (1-beta2) g g, m = beta1 m + (1-beta1) g, p -= weight_decay*lr*p,
 * p -= step_size * (rectified ? m / (sqrt(v) + eps) : m).  step_size is radam.py:95-105 (contains lr).  Hyper-parameters
 * are doubles like the python optimizer's; derived coefficients are rounded to fp32 once.
 * Guard (device side, no host synchronisation): when gnorm_sq_dev is given and *gnorm_sq_dev is NaN or Inf the whole update
 * is SKIPPED -- p, m, v keep their values -- and *skipped_dev (optional) is incremented: what GradScaler.step does for an fp16
 * overflow (train.py:330), extended to every non-finite global norm, so that one poisoned step (ft_poison_if_nonzero below)
 * can never reach the weights or the moments.
 * ft_sumsq: acc[0] += sum x^2, DETERMINISTIC (no float atomics: per-workgroup partial sums into `partials`, FT_SUMSQ_PARTIALS
 * floats of caller scratch, added in index order by one workgroup) -- data-parallel replicas stay bit-identical only if every
 * rank derives the same clip factor from the same reduced gradients. */
#define FT_SUMSQ_PARTIALS 1024
int ft_sumsq(const float* x, float* acc, int64_t n, float* partials, void* stream);
int ft_radam_step(float* p, const float* g, float* m, float* v, int64_t n,
                  const float* gnorm_sq_dev, double clip, double lr, double beta1, double beta2, double eps,
                  double weight_decay, double step_size, int rectified, int32_t* skipped_dev, void* stream);
/* The same update with the step count formed on the device (ABI 12): step = calls - *skipped_dev, where calls = the number of
 * optimizer steps issued so far (this one included) and *skipped_dev the updates the guard has dropped so far; step_size and the
 * rectification switch (radam.py:82-106) are evaluated from it in double inside the kernel.  The schedule then follows the APPLIED
 * updates exactly -- radam.py's state['step'] under GradScaler, which does not call step() after an overflow (train.py:330) --
 * without the host reading the drop decision: an optimizer that sets `_step_supports_amp_scaling` lets torch's GradScaler.step
 * hand over `found_inf` as a device tensor instead of synchronising on it (one host sync per fp16 step gone). */
int ft_radam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* gnorm_sq_dev, double clip,
                      double lr, double beta1, double beta2, double eps, double weight_decay, int calls,
                      int32_t* skipped_dev, void* stream);
/* if (*status_dev != 0) dst[0] = NaN.  Enqueued behind the last persistent recurrence launch of a backward pass on the first
 * element of a gradient bucket BEFORE its all-reduce / the norm reduction: a recurrence that reported a time-out (status word of
 * ft_lstm_persist_*) poisons the global gradient norm on EVERY rank, and the guard of ft_radam_step drops that step everywhere. */
int ft_poison_if_nonzero(const int32_t* status_dev, float* dst, void* stream);

/* ---- fp16-operand twins (FT_F16): same signatures, semantics and workspace queries as the entries they are named after;
 * 16-bit images / fragments made by a twin must only be fed to twins. ---- */
int ft_gemm_f16(const ft_gemm_args* a, void* stream);
int ft_bf16_image_f16(const float* src, int64_t ld, int64_t rows, int64_t cols, void* dst, void* stream);
int ft_bf16_image_colsum_f16(const float* src, int64_t ld, int64_t rows, int64_t cols, void* dst, float* colsum, void* stream);
int ft_gemm_img_f16(const ft_gemm_img_args* a, void* stream);
int ft_bf16_image_rows_f16(const float* src, int64_t ld, int64_t cap_rows, int64_t cols, void* dst, float* colsum,
                           const int32_t* rowmap, const int32_t* rows_dev, void* stream);
int ft_lstm_seq_fwd_f16(const float* gx, const float* w_hh, const int32_t* lens,
                    float* y, int64_t ldy, float* gates, float* cell, void* work,
                    int T, int B, int H, int reverse, int mode, void* stream);
int ft_lstm_seq_bwd_f16(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens,
                    const float* gates, const float* cell, float* dgx, void* work,
                    int T, int B, int H, int reverse, int mode, void* stream);
int ft_lstm_persist_bwd_f16(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens, const float* gates,
                        const float* cell, float* dgx, void* work, int32_t* status, int T, int B, int H, int ng, void* stream);
int ft_lstm_persist_bwd_img_f16(const float* dy, int64_t ldy, const float* w_hh, const int32_t* lens, const float* gates,
                            const float* cell, float* dgx, void* work, int32_t* status, int T, int B, int H, int ng,
                            void* dimg, int64_t dimg_ld, int64_t dimg_rows, float* dbias, void* stream);
int ft_lstm_roles_prepare_fwd_f16(const float* w_hh, void* wimg, int H, void* stream);
int ft_lstm_roles_prepare_bwd_f16(const float* w_hh, void* wimg, int H, void* stream);
int ft_lstm_roles_fwd_f16(const ft_lstm_fwd_role* roles, int n_roles, int rows_per_group, int reset_rows, void* ctx, int phase,
                          int32_t* status, int H, void* stream);
int ft_lstm_roles_bwd_f16(const ft_lstm_bwd_role* roles, int n_roles, int rows_per_group, int reset_rows, void* ctx, int phase,
                          int32_t* status, int H, void* stream);
int ft_lstm2_seq_fwd_f16(const float* gx0, const float* w_hh0, const float* w_ih1, const float* bias1, const float* w_hh1,
                     const int32_t* lens, float* y0, float* gates0, float* cell0, float* y1, float* gates1, float* cell1,
                     void* work, int T, int B, int H, void* stream);
int ft_lstm2_seq_bwd_f16(const float* dy1, const float* w_hh0, const float* w_ih1, const float* w_hh1, const int32_t* lens,
                     const float* gates0, const float* cell0, const float* gates1, const float* cell1,
                     float* dgx0, float* dgx1, void* work, int T, int B, int H, void* stream);
int ft_lstm_bidir_seq_fwd_f16(const float* gx_f, const float* gx_r, const float* w_hh_f, const float* w_hh_r,
                          const int32_t* lens, float* y, int64_t ldy, float* gates_f, float* gates_r,
                          float* cell_f, float* cell_r, void* work_f, void* work_r, int T, int B, int H, void* stream);
int ft_lstm_bidir_seq_bwd_f16(const float* dy, int64_t ldy, const float* w_hh_f, const float* w_hh_r, const int32_t* lens,
                          const float* gates_f, const float* gates_r, const float* cell_f, const float* cell_r,
                          float* dgx_f, float* dgx_r, void* work_f, void* work_r, int T, int B, int H, void* stream);
int ft_bilstm_persist_fwd_f16(const float* gx_f, const float* gx_r, const float* w_hh_f, const float* w_hh_r, const int32_t* lens,
                          float* y, int64_t ldy, float* gates_f, float* gates_r, float* cell_f, float* cell_r,
                          void* work, int32_t* status, int T, int B, int H, void* stream);
int ft_bilstm_persist_bwd_f16(const float* dy, int64_t ldy, const float* w_hh_f, const float* w_hh_r, const int32_t* lens,
                          const float* gates_f, const float* gates_r, const float* cell_f, const float* cell_r,
                          float* dgx_f, float* dgx_r, void* work, int32_t* status, int T, int B, int H, void* stream);

#ifdef __cplusplus
}
#endif
#endif
