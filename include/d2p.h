/*
 * d2p.h -- C ABI of libd2p_hip.so: the MI355X (gfx950) kernels under the
 * demo2program full-model training step.
 *
 * The reference (shaohua0116/demo2program) has no FFI / plugin interface: its hot
 * path sits behind Python classes (models/model_full.py, trainer.py) and the
 * arithmetic is whatever TensorFlow 1.3 dispatches.  Each entry point below
 * therefore cites the reference call site (file:line under /root/reference) whose
 * TF-1.3 op it replaces.  The Python surface that mirrors the reference classes is
 * demo2program_amd/{models/model_full.py,trainer.py}; it reaches these functions
 * through ctypes (demo2program_amd/lib.py).  INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers unless noted;
 *    fp32 / int32, contiguous unless a stride argument says otherwise;
 *  - the caller owns every buffer, including scratch (`ws`); the library allocates
 *    nothing.  Compute entry points keep no state between calls; the only mutable
 *    globals are (a) the thread-local error string, (b) the process-global TUNING /
 *    DEBUGGING switches marked as such below (d2p_gemm_set_option, d2p_conv_set_direct,
 *    d2p_lstm_set_fused / _set_persistent / _set_tiling / _debug_flags /
 *    _persist_set_trace / _persist_set_wgs_per_cu, d2p_prof_*): they select between
 *    implementations that return the same results (or, for the debug ones, instrument
 *    a launch), are not thread-safe against concurrent compute calls, and are meant to
 *    be set once at start-up or by benchmarking tools; and (c) one device-side status
 *    word of the persistent LSTM kernels (d2p_lstm_persist_error);
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*), performs
 *    no device synchronisation, and is hipGraph-capturable;
 *  - return 0 on success, a negative D2P_E* code for argument errors, or a positive
 *    hipError_t; never throws / aborts.  d2p_last_error() describes the failure.
 *  - activations are row-major [rows, features]; images NHWC; conv weights
 *    [3,3,Cin,Cout] (TF HWIO); LSTM kernel [(I+U), 4U] with gate order i, j, f, o.
 */
#ifndef D2P_H
#define D2P_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D2P_OK 0
#define D2P_EINVAL (-1)  /* bad argument (null pointer, negative size, bad enum) */
#define D2P_EWS (-2)     /* workspace too small */
#define D2P_EALIGN (-3)  /* pointer / stride alignment requirement violated */

typedef void* d2p_stream_t; /* hipStream_t */

/* ---- library ------------------------------------------------------------------ */
/* ABI version, currently 2.  History -- 1: rounds 1-4.  2 (round 5/6): d2p_gemm_set_corun removed; the bit layout of
 * d2p_gemm_set_option changed (the LDS-DMA grid moved from bits 8.. to bits 16.., bit 7 = the embedding gradient stays a
 * one-hot GEMM): a caller written against version 1 that passes a grid in bits 8.. must be updated. */
int d2p_version(void);
const char* d2p_last_error(void);      /* thread-local, never NULL */
/* Fills: name (<=255 chars), number of CUs, wavefront size, HBM bytes.  HOST pointers. */
int d2p_device_info(int device, char* name, int name_len, int* cus, int* wave, size_t* hbm_bytes);

/* ---- K5: fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32, LDS-tiled) ---------------------
 * Replaces the cuBLAS/Eigen matmuls TF dispatches for slim.fully_connected
 * (models/ops.py:152), BasicLSTMCell's [x,h]·W (models/model_full.py:244-246),
 * layers_core.Dense (models/model_full.py:463-464) and their gradients.
 *   nn: C[M,N] (=|+=) A[M,K]·B[K,N]      nt: A[M,K]·B[N,K]^T      tn: A[K,M]^T·B[K,N]
 * lda/ldb/ldc are row strides in floats.  bias: [N] or NULL.  act: 0 none, 1 lrelu(0.2).
 * accumulate != 0 adds the previous contents of C (before bias/act are applied:
 * C = act(A·B + C_old + bias)).
 * ws: scratch for split-K, at least d2p_gemm_ws_bytes(M,N,K) bytes (may be NULL if 0). */
size_t d2p_gemm_ws_bytes(int M, int N, int K);
/* nb1 x nb0 equally shaped problems in one launch (the relation network's h / c summaries and their
 * two fc1 halves, models/model_full.py:333-349): problem (i, j) uses A + i*sA1 + j*sA0, B, C and bias
 * likewise (strides in floats; a zero stride shares the operand).  kind: 0 = nn, 1 = nt, 2 = tn with
 * the operand layouts of the entry points above.  K is never split: no workspace. */
int d2p_gemm_f32_batched(int kind, int nb1, int nb0, int M, int N, int K, const float* A, long lda,
                         long sA1, long sA0, const float* B, long ldb, long sB1, long sB0, float* C,
                         long ldc, long sC1, long sC0, const float* bias, long sbias1, long sbias0,
                         int act, int accumulate, d2p_stream_t stream);
/* Tuning knobs for the dense entry points (bit mask): bit 0 lets long-K problems on the 64x64 tile
 * use 32-deep K slabs; bit 1 switches OFF the small-problem path (32x32 tiles whose four waves split
 * K and combine through LDS, picked when the ordinary plan would fill fewer than 128 workgroups);
 * bit 2 keeps the select between global load and LDS store even when K is a multiple of the slab
 * depth (the dense loaders then need none); bit 6: the A^T B products stay on the staged kernel (no
 * gemm_tn_direct_kernel); bit 7: the embedding gradient stays a one-hot GEMM (no rows_by_key_kernel); bit 8: the large
 * A^T B products stay on 64 x 64 tiles (no gemm_tn_direct128_kernel); bits 16 and up:
 * persistent grid size of the LDS-DMA kernel (0 = CUs x resident workgroups per CU). */
int d2p_gemm_set_option(int bk32);
/* Tuning experiments only: force the tile (0 64x64, 1 128x128, 2 128x32, 3 256x32, 4 128x64, 5-7 the K-split
 * forms; 8-12 the LDS-DMA pipeline: 64x64 with a 4- / 3-deep ring, 128x64 3-deep, 128x128 2- / 3-deep, which
 * falls back to the staged kernel of the same tile when the operands are not 16-byte aligned or K is not a
 * multiple of 32; -1 auto) and the split-K factor (0 auto) of the dense entry points. */
int d2p_gemm_force_plan(int tile, int splits);
int d2p_gemm_f32_nn(int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                    float* C, long ldc, const float* bias, int act, int accumulate,
                    void* ws, size_t ws_bytes, d2p_stream_t stream);
int d2p_gemm_f32_nt(int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                    float* C, long ldc, const float* bias, int act, int accumulate,
                    void* ws, size_t ws_bytes, d2p_stream_t stream);
int d2p_gemm_f32_tn(int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                    float* C, long ldc, const float* bias, int act, int accumulate,
                    void* ws, size_t ws_bytes, d2p_stream_t stream);
/* Products over a list of rows (the ACTIVE rows of a padded, time-major batch: rows past a sequence's length hold
 * zeros and nothing reads their results): row x of the product is computed from row rows[x] of A and stored at row
 * rows[x] of C; M = number of listed rows, unlisted rows of C are untouched.  kind 0: C = A . B + bias (B [K, N]);
 * kind 1: C = A . B^T + bias (B [N, K]).  Same summation order per element as d2p_gemm_f32_nn / _nt on the same
 * tile plan.  Workspace: d2p_gemm_ws_bytes(M, N, K). */
int d2p_gemm_f32_rows(int kind, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C,
                      long ldc, const float* bias, const int* rows, void* ws, size_t ws_bytes, d2p_stream_t stream);
/* C[m, n] (+)= sum_{x < K} A[rowsA[x]*lda + m] * B[rowsB[x]*ldb + n]: the 'tn' product (d2p_gemm_f32_tn: C = A^T B)
 * with BOTH operands read through lists of K row indices.  For the weight gradients X^T dZ / h[t-1]^T dz[t] of a
 * padded time-major batch over the rows inside their sequences only (rows past a sequence's length are zeros in
 * dZ; the reference multiplies them in, tf.gradients of models/model_full.py:243-277).  K % 32 == 0 takes the
 * select-free loaders: pad the lists with a row that is zero in B and finite in A.
 * Round 4: with M, N multiples of 4, K a multiple of 16 and >= 1024, 16-byte aligned operands and at least 128 output
 * tiles of 64 x 64 these products (and the dense d2p_gemm_f32_tn without bias / activation) run on
 * gemm_tn_direct_kernel: operands straight from memory into the MFMA registers, no LDS staging;
 * deterministic, another order of the K sum than the staged kernel.  d2p_gemm_set_option bit 6 or
 * D2P_GEMM_TN_DIRECT=0 keeps the staged kernel. */
int d2p_gemm_f32_tn_rows(int M, int N, int K, const float* A, long lda, const int* rowsA, const float* B, long ldb,
                         const int* rowsB, float* C, long ldc, int accumulate, void* ws, size_t ws_bytes,
                         d2p_stream_t stream);
/* Round 6: C[:M0] (+)= A0^T B and C[M0 : M0 + M1] (+)= A1^T B through ONE pair of row lists -- the input and the
 * recurrent half of an LSTM's kernel gradient, [X | H]^T dZ (tf.gradients of the `kernel` variable of
 * tf.contrib.rnn.BasicLSTMCell, models/model_full.py:243-246: one [I + U, 4U] tensor).  M0 % 128 == 0 and (M0 + M1) x N
 * = at least 256 tiles of 128 x 64 with K >= 1024: one launch of gemm_tn_direct128_kernel (a 128 x 64 output tile per wave: three
 * 16-byte loads per 32 MFMAs); anything else, or d2p_gemm_set_option bit 8: the two d2p_gemm_f32_tn_rows products one
 * after the other, the same values bit for bit.  ws >= d2p_gemm_ws_bytes(M0 + M1, N, K). */
int d2p_gemm_f32_tn_rows2(int M0, int M1, int N, int K, const float* A0, long lda0, const float* A1, long lda1,
                          const int* rowsA, const float* B, long ldb, const int* rowsB, float* C, long ldc, int accumulate,
                          void* ws, size_t ws_bytes, d2p_stream_t stream);
/* Round 6: two independent products of one shape through one pair of row lists, C0 (+)= A0^T B0 and C1 (+)= A1^T B1 -- the
 * recurrent halves of the action and the perception decoder's kernel gradients (models/model_full.py:530-599: one decoder
 * graph per demonstration index, all of them sharing these weights).  2 x (M / 128) x (N / 64) >= 256 tiles with M % 128 == 0,
 * K >= 1024: one launch of gemm_tn_direct128_kernel; anything else, or d2p_gemm_set_option bit 8: one after the other, the
 * same values bit for bit.  ws >= d2p_gemm_ws_bytes(M, N, K). */
int d2p_gemm_f32_tn_rows_x2(int M, int N, int K, const float* A0, long lda0, const float* B0, long ldb0, float* C0,
                            const float* A1, long lda1, const float* B1, long ldb1, float* C1, long ldc, const int* rowsA,
                            const int* rowsB, int accumulate, void* ws, size_t ws_bytes, d2p_stream_t stream);
/* out[c] = sum_r X[r*ld + c]  (bias gradients).  ws >= d2p_colsum_ws_bytes. */
size_t d2p_colsum_ws_bytes(int rows, int cols);
int d2p_colsum_f32(int rows, int cols, const float* X, long ld, float* out,
                   void* ws, size_t ws_bytes, d2p_stream_t stream);

/* ---- K1: NHWC 3x3 stride-2 TF-SAME convolution as implicit-im2col MFMA GEMM -------
 * Replaces slim.conv2d(3x3, stride 2, SAME) (models/ops.py:30, called from
 * State_Encoder, models/model_full.py:219-229) and its cuDNN gradients.
 * x: [N,H,W,Cin] (fp32, or uint8 when x_is_u8 != 0 -- widened on load);
 * w: [3,3,Cin,Cout]; y: [N,Ho,Wo,Cout], Ho=ceil(H/2), Wo=ceil(W/2).
 * SAME padding is asymmetric: even sizes pad (0 before, 1 after); odd sizes (1,1).
 * fwd applies  y = act(conv(x,w) + bias), act: 0 none / 1 lrelu(0.2). */
size_t d2p_conv_ws_bytes(int N, int H, int W, int Cin, int Cout);
int d2p_conv2d_nhwc_s2_same_fwd(int N, int H, int W, int Cin, int Cout,
                                const void* x, int x_is_u8, const float* w, const float* bias,
                                int act, float* y, d2p_stream_t stream);
/* dx[N,H,W,Cin] = conv_transpose(dy[N,Ho,Wo,Cout], w) */
int d2p_conv2d_nhwc_s2_same_dgrad(int N, int H, int W, int Cin, int Cout,
                                  const float* dy, const float* w, float* dx, d2p_stream_t stream);
/* dw[3,3,Cin,Cout] = im2col(x)^T · dy  (split-K over the N*Ho*Wo rows; deterministic) */
int d2p_conv2d_nhwc_s2_same_wgrad(int N, int H, int W, int Cin, int Cout,
                                  const void* x, int x_is_u8, const float* dy, float* dw,
                                  void* ws, size_t ws_bytes, d2p_stream_t stream);

/* Back-end selection for the three conv entry points (tuning / A-B measurement only): the
 * narrow layers (Cin,Cout) in {(4,16), (16,16), (16,32)} and the 2x2 -> 1x1 (32,48) layer
 * run on register-resident-filter direct kernels by default (selector 2: whole-frame LDS-staged
 * kernels for the Karel geometries, gather kernels otherwise; 1: gather kernels only); 0 routes
 * that direction through the implicit-im2col GEMM instead.  Results agree to fp32 rounding.
 * Round 6: selector 2 also covers the LDS-filter kernels of conv_wide.hip (the 48-channel layers in every direction; the
 * large 16 -> 32 layer's forward and input gradient in block form); fwd = 3 / dgrad = 3: as 2, with that 16 -> 32 layer
 * back on the round-5 gather / row-strip kernels (A/B measurements: tools/bench_conv2.py). */
int d2p_conv_set_direct(int fwd, int dgrad, int wgrad);
/* > 0 sets a knob, 0 leaves it, < 0 (fwd, wgrad) returns it to the automatic per-layer choice */
int d2p_conv_direct_tune(int fwd_tiles_per_wave, int dgrad_tiles_per_wave, int wgrad_workgroups);

/* ---- K2: training-mode batch norm over row groups (+ fused lrelu backward) ---------
 * Replaces tf.contrib.layers.batch_norm(is_training=True, decay=0.9,
 * updates_collections=None) (models/ops.py:20-23).  The reference calls the encoder
 * once per demonstration index (models/model_full.py:373-379), so statistics are per
 * GROUP: x is [R, C]; group(r) = (r / inner) % G; G = 1 is an ordinary batch norm.
 * Statistics: per (group, channel) mean and BIASED variance, eps = 1e-3, accumulated
 * in fp64.  mean/rstd: [G, C] outputs (kept for backward).
 * y = gamma*(x-mean)*rstd + beta.   var_out ([G,C] or NULL) receives the biased variance.
 * bwd: given dy, the forward input x and saved mean/rstd, writes
 *   dx = gamma*rstd*(dy - mean_g(dy) - xhat*mean_g(dy*xhat)) [* lrelu'(x) if act_bwd]
 *   dgamma[c] = sum_all dy*xhat,  dbeta[c] = sum_all dy.
 * act_bwd = 1: x is the lrelu OUTPUT feeding BN (conv/fc -> lrelu -> BN order,
 * models/ops.py:14-24), and dx is additionally multiplied by lrelu'(pre-activation),
 * whose sign equals sign(x): 1 (x>0), 0.2 (x<0), 0.6 (x==0, TF's abs'(0)=0).
 * dx_colsum (nullable, [C]): column sums of dx = the gradient of the bias that the layer under
 * the BN adds before the activation (slim.conv2d / slim.fully_connected biases), produced by
 * the same pass that writes dx. */
size_t d2p_bn_ws_bytes(int R, int C, int G);
int d2p_bn_group_fwd(int R, int C, int G, int inner, const float* x, const float* gamma,
                     const float* beta, float* y, float* mean, float* rstd, float* var_out,
                     float* moving_mean, float* moving_var, float decay,
                     void* ws, size_t ws_bytes, d2p_stream_t stream);
int d2p_bn_group_bwd(int R, int C, int G, int inner, const float* x, const float* dy,
                     const float* gamma, const float* mean, const float* rstd, int act_bwd,
                     float* dx, float* dgamma, float* dbeta, float* dx_colsum,
                     void* ws, size_t ws_bytes, d2p_stream_t stream);
/* The same with the partial sums [G][S_sums][C][2] (fp64: sum dy, sum dy * xhat per (group, slice)) left behind by the
 * launch that produced dy (d2p_conv2d_nhwc_s2_same_dgrad_bn): the partial-sum pass over (x, dy) is not run. */
int d2p_bn_group_bwd_sums(int R, int C, int G, int inner, const float* x, const float* dy,
                          const float* gamma, const float* mean, const float* rstd,
                          int act_bwd, float* dx, float* dgamma, float* dbeta, float* dx_colsum,
                          const double* sums, int S_sums, void* ws, size_t ws_bytes, d2p_stream_t stream);
/* nb independent, equally shaped batch-norm problems in ONE set of launches (grid.z = problem).  Problem b uses
 * every pointer of problem 0 plus b times a stride, in floats: xs for the [R, C] inputs (x, dy), ys for the
 * [R, C] outputs (y, dx), ps for per-channel parameters and their gradients (gamma, beta, dgamma, dbeta,
 * dx_colsum), ms for the moving statistics; mean / rstd / var_out are [nb][G][C]; the workspace holds
 * d2p_bn_batched_ws_bytes(nb, R, C, G).  Same arithmetic as nb separate calls (bit-identical).  Used for the two
 * relation networks (rn_h / rn_c: same shapes, own parameters, models/model_full.py:333-362). */
size_t d2p_bn_batched_ws_bytes(int nb, int R, int C, int G);
int d2p_bn_group_fwd_batched(int nb, long xs, long ys, long ps, long ms, int R, int C, int G, int inner,
                             const float* x, const float* gamma, const float* beta, float* y, float* mean,
                             float* rstd, float* var_out, float* moving_mean, float* moving_var, float decay,
                             void* ws, size_t ws_bytes, d2p_stream_t stream);
int d2p_bn_group_bwd_batched(int nb, long xs, long ys, long ps, int R, int C, int G, int inner, const float* x,
                             const float* dy, const float* gamma, const float* mean, const float* rstd,
                             int act_bwd, float* dx, float* dgamma, float* dbeta, float* dx_colsum, void* ws,
                             size_t ws_bytes, d2p_stream_t stream);
/* The Karel State_Encoder's forward pass in ONE launch (models/model_full.py:362-381 of the reference: three times
 * conv 3x3 stride 2 SAME -> +bias -> leaky ReLU 0.2 -> batch norm with the statistics of one demonstration index, on
 * 8x8x16 frames): frames x [B, G, T, 8, 8, 16] (fp32, or uint8 with x_is_u8), layer l's filters w[l] [3,3,Cin,Cout]
 * (16->16, 16->32, 32->48), bias / gamma / beta [Cout].  Writes what the separate launches write: a[l] the
 * activations before normalisation ([NF,4,4,16], [NF,2,2,32], [NF,48]; NF = B*G*T), y[0], y[1] the normalised
 * outputs of layers 1 and 2, feats_tm [T, B*G, 48] the normalised features time-major, mean / rstd / var [l] [G, Cout]
 * (biased variance; feed d2p_bn_update_moving).  w .. var are HOST arrays of three device pointers (y: two).
 * One workgroup holds the frames of a slice of the programs of one demonstration index in LDS through all three
 * layers; the workgroups of an index exchange their fp64 partial sums through ws (d2p_karel_encoder_ws_bytes)
 * and an arrival counter, so all G*S workgroups must be co-resident: D2P_EINVAL when B, G, T do not allow that
 * (T % 4 != 0, G > 32, more frames per workgroup than LDS holds) -- callers fall back to the separate launches.
 * A workgroup that waits too long for its group sets the status word of d2p_lstm_persist_error (code 0x7e). */
size_t d2p_karel_encoder_ws_bytes(int B, int G, int T);   /* 0: geometry not supported */
/* diagnostic: buf [workgroups][10] uint64 in device memory receives wall-clock stamps (100 MHz) of the following launches
 * (start, after each layer's products / statistics exchange / normalisation, end); null: off */
int d2p_karel_encoder_set_trace(void* buf);
int d2p_karel_encoder_fwd(int B, int G, int T, const void* x, int x_is_u8, const float* const* w,
                          const float* const* bias, const float* const* gamma, const float* const* beta,
                          float* const* a, float* const* y, float* feats_tm, float* const* mean,
                          float* const* rstd, float* const* var, void* ws, size_t ws_bytes, d2p_stream_t stream);
/* The Karel State_Encoder's BACKWARD pass in one launch + one combine launch (round 5; the reverse of
 * models/model_full.py:216-231 / models/ops.py:14-33 as tf.gradients runs it: per layer batch-norm backward with the
 * statistics of one demonstration index, lrelu', the conv's weight / bias gradient, and the input gradient of layers 3
 * and 2).  dfeat_tm [T, B*G, 48] is the gradient of d2p_karel_encoder_fwd's feats_tm (time-major: no transpose pass);
 * x, w, gamma, beta, a, mean, rstd are that launch's inputs and outputs (a = the activations before normalisation);
 * dw[l] [3,3,Cin,Cout], db / dgamma / dbeta [l] [Cout] receive the gradients (written, not accumulated).  All array
 * arguments are HOST arrays of three device pointers.  Same workgroup decomposition and co-residency requirement as
 * the forward launch (the S workgroups of an index meet once per layer for the batch-norm sums; 0x7d in the status
 * word of d2p_lstm_persist_error when one waits too long); every gradient is a sum over the workgroups' slabs in ws,
 * added in workgroup order by the second launch (deterministic).  D2P_EINVAL for geometries it does not take
 * (d2p_karel_encoder_bwd_ws_bytes == 0): callers run the separate launches. */
size_t d2p_karel_encoder_bwd_ws_bytes(int B, int G, int T);   /* 0: geometry not supported */
int d2p_karel_encoder_bwd(int B, int G, int T, const void* x, int x_is_u8, const float* dfeat_tm,
                          const float* const* w, const float* const* gamma, const float* const* beta,
                          const float* const* a, const float* const* mean, const float* const* rstd,
                          float* const* dw, float* const* db, float* const* dgamma, float* const* dbeta, void* ws,
                          size_t ws_bytes, d2p_stream_t stream);
/* The relation networks' pointwise chains around their two GEMMs in four launches (round 5; rn_pool of
 * models/model_full.py:333-349 for BOTH summaries: leading dimension 2 everywhere, parameters of the second summary
 * pstride floats behind the first's).  B programs, k demonstrations, U units; rows of a summary ordered (b, a, c).
 *   d2p_rn_fc1_fwd: y1a[b,a,c] = lrelu(P[b,c] + Q[b,a] + bias) (P, Q [2, B*k, U]: the two half-projections of fc1) and y1 =
 *       its batch norm over all B*k*k rows (training mode, one group): the sums come from recomputed values, y1a and y1 are
 *       written once; mean / rstd / var [2, U]; moving statistics [2, U] updated in the launch when non-null.
 *   d2p_rn_fc2_fwd: out[b] = mean over the pairs of batch norm(y2a)[b] (+ mean over k of feat [2, B*k, U] when non-null:
 *       the avg-pool branch) from ONE read of y2a [2, B*k*k, U] (batch norm is affine per column: it commutes with the
 *       mean); psum [2, B, U] = the per-program sums of y2a, for the backward.
 *   d2p_rn_fc2_bwd: dpre [2, B*k*k, U] = gradient of fc2's pre-activation from dout [2, B, U] (the batch-norm backward's
 *       sums are closed forms of dout and psum), dgamma / dbeta [U] (+ pstride); fc2's bias gradient is finished by
 *   d2p_rn_fc1_bwd: batch-norm backward of fc1 (+ lrelu') from dy1, summed over a into dP [2, B*k, U] and over c into dQ
 *       (the pre-activation gradient itself is never written); dgamma, dbeta, dbias of fc1 and dbias2 of fc2.
 * The B workgroups of a (summary, 128-column slice) exchange fp64 partial sums through ws (d2p_rn_ws_bytes; one ws for the
 * four calls of a step) and must be co-resident: D2P_EINVAL / d2p_rn_ws_bytes == 0 for geometries that are not taken
 * (U % 128, k > 32, more workgroups than two per CU) -- callers run the separate launches.  Same values as those up to the
 * order of the fp64 sums.  A workgroup that waits too long sets the status word of d2p_lstm_persist_error (code 0x7c). */
size_t d2p_rn_ws_bytes(int B, int k, int U);   /* 0: geometry not supported */
int d2p_rn_fc1_fwd(int B, int k, int U, const float* P, const float* Q, const float* bias, const float* gamma,
                   const float* beta, long pstride, float* y1a, float* y1, float* mean, float* rstd, float* var,
                   float* moving_mean, float* moving_var, float decay, void* ws, size_t ws_bytes, d2p_stream_t stream);
int d2p_rn_fc2_fwd(int B, int k, int U, const float* y2a, const float* gamma, const float* beta, long pstride,
                   const float* feat, float* out, float* psum, float* mean, float* rstd, float* var, float* moving_mean,
                   float* moving_var, float decay, void* ws, size_t ws_bytes, d2p_stream_t stream);
int d2p_rn_fc2_bwd(int B, int k, int U, const float* y2a, const float* dout, const float* psum, const float* gamma,
                   long pstride, const float* mean, const float* rstd, float* dpre, float* dgamma, float* dbeta, void* ws,
                   size_t ws_bytes, d2p_stream_t stream);
int d2p_rn_fc1_bwd(int B, int k, int U, const float* y1a, const float* dy1, const float* gamma, long pstride,
                   const float* mean, const float* rstd, float* dP, float* dQ, float* dgamma, float* dbeta, float* dbias,
                   float* dbias2, void* ws, size_t ws_bytes, d2p_stream_t stream);
/* Batch norm folded into the conv launches (round 5; the ViZDoom-size layers of models/model_full.py:216-231, whose
 * conv -> lrelu -> batch-norm chain of models/ops.py:14-33 otherwise writes and re-reads each activation three times).
 * Frames are ordered (program, demonstration index, step): the statistics of frame n belong to index g = (n / seq) % G.
 *   d2p_conv_bn_slices: slices S per index the folding forward launch of this geometry writes partial sums for
 *       (0: no folding kernel for it -- run d2p_conv2d_nhwc_s2_same_fwd + d2p_bn_group_fwd);
 *   d2p_conv2d_nhwc_s2_same_fwd_bn: the forward conv (+bias, act) that also leaves stats [G][S][Cout][2] fp64 = (sum, sum
 *       of squares) of its outputs per (index, slice), and -- in_scale / in_shift [G, Cin] non-null -- reads its input as
 *       x * in_scale[g] + in_shift[g]: x is the PREVIOUS layer's pre-norm activation and the affine that layer's
 *       batch-norm apply, so the normalised tensor is never written (zero padding applies to the normalised values).
 *       With the affine, x's N*H*W*Cin elements must be FOLLOWED by G*Cin floats, index g's "pad pixel" -in_shift[g] /
 *       in_scale[g] (d2p_bn_stats_from_partials writes them): out-of-image taps load it instead of being masked;
 *   d2p_conv2d_nhwc_s2_same_wgrad_bn: the weight gradient with the same input affine;
 *   d2p_bn_stats_from_partials: mean / rstd / var [G, C] (biased variance; feed d2p_bn_update_moving) from such partial
 *       sums over n_per_group values per (index, channel), and (scale, shift non-null) the folded affine scale = gamma *
 *       rstd (kept >= 1e-20 in magnitude), shift = beta - mean * scale, pad (nullable) = -shift / scale [G, C];
 *   d2p_bn_apply_fwd: the apply pass of d2p_bn_group_fwd alone, y = gamma * (x - mean[g]) * rstd[g] + beta.
 * Same values as the separate launches up to the order of the fp64 sums / one fp32 rounding of the affine. */
int d2p_conv_bn_slices(int N, int H, int W, int Cin, int Cout, int G, int seq);
/* 1 when this layer's folding forward and weight-gradient launches read their input through the previous layer's batch-norm
 * apply (in_scale / in_shift + the G pad pixels behind x): the caller then never writes the normalised tensor. */
int d2p_conv_bn_affine_ok(int N, int H, int W, int Cin, int Cout, int G, int seq);
int d2p_conv2d_nhwc_s2_same_fwd_bn(int N, int H, int W, int Cin, int Cout, const void* x, int x_is_u8, const float* w,
                                   const float* bias, int act, float* y, int G, int seq, const float* in_scale,
                                   const float* in_shift, double* stats, int S, d2p_stream_t stream);
int d2p_conv2d_nhwc_s2_same_wgrad_bn(int N, int H, int W, int Cin, int Cout, const void* x, int x_is_u8, const float* dy,
                                     float* dw, int G, int seq, const float* in_scale, const float* in_shift, void* ws,
                                     size_t ws_bytes, d2p_stream_t stream);
int d2p_bn_stats_from_partials(int n_per_group, int C, int G, int S, const double* partial, const float* gamma,
                               const float* beta, float* mean, float* rstd, float* var, float* scale, float* shift,
                               float* pad, d2p_stream_t stream);
int d2p_bn_apply_fwd(int R, int C, int G, int inner, const float* x, const float* gamma, const float* beta,
                     const float* mean, const float* rstd, float* y, d2p_stream_t stream);
/* The FIRST conv layer's batch-norm backward folded into its weight gradient (nothing needs that layer's input
 * gradient): d2p_bn_group_bwd_coef runs the sums of d2p_bn_group_bwd and leaves, instead of dx, the coefficients
 * coef [G, C, 4] = (k1, k2, k3, 0) of dx = (k1 * dy + k2 * x + k3) * lrelu'(x) (+ dgamma, dbeta [C]; ws as
 * d2p_bn_group_bwd); d2p_conv2d_nhwc_s2_same_wgrad_bnbwd forms that dx from act (= x, the pre-norm activation) and dy as
 * it loads them -- the apply pass and the materialised dx are gone -- and also returns dbias [Cout] = column sums of dx.
 * d2p_conv_bnbwd_ok: 1 when the geometry has the folding kernel (80-wide 4 -> 16 layers; ws: d2p_conv_ws_bytes). */
int d2p_bn_group_bwd_coef(int R, int C, int G, int inner, const float* x, const float* dy, const float* gamma,
                          const float* mean, const float* rstd, float* coef, float* dgamma, float* dbeta,
                          const double* sums, int S_sums, void* ws, size_t ws_bytes, d2p_stream_t stream);
/* ... and its partial sums from the PRODUCER of dy: the input-gradient launch of the next conv layer leaves stats
 * [G][S][Cin][2] fp64 = (sum dx, sum dx * xhat) per (index, slice) of what it writes (xhat from act, the pre-norm
 * activation dx belongs to, and that layer's mean / rstd [G, Cin]); pass them as `sums`, S as `S_sums` to
 * d2p_bn_group_bwd_coef (x, dy may then be null: no pass over them at all).  d2p_conv_dgrad_bn_slices: S, 0 when the
 * geometry has no such kernel. */
int d2p_conv_dgrad_bn_slices(int N, int H, int W, int Cin, int Cout, int G, int seq);
int d2p_conv2d_nhwc_s2_same_dgrad_bn(int N, int H, int W, int Cin, int Cout, const float* dy, const float* w, float* dx,
                                     const float* act, const float* mean, const float* rstd, int G, int seq,
                                     double* stats, int S, d2p_stream_t stream);
int d2p_conv_bnbwd_ok(int N, int H, int W, int Cin, int Cout);
int d2p_conv2d_nhwc_s2_same_wgrad_bnbwd(int N, int H, int W, int Cin, int Cout, const void* x, int x_is_u8,
                                        const float* act, const float* dy, const float* coef, int G, int seq, float* dw,
                                        float* dbias, void* ws, size_t ws_bytes, d2p_stream_t stream);
/* d2p_bn_group_fwd's moving_mean / moving_var ([C], nullable together): the G moving-average
 * updates of this call (one per group = one per reference BN call, in group order) are applied
 * by the statistics kernel itself; d2p_bn_update_moving below is the same update stand-alone. */
/* Inference mode (is_training=False, evaler.py:61): y = (x - moving_mean) * rsqrt(moving_var +
 * 1e-3) * gamma + beta per channel.  x, y: [R, C]. */
int d2p_bn_inference_fwd(int R, int C, const float* x, const float* gamma, const float* beta,
                         const float* moving_mean, const float* moving_var, float* y,
                         d2p_stream_t stream);
/* moving <- decay*moving + (1-decay)*batch, applied G times in group order (the
 * reference updates once per Demo_Encoder call).  moving_mean/var: [C]. */
int d2p_bn_update_moving(int C, int G, float decay, const float* mean, const float* var,
                         float* moving_mean, float* moving_var, d2p_stream_t stream);
/* Tuning / A-B switch (process-global, default 0 -- measured equal to slightly slower on MI355X): 1 = the finalize step of a training-mode batch norm with few
 * partial sums per group (the conv layers) runs inside the partial-sum launch -- the last workgroup of each group
 * folds that group, the last group updates the moving statistics / dgamma, dbeta -- and the bias-gradient column
 * sums inside the backward apply launch; 0 (default) = separate finalize launches.  Same sums in a fixed order either way. */
int d2p_bn_set_fold(int bits);   /* bit 1 set: the round-2 finalize kernels (one wavefront per channel over all groups) */

/* ---- K3/K4: LSTM gate pointwise, standalone ----------------------------------------
 * Replaces the elementwise tail of rnn.BasicLSTMCell.call (models/model_full.py:244-246):
 *   z = [i, j, f, o] pre-activations (row r at z + r*z_row_stride, 4U floats);
 *   c' = c*sigmoid(f+1) + sigmoid(i)*tanh(j);  h' = tanh(c')*sigmoid(o).
 * lens/t implement tf.nn.dynamic_rnn(sequence_length) (models/model_full.py:254-256):
 * rows with t >= lens[r] copy (c,h) through and emit h_out = 0.  lens == NULL: no masking.
 * Algorithmic bytes/row: fwd 14336 (U=512), bwd 26624. */
int d2p_lstm_gate_fwd(int M, int U, const float* z, long z_row_stride, const float* c_prev,
                      const float* h_prev, const int* lens, int t,
                      float* c_out, float* h_state_out, float* h_out, d2p_stream_t stream);
/* dh_in: gradient wrt the state h after this step ([M,U], may be NULL = 0);
 * dh_out_grad: gradient wrt the emitted output row ([M,U], may be NULL);
 * dc: in/out gradient wrt state c ([M,U]); dz: out [M rows, stride dz_row_stride, 4U].
 * For masked rows dz = 0, dc unchanged and dh_pass (if non-NULL) receives dh_in
 * (else 0), so that dh_prev = dz·Wh^T + dh_pass. */
int d2p_lstm_gate_bwd(int M, int U, const float* z, long z_row_stride, const float* c_prev,
                      const float* c, const float* dh_in, const float* dh_out_grad,
                      const int* lens, int t, float* dc, float* dz, long dz_row_stride,
                      float* dh_pass, d2p_stream_t stream);

/* ---- LSTM over a sequence (recurrent GEMM + gate per step) --------------------------
 * Replaces tf.nn.dynamic_rnn(BasicLSTMCell) (models/model_full.py:254-256,274-276) and
 * the BasicDecoder/TrainingHelper loop (models/model_full.py:413,465-471).
 * z: in = hoisted input projection x·Wx + b for every step; out = full pre-activations
 *    (element (row r, step t) at z + r*z_row_stride + t*z_t_stride, 4U floats).
 * Wh: [U, 4U] recurrent rows of the LSTM kernel.  h0/c0: [M,U] or NULL (zeros).
 * lens: [M] int32 or NULL.  hout: [n_steps, M, U] emitted outputs (0 past len).
 * cs: [n_steps, M, U] cell state after each step.  h_final/c_final: [M,U].
 * ws: >= d2p_lstm_ws_bytes(M,U). */
size_t d2p_lstm_ws_bytes(int M, int U);
/* Tuning knob (process-global): 1 (default) = fused recurrent-step kernels (one launch per
 * step: h·Wh MFMA + gates) when U in {64,128,256,512} and ws >= d2p_lstm_ws_bytes;
 * 0 = generic GEMM + gate kernel per step.  Both produce the same results (tests compare). */
int d2p_lstm_set_fused(int on);
/* Tuning knob (process-global): 1 (default) = persistent sequence kernels (lstm_persist.hip: ONE
 * launch runs all n_steps; the workgroup's slice of Wh stays in registers, the new h / dz rows
 * travel between workgroups of a row domain through write-through stores + flags) whenever the
 * fused path is eligible, U/8 (forward) resp. U/16 (backward) column tiles fit the device's CUs
 * and M <= 128 rows per row domain; 0 = one fused launch per step.  Same results (tests compare). */
int d2p_lstm_set_persistent(int on);
/* Synchronising status query (NOT stream-async, do not call during graph capture): 0 = every
 * persistent launch so far completed its hand-offs; non-zero = a bounded spin timed out (a
 * workgroup was not resident, e.g. the device was shared) and the results of that launch are
 * invalid: (code << 24) | 0x800000 | block.  reset != 0 clears the word. */
int d2p_lstm_persist_error(int reset);
/* Test hook (synchronising): sets the status word as a timed-out hand-off would (code 0x7f). */
int d2p_lstm_persist_inject_error(void);
/* Debugging (tools/trace_lstm_persist.py): workgroup `block` of every following persistent launch
 * writes shader-clock stamps of each phase into buf (device memory, >= 2*512*8 uint64; NULL = off):
 * [role][tick][8] with role 0 = MFMA wave 0 {start, first half issued, flags seen, partials written,
 * barrier passed}, role 1 = epilogue wave {start, barrier passed, stores issued, stores drained}. */
int d2p_lstm_persist_set_trace(void* buf, size_t bytes, int block);
/* Tuning knob (process-global): the backward persistent kernel runs a row domain in the deferred form -- the
 * product-dependent half of a phase's gate backward inside the NEXT phase's MFMA chain, one barrier per phase, rows
 * published one phase later -- when the domain has at least `from_phases` 16-row phases per step (<= 0: never, the
 * default: measured -4 % per phase at 5-7 phases in isolation and nothing in the training step).  Same results. */
int d2p_lstm_persist_set_bwd_defer(int from_phases);
/* A/B switch (default 1): the persistent backward kernel's global stores and operand prefetches go through buffer
 * descriptors with scalar step / gate offsets (a few VALU instructions per phase instead of ~60: beside an fp32 MFMA
 * chain every VALU instruction is paid in full); 0 = 64-bit pointers, round 3's form.  Same results. */
int d2p_lstm_persist_set_bwd_desc(int on);
/* Words of a d2p_lstm_*_desc.flags buffer; and the A/B switch of the direct launches (1 default; 0: every persistent
 * launch gets its preparation launch whatever the descriptor says). */
size_t d2p_lstm_flag_words(void);
/* A/B switch: 0 = ignore rowmap / slab_steps of d2p_lstm_bwd_desc (every domain runs all steps); default 1 */
int d2p_lstm_persist_set_sorted(int on);
/* How a wave of the persistent kernels waits for a hand-off flag (default 16).  Bits 0-1: pipelined polls -- 1 = in row
 * domains of two and more phases a waiting wave keeps two reads of its flag in flight instead of read / pause / read,
 * 2 = in single-phase domains as well (the kernels gain 5-9 % in isolation, the training step loses 1 %: off).  Bits
 * 2-4: n extra pauses between two polls, (1 + n) x s_sleep(4) -- n = 4 by default: fewer polls cost a waiting wave
 * little and leave the L2 to the GEMMs of the other queue (training step -1 %; n = 0: 3.010 ms, 3: 2.981, 4: 2.974 -
 * 2.982, 5: 2.981, 7: 3.027). */
int d2p_lstm_persist_set_poll(int pipelined);
/* Tuning knob of the length-sorted planner: cost model of a backward row domain per pass, max(us_per_phase * phases,
 * floor_us); defaults 3.3 / 6.3 (values <= 0 leave a parameter unchanged) */
int d2p_lstm_persist_set_plan_cost(double us_per_phase, double floor_us);
/* The packed weight images of n <= 8 cells (Wh[i]: [U, 4U] row-major) in one launch: Wf[i] / Wb[i] (4*U*U floats each,
 * NULL: skip) are what d2p_lstm_fwd_desc.wpack / d2p_lstm_bwd_desc.wpack take.  Wh, Wf, Wb: HOST arrays of device
 * pointers. */
int d2p_lstm_pack_weights(int n, int U, const float* const* Wh, float* const* Wf, float* const* Wb, d2p_stream_t stream);
int d2p_lstm_persist_set_direct(int on);
/* Tuning knob (process-global): workgroups per CU the persistent forward / backward kernels are
 * sized for (0 keeps the current value; default 1).  2 cuts the rows into twice as many domains so
 * that two workgroups share a CU and overlap each other's MFMA and epilogue phases. */
int d2p_lstm_persist_set_wgs_per_cu(int fwd, int bwd);
/* CUs the persistent launches are planned for (process-global; 0 = all of them, the default).  A smaller budget -- 224 of
 * 256: seven row domains of 32 column tiles instead of eight -- leaves whole CUs to other queues: four-wave workgroups
 * (GEMMs, a collective's kernels) cannot become resident on a CU that holds a recurrence's workgroup. */
int d2p_lstm_persist_set_cu_budget(int cus);
/* The wide-tile forward kernel (round 4: 16 units per column tile -- 8 row domains at U = 512 --, one to three
 * sequences per launch, length-sorted where a descriptor brings rowmap / slab_steps, row domains that find all their
 * workgroups on one XCD exchange through its L2).  on: 1 (default) / 0 = every forward launch goes to the 8-unit-tile
 * kernel.  la_from / defer_from: phases per row domain from which a domain requests its next rows ahead / runs the
 * deferred gate math (defaults 3 / 5; values below 2 / 3 leave them unchanged).  xcd_local (negative: unchanged): bit 0
 * = L2-local hand-offs in domains found on one XCD (default 1; 0: always write-through); bits 1 / 2 (experiments): the
 * look-ahead request after 2 (default) / 3 quarters of a phase's MFMA chain.  Results are bit-identical in every setting. */
int d2p_lstm_persist_set_fwd_wide(int on, int la_from, int defer_from, int xcd_local);
/* ... and two scale knobs of its planner's per-step cost table (psw_step_cost in lstm_persist.hip): the per-phase
 * costs (default 2.7 = as measured) and the single-phase step (default 4.6 us); values <= 0 leave a knob unchanged */
int d2p_lstm_persist_set_fwd_plan_cost(double us_per_phase, double floor_us);
/* Wide-tile forward launches so far that carried nseq = 1, 2, 3 sequences (nseq = 0: those with a length-sorted
 * sequence).  For tests: the path must not be skipped silently. */
int d2p_lstm_persist_wide_launches(int nseq);
/* Workgroups of wide-tile launches so far that found all 32 workgroups of their row domain on their own XCD and
 * switched to L2-local hand-offs (a statistic; synchronising; reset != 0 zeroes it). */
int d2p_lstm_persist_wide_local_wgs(int reset);
/* Number of persistent launches so far that carried TWO sequences (d2p_lstm_seq_{fwd,bwd}_multi with nseq == 2
 * puts both on disjoint workgroups of one launch when both shapes are taken and sharing the chip is expected to
 * beat two launches back to back).  For tests: the pair path must not be skipped silently. */
int d2p_lstm_persist_pair_launches(void);
/* Ablation knobs for tools/bench_lstm_step.py only (results are wrong when non-zero):
 * bit 0 skips the MFMA part of the fused step kernels, bit 1 skips their epilogue. */
int d2p_lstm_debug_flags(int flags);
/* Tuning knobs: workgroups per launch the fused step kernels aim at (0 keeps the current value);
 * fwd_pipelined 1 selects the register-lean forward kernel (2 workgroups/CU), 0 the all-loads-
 * up-front one, -1 keeps the current choice. */
int d2p_lstm_set_tiling(int fwd_wgs, int bwd_wgs, int fwd_pipelined);
int d2p_lstm_seq_fwd(int M, int U, int n_steps, float* z, long z_row_stride, long z_t_stride,
                     const float* Wh, const float* h0, const float* c0, const int* lens,
                     float* hout, float* cs, float* h_final, float* c_final,
                     void* ws, size_t ws_bytes, d2p_stream_t stream);
/* Backward through the sequence.  dhout: [n_steps,M,U] or NULL; dh_final/dc_final: [M,U]
 * or NULL.  dz: out, same addressing as z.  dh0/dc0: out [M,U] (may be NULL). */
int d2p_lstm_seq_bwd(int M, int U, int n_steps, const float* z, long z_row_stride, long z_t_stride,
                     const float* Wh, const float* c0, const int* lens,
                     const float* cs, const float* dhout, const float* dh_final,
                     const float* dc_final, float* dz, float* dh0, float* dc0,
                     void* ws, size_t ws_bytes, d2p_stream_t stream);

/* Several INDEPENDENT sequences (up to 3, same U) advanced together: launch j serves step j
 * of every forward sequence (step n_i-1-j of every backward sequence) that still has one, so
 * the action / perception / program decoders of the reference (models/model_full.py:497-599),
 * which do not depend on each other, share launches and CUs.  Results are identical to
 * calling d2p_lstm_seq_fwd / _bwd once per descriptor (that is also the fallback path).
 * Every descriptor carries its own workspace (>= d2p_lstm_ws_bytes(M, U)). */
typedef struct {
    int M, U, n_steps;
    float* z; long z_row_stride, z_t_stride;
    const float* Wh; const float* h0; const float* c0; const int* lens;
    float* hout; float* cs; float* h_final; float* c_final;
    void* ws; size_t ws_bytes;
    /* Optional -- a "direct" persistent launch, without the preparation launch (weight pack, initial-state pack,
     * flag reset) in front of it: `flags` = a device buffer of d2p_lstm_flag_words() 32-bit words that the caller
     * zeroes ONCE and then hands to this library only (one buffer per sequence slot and stream); `epoch` = a counter
     * the caller keeps per buffer, starting at 0 and raised by at least n_steps + 2 after every call that got the
     * buffer (wrap: zero the buffer again, restart at 0).  NULL: the preparation launch runs (needed under hipGraph
     * capture, where the epoch would be baked into the graph).  Ignored by the per-step back ends. */
    unsigned* flags; unsigned epoch;
    const float* wpack;   /* optional, with flags: the packed forward image of Wh (d2p_lstm_pack_weights, 4*U*U floats),
                           * kept up to date by the caller -- the kernel's prologue then reads contiguous fragments */
    /* optional, with flags and lens -- a length-sorted launch (as d2p_lstm_bwd_desc): the forward recurrence groups
     * rows of similar length into its row domains and runs each domain only for its longest row's steps; what the
     * skipped steps would have written (zeros in hout, the carried cell state in cs) is filled in, so every output
     * is bit-identical to the unsorted call. */
    const int* rowmap; const int* slab_steps;
} d2p_lstm_fwd_desc;
typedef struct {
    int M, U, n_steps;
    const float* z; long z_row_stride, z_t_stride;
    const float* Wh; const float* c0; const int* lens; const float* cs;
    const float* dhout; const float* dh_final; const float* dc_final;
    float* dz; float* dh0; float* dc0;
    void* ws; size_t ws_bytes;
    float* db;      /* optional [4U]: the cell's bias gradient = column sums of dz over all n_steps*M rows
                     * (needs z_t_stride == M * z_row_stride).  The persistent kernels produce it inside
                     * their launch (per-workgroup sums, folded by each column tile's last workgroup in a
                     * fixed order); the other back ends run d2p_colsum_f32 over dz behind the recurrence. */
    unsigned* flags; unsigned epoch;   /* as d2p_lstm_fwd_desc */
    const float* wpack;                /* ... the packed backward (Wh^T) image */
    /* optional, with flags -- a length-sorted launch: rowmap (DEVICE, M ints) lists the rows by decreasing length,
     * slab_steps (HOST, ceil(M/16) ints) holds the longest length among rows 16s .. 16s+15 of that order.  The kernel
     * then groups rows of similar length into its row domains and runs each domain only for its longest row's
     * steps (a masked step leaves nothing behind but zeros in dz, which are still written): results unchanged.
     * For a sequence without lens (a decoder whose loss masks the steps past a row's length, so that dhout is zero
     * there) the lengths are those of the loss mask. */
    const int* rowmap; const int* slab_steps;
} d2p_lstm_bwd_desc;
int d2p_lstm_seq_fwd_multi(int nseq, const d2p_lstm_fwd_desc* descs, d2p_stream_t stream);
int d2p_lstm_seq_bwd_multi(int nseq, const d2p_lstm_bwd_desc* descs, d2p_stream_t stream);

/* ---- K6: embedding gather with out-of-range -> 0, and its scatter-add gradient ------
 * Replaces tf.nn.embedding_lookup (models/model_full.py:294); the reference's <s> id is
 * token_dim+1, out of range for the [token_dim+1, E] table, which TF-GPU gathers as zeros
 * (models/model_full.py:288-291,448-450).
 * d2p_shift_tokens_tm builds the decoder input ids time-major:
 *   ids[t*R + r] = (t == 0) ? start_id : tokens[r*T + t-1]      (models/model_full.py:447-450) */
int d2p_shift_tokens_tm(int R, int T, const int* tokens, int start_id, int* ids, d2p_stream_t stream);
int d2p_embedding_gather_oob0(int n, int rows, int E, const int* ids, const float* table,
                              float* out, d2p_stream_t stream);
/* dtable[v,:] = sum_{i: ids[i]==v} dout[i,:]  (overwrites dtable).  Runs as a one-hot TN GEMM on
 * the MFMA pipe, split over the id list and combined in a fixed order (deterministic).
 * ws >= d2p_embedding_scatter_ws_bytes(n, rows, E). */
size_t d2p_embedding_scatter_ws_bytes(int n, int rows, int E);
int d2p_embedding_scatter_add_oob0(int n, int rows, int E, const int* ids, const float* dout,
                                   float* dtable, void* ws, size_t ws_bytes, d2p_stream_t stream);

/* ---- greedy decoding (evaluation path) -----------------------------------
 * Replaces BasicDecoder + GreedyEmbeddingHelper + dynamic_decode(maximum_iterations=L)
 * (models/model_full.py:424-435,465-490).  table_proj: [V+1, 4U] = embedding·Wx + b (the
 * caller computes it once with d2p_gemm_f32_nn; steps then only gather rows).  start_id is
 * token_dim (last table row), end_id the end token ('m)' = 3 for programs, A-1 for actions).
 * Outputs (time-major): logits [L, M, V] zero past the steps TF would have run, ids [L, M]
 * (argmax, first index on ties, 0 past the run), lengths [M] (1-based first end_id, L if none).
 * No host synchronisation inside.  ws >= d2p_greedy_ws_bytes(M, U, V). */
size_t d2p_greedy_ws_bytes(int M, int U, int V);
int d2p_greedy_decode(int M, int U, int V, int L, const float* table_proj, const float* Wh,
                      const float* proj, const float* h0, const float* c0, int start_id, int end_id,
                      float* logits, int* ids, int* lengths, void* ws, size_t ws_bytes,
                      d2p_stream_t stream);
/* ---- scheduled sampling step (models/model_full.py:59-67,414-423) -----------------------
 * seq2seq.ScheduledEmbeddingTrainingHelper.sample + next_inputs for one decoder step t:
 * next_ids[r] = Categorical(softmax(logits[r,:])) draw with probability *p_sample_dev, else
 * gt_next[r].  Noise: Gumbel-max over Philox4x32-10 keyed by rng_dev = {uint64 seed, uint64
 * step counter} (device memory, so a captured graph draws fresh noise per replay), t, row and
 * token.  sampled_flag (nullable): 1 where the draw was taken.  Parity with TF is statistical
 * only (different generator). */
int d2p_sched_sample(int M, int V, const float* logits, const int* gt_next, const float* p_sample_dev,
                     const void* rng_dev, int t, int* next_ids, int* sampled_flag, d2p_stream_t stream);
/* out[r] = argmax_v x[r*ld + v] (first index on ties, as tf.argmax) */
int d2p_argmax_rows(int rows, int V, const float* x, long ld, int* out, d2p_stream_t stream);

/* ---- K7: masked, count-normalised sequence cross-entropies --------------------------
 * Replaces Sequence_Loss (models/model_full.py:620-657): softmax_/sigmoid_cross_entropy_
 * with_logits, tf.sequence_mask and the sum(ce*mask)/sum(mask) normalisation.
 * logits: time-major [T, R, V] (row (t,r) at (t*R+r)*V).  labels: element (r,t,v) at
 * labels[r*lab_r_stride + t*lab_t_stride + v*lab_v_stride] (so both the reference's
 * [B,V,L] one-hot `program` and its [B,k,T,A] `a_h` are read in place).
 * lens: [R].  group(r) = r % G (G = k for action/per: one loss per demonstration index,
 * models/model_full.py:1018-1038; G = 1 for the program).  Rows with t >= n_steps have
 * logits treated as 0 (the reference zero-pads past max(len), models/model_full.py:476-484).
 * fwd: loss_num[g] = sum ce*mask, loss_den[g] = sum mask (fp32 outputs [G] each); two-stage
 *      deterministic reduction through ws (>= d2p_xent_ws_bytes(G)).
 * bwd: dlogits[(t,r),v] = scale/(G*den[g]) * mask * (softmax - labels)      (softmax)
 *                       = scale/(G*den[g]) * mask * (sigmoid - labels)/V    (sigmoid),
 *      written for t < n_steps only. */
size_t d2p_xent_ws_bytes(int G);
int d2p_softmax_xent_masked_fwd(int T, int R, int V, int G, int n_steps, const float* logits,
                                const float* labels, long lab_r_stride, long lab_t_stride,
                                long lab_v_stride, const int* lens, float* loss_num,
                                float* loss_den, void* ws, size_t ws_bytes, d2p_stream_t stream);
int d2p_softmax_xent_masked_bwd(int T, int R, int V, int G, int n_steps, const float* logits,
                                const float* labels, long lab_r_stride, long lab_t_stride,
                                long lab_v_stride, const int* lens, const float* loss_den,
                                float scale, float* dlogits, d2p_stream_t stream);
int d2p_sigmoid_xent_masked_fwd(int T, int R, int V, int G, int n_steps, const float* logits,
                                const float* labels, long lab_r_stride, long lab_t_stride,
                                long lab_v_stride, const int* lens, float* loss_num,
                                float* loss_den, void* ws, size_t ws_bytes, d2p_stream_t stream);
int d2p_sigmoid_xent_masked_bwd(int T, int R, int V, int G, int n_steps, const float* logits,
                                const float* labels, long lab_r_stride, long lab_t_stride,
                                long lab_v_stride, const int* lens, const float* loss_den,
                                float scale, float* dlogits, d2p_stream_t stream);
/* Loss backward of up to three decoders in ONE launch, through their output projections: per problem
 *   dlogits = d loss / d logits   (exactly d2p_softmax_ / d2p_sigmoid_xent_masked_bwd; first n_steps*R rows), and
 *   dhout   = dlogits . proj^T    (proj [U, V]: the Dense(use_bias=False) projection, models/model_full.py:463-464;
 *                                  dhout [n_steps*R, U] is what the backward recurrence reads)
 * -- the three loss-backward launches and the three K = V products of a training step as one.  V <= 64. */
typedef struct {
    int sigmoid;                  /* 0: softmax cross-entropy, 1: sigmoid cross-entropy (mean over V) */
    int R, V, G, n_steps, U;
    const float* logits; const float* labels; long label_rs, label_ts, label_vs;
    const int* lens; const float* den; float scale;
    float* dlogits; const float* proj; float* dhout;
    /* optional (round 4): hout [n_steps*R, U] != NULL -- the launch first computes the logits themselves, logits_out
     * [n_steps*R, V] = hout . proj (the Dense(use_bias=False) projection of models/model_full.py:463-464: three skinny
     * GEMM launches between the decoders' forward and backward recurrences otherwise), and differentiates those;
     * `logits` is then not read. */
    const float* hout; float* logits_out;
    /* optional (round 4): loss_part [ceil(n_steps*R / 16)][G] != NULL receives, per workgroup of 16 rows, the sums of the
     * rows' loss VALUES by loss group (exactly the terms d2p_*_xent_masked_fwd adds up): d2p_loss_from_partials turns
     * them into the loss -- a training step then needs no forward loss launches at all. */
    float* loss_part;
} d2p_xent_bwd_desc;
int d2p_xent_bwd_dhout_multi(int nprob, const d2p_xent_bwd_desc* descs, d2p_stream_t stream);
/* Round 6: the decoders' small gradient products in one launch for up to four decoders (descs: HOST array).
 * d2p_small_pair_products: with S [R <= 256, N4 = 4U] (the decoder's dz rows summed by input token / perception column),
 *   G1 [U, N4] = A^T S (A [R, U]: the embedding table or the perception rows' matrix H) -- the input half of the LSTM
 *   kernel's gradient -- and G2 [R, U] = S Wx^T (Wx [U, N4] row-major) -- the embedding gradient / the Q of
 *   d2p_per_fc_bn_bwd.  Replaces a d2p_gemm_f32_tn and a d2p_gemm_f32_nt call per decoder. */
typedef struct {
    int R, U, N4;
    const float* S; const float* A; const float* Wx; float* G1; float* G2;
} d2p_pair_products_desc;
int d2p_small_pair_products(int n, const d2p_pair_products_desc* descs, d2p_stream_t stream);
/* loss / term_losses / nums as d2p_loss_assemble writes them, from the loss_part arrays of d2p_xent_bwd_dhout_multi:
 * term j has groups[j] loss groups and nblocks[j] = ceil(n_steps_j * R_j / 16) partial rows at parts[j]; groups, nblocks,
 * parts: HOST arrays of n_terms (<= 3) entries, at most 64 groups in all; dens as for d2p_loss_assemble; nums may be NULL.
 * Sums in workgroup order (fixed): equal to the forward kernels' value up to the order of an fp32 sum. */
int d2p_loss_from_partials(int n_terms, const int* groups, const int* nblocks, const float* const* parts,
                           const float* dens, float* nums, float* loss, float* term_losses, d2p_stream_t stream);

/* loss[0] = sum_terms (1/G_j) * sum_g num_j[g]/den_j[g]   (models/model_full.py:932,1035-1038,
 * 1078-1079).  nums/dens: concatenated [G_0 + G_1 + ...]; groups: HOST array of n_terms ints. */
int d2p_loss_assemble(int n_terms, const int* groups, const float* nums, const float* dens,
                      float* loss, float* term_losses, d2p_stream_t stream);

/* ---- summarizer glue (models/model_full.py:333-362,380-404) -------------------------- */
/* mean over k: out[b,:] = mean_i x[b,i,:]; if bcast != NULL also bcast[b,i,:] = out[b,:]. */
int d2p_group_mean(int B, int k, int U, const float* x, float* out, float* bcast, d2p_stream_t stream);
/* dx[b,i,:] (=|+=) (dout[b,:] + sum_i dbcast[b,i,:]) / k   (dbcast may be NULL) */
int d2p_group_mean_bwd(int B, int k, int U, const float* dout, const float* dbcast, float* dx,
                       int accumulate, d2p_stream_t stream);
/* max over k with the winning index (ties: lowest), and its backward -- the 'maxpool' demo
 * aggregation of the synthesis baseline (models/baselines/model_synthesis.py:345-358) */
int d2p_group_max(int B, int k, int U, const float* x, float* out, int* arg, d2p_stream_t stream);
int d2p_group_max_bwd(int B, int k, int U, const float* dout, const int* arg, float* dx,
                      int accumulate, d2p_stream_t stream);
/* rn_pool first layer without materialising pairs:
 *   y[b,a,c,:] = lrelu(P[b,c,:] + Q[b,a,:] + bias)   with P = feat·W1[:U], Q = feat·W1[U:]
 * `scopes` summaries (h and c) may be stacked along B: programs [s*B/scopes, (s+1)*B/scopes) use
 * bias + s*bias_stride. */
int d2p_rn_pair_fwd(int B, int k, int U, const float* P, const float* Q, const float* bias,
                    int scopes, long bias_stride, float* y, d2p_stream_t stream);
/* dP[b,c,:] = sum_a dy[b,a,c,:], dQ[b,a,:] = sum_c dy[b,a,c,:]   (dy already wrt pre-activation) */
int d2p_rn_pair_bwd(int B, int k, int U, const float* dy, float* dP, float* dQ, d2p_stream_t stream);
/* out[b,:] = mean_{a,c} y[b,a,c,:] + base[b,:]  and its backward dy[b,a,c,:] = dout[b,:]/k^2 */
int d2p_pair_mean_fwd(int B, int kk, int U, const float* y, const float* base, float* out, d2p_stream_t stream);
int d2p_pair_mean_bwd(int B, int kk, int U, const float* dout, float* dy, d2p_stream_t stream);
/* small elementwise helpers: y (=|+=) a*x ; transposes used to re-lay host batches */
int d2p_axpy(size_t n, float a, const float* x, float* y, int accumulate, d2p_stream_t stream);
/* out[t, r, :] = in[r, t, :]  (R x T x C -> T x R x C) */
int d2p_transpose_rt(int R, int T, int C, const float* in, float* out, d2p_stream_t stream);
/* out[o, c, i] = c < C ? in[o, c, i] : 0 over [outer, Cp, inner] (unpad = 0), or the inverse
 * slice out[o, c, i] = in[o, c, i], c < C, over [outer, C, inner] (unpad = 1).  fp32 or uint8.
 * Brings 3-channel frames / conv1 weights to 4 channels for 16-byte (uint8x4) tap gathers. */
int d2p_pad_axis(long outer, int C, int Cp, int inner, const void* in, void* out, int is_u8,
                 int unpad, d2p_stream_t stream);
/* Measurement hook (tools/corun_probe.py): a launch of `blocks` x `threads` whose workgroups each write the 100 MHz
 * wall clock of their first instruction to out[block] (unsigned 64-bit); every wave holds about `regs` live VGPRs
 * (4 / 48 / 80 / 120 / 200) and the launch asks for lds_bytes of dynamic LDS -- shows when and where workgroups of a
 * second queue become resident beside a persistent recurrent launch.  src: >= 1024 floats, sink: >= threads floats. */
int d2p_probe_clock(int blocks, int threads, int regs, int lds_bytes, void* out, const float* src, float* sink,
                    d2p_stream_t stream);
/* zero logits rows (t, r) with t >= nsteps_g[r % G]  (per-demo dynamic padding,
 * models/model_full.py:476-484); nsteps_g[g] = min(T, max_{r%G==g} lens[r]) computed on device. */
int d2p_zero_past_group_steps(int T, int R, int V, int G, const int* lens, float* logits, d2p_stream_t stream);
/* Perception decoder input in factored form (Per_Encoder = fc + batch norm per demonstration index in front of the
 * decoder LSTM, models/model_full.py:308-316,573-599).  With A [rows, NCp] holding, for a row of demonstration
 * index g, per[row] in columns g*(P+1) .. g*(P+1)+P-1 and a 1 in column g*(P+1)+P, the batch-normed features are
 * pe = A . H: per_affine_rows writes H [NCp, U] from the fc weights, the batch-norm parameters and the batch
 * statistics; the decoder's input projection is then A . (H . Wx), its weight gradient H^T . (A^T dZ), and
 * per_fc_bn_bwd derives the fc / batch-norm gradients from Q = (A^T dZ) . Wx^T [NCp, U] and gram = A^T A
 * [NCp, NCp] -- no [rows, U] matrix is multiplied by Wx in either direction.  P <= 8. */
int d2p_per_affine_rows(int G, int P, int U, int NCp, const float* W, const float* b, const float* gamma,
                        const float* beta, const float* mean, const float* rstd, float* H, d2p_stream_t stream);
/* S [NCp, E] = A^T dz over the first `rows` rows, from the structure of A (d2p_per_affine_rows: row r holds
 * per[r, 0..P) in the columns g*(P+1) .. of its demonstration index g = r % G and a 1 at g*(P+1)+P): what
 * d2p_gemm_f32_tn(A, dz) computes, as a read of dz.  per [rows, P] (time-major rows, as dz), G <= 15, P <= 8,
 * rows % G == 0, E % 4 == 0; rows NC .. NCp of S are written as zeros.  Fixed summation order. */
size_t d2p_per_rows_tn_ws_bytes(int rows, int G, int NCp, int E);
int d2p_per_rows_tn(int rows, int G, int P, int NCp, int E, const float* per, const float* dz, float* S, void* ws,
                    size_t ws_bytes, d2p_stream_t stream);
/* z [rows, E] = A . HWx + bias over the first `rows` rows from the structure of A (as d2p_per_rows_tn): what
 * d2p_gemm_f32_nn(A, HWx, bias) computes -- the factored perception decoder's input projection -- as a write of z.
 * HWx [NCp, E], bias [E] or NULL, per [rows, P]; P <= 8, rows % G == 0, E % 4 == 0. */
int d2p_per_rows_nn(int rows, int G, int P, int NCp, int E, const float* per, const float* HWx, const float* bias,
                    float* z, d2p_stream_t stream);
/* Batch statistics of u = per . W + b per demonstration index (the Per_Encoder's fc + batch norm, models/model_full.py:
 * 383-398) from gram = A^T A alone (A as for d2p_per_affine_rows; its blocks hold per_g^T per_g and colsum(per_g)):
 * mean / rstd / var [G, U] (var biased, may be NULL), fp64 inside.  rows_per_group = rows of A per index. */
int d2p_per_fc_bn_stats(int G, int P, int U, int NCp, int rows_per_group, const float* W, const float* b,
                        const float* gram, float* mean, float* rstd, float* var, d2p_stream_t stream);
int d2p_per_fc_bn_bwd(int G, int P, int U, int NCp, int rows_per_group, const float* W, const float* b,
                      const float* gamma, const float* mean, const float* rstd, const float* Q, const float* gram,
                      float* dW, float* db, float* dgamma, float* dbeta, d2p_stream_t stream);

/* ---- K8: global-norm clip + Adam over one flat buffer -------------------------------
 * Replaces tf.contrib.layers.optimize_loss(clip_gradients=20.0, AdamOptimizer)
 * (trainer.py:102-109): clip_by_global_norm then Adam (beta1 .9, beta2 .999, eps 1e-8,
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t) supplied by the caller as lr_t).
 * d2p_l2norm_flat writes sum(g^2)*prescale^2 as fp64 to sumsq[0]; d2p_adam_clip_flat reads it
 * on device (no host sync): scale = prescale * clip / max(sqrt(sumsq), clip).
 * prescale = 1/world_size folds the data-parallel gradient average in. */
size_t d2p_l2norm_ws_bytes(size_t n);
int d2p_l2norm_flat(size_t n, const float* g, float prescale, double* sumsq,
                    void* ws, size_t ws_bytes, d2p_stream_t stream);
/* lr_t_dev: optional DEVICE pointer to one float; when non-NULL it overrides lr_t, so the
 * launch can sit inside a replayed hipGraph while the bias-corrected rate changes per step. */
int d2p_adam_clip_flat(size_t n, float* p, const float* g, float* m, float* v,
                       const double* sumsq, float prescale, float clip, float lr_t,
                       const float* lr_t_dev, float beta1, float beta2, float eps,
                       d2p_stream_t stream);
/* The guarded optimizer step (no reference counterpart: TF's session either runs a step or raises).
 * The persistent recurrent kernels report a hand-off they gave up in a device status word
 * (d2p_lstm_persist_error); the gradients of that step are then invalid.  The guarded form reads the
 * word ON THE DEVICE and SKIPS the update -- parameters and both moments untouched -- while it is set,
 * or while fail_slot[0] != 0 (NULL: not consulted).  fail_slot is one float the caller appends to the
 * buffer it all-reduces: d2p_step_status_publish writes 1.0 / 0.0 into it after backward, so after the
 * SUM every rank of a data-parallel job takes the same decision.  counters (device, two uint64,
 * caller-zeroed): [0] += 1 per applied step, [1] += 1 per skipped step; mirror (optional, two uint64 of
 * pinned HOST memory the device can write): receives both values after this step -- the host reads them
 * there once an event recorded behind this call has completed, resets the word, switches the recurrences
 * to the per-step kernels
 * (d2p_lstm_set_persistent(0)) and re-runs the skipped steps (demo2program_amd/trainer.py).
 * The batch-norm moving statistics are likewise left alone while the word is set. */
int d2p_step_status_publish(float* slot, d2p_stream_t stream);
int d2p_adam_clip_flat_guarded(size_t n, float* p, const float* g, float* m, float* v,
                               const double* sumsq, float prescale, float clip, float lr_t,
                               const float* lr_t_dev, float beta1, float beta2, float eps,
                               const float* fail_slot, unsigned long long* counters,
                               unsigned long long* mirror, d2p_stream_t stream);

/* ---- optional per-launch HIP-event timing (used by bench.py's roofline leg) -------------
 * When enabled, every GEMM / conv / LSTM-gate launch is bracketed by hipEvents recorded on
 * the launch stream.  key = family*8 + tag; families: 1 dense GEMM (work = FLOP),
 * 2 conv (FLOP), 3 LSTM gate fwd (algorithmic bytes), 4 LSTM gate bwd (bytes);
 * tag 1 = launched from inside the recurrence (d2p_lstm_seq_*), 0 otherwise.
 * d2p_prof_enable(on) clears all records.  d2p_prof_read synchronises on the recorded
 * events and returns launch count, summed milliseconds and summed work for one key
 * (HOST pointers).  Not hipGraph-capturable; off by default. */
int d2p_prof_enable(int on);
int d2p_prof_set_tag(int tag);
int d2p_prof_read(int key, int* count, double* total_ms, double* total_work);

#ifdef __cplusplus
}
#endif
#endif /* D2P_H */
