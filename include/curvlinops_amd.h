/*
 * curvlinops_amd.h -- C ABI of libclo_hip.so, the MI355X (gfx950) backend for the
 * curvature-matvec hot path of f-dangel/curvlinops.
 *
 * Every entry point takes raw DEVICE pointers (fp32 unless stated), explicit
 * sizes/strides in ELEMENTS, and a hipStream_t passed as void*.  The caller
 * (PyTorch, or any other host) owns all memory.  Return value: 0 on success,
 * a negative CLO_E* code otherwise; clo_last_error() gives the message for the
 * calling thread.  No global state besides that per-thread message.
 *
 * Each function names the reference (f-dangel/curvlinops) call site it replaces;
 * paths are relative to the reference repository root.
 */
#ifndef CURVLINOPS_AMD_H
#define CURVLINOPS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLO_OK 0
#define CLO_EINVAL -1   /* bad argument (shape / stride / enum)            */
#define CLO_EHIP -2     /* a HIP runtime call or kernel launch failed       */
#define CLO_ENOTPD -3   /* Cholesky: matrix not positive definite           */
#define CLO_EUNSUP -4   /* valid request this build does not implement      */
#define CLO_EASYNC -5   /* an EARLIER launch of this entry point timed out waiting for its own workgroups (GPU shared
                           with another process / CU mask): its results are invalid; the mode has been disabled on this
                           device and the call can simply be repeated (reported once)                                    */

/* Activation codes (elementwise nonlinearity after a Linear layer). */
#define CLO_ACT_IDENTITY 0
#define CLO_ACT_RELU 1
#define CLO_ACT_TANH 2
#define CLO_ACT_SIGMOID 3

/* Output-space curvature ("loss Hessian") kinds applied between J v and J^T. */
#define CLO_LOSS_MSE 0       /* w = s * u                         (ggn_utils.py:52-55) */
#define CLO_LOSS_CE 1        /* w = s * (p*u - p*(p.u))           (ggn_utils.py:56-75) */
#define CLO_LOSS_BCE 2       /* w = s * sig*(1-sig)*u             (ggn_utils.py:76-79) */
#define CLO_LOSS_RANK1 3     /* w = s * sum_m g_m*(g_m.u), g given per row (gradient_moments.py:48-87,
                                ggn.py:140-166: EF / MC pseudo-losses)                 */
/* Empirical Fisher with the per-sample loss gradient g_n computed IN the kernel from the prediction it already holds and
 * the TARGETS (aux, aux_rank = 1), as the reference re-derives it on every product (gradient_moments.py:48-87):
 * w = s * g (g.u).  aux: [N][C] floats for MSE / BCE, [N] class labels stored as floats for CE. */
#define CLO_LOSS_EF_MSE 4    /* g = 2 (f - y)                    */
#define CLO_LOSS_EF_CE 5     /* g = softmax(f) - onehot(label)   */
#define CLO_LOSS_EF_BCE 6    /* g = sigmoid(f) - y               */

int clo_version(void);
const char *clo_last_error(void);

/* Health of the modes that need a co-resident grid on device `dev`: bit 0 persistent MLP kernel, bit 1 persistent
 * tridiagonalisation panels, bit 2 stream-K GEMM schedule -- set = disabled after a timeout (see CLO_EASYNC); a pending,
 * not yet reported timeout is reported (and the mode disabled) by this call as well. */
int clo_persistent_status(int dev);
/* Bit mask (as above) of timeouts that have happened on `dev` and have NOT been reported yet: a peek for consumers that
 * have just synchronised with the device (the SciPy export copies the result to the host) -- nothing is cleared, the
 * affected entry point still returns CLO_EASYNC on its next call.  A launch that timed out also marks its own result
 * with NaN (mlp_mega.hip, gemm_v3.hip, sytrd.hip): invalid numbers are never plausible numbers. */
int clo_fault_pending(int dev);
/* Diagnostics for the fail-soft tests: the spin budget of every bounded wait (default 1 << 22 polls, ~seconds), and a
 * kernel that keeps `blocks` workgroups with `lds_bytes` of LDS each busy for `ticks` ticks of the 100 MHz wall clock. */
int clo_test_set_spin_limit(unsigned polls);
int clo_test_occupy(int blocks, int lds_bytes, long ticks, void *stream);

/* Optional per-kernel timing (HIP events on the launch stream) for the roofline leg of bench.py.
 * clo_prof_collect fills arrays of 8 entries indexed by tag (0 forward+JVP weight stream,
 * 1 loss / head backward, 2 backward data chain, 3 slab finish / head forward, 4 outer products):
 * summed milliseconds, launch counts, summed algorithmic bytes. */
int clo_prof_enable(int on);
int clo_prof_collect(double *ms, long *count, double *alg_bytes);


/* ------------------------------------------------------------------------- *
 * Dense fp32 GEMM on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32).
 *   C[b][m][n] = alpha * sum_k A_b(m,k) * B_b(k,n) + beta * C[b][m][n]
 * A_b(m,k) = A[b*sa_b + m*sa_m + k*sa_k], B_b(k,n) = B[b*sb_b + k*sb_k + n*sb_n],
 * C row-major with leading dimension ldc and batch stride sc_b.  Either stride of
 * an operand may be 1 (both layouts are loaded coalesced); arbitrary strides work
 * (slow path).  splitk > 1 splits K over grid.z: `ws` must then hold
 * batch*splitk*M*N floats and is reduced deterministically by a second kernel.
 * splitk == -1 asks for the stream-K schedule of the LDS-DMA engine (gemm_v3.hip: the (tile, k tile) units of
 * the whole problem in equal contiguous ranges, one per CU; split tiles are finished inside the kernel in a fixed
 * order): `ws` must then hold clo_gemm_streamk_ws_floats() floats.  Operands the engine cannot take run unsplit.
 * Replaces torch.einsum / @ in kronecker.py:141-171, eigh.py:84-105 and the
 * Linear-layer GEMMs that torch.func.jvp/vjp issue for ggn.py:61-71.
 * ------------------------------------------------------------------------- */
int clo_gemm_f32(int M, int N, int K, float alpha,
                 const float *A, long sa_m, long sa_k, long sa_b,
                 const float *B, long sb_k, long sb_n, long sb_b,
                 float beta, float *C, long ldc, long sc_b,
                 int batch, int splitk, float *ws, void *stream);

/* Suggested split-K factor for a (M,N,K,batch) problem (1 = none, -1 = stream-K, see clo_gemm_f32). */
int clo_gemm_suggest_splitk(int M, int N, int K, int batch);
/* Workspace (floats) of the stream-K schedule: one 128 x 128 partial accumulator and one flag per CU. */
long clo_gemm_streamk_ws_floats(void);
/* The same for the symmetric product of clo_syrk_accum_f32 / clo_im2col_syrk_accum_f32 (upper-triangular
 * tiles only: about twice the split of the full d x d product fills the chip). */
int clo_syrk_suggest_splitk(int d, long rows);

/* EKFAC eigenvalue correction (computers/ekfac_hooks.py:206-236, per-example-gradient
 * strategy without materialising the [batch, d_out, d_in] tensor):
 *   C[m][n] = beta*C[m][n] + alpha * sum_b ( sum_k A_b(m,k) B_b(k,n) )^2
 * Operand addressing as clo_gemm_f32.  `splits` partitions the batch range over grid.y
 * (`ws`: splits*M*N floats when splits > 1). */
int clo_gemm_sqsum_f32(int M, int N, int K, float alpha,
                       const float *A, long sa_m, long sa_k, long sa_b,
                       const float *B, long sb_k, long sb_n, long sb_b,
                       float beta, float *C, long ldc, int batch, int splits, float *ws,
                       void *stream);
int clo_gemm_sqsum_suggest_splits(int M, int N, int batch);

/* clo_gemm_f32 for a batch whose members of A and / or B lie at ARBITRARY addresses: A_ptrs / B_ptrs are host arrays of
 * `batch` <= 8 device pointers (NULL: that operand is strided from A / B as above); every member 16-byte aligned relative to
 * the first, all with the same strides.  Equal-shape Kronecker factors of different layers (kfac.py builds one block per
 * layer, block_diagonal.py loops over them) enter one batched launch where they lie -- no stacked copies to keep or refresh. */
int clo_gemm_ptrs_f32(int M, int N, int K, float alpha, const float *A, const float *const *A_ptrs, long sa_m, long sa_k,
                      long sa_b, const float *B, const float *const *B_ptrs, long sb_k, long sb_n, long sb_b, float beta,
                      float *C, long ldc, long sc_b, int batch, int splitk, float *ws, void *stream);

/* ------------------------------------------------------------------------- *
 * KFAC factor accumulation: C[d][d] = beta*C + alpha * X^T X for row-major
 * X[rows][ldx] (first d columns used).  If ones_col != 0 the matrix is treated
 * as [X | 1] (joint weight+bias, kfac_math.py:115-116; the ones are synthesised by the tile
 * loader, nothing is concatenated) and C is (d+1)x(d+1).
 * Only the upper block-triangle is computed on the MFMA pipe and mirrored.
 * `ws`: split-K workspace as for clo_gemm_f32 (may be NULL when splitk == 1).
 * Replaces einsum("b s i, b s j -> i j") in computers/kfac_hooks.py:350,390.
 * ------------------------------------------------------------------------- */
int clo_syrk_accum_f32(float *C, long ldc, const float *X, long rows, int d, long ldx,
                       int ones_col, float alpha, float beta,
                       int splitk, float *ws, void *stream);

/* Patch extraction for Conv2d input covariances (kfac_utils.py:78-121: unfold + transpose):
 * x [B][C][H][W] -> out [B*OH*OW][C*KH*KW], one launch for the whole mini-batch. */
int clo_im2col_f32(const float *x, float *out, int B, int C, int H, int W, int KH, int KW,
                   int SH, int SW, int PH, int PW, int DH, int DW, int OH, int OW, void *stream);

/* All small covariance products of one factor build in ONE launch (round 6; computers/kfac_hooks.py:335-393, one einsum per
 * layer and backpropagated vector): C_p = beta_p C_p + alpha_p [X_p | 1]^T [X_p | 1] for p < P <= clo_syrk_grouped_max_problems().
 * Every (problem, 64 x 64 upper tile, row chunk) is a work item of one grid; split-K partials are summed in-kernel by the last
 * arriver of a tile in chunk order (deterministic).  The C_p must be distinct matrices.  Pointer / size arrays are HOST arrays.
 * Workspace from clo_syrk_grouped_ws: `slab_floats` floats (uninitialised) and `counters` unsigned that must be ZERO on entry
 * (they are zero again on exit). */
int clo_syrk_grouped_max_problems(void);
int clo_syrk_grouped_ws(int P, const long *rows, const int *d, const int *ones_col, long *slab_floats, long *counters);
int clo_syrk_grouped_f32(int P, float *const *C, const long *ldc, const float *const *X, const long *rows, const int *d,
                         const long *ldx, const int *ones_col, const float *alpha, const float *beta, float *slab,
                         unsigned *counters, void *stream);

/* The two steps above in one: C = beta C + alpha [P | 1]^T [P | 1] with P = im2col(x) generated inside
 * the tile loader of the symmetric MFMA GEMM -- the [B*OH*OW][C*KH*KW] patch matrix is never written
 * (kfac_utils.py:78-121 + kfac_hooks.py:355-393; 604 MB for ResNet-18 layer1 at B = 4096).  Any patch
 * length (no alignment requirement).  ws: splitk * d * d floats if splitk > 1 (d = C*KH*KW + ones_col). */
int clo_im2col_syrk_accum_f32(float *C, long ldc, const float *x, int B, int Cc, int H, int W,
                              int KH, int KW, int SH, int SW, int PH, int PW, int DH, int DW,
                              int OH, int OW, int ones_col, float alpha, float beta,
                              int splitk, float *ws, void *stream);

/* Input covariance of a convolution from the PIXEL Gram matrix (round 5).  The patch product of
 * kfac_utils.py:78-121 + kfac_hooks.py:350 only multiplies pixels of the same sample, so
 *   A[(c1,t1)][(c2,t2)] = sum_{pos: both taps inside} Gam[(c1, pos+t1)][(c2, pos+t2)],  Gam = X^T X, X = x as [B][C*H*W]
 * -- for feature maps with (H W)^2 < OH OW (KH KW)^2 (every 3x3 layer of a CIFAR-sized ResNet) fewer flops than the patch
 * product and no patch matrix.  The caller computes Gam with clo_syrk_accum_f32 (alpha = 1) and, for joint weight + bias
 * factors, colsum[c*H*W + q] = sum_b x[b][c][q]; this entry folds them: C = beta C + alpha A, C [(C KH KW + ones)^2].
 * clo_patch_fold_supported: the LDS tile of one channel pair fits (H W <= ~120). */
int clo_patch_fold_supported(int Cc, int H, int W, int KH, int KW, int OH, int OW);
int clo_patch_fold_f32(float *C, long ldc, const float *Gam, long ldg, const float *colsum, int B, int Cc, int H, int W,
                       int KH, int KW, int SH, int SW, int PH, int PW, int DH, int DW, int OH, int OW, int ones_col,
                       float alpha, float beta, void *stream);

/* Tall-skinny Gram matrix C = beta C + alpha [X | 1]^T [X | 1] for rows >> d, d + ones_col <= 128 (KFAC
 * factors of convolution layers: G_l with few output channels against B*H*W rows, A_1 with C_in k^2 + 1
 * columns; the Gram passes of the Hutch++ range basis).  X is streamed once, linearly; per-block
 * partial Grams are summed in a fixed order.  clo_gram_tall_supported: whether this beats the tiled
 * SYRK (rows >= 32 (d + ones), d + ones <= 128).  ws: clo_gram_tall_ws_floats floats. */
int clo_gram_tall_supported(long rows, int d, int ones_col);
long clo_gram_tall_ws_floats(long rows, int d, int ones_col);
int clo_gram_tall_f32(float *C, long ldc, const float *X, long rows, int d, long ldx, int ones_col,
                      float alpha, float beta, float *ws, void *stream);

/* Tall-skinny algebra of the randomised trace estimators (trace/meyer2020hutch.py:86-102: the QR of the [D, N] sketch and
 * the projections Q (Q^T G); trace/epperly2024xtrace.py:52-101), one streaming pass per call:
 *   clo_tall_gram_f64   out[n1][ldo] (FLOAT64) = X^T Y for float32 X [m][ldx] (n1 columns), Y [m][ldy] (n2 columns),
 *                       n1, n2 <= 64; exact products, float64 accumulation (f64 MFMA).  Y == NULL: the symmetric Gram
 *                       X^T X.  ws: clo_tall_gram_ws_bytes(m, n1, n2) bytes.
 *   clo_tall_apply_f32  out[m][ldo] = beta G + Q C with Q [m][ldq] (k columns, k % 4 == 0, 16-byte aligned rows),
 *                       C [k][ldc] (n columns), k, n <= 64; G may be NULL (beta ignored) or alias out. */
long clo_tall_gram_ws_bytes(long m, int n1, int n2);
int clo_tall_gram_f64(double *out, long ldo, const float *X, long ldx, int n1, const float *Y, long ldy, int n2,
                      long m, void *ws, void *stream);
int clo_tall_apply_f32(float *out, long ldo, const float *G, long ldg, float beta, const float *Q, long ldq,
                       const float *C, long ldc, long m, int k, int n, void *stream);

/* ------------------------------------------------------------------------- *
 * Damped Cholesky inverse of a Kronecker factor (kronecker.py:328-373): the blocked algorithm
 * (panel solve, trailing update, triangular inverse, L^-T L^-1) runs on clo_gemm_f32; this entry
 * point factors ONE nb x nb (nb <= 128; up to 64: one wave, 65 ... 128: four waves of one workgroup) diagonal
 * block in place (lower) and writes the inverse of its triangular factor to Linv.  *status (device int, caller-zeroed) receives pivot_base + k + 1
 * if pivot k is not positive.  Driver: curvlinops_amd/_hip.py:cholesky_inverse.
 * ------------------------------------------------------------------------- */
int clo_potrf_diag_f32(float *A, long lda, int nb, float *Linv, long ldinv, int *status,
                       int pivot_base, void *stream);
/* Whole inverse in one call: out = (A + damping I)^-1 by the recursive blocked algorithm
 *   L11,L11^-1 = rec(A11); L21 = A21 L11^-T; S22 -= L21 L21^T; L22,L22^-1 = rec(S22);
 *   (L^-1)21 = -L22^-1 (L21 L11^-1);   A^-1 = L^-T L^-1
 * (nodes of <= 128 rows in LDS, everything else GEMM/SYRK on the MFMA pipe).  A is not modified.
 * ws: clo_cholesky_inverse_ws_floats(n) floats; *status (device int) = 0 or the failing pivot. */
int clo_cholesky_inverse_f32(const float *A, long lda, float *out, long ldo, int n, float damping,
                             float *ws, int *status, void *stream);
long clo_cholesky_inverse_ws_floats(int n);
/* The same for `batch` factors of EQUAL size n in one chain of launches (leaves: one workgroup per
 * factor, products: batched GEMMs): the factors of repeated layer shapes (transformer blocks,
 * ResNet stages; reference kfac.py:224-271 builds one Kronecker block per layer) share the many
 * small launches of the recursion.  A, lda, out, ldo, damping are HOST arrays of length batch;
 * ws: clo_cholesky_inverse_batched_ws_floats(n, batch) floats; status: batch device ints. */
int clo_cholesky_inverse_batched_f32(const float *const *A, const long *lda, float *const *out,
                                     const long *ldo, int n, int batch, const float *damping,
                                     float *ws, int *status, void *stream);
long clo_cholesky_inverse_batched_ws_floats(int n, int batch);

/* Householder reduction of a symmetric matrix to tridiagonal form, the 85 % of the reference's
 * torch.linalg.eigh (kronecker.py:294-301 eigendecomposed Kronecker factors; ekfac.py's eigenbases) that
 * rocSOLVER runs as ~5 dependent kernels per column; here ONE persistent launch per 64-column panel (the
 * workgroups own their matrix rows for the whole panel and hand the reflector data to each other between
 * columns) plus one MFMA rank-128 update per panel.  A: row-major [n][lda] FULL symmetric matrix, 16-byte
 * aligned rows zero-padded to a multiple of 4 columns; overwritten.  On return, LAPACK ssytrd(uplo='L') storage
 * of the column-major (== row-major, symmetric) matrix: row j, columns j+2.. = Householder vector j (unit entry
 * at j+1 implied), D[n] / E[n-1] the tridiagonal matrix, tau[n-1] the reflector scales -- the inputs of a
 * LAPACK-compatible tridiagonal solver / back-transformation (clo_eigh_* below).  3 <= n <= 8184.
 * ws: clo_sytrd_ws_bytes(n) bytes, 16-byte aligned.
 * max_blocks: cap on the workgroups of a panel launch (0 = 128).  All of them must be resident at once (they
 * wait for each other); two fit a CU, so callers that run k reductions side by side on different streams pass
 * <= 512 / k. */
int clo_sytrd_f32(float *A, long lda, int n, float *D, float *E, float *tau, float *ws, long ws_bytes,
                  int max_blocks, void *stream);
long clo_sytrd_ws_bytes(int n);

/* ------------------------------------------------------------------------- *
 * MLP fast path (Sequential of Linear + elementwise activation), one mini-batch,
 * K = 1 column.  All activations are row-major [N][d].
 *
 * clo_mlp_fwd_jvp_layer: fused forward + forward-mode (JVP) pass through one
 * Linear layer, reading W and VW exactly once:
 *   z  = a_in W^T + b ;  dz = da_in W^T + a_in VW^T + Vb
 *   a_out = act(z) ; da_out = act'(z) * dz ; dphi_out = act'(z)
 * da_in may be NULL (first layer: tangent of the input is 0); b/Vb may be NULL.
 * VW/Vb/da_in/da_out may all be NULL for a pure forward pass.
 * Replaces jvp(f) in ggn.py:61 for Linear layers.
 * ------------------------------------------------------------------------- */
int clo_mlp_fwd_jvp_layer(const float *W, const float *b, const float *VW, const float *Vb,
                          const float *a_in, const float *da_in,
                          float *a_out, float *da_out, float *dphi_out,
                          int N, int d_in, int d_out, int act, float *ws, void *stream);
/* `ws`: clo_mlp_fwd_ws_floats(N, d_in, d_out) floats (split-K slabs of narrow layers). */
long clo_mlp_fwd_ws_floats(int N, int d_in, int d_out);

/* Output-space curvature product for one mini-batch (jvp(jacrev(c)) in
 * ggn.py:64-65):  w[n][:] = scale * H(f[n], .) u[n][:], then multiplied
 * elementwise by dphi_last (if not NULL).  `aux`: for CLO_LOSS_RANK1 the per-row
 * vectors g[n][m][C], m < aux_rank (H_n = sum_m g_nm g_nm^T); ignored otherwise. */
int clo_loss_hessian_apply(int kind, const float *f, const float *aux, int aux_rank,
                           const float *u, const float *dphi_last, float *w, int N, int C,
                           float scale, void *stream);

/* Backward (VJP) through one Linear layer (vjp(f) in ggn.py:68-71):
 *   out_W[j][i] = beta*out_W[j][i] + alpha * sum_n delta[n][j] * a_prev[n][i]
 *   out_b[j]    = beta*out_b[j]    + alpha * sum_n delta[n][j]          (if out_b)
 *   delta_prev[n][i] = dphi_prev[n][i] * sum_j W[j][i] * delta[n][j]    (if delta_prev)
 * `ws` must hold clo_mlp_bwd_ws_floats(N, d_in, d_out) floats when delta_prev != NULL. */
int clo_mlp_bwd_layer(const float *W, const float *delta, const float *a_prev,
                      const float *dphi_prev, float *out_W, float *out_b, float *delta_prev,
                      float alpha, float beta, int N, int d_in, int d_out,
                      float *ws, void *stream);
long clo_mlp_bwd_ws_floats(int N, int d_in, int d_out);

/* Whole-network GGN-type matvec for one mini-batch, K = 1 (ggn.py:41-72 with
 * _torch_base.py:937-942 accumulation):  out += / = alpha * J^T H J v.
 *   L            number of Linear layers
 *   dims[L+1]    d_0 .. d_L
 *   acts[L]      activation after each layer (CLO_ACT_*)
 *   W,b,VW,Vb,OW,Ob  host arrays of L device pointers (b/Vb/Ob entries may be NULL)
 *   X [N][d_0]   input batch;  loss_kind/aux/loss_scale as clo_loss_hessian_apply
 *   ws           workspace of clo_mlp_ggn_ws_floats(L, dims, N) floats
 *   flags        CLO_MLP_* bits below: the kernel choice is an argument, not process state
 * Works for any N: <= 8 rows in one persistent launch (three layers, narrow head, widths as in
 * mlp_mega.hip; needs clo_mlp_ggn_ws_init) or on the VALU/MFMA streaming chain, 9 ... 64 rows on its
 * all-MFMA variant (narrow head, float4-complete layers), otherwise on the MFMA GEMM engine. */
#define CLO_MLP_DEFAULT 0        /* the library picks: persistent launch where the shapes qualify            */
#define CLO_MLP_NO_PERSISTENT 1  /* keep the launch chain (a grid that needs every CU must not share the chip
                                    with a collective or another long-running kernel of the caller)          */
int clo_mlp_ggn_matvec(int L, const int *dims, const int *acts,
                       const float *const *W, const float *const *b,
                       const float *const *VW, const float *const *Vb,
                       float *const *OW, float *const *Ob,
                       const float *X, int N, int loss_kind, const float *aux, int aux_rank,
                       float loss_scale, float alpha, float beta, int flags,
                       float *ws, void *stream);
long clo_mlp_ggn_ws_floats(int L, const int *dims, int N);
/* Must run once on a freshly allocated workspace before its first clo_mlp_ggn_matvec (and again if other
 * code wrote into it): three-layer nets with a narrow head and <= 8 rows run as ONE persistent launch whose
 * workgroups synchronise through counters at the end of `ws`; this call zeroes them (hipMemsetAsync on
 * `stream`).  The kernel keeps them consistent from call to call by itself.  No-op for other shapes. */
int clo_mlp_ggn_ws_init(int L, const int *dims, int N, float *ws, void *stream);

/* G[n][c] = scale * d l_n / d f_n[c] of the MSE / CE / BCE loss at f = net(X) (plain forward pass on the GEMM engine, then the
 * per-sample loss gradient from the targets: [N][C] floats, CE: [N] class labels stored as floats).  This is the `G`
 * operand of clo_mlp_hessian_matvec / _matmat (scale = the reduction factor of the mini-batch loss), computed on the device
 * from the live parameters on every product, as the reference re-evaluates model and loss (hessian.py:13-69).
 * ws: clo_mlp_jac_ws_floats(L, dims, N) floats. */
int clo_mlp_loss_grad(int L, const int *dims, const int *acts, const float *const *W, const float *const *b,
                      const float *X, int N, int loss_kind, const float *targets, float scale, float *G,
                      float *ws, void *stream);

/* Jacobian and transposed-Jacobian products of an MLP (reference jacobian.py:14-358): the
 * forward+JVP half and the VJP half of the GGN product.
 *   clo_mlp_jvp: JV [N][d_L] = J v                             (any widths)
 *   clo_mlp_vjp: out = beta out + alpha J^T U,  U [N][d_L]     (any widths)
 * ws: clo_mlp_jac_ws_floats(L, dims, N) floats for either. */
long clo_mlp_jac_ws_floats(int L, const int *dims, int N);
int clo_mlp_jvp(int L, const int *dims, const int *acts, const float *const *W, const float *const *b,
                const float *const *VW, const float *const *Vb, const float *X, int N, float *JV,
                float *ws, void *stream);
int clo_mlp_vjp(int L, const int *dims, const int *acts, const float *const *W, const float *const *b,
                float *const *OW, float *const *Ob, const float *X, int N, const float *U, float alpha,
                float beta, float *ws, void *stream);

/* Exact Hessian-vector product of the mini-batch loss for an MLP (reference hessian.py:13-69:
 * jvp of the gradient), computed by the R-operator: tangent forward pass, then backpropagation of
 * the gradient signal AND its directional derivative; every product on the MFMA GEMM engine.
 *   G [N][C]   gradient of the (reduced) mini-batch loss w.r.t. the model output f
 *   loss_kind / aux / aux_rank / loss_scale: the loss Hessian w.r.t. f as in clo_mlp_ggn_matvec
 * out = beta * out + alpha * H v.  Any layer widths (dims[0..L-1] % 4 == 0 with 16-byte aligned operands
 * take the vectorised kernel variants).  ws: clo_mlp_hessian_ws_floats(L, dims, N) floats. */
long clo_mlp_hessian_ws_floats(int L, const int *dims, int N);
int clo_mlp_hessian_matvec(int L, const int *dims, const int *acts, const float *const *W,
                           const float *const *b, const float *const *VW, const float *const *Vb,
                           float *const *OW, float *const *Ob, const float *X, int N, const float *G,
                           int loss_kind, const float *aux, int aux_rank, float loss_scale, float alpha,
                           float beta, float *ws, void *stream);

/* K probe columns in one call (reference: vmap over the trailing K axis, _torch_base.py:946-989):
 *   out[.., k] = beta * out[.., k] + alpha * (J^T H J) V[.., k],   k = 0..K-1
 * VW[l] / OW[l] point at element (0, 0, 0) of a [d_out][d_in][K] block whose rows (one per weight)
 * are ldk floats apart -- the rows of the reference's [D, K] matrix; Vb[l] / Ob[l] likewise
 * [d_out][K].  The tangent weights are streamed once in that layout, W is shared by all columns.
 * Returns CLO_EUNSUP unless K % 4 == 0, 4 <= K <= 64, ldk % 4 == 0, dims[0..L-1] % 4 == 0, aux_rank
 * <= 16 and all operands are 16-byte aligned (the caller then loops clo_mlp_ggn_matvec over columns).
 * ws: clo_mlp_ggn_matmat_ws_floats(L, dims, N, K) floats. */
long clo_mlp_ggn_matmat_ws_floats(int L, const int *dims, int N, int K);
int clo_mlp_ggn_matmat(int L, const int *dims, const int *acts, const float *const *W,
                       const float *const *b, const float *const *VW, const float *const *Vb,
                       float *const *OW, float *const *Ob, long ldk, const float *X, int N, int K,
                       int loss_kind, const float *aux, int aux_rank, float loss_scale, float alpha,
                       float beta, float *ws, void *stream);
/* Exact Hessian for K columns (hessian.py:66 under the vmap over the trailing axis, _torch_base.py:946-989): the pipeline
 * of clo_mlp_ggn_matmat plus the R-operator terms (gradient signal d_l, d_l V_l, d_l^T da_{l-1}).  G [N][C] = gradient of
 * the reduced mini-batch loss w.r.t. the model output.  Requirements as clo_mlp_ggn_matmat plus ldk == K, a linear last
 * layer, loss_kind in {MSE, CE, BCE} (else CLO_EUNSUP).  ws: clo_mlp_hessian_matmat_ws_floats(L, dims, N, K) floats. */
long clo_mlp_hessian_matmat_ws_floats(int L, const int *dims, int N, int K);
int clo_mlp_hessian_matmat(int L, const int *dims, const int *acts, const float *const *W,
                           const float *const *b, const float *const *VW, const float *const *Vb,
                           float *const *OW, float *const *Ob, long ldk, const float *X, int N, int K,
                           const float *G, int loss_kind, float loss_scale, float alpha, float beta,
                           float *ws, void *stream);

/* ------------------------------------------------------------------------- *
 * Streaming helpers (HBM-bound).
 * ------------------------------------------------------------------------- */
/* y = beta*y + alpha*x  (batch accumulate, _torch_base.py:942) */
int clo_axpby_f32(float *y, const float *x, long n, float alpha, float beta, void *stream);
/* out[r][c] = in[c][r]  ([D,K] <-> [K,D] probe-layout conversion) */
int clo_transpose_f32(float *out, const float *in, long rows, long cols, void *stream);
/* Canonical pack / unpack of one joint (W, b) parameter group fused with the K-trailing <-> K-major layout change of
 * the Kronecker matvec (replaces kfac_utils.py:280-306 `cat` and :338-385 slicing plus one transpose each):
 * parameter side w [rows * cols_w][K], bias [rows][K] (K trailing, the reference's layout); canonical side
 * [K][rows][cols_w + 1] with the bias as the last column.  bias == NULL: plain [n][K] <-> [K][n]. */
int clo_canonical_pack_f32(float *out_kmajor, const float *w, const float *bias, long rows, long cols_w, long K,
                           void *stream);
int clo_canonical_unpack_f32(float *w, float *bias, const float *in_kmajor, long rows, long cols_w, long K,
                             void *stream);
/* y[i] = s[i] * x[i*K + k] for all k (EighDecomposed scaling, eigh.py:103-105) */
int clo_rowscale_f32(float *y, const float *x, const float *s, long rows, long K, int reciprocal,
                     float shift, void *stream);
/* Probe packing (sampling.py:6-56, trace/hutchinson.py:71-75): fill out[D][K]
 * (K trailing) with Rademacher (+-1, dist=0) or standard normal (dist=1) draws
 * from a counter-based Philox4x32-10 stream keyed by (seed, element index). */
int clo_pack_probes_f32(float *out, long D, long K, uint64_t seed, int dist, void *stream);

/* out[0] = scale * <x, y> over n fp32 elements (Frobenius inner product of two packed [D, K] blocks:
 * the reductions of trace/hutchinson.py:75 and trace/meyer2020hutch.py:95-102).  64-bit indexing
 * (D * K >= 2^31 is the C5 regime where BLAS-backed reductions refuse), double accumulation of the
 * block partials, deterministic.  ws: clo_dot_ws_bytes() bytes of device memory. */
long clo_dot_ws_bytes(void);
int clo_dot_f32(const float *x, const float *y, long n, float scale, float *out, void *ws, void *stream);

/* Fused vector updates of conjugate gradients for one right-hand side (the consumer of the matvec in
 * reference inverse.py:54-140); step sizes are formed on the device from the scalars the previous
 * kernels left there, so an iteration needs no host synchronisation:
 *   clo_cg_update_f32   : a = rz / pap;  x += a p;  r -= a ap;  rr_out[0] = <r, r>   (ws as clo_dot)
 *   clo_cg_direction_f32: p = z + (num / den) p */
int clo_cg_update_f32(float *x, float *r, const float *p, const float *ap, long n, const float *rz,
                      const float *pap, float *rr_out, void *ws, void *stream);
int clo_cg_direction_f32(float *p, const float *z, long n, const float *num, const float *den,
                         void *stream);

/* ---- symmetric eigensolver behind clo_sytrd_f32 (csrc/eigh.hip; replaces rocSOLVER sstedc / sormtr behind
 * torch.linalg.eigh at computers/_base.py:355-372, kronecker.py:292-300): the building blocks that clo_eigh_f32
 * composes (rounds 2-3 drove them level by level from curvlinops_amd/eigh_native.py); every O(n^3) product is a
 * clo_gemm_f32.
 *   clo_larft_f32       : T factors (nb x nb, upper triangular) of `np` block reflectors from G = V^T V and tau
 *   clo_tql2_batched_f32: eigen-decomposition of `batch` symmetric tridiagonal matrices of order L <= 64
 *                         (implicit QL in float64, one wave each): lam ascending, eigenvectors in columns
 *   clo_dc_*            : one merge level of Cuppen's divide & conquer, batched over `nodes` of size s:
 *                         deflation scan (in place on D, z), secular roots (float64 bisection on the shifted
 *                         variable) + Gu-Eisenstat weights, eigenvector matrix MT [nodes][s][s] (transposed),
 *                         Givens rotations of the deflation. */
/* ---- the symmetric eigensolver in ONE call (csrc/eigh_driver.hip; replaces torch.linalg.eigh = rocSOLVER ssyevd at
 * computers/_base.py:355-372 and kronecker.py:292-300): clo_sytrd_f32 -> tridiagonal divide & conquer (every level of
 * the tree walked in C++, device-side sorts and gathers, no host synchronisation) -> block-reflector
 * back-transformation.  `batch` matrices of ONE order 1 <= n <= 8184:
 *   A   [batch][n][lda]  full symmetric fp32, 16-byte aligned rows with ZERO padding columns up to lda (>= pad4(n),
 *                        multiple of 4), batch stride strideA floats; OVERWRITTEN (reflectors of the reduction).
 *                        Normalise to max |A| = 1 first (the deflation tolerances are relative to the matrix scale).
 *   lam [batch][ld_lam]  eigenvalues, ascending (NaN if a leaf of the tridiagonal solver did not converge)
 *   Z   [batch][n][ldz]  eigenvectors in the ROWS (Z^T = torch.linalg.eigh(...).eigenvectors); ldz >= pad4(n),
 *                        multiple of 4; padding columns zero on return; batch stride strideZ floats
 *   ws  clo_eigh_ws_bytes(n, batch) bytes, 16-byte aligned;  max_blocks: as clo_sytrd_f32.
 * clo_stedc_f32: the tridiagonal stage alone (d[b][ldd] diagonal, e[b][ldd] sub-diagonal, n - 1 entries). */
long clo_eigh_ws_bytes(int n, int batch);
int clo_eigh_f32(float *A, long lda, int n, float *lam, float *Z, long ldz, void *ws, long ws_bytes,
                 int max_blocks, void *stream);
int clo_eigh_batched_f32(float *A, long lda, long strideA, int n, int batch, float *lam, long ld_lam, float *Z,
                         long ldz, long strideZ, void *ws, long ws_bytes, int max_blocks, void *stream);
long clo_stedc_ws_bytes(int n, int batch);
int clo_stedc_f32(const float *d, const float *e, long ldd, int n, int batch, float *lam, long ld_lam, float *Z,
                  long ldz, long strideZ, void *ws, long ws_bytes, void *stream);
int clo_larft_f32(const float *G, const float *tau, float *T, int np, int nb, void *stream);
/* ---- Kronecker-factored blocks in one call (csrc/kron.hip).  Replace the einsum over the factors at
 * kronecker.py:141-171 (KroneckerProductLinearOperator._matmat / _adjoint_matmat), the eigen-decomposed product at
 * eigh.py:84-105 with a Kronecker eigenbasis (EKFAC blocks, ekfac.py), and the loop over the blocks of a block-diagonal
 * KFAC / EKFAC operator (block_diagonal.py; kfac.py / ekfac.py build one block per layer).
 * Operands are K-major: X [K][a*b] -- column k is a row-major [a, b] matrix (for K == 1 the flat vector) --, Y [K][A*B]
 * likewise; factors are row-major S1 [A][ld1 >= a], S2 [B][ld2 >= b].
 *   clo_kron_matmat        : Y_k = E1 X_k E2^T with E_i = S_i, or S_i^T where bit i-1 of `trans` is set (trans = 3:
 *                            Y_k = S1^T X_k S2, X [K][A*B], Y [K][a*b]) -- adjoint operators hold transposed arrays
 *   clo_eigh_apply         : Y_k = Q1 (lam .* (Q1^T X_k Q2)) Q2^T with Q1 [n1][n1], Q2 [n2][n2], lam [n1*n2]; bit i-1 of
 *                            `rows`: array i holds its eigenvectors in the ROWS (as clo_eigh_f32 returns them)
 *   clo_kron_matmat_blocks : `nblocks` independent blocks (HOST arrays of device pointers / extents); lam may be NULL,
 *                            lam[i] != NULL marks block i as eigen-decomposed (S1 = Q1, S2 = Q2 square); trans[i] (may be
 *                            NULL = 0) are block i's flags as above
 *   ws: the largest clo_kron_ws_floats(A, a, B, b, K, eig) over the blocks, 16-byte aligned. */
/* EKFAC eigenvalue correction of ONE layer in one call (replaces computers/ekfac_hooks.py:25-238, both strategies):
 *   lam[i][j] = beta lam[i][j] + alpha sum_{v < V, n < B} ( Qg^T ( sum_{s < S} g[v][n][s][:] a[n][s][:]^T ) Qa )[i][j]^2
 * g [V][B][S][d_out] (V backpropagated vectors), a [B][S][d_in] (S weight-sharing positions), both contiguous;
 * Qg [d_out][ldg], Qa [d_in][lda] eigenvector arrays, bit 0 / 1 of `rows`: Qg / Qa hold their eigenvectors in the ROWS;
 * lam [d_out][ld_lam].  ws: clo_ekfac_correction_ws_floats(...) floats, 16-byte aligned. */
long clo_ekfac_correction_ws_floats(int V, int B, int S, int d_out, int d_in);
int clo_ekfac_correction_f32(float *lam, long ld_lam, const float *Qg, long ldg, const float *Qa, long lda, int rows,
                             const float *g, const float *a, int V, int B, int S, int d_out, int d_in, float alpha,
                             float beta, float *ws, long ws_floats, void *stream);
long clo_kron_ws_floats(int A, int a, int B, int b, int K, int eig);
int clo_kron_matmat(float *Y, const float *S1, long ld1, const float *S2, long ld2, const float *X, int A, int a, int B,
                    int b, int K, int trans, float *ws, long ws_floats, void *stream);
int clo_eigh_apply(float *Y, const float *Q1, long ld1, const float *Q2, long ld2, const float *lam, const float *X, int n1,
                   int n2, int K, int rows, float *ws, long ws_floats, void *stream);
int clo_kron_matmat_blocks(int nblocks, float *const *Y, const float *const *S1, const long *ld1, const float *const *S2,
                           const long *ld2, const float *const *lam, const float *const *X, const int *A, const int *a,
                           const int *B, const int *b, const int *trans, int K, float *ws, long ws_floats, void *stream);

/* Back-transformation Z <- Z Q^T: every ROW of Z [m][ldz] (ldz >= pad4(n), multiple of 4, 16-byte aligned) is
 * multiplied by Q = H_0 ... H_{n-2}, the reflectors clo_sytrd_f32 left in the rows of `work` [n][ldw] and `tau`
 * (LAPACK sormtr, side = left on the column-major eigenvector matrix).  ws: clo_ormtr_ws_floats(m, n) floats. */
long clo_ormtr_ws_floats(int m, int n);
int clo_ormtr_f32(const float *work, long ldw, const float *tau, float *Z, long ldz, int m, int n, float *ws,
                  long ws_floats, void *stream);
int clo_tql2_batched_f32(const float *d, const float *e, float *lam, float *Q, int L, int batch, int *status,
                         void *stream);
int clo_dc_deflate(double *D, double *z, const double *rho, int *type, int *rot_p, double *rot_c, double *rot_s,
                   int *K, int s, int nodes, double eps, void *stream);
int clo_dc_secular(const double *dk, const double *zk, const double *rho, const int *K, int *org, double *mu,
                   double *zh, int s, int nodes, int kmax, void *stream);
int clo_dc_build(const double *dk, const int *K, const int *org, const double *mu, const double *zh,
                 const int *spos, float *MT, int s, int nodes, int kmax, void *stream);
int clo_dc_rotate(float *MT, const int *rot_p, const double *rot_c, const double *rot_s, int s, int nodes,
                  void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CURVLINOPS_AMD_H */
