/* stemgnn_hip.h -- C ABI of libstemgnn_hip.so: the MI355X (gfx950) implementation of StemGNN's
 * spectral hot path (latent-correlation attention -> Laplacian -> Chebyshev/eigen basis -> GFT ->
 * DFT/GLU/iDFT spe_seq_cell -> IGFT + forecast/backcast heads), forward and backward.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major fp32 unless the name ends in _host;
 *   - all buffers (inputs, outputs, saved activations, workspaces) are owned by the caller; the
 *     library never allocates, frees or synchronises; every launch goes to `stream` (a hipStream_t
 *     passed as void*), so the calls are stream-ordered and graph-capturable;
 *   - return value: 0 on success, a negative hipError_t (-(int)err) on a launch failure, or
 *     SG_EINVAL (-10001) for a bad argument.  No exception crosses this boundary.
 *   - shapes: B batch, N nodes (units), W window (time_step), multi, H horizon, Wm = W*multi,
 *     M = B*N rows.
 *
 * Each entry point names the reference code it replaces (paths relative to microsoft/StemGNN).
 */
#ifndef STEMGNN_HIP_H
#define STEMGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG_EINVAL (-10001)

/* number of per-StockBlock parameter tensors, in state_dict order of one block:
 * 0 weight[1,4,1,Wm,Wm]; 1,2 forecast.{w,b}; 3,4 forecast_result.{w,b}; 5,6 backcast.{w,b} (NULL for
 * stack_cnt>0); 7,8 backcast_short_cut.{w,b}; then GLUs.g.linear_left.{w,b}, GLUs.g.linear_right.{w,b}
 * for g=0..5 (9 + 4*g + {0,1,2,3}).   (models/base_model.py:23-44) */
#define SG_BLOCK_NPARAMS 33

/* library / build identification: returns a static string "stemgnn_hip <version> gfx950". */
const char* stemgnn_version(void);
/* compute units of the current HIP device (256 on MI355X): the launch-sizing formulas use it, never a constant */
int stemgnn_num_cus(void);

/* ---- sizes (in floats) of the caller-allocated buffers ----------------------------------------- */
size_t stemgnn_table_floats(int W, int multi);                       /* DFT / C2R tables            */
size_t stemgnn_packed_floats(int W, int multi);                      /* packed weights of one block */
size_t stemgnn_saved_floats(int B, int N, int W, int multi);         /* saved activations, one block fwd */
size_t stemgnn_scratch_floats(int B, int N, int W, int multi);       /* backward scratch, one block */
size_t stemgnn_gradpart_floats(int W, int multi, int nsplit);        /* split-M weight-grad partials */
size_t stemgnn_attn_saved_floats(int B, int N);                      /* key,query,rowsum,A,deg      */
size_t stemgnn_attn_scratch_floats(int B, int N, int nchunk);        /* attention backward scratch  */
size_t stemgnn_scratch_offset_dG(int B, int N, int W, int multi);    /* float offset of dG [M,3W] inside the backward scratch */

/* Fill the constant DFT tables (double-precision trig rounded to fp32) into a HOST buffer of
 * stemgnn_table_floats() floats; the caller copies it to the device once per (W, multi).
 * Replaces the library FFT plans behind torch.rfft/irfft (models/base_model.py:49,58). */
int stemgnn_make_tables_host(int W, int multi, float* tables_host);

/* ---- latent-correlation attention + Laplacian  (models/base_model.py:139-147, 151-162) --------
 * h        [N, B, N]  GRU output exactly as nn.GRU returns it (seq-major; :137), h[s,b,i]
 * wk, wq   [N]        weight_key / weight_query
 * seed     device uint64[2] = {seed, offset} for the dropout Philox stream (ignored if !training or p==0)
 * saved    stemgnn_attn_saved_floats(): key[B,N] | query[B,N] | rowsum[B,N] | A[N,N] (batch-mean, un-symmetrised) | deg[N]
 * attention_out [N,N] = 0.5(A+A^T)  (the tensor Model.forward returns)
 * mul_L    [4,N,N]: slot 0 := 0 (T0 is zeros, :129) and slot 1 := L are written here
 * parts    bit 0: attention (-> A | deg, contiguous N*N + N floats at saved + 3*B*N), bit 1: Laplacian from (A, deg);
 *          3 = both.  The batch mean (:140) is the ONE cross-sample reduction of the path: a data-parallel caller that
 *          wants single-process semantics for a split batch runs part 1, averages A | deg over the ranks, runs part 2.
 */
int stemgnn_attn_laplacian_fwd(const float* h, const float* wk, const float* wq, float alpha,
                               float drop_p, int training, const uint64_t* seed,
                               int B, int N, float* saved, float* attention_out, float* mul_L,
                               int parts, void* stream);
/* dL [N,N] = gradient w.r.t. mul_L slot 1 (total).  Outputs dh [N,B,N], dwk [N], dwq [N].
 * scratch: stemgnn_attn_scratch_floats(B,N,nchunk).  parts bit 0: Laplacian backward -> dA / B in scratch[0 .. N*N)
 * (averaged over the ranks by an exact-mode data-parallel caller), bit 1: softmax / key / query backward; 3 = both.
 * parts bit 2 (with bit 1): FACTORED output -- stop at dkey | dquery ([B,N] each, left in scratch behind the [N,N] dA:
 * scratch + N*N and scratch + N*N + B*N); dh / dwk / dwq are not touched (may be NULL).  In the model
 * dh[s][b][i] = dkey[b][i] wk[s] + dquery[b][i] wq[s] (models/base_model.py:154-155): stemgnn_gru_bwd_rank2 consumes the two
 * factors directly, stemgnn_keyquery_wgrad forms dwk / dwq from them off the critical chain. */
int stemgnn_attn_laplacian_bwd(const float* dL, const float* h, const float* wk, const float* wq,
                               float alpha, float drop_p, int training, const uint64_t* seed,
                               int B, int N, const float* saved, float* scratch, int nchunk,
                               float* dh, float* dwk, float* dwq, int parts, void* stream);
/* test hook: write the 0/1 keep-mask [B,N,N] the kernels above generate for `seed`. */
int stemgnn_dropout_mask(float drop_p, const uint64_t* seed, int B, int N, float* mask, void* stream);
/* One step of a model's dropout stream (nn.Dropout draws a fresh mask per forward, models/base_model.py:101,142): used[0..1] :=
 * seed[0..1] (the {key, offset} the coming forward / backward pair reads), then seed[1] += 1 -- one launch, graph-capturable. */
int stemgnn_dropout_seed_next(uint64_t* seed, uint64_t* used, void* stream);

/* ---- Chebyshev basis  (models/base_model.py:121-134) --------------------------------------------
 * in: mul_L slot 1 = L;  out: slot 2 = 2LL, slot 3 = 2L(2LL) - L  (fp32 MFMA GEMMs). */
int stemgnn_cheb_fwd(float* mul_L, int N, void* stream);
/* dmul_L [4,N,N] gradient of all four slots (slot 0 ignored) -> dL [N,N]; scratch 2*N*N floats. */
int stemgnn_cheb_bwd(const float* mul_L, const float* dmul_L, float* dL, float* scratch, int N, void* stream);

/* ---- Laplacian eigendecomposition route (north-star; same function of L as stemgnn_cheb_fwd) --------
 * Symmetric N x N eigensolver L = U^T diag(lam) U, then slot k := sum_e p_k(lam_e) u_e u_e^T for k = 2,3 with
 * p = (2 l^2, 4 l^3 - l) (slot 0 = zeros and slot 1 = L are left as attn_laplacian_fwd wrote them).
 *   nsweeps <= 0 : direct solver, 7 launches, 3 <= N <= 2048: Householder tridiagonalisation as ONE persistent cluster
 *                  kernel (pending rank-2 update fused with the next symmetric mat-vec, one grid barrier per column) ->
 *                  multisection (Sturm counts, one wave per eigenvalue: 3 fp32 + 6 fp64 rounds of 65 points) -> fp64 inverse iteration (pivoted
 *                  tridiagonal LU) -> cluster pass (eigenvalues closer than 1e-9 |T|, repeated ones included, are
 *                  re-orthogonalised and re-iterated as LAPACK dstein does) -> reflector back-transform -> rebuild on
 *                  the MFMA GEMM core;
 *   nsweeps  > 0 : parallel one-sided Jacobi (one launch per tournament round, fp64 rotation parameters), one
 *                  Newton-Schulz re-orthogonalisation and Rayleigh quotients on MFMA.
 * lam [N] (ascending for the direct solver); U [N,N] with the eigenvectors in ROWS; scratch:
 * stemgnn_eigh_scratch_floats(N) (16-byte aligned).  stemgnn_eigh_status(): host-synchronous read-and-clear of the
 * solver's status word of the CURRENT device (0 ok, 2 = a grid-barrier wait of the tridiagonalisation timed out, 3 = a cluster
 * re-solve broke down).  stemgnn_eigh_cluster_fixes(): host-synchronous read-and-clear of the number of eigenvectors the
 * cluster pass re-orthogonalised on the current device (diagnostic; 0 for a spectrum without clusters). */
size_t stemgnn_eigh_scratch_floats(int N);
int stemgnn_eigh_fwd(float* mul_L, float* lam, float* U, float* scratch, int N, int nsweeps, void* stream);
/* The batched form north_star names (round 5; direct solver): `batch` matrices in ONE call, matrix m at mul_L + m 4 N^2
 * (slot 1 = L_m; slots 2 / 3 receive its rebuilt basis), lam + m N, U + m N^2, scratch + m stemgnn_eigh_scratch_floats(N).
 * N <= 256: every stage takes the batch as a grid dimension -- 7 launches whatever the batch, one workgroup per matrix in the
 * tridiagonalisation -- so 8 Laplacians of the PEMS07 size cost what one does; N > 256: the multi-workgroup
 * tridiagonalisation runs matrix by matrix, the other stages batched.  Results per matrix are bitwise those of stemgnn_eigh_fwd. */
int stemgnn_eigh_batched(float* mul_L, float* lam, float* U, float* scratch, int N, int batch, void* stream);
int stemgnn_eigh_status(void);
int stemgnn_eigh_cluster_fixes(void);

/* ---- split-bf16 GLU GEMM (experiment; BASELINE configs[1] "bf16/fp32", SURVEY 8b `_bf16` entry points) -----------
 * C[M,N] = A[M,K] B[N,K]^T -- the shape of one GLU layer (models/base_model.py:12-13, x W^T) -- with every fp32 operand
 * taken as the sum of `splits` bf16 numbers and the cross products evaluated on the bf16 MFMA pipe with fp32
 * accumulation: splits = 3 -> 6 products (~2^-24 relative, fp32 class), 2 -> 3 products (~2^-16), 1 -> plain bf16.
 * B is pre-split by stemgnn_split_weights_bf16 into `planes` (stemgnn_split_planes_floats floats, 16-byte aligned);
 * A is split on the fly.  K % 4 == 0.  stemgnn_glu_gemm_f32 is the same product on the exact-fp32 MFMA core. */
size_t stemgnn_split_planes_floats(int N, int K, int splits);
int stemgnn_split_weights_bf16(const float* B, int N, int K, int splits, void* planes, void* stream);
int stemgnn_glu_gemm_bf16(const float* A, const void* planes, float* C, int M, int N, int K, int splits, void* stream);
int stemgnn_glu_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, void* stream);

/* ---- stand-alone GLU (models/base_model.py:6-13) and the general fp32 GEMM it is composed from ---------------------
 * stemgnn_sgemm_f32: C[M,N] (+)= A B on the exact-fp32 MFMA core; element (i,k) of A at A[i*lda+k] (a_kcontig) or
 * A[k*lda+i]; element (k,j) of B at B[j*ldb+k] (b_kcontig) or B[k*ldb+j]; accumulate != 0 adds to C.
 * stemgnn_glu_combine_fwd: out = (U + bl) * sigmoid(V + br), saving gate = sigmoid(.) and lin = U + bl ([M,C] each);
 * stemgnn_glu_combine_bwd: dU = dout * gate, dV = dout * lin * gate * (1 - gate);  stemgnn_colsum: out[c] = sum_m X[m][c]
 * in a fixed order (bias gradients). */
int stemgnn_sgemm_f32(const float* A, int lda, int a_kcontig, const float* B, int ldb, int b_kcontig, float* C, int ldc,
                      int M, int N, int K, int accumulate, void* stream);
int stemgnn_glu_combine_fwd(const float* U, const float* V, const float* bl, const float* br, float* out, float* gate,
                            float* lin, int M, int C, void* stream);
int stemgnn_glu_combine_bwd(const float* dout, const float* lin, const float* gate, float* dU, float* dV, int M, int C,
                            void* stream);
int stemgnn_colsum(const float* X, int M, int C, float* out, void* stream);

/* ---- GRU front (models/base_model.py:92,137: nn.GRU(time_step, units) over the node axis) ------------
 * seq_len S (= N nodes), batch B, input size W, hidden size Hd (= N).  PyTorch gate order (r,z,n).
 * x [B,W,S] is the model input read in place (x_s[b,t] = x[b,t,s]); w_ih [3Hd,W], w_hh [3Hd,Hd],
 * b_ih, b_hh [3Hd]; h_ext [S+1,B,Hd]: slab 0 is set to zero (h_{-1}) and slabs 1..S are exactly nn.GRU's output
 * (so h_ext + B*Hd is the [S,B,Hd] tensor, and h_ext itself is "h of the previous step" row for row -- the
 * operand of the dW_hh GEMM); reserve keeps r,z,n,gh_n for backward.
 * Recurrence: P = 1..8 persistent workgroups per batch row keep their slice of w_hh resident in registers
 * for all S steps and exchange h (forward) / the gate gradients (backward) once per step through tagged
 * 8-byte granules (bounded spins; a timeout sets *status = 1, a device int the caller owns and zeroes);
 * hidden sizes beyond 512 run on ONE cluster of <= 224 workgroups that holds w_hh as MFMA A operands for up to 16
 * batch rows at a time (csrc/gru_wide.h; flags + write-through stores instead of granules). */
size_t stemgnn_gru_reserve_floats(int B, int S, int Hd);
size_t stemgnn_gru_fwd_scratch_floats(int B, int S, int Hd);
size_t stemgnn_gru_bwd_scratch_floats(int B, int S, int Hd, int W);
/* CUs (one workgroup each) the backward recurrence occupies for its whole run at this shape: a caller that overlaps other
 * kernels with it on another stream sizes them for the remaining CUs. */
int stemgnn_gru_bwd_cus(int B, int Hd);
int stemgnn_gru_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                    int B, int S, int Hd, int W, float* scratch, float* h_ext, float* reserve, int* status,
                    void* stream);
/* dh_all [S,B,Hd] = gradient of every output step -> dw_ih, dw_hh, db_ih, db_hh (x gets no gradient); everything is
 * ordered on `stream` (the side-stream schedules of round 2 were measured slower and removed in round 4). */
int stemgnn_gru_bwd(const float* dh_all, const float* x, const float* w_hh, const float* h_ext,
                    const float* reserve, int B, int S, int Hd, int W, float* scratch,
                    float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, int* status, void* stream);

/* dwk[s] = sum_{b,i} dkey[b,i] h[s,b,i] (dwq likewise) from the factors a parts | 4 call of stemgnn_attn_laplacian_bwd left
 * in `attn_scratch`. */
int stemgnn_keyquery_wgrad(const float* h, const float* attn_scratch, float* dwk, float* dwq, int B, int N, void* stream);
/* parts bit 3 (value 8, with bit 2) of stemgnn_attn_laplacian_bwd leaves dquery as per-chunk partials; this sums them (same fixed
 * order) into `out` [B, N], or into the scratch's own dquery slot when out == NULL.  stemgnn_keyquery_wgrad2: the key / query
 * weight gradients from explicit dkey / dquery buffers. */
int stemgnn_attn_dquery_reduce(float* attn_scratch, int B, int N, int nchunk, float* out, void* stream);
int stemgnn_keyquery_wgrad2(const float* h, const float* dkey, const float* dquery, float* dwk, float* dwq, int B, int N,
                            void* stream);
/* stemgnn_gru_bwd with the output gradient given as the rank-2 form dh[s][b][i] = dkey[b][i] wk[s] + dquery[b][i] wq[s]
 * (dkey, dquery [B,Hd]; wk, wq [S]); only where stemgnn_gru_bwd_rank2_ok(B, Hd) (the wave-specialised per-row clusters). */
int stemgnn_gru_bwd_rank2_ok(int B, int Hd);
int stemgnn_gru_bwd_rank2(const float* dkey, const float* dquery, const float* wk, const float* wq, const float* x,
                          const float* w_hh, const float* h_ext, const float* reserve, int B, int S, int Hd, int W,
                          float* scratch, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, int* status, void* stream);
/* The same with dquery still in the attention backward's per-chunk partials (stemgnn_attn_laplacian_bwd with parts bit 3 set):
 * `dquery` [B, Hd] is followed by the partials [B][nchunk][Hd], as in the attention scratch; the chunk sum (fixed order: the
 * bits of stemgnn_attn_laplacian_bwd's own reduction) runs inside the zero-fill launch ahead of the recurrence.
 * flags bit 0: the dW_hh product as three-term split-bf16 on the bf16 matrix pipe (STEMGNN_DTYPE=bf16x2; ~2^-16 relative per
 * product, fp32 accumulation, same fixed-order split reduction); 0: exact fp32. */
int stemgnn_gru_bwd_rank2_dq(const float* dkey, float* dquery, int nchunk, int flags, const float* wk, const float* wq, const float* x,
                             const float* w_hh, const float* h_ext, const float* reserve, int B, int S, int Hd, int W,
                             float* scratch, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, int* status, void* stream);
/* stemgnn_gru_bwd_rank2 as two calls, with the dW_hh product running BESIDE the recurrence instead of behind it (round 5):
 *   _begin  (stream):              the fill of the control words, the fork point, the recurrence -- which now stores the gate
 *                                  gradients write-through and counts finished chunks of 4 time steps per workgroup;
 *   _finish (side_stream, stream): on `side_stream`, ordered behind the fill, persistent workgroups (at most the CUs the
 *                                  recurrence leaves free) that take the weight-gradient work items in the order their rows
 *                                  become final, wait (BOUNDED) for the recurrence to get there, claim and compute them; on
 *                                  `stream`, behind the recurrence, the closing launch: the items of the last few time steps,
 *                                  anything the side launch did not claim, the fixed-order sums, dW_ih | db_ih.
 * The caller joins `side_stream` into `stream` afterwards.  Same arguments as stemgnn_gru_bwd_rank2 for both; dW_hh has the
 * SAME BITS as from the single call (one K partition, one summation order, whoever computes which item); a side launch that
 * cannot run beside the recurrence (a serialising graph executor) times out and the closing launch does everything.
 * Only where stemgnn_gru_bwd_overlap_ok (the rank-2 kernels, W <= 16, S >= 32, >= 32 free CUs, STEMGNN_GRU_WHH_OVERLAP=1 --
 * OFF by default: measured 1.102 against 1.105 ms per step at the headline shape, see csrc/gru.hip).
 * ctl: NULL -- the progress counters / claims / arrival counters live inside `scratch`, `_begin` zeroes them and `_finish`
 * makes `side_stream` wait for that fill through an event.  Inside a captured hipGraph that edge is NOT enough: the ROCm
 * executor makes a node with a parent on another branch wait for everything that branch has queued by then -- the recurrence
 * included (measured, profiles/r05_gru_whh_overlap.md).  So a graph caller passes its own buffer of
 * stemgnn_gru_bwd_ctl_words(S) zeroed words, zeroed (every step) at a point that is ordered ahead of both streams' use by
 * edges the step has anyway, and the side launch gets no parent on `stream` at all. */
int stemgnn_gru_bwd_overlap_ok(int B, int S, int Hd, int W);
size_t stemgnn_gru_bwd_ctl_words(int S);
int stemgnn_gru_bwd_rank2_begin(const float* dkey, const float* dquery, const float* wk, const float* wq, const float* x,
                                const float* w_hh, const float* h_ext, const float* reserve, int B, int S, int Hd, int W,
                                float* scratch, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, int* status,
                                unsigned* ctl, void* stream);
int stemgnn_gru_bwd_rank2_finish(const float* dkey, const float* dquery, const float* wk, const float* wq, const float* x,
                                 const float* w_hh, const float* h_ext, const float* reserve, int B, int S, int Hd, int W,
                                 float* scratch, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, int* status,
                                 unsigned* ctl, void* side_stream, void* stream);

/* ---- weight packing (per StockBlock, once per optimizer step) -----------------------------------
 * Folds the length-W DFT (:49-51) into the first GLU layer, drops the dead C2R bins,
 * folds the C2R inverse DFT (:58) into the graph-conv weight (:66-67), and lays the GLU weights out
 * as K-major "pair" panels for the MFMA kernels.  params_host: SG_BLOCK_NPARAMS device pointers
 * (host array). */
int stemgnn_block_pack(const float* const* params_host, const float* tables, float* packed,
                       int W, int multi, void* stream);
/* The pair panels, biases and the folded graph-conv weight only -- without the two fp32 stage streams of the fused fp32
 * kernels (stemgnn_glu_fused_repack writes those).  For a caller whose GLU forward / data gradients run on the split-bf16
 * kernels (stemgnn_glu_split_panels packs THEIR streams from the panels): nothing would read the fp32 streams. */
int stemgnn_block_pack_panels(const float* const* params_host, const float* tables, float* packed, int W, int multi,
                              void* stream);
/* The GLU weights of `packed` once more as the stage stream of the fused three-layer forward (csrc/glu_fused.h), written
 * behind the panels inside the same buffer; stemgnn_block_pack calls it (exported for callers that fill the panels
 * themselves).  No-op for (W, multi) the fused kernel does not cover (padded channel count 4 W multi > 256). */
int stemgnn_glu_fused_repack(float* packed, int W, int multi, void* stream);
/* adjoint of the above: scatter the weight gradients (slab 0 of every gradpart region, left complete by
 * stemgnn_block_wgrad or by the parts & 2 calls of stemgnn_igft_heads_bwd / stemgnn_spectral_glu_bwd) into the parameter
 * gradients grads_host[SG_BLOCK_NPARAMS] (entries may be NULL to skip).  nsplit = the value gradpart was sized with. */
int stemgnn_block_unpack_grads(const float* gradpart, int nsplit, const float* tables,
                               float* const* grads_host, int W, int multi, int has_backcast, void* stream);

/* ---- GFT  (models/base_model.py:62-64)  G[b,n,(k-1)W+t] = sum_m T_k[n,m] X[b,m,t], k=1..3 -----
 * X is addressed as X[b*xs_b + m*xs_n + t*xs_t]: block 0 reads the model input x[B,W,N] in place
 * (xs = W*N, 1, N), block 1 reads the backcast [B,N,W] (xs = N*W, W, 1).  G [M, 3W]. */
int stemgnn_gft_fwd(const float* mul_L, const float* X, long xs_b, long xs_n, long xs_t,
                    float* G, int B, int N, int W, void* stream);
/* dG [2][M,3W]: the two per-branch partial slabs stemgnn_spectral_glu_bwd leaves at stemgnn_scratch_offset_dG (their
 * sum is the gradient; added while loading) -> dX [B,N,W] (may be NULL) and dmul_L[1..3] (+= if accumulate; may be NULL: a
 * caller can run the two products of a block on different streams -- only the data gradient is on the backward's chain). */
int stemgnn_gft_bwd(const float* mul_L, const float* X, long xs_b, long xs_n, long xs_t,
                    const float* dG, float* dX, float* dmul_L, int accumulate,
                    int B, int N, int W, void* stream);
/* d(mul_L)[1..3] of BOTH blocks as one product (written, not accumulated): the reduction runs over block 0's (b, t) range,
 * then block 1's.  X0 / X1, dG0 / dG1 as in stemgnn_gft_bwd (reference: autograd of models/base_model.py:64, mul_L is shared
 * by the two blocks, :172). */
int stemgnn_gft_bwd_dt2(const float* X0, long xs0_b, long xs0_n, long xs0_t, const float* dG0, const float* X1, long xs1_b,
                        long xs1_n, long xs1_t, const float* dG1, float* dmul_L, int B, int N, int W, void* stream);

/* ---- spe_seq_cell: DFT -> 3x GLU on Re and Im (models/base_model.py:46-54, GLU :12-13) ----------
 * G = saved + offset(G) is read; GLU outputs and gates are written into `saved`. */
int stemgnn_spectral_glu_fwd(const float* packed, float* saved, int B, int N, int W, int multi, void* stream);
/* needs d(pre-activation) of the last GLU layer in scratch (written by igft_heads_bwd); produces dG in
 * scratch.dG and the GLU weight-gradient partials in gradpart.
 * parts: bit 0 = data-gradient chain (layer 2 -> 1 -> 0 -> dG), bit 1 = the weight-gradient GEMMs.  The two parts
 * only share read-only inputs once the chain has run, so a caller may issue part 2 later or on another stream
 * (it is off the critical path of the backward pass); 3 = both, in order.  bit 2 (with bit 1): the fused weight-gradient
 * launch's products as three-term split-bf16 (see stemgnn_block_wgrad_split). */
int stemgnn_spectral_glu_bwd(const float* packed, const float* saved, float* scratch, float* gradpart,
                             int nsplit, int parts, int B, int N, int W, int multi, void* stream);
/* Split-bf16 arithmetic for the same layers (BASELINE.json configs[1] "bf16/fp32"; reference data and weights are fp32,
 * models/base_model.py:12-13, 52-54): every fp32 operand is the exact sum of `splits` bf16 numbers (3: fp32-class
 * products from 6 bf16 MFMAs, 2: ~2^-16 relative from 3) with fp32 accumulation on v_mfma_f32_32x32x16_bf16.
 * stemgnn_glu_split_panels turns the packed fp32 panels of layers 1, 2 (both branches) into the bf16 plane sets the two
 * entry points read (`split`: caller-owned, 16-byte aligned, stemgnn_glu_split_floats floats; once per step, after
 * stemgnn_block_pack).  _fwd_split == stemgnn_spectral_glu_fwd, _dgrad_split == stemgnn_spectral_glu_bwd with parts = 1,
 * with layers 1, 2 on the split kernel (layer 0 and its data gradient stay exact fp32: K = 3W, traffic-bound); saved
 * activations, scratch layout and every downstream stage are unchanged.  Shapes that break the 16-byte rules of the
 * split kernel run the fp32 kernels. */
size_t stemgnn_glu_split_floats(int W, int multi, int splits);
int stemgnn_glu_split_panels(const float* packed, float* split, int W, int multi, int splits, void* stream);
int stemgnn_spectral_glu_fwd_split(const float* packed, const float* split, float* saved, int B, int N, int W, int multi,
                                   int splits, void* stream);
/* Warm-up of the fused forward: the same kernel instance the real launch of a [B, N] batch gets, over four row blocks (eight
 * workgroups, one per XCD and branch), writing into a caller-owned dummy `saved` of stemgnn_glu_warm_saved_floats floats whose
 * G region holds any finite values.  splits = 0: the fp32 kernel (`split` ignored), 2: the split-bf16 kernel.  Brings the
 * kernel's code and block's weight stream into the XCDs' L2 ahead of the first real launch of a step (-10 us at PEMS07); a
 * no-op returning 0 where the fused kernels do not apply. */
size_t stemgnn_glu_warm_saved_floats(int W, int multi);
int stemgnn_spectral_glu_fwd_warm(const float* packed, const float* split, float* saved, int B, int N, int W, int multi,
                                  int splits, void* stream);
int stemgnn_spectral_glu_dgrad_split(const float* packed, const float* split, const float* saved, float* scratch,
                                     int B, int N, int W, int multi, int splits, void* stream);
/* Round 5: with splits == 2 and a padded channel count 4 W multi <= 256 (every BASELINE configuration but configs[4])
 * stemgnn_spectral_glu_fwd_split is ONE launch for the three layers with the split-bf16 products inside
 * (csrc/glu_fused_bf16.h: the row block's activations resident in LDS as two bf16 planes, the pre-split weights -- written
 * by stemgnn_glu_split_panels into the same `split` buffer -- on a direct-to-LDS ring; layer 0 is split too), and
 * stemgnn_spectral_glu_dgrad_split is ONE launch for d(pre-activation) of layer 2 -> 1 -> 0 -> dG likewise (its layer-0 product
 * included).  The saved out / gate and every d(pre-activation) stay fp32, so weight gradients, heads and the rest are
 * unchanged.  stemgnn_glu_fused_bf16_ok tells (1 / 0) whether those forms apply (STEMGNN_GLU_FUSED=0 turns them off). */
int stemgnn_glu_fused_bf16_ok(int W, int multi, int splits);

/* ---- C2R iDFT + graph-conv weight + forecast / backcast heads (models/base_model.py:55-58, 65-74)
 * forecast [M,W]: written (accumulate=0) or added to (accumulate=1: result[0]+result[1], :174).
 * backcast [M,W] (block 0 only, else NULL); X as in gft_fwd (short-cut input, :71). */
int stemgnn_igft_heads_fwd(const float* const* params_host, const float* packed, float* saved,
                           const float* X, long xs_b, long xs_n, long xs_t,
                           float* forecast, int accumulate, float* backcast,
                           int B, int N, int W, int multi, void* stream);
/* dforecast [M,W], dbackcast [M,W] or NULL, backcast = forward output (for sigmoid').  parts bit 0: data path
 * (dpF, dpB, dig and d(pre-activation) of the last GLU layer into scratch); bit 1: the heads' / graph-conv
 * weight-gradient partials (needs bit 0's results). */
int stemgnn_igft_heads_bwd(const float* const* params_host, const float* packed, const float* saved,
                           const float* X, long xs_b, long xs_n, long xs_t,
                           const float* dforecast, const float* dbackcast, const float* backcast,
                           float* scratch, float* gradpart, int nsplit, int parts,
                           int B, int N, int W, int multi, void* stream);

/* Stand-alone block with a differentiable input only: the short-cut head's direct term (models/base_model.py:71-72),
 * dX [M,W] -= dpB BS_w, with dpB taken from the scratch the data part (parts & 1) of stemgnn_igft_heads_bwd filled. */
int stemgnn_shortcut_dx(const float* scratch, const float* bs_w, float* dX, int B, int N, int W, int multi, void* stream);

/* ALL weight gradients of one StockBlock (autograd of models/base_model.py:12-13, 66-74 via models/handler.py:164): the
 * six GLU products and the heads' FR / F / BC / graph-conv products in ONE launch of the fused weight-gradient kernel
 * (direct-to-LDS operand ring, in-kernel fixed-order split reduction), the short-cut head BS beside it.  Needs the data
 * parts (parts & 1) of stemgnn_igft_heads_bwd and stemgnn_spectral_glu_bwd; leaves complete gradients in slab 0 of every
 * gradpart region (-> stemgnn_block_unpack_grads).  has_bc: block 0 (backcast heads present).  cu_percent (10..100):
 * share of the CUs the big launch should fill -- lower it when latency-critical kernels run beside it on another stream. */
int stemgnn_block_wgrad(const float* const* params_host, const float* packed, const float* saved,
                        const float* X, long xs_b, long xs_n, long xs_t, const float* dforecast, int has_bc,
                        float* scratch, float* gradpart, int nsplit, int cu_percent,
                        int B, int N, int W, int multi, void* stream);
/* The same launch with the fused kernel's products as three-term split-bf16 (a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32
 * accumulation) on v_mfma_f32_32x32x16_bf16 when splits == 2 (STEMGNN_DTYPE=bf16x2; ~2^-16 relative per product); splits == 0
 * is stemgnn_block_wgrad.  Reference: the weight gradients autograd forms for models/base_model.py:12-13, 66-72. */
int stemgnn_block_wgrad_split(const float* const* params_host, const float* packed, const float* saved,
                              const float* X, long xs_b, long xs_n, long xs_t, const float* dforecast, int has_bc,
                              float* scratch, float* gradpart, int nsplit, int cu_percent, int B, int N, int W,
                              int multi, int splits, void* stream);

/* ---- callers on either side of the blocks, fused ------------------------------------------------
 * fc tail (models/base_model.py:97-101,174-179): fsum [B*N, W] (block forecast sum) ->
 * forecast [B,H,N] = Linear(W,H)(LeakyReLU_0.01(Linear(W,W)(fsum))) permuted.  stemgnn_fc_tail_supported(W,H)
 * tells whether the fused kernels cover (W,H) (registers / LDS); otherwise the caller keeps its own fc. */
int stemgnn_fc_tail_supported(int W, int H);
size_t stemgnn_fc_tail_scratch_floats(int B, int N, int W, int H);
int stemgnn_fc_tail_fwd(const float* fsum, const float* w0, const float* b0, const float* w2, const float* b2,
                        int B, int N, int W, int H, float* forecast, void* stream);
int stemgnn_fc_tail_bwd(const float* dforecast, const float* fsum, const float* w0, const float* b0, const float* w2,
                        int B, int N, int W, int H, float* scratch, float* dfsum, float* dw0, float* db0,
                        float* dw2, float* db2, void* stream);
/* training tail in two launches: fc forward -> nn.MSELoss(reduction='mean') against target [B,H,N] -> d(loss)/d(forecast)
 * (upstream gradient taken as 1: loss.backward(), models/handler.py:162-164) -> fc backward.  Writes the loss (float, and
 * += into *loss_accum, double, if given), dfsum [B*N, W] and the four fc parameter gradients; forecast [B,H,N] is also
 * written when non-NULL.  Same arithmetic as stemgnn_fc_tail_fwd + stemgnn_mse_fwd/_bwd + stemgnn_fc_tail_bwd. */
size_t stemgnn_fc_tail_train_scratch_floats(int B, int N, int W, int H);
int stemgnn_fc_tail_train(const float* fsum, const float* target, const float* w0, const float* b0, const float* w2,
                          const float* b2, int B, int N, int W, int H, float* scratch, float* forecast,
                          float* loss, double* loss_accum, float* dfsum, float* dw0, float* db0, float* dw2,
                          float* db2, void* stream);
/* The same two launches as separate calls: `_rows` = the per-row part (forecast, d(fsum), per-row-block partial sums in
 * `scratch`), `_finish` = the fixed-order sum of the partials into loss / loss_accum / the fc gradients.  Nothing on the
 * backward's chain reads what `_finish` writes, so a step driver may queue it on another stream; same bits as the one call. */
int stemgnn_fc_tail_train_rows(const float* fsum, const float* target, const float* w0, const float* b0, const float* w2,
                               const float* b2, int B, int N, int W, int H, float* scratch, float* forecast, float* dfsum,
                               void* stream);
int stemgnn_fc_tail_train_finish(const float* scratch, int B, int N, int W, int H, float* loss, double* loss_accum,
                                 float* dw0, float* db0, float* dw2, float* db2, void* stream);
/* Zero `bytes` bytes at `ptr` in stream order, as a KERNEL launch (the reference's zero_grad, models/handler.py:160, when it is
 * not fused into the optimizer kernel; control words).  The step path never uses hipMemsetAsync: inside a captured hipGraph
 * a memset node was seen to run into the kernel node that follows it. */
int stemgnn_fill_zero(void* ptr, size_t bytes, void* stream);
/* RMSprop step of the reference driver (models/handler.py:127,165; torch defaults alpha=0.99, momentum 0, not
 * centered) over flat, 16-byte aligned parameter / gradient / square_avg buffers of n floats; lr is read from
 * device memory; zero_grad != 0 also clears the gradients for the next step (handler.py:160); every gradient is
 * multiplied by grad_scale first (1 / world_size after a SUM all-reduce of the flat bucket: no separate averaging pass). */
int stemgnn_rmsprop_step(float* params, float* grads, float* square_avg, size_t n, const float* lr_dev,
                         float alpha, float eps, int zero_grad, float grad_scale, void* stream);

/* Adam step of the driver's other optimizer branch (models/handler.py:128-129; torch defaults weight_decay 0,
 * amsgrad off) over flat buffers of n floats; lr and the step count (step_dev[0], a float, incremented by the call) are
 * read from device memory, so the call is replayable inside a hipGraph; zero_grad / grad_scale as in rmsprop_step. */
int stemgnn_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, size_t n, const float* lr_dev,
                      float* step_dev, float beta1, float beta2, float eps, int zero_grad, float grad_scale,
                      void* stream);

/* ---- data path either side of the hot path ---------------------------------------------
 * normalized() (data_loader/forecast_dataloader.py:7-22): out[t,n] = (float)clip01?((raw[t,n]-sub[n])/div[n]) in IEEE
 * fp64 (z_score: sub=mean, div=std with 0->1; min_max: sub=min, div=max-min+1e-5, clip01=1).  raw [T,N] fp64. */
int stemgnn_normalize_series(const double* raw, const double* sub, const double* div, int clip01, float* out,
                             long T, int N, void* stream);
/* ForecastDataset.__getitem__ + default collate (forecast_dataloader.py:56-63, models/handler.py:136-138,158-159):
 * x[b,w,:] = series[hi[b]-W+w,:], y[b,h,:] = series[hi[b]+h,:]; series [T,N] fp32 resident, hi [B] int64 (device).
 * A window outside [0,T] is written as zeros and sets bit 0 of *status (device int, may be NULL). */
int stemgnn_window_gather(const float* series, const long long* hi, float* x, float* y, int B, int W, int H, int N,
                          long T, int* status, void* stream);
/* The same gather as an ITERATOR over one shuffled epoch (the DataLoader loop of models/handler.py:157-159): order [count]
 * int64 window-end rows (device), queue = device long long[4] {position, 0, count, wrap}, set by the caller when an epoch
 * is loaded (wrap != 0: a position from which no full batch is left goes back to 0 instead of running past the end --
 * start-up replays only).  Every call gathers windows order[position .. position+B) and advances position by B ON THE DEVICE, so a
 * captured hipGraph step needs no per-step index copy.  Past the end: zeros + bit 1 of *status. */
int stemgnn_window_gather_queue(const float* series, const long long* order, long long* queue, float* x, float* y, int B,
                                int W, int H, int N, long T, int* status, void* stream);
/* nn.MSELoss(reduction='mean') of the driver (models/handler.py:140,162): loss[0] = mean((forecast-target)^2) with
 * a fixed-order two-stage reduction; bwd: dforecast = grad_loss[0] * 2 (forecast-target)/n. */
size_t stemgnn_mse_scratch_floats(void);
int stemgnn_mse_fwd(const float* forecast, const float* target, size_t n, float* scratch, float* loss,
                    double* loss_accum, void* stream);   /* loss_accum (may be NULL): device double, += loss */
int stemgnn_mse_bwd(const float* forecast, const float* target, size_t n, const float* grad_loss, float* dforecast,
                    void* stream);
/* one iteration of the rolling inference (models/handler.py:56-61), out of place: inputs_next = inputs shifted left
 * by L with forecast [B,L,N] appended; forecast_steps[b, step + j, :] = forecast[b,j,:] for j < min(horizon-step, L).
 * L > W (which the reference fails on with a shape error) returns SG_EINVAL. */
int stemgnn_roll_window(const float* inputs, const float* forecast, float* inputs_next, float* forecast_steps,
                        int B, int W, int L, int N, int step, int horizon, void* stream);
/* evaluate() (utils/math_utils.py:24-74) with the optional de_normalized() (forecast_dataloader.py:25-38) in front:
 * target / forecast [count,H,N] fp32, v -> v*mul[n] + add[n] in fp64 when mul != NULL (z_score: std, mean; min_max:
 * max-min+1e-8, min).  out (fp64) = overall[3] | by_node[3][N] | by_step[3][H] | by_step_node[3][H][N], each triple
 * (MAPE with its +1e-5 and clip at 5, MAE, RMSE). */
size_t stemgnn_eval_scratch_doubles(long count, int H, int N);
size_t stemgnn_eval_out_doubles(int H, int N);
int stemgnn_eval_metrics(const float* target, const float* forecast, const double* mul, const double* add,
                         long count, int H, int N, double* scratch, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STEMGNN_HIP_H */
