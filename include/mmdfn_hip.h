/*
 * mmdfn_hip.h -- C ABI of libmmdfn_hip.so: the MI355X (gfx950) kernels behind
 * the MM-DFN graph dynamic-fusion hot path.
 *
 * The reference (zerohd4869/MM-DFN) has no FFI layer: its "operators" are
 * torch ops called from nn.Module.forward.  Each entry point below replaces
 * the reference call site cited next to it; the Python host side
 * (the mm_dfn_amd Python package) binds them with ctypes and keeps the reference's
 * nn.Module signatures (see INTEGRATION.md for the stub a maintainer adds).
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (the PyTorch
 *     caching allocator); kernels never allocate or free;
 *   - fp32 row-major contiguous data, int32 / int64 index arrays;
 *   - asynchronous enqueue on `stream` (a hipStream_t passed as void*), no
 *     internal synchronisation, no global state, re-entrant;
 *   - return value 0 = success, otherwise a hipError_t (or -1 for an invalid
 *     argument); nothing throws across the ABI.
 *
 * Dialogue layout shared by the graph kernels ("block-tile adjacency"):
 *   B dialogues of lengths dia_len[i]; row_start[i] = sum_{j<i} dia_len[j]
 *   (B+1 entries, row_start[B] = N).  M modalities.  Node (m, r) of the
 *   multimodal dialogue graph is row m*N + r of every (M*N, d) feature matrix
 *   (the reference's cat([a, v, l], 0) order, model_mm.py:98).
 *   The normalised adjacency is never materialised densely; it is stored as
 *     tiles : for dialogue i, modality m a dia_len[i] x dia_len[i] fp32 tile,
 *             row-major with leading dimension ld_i = round_up(dia_len[i], 4),
 *             at float offset tile_base[i] + m * dia_len[i] * ld_i
 *             (tile_base: B+1 int64 entries, multiples of 4); pad columns are 0;
 *     cross : for each unordered modality pair (m<n), in lexicographic order,
 *             N fp32 values: entry r is A[(m,r),(n,r)] = A[(n,r),(m,r)].
 */
#ifndef MMDFN_HIP_H
#define MMDFN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Library / device sanity: returns the ABI version (currently 17: 16 + the GRU backward launch's weight-gradient riders mmdfn_wgrad_riders_{stage,staged,flush,drain}, mmdfn_gru_seq_bwd_idle_cus, mmdfn_gru_seq_bwd_step_ns, and the dropout-flag draw as a rider of the GRU forward launch mmdfn_keep_flags_{stage,flush}, mmdfn_gru_seq_fwd_takes_flags; 16 = 15 + mmdfn_linear_planes_group, mmdfn_party_gather_bwd_colsum, mmdfn_party_combine_bwd_dst, mmdfn_prop_layer_fwd; 15 = 14 + mmdfn_weight_planes_workspace, mmdfn_cut_weight_planes, mmdfn_linear_planes; 14 = 13 + mmdfn_lstm_gate_{planes_workspace,cut_weights,fwd_pre,takes_planes}; 13 = 12 + mmdfn_gemm_tn_batch_ext, mmdfn_head_bwd_partial / _groups, mmdfn_colsum_partial; 12 = 11 + the segmented GRU recurrence mmdfn_gru_seq_{fwd,bwd}_seg, mmdfn_gru_tab_reduce, the strided forms mmdfn_lstm_gate_fwd_ld, mmdfn_gcnii_layer_bwd_ld, mmdfn_focal_loss_{fwd,bwd}_ignore and mmdfn_focal_loss_fwd_grad). */
int mmdfn_abi_version(void);

/* ---------------------------------------------------------------------------
 * K6  scatter-propagate  out = A_hat . H        (replaces torch.spmm(adj, input),
 *                                                model_GCN.py:178)
 *   out[(m,r),:] = sum_q tile_{i,m}[r,q] * H[(m,q),:] + sum_{n!=m} cross_{mn}[r] * H[(n,r),:]
 *   transpose != 0 uses tile^T (for dH = A^T . dO when tiles are not symmetric).
 *   H, out: M*N rows of d fp32 with row strides ldh / ldo floats (>= d, multiples of 4, so a
 *   column slice of a wider matrix can be read or written in place); d % 4 == 0.
 *   max_len = max_i dia_len[i].
 * ------------------------------------------------------------------------- */
int mmdfn_propagate(const float* tiles, const float* cross, const float* H, float* out,
                    const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                    int B, int M, int N, int d, int ldh, int ldo, int max_len, int transpose,
                    void* stream);

/* ---------------------------------------------------------------------------
 * K6 backward w.r.t. the adjacency, restricted to the stored pattern
 * (autograd of model_GCN.py:178 w.r.t. adj; the reference gets a dense
 * (MN x MN) gradient from SpmmBackward):
 *   dtiles_{i,m}[p,q] (+)= X[(m,p),:] . Y[(m,q),:]
 *   dcross_{mn}[r]    (+)= X[(m,r),:].Y[(n,r),:] + X[(n,r),:].Y[(m,r),:]
 *   with X = dOut, Y = H (row strides ldx / ldy floats).  accumulate != 0 adds into dtiles/dcross.
 *   Any d % 4 == 0 (wide d runs on the bf16-piece kernel); the adjacency build below needs D <= 512.
 * ------------------------------------------------------------------------- */
int mmdfn_tile_outer(const float* X, const float* Y, float* dtiles, float* dcross,
                     const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                     int B, int M, int N, int d, int ldx, int ldy, int max_len, int accumulate,
                     void* stream);

/* ---------------------------------------------------------------------------
 * K5  adjacency build (replaces MM_GCN.create_big_adj, model_mm.py:122-180)
 *   feats  : (M, N, D) fp32 (modality-major stack of the encoder outputs)
 *   unit   : (M, N, D) out, x / ||x||              (saved for backward)
 *   norm   : (M, N)    out, ||x||
 *   cosg   : tile-shaped out, raw cosine Gram G    (saved for backward)
 *   cdot   : (npairs, N) out, raw cross cosines    (saved for backward)
 *   rdeg   : (M, N)    out, degree^-1/2
 *   tiles, cross : the normalised adjacency (layout above)
 *   sim(c) = 1 - acos(0.99999 c)/pi ; cross entries are scaled by modal_weight.
 * ------------------------------------------------------------------------- */
int mmdfn_adj_build(const float* feats, float* unit, float* norm, float* cosg, float* cdot,
                    float* rdeg, float* tiles, float* cross,
                    const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                    int B, int M, int N, int D, int max_len, float modal_weight, void* stream);

/* Backward of mmdfn_adj_build: given dtiles / dcross (gradients of the stored
 * entries) produce dfeats (M, N, D).  `wsym` (tile-shaped), `etile`
 * (tile-shaped), `ecross` (npairs, N), `ddeg` (M, N) and `dunit` (M, N, D) are
 * caller-provided scratch.  * addend (mmdfn_adj_build_bwd, may be NULL): (M, N, D) gradient reaching the same features on another path (they are
 *   also the input of the GCN stack); dfeats = adjacency gradient + addend, so no separate accumulation launch runs.
 */
int mmdfn_adj_build_bwd(const float* dtiles, const float* dcross,
                        const float* unit, const float* norm, const float* cosg, const float* cdot,
                        const float* rdeg, const float* tiles, const float* cross,
                        float* wsym, float* etile, float* ecross, float* ddeg, float* dunit,
                        float* dfeats, const float* addend,
                        const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                        int B, int M, int N, int D, int max_len, float modal_weight, void* stream);

/* ---------------------------------------------------------------------------
 * K2  fused GRU recurrence (replaces the time loop inside nn.GRU for ``lstm_l``
 * and ``rnn_parties``: model.py:866,868, called at :1082,1112,1132,1145).
 * One layer, both directions, `ngroups` independent GRUs in one launch
 * (host arrays of device pointers, one entry per group; ngroups <= 4):
 *   gi[g]    : (T, rows, 2, 3H)  X W_ih^T + b_ih for both directions (dir-major columns)
 *   w_hh[2g + dir] : (3H, H), b_hh[2g + dir] : (3H)  per direction, exactly as nn.GRU stores
 *              weight_hh_l*[_reverse] / bias_hh_l*[_reverse] (arrays of 2*ngroups pointers); gate order r, z, n
 *   y[g]     : (T, rows, 2H) out; direction 1 runs t = T-1 .. 0; zero initial state
 *   gates[g] : (T, rows, 2, 4, H) out: r, z, n and W_hn h + b_hn (saved for backward)
 * H must be 100 (the reference hard-codes D_e = 100).
 * ------------------------------------------------------------------------- */
int mmdfn_gru_seq_fwd(int ngroups, const float* const* gi, const float* const* w_hh,
                      const float* const* b_hh, float* const* y, float* const* gates,
                      const int* rows, const int* T, int H, void* stream);

/* Backward through time of the same recurrence: given dy[g] (T, rows, 2H) writes
 *   dgi[g], dgh[g] : (T, rows, 2, 3H)  gradients of the input-side / hidden-side gate
 *   pre-activations (they differ only in the n gate: dgh_n = dgi_n * r).
 * Weight gradients are dense contractions of these done by the caller. */
int mmdfn_gru_seq_bwd(int ngroups, const float* const* dy, const float* const* y,
                      const float* const* gates, const float* const* w_hh,
                      float* const* dgi, float* const* dgh,
                      const int* rows, const int* T, int H, void* stream);

/* Valid-length ("segmented") form of the same recurrence for the speaker-party batch (model.py:1076-1087: per dialogue b and
 * speaker p the reference fills rows [:k_bp] of a zero (L, H) buffer, runs the party GRU over all L steps and scatters rows
 * [:k_bp] back).  Exact work removal, see csrc/gru.hip "Valid-length truncation".  Per group g (host arrays, one entry per group):
 *   rank[g]  : the gather's (T, BP) int32 rank array (>= 0 where speaker p talks) or NULL (every row runs all T steps);
 *              rows[g] must be a multiple of BP[g] (one (B, P) block per gathered modality), P[g] <= 16, P[g] * T[g] <= 2048
 *   tdir[g]  : -1, or the direction that stops at / starts from the valid length: its P sequences per (modality, dialogue)
 *              run back to back in one workgroup.  Rows with k = 0 are skipped in BOTH directions.
 *   ytab[g]  : NULL, or (tdir[g] must be 1) the (T, 1, 2H) outputs of the all-padding sequence (gi = b_ih at every step, run
 *              through mmdfn_gru_seq_fwd as a one-row group): a truncated reverse row starts at t = k - 1 from ytab[k] and
 *              its outputs at t >= k are copies of ytab[t]; with NULL a truncated row starts from 0 and y[t >= k] = 0.
 * Positions never visited hold ytab copies / zeros in y and are NOT written in gates.
 * Backward: dgi / dgh are zero at the positions never visited; with dhinit[g] (rows, H) and kout[g] (rows) int32 given
 * (tdir[g] == 1) the gradient wrt every truncated row's start state and its step count are written, and
 * mmdfn_gru_tab_reduce sums what reaches the all-padding sequence:
 *   dyt[t][dir half] = sum_{rows: k <= t} dy[t][row] + sum_{rows: k == t >= 1} dhinit[row]   (the other half zero),
 * the dy of that one-row group's own mmdfn_gru_seq_bwd. */
int mmdfn_gru_seq_fwd_seg(int ngroups, const float* const* gi, const float* const* w_hh,
                          const float* const* b_hh, float* const* y, float* const* gates,
                          const int* rows, const int* T, int H, const int32_t* const* rank, const int* P,
                          const int* BP, const int* tdir, const float* const* ytab, void* stream);
int mmdfn_gru_seq_bwd_seg(int ngroups, const float* const* dy, const float* const* y,
                          const float* const* gates, const float* const* w_hh,
                          float* const* dgi, float* const* dgh, const int* rows, const int* T, int H,
                          const int32_t* const* rank, const int* P, const int* BP, const int* tdir,
                          float* const* dhinit, int32_t* const* kout, void* stream);
int mmdfn_gru_tab_reduce(const float* dy, const int32_t* kout, const float* dhinit, float* dyt, int rows, int T,
                         int H, int dir, void* stream);

/* ---------------------------------------------------------------------------
 * K8  LSTM-cell gate math of the "reasoning" / dynamic-fusion module (replaces the pointwise part
 * of nn.LSTM with seq_len 1, model_GCN.py:433,466; gate order i, f, g, o):
 *   G (R, 4H) = x W_ih^T + b_ih + h W_hh^T + b_hh   (dense contractions done by the caller)
 *   c = sig(f) c_prev + sig(i) tanh(g) ;  h = sig(o) tanh(c).   c_prev may be NULL (zero state).
 * Backward: dG (R, 4H), dc_prev (R, H) from dh, dc_next (either may be NULL = zero).
 * ------------------------------------------------------------------------- */
int mmdfn_lstm_pointwise_fwd(const float* G, const float* c_prev, float* h_out, float* c_out,
                             int64_t R, int H, void* stream);
int mmdfn_lstm_pointwise_bwd(const float* G, const float* c_prev, const float* c_new,
                             const float* dh, const float* dc_next, float* dG, float* dc_prev,
                             int64_t R, int H, void* stream);

/* ---------------------------------------------------------------------------
 * K7  GCNII update (replaces model_GCN.py:179-186 variant branch + :469-472):
 *   S2 (R, 2d) = [A.x | h0],  P (R, d) = S2 . W  (contraction done by the caller)
 *   out = relu(theta P + (1-theta)((1-alpha) A.x + alpha h0)) * mask + q
 *   mask (dropout keep-mask already scaled by 1/(1-p)) and q may be NULL.
 * Backward: dP (R, d) and dS2 (R, 2d) from dout (dq = dout is the caller's).
 * ------------------------------------------------------------------------- */
int mmdfn_gcnii_combine_fwd(const float* P, const float* S2, const float* q, const float* mask,
                            float* out, float theta, float alpha, int64_t R, int d, void* stream);
int mmdfn_gcnii_combine_bwd(const float* P, const float* S2, const float* mask, const float* dout,
                            float* dP, float* dS2, float theta, float alpha, int64_t R, int d,
                            void* stream);

/* ---------------------------------------------------------------------------
 * K3 / K4  speaker-party gather / scatter + pad strip (replaces the per-(dialogue, speaker) Python
 * slice-assign loops model.py:1076-1087 (x3 modalities) and simple_batch_graphify model.py:553-565).
 *   X[m]   : Mn host-array entries, each a device (L, B, H) fp32 projection (Mn <= 4)
 *   qmask  : (L, B, P) fp32 speaker flags (non-zero = speaker p utters t in dialogue b)
 *   S      : (L, Mn*B*P, H) out: column ((m*B+b)*P+p) holds speaker p's utterances of dialogue b
 *            compacted to the front in time order, zero rows behind (the party-GRU input)
 *   bias   : NULL, or H floats added to EVERY row of S.  With it the gather can run after a bias-free projection
 *            (gather(X) W^T + b == gather(X W^T) + b, padding rows = b): the input contraction of the first party-GRU
 *            layer then runs over the L*B real utterances instead of the L*P*B mostly-zero party rows
 *   rank   : (L, B, P) int32 out: position of utterance t inside its party sequence, -1 otherwise
 * combine: out[m][n][:] = base[m][t,b,:] + weights[m] * E[rank[t,b,p*], ((m*B+b)*P+p*), :]
 *   with flat_idx[n] = t*B + b (dialogue-major order), p* = LAST flagged speaker (the reference
 *   scatters speaker by speaker), E = party-GRU output or NULL; out: (Mn, N, H).  E holds ONE column block per
 *   modality whose weight is non-zero, in modality order: (L, nact*B*P, H), column ((slot(m)*B+b)*P+p) with
 *   slot(m) = #{m' < m : weights[m'] != 0}.  (The reference also encodes the zero-weight modality and multiplies the
 *   result by 0, model.py:1121; that block is never needed.)
 *   `weights` is a HOST array of Mn floats (speaker_weights, model.py:816).
 * Backward entries: dX / dbase are Mn host-array entries of device (L, B, H) buffers;
 *   combine_bwd expects dbase and dE pre-zeroed (it writes only the rows that exist).
 *   gather_bwd: `addend` (NULL, or Mn entries each NULL or an (L, B, H) buffer) is added to dX -- the gradient that
 *   reaches X_m as the base of the combine stage, so that no separate accumulation launch is needed.
 * ------------------------------------------------------------------------- */
int mmdfn_party_gather(int Mn, const float* const* X, const float* qmask, const float* bias, float* S,
                       int32_t* rank, int L, int B, int P, int H, void* stream);
int mmdfn_party_gather_bwd(int Mn, const float* dS, const int32_t* rank, float* const* dX,
                           const float* const* addend, int L, int B, int P, int H, void* stream);
/* ABI 16: mmdfn_party_gather_bwd and mmdfn_colsum_partial of the SAME dS (viewed as (L Mn B P) x H; the party GRU's bias
 * gradient, model.py:1082) in one launch; returns the number of [H] slabs written to `workspace` (mmdfn_colsum_workspace(H)
 * floats) for mmdfn_gemm_tn_batch_ext to sum, negative = rejected. */
int mmdfn_party_gather_bwd_colsum(int Mn, const float* dS, const int32_t* rank, float* const* dX,
                                  const float* const* addend, int L, int B, int P, int H, float* workspace, void* stream);

/* Column sums out[c] = sum_r A[r][c] of an (R, H) matrix (row stride lda; H, lda % 4 == 0, 16-byte aligned), bit-reproducible
 * (slab partial sums + a small final launch): the bias gradient of gate pre-activations gathered after a bias-free
 * projection (replaces the autograd `sum` of the broadcast bias add, model.py:1082).
 * workspace: mmdfn_colsum_workspace(H) floats. */
/* Dropout keep flags (replaces the generator launch behind F.dropout / nn.Dropout / nn.GRU(dropout=p), model.py:866,868,1328,
 * model_GCN.py:453-470): out[i] = 1.0f with probability `keep`, else 0.0f, n % 4 == 0, 16-byte aligned; Philox4x32-10 keyed by
 * state[0] (seed) at counter state[1] + i / 8 (16 random bits per flag: the rate is exact to 2^-16).  `state`: three 64-bit words
 * in DEVICE memory (seed, offset, 0); the launch adds ceil(n / 8) rounded up to a multiple of 64 to the offset itself, so replays of a captured graph draw fresh flags. */
int mmdfn_keep_flags(float* out, int64_t n, float keep, void* state, void* stream);
/* ABI 17: the same draw as a RIDER of the first GRU layer's forward recurrence launch.  mmdfn_keep_flags_stage takes the arguments
 * of mmdfn_keep_flags and does not launch (a second staged draw is launched at once); the next mmdfn_gru_seq_fwd launch of the kind
 * mmdfn_gru_seq_fwd_takes_flags answers 1 for (one sequence per workgroup, fewer workgroups than CUs) runs the draw as extra
 * workgroups on the CUs the recurrence leaves idle -- the same flags bit for bit (which counter yields which flag depends on
 * neither the grid nor the block size); mmdfn_keep_flags_flush launches a draw that is still staged.  The flags must not be read
 * before the launch that carries them (their first consumer in the reference is the dropout behind that GRU layer,
 * model.py:866). */
int mmdfn_keep_flags_stage(float* out, int64_t n, float keep, void* state, void* stream);
int mmdfn_keep_flags_flush(void* stream);
int mmdfn_gru_seq_fwd_takes_flags(int ngroups, const int* rows);
int64_t mmdfn_colsum_workspace(int H);
int mmdfn_colsum(const float* A, int64_t R, int H, int lda, float* out, float* workspace, void* stream);
/* The first launch of mmdfn_colsum alone (ABI 13): returns the number (> 0) of [H] slabs left in `workspace` for
 * mmdfn_gemm_tn_batch_ext to sum (ext_N = 0, ext_M = H), negative = rejected. */
int mmdfn_colsum_partial(const float* A, int64_t R, int H, int lda, float* workspace, void* stream);
int mmdfn_party_combine(int Mn, const float* const* base, const float* E, const int32_t* rank,
                        const int64_t* flat_idx, float* out, const float* weights,
                        int L, int B, int P, int N, int H, void* stream);
int mmdfn_party_combine_bwd(int Mn, const float* dout, const int32_t* rank, const int64_t* flat_idx,
                            float* const* dbase, float* dE, const float* weights,
                            int L, int B, int P, int N, int H, void* stream);
/* ABI 16: the same gradients written destination by destination -- dbase / dE need NOT be zeroed by the caller (its fill was a
 * launch of its own).  inv: (L * B) int64, the row of (t, b) in the stripped order or -1 (the inverse of flat_idx).
 * Returns -2 when the shape is not covered (L > 2048): use mmdfn_party_combine_bwd on pre-zeroed buffers. */
int mmdfn_party_combine_bwd_dst(int Mn, const float* dout, const int32_t* rank, const int64_t* inv, float* const* dbase,
                                float* dE, const float* weights, int L, int B, int P, int N, int H, void* stream);


/* ---------------------------------------------------------------------------
 * Dropout as a multiply by precomputed keep flags, several tensors per launch (the inter-layer dropout of
 * nn.GRU(num_layers=2, dropout=p) for the context and the party encoder, model.py:866,868, one launch each way):
 *   out[g][i] = x[g][i] * mask[g][i] * scale,  i < n[g]   (n[g] % 4 == 0, 16-byte aligned buffers, ngroups <= 4).
 * The backward pass is the same call on the incoming gradients.  x, mask, out: HOST arrays of device pointers.
 * ------------------------------------------------------------------------- */
int mmdfn_mask_scale(int ngroups, const float* const* x, const float* const* mask, float* const* out,
                     const int64_t* n, float scale, void* stream);

/* ---------------------------------------------------------------------------
 * K1  dense projection, fp32 in / fp32 out (replaces nn.Linear / F.linear / torch.mm on the hot path:
 * model.py:1065,1094,1129; the hoisted nn.GRU input contraction; model_GCN.py:454,466,186):
 *   Y[r, n] = act( sum_k X[r, k] W[n, k] + bias[n] ) (+ Y[r, n] if accumulate)
 *   X: R rows of K floats, row stride ldx; W: (N, K) contiguous (nn.Linear layout); bias: N or NULL;
 *   Y: R rows of N floats, row stride ldy.  act: 0 = identity, 1 = ReLU.  K % 4 == 0, ldx % 4 == 0.
 * Arithmetic: exact-f32 MFMA (v_mfma_f32_16x16x4_f32); launches with >= 256 output tiles of 128 x 128 run on the
 * bf16 matrix path with every fp32 operand cut exactly into three bf16 pieces and six piece products per MAC
 * (fp32-level error, < 2e-6 relative to max|Y| on the tested shapes; the same scheme serves K6 / K6' on dialogues
 * of >= 128 utterances).
 * ------------------------------------------------------------------------- */
int mmdfn_linear(const float* X, const float* W, const float* bias, float* Y, int R, int K, int N,
                 int ldx, int ldy, int act, int accumulate, void* stream);
/* The same projection with the weight rows given as TWO blocks: output columns [0, N1) use W (N1, K) / bias,
 * columns [N1, N) use W2 (N - N1, K) / bias2.  A bidirectional nn.GRU layer keeps one weight_ih / bias_ih
 * parameter per direction (model.py:866,868); the hoisted input contraction of both directions runs as one
 * launch on the parameters themselves, without a concatenated copy per step.  N1 == N: W2 / bias2 unused. */
int mmdfn_linear2(const float* X, const float* W, const float* W2, int N1, const float* bias, const float* bias2,
                  float* Y, int R, int K, int N, int ldx, int ldy, int act, int accumulate, void* stream);

/* ---------------------------------------------------------------------------
 * K1p  the same projection against a weight that arrives as bf16 PIECE PLANES (linear_planes.hip; ABI 15).
 * Replaces the `nn.GRU` input products of `lstm_l` / `rnn_parties` (reference model.py:866,868,1082,1132) and their input
 * gradients at the row counts of the BASELINE configs.  A weight changes once per optimizer step
 * (run_train_erc.py:512), so its three exact bf16 pieces are cut once per step -- one grouped launch for up to 16 weights --
 * into MFMA B-fragment order, and the projection kernel cuts only its own X rows.
 *
 * mmdfn_weight_planes_workspace: bytes of the plane buffer of a B operand with N output columns and K contraction
 *   (ceil(N/32) x ceil(K/16) x 3 pieces x 1 KB; zero-filled outside N x K by the cut).
 * mmdfn_cut_weight_planes: weight i is given as up to two fp32 matrices w1[i] / w2[i] of row stride ld[i] (w2 may be null) and a
 *   mode[i] that says how the B operand (N[i] output columns, K[i] contraction) is read from them:
 *     0  B[n][k] = stored[n][k], the stored rows split at n1[i] (rows [0, n1) in w1, the rest in w2): a forward product against
 *        the two directions' own parameters;
 *     1  B[n][k] = stored[k][n], the stored rows (= k) split at n1[i]: the input gradient dX = dY . W on the same parameters
 *        (and y = x W for a weight stored (K, N): GraphConvolution.weight, model_GCN.py:172);
 *     2  B[n][k] = w1[k][n] for n < n1[i], w2[k][n - n1] beyond: two (K, .) matrices side by side, transposed;
 *     3  as 2 with the contraction index gate-interleaved, k = 4 u + g  <->  stored row g (K / 4) + u ([W_ih | W_hh] of an LSTM
 *        cell, K = 4H; used by the stack-kernel experiment under tools/gcn_planes/).
 *   planes[i]: 16-byte aligned, mmdfn_weight_planes_workspace(N[i], K[i]) bytes.
 * mmdfn_linear_planes:  Y = act(X B^T + bias) (+ Y);  X: R rows of K floats (row stride ldx, 16-byte aligned rows, K % 4 == 0),
 *   bias / bias2 split at n1 as in mmdfn_linear2 (either may be null), Y: R x N (row stride ldy), act: 0 identity, 1 ReLU.
 *   Arithmetic: six bf16 piece products per MAC, fp32 accumulation -- fp32-level error, as mmdfn_linear's many-row form.
 * mmdfn_linear_planes_group (ABI 16): n <= 4 such products in ONE launch (problem i: X[i], planes[i], ... as above; bias / bias2
 *   may be null arrays; R[i] == 0 skips a problem) -- the hoisted input contractions of the context and the party GRU of one
 *   layer (model.py:866-868, 1082, 1132) and their input gradients, which do not depend on each other.  Every problem keeps the
 *   tile form it takes alone, so its results are the bits of its own mmdfn_linear_planes launch.  mask (may be null, entries may
 *   be null): R[i] x N[i] keep flags (0 / 1, contiguous); Y_i is multiplied by mask_i * mask_scale after the activation -- the
 *   backward of the dropout between the GRU layers (nn.GRU(dropout=), model.py:866) folded into the input gradient's epilogue.
 * ------------------------------------------------------------------------- */
int64_t mmdfn_weight_planes_workspace(int N, int K);
int mmdfn_cut_weight_planes(int n, const float* const* w1, const float* const* w2, const int* n1, const int* ld,
                            const int* N, const int* K, const int* mode, void* const* planes, void* stream);
int mmdfn_linear_planes(const float* X, const void* planes, const float* bias, const float* bias2, int n1, float* Y, int R,
                        int K, int N, int ldx, int ldy, int act, int accumulate, void* stream);
int mmdfn_linear_planes_group(int n, const float* const* X, const void* const* planes, const float* const* bias,
                              const float* const* bias2, const int* n1, float* const* Y, const int* R, const int* K,
                              const int* N, const int* ldx, const int* ldy, int act, int accumulate,
                              const float* const* mask, float mask_scale, void* stream);

/* A GROUP of few-row projections in one launch (linear_small.hip; n <= 8 problems, K <= 768, K % 4 == 0):
 *   Y_p = act(X_p W_p^T + b_p) (+ Y_p)      X_p: R_p rows of K_p floats (stride ldx), Y_p: R_p x N_p (stride ldy)
 *   kmajor[p] == 0: W_p is (N, K) with k-contiguous rows (row stride ldw), given as two row blocks W / W2 split at
 *                   N1 (N1 == N: one block), biases likewise;
 *   kmajor[p] != 0: W_p is (K, N) with n-contiguous rows (row stride ldw): the input gradient dX = dY . Wcat of a dense
 *                   layer reads the stacked weight as stored.
 * Replaces the modality projections model.py:1065,1094,1129, the hoisted GRU input contractions model.py:1082,1132
 * and their autograd input gradients at BASELINE cfg2-cfg4 row counts.  mmdfn_linear_group_supported: 1 if covered. */
int mmdfn_linear_group_supported(int R, int K, int N);
int mmdfn_linear_group(int n, const float* const* X, const float* const* W, const float* const* W2, const int* N1,
                       const float* const* bias, const float* const* bias2, float* const* Y, const int* R,
                       const int* K, const int* N, const int* ldx, const int* ldw, const int* ldy,
                       const int* kmajor, const int* accumulate, int act, void* stream);
/* The same with an out-of-place addend: problem p computes Y_p = act(X_p W_p (+ b)) + Z_p for Z[p] != NULL (Z_p: R_p rows of
 * N_p floats, row stride ldz[p]; read only -- an autograd node adds onto an incoming gradient without writing into a tensor
 * it does not own), and behaves like mmdfn_linear_group for Z[p] == NULL (Z == NULL: for every problem). */
int mmdfn_linear_group_addend(int n, const float* const* X, const float* const* W, const float* const* W2, const int* N1,
                              const float* const* bias, const float* const* bias2, float* const* Y, const float* const* Z,
                              const int* ldz, const int* R, const int* K, const int* N, const int* ldx, const int* ldw,
                              const int* ldy, const int* kmajor, const int* accumulate, int act, void* stream);

/* ---------------------------------------------------------------------------
 * Secondary fusion modules (fusion.hip); their dense projections go through mmdfn_linear_group.
 *   MFN, reference model_fusion.py:62-120, per timestep and batch row:
 *     softmax_scale: att = softmax(z, dim=1) (saved), out = att * c          (:96-97; z, c, att, out: R x W)
 *     mfn_mem      : mem' = sigmoid(v1) mem + sigmoid(v2) tanh(u)            (:98-102; n elements; saved: 3 n floats)
 *   MMGatedAttention 'general', reference model.py:761-781, per modality pair (m, n) and row:
 *     gated_pair   : z = sigmoid(w . [x_m | x_n | x_m * x_n] + b) (w: 3 D floats, saved in zs),
 *                    out = z tanh(p_m) + (1 - z) tanh(p_n)   (x: R x D, p / out: R x C);
 *                    backward also writes dpre[r] = d loss / d (gate pre-activation)
 *     rowscale_colsum: out[0:3D] = sum_r s[r] [x_m | x_n | x_m * x_n][r], out[3D] = sum_r s[r]   (gate weight / bias gradient)
 * ------------------------------------------------------------------------- */
int mmdfn_softmax_scale_fwd(const float* z, const float* c, float* att, float* out, int R, int W, void* stream);
int mmdfn_softmax_scale_bwd(const float* att, const float* c, const float* dout, float* dz, float* dc, int R, int W,
                            void* stream);
int mmdfn_mfn_mem_fwd(const float* u, const float* v1, const float* v2, const float* mem, float* out, float* saved,
                      int64_t n, void* stream);
int mmdfn_mfn_mem_bwd(const float* saved, const float* mem, const float* dout, float* du, float* dv1, float* dv2,
                      float* dmem, int64_t n, void* stream);
int mmdfn_gated_pair_fwd(const float* xm, const float* xn, const float* w, const float* b, const float* pm,
                         const float* pn, float* out, float* zs, int R, int D, int C, void* stream);
int mmdfn_gated_pair_bwd(const float* xm, const float* xn, const float* w, const float* pm, const float* pn,
                         const float* zs, const float* dout, float* dxm, float* dxn, float* dpm, float* dpn,
                         float* dpre, int R, int D, int C, void* stream);
int mmdfn_rowscale_colsum(const float* s, const float* xm, const float* xn, float* out, int R, int D, void* stream);

/* ---------------------------------------------------------------------------
 * Weight-gradient contraction (autograd of the dense layers on the path: dW = dY^T X, db = sum_r dY):
 *   C[m, n] = sum_r A[r, m] B[r, n]      A: R rows of M floats (stride lda), B: R rows of N floats (ldb)
 *   colsum[m] = sum_r A[r, m]            (optional, NULL to skip)
 * Split over r across workgroups; `workspace` must hold splits*(M*N + M) floats
 * (splits from mmdfn_gemm_tn_splits).  M, N, lda, ldb multiples of 4.  C: M rows, stride ldc.
 * ------------------------------------------------------------------------- */
int mmdfn_gemm_tn_splits(int R, int M, int N);
int mmdfn_gemm_tn(const float* A, const float* B, float* C, float* colsum, float* workspace,
                  int R, int M, int N, int lda, int ldb, int ldc, int splits, void* stream);

/* ---------------------------------------------------------------------------
 * The same contraction for up to 8 independent problems in ONE launch pair, with an optional row shift on B:
 *   C_p[m, n] = sum_r A_p[r, m] * B_p[r + bshift_p, n]     (rows of B_p outside [0, R_p) count as zero)
 *   colsum_p[m] = sum_r A_p[r, m]                           (colsum may be NULL, or an array with NULL entries)
 * Replaces the recurrent-weight gradients of nn.GRU on the path (autograd of model.py:866,868: for every GRU,
 * layer and direction dW_hh = sum_t dgh_t (x) h_{t-1}, db_hh = sum_t dgh_t): h_{t-1} is the output sequence
 * shifted by one time step (bshift = -rows_per_step forward, +rows_per_step reverse), and with the shift inside
 * the kernel A spans every row so its column sums are the full bias gradient.  A, B, C, colsum, R .. bshift are
 * HOST arrays of length n (device pointers / ints).  workspace: mmdfn_gemm_tn_grouped_workspace(n, R, M, N) floats.
 * ------------------------------------------------------------------------- */
int64_t mmdfn_gemm_tn_grouped_workspace(int n, const int* R, const int* M, const int* N);
int mmdfn_gemm_tn_grouped(int n, const float* const* A, const float* const* B, float* const* C, float* const* colsum,
                          const int* R, const int* M, const int* N, const int* lda, const int* ldb, const int* ldc,
                          const int* bshift, float* workspace, void* stream);

/* ---------------------------------------------------------------------------
 * Fused stages of the GCNII "dynamic fusion" stack (GCNII_lyc.forward model_GCN.py:444-488, GraphConvolution.forward
 * :176-189): each is ONE launch whose dense contraction runs as exact-f32 MFMA with the pointwise work in its
 * prologue / epilogue (csrc/gcn_stack.hip).  R rows (= M * N graph nodes), H = hidden width (<= 100, multiple of 4),
 * F = input width (<= 256, multiple of 4); every mask is a float keep-mask (0 / 1) multiplied by mscale = 1/(1-p) where
 * it is used, NULL = ones.
 *
 * input stage (model_GCN.py:453-456):  xd = x (.) mx (row stride ldxd);  h0 = relu(xd W0^T + b0);  cur0 = h0 (.) m0
 *   bwd:  dpre = (dcur0 (.) m0 + dh0) (.) [h0 > 0]  (operand of dW0 / db0);  dx = (dpre W0 + dxd) (.) mx,
 *         dxd = gradient reaching xd directly (row stride lddxd) or NULL.
 * gate (K8, model_GCN.py:463-467, nn.LSTM seq_len 1, gate order i f g o, W_ih / W_hh (4H, H), bias = bsum + bsum2
 *   (b_ih, b_hh; bsum2 may be NULL):
 *   fwd:  (h_out, c_out) = LSTMCell(q, (h, c));  h = c = NULL is the zero state;  gates (R, 4H) = gate ACTIVATIONS.
 *   bwd:  dh' = dh_a + dh_b (either NULL), dc_next (NULL = 0) -> dG (R, 4H) pre-activation gradients (operand of
 *         dW_ih, dW_hh, db), dc_prev, dq = dG W_ih + dres (row stride lddres, NULL = 0), dh_prev = dG W_hh;  has_h = 0: the incoming
 *         state was zero, dc_prev / dh_prev are not produced.
 * layer (K7, model_GCN.py:178-186,469-472, W (2H, H) as stored by GraphConvolution):
 *   fwd:  pre = theta [hi | h0] W + (1-theta)((1-alpha) hi + alpha h0);  out = relu(pre) (.) m + q (row stride ldo, q
 *         NULL = 0);  gmask = m (.) [pre > 0].
 *   bwd:  dP = theta dout (.) gmask (operand of dW = [hi | h0]^T dP; dout row stride lddo);
 *         dhi = dP W[:H]^T + (1-theta)(1-alpha) gg,  dh0 (+)= dP W[H:]^T + (1-theta) alpha gg,  gg = dout (.) gmask;
 *         acc_h0 != 0 accumulates into dh0 (h0 feeds every layer).
 * ------------------------------------------------------------------------- */
int mmdfn_gcn_input_fwd(const float* x, const float* mx, const float* W0, const float* b0, const float* m0, float* xd,
                        float* h0, float* cur0, int R, int F, int H, int ldxd, float mscale, void* stream);
int mmdfn_gcn_input_bwd(const float* dcur0, const float* m0, const float* dh0, const float* h0, const float* W0,
                        const float* dxd, const float* mx, float* dpre, float* dx, int R, int F, int H, int lddxd,
                        float mscale, void* stream);
int mmdfn_lstm_gate_fwd(const float* q, const float* h, const float* c, const float* Wih, const float* Whh,
                        const float* bsum, const float* bsum2, float* gates, float* h_out, float* c_out, int R, int H,
                        void* stream);
/* the same with a row stride ldh (floats, >= H, % 4 == 0) on BOTH h and h_out: the layers' hidden states live as column blocks of
 * one (R, nl H) buffer, which is then the Y operand of ONE mmdfn_tile_outer(d = nl H) for the whole stack (model_GCN.py:461-472
 * shares `adj` across the layers: dA = sum_l dhi_l zin_l^T is one contraction of width nl H) */
int mmdfn_lstm_gate_fwd_ld(const float* q, const float* h, const float* c, const float* Wih, const float* Whh,
                           const float* bsum, const float* bsum2, float* gates, float* h_out, float* c_out, int R, int H,
                           int ldh, void* stream);
/* The cell's weights as bf16 piece planes for the many-row form of the forward launch (ABI 14).  The LSTM cell of the reasoning
 * module is shared by all layers of a stack and constant inside a step (model_GCN.py:466), so a caller whose launches take the
 * many-row form (mmdfn_lstm_gate_takes_planes(R, H) != 0) cuts it ONCE per forward pass -- mmdfn_lstm_gate_cut_weights into
 * mmdfn_lstm_gate_planes_workspace(H) floats -- and hands the planes to every layer's mmdfn_lstm_gate_fwd_pre (same operands
 * as mmdfn_lstm_gate_fwd_ld; planes = NULL, h = NULL or a launch of another form: identical to mmdfn_lstm_gate_fwd_ld). */
int64_t mmdfn_lstm_gate_planes_workspace(int H);
int mmdfn_lstm_gate_cut_weights(const float* Wih, const float* Whh, float* planes, int H, void* stream);
int mmdfn_lstm_gate_takes_planes(int R, int H);
int mmdfn_lstm_gate_fwd_pre(const float* q, const float* h, const float* c, const float* Wih, const float* Whh,
                            const float* bsum, const float* bsum2, float* gates, float* h_out, float* c_out, int R, int H,
                            int ldh, const float* planes, void* stream);
int mmdfn_lstm_gate_bwd(const float* gates, const float* c_prev, const float* c_new, const float* dh_a, const float* dh_b,
                        const float* dc_next, const float* Wih, const float* Whh, const float* dres, float* dG,
                        float* dc_prev, float* dq, float* dh_prev, int R, int H, int has_h, int lddres, void* stream);
int mmdfn_gcnii_layer_fwd(const float* hi, const float* h0, const float* W, const float* q, const float* m, float* out,
                          float* gmask, float theta, float alpha, int R, int H, int ldo, float mscale, void* stream);

/* ABI 16: K6 + K7 forward of one layer in ONE launch for short dialogues (L <= 128, M <= 3, H <= 100, H % 4 == 0, at most 4 096
 * (dialogue, modality, 32-row strip) workgroups):  hi = A_hat . z (written out: the backward pass contracts against it; z rows
 * have stride ldz), then mmdfn_gcnii_layer_fwd's update of the same rows.  Same operands as mmdfn_propagate (block-tile
 * adjacency, layout arrays) and mmdfn_gcnii_layer_fwd (h0, W, q, m: row stride H; out: row stride ldo; gmask).  Returns -2 when
 * the shape is not covered: the caller runs the two launches.  (csrc/gcn_small.hip; model_GCN.py:178-189) */
int mmdfn_prop_layer_fwd(const float* tiles, const float* cross, const float* zin, int ldz, const int32_t* dia_len,
                         const int32_t* row_start, const int64_t* tile_base, int B, int M, int N, int max_len,
                         const float* h0, const float* W, const float* q, const float* m, float* hi, float* out,
                         float* gmask, float theta, float alpha, int H, int ldo, float mscale, void* stream);
int mmdfn_gcnii_layer_bwd(const float* dout, const float* gmask, const float* W, float* dP, float* dhi, float* dh0,
                          float theta, float alpha, int R, int H, int lddo, int acc_h0, void* stream);
/* the same with a row stride lddhi on dhi (the X operand of the stack's single mmdfn_tile_outer, see mmdfn_lstm_gate_fwd_ld) */
int mmdfn_gcnii_layer_bwd_ld(const float* dout, const float* gmask, const float* W, float* dP, float* dhi, float* dh0,
                             float theta, float alpha, int R, int H, int lddo, int acc_h0, int lddhi, void* stream);

/* ---------------------------------------------------------------------------
 * Batch form: EVERY weight-gradient contraction of a training step in one launch pair (they feed nothing but the
 * optimizer, so the host queues them during backward and issues them at its end; run_train_erc.py:208 autograd).
 *   segment s (0 <= s < nseg):  A_s (R_s rows, stride lda_s), B_s (R_s rows, stride ldb_s), row shift bshift_s on B,
 *                               contributes  sum_r A_s[r, :]^T B_s[r + bshift_s, :]  to output out[s];
 *   output o (0 <= o < nout):   C_o (M_o x N_o, stride ldc_o) (+)= sum of its segments; colsum_o / colsum2_o (M_o
 *                               floats or NULL; two destinations because b_ih and b_hh of the LSTM gate share one
 *                               gradient) (+)= column sums of the segments' A;  accumulate_o != 0 adds to the
 *                               existing contents.  Segments of one output are extra splits of one slab stack: the
 *                               layer-shared LSTM gate (model_GCN.py:466) gets one segment per GCN layer and no
 *                               gradient-accumulation kernel.
 * nseg, nout <= 40.  All arrays are HOST arrays.  workspace: mmdfn_gemm_tn_batch_workspace(...) floats.
 * Kernel form per batch (an implementation choice, results do not depend on it beyond fp32 summation order): 64 x 64 / 64 x 112
 * output tiles with the rows split over workgroups, or -- when the batch holds enough long segments (>= 8192 rows, a multiple
 * of 16, 64-128 or 336-448 output rows, 64-112 columns, no shift) -- one workgroup per whole-output slab for those.
 * ------------------------------------------------------------------------- */
int64_t mmdfn_gemm_tn_batch_workspace(int nseg, const int* R, const int* out, int nout, const int* M, const int* N);
int mmdfn_gemm_tn_batch(int nseg, const float* const* A, const float* const* B, const int* R, const int* lda,
                        const int* ldb, const int* bshift, const int* out, int nout, float* const* C,
                        float* const* colsum, float* const* colsum2, const int* M, const int* N, const int* ldc,
                        const int* accumulate, float* workspace, void* stream);
/* The same batch whose reduction launch ALSO sums `next` slab stacks written by other kernels (ABI 13): stack e holds
 * ext_splits[e] slabs of ext_M[e] x ext_N[e] floats (ext_part[e], summed into ext_C[e] of row stride ext_ldc[e]; ext_N[e] = 0:
 * none) and / or of ext_M[e] floats (ext_colpart[e], summed into ext_colsum[e]; NULL: none), in slab order (bit-reproducible);
 * ext_accumulate[e] != 0 adds to the destination.  Producers: mmdfn_head_bwd_partial (the classifier's dW / db, reference
 * model.py:1337), mmdfn_colsum_partial (the bias gradient of the project-then-gather node, model.py:1082).  nseg = nout = 0 with
 * next > 0 runs the reduction alone; nout + next <= 40. */
int mmdfn_gemm_tn_batch_ext(int nseg, const float* const* A, const float* const* B, const int* R, const int* lda,
                            const int* ldb, const int* bshift, const int* out, int nout, float* const* C,
                            float* const* colsum, float* const* colsum2, const int* M, const int* N, const int* ldc,
                            const int* accumulate, float* workspace, int next, const float* const* ext_part,
                            const float* const* ext_colpart, float* const* ext_C, float* const* ext_colsum, const int* ext_M,
                            const int* ext_N, const int* ext_ldc, const int* ext_splits, const int* ext_accumulate,
                            void* stream);

/* Weight-gradient RIDERS of the GRU backward recurrence launch (ABI 17).  The backward recurrence of nn.GRU (reference
 * model.py:866-868: autograd of the context / party GRUs) keeps one CU per sequence busy for ~T x 0.75 us and leaves the other
 * CUs idle (IEMOCAP batch of 16: 160 of 256).  mmdfn_wgrad_riders_stage takes the arguments of mmdfn_gemm_tn_batch, plans the
 * batch and allocates its slabs exactly as that call would, but -- when the batch runs on the bf16-piece form, has at most 16
 * segments and nothing is staged yet -- does NOT launch it: the next mmdfn_gru_seq_bwd call on a one-sequence-per-workgroup
 * launch runs the batch's tiles as extra workgroups of the recurrence launch (never on a CU that holds a recurrence: the
 * launch's LDS request keeps every CU to one workgroup); the slab reduction follows later (mmdfn_wgrad_riders_drain).  A batch that cannot ride is
 * launched by the stage call itself.  mmdfn_wgrad_riders_staged: 1 while a batch waits; mmdfn_wgrad_riders_flush launches a
 * waiting batch the ordinary way (call it where no GRU backward launch will follow).  The operands and the workspace must stay
 * valid until the launch that consumes them has been issued on `stream` (the same stream for all three calls).  Results are
 * those of mmdfn_gemm_tn_batch bit for bit (same tiles, same slab order). */
int mmdfn_wgrad_riders_stage(int nseg, const float* const* A, const float* const* B, const int* R, const int* lda,
                             const int* ldb, const int* bshift, const int* out, int nout, float* const* C,
                             float* const* colsum, float* const* colsum2, const int* M, const int* N, const int* ldc,
                             const int* accumulate, float* workspace, void* stream);
int mmdfn_wgrad_riders_staged(void);
/* CUs the mmdfn_gru_seq_bwd launch of these groups leaves idle if it is of the kind that takes riders, else 0: stage a batch
 * only when this is > 0, and size it for that many CUs over the recurrence's ~0.75 us x T. */
int mmdfn_gru_seq_bwd_idle_cus(int ngroups, const int* rows);
int mmdfn_gru_seq_bwd_step_ns(int ngroups, const int* rows);    /* ns per recurrence step of that launch (750 / 2300: the two forms) */
int mmdfn_wgrad_riders_flush(void* stream);
/* The slab reduction of a batch that rode (or that mmdfn_wgrad_riders_flush launched) is not a launch of its own either: it joins the reduction launch of the next
 * mmdfn_gemm_tn_batch / _ext call on the stream (unless that call writes one of the same gradients: then it goes first, alone).
 * mmdfn_wgrad_riders_drain(stream, 0) reduces what is still waiting -- call it once at the end of the backward pass, behind the
 * last batch; the slabs (the staged batches' workspaces) must stay valid until then.  discard != 0 forgets staged and waiting
 * work instead (after a backward pass that raised). */
int mmdfn_wgrad_riders_drain(void* stream, int discard);

/* ---------------------------------------------------------------------------
 * Fused Adam step over flat fp32 buffers (replaces torch.optim.Adam(lr, weight_decay=l2).step(),
 * run_train_erc.py:512,212): L2 folded into the gradient, bias-corrected, `step` = 1, 2, ...
 * p, m, v updated in place; n elements (16-byte aligned buffers).
 * ------------------------------------------------------------------------- */
int mmdfn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, void* stream);

/* ---------------------------------------------------------------------------
 * K9  classifier head (reference model.py:1328-1337: dropout_ -> ReLU -> smax_fc -> log_softmax) as one launch each way:
 *   z = relu(F (.) mask * mscale);  logp = log_softmax(z W^T + b)       F: N rows of Wd floats (row stride ldf),
 *   mask: (N, Wd) 0 / 1 keep flags or NULL, W: (C, Wd) nn.Linear layout, C <= 8 classes, Wd % 4 == 0.
 *   bwd: dF (row stride lddf), dW (C, Wd), db (C) from dlogp; workspace: mmdfn_head_bwd_workspace(Wd, C) floats;
 *   dW / db are reduced in a fixed order (bit-reproducible).
 *   split = 0: F / dF are plain (N, Wd) matrices.  split = Wm > 0 (Wm % 4 == 0, Wd % Wm == 0): F / dF are Wd / Wm
 *   consecutive (N, Wm) matrices of row stride ldf / lddf, read as their column-wise concatenation -- the (M, N, Wm)
 *   output of the graph stack used as cat([F[0], .., F[M-1]], -1) (model_mm.py:113-117) without materialising it.
 * ------------------------------------------------------------------------- */
int mmdfn_head_fwd(const float* F, const float* mask, const float* W, const float* bias, float* logp, int64_t N, int Wd,
                   int C, int ldf, int split, float mscale, void* stream);
int64_t mmdfn_head_bwd_workspace(int Wd, int C);
int mmdfn_head_bwd(const float* dlogp, const float* logp, const float* F, const float* mask, const float* W, float* dF,
                   float* dW, float* db, float* workspace, int64_t N, int Wd, int C, int ldf, int lddf, int split,
                   float mscale, void* stream);
/* mmdfn_head_bwd without its slab reduction (ABI 13): dF is complete, `workspace` holds mmdfn_head_bwd_groups() slabs of dW
 * ([groups][C][Wd]) followed by as many of db ([groups][C]) for mmdfn_gemm_tn_batch_ext. */
int mmdfn_head_bwd_groups(void);
int mmdfn_head_bwd_partial(const float* dlogp, const float* logp, const float* F, const float* mask, const float* W, float* dF,
                           float* workspace, int64_t N, int Wd, int C, int ldf, int lddf, int split, float mscale,
                           void* stream);

/* ---------------------------------------------------------------------------
 * K10  FocalLoss (reference loss.py:14-34) as one launch each way:
 *   loss = reduce_i( -(1 - pt_i)^gamma * alpha[t_i] * log_prob[i, t_i] ),  pt_i = exp(log_prob[i, t_i]) held constant
 *   (the reference detaches it), reduce = mean if size_average else sum; alpha (C) may be NULL.
 *   fwd writes loss[0] and coef[i] = -(1 - pt_i)^gamma * alpha[t_i] * (1/N or 1); the sum is reduced in a fixed order
 *   (bit-reproducible).  bwd: dlogp[i, c] = (c == t_i) * coef[i] * dloss[0]   (all N*C entries written).
 *   log_prob: (N, C) contiguous fp32; target: N int64.
 * ------------------------------------------------------------------------- */
int mmdfn_focal_loss_fwd(const float* logp, const int64_t* target, const float* alpha, float* loss, float* coef,
                         int64_t N, int C, float gamma, int size_average, void* stream);
/* the same, additionally writing dlogp_unit (N, C) = d loss / d log_prob for an upstream gradient of exactly 1 (NULL: not
 * wanted): loss.backward() then needs no launch for the loss (mmdfn_focal_loss_bwd remains for other upstream gradients) */
int mmdfn_focal_loss_fwd_grad(const float* logp, const int64_t* target, const float* alpha, float* loss, float* coef,
                              float* dlogp_unit, int64_t N, int C, float gamma, int size_average, void* stream);
int mmdfn_focal_loss_bwd(const float* coef, const int64_t* target, const float* dloss, float* dlogp, int64_t N, int C,
                         void* stream);
/* the same with rows to leave out (target == ignore_index: no loss, zero gradient row; the mean divides by the rows that
 * count, on the device).  An extension of the reference's FocalLoss (loss.py has no ignore_index): train.StepGraphCache pads
 * a batch to its size bucket with a dummy dialogue labelled ignore_index, so one captured step serves batches with different
 * numbers of utterances.  scale_out: 1 float (1 / count, or 1 without size_average), handed to the backward. */
int mmdfn_focal_loss_fwd_ignore(const float* logp, const int64_t* target, const float* alpha, float* loss, float* coef,
                                float* scale_out, int64_t N, int C, float gamma, int size_average, int64_t ignore_index,
                                void* stream);
int mmdfn_focal_loss_bwd_ignore(const float* coef, const int64_t* target, const float* dloss, const float* scale,
                                float* dlogp, int64_t N, int C, int64_t ignore_index, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMDFN_HIP_H */
