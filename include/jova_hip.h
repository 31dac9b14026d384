/* jova_hip.h -- C ABI of libjova_hip.so: the MI355X (gfx950) kernels behind the JOVA multi-view DTI training hot path.
 *
 * The reference (bbrighttaer/jova) has no FFI: its hot path is stock torch ops called from the nn.Modules of
 * jova/nn/{layers,models}.py.  Each entry point below replaces the torch-op sequence of one of those modules
 * (cited as file:line relative to the reference root); jova_amd/nn binds them with ctypes and keeps the
 * reference's module names, constructor / forward signatures and state-dict keys.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its comment says "host"; tensors are dense row-major fp32 unless typed
 *   - `stream` is a hipStream_t (0 = default stream); every call only enqueues work: no allocation, no sync
 *   - return value 0 = ok, non-zero = error (jova_last_error() holds the message for the calling thread)
 *   - "accumulates" = adds into the destination (parameter gradients), everything else overwrites
 */
#ifndef JOVA_HIP_H
#define JOVA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JOVA_HIP_ABI_VERSION 1

const char* jova_last_error(void);
int jova_abi_version(void);

/* ---- GraphConvLayer: jova/nn/layers.py:282-347 ---------------------------------------------------------------
 * atoms [N, f_in] with row stride ld_atoms (int64 when atoms_is_i64, the featuriser's dtype for layer 1; else fp32),
 * sorted degree-major as
 * ConvMol.agglomerate_mols leaves them (jova/feat/mol_graphs.py:290-296); deg_slice int64 [11,2] (start,size);
 * adj: HOST array of 10 device pointers, adj[d-1] = int32 [n_d, d]; weight/bias: HOST arrays of 21 device pointers
 * in linear_layers order (rel_1, self_1, ..., rel_10, self_10, self_0), weight[k] = [f_out, f_in].
 * out [N, f_out] = act( W_rel[d] . sum_nbr x + b_rel[d] + W_self[d] . x + b_self[d] ).
 * bn_stats (optional, fp64 [2, f_out]): column sum / sum of squares of `out` for the BatchNorm1d that follows.
 * packed_ws: scratch of jova_graphconv_workspace_floats(f_in, f_out) floats.
 * deg_slice_host: optional HOST copy of deg_slice (see jova_graphconv_bwd).
 * N == 0 (an empty batch) is legal in every graph entry point: nothing is launched and the per-atom pointers may be
 * null. */
int jova_graphconv_fwd(const void* atoms, int atoms_is_i64, int64_t ld_atoms, int64_t n_atoms, int f_in, int f_out,
                       const int64_t* deg_slice, const int32_t* const* adj, const float* const* weight,
                       const float* const* bias, int relu, float* out, double* bn_stats, float* packed_ws,
                       const int64_t* deg_slice_host, void* stream);
/* int64 atom features [n, f] -> fp32 rows padded with zeros to ld floats (ld % 4 == 0 enables 16-byte row loads in
 * jova_graphconv_fwd / _bwd; done once per batch: the features of a batch never change) */
int jova_atoms_to_f32(const int64_t* atoms, int64_t n_atoms, int f, int ld, float* out, void* stream);
/* floats of scratch `packed_ws` (the weights re-packed into MFMA fragment order, rebuilt every call) */
size_t jova_graphconv_workspace_floats(int f_in, int f_out);

/* Backward of the above.  dy [N, f_out]; y = forward output when the layer applied ReLU (mask), else NULL.
 * dweight/dbias: HOST arrays of 21 device pointers, accumulate.  dx [N, f_in] (optional, fp32 atoms only) needs
 * gbuf [N, f_in] scratch and `weight`.  sym_flag: device int from jova_adj_symmetry (NULL or 0 -> atomic scatter).
 * deg_slice_host: optional HOST copy of deg_slice (the collator has it: mol_graphs.py:304-314); with it the work is
 * split on the host into equally long workgroups that are all resident at once, without it into fixed tile groups.
 * block_order (optional): as in jova_graphpool_fwd, for the neighbour-gather launch of dx. */
int jova_graphconv_bwd(const void* atoms, int atoms_is_i64, int64_t ld_atoms, int64_t n_atoms, int f_in, int f_out,
                       const int64_t* deg_slice, const int32_t* const* adj, const float* const* weight,
                       const float* dy, const float* y_or_null, float* const* dweight, float* const* dbias,
                       float* dx_or_null, float* gbuf_or_null, const int* sym_flag, const int64_t* deg_slice_host,
                       float* dw_scratch_or_null, const int32_t* block_order_or_null, void* stream);
/* floats of the optional `dw_scratch` above: an aligned image of the 21 weight gradients that the kernel accumulates in
 * when f_in is not a multiple of 32 (rows of 75 floats straddle 128-byte lines and slow the float atomics 10x);
 * contents on entry are ignored, the result is added to dweight by a fold kernel. */
size_t jova_graphconv_bwd_scratch_floats(void);

/* *flag <- 1 iff the adjacency is a simple undirected graph (no self loops / duplicates, i in nbr(j) <=> j in nbr(i)):
 * then the backward scatters above and below run as gathers.  Always true for ConvMol bond graphs. */
int jova_adj_symmetry(int64_t n_atoms, const int64_t* deg_slice, const int32_t* const* adj, int* flag, void* stream);

/* ---- GraphPool: jova/nn/layers.py:356-393 -------------------------------------------------------------------
 * out[i] = max(x[i], max over neighbours); ties -> first candidate (self, then adjacency order) like torch.max.
 * argmax uint8 [N, f]: 0 = self, k = k-th neighbour.  Backward: sym_flag = device flag of jova_adj_symmetry;
 * host_knows_symmetric = 1 when the caller has read that flag as 1 (selects the pure-gather kernel, no memset);
 * rev_pos (optional, from jova_adj_reverse_pos, symmetric graphs only) spares that kernel the search of every
 * neighbour's list.
 * block_order (optional, DEVICE int32 [ceil(N/16)], a permutation of the 16-row block numbers; jova_block_order
 * makes it on the host once per batch): which block each workgroup takes, arranged so that an atom's row and its
 * neighbours' reads of it meet in one XCD's L2.  Results do not depend on it; NULL = blocks in row order.
 * row_table (optional, DEVICE, from jova_row_table for the SAME block order and adjacency; takes the place of
 * block_order when given): row, degree, first four neighbours and their reverse positions of every row in launch
 * order, 32 bytes a row, so the gathers start after one read instead of four dependent ones.  Results do not depend
 * on it. */
int jova_graphpool_fwd(const float* x, int64_t n_atoms, int f, const int64_t* deg_slice, const int32_t* const* adj,
                       float* out, uint8_t* argmax, const int32_t* block_order_or_null,
                       const int32_t* row_table_or_null, void* stream);
int jova_graphpool_bwd(const float* dy, const uint8_t* argmax, int64_t n_atoms, int f, const int64_t* deg_slice,
                       const int32_t* const* adj, float* dx, const int* sym_flag, int host_knows_symmetric,
                       const uint8_t* rev_pos_or_null, const int32_t* block_order_or_null,
                       const int32_t* row_table_or_null, void* stream);
/* table int32 [16 * ceil(n_atoms/16), 8] (16-byte aligned), slot = launch block * 16 + row in block:
 * { row or -1, degree, neighbour 0..3 (last one repeated past the degree), rev_pos 0..3 one per byte (0xFF: none), 0 }.
 * rev_pos NULL: the packed positions are 0xFF (enough for the forward).  No counterpart in the reference: a latency
 * measure for the gather kernels, built once per batch. */
int jova_row_table(int64_t n_atoms, const int64_t* deg_slice, const int32_t* const* adj, const uint8_t* rev_pos_or_null,
                   const int32_t* block_order_or_null, int32_t* table, void* stream);
/* host-only (no GPU call): order_host receives ceil(n_atoms/16) ints from the HOST copy of deg_slice.  No
 * counterpart in the reference: an MI355X placement hint (8 XCDs, one L2 each). */
int jova_block_order(const int64_t* deg_slice_host, int64_t n_atoms, int32_t* order_host);
/* rev_pos uint8 [sum_d n_d * d] (laid out like adj_1 | adj_2 | .. | adj_10): for atom i's n-th neighbour j the 1-based
 * position of i in j's own list, 0xFF if absent.  A property of the batch graph (mol_graphs.py:340-378 builds both
 * directions of every bond); built once per batch. */
int jova_adj_reverse_pos(int64_t n_atoms, const int64_t* deg_slice, const int32_t* const* adj, uint8_t* rev_pos,
                         void* stream);

/* ---- batch merge: ConvMol.agglomerate_mols, jova/feat/mol_graphs.py:272-383 (called per mini-batch from
 * process_gconv_view_data, jova/data/data.py:290).  HOST-only, no GPU call; every pointer is a HOST pointer.
 * Per molecule m (atoms already degree-sorted as ConvMol.__init__ leaves them, mol_graphs.py:54-201): n_atoms[m],
 * feats[m] int64 [n, n_feat], deg_list[m] int64 [n] (non-decreasing), nbr_flat[m] int64 [sum deg] = the neighbour lists
 * in atom order, local ids, n_nbr[m] = length of nbr_flat[m] (checked against sum deg: an inconsistent molecule is an
 * error, not a read past the array).  Outputs (the caller's upload image, typically pinned): nodes [N, n_feat], deg_slice [11,2],
 * membership [N], adj = adj_1 | .. | adj_10 concatenated (adj_d int32 [n_d, d]), mol_off int32 [n_mols + 1] (prefix sums
 * of n_atoms), block_order (optional, as jova_block_order), symmetric (optional: 1 iff the adjacency is a simple
 * undirected graph, the host counterpart of jova_adj_symmetry), atoms_f32 (optional: the fp32 copy of nodes with rows
 * zero padded to ld_f32 floats, as jova_atoms_to_f32), rev_pos (optional, uint8 [sum_d n_d * d], as
 * jova_adj_reverse_pos).  Same integers as the reference, bit for bit. */
int jova_collate_gconv(int n_mols, const int32_t* n_atoms, const int64_t* const* feats, const int64_t* const* deg_list,
                       const int64_t* const* nbr_flat, const int64_t* n_nbr, int n_feat, int64_t* nodes_out,
                       int64_t* deg_slice_out,
                       int64_t* membership_out, int32_t* adj_out, int32_t* mol_off_out, int32_t* block_order_out_or_null,
                       int* symmetric_out_or_null, float* atoms_f32_out_or_null, int ld_f32,
                       uint8_t* rev_pos_out_or_null);

/* ---- GraphGather2D: jova/nn/layers.py:427-437 ---------------------------------------------------------------
 * mol_off int32 [batch+1] = prefix sums of n_atoms_lst; out [a_max, batch, g] = zero-padded contiguous chunks. */
int jova_gather2d_fwd(const float* x, const int32_t* mol_off, int a_max, int batch, int g, float* out, void* stream);
int jova_gather2d_bwd(const float* dout, const int32_t* mol_off, int a_max, int batch, int g, float* dx, void* stream);

/* ---- nonsat_activation: jova/nn/models.py:458-478 -----------------------------------------------------------
 * Newton-style iteration whose trip count depends on a global mean; decided on the device in ONE launch.
 * out = y_final, dfac = d y_final / d x (= 1 / (y_prev^2 + 1)), n_eval_dev int32 [3] = (evaluations, arrivals,
 * timeout flag).  workspace: jova_nonsat_workspace_bytes(n) bytes.  Backward: dx = dout * dfac (jova_mul).
 * The launch is one workgroup per CU at most (capped by an occupancy query) with an in-kernel barrier whose spins are
 * bounded: if the workgroups are not all resident (another process holds CUs) the give-up is collective, the flag
 * n_eval_dev[2] is raised and the outputs of the workgroups that gave up are filled with NaN, so the failure reaches
 * the loss / the non-finite check (proj/dti/train_joint_attn.py:468-470) instead of training on a torn tensor. */
size_t jova_nonsat_workspace_bytes(int64_t n);
int jova_nonsat_fwd(const float* x, int64_t n, float ep, int max_iter, float* out, float* dfac, int* n_eval_dev,
                    void* workspace, void* stream);
/* spin bound of the barrier above in ticks of the 100 MHz wall clock (0 restores the default, 2 s); process-wide.
 * No counterpart in the reference; tests use it to provoke the give-up path. */
void jova_nonsat_set_spin_limit(uint64_t ticks);
/* Products of nn.Linear (forward Y = X W^T, and the backward dX = dY W, dW = dY^T X of torch autograd,
 * proj/dti/train_joint_attn.py:456-466) of a GFLOP or more on aligned shapes run on the bf16 matrix pipe at fp32 accuracy:
 * every fp32 operand element split into three bf16 pieces, six piece products, fp32 accumulation (jova_gemm_f32 /
 * jova_linear_bwd_weight choose it by themselves).  mode 0 keeps every product on the fp32 MFMA kernel (as does
 * JOVA_GEMM_NO_BF16X3 in the environment), 1 = backward products only (JOVA_GEMM_BF16X3_BWD_ONLY; the round-2 rule),
 * 2 = forward products too (the default since round 3: DESIGN.md section 4 has the evidence from the imported reference);
 * process-wide.  No counterpart in the reference.
 * Operand range of that form: finite operands with |x| <= 3.3895e38 (the largest bfloat16; an fp32 value above it, i.e.
 * within 2^-8 of FLT_MAX, rounds its top piece to infinity and the outputs it feeds are NaN where an fp32 chain stays
 * finite), full fp32 accuracy for |x| >= 2^-110 (below that the smallest of the three pieces, 2^-16 |x|, leaves the normal
 * range and the product degrades gracefully towards bf16 x 2 accuracy); zeros are exact; a non-finite operand gives NaN
 * (fp32 would give +-inf).  tests/test_gpu_ops.py::test_bf16_pieces_operand_range pins both ends.  Gradients of a
 * training step sit many decades inside this range; data that do not: jova_gemm_set_bf16x3(0). */
void jova_gemm_set_bf16x3(int mode);
/* the switch's current mode (0 / 1 / 2), the environment default resolved: lets a caller put back what it found */
int jova_gemm_get_bf16x3(void);
/* Allocation-free variant of jova_nonsat_fwd.  NO counterpart in the reference: it exists for callers that cannot
 * allocate device memory (code under hipGraph capture, tight serving loops, harnesses that forbid hipMalloc).  Same
 * kernels and results as jova_nonsat_fwd, but the workspace lives in a built-in per-GPU scratch array of
 * 64 + 8 KiB + 8 * jova_nonsat_noalloc_max_n() bytes (16 MiB; n <= jova_nonsat_noalloc_max_n() = 2^21 elements,
 * larger n is rejected with an error).  The call itself allocates nothing.  All calls on one GPU share that array:
 * calls that overlap on one GPU (two streams, two threads) race; serialise them. */
int jova_nonsat_fwd_noalloc(const float* x, int64_t n, float ep, int max_iter, float* out, float* dfac,
                            int* n_eval_dev, void* stream);
int64_t jova_nonsat_noalloc_max_n(void);
int jova_mul(const float* a, const float* b, int64_t n, float* out, void* stream);

/* ---- dense contractions: nn.Linear / bmm call sites of jova/nn/models.py:740-771, :944-963, :996-998 and
 * proj/dti/train_joint_attn.py:167,229-238 -------------------------------------------------------------------
 * C[z](m,n) = epi(alpha * sum_k A[z](m,k) B[z](k,n)), A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn] (one unit
 * stride each), epi: + bias[n], + residual[m*ldr+n], ReLU (relu = 1).  relu = 2 is the ReLU BACKWARD mask: residual is
 * not added, the result is zeroed where residual[m*ldr+n] <= 0.  splitk > 1: partial sums ADDED to C with atomics
 * (no residual / relu).  Batch z = z1*nb2 + z2, offsets z1*s?1 + z2*s?2 (residual shares C's batch strides).
 * Operands are read with 16-byte loads when base, leading dimension and batch strides are multiples of 4 floats and
 * the contiguous extent is one too, or the leading dimension covers the extent rounded up to 4 (padded rows): then
 * the up to 3 floats past the extent of every row must be addressable; their values are ignored. */
int jova_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C,
                  int64_t ldc, int M, int N, int K, float alpha, const float* bias, const float* residual, int64_t ldr,
                  int relu, int splitk, int nb1, int nb2, int64_t sA1, int64_t sA2, int64_t sB1, int64_t sB2,
                  int64_t sC1, int64_t sC2, void* stream);
/* The same product with B split into its bf16 pieces beforehand, once per call instead of once per row tile and wave
 * (A K-contiguous, B a matrix all row tiles share: the forward products x W^T and the data gradients dY W of the
 * linear layers; whole 128 x 128 tiles, K % 32 == 0, no batch, no split-K, bf16-pieces mode on for the product).
 *   jova_gemm_b_image_bytes  size of the image jova_gemm_f32_img would use for this product, 0 when it would run on
 *                            another form (then the image argument is ignored and may be NULL);
 *   jova_gemm_b_image        writes the image (DEVICE memory, 16-byte aligned, 6 bytes per element) of B(k, n) =
 *                            B[k sbk + n sbn]; N % 128 == 0, K % 32 == 0; one small launch on the stream;
 *   jova_gemm_f32_img        jova_gemm_f32 with that image (NULL: exactly jova_gemm_f32).
 * Same arithmetic as the in-kernel split (three round-to-nearest pieces, the six products of weight >= 2^-16, fp32
 * accumulation); only the summation order inside a 32-deep k block differs.  No counterpart in the reference (its
 * products are torch.nn.Linear, jova/nn/models.py:915-960): a workspace-bringing variant for callers that can hold
 * 6 N K bytes per product. */
size_t jova_gemm_b_image_bytes(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, int M,
                               int N, int K, int nb1, int nb2, int splitk);
int jova_gemm_b_image(const float* B, int64_t sbk, int64_t sbn, int N, int K, void* image, void* stream);
int jova_gemm_f32_img(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C,
                      int64_t ldc, int M, int N, int K, float alpha, const float* bias, const float* residual,
                      int64_t ldr, int relu, int splitk, int nb1, int nb2, int64_t sA1, int64_t sA2, int64_t sB1,
                      int64_t sB2, int64_t sC1, int64_t sC2, const void* b_image_or_null, void* stream);

/* ---- nn.BatchNorm1d (training mode) call sites: proj/dti/train_joint_attn.py:158,163,168,232; models.py:926,972
 * stats fp64 [2,c] from jova_colstats (or from jova_graphconv_fwd's fused epilogue). */
int jova_colstats(const float* x, int64_t n_rows, int c, double* stats, void* stream);
int jova_batchnorm_fwd(const float* x, int64_t n_rows, int c, const double* stats, const float* gamma,
                       const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                       int64_t* num_batches, int relu, float* y, float* mean_rstd, void* stream);
/* y = forward output when ReLU was fused (mask), else NULL; dgamma/dbeta accumulate; sums_ws fp64 [2,c] scratch.
 * relu_input_mask != 0: x is itself a ReLU output (GraphConvLayer's built-in activation, layers.py:344-345, in front
 * of the BatchNorm, train_joint_attn.py:157-158) and dx is zeroed where x <= 0, i.e. dx is already the gradient in
 * front of that ReLU; the producer's backward then takes y_or_null = NULL (see jova_graphconv_bwd). */
int jova_batchnorm_bwd(const float* x, const float* dy, const float* y_or_null, const float* mean_rstd,
                       const float* gamma, int64_t n_rows, int c, int relu_input_mask, float* dx, float* dgamma,
                       float* dbeta, double* sums_ws, void* stream);
/* The two parameter gradients of an nn.Linear (jova/nn/models.py: every Linear of the path): dw[n_out, k_in] +=
 * dy[rows, n_out]^T x[rows, k_in] and, when db is given, db[n_out] += column sums of dy.  splitk > 1 splits the rows over
 * workgroups (dw accumulates atomically).  On the aligned GEMM form the column sums ride on the product's own reads of
 * dy; otherwise jova_colsum_add runs behind it. */
int jova_linear_bwd_weight(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, int64_t rows, int n_out, int k_in,
                           float* dw, float* db_or_null, int splitk, void* stream);
/* out[c] += column sums of x [n_rows, c] (bias gradients) */
int jova_colsum_add(const float* x, int64_t n_rows, int c, int64_t ld, float* out, void* stream);

/* ---- nn.LayerNorm with fused residual add: jova/nn/models.py:959-960, :1016-1017 ----------------------------- */
int jova_layernorm_fwd(const float* x, const float* residual_or_null, const float* gamma, const float* beta,
                       int64_t n_rows, int d, float eps, float* y, float* mean_rstd, void* stream);
/* dz = gradient w.r.t. (x + residual); dgamma/dbeta accumulate */
int jova_layernorm_bwd(const float* x, const float* residual_or_null, const float* dy, const float* gamma,
                       const float* mean_rstd, int64_t n_rows, int d, float* dz, float* dgamma, float* dbeta,
                       void* stream);

/* ---- softmax / dropout of nn.MultiheadAttention (jova/nn/models.py:952) and nn.Dropout ------------------------ */
int jova_softmax_fwd(const float* x, int64_t n_rows, int n, int64_t ld, float* y, void* stream);
/* p_dropped (optional) = dropout(p) as the forward made it (MHA's attention dropout sits between softmax and P V):
 * dp is then masked where p_dropped == 0 and scaled by keep_scale = 1 / (1 - prob) inside this kernel. */
int jova_softmax_bwd(const float* p, const float* dp, const float* p_dropped_or_null, float keep_scale, int64_t n_rows,
                     int n, int64_t ld, float* ds, void* stream);
/* y = x * keep / (1-p); keep is a pure function of (seed, offset, element index): call again on dy for backward */
int jova_dropout(const float* x, int64_t n_rows, int n, int64_t ld, float p, uint64_t seed, uint64_t offset, float* y,
                 void* stream);

/* ---- Prot2Vec: jova/nn/models.py:214-242 (_construct_embedding_indices) + :704-715 ---------------------------
 * table int32 [rows, window]: every protein profile, resident; prot_off/prot_len int32 [batch]: first row / segment
 * count of each pair's protein.  out [p_max, batch, window*e] = act(E[idx]); idx_out (optional) int64
 * [batch, p_max, window] = the reference's padded index cube (pad = vocab). */
int jova_prot2vec_fwd(const int32_t* table, const int32_t* prot_off, const int32_t* prot_len, const float* emb,
                      int batch, int p_max, int window, int e, int vocab, int relu, float* out,
                      int64_t* idx_out_or_null, void* stream);
/* demb [vocab+1, e] accumulates; the pad row receives nothing */
int jova_prot2vec_bwd(const int32_t* table, const int32_t* prot_off, const int32_t* prot_len, const float* dout,
                      const float* out, int batch, int p_max, int window, int e, int vocab, int relu, float* demb,
                      void* stream);

/* ---- JointAttention view pooling: jova/nn/models.py:966-971 ---------------------------------------------------
 * x [S, batch, l]; seg_off int32 [n_views+1]; out [batch, n_views*l] */
int jova_viewpool_fwd(const float* x, const int32_t* seg_off, int n_views, int batch, int l, float* out, void* stream);
int jova_viewpool_bwd(const float* dout, const int32_t* seg_off, int n_views, int batch, int l, int s_total, float* dx,
                      void* stream);

/* ---- loss and optimiser step: proj/dti/train_joint_attn.py:464-465, :485-486 ---------------------------------- */
int jova_mse_loss(const float* outputs, const float* weights, const float* target, int n, float* loss,
                  float* dout_or_null, void* stream);
/* Optimizer steps (torch.optim semantics on the flat buffers).  Common tail of the three entry points:
 * skip_flag_or_null = the device flag of jova_nonfinite_flag: while it is non-zero parameters and optimizer state are
 * left untouched (the reference leaves its loop at a NaN loss before backward / optimizer.step, :468-470, so its
 * weights stay at the last finite values; here that decision needs no device->host read); zero_grads != 0 clears the
 * gradient buffer once it has been consumed (optimizer.zero_grad() of the next step, :428, without a second pass). */
int jova_sgd_step(float* params, float* grads, float* momentum_buf_or_null, int64_t n, float lr,
                  float weight_decay, float momentum, float grad_scale, int first_step,
                  const int* skip_flag_or_null, int zero_grads, void* stream);
int jova_adamax_step(float* params, float* grads, float* exp_avg, float* exp_inf, int64_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, float grad_scale, int step,
                     const int* skip_flag_or_null, int zero_grads, void* stream);

/* torch.optim.Adam / RMSprop / Adagrad / Adadelta / ASGD / Rprop (with sgd and adamax above: the whole optimizer table of
 * proj/dti/train_joint_attn.py:276-285) on the flat parameter / gradient buffers.
 * kind 2 = adam (h0 beta1, h1 beta2, h2 eps; state1 exp_avg, state2 exp_avg_sq),
 * 3 = rmsprop (h0 alpha, h1 eps, h2 momentum; state1 square_avg, state2 momentum buffer or NULL when h2 == 0),
 * 4 = adagrad (h0 lr_decay, h1 eps; state1 sum), 5 = adadelta (h0 rho, h1 eps; state1 square_avg, state2 acc_delta),
 * 6 = asgd (h0 lambd, h1 alpha, h2 t0; state1 ax = the averaged iterate; eta and mu follow from `step`),
 * 7 = rprop (h0 eta-, h1 eta+, h3 / h4 smallest / largest step size; state1 previous gradient, state2 per-element step
 *     size, initialised to lr by the caller; weight_decay must be 0).
 * step = 1-based step count; gradient used = grads * grad_scale + weight_decay * params. */
int jova_optim_step(int kind, float* params, float* grads, float* state1, float* state2_or_null, int64_t n,
                    float lr, float weight_decay, float grad_scale, int step, float h0, float h1, float h2, float h3,
                    float h4, const int* skip_flag_or_null, int zero_grads, void* stream);
/* *flag |= 1 if x holds a NaN, |= 2 if it holds an infinity: the fail-fast of proj/dti/train_joint_attn.py:468-470
 * (which tests for NaN) without a device->host read per batch */
int jova_nonfinite_flag(const float* x, int64_t n, int* flag, void* stream);

/* dst[r][c] (+)= src[r][c] inside the source's extent, 0 outside, for r < dst_rows, c < dst_cols (row strides ld_*): pads
 * an operand of ProteinCNN's 242-wide layers (jova/nn/models.py:738-739, :768-771) to the aligned GEMM's 128-multiples,
 * crops a result, or adds the [242, 242] block of a padded weight gradient to the real one (accumulate = 1). */
int jova_copy2d(const float* src, int64_t src_rows, int src_cols, int64_t ld_src, float* dst, int64_t dst_rows,
                int dst_cols, int64_t ld_dst, int accumulate, void* stream);

/* ---- GraphNeuralNet / GraphNeuralNet2D (jova/nn/models.py:352-418), section 8(f3) ----------------------------- */
/* nn.Embedding of the fingerprint ids: out[r][:] = weight[idx[r]][:]; an id outside [0, vocab) (nn.Embedding raises) sets
 * *bad_flag |= 1 and writes NaN.  Backward: dweight[idx[r]][:] += dout[r][:] (float atomics). */
int jova_embedding_fwd(const float* weight, const int64_t* idx, int64_t n, int d, int64_t vocab, float* out,
                       int* bad_flag, void* stream);
int jova_embedding_bwd(const float* dout, const int64_t* idx, int64_t n, int d, int64_t vocab, float* dweight,
                       void* stream);
/* update_fingerprints (models.py:364-372) after the matmul: out = x + y / (m[row] - 1) ("mean"), out = x + y (m NULL,
 * "sum"); x NULL = the scaled y alone (backward). */
int jova_rowscale_add(const float* x_or_null, const float* y, const float* m_or_null, int64_t rows, int cols, float* out,
                      void* stream);
/* The matmul of update_fingerprints itself (models.py:366-371: torch.matmul(adjacency_matrices, hs)) for an adjacency kept
 * in CSR form: out [n_rows, d] = A x, A given by row_ptr int32 [n_rows + 1], col int32 [nnz] (rows of x, < n_src; an index
 * outside yields NaN), val [nnz] (NULL = all ones).  The reference builds the batch's block-diagonal matrix dense
 * ([N, N] fp32, jova/data/data.py:309-335); jova_amd.data keeps the per-molecule non-zeros.  Backward = the same call
 * with the CSR form of A^T. */
int jova_spmm_csr(const int32_t* row_ptr, const int32_t* col, const float* val_or_null, const float* x, int64_t n_rows,
                  int d, int64_t n_src, float* out, void* stream);

/* ---- ProteinRNN (jova/nn/models.py:657-682): single-layer one-directional nn.LSTM, zero initial state, section 8(f3) --
 * gx [T, B, 4H] = x W_ih^T + b_ih + b_hh for all steps (one jova_gemm_f32 by the caller; gate order i | f | g | o),
 * w_hh [4H, H].  Forward writes h [T, B, H], the activated gates [T, B, 4H] and the cell states [T, B, H] (kept for the
 * backward).  Backward takes dh [T, B, H] (gradient w.r.t. every h_t) and writes dgx [T, B, 4H], from which the caller
 * forms dW_ih, dW_hh (against h shifted by one step), the biases and dx with GEMMs.  hidden <= 64. */
int jova_lstm_fwd(const float* gx, const float* w_hh, int seq_len, int batch, int hidden, float* h_out, float* gates,
                  float* c_out, void* stream);
int jova_lstm_bwd(const float* dh, const float* w_hh, const float* gates, const float* c_seq, int seq_len, int batch,
                  int hidden, float* dgx, void* stream);

/* ---- Weave view (jova/nn/layers.py:72-232), section 8(f3) -------------------------------------------------------- */
/* out[r][:] = src[idx[r]][:] (atom_features[atom_to_pair], layers.py:141-142; an index outside [0, m) yields NaN) and
 * out[idx[r]][:] += src[r][:] (torch_scatter.scatter_add(src, index, dim=0), the function layers.py:135 and :191 call;
 * float atomics, out is accumulated into). */
int jova_gather_rows(const float* src, const int64_t* idx, int64_t n, int d, int64_t m, float* out, void* stream);
int jova_scatter_add_rows(const float* src, const int64_t* idx, int64_t n, int d, int64_t m, float* out, void* stream);
/* WeaveGather.gaussian_histogram (layers.py:198-207): x [n] -> out [n, 11]; backward dx [n] from dout [n, 11]. */
int jova_gauss_hist_fwd(const float* x, int64_t n, float eps, float* out, void* stream);
int jova_gauss_hist_bwd(const float* x, const float* dout, int64_t n, float eps, float* dx, void* stream);

/* ---- evaluation metrics ---------------------------------------------------------------------------------------- */
/* The metrics Jova.evaluate takes per evaluation batch (proj/dti/train_joint_attn.py:335-343 through
 * jova/metrics/all_metrics.py:80-104 and jova/metrics/cindex_measure.py:15-38, whose pair count is
 * ext_src/c_swapped.c:324-365), on n samples resident on the device.  Samples with w == 0 are left out
 * (all_metrics.py:651-652); labels and predictions are first un-z-scored as v * y_std + y_mean in double
 * (NormalizationTransformer.untransform).  out6 = { rms, concordance index, pearson r^2, nanmean([ci, r2]) - rms,
 * swapped pairs (ties in the prediction count 0.5), pairs with different labels }; a quantity the reference cannot
 * form (no pair with different labels, constant input, no sample) is NaN.  ws16 = 128 bytes of device scratch.
 * Pair counts are exact integers; n <= 2^18 (the count is brute force, n^2 / 2 comparisons). */
int jova_dti_metrics(const float* y_true, const float* y_pred, const float* w_or_null, int n, double y_std,
                     double y_mean, double* ws16, double* out6, void* stream);
/* Multi-task form (output_dim = len(tasks) > 1, proj/dti/train_joint_attn.py:1027; "full ToxCast"): y_true / y_pred / w are
 * [n, n_tasks] row-major.  Every task is measured as above on its own samples with w != 0, then the tasks are combined
 * as Metric.compute_metric does with its default arithmetic_mean = True and task_averager = np.nanmean
 * (jova/metrics/all_metrics.py:147, :517-600; jova/metrics/evaluate.py:17): sum over tasks with a finite value of
 * n_t * value_t / N, with n_t the kept samples of task t and N their sum over the tasks the metric does not exclude (the
 * concordance index excludes a task whose kept labels are all equal, :273-279).  out6 as above (swapped pairs and pairs
 * summed over the tasks); per_task (optional) receives [n_tasks, 6] = rms, ci, r2, n_t, swapped, pairs per task.
 * ws = 128 * n_tasks bytes of device scratch. */
int jova_dti_metrics_tasks(const float* y_true, const float* y_pred, const float* w_or_null, int n, int n_tasks,
                           double y_std, double y_mean, double* ws, double* per_task_or_null, double* out6, void* stream);

/* ---- element-wise helpers (nn.ReLU and gradient bookkeeping) --------------------------------------------------- */
int jova_relu_fwd(const float* x, int64_t n, float* y, void* stream);
int jova_relu_bwd(const float* dy, const float* y, int64_t n, float* dx, void* stream);
int jova_axpy(const float* x, float a, int64_t n, float* y, void* stream);
/* y = x * (*a_dev): the factor is a DEVICE scalar (the gradient that reaches nn.MSELoss's backward, train_joint_attn.py:465
 * and :485, without a host read or a broadcast copy) */
int jova_scale_dev(const float* x, const float* a_dev, int64_t n, float* y, void* stream);
/* p[0..n) = 0 as a kernel on the stream (optimizer.zero_grad(), proj/dti/train_joint_attn.py:428, on the flat gradient
 * buffer; a kernel rather than a memset so that it replays reliably inside a captured hipGraph) */
int jova_zero(float* p, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JOVA_HIP_H */
