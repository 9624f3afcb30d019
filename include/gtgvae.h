/* gtgvae.h — C ABI of libgtgvae_hip.so: the graph-VAE training hot path of
 * gtg-vae as hand-written HIP kernels for AMD MI355X (gfx950 / CDNA4).
 *
 * The reference (/root/reference, pure Python) has no FFI of its own; its
 * boundary is the Python class surface.  Each entry point below therefore
 * cites the reference function whose arithmetic it replaces, and
 * the Python modules under gtg-vae_amd/ mirror the reference's classes on top of this ABI.
 *
 * Conventions
 *   - every pointer is DEVICE memory owned by the caller (inputs, outputs and
 *     workspaces); the library allocates, frees and synchronises nothing;
 *   - `stream` is a hipStream_t (NULL = the default stream); all kernels of a
 *     call are launched on it in order, so calls can be captured in a HIP graph;
 *   - dense matrices are row-major fp32; indices are int32 unless stated;
 *   - return 0 on success, a negative GTG_ERR_* code otherwise, with a
 *     human-readable message in gtg_last_error() (thread-local);
 *   - n (nodes per padded graph) must be in [1, 64]; the encoder entry points take
 *     at most GTG_MAX_GRAPHS_PER_CALL graphs per call (larger batches: several calls);
 *   - *_workspace_bytes() return the exact size the matching call carves;
 *     pass a buffer at least that large (256-byte aligned).
 *   - all reductions are ordered: identical inputs give bit-identical outputs.
 */
#ifndef GTGVAE_H
#define GTGVAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GTG_VERSION 1

#define GTG_OK 0
#define GTG_ERR_INVALID (-1)
#define GTG_ERR_LAUNCH (-2)
#define GTG_ERR_UNSUPPORTED (-3)
#define GTG_ERR_WORKSPACE (-4)

/* Matrix pipe for the large (MFMA-bound) GEMMs of an entry point that takes a
 * `matrix_mode` argument.  Operands and results are fp32 in memory in every mode.
 *   GTG_MATRIX_F32     v_mfma_f32_32x32x2_f32 on the fp32 operands
 *   GTG_MATRIX_BF16X3  every operand value is split exactly into three bf16 pieces
 *                      (8 + 8 + 8 significant bits) on its way into LDS and the six
 *                      piece products of weight >= 2^-16 run on
 *                      v_mfma_f32_32x32x16_bf16 with fp32 accumulation: the error
 *                      against float64 equals the fp32 pipe's (about 1e-6 of max|C|),
 *                      at 1.5x its speed.  Inf operands give NaN (inf - inf in the split).
 *   GTG_MATRIX_F16X2   every operand is scaled by a power of two that puts its largest
 *                      magnitude below 2^14 (found by a pass over the operand right before
 *                      the product) and split into two fp16 pieces, each rounded to nearest
 *                      (11 + 11 significant bits and the sign of the remainder); the three
 *                      piece products of weight >= 2^-11 run on v_mfma_f32_32x32x16_f16 with
 *                      fp32 accumulation and the result is scaled back: the same error
 *                      against float64 as GTG_MATRIX_BF16X3 with half its matrix
 *                      instructions.  Values below 2^-28 of an operand's largest lose relative
 *                      precision (not absolute); Inf operands give NaN.  Where a producer
 *                      kernel hands the product its operands already split (the graph encoder's
 *                      backward, the LSTM recurrence) the scale is per operand row and comes
 *                      from that kernel.  The batched forms, and a GEMM call without the scratch
 *                      for the maxima, run the GTG_MATRIX_BF16X3 arithmetic in this mode.
 *   GTG_MATRIX_BF16    operands rounded to bf16 (round to nearest even), one product:
 *                      about 3e-3 of max|C|; for loss tolerances of 1e-3. */
/* The encoder entry points launch one grid row per graph: at most this many graphs per call. */
#define GTG_MAX_GRAPHS_PER_CALL 65000

#define GTG_MATRIX_F32 0
#define GTG_MATRIX_BF16 1
#define GTG_MATRIX_F16X2 2
#define GTG_MATRIX_BF16X3 3

int gtg_version(void);
const char* gtg_last_error(void);
int gtg_max_nodes(void);
/* Kernel launches issued by the library in this process so far (a host-side counter for benchmarks:
 * the difference around one eager step is the step's launch count). */
int64_t gtg_launch_count(void);

/* Optional timing of the GEMM and MST kernel launches with HIP event pairs on
 * the caller's stream (used by bench.py for the roofline figures; the only
 * entry points that create events or synchronise; not valid under graph
 * capture).  gtg_profile_report writes one line per distinct launch,
 * "<kernel> <dim0> <dim1> <dim2> <count> <total_ms>", clears the records and
 * returns the length of the full report. */
int gtg_profile_enable(int on);
int64_t gtg_profile_report(char* buf, size_t cap);

/* Streams of the caller's own.  No counterpart in the reference (which issues everything on PyTorch's
 * current stream, src/train.py:302-314); they exist for hosts whose stream objects come from a shared,
 * recycled pool (PyTorch hands out 32 pooled streams per device round-robin, its collectives' streams
 * included), where "a new stream" can be a handle some other thread is already issuing work on -- fatal
 * for a stream that is about to capture a HIP graph.  gtg_stream_create makes a non-blocking stream on
 * `device` with hipStreamCreateWithFlags, so its handle is shared with nobody; the current device of the
 * calling thread is left as it was.  gtg_stream_destroy releases it (the caller must have drained it).
 * These two are the only entry points that create or destroy a runtime object. */
int gtg_stream_create(int device, void** stream);
int gtg_stream_destroy(void* stream);

/* ---- maximum-spanning-tree read-out ------------------------------------
 * Replaces mst_graph / helper_mst / maximumSpanningTree / findMaxVertex /
 * printMaximumSpanningTree (src/maxspantree.py:118-163, 61-115, 7-31, 35-58),
 * one graph per 64-lane wavefront, bit-exact edge lists.
 *   edges (G, n-1, 2) int32: ordered (parent, vertex) pairs, unused pairs = -1
 *   count (G)        int32: number of pairs (V-1 for V present vertices, else 0)
 * _triu takes rows (G, n(n+1)/2): the decoder's upper-triangular vectors; the
 * symmetric matrix of GraphVAE.recover_adj_lower + recover_full_adj_from_lower
 * (src/graph_vae.py:61-69) is formed on chip.  _full takes (G, n, n) matrices
 * (row j / column m orientation of src/maxspantree.py:103-112 is kept). */
int gtg_mst_batched_triu(const float* rows, int64_t num_graphs, int n, int32_t* edges, int32_t* count, void* stream);
int gtg_mst_batched_full(const float* mats, int64_t num_graphs, int n, int32_t* edges, int32_t* count, void* stream);

/* Replaces GraphVAE.graph_statistics (src/graph_vae.py:75-92) after the MST:
 * gold_bits (G, n) uint64, bit b of word a set iff (a,b) is a gold edge.
 * recall/precision (G) fp32; both 1 when the gold set or the prediction is empty. */
int gtg_edge_prf(const int32_t* edges, const int32_t* count, const uint64_t* gold_bits, int64_t num_graphs, int n,
                 float* recall, float* precision, void* stream);

/* ---- normalised adjacency ----------------------------------------------
 * Replaces the gcn_norm step inside torch_geometric GCNConv as called at
 * src/graph_vae.py:107,110: self loops dropped, one unit self loop per node,
 * in-degree with multiplicity, D^-1/2 (C + I) D^-1/2.
 *   edge_index (2, E) int64, PyG batch numbering (node id = graph*n + local)
 *   counts_ws (B*n*n) int32 scratch, 16-byte aligned: ALL ZERO on entry -- zero
 *   it once when it is allocated -- and left all zero (the call that reads
 *   the counts last clears them, so a step needs no clearing launch);
 *   err_flag (1) int32, zeroed by the caller: set to 1 if an edge leaves its
 *   graph or is out of range, never cleared by the library;
 *   adj (B, n, n) fp32 output, adj[b,i,j] = weight with which node j feeds
 *   node i;  extents (B) int32 output, may be NULL: 1 + the highest node index
 *   a non-self edge of graph b touches (0: no edge), the row count the
 *   compact-row encoder stores for the graph. */
int gtg_adj_build(const int64_t* edge_index, int64_t num_edges, int64_t num_graphs, int n, int32_t* counts_ws,
                  int32_t* err_flag, float* adj, int32_t* extents, void* stream);

/* ---- encoder -------------------------------------------------------------
 * Replaces VGAE_encoder.forward (src/graph_vae.py:106-110: GCNConv, ReLU,
 * GCNConv) + global_max_pool (src/graph_vae.py:25) and their backward.
 *   x (B*n, f) node features, or NULL for stacked identities (then f == n,
 *     src/train.py:203-204) — the X*W1^T GEMM is then skipped;
 *   W1 (2d, f), b1 (2d), W2 (d, 2d), b2 (d)
 *   h1 (B*n, 2d) saved activations; pooled (B, d); argmax (B, d) row index of
 *   the pooled maximum (lowest row on ties). */
size_t gtg_encoder_fwd_workspace_bytes(int64_t num_graphs, int n, int f, int d, int has_x);
int gtg_encoder_fwd(const float* adj, const float* x, const float* W1, const float* b1, const float* W2,
                    const float* b2, int64_t num_graphs, int n, int f, int d, float* h1, float* pooled,
                    int32_t* argmax, int matrix_mode, void* ws, size_t ws_bytes, void* stream);
size_t gtg_encoder_bwd_workspace_bytes(int64_t num_graphs, int n, int f, int d, int has_x);
int gtg_encoder_bwd(const float* adj, const float* x, const float* W1, const float* W2, const float* h1,
                    const int32_t* argmax, const float* dpooled, int64_t num_graphs, int n, int f, int d, float* dW1,
                    float* db1, float* dW2, float* db2, int matrix_mode, void* ws, size_t ws_bytes, void* stream);

/* Encoder on compact node rows, identity node features only (x = I_n, f == n).
 * Rows [0, n_b) of graph b are stored back to back (n_b = 1 + the highest node
 * index a non-self edge touches, R rows in all); the rows past n_b are
 * identical in every graph, so they are stored once, as n "prototype" rows
 * right after the real ones (n extra nodes whose adjacency row is the unit
 * vector), and merged into the max-pool as a suffix maximum.  Results equal
 * gtg_encoder_fwd/bwd with x = NULL; the row-proportional GEMMs shrink from
 * B*n to R + n rows.
 *   extents (B) int32: as written by gtg_adj_build (may be the first B words of meta)
 *   meta    (2B+2)  int32 out: n_b per graph, B+1 exclusive row offsets (last = R), R + n
 *   adjc    (B*n + n, n) out: compact adjacency rows, then the n unit rows
 *   h1c     (B*n + n, 2d) out: first R + n rows valid */
size_t gtg_encoder_compact_fwd_workspace_bytes(int64_t num_graphs, int n, int d);
int gtg_encoder_compact_fwd(const float* adj, const int32_t* extents, const float* W1, const float* b1, const float* W2,
                            const float* b2, int64_t num_graphs, int n, int d, int32_t* meta, float* adjc, float* h1c,
                            float* pooled, int32_t* argmax, int matrix_mode, void* ws, size_t ws_bytes, void* stream);
size_t gtg_encoder_compact_bwd_workspace_bytes(int64_t num_graphs, int n, int d);
int gtg_encoder_compact_bwd(const float* adj, const float* adjc, const int32_t* meta, const float* W2, const float* h1c,
                            const int32_t* argmax, const float* dpooled, int64_t num_graphs, int n, int d, float* dW1,
                            float* db1, float* dW2, float* db2, int matrix_mode, void* ws, size_t ws_bytes, void* stream);

/* ---- latent head ---------------------------------------------------------
 * Replaces zg_mu / zg_var (src/gtvae.py:25-26,108-109), reparameterize
 * (src/gtvae.py:33-37, eps supplied by the caller) and loss_kl
 * (src/gtvae.py:41-42).  kl (1) = -0.5 * sum(1 + logvar - mu^2 - e^logvar) / B.
 * Backward: dz may be NULL (no reconstruction gradient); the KL term enters
 * with weight kl_w * (*kl_w_dev if non-NULL). */
size_t gtg_latent_workspace_bytes(int64_t num_graphs, int d);
int gtg_latent_fwd(const float* h, const float* Wmu, const float* bmu, const float* Wlv, const float* blv,
                   const float* eps, int64_t num_graphs, int d, float* mu, float* logvar, float* z, float* kl,
                   void* ws, size_t ws_bytes, void* stream);
int gtg_latent_bwd(const float* h, const float* Wmu, const float* Wlv, const float* mu, const float* logvar,
                   const float* eps, const float* dz, float kl_w, const float* kl_w_dev, int64_t num_graphs, int d,
                   float* dWmu, float* dbmu, float* dWlv, float* dblv, float* dh, void* ws, size_t ws_bytes,
                   void* stream);

/* The element-wise half alone: GTVAE.reparameterize + GTVAE.loss_kl
 * (src/gtvae.py:33-42) and their backward (dmu, dlogvar). */
int gtg_reparam_kl_fwd(const float* mu, const float* logvar, const float* eps, int64_t num_graphs, int d, float* z,
                       float* kl, void* ws, size_t ws_bytes, void* stream);
int gtg_reparam_kl_bwd(const float* mu, const float* logvar, const float* eps, const float* dz, float kl_w,
                       const float* kl_w_dev, int64_t num_graphs, int d, float* dmu, float* dlogvar, void* stream);

/* ---- decoder + reconstruction loss ---------------------------------------
 * gtg_decoder_fwd replaces MLP_VAE_plain.forward (src/graph_vae.py:126-131):
 *   y1 (B, hid) = relu(z D1^T + c1), out (B, m_out) = sigmoid(y1 D2^T + c2).
 * gtg_bce_loss_fwd replaces GraphVAE.loss + adj_recon_loss
 * (src/graph_vae.py:34-59, 71-73): the mean over graphs of the mean over the
 * n(n+1)/2 upper-triangular entries of BCE(out[b,k], adj_target[b,i,j]), log
 * terms clamped at -100, no mask.  adj_target (B, n, n) fp32.  dlogits
 * (B, n(n+1)/2), may be NULL: the gradient of that loss w.r.t. the decoder's
 * pre-sigmoid logits, written in the same pass (a training step then needs no
 * separate backward launch for the loss: hand it to gtg_decoder_bwd as dout
 * with out = NULL); it is multiplied by *gscale_dev when that device scalar is
 * given (NULL = 1): the weight B_local * world / B_global of this rank's shard
 * in a data-parallel step whose shards are unequal.  `loss` is never scaled.
 * gtg_bce_loss_bwd: gradient of that loss times gscale * (*gscale_dev if
 * non-NULL), w.r.t. `out` (through_sigmoid = 0) or the logits (= 1).
 * gtg_decoder_bwd: give either dout (gradient w.r.t. out; w.r.t. the logits
 * when out is NULL) or adj_target (the BCE backward is then run inside). */
int gtg_decoder_fwd(const float* z, const float* D1, const float* c1, const float* D2, const float* c2,
                    int64_t num_graphs, int d, int hid, int m_out, float* y1, float* out, void* stream);
size_t gtg_bce_workspace_bytes(int64_t num_graphs);
int gtg_bce_loss_fwd(const float* out, const float* adj_target, int64_t num_graphs, int n, float* loss, float* dlogits,
                     const float* gscale_dev, void* ws, size_t ws_bytes, void* stream);
int gtg_bce_loss_bwd(const float* out, const float* adj_target, float gscale, const float* gscale_dev,
                     int64_t num_graphs, int n, int through_sigmoid, float* grad, void* stream);

/* gtg_bce_loss_fwd without a workspace and without any allocation: same
 * arguments otherwise, same kernels, bit-identical loss and dlogits.  It has NO
 * counterpart in the reference; it exists for callers that cannot allocate
 * device memory around the call (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  The per-graph partial sums go to a
 * device array inside the library's code object (256 KB per GPU, held until the
 * process ends), so num_graphs is limited to
 * gtg_bce_loss_fwd_noalloc_max_graphs() = 65536; a larger batch is rejected
 * with GTG_ERR_UNSUPPORTED.  The first call on each device looks the array's
 * address up (no allocation; make that call outside graph capture), with
 * `stream` belonging to the current device.  All calls on one GPU share the
 * array: two calls that overlap on one GPU (different streams) race; calls
 * ordered on one stream, or on different GPUs, are safe. */
int64_t gtg_bce_loss_fwd_noalloc_max_graphs(void);
int gtg_bce_loss_fwd_noalloc(const float* out, const float* adj_target, int64_t num_graphs, int n, float* loss,
                             float* dlogits, const float* gscale_dev, void* stream);
size_t gtg_decoder_bwd_workspace_bytes(int64_t num_graphs, int d, int hid, int m_out);
int gtg_decoder_bwd(const float* z, const float* D1, const float* D2, const float* y1, const float* out,
                    const float* dout, const float* adj_target, int n, float gscale, const float* gscale_dev,
                    int64_t num_graphs, int d, int hid, int m_out, float* dD1, float* dc1, float* dD2, float* dc2,
                    float* dz, void* ws, size_t ws_bytes, void* stream);

/* The latent head, the decoder and the reconstruction loss of one training
 * step, forward and backward, as ONE call: what GTVAE.forward does after the
 * encoder (src/gtvae.py:108-116: zg_mu / zg_var, reparameterize, loss_kl,
 * decoder, GraphVAE.loss) followed by the backward of all of it
 * (loss.backward(), src/train.py:313).  Outputs are bit-identical to
 * gtg_latent_fwd, gtg_decoder_fwd, gtg_bce_loss_fwd(dlogits), gtg_decoder_bwd
 * (dout = dlogits, out = NULL) and gtg_latent_bwd(kl_w = 1, kl_w_dev) run in
 * that order, with one exception: at batch sizes whose decoder products run
 * the small-tile configurations (up to about 1500 graphs at n = 64) the
 * scalar `loss` is the sum of the same terms in another order and may differ
 * in its last bits.  The call exists because a call of its own must finish
 * inside its own launches: here the final sums of `loss` and `kl` ride in
 * spare blocks of a later launch, the loss terms and dlogits are computed in
 * the epilogue of the decoder's last product (the probabilities are never read
 * back) and the gradients w.r.t. mu / logvar in the epilogue of the product
 * that yields dz (seven launches where the five calls and the counter kernel of
 * gtg_adam_step take twelve).
 *   h (B, d) pooled encoder output; eps (B, d); adj_target (B, n, n);
 *   kl_w_dev: device scalar, weight of the KL term in the backward (NULL = 1);
 *   gscale_dev: device scalar multiplied into dlogits (NULL = 1), as in
 *   gtg_bce_loss_fwd; loss, kl: one float each, not scaled by either.
 *   adam_state16 (nullable): the 16-byte state of gtg_adam_step; when given,
 *   this call advances its step counter and bias corrections (with lr, beta1,
 *   beta2) and the step's update must then be gtg_adam_update, not
 *   gtg_adam_step.
 *   dz (B, d) and dh (B, d): gradients w.r.t. z and h.  m_out = n(n+1)/2. */
size_t gtg_head_workspace_bytes(int64_t num_graphs, int n, int d, int hid);
int gtg_head_fwd_bwd(const float* h, const float* Wmu, const float* bmu, const float* Wlv, const float* blv,
                     const float* D1, const float* c1, const float* D2, const float* c2, const float* eps,
                     const float* adj_target, int64_t num_graphs, int n, int d, int hid, const float* kl_w_dev,
                     const float* gscale_dev, float* mu, float* logvar, float* z, float* y1, float* out, float* dlogits,
                     float* loss, float* kl, float* dWmu, float* dbmu, float* dWlv, float* dblv, float* dD1, float* dc1,
                     float* dD2, float* dc2, float* dz, float* dh, void* adam_state16, float lr, float beta1, float beta2,
                     void* ws, size_t ws_bytes, void* stream);

/* Brings one batch into the static buffers of a graph-captured step in a single
 * launch (the loop of train.py:297-312 hands the model a fresh batch every
 * iteration): the (2, num_edges) int64 edge list into a (2, edge_capacity)
 * buffer, the tail filled with (0,0) self loops that gtg_adj_build drops, so
 * launch shapes do not depend on the edge count; the padded targets
 * (target_count floats) and eps (eps_count floats). */
int gtg_stage_batch(const int64_t* edge_index, int64_t num_edges, int64_t* edge_buf, int64_t edge_capacity,
                    const float* adj_target, float* target_buf, int64_t target_count, const float* eps,
                    float* eps_buf, int64_t eps_count, void* stream);

/* Batch assembly from a dataset resident in device memory (what the collate of
 * torch_geometric's DataLoader, train.py:549, does on the host for every
 * batch): the dataset holds every graph's edge list with local node ids,
 * concatenated as (2, total_edges) with num_dataset_graphs + 1 start offsets,
 * and the padded targets (num_dataset_graphs, n, n).  For the num_graphs ids in
 * graph_ids this writes the (2, edge_capacity) edge buffer (node id = slot * n +
 * local id, tail filled with (0,0) self loops) and the targets of the batch.
 * offsets: num_graphs + 1 int64 of scratch.  err (one int32, written by every
 * call): 0, or 2 when the batch's edges exceed edge_capacity (the excess is
 * dropped), or-ed with 4 when an id is outside the dataset (that slot becomes
 * an empty graph). */
int gtg_gather_batch(const int64_t* edges_local, int64_t total_edges, const int64_t* edge_start, const float* adj_all,
                     int64_t num_dataset_graphs, int n, const int64_t* graph_ids, int64_t num_graphs, int64_t* offsets,
                     int64_t* edge_buf, int64_t edge_capacity, float* target_buf, int32_t* err, void* stream);

/* ---- vocabulary projection + token cross-entropy (text decoder) -----------
 * Replaces, for training, the tail of DAE.decode (src/text_vae.py:65,
 * logits = self.proj(output)), DAE.loss_rec (src/text_vae.py:77-79,
 * F.cross_entropy with ignore_index = pad, reduction none) and the arg-max of
 * DAE.accuracy (src/text_vae.py:83-84) and their backward.
 *   H (rows, k) decoder outputs (rows = T*B), W (vocab, k), bias (vocab) or
 *   NULL, targets (rows) int64.
 *   loss (rows): per-token cross-entropy, 0 where targets == padidx;
 *   pred (rows) int32: arg-max of the logits (first maximum), -1 at pad rows.
 * Rows whose target is the pad symbol take no part in any GEMM: the kept rows
 * are packed to the front and their count stays on the device.  State handed
 * from forward to backward (caller-allocated, opaque):
 *   rowmap (2*rows + 1) int32, Hc (rows, k) the kept rows of H,
 *   P (rows, ldp), ldp % 4 == 0, ldp >= vocab, 16-byte aligned: on return
 *   softmax(logits) - onehot(target) for the kept rows.
 * Backward takes gloss (rows), the gradient of each token's loss, and writes
 * dH (rows, k) (may be NULL), dW (vocab, k), db (vocab) (may be NULL). */
size_t gtg_proj_ce_fwd_workspace_bytes(int64_t rows, int k, int vocab);
int gtg_proj_ce_fwd(const float* H, const float* W, const float* bias, const int64_t* targets, int64_t rows, int k,
                    int vocab, int64_t padidx, float* loss, int32_t* pred, int32_t* rowmap, float* Hc, float* P,
                    int64_t ldp, int matrix_mode, void* ws, size_t ws_bytes, void* stream);
size_t gtg_proj_ce_bwd_workspace_bytes(int64_t rows, int k, int vocab);
int gtg_proj_ce_bwd(const float* W, const float* gloss, const int32_t* rowmap, const float* Hc, const float* P,
                    int64_t ldp, int64_t rows, int k, int vocab, float* dH, float* dW, float* db, int matrix_mode,
                    void* ws, size_t ws_bytes, void* stream);

/* ---- one-layer LSTM over a padded sequence (text encoder / decoder) --------
 * Replaces nn.LSTM(512, 1024, 1) as DAE runs it (src/text_vae.py:37-40: self.E
 * bidirectional = one call per direction, self.G; called at :57 and :64) and
 * its backward: gate order i, f, g, o; c' = f c + i g; h' = o tanh(c').
 *   x (T, B, I), w_ih (4H, I), w_hh (4H, H), b_ih, b_hh (4H), h0, c0 (B, H) or
 *   NULL (zero state); reverse != 0 runs time T-1 .. 0 (the "_reverse" weights).
 *   hs (T, B, H): every step's hidden state in time order (the module's
 *   `output`; its h_n is hs[T-1], or hs[0] when reverse), cs (T, B, H): the cell
 *   states (c_n likewise), gates (T, B, 4H): the gate activations.  hs, cs and
 *   gates are the state the backward needs.
 * Backward takes d_hs (T, B, H), d_hT, d_cT (B, H) -- gradients of the output
 * sequence, of h_n and of c_n, each may be NULL -- and writes dw_ih (4H, I),
 * dw_hh (4H, H), db (4H) (the gradient of b_ih and of b_hh alike), and, where
 * not NULL, dx (T, B, I), dh0, dc0 (B, H).
 * I must be a multiple of 4 and H of 64 (GTG_ERR_UNSUPPORTED otherwise).
 * matrix_mode selects the pipe of the four large products outside the
 * recurrence (input projection, dW_ih, dW_hh, dx).  The per-step recurrent
 * products h_{t-1} W_hh^T and dG_t W_hh run on split 16-bit planes at fp32-level
 * error in EVERY mode: with GTG_MATRIX_F16X2 and H <= 1024 on two scaled fp16
 * planes per operand and three products (h at the fixed scale 2^13, a given
 * initial state and the gate gradients scaled per row by the kernel that writes
 * them, W_hh by one power of two found by a pass over it); otherwise --
 * GTG_MATRIX_F32, _BF16X3, _BF16, and F16X2 with H > 1024 -- on three bf16 planes
 * per operand and six products.  There is no fp32-MFMA recurrence.
 * The step kernel holds 68 KB of static LDS (a 64 x 64 fp32 tile per wave for the
 * K-split reduction): more than the 64 KB of earlier CDNA parts, so this file
 * builds for gfx950 (160 KB per CU) only. */
size_t gtg_lstm_fwd_workspace_bytes(int T, int B, int I, int H);
int gtg_lstm_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                 const float* h0, const float* c0, int T, int B, int I, int H, int reverse, float* hs, float* cs,
                 float* gates, int matrix_mode, void* ws, size_t ws_bytes, void* stream);
size_t gtg_lstm_bwd_workspace_bytes(int T, int B, int I, int H);
int gtg_lstm_bwd(const float* x, const float* w_ih, const float* w_hh, const float* h0, const float* c0, const float* hs,
                 const float* cs, const float* gates, const float* d_hs, const float* d_hT, const float* d_cT, int T,
                 int B, int I, int H, int reverse, float* dx, float* dw_ih, float* dw_hh, float* db, float* dh0,
                 float* dc0, int matrix_mode, void* ws, size_t ws_bytes, void* stream);

/* ---- text branch: word drop, embedding lookup + dropout, dropout -------------
 * Randomness in these calls is counter-based (Philox4x32-10): whether an element is kept is a pure function of
 * (seed, stream_id, element index), so a backward call given the forward call's (seed, stream_id) regenerates
 * the forward's mask and no mask is stored.  Use a fresh stream_id per forward call.  The sequences are not
 * PyTorch's or numpy's; the distribution (keep with probability 1 - p, kept values scaled by 1 / (1 - p)) is.
 *
 * gtg_word_drop replaces noise.word_drop (src/noise.py:12-22, reached from src/text_vae.py:70 with p = 0.3):
 * tokens, out (T, B) int64, time-major; per sentence (column) every token but the first is dropped with
 * probability p, the kept ones move up in order, the tail is filled with padidx.  `keep` (T, B) uint8,
 * nullable: an explicit keep mask instead of the random one (row 0 is kept either way).  out must not alias
 * tokens.
 *
 * gtg_embed_dropout_fwd replaces self.drop(self.embed(input)) (src/text_vae.py:56) and, with `add`,
 * self.drop(self.embed(input)) + self.z2emb(z) (src/text_vae.py:62): out[r, :] = drop(table[ids[r], :])
 * + add[r mod add_rows, :].  table (vocab, dim), ids (rows) int64, add (add_rows, dim) nullable, out (rows, dim);
 * dim a multiple of 4.  An id outside [0, vocab) sets *err (nullable int32, never cleared by the library) and
 * reads row 0.
 * gtg_embed_dropout_bwd replaces the backward of both (nn.Embedding's scatter-add, src/text_vae.py:35):
 * dtable[id, :] += sum over the rows r with ids[r] == id of drop'(dout[r, :]), the rows of an id added in a fixed
 * order (a sum per group of 128 consecutive rows, then the group sums in ascending order; a list of eight or more
 * addends is cut into four contiguous parts summed side by side and joined in part order): the result depends on
 * the inputs alone, bit for bit -- no float atomics.
 * dtable is accumulated into, not overwritten; rows whose id lies outside the table are skipped; at most
 * 262 144 rows per call.  d_add (add_rows, dim), nullable: the gradient of `add`, the sum of dout over the rows
 * that share an add row (no dropout on that path; overwritten, not accumulated).  dtable may be NULL when only
 * d_add is wanted.
 *
 * gtg_dropout replaces self.drop(output) (src/text_vae.py:64) and its backward, which is the same map applied to
 * the gradient: y[i] = keep(i) ? x[i] / (1 - p) : 0; count a multiple of 4; y may alias x. */
int gtg_word_drop(const int64_t* tokens, int T, int B, int64_t padidx, float p, uint64_t seed, uint64_t stream_id,
                  const uint8_t* keep, int64_t* out, void* stream);
int gtg_embed_dropout_fwd(const float* table, const int64_t* ids, int64_t rows, int64_t vocab, int dim, float p,
                          uint64_t seed, uint64_t stream_id, const float* add, int add_rows, float* out, int32_t* err,
                          void* stream);
size_t gtg_embed_dropout_bwd_workspace_bytes(int64_t rows, int64_t vocab, int dim, int add_rows);
int gtg_embed_dropout_bwd(const float* dout, const int64_t* ids, int64_t rows, int64_t vocab, int dim, float p,
                          uint64_t seed, uint64_t stream_id, float* dtable, float* d_add, int add_rows, void* ws,
                          size_t ws_bytes, void* stream);
int gtg_dropout(const float* x, int64_t count, float p, uint64_t seed, uint64_t stream_id, float* y, void* stream);

/* ---- optimiser -------------------------------------------------------------
 * Replaces optim.Adam(lr, betas=(0.5, 0.999)).step() (src/train.py:290,314)
 * over one flat fp32 buffer.  state16: 16 device bytes, zero-initialised by the
 * caller once; holds the step counter so that a captured HIP graph replays
 * correctly.  grad_scale multiplies the gradient first (1/world_size after a
 * sum all-reduce). */
int gtg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel, void* state16,
                  float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);

/* The parameter update of gtg_adam_step alone, for a step whose counter and
 * bias corrections were already advanced by gtg_head_fwd_bwd (adam_state16
 * given there): same arithmetic, one launch. */
int gtg_adam_update(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                    const void* state16, float beta1, float beta2, float eps, float grad_scale, void* stream);

/* ---- the fp32 MFMA GEMM underneath (exposed for tests and profiling) -------
 * C[M,N] = epilogue(op(A) op(B)); a_kc: A(m,k) at A[m*lda+k] else A[k*lda+m];
 * b_kc: B(k,n) at B[n*ldb+k] else B[k*ldb+n].  epilogue: 0 store(+bias),
 * 1 relu(+bias), 2 sigmoid(+bias), 3 mask by aux>0, 4 accumulate into C.
 * slabs (optional, gtg_gemm_splitk_workspace_bytes() device bytes) enables ordered
 * split-K for short-and-wide shapes and holds the operand maxima GTG_MATRIX_F16X2
 * needs; without it that mode runs the GTG_MATRIX_BF16X3 arithmetic. */
int gtg_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int a_kc,
                 int b_kc, const float* bias, int epilogue, const float* aux, int ldaux, void* slabs,
                 size_t slab_bytes, int matrix_mode, void* stream);
size_t gtg_gemm_splitk_workspace_bytes(int M, int N, int K);

/* The same GEMM with the row count read on the device: only the first
 * min(M, *rows_dev) rows of C are computed and written (rows_dev may be NULL).
 * When the tiles do not fill a whole number of rounds of resident blocks, the
 * last round's tiles are cut along K and summed in a fixed order (bitwise
 * reproducible).  This is what the compact-row encoder GEMMs run; exported for
 * tests and profiling.  ws: gtg_gemm_rows_workspace_bytes() device bytes. */
size_t gtg_gemm_rows_workspace_bytes(void);
int gtg_gemm_f32_rows(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int a_kc,
                      int b_kc, const float* bias, int epilogue, const float* aux, int ldaux, const int32_t* rows_dev,
                      int matrix_mode, void* ws, size_t ws_bytes, void* stream);

/* The three-plane bf16 split GTG_MATRIX_BF16X3 applies to an operand, as a kernel of its own, and the GEMM on
 * operands that arrive split.  An operand with R rows and depth K (a multiple of 16) is stored k-blocked:
 * planes (3, K/16, R, 16) bf16, element [p][k/16][r][k%16] = plane p of the value (r, k); the three planes sum
 * to the fp32 value exactly.  gtg_split_planes splits src (R = rows, K = cols) or, with transpose != 0, its
 * transpose (R = cols, K = rows).  gtg_gemm_planes_rows: C[:min(M, *rows_dev)] = epilogue(A B^T), A_planes of an
 * (M, K) operand, B_planes of an (N, K) operand: the six plane products of the BF16X3 mode with no split
 * arithmetic in the GEMM's loop (the planes reach LDS by global_load_lds, whole cache lines at a time).  What the compact-row encoder's backward runs for
 * dZ1 = (dT2 W2) masked by relu; exported for tests and profiling.  ws: gtg_gemm_rows_workspace_bytes(). */
int gtg_split_planes(const float* src, int64_t rows, int64_t cols, int64_t ld, int transpose, void* planes, void* stream);
int gtg_gemm_planes_rows(const void* A_planes, const void* B_planes, float* C, int M, int N, int K, int ldc,
                         const float* bias, int epilogue, const float* aux, int ldaux, const int32_t* rows_dev, void* ws,
                         size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GTGVAE_H */
