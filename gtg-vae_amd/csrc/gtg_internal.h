// Internal declarations shared by the HIP translation units of libgtgvae_hip.so.
// gfx950 (MI355X / CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gtgvae.h"  // GTG_OK / GTG_ERR_* codes
#include "side_jobs.h"

namespace gtg {

// ---- error plumbing (no C++ exceptions cross the C ABI) --------------------
int set_error(int code, const char* fmt, ...);
int check_launch(const char* what);

// ---- optional per-launch timing (off by default) -----------------------------
// When enabled through gtg_profile_enable(1), selected launch sites bracket
// their kernel with a pair of HIP events on the caller's stream;
// gtg_profile_report() synchronises them and aggregates by tag.
struct ProfScope {
    int slot = -1;
    hipStream_t stream;
    ProfScope(hipStream_t s, const char* tag, long long a, long long b, long long c);
    ~ProfScope();
};

// ---- fp32 MFMA GEMM ---------------------------------------------------------
// C[M,N] (+)= op(A)[M,K] * op(B)[K,N], fp32 in / fp32 accumulate on
// v_mfma_f32_32x32x2_f32.  Operand layouts:
//   a_kc  : A element (m,k) at A[m*lda + k]   (K contiguous);  else A[k*lda + m]
//   b_kc  : B element (k,n) at B[n*ldb + k]   (K contiguous);  else B[k*ldb + n]
enum GemmEpilogue : int {
    EPI_STORE = 0,     // C = acc (+ bias[n])
    EPI_RELU = 1,      // C = max(acc + bias[n], 0)
    EPI_SIGMOID = 2,   // C = 1/(1+exp(-(acc + bias[n])))
    EPI_MASK = 3,      // C = aux[m,n] > 0 ? acc : 0
    EPI_ACCUM = 4,     // C += acc
    EPI_POOL = 5,      // per batch: pool_val[b,n] = max_m (acc + bias[n]), pool_idx = argmax (lowest m on ties)
    // EPI_SIGMOID followed, on the values just stored, by the un-masked upper-triangular BCE against
    // fuse.bce_target (graph_ops.hip: bce_fwd_kernel, element for element the same arithmetic): the gradient
    // w.r.t. the logits goes to fuse.bce_dlogits (row pitch N) and every wave that stores leaves the sum of its
    // loss terms in fuse.bce_partial (gemm_bce_partials() floats, all of them written).  Row m is graph m, column
    // k the k-th element of the triangle, N = n(n+1)/2.  Batch 1, no split-K, no device-side extents.
    EPI_SIGMOID_BCE = 6,
    // EPI_STORE of dz followed by the backward of the reparameterisation and the KL term (graph_ops.hip:
    // latent_bwd_kernel, the same arithmetic): fuse.dmu / fuse.dlv receive the gradients w.r.t. mu and logvar at
    // row pitch fuse.ld_out.  Batch 1, no split-K.
    EPI_LATENT_BWD = 7,
};

// Operands of the fused epilogues EPI_SIGMOID_BCE and EPI_LATENT_BWD.
struct GemmFuse {
    const float* bce_target = nullptr;  // (M, n, n) padded targets
    int bce_n = 0;
    float bce_gscale = 0.f;             // 1 / (M N), times *bce_gscale_dev when given
    const float* bce_gscale_dev = nullptr;
    float* bce_dlogits = nullptr;       // (M, N)
    float* bce_partial = nullptr;
    const float* mu = nullptr;          // (M, N) each, row pitch N
    const float* lv = nullptr;
    const float* eps = nullptr;
    const float* kl_w_dev = nullptr;    // optional device scalar multiplied into kl_w
    float kl_w = 0.f, inv_b = 0.f;
    float* dmu = nullptr;
    float* dlv = nullptr;
    int ld_out = 0;
};

struct GemmDesc {
    const float* A = nullptr;
    const float* B = nullptr;
    float* C = nullptr;
    int M = 0, N = 0, K = 0;
    int lda = 0, ldb = 0, ldc = 0;
    bool a_kc = true, b_kc = true;
    int batch = 1;
    long long strideA = 0, strideB = 0, strideC = 0, strideBias = 0;  // element offsets per batch (may be negative)
    int epi = EPI_STORE;
    const float* bias = nullptr;
    const float* aux = nullptr;
    int ldaux = 0;
    long long strideAux = 0;
    float* pool_val = nullptr;   // (batch, N)
    int32_t* pool_idx = nullptr; // (batch, N)
    // ragged / device-sized extents (all optional, device pointers)
    const int* m_dev = nullptr;      // M = min(M, *m_dev): tiles past it exit at once
    const int* k_dev = nullptr;      // K = min(K, *k_dev)
    const int* batch_dim = nullptr;  // per batch: M = K = batch_dim[b]
    const int* batch_off = nullptr;  // per batch: row offset applied to B (off_b) and/or C and aux (off_c)
    bool off_b = false, off_c = false;
    // EPI_POOL: B holds M batch-independent "prototype" rows from row *proto_base on (off_b offsets aside); the
    // maximum over prototype rows batch_dim[b] .. M-1 (+ bias) is merged into batch b's pool
    const int* proto_base = nullptr;
    // split-K: when splits > 1 the kernel writes `splits` partial slabs of
    // M*N floats into `slabs` and a second kernel sums them in slab order into C
    // (bitwise reproducible; no float atomics).
    int splits = 1;
    float* slabs = nullptr;
    // tail split (128x128 tiles, batch 1, no split-K): the tiles of the last, partly
    // filled round of resident blocks are each cut along K into as many parts as
    // fill the idle block slots; the parts leave their accumulators in `tail_ws`
    // and the last one to arrive (counted in `tail_counters`) sums them in part
    // order and runs the epilogue.  tail_counters must be zero on entry and is
    // zero again on exit.  Sizes: gemm_tail_workspace().
    float* tail_ws = nullptr;
    int32_t* tail_counters = nullptr;
    // 128x128 configuration only: the epilogue also leaves the column sums of the rows each wave stored, as
    // colsum_part[(tile row * 2 + wave row)][N] (gemm_colsum_rows(M) rows; with m_dev only those of live tiles)
    float* colsum_part = nullptr;
    // Operands already split into three bf16 planes (the GTG_MATRIX_BF16X3 arithmetic without the split in the
    // GEMM's loop): A and B then point at k-blocked bf16 images [3][K/16][rows][16] (a_kc = b_kc = true), lda /
    // ldb are the images' `rows` extents and planeA / planeB plane distances in bf16 elements.  128x128 tiles, batch 1,
    // K a multiple of 16, no split-K; m_dev, the tail split, every store epilogue and colsum_part apply.
    bool planes = false;
    long long planeA = 0, planeB = 0;
    // The same with two fp16 planes per operand (images [2][K/16][rows][16]), every operand row scaled by a power of
    // two: planes_inv_a[M] / planes_inv_b[N] hold the inverse scales, which the epilogue multiplies back in.
    const float* planes_inv_a = nullptr;
    const float* planes_inv_b = nullptr;
    // GTG_MATRIX_F16X2 on the 128x128 / 256x128 configurations: GEMM_ABSMAX_BYTES of device scratch for the
    // operands' partial maxima (written by a launch of its own right before the GEMM's; calls on one stream may
    // share it).  Without it such a product runs the three-plane bf16 split instead.
    float* absmax_ws = nullptr;
    // optional: a bound the caller knows on the magnitudes of A / B (> 0), which saves that operand's pass
    float a_bound = 0.f, b_bound = 0.f;
    // absmax_ws already holds both operands' partial maxima (the producer of A left them): no pass at all
    bool absmax_ready = false;
    // absmax_ws already holds partial maxima that bound A (left by an earlier product on the same or a larger
    // array): only B is walked
    bool a_partials_ready = false;
    // Side jobs (side_jobs.h) that neither read what this product writes nor write what it reads: they ride in
    // extra blocks of its launch when the launch is one of the K-group configurations (batch-sized products) and
    // run as a launch of their own in front of it otherwise.  gemm_pair takes those of its first product.
    const SideJobs* side = nullptr;
    GemmFuse fuse;  // EPI_SIGMOID_BCE / EPI_LATENT_BWD
};
// Number of loss partials an EPI_SIGMOID_BCE product of this shape writes (one per storing wave).
long long gemm_bce_partials(int M, int N, int K);
// Whether a product of this shape (batch 1) runs a configuration that has the fused epilogues.
bool gemm_fused_epilogue_ok(int M, int N, int K);
constexpr int GEMM_ABSMAX_PARTS = 1024;
constexpr size_t GEMM_ABSMAX_BYTES = 2 * GEMM_ABSMAX_PARTS * sizeof(float);
// GEMM_ABSMAX_PARTS partial maxima of |x| over a (rows, cols) fp32 array with leading dimension ld, written to
// out[0 .. GEMM_ABSMAX_PARTS) (every entry is written: no clearing needed)
int absmax_partials(const float* x, long long rows, long long cols, long long ld, float* out, hipStream_t s);

#if defined(__HIPCC__)
// The power of two that puts a largest magnitude `mx` into [2^13, 2^14) -- where fp16 (up to 65504) has room for
// the rounding of every element and of sums the matrix pipe forms before its fp32 accumulation -- and its
// inverse; both stay normal numbers whatever mx is (zero, subnormal and huge maxima are clamped).
__device__ __forceinline__ void f16x2_scale(float mx, float& scale, float& inverse)
{
    int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);  // biased exponent
    e = e < 14 ? 14 : (e > 254 ? 254 : e);
    scale = __uint_as_float((unsigned)(267 - e) << 23);    // 2^(13 - (e - 127))
    inverse = __uint_as_float((unsigned)(e - 13) << 23);   // 2^((e - 127) - 13)
}
// The two fp16 planes of the pair (x0, x1) * scale in four instructions: w0 = the products rounded to fp16
// (low half x0, high half x1), w1 = what is left of them, rounded.  v_fma_mix* multiplies in fp32, adds an
// fp16 or fp32 operand and rounds once to fp16, into one half of the destination: the product x * scale is
// exact (a power of two), so w0 is its fp16 rounding and x * scale - w0, exact in fp32, is rounded once for
// w1 -- the same values as scaling, converting, converting back, subtracting and converting again, which
// the compiler builds from seven instructions per pair (it converts the first plane twice, once packed for
// the store and once per element as the subtrahend).
// Backward of z = eps exp(logvar / 2) + mu and of the KL term at one element: g = dL/dz, w = KL weight / B.
// Written with explicit fused multiply-adds so that the stand-alone kernel (graph_ops.hip) and the GEMM epilogue
// that computes the same thing (gemm_f32.hip, EPI_LATENT_BWD) round identically: left to the compiler, a sum of
// two products is contracted around either one.
__device__ __forceinline__ void latent_bwd_point(float g, float m, float l, float e, float w, float& gm, float& gl)
{
    gm = __fmaf_rn(w, m, g);
    const float t1 = __fmul_rn(__fmul_rn(0.5f, e), expf(__fmul_rn(0.5f, l)));
    const float t2 = __fmul_rn(__fmul_rn(w, 0.5f), __fsub_rn(expf(l), 1.f));
    gl = __fmaf_rn(g, t1, t2);
}
__device__ __forceinline__ void split_pair_f16(float x0, float x1, float scale, unsigned& w0, unsigned& w1)
{
    unsigned a, b;
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(a), "=&v"(b)
        : "v"(x0), "v"(x1), "v"(scale));
    w0 = a;
    w1 = b;
}
#endif
inline int gemm_colsum_rows(int M) { return 2 * ((M + 127) / 128); }

// Matrix pipe of the MFMA-bound (128x128-tile) GEMMs issued by the calling thread: GTG_MATRIX_F32
// (v_mfma_f32_32x32x2_f32), GTG_MATRIX_BF16 (operands rounded to bf16) or GTG_MATRIX_BF16X3 (exact
// three-way bf16 split of both operands, six products per k step, fp32 accumulate: fp32-level error).
// An entry point that takes a matrix_mode argument holds a MatrixModeScope for its duration.
extern thread_local int g_matrix_mode;
struct MatrixModeScope {
    int prev;
    explicit MatrixModeScope(int mode) : prev(g_matrix_mode) { g_matrix_mode = mode; }
    ~MatrixModeScope() { g_matrix_mode = prev; }
};
inline bool matrix_mode_valid(int m)
{
    return m == GTG_MATRIX_F32 || m == GTG_MATRIX_BF16 || m == GTG_MATRIX_BF16X3 || m == GTG_MATRIX_F16X2;
}
// the modes whose results carry fp32-level error through split operands on the 16-bit matrix pipe
inline bool matrix_mode_split(int m) { return m == GTG_MATRIX_BF16X3 || m == GTG_MATRIX_F16X2; }

// Tail-split workspace: *ws_bytes of float slabs and *counter_bytes of int32 counters.
void gemm_tail_workspace(size_t* ws_bytes, size_t* counter_bytes);

// Bytes of slab workspace gemm() needs for this shape (0 when it will not split).
size_t gemm_splitk_workspace(int M, int N, int K, int batch);
int gemm_pick_splits(int M, int N, int K, int batch);
int gemm(const GemmDesc& d, hipStream_t s);
// k-blocked planes [3][K/16][R][16] bf16 of the operand src (R = rows, K = cols) or, with transpose, of its
// transpose (R = cols, K = rows); src has row pitch ld; K must be a multiple of 16
int split_planes(const float* src, long long rows, long long cols, long long ld, bool transpose, void* planes, hipStream_t s);
// Two independent GEMMs; one launch when both run on the 32x32 K-group tile configuration (d0 transposed-A x
// plain-B, d1 plain-A x plain-B), otherwise one after the other.  Results equal gemm(d0), gemm(d1) bit for bit.
int gemm_pair(const GemmDesc& d0, const GemmDesc& d1, hipStream_t s);

// ---- element-wise / reduction kernels (graph_ops.hip) -----------------------
int adj_count(const int64_t* edge_index, int64_t E, int64_t B, int n, int32_t* counts, int32_t* err, hipStream_t s);
int stage_batch(const int64_t* ei_src, int64_t E, int64_t* ei_dst, int64_t E_cap, const float* a_src, float* a_dst,
                int64_t a_count, const float* e_src, float* e_dst, int64_t e_count, hipStream_t s);
int gather_batch(const int64_t* esrc, int64_t E_all, const int64_t* estart, const float* adj_all, int64_t G, int n,
                 const int64_t* idx, int B, int64_t* off, int64_t* ei_dst, int64_t E_cap, float* target_dst, int32_t* err,
                 hipStream_t s);
int clear_i32(int32_t* p, int64_t count, hipStream_t s);  // a kernel, not a memset: replays correctly in HIP graphs
// counts: zero on entry to adj_count, zero again after adj_normalize; extent (nullable): per graph, 1 + the
// highest node index an edge touches
int adj_normalize(int32_t* counts, int64_t B, int n, float* adj, int32_t* extent, hipStream_t s);
int pool_bwd_gather(const float* adj, const float* dpool, const int32_t* argmax, int64_t B, int n, int d,
                    float* dT2, hipStream_t s);
int colsum(const float* X, int64_t rows, int cols, int ld, float* partial, float* out, hipStream_t s);
size_t colsum_workspace(int64_t rows, int cols);
// column sums of a tall matrix and of a short one, the short one riding in the tall one's second-stage launch
int colsum_dev_and_short(const float* X, int64_t rows, const int* rows_dev, int cols, int ld, float* partial, float* out,
                         const float* X2, int64_t rows2, int cols2, int ld2, float* out2, hipStream_t s);
// out = sum of a GEMM epilogue's per-row-block column sums (GemmDesc::colsum_part), out2 = column sums of X2
int colsum_parts_and_short(const float* part, int part_rows, const int* gemm_rows_dev, int cols, float* out,
                           const float* X2, int64_t rows2, int cols2, int ld2, float* out2, float* partial,
                           hipStream_t s);
// two column sums in one launch when both matrices have at most 1024 rows (else two launches through `partial`)
int colsum_pair(const float* X0, int64_t rows0, int cols0, int ld0, float* out0, const float* X1, int64_t rows1,
                int cols1, int ld1, float* out1, float* partial, hipStream_t s);
// rows = min(rows, *rows_dev) when rows_dev is given (row count known only on the device)
int colsum_dev(const float* X, int64_t rows, const int* rows_dev, int cols, int ld, float* partial, float* out,
               hipStream_t s);
// compact node rows (identity features): meta = [n_b (B) | row offsets (B+1) | R + n], adjc = compact adjacency
// rows followed by n identity rows (the prototype rows)
// zero / zero_count: int32 words cleared by the same launches (the tail-split arrival counters)
int graph_meta(const int32_t* extent, const float* adj, int64_t B, int n, int32_t* meta, float* adjc, int32_t* zero,
               int zero_count, hipStream_t s);
// all of dT2c: real rows from the arg-max routing, prototype rows (after the R real ones) as the ordered sum over
// the graphs whose maximum they won; pg_partial: proto_grad_workspace() bytes
int pool_bwd_gather_compact(const float* adj, const float* dpool, const int32_t* argmax, const int32_t* meta,
                            int64_t B, int n, int d, float* dT2, float* pg_partial, int32_t* zero, int zero_count,
                            hipStream_t s);
size_t proto_grad_workspace(int n, int d);
// the same rows as the plane GEMM's A operand: k-blocked bf16 planes (3, d / 16, rows_alloc, 16), and the prototype
// rows also as fp32 (proto_rows, (n, d)); d a multiple of 16, at most 1024
bool pool_bwd_gather_planes_applies(int n, int d);
int pool_bwd_gather_planes(const float* adj, const float* dpool, const int32_t* argmax, const int32_t* meta, int64_t B,
                           int n, int d, long long rows_alloc, void* planes, float* proto_rows, float* pg_partial,
                           int32_t* zero, int zero_count, const float* w, int w_rows, int w_cols, int w_ld, void* w_planes,
                           float* inv_rows, float* inv_w, hipStream_t s);
// kl_out = reparam_kl_scale(B) * (the reparam_kl_partials(B, d) partial sums, added by final_sum_kernel); with
// kl_out == NULL the partials are left for the caller to sum (a SIDE_SUM job)
int reparam_kl(const float* mu, const float* logvar, const float* eps, int64_t B, int d, float* z,
               float* kl_partial, float* kl_out, hipStream_t s);
inline long long reparam_kl_partials(int64_t B, int d) { return ((long long)B * d + 255) / 256; }
inline float reparam_kl_scale(int64_t B) { return B > 0 ? -0.5f / (float)B : 0.f; }
// dmu / dlogvar rows have stride ld_out (d, or 2d when they are the two halves of one (B, 2d) matrix)
int latent_bwd_elem(const float* dz, const float* mu, const float* logvar, const float* eps, const float* kl_scale_dev,
                    float kl_scale, int64_t B, int d, int ld_out, float* dmu, float* dlogvar, hipStream_t s);
// dlogits (nullable): gradient of the mean loss w.r.t. the pre-sigmoid logits, written in the same pass
// loss = bce_scale(B, n) * (the B per-graph partial sums); with loss == NULL the partials are left for the caller
int bce_fwd(const float* out, const float* adj, int64_t B, int n, float* row_partial, float* loss, float* dlogits,
            const float* gscale_dev, hipStream_t s);
inline float bce_scale(int64_t B, int n)
{
    const int m = n * (n + 1) / 2;
    return (B > 0 && m > 0) ? 1.f / ((float)B * (float)m) : 0.f;
}
// the same with the per-graph partial sums in a file-scope device array (no workspace argument, nothing allocated)
int bce_fwd_static(const float* out, const float* adj, int64_t B, int n, float* loss, float* dlogits,
                   const float* gscale_dev, hipStream_t s);
int64_t bce_static_max_graphs();
// gdev (nullable) is a device scalar multiplied into gscale (an upstream gradient).
// through_sigmoid = 1: gradient w.r.t. the pre-sigmoid logits (BCE backward, then
// the sigmoid's backward); 0: gradient w.r.t. the probabilities only.
int bce_bwd(const float* out, const float* adj, int64_t B, int n, const float* gdev, float gscale,
            int through_sigmoid, float* grad, hipStream_t s);
// dy = dout * (1 - out) * out
int sigmoid_bwd(const float* out, const float* dout, int64_t total, float* dy, hipStream_t s);
// state: 16 device bytes {int32 step, float lr/bc1, float 1/sqrt(bc2), pad}; the
// step counter lives on the device so a captured HIP graph can be replayed.
// prepared: the step counter and bias corrections of this step were already advanced (a SIDE_ADAM_PREPARE job)
int adam_stateful(float* p, const float* g, float* m, float* v, int64_t n, void* state, float lr, float b1, float b2,
                  float eps, float gscale, bool prepared, hipStream_t s);
int edge_prf(const int32_t* edges, const int32_t* count, const uint64_t* gold_bits, int64_t G, int n,
             float* recall, float* precision, hipStream_t s);

// ---- first GCN layer on identity features as sparse row products (gcn_sparse.hip) ----
// h1c[r] = relu(b1 + sum_k adjc[r,k] W1[:,k]) for r < min(rows_max, *rows_dev); H a multiple of 4
int conv1_sparse(const float* adjc, int n, const float* W1, const float* b1, int H, const int* rows_dev, int rows_max,
                 float* h1c, float* absmax, const float* w_next, long long w_next_count, hipStream_t s);

// pooled (B, d) / argmax = max over the rows r < n_b of A_b T2c_b + b2, merged with the prototype rows' maximum;
// meta as written by graph_meta; d a multiple of 4
// dW2 = dT2c^T * H1c computed from the pooled gradient and the arg-max rows instead of the dense product
// (gcn_sparse.hip); dT2c supplies only its prototype rows.  partial: dw2_sparse_workspace() bytes.  With
// defer_reduce the partials are left for dw1_sparse to reduce (one launch fewer).
bool dw2_sparse_applies(int64_t B, int n, int d, int H);
size_t dw2_sparse_workspace(int d, int H);
// proto_rows (nullable): the prototype rows' gradient as an (n, d) block of its own instead of rows R.. of dT2c
int dw2_sparse(const float* adj, const float* h1c, int H, const float* dpool, const int32_t* argmax, const int32_t* meta,
               const float* dT2c, const float* proto_rows, int64_t B, int n, int d, float* partial, float* dW2,
               bool defer_reduce, hipStream_t s);
// dW1[c][i] = sum over the stored rows (count read on the device) of dZ1c[r][c] * adjc[r][i], walking the nonzeros
// of each adjacency row (gcn_sparse.hip).  partial: dw1_sparse_workspace() bytes.
bool dw1_sparse_applies(int64_t rows, int n, int H);
size_t dw1_sparse_workspace(int H);
// dw2_partial (nullable): partials of a dw2_sparse call made with defer_reduce, reduced into dW2 (d x H) by the
// launch that reduces dW1's.
int dw1_sparse(const float* adjc, int n, const float* dZ1c, int H, const int* rows_dev, int64_t rows_max, float* partial,
               float* dW1, const float* dw2_partial, int d, float* dW2, hipStream_t s, const SideJobs* side = nullptr);
int pool_sparse(const float* adj, const float* t2c, const int32_t* meta, const float* b2, int64_t B, int n, int d,
                float* pooled, int32_t* argmax, hipStream_t s);

// ---- vocabulary projection + cross-entropy (proj_ce.hip) ---------------------
// rowmap (2 R + 1 int32): [count | compact index of row r or -1 | original row of compact row i]
int ce_rowmap(const int64_t* targets, int64_t R, int64_t padidx, int32_t* rowmap, float* loss, int32_t* pred,
              hipStream_t s);
// mode 0: dst[i] = src[list[i]]; 1: dst[i] = g[list[i]] * src[i]; 2: dst[r] = map[r] >= 0 ? g[r] * src[map[r]] : 0
int ce_rows(int mode, const float* src, const float* g, const int32_t* rowmap, int64_t R, int K, float* dst,
            hipStream_t s);
// in place on the compact rows of P: logits -> softmax - onehot(target); loss / pred at the original row
int ce_softmax(float* P, int64_t ldp, int V, const int64_t* targets, const int32_t* rowmap, int64_t R, float* loss,
               int32_t* pred, hipStream_t s);
size_t ce_bias_grad_workspace(int V);
int ce_bias_grad(const float* P, int64_t ldp, int V, const float* g, const int32_t* rowmap, int64_t R, float* partial,
                 float* db, hipStream_t s);

// ---- one-layer LSTM over a padded sequence (lstm.hip) ------------------------
// x (T, B, I), w_ih (4H, I), w_hh (4H, H), biases (4H), h0 / c0 (B, H) nullable; hs / cs (T, B, H) and gates
// (T, B, 4H) (the gate activations) are outputs and the backward's saved state.  I % 4 == 0, H % 64 == 0.
bool lstm_applies(int T, int B, int I, int H);
size_t lstm_fwd_workspace(int T, int B, int I, int H);
size_t lstm_bwd_workspace(int T, int B, int I, int H);
int lstm_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* h0,
             const float* c0, int T, int B, int I, int H, bool reverse, float* hs, float* cs, float* gates, void* ws,
             size_t ws_bytes, hipStream_t s);
int lstm_bwd(const float* x, const float* w_ih, const float* w_hh, const float* h0, const float* c0, const float* hs,
             const float* cs, const float* gates, const float* d_hs, const float* d_hT, const float* d_cT, int T, int B,
             int I, int H, bool reverse, float* dx, float* dw_ih, float* dw_hh, float* db, float* dh0, float* dc0, void* ws,
             size_t ws_bytes, hipStream_t s);

// ---- MST (mst.hip) ----------------------------------------------------------
// text_ops.hip: word drop, embedding lookup + dropout (+ broadcast add) and its ordered scatter-add backward, dropout
int word_drop(const int64_t* x, int T, int B, int64_t padidx, float p, unsigned long long seed, unsigned long long sid,
              const uint8_t* keep, int64_t* out, hipStream_t s);
int embed_dropout_fwd(const float* table, const int64_t* ids, int64_t R, int64_t V, int D, float p, unsigned long long seed,
                      unsigned long long sid, const float* add, int B_add, float* out, int32_t* err, hipStream_t s);
size_t embed_dropout_bwd_workspace(int64_t R, int64_t V, int D);
int embed_dropout_bwd(const float* dout, const int64_t* ids, int64_t R, int64_t V, int D, float p, unsigned long long seed,
                      unsigned long long sid, float* dtable, void* ws, size_t ws_bytes, hipStream_t s);
int dropout_map(const float* x, int64_t n, float p, unsigned long long seed, unsigned long long sid, float* y, hipStream_t s);

int mst_triu(const float* rows, int64_t G, int n, int32_t* edges, int32_t* count, hipStream_t s);
int mst_full(const float* mats, int64_t G, int n, int32_t* edges, int32_t* count, hipStream_t s);

}  // namespace gtg
