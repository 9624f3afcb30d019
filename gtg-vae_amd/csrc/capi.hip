// extern "C" boundary of libgtgvae_hip.so (declared in include/gtgvae.h).
//
// Rules of the boundary: plain pointers and sizes only; every buffer
// (inputs, outputs, workspaces) is owned and allocated by the caller as device
// memory; nothing here allocates, frees or synchronises; every kernel is
// launched on the caller's stream, so a whole training step can be captured
// into a HIP graph.  Return value 0 = OK, negative = error, message via
// gtg_last_error().
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gtgvae.h"
#include "gtg_internal.h"

namespace gtg {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static std::atomic<long long> g_launches{0};

int check_launch(const char* what)
{
    g_launches.fetch_add(1, std::memory_order_relaxed);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GTG_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return GTG_OK;
}

// ---- per-launch timing ------------------------------------------------------
struct ProfRec {
    hipEvent_t a, b;
    std::string tag;
};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static bool g_prof_on = false;

ProfScope::ProfScope(hipStream_t s, const char* tag, long long a, long long b, long long c) : stream(s)
{
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    if (hipEventCreate(&r.a) != hipSuccess) return;
    if (hipEventCreate(&r.b) != hipSuccess) {
        (void)hipEventDestroy(r.a);
        return;
    }
    char buf[160];
    snprintf(buf, sizeof(buf), "%s %lld %lld %lld", tag, a, b, c);
    r.tag = buf;
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
    slot = (int)g_prof.size() - 1;
}

ProfScope::~ProfScope()
{
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (slot < (int)g_prof.size()) (void)hipEventRecord(g_prof[slot].b, stream);
}

namespace {

// Bump carver over a caller-provided workspace.  With base == nullptr it only
// measures, so the *_workspace_bytes entry points and the real calls share one
// layout.
struct Carver {
    char* base;
    size_t cap;
    size_t off = 0;
    bool overflow = false;
    Carver(void* b, size_t c) : base(reinterpret_cast<char*>(b)), cap(c) {}
    template <typename T>
    T* take(size_t count)
    {
        const size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
        const size_t at = off;
        off += bytes;
        if (base == nullptr) return nullptr;
        if (off > cap) {
            overflow = true;
            return nullptr;
        }
        return reinterpret_cast<T*>(base + at);
    }
};

inline size_t max2(size_t a, size_t b) { return a > b ? a : b; }

}  // namespace
}  // namespace gtg

using namespace gtg;

#define GTG_REQUIRE(cond, msg) \
    do {                       \
        if (!(cond)) return set_error(GTG_ERR_INVALID, "%s: %s", __func__, msg); \
    } while (0)
#define GTG_TRY(expr)           \
    do {                        \
        const int _rc = (expr); \
        if (_rc != GTG_OK) return _rc; \
    } while (0)

extern "C" {

int gtg_version(void) { return GTG_VERSION; }

const char* gtg_last_error(void) { return g_err; }

int gtg_max_nodes(void) { return 64; }

int64_t gtg_launch_count(void) { return (int64_t)g_launches.load(std::memory_order_relaxed); }

// Timing of the GEMM and MST launches with HIP events on the caller's stream.
// Not for use under graph capture.  gtg_profile_report writes one line per
// distinct launch, "<tag> <dim0> <dim1> <dim2> <count> <total_ms>", returns the
// number of bytes written (or needed), and clears the records.
int gtg_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    g_prof_on = on != 0;
    return GTG_OK;
}

int64_t gtg_profile_report(char* buf, size_t cap)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::map<std::string, std::pair<long long, double>> agg;
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto& e = agg[r.tag];
            e.first += 1;
            e.second += ms;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    std::string out;
    for (auto& kv : agg) {
        char line[256];
        snprintf(line, sizeof(line), "%s %lld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        out += line;
    }
    if (buf && cap > 0) {
        const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int64_t)out.size();
}

// A stream whose handle nobody else holds (see include/gtgvae.h): for graph capture and for copies issued
// from a loader thread.
int gtg_stream_create(int device, void** stream)
{
    GTG_REQUIRE(stream != nullptr, "null pointer");
    *stream = nullptr;
    int count = 0, prev = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
        return set_error(GTG_ERR_INVALID, "gtg_stream_create: device %d of %d", device, count);
    if (hipGetDevice(&prev) != hipSuccess) return set_error(GTG_ERR_LAUNCH, "gtg_stream_create: hipGetDevice failed");
    hipError_t e = hipSetDevice(device);
    hipStream_t s = nullptr;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return set_error(GTG_ERR_LAUNCH, "gtg_stream_create: %s", hipGetErrorString(e));
    *stream = (void*)s;
    return GTG_OK;
}

int gtg_stream_destroy(void* stream)
{
    if (stream == nullptr) return GTG_OK;
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) return set_error(GTG_ERR_LAUNCH, "gtg_stream_destroy: %s", hipGetErrorString(e));
    return GTG_OK;
}

// ---------------------------------------------------------------------------
// MST read-out and edge statistics
// ---------------------------------------------------------------------------
int gtg_mst_batched_triu(const float* rows, int64_t num_graphs, int n, int32_t* edges, int32_t* count, void* stream)
{
    return mst_triu(rows, num_graphs, n, edges, count, (hipStream_t)stream);
}

int gtg_mst_batched_full(const float* mats, int64_t num_graphs, int n, int32_t* edges, int32_t* count, void* stream)
{
    return mst_full(mats, num_graphs, n, edges, count, (hipStream_t)stream);
}

int gtg_edge_prf(const int32_t* edges, const int32_t* count, const uint64_t* gold_bits, int64_t num_graphs, int n,
                 float* recall, float* precision, void* stream)
{
    GTG_REQUIRE(n >= 0 && n <= 64 && num_graphs >= 0, "bad size");
    GTG_REQUIRE(num_graphs == 0 || (edges && count && gold_bits && recall && precision), "null pointer");
    return edge_prf(edges, count, gold_bits, num_graphs, n, recall, precision, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// normalised adjacency from the ragged edge list
// ---------------------------------------------------------------------------
int gtg_adj_build(const int64_t* edge_index, int64_t num_edges, int64_t num_graphs, int n, int32_t* counts_ws,
                  int32_t* err_flag, float* adj, int32_t* extents, void* stream)
{
    GTG_REQUIRE(num_edges >= 0 && num_graphs >= 0, "negative size");
    GTG_REQUIRE(n >= 1 && n <= 64, "n must be in [1, 64]");
    if (num_graphs == 0) return GTG_OK;
    GTG_REQUIRE(counts_ws && err_flag && adj && (num_edges == 0 || edge_index), "null pointer");
    hipStream_t s = (hipStream_t)stream;
    GTG_TRY(adj_count(edge_index, num_edges, num_graphs, n, counts_ws, err_flag, s));
    return adj_normalize(counts_ws, num_graphs, n, adj, extents, s);
}

// ---------------------------------------------------------------------------
// encoder: two GCN layers + max-pool over all n rows of each graph
// ---------------------------------------------------------------------------
static void enc_fwd_layout(Carver& c, int64_t B, int n, int d, bool has_x, float** xw1, float** t2)
{
    const size_t rows = (size_t)B * n;
    *xw1 = has_x ? c.take<float>(rows * 2 * d) : nullptr;
    *t2 = c.take<float>(rows * d);
}

size_t gtg_encoder_fwd_workspace_bytes(int64_t num_graphs, int n, int f, int d, int has_x)
{
    (void)f;
    Carver c(nullptr, 0);
    float *a, *b;
    enc_fwd_layout(c, num_graphs, n, d, has_x != 0, &a, &b);
    return c.off;
}

int gtg_encoder_fwd(const float* adj, const float* x, const float* W1, const float* b1, const float* W2,
                    const float* b2, int64_t num_graphs, int n, int f, int d, float* h1, float* pooled,
                    int32_t* argmax, int matrix_mode, void* ws, size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(matrix_mode_valid(matrix_mode), "matrix_mode must be GTG_MATRIX_F32, _BF16, _F16X2 or _BF16X3");
    MatrixModeScope matrix_scope(matrix_mode);
    GTG_REQUIRE(num_graphs >= 0 && f >= 1 && d >= 1, "bad size");
    GTG_REQUIRE(n >= 1 && n <= 64, "n must be in [1, 64]");
    GTG_REQUIRE(x != nullptr || f == n, "identity node features need input_dim == n");
    if (num_graphs == 0) return GTG_OK;
    GTG_REQUIRE(adj && W1 && b1 && W2 && b2 && h1 && pooled && argmax, "null pointer");
    GTG_REQUIRE((int64_t)num_graphs * n < (1ll << 31) / 4 && num_graphs <= GTG_MAX_GRAPHS_PER_CALL,
                "batch too large (at most 65000 graphs per call)");
    hipStream_t s = (hipStream_t)stream;
    const int B = (int)num_graphs, H = 2 * d, rows = B * n;
    Carver c(ws, ws_bytes);
    float *xw1, *t2;
    enc_fwd_layout(c, B, n, d, x != nullptr, &xw1, &t2);
    if (c.overflow || (c.off > 0 && ws == nullptr)) return set_error(GTG_ERR_WORKSPACE, "gtg_encoder_fwd: workspace too small");

    GemmDesc g;
    if (x) {
        // XW1 = X * W1^T
        g = GemmDesc();
        g.A = x; g.lda = f; g.a_kc = true;
        g.B = W1; g.ldb = f; g.b_kc = true;
        g.C = xw1; g.ldc = H;
        g.M = rows; g.N = H; g.K = f;
        GTG_TRY(gemm(g, s));
        // H1[b] = relu(A_b * XW1[b] + b1)
        g = GemmDesc();
        g.A = adj; g.lda = n; g.a_kc = true; g.strideA = (long long)n * n;
        g.B = xw1; g.ldb = H; g.b_kc = false; g.strideB = (long long)n * H;
        g.C = h1; g.ldc = H; g.strideC = (long long)n * H;
        g.M = n; g.N = H; g.K = n; g.batch = B;
        g.bias = b1; g.epi = EPI_RELU;
        GTG_TRY(gemm(g, s));
    } else {
        // X = I_n per graph: XW1[b] = W1^T, so H1[b] = relu(A_b * W1^T + b1)
        g = GemmDesc();
        g.A = adj; g.lda = n; g.a_kc = true; g.strideA = (long long)n * n;
        g.B = W1; g.ldb = f; g.b_kc = true; g.strideB = 0;
        g.C = h1; g.ldc = H; g.strideC = (long long)n * H;
        g.M = n; g.N = H; g.K = n; g.batch = B;
        g.bias = b1; g.epi = EPI_RELU;
        GTG_TRY(gemm(g, s));
    }
    // T2 = H1 * W2^T
    g = GemmDesc();
    g.A = h1; g.lda = H; g.a_kc = true;
    g.B = W2; g.ldb = H; g.b_kc = true;
    g.C = t2; g.ldc = d;
    g.M = rows; g.N = d; g.K = H;
    GTG_TRY(gemm(g, s));
    // pooled[b] = max_rows(A_b * T2[b] + b2)
    g = GemmDesc();
    g.A = adj; g.lda = n; g.a_kc = true; g.strideA = (long long)n * n;
    g.B = t2; g.ldb = d; g.b_kc = false; g.strideB = (long long)n * d;
    g.M = n; g.N = d; g.K = n; g.batch = B;
    g.bias = b2; g.epi = EPI_POOL; g.pool_val = pooled; g.pool_idx = argmax;
    return gemm(g, s);
}

struct EncBwdWs {
    float *dT2, *dZ1, *dXW1, *cs, *slabs;
};

static void enc_bwd_layout(Carver& c, int64_t B, int n, int f, int d, bool has_x, EncBwdWs* w)
{
    const size_t rows = (size_t)B * n;
    const int H = 2 * d;
    w->dT2 = c.take<float>(rows * d);
    w->dZ1 = c.take<float>(rows * H);
    w->dXW1 = has_x ? c.take<float>(rows * H) : nullptr;
    const size_t cs = max2(colsum_workspace(B, d), colsum_workspace((int64_t)rows, H));
    w->cs = reinterpret_cast<float*>(c.take<char>(cs));
    const size_t sl = max2(gemm_splitk_workspace(d, H, (int)rows, 1), gemm_splitk_workspace(H, f, (int)rows, 1));
    w->slabs = reinterpret_cast<float*>(c.take<char>(sl));
}

size_t gtg_encoder_bwd_workspace_bytes(int64_t num_graphs, int n, int f, int d, int has_x)
{
    Carver c(nullptr, 0);
    EncBwdWs w;
    enc_bwd_layout(c, num_graphs, n, f, d, has_x != 0, &w);
    return c.off;
}

int gtg_encoder_bwd(const float* adj, const float* x, const float* W1, const float* W2, const float* h1,
                    const int32_t* argmax, const float* dpooled, int64_t num_graphs, int n, int f, int d, float* dW1,
                    float* db1, float* dW2, float* db2, int matrix_mode, void* ws, size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(matrix_mode_valid(matrix_mode), "matrix_mode must be GTG_MATRIX_F32, _BF16, _F16X2 or _BF16X3");
    MatrixModeScope matrix_scope(matrix_mode);
    (void)W1;
    GTG_REQUIRE(num_graphs >= 1 && f >= 1 && d >= 1, "bad size");
    GTG_REQUIRE(n >= 1 && n <= 64, "n must be in [1, 64]");
    GTG_REQUIRE(x != nullptr || f == n, "identity node features need input_dim == n");
    GTG_REQUIRE(adj && W2 && h1 && argmax && dpooled && dW1 && db1 && dW2 && db2, "null pointer");
    GTG_REQUIRE((int64_t)num_graphs * n < (1ll << 31) / 4 && num_graphs <= GTG_MAX_GRAPHS_PER_CALL,
                "batch too large (at most 65000 graphs per call)");
    hipStream_t s = (hipStream_t)stream;
    const int B = (int)num_graphs, H = 2 * d, rows = B * n;
    Carver c(ws, ws_bytes);
    EncBwdWs w;
    enc_bwd_layout(c, B, n, f, d, x != nullptr, &w);
    if (c.overflow || ws == nullptr) return set_error(GTG_ERR_WORKSPACE, "gtg_encoder_bwd: workspace too small");

    // max-pool routing through the second propagate, and db2
    GTG_TRY(pool_bwd_gather(adj, dpooled, argmax, B, n, d, w.dT2, s));
    GTG_TRY(colsum(dpooled, B, d, d, w.cs, db2, s));
    GemmDesc g;
    // dW2 = dT2^T * H1
    g = GemmDesc();
    g.A = w.dT2; g.lda = d; g.a_kc = false;
    g.B = h1; g.ldb = H; g.b_kc = false;
    g.C = dW2; g.ldc = H;
    g.M = d; g.N = H; g.K = rows;
    g.splits = gemm_pick_splits(d, H, rows, 1); g.slabs = w.slabs;
    GTG_TRY(gemm(g, s));
    // dZ1 = (dT2 * W2) masked by relu
    g = GemmDesc();
    g.A = w.dT2; g.lda = d; g.a_kc = true;
    g.B = W2; g.ldb = H; g.b_kc = false;
    g.C = w.dZ1; g.ldc = H;
    g.M = rows; g.N = H; g.K = d;
    g.epi = EPI_MASK; g.aux = h1; g.ldaux = H;
    GTG_TRY(gemm(g, s));
    GTG_TRY(colsum(w.dZ1, rows, H, H, w.cs, db1, s));
    if (x) {
        // dXW1[b] = A_b^T * dZ1[b]
        g = GemmDesc();
        g.A = adj; g.lda = n; g.a_kc = false; g.strideA = (long long)n * n;
        g.B = w.dZ1; g.ldb = H; g.b_kc = false; g.strideB = (long long)n * H;
        g.C = w.dXW1; g.ldc = H; g.strideC = (long long)n * H;
        g.M = n; g.N = H; g.K = n; g.batch = B;
        GTG_TRY(gemm(g, s));
        // dW1 = dXW1^T * X
        g = GemmDesc();
        g.A = w.dXW1; g.lda = H; g.a_kc = false;
        g.B = x; g.ldb = f; g.b_kc = false;
        g.C = dW1; g.ldc = f;
        g.M = H; g.N = f; g.K = rows;
        g.splits = gemm_pick_splits(H, f, rows, 1); g.slabs = w.slabs;
        return gemm(g, s);
    }
    // X = I_n: dW1[c][i] = sum_{b,j} dZ1[b,j,c] * A[b,j,i]
    g = GemmDesc();
    g.A = w.dZ1; g.lda = H; g.a_kc = false;
    g.B = adj; g.ldb = n; g.b_kc = false;
    g.C = dW1; g.ldc = f;
    g.M = H; g.N = n; g.K = rows;
    g.splits = gemm_pick_splits(H, n, rows, 1); g.slabs = w.slabs;
    return gemm(g, s);
}

// ---------------------------------------------------------------------------
// encoder on compact node rows (identity node features only).
// Rows [0, n_b) of graph b (n_b = 1 + highest node index any non-self edge
// touches) are stored back to back; every row past n_b equals a
// batch-independent prototype row, evaluated once per call for all n indices
// and merged into the max-pool as a suffix maximum.  Same results as the dense
// path, with the row-proportional GEMMs shrunk to the real rows.
// ---------------------------------------------------------------------------
struct EncCFwdWs {
    float *t2c, *tail;
    int32_t* tail_cnt;
    size_t tail_cnt_count;
    float* absmax;
};

static void enc_cfwd_layout(Carver& c, int64_t B, int n, int d, EncCFwdWs* w)
{
    size_t tb, cb;
    gemm_tail_workspace(&tb, &cb);
    w->t2c = c.take<float>((size_t)(B * n + n) * d);
    w->tail = reinterpret_cast<float*>(c.take<char>(tb));
    w->tail_cnt = reinterpret_cast<int32_t*>(c.take<char>(cb));
    w->tail_cnt_count = cb / sizeof(int32_t);
    w->absmax = reinterpret_cast<float*>(c.take<char>(GEMM_ABSMAX_BYTES));
}

size_t gtg_encoder_compact_fwd_workspace_bytes(int64_t num_graphs, int n, int d)
{
    Carver c(nullptr, 0);
    EncCFwdWs w;
    enc_cfwd_layout(c, num_graphs, n, d, &w);
    return c.off;
}

int gtg_encoder_compact_fwd(const float* adj, const int32_t* extents, const float* W1, const float* b1, const float* W2,
                            const float* b2, int64_t num_graphs, int n, int d, int32_t* meta, float* adjc, float* h1c,
                            float* pooled, int32_t* argmax, int matrix_mode, void* ws, size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(matrix_mode_valid(matrix_mode), "matrix_mode must be GTG_MATRIX_F32, _BF16, _F16X2 or _BF16X3");
    MatrixModeScope matrix_scope(matrix_mode);
    GTG_REQUIRE(num_graphs >= 1 && d >= 1, "bad size");
    GTG_REQUIRE(n >= 1 && n <= 64, "n must be in [1, 64]");
    GTG_REQUIRE(adj && extents && W1 && b1 && W2 && b2 && meta && adjc && h1c && pooled && argmax, "null pointer");
    GTG_REQUIRE((int64_t)num_graphs * n < (1ll << 31) / 4 && num_graphs <= GTG_MAX_GRAPHS_PER_CALL,
                "batch too large (at most 65000 graphs per call)");
    hipStream_t s = (hipStream_t)stream;
    const int B = (int)num_graphs, H = 2 * d, rows = B * n + n;
    Carver c(ws, ws_bytes);
    EncCFwdWs w;
    enc_cfwd_layout(c, B, n, d, &w);
    if (c.overflow || ws == nullptr) return set_error(GTG_ERR_WORKSPACE, "gtg_encoder_compact_fwd: workspace too small");
    const int32_t* n_real = meta;
    const int32_t* row_off = meta + B;
    const int32_t* real_rows = meta + 2 * B;      // R
    const int32_t* all_rows = meta + 2 * B + 1;   // R + n

    GTG_TRY(graph_meta(extents, adj, B, n, meta, adjc, w.tail_cnt, (int)w.tail_cnt_count, s));
    GemmDesc g;
    bool conv1_maxima = false;
    // H1c = relu(adjc * W1^T + b1) over the R + n stored rows: a graph's compact adjacency rows are zero past
    // its n_b columns, so one product with K = n serves every graph, and the unit rows appended to adjc give
    // the prototype rows relu(W1^T + b1) in the same launch.  A row of adjc holds deg + 1 nonzeros: the product
    // runs as a sparse row kernel bound by the store of H1c (dense GEMM when 2d is not a multiple of 4).
    if ((H & 3) == 0) {
        // (in the two-plane fp16 mode the kernel also leaves the maxima of H1c and W2 that the next product scales by)
        conv1_maxima = matrix_mode == GTG_MATRIX_F16X2;
        GTG_TRY(conv1_sparse(adjc, n, W1, b1, H, all_rows, rows, h1c, conv1_maxima ? w.absmax : nullptr, W2, (long long)d * H, s));
    } else {
        g.A = adjc; g.lda = n; g.a_kc = true;
        g.B = W1; g.ldb = n; g.b_kc = true;
        g.C = h1c; g.ldc = H;
        g.M = rows; g.N = H; g.K = n; g.m_dev = all_rows;
        g.bias = b1; g.epi = EPI_RELU;
        g.tail_ws = w.tail; g.tail_counters = w.tail_cnt;
        GTG_TRY(gemm(g, s));
    }
    // T2c = H1c * W2^T over the R + n stored rows
    g = GemmDesc();
    g.A = h1c; g.lda = H; g.B = W2; g.ldb = H; g.C = w.t2c; g.ldc = d;
    g.M = rows; g.N = d; g.K = H; g.m_dev = all_rows;
    g.tail_ws = w.tail; g.tail_counters = w.tail_cnt;
    g.absmax_ws = w.absmax;
    g.absmax_ready = conv1_maxima;
    GTG_TRY(gemm(g, s));
    // pooled[b] = max( max_{i<n_b}(A_b * T2c[b] + b2), max of the prototype rows from n_b on (+ b2) ): the
    // epilogue reads those rows of T2c itself.  Sparse per-graph kernel when d is a multiple of 4 (the adjacency
    // rows hold 3 to 4 nonzeros), the batched dense product otherwise.
    if ((d & 3) == 0) return pool_sparse(adj, w.t2c, meta, b2, B, n, d, pooled, argmax, s);
    g = GemmDesc();
    g.A = adj; g.lda = n; g.a_kc = true; g.strideA = (long long)n * n;
    g.B = w.t2c; g.ldb = d; g.b_kc = false; g.off_b = true;
    g.M = n; g.N = d; g.K = n; g.batch = B;
    g.batch_dim = n_real; g.batch_off = row_off;
    g.bias = b2; g.epi = EPI_POOL; g.pool_val = pooled; g.pool_idx = argmax;
    g.proto_base = real_rows;
    return gemm(g, s);
}

struct EncCBwdWs {
    float *dT2c, *dZ1c, *pg, *cs, *slabs, *tail, *dw2, *dw1;
    float* proto_rows;   // plane path: the prototype rows' gradient as fp32
    void* w2t_planes;    // plane path: W2^T as k-blocked bf16 planes
    float *inv_rows, *inv_w;  // fp16 planes: inverse row scales of dT2c (rows) and of W2^T (H)
    int32_t* tail_cnt;
    size_t tail_cnt_count;
};

static void enc_cbwd_layout(Carver& c, int64_t B, int n, int d, EncCBwdWs* w)
{
    const size_t rows = (size_t)B * n + n;
    const int H = 2 * d;
    // fp32 rows, or (GTG_MATRIX_BF16X3, plane path) three bf16 planes: sized for the larger
    w->dT2c = reinterpret_cast<float*>(c.take<char>(rows * d * 6));
    w->proto_rows = c.take<float>((size_t)n * d);
    w->w2t_planes = c.take<char>((size_t)d * H * 6);
    w->inv_rows = c.take<float>(rows);
    w->inv_w = c.take<float>((size_t)H);
    w->dZ1c = c.take<float>(rows * H);
    w->pg = reinterpret_cast<float*>(c.take<char>(proto_grad_workspace(n, d)));
    w->cs = reinterpret_cast<float*>(c.take<char>(max2(max2(colsum_workspace(B, d), colsum_workspace((int64_t)rows, H)),
                                                       (size_t)gemm_colsum_rows((int)rows) * H * sizeof(float))));
    w->slabs = reinterpret_cast<float*>(
        c.take<char>(max2(gemm_splitk_workspace(d, H, (int)rows, 1), gemm_splitk_workspace(H, n, (int)rows, 1))));
    // the sparse dW2 kernel's partials
    w->dw2 = reinterpret_cast<float*>(c.take<char>(dw2_sparse_applies(B, n, d, H) ? dw2_sparse_workspace(d, H) : 0));
    // the sparse dW1 kernel's partials
    w->dw1 = reinterpret_cast<float*>(c.take<char>(dw1_sparse_applies((int64_t)rows, n, H) ? dw1_sparse_workspace(H) : 0));
    size_t tb, cb;
    gemm_tail_workspace(&tb, &cb);
    w->tail = reinterpret_cast<float*>(c.take<char>(tb));
    w->tail_cnt = reinterpret_cast<int32_t*>(c.take<char>(cb));
    w->tail_cnt_count = cb / sizeof(int32_t);
}

size_t gtg_encoder_compact_bwd_workspace_bytes(int64_t num_graphs, int n, int d)
{
    Carver c(nullptr, 0);
    EncCBwdWs w;
    enc_cbwd_layout(c, num_graphs, n, d, &w);
    return c.off;
}

int gtg_encoder_compact_bwd(const float* adj, const float* adjc, const int32_t* meta, const float* W2, const float* h1c,
                            const int32_t* argmax, const float* dpooled, int64_t num_graphs, int n, int d, float* dW1,
                            float* db1, float* dW2, float* db2, int matrix_mode, void* ws, size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(matrix_mode_valid(matrix_mode), "matrix_mode must be GTG_MATRIX_F32, _BF16, _F16X2 or _BF16X3");
    MatrixModeScope matrix_scope(matrix_mode);
    GTG_REQUIRE(num_graphs >= 1 && d >= 1, "bad size");
    GTG_REQUIRE(n >= 1 && n <= 64, "n must be in [1, 64]");
    GTG_REQUIRE(adj && adjc && meta && W2 && h1c && argmax && dpooled && dW1 && db1 && dW2 && db2, "null pointer");
    GTG_REQUIRE((int64_t)num_graphs * n < (1ll << 31) / 4 && num_graphs <= GTG_MAX_GRAPHS_PER_CALL,
                "batch too large (at most 65000 graphs per call)");
    hipStream_t s = (hipStream_t)stream;
    const int B = (int)num_graphs, H = 2 * d, rows = B * n + n;
    Carver c(ws, ws_bytes);
    EncCBwdWs w;
    enc_cbwd_layout(c, B, n, d, &w);
    if (c.overflow || ws == nullptr) return set_error(GTG_ERR_WORKSPACE, "gtg_encoder_compact_bwd: workspace too small");
    const int32_t* all_rows = meta + 2 * B + 1;  // R + n

    // The plane path: in the BF16X3 mode, at sizes where dZ1c = (dT2c W2) masked runs on the 128x128 tiles and dW2
    // needs no fp32 dT2c, the gather writes dT2c straight as the three bf16 planes that GEMM consumes, W2^T is split
    // once, and the GEMM's loop carries no split arithmetic.
    const bool planes = matrix_mode_split(matrix_mode) && pool_bwd_gather_planes_applies(n, d) && dw2_sparse_applies(B, n, d, H) &&
                        (long long)((rows + 127) / 128) * ((H + 127) / 128) >= 192;
    // dT2c: real rows from the max-pool routing, prototype rows from the graphs whose maximum they won
    const bool f16 = planes && matrix_mode == GTG_MATRIX_F16X2;  // two fp16 planes, three products
    if (planes) {
        GTG_TRY(pool_bwd_gather_planes(adj, dpooled, argmax, meta, B, n, d, rows, w.dT2c, w.proto_rows, w.pg, w.tail_cnt,
                                       (int)w.tail_cnt_count, W2, d, H, H, w.w2t_planes, f16 ? w.inv_rows : nullptr,
                                       f16 ? w.inv_w : nullptr, s));
    } else {
        GTG_TRY(pool_bwd_gather_compact(adj, dpooled, argmax, meta, B, n, d, w.dT2c, w.pg, w.tail_cnt, (int)w.tail_cnt_count, s));
    }
    GemmDesc g;
    // dW2 = dT2c^T * H1c: from the pooled gradient and the winning rows where the shape allows (one aggregated
    // row of H1c per graph and pooled column), else as the dense product
    const bool dw1_sp = dw1_sparse_applies(rows, n, H);
    const bool dw2_deferred = dw1_sp && dw2_sparse_applies(B, n, d, H);  // dW2's partials are reduced beside dW1's
    if (dw2_sparse_applies(B, n, d, H)) {
        GTG_TRY(dw2_sparse(adj, h1c, H, dpooled, argmax, meta, w.dT2c, planes ? w.proto_rows : nullptr, B, n, d, w.dw2, dW2,
                           dw2_deferred, s));
    } else {
        g.A = w.dT2c; g.lda = d; g.a_kc = false;
        g.B = h1c; g.ldb = H; g.b_kc = false;
        g.C = dW2; g.ldc = H; g.M = d; g.N = H; g.K = rows; g.k_dev = all_rows;
        g.splits = gemm_pick_splits(d, H, rows, 1); g.slabs = w.slabs;
        GTG_TRY(gemm(g, s));
    }
    // dZ1c = (dT2c * W2) masked by relu
    g = GemmDesc();
    if (planes) {
        g.planes = true;
        g.A = w.dT2c; g.lda = rows; g.planeA = (long long)d * rows;
        g.B = reinterpret_cast<const float*>(w.w2t_planes); g.ldb = H; g.planeB = (long long)d * H;
        if (f16) { g.planes_inv_a = w.inv_rows; g.planes_inv_b = w.inv_w; }
    } else {
        g.A = w.dT2c; g.lda = d; g.a_kc = true;
        g.B = W2; g.ldb = H; g.b_kc = false;
    }
    g.C = w.dZ1c; g.ldc = H; g.M = rows; g.N = H; g.K = d; g.m_dev = all_rows;
    g.epi = EPI_MASK; g.aux = h1c; g.ldaux = H;
    g.tail_ws = w.tail; g.tail_counters = w.tail_cnt;
    // db1 = column sums of dZ1c: the GEMM's epilogue leaves them per 64-row block (no pass over dZ1c for them), one
    // launch adds the blocks and, beside them, db2 = column sums of dpooled.  Needs the 128x128 configuration and at
    // most 1024 blocks and graphs; otherwise a two-stage column sum over dZ1c.
    const bool fused_db1 = gemm_colsum_rows(rows) <= 1024 && B <= 1024 && (long long)((rows + 127) / 128) * ((H + 127) / 128) >= 192;
    if (fused_db1) g.colsum_part = w.cs;
    GTG_TRY(gemm(g, s));
    // (where the sparse dW1 path follows, the two column sums ride in the front blocks of its reduce launch: they
    // are final here and nothing before that launch's end needs them)
    SideJobs bias_sums;
    const bool riding = fused_db1 && dw1_sp &&
                        bias_sums.add_colsum(w.cs, gemm_colsum_rows(rows), H, H, db1, all_rows, 128, 2) &&
                        bias_sums.add_colsum(dpooled, B, d, d, db2);
    if (riding) {
    } else if (fused_db1) {
        GTG_TRY(colsum_parts_and_short(w.cs, gemm_colsum_rows(rows), all_rows, H, db1, dpooled, B, d, d, db2, nullptr, s));
    } else {
        GTG_TRY(colsum_dev_and_short(w.dZ1c, rows, all_rows, H, H, w.cs, db1, dpooled, B, d, d, db2, s));
    }
    // dW1[c][i] = sum over stored rows dZ1c[r][c] * adjc[r][i]   (prototype rows carry unit adjacency rows):
    // through the nonzeros of the adjacency rows where the shape allows, else as the dense product
    if (dw1_sp)
        return dw1_sparse(adjc, n, w.dZ1c, H, all_rows, rows, w.dw1, dW1, dw2_deferred ? w.dw2 : nullptr, d, dW2, s,
                          riding ? &bias_sums : nullptr);
    g = GemmDesc();
    g.A = w.dZ1c; g.lda = H; g.a_kc = false;
    g.B = adjc; g.ldb = n; g.b_kc = false;
    g.C = dW1; g.ldc = n; g.M = H; g.N = n; g.K = rows; g.k_dev = all_rows;
    g.splits = gemm_pick_splits(H, n, rows, 1); g.slabs = w.slabs;
    return gemm(g, s);
}

// ---------------------------------------------------------------------------
// latent head: mu / logvar linears, reparameterisation with supplied eps, KL
// ---------------------------------------------------------------------------
size_t gtg_latent_workspace_bytes(int64_t num_graphs, int d)
{
    Carver c(nullptr, 0);
    c.take<float>((size_t)((num_graphs * d + 255) / 256) + 1);          // KL block partials
    c.take<float>((size_t)2 * num_graphs * d);                           // dmu, dlogvar
    c.take<char>(colsum_workspace(num_graphs, d));
    c.take<char>(gemm_splitk_workspace(d, d, (int)num_graphs, 1));
    return c.off;
}

int gtg_latent_fwd(const float* h, const float* Wmu, const float* bmu, const float* Wlv, const float* blv,
                   const float* eps, int64_t num_graphs, int d, float* mu, float* logvar, float* z, float* kl,
                   void* ws, size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(num_graphs >= 1 && d >= 1, "bad size");
    GTG_REQUIRE(h && Wmu && bmu && Wlv && blv && eps && mu && logvar && z && kl, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int B = (int)num_graphs;
    Carver c(ws, ws_bytes);
    float* partial = c.take<float>((size_t)((num_graphs * d + 255) / 256) + 1);
    if (c.overflow || ws == nullptr) return set_error(GTG_ERR_WORKSPACE, "gtg_latent_fwd: workspace too small");
    // both linears in one launch: a batch of two over the same input, the second operand set addressed by its
    // element distance from the first
    GemmDesc g;
    g.A = h; g.lda = d; g.B = Wmu; g.ldb = d; g.C = mu; g.ldc = d;
    g.M = B; g.N = d; g.K = d; g.bias = bmu;
    g.batch = 2; g.strideA = 0; g.strideB = Wlv - Wmu; g.strideC = logvar - mu; g.strideBias = blv - bmu;
    GTG_TRY(gemm(g, s));
    return reparam_kl(mu, logvar, eps, B, d, z, partial, kl, s);
}

// The latent head's backward; dboth: 2 B d floats, cs: colsum_workspace(B, d), slabs: gemm_splitk_workspace(d, d, B, 1).
static int latent_bwd_launches(const float* h, const float* Wmu, const float* Wlv, const float* mu, const float* logvar,
                               const float* eps, const float* dz, float kl_w, const float* kl_w_dev, int B, int d,
                               float* dWmu, float* dbmu, float* dWlv, float* dblv, float* dh, float* dboth, float* cs,
                               float* slabs, hipStream_t s, bool elem_done = false)
{
    // When the two weight matrices lie back to back in memory, [Wmu; Wlv] is one (2d, d) operand: dmu and dlogvar
    // are then kept as the halves of (B, 2d) rows and dh = [dmu | dlogvar] [Wmu; Wlv] is a single GEMM.
    const bool stacked = (Wlv == Wmu + (size_t)d * d);
    const int ld = stacked ? 2 * d : d;
    float* dmu = dboth;
    float* dlv = stacked ? dboth + d : dboth + (size_t)B * d;

    // (elem_done: the product that wrote dz already left dmu and dlogvar in dboth, latent_fuse() below)
    if (!elem_done) GTG_TRY(latent_bwd_elem(dz, mu, logvar, eps, kl_w_dev, kl_w, B, d, ld, dmu, dlv, s));
    const int splits = gemm_pick_splits(d, d, B, 1);
    GemmDesc g, gh;
    // dW = dY^T * h
    g.A = dmu; g.lda = ld; g.a_kc = false;
    g.B = h; g.ldb = d; g.b_kc = false;
    g.C = dWmu; g.ldc = d; g.M = d; g.N = d; g.K = B;
    // dh = dmu * Wmu + dlogvar * Wlv
    gh.A = dmu; gh.lda = ld; gh.a_kc = true;
    gh.B = Wmu; gh.ldb = d; gh.b_kc = false;
    gh.C = dh; gh.ldc = d; gh.M = B; gh.N = d; gh.K = stacked ? 2 * d : d;
    if (splits > 1) {
        g.splits = splits; g.slabs = slabs;
        GTG_TRY(gemm(g, s));
        g.A = dlv; g.C = dWlv;
        GTG_TRY(gemm(g, s));
        GTG_TRY(gemm(gh, s));
    } else {
        // both weight gradients as a batch of two, and dh beside them: one launch, in whose first blocks the two
        // bias gradients (column sums of dmu and dlogvar, which this launch only reads) ride as side jobs
        g.batch = 2; g.strideA = dlv - dmu; g.strideB = 0; g.strideC = dWlv - dWmu;
        SideJobs side;
        const bool riding = side.add_colsum(dmu, B, d, ld, dbmu) && side.add_colsum(dlv, B, d, ld, dblv);
        if (riding) g.side = &side;
        GTG_TRY(gemm_pair(g, gh, s));
        if (!stacked) {
            gh.A = dlv; gh.B = Wlv; gh.epi = EPI_ACCUM;
            GTG_TRY(gemm(gh, s));
        }
        if (riding) return GTG_OK;
        return colsum_pair(dmu, B, d, ld, dbmu, dlv, B, d, ld, dblv, cs, s);
    }
    if (!stacked) {
        gh.A = dlv; gh.B = Wlv; gh.epi = EPI_ACCUM;
        GTG_TRY(gemm(gh, s));
    }
    return colsum_pair(dmu, B, d, ld, dbmu, dlv, B, d, ld, dblv, cs, s);
}

int gtg_latent_bwd(const float* h, const float* Wmu, const float* Wlv, const float* mu, const float* logvar,
                   const float* eps, const float* dz, float kl_w, const float* kl_w_dev, int64_t num_graphs, int d,
                   float* dWmu, float* dbmu, float* dWlv, float* dblv, float* dh, void* ws, size_t ws_bytes,
                   void* stream)
{
    GTG_REQUIRE(num_graphs >= 1 && d >= 1, "bad size");
    GTG_REQUIRE(h && Wmu && Wlv && mu && logvar && eps && dWmu && dbmu && dWlv && dblv && dh, "null pointer");
    const int B = (int)num_graphs;
    Carver c(ws, ws_bytes);
    c.take<float>((size_t)((num_graphs * d + 255) / 256) + 1);
    float* dboth = c.take<float>((size_t)2 * B * d);
    float* cs = reinterpret_cast<float*>(c.take<char>(colsum_workspace(B, d)));
    float* slabs = reinterpret_cast<float*>(c.take<char>(gemm_splitk_workspace(d, d, B, 1)));
    if (c.overflow || ws == nullptr) return set_error(GTG_ERR_WORKSPACE, "gtg_latent_bwd: workspace too small");
    return latent_bwd_launches(h, Wmu, Wlv, mu, logvar, eps, dz, kl_w, kl_w_dev, B, d, dWmu, dbmu, dWlv, dblv, dh, dboth,
                               cs, slabs, (hipStream_t)stream);
}

// Stand-alone reparameterisation + KL (GTVAE.reparameterize / GTVAE.loss_kl as
// separate calls).  Workspace: gtg_latent_workspace_bytes.
int gtg_reparam_kl_fwd(const float* mu, const float* logvar, const float* eps, int64_t num_graphs, int d, float* z,
                       float* kl, void* ws, size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(num_graphs >= 1 && d >= 1, "bad size");
    GTG_REQUIRE(mu && logvar && eps && z && kl, "null pointer");
    Carver c(ws, ws_bytes);
    float* partial = c.take<float>((size_t)((num_graphs * d + 255) / 256) + 1);
    if (c.overflow || ws == nullptr) return set_error(GTG_ERR_WORKSPACE, "gtg_reparam_kl_fwd: workspace too small");
    return reparam_kl(mu, logvar, eps, num_graphs, d, z, partial, kl, (hipStream_t)stream);
}

int gtg_reparam_kl_bwd(const float* mu, const float* logvar, const float* eps, const float* dz, float kl_w,
                       const float* kl_w_dev, int64_t num_graphs, int d, float* dmu, float* dlogvar, void* stream)
{
    GTG_REQUIRE(num_graphs >= 1 && d >= 1, "bad size");
    GTG_REQUIRE(mu && logvar && eps && dmu && dlogvar, "null pointer");
    return latent_bwd_elem(dz, mu, logvar, eps, kl_w_dev, kl_w, num_graphs, d, d, dmu, dlogvar, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// MLP edge decoder + un-masked upper-triangular BCE
// ---------------------------------------------------------------------------
int gtg_decoder_fwd(const float* z, const float* D1, const float* c1, const float* D2, const float* c2,
                    int64_t num_graphs, int d, int hid, int m_out, float* y1, float* out, void* stream)
{
    GTG_REQUIRE(num_graphs >= 1 && d >= 1 && hid >= 1 && m_out >= 1, "bad size");
    GTG_REQUIRE(z && D1 && c1 && D2 && c2 && y1 && out, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int B = (int)num_graphs;
    GemmDesc g;
    g.A = z; g.lda = d; g.B = D1; g.ldb = d; g.C = y1; g.ldc = hid;
    g.M = B; g.N = hid; g.K = d; g.bias = c1; g.epi = EPI_RELU;
    GTG_TRY(gemm(g, s));
    g = GemmDesc();
    g.A = y1; g.lda = hid; g.B = D2; g.ldb = hid; g.C = out; g.ldc = m_out;
    g.M = B; g.N = m_out; g.K = hid; g.bias = c2; g.epi = EPI_SIGMOID;
    return gemm(g, s);
}

size_t gtg_bce_workspace_bytes(int64_t num_graphs)
{
    Carver c(nullptr, 0);
    c.take<float>((size_t)num_graphs + 1);
    return c.off;
}

int gtg_bce_loss_fwd(const float* out, const float* adj_target, int64_t num_graphs, int n, float* loss, float* dlogits,
                     const float* gscale_dev, void* ws, size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(num_graphs >= 1 && n >= 1, "bad size");
    GTG_REQUIRE(out && adj_target && loss, "null pointer");
    Carver c(ws, ws_bytes);
    float* partial = c.take<float>((size_t)num_graphs + 1);
    if (c.overflow || ws == nullptr) return set_error(GTG_ERR_WORKSPACE, "gtg_bce_loss_fwd: workspace too small");
    return bce_fwd(out, adj_target, num_graphs, n, partial, loss, dlogits, gscale_dev, (hipStream_t)stream);
}

// gtg_bce_loss_fwd for callers that cannot allocate: no workspace argument, the per-graph partial sums go to
// a device array that is part of the library's code object.
int64_t gtg_bce_loss_fwd_noalloc_max_graphs(void) { return bce_static_max_graphs(); }

int gtg_bce_loss_fwd_noalloc(const float* out, const float* adj_target, int64_t num_graphs, int n, float* loss,
                             float* dlogits, const float* gscale_dev, void* stream)
{
    GTG_REQUIRE(num_graphs >= 1 && n >= 1 && n <= 64, "bad size");
    GTG_REQUIRE(out && adj_target && loss, "null pointer");
    return bce_fwd_static(out, adj_target, num_graphs, n, loss, dlogits, gscale_dev, (hipStream_t)stream);
}

int gtg_bce_loss_bwd(const float* out, const float* adj_target, float gscale, const float* gscale_dev,
                     int64_t num_graphs, int n, int through_sigmoid, float* grad, void* stream)
{
    GTG_REQUIRE(num_graphs >= 1 && n >= 1, "bad size");
    GTG_REQUIRE(out && adj_target && grad, "null pointer");
    return bce_bwd(out, adj_target, num_graphs, n, gscale_dev, gscale, through_sigmoid, grad, (hipStream_t)stream);
}

struct DecBwdWs {
    float *dy2, *dy1, *cs, *slabs;
};

static void dec_bwd_layout(Carver& c, int64_t B, int d, int hid, int m_out, DecBwdWs* w)
{
    w->dy2 = c.take<float>((size_t)B * m_out);
    w->dy1 = c.take<float>((size_t)B * hid);
    w->cs = reinterpret_cast<float*>(c.take<char>(max2(colsum_workspace(B, m_out), colsum_workspace(B, hid))));
    w->slabs = reinterpret_cast<float*>(
        c.take<char>(max2(gemm_splitk_workspace(m_out, hid, (int)B, 1), gemm_splitk_workspace(hid, d, (int)B, 1))));
}

size_t gtg_decoder_bwd_workspace_bytes(int64_t num_graphs, int d, int hid, int m_out)
{
    Carver c(nullptr, 0);
    DecBwdWs w;
    dec_bwd_layout(c, num_graphs, d, hid, m_out, &w);
    return c.off;
}

static int decoder_bwd_launches(const float* z, const float* D1, const float* D2, const float* y1, const float* dy2, int B,
                                int d, int hid, int m_out, float* dD1, float* dc1, float* dD2, float* dc2, float* dz,
                                const DecBwdWs& w, const SideJobs* extra, const GemmFuse* latent, hipStream_t s);

// Exactly one of (dout) and (adj_target) is given:
//   dout        gradient w.r.t. the decoder's sigmoid output (generic autograd path)
//   adj_target  fuse the BCE backward: dy2 is produced straight from out/targets
int gtg_decoder_bwd(const float* z, const float* D1, const float* D2, const float* y1, const float* out,
                    const float* dout, const float* adj_target, int n, float gscale, const float* gscale_dev,
                    int64_t num_graphs, int d, int hid, int m_out, float* dD1, float* dc1, float* dD2, float* dc2,
                    float* dz, void* ws, size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(num_graphs >= 1 && d >= 1 && hid >= 1 && m_out >= 1, "bad size");
    GTG_REQUIRE(z && D1 && D2 && y1 && dD1 && dc1 && dD2 && dc2 && dz, "null pointer");
    GTG_REQUIRE((dout != nullptr) != (adj_target != nullptr), "give exactly one of dout / adj_target");
    GTG_REQUIRE(out != nullptr || dout != nullptr, "out may be NULL only when dout holds the gradient w.r.t. the logits");
    GTG_REQUIRE(adj_target == nullptr || m_out == n * (n + 1) / 2, "m_out must be n(n+1)/2 with fused BCE");
    hipStream_t s = (hipStream_t)stream;
    const int B = (int)num_graphs;
    Carver c(ws, ws_bytes);
    DecBwdWs w;
    dec_bwd_layout(c, B, d, hid, m_out, &w);
    if (c.overflow || ws == nullptr) return set_error(GTG_ERR_WORKSPACE, "gtg_decoder_bwd: workspace too small");

    const float* dy2 = w.dy2;
    if (adj_target)
        GTG_TRY(bce_bwd(out, adj_target, B, n, gscale_dev, gscale, 1, w.dy2, s));
    else if (out)
        GTG_TRY(sigmoid_bwd(out, dout, (int64_t)B * m_out, w.dy2, s));
    else
        dy2 = dout;  // already the gradient w.r.t. the logits (gtg_bce_loss_fwd's dlogits)
    return decoder_bwd_launches(z, D1, D2, y1, dy2, B, d, hid, m_out, dD1, dc1, dD2, dc2, dz, w, nullptr, nullptr, s);
}

// The decoder's backward from the gradient w.r.t. its logits on.  `extra` (nullable): side jobs of the caller that
// ride with the first product's launch.  `latent` (nullable): the product that writes dz goes on, in its epilogue,
// to the gradients w.r.t. mu and logvar (EPI_LATENT_BWD).
static int decoder_bwd_launches(const float* z, const float* D1, const float* D2, const float* y1, const float* dy2, int B,
                                int d, int hid, int m_out, float* dD1, float* dc1, float* dD2, float* dc2, float* dz,
                                const DecBwdWs& w, const SideJobs* extra, const GemmFuse* latent, hipStream_t s)
{
    // Each weight gradient is independent of the activation gradient beside it: gemm_pair puts the two in one
    // launch where their shapes allow.  The bias gradients are column sums of matrices these launches only read:
    // dc2 rides with the first, dc1 (of dy1, which the first one writes) with the second.
    SideJobs side2, side1;
    if (extra != nullptr) side2 = *extra;
    const bool riding = side2.add_colsum(dy2, B, m_out, m_out, dc2) && side1.add_colsum(w.dy1, B, hid, hid, dc1);
    if (!riding) {
        side2 = SideJobs();
        if (extra != nullptr) side2 = *extra;
    }
    GemmDesc g, ga;
    // dD2 = dy2^T * y1
    g.A = dy2; g.lda = m_out; g.a_kc = false;
    g.B = y1; g.ldb = hid; g.b_kc = false;
    g.C = dD2; g.ldc = hid; g.M = m_out; g.N = hid; g.K = B;
    g.splits = gemm_pick_splits(m_out, hid, B, 1); g.slabs = w.slabs;
    if (side2.blocks() > 0) g.side = &side2;
    // dy1 = (dy2 * D2) masked by relu
    ga.A = dy2; ga.lda = m_out; ga.a_kc = true;
    ga.B = D2; ga.ldb = hid; ga.b_kc = false;
    ga.C = w.dy1; ga.ldc = hid; ga.M = B; ga.N = hid; ga.K = m_out;
    ga.epi = EPI_MASK; ga.aux = y1; ga.ldaux = hid;
    GTG_TRY(gemm_pair(g, ga, s));
    // dD1 = dy1^T * z
    g = GemmDesc();
    g.A = w.dy1; g.lda = hid; g.a_kc = false;
    g.B = z; g.ldb = d; g.b_kc = false;
    g.C = dD1; g.ldc = d; g.M = hid; g.N = d; g.K = B;
    g.splits = gemm_pick_splits(hid, d, B, 1); g.slabs = w.slabs;
    if (riding) g.side = &side1;
    // dz = dy1 * D1
    ga = GemmDesc();
    ga.A = w.dy1; ga.lda = hid; ga.a_kc = true;
    ga.B = D1; ga.ldb = d; ga.b_kc = false;
    ga.C = dz; ga.ldc = d; ga.M = B; ga.N = d; ga.K = hid;
    if (latent != nullptr) {
        ga.epi = EPI_LATENT_BWD;
        ga.fuse = *latent;
    }
    GTG_TRY(gemm_pair(g, ga, s));
    if (riding) return GTG_OK;
    // (more than 1024 graphs) both bias gradients in a launch of their own
    return colsum_pair(dy2, B, m_out, m_out, dc2, w.dy1, B, hid, hid, dc1, w.cs, s);
}

// ---------------------------------------------------------------------------
// Latent head + decoder + reconstruction loss, forward and backward, as one call: the launches of
// gtg_latent_fwd, gtg_decoder_fwd, gtg_bce_loss_fwd, gtg_decoder_bwd and gtg_latent_bwd in that order, except that
// what those calls must finish inside their own boundaries rides here in launches that follow: the final sums of
// the two loss terms and (optionally) Adam's step-counter kernel become side jobs of the decoder's first
// weight-gradient launch.  Every output is bit-identical to the five calls'.
// ---------------------------------------------------------------------------
struct HeadWs {
    float *klp, *bcep, *dboth, *cs, *slabs;
    DecBwdWs dec;
};

static void head_layout(Carver& c, int64_t B, int d, int hid, int m_out, HeadWs* w)
{
    w->klp = c.take<float>((size_t)reparam_kl_partials(B, d) + 1);
    // loss partials: one per graph (bce_fwd_kernel) or one per wave of the decoder's last product (at most four
    // per 32x32 outputs: gemm_bce_partials)
    const size_t tiles = 4 * (size_t)((B + 31) / 32) * (size_t)((m_out + 31) / 32);
    w->bcep = c.take<float>((tiles > (size_t)B ? tiles : (size_t)B) + 1);
    w->dboth = c.take<float>((size_t)2 * B * d);
    w->cs = reinterpret_cast<float*>(c.take<char>(colsum_workspace(B, d)));
    w->slabs = reinterpret_cast<float*>(c.take<char>(gemm_splitk_workspace(d, d, (int)B, 1)));
    dec_bwd_layout(c, B, d, hid, m_out, &w->dec);
}

size_t gtg_head_workspace_bytes(int64_t num_graphs, int n, int d, int hid)
{
    Carver c(nullptr, 0);
    HeadWs w;
    head_layout(c, num_graphs, d, hid, n * (n + 1) / 2, &w);
    return c.off;
}

int gtg_head_fwd_bwd(const float* h, const float* Wmu, const float* bmu, const float* Wlv, const float* blv,
                     const float* D1, const float* c1, const float* D2, const float* c2, const float* eps,
                     const float* adj_target, int64_t num_graphs, int n, int d, int hid, const float* kl_w_dev,
                     const float* gscale_dev, float* mu, float* logvar, float* z, float* y1, float* out, float* dlogits,
                     float* loss, float* kl, float* dWmu, float* dbmu, float* dWlv, float* dblv, float* dD1, float* dc1,
                     float* dD2, float* dc2, float* dz, float* dh, void* adam_state16, float lr, float beta1, float beta2,
                     void* ws, size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(num_graphs >= 1 && n >= 1 && d >= 1 && hid >= 1, "bad size");
    GTG_REQUIRE(h && Wmu && bmu && Wlv && blv && D1 && c1 && D2 && c2 && eps && adj_target, "null input");
    GTG_REQUIRE(mu && logvar && z && y1 && out && dlogits && loss && kl, "null output");
    GTG_REQUIRE(dWmu && dbmu && dWlv && dblv && dD1 && dc1 && dD2 && dc2 && dz && dh, "null gradient");
    hipStream_t s = (hipStream_t)stream;
    const int B = (int)num_graphs;
    const int m_out = n * (n + 1) / 2;
    Carver c(ws, ws_bytes);
    HeadWs w;
    head_layout(c, B, d, hid, m_out, &w);
    if (c.overflow || ws == nullptr) return set_error(GTG_ERR_WORKSPACE, "gtg_head_fwd_bwd: workspace too small");

    // ---- forward
    GemmDesc g;
    g.A = h; g.lda = d; g.B = Wmu; g.ldb = d; g.C = mu; g.ldc = d;
    g.M = B; g.N = d; g.K = d; g.bias = bmu;
    g.batch = 2; g.strideA = 0; g.strideB = Wlv - Wmu; g.strideC = logvar - mu; g.strideBias = blv - bmu;
    GTG_TRY(gemm(g, s));
    GTG_TRY(reparam_kl(mu, logvar, eps, B, d, z, w.klp, nullptr, s));
    g = GemmDesc();
    g.A = z; g.lda = d; g.B = D1; g.ldb = d; g.C = y1; g.ldc = hid;
    g.M = B; g.N = hid; g.K = d; g.bias = c1; g.epi = EPI_RELU;
    GTG_TRY(gemm(g, s));
    g = GemmDesc();
    g.A = y1; g.lda = hid; g.B = D2; g.ldb = hid; g.C = out; g.ldc = m_out;
    g.M = B; g.N = m_out; g.K = hid; g.bias = c2; g.epi = EPI_SIGMOID;
    // On a batch-sized product the loss is the epilogue of the decoder's last layer: the probabilities are stored
    // and, from the registers that hold them, turned into loss terms and the gradient w.r.t. the logits -- element
    // for element what bce_fwd_kernel computes; the loss itself is summed from per-wave instead of per-graph
    // partials (the one output of this call that differs from the separate calls', in its last bits).
    long long bce_parts = B;
    if (gemm_fused_epilogue_ok(B, m_out, hid)) {
        g.epi = EPI_SIGMOID_BCE;
        g.fuse.bce_target = adj_target;
        g.fuse.bce_n = n;
        g.fuse.bce_gscale = 1.f / ((float)B * (float)m_out);
        g.fuse.bce_gscale_dev = gscale_dev;
        g.fuse.bce_dlogits = dlogits;
        g.fuse.bce_partial = w.bcep;
        bce_parts = gemm_bce_partials(B, m_out, hid);
        GTG_TRY(gemm(g, s));
    } else {
        GTG_TRY(gemm(g, s));
        GTG_TRY(bce_fwd(out, adj_target, B, n, w.bcep, nullptr, dlogits, gscale_dev, s));
    }

    // ---- backward; the loss terms' final sums and Adam's step counter ride with its first launch
    SideJobs extra;
    bool ok = extra.add_sum(w.klp, reparam_kl_partials(B, d), reparam_kl_scale(B), kl);
    ok = extra.add_sum(w.bcep, bce_parts, bce_scale(B, n), loss) && ok;
    if (adam_state16 != nullptr) ok = extra.add_adam_prepare(adam_state16, lr, beta1, beta2) && ok;
    if (!ok) return set_error(GTG_ERR_LAUNCH, "gtg_head_fwd_bwd: side job list full");
    // the gradients w.r.t. mu and logvar as the epilogue of the product that writes dz, where that product is a
    // batch-sized one (the same arithmetic as latent_bwd_kernel: bit-identical)
    const bool stacked = (Wlv == Wmu + (size_t)d * d);
    GemmFuse lat;
    lat.mu = mu; lat.lv = logvar; lat.eps = eps;
    lat.kl_w_dev = kl_w_dev; lat.kl_w = 1.f; lat.inv_b = 1.f / (float)B;
    lat.dmu = w.dboth;
    lat.dlv = stacked ? w.dboth + d : w.dboth + (size_t)B * d;
    lat.ld_out = stacked ? 2 * d : d;
    const bool fuse_lat = gemm_fused_epilogue_ok(B, d, hid) && d % 4 == 0 &&
                          ((reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(logvar) |
                            reinterpret_cast<uintptr_t>(eps) | reinterpret_cast<uintptr_t>(w.dboth)) & 15) == 0;
    GTG_TRY(decoder_bwd_launches(z, D1, D2, y1, dlogits, B, d, hid, m_out, dD1, dc1, dD2, dc2, dz, w.dec, &extra,
                                 fuse_lat ? &lat : nullptr, s));
    return latent_bwd_launches(h, Wmu, Wlv, mu, logvar, eps, dz, 1.f, kl_w_dev, B, d, dWmu, dbmu, dWlv, dblv, dh,
                               w.dboth, w.cs, w.slabs, s, fuse_lat);
}

// ---------------------------------------------------------------------------
// batch staging (one launch instead of four device-to-device copies)
// ---------------------------------------------------------------------------
int gtg_stage_batch(const int64_t* edge_index, int64_t num_edges, int64_t* edge_buf, int64_t edge_capacity,
                    const float* adj_target, float* target_buf, int64_t target_count, const float* eps,
                    float* eps_buf, int64_t eps_count, void* stream)
{
    GTG_REQUIRE(num_edges >= 0 && edge_capacity >= num_edges && target_count >= 0 && eps_count >= 0, "bad size");
    GTG_REQUIRE((edge_capacity == 0 || edge_buf) && (num_edges == 0 || edge_index), "null edge pointer");
    GTG_REQUIRE((target_count == 0 || (adj_target && target_buf)) && (eps_count == 0 || (eps && eps_buf)), "null pointer");
    return stage_batch(edge_index, num_edges, edge_buf, edge_capacity, adj_target, target_buf, target_count, eps, eps_buf,
                       eps_count, (hipStream_t)stream);
}

int gtg_gather_batch(const int64_t* edges_local, int64_t total_edges, const int64_t* edge_start, const float* adj_all,
                     int64_t num_dataset_graphs, int n, const int64_t* graph_ids, int64_t num_graphs, int64_t* offsets,
                     int64_t* edge_buf, int64_t edge_capacity, float* target_buf, int32_t* err, void* stream)
{
    GTG_REQUIRE(num_graphs >= 1 && num_graphs < (1 << 30) && n >= 1 && num_dataset_graphs >= 1 && total_edges >= 0 &&
                    edge_capacity >= 0,
                "bad size");
    GTG_REQUIRE(edge_start && adj_all && graph_ids && offsets && target_buf && err && (total_edges == 0 || edges_local) &&
                    (edge_capacity == 0 || edge_buf),
                "null pointer");
    return gather_batch(edges_local, total_edges, edge_start, adj_all, num_dataset_graphs, n, graph_ids, (int)num_graphs,
                        offsets, edge_buf, edge_capacity, target_buf, err, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// one-layer LSTM over a padded sequence
// ---------------------------------------------------------------------------
size_t gtg_lstm_fwd_workspace_bytes(int T, int B, int I, int H)
{
    return lstm_applies(T, B, I, H) ? lstm_fwd_workspace(T, B, I, H) : 0;
}

size_t gtg_lstm_bwd_workspace_bytes(int T, int B, int I, int H)
{
    return lstm_applies(T, B, I, H) ? lstm_bwd_workspace(T, B, I, H) : 0;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int gtg_lstm_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                 const float* h0, const float* c0, int T, int B, int I, int H, int reverse, float* hs, float* cs,
                 float* gates, int matrix_mode, void* ws, size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(matrix_mode_valid(matrix_mode), "matrix_mode must be GTG_MATRIX_F32, _BF16, _F16X2 or _BF16X3");
    MatrixModeScope matrix_scope(matrix_mode);
    GTG_REQUIRE(T >= 1 && B >= 1 && I >= 1 && H >= 1, "bad size");
    GTG_REQUIRE(x && w_ih && w_hh && b_ih && b_hh && hs && cs && gates, "null pointer");
    GTG_REQUIRE(aligned16(x) && aligned16(w_ih) && aligned16(w_hh) && aligned16(b_ih) && aligned16(b_hh) && aligned16(h0) &&
                    aligned16(c0) && aligned16(hs) && aligned16(cs) && aligned16(gates) && aligned16(ws),
                "every buffer must be 16-byte aligned");
    return lstm_fwd(x, w_ih, w_hh, b_ih, b_hh, h0, c0, T, B, I, H, reverse != 0, hs, cs, gates, ws, ws_bytes,
                    (hipStream_t)stream);
}

int gtg_lstm_bwd(const float* x, const float* w_ih, const float* w_hh, const float* h0, const float* c0, const float* hs,
                 const float* cs, const float* gates, const float* d_hs, const float* d_hT, const float* d_cT, int T,
                 int B, int I, int H, int reverse, float* dx, float* dw_ih, float* dw_hh, float* db, float* dh0,
                 float* dc0, int matrix_mode, void* ws, size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(matrix_mode_valid(matrix_mode), "matrix_mode must be GTG_MATRIX_F32, _BF16, _F16X2 or _BF16X3");
    MatrixModeScope matrix_scope(matrix_mode);
    GTG_REQUIRE(T >= 1 && B >= 1 && I >= 1 && H >= 1, "bad size");
    GTG_REQUIRE(x && w_ih && w_hh && hs && cs && gates && dw_ih && dw_hh && db, "null pointer");
    GTG_REQUIRE(aligned16(x) && aligned16(w_ih) && aligned16(w_hh) && aligned16(h0) && aligned16(c0) && aligned16(hs) &&
                    aligned16(cs) && aligned16(gates) && aligned16(d_hs) && aligned16(d_hT) && aligned16(d_cT) &&
                    aligned16(dx) && aligned16(dw_ih) && aligned16(dw_hh) && aligned16(db) && aligned16(dh0) &&
                    aligned16(dc0) && aligned16(ws),
                "every buffer must be 16-byte aligned");
    return lstm_bwd(x, w_ih, w_hh, h0, c0, hs, cs, gates, d_hs, d_hT, d_cT, T, B, I, H, reverse != 0, dx, dw_ih, dw_hh, db,
                    dh0, dc0, ws, ws_bytes, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// text branch: word drop, embedding lookup + dropout, dropout
// ---------------------------------------------------------------------------
int gtg_word_drop(const int64_t* tokens, int T, int B, int64_t padidx, float p, uint64_t seed, uint64_t stream_id,
                  const uint8_t* keep, int64_t* out, void* stream)
{
    GTG_REQUIRE(T >= 1 && B >= 1 && p >= 0.f && p <= 1.f, "bad size or probability");
    GTG_REQUIRE(tokens && out && tokens != out, "null pointer, or input and output alias");
    return word_drop(tokens, T, B, padidx, p, seed, stream_id, keep, out, (hipStream_t)stream);
}

int gtg_embed_dropout_fwd(const float* table, const int64_t* ids, int64_t rows, int64_t vocab, int dim, float p,
                          uint64_t seed, uint64_t stream_id, const float* add, int add_rows, float* out, int32_t* err,
                          void* stream)
{
    GTG_REQUIRE(rows >= 1 && rows < (1ll << 31) && vocab >= 1 && dim >= 4 && dim % 4 == 0 && p >= 0.f && p < 1.f,
                "bad size or probability (dim must be a multiple of 4)");
    GTG_REQUIRE(table && ids && out && (add == nullptr || add_rows >= 1), "null pointer");
    GTG_REQUIRE(aligned16(table) && aligned16(out) && aligned16(add), "table, add and out must be 16-byte aligned");
    return embed_dropout_fwd(table, ids, rows, vocab, dim, p, seed, stream_id, add, add_rows, out, err, (hipStream_t)stream);
}

size_t gtg_embed_dropout_bwd_workspace_bytes(int64_t rows, int64_t vocab, int dim, int add_rows)
{
    Carver c(nullptr, 0);
    c.take<char>(embed_dropout_bwd_workspace(rows, vocab, dim));
    if (add_rows > 0) c.take<char>(colsum_workspace(rows / add_rows, add_rows * dim));
    return c.off;
}

int gtg_embed_dropout_bwd(const float* dout, const int64_t* ids, int64_t rows, int64_t vocab, int dim, float p,
                          uint64_t seed, uint64_t stream_id, float* dtable, float* d_add, int add_rows, void* ws,
                          size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(rows >= 1 && vocab >= 1 && dim >= 4 && dim % 4 == 0 && p >= 0.f && p < 1.f,
                "bad size or probability (dim must be a multiple of 4)");
    GTG_REQUIRE(dout && ids && (dtable || d_add), "null pointer");
    GTG_REQUIRE(d_add == nullptr || (add_rows >= 1 && rows % add_rows == 0 && (int64_t)add_rows * dim < (1ll << 31)),
                "rows must be a multiple of add_rows");
    GTG_REQUIRE(aligned16(dout) && aligned16(dtable) && aligned16(d_add) && aligned16(ws),
                "dout, dtable, d_add and ws must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    Carver c(ws, ws_bytes);
    char* scatter_ws = c.take<char>(embed_dropout_bwd_workspace(rows, vocab, dim));
    float* cs = d_add ? reinterpret_cast<float*>(c.take<char>(colsum_workspace(rows / add_rows, add_rows * dim))) : nullptr;
    if (c.overflow || ws == nullptr) return set_error(GTG_ERR_WORKSPACE, "gtg_embed_dropout_bwd: workspace too small");
    if (dtable)
        GTG_TRY(embed_dropout_bwd(dout, ids, rows, vocab, dim, p, seed, stream_id, dtable, scatter_ws,
                                  embed_dropout_bwd_workspace(rows, vocab, dim), s));
    if (d_add)  // the added rows are shared by the rows r, r + add_rows, ...: column sums of dout seen as (rows / add_rows, add_rows * dim)
        GTG_TRY(colsum(dout, rows / add_rows, add_rows * dim, add_rows * dim, cs, d_add, s));
    return GTG_OK;
}

int gtg_dropout(const float* x, int64_t count, float p, uint64_t seed, uint64_t stream_id, float* y, void* stream)
{
    GTG_REQUIRE(count >= 0 && count % 4 == 0 && p >= 0.f && p < 1.f, "count must be a multiple of 4, p in [0, 1)");
    if (count == 0) return GTG_OK;
    GTG_REQUIRE(x && y && aligned16(x) && aligned16(y), "null or misaligned pointer");
    return dropout_map(x, count, p, seed, stream_id, y, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// optimiser
// ---------------------------------------------------------------------------
int gtg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel, void* state16,
                  float lr, float beta1, float beta2, float eps, float grad_scale, void* stream)
{
    GTG_REQUIRE(numel >= 0, "negative size");
    GTG_REQUIRE(state16 && (numel == 0 || (param && grad && exp_avg && exp_avg_sq)), "null pointer");
    return adam_stateful(param, grad, exp_avg, exp_avg_sq, numel, state16, lr, beta1, beta2, eps, grad_scale, false,
                         (hipStream_t)stream);
}

int gtg_adam_update(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                    const void* state16, float beta1, float beta2, float eps, float grad_scale, void* stream)
{
    GTG_REQUIRE(numel >= 0, "negative size");
    GTG_REQUIRE(state16 && (numel == 0 || (param && grad && exp_avg && exp_avg_sq)), "null pointer");
    return adam_stateful(param, grad, exp_avg, exp_avg_sq, numel, const_cast<void*>(state16), 0.f, beta1, beta2, eps,
                         grad_scale, true, (hipStream_t)stream);
}

// Plain GEMM entry (tests, and callers that want one of the three layouts).
int gtg_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int a_kc,
                 int b_kc, const float* bias, int epilogue, const float* aux, int ldaux, void* slabs,
                 size_t slab_bytes, int matrix_mode, void* stream)
{
    GTG_REQUIRE(matrix_mode_valid(matrix_mode), "matrix_mode must be GTG_MATRIX_F32, _BF16, _F16X2 or _BF16X3");
    MatrixModeScope matrix_scope(matrix_mode);
    GTG_REQUIRE(epilogue >= EPI_STORE && epilogue <= EPI_ACCUM, "unsupported epilogue");
    GemmDesc g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_kc = a_kc != 0; g.b_kc = b_kc != 0; g.bias = bias; g.epi = epilogue; g.aux = aux; g.ldaux = ldaux;
    // workspace: the split-K slabs (when the shape splits) followed by the scratch of the two-plane fp16 mode
    size_t used = 0;
    if (slabs && !bias && (epilogue == EPI_STORE || epilogue == EPI_ACCUM)) {
        const int sp = gemm_pick_splits(M, N, K, 1);
        if (sp > 1 && (size_t)sp * M * N * sizeof(float) <= slab_bytes) {
            g.splits = sp;
            g.slabs = reinterpret_cast<float*>(slabs);
            used = (size_t)sp * M * N * sizeof(float);
        }
    }
    if (slabs && used + GEMM_ABSMAX_BYTES <= slab_bytes)
        g.absmax_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(slabs) + used);
    return gemm(g, (hipStream_t)stream);
}

size_t gtg_gemm_splitk_workspace_bytes(int M, int N, int K) { return gemm_splitk_workspace(M, N, K, 1) + GEMM_ABSMAX_BYTES; }

// GEMM whose row count is read on the device (rows_dev, nullable), with the
// tail split: C[:min(M,*rows_dev)] = op(A) * op(B).  Test / profiling hook
// for the ragged encoder GEMMs.
size_t gtg_gemm_rows_workspace_bytes(void)
{
    size_t tb, cb;
    gemm_tail_workspace(&tb, &cb);
    return tb + cb + GEMM_ABSMAX_BYTES;
}

int gtg_gemm_f32_rows(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int a_kc,
                      int b_kc, const float* bias, int epilogue, const float* aux, int ldaux, const int32_t* rows_dev,
                      int matrix_mode, void* ws, size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(matrix_mode_valid(matrix_mode), "matrix_mode must be GTG_MATRIX_F32, _BF16, _F16X2 or _BF16X3");
    MatrixModeScope matrix_scope(matrix_mode);
    GTG_REQUIRE(epilogue >= EPI_STORE && epilogue <= EPI_ACCUM, "unsupported epilogue");
    size_t tb, cb;
    gemm_tail_workspace(&tb, &cb);
    GTG_REQUIRE(ws != nullptr && ws_bytes >= tb + cb + GEMM_ABSMAX_BYTES, "workspace too small");
    GemmDesc g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.a_kc = a_kc != 0; g.b_kc = b_kc != 0; g.bias = bias; g.epi = epilogue; g.aux = aux; g.ldaux = ldaux;
    g.m_dev = rows_dev;
    g.tail_ws = reinterpret_cast<float*>(ws);
    g.tail_counters = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + tb);
    g.absmax_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + tb + cb);
    GTG_TRY(clear_i32(g.tail_counters, (int64_t)(cb / sizeof(int32_t)), (hipStream_t)stream));
    return gemm(g, (hipStream_t)stream);
}

// The three-plane split and the GEMM on operands that arrive split (tests and profiling).
int gtg_split_planes(const float* src, int64_t rows, int64_t cols, int64_t ld, int transpose, void* planes, void* stream)
{
    GTG_REQUIRE(rows >= 0 && cols >= 0 && ld >= cols, "bad size");
    GTG_REQUIRE((transpose ? rows : cols) % 16 == 0, "the operand's depth must be a multiple of 16");
    GTG_REQUIRE(rows * cols == 0 || (src && planes), "null pointer");
    return split_planes(src, rows, cols, ld, transpose != 0, planes, (hipStream_t)stream);
}

int gtg_gemm_planes_rows(const void* A_planes, const void* B_planes, float* C, int M, int N, int K, int ldc,
                         const float* bias, int epilogue, const float* aux, int ldaux, const int32_t* rows_dev, void* ws,
                         size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(epilogue >= EPI_STORE && epilogue <= EPI_ACCUM, "unsupported epilogue");
    size_t tb, cb;
    gemm_tail_workspace(&tb, &cb);
    GTG_REQUIRE(ws != nullptr && ws_bytes >= tb + cb, "workspace too small");
    GemmDesc g;
    g.A = reinterpret_cast<const float*>(A_planes); g.B = reinterpret_cast<const float*>(B_planes); g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = M; g.ldb = N; g.ldc = ldc;
    g.planes = true; g.planeA = (long long)M * K; g.planeB = (long long)N * K;
    g.bias = bias; g.epi = epilogue; g.aux = aux; g.ldaux = ldaux;
    g.m_dev = rows_dev;
    g.tail_ws = reinterpret_cast<float*>(ws);
    g.tail_counters = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + tb);
    GTG_TRY(clear_i32(g.tail_counters, (int64_t)(cb / sizeof(int32_t)), (hipStream_t)stream));
    return gemm(g, (hipStream_t)stream);
}

// ---- vocabulary projection + token cross-entropy (text decoder) ---------------
static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

size_t gtg_proj_ce_fwd_workspace_bytes(int64_t rows, int k, int vocab)
{
    (void)rows; (void)k; (void)vocab;
    size_t tb, cb;
    gemm_tail_workspace(&tb, &cb);
    return align256(tb) + align256(cb) + align256(GEMM_ABSMAX_BYTES);
}

int gtg_proj_ce_fwd(const float* H, const float* W, const float* bias, const int64_t* targets, int64_t rows, int k,
                    int vocab, int64_t padidx, float* loss, int32_t* pred, int32_t* rowmap, float* Hc, float* P,
                    int64_t ldp, int matrix_mode, void* ws, size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(matrix_mode_valid(matrix_mode), "matrix_mode must be GTG_MATRIX_F32, _BF16, _F16X2 or _BF16X3");
    MatrixModeScope matrix_scope(matrix_mode);
    GTG_REQUIRE(rows >= 0 && k > 0 && vocab > 0, "bad size");
    GTG_REQUIRE(rows <= (1 << 30) / 2, "too many rows");
    if (rows == 0) return GTG_OK;
    GTG_REQUIRE(H && W && targets && loss && pred && rowmap && Hc && P, "null pointer");
    GTG_REQUIRE(ldp >= vocab && (ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(P) & 15) == 0,
                "P needs 16-byte alignment and a row stride that is a multiple of 4 and >= vocab");
    GTG_REQUIRE(ws != nullptr && ws_bytes >= gtg_proj_ce_fwd_workspace_bytes(rows, k, vocab), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    size_t tb, cb;
    gemm_tail_workspace(&tb, &cb);
    float* tail_ws = reinterpret_cast<float*>(ws);
    int32_t* tail_counters = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(ws) + align256(tb));
    float* absmax = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + align256(tb) + align256(cb));

    GTG_TRY(ce_rowmap(targets, rows, padidx, rowmap, loss, pred, s));
    GTG_TRY(ce_rows(0, H, nullptr, rowmap, rows, k, Hc, s));
    GTG_TRY(clear_i32(tail_counters, (int64_t)(cb / sizeof(int32_t)), s));
    GemmDesc g;  // P[:count] = Hc W^T + bias
    g.A = Hc; g.lda = k; g.a_kc = true;
    g.B = W; g.ldb = k; g.b_kc = true;
    g.C = P; g.ldc = (int)ldp;
    g.M = (int)rows; g.N = vocab; g.K = k;
    g.bias = bias;
    g.m_dev = rowmap;
    g.tail_ws = tail_ws; g.tail_counters = tail_counters;
    g.absmax_ws = absmax;
    GTG_TRY(gemm(g, s));
    return ce_softmax(P, ldp, vocab, targets, rowmap, rows, loss, pred, s);
}

// The input-gradient GEMM (kept rows x k, depth = vocab) has few output tiles and a very long K: it is
// cut along K so that the kept rows alone (often half of all rows) still fill the chip a few times over.
static int proj_ce_dh_splits(int64_t rows, int k)
{
    const long long t128 = ((rows + 127) / 128) * ((k + 127) / 128);
    long long sp = (2048 + t128 - 1) / t128;
    return (int)(sp < 1 ? 1 : sp > 8 ? 8 : sp);
}

size_t gtg_proj_ce_bwd_workspace_bytes(int64_t rows, int k, int vocab)
{
    size_t tb, cb;
    gemm_tail_workspace(&tb, &cb);
    const int sp = proj_ce_dh_splits(rows, k);
    return align256(tb) + align256(cb) + align256(GEMM_ABSMAX_BYTES) + 2 * align256((size_t)rows * k * sizeof(float)) +
           align256(ce_bias_grad_workspace(vocab)) + (sp > 1 ? align256((size_t)sp * rows * k * sizeof(float)) : 0);
}

int gtg_proj_ce_bwd(const float* W, const float* gloss, const int32_t* rowmap, const float* Hc, const float* P,
                    int64_t ldp, int64_t rows, int k, int vocab, float* dH, float* dW, float* db, int matrix_mode,
                    void* ws, size_t ws_bytes, void* stream)
{
    GTG_REQUIRE(matrix_mode_valid(matrix_mode), "matrix_mode must be GTG_MATRIX_F32, _BF16, _F16X2 or _BF16X3");
    MatrixModeScope matrix_scope(matrix_mode);
    GTG_REQUIRE(rows >= 0 && k > 0 && vocab > 0, "bad size");
    GTG_REQUIRE(rows <= (1 << 30) / 2, "too many rows");
    GTG_REQUIRE(W && dW, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (rows == 0) {
        GTG_TRY(clear_i32(reinterpret_cast<int32_t*>(dW), (int64_t)vocab * k, s));
        if (db) GTG_TRY(clear_i32(reinterpret_cast<int32_t*>(db), vocab, s));
        return GTG_OK;
    }
    GTG_REQUIRE(gloss && rowmap && Hc && P, "null pointer");
    GTG_REQUIRE(ldp >= vocab && (ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(P) & 15) == 0,
                "P needs 16-byte alignment and a row stride that is a multiple of 4 and >= vocab");
    GTG_REQUIRE(ws != nullptr && ws_bytes >= gtg_proj_ce_bwd_workspace_bytes(rows, k, vocab), "workspace too small");
    size_t tb, cb;
    gemm_tail_workspace(&tb, &cb);
    char* w = reinterpret_cast<char*>(ws);
    float* tail_ws = reinterpret_cast<float*>(w); w += align256(tb);
    int32_t* tail_counters = reinterpret_cast<int32_t*>(w); w += align256(cb);
    float* absmax = reinterpret_cast<float*>(w); w += align256(GEMM_ABSMAX_BYTES);
    float* Hs = reinterpret_cast<float*>(w); w += align256((size_t)rows * k * sizeof(float));
    float* dHc = reinterpret_cast<float*>(w); w += align256((size_t)rows * k * sizeof(float));
    float* db_partial = reinterpret_cast<float*>(w); w += align256(ce_bias_grad_workspace(vocab));
    float* dh_slabs = reinterpret_cast<float*>(w);
    const int* count = rowmap;

    // dW (vocab, k) = P^T diag(g) Hc over the kept rows (K extent on the device)
    GTG_TRY(ce_rows(1, Hc, gloss, rowmap, rows, k, Hs, s));
    {
        GemmDesc g;
        g.A = P; g.lda = (int)ldp; g.a_kc = false;
        g.B = Hs; g.ldb = k; g.b_kc = false;
        g.C = dW; g.ldc = k;
        g.M = vocab; g.N = k; g.K = (int)rows;
        g.k_dev = count;
        g.absmax_ws = absmax;
        g.a_bound = 1.f;  // probabilities minus one-hot
        GTG_TRY(gemm(g, s));
    }
    if (db) GTG_TRY(ce_bias_grad(P, ldp, vocab, gloss, rowmap, rows, db_partial, db, s));
    // dH[r] = g[r] * (P W)[map[r]], pad rows zero
    if (dH) {
        GTG_TRY(clear_i32(tail_counters, (int64_t)(cb / sizeof(int32_t)), s));
        GemmDesc g;  // dHc[:count] = P W
        g.A = P; g.lda = (int)ldp; g.a_kc = true;
        g.B = W; g.ldb = k; g.b_kc = false;
        g.C = dHc; g.ldc = k;
        g.M = (int)rows; g.N = k; g.K = vocab;
        g.m_dev = count;
        const int sp = proj_ce_dh_splits(rows, k);
        if (sp > 1) {
            g.splits = sp; g.slabs = dh_slabs;
        } else {
            g.tail_ws = tail_ws; g.tail_counters = tail_counters;
        }
        g.absmax_ws = absmax;
        g.a_bound = 1.f;
        GTG_TRY(gemm(g, s));
        GTG_TRY(ce_rows(2, dHc, gloss, rowmap, rows, k, dH, s));
    }
    return GTG_OK;
}

}  // extern "C"
