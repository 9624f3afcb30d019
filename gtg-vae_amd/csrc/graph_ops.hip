// HBM-bound kernels of the graph-VAE step: adjacency build from the ragged
// edge list, max-pool backward routing, column sums for bias gradients, the
// reparameterisation + KL block, the un-masked upper-triangular BCE and its
// backward, fused Adam, and edge precision/recall.
//
// Every reduction here is ordered (per-block tree in LDS, then a fixed-order
// pass over the partials): the same inputs give bit-identical outputs, and no
// float atomics are used.
#include <mutex>

#include "gtg_internal.h"

namespace gtg {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Sum over a 256-thread block; result valid in thread 0.
__device__ __forceinline__ float block_sum_256(float v, float* sm /* >= 4 floats */)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------
// adjacency: counts[b][i][j] = #edges (src=j -> dst=i), self loops dropped.
// Integer atomics: order-independent, exact.
// ---------------------------------------------------------------------------
// `counts` is all zero on entry (the caller zeroes it once, when it allocates it) and all zero again when
// adj_normalize_kernel, its last reader, is done: no clearing launch per step.
__global__ __launch_bounds__(256) void adj_count_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t B, int n,
                                                        int32_t* __restrict__ counts, int32_t* __restrict__ err)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t s = ei[e], t = ei[E + e];
    if (s == t) return;
    const int64_t total = B * n;
    if (s < 0 || t < 0 || s >= total || t >= total || s / n != t / n) {
        atomicOr(err, 1);  // edge leaves its graph: not a block-diagonal batch
        return;
    }
    const int64_t g = t / n;
    atomicAdd(&counts[g * n * n + (t % n) * n + (s % n)], 1);
}

// One launch that brings a batch into the step's static buffers: the (2, E) edge list into a (2, E_cap) buffer
// whose tail is filled with (0, 0) self loops (dropped by adj_count_kernel), the padded targets and eps.
// Work item = one int64 of the edge buffer, or one 16-byte piece (or scalar tail) of the float buffers.
__global__ __launch_bounds__(256) void stage_batch_kernel(const int64_t* __restrict__ ei_src, int64_t E,
                                                          int64_t* __restrict__ ei_dst, int64_t E_cap,
                                                          const float* __restrict__ a_src, float* __restrict__ a_dst,
                                                          int64_t a_count, const float* __restrict__ e_src,
                                                          float* __restrict__ e_dst, int64_t e_count, int vec)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * E_cap) {
        const int64_t row = i / E_cap, e = i - row * E_cap;
        ei_dst[i] = e < E ? ei_src[row * E + e] : 0;
        return;
    }
    i -= 2 * E_cap;
    const int64_t a_items = (a_count + 3) / 4, e_items = (e_count + 3) / 4;
    const float* src;
    float* dst;
    int64_t count;
    if (i < a_items) {
        src = a_src; dst = a_dst; count = a_count;
    } else if (i - a_items < e_items) {
        i -= a_items;
        src = e_src; dst = e_dst; count = e_count;
    } else {
        return;
    }
    if (vec && 4 * i + 3 < count) {
        *reinterpret_cast<f32x4*>(dst + 4 * i) = *reinterpret_cast<const f32x4*>(src + 4 * i);
    } else {
        for (int64_t j = 4 * i; j < count && j < 4 * i + 4; ++j) dst[j] = src[j];
    }
}

// ---------------------------------------------------------------------------
// Batches gathered from a dataset that lives in HBM.  The dataset keeps every
// graph's edge list with LOCAL node ids, concatenated (2, E_all) with (G+1)
// start offsets, and the padded targets (G, n, n).  A batch is a list of B
// graph ids: kernel 1 (one block) scans the selected graphs' edge counts into
// B+1 output offsets; kernel 2 (one block per slot, plus blocks for the tail)
// writes the (2, E_cap) edge buffer with ids shifted to slot * n and the tail
// filled with (0,0) self loops, and copies the targets.  No host involvement
// beyond the id list, so the launches depend on nothing the host must read back.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_offsets_kernel(const int64_t* __restrict__ estart,
                                                             const int64_t* __restrict__ idx, int B, int64_t G,
                                                             int64_t E_cap, int64_t* __restrict__ off,
                                                             int32_t* __restrict__ err)
{
    __shared__ int64_t part[256];
    const int t = threadIdx.x;
    const int per = (B + 255) / 256;
    const int lo = t * per, hi = min(B, lo + per);
    int64_t s = 0;
    bool bad = false;
    for (int b = lo; b < hi; ++b) {
        const int64_t g = idx[b];
        if (g < 0 || g >= G)
            bad = true;
        else
            s += estart[g + 1] - estart[g];
    }
    part[t] = s;
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if (t == 0) {
        int64_t run = 0;
        for (int i = 0; i < 256; ++i) {
            const int64_t v = part[i];
            part[i] = run;
            run += v;
        }
        off[B] = run;
        // this launch owns the word: 2 = the batch's edges do not fit the edge buffer, 4 = an id outside the dataset
        *err = (run > E_cap ? 2 : 0) | (any_bad ? 4 : 0);
    }
    __syncthreads();
    int64_t run = part[t];
    for (int b = lo; b < hi; ++b) {
        off[b] = run;
        const int64_t g = idx[b];
        if (g >= 0 && g < G) run += estart[g + 1] - estart[g];
    }
}

__global__ __launch_bounds__(256) void gather_batch_kernel(const int64_t* __restrict__ esrc, int64_t E_all,
                                                           const int64_t* __restrict__ estart,
                                                           const float* __restrict__ adj_all,
                                                           const int64_t* __restrict__ idx, int B, int64_t G, int n,
                                                           const int64_t* __restrict__ off, int64_t* __restrict__ ei_dst,
                                                           int64_t E_cap, float* __restrict__ target_dst)
{
    const int b = blockIdx.x;
    if (b >= B) {
        // tail of the edge buffer: (0, 0) self loops
        const int64_t total = min(off[B], E_cap);
        const int64_t tail_blocks = gridDim.x - B;
        for (int64_t e = total + (int64_t)(b - B) * 256 + threadIdx.x; e < E_cap; e += tail_blocks * 256) {
            ei_dst[e] = 0;
            ei_dst[E_cap + e] = 0;
        }
        return;
    }
    const int64_t g = idx[b];
    float* tdst = target_dst + (int64_t)b * n * n;
    if (g < 0 || g >= G) {  // flagged by the offsets kernel: leave an empty graph
        for (int i = threadIdx.x; i < n * n; i += 256) tdst[i] = 0.f;
        return;
    }
    const int64_t e0 = estart[g], cnt = estart[g + 1] - e0, o = off[b];
    const int64_t shift = (int64_t)b * n;
    for (int64_t e = threadIdx.x; e < cnt; e += 256) {
        if (o + e < E_cap) {
            ei_dst[o + e] = esrc[e0 + e] + shift;
            ei_dst[E_cap + o + e] = esrc[E_all + e0 + e] + shift;
        }
    }
    const float* tsrc = adj_all + g * n * n;
    for (int i = threadIdx.x; i < n * n; i += 256) tdst[i] = tsrc[i];
}

__global__ __launch_bounds__(256) void clear_i32_kernel(int32_t* __restrict__ p, int64_t count)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = 0;
}

// One block per graph (n <= 64): deg_i = 1 + sum_j counts[i][j];
// adj[i][j] = deg_i^-1/2 * (counts[i][j] + [i==j]) * deg_j^-1/2.
// On the way: extent[b] = 1 + the highest node index a (non-self) edge touches, 0 for a graph without
// edges -- the number of encoder rows the compact-row layout stores for this graph -- and the graph's
// counts are zeroed for the next batch (every element is read by the lane that then clears it).
__global__ __launch_bounds__(256) void adj_normalize_kernel(int32_t* __restrict__ counts, int n,
                                                            float* __restrict__ adj, int32_t* __restrict__ extent)
{
    __shared__ int cs[64 * 65];  // the graph's counts, row stride 65: rows and columns both read conflict-free
    __shared__ float dis[64];
    const int64_t b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t* c = counts + b * n * n;
    float* a = adj + b * n * n;
    const int lane_c = min(lane, n - 1);
    // four waves take the rows in turn: one coalesced read per row, then the row is cleared for the next batch
    // (unconditional loads at clamped indices; the rows' loads are all in flight before the first LDS write)
    int v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = c[min(wave + 4 * t, n - 1) * n + lane_c];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int i = wave + 4 * t;
        if (i < n && lane < n) {
            cs[i * 65 + lane] = v[t];
            c[i * n + lane] = 0;
        }
    }
    __syncthreads();
    if (wave == 0) {
        int deg = 1, col = 0;
        if (lane < n)
            for (int j = 0; j < n; ++j) {
                deg += cs[lane * 65 + j];  // row `lane`
                col |= cs[j * 65 + lane];  // column `lane`
            }
        dis[lane] = (lane < n) ? 1.0f / sqrtf((float)deg) : 0.f;
        const unsigned long long m = __ballot(deg > 1 || col != 0);
        if (extent != nullptr && lane == 0) extent[b] = m ? 64 - __clzll((long long)m) : 0;
    }
    __syncthreads();
    if (lane < n) {
        const float dj = dis[lane];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int i = wave + 4 * t;
            if (i < n) a[i * n + lane] = (dis[i] * (float)(cs[i * 65 + lane] + (i == lane ? 1 : 0))) * dj;
        }
    }
}

// ---------------------------------------------------------------------------
// max-pool backward routed through the second propagate:
// dT2[b,j,c] = adj[b, argmax[b,c], j] * dpool[b,c]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_bwd_gather_kernel(const float* __restrict__ adj,
                                                              const float* __restrict__ dpool,
                                                              const int32_t* __restrict__ argmax, int n, int d,
                                                              float* __restrict__ dT2)
{
    const int64_t b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    const float g = dpool[b * d + c];
    const int am = argmax[b * d + c];
    const float* arow = adj + b * n * n + (int64_t)am * n;
    float* out = dT2 + b * n * d + c;
    for (int j = 0; j < n; ++j) out[(int64_t)j * d] = arow[j] * g;
}

// ---------------------------------------------------------------------------
// column sums (bias gradients).  A block owns 64 columns (16 lanes x 4
// columns, 16-byte loads when the rows are aligned) and one chunk of rows,
// walked by 16 row lanes; the row lanes are combined through LDS in a fixed
// order.  With a single chunk the result goes straight to `out`, otherwise to
// partial[chunk][col] and a second launch adds the chunks in order.
// ---------------------------------------------------------------------------
constexpr int CS_COLS = 64;
__host__ __device__ inline int cs_chunk_rows(int64_t rows)
{
    return rows <= 1024 ? (int)(rows > 0 ? rows : 1) : 512;
}

// CS_RL row lanes: 16 (256 threads) for the chunks of a tall matrix, 64 (1024 threads) when the whole matrix is
// one short chunk and the few blocks of the launch need every load in flight at once
template <int CS_RL>
__global__ __launch_bounds__(16 * CS_RL) void colsum_kernel(const float* __restrict__ X, int64_t rows, int cols, int ld,
                                                            int chunk, int vec, const int* __restrict__ rows_dev,
                                                            float* __restrict__ dst)
{
    __shared__ f32x4 sm[CS_RL][CS_COLS / 4];
    if (rows_dev) rows = min(rows, (int64_t)rows_dev[0]);
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.x * CS_COLS + cg * 4;
    const int64_t r0 = (int64_t)blockIdx.y * chunk;
    const int64_t r1 = min(rows, r0 + chunk);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (col < cols) {
        if (vec && col + 3 < cols) {
#pragma unroll 8
            for (int64_t r = r0 + rl; r < r1; r += CS_RL) s += *reinterpret_cast<const f32x4*>(X + r * ld + col);
        } else {
            for (int64_t r = r0 + rl; r < r1; r += CS_RL)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (col + q < cols) s[q] += X[r * ld + col + q];
        }
    }
    sm[rl][cg] = s;
    __syncthreads();
    if (rl == 0 && col < cols) {
        f32x4 t = sm[0][cg];
#pragma unroll
        for (int k = 1; k < CS_RL; ++k) t += sm[k][cg];
        float* o = dst + (int64_t)blockIdx.y * cols + col;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (col + q < cols) o[q] = t[q];
    }
}

// Two short column sums in one launch (blockIdx.z picks the job): bias gradients that become available at
// different points of a backward pass but are needed only at its end share a kernel boundary.
struct ColsumJob {
    const float* X;
    float* out;
    int rows, cols, ld, vec;
    const int* rows_dev;  // optional: rows = min(rows, mult * ceil(*rows_dev / unit))
    int unit, mult;
};

__global__ __launch_bounds__(1024) void colsum_pair_kernel(const ColsumJob j0, const ColsumJob j1)
{
    constexpr int RL = 64;
    __shared__ f32x4 sm[RL][CS_COLS / 4];
    ColsumJob j = blockIdx.z == 0 ? j0 : j1;
    if (j.rows_dev) j.rows = min(j.rows, j.mult * ((j.rows_dev[0] + j.unit - 1) / j.unit));
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.x * CS_COLS + cg * 4;
    if (blockIdx.x * CS_COLS >= j.cols) return;  // block-uniform: this job is narrower than the grid
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (col < j.cols) {
        if (j.vec && col + 3 < j.cols) {
#pragma unroll 8
            for (int r = rl; r < j.rows; r += RL) s += *reinterpret_cast<const f32x4*>(j.X + (long long)r * j.ld + col);
        } else {
            for (int r = rl; r < j.rows; r += RL)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (col + q < j.cols) s[q] += j.X[(long long)r * j.ld + col + q];
        }
    }
    sm[rl][cg] = s;
    __syncthreads();
    if (rl == 0 && col < j.cols) {
        f32x4 t = sm[0][cg];
#pragma unroll
        for (int k = 1; k < RL; ++k) t += sm[k][cg];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (col + q < j.cols) j.out[col + q] = t[q];
    }
}

// ---------------------------------------------------------------------------
// z = eps * exp(0.5 logvar) + mu ;  KL = -0.5 * sum(1 + logvar - mu^2 - e^logvar) / B
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reparam_kl_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                         const float* __restrict__ eps, int64_t total,
                                                         float* __restrict__ z, float* __restrict__ partial)
{
    __shared__ float sm[4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float term = 0.f;
    if (i < total) {
        const float m = mu[i], l = lv[i];
        z[i] = eps[i] * expf(0.5f * l) + m;
        term = 1.f + l - m * m - expf(l);
    }
    const float s = block_sum_256(term, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// out[0] = scale * sum(partial[0..n)), summed by one block in a fixed order.
__global__ __launch_bounds__(256) void final_sum_kernel(const float* __restrict__ partial, int64_t n, float scale,
                                                        float* __restrict__ out)
{
    __shared__ float sm[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += partial[i];
    const float t = block_sum_256(s, sm);
    if (threadIdx.x == 0) out[0] = t * scale;
}

__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ mu,
                                                         const float* __restrict__ lv, const float* __restrict__ eps,
                                                         const float* __restrict__ kl_w_dev, float kl_w, float inv_b,
                                                         int64_t total, int d, int ld, float* __restrict__ dmu,
                                                         float* __restrict__ dlv)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float w = (kl_w_dev ? kl_w_dev[0] * kl_w : kl_w) * inv_b;
    const float m = mu[i], l = lv[i], g = dz ? dz[i] : 0.f;
    const int64_t o = (i / d) * ld + (i % d);  // the outputs may be the two halves of (B, 2d) rows
    float gm, gl;
    latent_bwd_point(g, m, l, eps[i], w, gm, gl);
    dmu[o] = gm;
    dlv[o] = gl;
}

// ---------------------------------------------------------------------------
// un-masked BCE over the padded upper triangle (diagonal included).
// Element k = i*n - i(i-1)/2 + (j-i), i <= j, of row b of `out` pairs with
// adj[b,i,j].  log terms clamp at -100 (binary_cross_entropy semantics).
// ---------------------------------------------------------------------------
// (dlogits, when given, receives the gradient of the mean loss w.r.t. the pre-sigmoid logits in the same
// pass: gl / (B m) with gl as in bce_bwd_kernel below.)
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ out, const float* __restrict__ adj,
                                                      int n, int m, float* __restrict__ row_partial, float gscale,
                                                      const float* __restrict__ gscale_dev,
                                                      float* __restrict__ dlogits)
{
    __shared__ float sm[4];
    if (gscale_dev != nullptr) gscale *= gscale_dev[0];
    const int64_t b = blockIdx.x;
    const float* o = out + b * m;
    const float* a = adj + b * n * n;
    float s = 0.f;
    // Threads walk the triangle's own index k (coalesced in `out` and in the gradient); (i, j) comes from
    // inverting k = i n - i(i-1)/2 + (j - i).  Four elements per round, their loads issued together with
    // clamped indices (a guarded load would be a branch with a full wait behind it).
    const float tn1 = (float)(2 * n + 1);
    for (int k0 = threadIdx.x; k0 < m; k0 += 4 * 256) {
        float p[4], t[4];
        int kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = min(k0 + u * 256, m - 1);
            int i = (int)((tn1 - sqrtf(tn1 * tn1 - 8.f * (float)k)) * 0.5f);
            i = max(0, min(i, n - 1));
            if (i * n - (i * (i - 1)) / 2 > k) --i;                    // float rounding: at most one row off
            else if (i + 1 < n && (i + 1) * n - ((i + 1) * i) / 2 <= k) ++i;
            const int j = i + (k - (i * n - (i * (i - 1)) / 2));
            kk[u] = k;
            p[u] = o[k];
            t[u] = a[i * n + j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 + u * 256 >= m) break;
            const float lp = fmaxf(logf(p[u]), -100.f);
            const float lq = fmaxf(logf(1.f - p[u]), -100.f);
            s += (t[u] - 1.f) * lq - t[u] * lp;
            if (dlogits != nullptr) {
                const float q = (1.f - p[u]) * p[u];
                dlogits[b * m + kk[u]] = (gscale * (p[u] - t[u]) / fmaxf(q, 1e-12f)) * q;
            }
        }
    }
    const float tot = block_sum_256(s, sm);
    if (threadIdx.x == 0) row_partial[b] = tot;
}

// dy2 = [g (p - t) / max((1-p) p, 1e-12)] * (1-p) p : BCE backward followed by
// the sigmoid's backward, in the order autograd applies them.
__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ out, const float* __restrict__ adj,
                                                      int n, int m, const float* __restrict__ gdev, float g,
                                                      int through_sigmoid, float* __restrict__ dy2)
{
    const int64_t b = blockIdx.x;
    const float* o = out + b * m;
    const float* a = adj + b * n * n;
    float* d = dy2 + b * m;
    const float gg = gdev ? gdev[0] * g : g;
    for (int idx = threadIdx.x; idx < n * n; idx += 256) {
        const int i = idx / n, j = idx - i * n;
        if (j < i) continue;
        const int k = i * n - (i * (i - 1)) / 2 + (j - i);
        const float p = o[k], t = a[idx];
        const float q = (1.f - p) * p;
        const float gi = gg * (p - t) / fmaxf(q, 1e-12f);
        d[k] = through_sigmoid ? gi * q : gi;
    }
}

// ---------------------------------------------------------------------------
// Compact node rows.  With identity node features, a node that no non-self
// edge touches has the same encoder rows in every graph (its adjacency row and
// column are the unit vector), so only rows [0, n_b) of graph b are stored and
// multiplied, n_b = 1 + the highest node index an edge touches; the rows past
// it are represented once per step by "prototype" rows.
// meta layout (int32): [0,B) n_b ; [B, 2B] exclusive row offsets, meta[2B] = R ;
// meta[2B+1] = R + n (the n prototype rows are stored right after the real rows, as
// n extra nodes whose adjacency row is the unit vector, so every row-wise GEMM and
// reduction of forward and backward covers them with no special case)
// ---------------------------------------------------------------------------
// Block b < B: row offset of graph b = sum of the extents before it (every block adds them up itself: B
// small integers, cheaper than a scan kernel in front of this one), meta[b] / meta[B + b], and
// adjc[off + j, :] = adj[b, j, :] for j < n_b.  Block B: the totals meta[2B], meta[2B + 1] and the n unit rows
// at R.  All blocks together clear `zero_count` int32 words at `zero`, the arrival counters of the GEMMs that
// follow (one launch less than a separate clear).
__global__ __launch_bounds__(256) void adj_compact_kernel(const float* __restrict__ adj, const int32_t* __restrict__ extent,
                                                         int32_t* __restrict__ meta, int B, int n,
                                                         float* __restrict__ adjc, int32_t* __restrict__ zero,
                                                         int zero_count)
{
    __shared__ int part[4];
    const int b = blockIdx.x;
    for (int i = b * 256 + threadIdx.x; i < zero_count; i += gridDim.x * 256) zero[i] = 0;
    int s = 0;
    for (int g = threadIdx.x; g < b; g += 256) s += extent[g];  // b == B: all of them
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    const int off = part[0] + part[1] + part[2] + part[3];
    if (b == B) {
        if (threadIdx.x == 0) {
            meta[2 * B] = off;          // R: rows of real nodes
            meta[2 * B + 1] = off + n;  // R + n: the n prototype rows follow them
        }
        for (int idx = threadIdx.x; idx < n * n; idx += 256)
            adjc[(int64_t)off * n + idx] = (idx / n == idx % n) ? 1.f : 0.f;
        return;
    }
    const int nb = extent[b];
    if (threadIdx.x == 0) {
        meta[b] = nb;
        meta[B + b] = off;
    }
    const float* src = adj + (int64_t)b * n * n;
    float* dst = adjc + (int64_t)off * n;
    for (int idx = threadIdx.x; idx < nb * n; idx += 256) dst[idx] = src[idx];
}

// gradient reaching the prototype rows: dQ[i][c] = sum over graphs whose pooled
// maximum in column c is row i >= n_b.  Stage 1: a wave (64 columns, one segment
// of the batch) walks its graphs in order, each thread adding into its own
// column of an LDS (n x 64) table; stage 2 adds the segments in order and
// writes the rows at offset R of dT2c.  Ordered: reproducible.  Stage 1 rides in
// the launch of pool_bwd_gather_compact_kernel (its first blocks): one kernel boundary less.
constexpr int PG_SEG = 32;
__device__ __forceinline__ void proto_grad_stage1_block(float* __restrict__ tab, int cb, int seg,
                                                        const float* __restrict__ dpool,
                                                        const int32_t* __restrict__ argmax,
                                                        const int32_t* __restrict__ meta, int64_t B, int n, int d,
                                                        float* __restrict__ partial)
{
    const int tc = threadIdx.x;
    if (tc >= 64) return;
    const int c = cb * 64 + tc;
    for (int i = 0; i < n; ++i) tab[i * 64 + tc] = 0.f;
    const int64_t per = (B + PG_SEG - 1) / PG_SEG;
    const int64_t b0 = seg * per, b1 = min(B, b0 + per);
    if (c < d) {
        // loads of eight graphs in flight, then the ordered adds into this thread's own LDS column
        for (int64_t b = b0; b < b1; b += 8) {
            int am[8], nb[8];
            float g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t bb = min(b + u, b1 - 1);
                am[u] = argmax[bb * d + c];
                nb[u] = meta[bb];
                g[u] = dpool[bb * d + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (b + u < b1 && am[u] >= nb[u]) tab[am[u] * 64 + tc] += g[u];
        }
        for (int i = 0; i < n; ++i) partial[((int64_t)seg * n + i) * d + c] = tab[i * 64 + tc];
    }
}

// max-pool backward for compact rows: dT2c[off_b + j, c] = adj[b, am, j] * dpool[b,c], j < n_b
// (zero when the winning row am is a prototype row: it has no edge to a stored row)
__global__ __launch_bounds__(256) void pool_bwd_gather_compact_kernel(const float* __restrict__ adj,
                                                                      const float* __restrict__ dpool,
                                                                      const int32_t* __restrict__ argmax,
                                                                      const int32_t* __restrict__ meta, int64_t B,
                                                                      int n, int d, float* __restrict__ dT2,
                                                                      int32_t* __restrict__ zero, int zero_count,
                                                                      float* __restrict__ pg_partial, int pg_rows)
{
    // the graph's adjacency block sits in LDS, one padded row per winning-row lookup
    __shared__ float sadj[64 * 65];
    // the first pg_rows rows of the grid are the prototype-gradient stage-1 blocks (they run longest: first out)
    if ((int)blockIdx.y < pg_rows) {
        const int q = blockIdx.y * gridDim.x + blockIdx.x, ncb = (d + 63) / 64;
        if (q < ncb * PG_SEG) proto_grad_stage1_block(sadj, q % ncb, q / ncb, dpool, argmax, meta, B, n, d, pg_partial);
        return;
    }
    const int64_t b = blockIdx.y - pg_rows;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    // also clears the arrival counters of the GEMM that follows (one launch less than a separate clear)
    for (int i = ((blockIdx.y - pg_rows) * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < zero_count;
         i += gridDim.x * (gridDim.y - pg_rows) * 256)
        zero[i] = 0;
    const int nb = meta[b], off = meta[B + b];
    const float* src = adj + b * n * n;
    for (int idx = threadIdx.x; idx < nb * n; idx += 256) sadj[(idx / n) * 65 + (idx % n)] = src[idx];
    __syncthreads();
    if (c >= d) return;
    const float g = dpool[b * d + c];
    const int am = argmax[b * d + c];
    // a prototype row that won the maximum has no edge to a stored row
    const float scale = (am < nb) ? g : 0.f;
    const float* arow = sadj + (am < nb ? am : 0) * 65;
    float* out = dT2 + (int64_t)off * d + c;
#pragma unroll 4
    for (int j = 0; j < nb; ++j) out[(int64_t)j * d] = arow[j] * scale;
}

__global__ __launch_bounds__(256) void proto_grad_stage2(const float* __restrict__ partial,
                                                         const int32_t* __restrict__ row_base, int n, int d,
                                                         float* __restrict__ dT2c)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * d) return;
    float s = 0.f;
    for (int k = 0; k < PG_SEG; ++k) s += partial[(int64_t)k * n * d + idx];
    dT2c[(int64_t)row_base[0] * d + idx] = s;
}

// ---- the same gradient rows as three bf16 planes -------------------------------------------------------
// dT2c is consumed by one GEMM, dZ1c = (dT2c W2) masked by relu.  In the GTG_MATRIX_BF16X3 mode that GEMM takes
// its operands as exact three-way bf16 splits; written that way here, in the GEMM's k-blocked operand layout
// planes[pl][c / 16][row][c % 16] (`rows_alloc` rows per k block), the split costs one pass in the producer
// instead of one per output tile in the consumer.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// x[0..7] -> the eight bf16 of plane 0, 1, 2 (truncation split: each plane the top 16 bits of what is left)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4_t& p0, u32x4_t& p1, u32x4_t& p2)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a0 = x[2 * q], b0 = x[2 * q + 1];
        p0[q] = __builtin_amdgcn_perm(__float_as_uint(b0), __float_as_uint(a0), 0x07060302u);
        const float a1 = a0 - __uint_as_float(__float_as_uint(a0) & 0xffff0000u);
        const float b1 = b0 - __uint_as_float(__float_as_uint(b0) & 0xffff0000u);
        p1[q] = __builtin_amdgcn_perm(__float_as_uint(b1), __float_as_uint(a1), 0x07060302u);
        const float a2 = a1 - __uint_as_float(__float_as_uint(a1) & 0xffff0000u);
        const float b2 = b1 - __uint_as_float(__float_as_uint(b1) & 0xffff0000u);
        p2[q] = __builtin_amdgcn_perm(__float_as_uint(b2), __float_as_uint(a2), 0x07060302u);
    }
}

// x[0..7] * scale -> the eight fp16 of the value rounded to nearest and of what is left, rounded to nearest
__device__ __forceinline__ void split8h(const float (&x)[8], float scale, u32x4_t& p0, u32x4_t& p1)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned hi, lo;
        split_pair_f16(x[2 * q], x[2 * q + 1], scale, hi, lo);
        p0[q] = hi;
        p1[q] = lo;
    }
}

// maximum of a non-negative value over the block's 256 threads (sm: four floats of LDS; ends with a barrier)
__device__ __forceinline__ float block_max256(float v, float* sm)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();  // an earlier use of sm has been read
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

constexpr int GP_MAX_D = 1024;
// blocks [0, pg_blocks): stage 1 of the prototype rows' gradient; block pg_blocks + b: graph b's rows
// dT2c[off_b + j, c] = adj[b, am, j] * dpool[b, c], j < n_b.  A thread takes (row j, eight columns): 16-byte
// stores, consecutive threads on consecutive rows of one k block (contiguous in the k-blocked layout).
// F16: two fp16 planes instead, every row of the graph times one power of two (from the largest pooled gradient
// times the largest adjacency entry of the graph, a bound on the rows' entries); inv_rows[row] = its inverse.
template <bool F16>
__global__ __launch_bounds__(256) void pool_bwd_gather_planes_kernel(
    const float* __restrict__ adj, const float* __restrict__ dpool, const int32_t* __restrict__ argmax,
    const int32_t* __restrict__ meta, int B, int n, int d, long long rows_alloc, unsigned short* __restrict__ planes,
    int32_t* __restrict__ zero, int zero_count, float* __restrict__ pg_partial, int pg_blocks, float* __restrict__ inv_rows)
{
    __shared__ float sadj[64 * 65];
    __shared__ float smax[4];
    __shared__ float gS[GP_MAX_D];
    __shared__ int amS[GP_MAX_D];
    if ((int)blockIdx.x < pg_blocks) {
        const int q = blockIdx.x, ncb = (d + 63) / 64;
        if (q < ncb * PG_SEG) proto_grad_stage1_block(sadj, q % ncb, q / ncb, dpool, argmax, meta, B, n, d, pg_partial);
        return;
    }
    const int b = blockIdx.x - pg_blocks, tid = threadIdx.x;
    for (int i = b * 256 + tid; i < zero_count; i += B * 256) zero[i] = 0;   // the following GEMM's arrival counters
    const int nb = meta[b], off = meta[B + b];
    // Every global load of the block is issued here, back to back and unconditionally (clamped indices; an entry
    // read twice is stored twice with the same value): the graph's whole adjacency block, the winning rows and
    // the pooled gradient.  As loops with data-dependent bounds and a gradient load that waited for the winning
    // row they were four memory latencies one after the other.
    const float* src = adj + (int64_t)b * n * n;
    const int nn_last = n * n - 1;
    float av[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) av[q] = src[min(tid + 256 * q, nn_last)];
    int amv[GP_MAX_D / 256];
    float gv[GP_MAX_D / 256];
#pragma unroll
    for (int q = 0; q < GP_MAX_D / 256; ++q) {
        const int c = min(tid + 256 * q, d - 1);
        amv[q] = argmax[(int64_t)b * d + c];
        gv[q] = dpool[(int64_t)b * d + c];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int idx = min(tid + 256 * q, nn_last);
        sadj[(idx / n) * 65 + (idx % n)] = av[q];
    }
#pragma unroll
    for (int q = 0; q < GP_MAX_D / 256; ++q) {
        const int c = tid + 256 * q;         // entries from d on are written (a repeat of column d - 1) and never read
        const bool real = amv[q] < nb;        // a prototype row that won the maximum has no edge to a stored row
        gS[c] = real ? gv[q] : 0.f;
        amS[c] = real ? amv[q] : 0;
    }
    __syncthreads();
    float scale = 1.f;
    if constexpr (F16) {
        float mg = 0.f, ma = 0.f;
        for (int c = tid; c < d; c += 256) mg = fmaxf(mg, fabsf(gS[c]));
        for (int idx = tid; idx < nb * n; idx += 256) ma = fmaxf(ma, fabsf(sadj[(idx / n) * 65 + (idx % n)]));
        mg = block_max256(mg, smax);
        ma = block_max256(ma, smax);
        float inverse;
        f16x2_scale(mg * ma, scale, inverse);
        for (int j = tid; j < nb; j += 256) inv_rows[off + j] = inverse;
    }
    const int kblocks = d / 16;
    const long long plane = (long long)kblocks * rows_alloc * 16;
    const int items = kblocks * nb * 2;
    for (int idx = tid; idx < items; idx += 256) {
        const int half = idx & 1, rk = idx >> 1, j = rk % nb, kb = rk / nb;
        const int c0 = kb * 16 + half * 8;
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = sadj[amS[c0 + u] * 65 + j] * gS[c0 + u];
        unsigned short* dst = planes + ((long long)kb * rows_alloc + off + j) * 16 + half * 8;
        if constexpr (F16) {
            u32x4_t p0, p1;
            split8h(x, scale, p0, p1);
            *reinterpret_cast<u32x4_t*>(dst) = p0;
            *reinterpret_cast<u32x4_t*>(dst + plane) = p1;
        } else {
            u32x4_t p0, p1, p2;
            split8(x, p0, p1, p2);
            *reinterpret_cast<u32x4_t*>(dst) = p0;
            *reinterpret_cast<u32x4_t*>(dst + plane) = p1;
            *reinterpret_cast<u32x4_t*>(dst + 2 * plane) = p2;
        }
    }
}

// stage 2 of the prototype rows' gradient: the ordered sum of the segments, kept as fp32 (proto_rows, (n, d):
// the sparse dW2 kernel reads it) and written as planes at rows R .. R + n - 1.  A block sums 256 consecutive
// elements of the (n, d) table (coalesced, the 32 segment loads in flight together), then 32 of its threads
// split eight each.  Blocks from stage2_blocks on split the weight matrix the same GEMM multiplies by
// (w: (w_rows, w_cols) fp32, leading dimension w_ld, used transposed: operand rows = w_cols, depth = w_rows),
// one output element per thread, so that both operands are ready after one launch.
__global__ __launch_bounds__(256) void proto_grad_stage2_planes(const float* __restrict__ partial,
                                                                const int32_t* __restrict__ row_base, int n, int d,
                                                                long long rows_alloc, float* __restrict__ proto_rows,
                                                                unsigned short* __restrict__ planes, int stage2_blocks,
                                                                const float* __restrict__ w, int w_rows, int w_cols,
                                                                int w_ld, unsigned short* __restrict__ w_planes)
{
    __shared__ float sm[256];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= stage2_blocks) {
        const long long total = (long long)w_rows * w_cols;
        const long long i = (long long)(blockIdx.x - stage2_blocks) * 256 + tid;   // output element
        if (i >= total) return;
        const long long kb = i / ((long long)w_cols * 16), row = (i / 16) % w_cols, k = kb * 16 + (i & 15);
        const float x0 = w[k * w_ld + row];
        const unsigned b0 = __float_as_uint(x0) & 0xffff0000u;
        const float x1 = x0 - __uint_as_float(b0);
        const unsigned b1 = __float_as_uint(x1) & 0xffff0000u;
        const float x2 = x1 - __uint_as_float(b1);
        w_planes[i] = (unsigned short)(b0 >> 16);
        w_planes[total + i] = (unsigned short)(b1 >> 16);
        w_planes[2 * total + i] = (unsigned short)(__float_as_uint(x2) >> 16);
        return;
    }
    const int total = n * d, idx = blockIdx.x * 256 + tid;
    float s = 0.f;
    if (idx < total) {
        float v[PG_SEG];
#pragma unroll
        for (int k = 0; k < PG_SEG; ++k) v[k] = partial[(int64_t)k * total + idx];
#pragma unroll
        for (int k = 0; k < PG_SEG; ++k) s += v[k];
        proto_rows[idx] = s;
    }
    sm[tid] = s;
    __syncthreads();
    const int e0 = blockIdx.x * 256 + tid * 8;
    if (tid >= 32 || e0 >= total) return;
    const int i = e0 / d, c0 = e0 % d;   // d is a multiple of 16: the eight share a row and half a k block
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = sm[tid * 8 + u];
    u32x4_t p0, p1, p2;
    split8(x, p0, p1, p2);
    const long long plane = (long long)(d / 16) * rows_alloc * 16;
    unsigned short* dst = planes + ((long long)(c0 / 16) * rows_alloc + row_base[0] + i) * 16 + (c0 & 8);
    *reinterpret_cast<u32x4_t*>(dst) = p0;
    *reinterpret_cast<u32x4_t*>(dst + plane) = p1;
    *reinterpret_cast<u32x4_t*>(dst + 2 * plane) = p2;
}

// The fp16 form of the kernel above.  Block i < n: prototype row i -- the ordered sum of its segments (kept as fp32
// in proto_rows), its largest magnitude, and the row times the matching power of two as two fp16 planes at row
// R + i; inv_rows[R + i] = the inverse.  Blocks from n on take four rows each of the weight operand (operand row r =
// column r of w, depth = w_rows <= GP_MAX_D): the row's largest magnitude, then its planes; inv_w[r] = the row's
// inverse scale.
__global__ __launch_bounds__(256) void proto_grad_stage2_planes_f16(const float* __restrict__ partial,
                                                                    const int32_t* __restrict__ row_base, int n, int d,
                                                                    long long rows_alloc, float* __restrict__ proto_rows,
                                                                    unsigned short* __restrict__ planes,
                                                                    float* __restrict__ inv_rows, const float* __restrict__ w,
                                                                    int w_rows, int w_cols, int w_ld,
                                                                    unsigned short* __restrict__ w_planes,
                                                                    float* __restrict__ inv_w)
{
    __shared__ float sm[GP_MAX_D];
    __shared__ float smax[4];
    __shared__ unsigned rowmax[16];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= n) {
        // four operand rows per block, 64 threads per row, up to 16 depth values per thread held in registers
        // (w_rows <= GP_MAX_D): every load is in flight before the first is used
        const int r = tid & 3, kq = tid >> 2, row = ((int)blockIdx.x - n) * 4 + r;
        if (tid < 4) rowmax[tid] = 0u;
        __syncthreads();
        const bool row_ok = row < w_cols;
        float v[GP_MAX_D / 64];
#pragma unroll
        for (int q = 0; q < GP_MAX_D / 64; ++q) {
            const int k = kq + 64 * q;
            v[q] = (row_ok && k < w_rows) ? w[(long long)k * w_ld + row] : 0.f;
        }
        float m = 0.f;
#pragma unroll
        for (int q = 0; q < GP_MAX_D / 64; ++q) m = fmaxf(m, fabsf(v[q]));
        atomicMax(&rowmax[r], __float_as_uint(m));  // non-negative floats order as their bit patterns
        __syncthreads();
        if (!row_ok) return;
        float scale, inverse;
        f16x2_scale(__uint_as_float(rowmax[r]), scale, inverse);
        if (kq == 0) inv_w[row] = inverse;
        const long long total = (long long)w_rows * w_cols;
#pragma unroll
        for (int q = 0; q < GP_MAX_D / 64; ++q) {
            const int k = kq + 64 * q;
            if (k >= w_rows) break;
            const float t = v[q] * scale;
            const _Float16 hi = (_Float16)t, lo = (_Float16)(t - (float)hi);
            const long long i = ((long long)(k >> 4) * w_cols + row) * 16 + (k & 15);
            w_planes[i] = __builtin_bit_cast(unsigned short, hi);
            w_planes[total + i] = __builtin_bit_cast(unsigned short, lo);
        }
        return;
    }
    const int i = blockIdx.x, total = n * d;
    float m = 0.f;
    for (int e = tid; e < d; e += 256) {
        float v[PG_SEG];
#pragma unroll
        for (int k = 0; k < PG_SEG; ++k) v[k] = partial[(int64_t)k * total + i * d + e];
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < PG_SEG; ++k) acc += v[k];
        proto_rows[i * d + e] = acc;
        sm[e] = acc;
        m = fmaxf(m, fabsf(acc));
    }
    m = block_max256(m, smax);
    float scale, inverse;
    f16x2_scale(m, scale, inverse);
    if (tid == 0) inv_rows[row_base[0] + i] = inverse;
    const long long plane = (long long)(d / 16) * rows_alloc * 16;
    for (int c0 = tid * 8; c0 < d; c0 += 256 * 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = sm[c0 + u];
        u32x4_t p0, p1;
        split8h(x, scale, p0, p1);
        unsigned short* dst = planes + ((long long)(c0 / 16) * rows_alloc + row_base[0] + i) * 16 + (c0 & 8);
        *reinterpret_cast<u32x4_t*>(dst) = p0;
        *reinterpret_cast<u32x4_t*>(dst + plane) = p1;
    }
}

__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ out,
                                                          const float* __restrict__ dout, int64_t total,
                                                          float* __restrict__ dy)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float p = out[i];
    dy[i] = dout[i] * ((1.f - p) * p);
}

// ---------------------------------------------------------------------------
// Adam (torch.optim.Adam arithmetic, no weight decay / amsgrad).
// state: [0] step (as float bits of an int32), [1] lr/bc1, [2] 1/sqrt(bc2)
// ---------------------------------------------------------------------------
__global__ void adam_prepare_kernel(AdamState* st, float lr, float b1, float b2)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int32_t step = st->step + 1;
    st->step = step;
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    st->step_size = (float)((double)lr / bc1);
    st->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   const AdamState* __restrict__ st, float b1, float b2, float eps,
                                                   float gscale)
{
    const float step_size = st->step_size, isb = st->inv_sqrt_bc2;
    const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 3 < n) {
        f32x4 pp = *reinterpret_cast<f32x4*>(p + i4);
        f32x4 gg = *reinterpret_cast<const f32x4*>(g + i4);
        f32x4 mm = *reinterpret_cast<f32x4*>(m + i4);
        f32x4 vv = *reinterpret_cast<f32x4*>(v + i4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gg[e] * gscale;
            mm[e] = mm[e] + (1.f - b1) * (ge - mm[e]);
            vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
            pp[e] = pp[e] - step_size * (mm[e] / (sqrtf(vv[e]) * isb + eps));
        }
        *reinterpret_cast<f32x4*>(p + i4) = pp;
        *reinterpret_cast<f32x4*>(m + i4) = mm;
        *reinterpret_cast<f32x4*>(v + i4) = vv;
    } else {
        for (int64_t i = i4; i < n; ++i) {
            const float ge = g[i] * gscale;
            const float me = m[i] + (1.f - b1) * (ge - m[i]);
            const float ve = b2 * v[i] + (1.f - b2) * ge * ge;
            m[i] = me;
            v[i] = ve;
            p[i] = p[i] - step_size * (me / (sqrtf(ve) * isb + eps));
        }
    }
}

// The side jobs of side_jobs.h as a launch of their own, for the cases where no GEMM launch can carry them.
__global__ __launch_bounds__(256) void side_jobs_kernel(const SideJobs sj)
{
    __shared__ __attribute__((aligned(16))) float smem[SIDE_SMEM_FLOATS];
    side_run<256>(sj, (int)blockIdx.x, smem);
}

// ---------------------------------------------------------------------------
// edge precision / recall against a gold edge bitmask.
// gold_bits[g*n + a] has bit b set iff (a,b) is a gold edge.
// per-graph: hits = #pred (a,b) with (a,b) or (b,a) in gold;
// recall = hits/|gold|, precision = hits/|pred|, both 1 when either is empty.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_prf_kernel(const int32_t* __restrict__ edges,
                                                       const int32_t* __restrict__ count,
                                                       const uint64_t* __restrict__ gold, int64_t G, int n,
                                                       float* __restrict__ recall, float* __restrict__ precision)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const uint64_t* gb = gold + g * n;
    int ngold = 0;
    for (int a = 0; a < n; ++a) ngold += __popcll(gb[a]);
    const int npred = count[g];
    const int32_t* e = edges + g * 2 * (n - 1);
    int hits = 0;
    for (int k = 0; k < npred; ++k) {
        const int a = e[2 * k], b = e[2 * k + 1];
        if (((gb[a] >> b) & 1ull) || ((gb[b] >> a) & 1ull)) ++hits;
    }
    if (ngold == 0 || npred == 0) {
        recall[g] = 1.f;
        precision[g] = 1.f;
    } else {
        recall[g] = (float)((double)hits / (double)ngold);
        precision[g] = (float)((double)hits / (double)npred);
    }
}

}  // namespace

// ===========================================================================
// host launchers
// ===========================================================================
int adj_count(const int64_t* ei, int64_t E, int64_t B, int n, int32_t* counts, int32_t* err, hipStream_t s)
{
    if (E == 0) return GTG_OK;
    hipLaunchKernelGGL(adj_count_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, ei, E, B, n, counts, err);
    return check_launch("adj_count_kernel");
}

int stage_batch(const int64_t* ei_src, int64_t E, int64_t* ei_dst, int64_t E_cap, const float* a_src, float* a_dst,
                int64_t a_count, const float* e_src, float* e_dst, int64_t e_count, hipStream_t s)
{
    const int64_t items = 2 * E_cap + (a_count + 3) / 4 + (e_count + 3) / 4;
    if (items == 0) return GTG_OK;
    const int vec = ((reinterpret_cast<uintptr_t>(a_src) | reinterpret_cast<uintptr_t>(a_dst) |
                      reinterpret_cast<uintptr_t>(e_src) | reinterpret_cast<uintptr_t>(e_dst)) & 15) == 0;
    hipLaunchKernelGGL(stage_batch_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, ei_src, E, ei_dst,
                       E_cap, a_src, a_dst, a_count, e_src, e_dst, e_count, vec);
    return check_launch("stage_batch_kernel");
}

int gather_batch(const int64_t* esrc, int64_t E_all, const int64_t* estart, const float* adj_all, int64_t G, int n,
                 const int64_t* idx, int B, int64_t* off, int64_t* ei_dst, int64_t E_cap, float* target_dst, int32_t* err,
                 hipStream_t s)
{
    hipLaunchKernelGGL(gather_offsets_kernel, dim3(1), dim3(256), 0, s, estart, idx, B, G, E_cap, off, err);
    int rc = check_launch("gather_offsets_kernel");
    if (rc) return rc;
    const int tail_blocks = (int)((E_cap + 4095) / 4096) + 1;
    hipLaunchKernelGGL(gather_batch_kernel, dim3((unsigned)(B + tail_blocks)), dim3(256), 0, s, esrc, E_all, estart,
                       adj_all, idx, B, G, n, off, ei_dst, E_cap, target_dst);
    return check_launch("gather_batch_kernel");
}

int clear_i32(int32_t* p, int64_t count, hipStream_t s)
{
    if (count <= 0) return GTG_OK;
    hipLaunchKernelGGL(clear_i32_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, p, count);
    return check_launch("clear_i32_kernel");
}

int adj_normalize(int32_t* counts, int64_t B, int n, float* adj, int32_t* extent, hipStream_t s)
{
    if (B == 0) return GTG_OK;
    hipLaunchKernelGGL(adj_normalize_kernel, dim3((unsigned)B), dim3(256), 0, s, counts, n, adj, extent);
    return check_launch("adj_normalize_kernel");
}

int pool_bwd_gather(const float* adj, const float* dpool, const int32_t* argmax, int64_t B, int n, int d, float* dT2,
                    hipStream_t s)
{
    if (B == 0 || d == 0) return GTG_OK;
    hipLaunchKernelGGL(pool_bwd_gather_kernel, dim3((d + 255) / 256, (unsigned)B), dim3(256), 0, s, adj, dpool, argmax,
                       n, d, dT2);
    return check_launch("pool_bwd_gather_kernel");
}

size_t colsum_workspace(int64_t rows, int cols)
{
    const int chunk = cs_chunk_rows(rows);
    const int64_t nchunks = (rows + chunk - 1) / chunk;
    return nchunks > 1 ? (size_t)nchunks * cols * sizeof(float) : 0;
}

int colsum(const float* X, int64_t rows, int cols, int ld, float* partial, float* out, hipStream_t s)
{
    return colsum_dev(X, rows, nullptr, cols, ld, partial, out, s);
}

int colsum_dev(const float* X, int64_t rows, const int* rows_dev, int cols, int ld, float* partial, float* out,
               hipStream_t s)
{
    if (cols == 0) return GTG_OK;
    const int chunk = cs_chunk_rows(rows);
    const int nchunks = (int)((rows + chunk - 1) / chunk);
    const int vec = ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && (ld % 4 == 0);
    const dim3 grid((cols + CS_COLS - 1) / CS_COLS, nchunks > 0 ? nchunks : 1);
    if (nchunks <= 1) {
        if (rows > 64)
            hipLaunchKernelGGL(colsum_kernel<64>, grid, dim3(1024), 0, s, X, rows, cols, ld, chunk, vec, rows_dev, out);
        else
            hipLaunchKernelGGL(colsum_kernel<16>, grid, dim3(256), 0, s, X, rows, cols, ld, chunk, vec, rows_dev, out);
        return check_launch("colsum_kernel");
    }
    hipLaunchKernelGGL(colsum_kernel<16>, grid, dim3(256), 0, s, X, rows, cols, ld, chunk, vec, rows_dev, partial);
    int rc = check_launch("colsum_kernel");
    if (rc) return rc;
    // second stage: the partials form an (nchunks x cols) matrix; same kernel, one chunk
    const int vec2 = ((reinterpret_cast<uintptr_t>(partial) & 15) == 0) && (cols % 4 == 0);
    hipLaunchKernelGGL(colsum_kernel<16>, dim3((cols + CS_COLS - 1) / CS_COLS, 1), dim3(256), 0, s, partial,
                       (int64_t)nchunks, cols, cols, nchunks, vec2, (const int*)nullptr, out);
    return check_launch("colsum_kernel");
}

// out0 = column sums of X0 (rows0 x cols0), out1 likewise, in one launch when both are a single chunk
// (at most 1024 rows); otherwise two ordinary column sums through `partial`.
int colsum_pair(const float* X0, int64_t rows0, int cols0, int ld0, float* out0, const float* X1, int64_t rows1,
                int cols1, int ld1, float* out1, float* partial, hipStream_t s)
{
    if (rows0 > 1024 || rows1 > 1024 || rows0 < 1 || rows1 < 1 || cols0 < 1 || cols1 < 1) {
        if (int rc = colsum(X0, rows0, cols0, ld0, partial, out0, s)) return rc;
        return colsum(X1, rows1, cols1, ld1, partial, out1, s);
    }
    ColsumJob j0{X0, out0, (int)rows0, cols0, ld0, ((reinterpret_cast<uintptr_t>(X0) & 15) == 0) && (ld0 % 4 == 0),
                 nullptr, 1, 1};
    ColsumJob j1{X1, out1, (int)rows1, cols1, ld1, ((reinterpret_cast<uintptr_t>(X1) & 15) == 0) && (ld1 % 4 == 0),
                 nullptr, 1, 1};
    const int wide = cols0 > cols1 ? cols0 : cols1;
    hipLaunchKernelGGL(colsum_pair_kernel, dim3((wide + CS_COLS - 1) / CS_COLS, 1, 2), dim3(1024), 0, s, j0, j1);
    return check_launch("colsum_pair_kernel");
}

// out = sum of the per-row-block column sums a GEMM epilogue left in `part` (gemm_colsum_rows(M) x cols; with a
// device-side row count only the blocks of live tiles are valid) and out2 = column sums of the short matrix X2.
int colsum_parts_and_short(const float* part, int part_rows, const int* gemm_rows_dev, int cols, float* out,
                           const float* X2, int64_t rows2, int cols2, int ld2, float* out2, float* partial,
                           hipStream_t s)
{
    if (part_rows > 1024 || part_rows < 1 || rows2 > 1024 || rows2 < 1 || cols < 1 || cols2 < 1)
        return set_error(GTG_ERR_UNSUPPORTED, "colsum_parts_and_short: more than 1024 rows");
    (void)partial;
    ColsumJob j0{part, out, part_rows, cols, cols, ((reinterpret_cast<uintptr_t>(part) & 15) == 0) && (cols % 4 == 0),
                 gemm_rows_dev, 128, 2};
    ColsumJob j1{X2, out2, (int)rows2, cols2, ld2, ((reinterpret_cast<uintptr_t>(X2) & 15) == 0) && (ld2 % 4 == 0),
                 nullptr, 1, 1};
    const int wide = cols > cols2 ? cols : cols2;
    hipLaunchKernelGGL(colsum_pair_kernel, dim3((wide + CS_COLS - 1) / CS_COLS, 1, 2), dim3(1024), 0, s, j0, j1);
    return check_launch("colsum_pair_kernel");
}

// Column sums of a tall matrix X (rows read on the device) and of a short one X2: the short one rides in the
// launch of the tall one's second stage (three launches become two).
int colsum_dev_and_short(const float* X, int64_t rows, const int* rows_dev, int cols, int ld, float* partial, float* out,
                         const float* X2, int64_t rows2, int cols2, int ld2, float* out2, hipStream_t s)
{
    const int chunk = cs_chunk_rows(rows);
    const int nchunks = (int)((rows + chunk - 1) / chunk);
    if (nchunks <= 1 || nchunks > 1024 || rows2 > 1024 || rows2 < 1 || cols < 1 || cols2 < 1) {
        if (int rc = colsum_dev(X, rows, rows_dev, cols, ld, partial, out, s)) return rc;
        return colsum(X2, rows2, cols2, ld2, partial, out2, s);  // partial is free again
    }
    const int vec = ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && (ld % 4 == 0);
    hipLaunchKernelGGL(colsum_kernel<16>, dim3((cols + CS_COLS - 1) / CS_COLS, nchunks), dim3(256), 0, s, X, rows, cols, ld,
                       chunk, vec, rows_dev, partial);
    if (int rc = check_launch("colsum_kernel")) return rc;
    return colsum_pair(partial, nchunks, cols, cols, out, X2, rows2, cols2, ld2, out2, nullptr, s);
}

int reparam_kl(const float* mu, const float* logvar, const float* eps, int64_t B, int d, float* z, float* kl_partial,
               float* kl_out, hipStream_t s)
{
    const int64_t total = B * d;
    const int64_t nb = (total + 255) / 256;
    if (nb > 0) {
        hipLaunchKernelGGL(reparam_kl_kernel, dim3((unsigned)nb), dim3(256), 0, s, mu, logvar, eps, total, z,
                           kl_partial);
        int rc = check_launch("reparam_kl_kernel");
        if (rc) return rc;
    }
    if (kl_out == nullptr) return GTG_OK;  // the caller sums the nb partials itself (reparam_kl_partials)
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, kl_partial, nb, reparam_kl_scale(B), kl_out);
    return check_launch("final_sum_kernel");
}

int latent_bwd_elem(const float* dz, const float* mu, const float* logvar, const float* eps, const float* kl_w_dev,
                    float kl_w, int64_t B, int d, int ld_out, float* dmu, float* dlogvar, hipStream_t s)
{
    const int64_t total = B * d;
    if (total == 0) return GTG_OK;
    hipLaunchKernelGGL(latent_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dz, mu, logvar, eps,
                       kl_w_dev, kl_w, 1.f / (float)B, total, d, ld_out, dmu, dlogvar);
    return check_launch("latent_bwd_kernel");
}

int bce_fwd(const float* out, const float* adj, int64_t B, int n, float* row_partial, float* loss, float* dlogits,
            const float* gscale_dev, hipStream_t s)
{
    const int m = n * (n + 1) / 2;
    if (B > 0) {
        hipLaunchKernelGGL(bce_fwd_kernel, dim3((unsigned)B), dim3(256), 0, s, out, adj, n, m, row_partial,
                           m > 0 ? 1.f / ((float)B * (float)m) : 0.f, gscale_dev, dlogits);
        int rc = check_launch("bce_fwd_kernel");
        if (rc) return rc;
    }
    if (loss == nullptr) return GTG_OK;  // the caller sums the B partials itself
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, row_partial, B, bce_scale(B, n), loss);
    return check_launch("final_sum_kernel");
}

// ---------------------------------------------------------------------------
// Allocation-free form of the reconstruction loss.  The per-graph partial sums
// (device memory beyond the call's inputs and outputs) live in a file-scope
// device array instead of a caller workspace.  Every GPU the library runs on
// holds its own copy of the array, at its own address: that address is looked
// up once per device, with that device current, and kept; `bce_scratch` itself
// is only the host-side handle the lookup takes and never a kernel argument.
// Nothing in the array has to start at zero (bce_fwd_kernel writes entry b of
// every graph b < B before final_sum_kernel reads entries [0, B)), so a call
// leaves no state behind that a later, smaller call could pick up.
// ---------------------------------------------------------------------------
constexpr int64_t BCE_STATIC_MAX_GRAPHS = 65536;
constexpr int BCE_STATIC_MAX_DEVICES = 64;
static __device__ __align__(256) float bce_scratch[BCE_STATIC_MAX_GRAPHS + 64];
static std::mutex g_bce_scratch_mu;
static float* g_bce_scratch_ptr[BCE_STATIC_MAX_DEVICES] = {};

int64_t bce_static_max_graphs() { return BCE_STATIC_MAX_GRAPHS; }

int bce_fwd_static(const float* out, const float* adj, int64_t B, int n, float* loss, float* dlogits,
                   const float* gscale_dev, hipStream_t s)
{
    if (B > BCE_STATIC_MAX_GRAPHS)
        return set_error(GTG_ERR_UNSUPPORTED, "gtg_bce_loss_fwd_noalloc: %lld graphs exceed the static scratch (%lld)",
                         (long long)B, (long long)BCE_STATIC_MAX_GRAPHS);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BCE_STATIC_MAX_DEVICES)
        return set_error(GTG_ERR_UNSUPPORTED, "gtg_bce_loss_fwd_noalloc: no current device, or device index >= %d",
                         BCE_STATIC_MAX_DEVICES);
    float* ptr;
    {
        std::lock_guard<std::mutex> lk(g_bce_scratch_mu);
        ptr = g_bce_scratch_ptr[dev];
        if (ptr == nullptr) {
            // first call on this device: the array's address in THIS device's copy of the code object
            const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(bce_scratch));
            if (e != hipSuccess || ptr == nullptr)
                return set_error(GTG_ERR_LAUNCH, "gtg_bce_loss_fwd_noalloc: hipGetSymbolAddress: %s", hipGetErrorString(e));
            g_bce_scratch_ptr[dev] = ptr;
        }
    }
    return bce_fwd(out, adj, B, n, ptr, loss, dlogits, gscale_dev, s);
}

int bce_bwd(const float* out, const float* adj, int64_t B, int n, const float* gdev, float gscale,
            int through_sigmoid, float* dy2, hipStream_t s)
{
    if (B == 0) return GTG_OK;
    const int m = n * (n + 1) / 2;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3((unsigned)B), dim3(256), 0, s, out, adj, n, m, gdev,
                       gscale / ((float)B * (float)m), through_sigmoid, dy2);
    return check_launch("bce_bwd_kernel");
}

int side_jobs_launch(const SideJobs& sj, hipStream_t s)
{
    if (sj.blocks() == 0) return GTG_OK;
    hipLaunchKernelGGL(side_jobs_kernel, dim3((unsigned)sj.blocks()), dim3(256), 0, s, sj);
    return check_launch("side_jobs_kernel");
}

int graph_meta(const int32_t* extent, const float* adj, int64_t B, int n, int32_t* meta, float* adjc, int32_t* zero,
               int zero_count, hipStream_t s)
{
    hipLaunchKernelGGL(adj_compact_kernel, dim3((unsigned)B + 1), dim3(256), 0, s, adj, extent, meta, (int)B, n, adjc, zero,
                       zero_count);
    return check_launch("adj_compact_kernel");
}

size_t proto_grad_workspace(int n, int d) { return (size_t)PG_SEG * n * d * sizeof(float); }

// dT2c, real rows and prototype rows: the gather and stage 1 of the prototype gradient in one launch, then stage 2.
int pool_bwd_gather_compact(const float* adj, const float* dpool, const int32_t* argmax, const int32_t* meta,
                            int64_t B, int n, int d, float* dT2, float* pg_partial, int32_t* zero, int zero_count,
                            hipStream_t s)
{
    if (B == 0 || d == 0) return clear_i32(zero, zero_count, s);
    const int gx = (d + 255) / 256;
    const int pg_rows = (((d + 63) / 64) * PG_SEG + gx - 1) / gx;
    hipLaunchKernelGGL(pool_bwd_gather_compact_kernel, dim3(gx, (unsigned)(B + pg_rows)), dim3(256), 0, s, adj, dpool,
                       argmax, meta, B, n, d, dT2, zero, zero_count, pg_partial, pg_rows);
    if (int rc = check_launch("pool_bwd_gather_compact_kernel")) return rc;
    hipLaunchKernelGGL(proto_grad_stage2, dim3((n * d + 255) / 256), dim3(256), 0, s, pg_partial, meta + 2 * B, n, d, dT2);
    return check_launch("proto_grad_stage2");
}

bool pool_bwd_gather_planes_applies(int n, int d) { return n >= 1 && n <= 64 && d % 16 == 0 && d <= GP_MAX_D; }

// dT2c as the plane GEMM's A operand: planes (3, d / 16, rows_alloc, 16) bf16, real rows and prototype rows; the
// prototype rows also as fp32 (proto_rows, (n, d)).  d a multiple of 16, at most 1024.  The second launch also
// splits the weight w ((w_rows, w_cols), leading dimension w_ld; w_rows a multiple of 16) into w_planes, laid out
// as the GEMM's B operand for a product with w: (3, w_rows / 16, w_cols, 16).
// With inv_rows / inv_w (both or neither): two fp16 planes per operand instead, each row scaled by a power of two,
// inv_rows[row] (rows_alloc floats) and inv_w[w_cols] receiving the inverse scales.
int pool_bwd_gather_planes(const float* adj, const float* dpool, const int32_t* argmax, const int32_t* meta, int64_t B,
                           int n, int d, long long rows_alloc, void* planes, float* proto_rows, float* pg_partial,
                           int32_t* zero, int zero_count, const float* w, int w_rows, int w_cols, int w_ld,
                           void* w_planes, float* inv_rows, float* inv_w, hipStream_t s)
{
    if (!pool_bwd_gather_planes_applies(n, d) || B < 1 || w_rows % 16 != 0 || w_rows < 16 || w_cols < 1 || w_ld < w_cols ||
        (inv_rows == nullptr) != (inv_w == nullptr))
        return set_error(GTG_ERR_INVALID, "pool_bwd_gather_planes: shape");
    const int pg_blocks = ((d + 63) / 64) * PG_SEG;
    if (inv_rows != nullptr) {
        hipLaunchKernelGGL(pool_bwd_gather_planes_kernel<true>, dim3((unsigned)(B + pg_blocks)), dim3(256), 0, s, adj, dpool,
                           argmax, meta, (int)B, n, d, rows_alloc, reinterpret_cast<unsigned short*>(planes), zero,
                           zero_count, pg_partial, pg_blocks, inv_rows);
        if (int rc = check_launch("pool_bwd_gather_planes_kernel")) return rc;
        if (w_rows > GP_MAX_D) return set_error(GTG_ERR_INVALID, "pool_bwd_gather_planes: weight depth above %d", GP_MAX_D);
        hipLaunchKernelGGL(proto_grad_stage2_planes_f16, dim3((unsigned)(n + (w_cols + 3) / 4)), dim3(256), 0, s, pg_partial,
                           meta + 2 * B, n, d, rows_alloc, proto_rows, reinterpret_cast<unsigned short*>(planes), inv_rows, w,
                           w_rows, w_cols, w_ld, reinterpret_cast<unsigned short*>(w_planes), inv_w);
        return check_launch("proto_grad_stage2_planes_f16");
    }
    hipLaunchKernelGGL(pool_bwd_gather_planes_kernel<false>, dim3((unsigned)(B + pg_blocks)), dim3(256), 0, s, adj, dpool,
                       argmax, meta, (int)B, n, d, rows_alloc, reinterpret_cast<unsigned short*>(planes), zero, zero_count,
                       pg_partial, pg_blocks, (float*)nullptr);
    if (int rc = check_launch("pool_bwd_gather_planes_kernel")) return rc;
    const int stage2_blocks = (n * d + 255) / 256;
    const long long w_blocks = ((long long)w_rows * w_cols + 255) / 256;
    hipLaunchKernelGGL(proto_grad_stage2_planes, dim3((unsigned)(stage2_blocks + w_blocks)), dim3(256), 0, s, pg_partial,
                       meta + 2 * B, n, d, rows_alloc, proto_rows, reinterpret_cast<unsigned short*>(planes), stage2_blocks, w,
                       w_rows, w_cols, w_ld, reinterpret_cast<unsigned short*>(w_planes));
    return check_launch("proto_grad_stage2_planes");
}

int sigmoid_bwd(const float* out, const float* dout, int64_t total, float* dy, hipStream_t s)
{
    if (total == 0) return GTG_OK;
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, out, dout, total, dy);
    return check_launch("sigmoid_bwd_kernel");
}

int adam_stateful(float* p, const float* g, float* m, float* v, int64_t n, void* state, float lr, float b1, float b2,
                  float eps, float gscale, bool prepared, hipStream_t s)
{
    AdamState* st = reinterpret_cast<AdamState*>(state);
    int rc = GTG_OK;
    if (!prepared) {
        hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, s, st, lr, b1, b2);
        rc = check_launch("adam_prepare_kernel");
    }
    if (rc || n == 0) return rc;
    const int64_t nv = (n + 3) / 4;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, st, b1, b2, eps,
                       gscale);
    return check_launch("adam_kernel");
}

int edge_prf(const int32_t* edges, const int32_t* count, const uint64_t* gold_bits, int64_t G, int n, float* recall,
             float* precision, hipStream_t s)
{
    if (G == 0) return GTG_OK;
    hipLaunchKernelGGL(edge_prf_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, edges, count, gold_bits, G,
                       n, recall, precision);
    return check_launch("edge_prf_kernel");
}

}  // namespace gtg
