// The two propagate steps of the encoder as sparse row products (gfx950): the first GCN layer on identity node
// features, GCNConv's propagate + bias + ReLU at /root/reference/src/graph_vae.py:107-108 (conv1, relu) for
// x = I_n, and conv2's propagate + bias fused with global_max_pool (graph_vae.py:110, :25).
//
// With identity features conv1 is H1 = relu(A W1^T + b1) where A is the
// normalised adjacency: an AMR graph's row of A has deg + 1 nonzeros (3 to 4 on
// average, out of N <= 64), so a dense GEMM with K = N multiplies mostly zeros
// and the step is bound by the store of the (rows, 2d) activations:
//
//   h1c[r, :] = relu(b1 + sum_{k : A[r,k] != 0} A[r,k] * W1[:, k])
//
// One wavefront takes one row at a time: lane k loads A[r, k] (one coalesced
// 256-byte read), a ballot gives the nonzero columns, and the wave walks them
// in ascending k with a scalar loop (v_readlane for the coefficient, one
// 16-byte LDS read of the transposed weight slab per lane).  Rows come from
// the compact-row layout: their count is read on the device.
#include "gtg_internal.h"

namespace gtg {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v where keep, +0 elsewhere, as a bit mask: written as a select the compiler moves the load that feeds
// it back under a branch
__device__ __forceinline__ float keep_if(float v, bool keep)
{
    return __int_as_float(__float_as_int(v) & -(int)keep);
}

constexpr int C1_THREADS = 512;
constexpr int C1_COLS = 256;       // output columns per block, 4 per lane
constexpr int C1_LD = C1_COLS + 4; // LDS row stride of the transposed weight slab (keeps 16-byte alignment)
constexpr int C1_UNROLL = 4;       // adjacency rows in flight per wave

// grid = slabs * chunks; block b: column slab b % slabs, row chunk b / slabs.
__global__ __launch_bounds__(C1_THREADS) void conv1_sparse_kernel(const float* __restrict__ adjc, int n,
                                                                  const float* __restrict__ W1,
                                                                  const float* __restrict__ b1, int H,
                                                                  const int* __restrict__ rows_dev, int rows_max,
                                                                  int slabs, float* __restrict__ h1c)
{
    __shared__ __attribute__((aligned(16))) float Wt[64 * C1_LD];  // Wt[k][c] = W1[c0 + c][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slab = blockIdx.x % slabs, chunk = blockIdx.x / slabs, chunks = gridDim.x / slabs;
    const int c0 = slab * C1_COLS;
    int rows = rows_max;
    if (rows_dev) rows = min(rows, rows_dev[0]);

    // lane = k: one coalesced read of W1's row c per wave instruction
    {
        constexpr int PER_WAVE = C1_COLS / (C1_THREADS / 64);
        float t[PER_WAVE];  // all of a wave's loads in flight before the first LDS write
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int c = wave + i * (C1_THREADS / 64);
            t[i] = (lane < n && c0 + c < H) ? W1[(long long)(c0 + c) * n + lane] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i)
            if (lane < n) Wt[lane * C1_LD + wave + i * (C1_THREADS / 64)] = t[i];
    }
    __syncthreads();

    const int per = (rows + chunks - 1) / chunks;
    const int r0 = chunk * per, r1 = min(rows, r0 + per);
    const int col = c0 + 4 * lane;
    const bool col_ok = col < H;  // H is a multiple of 4
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (col_ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bias[q] = b1[col + q];
    }
    constexpr int WAVES = C1_THREADS / 64;
    // the next group of adjacency rows is in flight while the current one is multiplied and stored
    // (loads are unconditional, row and lane clamped into range, and zeroed afterwards: a guarded load
    // becomes a branch with a full wait behind it)
    float a[C1_UNROLL], a_next[C1_UNROLL];
    int rb = r0 + wave * C1_UNROLL;
    const int lane_c = min(lane, n - 1), r_last = max(r1 - 1, 0);
    const bool lane_ok = lane < n;
#pragma unroll
    for (int u = 0; u < C1_UNROLL; ++u) {
        const float t = adjc[(long long)min(rb + u, r_last) * n + lane_c];
        a_next[u] = keep_if(t, rb + u < r1 && lane_ok);
    }
    for (; rb < r1; rb += WAVES * C1_UNROLL) {
        const int rn = rb + WAVES * C1_UNROLL;
#pragma unroll
        for (int u = 0; u < C1_UNROLL; ++u) {
            a[u] = a_next[u];
            const float t = adjc[(long long)min(rn + u, r_last) * n + lane_c];
            a_next[u] = keep_if(t, rn + u < r1 && lane_ok);
        }
#pragma unroll
        for (int u = 0; u < C1_UNROLL; ++u) {
            const int r = rb + u;
            if (r >= r1) break;  // wave-uniform
            unsigned long long mask = __ballot(a[u] != 0.f);
            f32x4 acc = bias;
            while (mask) {
                const int k = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const float ak = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[u]), k));
                const f32x4 w = *reinterpret_cast<const f32x4*>(&Wt[k * C1_LD + 4 * lane]);
                acc += ak * w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fmaxf(acc[q], 0.f);
            if (col_ok) *reinterpret_cast<f32x4*>(h1c + (long long)r * H + col) = acc;
        }
    }
}

// ---- second propagate + max-pool, per graph ---------------------------------------------
//   pooled[b, c] = max_{r < n_b} ( sum_k A_b[r, k] T2c[off_b + k, c] + b2[c] ), arg-max row (lowest on ties),
// merged with the maximum over the prototype rows n_b .. n-1 of T2c (a real row wins ties).
// A block takes one graph and 128 columns: the graph's rows of T2c (at most 64 x 128 floats) are staged in
// LDS with every load in flight at once, each wave walks every fourth adjacency row sparsely (lane k loads
// A_b[r, k], ballot, one 8-byte LDS read per nonzero and lane), keeping a running maximum for its two columns;
// the four waves' maxima are combined in row order.
constexpr int PS_COLS = 128;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void pool_sparse_kernel(const float* __restrict__ adj, const float* __restrict__ t2c,
                                                          const int32_t* __restrict__ meta, const float* __restrict__ b2,
                                                          int B, int n, int d, float* __restrict__ pooled,
                                                          int32_t* __restrict__ argmax)
{
    __shared__ __attribute__((aligned(16))) float T[64 * PS_COLS];
    __shared__ float wv[4][PS_COLS];
    __shared__ int wi[4][PS_COLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, c0 = blockIdx.x * PS_COLS;
    const int nb = meta[b], off = meta[B + b], R = meta[2 * B];

    {   // stage rows off .. off + nb - 1, columns c0 .. c0 + 127 (d is a multiple of 4)
        f32x4 v[8];
        const int c4 = tid & 31, col = c0 + 4 * c4;
        const int colc = min(col, d - 4), k_last = max(nb - 1, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = (tid >> 5) + 8 * i;
            v[i] = *reinterpret_cast<const f32x4*>(t2c + (long long)(off + min(k, k_last)) * d + colc);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = (tid >> 5) + 8 * i;
            if (k < nb && col < d) *reinterpret_cast<f32x4*>(&T[k * PS_COLS + 4 * c4]) = v[i];
        }
    }
    // this wave's adjacency rows, all in flight
    float a[16];
    const float* A = adj + (long long)b * n * n;
    const int lane_c = min(lane, n - 1);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int r = wave + 4 * t;
        a[t] = keep_if(A[min(r, n - 1) * n + lane_c], r < nb && lane < nb);
    }
    const int col = c0 + 2 * lane;
    f32x2 bias = {0.f, 0.f};
    if (col < d) bias[0] = b2[col];
    if (col + 1 < d) bias[1] = b2[col + 1];
    __syncthreads();

    f32x2 best = {0.f, 0.f};
    int bi0 = -1, bi1 = -1;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int r = wave + 4 * t;
        if (r >= nb) break;  // wave-uniform
        unsigned long long mask = __ballot(a[t] != 0.f);
        f32x2 acc = bias;
        while (mask) {
            const int k = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const float ak = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[t]), k));
            acc += ak * *reinterpret_cast<const f32x2*>(&T[k * PS_COLS + 2 * lane]);
        }
        if (bi0 < 0 || acc[0] > best[0]) {
            best[0] = acc[0];
            bi0 = r;
        }
        if (bi1 < 0 || acc[1] > best[1]) {
            best[1] = acc[1];
            bi1 = r;
        }
    }
    wv[wave][2 * lane] = best[0];
    wv[wave][2 * lane + 1] = best[1];
    wi[wave][2 * lane] = bi0;
    wi[wave][2 * lane + 1] = bi1;
    __syncthreads();
    if (tid >= PS_COLS) return;
    const int c = c0 + tid;
    if (c >= d) return;
    float bv = 0.f;
    int bidx = -1;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float v = wv[w][tid];
        const int i = wi[w][tid];
        if (i >= 0 && (bidx < 0 || v > bv || (v == bv && i < bidx))) {
            bv = v;
            bidx = i;
        }
    }
    // prototype rows n_b .. n-1, stored once after the R real rows
    const float bc = b2[c];
    float pv = 0.f;
    int pi = -1;
    const float* Q = t2c + (long long)R * d + c;
#pragma unroll 4
    for (int i = nb; i < n; ++i) {
        const float v = Q[(long long)i * d] + bc;
        if (pi < 0 || v > pv) {
            pv = v;
            pi = i;
        }
    }
    if (pi >= 0 && (bidx < 0 || pv > bv)) {
        bv = pv;
        bidx = pi;
    }
    pooled[(long long)b * d + c] = bv;
    argmax[(long long)b * d + c] = bidx;
}

}  // namespace

int conv1_sparse(const float* adjc, int n, const float* W1, const float* b1, int H, const int* rows_dev, int rows_max,
                 float* h1c, hipStream_t s)
{
    if (n < 1 || n > 64 || (H & 3) != 0) return set_error(GTG_ERR_INVALID, "conv1_sparse: n in [1, 64], H a multiple of 4");
    const int slabs = (H + C1_COLS - 1) / C1_COLS;
    int chunks = 512 / slabs;  // two blocks of 512 threads per CU
    if (chunks < 1) chunks = 1;
    ProfScope prof(s, "conv1_sparse", rows_max, H, n);
    hipLaunchKernelGGL(conv1_sparse_kernel, dim3(slabs * chunks), dim3(C1_THREADS), 0, s, adjc, n, W1, b1, H, rows_dev,
                       rows_max, slabs, h1c);
    return check_launch("conv1_sparse_kernel");
}

int pool_sparse(const float* adj, const float* t2c, const int32_t* meta, const float* b2, int64_t B, int n, int d,
                float* pooled, int32_t* argmax, hipStream_t s)
{
    if (n < 1 || n > 64 || (d & 3) != 0) return set_error(GTG_ERR_INVALID, "pool_sparse: n in [1, 64], d a multiple of 4");
    if (B == 0) return GTG_OK;
    ProfScope prof(s, "pool_sparse", B * n, d, n);
    hipLaunchKernelGGL(pool_sparse_kernel, dim3((d + PS_COLS - 1) / PS_COLS, (unsigned)B), dim3(256), 0, s, adj, t2c, meta,
                       b2, (int)B, n, d, pooled, argmax);
    return check_launch("pool_sparse_kernel");
}

}  // namespace gtg
