// The two propagate steps of the encoder as sparse row products (gfx950): the first GCN layer on identity node
// features, GCNConv's propagate + bias + ReLU at /root/reference/src/graph_vae.py:107-108 (conv1, relu) for
// x = I_n, and conv2's propagate + bias fused with global_max_pool (graph_vae.py:110, :25); and the gradient
// of conv2's weight taken through that max-pool's one-winner-per-column routing (autograd of the same lines).
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
typedef float f32x32 __attribute__((ext_vector_type(32)));

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
// With `absmax` (2 x GEMM_ABSMAX_PARTS floats): the partial maxima the two-plane fp16 GEMM that consumes h1c wants
// of its operands -- the block's largest stored activation in slot b, the largest magnitude of its share of the
// `w_next_count` weights at w_next in slot GEMM_ABSMAX_PARTS + b, zeros in the slots no block owns -- so that
// product needs no pass of its own over them.  The grid has at most GEMM_ABSMAX_PARTS blocks.
__global__ __launch_bounds__(C1_THREADS) void conv1_sparse_kernel(const float* __restrict__ adjc, int n,
                                                                  const float* __restrict__ W1,
                                                                  const float* __restrict__ b1, int H,
                                                                  const int* __restrict__ rows_dev, int rows_max,
                                                                  int slabs, float* __restrict__ h1c,
                                                                  float* __restrict__ absmax,
                                                                  const float* __restrict__ w_next, long long w_next_count)
{
    __shared__ __attribute__((aligned(16))) float Wt[64 * C1_LD];  // Wt[k][c] = W1[c0 + c][k]
    __shared__ float smax[2][C1_THREADS / 64];
    float amax = 0.f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // in a scalar register: row indices stay scalar
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
            if (col_ok) {
                *reinterpret_cast<f32x4*>(h1c + (long long)r * H + col) = acc;
                amax = fmaxf(amax, fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])));
            }
        }
    }
    if (absmax == nullptr) return;  // kernel argument: uniform
    float wmax = 0.f;
    {
        const long long per = (w_next_count + gridDim.x - 1) / gridDim.x;
        const long long e0 = (long long)blockIdx.x * per, e1 = min(w_next_count, e0 + per);
        for (long long e = e0 + tid; e < e1; e += C1_THREADS) wmax = fmaxf(wmax, fabsf(w_next[e]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        amax = fmaxf(amax, __shfl_xor(amax, o));
        wmax = fmaxf(wmax, __shfl_xor(wmax, o));
    }
    if (lane == 0) {
        smax[0][wave] = amax;
        smax[1][wave] = wmax;
    }
    __syncthreads();
    if (tid < 2) {
        float m = 0.f;
#pragma unroll
        for (int w = 0; w < C1_THREADS / 64; ++w) m = fmaxf(m, smax[tid][w]);
        float* out = absmax + tid * GEMM_ABSMAX_PARTS;
        out[blockIdx.x] = m;
        for (int sl = blockIdx.x + gridDim.x; sl < GEMM_ABSMAX_PARTS; sl += gridDim.x) out[sl] = 0.f;
    }
}

// ---- second propagate + max-pool, per graph ---------------------------------------------
//   pooled[b, c] = max_{r < n_b} ( sum_k A_b[r, k] T2c[off_b + k, c] + b2[c] ), arg-max row (lowest on ties),
// merged with the maximum over the prototype rows n_b .. n-1 of T2c (a real row wins ties).
// A block takes one graph and 128 columns: the graph's rows of T2c (at most 64 x 128 floats) are staged in
// LDS with every load in flight at once, each wave walks every fourth adjacency row sparsely (lane k loads
// A_b[r, k], ballot, one 8-byte LDS read per nonzero and lane) and takes every fourth prototype row as it is,
// keeping a running maximum for its two columns; the four waves' maxima are combined in row order.
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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // in a scalar register: row indices stay scalar
    const int b = blockIdx.y, c0 = blockIdx.x * PS_COLS;
    const int nb = meta[b], off = meta[B + b], R = meta[2 * B];

    // Every global load of the block is issued here, back to back, so that the block pays one memory latency:
    // the graph's rows of T2c for the LDS stage, this wave's adjacency rows, this lane's two columns of the
    // prototype rows n_b .. n-1 (stored once after the R real rows) and of the bias.  All of them are
    // unconditional loads at clamped indices: a load whose only use sits under a condition is moved under that
    // branch by the compiler and waited for on the spot.
    const int col = c0 + 2 * lane, colc = min(col, d - 2);  // (d is a multiple of 4: col < d implies col + 1 < d)
    f32x2 bias;
    bias[0] = b2[colc];
    bias[1] = b2[colc + 1];
    f32x4 v[8];
    const int c4 = tid & 31, scol = min(c0 + 4 * c4, d - 4), k_last = max(nb - 1, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = (tid >> 5) + 8 * i;
        v[i] = *reinterpret_cast<const f32x4*>(t2c + (long long)(off + min(k, k_last)) * d + scol);
    }
    float a[16];
    const float* A = adj + (long long)b * n * n;
    const int lane_c = min(lane, n - 1);
#pragma unroll
    for (int t = 0; t < 16; ++t) a[t] = A[min(wave + 4 * t, n - 1) * n + lane_c];
    f32x2 pr[16];
    const float* Q = t2c + (long long)R * d + colc;
#pragma unroll
    for (int t = 0; t < 16; ++t) pr[t] = *reinterpret_cast<const f32x2*>(Q + (long long)min(max(wave + 4 * t, nb), n - 1) * d);
    // (pins the adjacency and bias loads above the LDS stage: left alone, the compiler sinks the ones whose first
    // use is late in the walk below the barrier, where each is a memory latency of its own)
    asm volatile("" : "+v"(bias[0]), "+v"(bias[1]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]),
                 "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]),
                 "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
#pragma unroll
    for (int t = 0; t < 16; t += 4)
        asm volatile("" : "+v"(pr[t]), "+v"(pr[t + 1]), "+v"(pr[t + 2]), "+v"(pr[t + 3]));
    // rows past n_b and columns past d are staged too (repeats of valid memory) and never read back
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = (tid >> 5) + 8 * i;
        *reinterpret_cast<f32x4*>(&T[k * PS_COLS + 4 * c4]) = v[i];
    }
    const unsigned long long lanes_nb = nb >= 64 ? ~0ull : (1ull << nb) - 1;
    __syncthreads();

    // Rows wave, wave + 4, ...: real rows r < n_b are walked sparsely, prototype rows n_b <= r < n are
    // T2c's stored prototype rows plus the bias.  Strictly-greater updates in ascending row order keep the lowest
    // row on ties, and a real row (lower index) therefore wins a tie against a prototype row.
    f32x2 best = {0.f, 0.f};
    int bi0 = -1, bi1 = -1;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int r = wave + 4 * t;
        if (r >= n) break;  // wave-uniform
        f32x2 acc;
        if (r < nb) {
            unsigned long long mask = __ballot(a[t] != 0.f) & lanes_nb;
            acc = bias;
            while (mask) {
                const int k = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const float ak = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[t]), k));
                acc += ak * *reinterpret_cast<const f32x2*>(&T[k * PS_COLS + 2 * lane]);
            }
        } else {
            acc = pr[t] + bias;
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
        const float v2 = wv[w][tid];
        const int i = wi[w][tid];
        if (i >= 0 && (bidx < 0 || v2 > bv || (v2 == bv && i < bidx))) {
            bv = v2;
            bidx = i;
        }
    }
    pooled[(long long)b * d + c] = bv;
    argmax[(long long)b * d + c] = bidx;
}

// ---------------------------------------------------------------------------
// dW2 through the max-pool's sparsity.  T2 = H1 W2^T is aggregated and max-pooled, so the gradient that reaches
// W2 is dW2[c, :] = sum_b g[b, c] * Y_b[am_b(c), :], where g = dpooled, am = the winning row and
// Y_b = A_b H1_b is the aggregated hidden block of graph b (a prototype row aggregates to itself): one row of
// 2d numbers per graph and pooled column instead of the dense (rows x d)^T (rows x 2d) product, 1/n_b of
// its multiply-adds.  A block owns 64 columns of H1c and up to 512 pooled columns c and walks every G-th
// graph: it stages the graph's rows of the slab in LDS, aggregates them sparsely (lane k holds A_b[i, k],
// ballot, v_readlane coefficient, one LDS read per nonzero), then every lane -- one pooled column c each --
// adds g * Y[am, 0..63] to its 64 accumulators, two columns per 8-byte LDS read (row stride 66: lanes on
// different rows read different bank pairs unless the rows are 32 apart, lanes on the same row share a broadcast;
// 8-byte reads move twice the bytes per LDS cycle of 4-byte ones).  The items form a pipeline with ONE barrier
// each: phase t aggregates item t + 1, accumulates item t and stages item t + 2 (two staged and two aggregated
// tiles), and the three kinds of operand -- slab rows, adjacency rows, per-column gradient and winning row --
// are fetched two to three phases ahead of their use.  The columns won by a prototype row arrive as n more items:
// row i of the prototype block of dT2c (the sum of g over the graphs whose maximum row i won) times prototype
// row i of H1c.  Each group of graphs leaves a partial dW2, stored transposed (c along the lanes: whole 256-byte
// runs); dw_reduce_pair_kernel adds the groups in order and transposes back: reproducible.
// What bounds it is instruction issue, the scalar unit first (ablations: the parts' times add up; more
// instructions per nonzero for more reads in flight -- eight rows side by side, two nonzeros per trip, straight-line
// blocks of four -- all measured slower): every load is one buffer instruction (scalar row offset, item-independent
// lane offset, the range check of the resource instead of index clamps per load), rows are staged unconditionally,
// the walk's bit clear is one s_bitset0_b64.  70.8 -> 55.5 us at B = 512, n = 64, d = 512 (8-byte reads -8.5, the
// one-barrier pipeline -1, the leaner instruction stream -6), bit-identical results.
constexpr int DW_THREADS = 512;
constexpr int DW_COLS = 64;
constexpr int DW_LD = 66;   // row stride of the LDS tiles in floats: even, so that a lane reads two columns with one 8-byte load;
                             // such loads go over 64 banks, and 66 = 2 (mod 64) puts rows i and i + 32 only on the same pair
constexpr int DW_CPB = 512;   // pooled columns per block: 64 per wave
// 32 groups x 16 slabs at 2d = 1024: 512 blocks, two per CU (hence the bound of 4 waves per SIMD, 128 registers;
// 4 x 16.5 KB of LDS per block): a block spends much of an item waiting (the barrier, the dependent LDS reads of the
// adjacency walk) and the second block fills those waits -- worth more than the second copy of the partials costs
// (16 groups, one block per CU: 93 + 8 us; 32 groups: 13 us less per step).
constexpr int DW_GROUPS = 32;
constexpr int DW_INFLIGHT = 2;  // batches of four 8-byte LDS reads in flight while a lane accumulates (2 or 3)

__global__ __launch_bounds__(DW_THREADS, 4) void dw2_sparse_kernel(const float* __restrict__ adj, const float* __restrict__ h1c,
                                                                int ldh, const float* __restrict__ dpool,
                                                                const int32_t* __restrict__ argmax,
                                                                const int32_t* __restrict__ meta,
                                                                const float* __restrict__ dT2c,
                                                                const float* __restrict__ proto_rows, int B, int n, int d,
                                                                float* __restrict__ partial)
{
    // Four tiles: the staged rows and the aggregated rows of two items each, so that one barrier per item suffices
    // (phase t aggregates item t + 1, accumulates item t and stages item t + 2).  4 x 16.5 KB: two blocks per CU.
    __shared__ float Hs[2][64 * DW_LD];
    __shared__ __align__(16) float Y[2][64 * DW_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // in a scalar register: row indices stay scalar
    // Blocks go to the eight XCDs in turn (linear id mod 8), and each XCD has its own L2: the blocks of one group
    // of graphs -- one per 64-column slab -- are dealt to ONE XCD, so that the group's adjacency rows, gradients and
    // winning rows are fetched into that L2 once instead of once per slab, and a row's 256-byte pieces are asked
    // for together.  (DW_GROUPS is a multiple of 8.)
    const int nslab = ldh / DW_COLS;
    const int slab = (blockIdx.x >> 3) % nslab;
    const int G = DW_GROUPS, gr = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / nslab);
    const int c = blockIdx.z * DW_CPB + tid;      // this lane's pooled column while accumulating
    const bool c_ok = c < d;
    const int cc = min(c, d - 1);
    const bool wave_live = blockIdx.z * DW_CPB + wave * 64 < d;  // wave-uniform
    const int R = meta[2 * B];
    const int ng = gr < B ? (B - gr + G - 1) / G : 0;  // graphs gr, gr + G, ...
    const int np = gr < n ? (n - gr + G - 1) / G : 0;  // prototype rows gr, gr + G, ...
    const int items = ng + np;

    for (int i = tid; i < 64 * DW_LD; i += DW_THREADS) {
        Hs[0][i] = 0.f;  // rows past a graph's extent are read with a zero coefficient: keep them finite
        Hs[1][i] = 0.f;
        Y[0][i] = 0.f;
        Y[1][i] = 0.f;
    }
    float acc[DW_COLS];
#pragma unroll
    for (int k = 0; k < DW_COLS; ++k) acc[k] = 0.f;

    // Per item (this block's graphs, then its prototype rows), item t in lane t: extent (-1 marks a prototype
    // row) and first stored row.  Every load below is straight-line code at clamped indices, the same for both
    // kinds of item and for items past the end (a harmless repeat of the last one): with a branch around the
    // loads the compiler drains every outstanding load at the top of the loop, which undoes the prefetch.  (At
    // most 64 items per group: checked on the host.)
    int nbv = -1, offv = 0;
    if (lane < ng) {
        nbv = meta[gr + G * lane];
        offv = meta[B + gr + G * lane];
    } else if (lane < items) {
        offv = R + gr + G * (lane - ng);
    }
    // The three kinds of operand of an item are fetched a different number of phases ahead of their use, each
    // into one register set: its rows of the H1c slab (staged two phases before the item is accumulated), its
    // adjacency rows (walked one phase before) and its head -- per pooled column the gradient and the winning row.
    struct Head {
        float g;
        int am, nb;  // nb < 0: a prototype row
    };
    // A load is one buffer instruction: a resource that starts at the item's first row (adjacency block), a scalar
    // offset for row wave + 8 t -- clamped to n - 1: past the item's extent that is a row of the next graph or a
    // prototype row (the stored rows end with n prototype rows), loaded and never used -- and a per-lane offset
    // that depends on neither.
    float row[8], a[8];
    int srow[8], sadj[8];  // scalars (the wave index is in a scalar register)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int j = min(wave + 8 * t, n - 1);
        srow[t] = 4 * j * ldh;
        sadj[t] = 4 * j * n;
    }
    const int lane_r = 4 * (slab * DW_COLS + lane), lane_a = 4 * min(lane, n - 1);
    auto load_rows = [&](int it_raw) {
        const int it = min(it_raw, items - 1);
        const int nb = __builtin_amdgcn_readlane(nbv, it), off = __builtin_amdgcn_readlane(offv, it);
        // a prototype row is fetched with the whole prototype block (rows R .. R + n - 1)
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(h1c) + (long long)(nb < 0 ? R : off) * ldh, 0,
                                                          4 * n * ldh, 0x00020000);
#pragma unroll
        for (int t = 0; t < 8; ++t) row[t] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, lane_r, srow[t], 0));
    };
    auto load_adj = [&](int it_raw) {
        const int it = min(it_raw, items - 1);
        const int nb = __builtin_amdgcn_readlane(nbv, it);
        const int b = nb < 0 ? 0 : gr + G * it;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(adj) + (long long)b * n * n, 0, 4 * n * n, 0x00020000);
#pragma unroll
        for (int t = 0; t < 8; ++t) a[t] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, lane_a, sadj[t], 0));
    };
    auto load_head = [&](Head& h, int it_raw) {
        const int it = min(it_raw, items - 1);
        const int nb = __builtin_amdgcn_readlane(nbv, it), off = __builtin_amdgcn_readlane(offv, it);
        const bool proto = nb < 0;
        const int b = proto ? 0 : gr + G * it;
        const int am = argmax[(long long)b * d + cc];
        // a prototype row's gradient: row off of dT2c, or row off - R of a separate (n, d) block
        const float* gsrc = proto ? (proto_rows ? proto_rows + (long long)(off - R) * d : dT2c + (long long)off * d)
                                  : dpool + (long long)b * d;
        const bool real = proto || (am >= 0 && am < nb);
        h.g = keep_if(gsrc[cc], c_ok && real);
        h.am = proto ? off - R : (real ? am : 0);
        h.nb = nb;
    };
    // rows of the item in `row` -> Hb, all 64 of them (rows past the extent are never read; for a prototype row
    // the whole prototype block is staged and aggregate() picks the row)
    auto stage = [&](float* Hb) {
#pragma unroll
        for (int t = 0; t < 8; ++t) Hb[(wave + 8 * t) * DW_LD + lane] = row[t];
    };
    // Yc[i, :] = sum_k A[i, k] Hb[k, :] for the wave's rows i = wave, wave + 8, ...: the nonzeros of a row in ascending k
    auto aggregate = [&](const Head& h, const float* Hb, float* Yc) {
        const int nb = h.nb;
        if (nb < 0) {  // block-uniform
            if (wave == 0) Yc[h.am * DW_LD + lane] = Hb[h.am * DW_LD + lane];
            return;
        }
        const unsigned long long cols = nb >= 64 ? ~0ull : (1ull << nb) - 1ull;  // columns of the graph's own nodes
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int i = wave + 8 * t;
            if (i >= nb) break;  // wave-uniform
            unsigned long long mask = __ballot(a[t] != 0.f) & cols;
            float y = 0.f;
            while (mask) {
                const int k = __ffsll((long long)mask) - 1;
                asm("s_bitset0_b64 %0, %1" : "+s"(mask) : "s"(k));  // (the compiler takes three instructions for mask &= mask - 1)
                const float ak = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[t]), k));
                y = fmaf(ak, Hb[k * DW_LD + lane], y);
            }
            Yc[i * DW_LD + lane] = y;
        }
    };
    auto accumulate = [&](const Head& h, const float* Yc) {
        const float g = h.g;
        if (wave_live) {
            // The 32 reads are written out as single ds_read_b64 (left to itself the compiler pairs them into
            // ds_read2_b64, which goes over 32 banks at half the rate), in eight batches of four with DW_INFLIGHT
            // batches in flight; a wait carries the batch it retires as operands, so that no use can move above it.
            const unsigned ya = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)Yc + 4u * (unsigned)(h.am * DW_LD);
            f32x2 v[DW_INFLIGHT * 4];
#define DW_READ4(base, kb)                                                                                             \
    asm volatile("ds_read_b64 %0, %4 offset:%5\n\tds_read_b64 %1, %4 offset:%5+8\n\tds_read_b64 %2, %4 offset:%5+16\n\t"  \
                 "ds_read_b64 %3, %4 offset:%5+24"                                                                     \
                 : "=&v"(v[base]), "=&v"(v[base + 1]), "=&v"(v[base + 2]), "=&v"(v[base + 3])                           \
                 : "v"(ya), "n"(32 * (kb)))
#define DW_WAIT4(cnt, base)                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(%4)"                                                                               \
                 : "+v"(v[base]), "+v"(v[base + 1]), "+v"(v[base + 2]), "+v"(v[base + 3])                               \
                 : "n"(cnt))
#define DW_FMA4(base, kb)                                                                                              \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                    \
        acc[8 * (kb) + 2 * q] = fmaf(g, v[base + q][0], acc[8 * (kb) + 2 * q]);                                        \
        acc[8 * (kb) + 2 * q + 1] = fmaf(g, v[base + q][1], acc[8 * (kb) + 2 * q + 1]);                                \
    }
            // batch kb (columns 8 kb .. 8 kb + 7) lives in slot kb % DW_INFLIGHT
#define DW_STEP(kb)                                                                                                    \
    DW_WAIT4((DW_INFLIGHT - 1) * 4 < (7 - (kb)) * 4 ? (DW_INFLIGHT - 1) * 4 : (7 - (kb)) * 4, 4 * ((kb) % DW_INFLIGHT)); \
    DW_FMA4(4 * ((kb) % DW_INFLIGHT), kb);                                                                             \
    if ((kb) + DW_INFLIGHT < 8) DW_READ4(4 * ((kb) % DW_INFLIGHT), (kb) + DW_INFLIGHT < 8 ? (kb) + DW_INFLIGHT : 0)
            DW_READ4(0, 0);
            DW_READ4(4, 1);
            if (DW_INFLIGHT > 2) DW_READ4(4 * (2 % DW_INFLIGHT), 2);
            DW_STEP(0);
            DW_STEP(1);
            DW_STEP(2);
            DW_STEP(3);
            DW_STEP(4);
            DW_STEP(5);
            DW_STEP(6);
            DW_STEP(7);
#undef DW_STEP
#undef DW_READ4
#undef DW_WAIT4
#undef DW_FMA4
        }
    };
    // phase `it`: hc = head of item it, hn = head of item it + 1
    auto phase = [&](Head& hc, const Head& hn, int it) {
        if (it + 1 < items) aggregate(hn, Hs[(it + 1) & 1], Y[(it + 1) & 1]);
        load_adj(it + 2);            // a[] is free again; walked one phase from now
        accumulate(hc, Y[it & 1]);
        load_head(hc, it + 2);       // accumulated two phases from now
        if (it + 2 < items) stage(Hs[it & 1]);   // Hs[it & 1] was last read one phase ago
        load_rows(it + 3);
        __syncthreads();
    };
    Head h0, h1;
    h0.g = h1.g = 0.f;
    h0.am = h1.am = 0;
    h0.nb = h1.nb = 0;
    if (items > 0) {
        load_rows(0);
        load_adj(0);
        load_head(h0, 0);
        load_head(h1, 1);
    }
    __syncthreads();  // LDS cleared
    if (items > 0) {
        stage(Hs[0]);
        load_rows(1);
    }
    __syncthreads();
    if (items > 0) {
        aggregate(h0, Hs[0], Y[0]);
        load_adj(1);
        if (items > 1) stage(Hs[1]);
        load_rows(2);
    }
    __syncthreads();
    for (int it = 0; it < items; it += 2) {
        phase(h0, h1, it);
        if (it + 1 < items) phase(h1, h0, it + 1);
    }
    // partial[group][column of H1c][c]: c runs along the lanes, so every store instruction is one 256-byte run
    if (c_ok) {
        float* dst = partial + ((long long)gr * ldh + slab * DW_COLS) * d + c;
#pragma unroll
        for (int k = 0; k < DW_COLS; ++k) dst[(long long)k * d] = acc[k];
    }
}

// out[c][h] = sum over groups of partial[group][h][c], groups in order; 32 x 32 tiles transposed through LDS so
// that reads and writes are both whole 128-byte rows.  h_valid <= H columns of each output row are stored, ld_out
// apart (the first layer's gradient keeps 64 rows of partials whatever n is).  Two jobs share a launch: the
// first `blocks` blocks take job 0, the rest job 1 (a job with no blocks is skipped).
struct DwReduceJob {
    const float* partial;
    int d, H;
    float* out;
    int ld_out, h_valid;
    int bx, blocks;  // blocks along c; bx * ceil(H / 32) in all
};

__global__ __launch_bounds__(256) void dw_reduce_pair_kernel(const DwReduceJob j0, const DwReduceJob j1)
{
    __shared__ float tile[32][33];
    int b = blockIdx.x;
    const bool first = b < j0.blocks;  // block-uniform
    if (!first) b -= j0.blocks;
    const float* __restrict__ partial = first ? j0.partial : j1.partial;
    float* __restrict__ out = first ? j0.out : j1.out;
    const int d = first ? j0.d : j1.d, H = first ? j0.H : j1.H;
    const int ld_out = first ? j0.ld_out : j1.ld_out, h_valid = first ? j0.h_valid : j1.h_valid;
    const int bx = first ? j0.bx : j1.bx;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = (b % bx) * 32, h0 = (b / bx) * 32;
    const long long plane = (long long)H * d;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int h = h0 + ty + 8 * i, c = c0 + tx;
        // all groups' loads in flight at once (clamped indices: H is a multiple of 32, d need not be)
        const float* src = partial + (long long)min(h, H - 1) * d + min(c, d - 1);
        float v[DW_GROUPS];
#pragma unroll
        for (int g = 0; g < DW_GROUPS; ++g) v[g] = src[g * plane];
        float sum = v[0];
#pragma unroll
        for (int g = 1; g < DW_GROUPS; ++g) sum += v[g];
        tile[ty + 8 * i][tx] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, h = h0 + tx;
        if (c < d && h < h_valid) out[(long long)c * ld_out + h] = tile[tx][ty + 8 * i];
    }
}

// The same sums for widths d that are multiples of four (the step's): a lane reads four neighbouring c with one
// 16-byte load, so a wave instruction covers four whole 256-byte runs instead of two 128-byte ones and a block
// issues a quarter of the load instructions; tiles of 64 (c) x 16 (h), the groups in two batches of sixteen loads
// per lane, added in group order as above (bit-identical results).
constexpr int RV_C = 64, RV_H = 16;

// The first sj.blocks() blocks of the launch run side jobs (side_jobs.h): the encoder's bias gradients, column sums
// that are final before this launch and that nothing in it reads.
__global__ __launch_bounds__(256) void dw_reduce_pair_vec_kernel(const DwReduceJob j0, const DwReduceJob j1,
                                                                 const SideJobs sj)
{
    static_assert(SIDE_SMEM_FLOATS >= RV_H * (RV_C + 1), "the transposition tile lives in the side jobs' LDS");
    __shared__ __attribute__((aligned(16))) float smem[SIDE_SMEM_FLOATS];
    float (*const tile)[RV_C + 1] = reinterpret_cast<float (*)[RV_C + 1]>(smem);
    const int ns = sj.first_block[SIDE_MAX_JOBS];
    if ((int)blockIdx.x < ns) {  // block-uniform
        side_run<256>(sj, (int)blockIdx.x, smem);
        return;
    }
    int b = (int)blockIdx.x - ns;
    const bool first = b < j0.blocks;  // block-uniform
    if (!first) b -= j0.blocks;
    const float* __restrict__ partial = first ? j0.partial : j1.partial;
    float* __restrict__ out = first ? j0.out : j1.out;
    const int d = first ? j0.d : j1.d, H = first ? j0.H : j1.H;
    const int ld_out = first ? j0.ld_out : j1.ld_out, h_valid = first ? j0.h_valid : j1.h_valid;
    const int bx = first ? j0.bx : j1.bx;
    const int c0 = (b % bx) * RV_C, h0 = (b / bx) * RV_H;
    const long long plane = (long long)H * d;
    {
        const int cx = threadIdx.x & 15, hy = threadIdx.x >> 4;
        // (clamped indices: H is a multiple of 16, d a multiple of 4 but not of 64)
        const float* src = partial + (long long)min(h0 + hy, H - 1) * d + min(c0 + 4 * cx, d - 4);
        f32x4 v[DW_GROUPS / 2];
#pragma unroll
        for (int g = 0; g < DW_GROUPS / 2; ++g) v[g] = *reinterpret_cast<const f32x4*>(src + g * plane);
        f32x4 sum = v[0];
#pragma unroll
        for (int g = 1; g < DW_GROUPS / 2; ++g) sum += v[g];
#pragma unroll
        for (int g = 0; g < DW_GROUPS / 2; ++g) v[g] = *reinterpret_cast<const f32x4*>(src + (g + DW_GROUPS / 2) * plane);
#pragma unroll
        for (int g = 0; g < DW_GROUPS / 2; ++g) sum += v[g];
#pragma unroll
        for (int q = 0; q < 4; ++q) tile[hy][4 * cx + q] = sum[q];
    }
    __syncthreads();
    const int hx = threadIdx.x & 15, cy = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < RV_C / 16; ++i) {
        const int c = c0 + cy + 16 * i, h = h0 + hx;
        if (c < d && h < h_valid) out[(long long)c * ld_out + h] = tile[hx][cy + 16 * i];
    }
}

inline DwReduceJob dw_reduce_job(const float* partial, int d, int H, float* out, int ld_out, int h_valid)
{
    DwReduceJob j;
    j.partial = partial; j.d = d; j.H = H; j.out = out; j.ld_out = ld_out; j.h_valid = h_valid;
    j.bx = (d + 31) / 32;
    j.blocks = j.bx * ((H + 31) / 32);
    return j;
}

// `side` (nullable): side jobs that ride in front of the reduction's blocks (a launch of their own where the
// reduction runs the scalar kernel).
int dw_reduce_pair(const DwReduceJob& j0, const DwReduceJob& j1, hipStream_t s, const SideJobs* side = nullptr)
{
    SideJobs sj;
    if (side != nullptr) sj = *side;
    auto vec_ok = [](const DwReduceJob& j) {
        return j.blocks == 0 || (j.d % 4 == 0 && j.d >= 4 && j.H % RV_H == 0 && (reinterpret_cast<uintptr_t>(j.partial) & 15) == 0);
    };
    if (vec_ok(j0) && vec_ok(j1)) {
        DwReduceJob a = j0, b = j1;
        for (DwReduceJob* j : {&a, &b}) {
            if (j->blocks == 0) continue;
            j->bx = (j->d + RV_C - 1) / RV_C;
            j->blocks = j->bx * (j->H / RV_H);
        }
        hipLaunchKernelGGL(dw_reduce_pair_vec_kernel, dim3(a.blocks + b.blocks + sj.blocks()), dim3(256), 0, s, a, b, sj);
        return check_launch("dw_reduce_pair_vec_kernel");
    }
    if (sj.blocks() > 0)
        if (int rc = side_jobs_launch(sj, s)) return rc;
    hipLaunchKernelGGL(dw_reduce_pair_kernel, dim3(j0.blocks + j1.blocks), dim3(256), 0, s, j0, j1);
    return check_launch("dw_reduce_pair_kernel");
}

// ---------------------------------------------------------------------------
// dW1 through the adjacency's sparsity.  With identity node features the first layer is Z1 = A W1^T + b1, so
//   dW1[c][i] = sum over stored rows r of dZ1c[r][c] * adjc[r][i]
// and a row of adjc has deg + 1 nonzeros out of n <= 64: the dense product multiplies mostly zeros and is
// bound by its one pass over dZ1c.  Here that pass is all there is.  A block owns 64 columns c (one per lane; or
// 128, two per lane) and a chunk of the stored rows; each of its eight waves takes every eighth row of the chunk: the lane loads
// dZ1c[r][c] (one 256-byte run per wave) and, as lane k, adjc[r][k]; a ballot gives the nonzero columns and the
// wave walks them with a scalar loop (v_readlane for the coefficient), adding coefficient * dZ1c[r][c] to the
// accumulator of column i -- 64 accumulators per lane, indexed by a scalar.  D1_UNROLL rows are in flight per
// wave, the next batch requested before the current one is walked.  The eight waves' accumulators are added in
// a fixed order through LDS and the block stores partial[chunk][i][c] (c along the lanes: 256-byte runs);
// dw_reduce_pair_kernel adds the chunks in order and transposes: reproducible.
// ---------------------------------------------------------------------------
constexpr int D1_THREADS = 512;
constexpr int D1_WAVES = D1_THREADS / 64;
constexpr int D1_UNROLL = 8;
constexpr int D1_CHUNKS = DW_GROUPS;  // the reduce kernel adds DW_GROUPS partials

// NC columns per lane (1 or 2).  The walk is bound by the scalar instructions it issues per nonzero (find-first,
// clear, compare, branch, index on / off), so two columns per lane -- one 8-byte load, two multiply-adds under one
// index -- halve them per byte; the block then owns 128 columns and its 128 accumulators per lane leave
// room for two waves per SIMD, each of which issues at its own pace: 29.0 us against 32.0 us for one column per
// lane and four waves per SIMD at 512 graphs of 64 nodes, 2d = 1024 (the dense product: 41 us + 5 us).
template <int NC>
__global__ __launch_bounds__(D1_THREADS, NC == 1 ? 4 : 2) void dw1_sparse_kernel(const float* __restrict__ adjc, int n,
                                                                             const float* __restrict__ dz, int ldz,
                                                                             const int* __restrict__ rows_dev,
                                                                             int rows_max, float* __restrict__ partial)
{
    typedef float xvec __attribute__((ext_vector_type(NC)));
    __shared__ float red[D1_WAVES / 2][64][64 * NC];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows = min(rows_dev[0], rows_max);
    const int per = (rows + D1_CHUNKS - 1) / D1_CHUNKS;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    const int col = blockIdx.x * 64 * NC + lane * NC;
    const float* const zc = dz + col;
    const float* const ac = adjc + min(lane, n - 1);
    const unsigned long long lanes_n = n >= 64 ? ~0ull : (1ull << n) - 1;

    // The accumulators of the lane's first column sit in v[0:63] and those of its second in v[64:127]
    // (physical-register constraints on the statement that updates them), so one scalar index reaches all of
    // them; as plain arrays the compiler keeps half of them in scratch memory.
    f32x32 lo0, hi0, lo1, hi1;
#pragma unroll
    for (int k = 0; k < 32; ++k) lo0[k] = hi0[k] = lo1[k] = hi1[k] = 0.f;

    struct Batch {
        xvec x[D1_UNROLL];
        float a[D1_UNROLL];
    };
    // (clamped rows instead of a branch around the loads: a row past the chunk is read again with zero
    // coefficients, and the loads of a batch are all issued together)
    auto fetch = [&](Batch& b, int r) {
#pragma unroll
        for (int u = 0; u < D1_UNROLL; ++u) {
            const int rr = r + D1_WAVES * u;
            const int rc = max(min(rr, r1 - 1), 0);
            // 32-bit offsets: rows * ldz < 2^31 is checked on the host
            b.x[u] = *reinterpret_cast<const xvec*>(zc + rc * ldz);
            b.a[u] = ac[rc * n];
        }
    };
    // (the rows past the chunk and the lanes past n are taken out of the ballot, not out of the loaded value: a
    // select on the value makes the compiler branch around the load and wait for it on the spot)
    auto walk = [&](const Batch& b, int r) {
#pragma unroll
        for (int u = 0; u < D1_UNROLL; ++u) {
            unsigned long long mask = __ballot(b.a[u] != 0.f) & (r + D1_WAVES * u < r1 ? lanes_n : 0ull);
            while (mask) {
                const int k = __builtin_ctzll(mask);
                mask &= mask - 1;
                const float ak = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b.a[u]), k));
                // v[k] += ak * x: the index applies to the accumulator as source and as destination, not to
                // the scalar coefficient nor to x (SRC1 is not indexed)
                if constexpr (NC == 1) {
                    asm volatile("s_set_gpr_idx_on %3, gpr_idx(SRC2,DST)\n\t"
                                 "v_fma_f32 v0, %2, %4, v0\n\t"
                                 "s_set_gpr_idx_off"
                                 : "+{v[0:31]}"(lo0), "+{v[32:63]}"(hi0)
                                 : "s"(ak), "s"(k), "v"(b.x[u][0]));
                } else {
                    asm volatile("s_set_gpr_idx_on %5, gpr_idx(SRC2,DST)\n\t"
                                 "v_fma_f32 v0, %4, %6, v0\n\t"
                                 "v_fma_f32 v64, %4, %7, v64\n\t"
                                 "s_set_gpr_idx_off"
                                 : "+{v[0:31]}"(lo0), "+{v[32:63]}"(hi0), "+{v[64:95]}"(lo1), "+{v[96:127]}"(hi1)
                                 : "s"(ak), "s"(k), "v"(b.x[u][0]), "v"(b.x[u][NC - 1]));
                }
            }
        }
    };
    // NB batches rotate: NB - 1 are in flight while one is walked (four instead of two, which the two-column
    // form has the registers for, measured slower: 32.6 against 29.0 us -- the walk is bound by the instructions
    // a wave issues, not by the latency of its loads).
    constexpr int NB = 2;
    Batch b[NB];
    constexpr int STEP = D1_WAVES * D1_UNROLL;
    int r = r0 + wave;
    if (r0 < r1) {
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) fetch(b[i], r + i * STEP);
        for (; r < r1; r += NB * STEP) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                fetch(b[(i + NB - 1) % NB], r + (i + NB - 1) * STEP);
                walk(b[i], r + i * STEP);
            }
        }
    }
    // waves 4..7 -> 0..3, 2..3 -> 0..1, 1 -> 0
#pragma unroll
    for (int half = D1_WAVES / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
            float* o = &red[wave - half][0][lane * NC];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                o[k * 64 * NC] = lo0[k];
                o[(k + 32) * 64 * NC] = hi0[k];
                if constexpr (NC == 2) {
                    o[k * 64 * NC + 1] = lo1[k];
                    o[(k + 32) * 64 * NC + 1] = hi1[k];
                }
            }
        }
        __syncthreads();
        if (wave < half) {
            const float* o = &red[wave][0][lane * NC];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                lo0[k] += o[k * 64 * NC];
                hi0[k] += o[(k + 32) * 64 * NC];
                if constexpr (NC == 2) {
                    lo1[k] += o[k * 64 * NC + 1];
                    hi1[k] += o[(k + 32) * 64 * NC + 1];
                }
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* dst = partial + (long long)blockIdx.y * 64 * ldz + col;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if constexpr (NC == 1) {
                dst[k * ldz] = lo0[k];
                dst[(k + 32) * ldz] = hi0[k];
            } else {
                xvec a, b;
                a[0] = lo0[k]; a[NC - 1] = lo1[k];
                b[0] = hi0[k]; b[NC - 1] = hi1[k];
                *reinterpret_cast<xvec*>(dst + k * ldz) = a;
                *reinterpret_cast<xvec*>(dst + (k + 32) * ldz) = b;
            }
        }
    }
}

}  // namespace

int conv1_sparse(const float* adjc, int n, const float* W1, const float* b1, int H, const int* rows_dev, int rows_max,
                 float* h1c, float* absmax, const float* w_next, long long w_next_count, hipStream_t s)
{
    if (n < 1 || n > 64 || (H & 3) != 0) return set_error(GTG_ERR_INVALID, "conv1_sparse: n in [1, 64], H a multiple of 4");
    const int slabs = (H + C1_COLS - 1) / C1_COLS;
    int chunks = 512 / slabs;  // two blocks of 512 threads per CU
    if (chunks < 1) chunks = 1;
    ProfScope prof(s, "conv1_sparse", rows_max, H, n);
    if (slabs * chunks > GEMM_ABSMAX_PARTS) absmax = nullptr;  // (never: at most 512 blocks)
    hipLaunchKernelGGL(conv1_sparse_kernel, dim3(slabs * chunks), dim3(C1_THREADS), 0, s, adjc, n, W1, b1, H, rows_dev,
                       rows_max, slabs, h1c, absmax, w_next, w_next_count);
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

bool dw2_sparse_applies(int64_t B, int n, int d, int H)
{
    // small batches are launch-bound either way and keep the dense product
    return n >= 1 && n <= 64 && (H % DW_COLS) == 0 && B * (int64_t)d >= 65536 && (B + DW_GROUPS - 1) / DW_GROUPS + (n + DW_GROUPS - 1) / DW_GROUPS <= 64 &&
           (B * n + n) * (int64_t)H < (1ll << 31);
}

size_t dw2_sparse_workspace(int d, int H) { return (size_t)DW_GROUPS * d * H * sizeof(float); }

int dw2_sparse(const float* adj, const float* h1c, int H, const float* dpool, const int32_t* argmax, const int32_t* meta,
               const float* dT2c, const float* proto_rows, int64_t B, int n, int d, float* partial, float* dW2,
               bool defer_reduce, hipStream_t s)
{
    if (!dw2_sparse_applies(B, n, d, H)) return set_error(GTG_ERR_INVALID, "dw2_sparse: shape not supported");
    if ((reinterpret_cast<uintptr_t>(partial) & 15) != 0 || (reinterpret_cast<uintptr_t>(dW2) & 15) != 0)
        return set_error(GTG_ERR_INVALID, "dw2_sparse: outputs must be 16-byte aligned");
    ProfScope prof(s, "dw2_sparse", B * n, H, d);
    hipLaunchKernelGGL(dw2_sparse_kernel, dim3(H / DW_COLS * DW_GROUPS, 1, (d + DW_CPB - 1) / DW_CPB), dim3(DW_THREADS), 0, s,
                       adj, h1c, H, dpool, argmax, meta, dT2c, proto_rows, (int)B, n, d, partial);
    if (int rc = check_launch("dw2_sparse_kernel")) return rc;
    if (defer_reduce) return GTG_OK;
    DwReduceJob none = dw_reduce_job(nullptr, 1, 32, nullptr, 1, 0);
    none.blocks = 0;
    return dw_reduce_pair(dw_reduce_job(partial, d, H, dW2, H, H), none, s);
}

bool dw1_sparse_applies(int64_t rows, int n, int H)
{
    // small problems are launch-bound either way and keep the dense product
    return n >= 1 && n <= 64 && (H % 64) == 0 && rows >= 2048 && rows * (int64_t)H < (1ll << 31);
}

size_t dw1_sparse_workspace(int H) { return (size_t)D1_CHUNKS * 64 * H * sizeof(float); }

int dw1_sparse(const float* adjc, int n, const float* dZ1c, int H, const int* rows_dev, int64_t rows_max, float* partial,
               float* dW1, const float* dw2_partial, int d, float* dW2, hipStream_t s, const SideJobs* side)
{
    if (!dw1_sparse_applies(rows_max, n, H)) return set_error(GTG_ERR_INVALID, "dw1_sparse: shape not supported");
    ProfScope prof(s, "dw1_sparse", rows_max, H, n);
    // two columns per lane where that still leaves a block per CU (H / 128 slabs x 32 chunks)
    const bool two = (H % 128) == 0 && H >= 1024 && (reinterpret_cast<uintptr_t>(dZ1c) & 7) == 0 &&
                     (reinterpret_cast<uintptr_t>(partial) & 7) == 0;
    if (two)
        hipLaunchKernelGGL(dw1_sparse_kernel<2>, dim3(H / 128, D1_CHUNKS), dim3(D1_THREADS), 0, s, adjc, n, dZ1c, H,
                           rows_dev, (int)rows_max, partial);
    else
        hipLaunchKernelGGL(dw1_sparse_kernel<1>, dim3(H / 64, D1_CHUNKS), dim3(D1_THREADS), 0, s, adjc, n, dZ1c, H,
                           rows_dev, (int)rows_max, partial);
    if (int rc = check_launch("dw1_sparse_kernel")) return rc;
    // partial[chunk][i][c] -> dW1[c][i]; the deferred reduction of dW2's partials rides in the same launch
    DwReduceJob j2 = dw_reduce_job(dw2_partial, d, H, dW2, H, H);
    if (dw2_partial == nullptr) j2.blocks = 0;
    return dw_reduce_pair(j2, dw_reduce_job(partial, H, 64, dW1, n, n), s, side);
}

}  // namespace gtg
