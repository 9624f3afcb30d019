// One-layer LSTM (torch.nn.LSTM semantics: gate order i, f, g, o; c' = f c + i g, h' = o tanh(c')), forward
// and backward over a whole padded sequence, as the text branch runs it (/root/reference/src/text_vae.py:37-40,
// 54-66: nn.LSTM(512, 1024, 1), bidirectional for the encoder).
//
// Work split:
//   * everything that is one large product over all T steps goes through gemm(): the input projection
//     X W_ih^T + (b_ih + b_hh) before the recurrence, and dW_ih, dW_hh, dX after the backward recurrence;
//   * the recurrence itself is one launch per time step.  Forward: gates = h_{t-1} W_hh^T (a (B x H) x (H x 4H)
//     product, far too short in M to fill the large-tile GEMM) with the LSTM cell as its epilogue.  Operands
//     are three bf16 planes in the k-blocked layout [plane][K/16][rows][16] (the exact split x = x0 + x1 + x2,
//     six products per multiply-add, fp32 accumulate): W_hh is split once per call, and each step's epilogue
//     writes h_t directly as the planes the next step reads, so no step spends an instruction on splitting.
//     A 64 x 64 tile per block: 64 batch rows x (16 hidden units x 4 gates) -- the rows of W_hh are permuted
//     so that the four gates of a unit land in one tile and the cell needs no second pass -- which gives
//     (H/16) x ceil(B/64) blocks, 256 for H = 1024, B = 256: one per CU; the block's four waves split K.
//   * backward, per step: a pointwise kernel turns (dh_t, dc_t) into the four gate gradients (fp32 for the
//     large products afterwards, planes for the recurrence), then dh_{t-1} += dG_t W_hh with K = 4H cut into
//     slabs so that the launch again has a block per CU; the slabs are summed by the next step's pointwise
//     kernel.
//   * GTG_MATRIX_F16X2 (hidden sizes up to 1024): the recurrence's operands are two fp16 planes instead, every
//     operand row scaled by a power of two (h: |h| < 1, so 2^13 for every row; an initial state and the gate
//     gradients: from the row's largest magnitude, found by the kernel that writes the planes; W_hh: one scale
//     from its largest magnitude), three products per multiply-add, the product scaled back in the epilogue:
//     the same error with half the matrix instructions and two thirds of the operand bytes.
#include "gtg_internal.h"

namespace gtg {

namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float H_SCALE = 8192.f, H_INV = 1.f / 8192.f;  // fp16 planes of a hidden state: |h| < 1
constexpr int LT = 64;                        // tile edge (rows of A, rows of B)
constexpr int LKS = 16;                       // k extent of one k block of the plane layout
constexpr int LDEPTH = 4;                     // register sets of prefetched fragments per wave
constexpr int LDC = LT + 4;                   // row pitch of the accumulator tile staged for the epilogue

struct StepArgs {
    const __bf16* A;      // [3][K/16][rowsA][16] bf16, or [2][K/16][rowsA][16] fp16
    const __bf16* B;      // [3][K/16][rowsB][16]
    long long planeA, planeB;
    int rowsA, rowsB;
    int M, N;             // live rows of A (batch rows), rows of B (output columns)
    int kb_per_split;     // k blocks per blockIdx.z (0: the product is skipped, the accumulators stay zero)
    int H;
    // forward epilogue
    float* gates;         // (M, 4H): in: this step's input projection with both biases; out: the gate activations
    const float* c_prev;  // (M, H) or null (zero state)
    float* h_out;         // (M, H)
    float* c_out;         // (M, H)
    __bf16* h_planes;     // [3][H/16][rows_h][16]: h_t as the next step's A operand
    long long plane_h;
    int rows_h;
    // backward epilogue
    float* slabs;         // [gridDim.z][M][N]
    // fp16 planes: the product is multiplied by A's inverse row scale (per row, or one value for all when the
    // array is null) and by B's inverse scale (a device scalar)
    const float* inv_a_rows;
    float inv_a_const;
    const float* inv_b;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// four fp32 -> the four bf16 of plane 0, 1, 2 (truncation split: each plane the top 16 bits of what is left)
__device__ __forceinline__ void split4(const f32x4 x, u32x2& p0, u32x2& p1, u32x2& p2)
{
#pragma unroll
    for (int q = 0; q < 2; ++q) {
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

// store x (four consecutive k values k0 .. k0+3 of row `row`) into a k-blocked plane image
__device__ __forceinline__ void store_planes4(__bf16* img, long long plane, int rows, int row, int k0, const f32x4 x)
{
    u32x2 p0, p1, p2;
    split4(x, p0, p1, p2);
    __bf16* dst = img + ((long long)(k0 >> 4) * rows + row) * 16 + (k0 & 15);
    *reinterpret_cast<u32x2*>(dst) = p0;
    *reinterpret_cast<u32x2*>(dst + plane) = p1;
    *reinterpret_cast<u32x2*>(dst + 2 * plane) = p2;
}

// the same for two fp16 planes of x * scale: the value rounded to nearest, then what is left rounded to nearest
__device__ __forceinline__ void store_planes4h(__bf16* img, long long plane, int rows, int row, int k0, const f32x4 x, float scale)
{
    u32x2 p0, p1;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        unsigned hi, lo;
        split_pair_f16(x[2 * q], x[2 * q + 1], scale, hi, lo);
        p0[q] = hi;
        p1[q] = lo;
    }
    __bf16* dst = img + ((long long)(k0 >> 4) * rows + row) * 16 + (k0 & 15);
    *reinterpret_cast<u32x2*>(dst) = p0;
    *reinterpret_cast<u32x2*>(dst + plane) = p1;
}

// C tile (64 x 64) = A rows [m0, m0+64) x B rows [n0, n0+64) over k blocks [kb0, kb0 + nk).
//
// A block is alone on its CU (a block per CU fills the chip) with one wave per SIMD, so nothing but the wave
// itself can hide its latencies.  Hence: the four waves split K, not the tile.  Wave w multiplies the whole
// 64 x 64 tile (four 32 x 32 accumulators) over k blocks w, w + 4, ...  Nothing is shared between waves until
// the end, so the operands skip LDS: in the k-blocked plane layout the 16 bytes lane (r, h) feeds to an MFMA
// (row r, k half h) of 32 consecutive rows are one contiguous kilobyte, i.e. an MFMA operand is ONE coalesced
// 16-byte-per-lane load straight into registers.  LDEPTH register sets of fragments rotate: LDEPTH - 1 k blocks
// (12 KB each per wave) are in flight while the 24 MFMAs of one run, consecutive MFMAs go to different
// accumulators, and the loop holds no barrier and no LDS access.  The four partial tiles are summed through
// LDS by the epilogue.  FWD: the LSTM cell as epilogue; else raw slabs.
// NPL = 3: bf16 planes, six products; NPL = 2: fp16 planes, three products and the scales in the epilogue.
template <int NPL>
struct Frags {
    bf16x8 a[2][NPL], b[2][NPL];  // [32-row block][plane]
};

template <bool FWD, int NPL>
__global__ __launch_bounds__(256, 1) void lstm_step_kernel(const StepArgs a)
{
    constexpr int WAVE_LDS = LT * LDC * 4;  // a wave's partial tile
    __shared__ __attribute__((aligned(16))) char lds[4 * WAVE_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the pipeline's branches are per wave
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * LT, m0 = blockIdx.y * LT;
    const int nk = a.kb_per_split, kb0 = blockIdx.z * nk;
    const int row = tid >> 2, gr = m0 + row;  // the epilogue's row of this thread

    // FWD: the cell's inputs are fetched now and used after the product
    f32x4 pre[4], cp = {0.f, 0.f, 0.f, 0.f};
    const int u0 = (tid & 3) * 4, unit0 = blockIdx.x * 16 + u0;
    float* const grow = FWD ? a.gates + (long long)min(gr, a.M - 1) * 4 * a.H + unit0 : nullptr;
    if (FWD) {
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = *reinterpret_cast<const f32x4*>(grow + (long long)g * a.H);
        // unconditional (a null c_prev reads a dummy that the epilogue discards): a branch here would make the
        // wave wait for the load before it starts its pipeline
        const float* cpp = a.c_prev ? a.c_prev + (long long)min(gr, a.M - 1) * a.H + unit0 : grow;
        asm volatile("" : "+v"(cpp));  // keeps the load unconditional
        cp = *reinterpret_cast<const f32x4*>(cpp);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int mine = nk > wave ? (nk - wave + 3) / 4 : 0;  // k blocks of this wave
    if (mine > 0) {
        // Rows past the operand's last one are clamped (their products land in rows / columns the epilogue never
        // stores), k blocks past the wave's last one likewise (fetched again, never multiplied).
        const __bf16 *pA[2], *pB[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            pA[i] = a.A + ((long long)(kb0 + wave) * a.rowsA + min(m0 + 32 * i + r, a.M - 1)) * LKS + h * 8;
            pB[i] = a.B + ((long long)(kb0 + wave) * a.rowsB + min(n0 + 32 * i + r, a.N - 1)) * LKS + h * 8;
        }
        const long long stepA = 4ll * a.rowsA * LKS, stepB = 4ll * a.rowsB * LKS;  // to this wave's next k block
        auto load = [&](int j, Frags<NPL>& f) {
            const int jj = min(j, mine - 1);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f.a[i][pl] = *reinterpret_cast<const bf16x8*>(pA[i] + pl * a.planeA + jj * stepA);
                    f.b[i][pl] = *reinterpret_cast<const bf16x8*>(pB[i] + pl * a.planeB + jj * stepB);
                }
        };
        // products of plane pairs (pa, pb) with pa + pb < NPL, the small ones first
        auto mma = [&](const Frags<NPL>& f) {
#pragma unroll
            for (int sum = NPL - 1; sum >= 0; --sum)
#pragma unroll
                for (int pa = 0; pa <= sum; ++pa)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int jn = 0; jn < 2; ++jn)
                            if constexpr (NPL == 2)
                                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.a[i][pa]),
                                                                                    __builtin_bit_cast(f16x8, f.b[jn][sum - pa]),
                                                                                    acc[i][jn], 0, 0, 0);
                            else
                                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][pa], f.b[jn][sum - pa], acc[i][jn], 0, 0, 0);
        };
        // LDEPTH register sets rotate: LDEPTH - 1 k blocks are in flight while one is multiplied
        Frags<NPL> f[LDEPTH];
#pragma unroll
        for (int d = 0; d < LDEPTH - 1; ++d) load(d, f[d]);
        int j = 0;
        // One sub-step: fetch the set that was multiplied last, multiply the oldest one.  (The scheduler must
        // not move the loads down among the MFMAs: left to itself it issued the loads a set needs right before
        // the loop's back edge.)
        auto substep = [&](int jj, int u) {
            load(jj + LDEPTH - 1 + u, f[(LDEPTH - 1 + u) % LDEPTH]);
            __builtin_amdgcn_sched_barrier(0);
            mma(f[u]);
            __builtin_amdgcn_sched_barrier(0);
        };
        if (mine == 4 * LDEPTH) {
            // the common depth (hidden size 1024: 64 k blocks over four waves) as straight-line code: at a loop
            // header the compiler waits for EVERY load in flight, the set just requested included, which
            // exposes a full memory latency per trip; without the loop it counts the loads and waits for the
            // oldest set only
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < LDEPTH; ++u) substep(t * LDEPTH, u);
            j = 4 * LDEPTH;
        }
        for (; j + LDEPTH - 1 < mine; j += LDEPTH) {
#pragma unroll
            for (int u = 0; u < LDEPTH; ++u) substep(j, u);
        }
        // at most LDEPTH - 1 blocks are left, and they are the ones sets 0 .. LDEPTH - 2 hold
#pragma unroll
        for (int u = 0; u < LDEPTH - 1; ++u)
            if (j + u < mine) mma(f[u]);
    }

    // each wave leaves its partial tile in LDS (accumulator register e of lane (r, h) is element (row
    // (e & 3) + 8 (e >> 2) + 4 h, column r) of a 32 x 32 block); the four are summed in wave order
    float* const part = reinterpret_cast<float*>(lds + wave * WAVE_LDS);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) part[(32 * i + (e & 3) + 8 * (e >> 2) + 4 * h) * LDC + 32 * jn + r] = acc[i][jn][e];
    __syncthreads();
    if (gr >= a.M) return;
    float inv_a = 1.f, inv_b = 1.f;
    if constexpr (NPL == 2) {
        inv_a = a.inv_a_rows ? a.inv_a_rows[gr] : a.inv_a_const;
        inv_b = a.inv_b[0];
    }
    auto tile4 = [&](int col) {  // four consecutive columns of this thread's row, summed over the waves
        f32x4 v = *reinterpret_cast<const f32x4*>(lds + (row * LDC + col) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(lds + w * WAVE_LDS + (row * LDC + col) * 4);
        if constexpr (NPL == 2) v = (v * inv_a) * inv_b;  // powers of two: exact
        return v;
    };
    if (FWD) {
        // tile column g * 16 + u is gate g of hidden unit blockIdx.x * 16 + u; a thread takes four units of a row
        const int H = a.H;
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] += tile4(g * 16 + u0);
        if (a.c_prev == nullptr) cp = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 gi, gf, gg, go, c, hh;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gi[j] = sigmoidf_(pre[0][j]);
            gf[j] = sigmoidf_(pre[1][j]);
            gg[j] = tanhf(pre[2][j]);
            go[j] = sigmoidf_(pre[3][j]);
            c[j] = gf[j] * cp[j] + gi[j] * gg[j];
            hh[j] = go[j] * tanhf(c[j]);
        }
        *reinterpret_cast<f32x4*>(grow) = gi;
        *reinterpret_cast<f32x4*>(grow + (long long)H) = gf;
        *reinterpret_cast<f32x4*>(grow + 2ll * H) = gg;
        *reinterpret_cast<f32x4*>(grow + 3ll * H) = go;
        *reinterpret_cast<f32x4*>(a.c_out + (long long)gr * H + unit0) = c;
        *reinterpret_cast<f32x4*>(a.h_out + (long long)gr * H + unit0) = hh;
        if constexpr (NPL == 2)
            store_planes4h(a.h_planes, a.plane_h, a.rows_h, gr, unit0, hh, H_SCALE);
        else
            store_planes4(a.h_planes, a.plane_h, a.rows_h, gr, unit0, hh);
    } else {
        const int c0 = (tid & 3) * 16;
        float* out = a.slabs + ((long long)blockIdx.z * a.M + gr) * a.N + n0 + c0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (n0 + c0 + 4 * q < a.N) *reinterpret_cast<f32x4*>(out + 4 * q) = tile4(c0 + 4 * q);
    }
}

// W_hh (4H, H) fp32 -> one of the two plane images the recurrence multiplies by.
//   forward (BACK false): B operand of h W_hh^T, rows permuted so that a 64-row tile holds the four gates of 16
//                  units: [pl][H/16][4H rows p][16], p = (u / 16) * 64 + g * 16 + u % 16  <-  W_hh[g H + u][k]
//   backward (BACK true): B operand of dG W_hh (depth = the 4H gate rows in their own order):
//                  [pl][4H/16][H rows j][16]  <-  W_hh[q][j]
// A block walks tiles of 64 image rows x 16 depth values (one k block): thread t holds row t / 4 and the four
// consecutive depth values (t % 4) * 4 .., so a wave stores 512 contiguous bytes per plane and reads 64-byte runs
// of W_hh (forward: along a row of W_hh; backward: four rows, one element each, the neighbouring threads the
// neighbouring columns).  H is a multiple of 64.  F16: two fp16 planes of W_hh times the power of two its largest
// magnitude calls for (absmax: GEMM_ABSMAX_PARTS partial maxima of |W_hh|); inv_out[0] = its inverse.
template <bool F16, bool BACK>
__global__ __launch_bounds__(256) void lstm_wsplit_kernel(const float* __restrict__ w, int H, unsigned short* __restrict__ dst,
                                                          const float* __restrict__ absmax, float* __restrict__ inv_out)
{
    float scale = 1.f;
    if constexpr (F16) {
        __shared__ float sm[4];
        float m = 0.f;
#pragma unroll
        for (int q = 0; q < GEMM_ABSMAX_PARTS / 256; ++q) m = fmaxf(m, absmax[q * 256 + threadIdx.x]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        float inverse;
        f16x2_scale(m, scale, inverse);
        if (blockIdx.x == 0 && threadIdx.x == 0) inv_out[0] = inverse;
    }
    const long long total = 4ll * H * H;
    const int rl = threadIdx.x >> 2, kk0 = (threadIdx.x & 3) * 4;
    // tiles: (k block, 64-row group); forward: H/16 k blocks x 4H/64 groups, backward: 4H/16 k blocks x H/64 groups
    const int groups = BACK ? H / 64 : 4 * H / 64;
    const int ntiles = (BACK ? 4 * H / 16 : H / 16) * groups;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int kb = tile / groups, grp = tile - kb * groups;
        f32x4 x;
        long long out;   // element index of the thread's four values in a plane
        if constexpr (!BACK) {
            // group = 16 hidden units nb, image rows p = g * 16 + u of it
            const int g = rl >> 4, u = rl & 15;
            x = *reinterpret_cast<const f32x4*>(w + ((long long)g * H + grp * 16 + u) * H + kb * 16 + kk0);
            out = ((long long)kb * 4 * H + grp * 64 + rl) * 16 + kk0;
        } else {
            const int j = grp * 64 + rl;
            const float* src = w + (long long)(kb * 16 + kk0) * H + j;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = src[(long long)e * H];
            out = ((long long)kb * H + j) * 16 + kk0;
        }
        if constexpr (F16) {
            u32x2 p0, p1;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                unsigned short hs[2], ls[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float t = x[2 * q + e] * scale;
                    const _Float16 hi = (_Float16)t, lo = (_Float16)(t - (float)hi);
                    hs[e] = __builtin_bit_cast(unsigned short, hi);
                    ls[e] = __builtin_bit_cast(unsigned short, lo);
                }
                p0[q] = (unsigned)hs[0] | ((unsigned)hs[1] << 16);
                p1[q] = (unsigned)ls[0] | ((unsigned)ls[1] << 16);
            }
            *reinterpret_cast<u32x2*>(dst + out) = p0;
            *reinterpret_cast<u32x2*>(dst + total + out) = p1;
        } else {
            u32x2 p0, p1, p2;
            split4(x, p0, p1, p2);
            *reinterpret_cast<u32x2*>(dst + out) = p0;
            *reinterpret_cast<u32x2*>(dst + total + out) = p1;
            *reinterpret_cast<u32x2*>(dst + 2 * total + out) = p2;
        }
    }
}

// An initial hidden state (rows, cols <= 1024) as the first step's A operand in fp16 planes: one block per row,
// the row scaled by the power of two its largest magnitude calls for; inv_rows[row] = the inverse.
__global__ __launch_bounds__(256) void lstm_h0_planes_f16_kernel(const float* __restrict__ x, int rows, int cols,
                                                                 __bf16* __restrict__ planes, long long plane,
                                                                 float* __restrict__ inv_rows)
{
    __shared__ float sm[4];
    const int row = blockIdx.x, k0 = threadIdx.x * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k0 < cols) v = *reinterpret_cast<const f32x4*>(x + (long long)row * cols + k0);
    float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float scale, inverse;
    f16x2_scale(m, scale, inverse);
    if (threadIdx.x == 0) inv_rows[row] = inverse;
    if (k0 < cols) store_planes4h(planes, plane, rows, row, k0, v, scale);
}

// out = a + b (the two bias vectors), and optionally x (rows, cols) -> planes [3][cols/16][rows][16] (an initial
// hidden state as the first step's A operand); a thread takes four consecutive elements
__global__ __launch_bounds__(256) void lstm_prepare_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                           float* __restrict__ out, const float* __restrict__ x, int rows,
                                                           int cols, __bf16* __restrict__ planes, long long plane)
{
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i < n) {
        const f32x4 va = *reinterpret_cast<const f32x4*>(a + i), vb = *reinterpret_cast<const f32x4*>(b + i);
        *reinterpret_cast<f32x4*>(out + i) = va + vb;
    }
    if (x != nullptr && i < (long long)rows * cols) {
        const int row = (int)(i / cols), k0 = (int)(i % cols);
        store_planes4(planes, plane, rows, row, k0, *reinterpret_cast<const f32x4*>(x + i));
    }
}

// The cell's backward at one step, four hidden units of a row per thread.
//   dh = d_hs (if given) + d_extra (if given) + the Z slabs of dG_{t+1} W_hh (if Z > 0)
//   dc = dc_io (if !first) + d_cT (if first and given) + dh o (1 - tanh(c)^2)
// writes the pre-activation gradients dG (M, 4H) fp32 and as planes [3][4H/16][M][16], and dc_io = dc f.
// F16 (H <= 1024): a block takes whole rows (H / 4 threads each), the planes are two fp16 ones of the row times
// the power of two its largest gate gradient calls for, and inv_rows[row] receives the inverse.
template <bool F16>
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_t,
                                                            const float* __restrict__ c_prev, const float* __restrict__ d_hs,
                                                            const float* __restrict__ d_extra, const float* __restrict__ slabs,
                                                            int Z, const float* __restrict__ d_cT, int first,
                                                            float* __restrict__ dc_io, float* __restrict__ dG,
                                                            __bf16* __restrict__ dG_planes, long long plane, int M, int H,
                                                            float* __restrict__ inv_rows, unsigned* __restrict__ dg_max)
{
    __shared__ unsigned rowmax[16];
    const long long MH = (long long)M * H;
    long long i;
    int row, u0, rl = 0;
    bool active;
    if constexpr (F16) {
        const int tpr = H / 4, rpb = 256 / tpr;  // threads per row, rows per block
        rl = threadIdx.x / tpr;
        row = blockIdx.x * rpb + rl;
        u0 = (threadIdx.x % tpr) * 4;
        active = rl < rpb && row < M;
        i = (long long)row * H + u0;
        if (threadIdx.x < 16) rowmax[threadIdx.x] = 0u;
        __syncthreads();
    } else {
        i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
        active = i < MH;
        row = (int)(i / H);
        u0 = (int)(i % H);
    }
    if (!F16 && !active) return;
    if (!active) {  // (F16) an idle thread reads element 0 and stores nothing: it still meets the barriers
        i = 0;
        row = 0;
        u0 = 0;
    }
    // Every load is issued before the first is used.  Optional inputs are read through a stand-in address and
    // discarded (a branch around a load makes the wave wait for everything in flight at the join).
    auto opt = [&](const float* p) {
        const float* q = p ? p + i : c_t + i;
        asm volatile("" : "+v"(q));
        return *reinterpret_cast<const f32x4*>(q);
    };
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 v_dhs = opt(d_hs), v_ext = opt(d_extra), v_cp = opt(c_prev);
    const f32x4 v_dc = opt(first ? d_cT : dc_io);
    const float* grow = gates + (long long)row * 4 * H + u0;
    const f32x4 gi = *reinterpret_cast<const f32x4*>(grow), gf = *reinterpret_cast<const f32x4*>(grow + (long long)H);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(grow + 2ll * H), go = *reinterpret_cast<const f32x4*>(grow + 3ll * H);
    const f32x4 c = *reinterpret_cast<const f32x4*>(c_t + i);
    f32x4 dh = d_hs ? v_dhs : zero;
    if (d_extra) dh += v_ext;
    if (Z == 4) {  // the usual slab count: four loads in flight
        f32x4 t[4];
#pragma unroll
        for (int z = 0; z < 4; ++z) t[z] = *reinterpret_cast<const f32x4*>(slabs + z * MH + i);
#pragma unroll
        for (int z = 0; z < 4; ++z) dh += t[z];
    } else {
        for (int z = 0; z < Z; ++z) dh += *reinterpret_cast<const f32x4*>(slabs + z * MH + i);
    }
    const f32x4 dc = (first && d_cT == nullptr) ? zero : v_dc;
    const f32x4 cp = c_prev ? v_cp : zero;
    f32x4 di, df, dg, dO, dcp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float tc = tanhf(c[j]);
        const float dcj = dc[j] + dh[j] * go[j] * (1.f - tc * tc);
        dO[j] = dh[j] * tc * go[j] * (1.f - go[j]);
        di[j] = dcj * gg[j] * gi[j] * (1.f - gi[j]);
        df[j] = dcj * cp[j] * gf[j] * (1.f - gf[j]);
        dg[j] = dcj * gi[j] * (1.f - gg[j] * gg[j]);
        dcp[j] = dcj * gf[j];
    }
    if constexpr (F16) {
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) m = fmaxf(m, fmaxf(fmaxf(fabsf(di[j]), fabsf(df[j])), fmaxf(fabsf(dg[j]), fabsf(dO[j]))));
        if (active) atomicMax(&rowmax[rl], __float_as_uint(m));  // non-negative floats order as their bit patterns
        __syncthreads();
        if (!active) return;
        float scale, inverse;
        f16x2_scale(__uint_as_float(rowmax[rl]), scale, inverse);
        if (u0 == 0) {
            inv_rows[row] = inverse;
            // the running maxima of all steps' gate gradients, for the large products that follow the recurrence
            // (GEMM_ABSMAX_PARTS slots, cleared by the caller; non-negative floats order as their bit patterns)
            if (dg_max != nullptr) atomicMax(&dg_max[row & (GEMM_ABSMAX_PARTS - 1)], rowmax[rl]);
        }
        *reinterpret_cast<f32x4*>(dc_io + i) = dcp;
        float* drow = dG + (long long)row * 4 * H + u0;
        *reinterpret_cast<f32x4*>(drow) = di;
        *reinterpret_cast<f32x4*>(drow + (long long)H) = df;
        *reinterpret_cast<f32x4*>(drow + 2ll * H) = dg;
        *reinterpret_cast<f32x4*>(drow + 3ll * H) = dO;
        store_planes4h(dG_planes, plane, M, row, u0, di, scale);
        store_planes4h(dG_planes, plane, M, row, H + u0, df, scale);
        store_planes4h(dG_planes, plane, M, row, 2 * H + u0, dg, scale);
        store_planes4h(dG_planes, plane, M, row, 3 * H + u0, dO, scale);
        return;
    }
    *reinterpret_cast<f32x4*>(dc_io + i) = dcp;
    float* drow = dG + (long long)row * 4 * H + u0;
    *reinterpret_cast<f32x4*>(drow) = di;
    *reinterpret_cast<f32x4*>(drow + (long long)H) = df;
    *reinterpret_cast<f32x4*>(drow + 2ll * H) = dg;
    *reinterpret_cast<f32x4*>(drow + 3ll * H) = dO;
    store_planes4(dG_planes, plane, M, row, u0, di);
    store_planes4(dG_planes, plane, M, row, H + u0, df);
    store_planes4(dG_planes, plane, M, row, 2 * H + u0, dg);
    store_planes4(dG_planes, plane, M, row, 3 * H + u0, dO);
}

// out = the Z slabs summed in order
__global__ __launch_bounds__(256) void lstm_slab_sum_kernel(const float* __restrict__ slabs, int Z, long long n, float* __restrict__ out)
{
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < Z; ++z) s += *reinterpret_cast<const f32x4*>(slabs + z * n + i);
    *reinterpret_cast<f32x4*>(out + i) = s;
}

struct Carve {
    char* base;
    size_t off = 0, cap;
    Carve(void* b, size_t c) : base(reinterpret_cast<char*>(b)), cap(c) {}
    template <typename T>
    T* take(size_t count)
    {
        off = (off + 255) / 256 * 256;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
    bool overflow() const { return off > cap; }
};

inline size_t max2(size_t a, size_t b) { return a > b ? a : b; }
inline size_t min2(size_t a, size_t b) { return a < b ? a : b; }

// K slabs of the backward step product: enough (M / 64) x (H / 64) x Z blocks for a block per CU
inline int bwd_slabs(int B, int H)
{
    const long long tiles = (long long)((B + LT - 1) / LT) * ((H + LT - 1) / LT);
    int z = 1;
    while (z < 16 && tiles * z < 256 && ((4 * H / 16) % (2 * z)) == 0) z *= 2;
    return z;
}

struct FwdWs {
    float* bias;
    unsigned short* wf;
    __bf16* hp[2];
    float* absmax;  // operand maxima of the large products (GTG_MATRIX_F16X2)
    float* w_inv;   // fp16 planes: inverse scale of W_hh (one value), inverse row scales of the initial state
    float* a_inv;
};
void fwd_layout(Carve& c, int B, int H, FwdWs* w)
{
    w->bias = c.take<float>(4 * (size_t)H);
    w->wf = c.take<unsigned short>(12 * (size_t)H * H);
    w->hp[0] = c.take<__bf16>(3 * (size_t)B * H);
    w->hp[1] = c.take<__bf16>(3 * (size_t)B * H);
    w->absmax = c.take<float>(GEMM_ABSMAX_BYTES / sizeof(float));
    w->w_inv = c.take<float>(64);
    w->a_inv = c.take<float>(((size_t)B + 63) / 64 * 64);
}

struct BwdWs {
    unsigned short* wb;
    __bf16* dgp;
    float *slabs, *dc, *dG, *cs, *gemm_slabs;
    float* absmax;  // operand maxima of the large products (GTG_MATRIX_F16X2)
    float* w_inv;   // fp16 planes: inverse scale of W_hh (one value), inverse row scales of a step's gate gradients
    float* a_inv;
};
void bwd_layout(Carve& c, int T, int B, int I, int H, BwdWs* w)
{
    const size_t TB = (size_t)T * B;
    w->wb = c.take<unsigned short>(12 * (size_t)H * H);
    w->dgp = c.take<__bf16>(3 * (size_t)B * 4 * H);
    w->slabs = c.take<float>((size_t)bwd_slabs(B, H) * B * H);
    w->dc = c.take<float>((size_t)B * H);
    w->dG = c.take<float>(TB * 4 * H);
    w->cs = reinterpret_cast<float*>(c.take<char>(colsum_workspace((int64_t)TB, 4 * H)));
    w->gemm_slabs = reinterpret_cast<float*>(c.take<char>(
        max2(gemm_splitk_workspace(4 * H, I, (int)TB, 1), gemm_splitk_workspace(4 * H, H, (int)TB, 1))));
    w->absmax = c.take<float>(GEMM_ABSMAX_BYTES / sizeof(float));
    w->w_inv = c.take<float>(64);
    w->a_inv = c.take<float>(((size_t)B + 63) / 64 * 64);
}

// the recurrence on fp16 planes: that matrix mode, and rows that fit one block of the kernels that scale them
inline bool lstm_f16(int H) { return g_matrix_mode == GTG_MATRIX_F16X2 && H <= 1024; }

}  // namespace

bool lstm_applies(int T, int B, int I, int H)
{
    return T >= 1 && B >= 1 && I >= 4 && I % 4 == 0 && H >= 64 && H % 64 == 0 && (long long)T * B * 4 * H < (1ll << 31);
}

size_t lstm_fwd_workspace(int T, int B, int I, int H)
{
    (void)T; (void)I;
    Carve c(nullptr, 0);
    FwdWs w;
    fwd_layout(c, B, H, &w);
    return c.off;
}

size_t lstm_bwd_workspace(int T, int B, int I, int H)
{
    Carve c(nullptr, 0);
    BwdWs w;
    bwd_layout(c, T, B, I, H, &w);
    return c.off;
}

// x (T, B, I); gates (T, B, 4H), hs / cs (T, B, H) are outputs (gates: the activations, kept for the backward).
// Time runs T-1 .. 0 when `reverse`.  h0 / c0 (B, H) nullable = zero state.
int lstm_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* h0,
             const float* c0, int T, int B, int I, int H, bool reverse, float* hs, float* cs, float* gates, void* ws,
             size_t ws_bytes, hipStream_t s)
{
    if (!lstm_applies(T, B, I, H)) return set_error(GTG_ERR_UNSUPPORTED, "lstm: needs I %% 4 == 0 and H %% 64 == 0 (got I=%d H=%d)", I, H);
    Carve c(ws, ws_bytes);
    FwdWs w;
    fwd_layout(c, B, H, &w);
    if (ws == nullptr || c.overflow()) return set_error(GTG_ERR_WORKSPACE, "lstm_fwd: workspace too small");
    const long long plane_h = (long long)B * H;
    const bool f16 = lstm_f16(H);
    {
        const float* h0_bf16 = f16 ? nullptr : h0;
        const long long n4 = max2((size_t)4 * H, h0_bf16 ? (size_t)B * H : 0) / 4;
        hipLaunchKernelGGL(lstm_prepare_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, b_ih, b_hh, 4 * H, w.bias,
                           h0_bf16, B, H, w.hp[0], plane_h);
        if (int rc = check_launch("lstm_prepare_kernel")) return rc;
        const dim3 wgrid((unsigned)min2((size_t)(H / 16) * (4 * H / 64), (size_t)4096));
        if (f16) {
            if (int rc = absmax_partials(w_hh, 4ll * H, H, H, w.absmax, s)) return rc;
            hipLaunchKernelGGL((lstm_wsplit_kernel<true, false>), wgrid, dim3(256), 0, s, w_hh, H, w.wf, w.absmax, w.w_inv);
            if (h0) hipLaunchKernelGGL(lstm_h0_planes_f16_kernel, dim3(B), dim3(256), 0, s, h0, B, H, w.hp[0], plane_h, w.a_inv);
        } else {
            hipLaunchKernelGGL((lstm_wsplit_kernel<false, false>), wgrid, dim3(256), 0, s, w_hh, H, w.wf, (const float*)nullptr,
                               (float*)nullptr);
        }
        if (int rc = check_launch("lstm_wsplit_kernel")) return rc;
    }
    // the input half of the gates for all steps at once
    GemmDesc g;
    g.A = x; g.lda = I; g.a_kc = true;
    g.B = w_ih; g.ldb = I; g.b_kc = true;
    g.C = gates; g.ldc = 4 * H; g.M = T * B; g.N = 4 * H; g.K = I;
    g.bias = w.bias;
    g.absmax_ws = w.absmax;
    if (int rc = gemm(g, s)) return rc;

    StepArgs a{};
    a.B = reinterpret_cast<const __bf16*>(w.wf);
    a.planeB = 4ll * H * H; a.rowsB = 4 * H;
    a.planeA = plane_h; a.rowsA = B;
    a.M = B; a.N = 4 * H; a.H = H;
    a.plane_h = plane_h; a.rows_h = B;
    a.inv_b = w.w_inv; a.inv_a_const = H_INV;
    const dim3 grid(H / 16, (B + LT - 1) / LT, 1);
    for (int st = 0; st < T; ++st) {
        const int t = reverse ? T - 1 - st : st;
        const int tp = reverse ? t + 1 : t - 1;
        a.A = w.hp[st & 1];
        a.h_planes = w.hp[(st + 1) & 1];
        a.kb_per_split = (st == 0 && h0 == nullptr) ? 0 : H / 16;
        a.gates = gates + (long long)t * B * 4 * H;
        a.c_prev = st == 0 ? c0 : cs + (long long)tp * B * H;
        a.h_out = hs + (long long)t * B * H;
        a.c_out = cs + (long long)t * B * H;
        a.inv_a_rows = (st == 0 && h0 != nullptr) ? w.a_inv : nullptr;  // a given initial state has row scales of its own
        if (f16)
            hipLaunchKernelGGL((lstm_step_kernel<true, 2>), grid, dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((lstm_step_kernel<true, 3>), grid, dim3(256), 0, s, a);
    }
    return check_launch("lstm_step_kernel<fwd>");
}

// d_hs (T, B, H), d_hT, d_cT (B, H): gradients w.r.t. every step's output, the final hidden and the final cell
// state (each nullable = zero).  dx (T, B, I), dh0, dc0 nullable = not wanted.  db (4H) is the gradient of
// both bias vectors.
int lstm_bwd(const float* x, const float* w_ih, const float* w_hh, const float* h0, const float* c0, const float* hs,
             const float* cs, const float* gates, const float* d_hs, const float* d_hT, const float* d_cT, int T, int B,
             int I, int H, bool reverse, float* dx, float* dw_ih, float* dw_hh, float* db, float* dh0, float* dc0, void* ws,
             size_t ws_bytes, hipStream_t s)
{
    if (!lstm_applies(T, B, I, H)) return set_error(GTG_ERR_UNSUPPORTED, "lstm: needs I %% 4 == 0 and H %% 64 == 0 (got I=%d H=%d)", I, H);
    Carve c(ws, ws_bytes);
    BwdWs w;
    bwd_layout(c, T, B, I, H, &w);
    if (ws == nullptr || c.overflow()) return set_error(GTG_ERR_WORKSPACE, "lstm_bwd: workspace too small");
    const long long BH = (long long)B * H, total = 4ll * H * H;
    const bool f16 = lstm_f16(H);
    {
        const dim3 wgrid((unsigned)min2((size_t)(4 * H / 16) * (H / 64), (size_t)4096));
        if (f16) {
            if (int rc = absmax_partials(w_hh, 4ll * H, H, H, w.absmax, s)) return rc;
            hipLaunchKernelGGL((lstm_wsplit_kernel<true, true>), wgrid, dim3(256), 0, s, w_hh, H, w.wb, w.absmax, w.w_inv);
        } else {
            hipLaunchKernelGGL((lstm_wsplit_kernel<false, true>), wgrid, dim3(256), 0, s, w_hh, H, w.wb, (const float*)nullptr,
                               (float*)nullptr);
        }
        if (int rc = check_launch("lstm_wsplit_kernel")) return rc;
        // (W_hh's maxima have been consumed: the slots now collect the gate gradients' maxima step by step)
        if (f16)
            if (int rc = clear_i32(reinterpret_cast<int32_t*>(w.absmax), GEMM_ABSMAX_PARTS, s)) return rc;
    }

    const int Z = bwd_slabs(B, H);
    StepArgs a{};
    a.A = w.dgp; a.planeA = 4ll * BH; a.rowsA = B;
    a.B = reinterpret_cast<const __bf16*>(w.wb); a.planeB = total; a.rowsB = H;
    a.M = B; a.N = H; a.H = H;
    a.kb_per_split = 4 * H / 16 / Z;
    a.slabs = w.slabs;
    a.inv_b = w.w_inv; a.inv_a_rows = w.a_inv; a.inv_a_const = 1.f;
    const dim3 grid((H + LT - 1) / LT, (B + LT - 1) / LT, Z);
    const unsigned cell_blocks = (unsigned)((BH / 4 + 255) / 256);
    const int rows_per_block = 256 / (H / 4 > 256 ? 256 : H / 4);  // F16: whole rows per block
    const unsigned cell_blocks_f16 = (unsigned)((B + rows_per_block - 1) / rows_per_block);
    for (int st = T - 1; st >= 0; --st) {
        const int t = reverse ? T - 1 - st : st;
        const int tp = reverse ? t + 1 : t - 1;
        const bool first = st == T - 1;
        if (f16) {
            hipLaunchKernelGGL(lstm_cell_bwd_kernel<true>, dim3(cell_blocks_f16), dim3(256), 0, s, gates + (long long)t * 4 * BH,
                               cs + t * BH, st == 0 ? c0 : cs + tp * BH, d_hs ? d_hs + t * BH : nullptr, first ? d_hT : nullptr,
                               w.slabs, first ? 0 : Z, d_cT, first ? 1 : 0, w.dc, w.dG + (long long)t * 4 * BH, w.dgp, 4ll * BH, B,
                               H, w.a_inv, reinterpret_cast<unsigned*>(w.absmax));
            if (st > 0 || dh0 != nullptr) hipLaunchKernelGGL((lstm_step_kernel<false, 2>), grid, dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL(lstm_cell_bwd_kernel<false>, dim3(cell_blocks), dim3(256), 0, s, gates + (long long)t * 4 * BH,
                               cs + t * BH, st == 0 ? c0 : cs + tp * BH, d_hs ? d_hs + t * BH : nullptr, first ? d_hT : nullptr,
                               w.slabs, first ? 0 : Z, d_cT, first ? 1 : 0, w.dc, w.dG + (long long)t * 4 * BH, w.dgp, 4ll * BH, B,
                               H, (float*)nullptr, (unsigned*)nullptr);
            if (st > 0 || dh0 != nullptr) hipLaunchKernelGGL((lstm_step_kernel<false, 3>), grid, dim3(256), 0, s, a);
        }
    }
    if (int rc = check_launch("lstm backward recurrence")) return rc;
    if (dh0) {
        hipLaunchKernelGGL(lstm_slab_sum_kernel, dim3(cell_blocks), dim3(256), 0, s, w.slabs, Z, BH, dh0);
        if (int rc = check_launch("lstm_slab_sum_kernel")) return rc;
    }
    if (dc0) {  // a copy as a kernel (one "slab"): the library enqueues no memcpy / memset nodes
        hipLaunchKernelGGL(lstm_slab_sum_kernel, dim3(cell_blocks), dim3(256), 0, s, w.dc, 1, BH, dc0);
        if (int rc = check_launch("lstm_slab_sum_kernel")) return rc;
    }

    const int TB = T * B;
    if (int rc = colsum(w.dG, TB, 4 * H, 4 * H, w.cs, db, s)) return rc;
    // the gate gradients are the A operand of every product below: their maxima (two-plane fp16 mode) are taken once
    // (the fp16 recurrence has collected them step by step; otherwise one pass over dG)
    const bool dg_maxima = g_matrix_mode == GTG_MATRIX_F16X2;
    if (dg_maxima && !f16)
        if (int rc = absmax_partials(w.dG, TB, 4ll * H, 4ll * H, w.absmax, s)) return rc;
    GemmDesc g;
    // dW_ih = dG^T X
    g.A = w.dG; g.lda = 4 * H; g.a_kc = false;
    g.B = x; g.ldb = I; g.b_kc = false;
    g.C = dw_ih; g.ldc = I; g.M = 4 * H; g.N = I; g.K = TB;
    g.splits = gemm_pick_splits(4 * H, I, TB, 1); g.slabs = w.gemm_slabs;
    g.absmax_ws = w.absmax;
    g.a_partials_ready = dg_maxima;
    if (int rc = gemm(g, s)) return rc;
    // dW_hh = sum over steps of dG_t^T h_(previous step): a contiguous run of steps in both buffers, plus the
    // first step against h0
    if (T > 1) {
        g = GemmDesc();
        g.A = w.dG + (reverse ? 0 : 4 * BH); g.lda = 4 * H; g.a_kc = false;
        g.B = hs + (reverse ? BH : 0); g.ldb = H; g.b_kc = false;
        g.C = dw_hh; g.ldc = H; g.M = 4 * H; g.N = H; g.K = (T - 1) * B;
        g.splits = gemm_pick_splits(4 * H, H, (T - 1) * B, 1); g.slabs = w.gemm_slabs;
        g.absmax_ws = w.absmax;
        g.a_partials_ready = dg_maxima;  // (a bound for any run of its steps)
        g.b_bound = 1.f;  // hidden states
        if (int rc = gemm(g, s)) return rc;
    } else if (int rc = clear_i32(reinterpret_cast<int32_t*>(dw_hh), total, s)) {  // one step and no h0 term yet: zero
        return rc;
    }
    if (h0) {
        g = GemmDesc();
        g.A = w.dG + (reverse ? (long long)(T - 1) * 4 * BH : 0); g.lda = 4 * H; g.a_kc = false;
        g.B = h0; g.ldb = H; g.b_kc = false;
        g.C = dw_hh; g.ldc = H; g.M = 4 * H; g.N = H; g.K = B;
        g.epi = EPI_ACCUM;
        g.absmax_ws = w.absmax;
        g.a_partials_ready = dg_maxima;
        if (int rc = gemm(g, s)) return rc;
    }
    if (dx) {
        g = GemmDesc();
        g.A = w.dG; g.lda = 4 * H; g.a_kc = true;
        g.B = w_ih; g.ldb = I; g.b_kc = false;
        g.C = dx; g.ldc = I; g.M = TB; g.N = I; g.K = 4 * H;
        g.absmax_ws = w.absmax;
        g.a_partials_ready = dg_maxima;
        if (int rc = gemm(g, s)) return rc;
    }
    return GTG_OK;
}

}  // namespace gtg
