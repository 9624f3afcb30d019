// Batched maximum-spanning-tree read-out: one 64-lane wavefront per graph.
//
// Semantics (bit-exact target): Prim's maximum spanning tree over the
// sub-matrix of "present" vertices (diagonal > 0.5, ascending order), rooted
// at the first present vertex, exactly V-1 rounds of
//   m   = first unvisited vertex holding the strictly greatest reached weight,
//         or the LAST present vertex when no unvisited vertex has been reached;
//   for every unvisited j with T[j][m] != 0 and T[j][m] > weight[j]:
//         weight[j] = T[j][m], parent[j] = m
// and the ordered output (parent[i], i) for every present vertex but the root.
//
// Lane j owns vertex j: its weight, its parent and its row of the matrix.
// The visited set and the present set are 64-bit scalar masks.  "Never
// reached" is the float -2^63; an update needs t > weight[j], so a stored
// weight is always > -2^63 and never NaN, which makes the select a plain float
// max over the eligible lanes (DPP row reductions + two row broadcasts), the
// tie-break a ballot + find-first-set.
//
// Upper-triangular input (one decoder row of n(n+1)/2 floats per graph) is
// staged once, coalesced, into LDS and read as U[min(j,m)][max(j,m)]; the
// symmetric n x n matrix is never materialised.  Full n x n input is staged
// transposed (padded row stride) so that lane j reads T[j][m] conflict-free.
#include "gtg_internal.h"

namespace gtg {

namespace {

constexpr int MAXN = 64;
constexpr int WAVES = 4;  // graphs per 256-thread block
constexpr float NEG_SENTINEL = -9223372036854775808.0f;  // float(-(2^63 - 1))

typedef float f32x4 __attribute__((ext_vector_type(4)));

// dpp_ctrl encodings (gfx9): quad_perm, row_half_mirror, row_mirror, row_bcast15/31
constexpr int DPP_QUAD_XOR1 = 0xB1;
constexpr int DPP_QUAD_XOR2 = 0x4E;
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_ROW_BCAST31 = 0x143;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max(float v)
{
    const int t = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return fmaxf(v, __int_as_float(t));
}

// Maximum over the 64 lanes, returned as a wave-uniform value.
__device__ __forceinline__ float wave_max_uniform(float v)
{
    v = dpp_max<DPP_QUAD_XOR1, 0xF>(v);
    v = dpp_max<DPP_QUAD_XOR2, 0xF>(v);
    v = dpp_max<DPP_ROW_HALF_MIRROR, 0xF>(v);
    v = dpp_max<DPP_ROW_MIRROR, 0xF>(v);       // every lane: max of its row of 16
    v = dpp_max<DPP_ROW_BCAST15, 0xA>(v);      // rows 1,3 += rows 0,2
    v = dpp_max<DPP_ROW_BCAST31, 0xC>(v);      // rows 2,3 += rows 0,1 -> lane 63 holds all
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <bool TRIU>
__global__ __launch_bounds__(WAVES * 64) void mst_kernel(const float* __restrict__ in, int64_t G, int n,
                                                         int32_t* __restrict__ edges, int32_t* __restrict__ count)
{
    constexpr int PER_WAVE = TRIU ? (MAXN * (MAXN + 1) / 2) : (MAXN * (MAXN + 1));
    __shared__ __attribute__((aligned(16))) float lds[WAVES * PER_WAVE];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t g = (int64_t)blockIdx.x * WAVES + wave;
    const bool valid = g < G;
    float* T = lds + wave * PER_WAVE;
    const int m_tri = n * (n + 1) / 2;

    // ---- stage the graph's weights into LDS (coalesced global reads) --------
    if (valid) {
        if (TRIU) {
            const float* row = in + g * m_tri;
            if ((m_tri & 3) == 0 && ((reinterpret_cast<uintptr_t>(in) & 15) == 0)) {
                const f32x4* row4 = reinterpret_cast<const f32x4*>(row);
                for (int k = lane; k < (m_tri >> 2); k += 64) reinterpret_cast<f32x4*>(T)[k] = row4[k];
            } else {
                for (int k = lane; k < m_tri; k += 64) T[k] = row[k];
            }
        } else {
            // T[c*(MAXN+1) + r] = mat[r][c]
            const float* mat = in + g * n * n;
            for (int idx = lane; idx < n * n; idx += 64) {
                const int r = idx / n, c = idx - r * n;
                T[c * (MAXN + 1) + r] = mat[idx];
            }
        }
    }
    __syncthreads();
    if (!valid) return;

    const int j = lane;
    // ---- present set: diagonal > 0.5 ---------------------------------------
    float diag = 0.f;
    if (j < n) diag = TRIU ? T[j * n - (j * (j - 1)) / 2] : T[j * (MAXN + 1) + j];
    const unsigned long long present = __ballot(j < n && diag > 0.5f);
    const int V = __popcll(present);
    int32_t* eout = edges + g * 2 * (n - 1);

    if (V == 0) {
        if (j < n - 1) {
            eout[2 * j] = -1;
            eout[2 * j + 1] = -1;
        }
        if (j == 0) count[g] = 0;
        return;
    }

    const int root = __ffsll((long long)present) - 1;
    const int last = 63 - __clzll((long long)present);
    const bool pj = (present >> j) & 1ull;
    const int cj = j * n - (j * (j - 1)) / 2 - j;  // triu: U[j][m] (m > j) sits at cj + m

    float w = NEG_SENTINEL;
    int parent = root;
    unsigned long long visited = 0ull;

    for (int round = 0; round < V - 1; ++round) {
        int m;
        if (round == 0) {
            m = root;
        } else {
            const bool elig = pj && !((visited >> j) & 1ull) && (w > NEG_SENTINEL);
            const unsigned long long em = __ballot(elig);
            if (em == 0ull) {
                m = last;
            } else {
                const float mx = wave_max_uniform(elig ? w : -__builtin_inff());
                const unsigned long long eq = __ballot(elig && w == mx);
                m = __ffsll((long long)eq) - 1;
            }
        }
        m = __builtin_amdgcn_readfirstlane(m);
        visited |= (1ull << m);
        float t;
        if (TRIU) {
            const int cm = m * n - (m * (m - 1)) / 2 - m;
            const int idx = (j > m) ? (cm + j) : (cj + m);
            t = T[(j < n) ? idx : 0];
        } else {
            t = T[m * (MAXN + 1) + j];
        }
        const bool upd = pj && !((visited >> j) & 1ull) && (t != 0.f) && (t > w);
        if (upd) {
            w = t;
            parent = m;
        }
    }

    // ---- ordered output: slot (local index - 1) <- (parent, vertex) ---------
    if (pj && j != root) {
        const int li = __popcll(present & ((1ull << j) - 1ull));
        eout[2 * (li - 1)] = parent;
        eout[2 * (li - 1) + 1] = j;
    }
    if (j >= V - 1 && j < n - 1) {
        eout[2 * j] = -1;
        eout[2 * j + 1] = -1;
    }
    if (j == 0) count[g] = V - 1;
}

}  // namespace

static int mst_launch(bool triu, const float* in, int64_t G, int n, int32_t* edges, int32_t* count, hipStream_t s)
{
    if (G < 0 || n < 0) return set_error(GTG_ERR_INVALID, "mst: negative size");
    if (n > MAXN) return set_error(GTG_ERR_UNSUPPORTED, "mst: n=%d exceeds the one-wavefront-per-graph limit of %d", n, MAXN);
    if (G == 0 || n == 0) return GTG_OK;
    if (!in || !count || (n > 1 && !edges)) return set_error(GTG_ERR_INVALID, "mst: null pointer");
    const int64_t blocks = (G + WAVES - 1) / WAVES;
    if (blocks > 0x7fffffffLL) return set_error(GTG_ERR_INVALID, "mst: too many graphs for one launch");
    ProfScope prof(s, triu ? "mst_triu" : "mst_full", G, n, 0);
    if (triu)
        hipLaunchKernelGGL(mst_kernel<true>, dim3((unsigned)blocks), dim3(WAVES * 64), 0, s, in, G, n, edges, count);
    else
        hipLaunchKernelGGL(mst_kernel<false>, dim3((unsigned)blocks), dim3(WAVES * 64), 0, s, in, G, n, edges, count);
    return check_launch("mst_kernel");
}

int mst_triu(const float* rows, int64_t G, int n, int32_t* edges, int32_t* count, hipStream_t s)
{
    return mst_launch(true, rows, G, n, edges, count, s);
}

int mst_full(const float* mats, int64_t G, int n, int32_t* edges, int32_t* count, hipStream_t s)
{
    return mst_launch(false, mats, G, n, edges, count, s);
}

}  // namespace gtg
