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
#include "mst_fill_n64.inc"

namespace gtg {

namespace {

constexpr int MAXN = 64;
constexpr int WAVES = 4;  // graphs per 256-thread block
constexpr float NEG_SENTINEL = -9223372036854775808.0f;  // float(-(2^63 - 1))

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float f32x32 __attribute__((ext_vector_type(32)));

// Maximum over the 64 lanes, NaN lanes ignored (v_max_f32 returns the other
// operand for a quiet NaN), returned wave-uniform.  Six DPP steps: xor-1 and
// xor-2 inside quads, half-row and row mirrors (every lane then holds the max
// of its row of 16), row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2
// and 3; lane 63 ends up holding the wave maximum.  Written as one asm block
// so that each step is a single v_max_f32_dpp; the s_nop 1 in front of every
// step is the 2 wait states a DPP read needs after a VALU write of its source.
__device__ __forceinline__ float wave_max_uniform(float v)
{
    float mx, t;  // t: its own register, so the caller's value needs no copy before the chain
    asm volatile(
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_readlane_b32 %1, %0, 63\n\t"
        "s_nop 3"   // the compiler cannot see that a VALU instruction wrote this SGPR: pad its hazards here
        : "=&v"(t), "=s"(mx) : "v"(v));
    return mx;
}

// w with lane `lane` (wave-uniform) replaced by a quiet NaN: one v_writelane
// instead of a compare and a select.  This compiler has no builtin for the
// instruction, so the LLVM intrinsic is bound by name; going through the
// compiler (not inline asm) keeps its hazard padding automatic.
extern "C" __device__ int gtg_llvm_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ float mark_lane_nan(float w, int lane)
{
    // (all ones: a quiet NaN that is an inline constant, so the lane index needs no detour through M0)
    return __int_as_float(gtg_llvm_writelane(-1, lane, __float_as_int(w)));
}

// Asynchronous global -> LDS copy of one upper-triangular row (m_tri floats)
// into the wave's LDS region: LDS-DMA loads, no registers involved; the data
// is complete after the wave's next s_waitcnt vmcnt(0).  16-byte pieces when
// the row is 16-byte aligned, dword pieces otherwise.
template <bool N64>
__device__ __forceinline__ void stage_row_async(const float* __restrict__ row, float* T, int m_tri, bool vec16, int lane)
{
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    if (N64 && vec16) {
        // n = 64: 2080 floats = 8 full 1-KiB pieces + one 128-byte piece, no loop, no per-piece guard
        const float* src = row + 4 * lane;
#pragma unroll
        for (int it = 0; it < 8; ++it)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + 256 * it), (lptr_t)(T + 256 * it), 16, 0, 0);
        if (lane < 8) __builtin_amdgcn_global_load_lds((gptr_t)(src + 2048), (lptr_t)(T + 2048), 16, 0, 0);
    } else if (vec16) {
        const int m4 = m_tri >> 2;
#pragma unroll 1
        for (int base = 0; base < m4; base += 64) {
            const int k = base + lane;
            if (k < m4)
                __builtin_amdgcn_global_load_lds((gptr_t)(row + 4 * k), (lptr_t)(T + 4 * base), 16, 0, 0);
        }
    } else {
#pragma unroll 1
        for (int base = 0; base < m_tri; base += 64) {
            const int k = base + lane;
            if (k < m_tri) __builtin_amdgcn_global_load_lds((gptr_t)(row + k), (lptr_t)(T + base), 4, 0, 0);
        }
    }
}

// Persistent: every wave walks graphs g = wave id, wave id + #waves, ...  For
// upper-triangular input the next graph's row is already in flight (LDS-DMA)
// while the current graph's spanning tree is computed from registers.
// PIN: the row's 64 registers are pinned to v0..v63 for the one-move fetch of the Prim round (see GTG_MST_VISIT).
// Under that constraint the compiler builds the generic triangular fill with about twice the scalar
// instructions, which a graph of few nodes (few rounds) does not earn back: the launcher pins from 40 nodes on.
// CAP: the most nodes this instantiation takes, 64 or 32.  With 32 the row is 32 registers: half the fill, a quarter
// of the LDS, and the Prim round fetches its element with one indexed move (no second half, no select).
template <bool TRIU, bool N64, bool PIN, int CAP = MAXN>
__global__ __launch_bounds__(WAVES * 64, TRIU ? 4 : 2) void mst_kernel(const float* __restrict__ in, int64_t G, int n,
                                                         int32_t* __restrict__ edges, int32_t* __restrict__ count)
{
    static_assert(CAP == 32 || CAP == 64, "rows of 32 or 64 registers");
    static_assert(CAP == 64 || (!N64 && !PIN), "the pinned-register and n = 64 forms are written for 64-register rows");
    constexpr int PER_WAVE = TRIU ? (CAP * (CAP + 1) / 2) : (CAP * (CAP + 1));
    __shared__ __attribute__((aligned(16))) float lds[WAVES * PER_WAVE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WAVES;
    if (N64) n = 64;  // lets the triangular offsets below fold to constants
    float* T = lds + wave * PER_WAVE;
    const int m_tri = n * (n + 1) / 2;
    const bool vec16 = (m_tri & 3) == 0 && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);

    const int j = lane;
    const int jc = (j < n) ? j : 0;  // lanes past n read vertex 0's row; their values are never used
    const int cj = jc * n - (jc * (jc - 1)) / 2 - jc;

    int64_t g = (int64_t)blockIdx.x * WAVES + wave;
    if (TRIU && g < G) stage_row_async<N64>(in + g * m_tri, T, m_tri, vec16, lane);

    for (; g < G; g += nwaves) {
        if (TRIU) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this graph's row has landed
        } else {
            // full matrices: staged transposed, T[c*(CAP+1) + r] = mat[r][c]
            const float* mat = in + g * n * n;
            for (int idx = lane; idx < n * n; idx += 64) {
                const int r = idx / n, c = idx - r * n;
                T[c * (CAP + 1) + r] = mat[idx];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        // ---- lane j takes row j of the matrix into registers: T[j][m] ---------
        // (upper-triangular input: U[min(j,m)][max(j,m)], element (i,k), i <= k,
        //  at i*n - i(i-1)/2 + (k - i)).  The Prim loop needs no memory access.
        f32x32 lo, hi;
        // Element (j,c) of the symmetric matrix is U[min(j,c)][max(j,c)], at
        // triangular offset min(j + cc, cj + c) with cc = c*n - c(c-1)/2 - c and
        // cj likewise for j: the wrong candidate always lies past the right one
        // because every row of the triangle is longer than the distance walked
        // back into it.  So one add, one min and one read per element, no
        // compare masks (n = 64 has a cheaper form still: mst_fill_n64.inc).  The two lane constants are laundered through an empty
        // asm so that the 64 addresses are not hoisted out of the graph loop
        // (they would cost 64 registers).
        typedef __attribute__((address_space(3))) const float* lds_cptr;
        int pj4 = (int)(uintptr_t)(__attribute__((address_space(3))) float*)T + 4 * jc;
        int pcj4 = (int)(uintptr_t)(__attribute__((address_space(3))) float*)T + 4 * cj;
        asm volatile("" : "+v"(pj4), "+v"(pcj4));
        if constexpr (TRIU && N64) {
            // two reads under a compile-time execution mask instead of the add + min
            unsigned long long saved_exec;
#define GTG_MST_FILL16(BLK_, V_, O_)                                                              \
            {                                                                                     \
                float t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15;       \
                BLK_(t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15, saved_exec, pj4, pcj4); \
                V_[O_ + 0] = t0, V_[O_ + 1] = t1, V_[O_ + 2] = t2, V_[O_ + 3] = t3;               \
                V_[O_ + 4] = t4, V_[O_ + 5] = t5, V_[O_ + 6] = t6, V_[O_ + 7] = t7;               \
                V_[O_ + 8] = t8, V_[O_ + 9] = t9, V_[O_ + 10] = t10, V_[O_ + 11] = t11;           \
                V_[O_ + 12] = t12, V_[O_ + 13] = t13, V_[O_ + 14] = t14, V_[O_ + 15] = t15;       \
            }
            GTG_MST_FILL16(GTG_MST_FILL_BLOCK_0, lo, 0)
            GTG_MST_FILL16(GTG_MST_FILL_BLOCK_1, lo, 16)
            GTG_MST_FILL16(GTG_MST_FILL_BLOCK_2, hi, 0)
            GTG_MST_FILL16(GTG_MST_FILL_BLOCK_3, hi, 16)
#undef GTG_MST_FILL16
        } else {
#pragma unroll
            for (int c = 0; c < CAP; ++c) {
                float v;
                if (TRIU) {
                    const int cc = c * n - (c * (c - 1)) / 2 - c;
                    const int addr = min(pj4 + 4 * (cc - c), pcj4) + 4 * c;
                    v = *(lds_cptr)(uintptr_t)addr;
                } else {
                    v = T[c * (CAP + 1) + jc];
                }
                if (c < 32)
                    lo[c] = v;
                else
                    hi[c - 32] = v;
                // bound the reads in flight: their destinations count against the 128-register budget
                if ((c & 15) == 15) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the triangular form goes through U + U^T - diag(U) in the reference (graph_vae.py:67-69): the diagonal is
        // (d + d) - d, which is d except that +inf gives NaN (vertex absent) and a d above FLT_MAX / 2 gives +inf
        const float draw = TRIU ? T[jc * n - (jc * (jc - 1)) / 2] : T[jc * (CAP + 1) + jc];
        const float diag = TRIU ? (draw + draw) - draw : draw;
        // every LDS read above has returned (waited for just below): the
        // region may be overwritten by the next graph's row now
        if (TRIU) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int64_t gn = g + nwaves;
            if (gn < G) stage_row_async<N64>(in + gn * m_tri, T, m_tri, vec16, lane);
        }

        // ---- present set: diagonal > 0.5 -----------------------------------
        const unsigned long long present = __ballot(j < n && diag > 0.5f);
        const int V = __popcll(present);
        int32_t* eout = edges + g * 2 * (n - 1);
        if (V == 0) {
            if (j < n - 1) {
                eout[2 * j] = -1;
                eout[2 * j + 1] = -1;
            }
            if (j == 0) count[g] = 0;
            continue;
        }
        const int root = __ffsll((long long)present) - 1;
        const int last = 63 - __clzll((long long)present);
        const bool pj = (present >> j) & 1ull;

        // w: best weight reaching vertex j so far; -2^63 = not reached yet;
        // quiet NaN = visited or absent (never selected: max skips NaN; never
        // updated: t > NaN is false).
        float w = pj ? NEG_SENTINEL : __builtin_nanf("");
        int parent = root;
        // One round: visit m, then let column m improve every unvisited vertex.
        // Row element m comes from ONE indexed move: the asm statement asks for the two register halves of the
        // row in v[0:31] and v[32:63] (physical-register constraints, which the register allocator honours by
        // keeping the row there for the whole loop), so the 64 elements are consecutive registers and index m
        // (0..63) from v0 reaches all of them -- where an indexed move per half and a select would take three
        // vector and three scalar instructions.
#define GTG_MST_VISIT(m_)                                                    \
        {                                                                        \
            const int mm = (m_);                                                 \
            float t;                                                             \
            if constexpr (PIN) {                                                 \
                asm volatile("s_set_gpr_idx_on %3, gpr_idx(SRC0)\n\t"              \
                             "v_mov_b32 %0, v0\n\t"                               \
                             "s_set_gpr_idx_off"                                 \
                             : "=v"(t) : "{v[0:31]}"(lo), "{v[32:63]}"(hi), "s"(mm)); \
            } else if constexpr (CAP == 32) {                                    \
                t = lo[mm & 31];                                                 \
            } else {                                                             \
                float tl = lo[mm & 31], th = hi[mm & 31];                        \
                asm volatile("" : "+v"(tl), "+v"(th));                           \
                t = (mm < 32) ? tl : th;                                         \
            }                                                                    \
            w = mark_lane_nan(w, mm);                                            \
            const bool upd = (t != 0.f) && (t > w);                              \
            w = upd ? t : w;                                                     \
            parent = upd ? mm : parent;                                          \
        }
        if (V > 1) GTG_MST_VISIT(root)  // round 0: weights[root] = +maxsize wins
        // (eq is never empty: a round starts with at least two unvisited present vertices, whose weights are
        // numbers; nothing reached yet -> Python's index -1 = the last present vertex)
#define GTG_MST_ROUND()                                                                      \
        {                                                                                        \
            const float mx = wave_max_uniform(w);                                                \
            const unsigned long long eq = __ballot(w == mx);                                     \
            const int first = __builtin_ctzll(eq);                                               \
            const int m = __builtin_amdgcn_readfirstlane(                                        \
                (__float_as_int(mx) == __float_as_int(NEG_SENTINEL)) ? last : first);            \
            GTG_MST_VISIT(m)                                                                     \
        }
        // two rounds per trip: the loop's own counter, compare and branch are a tenth of a round's instructions
        int round = 1;
        for (; round + 1 < V - 1; round += 2) {
            GTG_MST_ROUND()
            GTG_MST_ROUND()
        }
        if (round < V - 1) GTG_MST_ROUND()
#undef GTG_MST_ROUND
#undef GTG_MST_VISIT

        // ---- ordered output: slot (local index - 1) <- (parent, vertex) -----
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
}

// ---------------------------------------------------------------------------------------------------------
// Graphs of at most 16 nodes (the reference's default is 10, train.py:544): FOUR graphs per wavefront, one per
// DPP row of 16 lanes.  Lane (s, j) = (lane / 16, lane % 16) owns vertex j of the wave's s-th graph.  A row
// reduction needs no step across rows -- four single-instruction DPP steps (two inside quads, half-row mirror,
// row mirror) leave every lane holding its own row's result -- so neither the maximum nor the chosen vertex ever
// goes through a scalar register: the arg-max is a float max over the row followed by an unsigned min over
// (lane holds the max ? j : 16), and the chosen vertex m is a per-lane value.  Because m differs between the
// four graphs of a wave, T[j][m] cannot come from an indexed register; each graph's matrix is kept as a full
// 16 x 17 tile in LDS (rebuilt from the triangular row when the input is triangular) and the round reads one
// element of it.  Same semantics, same output as mst_kernel.
// (The same scheme with two graphs per wavefront for 17..32 nodes -- two DPP rows per graph joined by one
// exchange with the lane 16 away, 32 x 33 tiles -- measured slower than one graph per wave through mst_kernel,
// 0.085 against 0.069 ms for 65 536 graphs of 32 nodes: the tile build and the LDS it occupies cost more than the
// shared instructions save.  It is not kept.)
constexpr int SMALL_N = 16;
constexpr int SMALL_LD = SMALL_N + 1;
constexpr int SMALL_TRI = SMALL_N * (SMALL_N + 1) / 2;

__device__ __forceinline__ float row16_max(float v)
{
    float t;
    asm volatile(
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "=&v"(t) : "v"(v));
    return t;
}

__device__ __forceinline__ unsigned row16_min_u32(unsigned v)
{
    unsigned t;
    asm volatile(
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "=&v"(t) : "v"(v));
    return t;
}

template <bool TRIU>
__global__ __launch_bounds__(WAVES * 64) void mst_small_kernel(const float* __restrict__ in, int64_t G, int n,
                                                                int32_t* __restrict__ edges, int32_t* __restrict__ count)
{
    // per wave: the four graphs' input as it lies in memory (4 * per floats, contiguous), then four 16 x 17 tiles
    constexpr int RAW = 4 * (TRIU ? SMALL_TRI : SMALL_N * SMALL_N);
    constexpr int PER_WAVE = RAW + 4 * SMALL_N * SMALL_LD;
    __shared__ __attribute__((aligned(16))) float lds[WAVES * PER_WAVE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane >> 4, j = lane & 15;
    const int64_t nwaves = (int64_t)gridDim.x * WAVES;
    const int per = TRIU ? n * (n + 1) / 2 : n * n;   // floats per graph in memory
    float* raw = lds + wave * PER_WAVE;
    float* tile = raw + RAW + sub * (SMALL_N * SMALL_LD);
    const int jc = (j < n) ? j : 0;  // lanes past n read vertex 0's row; their values are never used
    const int cj = jc * n - (jc * (jc - 1)) / 2 - jc;

    for (int64_t q = (int64_t)blockIdx.x * WAVES + wave; q * 4 < G; q += nwaves) {
        const int64_t g = q * 4 + sub;
        const bool valid = g < G;
        // the wave's four graphs are contiguous in memory: one coalesced copy
        const int64_t left = G - q * 4;
        const int total = (int)(left < 4 ? left : 4) * per;
        const float* src = in + q * 4 * per;
        for (int idx = lane; idx < total; idx += 64) raw[idx] = src[idx];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const float* mine = raw + sub * per;
        // lane (s, j) writes row j of its graph's tile: tile[j][c] = T[j][c]
        float diag = 0.f;
        if (valid) {
            for (int c = 0; c < n; ++c) {   // wave-uniform trip count
                float v;
                if (TRIU) {
                    // U[min(j,c)][max(j,c)]: the right one of the two candidate offsets is the smaller (see mst_kernel)
                    const int cc = c * n - (c * (c - 1)) / 2 - c;
                    v = mine[min(jc + cc, cj + c)];
                } else {
                    v = mine[jc * n + c];
                }
                tile[jc * SMALL_LD + c] = v;   // lanes past n rewrite row 0 with the same values
                if (c == jc) diag = v;
            }
            if (TRIU) diag = (diag + diag) - diag;   // U + U^T - diag(U), graph_vae.py:67-69 (inf -> NaN, huge -> inf)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        const unsigned long long pres_all = __ballot(valid && j < n && diag > 0.5f);
        const unsigned pres = (unsigned)(pres_all >> (16 * sub)) & 0xffffu;   // this graph's present set
        const int V = __popc(pres);
        const int root = V ? __ffs(pres) - 1 : 0;
        const int last = V ? 31 - __clz(pres) : 0;
        const bool pj = (pres >> j) & 1u;

        // w as in mst_kernel (NaN = visited or absent, -2^63 = not reached); the root starts at +2^63, the
        // reference's weights[0] = +maxsize, so that the first round selects it through the common path
        float w = pj ? (j == root ? -NEG_SENTINEL : NEG_SENTINEL) : __builtin_nanf("");
        int parent = root;
        // every graph of the wave runs its own V - 1 rounds; the wave loops until the largest is done
        for (int r = 0; __ballot(r < V - 1) != 0ull; ++r) {
            const bool live = r < V - 1;
            const float mx = row16_max(w);
            const unsigned first = row16_min_u32((w == mx) ? (unsigned)j : 16u);
            // nothing reached yet -> Python's index -1 = the last present vertex
            const int m = (__float_as_int(mx) == __float_as_int(NEG_SENTINEL)) ? last : (int)first;
            const float t = tile[jc * SMALL_LD + m];   // m <= 16: column 16 is the tile's pad (a finished graph's rows)
            if (live) {
                if (j == m) w = __builtin_nanf("");
                const bool upd = (t != 0.f) && (t > w);
                w = upd ? t : w;
                parent = upd ? m : parent;
            }
        }

        if (valid) {
            int32_t* eout = edges + g * 2 * (n - 1);
            if (pj && j != root) {
                const int li = __popc(pres & ((1u << j) - 1u));
                eout[2 * (li - 1)] = parent;
                eout[2 * (li - 1) + 1] = j;
            }
            const int used = V > 0 ? V - 1 : 0;
            if (j >= used && j < n - 1) {
                eout[2 * j] = -1;
                eout[2 * j + 1] = -1;
            }
            if (j == 0) count[g] = used;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the next quad's copy overwrites raw and the tiles
    }
}

}  // namespace

static int num_cus()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus = v;
        else
            cus = 256;
    }
    return cus;
}

static int mst_launch(bool triu, const float* in, int64_t G, int n, int32_t* edges, int32_t* count, hipStream_t s)
{
    if (G < 0 || n < 0) return set_error(GTG_ERR_INVALID, "mst: negative size");
    if (n > MAXN) return set_error(GTG_ERR_UNSUPPORTED, "mst: n=%d exceeds the one-wavefront-per-graph limit of %d", n, MAXN);
    if (G == 0 || n == 0) return GTG_OK;
    if (!in || !count || (n > 1 && !edges)) return set_error(GTG_ERR_INVALID, "mst: null pointer");
    // persistent grid: LDS admits 4 blocks of the triangular kernel (2 of the full-matrix
    // one) per CU; more blocks than that only queue
    int64_t blocks = (G + WAVES - 1) / WAVES;
    const int64_t resident = (int64_t)num_cus() * (triu ? 4 : 2);
    if (blocks > resident) blocks = resident;
    ProfScope prof(s, triu ? "mst_triu" : "mst_full", G, n, 0);
    if (n <= SMALL_N) {
        // four graphs per wavefront, 16 per block; 7.5 KB (triangular) / 8.4 KB (full) of LDS per wave
        int64_t small_blocks = (G + 4 * WAVES - 1) / (4 * WAVES);
        const int64_t small_resident = (int64_t)num_cus() * 4;
        if (small_blocks > small_resident) small_blocks = small_resident;
        if (triu)
            hipLaunchKernelGGL((mst_small_kernel<true>), dim3((unsigned)small_blocks), dim3(WAVES * 64), 0, s, in, G, n, edges, count);
        else
            hipLaunchKernelGGL((mst_small_kernel<false>), dim3((unsigned)small_blocks), dim3(WAVES * 64), 0, s, in, G, n, edges, count);
        return check_launch("mst_small_kernel");
    }
    if (n <= 32) {
        // rows of 32 registers, 8.4 KB (triangular) of LDS per block: more blocks fit a CU than the 64-node form's four
        int64_t b32 = (G + WAVES - 1) / WAVES;
        const int64_t r32 = (int64_t)num_cus() * (triu ? 8 : 4);
        if (b32 > r32) b32 = r32;
        if (triu)
            hipLaunchKernelGGL((mst_kernel<true, false, false, 32>), dim3((unsigned)b32), dim3(WAVES * 64), 0, s, in, G, n, edges, count);
        else
            hipLaunchKernelGGL((mst_kernel<false, false, false, 32>), dim3((unsigned)b32), dim3(WAVES * 64), 0, s, in, G, n, edges, count);
        return check_launch("mst_kernel<32>");
    }
    if (triu && n == 64)
        hipLaunchKernelGGL((mst_kernel<true, true, true>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, s, in, G, n, edges, count);
    else if (triu && n >= 40)
        hipLaunchKernelGGL((mst_kernel<true, false, true>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, s, in, G, n, edges, count);
    else if (triu)
        hipLaunchKernelGGL((mst_kernel<true, false, false>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, s, in, G, n, edges, count);
    else
        hipLaunchKernelGGL((mst_kernel<false, false, true>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, s, in, G, n, edges, count);
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
