// Element-wise and gather/scatter pieces of the text branch (/root/reference/src/text_vae.py:35-36, 55-56,
// 62-65; src/noise.py:12-22), gfx950 only:
//
//   word drop          every token of a sentence but the first is dropped with probability p, the kept ones
//                      move up in order, the tail is padded (noise.py:12-22: a Python loop over sentences
//                      with a .tolist() round trip per sentence in the reference);
//   embedding + dropout  out[r, :] = drop(table[ids[r], :]) (+ add[r mod B, :]), text_vae.py:56 and :62
//                      (self.drop(self.embed(input)), plus self.z2emb(z) broadcast over time in the decoder);
//   its backward       dtable[id, :] += sum over the rows r with ids[r] = id of the dropped gradient, in a
//                      FIXED order (no float atomics, no sort): the rows of a batch are few (T*B ~ 10^4), so
//                      every row's wavefront finds the other rows of its id by comparing id lists, first inside
//                      its group of 128 rows, then across groups (embed_bwd_group_kernel / _final_kernel);
//   dropout            y = x * keep / (1 - p) and the same on the gradient (text_vae.py:64).
//
// All of it is HBM- or latency-bound byte shuffling; nothing here goes near the matrix cores.
//
// Randomness: a keep decision is a pure function of (seed, stream id, element index) -- Philox4x32-7, four
// 32-bit words per call, one call per four consecutive elements -- so the backward pass regenerates the mask
// of the forward pass from the same three numbers and no mask is ever stored.  The sequences differ from
// PyTorch's (and from numpy's, which the reference uses for word drop): only the distribution is the same.
#include "gtg_internal.h"

namespace gtg {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct U4 {
    unsigned x, y, z, w;
};

// Philox4x32-7 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; seven rounds are the fewest
// the paper reports as passing BigCrush, ten its default): counter (c0..c3), key (k0, k1).  The generator sits on
// the serial path of the ordered scatter-add below (a row's mask is regenerated before the row is added), so its
// cost is not hidden behind memory there.
__device__ __forceinline__ U4 philox4x32_7(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1)
{
    constexpr unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const unsigned hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const unsigned hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return U4{c0, c1, c2, c3};
}

// the four uniform words of element group `quad` (elements 4 quad .. 4 quad + 3) of random stream `sid`
__device__ __forceinline__ U4 rand4(unsigned long long seed, unsigned long long sid, unsigned long long quad)
{
    return philox4x32_7((unsigned)quad, (unsigned)(quad >> 32), (unsigned)sid, (unsigned)(sid >> 32), (unsigned)seed,
                         (unsigned)(seed >> 32));
}

// keep with probability 1 - p: the word, as a fraction of 2^32, is >= p.  thresh = min(p * 2^32, 2^32 - 1).
__device__ __forceinline__ bool keep_word(unsigned word, unsigned thresh) { return word >= thresh; }

inline unsigned drop_threshold(float p)
{
    const double t = (double)p * 4294967296.0;
    if (t <= 0.0) return 0u;
    if (t >= 4294967295.0) return 4294967295u;
    return (unsigned)t;
}

// ---- word drop ----------------------------------------------------------------------------------------
// x, out (T, B) int64, time-major as the reference feeds them.  One wavefront per sentence (column b): 64 tokens
// per pass, a ballot of the keep flags, the lane's rank among the kept ones is its place in the output.
// keep_in (T, B) uint8 nullable: an explicit keep mask (tests; row 0 is forced to "keep" either way).
__global__ __launch_bounds__(256) void word_drop_kernel(const int64_t* __restrict__ x, int T, int B, int64_t padidx,
                                                        unsigned thresh, unsigned long long seed, unsigned long long sid,
                                                        const uint8_t* __restrict__ keep_in, int64_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;  // wave-uniform
    int placed = 0;
    for (int base = 0; base < T; base += 64) {
        const int t = base + lane;
        bool keep = false;
        int64_t tok = padidx;
        if (t < T) {
            const long long e = (long long)t * B + b;
            tok = x[e];
            if (t == 0)
                keep = true;
            else if (keep_in)
                keep = keep_in[e] != 0;
            else {
                const U4 r = rand4(seed, sid, (unsigned long long)e >> 2);
                const unsigned w = (e & 3) == 0 ? r.x : (e & 3) == 1 ? r.y : (e & 3) == 2 ? r.z : r.w;
                keep = keep_word(w, thresh);
            }
        }
        const unsigned long long bal = __ballot(keep);
        if (keep) out[(long long)(placed + __popcll(bal & ((1ull << lane) - 1ull))) * B + b] = tok;
        placed += __popcll(bal);
    }
    for (int t = placed + lane; t < T; t += 64) out[(long long)t * B + b] = padidx;
}

// ---- embedding lookup + dropout (+ broadcast add) --------------------------------------------------------
// One block per output row r: D / 4 float4 per row.  `add` (B_add, D) nullable: row r takes add[r mod B_add].
// ids outside [0, V) set *err and read row 0.
__global__ __launch_bounds__(128) void embed_dropout_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids,
                                                                long long V, int D, unsigned thresh, float scale,
                                                                unsigned long long seed, unsigned long long sid,
                                                                const float* __restrict__ add, int B_add,
                                                                float* __restrict__ out, int32_t* __restrict__ err)
{
    const long long r = blockIdx.x;
    long long id = ids[r];
    if (id < 0 || id >= V) {
        if (threadIdx.x == 0 && err) *err = 1;
        id = 0;
    }
    const f32x4* src = reinterpret_cast<const f32x4*>(table + id * D);
    const f32x4* ad = add ? reinterpret_cast<const f32x4*>(add + (r % B_add) * (long long)D) : nullptr;
    f32x4* dst = reinterpret_cast<f32x4*>(out + r * D);
    const int d4 = D >> 2;
    for (int c = threadIdx.x; c < d4; c += 128) {
        f32x4 v = src[c];
        if (thresh) {
            const U4 k = rand4(seed, sid, (unsigned long long)(r * d4 + c));
            v[0] = keep_word(k.x, thresh) ? v[0] * scale : 0.f;
            v[1] = keep_word(k.y, thresh) ? v[1] * scale : 0.f;
            v[2] = keep_word(k.z, thresh) ? v[2] * scale : 0.f;
            v[3] = keep_word(k.w, thresh) ? v[3] * scale : 0.f;
        }
        if (ad) v += ad[c];
        dst[c] = v;
    }
}

// Backward of the lookup, dtable[id, :] += sum of the dropped gradients of the rows with that id, in a fixed order
// (ascending rows inside a group, then ascending groups: bit-reproducible; no float atomics, no sort), in two levels
// so that an id with thousands of rows (the pad symbol, the sentence delimiters) is not one block's serial sum:
//   level 1  rows are cut into fixed groups of EG consecutive rows.  The block of row r compares the ids of
//            its group with its own (a ballot per wavefront); if r is the first row of its id in the group it sums the
//            group's rows of that id in order into part[r, :] and sets flag[r] = 2 -- or flag[r] = 1 and no sum
//            when it is the only one (its gradient is then read in place by level 2); every other row gets 0.
//   level 2  the block of a flagged row scans the int32 copy of the id list that level 1 left (L2-resident:
//            4 R bytes): a row of its id in an earlier group -> some other row is the first of all, done;
//            otherwise it adds, group after group, the flagged rows of its id (at most one per group) onto
//            dtable[id, :], which it owns.
// Both kernels: one block per row; a list of rows is summed by the whole block (block_sum_rows), loads of up to EB
// rows in flight per thread.
constexpr int EG = 128;    // rows per group
constexpr int EP = 4;      // a list of rows is cut into EP contiguous parts, one per EG threads of a block, summed in part order
constexpr int ET = EG * EP;  // threads per block
constexpr int EB = 16;     // rows whose loads are issued together
constexpr int EMAXG = 2048;  // groups per call: at most EMAXG * EG rows

__device__ __forceinline__ f32x4 dropped(const float* __restrict__ dout, long long q, int d4, int c, unsigned thresh, float scale,
                                         unsigned long long seed, unsigned long long sid)
{
    f32x4 g = reinterpret_cast<const f32x4*>(dout)[q * d4 + c];
    if (thresh) {
        const U4 k = rand4(seed, sid, (unsigned long long)(q * d4 + c));
        g[0] = keep_word(k.x, thresh) ? g[0] * scale : 0.f;
        g[1] = keep_word(k.y, thresh) ? g[1] * scale : 0.f;
        g[2] = keep_word(k.z, thresh) ? g[2] * scale : 0.f;
        g[3] = keep_word(k.w, thresh) ? g[3] * scale : 0.f;
    }
    return g;
}

// acc (one float4 column per thread and pass) += the listed rows, in list order.  The loads of a batch are issued
// together and added afterwards; batches of EB rows, then one of 8, 4, 2, 1 for what is left (a list of one row,
// by far the commonest, is one load and one mask evaluation).  from_part(q) says whether row q is read from `part`
// (a level-1 sum) or from dout through the dropout mask.
template <int NB, typename FromPart>
__device__ __forceinline__ f32x4 sum_batch(f32x4 acc, const int* rows, const float* __restrict__ dout,
                                           const float* __restrict__ part, int d4, int c, unsigned thresh, float scale,
                                           unsigned long long seed, unsigned long long sid, FromPart from_part)
{
    f32x4 v[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int q = rows[k];
        v[k] = from_part(q) ? reinterpret_cast<const f32x4*>(part)[(long long)q * d4 + c]
                            : dropped(dout, q, d4, c, thresh, scale, seed, sid);
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) acc += v[k];
    return acc;
}

template <typename FromPart>
__device__ __forceinline__ f32x4 sum_rows(f32x4 acc, const int* rows, int n, const float* __restrict__ dout,
                                          const float* __restrict__ part, int d4, int c, unsigned thresh, float scale,
                                          unsigned long long seed, unsigned long long sid, FromPart from_part)
{
    int i = 0;
    for (; i + EB <= n; i += EB) acc = sum_batch<EB>(acc, rows + i, dout, part, d4, c, thresh, scale, seed, sid, from_part);
    if (n - i >= 8) { acc = sum_batch<8>(acc, rows + i, dout, part, d4, c, thresh, scale, seed, sid, from_part); i += 8; }
    if (n - i >= 4) { acc = sum_batch<4>(acc, rows + i, dout, part, d4, c, thresh, scale, seed, sid, from_part); i += 4; }
    if (n - i >= 2) { acc = sum_batch<2>(acc, rows + i, dout, part, d4, c, thresh, scale, seed, sid, from_part); i += 2; }
    if (n - i >= 1) acc = sum_batch<1>(acc, rows + i, dout, part, d4, c, thresh, scale, seed, sid, from_part);
    return acc;
}

// The whole block sums the n listed rows into one row of D floats: thread (p, t) = (tid / EG, tid % EG) adds part p
// of the list, rows [p n / EP, (p + 1) n / EP), in order into its float4 column(s) t, t + EG, ...; the EP partial
// columns meet in LDS and part 0's thread adds them in part order onto `init` and hands the result to store(c, v).
// Short lists (n < 2 EP) are summed by part 0 alone.  Every thread of the block must call this.
template <typename FromPart, typename Init, typename Store>
__device__ __forceinline__ void block_sum_rows(const int* rows, int n, const float* __restrict__ dout,
                                               const float* __restrict__ part, int d4, unsigned thresh, float scale,
                                               unsigned long long seed, unsigned long long sid, FromPart from_part, Init init,
                                               Store store)
{
    __shared__ f32x4 s_part[(EP > 1 ? EP - 1 : 1) * EG];
    const int tid = threadIdx.x, p = tid / EG, t = tid % EG;
    const bool split = n >= 2 * EP;
    const int lo = split ? (int)((long long)p * n / EP) : 0, hi = split ? (int)((long long)(p + 1) * n / EP) : n;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < d4; c0 += EG) {
        const int c = c0 + t;
        f32x4 acc = zero;
        if (c < d4 && (split || p == 0)) {
            if (p == 0) acc = init(c);
            acc = sum_rows(acc, rows + lo, hi - lo, dout, part, d4, c, thresh, scale, seed, sid, from_part);
        }
        if (split) {   // block-uniform
            if (p > 0) s_part[(p - 1) * EG + t] = acc;
            __syncthreads();
            if (p == 0) {
#pragma unroll
                for (int k = 1; k < EP; ++k) acc += s_part[(k - 1) * EG + t];
            }
            __syncthreads();
        }
        if (p == 0 && c < d4) store(c, acc);
    }
}

__global__ __launch_bounds__(ET) void embed_bwd_group_kernel(const float* __restrict__ dout, const int64_t* __restrict__ ids,
                                                             int R, long long V, int D, unsigned thresh, float scale,
                                                             unsigned long long seed, unsigned long long sid,
                                                             int32_t* __restrict__ ids32, uint8_t* __restrict__ flag,
                                                             int32_t* __restrict__ id_count, float* __restrict__ part)
{
    __shared__ int s_rows[EG];
    __shared__ unsigned long long s_bal[EG / 64];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g0 = (r / EG) * EG;
    const long long id = ids[r];
    const bool valid = id >= 0 && id < V;  // an id outside the table was flagged by the forward; it gets no gradient
    if (tid == 0) ids32[r] = valid ? (int)id : -1;
    // the group's rows of this id, in ascending order: thread t < EG looks at row g0 + t
    const int q = g0 + tid;
    const bool hit = tid < EG && valid && q < R && ids[q] == id;
    const unsigned long long bal = __ballot(hit);
    if (lane == 0 && wave < EG / 64) s_bal[wave] = bal;
    __syncthreads();
    int n = 0, before = 0, my_rank = __popcll(bal & ((1ull << lane) - 1ull));
#pragma unroll
    for (int w = 0; w < EG / 64; ++w) {
        const unsigned long long b = s_bal[w];
        const int lo = r - g0 - 64 * w;   // rows of this wave's chunk that lie before r
        before += lo >= 64 ? __popcll(b) : lo > 0 ? __popcll(b & ((1ull << lo) - 1ull)) : 0;
        if (w < wave) my_rank += __popcll(b);
        n += __popcll(b);
    }
    // id_count[id] (zero on entry): how many groups hold this id, i.e. how many flagged rows level 2 will find
    // for it (integer atomics: the outcome does not depend on the order of arrival)
    if (tid == 0 && valid && before == 0) atomicAdd(&id_count[id], 1);
    if (!valid || before > 0 || n == 1) {
        if (tid == 0) flag[r] = (valid && before == 0) ? 1 : 0;
        return;
    }
    if (tid == 0) flag[r] = 2;
    if (hit) s_rows[my_rank] = q;
    __syncthreads();
    const int d4 = D >> 2;
    f32x4* prow = reinterpret_cast<f32x4*>(part) + (long long)r * d4;
    block_sum_rows(s_rows, n, dout, part, d4, thresh, scale, seed, sid, [](int) { return false; },
                   [](int) { const f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }, [prow](int c, f32x4 v) { prow[c] = v; });
}

__global__ __launch_bounds__(ET) void embed_bwd_final_kernel(const float* __restrict__ dout, const int32_t* __restrict__ ids32,
                                                             const uint8_t* __restrict__ flag,
                                                             const int32_t* __restrict__ id_count, const float* __restrict__ part,
                                                             int R, int D, unsigned thresh, float scale, unsigned long long seed,
                                                             unsigned long long sid, float* __restrict__ dtable)
{
    __shared__ int s_grp[EMAXG];   // the flagged row of this id in each group, -1 if none
    __shared__ int s_rows[EMAXG];
    __shared__ int s_n;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (flag[r] == 0) return;      // block-uniform
    const int id = ids32[r];
    const int d4 = D >> 2;
    f32x4* drow = reinterpret_cast<f32x4*>(dtable + (long long)id * D);
    if (id_count[id] == 1) {
        // the id lives in this group alone (most ids of a batch occur once): no search for partners
        if (tid == 0) s_rows[0] = r;
        __syncthreads();
        block_sum_rows(s_rows, 1, dout, part, d4, thresh, scale, seed, sid, [flag](int q) { return flag[q] == 2; },
                       [drow](int c) { return drow[c]; }, [drow](int c, f32x4 v) { drow[c] = v; });
        return;
    }
    const int g0 = (r / EG) * EG;
    const int ngroups = (R + EG - 1) / EG, mygroup = r / EG;
    for (int g = tid; g < ngroups; g += ET) s_grp[g] = -1;
    __syncthreads();
    // One pass over the id list (int32, L2-resident), four ids per thread and step, every load independent of the
    // others: a row of this id in an earlier group means some other row is the first of all; in a later group it
    // is a candidate partner, recorded per group if it is flagged (at most one per group is).
    bool earlier = false;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const int r4 = R & ~3;
    for (int q = 4 * tid; q < R; q += 4 * ET) {
        i32x4 v;
        if (q < r4) {
            v = *reinterpret_cast<const i32x4*>(ids32 + q);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = q + k < R ? ids32[q + k] : -2;
        }
        if (v[0] == id || v[1] == id || v[2] == id || v[3] == id) {
            if (q < g0) {
                earlier = true;       // whole vectors lie on one side of g0 (a multiple of 4) and of g0 + EG
            } else if (q >= g0 + EG) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (v[k] == id && flag[q + k] != 0) s_grp[(q + k) / EG] = q + k;
            }
        }
    }
    if (__syncthreads_or(earlier)) return;
    if (tid < 64) {    // ascending compaction by the first wavefront
        int n = 1;
        if (lane == 0) s_rows[0] = r;
        for (int base = mygroup + 1; base < ngroups; base += 64) {
            const int g = base + lane;
            const int q = g < ngroups ? s_grp[g] : -1;
            const unsigned long long bal = __ballot(q >= 0);
            if (q >= 0) s_rows[n + __popcll(bal & ((1ull << lane) - 1ull))] = q;
            n += __popcll(bal);
        }
        if (lane == 0) s_n = n;
    }
    __syncthreads();
    block_sum_rows(s_rows, s_n, dout, part, d4, thresh, scale, seed, sid, [flag](int q) { return flag[q] == 2; },
                   [drow](int c) { return drow[c]; }, [drow](int c, f32x4 v) { drow[c] = v; });
}

// ---- dropout on a flat array (forward and backward are the same map) ---------------------------------------
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, long long n4, unsigned thresh, float scale,
                                                      unsigned long long seed, unsigned long long sid, float* __restrict__ y)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        const U4 k = rand4(seed, sid, (unsigned long long)i);
        v[0] = keep_word(k.x, thresh) ? v[0] * scale : 0.f;
        v[1] = keep_word(k.y, thresh) ? v[1] * scale : 0.f;
        v[2] = keep_word(k.z, thresh) ? v[2] * scale : 0.f;
        v[3] = keep_word(k.w, thresh) ? v[3] * scale : 0.f;
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

}  // namespace

int word_drop(const int64_t* x, int T, int B, int64_t padidx, float p, unsigned long long seed, unsigned long long sid,
              const uint8_t* keep, int64_t* out, hipStream_t s)
{
    hipLaunchKernelGGL(word_drop_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, x, T, B, padidx, drop_threshold(p), seed,
                       sid, keep, out);
    return check_launch("word_drop_kernel");
}

int embed_dropout_fwd(const float* table, const int64_t* ids, int64_t R, int64_t V, int D, float p, unsigned long long seed,
                      unsigned long long sid, const float* add, int B_add, float* out, int32_t* err, hipStream_t s)
{
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    hipLaunchKernelGGL(embed_dropout_fwd_kernel, dim3((unsigned)R), dim3(128), 0, s, table, ids, (long long)V, D,
                       drop_threshold(p), scale, seed, sid, add, B_add, out, err);
    return check_launch("embed_dropout_fwd_kernel");
}

size_t embed_dropout_bwd_workspace(int64_t R, int64_t V, int D)
{
    // ids32 (R) | flag (R) | id_count (V) | part (R, D), each 256-byte aligned
    auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
    return up((size_t)R * 4) + up((size_t)R) + up((size_t)V * 4) + up((size_t)R * D * 4);
}

int embed_dropout_bwd(const float* dout, const int64_t* ids, int64_t R, int64_t V, int D, float p, unsigned long long seed,
                      unsigned long long sid, float* dtable, void* ws, size_t ws_bytes, hipStream_t s)
{
    if (R > (long long)EMAXG * EG) return set_error(GTG_ERR_UNSUPPORTED, "embed_dropout_bwd: at most %d rows per call", EMAXG * EG);
    if (V >= (1ll << 31)) return set_error(GTG_ERR_UNSUPPORTED, "embed_dropout_bwd: vocabulary too large");
    if (ws == nullptr || ws_bytes < embed_dropout_bwd_workspace(R, V, D))
        return set_error(GTG_ERR_WORKSPACE, "embed_dropout_bwd: workspace too small");
    auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
    char* base = reinterpret_cast<char*>(ws);
    int32_t* ids32 = reinterpret_cast<int32_t*>(base);
    uint8_t* flag = reinterpret_cast<uint8_t*>(base + up((size_t)R * 4));
    int32_t* id_count = reinterpret_cast<int32_t*>(base + up((size_t)R * 4) + up((size_t)R));
    float* part = reinterpret_cast<float*>(base + up((size_t)R * 4) + up((size_t)R) + up((size_t)V * 4));
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    if (int rc = clear_i32(id_count, V, s)) return rc;
    hipLaunchKernelGGL(embed_bwd_group_kernel, dim3((unsigned)R), dim3(ET), 0, s, dout, ids, (int)R, (long long)V, D,
                       drop_threshold(p), scale, seed, sid, ids32, flag, id_count, part);
    if (int rc = check_launch("embed_bwd_group_kernel")) return rc;
    hipLaunchKernelGGL(embed_bwd_final_kernel, dim3((unsigned)R), dim3(ET), 0, s, dout, ids32, flag, id_count, part, (int)R, D,
                       drop_threshold(p), scale, seed, sid, dtable);
    return check_launch("embed_bwd_final_kernel");
}

int dropout_map(const float* x, int64_t n, float p, unsigned long long seed, unsigned long long sid, float* y, hipStream_t s)
{
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n4, drop_threshold(p), scale, seed, sid, y);
    return check_launch("dropout_kernel");
}

}  // namespace gtg
