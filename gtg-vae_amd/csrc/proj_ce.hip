// Vocabulary projection + token cross-entropy of the text decoder
// (/root/reference/src/text_vae.py:65 self.proj, :77-79 loss_rec with
// ignore_index = pad, :81-84 the arg-max of accuracy), gfx950 only.
//
// Tokens whose target is the pad symbol contribute neither loss nor gradient,
// so their rows never enter the GEMMs: the rows that count are packed to the
// front ("compact rows"), their number stays on the device, and the three
// GEMMs (logits, weight gradient, input gradient) read it there.
//
//   rowmap (2 R + 1 int32): [0] = count of kept rows,
//                           [1 + r]     = compact index of row r, -1 for a pad row,
//                           [1 + R + i] = original row of compact row i.
//
// The HBM-bound kernels here: row map (one block, scan), row gather / scale /
// scatter over (R, K), the soft-max kernel that turns a row of logits into
// (probabilities - one-hot) in place with one read and one write per element,
// and the weighted column sum for the bias gradient.
#include "gtg_internal.h"

namespace gtg {
namespace {

constexpr int CE_THREADS = 512;
constexpr int CE_CACHE_V4 = 16;  // float4 per thread held in registers: rows up to 32 768 columns

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- row map ------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ce_rowmap_kernel(const int64_t* __restrict__ targets, int R, int64_t padidx,
                                                         int32_t* __restrict__ rowmap, float* __restrict__ loss,
                                                         int32_t* __restrict__ pred)
{
    __shared__ int wave_tot[16];
    __shared__ int running;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) running = 0;
    __syncthreads();
    int32_t* map = rowmap + 1;
    int32_t* list = rowmap + 1 + R;
    for (int base = 0; base < R; base += 1024) {
        const int r = base + tid;
        const bool keep = r < R && targets[r] != padidx;
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (r < R) {
            if (keep) {
                map[r] = off + before;
                list[off + before] = r;
            } else {
                map[r] = -1;
                loss[r] = 0.f;
                pred[r] = -1;
            }
        }
        __syncthreads();
        if (tid == 0) {
            int t = running;
            for (int w = 0; w < 16; ++w) t += wave_tot[w];
            running = t;
        }
        __syncthreads();
    }
    if (tid == 0) rowmap[0] = running;
}

// ---- (R, K) row movers -----------------------------------------------------------
// MODE 0: dst[i] = src[list[i]]            (gather the kept rows)
// MODE 1: dst[i] = g[list[i]] * src[i]     (upstream gradient folded into the compact rows)
// MODE 2: dst[r] = map[r] >= 0 ? g[r] * src[map[r]] : 0   (scatter back, pad rows zero)
template <int MODE>
__global__ __launch_bounds__(CE_THREADS) void ce_rows_kernel(const float* __restrict__ src, const float* __restrict__ g,
                                                             const int32_t* __restrict__ rowmap, int R, int K,
                                                             float* __restrict__ dst)
{
    const int i = blockIdx.x;
    const int count = rowmap[0];
    const float* s;
    float scale = 1.f;
    if (MODE == 2) {
        const int m = rowmap[1 + i];
        s = (m >= 0) ? src + (long long)m * K : nullptr;
        if (m >= 0) scale = g[i];
    } else {
        if (i >= count) return;
        const int r = rowmap[1 + R + i];
        s = src + (long long)(MODE == 0 ? r : i) * K;
        if (MODE == 1) scale = g[r];
    }
    float* d = dst + (long long)i * K;
    if ((K & 3) == 0) {
        for (int c = threadIdx.x; c < (K >> 2); c += CE_THREADS) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (s) v = reinterpret_cast<const f32x4*>(s)[c];
            if (MODE != 0) v *= scale;
            reinterpret_cast<f32x4*>(d)[c] = v;
        }
    } else {
        for (int c = threadIdx.x; c < K; c += CE_THREADS) d[c] = s ? s[c] * scale : 0.f;
    }
}

// ---- soft-max cross-entropy over one row of logits ---------------------------------
__device__ __forceinline__ void block_max_arg(float& v, int& idx, float* sv, int* si)
{
    // larger value wins; on equal values the lower index (torch.argmax returns the first maximum)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(idx, o);
        if (ov > v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sv[wave] = v;
        si[wave] = idx;
    }
    __syncthreads();
    v = sv[0];
    idx = si[0];
#pragma unroll
    for (int w = 1; w < CE_THREADS / 64; ++w) {
        if (sv[w] > v || (sv[w] == v && si[w] < idx)) {
            v = sv[w];
            idx = si[w];
        }
    }
    __syncthreads();
}

__device__ __forceinline__ float block_sum(float v, float* sv)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sv[wave] = v;
    __syncthreads();
    float t = sv[0];
#pragma unroll
    for (int w = 1; w < CE_THREADS / 64; ++w) t += sv[w];
    __syncthreads();
    return t;
}

// One block per compact row.  CACHED: the row lives in registers between the
// three passes (max / sum of exponentials / write), so memory sees one read and
// one write; otherwise the passes re-read it.  ldp is a multiple of 4 and P is
// 16-byte aligned; columns [V, ldp) are never read as data nor written.
template <bool CACHED>
__global__ __launch_bounds__(CE_THREADS, 4) void ce_softmax_kernel(float* __restrict__ P, long long ldp, int V,
                                                                const int64_t* __restrict__ targets,
                                                                const int32_t* __restrict__ rowmap, int R,
                                                                float* __restrict__ loss, int32_t* __restrict__ pred)
{
    __shared__ float sv[CE_THREADS / 64];
    __shared__ int si[CE_THREADS / 64];
    const int i = blockIdx.x;
    if (i >= rowmap[0]) return;
    const int r = rowmap[1 + R + i];
    const int tgt = (int)targets[r];
    float* row = P + (long long)i * ldp;
    const int nv4 = (V + 3) >> 2;
    const float NEG_INF = -__builtin_inff();

    f32x4 x[CACHED ? CE_CACHE_V4 : 1];
    float mx = NEG_INF;
    int arg = 0x7fffffff;
    if (CACHED) {
#pragma unroll
        for (int t = 0; t < CE_CACHE_V4; ++t) {
            const int c4 = threadIdx.x + t * CE_THREADS;
            f32x4 v = {NEG_INF, NEG_INF, NEG_INF, NEG_INF};
            if (c4 < nv4) v = reinterpret_cast<const f32x4*>(row)[c4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * c4 + q >= V) v[q] = NEG_INF;
            x[t] = v;
        }
#pragma unroll
        for (int t = 0; t < CE_CACHE_V4; ++t) {
            const int c4 = threadIdx.x + t * CE_THREADS;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (x[t][q] > mx) {  // strictly greater: a thread's columns ascend, the first maximum stays
                    mx = x[t][q];
                    arg = 4 * c4 + q;
                }
        }
    } else {
        for (int c4 = threadIdx.x; c4 < nv4; c4 += CE_THREADS) {
            const f32x4 v = reinterpret_cast<const f32x4*>(row)[c4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * c4 + q < V && v[q] > mx) {
                    mx = v[q];
                    arg = 4 * c4 + q;
                }
        }
    }
    block_max_arg(mx, arg, sv, si);

    float sum = 0.f;
    if (CACHED) {
#pragma unroll
        for (int t = 0; t < CE_CACHE_V4; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float e = expf(x[t][q] - mx);  // exp(-inf) = 0 for the columns past V
                x[t][q] = e;
                sum += e;
            }
        }
    } else {
        for (int c4 = threadIdx.x; c4 < nv4; c4 += CE_THREADS) {
            const f32x4 v = reinterpret_cast<const f32x4*>(row)[c4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * c4 + q < V) sum += expf(v[q] - mx);
        }
    }
    sum = block_sum(sum, sv);
    const float inv = 1.f / sum;

    // loss = log(sum) + max - logit[target]; the target's logit is read back from
    // the row before it is overwritten (one thread, one dword)
    if (threadIdx.x == 0) {
        // a target outside [0, V) (torch raises for it) yields NaN instead of an out-of-bounds read
        loss[r] = (tgt >= 0 && tgt < V) ? logf(sum) + mx - row[tgt] : __builtin_nanf("");
        pred[r] = arg;
    }
    __syncthreads();

    if (CACHED) {
#pragma unroll
        for (int t = 0; t < CE_CACHE_V4; ++t) {
            const int c4 = threadIdx.x + t * CE_THREADS;
            if (c4 >= nv4) continue;
            f32x4 p = x[t] * inv;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * c4 + q == tgt) p[q] -= 1.f;
            if (4 * c4 + 3 < V) {
                reinterpret_cast<f32x4*>(row)[c4] = p;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (4 * c4 + q < V) row[4 * c4 + q] = p[q];
            }
        }
    } else {
        for (int c4 = threadIdx.x; c4 < nv4; c4 += CE_THREADS) {
            f32x4 p = reinterpret_cast<const f32x4*>(row)[c4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                p[q] = expf(p[q] - mx) * inv;
                if (4 * c4 + q == tgt) p[q] -= 1.f;
            }
            if (4 * c4 + 3 < V) {
                reinterpret_cast<f32x4*>(row)[c4] = p;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (4 * c4 + q < V) row[4 * c4 + q] = p[q];
            }
        }
    }
}

// ---- bias gradient: db[v] = sum_i g[list[i]] * P[i][v] -----------------------------
// Stage 1: block (column chunk of 256, row segment) -> partial[seg][v], rows in
// ascending order; stage 2 sums the segments in order (bitwise reproducible).
constexpr int DB_SEGS = 16;

__global__ __launch_bounds__(256) void ce_bias_grad_stage1(const float* __restrict__ P, long long ldp, int V,
                                                           const float* __restrict__ g,
                                                           const int32_t* __restrict__ rowmap, int R,
                                                           float* __restrict__ partial)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int seg = blockIdx.y;
    const int count = rowmap[0];
    const int per = (count + DB_SEGS - 1) / DB_SEGS;
    const int i0 = seg * per, i1 = min(count, i0 + per);
    const int32_t* list = rowmap + 1 + R;
    float acc = 0.f;
    if (v < V) {
        int i = i0;
        for (; i + 8 <= i1; i += 8) {
            float x[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x[u] = P[(long long)(i + u) * ldp + v];
                w[u] = g[list[i + u]];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += w[u] * x[u];
        }
        for (; i < i1; ++i) acc += g[list[i]] * P[(long long)i * ldp + v];
        partial[(long long)seg * V + v] = acc;
    }
}

__global__ __launch_bounds__(256) void ce_bias_grad_stage2(const float* __restrict__ partial, int V, float* __restrict__ db)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < DB_SEGS; ++k) s += partial[(long long)k * V + v];
    db[v] = s;
}

}  // namespace

int ce_rowmap(const int64_t* targets, int64_t R, int64_t padidx, int32_t* rowmap, float* loss, int32_t* pred, hipStream_t s)
{
    hipLaunchKernelGGL(ce_rowmap_kernel, dim3(1), dim3(1024), 0, s, targets, (int)R, padidx, rowmap, loss, pred);
    return check_launch("ce_rowmap_kernel");
}

int ce_rows(int mode, const float* src, const float* g, const int32_t* rowmap, int64_t R, int K, float* dst, hipStream_t s)
{
    const dim3 grid((unsigned)R), block(CE_THREADS);
    if (mode == 0)
        hipLaunchKernelGGL(ce_rows_kernel<0>, grid, block, 0, s, src, g, rowmap, (int)R, K, dst);
    else if (mode == 1)
        hipLaunchKernelGGL(ce_rows_kernel<1>, grid, block, 0, s, src, g, rowmap, (int)R, K, dst);
    else
        hipLaunchKernelGGL(ce_rows_kernel<2>, grid, block, 0, s, src, g, rowmap, (int)R, K, dst);
    return check_launch("ce_rows_kernel");
}

int ce_softmax(float* P, int64_t ldp, int V, const int64_t* targets, const int32_t* rowmap, int64_t R, float* loss,
               int32_t* pred, hipStream_t s)
{
    ProfScope prof(s, "ce_softmax", R, V, 0);
    const dim3 grid((unsigned)R), block(CE_THREADS);
    if ((V + 3) / 4 <= CE_CACHE_V4 * CE_THREADS)
        hipLaunchKernelGGL(ce_softmax_kernel<true>, grid, block, 0, s, P, (long long)ldp, V, targets, rowmap, (int)R, loss, pred);
    else
        hipLaunchKernelGGL(ce_softmax_kernel<false>, grid, block, 0, s, P, (long long)ldp, V, targets, rowmap, (int)R, loss, pred);
    return check_launch("ce_softmax_kernel");
}

size_t ce_bias_grad_workspace(int V) { return (size_t)DB_SEGS * V * sizeof(float); }

int ce_bias_grad(const float* P, int64_t ldp, int V, const float* g, const int32_t* rowmap, int64_t R, float* partial,
                 float* db, hipStream_t s)
{
    const int chunks = (V + 255) / 256;
    hipLaunchKernelGGL(ce_bias_grad_stage1, dim3(chunks, DB_SEGS), dim3(256), 0, s, P, (long long)ldp, V, g, rowmap, (int)R, partial);
    if (int rc = check_launch("ce_bias_grad_stage1")) return rc;
    hipLaunchKernelGGL(ce_bias_grad_stage2, dim3(chunks), dim3(256), 0, s, partial, V, db);
    return check_launch("ce_bias_grad_stage2");
}

}  // namespace gtg
