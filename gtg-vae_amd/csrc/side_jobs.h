// Side jobs: small reductions that do not depend on the launch that carries them and ride in extra blocks at
// the front of its grid instead of paying a kernel boundary of their own (a dependent launch in a replayed HIP
// graph costs about 4.5 us however little it does).  A carrier is a batch-sized GEMM launch (gemm_f32.hip) or,
// when no launch can take them, side_jobs_kernel (graph_ops.hip).  Every job computes exactly what its
// stand-alone kernel computes, in the same order of additions, whatever the carrier's block size:
//   SIDE_COLSUM        out[c] = sum over rows of X[r][c]        (colsum_pair_kernel: 64 row lanes, lanes summed in order)
//   SIDE_SUM           out[0] = scale * sum(x[0..n))            (final_sum_kernel: 256 strided sums, wave tree, (0+1)+(2+3))
//   SIDE_ADAM_PREPARE  step counter and bias corrections of Adam (adam_prepare_kernel)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gtg {

enum SideKind : int { SIDE_NONE = 0, SIDE_COLSUM = 1, SIDE_SUM = 2, SIDE_ADAM_PREPARE = 3 };

// Device-resident Adam state (16 bytes): the step counter lives on the device so that a captured step replays.
struct AdamState {
    int32_t step;
    float step_size;     // lr / (1 - beta1^step)
    float inv_sqrt_bc2;  // 1 / sqrt(1 - beta2^step)
    float pad;
};

struct SideJob {
    int kind;
    int rows, cols, ld, vec;  // SIDE_COLSUM: X is rows x cols with row pitch ld; vec: 16-byte loads allowed
    const int* rows_dev;      // SIDE_COLSUM, optional: rows = min(rows, mult * ceil(*rows_dev / unit))
    int unit, mult;
    const float* x;           // SIDE_COLSUM: X; SIDE_SUM: the addends
    float* out;               // SIDE_ADAM_PREPARE: the AdamState
    long long n;              // SIDE_SUM: number of addends
    float scale;              // SIDE_SUM
    float lr, b1, b2;         // SIDE_ADAM_PREPARE
};

constexpr int SIDE_MAX_JOBS = 6;
constexpr int SIDE_COLS = 64;         // columns per block of a column sum
constexpr int SIDE_MAX_ROWS = 1024;   // a column sum is one chunk of rows
constexpr int SIDE_SMEM_FLOATS = 64 * (SIDE_COLS / 4) * 4;  // LDS a carrier must offer (16 KB)

struct SideJobs {
    int count = 0;
    int first_block[SIDE_MAX_JOBS + 1] = {0};  // job j owns blocks [first_block[j], first_block[j + 1])
    SideJob job[SIDE_MAX_JOBS];

    int blocks() const { return first_block[count]; }
    bool full() const { return count >= SIDE_MAX_JOBS; }
    // each returns false when the job cannot ride (the caller then runs the stand-alone kernel)
    // rows_dev (optional, device scalar): only min(rows, mult * ceil(*rows_dev / unit)) rows are valid -- the
    // per-row-block column sums a GEMM with a device-side row count leaves (two per live 128-row tile)
    bool add_colsum(const float* X, long long rows, int cols, int ld, float* out, const int* rows_dev = nullptr,
                    int unit = 1, int mult = 1)
    {
        if (full() || rows < 1 || rows > SIDE_MAX_ROWS || cols < 1 || unit < 1) return false;
        SideJob& j = job[count];
        j = SideJob();
        j.kind = SIDE_COLSUM;
        j.x = X;
        j.out = out;
        j.rows = (int)rows;
        j.cols = cols;
        j.ld = ld;
        j.vec = ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && (ld % 4 == 0);
        j.rows_dev = rows_dev;
        j.unit = unit;
        j.mult = mult;
        push((cols + SIDE_COLS - 1) / SIDE_COLS);
        return true;
    }
    bool add_sum(const float* x, long long n, float scale, float* out)
    {
        if (full() || n < 0) return false;
        SideJob& j = job[count];
        j = SideJob();
        j.kind = SIDE_SUM;
        j.x = x;
        j.out = out;
        j.n = n;
        j.scale = scale;
        push(1);
        return true;
    }
    bool add_adam_prepare(void* state16, float lr, float b1, float b2)
    {
        if (full() || state16 == nullptr) return false;
        SideJob& j = job[count];
        j = SideJob();
        j.kind = SIDE_ADAM_PREPARE;
        j.out = reinterpret_cast<float*>(state16);
        j.lr = lr;
        j.b1 = b1;
        j.b2 = b2;
        push(1);
        return true;
    }

private:
    void push(int nblocks)
    {
        first_block[count + 1] = first_block[count] + nblocks;
        ++count;
        for (int k = count + 1; k <= SIDE_MAX_JOBS; ++k) first_block[k] = first_block[count];
    }
};

#if defined(__HIPCC__)
// Block `block` (< sj.blocks()) of the side jobs, run by all NT threads of a block (NT a multiple of 256, at most
// 1024); `smem` offers SIDE_SMEM_FLOATS floats, 16-byte aligned.  Block-uniform control flow throughout.
template <int NT>
__device__ __forceinline__ void side_run(const SideJobs& sj, const int block, float* const smem)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    static_assert(NT % 256 == 0 && NT <= 1024, "side jobs are written for blocks of 256, 512 or 1024 threads");
    int ji = 0;
#pragma unroll
    for (int k = 1; k < SIDE_MAX_JOBS; ++k)
        if (k < sj.count && block >= sj.first_block[k]) ji = k;
    const SideJob& j = sj.job[ji];
    const int jb = block - sj.first_block[ji];
    const int tid = (int)threadIdx.x;
    if (j.kind == SIDE_COLSUM) {
        // 64 virtual row lanes (lane v sums rows v, v + 64, ...), NT / 16 of them per pass of the block's threads
        constexpr int RL = NT / 16, PASSES = 64 / RL;
        f32x4* const sm = reinterpret_cast<f32x4*>(smem);  // [64][16]
        const int cg = tid & 15, rl = tid >> 4;
        const int col = jb * SIDE_COLS + cg * 4;
        f32x4 s[PASSES];
#pragma unroll
        for (int u = 0; u < PASSES; ++u) s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        int rows = j.rows;
        if (j.rows_dev != nullptr) rows = min(rows, j.mult * ((j.rows_dev[0] + j.unit - 1) / j.unit));
        if (col < j.cols) {
            if (j.vec && col + 3 < j.cols) {
                // whole groups of 64 rows without a guard, several groups' loads in flight per thread (a guarded
                // load waits for its predecessor); the additions into each row lane keep their order
                const float* x = j.x + (long long)rl * j.ld + col;
                int r0 = 0;
#pragma unroll 4
                for (; r0 + 64 <= rows; r0 += 64) {
#pragma unroll
                    for (int u = 0; u < PASSES; ++u)
                        s[u] += *reinterpret_cast<const f32x4*>(x + (long long)(r0 + u * RL) * j.ld);
                }
#pragma unroll
                for (int u = 0; u < PASSES; ++u) {
                    const int r = r0 + u * RL + rl;
                    if (r < rows) s[u] += *reinterpret_cast<const f32x4*>(j.x + (long long)r * j.ld + col);
                }
            } else {
                for (int r0 = 0; r0 < rows; r0 += 64) {
#pragma unroll
                    for (int u = 0; u < PASSES; ++u) {
                        const int r = r0 + u * RL + rl;
                        if (r < rows) {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (col + q < j.cols) s[u][q] += j.x[(long long)r * j.ld + col + q];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PASSES; ++u) sm[(u * RL + rl) * 16 + cg] = s[u];
        __syncthreads();
        if (rl == 0 && col < j.cols) {
            f32x4 t = sm[cg];
#pragma unroll 8
            for (int k = 1; k < 64; ++k) t += sm[k * 16 + cg];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (col + q < j.cols) j.out[col + q] = t[q];
        }
    } else if (j.kind == SIDE_SUM) {
        // the first 256 threads take the strided sums; their four waves are combined as (0 + 1) + (2 + 3)
        float s = 0.f;
        if (tid < 256)
            for (long long i = tid; i < j.n; i += 256) s += j.x[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (tid < 256 && (tid & 63) == 0) smem[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) j.out[0] = ((smem[0] + smem[1]) + (smem[2] + smem[3])) * j.scale;
    } else if (j.kind == SIDE_ADAM_PREPARE) {
        if (tid == 0) {
            AdamState* st = reinterpret_cast<AdamState*>(j.out);
            const int32_t step = st->step + 1;
            st->step = step;
            const double bc1 = 1.0 - pow((double)j.b1, (double)step);
            const double bc2 = 1.0 - pow((double)j.b2, (double)step);
            st->step_size = (float)((double)j.lr / bc1);
            st->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
        }
    }
}
#endif

// the side jobs as a launch of their own (graph_ops.hip)
int side_jobs_launch(const SideJobs& sj, hipStream_t s);

}  // namespace gtg
