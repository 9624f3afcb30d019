// fp32-in / fp32-accumulate GEMM on the gfx950 matrix cores
// (v_mfma_f32_32x32x2_f32), LDS-tiled, register-staged double buffering.
//
// One kernel template serves every dense contraction of the graph-VAE step:
//   NT  C = A * W^T        activations x Linear / GCN weights (both K-contiguous)
//   NN  C = A * B          dY * W (input gradients), A_hat (x) T (block-diagonal
//                          normalised adjacency times node features, batched)
//   TN  C = A^T * B        weight gradients (reduction over B*N rows, split-K)
// Tiles are 64-lane-wavefront shaped: every wave owns TM x TN sub-tiles of
// 32x32 whose accumulators live in 16 registers each; operand fragments are
// one float per lane per MFMA (lane l: row l&31, k-slot l>>5).
//
// k order inside an 8-wide k group is permuted (lane half h, step s -> k = 4h+s)
// so that K-contiguous tiles are read from LDS with one 16-byte ds_read_b128
// per 4 MFMAs; A and B use the same permutation, so the product is unchanged.
#include "gtg_internal.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace gtg {

thread_local int g_matrix_mode = GTG_MATRIX_F32;

namespace {

constexpr int BK = 32;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct KArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* aux;
    float* pool_val;
    int32_t* pool_idx;
    long long strideA, strideB, strideC, strideAux, strideBias;
    int M, N, K;
    int lda, ldb, ldc, ldaux;
    int tiles_m, tiles_n, splits, k_per_split;
    int epi;
    int vecA, vecB, vecC, vecAux;
    const int* m_dev;      // optional device row count (M = min(M, *m_dev))
    const int* k_dev;      // optional device depth (K = min(K, *k_dev))
    const int* batch_dim;  // optional per-batch M = K (ragged block-diagonal operands)
    const int* batch_off;  // optional per-batch row offset of B and/or C (compact node rows)
    int off_b, off_c;
    const int* proto_base;   // pooled epilogue: row of B where the M prototype rows start (device scalar)
    int to_slabs;  // write raw partials to C viewed as [batch*splits][M][N]
    float* tail_ws;         // tail split: accumulator slabs, one per (tail tile, part)
    int32_t* tail_counters; // arrivals per tail tile
    int tail_slots;         // resident block slots of the device for this kernel
    float* colsum_part;     // optional: column sums of C per wave row-block, (tiles_m * WAVES_M, N)
    // PREC 4: A and B point at bf16 plane images [3][K/16][rows][16]; lda / ldb are then the `rows` extents of the
    // images and these the distances between planes, in bf16 elements
    long long planeA, planeB;
    // PREC 2: ABSMAX_PARTS partial maxima of |A| followed by as many of |B| (absmax_pair_kernel), from which every
    // block derives the two power-of-two operand scales
    const float* absmax;
    // PREC 5 (fp16 plane images [2][K/16][rows][16]): inverse row scales of A (M) and of B (N)
    const float* inv_a_rows;
    const float* inv_b_rows;
    GemmFuse fuse;  // EPI_SIGMOID_BCE / EPI_LATENT_BWD
};

constexpr int ABSMAX_PARTS = GEMM_ABSMAX_PARTS;

// Blocks are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).
// Give each XCD a contiguous run of tile ids so that the tiles that re-read
// one operand panel sit behind one L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, rem = nwg & 7;
    const int xcd = bid & 7, within = bid >> 3;
    const int base = (xcd < rem) ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    return base + within;
}

// One operand tile: ROWS (the M or N extent) x BK.
//   KC  : K contiguous in memory; LDS image [ROWS][BK+4]
//   !KC : ROWS contiguous in memory; LDS image [BK][ROWS+4]
// The +4 float pad keeps 16-byte alignment and makes both the ds_write_b128
// of staging and the ds_read_b128 / ds_read_b32 of the fragment reads
// bank-conflict-free.
template <int ROWS, bool KC, int NTHREADS>
struct Tile {
    static constexpr int LD = KC ? (BK + 4) : (ROWS + 4);
    static constexpr int SIZE = KC ? ROWS * LD : BK * LD;
    static constexpr int NV = ROWS * BK / 4 / NTHREADS;
    static_assert(NV >= 1 && NV * NTHREADS * 4 == ROWS * BK, "tile must split evenly over the staging threads");
    static constexpr int V4 = ROWS / 4;

    // Interior tiles with 16-byte-aligned rows take the branch-free path: NV
    // independent 16-byte loads issued back to back.
    __device__ static __forceinline__ void fetch_fast(f32x4 (&reg)[NV], const float* __restrict__ base, int ld, int row0,
                                                      int k0, int tid)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * NTHREADS;
            if (KC) {
                const int row = idx >> 3, c4 = idx & 7;
                reg[i] = *reinterpret_cast<const f32x4*>(base + (long long)(row0 + row) * ld + (k0 + c4 * 4));
            } else {
                const int kk = idx / V4, c4 = idx % V4;
                reg[i] = *reinterpret_cast<const f32x4*>(base + (long long)(k0 + kk) * ld + (row0 + c4 * 4));
            }
        }
    }

    __device__ static __forceinline__ void fetch(f32x4 (&reg)[NV], const float* __restrict__ base, int ld, int row0,
                                                 int rows_total, int k0, int k_end, bool vec, int tid)
    {
        if (vec && row0 + ROWS <= rows_total && k0 + BK <= k_end) {  // block-uniform
            fetch_fast(reg, base, ld, row0, k0, tid);
            return;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * NTHREADS;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (KC) {
                const int row = idx >> 3, c4 = idx & 7;
                const int gr = row0 + row, gk = k0 + c4 * 4;
                if (gr < rows_total && gk < k_end) {
                    const float* p = base + (long long)gr * ld + gk;
                    if (vec && gk + 3 < k_end) {
                        v = *reinterpret_cast<const f32x4*>(p);
                    } else {
                        v[0] = p[0];
                        if (gk + 1 < k_end) v[1] = p[1];
                        if (gk + 2 < k_end) v[2] = p[2];
                        if (gk + 3 < k_end) v[3] = p[3];
                    }
                }
            } else {
                const int kk = idx / V4, c4 = idx % V4;
                const int gk = k0 + kk, gr = row0 + c4 * 4;
                if (gk < k_end && gr < rows_total) {
                    const float* p = base + (long long)gk * ld + gr;
                    if (vec && gr + 3 < rows_total) {
                        v = *reinterpret_cast<const f32x4*>(p);
                    } else {
                        v[0] = p[0];
                        if (gr + 1 < rows_total) v[1] = p[1];
                        if (gr + 2 < rows_total) v[2] = p[2];
                        if (gr + 3 < rows_total) v[3] = p[3];
                    }
                }
            }
            reg[i] = v;
        }
    }

    __device__ static __forceinline__ void stage(float* s, const f32x4 (&reg)[NV], int tid)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * NTHREADS;
            if (KC) {
                const int row = idx >> 3, c4 = idx & 7;
                *reinterpret_cast<f32x4*>(&s[row * LD + c4 * 4]) = reg[i];
            } else {
                const int kk = idx / V4, c4 = idx % V4;
                *reinterpret_cast<f32x4*>(&s[kk * LD + c4 * 4]) = reg[i];
            }
        }
    }

    // The 4 k values (k = 8g + 4h + s, s = 0..3) of row rbase + r.
    __device__ static __forceinline__ f32x4 frag(const float* s, int rbase, int g, int r, int h)
    {
        if (KC) {
            return *reinterpret_cast<const f32x4*>(&s[(rbase + r) * LD + 8 * g + 4 * h]);
        } else {
            const float* p = &s[(8 * g + 4 * h) * LD + rbase + r];
            f32x4 v = {p[0], p[LD], p[2 * LD], p[3 * LD]};
            return v;
        }
    }
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// One operand tile for the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16): ROWS x BK
// values kept as PLANES bf16 "planes" whose sum is the fp32 value.
//   PLANES = 1: the value rounded to bf16 (round to nearest even).
//   PLANES = 3: an exact three-way split by truncation, x = h0 + h1 + h2 with
//               8 significant bits each (fp32 has 24), so that the six products
//               h_i*h_j, i + j <= 2, summed in fp32 reproduce the fp32 product
//               to ~2^-22 relative.
//   PLANES = 2: the planes hold fp16, not bf16: x * scale = h0 + h1 + (at most 2^-22 of it), h0 the value
//               rounded to nearest, h1 what is left rounded to nearest (11 significant bits each and the sign
//               of the remainder); the three products h0*h0, h0*h1, h1*h0 reproduce the fp32 product as
//               closely as the six of PLANES = 3.  `scale` is a power of two that puts the operand's largest
//               magnitude just below 2^14 (fp16 reaches 65504), so nothing overflows and only values 2^-28 of
//               the largest and smaller lose relative precision.
// Either way a lane's MFMA fragment is (row r, 8 consecutive k).
//   KC  (K contiguous in memory): every thread owns pieces of (row, 8 k), two 16-byte loads each; LDS image
//        per plane [ROWS][TK bf16 + 16 B pad], a fragment is one conflict-free ds_read_b128.
//   !KC (rows contiguous in memory): every thread owns pieces of (k, 4 rows), one 16-byte load each (a wave
//        instruction covers two k values x 128 rows, 512 contiguous bytes apiece); the LDS image keeps that
//        orientation, [TK][ROWS bf16 + 64 B pad], so staging is one 8-byte store per piece and plane, and the
//        transposition happens in the read: ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a
//        4 (k) x 16 (rows) block, i.e. four consecutive k of one row; two of them make the fragment.  The 64-byte
//        pad puts the four k rows of a block in disjoint bank windows (conflict-free per 32-lane half).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int ROWS, bool KC, int NTHREADS, int PLANES, int TK>
struct TileB {
    static constexpr int ROWB = KC ? 2 * TK + 16 : 2 * ROWS + 64;  // bytes per LDS row per plane
    static constexpr int PLANE = (KC ? ROWS : TK) * ROWB / 4;      // floats per plane
    static constexpr int SIZE = PLANES * PLANE;                    // floats per tile
    static constexpr int NV = ROWS * TK / 4 / NTHREADS;            // 16-byte registers per thread
    // K-contiguous pieces hold 8 k values (two registers, one 16-byte LDS store per plane) when a thread owns an
    // even number of registers, else 4 (one register, an 8-byte store)
    static constexpr int KP = (NV % 2 == 0) ? 8 : 4;
    static constexpr int NP = KC ? NV * 4 / KP : NV;               // pieces per thread
    static_assert(NV >= 1 && NV * NTHREADS * 4 == ROWS * TK, "tile must split evenly over the staging threads");
    static_assert(KC || ROWS % 128 == 0, "the transposed image's bank layout assumes 256-byte rows");

    // KC: piece = (row, KP k starting at KP kb); !KC: piece = (k = kb, 4 rows starting at row)
    __device__ static __forceinline__ void piece(int p, int& row, int& kb)
    {
        if (KC) {
            row = p / (TK / KP);
            kb = p % (TK / KP);
        } else {
            row = 4 * (p % (ROWS / 4));
            kb = p / (ROWS / 4);
        }
    }

    __device__ static __forceinline__ void fetch(f32x4 (&reg)[NV], const float* __restrict__ base, int ld, int row0,
                                                 int rows_total, int k0, int k_end, bool vec, int tid)
    {
        const bool interior = row0 + ROWS <= rows_total && k0 + TK <= k_end;  // block-uniform
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int row, kb;
            piece(tid + i * NTHREADS, row, kb);
            if (KC && KP == 4) {
                const int gr = row0 + row, gk = k0 + 4 * kb;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                const float* p = base + (long long)gr * ld + gk;
                if (interior && vec) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else if (gr < rows_total) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (gk + q < k_end) v[q] = p[q];
                }
                reg[i] = v;
            } else if (KC) {
                const int gr = row0 + row, gk = k0 + 8 * kb;
                f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
                const float* p = base + (long long)gr * ld + gk;
                if (interior && vec) {
                    lo = *reinterpret_cast<const f32x4*>(p);
                    hi = *reinterpret_cast<const f32x4*>(p + 4);
                } else if (gr < rows_total) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (gk + q < k_end) lo[q] = p[q];
                        if (gk + 4 + q < k_end) hi[q] = p[4 + q];
                    }
                }
                reg[2 * i] = lo;
                reg[2 * i + 1] = hi;
            } else {
                const int gr = row0 + row, gk = k0 + kb;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                const float* p = base + (long long)gk * ld + gr;
                if (interior && vec) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else if (gk < k_end) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (gr + q < rows_total) v[q] = p[q];
                }
                reg[i] = v;
            }
        }
    }

    // interior tile, aligned rows: no guards, no branches (the waits on these loads can then be counted)
    __device__ static __forceinline__ void fetch_fast(f32x4 (&reg)[NV], const float* __restrict__ base, int ld, int row0,
                                                      int k0, int tid)
    {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int row, kb;
            piece(tid + i * NTHREADS, row, kb);
            if (KC && KP == 4) {
                reg[i] = *reinterpret_cast<const f32x4*>(base + (long long)(row0 + row) * ld + (k0 + 4 * kb));
            } else if (KC) {
                const float* p = base + (long long)(row0 + row) * ld + (k0 + 8 * kb);
                reg[2 * i] = *reinterpret_cast<const f32x4*>(p);
                reg[2 * i + 1] = *reinterpret_cast<const f32x4*>(p + 4);
            } else {
                reg[i] = *reinterpret_cast<const f32x4*>(base + (long long)(k0 + kb) * ld + (row0 + row));
            }
        }
    }

    // bf16 pair (low half = x0, high half = x1) of the top 16 bits of two fp32 words
    __device__ static __forceinline__ unsigned top_halves(float x0, float x1)
    {
        return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
    }

    // x0, x1 -> their bf16 pair in plane 0 (1, 2) and what is left of them
    __device__ static __forceinline__ unsigned take_plane(float& x0, float& x1)
    {
        const unsigned w = top_halves(x0, x1);
        x0 -= __uint_as_float(__float_as_uint(x0) & 0xffff0000u);  // exact
        x1 -= __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
        return w;
    }

    // x0, x1 -> their fp16 pair rounded to nearest and what is left of them (exact)
    __device__ static __forceinline__ unsigned take_half(float& x0, float& x1)
    {
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        const f16x2 hpair = {(_Float16)x0, (_Float16)x1};
        x0 -= (float)hpair[0];
        x1 -= (float)hpair[1];
        return __builtin_bit_cast(unsigned, hpair);
    }
    __device__ static __forceinline__ unsigned round_half(float x0, float x1)
    {
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        const f16x2 hpair = {(_Float16)x0, (_Float16)x1};
        return __builtin_bit_cast(unsigned, hpair);
    }

    __device__ static __forceinline__ void stage(float* s, const f32x4 (&reg)[NV], int tid, float scale = 1.f)
    {
        char* sb = reinterpret_cast<char*>(s);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int row, kb;
            piece(tid + i * NTHREADS, row, kb);
            if (KC && KP == 8) {
                float x[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    x[q] = reg[2 * i][q];
                    x[4 + q] = reg[2 * i + 1][q];
                }
                char* dst = sb + row * ROWB + kb * 16;
                if (PLANES == 1) {
                    u32x4 w;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned pk;
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(x[2 * q]), "v"(x[2 * q + 1]));
                        w[q] = pk;
                    }
                    *reinterpret_cast<u32x4*>(dst) = w;
                } else if (PLANES == 2) {
                    u32x4 w0, w1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned p0, p1;
                        split_pair_f16(x[2 * q], x[2 * q + 1], scale, p0, p1);
                        w0[q] = p0;
                        w1[q] = p1;
                    }
                    *reinterpret_cast<u32x4*>(dst) = w0;
                    *reinterpret_cast<u32x4*>(dst + PLANE * 4) = w1;
                } else {
                    u32x4 w0, w1, w2;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        w0[q] = take_plane(x[2 * q], x[2 * q + 1]);
                        w1[q] = take_plane(x[2 * q], x[2 * q + 1]);
                        w2[q] = top_halves(x[2 * q], x[2 * q + 1]);  // <= 8 bits left
                    }
                    *reinterpret_cast<u32x4*>(dst) = w0;
                    *reinterpret_cast<u32x4*>(dst + PLANE * 4) = w1;
                    *reinterpret_cast<u32x4*>(dst + 2 * PLANE * 4) = w2;
                }
            } else {
                // four values: consecutive k of one row (K contiguous) or consecutive rows at one k
                float x[4] = {reg[i][0], reg[i][1], reg[i][2], reg[i][3]};
                char* dst = KC ? sb + row * ROWB + kb * 8 : sb + kb * ROWB + row * 2;
                if (PLANES == 1) {
                    u32x2 w;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        unsigned pk;
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(x[2 * q]), "v"(x[2 * q + 1]));
                        w[q] = pk;
                    }
                    *reinterpret_cast<u32x2*>(dst) = w;
                } else if (PLANES == 2) {
                    u32x2 w0, w1;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        unsigned p0, p1;
                        split_pair_f16(x[2 * q], x[2 * q + 1], scale, p0, p1);
                        w0[q] = p0;
                        w1[q] = p1;
                    }
                    *reinterpret_cast<u32x2*>(dst) = w0;
                    *reinterpret_cast<u32x2*>(dst + PLANE * 4) = w1;
                } else {
                    u32x2 w0, w1, w2;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        w0[q] = take_plane(x[2 * q], x[2 * q + 1]);
                        w1[q] = take_plane(x[2 * q], x[2 * q + 1]);
                        w2[q] = top_halves(x[2 * q], x[2 * q + 1]);
                    }
                    *reinterpret_cast<u32x2*>(dst) = w0;
                    *reinterpret_cast<u32x2*>(dst + PLANE * 4) = w1;
                    *reinterpret_cast<u32x2*>(dst + 2 * PLANE * 4) = w2;
                }
            }
        }
    }

    // plane pl, row rbase + r, the 8 k values 16*g + 8*h .. +7.  (The transposed read needs every lane of the
    // wave active: it is only called from the block-uniform main loop.)
    __device__ static __forceinline__ bf16x8 frag(const float* s, int pl, int rbase, int g, int r, int h)
    {
        if (KC) {
            const char* p = reinterpret_cast<const char*>(s) + pl * (PLANE * 4) + (rbase + r) * ROWB + (2 * g + h) * 16;
            return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
        } else {
            // lane (r, h) sits in the 16-lane group 2h + (r >> 4) as lane r & 15: it receives row rbase + r; it
            // supplies the address of k row (r >> 2) & 3 of the group's 4 x 16 block, columns 4 (r & 3) .. + 3
            const char* p = reinterpret_cast<const char*>(s) + pl * (PLANE * 4) + (16 * g + 8 * h + ((r >> 2) & 3)) * ROWB +
                            (rbase + (r & 16) + 4 * (r & 3)) * 2;
            typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 4 * ROWB));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, v);
        }
    }
};

// Operands that arrive already split: three bf16 planes per operand, made once by split_planes_kernel (weights)
// or written that way by the kernel that produces the operand, in the k-blocked layout
//     [plane][K / 16][rows][16]   (bf16)
// so that the 16 k values of one k step of ALL rows of a tile are one contiguous run in memory (32 bytes per
// row): a tile's share of a k step is fetched in whole cache lines, one contiguous kilobyte per wave
// instruction, instead of 32 bytes out of each of 32 lines as a [rows][K] image would give.  A stage of the
// LDS ring holds the BM + BN rows of all six planes, rows back to back (no padding: lane (r, h) reads the 16
// bytes at row r, half h, so a wave's fragment read is one contiguous kilobyte, conflict-free).  The stage is
// filled by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes = 32 rows per instruction, no register and
// no vector instruction spent), three stages deep.
template <int BM, int BN, int NPL = 3>
struct PlaneStage {
    static constexpr int KS = 16;
    static constexpr int STAGES = 3;
    static constexpr int ROWB = 2 * KS;                         // bytes per row per plane
    static constexpr int PLANE_A = BM * ROWB, PLANE_B = BN * ROWB;
    static constexpr int STAGE_BYTES = NPL * (PLANE_A + PLANE_B);
    static constexpr int PIECES = STAGE_BYTES / 1024;           // wave instructions per stage
};

// One 32x32x16 product on 16-bit fragments: bf16, or fp16 (the fragments are carried as bf16x8 bit patterns).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x16 mfma_16bit(bf16x8 x, bf16x8 y, f32x16 c)
{
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
}

// WAVES_M x WAVES_N waves share one BM x BN tile; KG such wave groups split the
// block's K range between them (each with its own LDS buffers and its own
// loads) and their accumulators are summed through LDS, in group order, before
// the epilogue.  KG > 1 shortens the dependent load -> LDS -> MFMA chain of
// short-and-narrow GEMMs (batch-sized M, K of a few hundred) that have too few
// tiles to hide it behind other blocks.
// PREC selects the matrix pipe: 0 = fp32 operands (v_mfma_f32_32x32x2_f32), 1 = operands rounded to
// bf16, 3 = the exact three-plane bf16 split with six products per k step (both on
// v_mfma_f32_32x32x16_bf16, fp32 accumulate), 2 = operands scaled by a power of two and split into two fp16
// planes, three products per k step (v_mfma_f32_32x32x16_f16), the result scaled back.
// LDS floats of one block: the operand double buffers of every wave group, reused by the epilogue to
// stage the accumulators.
// Operand buffers per wave group: two (the next tile is staged while the current one is multiplied), except where
// a K group is a single wave -- its buffer is private to the wave, LDS executes a wave's accesses in order, so
// one buffer needs no barrier at all and half the LDS lets twice as many blocks share a CU.
// The fp16-plane kernel stages its accumulators for the epilogue in two halves (half the staging area).
template <int KG, int PREC>
constexpr bool gemm_epilogue_halves()
{
    return KG == 1 && PREC == 5;
}

template <int WAVES_M, int WAVES_N, int KG, int PREC>
constexpr int gemm_group_buffers()
{
    return (PREC == 0 && KG > 1 && WAVES_M * WAVES_N == 1) ? 1 : 2;
}

// Waves per SIMD the register allocation must leave room for: four where a K group is one wave, so that four
// blocks of four waves share a CU (their LDS allows it).
template <int WAVES_M, int WAVES_N, int KG, int PREC>
constexpr int gemm_min_waves()
{
    return (PREC == 0 && KG > 1 && WAVES_M * WAVES_N == 1) ? 4 : (PREC == 5 ? 3 : 1);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int KG, bool AKC, bool BKC, int PREC>
constexpr int gemm_smem_floats()
{
    constexpr int GT = 64 * WAVES_M * WAVES_N;
    constexpr int KS = (PREC == 3) ? 16 : BK;
    constexpr int TPL = (PREC == 0 || PREC == 4 || PREC == 5) ? 1 : PREC;
    using TA = typename std::conditional<PREC == 0, Tile<BM, AKC, GT>, TileB<BM, AKC, GT, TPL, KS>>::type;
    using TB = typename std::conditional<PREC == 0, Tile<BN, BKC, GT>, TileB<BN, BKC, GT, TPL, KS>>::type;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int MAIN = (PREC == 4)   ? PlaneStage<BM, BN, 3>::STAGES * PlaneStage<BM, BN, 3>::STAGE_BYTES / 4
                         : (PREC == 5) ? PlaneStage<BM, BN, 2>::STAGES * PlaneStage<BM, BN, 2>::STAGE_BYTES / 4
                                       : KG * gemm_group_buffers<WAVES_M, WAVES_N, KG, PREC>() * (TA::SIZE + TB::SIZE),
                  EPI = KG * WAVES_M * WAVES_N * (WTM / (gemm_epilogue_halves<KG, PREC>() && WTM == 64 ? 2 : 1)) * (WTN + 4);
    // + 4: the tail split's "last arriver" word lives behind the buffers, in the same array (a second __shared__
    // object beside an array that LDS-DMA fills makes the compiler drain all loads before every LDS read).  The
    // two-plane fp16 tiles (k step 32) fill exactly half a CU's LDS with the 128x128 tile's four buffers, and the
    // word sits in the unused pad that ends the last one, so that two blocks still share a CU.
    return (MAIN > EPI ? MAIN : EPI) + ((PREC == 2 && MAIN >= EPI + 4) ? 0 : 4);
}

// One block's share of a GEMM launch: block `block_x` of `grid_x`, LDS at `smem` (gemm_smem_floats() floats).
// A kernel of its own (gemm_kernel) or one of the two jobs of gemm_pair_kernel.
template <int BM, int BN, int WAVES_M, int WAVES_N, int KG, bool AKC, bool BKC, int PREC>
__device__ __forceinline__ void gemm_body(const KArgs& a, float* const smem, const int block_x, const int grid_x)
{
    constexpr int GT = 64 * WAVES_M * WAVES_N;  // threads of one wave group
    constexpr int GW = WAVES_M * WAVES_N;
    // k extent of one pipeline step: the three-plane tiles take 16 so that two blocks still fit a CU's LDS
    constexpr int KS = (PREC == 3 || PREC == 4 || PREC == 5) ? 16 : BK;
    constexpr int TPL = (PREC == 0 || PREC == 4 || PREC == 5) ? 1 : PREC;  // planes of the register-staged bf16 tiles
    using TA = typename std::conditional<PREC == 0, Tile<BM, AKC, GT>, TileB<BM, AKC, GT, TPL, KS>>::type;
    using TB = typename std::conditional<PREC == 0, Tile<BN, BKC, GT>, TileB<BN, BKC, GT, TPL, KS>>::type;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");

    // operand double buffers per wave group; reused by the epilogue to stage the accumulators
    constexpr int LDC = WTN + 4;
    constexpr int SMEM_GROUP = gemm_group_buffers<WAVES_M, WAVES_N, KG, PREC>() * (TA::SIZE + TB::SIZE);

    const int kg = (KG > 1) ? (int)threadIdx.x / GT : 0;
    const int tid = (KG > 1) ? (int)threadIdx.x % GT : (int)threadIdx.x;
    float* const sA0 = smem + kg * SMEM_GROUP;
    float* const sB0 = sA0 + TA::SIZE;
    float* const sA1 = sA0 + TA::SIZE + TB::SIZE;
    float* const sB1 = sA1 + TA::SIZE;

    // (the wave index in a scalar register: what depends on it alone -- tile offsets, the operand a wave stages --
    // is then scalar arithmetic and scalar branches instead of per-lane selects)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = (wave / WAVES_N) * WTM, wn = (wave % WAVES_N) * WTN;

    // effective extents: static, or read from device memory (compact node rows)
    int Mr = a.M, Kr = a.K;
    if (a.m_dev) Mr = min(Mr, a.m_dev[0]);
    if (a.k_dev) Kr = min(Kr, a.k_dev[0]);
    // With a device-side row count only the first ceil(Mr/BM) tile rows have
    // work.  The launch is sized for the static M; blocks past the live tile
    // count leave at once, and the live ones (block ids 0..live-1, dealt evenly
    // over the XCDs) are remapped among themselves.
    int live = grid_x;
    int tiles_m = a.tiles_m;
    if (a.m_dev) {
        tiles_m = (Mr + BM - 1) / BM;
        live = tiles_m * a.tiles_n * a.splits;  // batch == 1 whenever m_dev is set
    }
    // Tail split: `live` tiles over `slots` resident blocks leave a last round of
    // tail = live % slots tiles; cut each of those into `parts` K ranges so that
    // the round fills the slots and takes 1/parts as long.
    int parts = 1, part = 0, tail_tile = 0;
    int bid;
    if (KG == 1 && a.tail_ws != nullptr) {
        live = tiles_m * a.tiles_n;  // batch 1, no split-K in this mode; the launch is padded by `slots` blocks
        const int slots = a.tail_slots;
        const int full = live / slots * slots, tail = live - full;
        const int nk_all = (Kr + BK - 1) / BK;
        int s = tail > 0 ? slots / tail : 1;
        s = min(s, nk_all / 4);  // a part keeps at least four k steps
        s = max(s, 1);
        if (block_x >= full + tail * s) return;
        if (block_x < full) {
            bid = xcd_remap(block_x, full);
        } else {
            const int u = xcd_remap(block_x - full, tail * s);  // the parts of a tile share an XCD
            tail_tile = u / s;
            part = u % s;
            parts = s;
            bid = full + tail_tile;
        }
    } else {
        if (block_x >= live) return;
        bid = xcd_remap(block_x, live);
    }
    const int tn = bid % a.tiles_n;
    bid /= a.tiles_n;
    const int tm = bid % tiles_m;
    bid /= tiles_m;
    const int split = bid % a.splits;
    const int batch = bid / a.splits;

    const int m0 = tm * BM, n0 = tn * BN;
    long long row_off = 0;
    if (a.batch_dim) {
        const int nb = a.batch_dim[batch];
        Mr = min(Mr, nb);
        Kr = min(Kr, nb);
    }
    if (a.batch_off) row_off = a.batch_off[batch];
    // split-K over the effective depth, so that every split has an equal share
    int kps = a.k_per_split;
    if (a.k_dev && a.splits > 1) kps = ((Kr + a.splits - 1) / a.splits + BK - 1) / BK * BK;
    int k_begin = split * kps;
    int k_end = min(Kr, k_begin + kps);
    if (parts > 1) {
        const int per = (((Kr + BK - 1) / BK + parts - 1) / parts) * BK;
        k_begin = part * per;
        k_end = min(Kr, k_begin + per);
    }
    // every wave group runs the same number of iterations (block-wide barriers); a group whose share is
    // empty or short multiplies zero tiles
    int nk = (k_end - k_begin + KS - 1) / KS;
    if (KG > 1) {
        nk = (nk + KG - 1) / KG;
        k_begin += kg * nk * KS;
        k_end = min(k_end, k_begin + nk * KS);
    }
    const float* __restrict__ A = a.A + batch * a.strideA;
    const float* __restrict__ B = a.B + batch * a.strideB + (a.off_b ? row_off * a.ldb : 0);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // PREC 2: the power of two that puts each operand's largest magnitude in [2^13, 2^14), and its inverse.
    // Every block reduces the same partial maxima (left by the launch before this one) in the same order.
    float sc_a = 1.f, sc_b = 1.f, inv_a = 1.f, inv_b = 1.f;
    if constexpr (PREC == 2) {
        static_assert(KG == 1, "the scale prologue is written for one wave group");
        float ma = 0.f, mb = 0.f;
#pragma unroll
        for (int i = 0; i < ABSMAX_PARTS / GT; ++i) {
            ma = fmaxf(ma, a.absmax[i * GT + tid]);
            mb = fmaxf(mb, a.absmax[ABSMAX_PARTS + i * GT + tid]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ma = fmaxf(ma, __shfl_xor(ma, o));
            mb = fmaxf(mb, __shfl_xor(mb, o));
        }
        if (lane == 0) {
            smem[2 * wave] = ma;
            smem[2 * wave + 1] = mb;
        }
        __syncthreads();
        ma = smem[0];
        mb = smem[1];
#pragma unroll
        for (int w = 1; w < GW; ++w) {
            ma = fmaxf(ma, smem[2 * w]);
            mb = fmaxf(mb, smem[2 * w + 1]);
        }
        __syncthreads();  // before the first tile is staged over these words
        f16x2_scale(ma, sc_a, inv_a);
        f16x2_scale(mb, sc_b, inv_b);
    }

    if constexpr (PREC == 4 || PREC == 5) {
        static_assert(KG == 1 && WAVES_M * WAVES_N == 4, "the plane pipeline is written for four waves");
        constexpr int NPL = PREC == 4 ? 3 : 2;  // three bf16 planes, or two fp16 ones
        using PS = PlaneStage<BM, BN, NPL>;
        constexpr int PPW = PS::PIECES / 4;  // pieces (1 KB wave instructions) per wave and stage
        static_assert(PPW * 4 == PS::PIECES, "a stage must split evenly over the four waves");
        static_assert(PS::STAGES == 3 || PS::STAGES == 4, "the waits below are written out for rings of three and four");
        // (A ring of four 16 KB stages with the fragments of step kt + 1 read while the MFMAs of step kt run -- two
        // register sets, a branch-free steady-state loop -- took the fp16-plane kernel from 95.6 to 93.0 us at two
        // blocks per CU; three blocks per CU on a ring of three do better, 87.5 us, and gain nothing from that
        // double buffering (88.9 us), so the loop reads the stage it multiplies.)
        char* const lds = reinterpret_cast<char*>(smem);
        // this wave's pieces: piece q < NPL BM/32 is plane q / (BM/32), row block q % (BM/32) of A; the rest B alike.
        // Lane l fetches 16 bytes: row (l >> 1) of the block, k half (l & 1).  Rows past the operand's last one are
        // clamped (their products land in rows / columns the epilogue never stores).
        const __bf16* src[PPW];
        int dst[PPW];
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            const int q = wave * PPW + t;
            const bool isA = q < NPL * (BM / 32);
            const int qq = isA ? q : q - NPL * (BM / 32);
            const int rblocks = isA ? BM / 32 : BN / 32;
            const int pl = qq / rblocks, rb = qq % rblocks;
            const int row = rb * 32 + (lane >> 1);
            if (isA) {
                const int gr = min(m0 + row, Mr - 1);
                src[t] = reinterpret_cast<const __bf16*>(A) + pl * a.planeA +
                         ((long long)(k_begin / KS) * a.lda + gr) * KS + (lane & 1) * 8;
                dst[t] = pl * PS::PLANE_A + rb * 1024;
            } else {
                const int gc = min(n0 + row, a.N - 1);
                src[t] = reinterpret_cast<const __bf16*>(B) + pl * a.planeB +
                         ((long long)(k_begin / KS) * a.ldb + gc) * KS + (lane & 1) * 8;
                dst[t] = NPL * PS::PLANE_A + pl * PS::PLANE_B + rb * 1024;
            }
        }
        const long long stepA = (long long)a.lda * KS, stepB = (long long)a.ldb * KS;  // elements per k step
        auto issue = [&](int kt) {
            char* stage = lds + (kt % PS::STAGES) * PS::STAGE_BYTES;
#pragma unroll
            for (int t = 0; t < PPW; ++t) {
                const bool isA = wave * PPW + t < NPL * (BM / 32);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src[t] + kt * (isA ? stepA : stepB)),
                    (__attribute__((address_space(3))) void*)(stage + dst[t]), 16, 0, 0);
            }
        };
#pragma unroll
        for (int t = 0; t < PS::STAGES - 1; ++t)
            if (nk > t) issue(t);
        for (int kt = 0; kt < nk; ++kt) {
            // stage kt was issued STAGES - 1 iterations ago; the only younger loads are those of the up to
            // STAGES - 2 stages after it
            const int younger = min(PS::STAGES - 2, nk - 1 - kt);
            if (younger >= 2)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            else if (younger == 1)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // every wave's share of stage kt has landed, and every wave has finished reading stage kt - 1, whose
            // buffer stage kt + STAGES - 1 now overwrites
            __builtin_amdgcn_s_barrier();
            if (kt + PS::STAGES - 1 < nk) issue(kt + PS::STAGES - 1);
            const char* stage = lds + (kt % PS::STAGES) * PS::STAGE_BYTES;
            bf16x8 fa[TM][NPL], fb[TN][NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i][pl] = __builtin_bit_cast(
                        bf16x8, *reinterpret_cast<const u32x4*>(stage + pl * PS::PLANE_A + (wm + 32 * i + r) * PS::ROWB + h * 16));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j][pl] = __builtin_bit_cast(
                        bf16x8, *reinterpret_cast<const u32x4*>(stage + NPL * PS::PLANE_A + pl * PS::PLANE_B +
                                                                (wn + 32 * j + r) * PS::ROWB + h * 16));
            }
            // products of plane pairs (pa, pb) with pa + pb < NPL, the small ones first
#pragma unroll
            for (int sum = NPL - 1; sum >= 0; --sum)
#pragma unroll
                for (int pa = 0; pa <= sum; ++pa)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = mfma_16bit<NPL == 2>(fa[i][pa], fb[j][sum - pa], acc[i][j]);
        }
        __syncthreads();  // the epilogue reuses the stages
    } else if constexpr (PREC == 0 && KG > 1 && GW == 1) {
        // A K group of one wave (32x32 tiles): the group's staging buffer is private to its wave.  LDS executes a
        // wave's accesses in program order, so the tile written by the wave's own ds_writes is read back by its
        // ds_reads, and overwritten by the next tile's writes, without a barrier or a second buffer: the loop has
        // no block-wide synchronisation, the waves of a block drift freely, and with half the LDS and a ring of
        // two prefetched tiles in registers four blocks share a CU instead of two -- these launches are 2-4 blocks
        // per CU, so all of them start at once and the memory latency is paid once per launch, not once per round
        // of resident blocks.  The order of the products into each accumulator is unchanged.
        constexpr int FDEPTH = 2;
        f32x4 ra[FDEPTH][TA::NV], rb[FDEPTH][TB::NV];
#pragma unroll
        for (int t = 0; t < FDEPTH; ++t) {
            if (t < nk) {
                TA::fetch(ra[t], A, a.lda, m0, Mr, k_begin + t * BK, k_end, a.vecA, tid);
                TB::fetch(rb[t], B, a.ldb, n0, a.N, k_begin + t * BK, k_end, a.vecB, tid);
            }
        }
        for (int kt0 = 0; kt0 < nk; kt0 += FDEPTH) {
#pragma unroll
            for (int u = 0; u < FDEPTH; ++u) {
                const int kt = kt0 + u;
                if (kt >= nk) break;  // uniform over the wave
                TA::stage(sA0, ra[u], tid);
                TB::stage(sB0, rb[u], tid);
                if (kt + FDEPTH < nk) {
                    const int k0 = k_begin + (kt + FDEPTH) * BK;
                    TA::fetch(ra[u], A, a.lda, m0, Mr, k0, k_end, a.vecA, tid);
                    TB::fetch(rb[u], B, a.ldb, n0, a.N, k0, k_end, a.vecB, tid);
                }
                f32x4 fa[BK / 8], fb[BK / 8];
#pragma unroll
                for (int g = 0; g < BK / 8; ++g) {
                    fa[g] = TA::frag(sA0, wm, g, r, h);
                    fb[g] = TB::frag(sB0, wn, g, r, h);
                }
#pragma unroll
                for (int g = 0; g < BK / 8; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g][e], fb[g][e], acc[0][0], 0, 0, 0);
            }
        }
        __syncthreads();  // the epilogue stages the accumulators over every group's buffer
    } else if constexpr (PREC == 0 && KG > 1) {
        // K-group configurations run batch-sized GEMMs: a block sees only a handful of k steps, each far
        // shorter than a global load's latency, so the next FDEPTH tiles wait in registers (tile t in slot
        // t % FDEPTH, refilled with tile t + FDEPTH as soon as it has been written to LDS): the block pays
        // the memory latency about once instead of once per k step.  Three tiles: a fourth costs the
        // 32x32 configuration a resident block per CU in registers and measured slower.
        constexpr int FDEPTH = 3;
        f32x4 ra[FDEPTH][TA::NV], rb[FDEPTH][TB::NV];
#pragma unroll
        for (int t = 0; t < FDEPTH; ++t) {
            if (t < nk) {
                TA::fetch(ra[t], A, a.lda, m0, Mr, k_begin + t * BK, k_end, a.vecA, tid);
                TB::fetch(rb[t], B, a.ldb, n0, a.N, k_begin + t * BK, k_end, a.vecB, tid);
            }
        }
        if (nk > 0) {
            TA::stage(sA0, ra[0], tid);
            TB::stage(sB0, rb[0], tid);
            if (FDEPTH < nk) {
                TA::fetch(ra[0], A, a.lda, m0, Mr, k_begin + FDEPTH * BK, k_end, a.vecA, tid);
                TB::fetch(rb[0], B, a.ldb, n0, a.N, k_begin + FDEPTH * BK, k_end, a.vecB, tid);
            }
        }
        __syncthreads();
        for (int kt0 = 0; kt0 < nk; kt0 += FDEPTH) {
#pragma unroll
            for (int u = 0; u < FDEPTH; ++u) {
                const int kt = kt0 + u;
                if (kt >= nk) break;  // block-uniform
                const int nslot = (u + 1) % FDEPTH;
                const float* cA = (kt & 1) ? sA1 : sA0;
                const float* cB = (kt & 1) ? sB1 : sB0;
#pragma unroll
                for (int g = 0; g < BK / 8; ++g) {
                    f32x4 fa[TM], fb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[i] = TA::frag(cA, wm + 32 * i, g, r, h);
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[j] = TB::frag(cB, wn + 32 * j, g, r, h);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
                    if (g == 0) {
                        if (kt + 1 < nk) {
                            TA::stage((kt & 1) ? sA0 : sA1, ra[nslot], tid);
                            TB::stage((kt & 1) ? sB0 : sB1, rb[nslot], tid);
                        }
                        if (kt + 1 + FDEPTH < nk) {
                            const int k0 = k_begin + (kt + 1 + FDEPTH) * BK;
                            TA::fetch(ra[nslot], A, a.lda, m0, Mr, k0, k_end, a.vecA, tid);
                            TB::fetch(rb[nslot], B, a.ldb, n0, a.N, k0, k_end, a.vecB, tid);
                        }
                    }
                }
                __syncthreads();
            }
        }
    } else if constexpr (PREC == 0) {
        // Software pipeline: the registers ra/rb hold tile kt+1 (fetched one full
        // iteration earlier); they are written to the idle LDS buffer after the
        // first k group's MFMAs have been issued, and re-used at once for the
        // fetch of tile kt+2, so the staging traffic sits in the shadow of the
        // MFMA stream and a global load has a whole iteration to land.
        f32x4 ra[TA::NV], rb[TB::NV];
        if (nk > 0) {
            TA::fetch(ra, A, a.lda, m0, Mr, k_begin, k_end, a.vecA, tid);
            TB::fetch(rb, B, a.ldb, n0, a.N, k_begin, k_end, a.vecB, tid);
            TA::stage(sA0, ra, tid);
            TB::stage(sB0, rb, tid);
            if (nk > 1) {
                TA::fetch(ra, A, a.lda, m0, Mr, k_begin + KS, k_end, a.vecA, tid);
                TB::fetch(rb, B, a.ldb, n0, a.N, k_begin + KS, k_end, a.vecB, tid);
            }
        }
        __syncthreads();

        for (int kt = 0; kt < nk; ++kt) {
            const float* cA = (kt & 1) ? sA1 : sA0;
            const float* cB = (kt & 1) ? sB1 : sB0;
#pragma unroll
            for (int g = 0; g < BK / 8; ++g) {
                f32x4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = TA::frag(cA, wm + 32 * i, g, r, h);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = TB::frag(cB, wn + 32 * j, g, r, h);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
                if (g == 0) {
                    if (kt + 1 < nk) {
                        TA::stage((kt & 1) ? sA0 : sA1, ra, tid);
                        TB::stage((kt & 1) ? sB0 : sB1, rb, tid);
                    }
                    if (kt + 2 < nk) {
                        const int k0 = k_begin + (kt + 2) * BK;
                        TA::fetch(ra, A, a.lda, m0, Mr, k0, k_end, a.vecA, tid);
                        TB::fetch(rb, B, a.ldb, n0, a.N, k0, k_end, a.vecB, tid);
                    }
                }
            }
            __syncthreads();
        }
    } else {
        // The bf16 pipes finish a k step in a fraction of a global load's latency, so the register stage is a
        // ring of DEPTH tiles: tile t waits in slot t % DEPTH, is split into planes and written to LDS buffer
        // t & 1 during step t - 1, and its slot is refilled at once with tile t + DEPTH.  The loop is unrolled
        // by DEPTH so that the slots are fixed registers.  Interior tiles (whole rows, whole k steps, aligned)
        // run a copy of the loop without a single guard or branch in the step, so that the compiler can count
        // the outstanding loads instead of draining them and can place the split arithmetic between the MFMAs.
        constexpr int PL = PREC;
        // Ring depth: three tiles (deeper rings, 4 and 5, measured no faster); two for the fp16-split NT product on
        // 128x128 tiles, whose 32 registers go to the fragments of the k step's second half instead (FRAGS_AHEAD in
        // step(): requested up front, their LDS latency passes behind the first half's MFMAs and the staging
        // arithmetic): forward product of the graph encoder 98.7 -> 97.2 us, the text branch's logits product
        // 1.705 -> 1.644 ms, its LSTM input projection 0.209 -> 0.198 ms; the NN / TN forms measured no gain
        constexpr int DEPTH = (PL == 2 && BM == 128 && AKC && BKC) ? 2 : 3;
        f32x4 ra[DEPTH][TA::NV], rb[DEPTH][TB::NV];
        const bool fast = m0 + BM <= Mr && n0 + BN <= a.N && (k_end - k_begin) % KS == 0 && nk > 0 && a.vecA && a.vecB;
        auto step = [&](auto fast_tag, int kt, auto slot_tag) {
            constexpr bool FAST = decltype(fast_tag)::value;
            constexpr int slot = decltype(slot_tag)::value;
            const float* cA = (kt & 1) ? sA1 : sA0;
            const float* cB = (kt & 1) ? sB1 : sB0;
            constexpr bool FRAGS_AHEAD = (PL == 2 && KS == 32 && BM == 128 && AKC && BKC);
            bf16x8 fa2[FRAGS_AHEAD ? TM : 1][PL], fb2[FRAGS_AHEAD ? TN : 1][PL];
            if constexpr (FRAGS_AHEAD) {
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa2[i][pl] = TA::frag(cA, pl, wm + 32 * i, 1, r, h);
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb2[j][pl] = TB::frag(cB, pl, wn + 32 * j, 1, r, h);
                }
            }
#pragma unroll
            for (int g = 0; g < KS / 16; ++g) {
                bf16x8 fa[TM][PL], fb[TN][PL];
#pragma unroll
                for (int pl = 0; pl < PL; ++pl) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[i][pl] = (FRAGS_AHEAD && g == 1) ? fa2[FRAGS_AHEAD ? i : 0][pl] : TA::frag(cA, pl, wm + 32 * i, g, r, h);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        fb[j][pl] = (FRAGS_AHEAD && g == 1) ? fb2[FRAGS_AHEAD ? j : 0][pl] : TB::frag(cB, pl, wn + 32 * j, g, r, h);
                }
                // products of plane pairs (pa, pb) with pa + pb < PL, the small ones first
#pragma unroll
                for (int sum = PL - 1; sum >= 0; --sum)
#pragma unroll
                    for (int pa = 0; pa <= sum; ++pa)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = mfma_16bit<PREC == 2>(fa[i][pa], fb[j][sum - pa], acc[i][j]);
                if (g == 0) {
                    // tile kt + 1 waits in `slot`; staging is unconditional (past the last tile it rewrites
                    // a buffer nobody reads again)
                    TA::stage((kt & 1) ? sA0 : sA1, ra[slot], tid, sc_a);
                    TB::stage((kt & 1) ? sB0 : sB1, rb[slot], tid, sc_b);
                    if (FAST) {
                        if (PL == 2) {
#pragma unroll
                            for (int q = 0; q < 3 * TM * TN; ++q) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                            }
                        }
                        if (PL == 3) {
                            // one MFMA, then four of the split's vector instructions in its shadow
#pragma unroll
                            for (int q = 0; q < 6 * TM * TN; ++q) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                            }
                        }
                        // past the end the last tile is fetched again (never staged): no branch
                        const int k0 = k_begin + min(kt + 1 + DEPTH, nk - 1) * KS;
                        TA::fetch_fast(ra[slot], A, a.lda, m0, k0, tid);
                        TB::fetch_fast(rb[slot], B, a.ldb, n0, k0, tid);
                    } else if (kt + 1 + DEPTH < nk) {
                        const int k0 = k_begin + (kt + 1 + DEPTH) * KS;
                        TA::fetch(ra[slot], A, a.lda, m0, Mr, k0, k_end, a.vecA, tid);
                        TB::fetch(rb[slot], B, a.ldb, n0, a.N, k0, k_end, a.vecB, tid);
                    }
                }
            }
            __syncthreads();
        };
        auto run = [&](auto fast_tag) {
            constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
            for (int t = 0; t < DEPTH; ++t) {
                if (FAST) {
                    const int k0 = k_begin + min(t, nk - 1) * KS;
                    TA::fetch_fast(ra[t], A, a.lda, m0, k0, tid);
                    TB::fetch_fast(rb[t], B, a.ldb, n0, k0, tid);
                } else if (t < nk) {
                    TA::fetch(ra[t], A, a.lda, m0, Mr, k_begin + t * KS, k_end, a.vecA, tid);
                    TB::fetch(rb[t], B, a.ldb, n0, a.N, k_begin + t * KS, k_end, a.vecB, tid);
                }
            }
            if (nk > 0) {
                TA::stage(sA0, ra[0], tid, sc_a);
                TB::stage(sB0, rb[0], tid, sc_b);
                if (FAST) {
                    const int k0 = k_begin + min(DEPTH, nk - 1) * KS;
                    TA::fetch_fast(ra[0], A, a.lda, m0, k0, tid);
                    TB::fetch_fast(rb[0], B, a.ldb, n0, k0, tid);
                } else if (DEPTH < nk) {
                    TA::fetch(ra[0], A, a.lda, m0, Mr, k_begin + DEPTH * KS, k_end, a.vecA, tid);
                    TB::fetch(rb[0], B, a.ldb, n0, a.N, k_begin + DEPTH * KS, k_end, a.vecB, tid);
                }
            }
            __syncthreads();
            for (int kt0 = 0; kt0 < nk; kt0 += DEPTH) {
                if (kt0 < nk) step(fast_tag, kt0, std::integral_constant<int, 1 % DEPTH>());
                if (kt0 + 1 < nk) step(fast_tag, kt0 + 1, std::integral_constant<int, 2 % DEPTH>());
                if constexpr (DEPTH > 2)
                    if (kt0 + 2 < nk) step(fast_tag, kt0 + 2, std::integral_constant<int, 3 % DEPTH>());
                if constexpr (DEPTH > 3)
                    if (kt0 + 3 < nk) step(fast_tag, kt0 + 3, std::integral_constant<int, 4 % DEPTH>());
                if constexpr (DEPTH > 4)
                    if (kt0 + 4 < nk) step(fast_tag, kt0 + 4, std::integral_constant<int, 5 % DEPTH>());
                if constexpr (DEPTH > 5)
                    if (kt0 + 5 < nk) step(fast_tag, kt0 + 5, std::integral_constant<int, 6 % DEPTH>());
            }
        };
        static_assert(DEPTH >= 2 && DEPTH <= 6, "the step calls above are written out for rings of two to six");
        if (fast)
            run(std::true_type());
        else
            run(std::false_type());
    }

    if constexpr (PREC == 2) {
        // undo the operand scales (two multiplications by powers of two: exact, and neither factor can leave
        // the normal range the way their product could)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = (acc[i][j][e] * inv_a) * inv_b;
    }

    // ---- epilogue.  Accumulator register e of lane (r,h) is element
    // (row = (e&3) + 8*(e>>2) + 4h, col = r) of its 32x32 tile.
    if (a.epi == EPI_POOL) {
        if (WAVES_M == 1 && KG == 1) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn + 32 * j + r;
                const float bias = (a.bias != nullptr && col < a.N) ? a.bias[col] : 0.f;
                float best = 0.f;
                int bi = -1;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                        const float v = acc[i][j][e] + bias;
                        if (row < Mr && (bi < 0 || v > best)) {
                            best = v;
                            bi = row;
                        }
                    }
                const float ob = __shfl_xor(best, 32);
                const int oi = __shfl_xor(bi, 32);
                if (oi >= 0 && (bi < 0 || ob > best || (ob == best && oi < bi))) {
                    best = ob;
                    bi = oi;
                }
                if (a.proto_base != nullptr) {
                    // rows past this graph's real nodes all equal batch-independent prototype rows, stored once
                    // after the real rows of the B operand: take their maximum from row n_b on (lowest index on
                    // ties; the two half-waves share a column and split the rows) and merge it -- a real row
                    // wins ties (lower index).  The bias is added as to the real rows.
                    const int nb = a.batch_dim[batch];
                    float pv = 0.f;
                    int pi = -1;
                    if (col < a.N) {
                        const float* Q = a.B + (long long)a.proto_base[0] * a.ldb + col;
#pragma unroll 4
                        for (int i = nb + h; i < a.M; i += 2) {
                            const float v = Q[(long long)i * a.ldb] + bias;
                            if (pi < 0 || v > pv) {
                                pv = v;
                                pi = i;
                            }
                        }
                    }
                    const float qv = __shfl_xor(pv, 32);
                    const int qi = __shfl_xor(pi, 32);
                    if (qi >= 0 && (pi < 0 || qv > pv || (qv == pv && qi < pi))) {
                        pv = qv;
                        pi = qi;
                    }
                    if (pi >= 0 && (bi < 0 || pv > best)) {
                        best = pv;
                        bi = pi;
                    }
                }
                if (h == 0 && col < a.N) {
                    a.pool_val[(long long)batch * a.N + col] = best;
                    a.pool_idx[(long long)batch * a.N + col] = bi;
                }
            }
        }
        return;
    }

    if (KG == 1 && parts > 1) {
        // Leave this part's accumulators in its slab; the last part to arrive sums all of them, its own
        // included, in part order, so the result does not depend on which block that is.
        // The hand-off between the blocks (they may run on different XCDs, whose L2s are not coherent, and a
        // CU's vector L1 is never refreshed by another CU's stores):
        //   writer: every slab byte leaves by a 16-byte WRITE-THROUGH store (buffer_store_dwordx4 sc1: it goes
        //           to the point of coherence between the XCDs and leaves no dirty L2 line behind, so no cache
        //           write-back is needed); every storing wave then waits for its stores (vmcnt(0)), the block
        //           meets at a barrier, and only then one lane counts the block in (device-scope atomic add);
        //   reader: the block whose add returned parts - 1 knows every other part's stores have completed; one
        //           lane runs a device-scope acquire (drops this CU's L1 lines), waits for it, the block meets
        //           at a barrier, and the slabs are read by loads that bypass the L1 as well (sc1).
        // No release fence on the writer: this is the fence-free hand-off of the CDNA HIP programming guide,
        // section 6, Guideline 16 ("inter-workgroup communication"), rule R1, and of the MI355X micro-architecture
        // notes' visibility table (rows "stores of each flavour": sc1 stores need no buffer_wbl2 release; "valid
        // forms": sc1 payload + vmcnt(0) in every storing wave + an atomic-raised counter + sc1 loads).  It rests on
        // the cache policy of sc1 accesses on gfx950 as measured there, not on the C++ memory model.
        // 16-byte stores: register e of a 32x32 accumulator holds row (e&3) + 8(e>>2) + 4h, so registers
        // 4q..4q+3 go out as one store; every store instruction of a wave is one contiguous 1 KB run.
        int& s_last = *reinterpret_cast<int*>(smem + gemm_smem_floats<BM, BN, WAVES_M, WAVES_N, KG, AKC, BKC, PREC>() - 4);
        const long long slab_off = ((long long)tail_tile * parts) * (BM * BN);
        const auto slabs = __builtin_amdgcn_make_buffer_rsrc(
            a.tail_ws + (long long)__builtin_amdgcn_readfirstlane((int)(slab_off >> 8)) * 256, 0,
            __builtin_amdgcn_readfirstlane(parts) * (BM * BN * 4), 0x00020000);
        const int mine = part * (BM * BN * 4) + tid * 16;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), slabs,
                                                           mine + ((i * TN + j) * 4 + q) * (GT * 16), 0, 16 /* sc1 */);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave, before the block counts itself in
        __syncthreads();
        if (tid == 0) {
            const int seen = __hip_atomic_fetch_add(&a.tail_counters[tail_tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (seen == parts - 1) ? 1 : 0;
            if (seen == parts - 1) {
                // ready for the next launch
                __hip_atomic_store(&a.tail_counters[tail_tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the invalidate has completed before the barrier opens
            }
        }
        __syncthreads();
        if (!s_last) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int p = 0; p < parts; ++p) {
            const int src = p * (BM * BN * 4) + tid * 16;
            f32x4 t[TM][TN][4];  // all loads of a part in flight before the first add
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        t[i][j][q] = __builtin_bit_cast(
                            f32x4, __builtin_amdgcn_raw_buffer_load_b128(slabs, src + ((i * TN + j) * 4 + q) * (GT * 16), 0,
                                                                         16 /* sc1 */));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += t[i][j][q][e];
        }
    }

    float* __restrict__ C;
    int ldc;
    if (a.to_slabs) {
        C = a.C + ((long long)batch * a.splits + split) * (long long)a.M * a.N;
        ldc = a.N;
    } else {
        C = a.C + batch * a.strideC + (a.off_c ? row_off * a.ldc : 0);
        ldc = a.ldc;
    }
    const float* __restrict__ aux = a.aux ? a.aux + batch * a.strideAux + (a.off_c ? row_off * a.ldaux : 0) : nullptr;
    const int epi = a.to_slabs ? (int)EPI_STORE : a.epi;
    const float* __restrict__ bias = (a.to_slabs || a.bias == nullptr) ? nullptr : a.bias + batch * a.strideBias;

    constexpr int LPR = WTN / 4;   // lanes per row
    constexpr int RPI = 64 / LPR;  // rows per wave instruction
    constexpr int NIT = WTM / RPI;
    const int c4 = lane % LPR, rsub = lane / LPR;
    const int gcol = n0 + wn + c4 * 4;
    const bool vec_ok = a.vecC && gcol + 3 < a.N;
    f32x4 colacc = {0.f, 0.f, 0.f, 0.f};  // this lane's share of the column sums of the stored values
    constexpr bool HALVES = gemm_epilogue_halves<KG, PREC>() && TM == 2;
    if constexpr (!HALVES) {
    // Stage the wave's WTM x WTN accumulator tile through LDS (the operand
    // buffers are free after the last barrier of the main loop) so that global
    // stores -- and the aux / C loads of the mask and accumulate epilogues --
    // are whole 16-byte row segments, 4 or 8 full rows per wave instruction,
    // instead of one dword per lane per accumulator register.
    float* sC = smem + (kg * GW + wave) * (WTM * LDC);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                sC[(32 * i + (e & 3) + 8 * (e >> 2) + 4 * h) * LDC + 32 * j + r] = acc[i][j][e];
    __syncthreads();

    // K-group configurations: the groups' waves share the store loop (iteration `it` belongs to group it % KG),
    // each summing every group's staged tile for its rows, in group order
    constexpr int NOWN = (KG > 1) ? (NIT + KG - 1) / KG : NIT;
    const float* const sR = smem + wave * (WTM * LDC);  // group 0's staged tile at this wave's position
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (gcol + q < a.N) bv[q] = bias[gcol + q];
    }
    f32x4 xv[NOWN];
    if (epi == EPI_MASK || epi == EPI_ACCUM) {
        const float* __restrict__ src = (epi == EPI_MASK) ? aux : C;
        const int lds = (epi == EPI_MASK) ? a.ldaux : ldc;
        const bool vsrc = (epi == EPI_MASK) ? (a.vecAux && gcol + 3 < a.N) : vec_ok;
#pragma unroll
        for (int u = 0; u < NOWN; ++u) {
            const int it = KG > 1 ? u * KG + kg : u;
            const int grow = m0 + wm + it * RPI + rsub;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (it < NIT && grow < Mr && gcol < a.N) {
                const float* p = src + (long long)grow * lds + gcol;
                if (vsrc) {
                    t = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (gcol + q < a.N) t[q] = p[q];
                }
            }
            xv[u] = t;
        }
    }
    f32x4 ib = {1.f, 1.f, 1.f, 1.f};
    if constexpr (PREC == 5) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (gcol + q < a.N) ib[q] = a.inv_b_rows[gcol + q];
    }
    // PREC 5: the inverse scales of this lane's rows, all fetched before the store loop (inside it every row
    // would wait for its own load)
    float ia[NOWN];
    if constexpr (PREC == 5) {
#pragma unroll
        for (int u = 0; u < NOWN; ++u) ia[u] = a.inv_a_rows[max(min(m0 + wm + u * RPI + rsub, Mr - 1), 0)];  // KG == 1
    }
    // EPI_SIGMOID_BCE: where in a graph's (n, n) target the lane's four triangle elements lie (bce_fwd_kernel's
    // inversion of k = i n - i(i-1)/2 + (j - i)), and the targets of all its rows, fetched before the store loop
    // (both fused epilogues exist in the K-group configurations only: their batch-sized products are the ones
    // that use them, and the large tiles' register budget has no room for the prefetched operands)
    constexpr bool FUSE = (KG > 1 && PREC == 0);
    constexpr int FN = FUSE ? NOWN : 1;
    int tij[4] = {0, 0, 0, 0};
    f32x4 tv[FN];
    float bce_sum = 0.f, bce_g = 0.f;
    if (FUSE && epi == EPI_SIGMOID_BCE) {
        const int n = a.fuse.bce_n;
        const float tn1 = (float)(2 * n + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = min(gcol + q, a.N - 1);
            int i = (int)((tn1 - sqrtf(tn1 * tn1 - 8.f * (float)k)) * 0.5f);
            i = max(0, min(i, n - 1));
            if (i * n - (i * (i - 1)) / 2 > k) --i;
            else if (i + 1 < n && (i + 1) * n - ((i + 1) * i) / 2 <= k) ++i;
            tij[q] = i * n + i + (k - (i * n - (i * (i - 1)) / 2));
        }
#pragma unroll
        for (int u = 0; u < FN; ++u) {
            const int grow = max(min(m0 + wm + (u * KG + kg) * RPI + rsub, Mr - 1), 0);
            const float* trow = a.fuse.bce_target + (long long)grow * (n * n);
#pragma unroll
            for (int q = 0; q < 4; ++q) tv[u][q] = trow[tij[q]];
        }
        bce_g = a.fuse.bce_gscale;
        if (a.fuse.bce_gscale_dev != nullptr) bce_g *= a.fuse.bce_gscale_dev[0];
    }
    // EPI_LATENT_BWD: mu, logvar and eps of the lane's elements, fetched before the store loop
    f32x4 lmu[FN], llv[FN], lep[FN];
    float lat_w = 0.f;
    if (FUSE && epi == EPI_LATENT_BWD) {
#pragma unroll
        for (int u = 0; u < FN; ++u) {
            const long long o = (long long)max(min(m0 + wm + (u * KG + kg) * RPI + rsub, Mr - 1), 0) * a.N +
                                max(min(gcol, a.N - 4), 0);
            lmu[u] = *reinterpret_cast<const f32x4*>(a.fuse.mu + o);
            llv[u] = *reinterpret_cast<const f32x4*>(a.fuse.lv + o);
            lep[u] = *reinterpret_cast<const f32x4*>(a.fuse.eps + o);
        }
        lat_w = (a.fuse.kl_w_dev ? a.fuse.kl_w_dev[0] * a.fuse.kl_w : a.fuse.kl_w) * a.fuse.inv_b;
    }
    // The store loop is instantiated once per epilogue and chosen by ONE switch: with the switch inside the
    // loop the compiler keeps it there, a chain of scalar compares and branches per element.
    auto store_rows = [&](auto epi_tag) {
        constexpr int E = decltype(epi_tag)::value;
#pragma unroll
        for (int u = 0; u < NOWN; ++u) {
            const int it = KG > 1 ? u * KG + kg : u;
            if (NIT % (KG > 1 ? KG : 1) != 0 && it >= NIT) continue;
            const int lrow = it * RPI + rsub;
            const int grow = m0 + wm + lrow;
            f32x4 v = *reinterpret_cast<const f32x4*>(&sR[lrow * LDC + c4 * 4]);
#pragma unroll
            for (int g = 1; g < KG; ++g) v += *reinterpret_cast<const f32x4*>(&sR[g * GW * WTM * LDC + lrow * LDC + c4 * 4]);
            if (grow >= Mr || gcol >= a.N) continue;
            if constexpr (PREC == 5) {
                // undo the operands' row scales (powers of two: exact)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (v[q] * ia[u]) * ib[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float y = v[q] + bv[q];
                if constexpr (E == EPI_RELU) y = fmaxf(y, 0.f);
                if constexpr (E == EPI_SIGMOID || (FUSE && E == EPI_SIGMOID_BCE)) y = 1.f / (1.f + expf(-y));
                if constexpr (E == EPI_MASK) y = (xv[u][q] > 0.f) ? y : 0.f;
                if constexpr (E == EPI_ACCUM) y += xv[u][q];
                v[q] = y;
            }
            colacc += v;
            float* dst = C + (long long)grow * ldc + gcol;
            if (vec_ok) {
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (gcol + q < a.N) dst[q] = v[q];
            }
            if constexpr (FUSE && E == EPI_SIGMOID_BCE) {
                f32x4 dl;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float pq = v[q], tq = tv[u][q];
                    const float lp = fmaxf(logf(pq), -100.f);
                    const float lq = fmaxf(logf(1.f - pq), -100.f);
                    if (gcol + q < a.N) bce_sum += (tq - 1.f) * lq - tq * lp;
                    const float qq = (1.f - pq) * pq;
                    dl[q] = (bce_g * (pq - tq) / fmaxf(qq, 1e-12f)) * qq;
                }
                float* gd = a.fuse.bce_dlogits + (long long)grow * a.N + gcol;
                if (vec_ok) {
                    *reinterpret_cast<f32x4*>(gd) = dl;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (gcol + q < a.N) gd[q] = dl[q];
                }
            }
            if constexpr (FUSE && E == EPI_LATENT_BWD) {
                // (N is a multiple of 4 and the rows 16-byte aligned in this mode: gemm_plan checks)
                f32x4 gm, gl;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float pm, pl;
                    latent_bwd_point(v[q], lmu[u][q], llv[u][q], lep[u][q], lat_w, pm, pl);
                    gm[q] = pm;
                    gl[q] = pl;
                }
                const long long o = (long long)grow * a.fuse.ld_out + gcol;
                *reinterpret_cast<f32x4*>(a.fuse.dmu + o) = gm;
                *reinterpret_cast<f32x4*>(a.fuse.dlv + o) = gl;
            }
        }
    };
    switch (epi) {
        case EPI_RELU: store_rows(std::integral_constant<int, EPI_RELU>{}); break;
        case EPI_SIGMOID: store_rows(std::integral_constant<int, EPI_SIGMOID>{}); break;
        case EPI_MASK: store_rows(std::integral_constant<int, EPI_MASK>{}); break;
        case EPI_ACCUM: store_rows(std::integral_constant<int, EPI_ACCUM>{}); break;
        case EPI_SIGMOID_BCE:
            if constexpr (FUSE) store_rows(std::integral_constant<int, EPI_SIGMOID_BCE>{});
            break;
        case EPI_LATENT_BWD:
            if constexpr (FUSE) store_rows(std::integral_constant<int, EPI_LATENT_BWD>{});
            break;
        default: store_rows(std::integral_constant<int, EPI_STORE>{}); break;
    }
    if (FUSE && epi == EPI_SIGMOID_BCE) {
        // the wave's loss terms, lanes added in a fixed order; one slot per wave of every tile
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bce_sum += __shfl_xor(bce_sum, o);
        if (lane == 0) a.fuse.bce_partial[(((long long)tm * a.tiles_n + tn) * GW + wave) * KG + kg] = bce_sum;
    }
    } else {
    // Stage the wave's WTM x WTN accumulator tile through LDS (the operand
    // buffers are free after the last barrier of the main loop) so that global
    // stores -- and the aux / C loads of the mask and accumulate epilogues --
    // are whole 16-byte row segments, 4 or 8 full rows per wave instruction,
    // instead of one dword per lane per accumulator register.
    // The fp16-plane kernel stages and stores its tile in two halves of 32 rows (HALVES): a wave's staging area
    // is its own (LDS executes a wave's accesses in order, so the second half is written over the first with no
    // barrier), and half the staging area is what lets a third block share a CU's LDS.
    constexpr int NH = HALVES ? 2 : 1;
    constexpr int HROWS = WTM / NH;  // rows staged at a time
    constexpr int NIT_H = NIT / NH;  // store-loop iterations per staged part
    // K-group configurations: the groups' waves share the store loop (iteration `it` belongs to group it % KG),
    // each summing every group's staged tile for its rows, in group order
    constexpr int NOWN = (KG > 1) ? (NIT + KG - 1) / KG : NIT_H;
    static_assert(!HALVES || (KG == 1 && NIT % 2 == 0), "staging in halves is written for one wave group");
    float* const sC = smem + (kg * GW + wave) * (HROWS * LDC);
    const float* const sR = smem + wave * (HROWS * LDC);  // group 0's staged tile at this wave's position
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    f32x4 ib = {1.f, 1.f, 1.f, 1.f};
    // EPI_SIGMOID_BCE: where in a graph's (n, n) target the lane's four triangle elements lie (bce_fwd_kernel's
    // inversion of k = i n - i(i-1)/2 + (j - i)), and the targets of all its rows, fetched before the store loop
    // (both fused epilogues exist in the K-group configurations only: their batch-sized products are the ones
    // that use them, and the large tiles' register budget has no room for the prefetched operands)
    constexpr bool FUSE = (KG > 1 && PREC == 0);
    constexpr int FN = FUSE ? NOWN : 1;
    int tij[4] = {0, 0, 0, 0};
    f32x4 tv[FN];
    float bce_sum = 0.f, bce_g = 0.f;
    // EPI_LATENT_BWD: mu, logvar and eps of the lane's elements, fetched before the store loop
    f32x4 lmu[FN], llv[FN], lep[FN];
    float lat_w = 0.f;
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
        // this part's accumulators into the staging area: all TM row tiles at once, or row tile hf
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (HALVES && i != hf) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    sC[((HALVES ? 0 : 32 * i) + (e & 3) + 8 * (e >> 2) + 4 * h) * LDC + 32 * j + r] = acc[i][j][e];
        }
        if constexpr (!HALVES) __syncthreads();  // (halves: the wave reads back only what it staged itself)
        const int it0 = hf * NIT_H;  // first store-loop iteration of this part
        if (hf == 0) {
            // what every part of the store loop needs from global memory, requested once (after the first staging:
            // placed in front of it, the same loads made the in-loop-split kernels' epilogue slower)
            if (bias != nullptr) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (gcol + q < a.N) bv[q] = bias[gcol + q];
            }
            if constexpr (PREC == 5) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (gcol + q < a.N) ib[q] = a.inv_b_rows[gcol + q];
            }
            if (FUSE && epi == EPI_SIGMOID_BCE) {
                const int n = a.fuse.bce_n;
                const float tn1 = (float)(2 * n + 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = min(gcol + q, a.N - 1);
                    int i = (int)((tn1 - sqrtf(tn1 * tn1 - 8.f * (float)k)) * 0.5f);
                    i = max(0, min(i, n - 1));
                    if (i * n - (i * (i - 1)) / 2 > k) --i;
                    else if (i + 1 < n && (i + 1) * n - ((i + 1) * i) / 2 <= k) ++i;
                    tij[q] = i * n + i + (k - (i * n - (i * (i - 1)) / 2));
                }
#pragma unroll
                for (int u = 0; u < FN; ++u) {
                    const int grow = max(min(m0 + wm + (u * KG + kg) * RPI + rsub, Mr - 1), 0);
                    const float* trow = a.fuse.bce_target + (long long)grow * (n * n);
#pragma unroll
                    for (int q = 0; q < 4; ++q) tv[u][q] = trow[tij[q]];
                }
                bce_g = a.fuse.bce_gscale;
                if (a.fuse.bce_gscale_dev != nullptr) bce_g *= a.fuse.bce_gscale_dev[0];
            }
            if (FUSE && epi == EPI_LATENT_BWD) {
#pragma unroll
                for (int u = 0; u < FN; ++u) {
                    const long long o = (long long)max(min(m0 + wm + (u * KG + kg) * RPI + rsub, Mr - 1), 0) * a.N +
                                        max(min(gcol, a.N - 4), 0);
                    lmu[u] = *reinterpret_cast<const f32x4*>(a.fuse.mu + o);
                    llv[u] = *reinterpret_cast<const f32x4*>(a.fuse.lv + o);
                    lep[u] = *reinterpret_cast<const f32x4*>(a.fuse.eps + o);
                }
                lat_w = (a.fuse.kl_w_dev ? a.fuse.kl_w_dev[0] * a.fuse.kl_w : a.fuse.kl_w) * a.fuse.inv_b;
            }
        }
        f32x4 xv[NOWN];
        if (epi == EPI_MASK || epi == EPI_ACCUM) {
            const float* __restrict__ src = (epi == EPI_MASK) ? aux : C;
            const int lds = (epi == EPI_MASK) ? a.ldaux : ldc;
            const bool vsrc = (epi == EPI_MASK) ? (a.vecAux && gcol + 3 < a.N) : vec_ok;
#pragma unroll
            for (int u = 0; u < NOWN; ++u) {
                const int it = KG > 1 ? u * KG + kg : it0 + u;
                const int grow = m0 + wm + it * RPI + rsub;
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (it < NIT && grow < Mr && gcol < a.N) {
                    const float* p = src + (long long)grow * lds + gcol;
                    if (vsrc) {
                        t = *reinterpret_cast<const f32x4*>(p);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (gcol + q < a.N) t[q] = p[q];
                    }
                }
                xv[u] = t;
            }
        }
        // PREC 5: the inverse scales of this lane's rows, all fetched before the store loop (inside it every row
        // would wait for its own load)
        float ia[NOWN];
        if constexpr (PREC == 5) {
#pragma unroll
            for (int u = 0; u < NOWN; ++u)
                ia[u] = a.inv_a_rows[max(min(m0 + wm + (it0 + u) * RPI + rsub, Mr - 1), 0)];  // KG == 1
        }
        // The store loop is instantiated once per epilogue and chosen by ONE switch: with the switch inside the
        // loop the compiler keeps it there, a chain of scalar compares and branches per element.
        auto store_rows = [&](auto epi_tag) {
            constexpr int E = decltype(epi_tag)::value;
#pragma unroll
            for (int u = 0; u < NOWN; ++u) {
                const int it = KG > 1 ? u * KG + kg : it0 + u;
                if (NIT % (KG > 1 ? KG : 1) != 0 && it >= NIT) continue;
                const int lrow = it * RPI + rsub;       // row of the wave's tile
                const int srow = lrow - hf * HROWS * (HALVES ? 1 : 0);  // row of the staging area
                const int grow = m0 + wm + lrow;
                f32x4 v = *reinterpret_cast<const f32x4*>(&sR[srow * LDC + c4 * 4]);
#pragma unroll
                for (int g = 1; g < KG; ++g) v += *reinterpret_cast<const f32x4*>(&sR[g * GW * WTM * LDC + srow * LDC + c4 * 4]);
                if (grow >= Mr || gcol >= a.N) continue;
                if constexpr (PREC == 5) {
                    // undo the operands' row scales (powers of two: exact)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (v[q] * ia[u]) * ib[q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float y = v[q] + bv[q];
                    if constexpr (E == EPI_RELU) y = fmaxf(y, 0.f);
                    if constexpr (E == EPI_SIGMOID || (FUSE && E == EPI_SIGMOID_BCE)) y = 1.f / (1.f + expf(-y));
                    if constexpr (E == EPI_MASK) y = (xv[u][q] > 0.f) ? y : 0.f;
                    if constexpr (E == EPI_ACCUM) y += xv[u][q];
                    v[q] = y;
                }
                colacc += v;
                float* dst = C + (long long)grow * ldc + gcol;
                if (vec_ok) {
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (gcol + q < a.N) dst[q] = v[q];
                }
                if constexpr (FUSE && E == EPI_SIGMOID_BCE) {
                    f32x4 dl;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float pq = v[q], tq = tv[u][q];
                        const float lp = fmaxf(logf(pq), -100.f);
                        const float lq = fmaxf(logf(1.f - pq), -100.f);
                        if (gcol + q < a.N) bce_sum += (tq - 1.f) * lq - tq * lp;
                        const float qq = (1.f - pq) * pq;
                        dl[q] = (bce_g * (pq - tq) / fmaxf(qq, 1e-12f)) * qq;
                    }
                    float* gd = a.fuse.bce_dlogits + (long long)grow * a.N + gcol;
                    if (vec_ok) {
                        *reinterpret_cast<f32x4*>(gd) = dl;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (gcol + q < a.N) gd[q] = dl[q];
                    }
                }
                if constexpr (FUSE && E == EPI_LATENT_BWD) {
                    // (N is a multiple of 4 and the rows 16-byte aligned in this mode: gemm_plan checks)
                    f32x4 gm, gl;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float pm, pl;
                        latent_bwd_point(v[q], lmu[u][q], llv[u][q], lep[u][q], lat_w, pm, pl);
                        gm[q] = pm;
                        gl[q] = pl;
                    }
                    const long long o = (long long)grow * a.fuse.ld_out + gcol;
                    *reinterpret_cast<f32x4*>(a.fuse.dmu + o) = gm;
                    *reinterpret_cast<f32x4*>(a.fuse.dlv + o) = gl;
                }
            }
        };
        switch (epi) {
            case EPI_RELU: store_rows(std::integral_constant<int, EPI_RELU>{}); break;
            case EPI_SIGMOID: store_rows(std::integral_constant<int, EPI_SIGMOID>{}); break;
            case EPI_MASK: store_rows(std::integral_constant<int, EPI_MASK>{}); break;
            case EPI_ACCUM: store_rows(std::integral_constant<int, EPI_ACCUM>{}); break;
            case EPI_SIGMOID_BCE:
                if constexpr (FUSE) store_rows(std::integral_constant<int, EPI_SIGMOID_BCE>{});
                break;
            case EPI_LATENT_BWD:
                if constexpr (FUSE) store_rows(std::integral_constant<int, EPI_LATENT_BWD>{});
                break;
            default: store_rows(std::integral_constant<int, EPI_STORE>{}); break;
        }
    }
    if (FUSE && epi == EPI_SIGMOID_BCE) {
        // the wave's loss terms, lanes added in a fixed order; one slot per wave of every tile
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bce_sum += __shfl_xor(bce_sum, o);
        if (lane == 0) a.fuse.bce_partial[(((long long)tm * a.tiles_n + tn) * GW + wave) * KG + kg] = bce_sum;
    }
    }
    // Column sums of what this wave stored (its WTM rows x WTN columns), rows in a fixed order: a bias
    // gradient then needs no pass of its own over C, only the sum of these per-row-block partials.
    if (a.colsum_part != nullptr && !a.to_slabs) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
            for (int q = 0; q < 4; ++q) colacc[q] += __shfl_xor(colacc[q], o);
        if (rsub == 0 && gcol < a.N) {
            float* dst = a.colsum_part + (long long)(tm * WAVES_M + wm / WTM) * a.N + gcol;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (gcol + q < a.N) dst[q] = colacc[q];
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int KG, bool AKC, bool BKC, int PREC>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * KG, (gemm_min_waves<WAVES_M, WAVES_N, KG, PREC>())) void gemm_kernel(const KArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[gemm_smem_floats<BM, BN, WAVES_M, WAVES_N, KG, AKC, BKC, PREC>()];
    gemm_body<BM, BN, WAVES_M, WAVES_N, KG, AKC, BKC, PREC>(a, smem, (int)blockIdx.x, (int)gridDim.x);
}

// Two independent batch-sized GEMMs in one launch: blocks [0, n0) run job 0, the rest job 1.  Such GEMMs
// are latency-bound (a block's few k steps, prologue and epilogue; their duration hardly moves when the
// tile count doubles), so two that do not depend on each other cost little more than one, and a kernel
// boundary is saved besides.  Each job is computed exactly as by a launch of its own.
template <int BM_, int BN_, int WM_, int WN_, int KG_, bool AKC_, bool BKC_>
struct GemmJob {
    static constexpr int THREADS = 64 * WM_ * WN_ * KG_;
    static constexpr int SMEM = gemm_smem_floats<BM_, BN_, WM_, WN_, KG_, AKC_, BKC_, 0>();
    static constexpr int MIN_WAVES = gemm_min_waves<WM_, WN_, KG_, 0>();
    static __device__ __forceinline__ void run(const KArgs& a, float* smem, int block_x, int grid_x)
    {
        gemm_body<BM_, BN_, WM_, WN_, KG_, AKC_, BKC_, 0>(a, smem, block_x, grid_x);
    }
};

// The first sj.blocks() blocks of the launch run side jobs (side_jobs.h): short reductions that do not depend on
// the products, placed in front so that they are dispatched first and are gone by the time the products' blocks
// need their slots.
template <class J0, class J1>
__global__ __launch_bounds__(J0::THREADS, J0::MIN_WAVES) void gemm_pair_kernel(const KArgs a0, const KArgs a1, const int n0,
                                                                               const int lead1, const SideJobs sj)
{
    static_assert(J0::THREADS == J1::THREADS, "the jobs of a pair share the block size");
    constexpr int SMEM = J0::SMEM > J1::SMEM ? J0::SMEM : J1::SMEM;
    static_assert(SMEM >= SIDE_SMEM_FLOATS, "the side jobs stage through the products' LDS");
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    const int ns = sj.first_block[SIDE_MAX_JOBS];
    const int bx = (int)blockIdx.x - ns;
    const int n1 = (int)gridDim.x - ns - n0;
    // the job whose blocks run longer (the deeper K) is dispatched first, so that its blocks do not form the
    // launch's tail; which block computes a tile changes nothing in the tile
    if (bx < 0)
        side_run<J0::THREADS>(sj, (int)blockIdx.x, smem);
    else if (lead1 ? bx >= n1 : bx < n0)
        J0::run(a0, smem, lead1 ? bx - n1 : bx, n0);
    else
        J1::run(a1, smem, lead1 ? bx : bx - n0, n1);
}

// One batch-sized product (a K-group configuration, fp32 pipe) with side jobs in front of its blocks.
template <int BM, int BN, int WAVES_M, int WAVES_N, int KG, bool AKC, bool BKC>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * KG, (gemm_min_waves<WAVES_M, WAVES_N, KG, 0>())) void gemm_side_kernel(const KArgs a, const SideJobs sj)
{
    constexpr int SMEM = gemm_smem_floats<BM, BN, WAVES_M, WAVES_N, KG, AKC, BKC, 0>();
    static_assert(SMEM >= SIDE_SMEM_FLOATS, "the side jobs stage through the product's LDS");
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    const int ns = sj.first_block[SIDE_MAX_JOBS];
    const int bx = (int)blockIdx.x - ns;
    if (bx < 0)
        side_run<64 * WAVES_M * WAVES_N * KG>(sj, (int)blockIdx.x, smem);
    else
        gemm_body<BM, BN, WAVES_M, WAVES_N, KG, AKC, BKC, 0>(a, smem, bx, (int)gridDim.x - ns);
}

// C (+)= sum over slabs, in slab order.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, long long mn,
                                                            int N, float* __restrict__ C, int ldc, int accumulate)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= mn) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += slabs[(long long)k * mn + i];
    const long long row = i / N, col = i % N;
    float* dst = C + row * ldc + col;
    *dst = accumulate ? (*dst + s) : s;
}

// The exact three-way bf16 split of TileB::stage, done once for a whole matrix: x = h0 + h1 + h2, each h the
// top 16 bits of what is left (truncation), 8 significant bits apiece, written in the k-blocked operand layout
// of the plane GEMM: the operand has R rows and depth K (K a multiple of 16), planes[pl][k / 16][row][k % 16].
// The operand is src itself (R = rows, K = cols: an activation matrix or a weight used as W^T), or with
// `transpose` its transpose (R = cols, K = rows: a weight matrix stored (K, N) used as the B operand of X * W).
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, long long rows, long long cols,
                                                           long long ld, int transpose,
                                                           unsigned short* __restrict__ planes)
{
    const long long total = rows * cols;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // output element
    if (i >= total) return;
    const long long R = transpose ? cols : rows;
    const long long kb = i / (R * 16), row = (i / 16) % R, k = kb * 16 + (i & 15);
    const long long r = transpose ? k : row, c = transpose ? row : k;
    const float x0 = src[r * ld + c];
    const unsigned b0 = __float_as_uint(x0) & 0xffff0000u;
    const float x1 = x0 - __uint_as_float(b0);
    const unsigned b1 = __float_as_uint(x1) & 0xffff0000u;
    const float x2 = x1 - __uint_as_float(b1);
    planes[i] = (unsigned short)(b0 >> 16);
    planes[total + i] = (unsigned short)(b1 >> 16);
    planes[2 * total + i] = (unsigned short)(__float_as_uint(x2) >> 16);
}

// Partial maxima of |A| and |B| for the two-plane fp16 path: blocks [0, ABSMAX_PARTS) walk operand A, the rest
// operand B; every block writes its maximum (0 when it has no rows), so the output needs no clearing and the
// result does not depend on the order the blocks run in.  An operand is `rows` x `cols` with leading dimension
// `ld`, either extent optionally capped by a count in device memory (what lies past it is never read).
struct AbsmaxJob {
    const float* p;
    long long ld;
    int rows, cols;
    const int* rows_dev;
    const int* cols_dev;
    float bound;  // > 0: the caller's bound on the operand's magnitudes is taken instead of a pass over it;
                  // < 0: the operand's slots already hold valid maxima and are left alone
};

__global__ __launch_bounds__(256) void absmax_pair_kernel(const AbsmaxJob ja, const AbsmaxJob jb, float* __restrict__ out)
{
    __shared__ float sm[4];
    const bool second = (int)blockIdx.x >= ABSMAX_PARTS;
    const AbsmaxJob j = second ? jb : ja;
    const int part = (int)blockIdx.x - (second ? ABSMAX_PARTS : 0), tid = threadIdx.x;
    if (j.bound != 0.f) {
        if (tid == 0 && j.bound > 0.f) out[blockIdx.x] = j.bound;
        return;
    }
    int rows = j.rows, cols = j.cols;
    if (j.rows_dev) rows = min(rows, j.rows_dev[0]);
    if (j.cols_dev) cols = min(cols, j.cols_dev[0]);
    const bool vec = (j.ld % 4 == 0) && (reinterpret_cast<uintptr_t>(j.p) & 15) == 0;
    // threads per row: enough for 16 bytes each, a power of two up to the block; the block takes 256 / tpr rows
    // a pass, four passes' loads in flight together
    int tpr = 1;
    while (tpr < 256 && tpr * 4 < cols) tpr <<= 1;
    const int rpb = 256 / tpr, tr = tid / tpr, tc = tid % tpr;
    const int c4 = vec ? (cols & ~3) : 0;
    const long long pass = (long long)ABSMAX_PARTS * rpb;
    float m = 0.f;
    for (int c = tc * 4; c < c4; c += tpr * 4) {
        long long row = (long long)part * rpb + tr;
        for (; row + 3 * pass < rows; row += 4 * pass) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(j.p + (row + u * pass) * j.ld + c);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
        }
        for (; row < rows; row += pass) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(j.p + row * j.ld + c);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
    for (long long row = (long long)part * rpb + tr; row < rows; row += pass)
        for (int c = c4 + tc; c < cols; c += tpr) m = fmaxf(m, fabsf(j.p[row * j.ld + c]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) sm[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

enum TileCfg { CFG_128, CFG_64x128, CFG_64, CFG_64_K2, CFG_32_K4, CFG_256x128, CFG_COUNT };

inline TileCfg pick_cfg(int M, int N, int K, int batch, int epi)
{
    if (epi == EPI_POOL) return CFG_64x128;
    if (batch > 1 && M <= 64) return CFG_64x128;
    const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * batch;
    if (t128 >= 192) return CFG_128;
    // short-and-wide weight-gradient shapes: the big tile still fills the chip once K is split
    if (batch == 1 && M >= 128 && N >= 128 && t128 * (K / 1024) >= 192) return CFG_128;
    const long long t64 = (long long)((M + 63) / 64) * ((N + 63) / 64) * batch;
    // few tiles: 32x32 tiles put one on every CU and four wave groups quarter the dependent K chain
    if (K >= 256 && t64 <= 288) return CFG_32_K4;
    return K >= 128 ? CFG_64_K2 : CFG_64;
}

inline void cfg_dims(TileCfg c, int& bm, int& bn)
{
    switch (c) {
        case CFG_128: bm = 128; bn = 128; break;
        case CFG_64x128: bm = 64; bn = 128; break;
        case CFG_32_K4: bm = 32; bn = 32; break;
        case CFG_256x128: bm = 256; bn = 128; break;
        default: bm = 64; bn = 64; break;
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int KG, int PREC = 0>
int launch_cfg(const KArgs& k, bool akc, bool bkc, int grid, hipStream_t s)
{
    const dim3 block(64 * WAVES_M * WAVES_N * KG);
    if (akc && bkc)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, WAVES_M, WAVES_N, KG, true, true, PREC>), dim3(grid), block, 0, s, k);
    else if (akc && !bkc)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, WAVES_M, WAVES_N, KG, true, false, PREC>), dim3(grid), block, 0, s, k);
    else if (!akc && !bkc)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, WAVES_M, WAVES_N, KG, false, false, PREC>), dim3(grid), block, 0, s, k);
    else
        return set_error(GTG_ERR_UNSUPPORTED, "gemm: A M-contiguous with B K-contiguous is not instantiated");
    return check_launch("gemm_kernel");
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int KG>
int launch_cfg_side(const KArgs& k, const SideJobs& sj, bool akc, bool bkc, int grid, hipStream_t s)
{
    const dim3 block(64 * WAVES_M * WAVES_N * KG), g(grid + sj.blocks());
    if (akc && bkc)
        hipLaunchKernelGGL((gemm_side_kernel<BM, BN, WAVES_M, WAVES_N, KG, true, true>), g, block, 0, s, k, sj);
    else if (akc && !bkc)
        hipLaunchKernelGGL((gemm_side_kernel<BM, BN, WAVES_M, WAVES_N, KG, true, false>), g, block, 0, s, k, sj);
    else if (!akc && !bkc)
        hipLaunchKernelGGL((gemm_side_kernel<BM, BN, WAVES_M, WAVES_N, KG, false, false>), g, block, 0, s, k, sj);
    else
        return set_error(GTG_ERR_UNSUPPORTED, "gemm: A M-contiguous with B K-contiguous is not instantiated");
    return check_launch("gemm_side_kernel");
}

// C = A B^T with both operands given as three bf16 planes (GemmDesc::planes).
int launch_planes(const KArgs& k, int grid, hipStream_t s)
{
    if (k.inv_a_rows != nullptr)
        hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 2, 1, true, true, 5>), dim3(grid), dim3(256), 0, s, k);
    else
        hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 2, 1, true, true, 4>), dim3(grid), dim3(256), 0, s, k);
    return check_launch("gemm_kernel(planes)");
}

// Resident 128x128-tile blocks of the device: two per CU (LDS bound).
inline int tail_slots()
{
    static const int v = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int v2 = 2 * (cus > 0 ? cus : 256);
        return v2 > 1024 ? 1024 : v2;  // the bound gemm_tail_workspace() sizes for
    }();
    return v;
}

}  // namespace

int split_planes(const float* src, long long rows, long long cols, long long ld, bool transpose, void* planes, hipStream_t s)
{
    const long long total = rows * cols;
    if (total <= 0) return GTG_OK;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, rows, cols, ld,
                       transpose ? 1 : 0, reinterpret_cast<unsigned short*>(planes));
    return check_launch("split_planes_kernel");
}

int absmax_partials(const float* x, long long rows, long long cols, long long ld, float* out, hipStream_t s)
{
    if (rows > 0x7fffffffLL || cols > 0x7fffffffLL) return set_error(GTG_ERR_INVALID, "absmax: array too large");
    AbsmaxJob ja, jb;
    ja.p = x;
    ja.ld = ld;
    ja.rows = (int)rows;
    ja.cols = (int)cols;
    ja.rows_dev = ja.cols_dev = nullptr;
    ja.bound = 0.f;
    jb = ja;
    hipLaunchKernelGGL(absmax_pair_kernel, dim3(ABSMAX_PARTS), dim3(256), 0, s, ja, jb, out);  // operand A's blocks only
    return check_launch("absmax_pair_kernel");
}

void gemm_tail_workspace(size_t* ws_bytes, size_t* counter_bytes)
{
    const size_t slots = 1024;  // upper bound on tail_slots(): sized without touching the device
    *ws_bytes = slots * 128 * 128 * sizeof(float);
    *counter_bytes = slots * sizeof(int32_t);
}

int gemm_pick_splits(int M, int N, int K, int batch)
{
    if (batch != 1) return 1;
    int bm, bn;
    cfg_dims(pick_cfg(M, N, K, batch, EPI_STORE), bm, bn);
    const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    // (256 tiles are half a round of the 512 resident block slots: still worth cutting K in two)
    if (tiles > 256 || K < 1024) return 1;
    // one full round of resident blocks (two per CU on the 256 CUs for every configuration that gets here)
    long long want = 512 / tiles;
    const long long maxs = K / 256;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    int kps = (int)((K + want - 1) / want);
    kps = (kps + BK - 1) / BK * BK;
    return (K + kps - 1) / kps;
}

bool gemm_fused_epilogue_ok(int M, int N, int K)
{
    const TileCfg cfg = pick_cfg(M, N, K, 1, EPI_STORE);
    return cfg == CFG_32_K4 || cfg == CFG_64_K2;
}

long long gemm_bce_partials(int M, int N, int K)
{
    const TileCfg cfg = pick_cfg(M, N, K, 1, EPI_SIGMOID_BCE);
    int bm, bn;
    cfg_dims(cfg, bm, bn);
    const int waves = cfg == CFG_32_K4 ? 4 : 8;  // waves per tile: 1 x 4 K groups, or 4 x 2 (every wave stores rows)
    return (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * waves;
}

size_t gemm_splitk_workspace(int M, int N, int K, int batch)
{
    const int s = gemm_pick_splits(M, N, K, batch);
    return s > 1 ? (size_t)s * (size_t)M * (size_t)N * sizeof(float) : 0;
}

namespace {

// A validated launch: kernel arguments, tile configuration and grid of one GemmDesc.
struct GemmPlan {
    KArgs k;
    TileCfg cfg;
    long long grid;
    int splits;
    bool empty;  // nothing to compute
};

int gemm_plan(const GemmDesc& d, GemmPlan& p)
{
    p.empty = true;
    if (d.M < 0 || d.N < 0 || d.K < 0 || d.batch < 0) return set_error(GTG_ERR_INVALID, "gemm: negative dimension");
    if (d.M == 0 || d.N == 0 || d.batch == 0) return GTG_OK;
    p.empty = false;
    if (!d.A || !d.B) return set_error(GTG_ERR_INVALID, "gemm: null operand");
    if (d.epi == EPI_POOL) {
        if (d.M > 64 || !d.pool_val || !d.pool_idx)
            return set_error(GTG_ERR_INVALID, "gemm: pooled epilogue needs M <= 64 and pool outputs");
    } else if (!d.C) {
        return set_error(GTG_ERR_INVALID, "gemm: null output");
    }
    if (d.epi == EPI_MASK && !d.aux) return set_error(GTG_ERR_INVALID, "gemm: mask epilogue without aux");
    if (d.epi == EPI_SIGMOID_BCE || d.epi == EPI_LATENT_BWD) {
        if (d.batch != 1 || d.splits > 1 || d.m_dev != nullptr || d.k_dev != nullptr || d.batch_dim != nullptr ||
            d.tail_ws != nullptr || d.planes)
            return set_error(GTG_ERR_INVALID, "gemm: a fused loss epilogue needs batch 1, no split-K, static extents");
        const GemmFuse& f = d.fuse;
        if (d.epi == EPI_SIGMOID_BCE &&
            (!f.bce_target || !f.bce_dlogits || !f.bce_partial || f.bce_n < 1 || d.N != f.bce_n * (f.bce_n + 1) / 2))
            return set_error(GTG_ERR_INVALID, "gemm: BCE epilogue needs targets, outputs and N = n(n+1)/2");
        if (d.epi == EPI_LATENT_BWD &&
            (!f.mu || !f.lv || !f.eps || !f.dmu || !f.dlv || d.N % 4 != 0 || f.ld_out % 4 != 0 ||
             ((reinterpret_cast<uintptr_t>(f.mu) | reinterpret_cast<uintptr_t>(f.lv) | reinterpret_cast<uintptr_t>(f.eps) |
               reinterpret_cast<uintptr_t>(f.dmu) | reinterpret_cast<uintptr_t>(f.dlv)) & 15) != 0))
            return set_error(GTG_ERR_INVALID, "gemm: latent-backward epilogue needs 16-byte aligned rows of a multiple of 4");
    }

    TileCfg cfg = d.planes ? CFG_128 : pick_cfg(d.M, d.N, d.K, d.batch, d.epi);
    if ((d.epi == EPI_SIGMOID_BCE || d.epi == EPI_LATENT_BWD) && cfg != CFG_32_K4 && cfg != CFG_64_K2)
        return set_error(GTG_ERR_INVALID, "gemm: the fused loss epilogues exist for batch-sized products only "
                                          "(gemm_fused_epilogue_ok)");
    // Large products on the bf16 pipes: a 256x128 tile shared by eight waves (one block per CU) moves a quarter
    // less operand data per multiply-add than the 128x128 tile, through global memory, the split arithmetic and
    // the LDS stores alike.  Taken when it still gives every CU at least two tiles; a caller that offers a tail
    // split (few rounds of tiles, or a row count known only on the device) keeps the 128x128 tile until the
    // product has eight rounds of the large one, where the last round's imbalance no longer matters.
    // Measured on the text branch's products: a weight gradient (TN, both operands row-contiguous) gains 7-10 %,
    // products with a K-contiguous A lose about as much, so only the former take it.
    if (cfg == CFG_128 && !d.planes && !d.a_kc && !d.b_kc &&
        (g_matrix_mode == GTG_MATRIX_BF16X3 || g_matrix_mode == GTG_MATRIX_BF16 || g_matrix_mode == GTG_MATRIX_F16X2) && d.batch == 1 && d.splits <= 1 && d.colsum_part == nullptr && d.epi != EPI_POOL &&
        (long long)((d.M + 255) / 256) * ((d.N + 127) / 128) >= (d.tail_ws != nullptr ? 2048 : 512))
        cfg = CFG_256x128;
    int bm, bn;
    cfg_dims(cfg, bm, bn);
    if (d.planes) {
        if (d.batch != 1 || d.splits > 1 || d.epi == EPI_POOL || d.k_dev != nullptr || d.batch_dim != nullptr ||
            d.K % 16 != 0 || d.lda < d.M || d.ldb < d.N || d.planeA % 8 != 0 || d.planeB % 8 != 0 ||
            (reinterpret_cast<uintptr_t>(d.A) & 15) != 0 || (reinterpret_cast<uintptr_t>(d.B) & 15) != 0)
            return set_error(GTG_ERR_INVALID, "gemm: plane operands need batch 1, no split-K, K a multiple of 16 and "
                                              "16-byte aligned rows and planes");
        if ((d.planes_inv_a == nullptr) != (d.planes_inv_b == nullptr))
            return set_error(GTG_ERR_INVALID, "gemm: fp16 plane operands need the inverse scales of both");
    }

    KArgs& k = p.k;
    k.planeA = d.planeA;
    k.planeB = d.planeB;
    k.absmax = d.absmax_ws;
    k.inv_a_rows = d.planes ? d.planes_inv_a : nullptr;
    k.inv_b_rows = d.planes ? d.planes_inv_b : nullptr;
    k.fuse = d.fuse;
    k.A = d.A;
    k.B = d.B;
    k.C = d.C;
    k.bias = d.bias;
    k.aux = d.aux;
    k.pool_val = d.pool_val;
    k.pool_idx = d.pool_idx;
    k.strideA = d.strideA;
    k.strideB = d.strideB;
    k.strideC = d.strideC;
    k.strideAux = d.strideAux;
    k.strideBias = d.strideBias;
    k.M = d.M;
    k.N = d.N;
    k.K = d.K;
    k.lda = d.lda;
    k.ldb = d.ldb;
    k.ldc = d.ldc;
    k.ldaux = d.ldaux;
    k.tiles_m = (d.M + bm - 1) / bm;
    k.tiles_n = (d.N + bn - 1) / bn;
    k.epi = d.epi;
    k.m_dev = d.m_dev;
    k.k_dev = d.k_dev;
    k.batch_dim = d.batch_dim;
    k.batch_off = d.batch_off;
    k.off_b = d.off_b ? 1 : 0;
    k.off_c = d.off_c ? 1 : 0;
    k.proto_base = d.proto_base;
    k.vecA = ((reinterpret_cast<uintptr_t>(d.A) & 15) == 0) && (d.lda % 4 == 0) && (d.strideA % 4 == 0);
    k.vecB = ((reinterpret_cast<uintptr_t>(d.B) & 15) == 0) && (d.ldb % 4 == 0) && (d.strideB % 4 == 0);

    int splits = 1;
    if (d.splits > 1) {
        if (d.batch != 1 || !d.slabs || (d.epi != EPI_STORE && d.epi != EPI_ACCUM) || d.bias)
            return set_error(GTG_ERR_INVALID, "gemm: split-K needs batch 1, a slab workspace and a plain epilogue");
        splits = d.splits;
    }
    int kps = (d.K + splits - 1) / splits;
    kps = (kps + BK - 1) / BK * BK;
    if (kps == 0) kps = BK;
    splits = d.K > 0 ? (d.K + kps - 1) / kps : 1;
    if (splits > 1 && splits > d.splits) return set_error(GTG_ERR_INVALID, "gemm: split count grew past the workspace");
    k.splits = splits;
    k.k_per_split = kps;
    k.to_slabs = splits > 1 ? 1 : 0;
    if (k.to_slabs) k.C = d.slabs;
    if (k.to_slabs)
        k.vecC = ((reinterpret_cast<uintptr_t>(d.slabs) & 15) == 0) && (d.N % 4 == 0) && (((long long)d.M * d.N) % 4 == 0);
    else
        k.vecC = ((reinterpret_cast<uintptr_t>(d.C) & 15) == 0) && (d.ldc % 4 == 0) && (d.strideC % 4 == 0);
    k.vecAux = d.aux && ((reinterpret_cast<uintptr_t>(d.aux) & 15) == 0) && (d.ldaux % 4 == 0) && (d.strideAux % 4 == 0);

    k.tail_ws = nullptr;
    k.tail_counters = nullptr;
    k.tail_slots = 0;
    k.colsum_part = nullptr;
    if (d.colsum_part != nullptr) {
        if (cfg != CFG_128 || splits != 1 || d.batch != 1 || d.epi == EPI_POOL)
            return set_error(GTG_ERR_INVALID, "gemm: column sums ride only on the 128x128 configuration, batch 1, no split-K");
        k.colsum_part = d.colsum_part;
    }
    long long grid = (long long)k.tiles_m * k.tiles_n * splits * d.batch;
    if (d.tail_ws != nullptr && d.tail_counters != nullptr && cfg == CFG_128 && d.batch == 1 && splits == 1 &&
        d.epi != EPI_POOL) {
        k.tail_ws = d.tail_ws;
        k.tail_counters = d.tail_counters;
        // resident blocks of the kernel that will run: two per CU, three of the fp16-plane kernel
        k.tail_slots = (d.planes && d.planes_inv_a != nullptr) ? tail_slots() / 2 * 3 : tail_slots();
        if (k.tail_slots > 1024) k.tail_slots = 1024;  // what gemm_tail_workspace() sizes the slabs and counters for
        grid += k.tail_slots;  // the split tail adds fewer than tail_slots blocks
    }
    if (grid > 0x7fffffffLL) return set_error(GTG_ERR_INVALID, "gemm: grid too large");
    p.cfg = cfg;
    p.grid = grid;
    p.splits = splits;
    return GTG_OK;
}

}  // namespace

int gemm(const GemmDesc& d, hipStream_t s)
{
    GemmPlan p;
    if (int rc = gemm_plan(d, p)) return rc;
    const bool has_side = d.side != nullptr && d.side->blocks() > 0;
    // side jobs ride in the launch of a K-group configuration; in front of any other product they run alone
    const bool carry = has_side && !p.empty && p.splits == 1 && (p.cfg == CFG_64_K2 || p.cfg == CFG_32_K4) &&
                       p.grid + d.side->blocks() <= 0x7fffffffLL;
    if (has_side && !carry)
        if (int rc = side_jobs_launch(*d.side, s)) return rc;
    if (p.empty) return GTG_OK;
    const KArgs& k = p.k;
    const TileCfg cfg = p.cfg;
    const long long grid = p.grid;
    const int splits = p.splits;

    static const char* const kNames[CFG_COUNT][3] = {
        {"gemm128_nt", "gemm128_nn", "gemm128_tn"},
        {"gemm64x128_nt", "gemm64x128_nn", "gemm64x128_tn"},
        {"gemm64_nt", "gemm64_nn", "gemm64_tn"},
        {"gemm64k2_nt", "gemm64k2_nn", "gemm64k2_tn"},
        {"gemm32k4_nt", "gemm32k4_nn", "gemm32k4_tn"},
        {"gemm256x128_nt", "gemm256x128_nn", "gemm256x128_tn"}};
    const int lay = d.a_kc ? (d.b_kc ? 0 : 1) : 2;
    // a launch whose row count (":m"), depth (":k") or per-graph extent (":g") is read on the device is tagged so:
    // the dimensions recorded with it are then upper bounds, not the work done
    char tag[32];
    snprintf(tag, sizeof(tag), "%s%s%s", kNames[(int)cfg][lay], d.planes ? "_planes" : "",
             d.m_dev ? ":m" : d.k_dev ? ":k" : d.batch_dim ? ":g" : "");
    ProfScope prof(s, tag, (long long)d.M * d.batch, d.N, d.K);
    // the matrix pipe applies to the 128x128 configuration (the MFMA-bound GEMMs); the small-tile
    // configurations are latency-bound and stay on the fp32 pipe
    int prec = (cfg == CFG_128 || cfg == CFG_256x128) ? g_matrix_mode : 0;
    if (prec == GTG_MATRIX_F16X2 && !d.planes) {
        // the two-plane fp16 split needs the operands' largest magnitudes first; a caller without the scratch
        // for them, and the batched forms, get the three-plane bf16 split (the same accuracy class)
        if (d.absmax_ws == nullptr || d.batch != 1 || d.batch_dim != nullptr || d.batch_off != nullptr) {
            prec = GTG_MATRIX_BF16X3;
        } else if (!d.absmax_ready) {
            AbsmaxJob ja, jb;
            ja.p = d.A;
            ja.ld = d.lda;
            ja.rows = d.a_kc ? d.M : d.K;
            ja.cols = d.a_kc ? d.K : d.M;
            ja.rows_dev = d.a_kc ? d.m_dev : d.k_dev;
            ja.cols_dev = d.a_kc ? d.k_dev : d.m_dev;
            ja.bound = d.a_partials_ready ? -1.f : d.a_bound;
            jb.bound = d.b_bound;
            jb.p = d.B;
            jb.ld = d.ldb;
            jb.rows = d.b_kc ? d.N : d.K;
            jb.cols = d.b_kc ? d.K : d.N;
            jb.rows_dev = d.b_kc ? nullptr : d.k_dev;
            jb.cols_dev = d.b_kc ? d.k_dev : nullptr;
            ProfScope prof_absmax(s, "absmax_pair", (long long)d.M, d.N, d.K);
            hipLaunchKernelGGL(absmax_pair_kernel, dim3(2 * ABSMAX_PARTS), dim3(256), 0, s, ja, jb, d.absmax_ws);
            if (int rc0 = check_launch("absmax_pair_kernel")) return rc0;
        }
    }
    int rc;
    switch (cfg) {
        case CFG_128:
            if (d.planes)
                rc = launch_planes(k, (int)grid, s);
            else if (prec == 3)
                rc = launch_cfg<128, 128, 2, 2, 1, 3>(k, d.a_kc, d.b_kc, (int)grid, s);
            else if (prec == 2)
                rc = launch_cfg<128, 128, 2, 2, 1, 2>(k, d.a_kc, d.b_kc, (int)grid, s);
            else if (prec == 1)
                rc = launch_cfg<128, 128, 2, 2, 1, 1>(k, d.a_kc, d.b_kc, (int)grid, s);
            else
                rc = launch_cfg<128, 128, 2, 2, 1>(k, d.a_kc, d.b_kc, (int)grid, s);
            break;
        case CFG_64x128: rc = launch_cfg<64, 128, 1, 4, 1>(k, d.a_kc, d.b_kc, (int)grid, s); break;
        case CFG_256x128:
            if (prec == 3)
                rc = launch_cfg<256, 128, 4, 2, 1, 3>(k, d.a_kc, d.b_kc, (int)grid, s);
            else if (prec == 2)
                rc = launch_cfg<256, 128, 4, 2, 1, 2>(k, d.a_kc, d.b_kc, (int)grid, s);
            else
                rc = launch_cfg<256, 128, 4, 2, 1, 1>(k, d.a_kc, d.b_kc, (int)grid, s);
            break;
        case CFG_64_K2:
            rc = carry ? launch_cfg_side<64, 64, 2, 2, 2>(k, *d.side, d.a_kc, d.b_kc, (int)grid, s)
                       : launch_cfg<64, 64, 2, 2, 2>(k, d.a_kc, d.b_kc, (int)grid, s);
            break;
        case CFG_32_K4:
            rc = carry ? launch_cfg_side<32, 32, 1, 1, 4>(k, *d.side, d.a_kc, d.b_kc, (int)grid, s)
                       : launch_cfg<32, 32, 1, 1, 4>(k, d.a_kc, d.b_kc, (int)grid, s);
            break;
        default: rc = launch_cfg<64, 64, 2, 2, 1>(k, d.a_kc, d.b_kc, (int)grid, s); break;
    }
    if (rc != GTG_OK) return rc;
    if (splits > 1) {
        const long long mn = (long long)d.M * d.N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, s, d.slabs, splits, mn,
                           d.N, d.C, d.ldc, d.epi == EPI_ACCUM ? 1 : 0);
        return check_launch("splitk_reduce_kernel");
    }
    return GTG_OK;
}

int gemm_pair(const GemmDesc& d0, const GemmDesc& d1, hipStream_t s)
{
    GemmPlan p0, p1;
    if (int rc = gemm_plan(d0, p0)) return rc;
    if (int rc = gemm_plan(d1, p1)) return rc;
    const bool plain = !p0.empty && !p1.empty && p0.splits == 1 && p1.splits == 1 && p0.k.tail_ws == nullptr &&
                       p1.k.tail_ws == nullptr && p0.grid + p1.grid <= 0x7fffffffLL;
    const bool tn0 = !d0.a_kc && !d0.b_kc, nn1 = d1.a_kc && !d1.b_kc;
    // A 64x64 job beside a 32x32 one (the latter over eight wave groups to share the block size) measured
    // slower than the two launches: its LDS leaves one block per CU.
    if (!(plain && tn0 && nn1 && p0.cfg == CFG_32_K4 && p1.cfg == CFG_32_K4)) {
        if (int rc = gemm(d0, s)) return rc;
        return gemm(d1, s);
    }
    if (d1.side != nullptr && d1.side->blocks() > 0)
        if (int rc = side_jobs_launch(*d1.side, s)) return rc;
    SideJobs sj;
    if (d0.side != nullptr) sj = *d0.side;
    ProfScope prof(s, "gemm_pair32k4_tn+nn", (long long)d0.M * d0.batch, d0.N, d0.K);
    using J0 = GemmJob<32, 32, 1, 1, 4, false, false>;
    using J1 = GemmJob<32, 32, 1, 1, 4, true, false>;
    hipLaunchKernelGGL((gemm_pair_kernel<J0, J1>), dim3((unsigned)(p0.grid + p1.grid + sj.blocks())), dim3(J0::THREADS), 0,
                       s, p0.k, p1.k, (int)p0.grid, d1.K > d0.K ? 1 : 0, sj);
    return check_launch("gemm_pair_kernel");
}

}  // namespace gtg
