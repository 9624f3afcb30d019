// Sustained rate of the matrix pipes on this GPU with operands in registers (no memory traffic):
// the practical ceiling against which the GEMM kernels' MFMA utilisation is read.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed)
{
    f32x16 acc[4] = {};
    u32x4 ua = {seed + threadIdx.x, seed * 3 + 1, seed ^ 0x3f803f80u, 0x3f813f7fu};
    u32x4 ub = {0x3f803e80u + threadIdx.x, 0x3e803f80u, seed | 0x3c003c00u, 0x3f003f00u};
    float fa = 1.0f + threadIdx.x * 1e-3f, fb = 0.5f + seed * 1e-9f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0)
                acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[u & 3], 0, 0, 0);
            else
                acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub),
                                                                     acc[u & 3], 0, 0, 0);
            if (MODE == 2) {
                // the vector work of the three-way split that accompanies one MFMA in the GEMM (about 5.4 instructions)
                float x0 = fa, x1 = fb;
                const unsigned p0 = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
                x0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xffff0000u);
                x1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
                fa = x0 + 1.0f;
                fb = x1 + 0.5f;
                ua[u & 3] ^= p0;
            }
        }
    }
    float s = 0.f;
    for (int u = 0; u < 4; ++u)
        for (int e = 0; e < 16; ++e) s += acc[u][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int wpc = 1; wpc <= 2; ++wpc) {  // blocks (of 4 waves) per CU
            const int grid = 256 * wpc, iters = mode ? 40000 : 20000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0)
                    hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, 12345u);
                else if (mode == 1)
                    hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, 12345u);
                else
                    hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, 12345u);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)grid * 4 * iters * 8 * (mode ? 32.0 * 32 * 16 * 2 : 32.0 * 32 * 2 * 2);
            printf("%s  %d waves/SIMD  %.1f ms  %.0f TFLOP/s\n", mode == 2 ? "bf16 MFMA + split VALU    " : mode ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32  ", wpc, ms,
                   flops / ms / 1e9);
        }
    return 0;
}
