// Which MFMA shape delivers more bf16 FLOP/s under the package power cap?  Pure register-operand MFMA streams (no LDS, no memory), operands =
// random bf16 values rotated over eight fragment pairs (so consecutive MFMAs see different data, as in the convolution K loops), four or
// eight waves per CU, run long enough (~4 s per variant) for the power manager to settle.  Prints TFLOP/s per variant; sample rocm-smi next to it:
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/mfma_power_probe.hip -o scripts/probe/mfma_power_probe && scripts/probe/mfma_power_probe [zero]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>      // 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(256) void mfma_loop(const bf16x8* __restrict__ src, float* __restrict__ out, int iters)
{
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x + 256 * i) & 4095]; b[i] = src[(threadIdx.x + 256 * i + 2048) & 4095]; }
    if constexpr (SHAPE == 0) {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + k) & 7], b[i], acc[k], 0, 0, 0);
        }
        float s = 0.f;
        for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x4 acc[8];
        for (int k = 0; k < 8; ++k) for (int e = 0; e < 4; ++e) acc[k][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + k) & 7], b[i], acc[k], 0, 0, 0);
        }
        float s = 0.f;
        for (int k = 0; k < 8; ++k) for (int e = 0; e < 4; ++e) s += acc[k][e];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

int main(int argc, char** argv)
{
    const bool zero = argc > 1 && !strcmp(argv[1], "zero");
    std::vector<unsigned short> h(4096 * 8);
    unsigned s = 777u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; const float f = ((s >> 8) & 0xffff) / 65536.f - 0.5f; v = zero ? 0 : (unsigned short)(__builtin_bit_cast(unsigned, f) >> 16); }
    void *src, *out;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, 1024 * 256 * 4);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape = 0; shape < 2; ++shape)
        for (int wgs = 256; wgs <= 512; wgs *= 2) {       // one / two 4-wave workgroups per CU
            const int iters = 20000;
            const double flop_per_iter = shape == 0 ? 32.0 * 2 * 32 * 32 * 16 : 64.0 * 2 * 16 * 16 * 32;      // per wave and outer iteration
            float ms = 0.f;
            double best = 0;
            for (int rep = 0; rep < 12; ++rep) {          // ~4 s in total per variant
                hipEventRecord(e0, nullptr);
                if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(wgs), dim3(256), 0, nullptr, (const bf16x8*)src, (float*)out, iters);
                else hipLaunchKernelGGL(mfma_loop<1>, dim3(wgs), dim3(256), 0, nullptr, (const bf16x8*)src, (float*)out, iters);
                hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                best = flop_per_iter * iters * 4.0 * wgs / (ms * 1e-3) * 1e-12;      // the LAST repetition: settled clocks
            }
            printf("%s operands, %s, %d workgroups of 4 waves: %.1f ms per launch, %.0f TFLOP/s (last of 12 launches)\n", zero ? "zero" : "random",
                   shape == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", wgs, ms, best);
            fflush(stdout);
        }
    return 0;
}
