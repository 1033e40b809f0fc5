// In-kernel cycle budget of the wave-specialised persistent halo-staged convolution (conv_hdmaw.hpp) on one ResNet-34 shape, no Python:
// builds the kernel with LBC_HDMAW_PROF (per-wave s_memtime sums) and, optionally, one of the timing-experiment switches
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ilearningbycheating_amd/csrc -Iscripts/probe [-DLBC_HDMAW_ABL_NOMFMA ...] scripts/probe/hdmaw_prof.hip
//         learningbycheating_amd/csrc/lbc_util.cpp -o scripts/probe/hdmaw_prof[_variant]
//   scripts/probe/hdmaw_prof [H W C K N fill]     (default 10 24 256 256 256 1: layer 3 at batch 256, random operands; fill 0 = zeros, 2 = half zeros)
// Prints the launch time (HIP events, 20 launches) and the stamp sums averaged over workgroups, per K-tile.
#define LBC_HDMAW_PROF 1
#include "conv_hdmaw.hpp"       // (this directory: the experiment is not part of the library)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int H = argc > 1 ? atoi(argv[1]) : 10, W = argc > 2 ? atoi(argv[2]) : 24, C = argc > 3 ? atoi(argv[3]) : 256, K = argc > 4 ? atoi(argv[4]) : 256;
    const int N = argc > 5 ? atoi(argv[5]) : 256;
    const int fill = argc > 6 ? atoi(argv[6]) : 1;       // operands: 0 zeros, 1 uniform (-0.5, 0.5), 2 the activations half zeros (post-ReLU-like)
    const int M = N * H * W;
    std::vector<unsigned short> hx((size_t)M * C), hw((size_t)K * 9 * C);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; const float f = ((s >> 8) & 0xffff) / 65536.f - 0.5f; return (unsigned short)(__builtin_bit_cast(unsigned, f) >> 16); };
    for (auto& v : hx) { v = rnd(); if (fill == 0 || (fill == 2 && (v & 0x8000))) v = 0; }
    for (auto& v : hw) { v = rnd(); if (fill == 0) v = 0; }
    void *x, *w, *y, *zero;
    CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&w, hw.size() * 2)); CK(hipMalloc(&y, (size_t)M * K * 2)); CK(hipMalloc(&zero, 256));
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(zero, 0, 256));
    IgemmArgs a = IgemmArgs();
    a.x = x; a.w = w; a.y = y;
    a.N = N; a.H = H; a.W = W; a.C = C; a.OH = H; a.OW = W; a.K = K; a.KH = 3; a.KW = 3; a.S = 1; a.P = 1; a.M = M; a.LH = H; a.LW = W; a.ostep = 1; a.nphase = 1;
    a.bf16 = 1; a.act_bf16 = 1; a.w_bf16 = 1;
    const int ntiles = lbc_cdiv(M, 256) * (K / 128);
    const int tpw = lbc_cdiv(ntiles, 256);
    const unsigned grid = (unsigned)lbc_cdiv(ntiles, tpw);
    const bool big = 256 + 2 * W + 2 > 320 - 8;
    auto launch = [&]() {
        return big ? conv_hdmaw_launch_shape<256, 128, 368>(a, 0, zero, ntiles, tpw, dim3(grid), nullptr)
                   : conv_hdmaw_launch_shape<256, 128, 320>(a, 0, zero, ntiles, tpw, dim3(grid), nullptr);
    };
    for (int i = 0; i < 3; ++i) if (launch()) { fprintf(stderr, "launch failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20;
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters, flop = 2.0 * M * K * 9.0 * C;
    std::vector<unsigned long long> p(256 * 8 * 4);
    CK(hipMemcpyFromSymbol(p.data(), HIP_SYMBOL(g_hdmaw_prof), p.size() * 8));
    const int ktiles = tpw * 9 * (C / 64);
    printf("fill %d; H %d W %d C %d K %d N %d: %d tiles, %d per workgroup, %u workgroups, %d K-tiles per workgroup\n", fill, H, W, C, K, N, ntiles, tpw, grid, ktiles);
    printf("launch %.1f us = %.0f TF/s\n", us, flop / us * 1e-6);
    double c[4] = {0, 0, 0, 0}, l[4] = {0, 0, 0, 0}, cmax = 0;
    const unsigned full = (unsigned)(ntiles / tpw);      // workgroups with a full tile count
    for (unsigned b = 0; b < full; ++b)
        for (int wv = 0; wv < 8; ++wv)
            for (int k = 0; k < 4; ++k) {
                const double v = (double)p[(b * 8 + wv) * 4 + k];
                (wv < 4 ? c : l)[k] += v / (4.0 * full);
                if (wv < 4 && k == 3 && v > cmax) cmax = v;
            }
    printf("multiplying waves (s_memtime ticks, mean over %u workgroups x 4 waves): total %.0f (max %.0f) = %.0f per K-tile; K-tile segments %.0f per K-tile, in the K-tile barrier %.0f per K-tile, epilogue %.0f per tile\n",
           full, c[3], cmax, c[3] / ktiles, c[0] / ktiles, c[1] / ktiles, c[2] / tpw);
    printf("loading waves: total %.0f; requests %.0f per K-tile, counted vmcnt wait %.0f per K-tile, barrier %.0f per K-tile\n", l[3], l[0] / ktiles, l[1] / ktiles, l[2] / ktiles);
    printf("effective clock if a tick is a shader cycle: %.2f GHz over the launch (total ticks / launch time)\n", cmax / us * 1e-3);
    return 0;
}
