// Launch policy of the persistent halo-staged convolution (kernel: conv_hdmap.hpp; one translation unit per tile shape so that
// the 15 instantiations compile in parallel).
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include "conv_hdmap.hpp"      // (nothing is instantiated here)

int lbc_conv_hdmap_launch_256x128_320(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, unsigned grid, hipStream_t s);
int lbc_conv_hdmap_launch_256x128_384(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, unsigned grid, hipStream_t s);
int lbc_conv_hdmap_launch_128x256_192(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, unsigned grid, hipStream_t s);
int lbc_conv_hdmap_launch_128x64_192(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, unsigned grid, hipStream_t s, int nsplit, int kgroups);

namespace {

// Second launch of a split-K convolution (conv_hdmap_k<.., EPI 3> left f32 partial tiles [range][M][K] in IgemmArgs::split_ws): sums
// the ranges in the order 0 .. nsplit - 1 and does what the unsplit kernel's epilogue does with its accumulators -- affine, bias, residual add, ReLU,
// the bf16 store, and per 128-row block (the unsplit launch's tile rows: the same number of partial rows) either the (sum, sum of
// squares) of the f32 value or, FUSED, the BatchNorm-backward sums of the stored (rounded, masked) gradient (IgemmArgs::bnb_*).
// One workgroup per (128 rows, 64 columns): thread = (row group of 32, 8-column segment), four rows each.
template <bool FUSED>
__global__ __launch_bounds__(256) void conv_split_epilogue_k(IgemmArgs a, const int nsplit)
{
    __shared__ float red[2][32][64];
    const int tid = threadIdx.x, rg = tid >> 3, seg = tid & 7;
    const int mt = blockIdx.x, n0 = blockIdx.y * 64, col = n0 + seg * 8;
    const size_t plane = (size_t)a.M * (size_t)a.K;
    __bf16* y = static_cast<__bf16*>(a.y);
    const __bf16* resid = static_cast<const __bf16*>(a.resid);
    const __bf16* by = static_cast<const __bf16*>(a.bnb_y);
    const __bf16* bmask = static_cast<const __bf16*>(a.bnb_mask);      // (nullptr: the mask is bn(bnb_y) > 0)
    size_t o[4];
    bool live[4];
    f32x8 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = mt * 128 + rg + 32 * q;
        live[q] = m < a.M;
        o[q] = (size_t)(live[q] ? m : 0) * (size_t)a.K + (size_t)col;
        v[q] = ParamVec<8>::ld(a.split_ws + o[q]);
    }
    for (int r = 1; r < nsplit; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += ParamVec<8>::ld(a.split_ws + (size_t)r * plane + o[q]);
    f32x8 u1 = ParamVec<8>::splat(0.f), u2 = u1;
    f32x8 bsc = u1, bsh = u1, bmu = u1, biv = u1;
    if constexpr (FUSED) {
        if (!bmask) { bsc = ParamVec<8>::ld(a.bnb_scale + col); bsh = ParamVec<8>::ld(a.bnb_shift + col); }
        bmu = ParamVec<8>::ld(a.bnb_mean + col); biv = ParamVec<8>::ld(a.bnb_invstd + col);
    }
    const f32x8 psc = a.post_scale ? ParamVec<8>::ld(a.post_scale + col) : ParamVec<8>::splat(1.f);
    const f32x8 psh = a.post_scale ? ParamVec<8>::ld(a.post_shift + col) : ParamVec<8>::splat(0.f);
    const f32x8 bia = a.bias ? ParamVec<8>::ld(a.bias + col) : ParamVec<8>::splat(0.f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x8 t = v[q];
        if (a.post_scale) t = t * psc + psh;
        if (a.bias) t += bia;
        if (resid) t += __builtin_convertvector(*reinterpret_cast<const bf16x8*>(resid + o[q]), f32x8);
        if (a.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = fmaxf(t[e], 0.f);
        }
        bf16x8 ch = __builtin_convertvector(t, bf16x8);
        if constexpr (FUSED) {
            const f32x8 yf = __builtin_convertvector(*reinterpret_cast<const bf16x8*>(by + o[q]), f32x8);
            f32x8 g = __builtin_convertvector(ch, f32x8);
            const f32x8 z = bmask ? __builtin_convertvector(*reinterpret_cast<const bf16x8*>(bmask + o[q]), f32x8) : yf * bsc + bsh;
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = z[e] > 0.f ? g[e] : 0.f;
            ch = __builtin_convertvector(g, bf16x8);
            if (live[q]) { u1 += g; u2 += g * (yf - bmu) * biv; }
        } else if (live[q]) {
            u1 += t; u2 += t * t;
        }
        if (live[q]) *reinterpret_cast<bf16x8*>(y + o[q]) = ch;
    }
    if (a.stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[0][rg][seg * 8 + e] = u1[e]; red[1][rg][seg * 8 + e] = u2[e]; }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63;
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 32; ++r) sum += red[which][r][c];
            a.stats[(size_t)(a.stat_row0 + mt) * 2 * (size_t)a.K + (size_t)which * (size_t)a.K + (size_t)(n0 + c)] = sum;
        }
    }
}

}  // namespace

// Split-K ranges of a launch of the four-wave 128 x 64 shape (1 = none).  Launches with few tiles leave CUs idle AND run one wave per
// SIMD where they run (nothing hides a wave's LDS / DMA latencies): layer 4 at 32 images is 120 tiles of 72 K-tiles each.  With the
// gathered channels cut into ranges the launch has tiles x ranges workgroups of 1 / ranges the K-tiles, and a second, elementwise
// launch (conv_split_epilogue_k).  Needs the caller's scratch (IgemmArgs::split_ws).
// Measured (profiles/r04_run14_split_k.log; per launch, forward = input gradient): the second launch and the partial tiles cost about
// what half the K loop saves.  512 channels (72 K-tiles per tile), 32 images: 22 -> 19 us in 2 or 4 ranges, 28 in 8; 16 images: 21 -> 15;
// 64 images (240 tiles): 24 -> 27.  256 channels (36 K-tiles): 15 -> 21 us at 32 images, 13 -> 16 at 16.  Inside the training step
// (r04_run15_split_k_policy.log) the 3 us per launch at 32 images do not show (4.22 vs 4.23 / 4.26 ms: the timing loop of a lone
// launch hides less of a launch's fixed cost than the stream does); at 16 images the step gains 3.7 % (3.52 -> 3.39 ms).  Hence the
// policy: four ranges for launches of at most 64 tiles that contract 512 channels or more -- layer 4 at up to 16 images per GPU.
int lbc_conv_hdmap_nsplit(const IgemmArgs& a, int mode, int cfg)
{
    (void)mode;
    if (cfg != kLbcCfgHdma + 4 || !a.split_ws || a.pre_scale || a.nphase == 4) return 1;
    const long long opt = lbc_opt(kOptHdmapSplit);
    if (opt == 0) return 1;
    const int nslab = a.C / 64;
    const long long tiles = (long long)lbc_cdiv(a.M, 128) * (a.K / 64), elems = (long long)a.M * a.K;
    auto fits = [&](long long n) { return n > 1 && nslab % n == 0 && elems * n <= a.split_ws_floats && elems * n < (1ll << 31); };
    if (opt > 1) return fits(opt) ? (int)opt : 1;
    if (tiles > 64 || nslab < 8) return 1;
    return fits(4) ? 4 : 1;
}

// Which launches the persistent kernel takes (cfg 1: 256 x 128, 2: 128 x 256, 4: 128 x 64 on four waves)
// the stride-2 transposed launches whose four output-parity phases conv_hdmap_k<.., MODE 2> computes in one tile (same predicate as
// conv_glds.hip's phased form: input gradient of a stride-2 3x3 convolution, ConvTranspose2d forward; plain epilogue)
bool lbc_conv_hdmap_phased(const IgemmArgs& a, int mode)
{
    return mode == 1 && a.nphase == 4 && a.S == 2 && a.ostep == 2 && a.KH == 3 && a.KW == 3 && a.P == 1 && a.H == a.LH && a.W == a.LW &&
           a.OH == 2 * a.LH && a.OW == 2 * a.LW && a.M == a.N * a.LH * a.LW && !a.resid && !a.bnb_y && !a.pre_scale && a.w_bf16 && a.act_bf16 &&
           a.C % 64 == 0 && a.K % 64 == 0 && 128 + a.W + 2 <= 192 - 8 && (long long)a.N * a.OH * a.OW * a.K < (1ll << 31) &&
           (long long)a.N * a.H * a.W * a.C < (1ll << 31) && (long long)a.K * 9 * a.C < (1ll << 31) && !lbc_opt_on(kOptNoGldsPhased);
}

bool lbc_conv_hdmap_eligible(const IgemmArgs& a, int mode, int cfg)
{
    if (a.nphase == 4) return cfg == kLbcCfgHdma + 4 && lbc_conv_hdmap_phased(a, mode);
    if (mode != 0 && mode != 1) return false;
    if (a.pre_scale) return false;                       // no BatchNorm-on-load form (conv_hdma.hip)
    if (a.bnb_y && (mode != 1 || (a.resid != nullptr) != (a.bnb_mask != nullptr))) return false;      // (form 2: no residual; form 4: residual + mask tensor)
    // the halo (BM + 2W + 2 rows) must end at least 8 rows before its LDS buffer does: the last 8-row DMA piece then comes from the
    // zero page as a whole and holds the zero row of the border select
    if (cfg == kLbcCfgHdma + 1) return 256 + 2 * a.W + 2 <= 384 - 8;
    if (cfg == kLbcCfgHdma + 2 || cfg == kLbcCfgHdma + 4) return 128 + 2 * a.W + 2 <= 192 - 8;
    return false;
}

int lbc_conv_hdmap_launch(const IgemmArgs& a, int mode, int cfg, hipStream_t s)
{
    LBC_REQUIRE(lbc_conv_hdmap_eligible(a, mode, cfg), "conv_hdmap: launch not eligible");
    const int bm = cfg == kLbcCfgHdma + 1 ? 256 : 128, bn = cfg == kLbcCfgHdma + 1 ? 128 : (cfg == kLbcCfgHdma + 4 ? 64 : 256);
    LBC_REQUIRE(a.K % bn == 0 && a.C % 64 == 0, "conv_hdmap: shape not tileable");
    const void* zero = nullptr;
    int rc = lbc_zero_page(&zero);
    if (rc) return rc;
    const int ntiles = lbc_cdiv(a.M, bm) * (a.K / bn);
    // one workgroup per CU (LDS; two for the four-wave shape); tiles per workgroup so that a grid of <= `cap` workgroups covers the launch
    const int cap = lbc_opt(kOptHdmaPersistWgs) > 0 ? (int)lbc_opt(kOptHdmaPersistWgs) : (cfg == kLbcCfgHdma + 4 ? 512 : 256);
    if (a.nphase == 4) {       // phased transposed form (MODE 2): persistent, several tiles per workgroup, no K split
        const int tpwp = lbc_cdiv(ntiles, cap);
        return lbc_conv_hdmap_launch_128x64_192(a, 2, zero, ntiles, tpwp, (unsigned)lbc_cdiv(ntiles, tpwp), s, 1, 1);
    }
    const int nsplit = lbc_conv_hdmap_nsplit(a, mode, cfg);
    if (nsplit > 1) {
        rc = lbc_conv_hdmap_launch_128x64_192(a, mode, zero, ntiles, 1, (unsigned)(ntiles * nsplit), s, nsplit, 1);
        if (rc) return rc;
        const dim3 eg((unsigned)lbc_cdiv(a.M, 128), (unsigned)(a.K / 64));
        if (a.bnb_y) hipLaunchKernelGGL(conv_split_epilogue_k<true>, eg, dim3(256), 0, s, a, nsplit);
        else hipLaunchKernelGGL(conv_split_epilogue_k<false>, eg, dim3(256), 0, s, a, nsplit);
        return lbc_check_launch("conv_split_epilogue");
    }
    // In-workgroup K split of the four-wave shape (conv_hdmap_k<.., KG = 2>): a launch of at most one tile per CU would run one wave per
    // SIMD; as eight-wave workgroups of two K-range instances it runs two, on half the K loop each.  LBC_HDMAP_SPLIT=0: never, = n > 1: only the cross-workgroup ranges (tests
    // that compare launches bit for bit); LBC_HDMA_PERSIST_WGS (tests: several tiles per workgroup) keeps the plain form too.
    if (cfg == kLbcCfgHdma + 4 && ntiles <= 256 && (a.C / 64) % 2 == 0 && (lbc_opt(kOptHdmapSplit) < 0 || lbc_opt(kOptHdmapSplit) == 1) && lbc_opt(kOptHdmaPersistWgs) <= 0)
        return lbc_conv_hdmap_launch_128x64_192(a, mode, zero, ntiles, 1, (unsigned)ntiles, s, 1, 2);
    const int tpw = lbc_cdiv(ntiles, cap);
    const unsigned grid = (unsigned)lbc_cdiv(ntiles, tpw);
    if (cfg == kLbcCfgHdma + 1) {
        if (256 + 2 * a.W + 2 <= 320 - 8) return lbc_conv_hdmap_launch_256x128_320(a, mode, zero, ntiles, tpw, grid, s);   // W <= 27: layers 3 / 4
        return lbc_conv_hdmap_launch_256x128_384(a, mode, zero, ntiles, tpw, grid, s);
    }
    if (cfg == kLbcCfgHdma + 4) return lbc_conv_hdmap_launch_128x64_192(a, mode, zero, ntiles, tpw, grid, s, 1, 1);
    return lbc_conv_hdmap_launch_128x256_192(a, mode, zero, ntiles, tpw, grid, s);
}
