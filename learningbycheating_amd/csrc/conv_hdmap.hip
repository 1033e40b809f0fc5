// Launch policy of the persistent halo-staged convolution (kernel: conv_hdmap.hpp; one translation unit per tile shape so that
// the 15 instantiations compile in parallel).
#include "lbc_common.hpp"

int lbc_conv_hdmap_launch_256x128_320(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, unsigned grid, hipStream_t s);
int lbc_conv_hdmap_launch_256x128_384(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, unsigned grid, hipStream_t s);
int lbc_conv_hdmap_launch_128x256_192(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, unsigned grid, hipStream_t s);
int lbc_conv_hdmap_launch_128x64_192(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, unsigned grid, hipStream_t s);

// Persistent form of conv_hdma.hip's cfg 1 (256 x 128) and cfg 2 (128 x 256); false = the launch keeps conv_hdma_k
// (BatchNorm-on-load, the 256 x 256 test shape, LBC_NO_HDMA_PERSIST=1).
bool lbc_conv_hdmap_eligible(const IgemmArgs& a, int mode, int cfg)
{
    if (lbc_opt_on(kOptNoHdmaPersist) || a.pre_scale || (mode != 0 && mode != 1)) return false;
    if (a.bnb_y && (mode != 1 || a.resid)) return false;
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
    const int tpw = lbc_cdiv(ntiles, cap);
    const unsigned grid = (unsigned)lbc_cdiv(ntiles, tpw);
    if (cfg == kLbcCfgHdma + 1) {
        if (256 + 2 * a.W + 2 <= 320 - 8) return lbc_conv_hdmap_launch_256x128_320(a, mode, zero, ntiles, tpw, grid, s);   // W <= 30: layers 3 / 4
        return lbc_conv_hdmap_launch_256x128_384(a, mode, zero, ntiles, tpw, grid, s);
    }
    if (cfg == kLbcCfgHdma + 4) return lbc_conv_hdmap_launch_128x64_192(a, mode, zero, ntiles, tpw, grid, s);
    return lbc_conv_hdmap_launch_128x256_192(a, mode, zero, ntiles, tpw, grid, s);
}
