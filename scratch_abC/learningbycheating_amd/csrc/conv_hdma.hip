// 3x3 / stride-1 / pad-1 convolution (forward and input gradient) on bf16 tensors for gfx950 with the ACTIVATION HALO staged
// once per 64-channel slab: the third member of the LDS-DMA family (conv_glds.hip explains the family).
// reference arithmetic: BasicBlock conv1 / conv2, bird_view/models/resnet.py:15-22,38-54, and their autograd.
//
// Why: conv_glds2_k streams the activation tile of every (tap, slab) K-tile separately -- nine shifted copies of the same
// rows -- and its timing experiments put ~20 us of a 72 us layer-3 launch on the latency of that stream.
// Here a workgroup stages, per 64-channel slab, the BM + 2W + 2 input rows its BM output pixels can touch ONCE (LDS-DMA, two
// buffers: slab c + 1 lands while slab c is multiplied); the nine taps are row offsets of the fragment reads, exactly as in
// conv_halo.hip.  Only the weights still stream per (tap, slab): a ring of NBUFB tiles of BN x 64, as in conv_glds2_k.
//   * LDS rows are 128 bytes, 16-byte slot XOR-ed with (row >> 1) & 7 (on the DMA source and on the read) -> conflict-free
//     ds_read_b128; a tap changes the row, hence the XOR term: 2 VALU per fragment read (one v_xor, one v_lshl_add).
//   * Image borders: a lane whose tap leaves the image reads a 128-byte ZERO ROW instead (the last row of the halo buffer, which
//     lies past the halo and is filled from the zero page; one select on the row base per (tap, 32-row block)) -- no masking
//     of the fragments themselves.
// The kernels: conv_hdmap.hpp (persistent; one translation unit per tile shape) and conv_c64p.hip (the 64-channel layer, weights in
// registers).  This file is their launch POLICY.  (Round 2's one-tile-per-workgroup kernel conv_hdma_k, its BatchNorm-on-load and
// early-read variants and its timing-experiment builds were measured against the persistent form in rounds 3-4 and lost every time:
// removed in round 5 together with their switches.)
//   * Synchronisation as conv_glds2_k with 64-channel K-tiles: one barrier per K-tile in front of its last depth step; the
//     wave's own weight pieces of the next K-tile are waited for with a counted vmcnt (halo pieces are always OLDER in the
//     wave's DMA queue than the first weight tile of their slab: they are issued in the first 9 - NBUFB taps of the
//     previous slab, that weight tile after them -- so the same wait covers them).
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include "conv_lds_dma.hpp"


namespace {

struct HdmaCfg { int bm, bn, hrmax; };
// cfg ids kLbcCfgHdma + i.  0 (256 x 256) existed in the one-tile-per-workgroup kernel only: retired, the id stays reserved.
// 3: conv_c64p.hip; 4: four waves, two workgroups per CU.  (Round 5 also tried eight waves on 128 x 128 tiles, 124 KB of LDS, for
// launches whose 256 x 128 tiling leaves half the CUs idle and whose image rows do not fit the four-wave halo -- layer 2 at 32 images per
// GPU, 240 instead of 120 workgroups: 4.24 vs 4.22 ms per step, profiles/r05_call3_*: not kept.)
const HdmaCfg kHdmaCfg[kLbcHdmaCfgs] = {{0, 0, 0}, {256, 128, 384}, {128, 256, 192}, {256, 64, 456}, {128, 64, 192}};
constexpr long long kHdmaSmallMinTiles = 48;      // fill threshold of the four-wave shape (r03_run13: below it the 64 x 64 register-staged tiles win)

}  // namespace

// Tile configuration for a launch, or -1 when the launch keeps conv_glds.hip / conv_igemm.hip.
int lbc_conv_hdma_pick(const IgemmArgs& a, int mode)
{
    if (lbc_opt_on(kOptNoHdma) || lbc_opt_on(kOptNoGemm256)) return -1;
    if (a.nphase == 4) {
        // Stride-2 transposed launches (round 5): the four-wave persistent kernel with the 2 x 2-neighbourhood halo and one accumulator set
        // per output-parity phase (conv_hdmap_k<.., MODE 2>), from the same fill threshold as its stride-1 use; a pinned shape of the
        // per-tap kernel (LBC_GEMM256_CFG: its tests) keeps the launch there
        if (!lbc_conv_hdmap_phased(a, mode) || lbc_opt(kOptGemm256Cfg) >= 0) return -1;
        const long long forcedp = lbc_opt(kOptHdmaCfg);
        if (forcedp >= 0 && forcedp != 4) return -1;
        const long long tiles = (long long)lbc_cdiv(a.M, 128) * (a.K / 64);
        const long long small_fill = lbc_opt(kOptGemm256MinTiles) > 0 ? lbc_opt(kOptGemm256MinTiles) : kHdmaSmallMinTiles;
        return tiles >= small_fill ? kLbcCfgHdma + 4 : -1;
    }
    if (!(a.w_bf16 && a.act_bf16) || a.ostep != 1 || a.nphase > 1 || a.oy0 || a.ox0) return -1;
    // BatchNorm-on-load: only the 64-channel layer's kernel has it (conv_c64p_k<0, 0, true>: an in-place transform of the landed halo,
    // once per tile).  Inside conv_hdmap_k it was built twice (round 2 non-persistent, round 4 persistent) and measured slower than the
    // separate bn_apply pass both times (profiles/r05_call1_hdmap_pre_land_or_kill.txt): those launches keep conv_igemm.hip
    const bool c64 = a.C == 64 && a.K == 64;
    if (a.pre_scale && (mode != 0 || !c64 || lbc_opt_on(kOptNoC64pPre) || a.resid != nullptr)) return -1;
    if (a.KH != 3 || a.KW != 3 || a.P != 1 || a.S != 1 || a.C % 64 || (mode != 0 && mode != 1)) return -1;
    if (a.H != a.OH || a.W != a.OW || a.M != a.N * a.H * a.W || (long long)a.N * a.H * a.W * a.C >= (1ll << 31)) return -1;
    if ((long long)a.K * 9 * a.C >= (1ll << 31)) return -1;
    // one workgroup per CU; worth it from about 96 tiles (measured at 120 tiles = layer 2 at batch 32: 0.020 ms against 0.027 ms for
    // the 64 x 64 register-staged tiles; at 60 tiles = layer 3 at batch 32 it loses, 0.029 vs 0.027)
    const long long fill = lbc_opt(kOptGemm256MinTiles) > 0 ? lbc_opt(kOptGemm256MinTiles) : 96;
    const long long forced = lbc_opt(kOptHdmaCfg);          // tests / tuning: pin one shape
    if (c64) {                                              // the 64-channel layer: conv_c64p.hip (persistent, weights in registers)
        if (forced >= 0 && forced != 3) return -1;
        if (256 + 2 * a.W + 2 >= kHdmaCfg[3].hrmax || lbc_cdiv(a.M, 256) < fill) return -1;
        return kLbcCfgHdma + 3;
    }
    int best = -1;
    long long best_tiles = 0;
    double best_score = 0.0;
    for (int i = 1; i < 3; ++i) {
        const HdmaCfg& c = kHdmaCfg[i];
        if (a.K % c.bn) continue;
        if (forced >= 0 && forced != i) continue;
        if (!lbc_conv_hdmap_eligible(a, mode, kLbcCfgHdma + i)) continue;     // (the halo of a tile + one zero piece must fit its LDS buffer)
        const long long tiles = (long long)lbc_cdiv(a.M, c.bm) * (a.K / c.bn);
        if (tiles < fill) continue;
        const double score = (double)tiles / (double)(((tiles + 255) / 256) * 256);
        if (score > best_score) { best_score = score; best = i; best_tiles = tiles; }
    }
    // An eight-wave launch that leaves half the CUs idle runs faster as four-wave 128 x 64 tiles, two workgroups per CU and four times the
    // workgroups, where that shape's 184-row halo holds the image rows (layers 3 / 4): 120 tiles = layer 3 at 64 images 29 -> 22 us per
    // launch (the step 7.20 -> 6.89 ms on that box), layer 4 at 128 images 50 -> 38 us (10.34 -> 10.14 ms); at 240 tiles (layer 3 at 128
    // images) it loses, 35 -> 41 us (profiles/r04_run16_small_tiles_at_120.log).  Nothing in between was measured: the threshold sits at
    // 160 tiles (62 % of the CUs).  LBC_HDMA_SMALL_BELOW=0: never.
    const long long below = lbc_opt(kOptHdmaSmallBelow) >= 0 ? lbc_opt(kOptHdmaSmallBelow) : 160;
    // (not under LBC_GEMM256_MIN_TILES: the tests' switch that sends small launches to the eight-wave shapes keeps its meaning)
    const bool prefer_small = best >= 0 && forced < 0 && best_tiles < below && a.K % 64 == 0 && lbc_opt(kOptGemm256MinTiles) <= 0 &&
                              lbc_conv_hdmap_eligible(a, mode, kLbcCfgHdma + 4);
    if (best >= 0 && !prefer_small) return kLbcCfgHdma + best;
    // Few rows (the per-GPU load of the 8-GPU run: layer 3 / 4 at 32 images have 7680 / 1920 output pixels): 128 x 64 tiles, four waves,
    // two workgroups per CU (conv_hdmap.hpp) instead of the 64 x 64 register-staged tiles of conv_igemm.hip (31 us per 9-GFLOP launch)
    if ((forced < 0 || forced == 4) && a.K % 64 == 0 && lbc_conv_hdmap_eligible(a, mode, kLbcCfgHdma + 4)) {
        const long long tiles = (long long)lbc_cdiv(a.M, 128) * (a.K / 64);
        // (LBC_GEMM256_MIN_TILES -- which tests set to 1 -- applies here too)
        const long long small_fill = lbc_opt(kOptGemm256MinTiles) > 0 ? lbc_opt(kOptGemm256MinTiles) : kHdmaSmallMinTiles;
        if (tiles >= small_fill) return kLbcCfgHdma + 4;
    }
    return -1;
}

int lbc_conv_hdma_rows(const IgemmArgs& a, int cfg)
{
    if (cfg == kLbcCfgHdma + 3) return lbc_conv_c64p_rows(a);         // one row per persistent workgroup
    return lbc_cdiv(a.M, kHdmaCfg[cfg - kLbcCfgHdma].bm);
}

int lbc_conv_hdma_launch(const IgemmArgs& a, int mode, int cfg, hipStream_t s)
{
    LBC_REQUIRE(cfg > kLbcCfgHdma && cfg < kLbcCfgHdma + kLbcHdmaCfgs, "conv_hdma: bad cfg %d", cfg);
    const HdmaCfg c = kHdmaCfg[cfg - kLbcCfgHdma];
    LBC_REQUIRE(a.K % c.bn == 0 && a.C % 64 == 0 && c.bm + (a.nphase == 4 ? 1 : 2) * a.W + 2 < c.hrmax, "conv_hdma: shape not tileable");
    if (cfg == kLbcCfgHdma + 3) return lbc_conv_c64p_launch(a, mode, s);
    return lbc_conv_hdmap_launch(a, mode, cfg, s);
}
