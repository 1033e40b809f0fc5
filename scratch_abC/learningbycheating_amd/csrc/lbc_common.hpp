// Shared declarations for the gfx950 kernels of the LbC sensorimotor hot path.
// All activations are NHWC fp32 in HBM; conv weights are read in the memory
// order of a channels_last torch tensor: Conv2d (O,I,kh,kw) -> [O][kh][kw][I],
// ConvTranspose2d (I,O,kh,kw) -> [I][kh][kw][O].
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LBC_OK 0
#define LBC_EINVAL (-1)
#define LBC_ELAUNCH (-2)
#define LBC_ESTATE (-3)

void lbc_set_error(const char* fmt, ...);
int lbc_check_launch(const char* what);

#define LBC_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            lbc_set_error(__VA_ARGS__);        \
            return LBC_EINVAL;                 \
        }                                      \
    } while (0)

static inline int lbc_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Runtime options (A/B switches, tuning knobs, test hooks).  ONE table: initialised from the LBC_* environment variables when
// the library is loaded and changed afterwards only through lbc_config_set() (include/lbc_hip.h) -- a launch path never reads
// the environment, and every option can be toggled inside one process.  -1 = unset (the built-in policy applies).
enum LbcOpt {
    // (round 5: 50 -> 18.  Every variant that was measured equal or slower is gone with its kernel -- conv_hdma_k, conv_glds_k, the first
    //  bf16 stem, BatchNorm-on-load inside the halo-staged kernels, schedule variants, timing-experiment builds -- and every tuning knob
    //  whose sweep ended is a constant next to its use.  What is left: switches the tests use to reach a kernel at test sizes or to
    //  compare a specialised kernel with the generic one, and two the bench uses.)
    kOptForceCfg = 0,      // LBC_FORCE_CFG: tile policy of the generic convolution (0: 128x64, 1: 128x128, 2: 64x64)
    kOptHaloBlocks,        // LBC_HALO_BLOCKS: cap on the persistent workgroups of the 64-channel kernels (tests: force multi-tile workgroups)
    kOptHeadNoMfma,        // LBC_HEAD_NO_MFMA: 1 = the f32 head kernels in the bf16 mode too (tests)
    kOptNoSideStream,      // LBC_NO_SIDE_STREAM (read when a network is created)
    kOptNoGemm256,         // LBC_NO_GEMM256: 1 = never use the LDS-DMA convolutions (conv_glds.hip, conv_hdmap.hpp, conv_c64p.hip)
    kOptGemm256MinTiles,   // LBC_GEMM256_MIN_TILES: minimum tile count for those kernels (default 96 / 192; tests set 1)
    kOptGemm256Cfg,        // LBC_GEMM256_CFG: pin the tile shape of conv_glds2_k (0: 256x256, 1: 256x128, 2: 128x256, 3: 512x128, 4: 512x64)
    kOptNoBnBwdFuse,       // LBC_NO_BN_BWD_FUSE: 1 = BatchNorm-backward reduce always as its own pass (A/B, tests); 2 = only bn2's (the tensor-masked form of round 5) as its own pass
    kOptNoHdma,            // LBC_NO_HDMA: 1 = never use the halo-staged LDS-DMA convolution (conv_hdmap.hpp / conv_c64p.hip)
    kOptHdmaCfg,           // LBC_HDMA_CFG: pin its tile shape (1: 256x128, 2: 128x256, 3: the 64-channel kernel, 4: 128x64 four waves)
    kOptNoGldsPhased,      // LBC_NO_GLDS_PHASED: 1 = the stride-2 transposed launches keep conv_igemm.hip (tests compare the two)
    kOptHdmaPersistWgs,    // LBC_HDMA_PERSIST_WGS: cap on the persistent workgroups of conv_hdmap_k (default one / two per CU; tests: fewer)
    kOptWgradTr2MinWgs,    // LBC_WGRAD_TR2_MIN_WGS: the stride-2 tap-fused weight gradient takes a launch that yields at least this many workgroups of 16 chunks (default 192; tests: 1 = always, a huge value = never)
    kOptNoC64pPre,         // LBC_NO_C64P_PRE: 1 = forward launches of the 64-channel layer with BatchNorm-on-load stay on conv_halo.hip (tests compare the two)
    kOptNoBnFold,          // LBC_NO_BN_FOLD: 1 = every BatchNorm finalize is its own launch (A/B, tests); default: folded into the consuming elementwise pass where the partial rows are few
    kOptC64pBm,            // LBC_C64P_BM: tile rows of conv_c64p_k: 256 = eight waves, double-buffered halo, one workgroup per CU; 128 = four waves, ring halo, two per CU; unset = policy
    kOptHdmapSplit,        // LBC_HDMAP_SPLIT: split-K of the four-wave persistent convolution for launches with few tiles (IgemmArgs::split_ws): 0 = never, n > 1 = n ranges wherever they divide the slabs (tests, A/B), unset / 1 = policy (lbc_conv_hdmap_nsplit)
    kOptHdmaSmallBelow,    // LBC_HDMA_SMALL_BELOW: launches whose best eight-wave shape has fewer tiles than this take the four-wave 128 x 64 shape instead where it fits (default 160; 0 = never)
    kOptCount
};
long long lbc_opt(LbcOpt o);
static inline bool lbc_opt_on(LbcOpt o) { return lbc_opt(o) == 1; }

// Built-in launch profiler (lbc_util.cpp): when enabled through lbc_profile_enable(1) every launcher
// brackets its kernel with two HIP events on the launch stream and books the kernel's ALGORITHMIC
// flops / HBM bytes under a class name; lbc_profile_report() sums them.  Disabled = zero overhead.
bool lbc_prof_on();
void lbc_prof_begin(const char* name, double flops, double bytes, hipStream_t s);
void lbc_prof_end(hipStream_t s);
void lbc_prof_note(const char* name, double bytes);
struct LbcProfScope {
    hipStream_t s; bool on;
    LbcProfScope(const char* name, double flops, double bytes, hipStream_t st) : s(st), on(lbc_prof_on()) {
        if (on) lbc_prof_begin(name, flops, bytes, s);
    }
    ~LbcProfScope() { if (on) lbc_prof_end(s); }
};

// ---------------------------------------------------------------------------
// Implicit-GEMM convolution (conv_igemm.hip).
//   rows    m = (n, ly, lx) over a lattice of output pixels
//   columns k = output channel
//   depth     = (tap, c) over the gathered tensor's channels
// mode 0 (gather):      iy = oy*S + r - P                 (Conv2d forward,
//                                                          ConvTranspose2d dgrad)
// mode 1 (transposed):  iy = (oy + P - r)/S when divisible (Conv2d dgrad,
//                                                          ConvTranspose2d forward);
//                       for S==2 one launch per output parity phase.
// wmajor 1: w[k][tap][c] (depth-contiguous), wmajor 0: w[c][tap][k].
// ---------------------------------------------------------------------------
struct IgemmArgs {
    const void* x;        // gathered tensor, NHWC [N][H][W][C]; f32, or bf16 when act_bf16
    const void* w;        // weights: f32, or bf16 when w_bf16 (needs bf16 = 1; see lbc_weight_prep)
    void* y;              // NHWC [N][OH][OW][K]; same element type as x
    const float* post_scale;   // [K] or nullptr: per-output-channel affine applied to the accumulator first (eval-mode
    const float* post_shift;   //   BatchNorm folded into the producing convolution): v = acc * post_scale + post_shift
    const float* bias;    // [K] or nullptr
    const void* resid;    // like y (may alias y) or nullptr; added before relu
    float* stats;         // [rows][2][K] per-block (sum, sum of squares) of the stored value, or nullptr
    // fused BatchNorm-on-load of x (per gathered channel): x' = relu?(x*ps[c] + pt[c]); nullptr = identity
    const float* pre_scale;
    const float* pre_shift;
    int pre_relu;
    int N, H, W, C;
    int OH, OW, K;
    int KH, KW, S, P;
    int M;                // rows handled by this launch
    int LH, LW;           // lattice extents; pixel = (ly*ostep + oy0, lx*ostep + ox0)
    int oy0, ox0, ostep;
    int nphase;           // 4: one launch covers the four output-parity phases of a stride-2 transposed launch (oy0/ox0 ignored;
                          //    workgroups [ph * tiles, (ph+1) * tiles) serve phase ph = 2*oy0 + ox0; statistics rows follow); else 1
    int relu;
    int stat_row0;
    int bf16;             // 1: bf16 MFMA operands (f32 accumulation), needs wmajor weights and C % 64 == 0
    int act_bf16;         // 1: x / y / resid are bf16 tensors (requires bf16 = 1)
    int w_bf16;           // 1: w is a bf16 copy of the weights (requires bf16 = 1, depth-contiguous)
    // Fused BatchNorm-backward reduce (input-gradient launches whose output is the gradient wrt relu(bn(bnb_y))): the epilogue
    // stores g = out * (bnb_y * bnb_scale + bnb_shift > 0) and writes the partial rows (sum g, sum g * xhat),
    // xhat = (bnb_y - bnb_mean) * bnb_invstd, to `stats` -- what channel_reduce_k (op 1) would compute in a pass of its own.
    // Only kernels for which lbc_igemm_fuses_bn_bwd() is true honour it.
    const void* bnb_y;    // like y, or nullptr
    const float* bnb_scale;
    const float* bnb_shift;
    const float* bnb_mean;
    const float* bnb_invstd;
    // ... with the ReLU mask given as a tensor (round 5): the output is the gradient wrt relu(bn(bnb_y) + identity) -- the block output of
    // a BasicBlock, resnet.py:51-54 -- so the mask is bnb_mask > 0 (bnb_mask = that block output, like y) instead of bn(bnb_y) > 0, and the
    // launch may carry a residual (the identity-path gradient of the block BEHIND it).  bnb_scale / bnb_shift are not read then.
    // Only kernels for which lbc_igemm_fuses_bn_bwd_masked() is true honour it.
    const void* bnb_mask;
    // Split-K scratch (nullable): launches with few output tiles may cut the gathered channels into ranges, one workgroup per (tile,
    // range), f32 partial tiles [range][M][K] here and a second launch that sums them and does the epilogue (conv_hdmap.hip).
    float* split_ws;
    long long split_ws_floats;
};
// true when the kernel a (cfg, wmajor, mode) launch takes implements IgemmArgs::bnb_*
bool lbc_igemm_fuses_bn_bwd(const IgemmArgs& a, int wmajor, int mode, int cfg);
// ... and IgemmArgs::bnb_mask (+ a residual): conv_hdmap_k<.., EPI 4> and the split-K epilogue
bool lbc_igemm_fuses_bn_bwd_masked(const IgemmArgs& a, int wmajor, int mode, int cfg);

// One launch converts every convolution weight of a network to bf16, in its own layout w[A][T][B] and transposed
// wt[B][T][A] (the depth-contiguous operand of the input-gradient / transposed-convolution GEMMs).
struct WeightPrepItem {
    const float* w;
    void* wn;             // bf16 [A][T][B]
    void* wt;             // bf16 [B][T][A]
    int A, T, B;
    int tile_begin;       // first tile of this tensor in the launch (lbc_weight_prep_tiles() tiles per tensor)
};
// tiles (workgroups) a tensor of [A][T][B] weights takes in lbc_weight_prep: 64 x 64 (a, b) tiles per tap, zero-padded at the edges
static inline int lbc_weight_prep_tiles(int A, int T, int B) { return lbc_cdiv(A, 64) * lbc_cdiv(B, 64) * T; }
struct WeightPrepArgs {
    static const int kMax = 48;
    WeightPrepItem item[kMax];
    int count;
    int tiles;
};
int lbc_weight_prep(const WeightPrepArgs& a, hipStream_t s);

int lbc_igemm_rows(const IgemmArgs& a, int cfg);   // number of M tiles (= stats rows) of a launch
int lbc_igemm_pick(long long M, int K);            // tile configuration 0..2 of conv_igemm.hip from the GEMM extents alone
// tile configuration for a fully described launch: conv_glds.hip's (kLbcCfgGlds + 0..2) when eligible, else lbc_igemm_pick
int lbc_igemm_pick_for(const IgemmArgs& a, int mode);
constexpr int kLbcCfgGlds = 3;
constexpr int kLbcGldsCfgs = 7;
constexpr int kLbcCfgHdma = kLbcCfgGlds + kLbcGldsCfgs;     // conv_hdma.hip (policy): {0: retired, 1: 256x128, 2: 128x256, 3: the 64-channel kernel conv_c64p.hip (C = K = 64)}
constexpr int kLbcHdmaCfgs = 5;                             // ... 4: 128 x 64, four waves, two workgroups per CU (launches with few rows)
int lbc_conv_hdma_pick(const IgemmArgs& a, int mode);
int lbc_conv_hdma_rows(const IgemmArgs& a, int cfg);
int lbc_conv_hdma_launch(const IgemmArgs& a, int mode, int cfg, hipStream_t s);
int lbc_conv_c64p_launch(const IgemmArgs& a, int mode, hipStream_t s);     // conv_c64p.hip: cfg kLbcCfgHdma + 3 (C = K = 64)
int lbc_conv_c64p_rows(const IgemmArgs& a);                                  // statistics rows it writes: one per persistent workgroup
bool lbc_conv_hdmap_phased(const IgemmArgs& a, int mode);                  // conv_hdmap.hip: a stride-2 transposed launch the persistent kernel's MODE 2 takes (all four parity phases per tile)
bool lbc_conv_hdmap_eligible(const IgemmArgs& a, int mode, int cfg);       // conv_hdmap.hip: the persistent kernel takes cfg 1 / 2 / 4
int lbc_conv_hdmap_launch(const IgemmArgs& a, int mode, int cfg, hipStream_t s);
int lbc_conv_hdmap_nsplit(const IgemmArgs& a, int mode, int cfg);          // split-K ranges of that launch (1 = none)
int lbc_conv_glds_pick(const IgemmArgs& a, int mode);      // kLbcCfgGlds + {0: 256x256, 1: 256x128, 2: 128x256, 3: 512x128, 4: 512x64, 5: 256x64 and 6: 128x128 on four waves} or -1
int lbc_conv_glds_rows(const IgemmArgs& a, int cfg);
int lbc_conv_glds_launch(const IgemmArgs& a, int mode, int cfg, hipStream_t s);
// 256 zero bytes in device memory (per device, allocated on first use): source of the zero padding of LDS-DMA staging
int lbc_zero_page(const void** p);
int lbc_igemm_launch(const IgemmArgs& a, int wmajor, int mode, int cfg, hipStream_t s);
int lbc_weight_transpose(const float* w, float* wt, int A, int T, int B, hipStream_t s);   // w[A][T][B] -> wt[B][T][A]
// 3x3 / stride-1 / pad-1, C = K = 64 launches on bf16 tensors + bf16 weight copies take the halo-staged,
// weight-stationary kernel (conv_halo.hip); statistics rows are per 128-pixel tile like the 128-row igemm tiles
bool lbc_conv3x3_halo_eligible(const IgemmArgs& a, int mode);
int lbc_conv3x3_halo_launch(const IgemmArgs& a, int mode, hipStream_t s);

// Weight-gradient GEMM (conv_wgrad.hip):
//   out[p][tap][q] = sum_m  P[m][p] * Q[gather(m, tap)][q]
// P rows m = (n, oy, ox) dense NHWC [N][OH][OW][CP]; Q NHWC [N][H][W][CQ] gathered at
// (oy*S + r - P, ox*S + s - P).  Split over m into `nsplit` partial slabs which
// lbc_splitk_reduce sums in a fixed order (deterministic).
struct WgradArgs {
    const void* p;        // f32, or bf16 when act_bf16
    const void* q;
    float* partial;       // [nsplit][CP][T][CQ]
    // optional fused transforms on load (BatchNorm apply [+relu]) per channel
    const float* p_scale;
    const float* p_shift;
    const float* q_scale;
    const float* q_shift;
    int q_relu;
    int N, OH, OW, CP;
    int H, W, CQ;
    int KH, KW, S, P;
    int nsplit;
    int bf16;             // 1: bf16 MFMA operands (f32 accumulation)
    int act_bf16;         // 1: p / q are bf16 tensors (requires bf16 = 1)
};
int lbc_wgrad_pick_split(const WgradArgs& a);      // needs bf16 / act_bf16 set: the tap-fused kernel has its own split policy
int lbc_wgrad_launch(const WgradArgs& a, hipStream_t s);
// 3x3 / stride-1 weight gradients on bf16 tensors: all nine taps per workgroup, transpose reads (conv_wgrad_tr.hip)
bool lbc_wgrad_tr_eligible(const WgradArgs& a);
int lbc_wgrad_tr_pick_split(const WgradArgs& a);
int lbc_wgrad_tr_launch(const WgradArgs& a, hipStream_t s);
// The same kernel over n same-shaped convolutions in ONE launch (a ResNet stage's 3x3 convolutions: 6 / 7 / 11 / 5 of them).  A launch
// wants ~512 workgroups whatever it computes and every workgroup leaves a 64 x 9 x 64 f32 partial tile: one launch per convolution
// writes (and splitk_reduce re-reads) 75 MB of slabs per convolution; the grouped launch splits the pixel range n times less.
// `a` carries the geometry, the transform flags (q_scale != nullptr: on) and nsplit (lbc_wgrad_tr_group_split); the tensors come from g.
constexpr int kLbcWgradGroupMax = 12;
struct WgradGroup {
    int n;
    int linear_order;       // 1: workgroup id = logical id (tests / A/B scripts set it through the C ABI); 0: XCD-major logical order
    const void* p[kLbcWgradGroupMax];
    const void* q[kLbcWgradGroupMax];
    const float* q_scale[kLbcWgradGroupMax];
    const float* q_shift[kLbcWgradGroupMax];
    float* out[kLbcWgradGroupMax];      // slab 0 of member i (slab stride CP * 9 * CQ floats); the gradient itself when nsplit == 1
};
int lbc_wgrad_tr_group_split(const WgradArgs& a, int n);
int lbc_wgrad_tr_group_launch(const WgradArgs& a, const WgradGroup& g, hipStream_t s);
// out[i] = sum over the nsplit slabs at partial + i * nsplit * count, i < n (one launch)
int lbc_splitk_reduce_group(const float* partial, int nsplit, long long count, int n, float* const* out, hipStream_t s);
// 3x3 / stride-2 weight gradients (and the transposed convolutions') on bf16 tensors: all nine taps per workgroup over a ring of
// high-resolution rows that advances 128 rows per 32 low-resolution pixels (conv_wgrad_tr2.hip)
bool lbc_wgrad_tr2_eligible(const WgradArgs& a);
int lbc_wgrad_tr2_pick_split(const WgradArgs& a);
int lbc_wgrad_tr2_launch(const WgradArgs& a, hipStream_t s);
int lbc_splitk_reduce(const float* partial, int nsplit, long long count, float* out, float beta, hipStream_t s);
