// Internal launcher interfaces of the HBM-bound kernels (bn.hip, pool.hip, head.hip,
// loss.hip, adam.hip, stem.hip).  All tensors NHWC fp32 unless stated otherwise.
#pragma once
#include "lbc_common.hpp"

// ---- BatchNorm forward -----------------------------------------------------------
// the finalize kernels sum up to this many partial rows themselves; beyond it lbc_partial_reduce runs first
constexpr int kLbcFinalizeRows = 1024;
struct BnFinalizeArgs {
    const float* partial;        // [rows][2][C] (sum, sum^2); unused in eval
    int rows, C;
    long long count;             // N*H*W
    // SyncBN (nullable): *nsum = the batch size summed over all ranks (one float behind the all-reduced sums), n_local = this
    // rank's batch: the statistics are over count / n_local * *nsum elements -- ranks may run different batch sizes
    const float* nsum; int n_local;
    const float* gamma;          // nullable (=1)
    const float* beta;           // nullable (=0)
    float* running_mean;         // train: updated (nullable); eval: read
    float* running_var;
    long long* num_batches_tracked;   // nullable
    // train: further BatchNorms that see the SAME tensor (the waypoint head's four branches, image.py:54-60): their running
    // statistics and counters get the same update in this launch (nullable entries end the list)
    float* more_running_mean[3]; float* more_running_var[3]; long long* more_num_batches_tracked[3];
    float momentum, eps;
    int train;
    float* scale;                // out: gamma*invstd
    float* shift;                // out: beta - mean*scale
    float* save_mean;            // out (nullable)
    float* save_invstd;
};
// Eval mode: scale / shift / mean / invstd of every BatchNorm of a network from its running statistics, one launch.
struct BnEvalItem {
    const float* gamma; const float* beta; const float* running_mean; const float* running_var;
    float* scale; float* shift; float* mean; float* invstd;
    int C;
};
struct BnEvalArgs {
    static const int kMax = 48;
    BnEvalItem item[kMax];
    int count;
    float eps;
};
int lbc_bn_eval_prep(const BnEvalArgs& a, hipStream_t s);
// copy_lo / copy_hi (nullable, out_rows == 1 only): the first / second half of the output row is also written there
// tail >= 0 (out_rows == 1 only): out[cols] = tail (SyncBN: this rank's batch size travels behind the sums)
int lbc_partial_reduce(const float* in, int rows, int cols, float* out, int out_rows, hipStream_t s, float* copy_lo = nullptr,
                       float* copy_hi = nullptr, float tail = -1.f);
int lbc_bn_finalize(const BnFinalizeArgs& a, hipStream_t s);

struct BnApplyArgs {
    const void* x; void* y;      // activations: f32 or bf16 (act_bf16)
    long long pixels; int C;
    const float* scale; const float* shift;
    const void* resid;           // nullable
    const float* rscale;         // nullable: residual is itself BatchNorm'ed (downsample path)
    const float* rshift;
    int relu;
    int act_bf16;
    // Folded finalize (training mode, few partial rows: lbc_bn_fold_ok): the launch does the work of lbc_bn_finalize(fin) itself --
    // every workgroup re-derives scale / shift of all C channels from the partial rows (a few KB from L2), workgroup 0 also writes
    // what the finalize kernel writes (scale, shift, saved statistics, running statistics, counter).  One 8-us launch less per
    // BatchNorm where launches, not bytes, are the cost (the per-GPU batch of the 8-GPU run).  rfold: same for the residual's BatchNorm.
    int fold, rfold;
    BnFinalizeArgs fin, rfin;
};
int lbc_bn_apply(const BnApplyArgs& a, hipStream_t s);
// may a consumer fold the finalize of `rows` partial rows of a C-channel BatchNorm? (LBC_NO_BN_FOLD=1: never)
bool lbc_bn_fold_ok(int rows, int C);
// the largest row count lbc_bn_fold_ok accepts for C channels (producers that choose their own row count aim below it)
int lbc_bn_fold_max_rows(int C);

// ---- per-channel reductions --------------------------------------------------------
struct ChanReduceArgs {
    const void* x;               // op0: tensor to take statistics of; op1: pre-BN activation (nullable)
    const void* dz;              // op1: upstream gradient
    const void* mask;            // op1: ReLU mask source (g = dz where mask > 0), nullable
    const float* mask_scale;     // op1: optional per-channel affine applied to the mask source first
    const float* mask_shift;     //      (mask = pre-BN activation, affine = that BN: relu(bn(y)) > 0)
    void* g_out;                 // op1: optional store of the masked gradient (may alias dz)
    const float* mean;           // op1: nullable
    const float* invstd;
    float* partial;              // [rows][2][C]
    long long pixels; int C;
    long long pix_per_block;     // filled by the launcher
    int act_bf16;
    int max_rows;                // > 0: at most this many partial rows (the consumer folds the finalize); 0: the launcher's own policy
};
int lbc_chan_reduce_rows(long long pixels, int C, int max_rows = 0);
int lbc_chan_reduce(ChanReduceArgs a, int op, hipStream_t s);

struct BnBwdFinalizeArgs {
    const float* partial; int rows, C; long long count;
    const float* nsum; int n_local;       // SyncBN (nullable): as BnFinalizeArgs
    const float* gamma; const float* mean; const float* invstd;
    int train;
    float* dgamma; float* dbeta;          // nullable
    float* coefA; float* coefB; float* coefD;   // nullable (all or none): A = gamma*invstd, k1, k2
};
int lbc_bn_bwd_finalize(const BnBwdFinalizeArgs& a, hipStream_t s);

struct BnBwdApplyArgs {
    const void* g; const void* mask; const void* x;
    const float* coefA; const float* coefB; const float* coefD;   // A, k1, k2
    const float* mean; const float* invstd;
    void* dx;                    // [pixels][Cout]
    long long pixels; int C, Cout;
    int accum;                   // dx += ...
    int act_bf16;
    // folded finalize (as BnApplyArgs::fold; plain form only: no mask, no accumulation): the launch does lbc_bn_bwd_finalize(fin)
    int fold;
    BnBwdFinalizeArgs fin;
};
int lbc_bn_bwd_apply(const BnBwdApplyArgs& a, hipStream_t s);

int lbc_copy_f32(const float* src, float* dst, long long n, hipStream_t s);     // small device-to-device copies as a kernel
int lbc_concat_velocity(const void* t, const float* vel, void* h, int N, int hw, int Ct, int Cv, int act_bf16, hipStream_t s);

// ---- stem: input preparation, 7x7/2 convolution, BN+ReLU+maxpool -------------------
// prep: NCHW fp32 image -> (optionally ImageNet-normalised) NHWC fp32 with a 3-pixel
// zero border: xp[N][H+6][W+6][C]
struct NormConst { float mean[8]; float stdv[8]; int enabled; };
int lbc_prep_input(const float* img_nchw, void* xp, int xp_bf16, int N, int C, int H, int W, const NormConst& nc, hipStream_t s);
int lbc_prep_input_u8(const unsigned char* img_nhwc, void* xp, int xp_bf16, int N, int C, int H, int W, const NormConst& nc, hipStream_t s);
struct StemArgs {
    const void* xp;              // [N][H+6][W+6][Cin] zero-bordered image: f32, or bf16 (xp_bf16; the bf16 kernels)
    int xp_bf16;
    const float* w;              // [64][7][7][Cin]
    void* y;                     // [N][H/2][W/2][64] f32 or bf16 (act_bf16)
    float* stats;                // [rows][2][64] or nullptr
    int N, H, W, Cin;
    int act_bf16;
    int bf16;                    // 1: bf16 MFMA operands (f32 accumulation)
};
int lbc_stem_rows(const StemArgs& a);
int lbc_stem_fwd(const StemArgs& a, hipStream_t s);
struct StemWgradArgs {
    const void* xp; int xp_bf16; const void* dy; float* partial;   // partial [nsplit][64][7][7*Cin]; dy f32 or bf16
    int N, H, W, Cin, nsplit;
    int act_bf16;
    int bf16;                    // 1: bf16 MFMA operands (f32 accumulation)
    // Fused BatchNorm-backward apply (nullable; honoured when lbc_stem_wgrad_fuses_bn_bwd()): dy holds the masked gradient g wrt
    // bn1's output and the kernel forms dy' = A (g - k1 - xhat k2), xhat = (bn_y - mean) invstd, as it stages dy
    const void* bn_y;            // [N][H/2][W/2][64] pre-BN stem output (same element type as dy)
    const float* bn_coefA; const float* bn_coefB; const float* bn_coefD; const float* bn_mean; const float* bn_invstd;
};
bool lbc_stem_wgrad_fuses_bn_bwd(int Cin, int bf16);
int lbc_stem_wgrad_split(int N, int H, int W, int Cin, int bf16);   // partial slabs of the kernel a (Cin, bf16) launch takes
int lbc_stem_wgrad(const StemWgradArgs& a, hipStream_t s);

struct PoolFwdArgs {
    const void* y;               // [N][H][W][C] pre-BN stem output
    const float* scale; const float* shift;
    void* p;                     // [N][H/2][W/2][C]
    unsigned char* idx;          // [N][H/2][W/2][C] arg-max tap (0..8), nullable
    int N, H, W, C;
    int act_bf16;
};
int lbc_bn_relu_maxpool_fwd(const PoolFwdArgs& a, hipStream_t s);
struct PoolBwdArgs {
    const void* dp;              // [N][H/2][W/2][C]
    const unsigned char* idx;
    const void* y;               // [N][H][W][C]
    const float* scale; const float* shift;   // forward BN affine (ReLU mask)
    const float* mean; const float* invstd;
    void* g;                     // [N][H][W][C] masked gradient wrt the BN output
    float* partial;              // [rows][2][C]
    int N, H, W, C;
    long long pix_per_block;
    int act_bf16;
};
int lbc_pool_bwd_rows(int N, int H, int W, int C);
int lbc_maxpool_relu_bwd_reduce(PoolBwdArgs a, hipStream_t s);

// ---- waypoint head: 4 x (BN64 -> 1x1 conv 64->5 -> spatial softmax) + branch select --
struct HeadArgs {
    const void* h;               // [N][HW][64] decoder output, f32 or bf16 (act_bf16)
    int act_bf16;
    const float* mean[4];        // per branch BatchNorm statistics [64] (train: all four point at the batch stats)
    const float* invstd[4];
    const float* gamma[4];       // [64]
    const float* beta[4];
    const float* w[4];           // [5][64]
    const float* bias[4];        // [5]
    const float* cmd;            // [N][4] one-hot command
    const float* pos_x[4];       // SpatialSoftmax buffers of each branch, [OH*OW]
    const float* pos_y[4];
    float* pred_all;             // [N][4][5][2]
    float* pred_sel;             // [N][5][2] (nullable)
    float* rowstat;              // [N][4][5][2] (max, sum exp) saved for backward (nullable)
    int N, OH, OW;               // softmax map is OH x OW (pos_x over OW, pos_y over OH)
    float* scratch;              // >= N * 16 * 20 * 4 floats: per-slice soft-argmax partials of small-batch launches (nullable)
    int nslice;                  // set by lbc_head_fwd
    int wsplit;                  // set by the launchers (bf16 activations): folded weights as a bf16 high + low pair (see fold_branch)
};
int lbc_head_fwd(const HeadArgs& a, hipStream_t s);
struct HeadBwdArgs {
    HeadArgs f;
    const float* d_all;          // [N][4][5][2] nullable
    const float* d_sel;          // [N][5][2] nullable
    float* s_partial;            // [lbc_head_bwd_rows][20*65] : per (branch,step): sum dlogit*h[c] (64) and sum dlogit
    void* dh;                    // [N][HW][64] (same element type as h)
    const float* chan_coef;      // pass 2: [2][64] per-channel coefficients (see head.hip)
};
int lbc_head_bwd_rows(const HeadArgs& f);      // rows of s_partial that lbc_head_bwd_reduce writes (N, or N x slices on the MFMA path)
int lbc_head_bwd_max_rows(int max_batch);      // bound of the above over N <= max_batch
int lbc_head_bwd_reduce(const HeadBwdArgs& a, hipStream_t s);
struct HeadBwdFinalizeArgs {
    const float* s_partial; int rows; long long count;   // count = N*HW
    const float* nsum; int n_local;                      // SyncBN (nullable): as BnFinalizeArgs
    const float* gamma[4]; const float* beta[4]; const float* w[4];
    const float* mean; const float* invstd;              // shared batch statistics [64]
    float* dgamma[4]; float* dbeta[4]; float* dw[4]; float* dbias[4];
    float* chan_coef;            // [2][64]: invstd*k1, invstd*k2
    int coef_only;               // 1: write chan_coef only (SyncBN: second pass over the all-reduced sums)
};
int lbc_head_bwd_finalize(const HeadBwdFinalizeArgs& a, hipStream_t s);
int lbc_head_bwd_apply(const HeadBwdArgs& a, hipStream_t s);

// ---- device-side input pipeline (data.hip) ----------------------------------------------------------------
enum { kAugBlur = 0, kAugNoise = 1, kAugCoarseDropout = 2, kAugDropout = 3, kAugAdd = 4, kAugMultiply = 5, kAugContrast = 6 };
struct WarpParams {              // one per image; layout = lbc_warp_params of include/lbc_hip.h
    double im[6];                // the inverted 2 x 3 matrix: source (X, Y) = (im0 x + im1 y + im2, im3 x + im4 y + im5)
    int y0, x0;                  // origin of the output window in the warped image
};
int lbc_warp_crop_u8(const unsigned char* src, unsigned char* dst, const WarpParams* params_dev, int N, int SH, int SW, int C, int H, int W, hipStream_t s);
struct AugParams {               // one per image; layout = lbc_aug_params of include/lbc_hip.h
    int order[8];                // operator ids in application order (n_ops valid entries)
    int n_ops;
    int blur_pos;                // index of kAugBlur in order[], or n_ops when the sequence has no blur
    unsigned seed;               // per-image stream of the per-pixel hash
    float blur_sigma;
    float noise_scale; int noise_pc;
    float cd_p; int cd_h, cd_w, cd_pc;
    float do_p; int do_pc;
    float add_v[3], mul_v[3], con_a[3];
};
int lbc_crop_u8(const unsigned char* src, unsigned char* dst, int N, int SH, int SW, int C, int y0, int x0, int H, int W, hipStream_t s);
int lbc_augment_u8(unsigned char* img, const AugParams* params_dev, float* tmp, int N, int H, int W, int any_blur, hipStream_t s);

// ---- losses (phase 0 / phase 1 / bird-view) -------------------------------------------
struct LossArgs {
    const float* pred;           // student output, normalised [-1,1]
    const float* target;         // teacher output / targets
    float* loss_per_sample;      // [N]
    float* dpred;                // gradient of mean(loss) * grad_scale wrt pred
    int N, R;                    // R = rows per sample (5 or 20), each row = (x, y)
    float grad_scale;            // 1/N (times 1/world_size under data parallelism)
    // camera model (reference training/train_image_phase{0,1}.py CoordConverter)
    float w, h, fov, world_y, fixed_offset, pixels_per_meter, crop_size;
};
int lbc_loss_phase1(const LossArgs& a, hipStream_t s);   // differentiable unprojection + L1 in map space
int lbc_loss_phase0(const LossArgs& a, hipStream_t s);   // teacher map -> image projection (clip) + L1 in image space
int lbc_loss_l1(const LossArgs& a, float target_scale, float target_shift, hipStream_t s);   // bird-view BC loss
int lbc_phase2_weight_launch(const LossArgs& a, hipStream_t s);   // DAgger resampling weights (selected branch, map space)

// ---- Adam (multi-tensor) ------------------------------------------------------------------
struct AdamChunk { float* p; const float* g; float* m; float* v; int n; int pad; };
int lbc_adam_launch(const AdamChunk* chunks_dev, int nchunks, double lr, double beta1, double beta2, double eps,
                    double weight_decay, int step, hipStream_t s);
