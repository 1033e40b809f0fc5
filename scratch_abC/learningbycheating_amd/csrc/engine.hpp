// Native executor of the LbC policy networks (ImagePolicyModelSS / BirdViewPolicyModelSS:
// ResNet-18/34 BasicBlock trunk + velocity late fusion + 3 x (BN, ConvTranspose, ReLU)
// decoder + 4 command branches of (BN, 1x1 conv, spatial softmax)).
// reference topology: bird_view/models/image.py:22-89, birdview.py:34-79, resnet.py:95-159.
//
// The executor owns nothing but a plan: tensor table (names = the reference's
// state_dict keys), per-layer geometry and offsets into ONE caller-provided HBM
// workspace.  forward()/backward() enqueue the whole kernel sequence on the given
// stream with no host synchronisation, so a training step can be graph-captured.
#pragma once
#include <string>
#include <vector>

#include "lbc_common.hpp"
#include "lbc_hip.h"
#include "lbc_kernels.hpp"

namespace lbc {

enum TensorKind { kParam = 0, kBufferF32 = 1, kBufferI64 = 2 };

struct TensorInfo {
    std::string name;
    int kind;
    int ndim;
    int shape[4];
    long long numel;
    void* ptr = nullptr;     // bound device pointer
    float* grad = nullptr;   // bound gradient pointer (params only)
};

struct BN {
    int C = 0;
    int g = -1, b = -1, rm = -1, rv = -1, nbt = -1;          // tensor indices
    size_t scale = 0, shift = 0, mean = 0, invstd = 0;       // workspace offsets (floats)
    size_t cA = 0, cB = 0, cD = 0;
};

struct Conv {   // nn.Conv2d without bias
    int w = -1;
    int Cin = 0, Cout = 0, H = 0, W = 0, k = 0, s = 1, p = 0, OH = 0, OW = 0;
    size_t y = 0;            // raw output (pre-BN)
    size_t wn = 0, wt = 0;   // precision 2: per-step bf16 copies of the weight, [Cout][T][Cin] and transposed [Cin][T][Cout]
};

struct Block {
    Conv c1, c2, ds;
    BN b1, b2, bd;
    bool has_ds = false;
    bool fuse_z1 = true;     // conv2 / wgrad2 / bn1-backward read y1 with bn1(+ReLU) applied on load; z1 is never written
    size_t z1 = 0, out = 0;
};

struct Deconv {   // BN -> ConvTranspose2d(k3,s2,p1,op1) -> ReLU
    BN bn;
    int w = -1, bias = -1;
    int Cin = 0, Cout = 0, H = 0, W = 0;   // input spatial size
    size_t u = 0;                          // relu output [N,2H,2W,Cout]
    size_t wn = 0, wt = 0;                 // precision 2: bf16 copies [Cin][T][Cout] and transposed [Cout][T][Cin]
};

class Net {
public:
    explicit Net(const lbc_net_desc& d);
    ~Net();
    const lbc_net_desc& desc() const { return d_; }
    std::vector<TensorInfo>& tensors() { return t_; }
    size_t workspace_bytes() const { return ws_floats_ * sizeof(float); }
    void set_workspace(void* p) { ws_ = static_cast<float*>(p); }
    int check_bound(bool need_grads) const;

    // image: f32 NCHW in [0,1] (the reference signature) or, with image_u8, uint8 NHWC frames (0..255)
    int forward(int N, int train, const void* image, int image_u8, const float* velocity, const float* command, float* pred_sel,
                float* pred_all, hipStream_t s);
    // stage: -1 = everything; otherwise 0 = head+decoder, 1..4 = layer4..layer1, 5 = stem (call in order)
    int backward(const float* d_sel, const float* d_all, int stage, hipStream_t s);
    static const int kNumStages = 6;
    // Activations of the last training-mode forward as they lie in the workspace (introspection for parity tests: the masks a
    // float64 checker must freeze to differentiate the same piecewise-linear function).  NHWC [max_batch rows used: N][H][W][C];
    // elem: 4 = f32, 2 = bf16 (precision 2), 1 = uint8 (max-pool arg-max taps)
    struct ActInfo { std::string name; size_t offset_bytes; int H, W, C, elem; };
    std::vector<ActInfo> activations() const;
    // what backward() would differentiate: batch size and mode of the last forward, and how many forwards ran before it
    void set_frozen(bool f) { frozen_ = f; derived_valid_ = false; }
    void invalidate_derived() { derived_valid_ = false; }
    int last_batch() const { return lastN_; }
    int last_train() const { return last_train_; }
    long long generation() const { return generation_; }
    // Synchronized BatchNorm for data parallelism: every BatchNorm's batch sums (forward) and gradient sums (backward) are
    // all-reduced through `fn` before they are finalized, so the ranks normalise with the statistics of the global batch.
    // buf: device scratch of >= kSyncFloats floats the callback reduces in place.  fn == nullptr: local BatchNorm.
    static const int kSyncFloats = 1536;     // >= 20 * 65 + 1 (head) and 2 * 640 + 1 (the widest BatchNorm) sums + the batch size
    int set_sync_bn(lbc_allreduce_fn fn, void* ctx, int world, float* buf, int buf_floats);

private:
    int add_tensor(const std::string& name, int kind, std::initializer_list<int> shape);
    size_t alloc(size_t nfloats);
    size_t alloc_act(size_t nelems);
    BN make_bn(const std::string& prefix, int C);
    Conv make_conv(const std::string& name, int Cin, int Cout, int H, int W, int k, int s, int p);
    float* W(size_t off) const { return ws_ + off; }
    float* P(int ti) const { return static_cast<float*>(t_[ti].ptr); }
    float* G(int ti) const { return t_[ti].grad; }

    // pre: BatchNorm(+ReLU) of the producer applied on load.  Eval mode folds the consumer-side BatchNorm into the epilogue:
    // y = relu?(conv * post.scale + post.shift + resid), written to `out` (default: the layer's raw-output buffer).
    int conv_fwd(const Conv& c, const float* x, int N, bool stats, int* rows, hipStream_t s, const BN* pre = nullptr,
                 const BN* post = nullptr, const float* resid = nullptr, bool relu = false, float* out = nullptr, float* stats_buf = nullptr,
                 size_t stats_cap = 0);
    // the arguments of a training-mode forward finalize of `bn` over `rows` partial rows at `part` (one launch of its own, or folded
    // into the consuming bn_apply pass: BnApplyArgs::fold)
    BnFinalizeArgs fin_args(const BN& bn, const float* part, int rows, long long count, bool update_running = true) const;
    // may the consumer of this BatchNorm's coefficients do the finalize itself? (local statistics, few rows: lbc_bn_fold_ok)
    bool can_fold(int rows, int C) const { return !sync_fn_ && rows <= kLbcFinalizeRows && lbc_bn_fold_ok(rows, C); }
    // row cap for a statistics / gradient-sum reduction whose consumer may fold the finalize: only for tensors small enough that
    // ~100 workgroups still stream them at launch-latency cost (8 MB); 0 = the reduction's own policy
    int fold_rows_for(long long pixels, int C) const
    {
        if (sync_fn_ || pixels * C * (act_bf16_ ? 2 : 4) > (8ll << 20)) return 0;
        return lbc_bn_fold_ok(1, C) ? lbc_bn_fold_max_rows(C) : 0;
    }
    int bn_eval_prep(hipStream_t s);
    int conv_wgrad_pre(const Conv& c, const float* x, const BN* pre, const float* dy, int N, hipStream_t s);
    bool dgrad_wt_ = false;  // bf16 operands on f32 tensors (precision 1): input-gradient GEMMs read a per-step transposed copy of the weights
    size_t wt_ = 0;
    bool bf16_ = false;      // precision >= 1: bf16 MFMA operands in the convolution family
    bool act_bf16_ = false;  // precision 2: activations and activation gradients are stored as bf16 in HBM
    int weight_prep(hipStream_t s);
    bool conv_takes_glds(const Conv& c, int N, bool with_prologue = false) const;
    // synced: partial_ rows were all-reduced already by sync_rows() (several BatchNorms finalized from the same sums)
    int bn_finalize(const BN& bn, int rows, long long count, int n_local, int train, hipStream_t s, bool update_running = true,
                    const float* synced = nullptr);
    // SyncBN: part[rows][width] -> one row summed over every rank (in sync_buf_); no-op (returns part) when not enabled
    // local_lo / local_hi (nullable): the halves of this rank's own row are also written there, before the exchange
    // the batch size of this rank travels behind the sums (element `width` of the all-reduced row): finalizes divide by the GLOBAL count
    int sync_rows(const float*& part, int& rows, int width, int n_local, hipStream_t s, float* local_lo = nullptr, float* local_hi = nullptr);
    int bn_bwd_finalize(BnBwdFinalizeArgs f, hipStream_t s);
    lbc_allreduce_fn sync_fn_ = nullptr;
    void* sync_ctx_ = nullptr;
    int sync_world_ = 1;
    float* sync_buf_ = nullptr;
    // reduced_rows > 0: dz is already masked and partial_ holds that many rows of (sum g, sum g * xhat) (fused into the producer)
    int bn_backward(const BN& bn, const float* dz, const float* mask, float* g_out, const float* x, long long pixels,
                    float* dx, int Cout, hipStream_t s, const BN* mask_bn = nullptr, bool join_before_apply = false, int reduced_rows = 0);
    // Backward runs the weight gradients of the residual blocks on an internal side stream, next to the input gradients that
    // consume the same dY (independent work; it fills the chip at small per-GPU batches).  fork: the side stream waits for
    // everything enqueued on s so far; join: s waits for the side stream.  Inactive while the launch profiler is on.
    int fork(hipStream_t s);
    int join(hipStream_t s);
    hipStream_t wstream(hipStream_t s) const { return side_on_ ? side_ : s; }
    void split_scratch(IgemmArgs& a) const;
    hipStream_t side_ = nullptr;
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
    bool side_allowed_ = true, side_on_ = false, side_dirty_ = false;
    int conv_wgrad(const Conv& c, const float* x, const float* dy, int N, hipStream_t s);
    // bnb (+ bnb_y): fuse the reduce pass of that BatchNorm's backward into the epilogue when the kernel can (*fused_rows = partial
    // rows written to partial_, else 0)
    // bnb / bnb_y: fuse the BatchNorm-backward reduce of relu(bnb(bnb_y)) into the epilogue (mask = bnb(bnb_y) > 0); with bnb_mask the mask
    // is that tensor > 0 (the ReLU output of a block: relu(bnb(bnb_y) + identity)) and the launch may carry a residual.  *fused_rows > 0:
    // the launch did it (dx is the masked gradient, partial_ holds that many rows of sums)
    int conv_dgrad(const Conv& c, const float* dy, const float* resid, float* dx, int N, hipStream_t s, const BN* bnb = nullptr,
                   const float* bnb_y = nullptr, int* fused_rows = nullptr, const float* bnb_mask = nullptr);
    int block_backward(Block& b, float*& D, float*& Gbuf, float* E, float* F, hipStream_t s);
    // Stage-deferred weight gradients (bf16 tensors): the 3x3 weight gradients of a stage wait for the stage's last block and leave as
    // one grouped launch per shape (lbc_wgrad_tr_group_launch); until then every dY keeps its own slot of the dy arena.
    struct PendingWgrad { const Conv* c; const float* x; const BN* pre; const float* dy; };
    std::vector<PendingWgrad> pending_;
    bool defer_wgrad_ = false;
    size_t dy_arena_ = 0, dy_arena_floats_ = 0, dy_used_ = 0;
    size_t wg_floats_ = 0;               // size of the split-K slab arena (planned at max_batch)
    float* dy_slot(long long elems);
    int flush_wgrads(int N, hipStream_t s);
    int backward_impl(const float* d_sel, const float* d_all, int stage, hipStream_t s);
    long long generation_ = 0;

    lbc_net_desc d_;
    std::vector<TensorInfo> t_;
    size_t ws_floats_ = 0;
    float* ws_ = nullptr;

    // topology
    int stem_w_ = -1;
    BN stem_bn_;
    int H1_ = 0, W1_ = 0;                  // after stem + pool
    std::vector<Block> blocks_;
    std::vector<int> stage_first_block_;   // index of the first block of layer1..4
    Deconv dec_[3];
    BN head_bn_[4];
    int head_w_[4], head_b_[4], head_px_[4], head_py_[4];
    int HH_ = 0, HW_ = 0;                  // head map size

    // workspace offsets
    size_t xp_ = 0, y0_ = 0, p0_ = 0, idx_ = 0, hcat_ = 0, cmd_ = 0;
    size_t partial_ = 0, partial2_ = 0, wg_partial_ = 0, head_partial_ = 0, head_coef_ = 0, head_stats_ = 0;
    size_t pred_all_ = 0, rowstat_ = 0;
    size_t gD_ = 0, gE_ = 0, gF_ = 0, gG_ = 0, g0_ = 0;
    size_t partial_floats_ = 0, partial2_floats_ = 0;      // capacities of partial_ / partial2_ (checked against a launch's row count before it runs)

    // state of the last forward
    bool frozen_ = false, derived_valid_ = false;     // lbc_net_set_frozen: weight copies / eval-mode affines derived once
    int lastN_ = 0;
    int last_train_ = 0;
    float* bwd_D_ = nullptr;   // running "gradient wrt block output" buffer between stages
    long long bwd_pre_pix_ = 0; int bwd_pre_C_ = 0;   // ... what those rows were summed over (pixels, channels of the consuming block's bn2): checked by the consumer
    int bwd_pre_rows_ = 0;     // > 0: the gradient in bwd_D_ is already masked with its block's ReLU output and partial_ holds that many rows of bn2's backward sums (the producing input gradient did both: IgemmArgs::bnb_mask)
    float* bwd_G_ = nullptr;
};

}  // namespace lbc
