/* lbc_hip.h -- C ABI of the MI355X-native LbC sensorimotor hot path.
 *
 * Plain pointers and sizes only (no torch types).  All device pointers are HBM
 * addresses on the current device; `stream` is a hipStream_t passed as void*.
 * Activations are NHWC fp32.  Convolution weights are read in the memory order
 * of a channels_last tensor with the reference's logical shapes:
 *   nn.Conv2d          (O,I,kh,kw) -> [O][kh][kw][I]
 *   nn.ConvTranspose2d (I,O,kh,kw) -> [I][kh][kw][O]
 * Every function returns 0 on success or a negative LBC_E* code; the message is
 * available from lbc_last_error().  Nothing here ever calls abort().
 *
 * The reference (dotchen/LearningByCheating) has no FFI layer: its per-step
 * arithmetic is delegated to torch.nn modules.  Each entry point below names
 * the reference call site whose arithmetic it replaces.
 */
#ifndef LBC_HIP_H
#define LBC_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* lbc_stream_t;

const char* lbc_last_error(void);
const char* lbc_backend(void);   /* "hip-gfx950" for the product library */
/* ABI version of THIS header; lbc_version() returns the one the library was built with -- a host compares the two at load time.
 * 100: rounds 1-3.  101: lbc_conv_desc grew split_workspace / split_workspace_bytes (round 4; the library still answered 100).
 * 200: lbc_conv_desc starts with struct_size, which every entry point checks (a descriptor from an older header, or one that was not
 *      initialised, is refused with LBC_EINVAL instead of being read past); lbc_adam_profile_elems.  Accepted: every struct_size from the
 *      ABI-200 layout (through split_workspace_bytes) up to the library's own sizeof -- fields appended later are optional for older hosts.
 * The size_t-returning *_workspace() queries and the int-returning *_supported() queries answer 0 for "none / no" AND for a refused
 * descriptor: a host that gets 0 checks lbc_last_error() (empty = a genuine 0), as tests/c_host/host.c does. */
#define LBC_HIP_ABI_VERSION 200
int lbc_version(void);

typedef struct lbc_conv_desc {
    unsigned struct_size;   /* = sizeof(lbc_conv_desc): start every descriptor as `lbc_conv_desc d = LBC_CONV_DESC_INIT;` -- all other
                               fields zero (no scratch, no fused ReLU, exact f32), then fill in the geometry */
    int N, H, W, C;     /* input tensor (NHWC) */
    int K;              /* output channels */
    int KH, KW, S, P;   /* filter size, stride, padding */
    int relu;           /* fuse ReLU into the epilogue */
    int bf16;           /* 0: exact f32 MFMA.  1: MFMA operands rounded to bf16 (RNE), f32 accumulate; tensors stay f32.
                           2: as 1, and the activation tensors (x, y, resid, dy, dx: the `void*` arguments) are bf16 in
                           HBM; weights, bias, statistics and weight gradients stay f32.
                           3: as 2, and `w` of the forward / input-gradient entry points is a bf16 copy of the weights in
                           the same (depth-contiguous) layout; weight gradients are still produced in f32 */
    int w_transposed;   /* lbc_conv2d_dgrad / lbc_deconv3x3s2_fwd only: `w` is the lbc_weight_transpose()d copy (depth-
                           contiguous for these GEMMs).  Required when bf16 = 1. */
    void* split_workspace;          /* optional (NULL = none): device scratch that lets lbc_conv2d_fwd / lbc_conv2d_dgrad launches with */
    size_t split_workspace_bytes;   /* few output tiles (3x3 / stride 1, bf16 = 3, small N*H*W) cut the channel contraction into ranges:
                                       f32 partial tiles here, summed in a fixed order by a second launch that does the epilogue.  A
                                       launch uses at most 8 * N*OH*OW*K * 4 bytes and ignores a scratch that is too small.  Results
                                       differ from the unsplit launch by f32 summation order only. */
} lbc_conv_desc;
#define LBC_CONV_DESC_INIT { (unsigned)sizeof(lbc_conv_desc) }

/* nn.Conv2d forward (reference bird_view/models/resnet.py:15-22,102; image.py:57).
 * y[N,OH,OW,K] = conv(x', w) (+bias) (+resid) (relu), x' = relu?(x*pre_scale+pre_shift) when
 * pre_scale != NULL (the producing BatchNorm applied on load; zero padding stays zero).
 * stats (nullable): per-workgroup partial (sum, sum^2) of y per channel, [rows][2][K];
 * *stats_rows receives the number of rows written.  Query: call with y == NULL (nothing is launched) and the SAME descriptor, resid
 * and pre_scale (NULL or not) as the real call -- they select the kernel, and the kernel sets the row count. */
int lbc_conv2d_fwd(const lbc_conv_desc* d, const void* x, const void* w, const float* bias,
                   const void* resid, const float* pre_scale, const float* pre_shift, int pre_relu,
                   void* y, float* stats, int* stats_rows, lbc_stream_t stream);

/* w[A][T][B] -> wt[B][T][A] (fp32).  Conv2d weights [K][T][C] -> [C][T][K] for lbc_conv2d_dgrad, ConvTranspose2d weights
 * [C][T][K] -> [K][T][C] for lbc_deconv3x3s2_fwd, when lbc_conv_desc.w_transposed = 1. */
int lbc_weight_transpose_f32(const float* w, float* wt, int A, int T, int B, lbc_stream_t stream);

/* Input gradient of nn.Conv2d (autograd of the call sites above; loss.backward() at
 * training/train_image_phase1.py:204).  dx[N,H,W,C] = dgrad(dy[N,OH,OW,K], w) (+resid). */
int lbc_conv2d_dgrad(const lbc_conv_desc* d, const void* dy, const void* w, const void* resid,
                     void* dx, lbc_stream_t stream);

/* Weight gradient of nn.Conv2d.  dw[K][KH][KW][C] = beta*dw + sum_m dy[m][k] * x'[gather(m)][c].
 * workspace must hold lbc_conv2d_wgrad_workspace(d) bytes. */
size_t lbc_conv2d_wgrad_workspace(const lbc_conv_desc* d);
int lbc_conv2d_wgrad(const lbc_conv_desc* d, const void* x, const void* dy,
                     const float* pre_scale, const float* pre_shift, int pre_relu,
                     float* dw, float beta, void* workspace, lbc_stream_t stream);

/* The weight gradients of n same-shaped 3x3 / stride-1 / pad-1 convolutions on bf16 tensors (d->bf16 >= 2) in ONE launch: the
 * BasicBlock convolutions of one ResNet stage (bird_view/models/resnet.py:15-22: layer1..4 hold 6 / 7 / 11 / 5 of one shape),
 * whose gradients autograd produces one by one behind loss.backward() (training/train_image_phase1.py:204).  dw[i][K][3][3][C] =
 * sum_m dy[i][m][k] * x'[i][gather(m)][c]; pre_scale / pre_shift: nullptr, or n per-channel vectors (x' = relu?(x * scale + shift)).
 * n <= 12; workspace: lbc_conv2d_wgrad_group_workspace(d, n) bytes.  lbc_conv2d_wgrad_group_supported(d): 1 when d's geometry and
 * dtype have the grouped kernel (otherwise call lbc_conv2d_wgrad per convolution). */
int lbc_conv2d_wgrad_group_supported(const lbc_conv_desc* d);
size_t lbc_conv2d_wgrad_group_workspace(const lbc_conv_desc* d, int n);
int lbc_conv2d_wgrad_group(const lbc_conv_desc* d, int n, const void* const* x, const void* const* dy,
                           const float* const* pre_scale, const float* const* pre_shift, int pre_relu,
                           float* const* dw, void* workspace, lbc_stream_t stream);

/* nn.ConvTranspose2d(C,K,3,2,1,1) forward (reference bird_view/models/image.py:39,42,45;
 * birdview.py:37,40,43).  x[N,H,W,C] -> y[N,2H,2W,K]; d->KH=KW=3, S=2, P=1 required. */
int lbc_deconv3x3s2_fwd(const lbc_conv_desc* d, const void* x, const void* w, const float* bias,
                        const float* pre_scale, const float* pre_shift, int pre_relu,
                        void* y, float* stats, int* stats_rows, lbc_stream_t stream);
int lbc_deconv3x3s2_dgrad(const lbc_conv_desc* d, const void* dy, const void* w, void* dx, lbc_stream_t stream);
size_t lbc_deconv3x3s2_wgrad_workspace(const lbc_conv_desc* d);
int lbc_deconv3x3s2_wgrad(const lbc_conv_desc* d, const void* x, const void* dy,
                          const float* pre_scale, const float* pre_shift, int pre_relu,
                          float* dw, float beta, void* workspace, lbc_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Whole-network executor: the LbC policy networks behind the reference's module API
 * (bird_view/models/image.py:22-89 ImagePolicyModelSS, birdview.py:47-79 BirdViewPolicyModelSS).
 * The executor keeps no device memory of its own: parameters/buffers are bound by pointer
 * under the reference's state_dict names, activations live in one caller-provided workspace.
 * ---------------------------------------------------------------------------------------- */
typedef struct lbc_net_desc {
    int arch;          /* 18 or 34 (BasicBlock ResNets; reference resnet.py:162-168) */
    int in_channels;   /* 3 (RGB, ImageNet-normalised on load) or 7 (bird-view) */
    int H, W;          /* input image size, multiples of 32 (160x384 / 192x192) */
    int normalize;     /* 1: (x-mean)/std with the ImageNet constants of image.py:32-35 */
    int max_batch;
    int precision;     /* 0: f32 everywhere (exact-f32 MFMA; the parity path).  1: convolution MFMA operands rounded to
                          bf16 with f32 accumulation; tensors, BatchNorm, softmax, loss, Adam and the stem stay f32.
                          2: as 1, and activations / activation gradients are stored as bf16 in the workspace (all
                          arithmetic on them stays f32; parameters, gradients, statistics, outputs stay f32) */
} lbc_net_desc;
typedef struct lbc_net lbc_net;

int lbc_net_create(const lbc_net_desc* d, lbc_net** out);
void lbc_net_destroy(lbc_net* net);
int lbc_net_num_tensors(const lbc_net* net);
/* kind: 0 = parameter (fp32), 1 = buffer fp32, 2 = buffer int64.  name = state_dict key. */
int lbc_net_tensor_info(const lbc_net* net, int i, char* name, int name_cap, int* kind, int* ndim, int* shape4);
size_t lbc_net_workspace_bytes(const lbc_net* net);
/* Introspection for parity tests: where the activations of the last training-mode forward lie in the workspace (NHWC
 * [N][H][W][C] from offset_bytes; elem_bytes 4 = f32, 2 = bf16 (precision 2), 1 = uint8).  Names follow the reference's module
 * paths: "conv.conv1" (raw stem output), "conv.maxpool", "conv.maxpool.idx" (arg-max tap 3 r + s of MaxPool2d(3,2,1), resnet.py:106),
 * "conv.layerL.B.conv1" / ".conv2" / ".downsample.0" (raw convolution outputs, resnet.py:41-49), "conv.layerL.B" (block output,
 * resnet.py:51-52), "conv.layerL.B.bn1.scale" / ".shift" ([C] f32 vectors, H = W = 1: bn1 with the batch statistics folded, so that
 * relu(bn1(.)) is positive exactly where conv1 * scale + shift > 0), "deconv.2" / ".5" / ".8" (decoder ReLU outputs, image.py:40,43,46).  A float64 checker that freezes the ReLU
 * masks and pooling choices read from here differentiates the same piecewise-linear function as lbc_net_backward. */
int lbc_net_num_activations(const lbc_net* net);
int lbc_net_activation_info(const lbc_net* net, int i, char* name, int name_cap, size_t* offset_bytes, int* hwc3, int* elem_bytes);
/* tensor_ptrs[i] / grad_ptrs[i] in lbc_net_tensor_info order; grad_ptrs may be NULL (inference only)
 * and its entries for buffers are ignored.  4-D weights must be in channels_last memory order. */
int lbc_net_bind(lbc_net* net, void* workspace, void* const* tensor_ptrs, float* const* grad_ptrs);
/* forward(image, velocity, command) of the reference modules.  image: NCHW fp32 [N,C,H,W] (the
 * reference signature); velocity [N]; command [N,4] one-hot.  Outputs: pred_sel [N,5,2] and
 * pred_all [N,4,5,2] (normalised [-1,1] camera/map coordinates).  train != 0: batch statistics,
 * running-stat update, activations kept for backward. */
int lbc_net_forward(lbc_net* net, int N, int train, const float* image, const float* velocity,
                    const float* command, float* pred_sel, float* pred_all, lbc_stream_t stream);
/* Same with the frames as the dataset stores them: uint8 NHWC [N,H,W,C], 0..255 (reference
 * bird_view/utils/datasets/image_lmdb.py:128-222 converts them to f32 CHW /255 on the host: 4x the H2D and input bytes).
 * /255, the ImageNet normalisation and the NHWC repack are fused into the stem's input pass. */
int lbc_net_forward_u8(lbc_net* net, int N, int train, const unsigned char* image_nhwc, const float* velocity,
                       const float* command, float* pred_sel, float* pred_all, lbc_stream_t stream);
/* Backward of the last training-mode forward: gradients of a scalar wrt pred_sel / pred_all
 * (either may be NULL) -> every bound parameter gradient (overwritten, not accumulated).
 * stage = -1 runs everything; stages 0..lbc_net_num_stages()-1 run in order (head+decoder,
 * layer4, layer3, layer2, layer1, stem) so gradient buckets can be all-reduced while the
 * remaining stages execute.  The residual blocks' weight gradients run on an internal side stream next to their input
 * gradients; every stage joins it before returning control of `stream` (LBC_NO_SIDE_STREAM=1 keeps everything on `stream`). */
int lbc_net_num_stages(void);
/* State lbc_net_backward() would differentiate: batch size and mode of the last forward, and a counter that every forward
 * increments -- a caller that holds several forward results (autograd) can detect that the workspace has moved on. */
int lbc_net_last_forward(const lbc_net* net, int* batch, int* train, long long* generation);
/* frozen = 1: the caller promises not to change parameters or buffers until it says otherwise (the frozen privileged teacher of
 * reference training/train_image_phase1.py:244-248, phase2_utils.py:70-77).  Eval-mode forwards then derive what they derive from
 * the weights alone -- the bf16 weight copies (precision 2) and every BatchNorm's folded affine -- on the first forward only instead of
 * on every one.  frozen = 0 (default) or any lbc_net_bind: derived again on the next forward. */
int lbc_net_set_frozen(lbc_net* net, int frozen);
int lbc_net_backward(lbc_net* net, const float* d_sel, const float* d_all, int stage, lbc_stream_t stream);
/* Synchronized BatchNorm for data-parallel training (not in the reference, which is single-device; the equivalent of wrapping
 * its modules in torch.nn.SyncBatchNorm): every BatchNorm of a training-mode forward normalises with the statistics of the
 * GLOBAL batch, and its backward uses the global gradient sums -- a per-GPU batch of 32 then trains like the 256-image batch
 * of BASELINE.json instead of eight 32-image batches.  Before each finalize the net reduces that layer's per-channel sums to
 * one row of `count` floats in `buf` and calls fn(ctx, buf, count, stream), which must enqueue an in-place SUM all-reduce
 * over the data-parallel group on `stream` (ncclAllReduce of RCCL takes exactly these arguments) and return 0, or non-zero
 * to abort the step (LBC_ELAUNCH).  dgamma / dbeta stay local sums, as every other parameter gradient: the caller's gradient
 * all-reduce completes them.  buf: device memory, buf_floats >= 1536.  fn == NULL switches back to local statistics.
 * Eval-mode forwards never call fn. */
typedef int (*lbc_allreduce_fn)(void* ctx, float* buf, int count, lbc_stream_t stream);
int lbc_net_set_sync_bn(lbc_net* net, lbc_allreduce_fn fn, void* ctx, int world_size, float* buf, int buf_floats);
/* The RCCL communicator that callback normally is (csrc/comm.cpp): one per process/GPU, built from an id that rank 0 creates and
 * the host distributes over whatever channel it has (torch.distributed broadcast, MPI, a file).  lbc_comm_create is collective
 * over the world and binds the calling thread's current HIP device.  lbc_comm_allreduce_f32 is an lbc_allreduce_fn with
 * ctx = the lbc_comm: ncclAllReduce(buf, buf, count, ncclFloat32, ncclSum) enqueued on `stream`, nothing else.
 * RCCL is bound at run time (the librccl.so already in the process, else the system one): without it these return
 * LBC_EINVAL with the reason in lbc_last_error(), the rest of the library is unaffected. */
#define LBC_COMM_ID_BYTES 128
typedef struct lbc_comm lbc_comm;
int lbc_comm_unique_id(unsigned char* id /* [LBC_COMM_ID_BYTES] */);
int lbc_comm_create(const unsigned char* id, int rank, int world_size, lbc_comm** out);
void lbc_comm_destroy(lbc_comm* comm);
int lbc_comm_world_size(const lbc_comm* comm);
int lbc_comm_allreduce_f32(void* comm, float* buf, int count, lbc_stream_t stream);

/* Losses (forward value per sample + gradient wrt pred), reference training/train_image_phase1.py:35-70,
 * train_image_phase0.py:36-89, train_birdview.py:33-54.  kind: 0 phase-0, 1 phase-1, 2 bird-view L1 (pixel targets), 3 L1 vs normalised targets.
 * rows = waypoints per sample (5 or 20).  dpred = grad_scale * d(sum_n loss[n])/dpred. */
typedef struct lbc_camera { float w, h, fov, world_y, fixed_offset, pixels_per_meter, crop_size; } lbc_camera;
int lbc_loss(int kind, const lbc_camera* cam, const float* pred, const float* target, int N, int rows,
             float grad_scale, float* loss_per_sample, float* dpred, lbc_stream_t stream);

/* Phase-2 (DAgger) resampling weight per sample: reference training/phase2_utils.py:50-59 (get_weight) on the selected
 * branch, applied as in train_image_phase2.py:203-206.  pred_sel [N,5,2] camera space, teacher_sel [N,5,2] map space. */
int lbc_phase2_weight(const lbc_camera* cam, const float* pred_sel, const float* teacher_sel, int N, float* weights,
                      lbc_stream_t stream);

/* Multi-tensor Adam (torch.optim.Adam semantics; reference training/train_image_phase1.py:252).
 * chunk table lives in device memory: see lbc_adam_chunk. */
typedef struct lbc_adam_chunk { float* p; const float* g; float* m; float* v; int n; int pad; } lbc_adam_chunk;
int lbc_adam_step(const lbc_adam_chunk* chunks_dev, int nchunks, double lr, double beta1, double beta2,
                  double eps, double weight_decay, int step, lbc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Single-operator entry points of the HBM-bound kernels: what the executor above launches between the
 * convolutions, exported so that every kernel has its own parity test against torch CPU (tests/test_ops.py).
 * act_bf16 = 1: the activation tensors (void*) are bf16 in HBM, arithmetic stays f32.
 * ---------------------------------------------------------------------------------------- */

/* nn.BatchNorm2d (reference resnet.py:31,34,104,137; image.py:38,41,44,56).
 * lbc_bn_stats: per-workgroup partial (sum, sum^2) rows of x[pixels][C] -> partial[rows][2][C] (the convolution epilogues
 *   produce the same rows for their outputs); *rows receives the row count (query with x == NULL allowed; <= 1024).
 * lbc_bn_finalize_stats: partial rows -> scale = gamma*invstd, shift = beta - mean*scale, saved mean / invstd; train != 0
 *   also updates running_mean / running_var (momentum, unbiased variance) and num_batches_tracked (all nullable);
 *   train == 0 takes the statistics from running_mean / running_var.
 * lbc_bn_apply_relu_add_fwd: y = relu?(x*scale + shift (+ resid [*rscale + rshift])). */
int lbc_bn_stats(const void* x, long long pixels, int C, int act_bf16, float* partial, int* rows, lbc_stream_t stream);
int lbc_bn_finalize_stats(const float* partial, int rows, int C, long long count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                          int train, float* scale, float* shift, float* save_mean, float* save_invstd, lbc_stream_t stream);
int lbc_bn_apply_relu_add_fwd(const void* x, void* y, long long pixels, int C, const float* scale, const float* shift,
                              const void* resid, const float* rscale, const float* rshift, int relu, int act_bf16,
                              lbc_stream_t stream);
/* BatchNorm2d backward (autograd of the call sites above), fused with the backward of a following ReLU:
 *   g = dz * (mask > 0)   (mask nullable; mask_scale/mask_shift nullable: mask := mask*mask_scale + mask_shift first)
 *   dbeta = sum g, dgamma = sum g*xhat, dx = gamma*invstd*(g - dbeta/n - xhat*dgamma/n) over the first Cout channels.
 * g_out (nullable, may alias dz) receives g.  workspace: lbc_bn_bwd_workspace(C) bytes. */
size_t lbc_bn_bwd_workspace(int C);
int lbc_bn_bwd(const void* x, const void* dz, const void* mask, const float* mask_scale, const float* mask_shift,
               void* g_out, const float* gamma, const float* mean, const float* invstd, long long pixels, int C, int Cout,
               float* dgamma, float* dbeta, void* dx, float* workspace, int act_bf16, lbc_stream_t stream);

/* bn1 -> relu -> nn.MaxPool2d(3,2,1) of the ResNet stem (reference resnet.py:149-152, :106) and its backward.
 * fwd: y[N,H,W,C] (pre-BN) -> p[N,H/2,W/2,C], idx = arg-max tap 0..8 per output element (u8, nullable).
 * bwd: dp -> g[N,H,W,C] = gradient wrt the BatchNorm output (ReLU mask applied) + BatchNorm-backward partial rows
 *   (sum g, sum g*xhat) in partial[rows][2][C]; *rows receives the row count (query with dp == NULL allowed). */
int lbc_maxpool3x3s2_fwd(const void* y, const float* scale, const float* shift, void* p, unsigned char* idx, int N, int H, int W,
                         int C, int act_bf16, lbc_stream_t stream);
int lbc_maxpool3x3s2_bwd(const void* dp, const unsigned char* idx, const void* y, const float* scale, const float* shift,
                         const float* mean, const float* invstd, void* g, float* partial, int* rows, int N, int H, int W, int C,
                         int act_bf16, lbc_stream_t stream);

/* Waypoint head: 4 x (BatchNorm2d(64) -> Conv2d(64,5,1) -> SpatialSoftmax) -> stack -> select_branch
 * (reference image.py:54-60,82-84; common.py:29-35,136-152).  All per-branch parameter arrays are [4] pointers. */
typedef struct lbc_head_desc {
    const void* h;             /* decoder output [N][OH*OW][64], f32 or bf16 */
    int N, OH, OW, act_bf16;
    const float* mean[4];      /* BatchNorm statistics per branch [64]; training mode: the four entries are the same batch statistics */
    const float* invstd[4];
    const float* gamma[4];
    const float* beta[4];
    const float* w[4];         /* [5][64] */
    const float* bias[4];      /* [5] */
    const float* pos_x[4];     /* SpatialSoftmax buffers [OH*OW] */
    const float* pos_y[4];
    const float* cmd;          /* [N][4] one-hot */
} lbc_head_desc;
/* workspace (floats): N*40 (row statistics, fwd -> bwd) + N*20*65 + 8*20*65 + 3*64 + N*16*20*4 */
size_t lbc_head_workspace(int N);
int lbc_head_fwd(const lbc_head_desc* d, float* pred_all, float* pred_sel, float* workspace, lbc_stream_t stream);
/* backward of the last lbc_head_fwd on the same workspace (training-mode statistics): d_all [N,4,5,2] / d_sel [N,5,2]
 * (either nullable) -> dh (like h) and the parameter gradients dgamma/dbeta [4][64], dw [4][5*64], dbias [4][5]. */
int lbc_head_bwd(const lbc_head_desc* d, const float* pred_all, const float* d_all, const float* d_sel, void* dh,
                 float* const* dgamma, float* const* dbeta, float* const* dw, float* const* dbias, float* workspace,
                 lbc_stream_t stream);

/* Stem: input pass + 7x7/2 convolution 3|7 -> 64 (reference resnet.py:102; common.py:101-109 NormalizeV2 fused).
 * lbc_nchw_to_input / lbc_u8nhwc_to_input: image -> xp[N][H+6][W+6][C] (3-pixel zero border; bf16 when xp_bf16),
 *   optionally ImageNet-normalised (normalize = 1, C = 3).
 * lbc_stem_fwd: xp, w[64][7][7][C] -> y[N][H/2][W/2][64] (+ statistics partial rows, nullable).
 * lbc_stem_wgrad: xp, dy -> dw[64][7][7][C]; workspace lbc_stem_wgrad_workspace() bytes.  bf16: lbc_conv_desc.bf16 modes 0/1/2. */
int lbc_nchw_to_input(const float* image_nchw, void* xp, int xp_bf16, int N, int C, int H, int W, int normalize, lbc_stream_t stream);
int lbc_u8nhwc_to_input(const unsigned char* image_nhwc, void* xp, int xp_bf16, int N, int C, int H, int W, int normalize,
                        lbc_stream_t stream);
int lbc_stem_fwd(const void* xp, const float* w, void* y, float* stats, int* stats_rows, int N, int H, int W, int C, int bf16,
                 lbc_stream_t stream);
size_t lbc_stem_wgrad_workspace(int N, int H, int W, int C);
int lbc_stem_wgrad(const void* xp, const void* dy, float* dw, void* workspace, int N, int H, int W, int C, int bf16,
                   lbc_stream_t stream);

/* Device-side input pipeline on the dataset's uint8 frames (what the reference does per sample in CPU dataloader workers).
 * lbc_birdview_crop_u8: the fixed crop of the stored 320 x 320 x 7 bird-view (reference image_lmdb.py:150-163: rows
 *   [58:250], cols [64:256]); generic window copy src[N][SH][SW][C] -> dst[N][H][W][C].
 * lbc_augment_rgb_u8: the "super_hard" colour augmentation recipe (reference bird_view/augmenter.py:227-279) on a batch
 *   of RGB frames [N][H][W][3], in place; per-image parameters (operator order, magnitudes, seed) come from the host
 *   (learningbycheating_amd/bird_view/augmenter.py), per-pixel randomness from a counter-based hash.  scratch: N*H*W*3
 *   floats (needed when any image's sequence contains the blur). */
typedef struct lbc_aug_params {
    int order[8];                /* 0 blur, 1 gaussian noise, 2 coarse dropout, 3 dropout, 4 add, 5 multiply, 6 contrast */
    int n_ops, blur_pos;         /* blur_pos = index of the blur in order[], n_ops if absent */
    unsigned seed;
    float blur_sigma;
    float noise_scale; int noise_per_channel;
    float coarse_p; int coarse_h, coarse_w, coarse_per_channel;
    float dropout_p; int dropout_per_channel;
    float add[3], multiply[3], contrast[3];
} lbc_aug_params;
int lbc_birdview_crop_u8(const unsigned char* src, unsigned char* dst, int N, int SH, int SW, int C, int y0, int x0, int H, int W,
                         lbc_stream_t stream);
/* lbc_birdview_warp_crop_u8: the jittered sample of the privileged agent's loader (reference bird_view/utils/datasets/birdview_lmdb.py:
 *   103-125): per image, cv2.warpAffine(bird_view, cv2.getRotationMatrix2D((160, 260), delta_angle, 1.0), (320, 320), INTER_LINEAR) and
 *   the 192 x 192 window at (y0, x0) of the result, in one pass.  params[n].im = the INVERTED 2 x 3 matrix (what warpAffine derives from
 *   its argument), source (X, Y) = (im[0] x + im[1] y + im[2], im[3] x + im[4] y + im[5]); OpenCV's 8-bit bilinear arithmetic (1/32-pixel
 *   coordinates, 15-bit weight table, zero border).  src [N][SH][SW][C] -> dst [N][H][W][C]. */
typedef struct lbc_warp_params { double im[6]; int y0, x0; } lbc_warp_params;
int lbc_birdview_warp_crop_u8(const unsigned char* src, unsigned char* dst, const lbc_warp_params* params_dev, int N, int SH, int SW, int C,
                              int H, int W, lbc_stream_t stream);
int lbc_augment_rgb_u8(unsigned char* images, const lbc_aug_params* params_dev, float* scratch, int N, int H, int W, int any_blur,
                       lbc_stream_t stream);

/* Runtime options (A/B switches, tuning knobs, test hooks): names are the LBC_* environment variables that initialise the
 * table at load time; -1 = unset.  The one option read when a network is created (LBC_NO_SIDE_STREAM) applies to
 * networks created afterwards.  An LBC_* environment variable that is not in the table (a switch of an earlier round, a typo) is reported
 * on stderr when the library loads -- it would otherwise be ignored silently and an A/B script would measure nothing. */
int lbc_config_set(const char* name, long long value);
long long lbc_config_get(const char* name);

/* Built-in launch profiler: HIP-event timing of every kernel launch on its own stream, booked per
 * kernel class together with the launch's algorithmic flops and HBM bytes.  report() writes one
 * line per class "name count total_ms total_flops total_bytes" and resets the log. */
int lbc_profile_enable(int on);
int lbc_profile_report(char* buf, int cap);
/* parameter elements behind the optimizer's chunk table: only books the profiler's bytes of an lbc_adam_step launch (28 B per element) */
void lbc_adam_profile_elems(long long n);

#ifdef __cplusplus
}
#endif
#endif /* LBC_HIP_H */
