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
int lbc_version(void);

typedef struct lbc_conv_desc {
    int N, H, W, C;     /* input tensor (NHWC) */
    int K;              /* output channels */
    int KH, KW, S, P;   /* filter size, stride, padding */
    int relu;           /* fuse ReLU into the epilogue */
} lbc_conv_desc;

/* nn.Conv2d forward (reference bird_view/models/resnet.py:15-22,102; image.py:57).
 * y[N,OH,OW,K] = conv(x', w) (+bias) (+resid) (relu), x' = relu?(x*pre_scale+pre_shift) when
 * pre_scale != NULL (the producing BatchNorm applied on load; zero padding stays zero).
 * stats (nullable): per-workgroup partial (sum, sum^2) of y per channel, [rows][2][K];
 * *stats_rows receives the number of rows written (query with stats == NULL allowed). */
int lbc_conv2d_fwd(const lbc_conv_desc* d, const float* x, const float* w, const float* bias,
                   const float* resid, const float* pre_scale, const float* pre_shift, int pre_relu,
                   float* y, float* stats, int* stats_rows, lbc_stream_t stream);

/* Input gradient of nn.Conv2d (autograd of the call sites above; loss.backward() at
 * training/train_image_phase1.py:204).  dx[N,H,W,C] = dgrad(dy[N,OH,OW,K], w) (+resid). */
int lbc_conv2d_dgrad(const lbc_conv_desc* d, const float* dy, const float* w, const float* resid,
                     float* dx, lbc_stream_t stream);

/* Weight gradient of nn.Conv2d.  dw[K][KH][KW][C] = beta*dw + sum_m dy[m][k] * x'[gather(m)][c].
 * workspace must hold lbc_conv2d_wgrad_workspace(d) bytes. */
size_t lbc_conv2d_wgrad_workspace(const lbc_conv_desc* d);
int lbc_conv2d_wgrad(const lbc_conv_desc* d, const float* x, const float* dy,
                     const float* pre_scale, const float* pre_shift, int pre_relu,
                     float* dw, float beta, void* workspace, lbc_stream_t stream);

/* nn.ConvTranspose2d(C,K,3,2,1,1) forward (reference bird_view/models/image.py:39,42,45;
 * birdview.py:37,40,43).  x[N,H,W,C] -> y[N,2H,2W,K]; d->KH=KW=3, S=2, P=1 required. */
int lbc_deconv3x3s2_fwd(const lbc_conv_desc* d, const float* x, const float* w, const float* bias,
                        const float* pre_scale, const float* pre_shift, int pre_relu,
                        float* y, float* stats, int* stats_rows, lbc_stream_t stream);
int lbc_deconv3x3s2_dgrad(const lbc_conv_desc* d, const float* dy, const float* w, float* dx, lbc_stream_t stream);
size_t lbc_deconv3x3s2_wgrad_workspace(const lbc_conv_desc* d);
int lbc_deconv3x3s2_wgrad(const lbc_conv_desc* d, const float* x, const float* dy,
                          const float* pre_scale, const float* pre_shift, int pre_relu,
                          float* dw, float beta, void* workspace, lbc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LBC_HIP_H */
