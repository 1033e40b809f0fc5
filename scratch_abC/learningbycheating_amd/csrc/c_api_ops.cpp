// C-ABI entry points of the HBM-bound single operators (include/lbc_hip.h, "Single-operator entry points"): thin argument
// marshalling over the launchers the network executor uses (lbc_kernels.hpp), so every kernel can be driven -- and parity
// tested against torch CPU -- on its own.
#include "lbc_common.hpp"
#include "lbc_hip.h"
#include "lbc_kernels.hpp"
#include <string.h>
#include <algorithm>

namespace {
size_t up64(size_t n) { return (n + 63) / 64 * 64; }
NormConst imagenet(int enabled)
{
    NormConst nc;
    memset(&nc, 0, sizeof(nc));
    nc.enabled = enabled;
    const float m3[3] = {0.485f, 0.456f, 0.406f}, s3[3] = {0.229f, 0.224f, 0.225f};   // reference image.py:32-35
    for (int i = 0; i < 3; ++i) { nc.mean[i] = m3[i]; nc.stdv[i] = s3[i]; }
    return nc;
}
#define LBC_TRY(expr)            \
    do {                         \
        int rc__ = (expr);       \
        if (rc__) return rc__;   \
    } while (0)
}  // namespace

extern "C" {

int lbc_bn_stats(const void* x, long long pixels, int C, int act_bf16, float* partial, int* rows, lbc_stream_t stream)
{
    LBC_REQUIRE(pixels > 0 && C > 0 && C % 8 == 0 && C / 4 <= 256, "bn_stats: bad shape (%lld pixels, C = %d)", pixels, C);
    if (rows) *rows = lbc_chan_reduce_rows(pixels, C);
    if (!x) return LBC_OK;
    LBC_REQUIRE(partial, "bn_stats: null partial buffer");
    ChanReduceArgs r;
    memset(&r, 0, sizeof(r));
    r.x = x; r.partial = partial; r.pixels = pixels; r.C = C; r.act_bf16 = act_bf16;
    return lbc_chan_reduce(r, 0, (hipStream_t)stream);
}

int lbc_bn_finalize_stats(const float* partial, int rows, int C, long long count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                          int train, float* scale, float* shift, float* save_mean, float* save_invstd, lbc_stream_t stream)
{
    LBC_REQUIRE(scale && shift && C > 0, "bn_finalize_stats: null output");
    LBC_REQUIRE(!train || (partial && rows >= 1 && rows <= kLbcFinalizeRows && count > 0), "bn_finalize_stats: training mode needs 1..%d partial rows",
                kLbcFinalizeRows);
    LBC_REQUIRE(train || (running_mean && running_var), "bn_finalize_stats: eval mode needs the running statistics");
    LBC_REQUIRE((running_mean == nullptr) == (running_var == nullptr) && (save_mean == nullptr) == (save_invstd == nullptr),
                "bn_finalize_stats: running_mean/var and save_mean/invstd come in pairs");
    BnFinalizeArgs f;
    memset(&f, 0, sizeof(f));
    f.partial = partial; f.rows = rows; f.C = C; f.count = count; f.gamma = gamma; f.beta = beta;
    f.running_mean = running_mean; f.running_var = running_var; f.num_batches_tracked = num_batches_tracked;
    f.momentum = momentum; f.eps = eps; f.train = train;
    f.scale = scale; f.shift = shift; f.save_mean = save_mean; f.save_invstd = save_invstd;
    return lbc_bn_finalize(f, (hipStream_t)stream);
}

int lbc_bn_apply_relu_add_fwd(const void* x, void* y, long long pixels, int C, const float* scale, const float* shift,
                              const void* resid, const float* rscale, const float* rshift, int relu, int act_bf16,
                              lbc_stream_t stream)
{
    LBC_REQUIRE(x && y && scale && shift, "bn_apply: null argument");
    LBC_REQUIRE((rscale == nullptr) == (rshift == nullptr) && (!rscale || resid), "bn_apply: residual affine without a residual");
    BnApplyArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.pixels = pixels; a.C = C; a.scale = scale; a.shift = shift;
    a.resid = resid; a.rscale = rscale; a.rshift = rshift; a.relu = relu; a.act_bf16 = act_bf16;
    return lbc_bn_apply(a, (hipStream_t)stream);
}

size_t lbc_bn_bwd_workspace(int C) { return (up64((size_t)kLbcFinalizeRows * 2 * C) + 3 * up64((size_t)C)) * sizeof(float); }

int lbc_bn_bwd(const void* x, const void* dz, const void* mask, const float* mask_scale, const float* mask_shift,
               void* g_out, const float* gamma, const float* mean, const float* invstd, long long pixels, int C, int Cout,
               float* dgamma, float* dbeta, void* dx, float* workspace, int act_bf16, lbc_stream_t stream)
{
    LBC_REQUIRE(x && dz && mean && invstd && dx && workspace, "bn_bwd: null argument");
    LBC_REQUIRE(pixels > 0 && C % 8 == 0 && C / 4 <= 256 && Cout % 8 == 0 && Cout <= C, "bn_bwd: bad shape");
    LBC_REQUIRE(g_out || !mask_scale, "bn_bwd: an affine mask needs g_out (the apply pass reads the stored masked gradient)");
    hipStream_t s = (hipStream_t)stream;
    float* partial = workspace;                                   // lbc_chan_reduce_rows() <= kLbcFinalizeRows rows
    float* cA = partial + up64((size_t)kLbcFinalizeRows * 2 * C);
    float* cB = cA + up64((size_t)C);
    float* cD = cB + up64((size_t)C);
    ChanReduceArgs r;
    memset(&r, 0, sizeof(r));
    r.x = x; r.dz = dz; r.mask = mask; r.mask_scale = mask_scale; r.mask_shift = mask_shift; r.g_out = g_out;
    r.mean = mean; r.invstd = invstd; r.partial = partial; r.pixels = pixels; r.C = C; r.act_bf16 = act_bf16;
    LBC_TRY(lbc_chan_reduce(r, 1, s));
    const int rows = lbc_chan_reduce_rows(pixels, C);
    BnBwdFinalizeArgs f;
    memset(&f, 0, sizeof(f));
    f.partial = partial; f.rows = rows; f.C = C; f.count = pixels; f.gamma = gamma; f.mean = mean; f.invstd = invstd; f.train = 1;
    f.dgamma = dgamma; f.dbeta = dbeta; f.coefA = cA; f.coefB = cB; f.coefD = cD;
    LBC_TRY(lbc_bn_bwd_finalize(f, s));
    BnBwdApplyArgs ap;
    memset(&ap, 0, sizeof(ap));
    ap.g = g_out ? g_out : dz; ap.mask = g_out ? nullptr : mask; ap.x = x;
    ap.coefA = cA; ap.coefB = cB; ap.coefD = cD; ap.mean = mean; ap.invstd = invstd;
    ap.dx = dx; ap.pixels = pixels; ap.C = C; ap.Cout = Cout; ap.act_bf16 = act_bf16;
    return lbc_bn_bwd_apply(ap, s);
}

int lbc_maxpool3x3s2_fwd(const void* y, const float* scale, const float* shift, void* p, unsigned char* idx, int N, int H, int W,
                         int C, int act_bf16, lbc_stream_t stream)
{
    LBC_REQUIRE(y && scale && shift && p && N > 0, "maxpool_fwd: null argument");
    PoolFwdArgs a;
    memset(&a, 0, sizeof(a));
    a.y = y; a.scale = scale; a.shift = shift; a.p = p; a.idx = idx; a.N = N; a.H = H; a.W = W; a.C = C; a.act_bf16 = act_bf16;
    return lbc_bn_relu_maxpool_fwd(a, (hipStream_t)stream);
}

int lbc_maxpool3x3s2_bwd(const void* dp, const unsigned char* idx, const void* y, const float* scale, const float* shift,
                         const float* mean, const float* invstd, void* g, float* partial, int* rows, int N, int H, int W, int C,
                         int act_bf16, lbc_stream_t stream)
{
    LBC_REQUIRE(N > 0 && H > 0 && W > 0 && C % 8 == 0 && C / 4 <= 256, "maxpool_bwd: bad shape");
    if (rows) *rows = lbc_pool_bwd_rows(N, H, W, C);
    if (!dp) return LBC_OK;
    LBC_REQUIRE(idx && y && scale && shift && mean && invstd && g && partial, "maxpool_bwd: null argument");
    PoolBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.dp = dp; a.idx = idx; a.y = y; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.g = g; a.partial = partial;
    a.N = N; a.H = H; a.W = W; a.C = C; a.act_bf16 = act_bf16;
    return lbc_maxpool_relu_bwd_reduce(a, (hipStream_t)stream);
}

// ---- waypoint head --------------------------------------------------------------------------
namespace {
struct HeadWs { float* rowstat; float* s_partial; float* partial2; float* coef; float* scratch; };
HeadWs head_ws(float* w, int N)
{
    HeadWs h;
    h.rowstat = w;
    h.s_partial = h.rowstat + up64((size_t)N * 40);
    h.partial2 = h.s_partial + up64((size_t)lbc_head_bwd_max_rows(N) * 20 * 65);
    h.coef = h.partial2 + up64((size_t)8 * 20 * 65);
    h.scratch = h.coef + up64((size_t)3 * 64);
    return h;
}
void head_args(const lbc_head_desc* d, HeadArgs& ha)
{
    memset(&ha, 0, sizeof(ha));
    ha.h = d->h; ha.act_bf16 = d->act_bf16; ha.N = d->N; ha.OH = d->OH; ha.OW = d->OW; ha.cmd = d->cmd;
    for (int b = 0; b < 4; ++b) {
        ha.mean[b] = d->mean[b]; ha.invstd[b] = d->invstd[b]; ha.gamma[b] = d->gamma[b]; ha.beta[b] = d->beta[b];
        ha.w[b] = d->w[b]; ha.bias[b] = d->bias[b]; ha.pos_x[b] = d->pos_x[b]; ha.pos_y[b] = d->pos_y[b];
    }
}
}  // namespace

size_t lbc_head_workspace(int N)
{
    if (N < 1) return 0;
    return (up64((size_t)N * 40) + up64((size_t)lbc_head_bwd_max_rows(N) * 20 * 65) + up64((size_t)8 * 20 * 65) + up64((size_t)3 * 64) +
            up64((size_t)N * 16 * 20 * 4)) * sizeof(float);
}

int lbc_head_fwd(const lbc_head_desc* d, float* pred_all, float* pred_sel, float* workspace, lbc_stream_t stream)
{
    LBC_REQUIRE(d && d->h && d->cmd && pred_all && workspace, "head_fwd: null argument");
    HeadArgs ha;
    head_args(d, ha);
    const HeadWs ws = head_ws(workspace, d->N);
    ha.pred_all = pred_all; ha.pred_sel = pred_sel; ha.rowstat = ws.rowstat; ha.scratch = ws.scratch;
    return lbc_head_fwd(ha, (hipStream_t)stream);
}

int lbc_head_bwd(const lbc_head_desc* d, const float* pred_all, const float* d_all, const float* d_sel, void* dh,
                 float* const* dgamma, float* const* dbeta, float* const* dw, float* const* dbias, float* workspace,
                 lbc_stream_t stream)
{
    LBC_REQUIRE(d && d->h && d->cmd && pred_all && dh && dgamma && dbeta && dw && dbias && workspace, "head_bwd: null argument");
    hipStream_t s = (hipStream_t)stream;
    const int N = d->N;
    const HeadWs ws = head_ws(workspace, N);
    HeadBwdArgs hb;
    memset(&hb, 0, sizeof(hb));
    head_args(d, hb.f);
    hb.f.pred_all = const_cast<float*>(pred_all); hb.f.rowstat = ws.rowstat;
    hb.d_all = d_all; hb.d_sel = d_sel; hb.s_partial = ws.s_partial; hb.dh = dh; hb.chan_coef = ws.coef;
    LBC_TRY(lbc_head_bwd_reduce(hb, s));
    HeadBwdFinalizeArgs hf;
    memset(&hf, 0, sizeof(hf));
    hf.s_partial = ws.s_partial; hf.rows = lbc_head_bwd_rows(hb.f); hf.count = (long long)N * d->OH * d->OW;
    if (hf.rows > 8) {
        LBC_TRY(lbc_partial_reduce(ws.s_partial, hf.rows, 20 * 65, ws.partial2, 8, s));
        hf.s_partial = ws.partial2; hf.rows = 8;
    }
    for (int b = 0; b < 4; ++b) {
        hf.gamma[b] = d->gamma[b]; hf.beta[b] = d->beta[b]; hf.w[b] = d->w[b];
        hf.dgamma[b] = dgamma[b]; hf.dbeta[b] = dbeta[b]; hf.dw[b] = dw[b]; hf.dbias[b] = dbias[b];
    }
    hf.mean = d->mean[0]; hf.invstd = d->invstd[0]; hf.chan_coef = ws.coef;
    LBC_TRY(lbc_head_bwd_finalize(hf, s));
    return lbc_head_bwd_apply(hb, s);
}

// ---- stem -----------------------------------------------------------------------------------
int lbc_nchw_to_input(const float* image_nchw, void* xp, int xp_bf16, int N, int C, int H, int W, int normalize, lbc_stream_t stream)
{
    LBC_REQUIRE(image_nchw && xp && N > 0 && (!normalize || C == 3), "nchw_to_input: bad arguments");
    return lbc_prep_input(image_nchw, xp, xp_bf16, N, C, H, W, imagenet(normalize), (hipStream_t)stream);
}

int lbc_u8nhwc_to_input(const unsigned char* image_nhwc, void* xp, int xp_bf16, int N, int C, int H, int W, int normalize,
                        lbc_stream_t stream)
{
    LBC_REQUIRE(image_nhwc && xp && N > 0 && (!normalize || C == 3), "u8nhwc_to_input: bad arguments");
    return lbc_prep_input_u8(image_nhwc, xp, xp_bf16, N, C, H, W, imagenet(normalize), (hipStream_t)stream);
}

int lbc_stem_fwd(const void* xp, const float* w, void* y, float* stats, int* stats_rows, int N, int H, int W, int C, int bf16,
                 lbc_stream_t stream)
{
    LBC_REQUIRE(N > 0 && bf16 >= 0 && bf16 <= 2, "stem_fwd: bad arguments");
    StemArgs st;
    memset(&st, 0, sizeof(st));
    st.xp = xp; st.xp_bf16 = bf16 != 0; st.w = w; st.y = y; st.stats = stats;
    st.N = N; st.H = H; st.W = W; st.Cin = C; st.act_bf16 = bf16 == 2; st.bf16 = bf16 != 0;
    if (stats_rows) *stats_rows = lbc_stem_rows(st);
    if (!y) return LBC_OK;
    LBC_REQUIRE(xp && w, "stem_fwd: null argument");
    return lbc_stem_fwd(st, (hipStream_t)stream);
}

size_t lbc_stem_wgrad_workspace(int N, int H, int W, int C)
{
    // the larger of the two kernels' slab counts (the precision is not known here)
    const int ns = std::max(lbc_stem_wgrad_split(N, H, W, C, 0), lbc_stem_wgrad_split(N, H, W, C, 1));
    return (size_t)ns * 64 * 49 * C * sizeof(float);
}

int lbc_stem_wgrad(const void* xp, const void* dy, float* dw, void* workspace, int N, int H, int W, int C, int bf16,
                   lbc_stream_t stream)
{
    LBC_REQUIRE(xp && dy && dw && workspace && N > 0 && bf16 >= 0 && bf16 <= 2, "stem_wgrad: bad arguments");
    StemWgradArgs sw;
    memset(&sw, 0, sizeof(sw));
    sw.xp = xp; sw.xp_bf16 = bf16 != 0; sw.dy = dy; sw.partial = static_cast<float*>(workspace);
    sw.N = N; sw.H = H; sw.W = W; sw.Cin = C; sw.act_bf16 = bf16 == 2; sw.bf16 = bf16 != 0;
    sw.nsplit = lbc_stem_wgrad_split(N, H, W, C, bf16 != 0);
    LBC_TRY(lbc_stem_wgrad(sw, (hipStream_t)stream));
    return lbc_splitk_reduce(sw.partial, sw.nsplit, (long long)64 * 49 * C, dw, 0.f, (hipStream_t)stream);
}

// ---- device-side input pipeline ------------------------------------------------------------------------------
int lbc_birdview_crop_u8(const unsigned char* src, unsigned char* dst, int N, int SH, int SW, int C, int y0, int x0, int H, int W,
                         lbc_stream_t stream)
{
    return lbc_crop_u8(src, dst, N, SH, SW, C, y0, x0, H, W, (hipStream_t)stream);
}

int lbc_birdview_warp_crop_u8(const unsigned char* src, unsigned char* dst, const lbc_warp_params* params_dev, int N, int SH, int SW, int C,
                              int H, int W, lbc_stream_t stream)
{
    static_assert(sizeof(lbc_warp_params) == sizeof(WarpParams), "warp parameter layout");
    return lbc_warp_crop_u8(src, dst, reinterpret_cast<const WarpParams*>(params_dev), N, SH, SW, C, H, W, (hipStream_t)stream);
}

int lbc_augment_rgb_u8(unsigned char* images, const lbc_aug_params* params_dev, float* scratch, int N, int H, int W, int any_blur,
                       lbc_stream_t stream)
{
    static_assert(sizeof(lbc_aug_params) == sizeof(AugParams), "augmentation parameter layout");
    return lbc_augment_u8(images, reinterpret_cast<const AugParams*>(params_dev), scratch, N, H, W, any_blur, (hipStream_t)stream);
}

}  // extern "C"
