// C-ABI entry points (include/lbc_hip.h) over the internal launchers.
#include "lbc_common.hpp"
#include "lbc_hip.h"
#include <stddef.h>
#include <string.h>

extern "C" {

const char* lbc_backend(void)
{
#ifdef LBC_HIP_EMULATED_FOR_TESTS
    return "emu-cpu";
#else
    return "hip-gfx950";
#endif
}
int lbc_version(void) { return LBC_HIP_ABI_VERSION; }

// Every entry point that takes a descriptor checks it: struct_size must cover the fields of the first checked layout (ABI 200: everything up to
// and including split_workspace_bytes) and must not exceed this library's struct.  A host built against an OLDER header of the same major
// ABI (fewer trailing fields) stays valid: whoever appends a field must read it only where struct_size covers it (today the checked
// layout IS the struct: the two bounds coincide).  A host that forgot LBC_CONV_DESC_INIT (struct_size 0 / garbage) is refused.
static const unsigned kDescMinSize = (unsigned)(offsetof(lbc_conv_desc, split_workspace_bytes) + sizeof(((lbc_conv_desc*)0)->split_workspace_bytes));
static bool desc_ok(const lbc_conv_desc* d, const char* who)
{
    if (!d) { lbc_set_error("%s: null descriptor", who); return false; }
    if (d->struct_size < kDescMinSize || d->struct_size > sizeof(lbc_conv_desc)) {
        lbc_set_error("%s: lbc_conv_desc.struct_size is %u, this library (ABI %d) accepts %u .. %zu -- initialise the descriptor with LBC_CONV_DESC_INIT "
                      "and build the host against a lbc_hip.h of this ABI", who, d->struct_size, LBC_HIP_ABI_VERSION, kDescMinSize, sizeof(lbc_conv_desc));
        return false;
    }
    return true;
}
#define LBC_DESC(d, who) do { if (!desc_ok((d), (who))) return LBC_EINVAL; } while (0)

static IgemmArgs conv_args(const lbc_conv_desc* d)
{
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C; a.K = d->K;
    a.KH = d->KH; a.KW = d->KW; a.S = d->S; a.P = d->P;
    a.relu = d->relu;
    a.bf16 = d->bf16 != 0; a.act_bf16 = d->bf16 >= 2; a.w_bf16 = d->bf16 == 3;
    a.split_ws = static_cast<float*>(d->split_workspace); a.split_ws_floats = (long long)(d->split_workspace_bytes / 4);
    return a;
}

int lbc_conv2d_fwd(const lbc_conv_desc* d, const void* x, const void* w, const float* bias,
                   const void* resid, const float* pre_scale, const float* pre_shift, int pre_relu,
                   void* y, float* stats, int* stats_rows, lbc_stream_t stream)
{
    LBC_DESC(d, "conv2d_fwd");
    IgemmArgs a = conv_args(d);
    a.OH = (d->H + 2 * d->P - d->KH) / d->S + 1;
    a.OW = (d->W + 2 * d->P - d->KW) / d->S + 1;
    a.LH = a.OH; a.LW = a.OW; a.ostep = 1;
    a.M = d->N * a.OH * a.OW;
    a.pre_scale = pre_scale; a.pre_shift = pre_shift; a.pre_relu = pre_relu;
    a.resid = resid;
    // NB a row-count query must describe the launch it is for: pass the same pre_scale and resid (NULL or not) as the real call
    const int cfg = lbc_igemm_pick_for(a, 0);
    if (stats_rows) *stats_rows = lbc_igemm_rows(a, cfg);
    if (!y) return LBC_OK;   // query only
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.resid = resid; a.stats = stats;
    return lbc_igemm_launch(a, /*wmajor=*/1, /*mode=*/0, cfg, (hipStream_t)stream);
}

// dgrad of a Conv2d with geometry d: gathered tensor = dy [N,OH,OW,K], output = dx [N,H,W,C]
static int conv_dgrad_impl(const lbc_conv_desc* d, const void* dy, const void* w, int wmajor, const void* resid,
                           const float* bias, const float* pre_scale, const float* pre_shift, int pre_relu,
                           int relu, void* dx, float* stats, int* stats_rows, hipStream_t s)
{
    const int OH = (d->H + 2 * d->P - d->KH) / d->S + 1;
    const int OW = (d->W + 2 * d->P - d->KW) / d->S + 1;
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = dy; a.w = w; a.y = dx; a.resid = resid; a.bias = bias; a.relu = relu; a.bf16 = d->bf16 != 0; a.act_bf16 = d->bf16 >= 2; a.w_bf16 = d->bf16 == 3;
    a.pre_scale = pre_scale; a.pre_shift = pre_shift; a.pre_relu = pre_relu;
    a.split_ws = static_cast<float*>(d->split_workspace); a.split_ws_floats = (long long)(d->split_workspace_bytes / 4);
    a.N = d->N; a.H = OH; a.W = OW; a.C = d->K;
    a.OH = d->H; a.OW = d->W; a.K = d->C;
    a.KH = d->KH; a.KW = d->KW; a.S = d->S; a.P = d->P;
    a.stats = stats;
    int rows = 0;
    if (d->S == 1) {
        a.LH = a.OH; a.LW = a.OW; a.ostep = 1; a.oy0 = 0; a.ox0 = 0;
        a.M = d->N * a.LH * a.LW;
        const int cfg = wmajor ? lbc_igemm_pick_for(a, 1) : lbc_igemm_pick(a.M, a.K);
        rows = lbc_igemm_rows(a, cfg);
        if (stats_rows) *stats_rows = rows;
        if (!dx) return LBC_OK;
        return lbc_igemm_launch(a, wmajor, 1, cfg, s);
    }
    // stride 2: one launch per output parity phase
    LBC_REQUIRE(d->H % 2 == 0 && d->W % 2 == 0, "dgrad s2: odd spatial size %dx%d", d->H, d->W);
    a.LH = d->H / 2; a.LW = d->W / 2; a.ostep = 2;
    a.M = d->N * a.LH * a.LW;
    a.nphase = 4;        // one launch for the four output-parity phases (statistics rows ph * per + tile)
    const int cfg = wmajor ? lbc_igemm_pick_for(a, 1) : lbc_igemm_pick(a.M, a.K);
    const int per = lbc_igemm_rows(a, cfg);
    if (stats_rows) *stats_rows = 4 * per;
    if (!dx) return LBC_OK;
    return lbc_igemm_launch(a, wmajor, 1, cfg, s);
}

int lbc_weight_transpose_f32(const float* w, float* wt, int A, int T, int B, lbc_stream_t stream)
{
    LBC_REQUIRE(w && wt && A > 0 && T > 0 && B > 0, "weight_transpose: bad arguments");
    return lbc_weight_transpose(w, wt, A, T, B, (hipStream_t)stream);
}

int lbc_conv2d_dgrad(const lbc_conv_desc* d, const void* dy, const void* w, const void* resid,
                     void* dx, lbc_stream_t stream)
{
    LBC_DESC(d, "conv2d_dgrad");
    LBC_REQUIRE(!d->bf16 || d->w_transposed, "conv2d_dgrad: bf16 needs w_transposed = 1 (see lbc_weight_transpose_f32)");
    // weights [K][T][C]: depth index (gathered channel) = k is the slow axis -> wmajor 0 with row length C;
    // the transposed copy [C][T][K] is depth-contiguous -> wmajor 1
    return conv_dgrad_impl(d, dy, w, /*wmajor=*/d->w_transposed ? 1 : 0, resid, nullptr, nullptr, nullptr, 0, 0, dx, nullptr, nullptr,
                           (hipStream_t)stream);
}

// ConvTranspose2d(C->K, 3, 2, 1, 1) forward == dgrad of a Conv2d(K->C, 3, 2, 1) whose weight tensor
// [C_T][kh][kw][K_T] is exactly the transposed conv's channels_last weight.
static lbc_conv_desc deconv_as_conv(const lbc_conv_desc* d)
{
    lbc_conv_desc c = *d;
    c.N = d->N; c.H = 2 * d->H; c.W = 2 * d->W; c.C = d->K;   // the conv's input is the deconv's output
    c.K = d->C;
    return c;
}

int lbc_deconv3x3s2_fwd(const lbc_conv_desc* d, const void* x, const void* w, const float* bias,
                        const float* pre_scale, const float* pre_shift, int pre_relu,
                        void* y, float* stats, int* stats_rows, lbc_stream_t stream)
{
    LBC_DESC(d, "deconv3x3s2_fwd");
    LBC_REQUIRE(d->KH == 3 && d->KW == 3 && d->S == 2 && d->P == 1, "deconv3x3s2_fwd: geometry must be k3 s2 p1 op1");
    LBC_REQUIRE(!d->bf16 || d->w_transposed, "deconv3x3s2_fwd: bf16 needs w_transposed = 1 (see lbc_weight_transpose_f32)");
    lbc_conv_desc c = deconv_as_conv(d);
    return conv_dgrad_impl(&c, x, w, /*wmajor=*/d->w_transposed ? 1 : 0, nullptr, bias, pre_scale, pre_shift, pre_relu, d->relu, y, stats,
                           stats_rows, (hipStream_t)stream);
}

int lbc_deconv3x3s2_dgrad(const lbc_conv_desc* d, const void* dy, const void* w, void* dx, lbc_stream_t stream)
{
    LBC_DESC(d, "deconv3x3s2_dgrad");
    LBC_REQUIRE(d->KH == 3 && d->KW == 3 && d->S == 2 && d->P == 1, "deconv3x3s2_dgrad: geometry must be k3 s2 p1 op1");
    lbc_conv_desc c = deconv_as_conv(d);
    c.relu = 0;
    // forward gather conv over dy with the deconv weight read as [O=C_T][kh][kw][I=K_T]
    return lbc_conv2d_fwd(&c, dy, w, nullptr, nullptr, nullptr, nullptr, 0, dx, nullptr, nullptr, stream);
}


static WgradArgs conv_wgrad_args(const lbc_conv_desc* d)
{
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.N = d->N;
    a.OH = (d->H + 2 * d->P - d->KH) / d->S + 1;
    a.OW = (d->W + 2 * d->P - d->KW) / d->S + 1;
    a.CP = d->K;
    a.H = d->H; a.W = d->W; a.CQ = d->C;
    a.KH = d->KH; a.KW = d->KW; a.S = d->S; a.P = d->P;
    a.bf16 = d->bf16 != 0; a.act_bf16 = d->bf16 >= 2;
    a.nsplit = lbc_wgrad_pick_split(a);
    return a;
}

size_t lbc_conv2d_wgrad_workspace(const lbc_conv_desc* d)
{
    if (!desc_ok(d, "conv2d_wgrad_workspace")) return 0;
    WgradArgs a = conv_wgrad_args(d);
    return (size_t)a.nsplit * (size_t)a.CP * (size_t)(a.KH * a.KW) * (size_t)a.CQ * sizeof(float);
}

int lbc_conv2d_wgrad(const lbc_conv_desc* d, const void* x, const void* dy,
                     const float* pre_scale, const float* pre_shift, int pre_relu,
                     float* dw, float beta, void* workspace, lbc_stream_t stream)
{
    LBC_DESC(d, "conv2d_wgrad");
    LBC_REQUIRE(workspace, "conv2d_wgrad: null workspace");
    WgradArgs a = conv_wgrad_args(d);
    a.p = dy; a.q = x; a.partial = (float*)workspace;
    a.q_scale = pre_scale; a.q_shift = pre_shift; a.q_relu = pre_relu;
    if (a.nsplit == 1 && beta == 0.f) { a.partial = dw; return lbc_wgrad_launch(a, (hipStream_t)stream); }
    int rc = lbc_wgrad_launch(a, (hipStream_t)stream);
    if (rc) return rc;
    return lbc_splitk_reduce(a.partial, a.nsplit, (long long)a.CP * a.KH * a.KW * a.CQ, dw, beta, (hipStream_t)stream);
}

// n same-shaped 3x3 / stride-1 convolutions on bf16 tensors in one launch (+ one reduce launch)
int lbc_conv2d_wgrad_group_supported(const lbc_conv_desc* d)
{
    if (!desc_ok(d, "conv2d_wgrad_group_supported")) return 0;
    WgradArgs a = conv_wgrad_args(d);
    a.p = d; a.q = d;     // (eligibility looks at geometry and flags only)
    return lbc_wgrad_tr_eligible(a) ? 1 : 0;
}

size_t lbc_conv2d_wgrad_group_workspace(const lbc_conv_desc* d, int n)
{
    if (!d || n < 1 || !lbc_conv2d_wgrad_group_supported(d)) return 0;
    WgradArgs a = conv_wgrad_args(d);
    return (size_t)lbc_wgrad_tr_group_split(a, n) * n * (size_t)a.CP * 9 * (size_t)a.CQ * sizeof(float);
}

int lbc_conv2d_wgrad_group(const lbc_conv_desc* d, int n, const void* const* x, const void* const* dy,
                           const float* const* pre_scale, const float* const* pre_shift, int pre_relu,
                           float* const* dw, void* workspace, lbc_stream_t stream)
{
    LBC_DESC(d, "conv2d_wgrad_group");
    LBC_REQUIRE(x && dy && dw && workspace && n >= 1 && n <= kLbcWgradGroupMax, "conv2d_wgrad_group: null argument or group size %d outside [1,%d]", n, kLbcWgradGroupMax);
    LBC_REQUIRE(lbc_conv2d_wgrad_group_supported(d), "conv2d_wgrad_group: 3x3 / stride 1 / pad 1 on bf16 tensors (bf16 mode >= 2), channels multiples of 64");
    WgradArgs a = conv_wgrad_args(d);
    a.nsplit = lbc_wgrad_tr_group_split(a, n);
    a.q_relu = pre_relu;
    const size_t count = (size_t)a.CP * 9 * (size_t)a.CQ;
    WgradGroup g;
    memset(&g, 0, sizeof(g));
    g.n = n;
    for (int i = 0; i < n; ++i) {
        g.p[i] = dy[i]; g.q[i] = x[i];
        g.q_scale[i] = pre_scale ? pre_scale[i] : nullptr; g.q_shift[i] = pre_shift ? pre_shift[i] : nullptr;
        g.out[i] = a.nsplit == 1 ? dw[i] : (float*)workspace + (size_t)i * a.nsplit * count;
    }
    a.q_scale = g.q_scale[0]; a.q_shift = g.q_shift[0];
    const int rc = lbc_wgrad_tr_group_launch(a, g, (hipStream_t)stream);
    if (rc || a.nsplit == 1) return rc;
    return lbc_splitk_reduce_group((const float*)workspace, a.nsplit, (long long)count, n, dw, (hipStream_t)stream);
}

// ConvTranspose2d wgrad: dw[c][kh][kw][k] = sum x'[n,iy,ix,c] * dy[n,2iy-1+kh,2ix-1+kw,k]
// == Conv2d wgrad with P-side = x (dense rows) and Q-side = dy gathered with stride 2.
static WgradArgs deconv_wgrad_args(const lbc_conv_desc* d)
{
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.N = d->N; a.OH = d->H; a.OW = d->W; a.CP = d->C;
    a.H = 2 * d->H; a.W = 2 * d->W; a.CQ = d->K;
    a.KH = 3; a.KW = 3; a.S = 2; a.P = 1;
    a.bf16 = d->bf16 != 0; a.act_bf16 = d->bf16 >= 2;
    a.nsplit = lbc_wgrad_pick_split(a);
    return a;
}

size_t lbc_deconv3x3s2_wgrad_workspace(const lbc_conv_desc* d)
{
    if (!desc_ok(d, "deconv3x3s2_wgrad_workspace")) return 0;
    WgradArgs a = deconv_wgrad_args(d);
    return (size_t)a.nsplit * (size_t)a.CP * 9 * (size_t)a.CQ * sizeof(float);
}

int lbc_deconv3x3s2_wgrad(const lbc_conv_desc* d, const void* x, const void* dy,
                          const float* pre_scale, const float* pre_shift, int pre_relu,
                          float* dw, float beta, void* workspace, lbc_stream_t stream)
{
    LBC_DESC(d, "deconv3x3s2_wgrad");
    LBC_REQUIRE(workspace, "deconv_wgrad: null workspace");
    LBC_REQUIRE(!pre_relu, "deconv_wgrad: ReLU-on-load of the dense operand is not supported");
    WgradArgs a = deconv_wgrad_args(d);
    a.p = x; a.q = dy; a.partial = (float*)workspace;
    a.p_scale = pre_scale; a.p_shift = pre_shift;
    if (a.nsplit == 1 && beta == 0.f) { a.partial = dw; return lbc_wgrad_launch(a, (hipStream_t)stream); }
    int rc = lbc_wgrad_launch(a, (hipStream_t)stream);
    if (rc) return rc;
    return lbc_splitk_reduce(a.partial, a.nsplit, (long long)a.CP * 9 * a.CQ, dw, beta, (hipStream_t)stream);
}

}  // extern "C"
