// Network executor: plans and enqueues the forward / backward kernel sequences of the
// LbC policy networks.  See engine.hpp for the contract; every numbered comment cites
// the reference line whose arithmetic the kernels at that point reproduce.
#include "engine.hpp"

#include <algorithm>
#include <stdlib.h>
#include <string.h>

namespace lbc {

namespace {
constexpr float kBnMomentum = 0.1f;   // torch.nn.BatchNorm2d defaults used throughout the reference
constexpr float kBnEps = 1e-5f;
size_t up64(size_t n) { return (n + 63) / 64 * 64; }
#define LBC_TRY(expr)            \
    do {                         \
        int rc__ = (expr);       \
        if (rc__) return rc__;   \
    } while (0)
}  // namespace

int Net::add_tensor(const std::string& name, int kind, std::initializer_list<int> shape)
{
    TensorInfo ti;
    ti.name = name;
    ti.kind = kind;
    ti.ndim = (int)shape.size();
    ti.numel = 1;
    int i = 0;
    for (int s : shape) { ti.shape[i++] = s; ti.numel *= s; }
    for (; i < 4; ++i) ti.shape[i] = 1;
    t_.push_back(ti);
    return (int)t_.size() - 1;
}

size_t Net::alloc(size_t nfloats)
{
    const size_t off = ws_floats_;
    ws_floats_ += up64(nfloats);
    return off;
}

// activation / activation-gradient tensor of n elements: f32, or bf16 (half the space) in precision 2
size_t Net::alloc_act(size_t n) { return alloc(act_bf16_ ? (n + 1) / 2 : n); }

BN Net::make_bn(const std::string& p, int C)
{
    BN bn;
    bn.C = C;
    bn.g = add_tensor(p + ".weight", kParam, {C});
    bn.b = add_tensor(p + ".bias", kParam, {C});
    bn.rm = add_tensor(p + ".running_mean", kBufferF32, {C});
    bn.rv = add_tensor(p + ".running_var", kBufferF32, {C});
    bn.nbt = add_tensor(p + ".num_batches_tracked", kBufferI64, {});
    bn.scale = alloc(C); bn.shift = alloc(C); bn.mean = alloc(C); bn.invstd = alloc(C);
    bn.cA = alloc(C); bn.cB = alloc(C); bn.cD = alloc(C);
    return bn;
}

Conv Net::make_conv(const std::string& name, int Cin, int Cout, int H, int W, int k, int s, int p)
{
    Conv c;
    c.w = add_tensor(name, kParam, {Cout, Cin, k, k});
    c.Cin = Cin; c.Cout = Cout; c.H = H; c.W = W; c.k = k; c.s = s; c.p = p;
    c.OH = (H + 2 * p - k) / s + 1;
    c.OW = (W + 2 * p - k) / s + 1;
    c.y = alloc_act((size_t)d_.max_batch * c.OH * c.OW * Cout);
    if (act_bf16_) { c.wn = alloc(((size_t)Cout * Cin * k * k + 1) / 2); c.wt = alloc(((size_t)Cout * Cin * k * k + 1) / 2); }
    return c;
}

Net::Net(const lbc_net_desc& d) : d_(d)
{
    const size_t NB = (size_t)d.max_batch;
    const int Cin = d.in_channels, H0 = d.H, W0 = d.W;
    side_allowed_ = !lbc_opt_on(kOptNoSideStream);
    bf16_ = d.precision >= 1;
    act_bf16_ = d.precision == 2;
    defer_wgrad_ = act_bf16_;
    if (defer_wgrad_) side_allowed_ = false;      // nothing is left for a side stream: the deferred launches fill the chip by themselves
    if (bf16_) dgrad_wt_ = true;   // the bf16 tiles are [row][depth] only: every weight operand must be depth-contiguous

    // ---- stem (resnet.py:102-106) ----
    stem_w_ = add_tensor("conv.conv1.weight", kParam, {64, Cin, 7, 7});
    stem_bn_ = make_bn("conv.bn1", 64);
    xp_ = alloc(NB * (H0 + 6) * (W0 + 6) * Cin);
    y0_ = alloc_act(NB * (H0 / 2) * (W0 / 2) * 64);
    H1_ = H0 / 4; W1_ = W0 / 4;
    p0_ = alloc_act(NB * H1_ * W1_ * 64);
    idx_ = alloc(NB * H1_ * W1_ * 64 / 4);

    // ---- residual layers (resnet.py:107-110,124-146,162-168) ----
    const int n34[4] = {3, 4, 6, 3}, n18[4] = {2, 2, 2, 2};
    const int* nb = d.arch == 34 ? n34 : n18;
    int inpl = 64, h = H1_, w = W1_;
    size_t max_act = NB * H1_ * W1_ * 64;
    for (int li = 0; li < 4; ++li) {
        const int planes = 64 << li;
        stage_first_block_.push_back((int)blocks_.size());
        for (int bi = 0; bi < nb[li]; ++bi) {
            const int stride = (li > 0 && bi == 0) ? 2 : 1;
            const std::string pre = "conv.layer" + std::to_string(li + 1) + "." + std::to_string(bi);
            Block b;
            b.c1 = make_conv(pre + ".conv1.weight", inpl, planes, h, w, 3, stride, 1);
            b.b1 = make_bn(pre + ".bn1", planes);
            const int oh = b.c1.OH, ow = b.c1.OW;
            b.c2 = make_conv(pre + ".conv2.weight", planes, planes, oh, ow, 3, 1, 1);
            b.b2 = make_bn(pre + ".bn2", planes);
            if (stride != 1 || inpl != planes) {
                b.has_ds = true;
                b.ds = make_conv(pre + ".downsample.0.weight", inpl, planes, h, w, 1, stride, 0);
                b.bd = make_bn(pre + ".downsample.1", planes);
            }
            // z1 = relu(bn1(y1)): applied on load by its three consumers (never written), unless conv2 would then lose the LDS-DMA
            // kernel (conv_glds.hip cannot transform what it stages): there one bn_apply pass (read + write 2 bytes per element)
            // costs less than the register-staged convolution does (batch 256: layer 2 144 -> 95 + 29 us, layer 3 120 -> 72 + 14 us)
            // (the 64-channel layer keeps bn1 on load: conv_halo.hip transforms its register-staged halo for free, while the LDS-DMA kernel
            //  for that layer -- conv_c64p.hip -- would need the extra pass; its other launches take that kernel)
            b.fuse_z1 = planes == 64 || !conv_takes_glds(b.c2, (int)NB) || conv_takes_glds(b.c2, (int)NB, true);
            b.z1 = b.fuse_z1 ? 0 : alloc_act(NB * oh * ow * planes);
            b.out = alloc_act(NB * oh * ow * planes);
            blocks_.push_back(b);
            inpl = planes; h = oh; w = ow;
        }
    }

    // ---- velocity fusion + decoder (image.py:37-47,77-80) ----
    hcat_ = alloc_act(NB * h * w * 640);
    const int dc[4] = {640, 256, 128, 64};
    const char* bn_name[3] = {"deconv.0", "deconv.3", "deconv.6"};
    const char* ct_name[3] = {"deconv.1", "deconv.4", "deconv.7"};
    int dh = h, dw = w;
    for (int i = 0; i < 3; ++i) {
        Deconv& D = dec_[i];
        D.bn = make_bn(bn_name[i], dc[i]);
        D.w = add_tensor(std::string(ct_name[i]) + ".weight", kParam, {dc[i], dc[i + 1], 3, 3});
        D.bias = add_tensor(std::string(ct_name[i]) + ".bias", kParam, {dc[i + 1]});
        D.Cin = dc[i]; D.Cout = dc[i + 1]; D.H = dh; D.W = dw;
        D.u = alloc_act(NB * (2 * dh) * (2 * dw) * dc[i + 1]);
        if (act_bf16_) { D.wn = alloc(((size_t)dc[i] * dc[i + 1] * 9 + 1) / 2); D.wt = alloc(((size_t)dc[i] * dc[i + 1] * 9 + 1) / 2); }
        dh *= 2; dw *= 2;
        max_act = std::max(max_act, NB * dh * dw * (size_t)dc[i + 1]);
    }
    HH_ = dh; HW_ = dw;

    // ---- 4 command branches (image.py:54-60) ----
    for (int b = 0; b < 4; ++b) {
        const std::string pre = "location_pred." + std::to_string(b);
        head_bn_[b] = make_bn(pre + ".0", 64);
        head_w_[b] = add_tensor(pre + ".1.weight", kParam, {5, 64, 1, 1});
        head_b_[b] = add_tensor(pre + ".1.bias", kParam, {5});
        head_px_[b] = add_tensor(pre + ".2.pos_x", kBufferF32, {HH_ * HW_});
        head_py_[b] = add_tensor(pre + ".2.pos_y", kBufferF32, {HH_ * HW_});
    }
    head_stats_ = alloc(2 * 64);   // shared batch mean / invstd of the decoder output
    head_coef_ = alloc(3 * 64);
    head_partial_ = alloc(std::max((size_t)lbc_head_bwd_max_rows((int)NB) * 20 * 65, NB * 16 * 20 * 4));   // backward partial rows; forward slice scratch
    pred_all_ = alloc(NB * 40);
    rowstat_ = alloc(NB * 40);
    cmd_ = alloc(NB * 4);

    // ---- scratch: statistics partials, split-K slabs, gradient ping-pong buffers ----
    size_t pf = 1024 * 2 * 640;                                              // channel_reduce rows <= 1024
    pf = std::max(pf, (size_t)lbc_cdiv((long long)NB * (H0 / 2) * (W0 / 2), 128) * 128);   // stem epilogue
    for (const Block& b : blocks_) {
        pf = std::max(pf, (size_t)lbc_cdiv((long long)NB * b.c1.OH * b.c1.OW, 64) * 2 * b.c1.Cout);
    }
    for (int i = 0; i < 3; ++i)
        pf = std::max(pf, (size_t)4 * lbc_cdiv((long long)NB * dec_[i].H * dec_[i].W, 64) * 2 * dec_[i].Cout);
    partial_floats_ = pf;
    partial_ = alloc(pf);
    partial2_floats_ = (size_t)64 * 2 * 640;
    partial2_ = alloc(partial2_floats_);

    size_t wg = 0;
    auto wg_need = [&](int N, int OH, int OW, int CP, int Hq, int Wq, int CQ, int k, int s, int p) {
        WgradArgs a;
        memset(&a, 0, sizeof(a));
        a.N = N; a.OH = OH; a.OW = OW; a.CP = CP; a.H = Hq; a.W = Wq; a.CQ = CQ; a.KH = k; a.KW = k; a.S = s; a.P = p;
        a.bf16 = bf16_; a.act_bf16 = act_bf16_;      // the split policy depends on the kernel that will run
        return (size_t)lbc_wgrad_pick_split(a) * CP * k * k * CQ;
    };
    for (const Block& b : blocks_) {
        wg = std::max(wg, wg_need((int)NB, b.c1.OH, b.c1.OW, b.c1.Cout, b.c1.H, b.c1.W, b.c1.Cin, 3, b.c1.s, 1));
        wg = std::max(wg, wg_need((int)NB, b.c2.OH, b.c2.OW, b.c2.Cout, b.c2.H, b.c2.W, b.c2.Cin, 3, 1, 1));
        if (b.has_ds) wg = std::max(wg, wg_need((int)NB, b.ds.OH, b.ds.OW, b.ds.Cout, b.ds.H, b.ds.W, b.ds.Cin, 1, b.ds.s, 0));
    }
    for (int i = 0; i < 3; ++i)
        wg = std::max(wg, wg_need((int)NB, dec_[i].H, dec_[i].W, dec_[i].Cin, 2 * dec_[i].H, 2 * dec_[i].W, dec_[i].Cout, 3, 2, 1));
    wg = std::max(wg, (size_t)lbc_stem_wgrad_split((int)NB, H0, W0, Cin, bf16_) * 64 * 49 * Cin);
    if (defer_wgrad_) {
        // a stage's 3x3 / stride-1 convolutions share one shape: the slabs of their grouped launch, and one dY slot per convolution
        for (size_t li = 0; li < stage_first_block_.size(); ++li) {
            const size_t first = stage_first_block_[li], last = li + 1 < stage_first_block_.size() ? (size_t)stage_first_block_[li + 1] : blocks_.size();
            size_t members = 0, dy = 0;
            for (size_t bi = first; bi < last; ++bi) {
                const Block& b = blocks_[bi];
                const size_t slot = (((size_t)NB * b.c1.OH * b.c1.OW * b.c1.Cout + 1) / 2 + 63) / 64 * 64;
                members += 1 + (b.c1.s == 1 ? 1 : 0);
                dy += (b.has_ds ? 3 : 2) * slot;
            }
            const Conv& c = blocks_[last - 1].c2;
            WgradArgs a;
            memset(&a, 0, sizeof(a));
            a.N = (int)NB; a.OH = c.OH; a.OW = c.OW; a.CP = c.Cout; a.H = c.H; a.W = c.W; a.CQ = c.Cin; a.KH = 3; a.KW = 3; a.S = 1; a.P = 1;
            a.bf16 = 1; a.act_bf16 = 1; a.p = &a; a.q = &a;
            if (lbc_wgrad_tr_eligible(a)) {
                // (the members split into groups by their on-load transform: any group size up to the member count can occur)
                for (int n = 1; n <= (int)std::min(members, (size_t)kLbcWgradGroupMax); ++n)
                    wg = std::max(wg, (size_t)lbc_wgrad_tr_group_split(a, n) * n * c.Cout * 9 * c.Cin);
            }
            dy_arena_floats_ = std::max(dy_arena_floats_, dy);
        }
        dy_arena_ = alloc(dy_arena_floats_);
    }
    wg_partial_ = alloc(wg);
    wg_floats_ = wg;

    wt_ = alloc((size_t)640 * 512 * 9);
    gD_ = alloc_act(max_act); gE_ = alloc_act(max_act); gF_ = alloc_act(max_act); gG_ = alloc_act(max_act);
    g0_ = alloc_act(NB * (H0 / 2) * (W0 / 2) * 64);
}

std::vector<Net::ActInfo> Net::activations() const
{
    std::vector<ActInfo> v;
    const int ae = act_bf16_ ? 2 : 4;
    auto add = [&](const std::string& name, size_t off_floats, int H, int W, int C, int elem) {
        v.push_back(ActInfo{name, off_floats * sizeof(float), H, W, C, elem});
    };
    add("conv.conv1", y0_, d_.H / 2, d_.W / 2, 64, ae);                       // raw stem output (pre-BN)
    add("conv.maxpool", p0_, H1_, W1_, 64, ae);                               // maxpool(relu(bn1(.)))
    add("conv.maxpool.idx", idx_, H1_, W1_, 64, 1);                           // arg-max tap 3 r + s: input pixel (2 oy - 1 + r, 2 ox - 1 + s)
    int li = 0;
    for (size_t i = 0; i < blocks_.size(); ++i) {
        while (li + 1 < (int)stage_first_block_.size() && (int)i >= stage_first_block_[li + 1]) ++li;
        const Block& b = blocks_[i];
        const std::string p = "conv.layer" + std::to_string(li + 1) + "." + std::to_string((int)i - stage_first_block_[li]);
        add(p + ".conv1", b.c1.y, b.c1.OH, b.c1.OW, b.c1.Cout, ae);           // raw conv outputs (pre-BN)
        add(p + ".bn1.scale", b.b1.scale, 1, 1, b.b1.C, 4);                   // z1 = relu(conv1 * scale + shift) (batch statistics folded)
        add(p + ".bn1.shift", b.b1.shift, 1, 1, b.b1.C, 4);
        add(p + ".conv2", b.c2.y, b.c2.OH, b.c2.OW, b.c2.Cout, ae);
        if (b.has_ds) add(p + ".downsample.0", b.ds.y, b.ds.OH, b.ds.OW, b.ds.Cout, ae);
        add(p, b.out, b.c2.OH, b.c2.OW, b.c2.Cout, ae);                       // block output relu(bn2(.) + identity)
    }
    for (int i = 0; i < 3; ++i) add("deconv." + std::to_string(3 * i + 2), dec_[i].u, 2 * dec_[i].H, 2 * dec_[i].W, dec_[i].Cout, ae);
    return v;
}

Net::~Net()
{
    if (side_) (void)hipStreamDestroy(side_);
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_join_) (void)hipEventDestroy(ev_join_);
}

int Net::fork(hipStream_t s)
{
    if (!side_on_) return LBC_OK;
    if (hipEventRecord(ev_fork_, s) != hipSuccess || hipStreamWaitEvent(side_, ev_fork_, 0) != hipSuccess) {
        lbc_set_error("net.backward: side-stream fork failed");
        return LBC_ELAUNCH;
    }
    side_dirty_ = true;
    return LBC_OK;
}

int Net::join(hipStream_t s)
{
    if (!side_on_ || !side_dirty_) return LBC_OK;
    if (hipEventRecord(ev_join_, side_) != hipSuccess || hipStreamWaitEvent(s, ev_join_, 0) != hipSuccess) {
        lbc_set_error("net.backward: side-stream join failed");
        return LBC_ELAUNCH;
    }
    side_dirty_ = false;
    return LBC_OK;
}

int Net::check_bound(bool need_grads) const
{
    if (!ws_) { lbc_set_error("net: workspace not bound"); return LBC_ESTATE; }
    for (const TensorInfo& t : t_) {
        if (!t.ptr) { lbc_set_error("net: tensor %s not bound", t.name.c_str()); return LBC_ESTATE; }
        if (need_grads && t.kind == kParam && !t.grad) { lbc_set_error("net: gradient of %s not bound", t.name.c_str()); return LBC_ESTATE; }
    }
    return LBC_OK;
}

// ---------------------------------------------------------------------------------------
int Net::conv_fwd(const Conv& c, const float* x, int N, bool stats, int* rows, hipStream_t s, const BN* pre, const BN* post,
                  const float* resid, bool relu, float* out, float* stats_buf, size_t stats_cap)
{
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = P(c.w); a.y = out ? out : W(c.y);
    if (pre) { a.pre_scale = W(pre->scale); a.pre_shift = W(pre->shift); a.pre_relu = 1; }
    if (post) { a.post_scale = W(post->scale); a.post_shift = W(post->shift); }
    a.resid = resid; a.relu = relu ? 1 : 0;
    a.N = N; a.H = c.H; a.W = c.W; a.C = c.Cin;
    a.OH = c.OH; a.OW = c.OW; a.K = c.Cout;
    a.KH = c.k; a.KW = c.k; a.S = c.s; a.P = c.p;
    a.M = N * c.OH * c.OW; a.LH = c.OH; a.LW = c.OW; a.ostep = 1;
    a.bf16 = bf16_; a.act_bf16 = act_bf16_;
    if (act_bf16_) { a.w = W(c.wn); a.w_bf16 = 1; }
    split_scratch(a);
    const int cfg = lbc_igemm_pick_for(a, 0);
    *rows = lbc_igemm_rows(a, cfg);
    a.stats = stats ? (stats_buf ? stats_buf : W(partial_)) : nullptr;
    // (the row count is the kernel's, known before the launch: a caller-provided statistics buffer is checked BEFORE anything is written)
    LBC_REQUIRE(!a.stats || (size_t)*rows * 2 * (size_t)c.Cout <= (stats_buf ? stats_cap : partial_floats_),
                "net: %d statistics rows of %d channels exceed their buffer", *rows, c.Cout);
    return lbc_igemm_launch(a, 1, 0, cfg, s);
}

// Split-K scratch of a convolution launch (IgemmArgs::split_ws): the weight gradients' slab arena.  Its other users are the weight-gradient
// launches; with the deferred grouped weight gradients (the bf16 mode's default) they sit on the same stream as every forward / input-gradient
// launch, so the arena is free whenever one of those runs.  With a side stream for the weight gradients it is not: no split there.
void Net::split_scratch(IgemmArgs& a) const
{
    if (!defer_wgrad_ || side_on_ || !wg_floats_) return;
    a.split_ws = W(wg_partial_);
    a.split_ws_floats = (long long)wg_floats_;
}

// would a training forward of this convolution at batch N take conv_glds.hip when its input needs no transform on load?
bool Net::conv_takes_glds(const Conv& c, int N, bool with_prologue) const
{
    if (!act_bf16_) return false;
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.N = N; a.H = c.H; a.W = c.W; a.C = c.Cin; a.OH = c.OH; a.OW = c.OW; a.K = c.Cout;
    a.KH = c.k; a.KW = c.k; a.S = c.s; a.P = c.p; a.M = N * c.OH * c.OW; a.LH = c.OH; a.LW = c.OW; a.ostep = 1;
    a.bf16 = 1; a.act_bf16 = 1; a.w_bf16 = 1;
    if (with_prologue) { a.pre_scale = reinterpret_cast<const float*>(this); a.pre_shift = a.pre_scale; a.pre_relu = 1; }   // (only tested for null)
    return lbc_igemm_pick_for(a, 0) >= kLbcCfgGlds;        // conv_glds.hip or conv_hdma.hip
}

int Net::weight_prep(hipStream_t s)
{
    WeightPrepArgs a;
    memset(&a, 0, sizeof(a));
    int n = 0, tiles = 0;
    auto add = [&](const float* w, size_t wn, size_t wt, int A, int T, int B) {
        if (n >= WeightPrepArgs::kMax) return false;
        WeightPrepItem& it = a.item[n++];
        it.w = w; it.wn = W(wn); it.wt = W(wt); it.A = A; it.T = T; it.B = B; it.tile_begin = tiles;
        tiles += lbc_weight_prep_tiles(A, T, B);
        return true;
    };
    bool ok = true;
    for (const Block& b : blocks_) {
        ok = ok && add(P(b.c1.w), b.c1.wn, b.c1.wt, b.c1.Cout, 9, b.c1.Cin);
        ok = ok && add(P(b.c2.w), b.c2.wn, b.c2.wt, b.c2.Cout, 9, b.c2.Cin);
        if (b.has_ds) ok = ok && add(P(b.ds.w), b.ds.wn, b.ds.wt, b.ds.Cout, 1, b.ds.Cin);
    }
    for (int i = 0; i < 3; ++i) ok = ok && add(P(dec_[i].w), dec_[i].wn, dec_[i].wt, dec_[i].Cin, 9, dec_[i].Cout);
    LBC_REQUIRE(ok, "net: more than %d convolution weights", WeightPrepArgs::kMax);
    a.count = n; a.tiles = tiles;
    return lbc_weight_prep(a, s);
}

// eval mode: scale / shift / mean / invstd of all BatchNorms from the running statistics, one launch
int Net::bn_eval_prep(hipStream_t s)
{
    BnEvalArgs a;
    memset(&a, 0, sizeof(a));
    int n = 0;
    bool ok = true;
    auto add = [&](const BN& bn) {
        if (n >= BnEvalArgs::kMax) { ok = false; return; }
        BnEvalItem& it = a.item[n++];
        it.gamma = P(bn.g); it.beta = P(bn.b); it.running_mean = P(bn.rm); it.running_var = P(bn.rv);
        it.scale = W(bn.scale); it.shift = W(bn.shift); it.mean = W(bn.mean); it.invstd = W(bn.invstd);
        it.C = bn.C;
    };
    add(stem_bn_);
    for (const Block& b : blocks_) { add(b.b1); add(b.b2); if (b.has_ds) add(b.bd); }
    for (int i = 0; i < 3; ++i) add(dec_[i].bn);
    for (int b = 0; b < 4; ++b) add(head_bn_[b]);
    LBC_REQUIRE(ok, "net: more than %d BatchNorms", BnEvalArgs::kMax);
    a.count = n; a.eps = kBnEps;
    return lbc_bn_eval_prep(a, s);
}

int Net::set_sync_bn(lbc_allreduce_fn fn, void* ctx, int world, float* buf, int buf_floats)
{
    if (!fn) { sync_fn_ = nullptr; sync_ctx_ = nullptr; sync_world_ = 1; sync_buf_ = nullptr; return LBC_OK; }
    LBC_REQUIRE(world >= 1 && buf && buf_floats >= kSyncFloats, "net.set_sync_bn: world_size >= 1 and a buffer of >= %d floats",
                kSyncFloats);
    sync_fn_ = fn; sync_ctx_ = ctx; sync_world_ = world; sync_buf_ = buf;
    return LBC_OK;
}

int Net::sync_rows(const float*& part, int& rows, int width, int n_local, hipStream_t s, float* local_lo, float* local_hi)
{
    if (!sync_fn_) return LBC_OK;
    LBC_REQUIRE(width + 1 <= kSyncFloats, "net: %d sums do not fit the SyncBN buffer", width);
    if (rows > kLbcFinalizeRows) {      // (one thread per column walking thousands of rows would be the longest kernel of the layer)
        LBC_TRY(lbc_partial_reduce(part, rows, width, W(partial2_), 64, s));
        part = W(partial2_); rows = 64;
    }
    // element `width` = this rank's batch size: its sum over the ranks gives every finalize the global element count, so no rank
    // has to know (or verify with a collective of its own) what batch the others run
    LBC_TRY(lbc_partial_reduce(part, rows, width, sync_buf_, 1, s, local_lo, local_hi, (float)n_local));
    if (sync_fn_(sync_ctx_, sync_buf_, width + 1, s) != 0) {
        lbc_set_error("net: the SyncBN all-reduce callback failed");
        return LBC_ELAUNCH;
    }
    part = sync_buf_; rows = 1;
    return LBC_OK;
}

int Net::bn_finalize(const BN& bn, int rows, long long count, int n_local, int train, hipStream_t s, bool update_running, const float* synced)
{
    const float* part = W(partial_);
    const float* nsum = nullptr;
    if (train && synced) {
        part = synced; rows = 1; nsum = synced + 2 * bn.C;
    } else if (train && sync_fn_) {
        LBC_TRY(sync_rows(part, rows, 2 * bn.C, n_local, s));
        nsum = part + 2 * bn.C;
    } else if (train && rows > kLbcFinalizeRows) {
        LBC_TRY(lbc_partial_reduce(W(partial_), rows, 2 * bn.C, W(partial2_), 64, s));
        part = W(partial2_);
        rows = 64;
    }
    BnFinalizeArgs f = fin_args(bn, part, rows, count, update_running);
    f.nsum = nsum; f.n_local = n_local; f.train = train;
    if (!train) { f.running_mean = P(bn.rm); f.running_var = P(bn.rv); f.num_batches_tracked = nullptr; }
    return lbc_bn_finalize(f, s);
}

BnFinalizeArgs Net::fin_args(const BN& bn, const float* part, int rows, long long count, bool update_running) const
{
    BnFinalizeArgs f;
    memset(&f, 0, sizeof(f));
    f.partial = part; f.rows = rows; f.C = bn.C; f.count = count;
    f.gamma = P(bn.g); f.beta = P(bn.b);
    f.running_mean = update_running ? P(bn.rm) : nullptr;
    f.running_var = update_running ? P(bn.rv) : nullptr;
    f.num_batches_tracked = update_running ? static_cast<long long*>(t_[bn.nbt].ptr) : nullptr;
    f.momentum = kBnMomentum; f.eps = kBnEps; f.train = 1;
    f.scale = W(bn.scale); f.shift = W(bn.shift); f.save_mean = W(bn.mean); f.save_invstd = W(bn.invstd);
    return f;
}

int Net::forward(int N, int train, const void* image, int image_u8, const float* velocity, const float* command, float* pred_sel,
                 float* pred_all, hipStream_t s)
{
    LBC_REQUIRE(N >= 1 && N <= d_.max_batch, "net.forward: batch %d outside [1,%d]", N, d_.max_batch);
    LBC_REQUIRE(image && velocity && command && pred_all, "net.forward: null argument");
    LBC_TRY(check_bound(false));
    lastN_ = N; last_train_ = train; ++generation_;
    const int H0 = d_.H, W0 = d_.W, Cin = d_.in_channels;
    const bool tr = train != 0;
    int rows = 0;

    LBC_TRY(lbc_copy_f32(command, W(cmd_), 4 * N, s));    // (kernels, not memcpy nodes: a captured forward is kernel nodes only)
    // image.py:71 / common.py:108-109: (x - mean) / std, fused into the NCHW -> padded NHWC repack
    NormConst nc;
    memset(&nc, 0, sizeof(nc));
    nc.enabled = d_.normalize;
    const float m3[3] = {0.485f, 0.456f, 0.406f}, s3[3] = {0.229f, 0.224f, 0.225f};
    for (int i = 0; i < 3; ++i) { nc.mean[i] = m3[i]; nc.stdv[i] = s3[i]; }
    // bf16 modes: the padded image is bf16 too
    if (image_u8) LBC_TRY(lbc_prep_input_u8(static_cast<const unsigned char*>(image), W(xp_), bf16_, N, Cin, H0, W0, nc, s));
    else          LBC_TRY(lbc_prep_input(static_cast<const float*>(image), W(xp_), bf16_, N, Cin, H0, W0, nc, s));
    // derived from the weights alone: skipped when the caller declared them frozen and an eval-mode forward has derived them already
    const bool reuse = frozen_ && derived_valid_ && !tr;
    if (act_bf16_ && !reuse) LBC_TRY(weight_prep(s));   // the caller's optimizer may have stepped: refresh the bf16 weight copies
    if (!tr && !reuse) LBC_TRY(bn_eval_prep(s));        // eval: every BatchNorm's affine from its running statistics, one launch
    derived_valid_ = frozen_ && !tr;                    // (a training forward moves the running statistics: the eval-mode affines are stale after it)

    // resnet.py:148-152: conv1 -> bn1 -> relu -> maxpool
    StemArgs st;
    st.xp = W(xp_); st.xp_bf16 = bf16_; st.w = P(stem_w_); st.y = W(y0_); st.stats = tr ? W(partial_) : nullptr;
    st.N = N; st.H = H0; st.W = W0; st.Cin = Cin; st.act_bf16 = act_bf16_; st.bf16 = bf16_;
    LBC_TRY(lbc_stem_fwd(st, s));
    if (tr) LBC_TRY(bn_finalize(stem_bn_, lbc_stem_rows(st), (long long)N * (H0 / 2) * (W0 / 2), N, train, s));
    PoolFwdArgs pf;
    pf.y = W(y0_); pf.scale = W(stem_bn_.scale); pf.shift = W(stem_bn_.shift); pf.p = W(p0_);
    pf.idx = reinterpret_cast<unsigned char*>(W(idx_));
    pf.N = N; pf.H = H0 / 2; pf.W = W0 / 2; pf.C = 64; pf.act_bf16 = act_bf16_;
    LBC_TRY(lbc_bn_relu_maxpool_fwd(pf, s));

    // resnet.py:38-54 BasicBlock.forward x 8/16
    const float* x = W(p0_);
    for (Block& b : blocks_) {
        const long long pix = (long long)N * b.c1.OH * b.c1.OW;
        if (!tr) {
            // eval: BatchNorms folded into the convolution epilogues -- conv1 writes z1 = relu(bn1(.)), the downsample
            // writes bn_d(.), conv2 writes relu(bn2(.) + identity) straight into the block output: no BatchNorm pass at all
            LBC_TRY(conv_fwd(b.c1, x, N, false, &rows, s, nullptr, &b.b1, nullptr, true));
            const float* identity = x;
            if (b.has_ds) {
                LBC_TRY(conv_fwd(b.ds, x, N, false, &rows, s, nullptr, &b.bd, nullptr, false));
                identity = W(b.ds.y);
            }
            LBC_TRY(conv_fwd(b.c2, W(b.c1.y), N, false, &rows, s, nullptr, &b.b2, identity, true, W(b.out)));
            x = W(b.out);
            continue;
        }
        // Where a BatchNorm's partial rows are few (small per-GPU batches) the elementwise pass that consumes its coefficients also
        // does its finalize (BnApplyArgs::fold): bn1 in the z1 pass, bn2 and the downsample's BatchNorm in the block-output pass
        LBC_TRY(conv_fwd(b.c1, x, N, tr, &rows, s));
        const bool fold1 = !b.fuse_z1 && can_fold(rows, b.b1.C);
        if (!fold1) LBC_TRY(bn_finalize(b.b1, rows, pix, N, train, s));
        BnApplyArgs ap;
        if (b.fuse_z1) {
            LBC_TRY(conv_fwd(b.c2, W(b.c1.y), N, tr, &rows, s, &b.b1));
        } else {
            memset(&ap, 0, sizeof(ap));
            ap.x = W(b.c1.y); ap.y = W(b.z1); ap.pixels = pix; ap.C = b.c1.Cout;
            ap.scale = W(b.b1.scale); ap.shift = W(b.b1.shift); ap.relu = 1; ap.act_bf16 = act_bf16_;
            if (fold1) { ap.fold = 1; ap.fin = fin_args(b.b1, W(partial_), rows, pix); }
            LBC_TRY(lbc_bn_apply(ap, s));
            LBC_TRY(conv_fwd(b.c2, W(b.z1), N, tr, &rows, s));
        }
        // (the downsample's rows must not overwrite conv2's before the folded pass has read them: they go to partial2_ -- idle while
        //  no row count needs pre-reduction -- when they are certain to fit; otherwise bn2 is finalized by its own launch as before)
        const bool fold2 = can_fold(rows, b.b2.C) && (!b.has_ds || (size_t)lbc_cdiv(pix, 64) * 2 * b.bd.C <= partial2_floats_);
        const int rows2 = rows;
        if (!fold2) LBC_TRY(bn_finalize(b.b2, rows, pix, N, train, s));
        memset(&ap, 0, sizeof(ap));
        ap.x = W(b.c2.y); ap.y = W(b.out); ap.pixels = pix; ap.C = b.c2.Cout;
        ap.scale = W(b.b2.scale); ap.shift = W(b.b2.shift); ap.relu = 1; ap.act_bf16 = act_bf16_;
        if (fold2) { ap.fold = 1; ap.fin = fin_args(b.b2, W(partial_), rows2, pix); }
        if (b.has_ds) {
            float* dsp = fold2 ? W(partial2_) : W(partial_);
            LBC_TRY(conv_fwd(b.ds, x, N, tr, &rows, s, nullptr, nullptr, nullptr, false, nullptr, dsp, fold2 ? partial2_floats_ : partial_floats_));
            if (fold2 && can_fold(rows, b.bd.C)) { ap.rfold = 1; ap.rfin = fin_args(b.bd, dsp, rows, pix); }
            else if (fold2) LBC_TRY(lbc_bn_finalize(fin_args(b.bd, dsp, rows, pix), s));     // (can_fold implies local statistics and <= 1024 rows)
            else LBC_TRY(bn_finalize(b.bd, rows, pix, N, train, s));
            ap.resid = W(b.ds.y); ap.rscale = W(b.bd.scale); ap.rshift = W(b.bd.shift);
        } else {
            ap.resid = x;
        }
        LBC_TRY(lbc_bn_apply(ap, s));
        x = W(b.out);
    }

    // image.py:77-79 velocity late fusion; image.py:37-47 decoder (BN -> ConvT -> ReLU) x 3
    const Block& last = blocks_.back();
    const int th = last.c2.OH, tw = last.c2.OW;
    LBC_TRY(lbc_concat_velocity(x, velocity, W(hcat_), N, th * tw, 512, 128, act_bf16_, s));
    if (tr) {
        ChanReduceArgs cr;
        memset(&cr, 0, sizeof(cr));
        cr.x = W(hcat_); cr.partial = W(partial_); cr.pixels = (long long)N * th * tw; cr.C = 640; cr.act_bf16 = act_bf16_;
        cr.max_rows = fold_rows_for(cr.pixels, 640);
        LBC_TRY(lbc_chan_reduce(cr, 0, s));
        rows = lbc_chan_reduce_rows(cr.pixels, 640, cr.max_rows);
    }
    // the finalize of stage i's BatchNorm is issued in iteration i: folded into the bn_apply pass over the stage's input where there is one
    int pend_rows = rows;
    long long pend_count = (long long)N * th * tw;
    const float* din = W(hcat_);
    for (int i = 0; i < 3; ++i) {
        Deconv& D = dec_[i];
        IgemmArgs a;
        memset(&a, 0, sizeof(a));
        a.x = din; a.w = P(D.w); a.y = W(D.u); a.bias = P(D.bias); a.relu = 1;
        a.pre_scale = W(D.bn.scale); a.pre_shift = W(D.bn.shift);
        a.N = N; a.H = D.H; a.W = D.W; a.C = D.Cin;
        a.OH = 2 * D.H; a.OW = 2 * D.W; a.K = D.Cout;
        a.KH = 3; a.KW = 3; a.S = 2; a.P = 1;
        a.LH = D.H; a.LW = D.W; a.ostep = 2;
        a.M = N * D.H * D.W;
        a.stats = tr ? W(partial_) : nullptr;
        a.act_bf16 = act_bf16_;
        int wmajor = 0;
        if (act_bf16_) {
            a.w = W(D.wt); a.w_bf16 = 1; a.bf16 = 1; wmajor = 1;
        } else if (bf16_) {
            // w[Cin][T][Cout] -> wt[Cout][T][Cin]: depth-contiguous for the bf16 tiles
            LBC_TRY(lbc_weight_transpose(P(D.w), W(wt_), D.Cin, 9, D.Cout, s));
            a.w = W(wt_); a.bf16 = 1; wmajor = 1;
        }
        a.nphase = 4;                       // the four output-parity phases in one launch; statistics rows ph * per + tile
        int cfg = lbc_igemm_pick(a.M, a.K);
        if (act_bf16_) {
            // the LDS-DMA kernel cannot apply the BatchNorm on load: where it would take the launch otherwise, one bn_apply pass
            // over the (small) decoder input -- into a gradient ping-pong buffer, idle during the forward -- buys it
            IgemmArgs b = a;
            b.pre_scale = nullptr; b.pre_shift = nullptr; b.x = W(gF_);
            const int c2 = lbc_igemm_pick_for(b, 1);
            // (the 64-channel last stage too: 244 -> 167 + 25 us at 256 images, the teacher's 152 -> 112 + 15)
            if (c2 >= kLbcCfgGlds) {
                BnApplyArgs ap;
                memset(&ap, 0, sizeof(ap));
                ap.x = din; ap.y = W(gF_); ap.pixels = (long long)N * D.H * D.W; ap.C = D.Cin;
                ap.scale = W(D.bn.scale); ap.shift = W(D.bn.shift); ap.relu = 0; ap.act_bf16 = 1;
                if (tr && can_fold(pend_rows, D.Cin)) { ap.fold = 1; ap.fin = fin_args(D.bn, W(partial_), pend_rows, pend_count); pend_rows = 0; }
                else if (tr) { LBC_TRY(bn_finalize(D.bn, pend_rows, pend_count, N, train, s)); pend_rows = 0; }
                LBC_TRY(lbc_bn_apply(ap, s));
                a = b; cfg = c2;
            }
        }
        if (tr && pend_rows > 0) LBC_TRY(bn_finalize(D.bn, pend_rows, pend_count, N, train, s));
        const int per = lbc_igemm_rows(a, cfg);
        LBC_TRY(lbc_igemm_launch(a, wmajor, 1, cfg, s));
        const long long opix = (long long)N * 4 * D.H * D.W;
        if (!tr) {
            // eval: statistics come from bn_eval_prep
        } else if (i < 2) {
            pend_rows = 4 * per; pend_count = opix;
        } else {
            // image.py:54-60: the four branch BatchNorms see the same tensor -> same batch statistics
            const float* part = W(partial_);
            int prow = 4 * per;
            // ONE finalize: the head kernels read the batch statistics of branch 0's slot (every branch folds its own gamma / beta
            // into the projection); the other three branches only need their running statistics and counters moved along
            LBC_TRY(sync_rows(part, prow, 2 * 64, N, s));     // SyncBN: one all-reduce
            {
                BnFinalizeArgs f = fin_args(head_bn_[0], part, prow, opix);
                if (sync_fn_) { f.nsum = part + 2 * 64; f.n_local = N; }
                else if (prow > kLbcFinalizeRows) {
                    LBC_TRY(lbc_partial_reduce(W(partial_), prow, 2 * 64, W(partial2_), 64, s));
                    f.partial = W(partial2_); f.rows = 64;
                }
                for (int b = 1; b < 4; ++b) {
                    f.more_running_mean[b - 1] = P(head_bn_[b].rm); f.more_running_var[b - 1] = P(head_bn_[b].rv);
                    f.more_num_batches_tracked[b - 1] = static_cast<long long*>(t_[head_bn_[b].nbt].ptr);
                }
                LBC_TRY(lbc_bn_finalize(f, s));
            }
        }
        din = W(D.u);
    }

    // image.py:82-84 + common.py:29-35,136-152
    HeadArgs ha;
    memset(&ha, 0, sizeof(ha));
    ha.h = W(dec_[2].u);
    for (int b = 0; b < 4; ++b) {
        ha.mean[b] = W(head_bn_[tr ? 0 : b].mean);
        ha.invstd[b] = W(head_bn_[tr ? 0 : b].invstd);
        ha.gamma[b] = P(head_bn_[b].g); ha.beta[b] = P(head_bn_[b].b);
        ha.w[b] = P(head_w_[b]); ha.bias[b] = P(head_b_[b]);
        ha.pos_x[b] = P(head_px_[b]); ha.pos_y[b] = P(head_py_[b]);
    }
    ha.cmd = W(cmd_);
    ha.pred_all = W(pred_all_); ha.pred_sel = pred_sel; ha.rowstat = W(rowstat_);
    ha.N = N; ha.OH = HH_; ha.OW = HW_; ha.act_bf16 = act_bf16_;
    ha.scratch = W(head_partial_);       // max_batch * 20 * 65 floats >= N * 16 * 20 * 4
    LBC_TRY(lbc_head_fwd(ha, s));
    return lbc_copy_f32(W(pred_all_), pred_all, 40 * N, s);
}

// ---------------------------------------------------------------------------------------
// BatchNorm backward finalize.  SyncBN: dgamma / dbeta stay this rank's sums (the gradient all-reduce completes them like
// every other parameter gradient) -- they are the one-row reduction itself, written on its way into the exchange buffer;
// the coefficients k1 = sum(g)/n, k2 = sum(g xhat)/n of the input gradient come from the global sums.
int Net::bn_bwd_finalize(BnBwdFinalizeArgs f, hipStream_t s)
{
    if (sync_fn_) {
        LBC_TRY(sync_rows(f.partial, f.rows, 2 * f.C, lastN_, s, f.dbeta, f.dgamma));
        f.nsum = f.partial + 2 * f.C; f.n_local = lastN_;
        f.dgamma = nullptr; f.dbeta = nullptr;
    }
    return lbc_bn_bwd_finalize(f, s);
}

// BatchNorm backward: reduce (sum g, sum g*xhat) -> dgamma/dbeta + coefficients -> dx
int Net::bn_backward(const BN& bn, const float* dz, const float* mask, float* g_out, const float* x, long long pixels,
                     float* dx, int Cout, hipStream_t s, const BN* mask_bn, bool join_before_apply, int reduced_rows)
{
    int rows = reduced_rows;
    if (reduced_rows <= 0) {
        ChanReduceArgs r;
        memset(&r, 0, sizeof(r));
        r.x = x; r.dz = dz; r.mask = mask; r.g_out = g_out; r.mean = W(bn.mean); r.invstd = W(bn.invstd);
        if (mask_bn) { r.mask_scale = W(mask_bn->scale); r.mask_shift = W(mask_bn->shift); }
        r.partial = W(partial_); r.pixels = pixels; r.C = bn.C; r.act_bf16 = act_bf16_;
        r.max_rows = fold_rows_for(pixels, bn.C);      // small tensors: few enough rows for the apply pass to fold the finalize
        LBC_TRY(lbc_chan_reduce(r, 1, s));
        rows = lbc_chan_reduce_rows(pixels, bn.C, r.max_rows);
    } else {
        mask = nullptr;          // dz is the masked gradient already
        g_out = const_cast<float*>(dz);
    }
    const float* part = W(partial_);
    if (rows > kLbcFinalizeRows) {
        LBC_TRY(lbc_partial_reduce(W(partial_), rows, 2 * bn.C, W(partial2_), 64, s));
        part = W(partial2_); rows = 64;
    }
    BnBwdFinalizeArgs f;
    memset(&f, 0, sizeof(f));
    f.partial = part; f.rows = rows; f.C = bn.C; f.count = pixels;
    f.gamma = P(bn.g); f.mean = W(bn.mean); f.invstd = W(bn.invstd); f.train = 1;
    f.dgamma = G(bn.g); f.dbeta = G(bn.b);
    f.coefA = W(bn.cA); f.coefB = W(bn.cB); f.coefD = W(bn.cD);
    // few rows: the apply pass does the finalize itself (BnBwdApplyArgs::fold), one launch less per BatchNorm
    const bool fold = can_fold(rows, bn.C) && (g_out || !mask);
    if (!fold) LBC_TRY(bn_bwd_finalize(f, s));
    BnBwdApplyArgs ap;
    memset(&ap, 0, sizeof(ap));
    if (fold) { ap.fold = 1; ap.fin = f; }
    ap.g = g_out ? g_out : dz; ap.mask = g_out ? nullptr : mask; ap.x = x;
    ap.coefA = W(bn.cA); ap.coefB = W(bn.cB); ap.coefD = W(bn.cD);
    ap.mean = W(bn.mean); ap.invstd = W(bn.invstd);
    ap.dx = dx; ap.pixels = pixels; ap.C = bn.C; ap.Cout = Cout; ap.act_bf16 = act_bf16_;
    if (join_before_apply) LBC_TRY(join(s));     // dx is still being read by a weight gradient on the side stream
    return lbc_bn_bwd_apply(ap, s);
}

float* Net::dy_slot(long long elems)
{
    const size_t floats = ((size_t)(elems + 1) / 2 + 63) / 64 * 64;      // bf16 tensors (defer_wgrad_ implies act_bf16_)
    if (dy_used_ + floats > dy_arena_floats_) return nullptr;
    float* p = W(dy_arena_) + dy_used_;
    dy_used_ += floats;
    return p;
}

// The stage's pending weight gradients: same-shaped 3x3 / stride-1 convolutions in grouped launches, the rest (the stage's first
// stride-2 convolution, its 1x1 downsample) one by one.
int Net::flush_wgrads(int N, hipStream_t s)
{
    std::vector<char> done(pending_.size(), 0);
    for (size_t i = 0; i < pending_.size(); ++i) {
        if (done[i]) continue;
        const PendingWgrad& pi = pending_[i];
        const Conv& c = *pi.c;
        WgradArgs a;
        memset(&a, 0, sizeof(a));
        a.p = pi.dy; a.q = pi.x;
        if (pi.pre) { a.q_scale = W(pi.pre->scale); a.q_shift = W(pi.pre->shift); a.q_relu = 1; }
        a.bf16 = bf16_; a.act_bf16 = act_bf16_;
        a.N = N; a.OH = c.OH; a.OW = c.OW; a.CP = c.Cout; a.H = c.H; a.W = c.W; a.CQ = c.Cin;
        a.KH = c.k; a.KW = c.k; a.S = c.s; a.P = c.p;
        if (!lbc_wgrad_tr_eligible(a)) {
            LBC_TRY(conv_wgrad_pre(c, pi.x, pi.pre, pi.dy, N, s));
            done[i] = 1;
            continue;
        }
        WgradGroup g;
        memset(&g, 0, sizeof(g));
        float* grads[kLbcWgradGroupMax];
        for (size_t j = i; j < pending_.size() && g.n < kLbcWgradGroupMax; ++j) {
            const PendingWgrad& pj = pending_[j];
            const Conv& cj = *pj.c;
            if (done[j] || cj.Cout != c.Cout || cj.Cin != c.Cin || cj.H != c.H || cj.W != c.W || cj.k != c.k || cj.s != c.s || cj.p != c.p ||
                (pj.pre != nullptr) != (pi.pre != nullptr)) continue;
            g.p[g.n] = pj.dy; g.q[g.n] = pj.x;
            if (pj.pre) { g.q_scale[g.n] = W(pj.pre->scale); g.q_shift[g.n] = W(pj.pre->shift); }
            grads[g.n] = G(cj.w);
            ++g.n;
            done[j] = 1;
        }
        a.nsplit = lbc_wgrad_tr_group_split(a, g.n);
        const size_t count = (size_t)c.Cout * 9 * c.Cin;
        // (the slab arena was planned at max_batch; the split policy is not monotone in the batch: never more slabs than it holds)
        if ((size_t)a.nsplit * g.n * count > wg_floats_) a.nsplit = (int)std::max<size_t>(1, wg_floats_ / (g.n * count));
        for (int m = 0; m < g.n; ++m) g.out[m] = a.nsplit == 1 ? grads[m] : W(wg_partial_) + (size_t)m * a.nsplit * count;
        LBC_TRY(lbc_wgrad_tr_group_launch(a, g, s));
        if (a.nsplit > 1) LBC_TRY(lbc_splitk_reduce_group(W(wg_partial_), a.nsplit, (long long)count, g.n, grads, s));
    }
    pending_.clear();
    dy_used_ = 0;
    return LBC_OK;
}

int Net::conv_wgrad(const Conv& c, const float* x, const float* dy, int N, hipStream_t s)
{
    return conv_wgrad_pre(c, x, nullptr, dy, N, s);
}

int Net::conv_wgrad_pre(const Conv& c, const float* x, const BN* pre, const float* dy, int N, hipStream_t s)
{
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.p = dy; a.q = x; a.partial = W(wg_partial_);
    if (pre) { a.q_scale = W(pre->scale); a.q_shift = W(pre->shift); a.q_relu = 1; }
    a.bf16 = bf16_; a.act_bf16 = act_bf16_;
    a.N = N; a.OH = c.OH; a.OW = c.OW; a.CP = c.Cout;
    a.H = c.H; a.W = c.W; a.CQ = c.Cin;
    a.KH = c.k; a.KW = c.k; a.S = c.s; a.P = c.p;
    a.nsplit = lbc_wgrad_pick_split(a);
    {   // (planned at max_batch; kernel choice and split policy depend on the batch: never more slabs than the arena holds)
        const size_t count = (size_t)c.Cout * c.k * c.k * c.Cin;
        if ((size_t)a.nsplit * count > wg_floats_) a.nsplit = (int)std::max<size_t>(1, wg_floats_ / count);
    }
    if (a.nsplit == 1) { a.partial = G(c.w); return lbc_wgrad_launch(a, s); }   // a single slab is the gradient itself
    LBC_TRY(lbc_wgrad_launch(a, s));
    return lbc_splitk_reduce(a.partial, a.nsplit, (long long)c.Cout * c.k * c.k * c.Cin, G(c.w), 0.f, s);
}

// dx[N,H,W,Cin] = dgrad(dy) (+ resid).  For the 1x1/2 downsample only the even-even phase is touched.
int Net::conv_dgrad(const Conv& c, const float* dy, const float* resid, float* dx, int N, hipStream_t s, const BN* bnb,
                    const float* bnb_y, int* fused_rows, const float* bnb_mask)
{
    if (fused_rows) *fused_rows = 0;
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = dy; a.w = P(c.w); a.y = dx; a.resid = resid;
    a.N = N; a.H = c.OH; a.W = c.OW; a.C = c.Cout;
    a.OH = c.H; a.OW = c.W; a.K = c.Cin;
    a.KH = c.k; a.KW = c.k; a.S = c.s; a.P = c.p;
    int wmajor = 0;
    a.bf16 = bf16_; a.act_bf16 = act_bf16_;
    if (act_bf16_) {
        a.w = W(c.wt); a.w_bf16 = 1; wmajor = 1;
    } else if (dgrad_wt_ && (c.k == 3 || bf16_)) {
        // w[Cout][T][Cin] -> wt[Cin][T][Cout]: output channel (Cin) major, gathered channel (Cout) contiguous
        LBC_TRY(lbc_weight_transpose(P(c.w), W(wt_), c.Cout, c.k * c.k, c.Cin, s));
        a.w = W(wt_);
        wmajor = 1;
    }
    if (c.s == 1) {
        a.LH = c.H; a.LW = c.W; a.ostep = 1;
        a.M = N * c.H * c.W;
        split_scratch(a);
        const int cfg = wmajor ? lbc_igemm_pick_for(a, 1) : lbc_igemm_pick(a.M, a.K);
        if (bnb && bnb_y && fused_rows &&
            (bnb_mask ? lbc_igemm_fuses_bn_bwd_masked(a, wmajor, 1, cfg) : lbc_igemm_fuses_bn_bwd(a, wmajor, 1, cfg))) {
            a.bnb_y = bnb_y; a.bnb_scale = W(bnb->scale); a.bnb_shift = W(bnb->shift);
            a.bnb_mean = W(bnb->mean); a.bnb_invstd = W(bnb->invstd);
            a.bnb_mask = bnb_mask;
            a.stats = W(partial_);
            *fused_rows = lbc_igemm_rows(a, cfg);
            LBC_REQUIRE((size_t)*fused_rows * 2 * (size_t)a.K <= partial_floats_, "net: %d rows of BatchNorm-backward sums exceed their buffer", *fused_rows);
        }
        return lbc_igemm_launch(a, wmajor, 1, cfg, s);
    }
    a.LH = c.H / 2; a.LW = c.W / 2; a.ostep = 2;
    a.M = N * a.LH * a.LW;
    a.nphase = c.k == 1 ? 1 : 4;            // 1x1: only the even-even phase; 3x3: all four parity phases in one launch
    const int cfg = (wmajor && act_bf16_) ? lbc_igemm_pick_for(a, 1) : lbc_igemm_pick(a.M, a.K);
    return lbc_igemm_launch(a, wmajor, 1, cfg, s);
}

// D: gradient wrt the block output (consumed; becomes the masked gradient); on return D points at the
// gradient wrt the block input.
int Net::block_backward(Block& b, float*& D, float*& Gbuf, float* E, float* F, hipStream_t s)
{
    const int N = lastN_;
    const long long pix = (long long)N * b.c1.OH * b.c1.OW;
    const float* xin = (&b == &blocks_.front()) ? W(p0_) : W((&b - 1)->out);
    if (defer_wgrad_) {
        // every dY in a slot of its own until the stage's grouped weight-gradient launches have read it (flush_wgrads)
        float* E2 = dy_slot(pix * b.b2.C);
        float* E1 = dy_slot(pix * b.b1.C);
        LBC_REQUIRE(E1 && E2, "net.backward: dY arena exhausted");
        // (D may arrive masked and reduced: the input gradient that produced it -- conv1 of the block behind this one -- did bn2's reduce
        //  pass in its epilogue, bwd_pre_rows_ rows of sums in partial_)
        const int pre_rows = bwd_pre_rows_;
        bwd_pre_rows_ = 0;
        // (partial_ is LIVE across the block boundary then: nothing may write it between the producing conv_dgrad of the block behind this
        //  one and the bn_backward below, and the sums were taken over THIS block's bn2 extent -- checked, not assumed)
        LBC_REQUIRE(pre_rows == 0 || (bwd_pre_pix_ == pix && bwd_pre_C_ == b.b2.C), "net.backward: pre-reduced rows of %lld x %d do not match this block's bn2 (%lld x %d)",
                    bwd_pre_pix_, bwd_pre_C_, pix, b.b2.C);
        LBC_TRY(bn_backward(b.b2, D, W(b.out), D, W(b.c2.y), pix, E2, b.b2.C, s, nullptr, false, pre_rows));   // E2 = dY2
        int fr = 0;
        LBC_TRY(conv_dgrad(b.c2, E2, nullptr, F, N, s, &b.b1, W(b.c1.y), &fr));                       // F = dZ1 (masked when fr > 0)
        if (b.fuse_z1) {
            pending_.push_back({&b.c2, W(b.c1.y), &b.b1, E2});
            LBC_TRY(bn_backward(b.b1, F, W(b.c1.y), F, W(b.c1.y), pix, E1, b.b1.C, s, &b.b1, false, fr));   // E1 = dY1
        } else {
            pending_.push_back({&b.c2, W(b.z1), nullptr, E2});
            LBC_TRY(bn_backward(b.b1, F, W(b.z1), F, W(b.c1.y), pix, E1, b.b1.C, s, nullptr, false, fr));
        }
        pending_.push_back({&b.c1, xin, nullptr, E1});
        if (!b.has_ds) {
            // G = dgrad + identity gradient = the gradient wrt the PREVIOUS block's output relu(bn2(y2) + identity) (resnet.py:51-54): where
            // the kernel can, its epilogue also masks G with that output and sums bn2's backward reduce -- the previous block's
            // channel_reduce pass (g, out, y2 read, g written) disappears
            Block* pb = (&b == &blocks_.front()) ? nullptr : (&b - 1);
            int pr = 0;
            if (pb) LBC_TRY(conv_dgrad(b.c1, E1, D, Gbuf, N, s, &pb->b2, W(pb->c2.y), &pr, W(pb->out)));
            else    LBC_TRY(conv_dgrad(b.c1, E1, D, Gbuf, N, s));
            bwd_pre_rows_ = pr;
            if (pr > 0) { bwd_pre_pix_ = (long long)N * pb->c2.OH * pb->c2.OW; bwd_pre_C_ = pb->b2.C; }
        } else {
            float* Fd = dy_slot(pix * b.bd.C);
            LBC_REQUIRE(Fd, "net.backward: dY arena exhausted");
            LBC_TRY(conv_dgrad(b.c1, E1, nullptr, Gbuf, N, s));
            LBC_TRY(bn_backward(b.bd, D, nullptr, nullptr, W(b.ds.y), pix, Fd, b.bd.C, s));
            pending_.push_back({&b.ds, xin, nullptr, Fd});
            LBC_TRY(conv_dgrad(b.ds, Fd, Gbuf, Gbuf, N, s));                              // G += dgrad_1x1 (even pixels)
        }
        std::swap(D, Gbuf);
        return LBC_OK;
    }
    // Weight gradients go to the side stream (wstream), next to the input gradient that consumes the same dY.  E and F are
    // rewritten by the BatchNorm-backward apply passes / dgrads of the main stream: every apply that overwrites a buffer a
    // pending weight gradient may still read joins first (join_before_apply), every weight gradient forks after its dY exists.
    // out = relu(bn2(y2) + identity): mask by out, keep masked gradient in D for the identity path
    LBC_TRY(bn_backward(b.b2, D, W(b.out), D, W(b.c2.y), pix, E, b.b2.C, s, nullptr, true));   // E = dY2
    LBC_TRY(fork(s));
    // the reduce pass of bn1's backward rides on conv2's input-gradient epilogue where that kernel can (mask = bn1(y1) > 0 and
    // xhat both come from y1: one extra read instead of a 4-tensor pass)
    int fr = 0;
    if (b.fuse_z1) {
        LBC_TRY(conv_wgrad_pre(b.c2, W(b.c1.y), &b.b1, E, N, wstream(s)));
        LBC_TRY(conv_dgrad(b.c2, E, nullptr, F, N, s, &b.b1, W(b.c1.y), &fr));        // F = dZ1 (masked when fr > 0)
        LBC_TRY(bn_backward(b.b1, F, W(b.c1.y), F, W(b.c1.y), pix, E, b.b1.C, s, &b.b1, true, fr));   // E = dY1 (mask = bn1(y1) > 0)
    } else {
        LBC_TRY(conv_wgrad(b.c2, W(b.z1), E, N, wstream(s)));
        LBC_TRY(conv_dgrad(b.c2, E, nullptr, F, N, s, &b.b1, W(b.c1.y), &fr));        // F = dZ1 (masked when fr > 0)
        LBC_TRY(bn_backward(b.b1, F, W(b.z1), F, W(b.c1.y), pix, E, b.b1.C, s, nullptr, true, fr));   // E = dY1
    }
    LBC_TRY(fork(s));
    LBC_TRY(conv_wgrad(b.c1, xin, E, N, wstream(s)));
    if (!b.has_ds) {
        LBC_TRY(conv_dgrad(b.c1, E, D, Gbuf, N, s));                                  // G = dgrad + identity gradient
    } else {
        LBC_TRY(conv_dgrad(b.c1, E, nullptr, Gbuf, N, s));
        LBC_TRY(bn_backward(b.bd, D, nullptr, nullptr, W(b.ds.y), pix, F, b.bd.C, s)); // F = dYd (no pending reader of F)
        LBC_TRY(fork(s));
        LBC_TRY(conv_wgrad(b.ds, xin, F, N, wstream(s)));
        LBC_TRY(conv_dgrad(b.ds, F, Gbuf, Gbuf, N, s));                               // G += dgrad_1x1 (even pixels)
    }
    std::swap(D, Gbuf);
    return LBC_OK;
}

int Net::backward(const float* d_sel, const float* d_all, int stage, hipStream_t s)
{
    const int rc = backward_impl(d_sel, d_all, stage, s);
    if (rc != LBC_OK) { pending_.clear(); dy_used_ = 0; bwd_pre_rows_ = 0; }
    if (rc != LBC_OK && side_dirty_) {
        // an early return between fork() and join(): weight gradients may still be in flight on the side stream; let them
        // finish before anything (a retry, the next forward) reuses the gradient ping-pong buffers or the split-K slabs
        (void)hipStreamSynchronize(side_);
        side_dirty_ = false;
    }
    return rc;
}

int Net::backward_impl(const float* d_sel, const float* d_all, int stage, hipStream_t s)
{
    LBC_REQUIRE(lastN_ > 0 && last_train_, "net.backward: needs a preceding training-mode forward");
    LBC_REQUIRE(stage >= -1 && stage < kNumStages, "net.backward: bad stage %d", stage);
    LBC_TRY(check_bound(true));
    const int N = lastN_;
    float* E = W(gE_);
    float* F = W(gF_);
    side_on_ = side_allowed_ && !lbc_prof_on();     // the profiler's per-launch events want kernels that run alone
    if (side_on_ && !side_) {
        // the side stream belongs to the device that owns the workspace, whatever device is current in the calling thread
        int cur = 0, want = 0;
        (void)hipGetDevice(&cur);
        want = cur;
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, ws_) == hipSuccess) want = attr.device;
        if (want != cur) (void)hipSetDevice(want);
        const bool ok = hipStreamCreateWithFlags(&side_, hipStreamNonBlocking) == hipSuccess &&
                        hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming) == hipSuccess &&
                        hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming) == hipSuccess;
        if (want != cur) (void)hipSetDevice(cur);
        if (!ok) {
            lbc_set_error("net.backward: cannot create the side stream");
            return LBC_ELAUNCH;
        }
    }

    if (stage == -1 || stage == 0) {
        bwd_D_ = W(gD_); bwd_G_ = W(gG_);
        // ---- head ----
        HeadBwdArgs hb;
        memset(&hb, 0, sizeof(hb));
        HeadArgs& ha = hb.f;
        ha.h = W(dec_[2].u);
        for (int b = 0; b < 4; ++b) {
            ha.mean[b] = W(head_bn_[0].mean); ha.invstd[b] = W(head_bn_[0].invstd);
            ha.gamma[b] = P(head_bn_[b].g); ha.beta[b] = P(head_bn_[b].b);
            ha.w[b] = P(head_w_[b]); ha.bias[b] = P(head_b_[b]);
            ha.pos_x[b] = P(head_px_[b]); ha.pos_y[b] = P(head_py_[b]);
        }
        ha.cmd = W(cmd_); ha.pred_all = W(pred_all_); ha.rowstat = W(rowstat_);
        ha.N = N; ha.OH = HH_; ha.OW = HW_; ha.act_bf16 = act_bf16_;
        hb.d_all = d_all; hb.d_sel = d_sel; hb.s_partial = W(head_partial_); hb.dh = E; hb.chan_coef = W(head_coef_);
        LBC_TRY(lbc_head_bwd_reduce(hb, s));
        HeadBwdFinalizeArgs hf;
        memset(&hf, 0, sizeof(hf));
        hf.s_partial = W(head_partial_); hf.rows = lbc_head_bwd_rows(hb.f); hf.count = (long long)N * HH_ * HW_;
        if (hf.rows > 8) {   // the finalize kernel is one workgroup walking the rows serially: hand it 8 pre-reduced rows
            LBC_TRY(lbc_partial_reduce(W(head_partial_), hf.rows, 20 * 65, W(partial2_), 8, s));
            hf.s_partial = W(partial2_); hf.rows = 8;
        }
        for (int b = 0; b < 4; ++b) {
            hf.gamma[b] = P(head_bn_[b].g); hf.beta[b] = P(head_bn_[b].b); hf.w[b] = P(head_w_[b]);
            hf.dgamma[b] = G(head_bn_[b].g); hf.dbeta[b] = G(head_bn_[b].b);
            hf.dw[b] = G(head_w_[b]); hf.dbias[b] = G(head_b_[b]);
        }
        hf.mean = W(head_bn_[0].mean); hf.invstd = W(head_bn_[0].invstd); hf.chan_coef = W(head_coef_);
        LBC_TRY(lbc_head_bwd_finalize(hf, s));
        if (sync_fn_) {      // second pass on the global sums: only the per-channel coefficients of the input gradient
            LBC_TRY(sync_rows(hf.s_partial, hf.rows, 20 * 65, N, s));
            hf.nsum = hf.s_partial + 20 * 65; hf.n_local = N; hf.coef_only = 1;
            LBC_TRY(lbc_head_bwd_finalize(hf, s));
        }
        LBC_TRY(lbc_head_bwd_apply(hb, s));   // E = dU3

        // ---- decoder, last to first ----
        for (int i = 2; i >= 0; --i) {
            Deconv& D = dec_[i];
            const float* xin = i == 0 ? W(hcat_) : W(dec_[i - 1].u);
            const long long opix = (long long)N * 4 * D.H * D.W;
            const long long ipix = (long long)N * D.H * D.W;
            // ReLU backward in place + bias gradient
            ChanReduceArgs r;
            memset(&r, 0, sizeof(r));
            r.dz = E; r.mask = W(D.u); r.g_out = E; r.partial = W(partial_); r.pixels = opix; r.C = D.Cout; r.act_bf16 = act_bf16_;
            LBC_TRY(lbc_chan_reduce(r, 1, s));
            int rows = lbc_chan_reduce_rows(opix, D.Cout);
            const float* part = W(partial_);
            if (rows > kLbcFinalizeRows) {
                LBC_TRY(lbc_partial_reduce(W(partial_), rows, 2 * D.Cout, W(partial2_), 64, s));
                part = W(partial2_); rows = 64;
            }
            BnBwdFinalizeArgs f;
            memset(&f, 0, sizeof(f));
            f.partial = part; f.rows = rows; f.C = D.Cout; f.count = opix; f.dbeta = G(D.bias);
            LBC_TRY(lbc_bn_bwd_finalize(f, s));
            // weight gradient: P = bn(x) (dense), Q = g gathered with stride 2
            WgradArgs wa;
            memset(&wa, 0, sizeof(wa));
            wa.p = xin; wa.q = E; wa.partial = W(wg_partial_);
            wa.p_scale = W(D.bn.scale); wa.p_shift = W(D.bn.shift);
            wa.N = N; wa.OH = D.H; wa.OW = D.W; wa.CP = D.Cin;
            wa.H = 2 * D.H; wa.W = 2 * D.W; wa.CQ = D.Cout; wa.KH = 3; wa.KW = 3; wa.S = 2; wa.P = 1;
            wa.bf16 = bf16_; wa.act_bf16 = act_bf16_;
            wa.nsplit = lbc_wgrad_pick_split(wa);
            if ((size_t)wa.nsplit * D.Cin * 9 * D.Cout > wg_floats_) wa.nsplit = (int)std::max<size_t>(1, wg_floats_ / ((size_t)D.Cin * 9 * D.Cout));
            if (wa.nsplit == 1) wa.partial = G(D.w);
            LBC_TRY(lbc_wgrad_launch(wa, s));
            if (wa.nsplit > 1) LBC_TRY(lbc_splitk_reduce(wa.partial, wa.nsplit, (long long)D.Cin * 9 * D.Cout, G(D.w), 0.f, s));
            // input gradient: a stride-2 gather convolution over g with the same weight tensor
            IgemmArgs a;
            memset(&a, 0, sizeof(a));
            a.x = E; a.w = P(D.w); a.y = F;
            a.N = N; a.H = 2 * D.H; a.W = 2 * D.W; a.C = D.Cout;
            a.OH = D.H; a.OW = D.W; a.K = D.Cin; a.KH = 3; a.KW = 3; a.S = 2; a.P = 1;
            a.M = N * D.H * D.W; a.LH = D.H; a.LW = D.W; a.ostep = 1;
            a.bf16 = bf16_; a.act_bf16 = act_bf16_;
            if (act_bf16_) { a.w = W(D.wn); a.w_bf16 = 1; }
            LBC_TRY(lbc_igemm_launch(a, 1, 0, act_bf16_ ? lbc_igemm_pick_for(a, 0) : lbc_igemm_pick(a.M, a.K), s));   // F = d bn(x)
            // BatchNorm backward; for the first decoder stage only the 512 trunk channels carry on
            float* dst = i == 0 ? bwd_D_ : E;
            LBC_TRY(bn_backward(D.bn, F, nullptr, nullptr, xin, ipix, dst, i == 0 ? 512 : D.Cin, s));
        }
    }

    for (int li = 3; li >= 0; --li) {
        const int st = 4 - li;   // layer4 -> stage 1 ... layer1 -> stage 4
        if (stage != -1 && stage != st) continue;
        bwd_pre_rows_ = 0;       // (a stage's last block gets its gradient from the stage behind it through a downsample block or the decoder: never pre-reduced)
        const int first = stage_first_block_[li];
        const int lastb = li == 3 ? (int)blocks_.size() : stage_first_block_[li + 1];
        for (int bi = lastb - 1; bi >= first; --bi) LBC_TRY(block_backward(blocks_[bi], bwd_D_, bwd_G_, E, F, s));
        if (defer_wgrad_) LBC_TRY(flush_wgrads(N, s));
        LBC_TRY(join(s));     // the stage's gradients are complete on s (the caller all-reduces them behind an event on s)
    }

    if (stage == -1 || stage == 5) {
        // resnet.py:149-152 backward: maxpool -> relu -> bn1 -> conv1 weight gradient
        const int H0 = d_.H, W0 = d_.W;
        PoolBwdArgs pb;
        memset(&pb, 0, sizeof(pb));
        pb.dp = bwd_D_; pb.idx = reinterpret_cast<const unsigned char*>(W(idx_)); pb.y = W(y0_);
        pb.scale = W(stem_bn_.scale); pb.shift = W(stem_bn_.shift);
        pb.mean = W(stem_bn_.mean); pb.invstd = W(stem_bn_.invstd);
        pb.g = W(g0_); pb.partial = W(partial_);
        pb.N = N; pb.H = H0 / 2; pb.W = W0 / 2; pb.C = 64; pb.act_bf16 = act_bf16_;
        LBC_TRY(lbc_maxpool_relu_bwd_reduce(pb, s));
        int rows = lbc_pool_bwd_rows(N, H0 / 2, W0 / 2, 64);
        const long long pix = (long long)N * (H0 / 2) * (W0 / 2);
        const float* part = W(partial_);
        if (rows > kLbcFinalizeRows) {
            LBC_TRY(lbc_partial_reduce(W(partial_), rows, 128, W(partial2_), 64, s));
            part = W(partial2_); rows = 64;
        }
        BnBwdFinalizeArgs f;
        memset(&f, 0, sizeof(f));
        f.partial = part; f.rows = rows; f.C = 64; f.count = pix;
        f.gamma = P(stem_bn_.g); f.mean = W(stem_bn_.mean); f.invstd = W(stem_bn_.invstd); f.train = 1;
        f.dgamma = G(stem_bn_.g); f.dbeta = G(stem_bn_.b);
        f.coefA = W(stem_bn_.cA); f.coefB = W(stem_bn_.cB); f.coefD = W(stem_bn_.cD);
        LBC_TRY(bn_bwd_finalize(f, s));
        StemWgradArgs sw;
        memset(&sw, 0, sizeof(sw));
        if (lbc_stem_wgrad_fuses_bn_bwd(d_.in_channels, bf16_)) {
            // the apply pass dx = A (g - k1 - xhat k2) feeds only the stem's weight gradient (nothing lies below conv1): that kernel
            // forms it while it stages g (-1 read and -1 write of the largest activation, +1 read of y0 there)
            sw.bn_y = W(y0_); sw.bn_coefA = W(stem_bn_.cA); sw.bn_coefB = W(stem_bn_.cB); sw.bn_coefD = W(stem_bn_.cD);
            sw.bn_mean = W(stem_bn_.mean); sw.bn_invstd = W(stem_bn_.invstd);
        } else {
            BnBwdApplyArgs ap;
            memset(&ap, 0, sizeof(ap));
            ap.g = W(g0_); ap.x = W(y0_); ap.coefA = W(stem_bn_.cA); ap.coefB = W(stem_bn_.cB); ap.coefD = W(stem_bn_.cD);
            ap.mean = W(stem_bn_.mean); ap.invstd = W(stem_bn_.invstd);
            ap.dx = W(g0_); ap.pixels = pix; ap.C = 64; ap.Cout = 64; ap.act_bf16 = act_bf16_;
            LBC_TRY(lbc_bn_bwd_apply(ap, s));
        }
        sw.xp = W(xp_); sw.xp_bf16 = bf16_; sw.dy = W(g0_); sw.partial = W(wg_partial_);
        sw.N = N; sw.H = H0; sw.W = W0; sw.Cin = d_.in_channels; sw.act_bf16 = act_bf16_; sw.bf16 = bf16_;
        sw.nsplit = lbc_stem_wgrad_split(N, H0, W0, d_.in_channels, bf16_);
        LBC_TRY(lbc_stem_wgrad(sw, s));
        LBC_TRY(lbc_splitk_reduce(sw.partial, sw.nsplit, (long long)64 * 49 * d_.in_channels, G(stem_w_), 0.f, s));
    }
    return LBC_OK;
}

}  // namespace lbc

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
struct lbc_net { lbc::Net impl; explicit lbc_net(const lbc_net_desc& d) : impl(d) {} };

extern "C" {

int lbc_net_create(const lbc_net_desc* d, lbc_net** out)
{
    LBC_REQUIRE(d && out, "net_create: null argument");
    LBC_REQUIRE(d->arch == 18 || d->arch == 34, "net_create: arch %d unsupported (BasicBlock ResNet-18/34 only)", d->arch);
    LBC_REQUIRE(d->in_channels == 3 || d->in_channels == 7, "net_create: in_channels %d unsupported", d->in_channels);
    LBC_REQUIRE(d->H > 0 && d->W > 0 && d->H % 32 == 0 && d->W % 32 == 0, "net_create: image %dx%d must be a multiple of 32", d->H, d->W);
    LBC_REQUIRE(d->max_batch >= 1, "net_create: max_batch %d", d->max_batch);
    LBC_REQUIRE(!d->normalize || d->in_channels == 3, "net_create: ImageNet normalisation needs 3 channels");
    LBC_REQUIRE(d->precision >= 0 && d->precision <= 2,
                "net_create: precision %d unknown (0 = f32, 1 = bf16 MFMA operands, 2 = bf16 operands + bf16 activation storage)", d->precision);
    LBC_REQUIRE((long long)d->max_batch * (d->H / 2) * (d->W / 2) * 64 < (1ll << 31), "net_create: batch too large for 32-bit indexing");
    *out = new lbc_net(*d);
    return LBC_OK;
}
void lbc_net_destroy(lbc_net* net) { delete net; }
int lbc_net_num_tensors(const lbc_net* net) { return net ? (int)const_cast<lbc_net*>(net)->impl.tensors().size() : 0; }
int lbc_net_tensor_info(const lbc_net* net, int i, char* name, int name_cap, int* kind, int* ndim, int* shape4)
{
    LBC_REQUIRE(net, "tensor_info: null net");
    auto& ts = const_cast<lbc_net*>(net)->impl.tensors();
    LBC_REQUIRE(i >= 0 && i < (int)ts.size(), "tensor_info: index %d out of range", i);
    const lbc::TensorInfo& t = ts[(size_t)i];
    if (name && name_cap > 0) { strncpy(name, t.name.c_str(), (size_t)name_cap - 1); name[name_cap - 1] = 0; }
    if (kind) *kind = t.kind;
    if (ndim) *ndim = t.ndim;
    if (shape4) for (int k = 0; k < 4; ++k) shape4[k] = t.shape[k];
    return LBC_OK;
}
size_t lbc_net_workspace_bytes(const lbc_net* net) { return net ? net->impl.workspace_bytes() : 0; }
int lbc_net_num_activations(const lbc_net* net) { return net ? (int)net->impl.activations().size() : 0; }
int lbc_net_activation_info(const lbc_net* net, int i, char* name, int name_cap, size_t* offset_bytes, int* hwc3, int* elem_bytes)
{
    LBC_REQUIRE(net, "activation_info: null net");
    const auto acts = net->impl.activations();
    LBC_REQUIRE(i >= 0 && i < (int)acts.size(), "activation_info: index %d out of range", i);
    const auto& a = acts[(size_t)i];
    if (name && name_cap > 0) { strncpy(name, a.name.c_str(), (size_t)name_cap - 1); name[name_cap - 1] = 0; }
    if (offset_bytes) *offset_bytes = a.offset_bytes;
    if (hwc3) { hwc3[0] = a.H; hwc3[1] = a.W; hwc3[2] = a.C; }
    if (elem_bytes) *elem_bytes = a.elem;
    return LBC_OK;
}
int lbc_net_bind(lbc_net* net, void* workspace, void* const* tensor_ptrs, float* const* grad_ptrs)
{
    LBC_REQUIRE(net && workspace && tensor_ptrs, "net_bind: null argument");
    net->impl.set_workspace(workspace);
    net->impl.invalidate_derived();
    auto& ts = net->impl.tensors();
    for (size_t i = 0; i < ts.size(); ++i) {
        ts[i].ptr = tensor_ptrs[i];
        ts[i].grad = (grad_ptrs && ts[i].kind == lbc::kParam) ? grad_ptrs[i] : nullptr;
    }
    return LBC_OK;
}
int lbc_net_forward(lbc_net* net, int N, int train, const float* image, const float* velocity, const float* command,
                    float* pred_sel, float* pred_all, lbc_stream_t stream)
{
    LBC_REQUIRE(net, "net_forward: null net");
    return net->impl.forward(N, train, image, 0, velocity, command, pred_sel, pred_all, (hipStream_t)stream);
}
int lbc_net_forward_u8(lbc_net* net, int N, int train, const unsigned char* image_nhwc, const float* velocity, const float* command,
                       float* pred_sel, float* pred_all, lbc_stream_t stream)
{
    LBC_REQUIRE(net, "net_forward_u8: null net");
    return net->impl.forward(N, train, image_nhwc, 1, velocity, command, pred_sel, pred_all, (hipStream_t)stream);
}
int lbc_net_num_stages(void) { return lbc::Net::kNumStages; }
int lbc_net_last_forward(const lbc_net* net, int* batch, int* train, long long* generation)
{
    LBC_REQUIRE(net, "net_last_forward: null net");
    if (batch) *batch = net->impl.last_batch();
    if (train) *train = net->impl.last_train();
    if (generation) *generation = net->impl.generation();
    return LBC_OK;
}
int lbc_net_set_frozen(lbc_net* net, int frozen)
{
    LBC_REQUIRE(net, "net_set_frozen: null net");
    net->impl.set_frozen(frozen != 0);
    return LBC_OK;
}
int lbc_net_set_sync_bn(lbc_net* net, lbc_allreduce_fn fn, void* ctx, int world_size, float* buf, int buf_floats)
{
    LBC_REQUIRE(net, "net_set_sync_bn: null net");
    return net->impl.set_sync_bn(fn, ctx, world_size, buf, buf_floats);
}
int lbc_net_backward(lbc_net* net, const float* d_sel, const float* d_all, int stage, lbc_stream_t stream)
{
    LBC_REQUIRE(net, "net_backward: null net");
    return net->impl.backward(d_sel, d_all, stage, (hipStream_t)stream);
}

int lbc_loss(int kind, const lbc_camera* cam, const float* pred, const float* target, int N, int rows, float grad_scale,
             float* loss_per_sample, float* dpred, lbc_stream_t stream)
{
    LBC_REQUIRE(pred && target && loss_per_sample, "loss: null argument");
    LossArgs a;
    memset(&a, 0, sizeof(a));
    a.pred = pred; a.target = target; a.loss_per_sample = loss_per_sample; a.dpred = dpred;
    a.N = N; a.R = rows; a.grad_scale = grad_scale;
    if (cam) {
        a.w = cam->w; a.h = cam->h; a.fov = cam->fov; a.world_y = cam->world_y; a.fixed_offset = cam->fixed_offset;
        a.pixels_per_meter = cam->pixels_per_meter; a.crop_size = cam->crop_size;
    }
    if (kind == 1) { LBC_REQUIRE(cam, "loss: phase-1 needs a camera"); return lbc_loss_phase1(a, (hipStream_t)stream); }
    if (kind == 0) { LBC_REQUIRE(cam, "loss: phase-0 needs a camera"); return lbc_loss_phase0(a, (hipStream_t)stream); }
    if (kind == 2) {
        // train_birdview.py:48-52: gt / (0.5*size) - 1 with w = h = crop_size
        LBC_REQUIRE(cam, "loss: bird-view L1 needs crop_size");
        return lbc_loss_l1(a, 1.f / (0.5f * cam->crop_size), -1.f, (hipStream_t)stream);
    }
    if (kind == 3) return lbc_loss_l1(a, 1.f, 0.f, (hipStream_t)stream);   // targets already normalised
    lbc_set_error("loss: unknown kind %d", kind);
    return LBC_EINVAL;
}

int lbc_phase2_weight(const lbc_camera* cam, const float* pred_sel, const float* teacher_sel, int N, float* weights,
                      lbc_stream_t stream)
{
    LBC_REQUIRE(cam && pred_sel && teacher_sel && weights, "phase2_weight: null argument");
    LossArgs a;
    memset(&a, 0, sizeof(a));
    a.pred = pred_sel; a.target = teacher_sel; a.loss_per_sample = weights; a.N = N; a.R = 5;
    a.w = cam->w; a.h = cam->h; a.fov = cam->fov; a.world_y = cam->world_y; a.fixed_offset = cam->fixed_offset;
    a.pixels_per_meter = cam->pixels_per_meter; a.crop_size = cam->crop_size;
    return lbc_phase2_weight_launch(a, (hipStream_t)stream);
}

int lbc_adam_step(const lbc_adam_chunk* chunks_dev, int nchunks, double lr, double beta1, double beta2,
                  double eps, double weight_decay, int step, lbc_stream_t stream)
{
    static_assert(sizeof(lbc_adam_chunk) == sizeof(AdamChunk), "chunk layout");
    return lbc_adam_launch(reinterpret_cast<const AdamChunk*>(chunks_dev), nchunks, lr, beta1, beta2, eps, weight_decay,
                           step, (hipStream_t)stream);
}

}  // extern "C"
