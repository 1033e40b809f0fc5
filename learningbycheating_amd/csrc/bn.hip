// BatchNorm2d (training and eval), ReLU, residual add and their gradients for
// NHWC fp32 tensors on gfx950.  These are the HBM-bound kernels of the path:
// every kernel moves 16 bytes per lane, walks pixels with a grid-stride loop and
// keeps per-channel coefficients in registers (a thread's channel group is fixed).
//
// Semantics follow torch.nn.BatchNorm2d as used by the reference
// (bird_view/models/resnet.py:31,34,104,137; image.py:38,41,44,56):
//   train: mean / biased variance over (N,H,W); running_var gets the unbiased one,
//          momentum 0.1, eps 1e-5, num_batches_tracked += 1.
//   eval:  running statistics.
// Statistics arrive as per-workgroup partial (sum, sum^2) rows written by the
// producing convolution's epilogue (or by channel_stats below); they are reduced
// here in double precision in a fixed order, so results are run-to-run identical.
#include "lbc_common.hpp"
#include "lbc_kernels.hpp"

namespace {

// ---- partial-row pre-reduction: in[rows][cols] -> out[R][cols] ---------------
__global__ __launch_bounds__(256) void partial_reduce_k(const float* __restrict__ in, int rows, int cols, float* __restrict__ out)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    // four independent accumulators keep four loads in flight (the loop is latency bound otherwise); fixed order -> deterministic
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const int step = gridDim.y;
    int r = blockIdx.y;
    for (; r + 3 * step < rows; r += 4 * step) {
        s0 += (double)in[(size_t)r * cols + c];
        s1 += (double)in[(size_t)(r + step) * cols + c];
        s2 += (double)in[(size_t)(r + 2 * step) * cols + c];
        s3 += (double)in[(size_t)(r + 3 * step) * cols + c];
    }
    for (; r < rows; r += step) s0 += (double)in[(size_t)r * cols + c];
    out[(size_t)blockIdx.y * cols + c] = (float)((s0 + s1) + (s2 + s3));
}

// Sums column c of a [rows][2][C] partial buffer.  The finalize kernels run 256 threads = 64 channels x 4 row-lanes: lane q
// takes rows q, q+4, ... with two loads in flight, the four lanes are combined through LDS in a fixed order (deterministic).
__device__ __forceinline__ void sum_rows2(const float* __restrict__ partial, int rows, int C, int c, bool valid, double& o1, double& o2)
{
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    if (valid) {
        int r = q;
        for (; r + 4 < rows; r += 8) {
            const float* p = partial + (size_t)r * 2 * C + c;
            a0 += (double)p[0];             b0 += (double)p[C];
            a1 += (double)p[8 * C];         b1 += (double)p[9 * C];
        }
        for (; r < rows; r += 4) {
            a0 += (double)partial[(size_t)r * 2 * C + c];
            b0 += (double)partial[(size_t)r * 2 * C + C + c];
        }
    }
    red[0][q][cl] = a0 + a1;
    red[1][q][cl] = b0 + b1;
    __syncthreads();
    o1 = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    o2 = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
}

// ---- forward finalize ----------------------------------------------------------
__global__ __launch_bounds__(256) void bn_finalize_k(BnFinalizeArgs a)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const bool lead = threadIdx.x < 64;
    if (c == 0 && lead && a.num_batches_tracked && a.train) *a.num_batches_tracked += 1;
    double s1 = 0.0, s2 = 0.0;
    if (a.train) sum_rows2(a.partial, a.rows, a.C, c, c < a.C, s1, s2);
    if (c >= a.C || !lead) return;
    float mean, invstd;
    if (a.train) {
        const double n = (double)a.count;
        const double m = s1 / n;
        double var = s2 / n - m * m;
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)a.eps));
        if (a.running_mean) {
            const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
            a.running_mean[c] = (float)((1.0 - a.momentum) * (double)a.running_mean[c] + a.momentum * m);
            a.running_var[c] = (float)((1.0 - a.momentum) * (double)a.running_var[c] + a.momentum * unb);
        }
    } else {
        mean = a.running_mean[c];
        invstd = 1.0f / sqrtf(a.running_var[c] + a.eps);
    }
    if (a.save_mean) { a.save_mean[c] = mean; a.save_invstd[c] = invstd; }
    const float g = a.gamma ? a.gamma[c] : 1.f;
    const float b = a.beta ? a.beta[c] : 0.f;
    const float sc = g * invstd;
    a.scale[c] = sc;
    a.shift[c] = b - mean * sc;
}

// ---- elementwise apply: y = relu?(x*s + t (+ r [* rs + rt])) ------------------
__global__ __launch_bounds__(256) void bn_apply_k(BnApplyArgs a)
{
    const int c4n = a.C / 4;
    const long long total4 = a.pixels * c4n;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        const int c = (int)(i % c4n) * 4;
        float4 v = reinterpret_cast<const float4*>(a.x)[i];
        const float4 s = *reinterpret_cast<const float4*>(a.scale + c);
        const float4 t = *reinterpret_cast<const float4*>(a.shift + c);
        v.x = v.x * s.x + t.x; v.y = v.y * s.y + t.y; v.z = v.z * s.z + t.z; v.w = v.w * s.w + t.w;
        if (a.resid) {
            float4 r = reinterpret_cast<const float4*>(a.resid)[i];
            if (a.rscale) {
                const float4 rs = *reinterpret_cast<const float4*>(a.rscale + c);
                const float4 rt = *reinterpret_cast<const float4*>(a.rshift + c);
                r.x = r.x * rs.x + rt.x; r.y = r.y * rs.y + rt.y; r.z = r.z * rs.z + rt.z; r.w = r.w * rs.w + rt.w;
            }
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        reinterpret_cast<float4*>(a.y)[i] = v;
    }
}

// ---- per-channel reductions over pixels ----------------------------------------
// Block layout: 256 threads = (C/4 channel groups) x RL pixel lanes (RL = 256/(C/4),
// C/4 <= 256).  Each block owns a contiguous pixel range and writes one partial row
// [2][C]:  row0 = sum g, row1 = sum g*q  where the meaning of g, q depends on the op.
template <int OP>
__global__ __launch_bounds__(256) void channel_reduce_k(ChanReduceArgs a)
{
    __shared__ __attribute__((aligned(16))) float red[2 * 256 * 4];
    const int c4n = a.C / 4;
    const int rl = 256 / c4n;
    const int cg = threadIdx.x % c4n;
    const int pl = threadIdx.x / c4n;
    const int c = cg * 4;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (pl < rl) {
        float4 mean = make_float4(0.f, 0.f, 0.f, 0.f), inv = make_float4(1.f, 1.f, 1.f, 1.f);
        float4 msc = make_float4(1.f, 1.f, 1.f, 1.f), msh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (OP == 1 && a.mean) {
            mean = *reinterpret_cast<const float4*>(a.mean + c);
            inv = *reinterpret_cast<const float4*>(a.invstd + c);
        }
        if (OP == 1 && a.mask_scale) {
            msc = *reinterpret_cast<const float4*>(a.mask_scale + c);
            msh = *reinterpret_cast<const float4*>(a.mask_shift + c);
        }
        const long long p0 = (long long)blockIdx.x * a.pix_per_block;
        long long p1 = p0 + a.pix_per_block;
        if (p1 > a.pixels) p1 = a.pixels;
        for (long long p = p0 + pl; p < p1; p += rl) {
            const long long i = p * c4n + cg;
            if (OP == 0) {            // plain statistics of x: sum x, sum x^2
                const float4 v = reinterpret_cast<const float4*>(a.x)[i];
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
            } else {                  // backward: g = dz * (mask > 0); sum g, sum g * xhat
                float4 g = reinterpret_cast<const float4*>(a.dz)[i];
                if (a.mask) {
                    float4 m = reinterpret_cast<const float4*>(a.mask)[i];
                    if (a.mask_scale) {
                        m.x = m.x * msc.x + msh.x; m.y = m.y * msc.y + msh.y;
                        m.z = m.z * msc.z + msh.z; m.w = m.w * msc.w + msh.w;
                    }
                    g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f;
                    g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f;
                }
                if (a.g_out) reinterpret_cast<float4*>(a.g_out)[i] = g;
                s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
                if (a.x) {
                    const float4 v = reinterpret_cast<const float4*>(a.x)[i];
                    s2.x += g.x * (v.x - mean.x) * inv.x; s2.y += g.y * (v.y - mean.y) * inv.y;
                    s2.z += g.z * (v.z - mean.z) * inv.z; s2.w += g.w * (v.w - mean.w) * inv.w;
                }
            }
        }
    }
    reinterpret_cast<float4*>(red)[threadIdx.x] = s1;
    reinterpret_cast<float4*>(red)[256 + threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < c4n) {
        float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
        for (int k = 0; k < rl; ++k) {
            const float4 u = reinterpret_cast<const float4*>(red)[k * c4n + threadIdx.x];
            const float4 w = reinterpret_cast<const float4*>(red)[256 + k * c4n + threadIdx.x];
            t1.x += u.x; t1.y += u.y; t1.z += u.z; t1.w += u.w;
            t2.x += w.x; t2.y += w.y; t2.z += w.z; t2.w += w.w;
        }
        float* dst = a.partial + (size_t)blockIdx.x * 2 * a.C;
        *reinterpret_cast<float4*>(dst + c) = t1;
        *reinterpret_cast<float4*>(dst + a.C + c) = t2;
    }
}

// ---- backward finalize: partial rows -> dgamma, dbeta and the apply coefficients --
//   dx = A*(g - k1 - xhat*k2),  A = gamma*invstd, k1 = sum(g)/n, k2 = sum(g*xhat)/n,
//   xhat recomputed per element as (x-mean)*invstd (factoring it into B*x + D would put a
//   systematic per-channel rounding offset on dx that downstream channel sums amplify).
__global__ __launch_bounds__(256) void bn_bwd_finalize_k(BnBwdFinalizeArgs a)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double s1, s2;
    sum_rows2(a.partial, a.rows, a.C, c, c < a.C, s1, s2);
    if (c >= a.C || threadIdx.x >= 64) return;
    if (a.dbeta) a.dbeta[c] = (float)s1;
    if (a.dgamma) a.dgamma[c] = (float)s2;
    if (a.coefA) {
        const double n = (double)a.count;
        const double g = a.gamma ? (double)a.gamma[c] : 1.0;
        const double inv = (double)a.invstd[c];
        const double mean = (double)a.mean[c];
        const double A = g * inv;
        (void)mean;
        a.coefA[c] = (float)A;
        a.coefB[c] = a.train ? (float)(s1 / n) : 0.f;   // k1
        a.coefD[c] = a.train ? (float)(s2 / n) : 0.f;   // k2 (eval-mode BN is a fixed affine map: k1 = k2 = 0)
    }
}

// ---- backward apply: dx = A*(g - k1 - xhat*k2) over the first Cout channels --------
__global__ __launch_bounds__(256) void bn_bwd_apply_k(BnBwdApplyArgs a)
{
    const int c4n = a.C / 4;
    const int o4n = a.Cout / 4;
    const long long total4 = a.pixels * o4n;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        const long long p = i / o4n;
        const int cg = (int)(i - p * o4n);
        const int c = cg * 4;
        const long long j = p * c4n + cg;
        float4 g = reinterpret_cast<const float4*>(a.g)[j];
        if (a.mask) {
            const float4 m = reinterpret_cast<const float4*>(a.mask)[j];
            g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f;
            g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f;
        }
        const float4 v = reinterpret_cast<const float4*>(a.x)[j];
        const float4 A = *reinterpret_cast<const float4*>(a.coefA + c);
        const float4 K1 = *reinterpret_cast<const float4*>(a.coefB + c);
        const float4 K2 = *reinterpret_cast<const float4*>(a.coefD + c);
        const float4 mu = *reinterpret_cast<const float4*>(a.mean + c);
        const float4 iv = *reinterpret_cast<const float4*>(a.invstd + c);
        float4 o;
        o.x = A.x * (g.x - K1.x - (v.x - mu.x) * iv.x * K2.x); o.y = A.y * (g.y - K1.y - (v.y - mu.y) * iv.y * K2.y);
        o.z = A.z * (g.z - K1.z - (v.z - mu.z) * iv.z * K2.z); o.w = A.w * (g.w - K1.w - (v.w - mu.w) * iv.w * K2.w);
        if (a.accum) {
            const float4 q = reinterpret_cast<const float4*>(a.dx)[i];
            o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
        }
        reinterpret_cast<float4*>(a.dx)[i] = o;
    }
}

// ---- velocity late fusion: h = cat(trunk, speed broadcast to 128 channels) --------
// reference bird_view/models/image.py:77-79 / birdview.py:67-69
__global__ __launch_bounds__(256) void concat_velocity_k(const float* __restrict__ t, const float* __restrict__ vel,
                                                         float* __restrict__ h, long long pixels, int hw, int Ct, int Cv)
{
    const int C = Ct + Cv;
    const int c4n = C / 4;
    const long long total4 = pixels * c4n;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        const long long p = i / c4n;
        const int c = (int)(i - p * c4n) * 4;
        float4 v;
        if (c < Ct) v = *reinterpret_cast<const float4*>(t + p * Ct + c);
        else { const float s = vel[p / hw]; v = make_float4(s, s, s, s); }
        reinterpret_cast<float4*>(h)[i] = v;
    }
}

int grid_for(long long total4)
{
    long long b = (total4 + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

int lbc_partial_reduce(const float* in, int rows, int cols, float* out, int out_rows, hipStream_t s)
{
    dim3 grid((unsigned)lbc_cdiv(cols, 256), (unsigned)out_rows);
    LbcProfScope prof("partial_reduce", 0.0, 4.0 * (double)rows * cols, s);
    hipLaunchKernelGGL(partial_reduce_k, grid, dim3(256), 0, s, in, rows, cols, out);
    return lbc_check_launch("partial_reduce");
}

int lbc_bn_finalize(const BnFinalizeArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.C > 0 && a.scale && a.shift, "bn_finalize: bad args");
    LbcProfScope prof("bn_finalize", 0.0, 4.0 * (double)a.rows * 2 * a.C, s);
    hipLaunchKernelGGL(bn_finalize_k, dim3((unsigned)lbc_cdiv(a.C, 64)), dim3(256), 0, s, a);
    return lbc_check_launch("bn_finalize");
}

int lbc_bn_apply(const BnApplyArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.C % 4 == 0 && a.pixels > 0, "bn_apply: bad shape");
    LbcProfScope prof("bn_apply", 0.0, 4.0 * (double)a.pixels * a.C * (a.resid ? 3 : 2), s);
    hipLaunchKernelGGL(bn_apply_k, dim3((unsigned)grid_for(a.pixels * (a.C / 4))), dim3(256), 0, s, a);
    return lbc_check_launch("bn_apply");
}

int lbc_chan_reduce_rows(long long pixels, int C)
{
    const int rl = 256 / (C / 4);
    // at most 1024 workgroups, at least 8 pixels per pixel-lane; small tensors still get >= ~512 workgroups when they can
    long long ppb = (long long)rl * 64;
    long long rows = (pixels + ppb - 1) / ppb;
    if (rows < 512) {
        ppb = (long long)rl * 8;
        rows = (pixels + ppb - 1) / ppb;
        if (rows > 512) rows = 512;
    }
    if (rows > 1024) rows = 1024;
    if (rows < 1) rows = 1;
    return (int)rows;
}

int lbc_chan_reduce(ChanReduceArgs a, int op, hipStream_t s)
{
    LBC_REQUIRE(a.C % 4 == 0 && a.C / 4 <= 256, "chan_reduce: C=%d unsupported", a.C);
    const int rows = lbc_chan_reduce_rows(a.pixels, a.C);
    a.pix_per_block = (a.pixels + rows - 1) / rows;
    LbcProfScope prof(op == 0 ? "channel_stats" : "bn_bwd_reduce", 0.0,
                      4.0 * (double)a.pixels * a.C * (op == 0 ? 1 : (1 + (a.mask ? 1 : 0) + (a.x ? 1 : 0) + (a.g_out ? 1 : 0))), s);
    if (op == 0) hipLaunchKernelGGL((channel_reduce_k<0>), dim3((unsigned)rows), dim3(256), 0, s, a);
    else         hipLaunchKernelGGL((channel_reduce_k<1>), dim3((unsigned)rows), dim3(256), 0, s, a);
    return lbc_check_launch("channel_reduce");
}

int lbc_bn_bwd_finalize(const BnBwdFinalizeArgs& a, hipStream_t s)
{
    LbcProfScope prof("bn_bwd_finalize", 0.0, 4.0 * (double)a.rows * 2 * a.C, s);
    hipLaunchKernelGGL(bn_bwd_finalize_k, dim3((unsigned)lbc_cdiv(a.C, 64)), dim3(256), 0, s, a);
    return lbc_check_launch("bn_bwd_finalize");
}

int lbc_bn_bwd_apply(const BnBwdApplyArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.C % 4 == 0 && a.Cout % 4 == 0 && a.Cout <= a.C, "bn_bwd_apply: bad channels");
    LbcProfScope prof("bn_bwd_apply", 0.0, 4.0 * (double)a.pixels * (a.C * (a.mask ? 3.0 : 2.0) + a.Cout * (a.accum ? 2.0 : 1.0)), s);
    hipLaunchKernelGGL(bn_bwd_apply_k, dim3((unsigned)grid_for(a.pixels * (a.Cout / 4))), dim3(256), 0, s, a);
    return lbc_check_launch("bn_bwd_apply");
}

int lbc_concat_velocity(const float* t, const float* vel, float* h, int N, int hw, int Ct, int Cv, hipStream_t s)
{
    LBC_REQUIRE(Ct % 4 == 0 && Cv % 4 == 0, "concat_velocity: channels must be multiples of 4");
    const long long pixels = (long long)N * hw;
    LbcProfScope prof("concat_velocity", 0.0, 4.0 * (double)pixels * (Ct + Ct + Cv), s);
    hipLaunchKernelGGL(concat_velocity_k, dim3((unsigned)grid_for(pixels * ((Ct + Cv) / 4))), dim3(256), 0, s, t, vel, h,
                       pixels, hw, Ct, Cv);
    return lbc_check_launch("concat_velocity");
}
