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
#include "lbc_act.hpp"
#include "lbc_kernels.hpp"

namespace {

// ---- partial-row pre-reduction: in[rows][cols] -> out[R][cols] ---------------
__global__ __launch_bounds__(256) void partial_reduce_k(const float* __restrict__ in, int rows, int cols, float* __restrict__ out,
                                                        float* __restrict__ copy_lo, float* __restrict__ copy_hi, float tail)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && blockIdx.y == 0 && tail >= 0.f) out[cols] = tail;
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
    const float v = (float)((s0 + s1) + (s2 + s3));
    out[(size_t)blockIdx.y * cols + c] = v;
    // one output row of [2][C] sums: the halves are also the local dbeta / dgamma (SyncBN backward, engine.cpp)
    if (copy_lo && c < cols / 2) copy_lo[c] = v;
    if (copy_hi && c >= cols / 2) copy_hi[c - cols / 2] = v;
}

// Sums column c of a [rows][2][C] partial buffer.  The finalize kernels run 1024 threads = 16 channels x 64 row-lanes: lane q
// takes rows q, q+64, ... with four row pairs in flight, the 64 lanes are combined through LDS in a fixed order
// (deterministic).  Up to ~1024 rows this is one short launch; larger row counts are pre-reduced by partial_reduce_k.
constexpr int kFinCh = 16, kFinLanes = 64, kFinWaves = kFinCh * kFinLanes / 64;
__device__ __forceinline__ void sum_rows2(const float* __restrict__ partial, int rows, int C, int c, bool valid, double& o1, double& o2)
{
    __shared__ double red[2][kFinWaves][kFinCh];
    const int cl = threadIdx.x & (kFinCh - 1), q = threadIdx.x / kFinCh;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    if (valid) {
        const size_t rs = (size_t)2 * C;
        int r = q;
        for (; r + 3 * kFinLanes < rows; r += 4 * kFinLanes) {
            const float* p = partial + (size_t)r * rs + c;
            a0 += (double)p[0];                      b0 += (double)p[C];
            a1 += (double)p[kFinLanes * rs];         b1 += (double)p[kFinLanes * rs + C];
            a2 += (double)p[2 * kFinLanes * rs];     b2 += (double)p[2 * kFinLanes * rs + C];
            a3 += (double)p[3 * kFinLanes * rs];     b3 += (double)p[3 * kFinLanes * rs + C];
        }
        for (; r < rows; r += kFinLanes) {
            a0 += (double)partial[(size_t)r * rs + c];
            b0 += (double)partial[(size_t)r * rs + C + c];
        }
    }
    // a wave holds 4 row-lanes x 16 channels: fold the row-lanes in registers, one LDS row per wave, 16 rows for the lead lanes
    // (the 64-step serial LDS walk this replaces was most of the kernel's 8 us)
    double v1 = (a0 + a1) + (a2 + a3), v2 = (b0 + b1) + (b2 + b3);
    v1 += __shfl_xor(v1, 16); v2 += __shfl_xor(v2, 16);
    v1 += __shfl_xor(v1, 32); v2 += __shfl_xor(v2, 32);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < kFinCh) { red[0][wave][cl] = v1; red[1][wave][cl] = v2; }
    __syncthreads();
    o1 = 0.0; o2 = 0.0;
    if (q == 0) {
#pragma unroll
        for (int k = 0; k < kFinWaves; ++k) { o1 += red[0][k][cl]; o2 += red[1][k][cl]; }
    }
}

// ---- forward finalize ----------------------------------------------------------
__global__ __launch_bounds__(1024) void bn_finalize_k(BnFinalizeArgs a)
{
    const int c = blockIdx.x * kFinCh + (threadIdx.x & (kFinCh - 1));
    const bool lead = threadIdx.x < kFinCh;
    if (c == 0 && lead && a.num_batches_tracked && a.train) {
        *a.num_batches_tracked += 1;
        for (int k = 0; k < 3 && a.more_num_batches_tracked[k]; ++k) *a.more_num_batches_tracked[k] += 1;
    }
    double s1 = 0.0, s2 = 0.0;
    if (a.train) sum_rows2(a.partial, a.rows, a.C, c, c < a.C, s1, s2);
    if (c >= a.C || !lead) return;
    float mean, invstd;
    if (a.train) {
        const double n = a.nsum ? (double)a.count / (double)a.n_local * (double)*a.nsum : (double)a.count;
        const double m = s1 / n;
        double var = s2 / n - m * m;
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)a.eps));
        if (a.running_mean) {
            const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
            a.running_mean[c] = (float)((1.0 - a.momentum) * (double)a.running_mean[c] + a.momentum * m);
            a.running_var[c] = (float)((1.0 - a.momentum) * (double)a.running_var[c] + a.momentum * unb);
            for (int k = 0; k < 3 && a.more_running_mean[k]; ++k) {
                a.more_running_mean[k][c] = (float)((1.0 - a.momentum) * (double)a.more_running_mean[k][c] + a.momentum * m);
                a.more_running_var[k][c] = (float)((1.0 - a.momentum) * (double)a.more_running_var[k][c] + a.momentum * unb);
            }
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

// ---- folded finalize: the consumer's workgroup derives the coefficients itself -------------------------------------------------
// Column sums of partial[rows][2C] -> colsum[2C] (double, LDS), by all 256 threads: a thread owns one 16-byte column group and a
// row lane, loads up to 16 rows at once (every load of a batch in flight before the first use: at most two batches under the
// lbc_bn_fold_ok limit of 128 KB per BatchNorm -- a dependent chain of row reads was the whole cost of a first version that walked
// the rows four at a time), row lanes combined through LDS in a fixed order.  Every workgroup of a launch runs the same code on
// the same rows: all of them get the same bits.  Ends with a barrier.
constexpr int kFoldMaxC = 640;
__device__ __forceinline__ void fold_colsums(const float* __restrict__ partial, int rows, int C, double* colsum, double* scratch /*[256 * 4]*/)
{
    const int ncg = (2 * C) / 4;
    for (int cg0 = 0; cg0 < ncg; cg0 += 256) {
        const int width = ncg - cg0 < 256 ? ncg - cg0 : 256;      // 32, 64, 128 or 256 (C = 640: 256 then 64)
        const int nl = 256 / width;
        const int cg = cg0 + (int)threadIdx.x % width, lane = (int)threadIdx.x / width;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        const f32x4* base = reinterpret_cast<const f32x4*>(partial) + cg;
        for (int r = lane; r < rows; r += 16 * nl) {
            f32x4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int rr = r + u * nl;
                v[u] = base[(size_t)(rr < rows ? rr : r) * ncg];       // (clamped: always a valid row; dropped below)
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (r + u * nl < rows) { a0 += (double)v[u][0]; a1 += (double)v[u][1]; a2 += (double)v[u][2]; a3 += (double)v[u][3]; }
            }
        }
        if (nl > 1) {
            double* sp = scratch + (size_t)threadIdx.x * 4;
            sp[0] = a0; sp[1] = a1; sp[2] = a2; sp[3] = a3;
            __syncthreads();
            if (lane == 0) {
                for (int k = 1; k < nl; ++k) {
                    const double* q = scratch + (size_t)(k * width + (int)threadIdx.x) * 4;
                    a0 += q[0]; a1 += q[1]; a2 += q[2]; a3 += q[3];
                }
            }
        }
        if (lane == 0) { double* d = colsum + (size_t)cg * 4; d[0] = a0; d[1] = a1; d[2] = a2; d[3] = a3; }
        __syncthreads();
    }
}

// the forward finalize of one BatchNorm inside a consumer: scale / shift of all C channels -> sc[], sh[] (LDS); `writer` (workgroup 0)
// also does bn_finalize_k's global writes.  Ends with a barrier.
__device__ __forceinline__ void fold_finalize(const BnFinalizeArgs& f, float* sc, float* sh, double* colsum, double* scratch, bool writer)
{
    const int C = f.C;
    fold_colsums(f.partial, f.rows, C, colsum, scratch);
    if (writer && threadIdx.x == 0 && f.num_batches_tracked) *f.num_batches_tracked += 1;
    for (int c = threadIdx.x; c < C; c += 256) {
        const double s1 = colsum[c], s2 = colsum[C + c];
        const double n = (double)f.count;
        const double m = s1 / n;
        double var = s2 / n - m * m;
        if (var < 0.0) var = 0.0;
        const float mean = (float)m;
        const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
        const float g = f.gamma ? f.gamma[c] : 1.f;
        const float b = f.beta ? f.beta[c] : 0.f;
        const float scv = g * invstd;
        sc[c] = scv;
        sh[c] = b - mean * scv;
        if (writer) {
            if (f.running_mean) {
                const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
                f.running_mean[c] = (float)((1.0 - f.momentum) * (double)f.running_mean[c] + f.momentum * m);
                f.running_var[c] = (float)((1.0 - f.momentum) * (double)f.running_var[c] + f.momentum * unb);
            }
            if (f.save_mean) { f.save_mean[c] = mean; f.save_invstd[c] = invstd; }
            f.scale[c] = scv;
            f.shift[c] = b - mean * scv;
        }
    }
    __syncthreads();
}

// the backward finalize inside bn_bwd_apply: A, k1, k2 of all C channels -> LDS; workgroup 0 writes dgamma / dbeta (and the
// coefficient vectors, which nothing else reads in this form but the introspection / tests may)
__device__ __forceinline__ void fold_bwd_finalize(const BnBwdFinalizeArgs& f, float* sA, float* sK1, float* sK2, double* colsum, double* scratch,
                                                  bool writer)
{
    const int C = f.C;
    fold_colsums(f.partial, f.rows, C, colsum, scratch);
    for (int c = threadIdx.x; c < C; c += 256) {
        const double s1 = colsum[c], s2 = colsum[C + c];
        const double n = (double)f.count;
        const double g = f.gamma ? (double)f.gamma[c] : 1.0;
        const float A = (float)(g * (double)f.invstd[c]);
        const float k1 = f.train ? (float)(s1 / n) : 0.f, k2 = f.train ? (float)(s2 / n) : 0.f;
        sA[c] = A; sK1[c] = k1; sK2[c] = k2;
        if (writer) {
            if (f.dbeta) f.dbeta[c] = (float)s1;
            if (f.dgamma) f.dgamma[c] = (float)s2;
            if (f.coefA) { f.coefA[c] = A; f.coefB[c] = k1; f.coefD[c] = k2; }
        }
    }
    __syncthreads();
}

// ---- eval mode: every BatchNorm of the network in one launch (blockIdx.x = BatchNorm, threads stride channels) ----------
__global__ __launch_bounds__(256) void bn_eval_prep_k(BnEvalArgs a)
{
    const BnEvalItem& it = a.item[blockIdx.x];
    for (int c = threadIdx.x; c < it.C; c += 256) {
        const float mean = it.running_mean[c];
        const float invstd = 1.0f / sqrtf(it.running_var[c] + a.eps);     // same expression as bn_finalize_k's eval branch
        const float g = it.gamma ? it.gamma[c] : 1.f;
        const float b = it.beta ? it.beta[c] : 0.f;
        const float sc = g * invstd;
        it.scale[c] = sc;
        it.shift[c] = b - mean * sc;
        it.mean[c] = mean;
        it.invstd[c] = invstd;
    }
}

// ---- elementwise apply: y = relu?(x*s + t (+ r [* rs + rt])) ------------------
// The launcher makes the grid stride a whole number of pixels, so a thread stays on one channel group: its coefficient vectors are
// loaded once (per iteration they were 4 + 4 more 16-byte loads next to the 1 + 1 that carry data), and both data loads of an
// iteration are issued before either is used (RES is a template flag: a runtime `if (resid)` between them split the loop body and
// put an `s_waitcnt vmcnt(0)` behind each load -- one memory latency per 16 bytes and wave, ~4.5 TB/s with every wave slot taken).
template <typename T, bool RES, bool FOLD = false>
__global__ __launch_bounds__(256) void bn_apply_k(BnApplyArgs a)
{
    constexpr int V = Act<T>::kVec;          // 16-byte accesses: 4 f32 or 8 bf16 channels per thread
    using vec = typename Act<T>::vec;
    using PV = ParamVec<V>;
    const T* x = static_cast<const T*>(a.x);
    const T* resid = static_cast<const T*>(a.resid);
    T* y = static_cast<T*>(a.y);
    const int cvn = a.C / V;
    const long long total = a.pixels * cvn;
    const long long stride = (long long)gridDim.x * blockDim.x;      // a multiple of cvn (lbc_bn_apply)
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(i0 % cvn) * V;
    const float* scp = a.scale; const float* shp = a.shift; const float* rscp = a.rscale; const float* rshp = a.rshift;
    if constexpr (FOLD) {
        // this launch is also the finalize of its BatchNorm(s): coefficients from the partial rows, per workgroup (lbc_kernels.hpp)
        __shared__ __attribute__((aligned(16))) float fsc[2][kFoldMaxC], fsh[2][kFoldMaxC];
        __shared__ double fcol[2 * kFoldMaxC], fscr[256 * 4];
        if (a.fold) { fold_finalize(a.fin, fsc[0], fsh[0], fcol, fscr, blockIdx.x == 0); scp = fsc[0]; shp = fsh[0]; }
        if (RES && a.rfold) { fold_finalize(a.rfin, fsc[1], fsh[1], fcol, fscr, blockIdx.x == 0); rscp = fsc[1]; rshp = fsh[1]; }
    }
    const vec sc = PV::ld(scp + c), sh = PV::ld(shp + c);
    vec rsc = PV::splat(1.f), rsh = PV::splat(0.f);
    if (RES && rscp) { rsc = PV::ld(rscp + c); rsh = PV::ld(rshp + c); }
    const bool relu = a.relu != 0;
    for (long long i = i0; i < total; i += stride) {
        // (x and the residual are not read again before the backward pass: nontemporal, they need not displace what the next
        //  convolution is about to read -- measured 15.19 -> 15.12 ms per step, profiles/r04_run9_*)
        vec v = Act<T>::cvt(__builtin_nontemporal_load(reinterpret_cast<const typename Act<T>::raw*>(x + i * V)));
        vec r = v;
        if (RES) r = Act<T>::cvt(__builtin_nontemporal_load(reinterpret_cast<const typename Act<T>::raw*>(resid + i * V)));
        v = v * sc + sh;
        if (RES) v += r * rsc + rsh;
        if (relu) {
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        Act<T>::stv(y + i * V, v);
    }
}

// ---- per-channel reductions over pixels ----------------------------------------
// Block layout: 256 threads = (C/4 channel groups) x RL pixel lanes (RL = 256/(C/4),
// C/4 <= 256).  Each block owns a contiguous pixel range and writes one partial row
// [2][C]:  row0 = sum g, row1 = sum g*q  where the meaning of g, q depends on the op.
// MASK / HASX / GOUT (which of a.mask, a.x, a.g_out are present) are template flags: as runtime null checks they sat between the
// loads of a batch and the compiler answered each with a branch and an `s_waitcnt vmcnt(1)` -- the batch was loaded one pair at a time.
template <int OP, typename T, bool MASK, bool HASX, bool GOUT>
__global__ __launch_bounds__(256) void channel_reduce_k(ChanReduceArgs a)
{
    constexpr int V = Act<T>::kVec;
    using vec = typename Act<T>::vec;
    using PV = ParamVec<V>;
    __shared__ __attribute__((aligned(16))) float red[2 * 256 * V];
    const T* xx = static_cast<const T*>(a.x);
    const T* dz = static_cast<const T*>(a.dz);
    const T* mask = static_cast<const T*>(a.mask);
    T* g_out = static_cast<T*>(a.g_out);
    const int cvn = a.C / V;
    const int rl = 256 / cvn;
    const int cg = threadIdx.x % cvn;
    const int pl = threadIdx.x / cvn;
    const int c = cg * V;
    vec s1 = PV::splat(0.f), s2 = s1;
    if (pl < rl) {
        vec mean = PV::splat(0.f), inv = PV::splat(1.f);
        vec msc = PV::splat(1.f), msh = PV::splat(0.f);
        if (OP == 1 && a.mean) { mean = PV::ld(a.mean + c); inv = PV::ld(a.invstd + c); }
        if (OP == 1 && a.mask_scale) { msc = PV::ld(a.mask_scale + c); msh = PV::ld(a.mask_shift + c); }
        const long long p0 = (long long)blockIdx.x * a.pix_per_block;
        long long p1 = p0 + a.pix_per_block;
        if (p1 > a.pixels) p1 = a.pixels;
        // batches of kU pixel steps with every load in front of the first store (g_out aliases dz in the executor)
        constexpr int kU = 4;
        auto one = [&](long long i, vec g, vec m, vec v) {
            if (OP == 0) {            // plain statistics of x: sum x, sum x^2
                s1 += v;
                s2 += v * v;
            } else {                  // backward: g = dz * (mask > 0); sum g, sum g * xhat
                if (MASK) {
                    m = m * msc + msh;                 // (1, 0) without a mask transform: exact
#pragma unroll
                    for (int e = 0; e < V; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
                }
                if (GOUT) Act<T>::stv(g_out + i, g);
                s1 += g;
                if (HASX) s2 += g * (v - mean) * inv;
            }
        };
        long long p = p0 + pl;
        for (; p + (kU - 1) * rl < p1; p += kU * rl) {
            typename Act<T>::raw g[kU], m[kU], v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const long long i = ((p + u * rl) * cvn + cg) * V;
                if (OP == 0) { v[u] = Act<T>::ldr(xx + i); g[u] = v[u]; m[u] = v[u]; }
                else {
                    g[u] = Act<T>::ldr(dz + i);
                    m[u] = MASK ? Act<T>::ldr(mask + i) : g[u];
                    v[u] = HASX ? Act<T>::ldr(xx + i) : g[u];
                }
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) one(((p + u * rl) * cvn + cg) * V, Act<T>::cvt(g[u]), Act<T>::cvt(m[u]), Act<T>::cvt(v[u]));
        }
        for (; p < p1; p += rl) {
            const long long i = (p * cvn + cg) * V;
            if (OP == 0) { const vec v = Act<T>::ldv(xx + i); one(i, v, v, v); }
            else {
                const vec g = Act<T>::ldv(dz + i);
                one(i, g, MASK ? Act<T>::ldv(mask + i) : g, HASX ? Act<T>::ldv(xx + i) : g);
            }
        }
    }
    PV::st(red + threadIdx.x * V, s1);
    PV::st(red + (256 + threadIdx.x) * V, s2);
    __syncthreads();
    if (threadIdx.x < cvn) {
        vec t1 = PV::splat(0.f), t2 = t1;
        for (int k = 0; k < rl; ++k) {
            t1 += PV::ld(red + (k * cvn + threadIdx.x) * V);
            t2 += PV::ld(red + (256 + k * cvn + threadIdx.x) * V);
        }
        float* dst = a.partial + (size_t)blockIdx.x * 2 * a.C;
        PV::st(dst + c, t1);
        PV::st(dst + a.C + c, t2);
    }
}

// ---- backward finalize: partial rows -> dgamma, dbeta and the apply coefficients --
//   dx = A*(g - k1 - xhat*k2),  A = gamma*invstd, k1 = sum(g)/n, k2 = sum(g*xhat)/n,
//   xhat recomputed per element as (x-mean)*invstd (factoring it into B*x + D would put a
//   systematic per-channel rounding offset on dx that downstream channel sums amplify).
__global__ __launch_bounds__(1024) void bn_bwd_finalize_k(BnBwdFinalizeArgs a)
{
    const int c = blockIdx.x * kFinCh + (threadIdx.x & (kFinCh - 1));
    double s1, s2;
    sum_rows2(a.partial, a.rows, a.C, c, c < a.C, s1, s2);
    if (c >= a.C || threadIdx.x >= kFinCh) return;
    if (a.dbeta) a.dbeta[c] = (float)s1;
    if (a.dgamma) a.dgamma[c] = (float)s2;
    if (a.coefA) {
        const double n = a.nsum ? (double)a.count / (double)a.n_local * (double)*a.nsum : (double)a.count;
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
// Same structure as bn_apply_k: grid stride = whole pixels (of the Cout / V output groups), the five coefficient vectors loaded
// once (they were 10 of the 12 loads of an iteration in the bf16 kernel), no 64-bit division per iteration, and every data load of
// an iteration issued before the first use (MASK / ACCUM are template flags).
template <typename T, bool MASK, bool ACCUM, bool FOLD = false>
__global__ __launch_bounds__(256, (MASK || ACCUM || FOLD) ? 1 : 8) void bn_bwd_apply_k(BnBwdApplyArgs a)
// (8 waves per SIMD = at most 64 VGPRs for the plain variant: its bf16 form sits at 66 without the bound and fits without scratch with it;
// the MASK / ACCUM forms would spill and keep 6-7 waves)
{
    constexpr int V = Act<T>::kVec;
    using vec = typename Act<T>::vec;
    using PV = ParamVec<V>;
    const T* gp = static_cast<const T*>(a.g);
    const T* mask = static_cast<const T*>(a.mask);
    const T* xx = static_cast<const T*>(a.x);
    T* dx = static_cast<T*>(a.dx);
    const int cvn = a.C / V;
    const int ovn = a.Cout / V;
    const long long total = a.pixels * ovn;
    const long long stride = (long long)gridDim.x * blockDim.x;      // a multiple of ovn (lbc_bn_bwd_apply)
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = (int)(i0 % ovn), c = cg * V;
    const float* pA = a.coefA; const float* pK1 = a.coefB; const float* pK2 = a.coefD;
    if constexpr (FOLD) {
        __shared__ __attribute__((aligned(16))) float fA[kFoldMaxC], fK1[kFoldMaxC], fK2[kFoldMaxC];
        __shared__ double fcol[2 * kFoldMaxC], fscr[256 * 4];
        fold_bwd_finalize(a.fin, fA, fK1, fK2, fcol, fscr, blockIdx.x == 0);
        pA = fA; pK1 = fK1; pK2 = fK2;
    }
    const vec cA = PV::ld(pA + c), k1 = PV::ld(pK1 + c), k2 = PV::ld(pK2 + c);
    const vec mean = PV::ld(a.mean + c), inv = PV::ld(a.invstd + c);
    long long j = ((i0 / ovn) * cvn + cg) * V;                       // element offset in the C-channel tensors
    const long long dj = (stride / ovn) * cvn * V;
    for (long long i = i0; i < total; i += stride, j += dj) {
        vec g = Act<T>::ldv(gp + j);
        const vec v = Act<T>::cvt(__builtin_nontemporal_load(reinterpret_cast<const typename Act<T>::raw*>(xx + j)));     // (the pre-BatchNorm activation: dead after this pass)
        vec m = g, old = g;
        if (MASK) m = Act<T>::ldv(mask + j);
        if (ACCUM) old = Act<T>::ldv(dx + i * V);
        if (MASK) {
#pragma unroll
            for (int e = 0; e < V; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
        }
        vec o = cA * (g - k1 - (v - mean) * inv * k2);
        if (ACCUM) o += old;
        Act<T>::stv(dx + i * V, o);
    }
}

// ---- velocity late fusion: h = cat(trunk, speed broadcast to 128 channels) --------
// reference bird_view/models/image.py:77-79 / birdview.py:67-69
template <typename T>
__global__ __launch_bounds__(256) void concat_velocity_k(const void* tv, const float* __restrict__ vel, void* hv, long long pixels,
                                                         int hw, int Ct, int Cv)
{
    const T* t = static_cast<const T*>(tv);
    T* h = static_cast<T*>(hv);
    const int C = Ct + Cv;
    const int c4n = C / 4;
    const long long total4 = pixels * c4n;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
        const long long p = i / c4n;
        const int c = (int)(i - p * c4n) * 4;
        f32x4 v;
        if (c < Ct) v = Act<T>::ld4(t + p * Ct + c);
        else { const float s = vel[p / hw]; v = f32x4{s, s, s, s}; }
        Act<T>::st4(h + i * 4, v);
    }
}

__global__ __launch_bounds__(256) void copy_f32_k(const float* __restrict__ src, float* __restrict__ dst, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

int grid_for(long long total4)
{
    long long b = (total4 + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

// as grid_for, with grid * 256 a multiple of `groups` (the 16-byte channel groups of a pixel): a thread of a grid-stride loop then
// stays on one channel group.  groups = 8 .. 64 divide 256; 80 (640 channels) needs a multiple of 5 blocks.
int grid_for_groups(long long total, int groups)
{
    int g = 256, r = groups;
    while (r) { const int t = g % r; g = r; r = t; }       // g = gcd(256, groups)
    const int unit = groups / g;
    long long b = grid_for(total);
    b = b / unit * unit;
    if (b < unit) b = unit;
    return (int)b;
}

}  // namespace

int lbc_copy_f32(const float* src, float* dst, long long n, hipStream_t s)
{
    LBC_REQUIRE(src && dst && n > 0, "copy_f32: bad arguments");
    hipLaunchKernelGGL(copy_f32_k, dim3((unsigned)grid_for(n)), dim3(256), 0, s, src, dst, n);
    return lbc_check_launch("copy_f32");
}

int lbc_partial_reduce(const float* in, int rows, int cols, float* out, int out_rows, hipStream_t s, float* copy_lo, float* copy_hi, float tail)
{
    LBC_REQUIRE((!copy_lo && !copy_hi) || (out_rows == 1 && cols % 2 == 0), "partial_reduce: the copy outputs need one output row");
    LBC_REQUIRE(tail < 0.f || out_rows == 1, "partial_reduce: the tail value needs one output row");
    dim3 grid((unsigned)lbc_cdiv(cols, 256), (unsigned)out_rows);
    LbcProfScope prof("partial_reduce", 0.0, 4.0 * (double)rows * cols, s);
    hipLaunchKernelGGL(partial_reduce_k, grid, dim3(256), 0, s, in, rows, cols, out, copy_lo, copy_hi, tail);
    return lbc_check_launch("partial_reduce");
}

int lbc_bn_finalize(const BnFinalizeArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.C > 0 && a.scale && a.shift, "bn_finalize: bad args");
    LbcProfScope prof("bn_finalize", 0.0, 4.0 * (double)a.rows * 2 * a.C, s);
    hipLaunchKernelGGL(bn_finalize_k, dim3((unsigned)lbc_cdiv(a.C, kFinCh)), dim3(1024), 0, s, a);
    return lbc_check_launch("bn_finalize");
}

int lbc_bn_eval_prep(const BnEvalArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.count >= 1 && a.count <= BnEvalArgs::kMax, "bn_eval_prep: bad table");
    LbcProfScope prof("bn_eval_prep", 0.0, 0.0, s);
    hipLaunchKernelGGL(bn_eval_prep_k, dim3((unsigned)a.count), dim3(256), 0, s, a);
    return lbc_check_launch("bn_eval_prep");
}

// Folding a finalize into its consumer: every workgroup reads rows x 2C floats from L2 (64 B / clk / CU) -- worth it where that is
// a microsecond and the consumer's grid is small, i.e. where launches are the cost
// (round 5 swept both at 32 / 64 images per GPU -- grid 256 / 512 / 1024, 64 / 128 / 256 KB of rows, no fold at all: every arm within
//  0.5 % of the others, profiles/r05_call5_*: the fold costs what the finalize launches cost)
constexpr long long kFoldBytes = 128 * 1024;
constexpr int kFoldGrid = 512;           // workgroups of a folding consumer (two per CU)
int lbc_bn_fold_max_rows(int C) { return (int)(kFoldBytes / (8ll * C)); }
bool lbc_bn_fold_ok(int rows, int C)
{
    const int ncg = C / 2, tail = ncg % 256;          // 16-byte column groups of a row; every pass of fold_colsums needs a width that divides 256
    return !lbc_opt_on(kOptNoBnFold) && C % 8 == 0 && C <= kFoldMaxC && (tail == 0 || 256 % tail == 0) && rows >= 1 && rows <= lbc_bn_fold_max_rows(C);
}

int lbc_bn_apply(const BnApplyArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.C % 8 == 0 && a.pixels > 0, "bn_apply: bad shape");
    const bool fold = a.fold || a.rfold;
    LBC_REQUIRE(!fold || ((!a.fold || lbc_bn_fold_ok(a.fin.rows, a.C)) && (!a.rfold || (a.resid && lbc_bn_fold_ok(a.rfin.rows, a.C)))),
                "bn_apply: folded finalize outside its limits");
    LBC_REQUIRE(!fold || ((!a.fold || (a.fin.C == a.C && a.fin.train && !a.fin.nsum)) && (!a.rfold || (a.rfin.C == a.C && a.rfin.train && !a.rfin.nsum))),
                "bn_apply: folded finalize needs a local training-mode BatchNorm of the same width");
    LbcProfScope prof("bn_apply", 0.0, (a.act_bf16 ? 2.0 : 4.0) * (double)a.pixels * a.C * (a.resid ? 3 : 2), s);
    const int groups = a.C / (a.act_bf16 ? 8 : 4);
    int grid = grid_for_groups(a.pixels * groups, groups);
    if (fold && grid > kFoldGrid) {          // (a multiple of the channel-group unit below the cap)
        int g = 256, r = groups;
        while (r) { const int t = g % r; g = r; r = t; }
        const int unit = groups / g;
        grid = kFoldGrid / unit * unit;
        if (grid < unit) grid = unit;
    }
#define LBC_K(T, g)                                                                                                          \
    do {                                                                                                                     \
        if (fold) {                                                                                                          \
            if (a.resid) hipLaunchKernelGGL((bn_apply_k<T, true, true>), dim3((unsigned)(g)), dim3(256), 0, s, a);           \
            else         hipLaunchKernelGGL((bn_apply_k<T, false, true>), dim3((unsigned)(g)), dim3(256), 0, s, a);          \
        } else {                                                                                                             \
            if (a.resid) hipLaunchKernelGGL((bn_apply_k<T, true>), dim3((unsigned)(g)), dim3(256), 0, s, a);                 \
            else         hipLaunchKernelGGL((bn_apply_k<T, false>), dim3((unsigned)(g)), dim3(256), 0, s, a);                \
        }                                                                                                                    \
    } while (0)
    LBC_DISPATCH_ACT(a.act_bf16, LBC_K, grid);
#undef LBC_K
    return lbc_check_launch("bn_apply");
}

int lbc_chan_reduce_rows(long long pixels, int C, int max_rows)
{
    if (max_rows > 0) {          // a consumer folds the finalize (lbc_bn_fold_ok): few rows, whole pixel-lane groups per workgroup
        const int rl0 = 256 / (C / 4);
        long long rows0 = (pixels + rl0 - 1) / rl0;
        if (rows0 > max_rows) rows0 = max_rows;
        return (int)(rows0 < 1 ? 1 : rows0);
    }
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
    LBC_REQUIRE(a.C % 8 == 0 && a.C / 4 <= 256, "chan_reduce: C=%d unsupported", a.C);
    const int rows = lbc_chan_reduce_rows(a.pixels, a.C, a.max_rows);
    a.pix_per_block = (a.pixels + rows - 1) / rows;
    LbcProfScope prof(op == 0 ? "channel_stats" : "bn_bwd_reduce", 0.0,
                      (a.act_bf16 ? 2.0 : 4.0) * (double)a.pixels * a.C * (op == 0 ? 1 : (1 + (a.mask ? 1 : 0) + (a.x ? 1 : 0) + (a.g_out ? 1 : 0))), s);
#define LBC_KF(T, OPV, M, X, G) hipLaunchKernelGGL((channel_reduce_k<OPV, T, M, X, G>), dim3((unsigned)rows), dim3(256), 0, s, a)
#define LBC_K(T, OPV)                                                                                    \
    do {                                                                                                 \
        const int f = OPV == 0 ? 2 : (a.mask ? 4 : 0) | (a.x ? 2 : 0) | (a.g_out ? 1 : 0);              \
        switch (f) {                                                                                     \
        case 0: LBC_KF(T, OPV, false, false, false); break;                                              \
        case 1: LBC_KF(T, OPV, false, false, true); break;                                               \
        case 2: LBC_KF(T, OPV, false, true, false); break;                                               \
        case 3: LBC_KF(T, OPV, false, true, true); break;                                                \
        case 4: LBC_KF(T, OPV, true, false, false); break;                                               \
        case 5: LBC_KF(T, OPV, true, false, true); break;                                                \
        case 6: LBC_KF(T, OPV, true, true, false); break;                                                \
        default: LBC_KF(T, OPV, true, true, true); break;                                                \
        }                                                                                                \
    } while (0)
    if (op == 0) {
        LBC_REQUIRE(a.x, "channel_stats: null input");
        if (a.act_bf16) LBC_KF(__bf16, 0, false, true, false);
        else            LBC_KF(float, 0, false, true, false);
    } else {
        LBC_DISPATCH_ACT(a.act_bf16, LBC_K, 1);
    }
#undef LBC_K
#undef LBC_KF
    return lbc_check_launch("channel_reduce");
}

int lbc_bn_bwd_finalize(const BnBwdFinalizeArgs& a, hipStream_t s)
{
    LbcProfScope prof("bn_bwd_finalize", 0.0, 4.0 * (double)a.rows * 2 * a.C, s);
    hipLaunchKernelGGL(bn_bwd_finalize_k, dim3((unsigned)lbc_cdiv(a.C, kFinCh)), dim3(1024), 0, s, a);
    return lbc_check_launch("bn_bwd_finalize");
}

int lbc_bn_bwd_apply(const BnBwdApplyArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.C % 8 == 0 && a.Cout % 8 == 0 && a.Cout <= a.C, "bn_bwd_apply: bad channels");
    if (a.fold) {
        LBC_REQUIRE(!a.mask && !a.accum && a.fin.C == a.C && !a.fin.nsum && lbc_bn_fold_ok(a.fin.rows, a.C), "bn_bwd_apply: folded finalize outside its limits");
        LbcProfScope prof("bn_bwd_apply", 0.0, (a.act_bf16 ? 2.0 : 4.0) * (double)a.pixels * (a.C * 2.0 + a.Cout), s);
        const int groups = a.Cout / (a.act_bf16 ? 8 : 4);
        int grid = grid_for_groups(a.pixels * groups, groups);
        if (grid > kFoldGrid) {
            int g = 256, r = groups;
            while (r) { const int t = g % r; g = r; r = t; }
            const int unit = groups / g;
            grid = kFoldGrid / unit * unit;
            if (grid < unit) grid = unit;
        }
#define LBC_K(T, g) hipLaunchKernelGGL((bn_bwd_apply_k<T, false, false, true>), dim3((unsigned)(g)), dim3(256), 0, s, a)
        LBC_DISPATCH_ACT(a.act_bf16, LBC_K, grid);
#undef LBC_K
        return lbc_check_launch("bn_bwd_apply");
    }
    LbcProfScope prof("bn_bwd_apply", 0.0, (a.act_bf16 ? 2.0 : 4.0) * (double)a.pixels * (a.C * (a.mask ? 3.0 : 2.0) + a.Cout * (a.accum ? 2.0 : 1.0)), s);
    const int groups = a.Cout / (a.act_bf16 ? 8 : 4);
#define LBC_K(T, g)                                                                                                      \
    do {                                                                                                                 \
        if (a.mask && a.accum)  hipLaunchKernelGGL((bn_bwd_apply_k<T, true, true>), dim3((unsigned)(g)), dim3(256), 0, s, a);   \
        else if (a.mask)        hipLaunchKernelGGL((bn_bwd_apply_k<T, true, false>), dim3((unsigned)(g)), dim3(256), 0, s, a);  \
        else if (a.accum)       hipLaunchKernelGGL((bn_bwd_apply_k<T, false, true>), dim3((unsigned)(g)), dim3(256), 0, s, a);  \
        else                    hipLaunchKernelGGL((bn_bwd_apply_k<T, false, false>), dim3((unsigned)(g)), dim3(256), 0, s, a); \
    } while (0)
    LBC_DISPATCH_ACT(a.act_bf16, LBC_K, grid_for_groups(a.pixels * groups, groups));
#undef LBC_K
    return lbc_check_launch("bn_bwd_apply");
}

int lbc_concat_velocity(const void* t, const float* vel, void* h, int N, int hw, int Ct, int Cv, int act_bf16, hipStream_t s)
{
    LBC_REQUIRE(Ct % 4 == 0 && Cv % 4 == 0, "concat_velocity: channels must be multiples of 4");
    const long long pixels = (long long)N * hw;
    LbcProfScope prof("concat_velocity", 0.0, (act_bf16 ? 2.0 : 4.0) * (double)pixels * (Ct + Ct + Cv), s);
#define LBC_K(T, g) hipLaunchKernelGGL((concat_velocity_k<T>), dim3((unsigned)(g)), dim3(256), 0, s, t, vel, h, pixels, hw, Ct, Cv)
    LBC_DISPATCH_ACT(act_bf16, LBC_K, grid_for(pixels * ((Ct + Cv) / 4)));
#undef LBC_K
    return lbc_check_launch("concat_velocity");
}
