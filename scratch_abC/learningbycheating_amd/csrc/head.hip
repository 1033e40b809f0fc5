// Conditional-branch waypoint head for gfx950: for each of the 4 command branches
//   BatchNorm2d(64) -> Conv2d(64,5,1) -> SpatialSoftmax -> (x^, y^) per step,
// stacked to (N,4,5,2) and reduced with the one-hot command to (N,5,2).
// reference: bird_view/models/image.py:54-60,82-84 ; birdview.py:53-58,72-75 ;
// common.py:29-35 (select_branch), 112-152 (SpatialSoftmax).
//
// The BatchNorm is folded into the 1x1 weights per workgroup (64x5 per branch), the
// decoder output is streamed once through LDS in 256-pixel tiles, and the softmax
// expectation is computed online (running max / sum / sum*pos), so no logit map is
// ever written to HBM.  The backward recomputes the logits from the same tiles.
//
// Backward algebra (training mode: the 4 BatchNorms see the same batch statistics
// but have their own gamma/beta).  With xh = (h-mean)*invstd, G = upstream gradient
// of (x^,y^) incl. the branch-select term, p = softmax probability:
//   dlogit[b,s,pix] = p * ((Gx*px + Gy*py) - (Gx*x^ + Gy*y^))
//   S0[b,s] = sum dlogit ,  S1[b,s,c] = sum dlogit * xh[c]
//   dW = gamma*S1 + beta*S0 ; dbias = S0 ; dgamma_b = sum_s W*S1 ; dbeta_b = sum_s W*S0
//   dh[pix,c] = sum_{b,s} dlogit * (W*gamma*invstd) - invstd*(k1 + k2*xh[pix,c])
//   k1 = sum_b gamma_b*dbeta_b / n, k2 = sum_b gamma_b*dgamma_b / n.
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include "lbc_kernels.hpp"
#include <stdlib.h>

namespace {

constexpr int TP = 256;        // pixels per tile
constexpr int LDH = 68;        // padded LDS row of 64 channels

struct SoftAcc { float m, l, sx, sy; };

__device__ __forceinline__ void soft_merge(SoftAcc& a, const SoftAcc& b)
{
    const float M = fmaxf(a.m, b.m);
    if (M == -INFINITY) return;
    const float fa = __expf(a.m - M), fb = __expf(b.m - M);
    a.l = a.l * fa + b.l * fb;
    a.sx = a.sx * fa + b.sx * fb;
    a.sy = a.sy * fa + b.sy * fb;
    a.m = M;
}

template <typename T>
__device__ __forceinline__ void load_tile(const void* hv, float* sH, int n, int HW, int tile, int tid)
{
    const T* h = static_cast<const T*>(hv);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int idx = tid + 256 * j;
        const int row = idx >> 4, sg = idx & 15;
        const int p = tile * TP + row;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (p < HW) v = Act<T>::ld4(h + ((size_t)n * HW + (size_t)p) * 64 + (size_t)(sg * 4));
        *reinterpret_cast<f32x4*>(&sH[row * LDH + sg * 4]) = v;
    }
}

// folded weights of one branch: Wf[s][c] = W*gamma*invstd, bf[s] = bias + sum_c W*(beta - gamma*mean*invstd)
__device__ __forceinline__ void fold_branch(const HeadArgs& a, int b, float* sW /*[5*64]*/, float* sB /*[5]*/, int tid, int nthr)
{
    const float* mean = a.mean[b];
    const float* inv = a.invstd[b];
    for (int idx = tid; idx < 320; idx += nthr) {
        const int c = idx & 63;
        const float wf = a.w[b][idx] * a.gamma[b][c] * inv[c];
        // bf16 activations: the forward multiplies on the bf16 MFMA, so the folded weight is a value that MFMA can multiply with
        // everywhere (forward, and the backward's recomputation of the logits) -- the saved soft-max statistics stay consistent.
        // wsplit (default): the sum of a bf16 high part and a bf16 low part (two MFMAs per k-block: ~16 significant bits, the
        // head is as accurate as the f32 kernels on the same bf16 decoder output; it streams h from HBM, the second MFMA is
        // free); otherwise one bf16 value (2^-9 relative: a FIXED perturbation of the 64 -> 5 projection that no master-weight
        // update below half a bf16 ulp can reach)
        float v = wf;
        if (a.act_bf16) {
            const float hi = (float)(__bf16)wf;
            v = a.wsplit ? hi + (float)(__bf16)(wf - hi) : hi;
        }
        sW[idx] = v;
    }
    if (tid < 5) {
        float t = a.bias[b][tid];
        for (int c = 0; c < 64; ++c)
            t += a.w[b][tid * 64 + c] * (a.beta[b][c] - a.gamma[b][c] * mean[c] * inv[c]);
        sB[tid] = t;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void head_fwd_k(HeadArgs a)
{
    __shared__ __attribute__((aligned(16))) float sH[TP * LDH];
    __shared__ __attribute__((aligned(16))) float sW[320];
    __shared__ float sB[8];
    __shared__ float sRed[4 * 5 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x, b = blockIdx.y;
    const int HW = a.OH * a.OW;
    fold_branch(a, b, sW, sB, tid, 256);
    __syncthreads();

    SoftAcc st[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) { st[s].m = -INFINITY; st[s].l = 0.f; st[s].sx = 0.f; st[s].sy = 0.f; }

    const int ntile = (HW + TP - 1) / TP;
    for (int tile = 0; tile < ntile; ++tile) {
        load_tile<T>(a.h, sH, n, HW, tile, tid);
        __syncthreads();
        const int p = tile * TP + tid;
        if (p < HW) {
            float lg[5];
#pragma unroll
            for (int s = 0; s < 5; ++s) lg[s] = sB[s];
#pragma unroll
            for (int c4 = 0; c4 < 16; ++c4) {
                const float4 f = *reinterpret_cast<const float4*>(&sH[tid * LDH + c4 * 4]);
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    const float4 w = *reinterpret_cast<const float4*>(&sW[s * 64 + c4 * 4]);
                    lg[s] += f.x * w.x + f.y * w.y + f.z * w.z + f.w * w.w;
                }
            }
            const float px = a.pos_x[b][p], py = a.pos_y[b][p];
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                SoftAcc o; o.m = lg[s]; o.l = 1.f; o.sx = px; o.sy = py;
                soft_merge(st[s], o);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < 5; ++s) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            SoftAcc o;
            o.m = __shfl_xor(st[s].m, off); o.l = __shfl_xor(st[s].l, off);
            o.sx = __shfl_xor(st[s].sx, off); o.sy = __shfl_xor(st[s].sy, off);
            soft_merge(st[s], o);
        }
        if (lane == 0) {
            sRed[(wave * 5 + s) * 4 + 0] = st[s].m; sRed[(wave * 5 + s) * 4 + 1] = st[s].l;
            sRed[(wave * 5 + s) * 4 + 2] = st[s].sx; sRed[(wave * 5 + s) * 4 + 3] = st[s].sy;
        }
    }
    __syncthreads();
    if (tid < 5) {
        SoftAcc t; t.m = sRed[tid * 4]; t.l = sRed[tid * 4 + 1]; t.sx = sRed[tid * 4 + 2]; t.sy = sRed[tid * 4 + 3];
        for (int w = 1; w < 4; ++w) {
            SoftAcc o; o.m = sRed[(w * 5 + tid) * 4]; o.l = sRed[(w * 5 + tid) * 4 + 1];
            o.sx = sRed[(w * 5 + tid) * 4 + 2]; o.sy = sRed[(w * 5 + tid) * 4 + 3];
            soft_merge(t, o);
        }
        const size_t o2 = (((size_t)n * 4 + b) * 5 + tid) * 2;
        a.pred_all[o2] = t.sx / t.l;
        a.pred_all[o2 + 1] = t.sy / t.l;
        if (a.rowstat) { a.rowstat[o2] = t.m; a.rowstat[o2 + 1] = t.l; }
    }
}

// bf16 activations: the 64 -> 4 x 5 projection of all four branches on v_mfma_f32_32x32x16_bf16 (20 of 32 columns
// used), A fragments straight from HBM (a lane's 8 consecutive channels of one pixel are one 16-byte load: no LDS
// staging at all), folded weights stationary in registers, online soft-argmax per column.  One workgroup per image.
__global__ __launch_bounds__(256) void head_fwd_mfma_k(HeadArgs a)
{
    __shared__ float sWf[4 * 320];
    __shared__ float sBf[4 * 8];
    __shared__ float sRed[4 * 32 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int n = blockIdx.x;
    const int HW = a.OH * a.OW;
    const int S = a.nslice, slice = blockIdx.y;      // small batches: the image's pixel groups are split over S workgroups
    for (int b = 0; b < 4; ++b) fold_branch(a, b, sWf + b * 320, sBf + b * 8, tid, 256);
    __syncthreads();
    const bool colok = l31 < 20;
    const int cb = colok ? l31 / 5 : 0, cs = colok ? l31 - 5 * cb : 0;     // branch and step of this lane's column
    bf16x8 wb[4], wl[4];            // high and low bf16 parts of the folded weights (wl = 0 without wsplit)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = colok ? sWf[cb * 320 + cs * 64 + g * 16 + kh * 8 + j] : 0.f;
            wb[g][j] = (__bf16)v;
            wl[g][j] = (__bf16)(v - (float)wb[g][j]);
        }
    const bool split = a.wsplit != 0;
    const float bias = colok ? sBf[cb * 8 + cs] : 0.f;
    const float* posx = a.pos_x[cb];
    const float* posy = a.pos_y[cb];
    const __bf16* h = static_cast<const __bf16*>(a.h) + (size_t)n * HW * 64;

    SoftAcc st; st.m = -INFINITY; st.l = 0.f; st.sx = 0.f; st.sy = 0.f;
    const int ngroup = (HW + 31) / 32;
    for (int grp = slice * 4 + wave; grp < ngroup; grp += 4 * S) {
        const int pbase = grp * 32;
        const int pl = pbase + l31 < HW ? pbase + l31 : HW - 1;       // rows past the map are computed on a valid pixel and skipped below
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        bf16x8 af[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) af[g] = *reinterpret_cast<const bf16x8*>(h + (size_t)pl * 64 + (size_t)(g * 16 + kh * 8));
#pragma unroll
        for (int g = 0; g < 4; ++g) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g], wb[g], acc, 0, 0, 0);
        if (split) {
#pragma unroll
            for (int g = 0; g < 4; ++g) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g], wl[g], acc, 0, 0, 0);
        }
        // the group's 16 logits of this lane merged in ONE step of the online soft-argmax: maximum first, one rescale of the running
        // sums, 16 exponentials (element by element it took two exponentials and a rescale per logit: the kernel was VALU-bound)
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int p = pbase + (e & 3) + 8 * (e >> 2) + 4 * kh;
            acc[e] = (p < HW && colok) ? acc[e] + bias : -INFINITY;
            mx = fmaxf(mx, acc[e]);
        }
        if (mx != -INFINITY) {
            const float M = fmaxf(st.m, mx);
            const float f = __expf(st.m - M);              // (0 while nothing has been merged yet: st.m = -inf)
            st.l *= f; st.sx *= f; st.sy *= f; st.m = M;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int p = pbase + (e & 3) + 8 * (e >> 2) + 4 * kh;
                const int pc = p < HW ? p : HW - 1;
                const float w = __expf(acc[e] - M);        // (exp(-inf) = 0: padding rows and columns contribute nothing)
                st.l += w; st.sx += w * posx[pc]; st.sy += w * posy[pc];
            }
        }
    }
    {   // the two half-waves hold different pixel rows of the same column
        SoftAcc o;
        o.m = __shfl_xor(st.m, 32); o.l = __shfl_xor(st.l, 32); o.sx = __shfl_xor(st.sx, 32); o.sy = __shfl_xor(st.sy, 32);
        soft_merge(st, o);
    }
    if (kh == 0) {
        sRed[(wave * 32 + l31) * 4 + 0] = st.m; sRed[(wave * 32 + l31) * 4 + 1] = st.l;
        sRed[(wave * 32 + l31) * 4 + 2] = st.sx; sRed[(wave * 32 + l31) * 4 + 3] = st.sy;
    }
    __syncthreads();
    if (tid < 20) {
        SoftAcc t; t.m = sRed[tid * 4]; t.l = sRed[tid * 4 + 1]; t.sx = sRed[tid * 4 + 2]; t.sy = sRed[tid * 4 + 3];
        for (int w = 1; w < 4; ++w) {
            SoftAcc o; o.m = sRed[(w * 32 + tid) * 4]; o.l = sRed[(w * 32 + tid) * 4 + 1];
            o.sx = sRed[(w * 32 + tid) * 4 + 2]; o.sy = sRed[(w * 32 + tid) * 4 + 3];
            soft_merge(t, o);
        }
        if (S > 1) {
            float* dst = a.scratch + (((size_t)n * S + slice) * 20 + tid) * 4;
            dst[0] = t.m; dst[1] = t.l; dst[2] = t.sx; dst[3] = t.sy;
        } else {
            const size_t o2 = ((size_t)n * 20 + tid) * 2;      // column = branch * 5 + step
            a.pred_all[o2] = t.sx / t.l;
            a.pred_all[o2 + 1] = t.sy / t.l;
            if (a.rowstat) { a.rowstat[o2] = t.m; a.rowstat[o2 + 1] = t.l; }
        }
    }
}

// merges the per-slice partials of head_fwd_mfma_k in slice order
__global__ __launch_bounds__(256) void head_merge_k(HeadArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;   // over N*20
    if (i >= a.N * 20) return;
    const int n = i / 20, col = i - n * 20;
    const float* src = a.scratch + ((size_t)n * a.nslice * 20 + col) * 4;
    SoftAcc t; t.m = src[0]; t.l = src[1]; t.sx = src[2]; t.sy = src[3];
    for (int k = 1; k < a.nslice; ++k) {
        const float* p = src + (size_t)k * 80;
        SoftAcc o; o.m = p[0]; o.l = p[1]; o.sx = p[2]; o.sy = p[3];
        soft_merge(t, o);
    }
    a.pred_all[(size_t)i * 2] = t.sx / t.l;
    a.pred_all[(size_t)i * 2 + 1] = t.sy / t.l;
    if (a.rowstat) { a.rowstat[(size_t)i * 2] = t.m; a.rowstat[(size_t)i * 2 + 1] = t.l; }
}

__global__ __launch_bounds__(256) void select_branch_k(const float* __restrict__ all, const float* __restrict__ cmd,
                                                       float* __restrict__ sel, int N)
{
    const int i = blockIdx.x * 256 + threadIdx.x;   // over N*10
    if (i >= N * 10) return;
    const int n = i / 10, r = i - n * 10;
    float t = 0.f;
    for (int b = 0; b < 4; ++b) t += cmd[n * 4 + b] * all[(n * 4 + b) * 10 + r];
    sel[i] = t;
}

// upstream gradient of one (n, b, s) row incl. the branch-select path
__device__ __forceinline__ void row_grad(const HeadBwdArgs& a, int n, int b, int s, float& gx, float& gy)
{
    gx = 0.f; gy = 0.f;
    if (a.d_all) { gx = a.d_all[((n * 4 + b) * 5 + s) * 2]; gy = a.d_all[((n * 4 + b) * 5 + s) * 2 + 1]; }
    if (a.d_sel) {
        const float cm = a.f.cmd[n * 4 + b];
        gx += cm * a.d_sel[(n * 5 + s) * 2]; gy += cm * a.d_sel[(n * 5 + s) * 2 + 1];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void head_bwd_reduce_k(HeadBwdArgs a)
{
    __shared__ __attribute__((aligned(16))) float sH[TP * LDH];
    __shared__ __attribute__((aligned(16))) float sW[320];
    __shared__ float sB[8];
    __shared__ float sRow[5 * 5];        // per step: Gx, Gy, cst, M, 1/l
    __shared__ float sD[5 * TP];
    __shared__ float sAcc[4 * 5 * 64];
    __shared__ float sS0[4 * 5];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x, b = blockIdx.y;
    const int HW = a.f.OH * a.f.OW;
    fold_branch(a.f, b, sW, sB, tid, 256);
    if (tid < 5) {
        float gx, gy;
        row_grad(a, n, b, tid, gx, gy);
        const size_t o2 = (((size_t)n * 4 + b) * 5 + tid) * 2;
        sRow[tid * 5 + 0] = gx; sRow[tid * 5 + 1] = gy;
        sRow[tid * 5 + 2] = gx * a.f.pred_all[o2] + gy * a.f.pred_all[o2 + 1];
        sRow[tid * 5 + 3] = a.f.rowstat[o2];
        sRow[tid * 5 + 4] = 1.f / a.f.rowstat[o2 + 1];
    }
    __syncthreads();

    const int c = tid & 63, q = tid >> 6;
    float acc[5], s0[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) { acc[s] = 0.f; s0[s] = 0.f; }

    const int ntile = (HW + TP - 1) / TP;
    for (int tile = 0; tile < ntile; ++tile) {
        load_tile<T>(a.f.h, sH, n, HW, tile, tid);
        __syncthreads();
        const int p = tile * TP + tid;
        float dl[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) dl[s] = 0.f;
        if (p < HW) {
            float lg[5];
#pragma unroll
            for (int s = 0; s < 5; ++s) lg[s] = sB[s];
#pragma unroll
            for (int c4 = 0; c4 < 16; ++c4) {
                const float4 f = *reinterpret_cast<const float4*>(&sH[tid * LDH + c4 * 4]);
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    const float4 w = *reinterpret_cast<const float4*>(&sW[s * 64 + c4 * 4]);
                    lg[s] += f.x * w.x + f.y * w.y + f.z * w.z + f.w * w.w;
                }
            }
            const float px = a.f.pos_x[b][p], py = a.f.pos_y[b][p];
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const float pr = __expf(lg[s] - sRow[s * 5 + 3]) * sRow[s * 5 + 4];
                dl[s] = pr * ((sRow[s * 5 + 0] * px + sRow[s * 5 + 1] * py) - sRow[s * 5 + 2]);
                s0[s] += dl[s];
            }
        }
#pragma unroll
        for (int s = 0; s < 5; ++s) sD[s * TP + tid] = dl[s];
        __syncthreads();
        for (int pp = q * 64; pp < q * 64 + 64; ++pp) {
            const float hv = sH[pp * LDH + c];
#pragma unroll
            for (int s = 0; s < 5; ++s) acc[s] += sD[s * TP + pp] * hv;
        }
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        sAcc[(q * 5 + s) * 64 + c] = acc[s];
        float t = s0[s];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
        if (lane == 0) sS0[wave * 5 + s] = t;
    }
    __syncthreads();
    float* out = a.s_partial + (size_t)n * (20 * 65) + (size_t)(b * 5) * 65;
    for (int idx = tid; idx < 320; idx += 256) {
        const int s = idx >> 6, cc = idx & 63;
        out[s * 65 + cc] = sAcc[(0 * 5 + s) * 64 + cc] + sAcc[(1 * 5 + s) * 64 + cc] + sAcc[(2 * 5 + s) * 64 + cc] +
                           sAcc[(3 * 5 + s) * 64 + cc];
    }
    if (tid < 5) out[tid * 65 + 64] = sS0[tid] + sS0[5 + tid] + sS0[10 + tid] + sS0[15 + tid];
}

__global__ __launch_bounds__(256) void head_bwd_finalize_k(HeadBwdFinalizeArgs a)
{
    __shared__ float sS[20 * 65];
    __shared__ float sDG[4 * 64], sDB[4 * 64];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 20 * 65; idx += 256) {
        double t = 0.0;
        for (int r = 0; r < a.rows; ++r) t += (double)a.s_partial[(size_t)r * (20 * 65) + idx];
        sS[idx] = (float)t;
    }
    __syncthreads();
    // H1 -> S1 = invstd * (H1 - mean * S0)
    for (int idx = tid; idx < 20 * 64; idx += 256) {
        const int bs = idx >> 6, c = idx & 63;
        const float s0 = sS[bs * 65 + 64];
        sS[bs * 65 + c] = a.invstd[c] * (sS[bs * 65 + c] - a.mean[c] * s0);
    }
    __syncthreads();
    for (int idx = tid; idx < 20 * 64; idx += 256) {
        const int bs = idx >> 6, c = idx & 63, b = bs / 5;
        if (!a.coef_only) a.dw[b][(bs - b * 5) * 64 + c] = a.gamma[b][c] * sS[bs * 65 + c] + a.beta[b][c] * sS[bs * 65 + 64];
    }
    if (tid < 20 && !a.coef_only) a.dbias[tid / 5][tid % 5] = sS[tid * 65 + 64];
    {
        const int b = tid >> 6, c = tid & 63;   // 256 threads = 4 x 64
        float dg = 0.f, db = 0.f;
        for (int s = 0; s < 5; ++s) {
            const float w = a.w[b][s * 64 + c];
            dg += w * sS[(b * 5 + s) * 65 + c];
            db += w * sS[(b * 5 + s) * 65 + 64];
        }
        if (!a.coef_only) { a.dgamma[b][c] = dg; a.dbeta[b][c] = db; }
        sDG[tid] = dg; sDB[tid] = db;
    }
    __syncthreads();
    if (tid < 64) {
        double k1 = 0.0, k2 = 0.0;
        for (int b = 0; b < 4; ++b) {
            k1 += (double)a.gamma[b][tid] * (double)sDB[b * 64 + tid];
            k2 += (double)a.gamma[b][tid] * (double)sDG[b * 64 + tid];
        }
        const double n = a.nsum ? (double)a.count / (double)a.n_local * (double)*a.nsum : (double)a.count;
        k1 /= n; k2 /= n;
        const double inv = (double)a.invstd[tid];
        a.chan_coef[tid] = (float)(inv * k1);
        a.chan_coef[64 + tid] = (float)(inv * k2);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void head_bwd_apply_k(HeadBwdArgs a)
{
    __shared__ __attribute__((aligned(16))) float sH[TP * LDH];
    __shared__ __attribute__((aligned(16))) float sW[20 * 64];
    __shared__ float sB[4 * 8];
    __shared__ float sRow[20 * 5];
    __shared__ float sD[20 * TP];
    const int tid = threadIdx.x;
    const int n = blockIdx.x, tile = blockIdx.y;
    const int HW = a.f.OH * a.f.OW;
    for (int b = 0; b < 4; ++b) fold_branch(a.f, b, sW + b * 320, sB + b * 8, tid, 256);
    if (tid < 20) {
        const int b = tid / 5, s = tid - b * 5;
        float gx, gy;
        row_grad(a, n, b, s, gx, gy);
        const size_t o2 = (((size_t)n * 4 + b) * 5 + s) * 2;
        sRow[tid * 5 + 0] = gx; sRow[tid * 5 + 1] = gy;
        sRow[tid * 5 + 2] = gx * a.f.pred_all[o2] + gy * a.f.pred_all[o2 + 1];
        sRow[tid * 5 + 3] = a.f.rowstat[o2];
        sRow[tid * 5 + 4] = 1.f / a.f.rowstat[o2 + 1];
    }
    load_tile<T>(a.f.h, sH, n, HW, tile, tid);
    __syncthreads();
    const int p = tile * TP + tid;
    {
        float lg[20];
#pragma unroll
        for (int bs = 0; bs < 20; ++bs) lg[bs] = sB[(bs / 5) * 8 + (bs % 5)];
        if (p < HW) {
#pragma unroll
            for (int c4 = 0; c4 < 16; ++c4) {
                const float4 f = *reinterpret_cast<const float4*>(&sH[tid * LDH + c4 * 4]);
#pragma unroll
                for (int bs = 0; bs < 20; ++bs) {
                    const float4 w = *reinterpret_cast<const float4*>(&sW[bs * 64 + c4 * 4]);
                    lg[bs] += f.x * w.x + f.y * w.y + f.z * w.z + f.w * w.w;
                }
            }
        }
#pragma unroll
        for (int bs = 0; bs < 20; ++bs) {
            float d = 0.f;
            if (p < HW) {
                const int b = bs / 5;
                const float px = a.f.pos_x[b][p], py = a.f.pos_y[b][p];
                const float pr = __expf(lg[bs] - sRow[bs * 5 + 3]) * sRow[bs * 5 + 4];
                d = pr * ((sRow[bs * 5 + 0] * px + sRow[bs * 5 + 1] * py) - sRow[bs * 5 + 2]);
            }
            sD[bs * TP + tid] = d;
        }
    }
    __syncthreads();
    const int c = tid & 63, q = tid >> 6;
    float wf[20];
#pragma unroll
    for (int bs = 0; bs < 20; ++bs) wf[bs] = sW[bs * 64 + c];
    const float c1 = a.chan_coef[c], c2 = a.chan_coef[64 + c];
    const float mu = a.f.mean[0][c], iv = a.f.invstd[0][c];
    for (int pp = q * 64; pp < q * 64 + 64; ++pp) {
        const int pg = tile * TP + pp;
        if (pg >= HW) break;
        float o = -c1 - c2 * ((sH[pp * LDH + c] - mu) * iv);
#pragma unroll
        for (int bs = 0; bs < 20; ++bs) o += sD[bs * TP + pp] * wf[bs];
        Act<T>::st1(static_cast<T*>(a.dh) + ((size_t)n * HW + (size_t)pg) * 64 + c, o);
    }
}


// ---- bf16 activations: the backward on v_mfma_f32_32x32x16_bf16 ---------------------------------------------------------------
// The scalar kernels above spend 20 x 64 LDS-fed FMAs per pixel twice (0.43 + 0.29 ms at batch 256 for 0.25 GB of traffic).
// Here a wave walks 32-pixel groups; the folded 20 x 64 projection (bf16 values, as in the forward) is stationary in registers.
//   reduce:  logits[p][bs] = h Wf^T (A = h rows straight from HBM, as head_fwd_mfma_k) -> d[p][bs] in the accumulator layout, which
//            IS the B-operand layout of the next product  H1^T[c][bs] += sum_p h^T[c][p] d[p][bs]  once the contraction index is
//            permuted consistently: MFMA j's k-slot i of half kh stands for pixel (e & 3) + 8 (e >> 2) + 4 kh, e = 8 j + i.  The
//            h^T fragments in that order come from the group's [pixel][channel] LDS image through ds_read_b64_tr_b16.
//   apply:   logits^T[bs][p] = Wf h^T (operands swapped: same loads) -> d^T in the accumulator layout = B operand of
//            dh^T[c][p] = sum_bs Wf^T[c][bs] d^T[bs][p] - c1'[c] - c2'[c] h^T[c][p];  the two BatchNorm-backward terms ride on
//            the same MFMAs: -c1' through two constant-one k-slots (bf16 high + low part), -c2' h as a diagonal A operand
//            against the h fragments already in registers.  A lane ends up with 4 consecutive channels of its pixel: 8-byte stores.
// H1 is built from the bf16-rounded d; the mean S0 term head_bwd_finalize_k subtracts is corrected for that rounding (see the end of
// the reduce kernel), while the stored S0 stays the exact f32 sum.
constexpr int kHTS = 80;           // LDS row stride (elements) of a wave's 32 x 64 h image: as conv_wgrad_tr.hip's transpose reads

struct HeadRowConst { float k0, gx, gy, cst; };

// per (branch, step) row constants of image n: d = exp(logit_nobias + k0) * (gx px + gy py - cst)
__device__ __forceinline__ void head_row_consts(const HeadBwdArgs& a, int n, const float* sBf, float* sRow /*[20][4]*/, int tid)
{
    if (tid < 20) {
        const int b = tid / 5, st = tid - b * 5;
        float gx, gy;
        row_grad(a, n, b, st, gx, gy);
        const size_t o2 = (((size_t)n * 4 + b) * 5 + st) * 2;
        sRow[tid * 4 + 0] = sBf[b * 8 + st] - a.f.rowstat[o2] - logf(a.f.rowstat[o2 + 1]);
        sRow[tid * 4 + 1] = gx;
        sRow[tid * 4 + 2] = gy;
        sRow[tid * 4 + 3] = gx * a.f.pred_all[o2] + gy * a.f.pred_all[o2 + 1];
    }
}

__global__ __launch_bounds__(256, 2) void head_bwd_reduce_mfma_k(HeadBwdArgs a, int S)
{
    __shared__ float sWf[4 * 320];
    __shared__ float sBf[4 * 8];
    __shared__ float sRow[20 * 4];
    __shared__ __attribute__((aligned(16))) __bf16 sT[4][32 * kHTS];
    __shared__ float sRed[4][20 * 65];
    __shared__ float sDiff[4][20];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5, G1 = (lane >> 4) & 1, t16 = lane & 15;
    const int n = blockIdx.x, slice = blockIdx.y;
    const int HW = a.f.OH * a.f.OW;
    for (int b = 0; b < 4; ++b) fold_branch(a.f, b, sWf + b * 320, sBf + b * 8, tid, 256);
    __syncthreads();
    head_row_consts(a, n, sBf, sRow, tid);
    __syncthreads();
    const bool colok = l31 < 20;
    const int cb = colok ? l31 / 5 : 0, cs = colok ? l31 - 5 * cb : 0;
    bf16x8 wb[4], wl[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = colok ? sWf[cb * 320 + cs * 64 + g * 16 + kh * 8 + j] : 0.f;
            wb[g][j] = (__bf16)v;
            wl[g][j] = (__bf16)(v - (float)wb[g][j]);
        }
    const bool split = a.f.wsplit != 0;
    const float k0 = colok ? sRow[l31 * 4 + 0] : 0.f, gx = colok ? sRow[l31 * 4 + 1] : 0.f;
    const float gy = colok ? sRow[l31 * 4 + 2] : 0.f, cst = colok ? sRow[l31 * 4 + 3] : 0.f;
    const float* posx = a.f.pos_x[cb];
    const float* posy = a.f.pos_y[cb];
    const __bf16* h = static_cast<const __bf16*>(a.f.h) + (size_t)n * HW * 64;
    __bf16* tile = sT[wave];

    f32x16 accH[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) accH[c][e] = 0.f;
    float s0 = 0.f, s0r = 0.f;        // sum of d, and of the bf16-rounded d the MFMA consumes
    const int ngroup = (HW + 31) / 32;
    for (int grp = slice * 4 + wave; grp < ngroup; grp += 4 * S) {
        const int pbase = grp * 32;
        const int pl = pbase + l31 < HW ? pbase + l31 : HW - 1;
        bf16x8 af[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) af[g] = *reinterpret_cast<const bf16x8*>(h + (size_t)pl * 64 + (size_t)(g * 16 + kh * 8));
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<bf16x8*>(&tile[l31 * kHTS + g * 16 + kh * 8]) = af[g];
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g], wb[g], acc, 0, 0, 0);
        if (split) {
#pragma unroll
            for (int g = 0; g < 4; ++g) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g], wl[g], acc, 0, 0, 0);
        }
        bf16x8 db[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int p = pbase + (e & 3) + 8 * (e >> 2) + 4 * kh;
            float d = 0.f;
            if (p < HW && colok) d = __expf(acc[e] + k0) * ((gx * posx[p] + gy * posy[p]) - cst);
            const __bf16 dr = (__bf16)d;
            db[e >> 3][e & 7] = dr;
            s0 += d;
            s0r += (float)dr;
        }
        // h^T fragments: rows = channels 32 cblk + l31, k-slot i of MFMA j = pixel 16 j + 8 (i >> 2) + 4 kh + (i & 3)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 16 * j + 4 * kh + (t16 >> 2);
                const int col = 32 * c + 16 * G1 + (t16 & 3) * 4;
                const bf16x4 a0 = lds_read_tr16(&tile[row * kHTS + col]);
                const bf16x4 a1 = lds_read_tr16(&tile[(row + 8) * kHTS + col]);
                const bf16x8 at = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                accH[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, db[j], accH[c], 0, 0, 0);
            }
    }
    // accH[c][e]: channel 32 c + (e & 3) + 8 (e >> 2) + 4 kh, column bs = l31; S0: the two half-waves hold different pixels
    s0 += __shfl_xor(s0, 32);
    s0r += __shfl_xor(s0r, 32);
    if (colok) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) sRed[wave][l31 * 65 + 32 * c + (e & 3) + 8 * (e >> 2) + 4 * kh] = accH[c][e];
        if (kh == 0) { sRed[wave][l31 * 65 + 64] = s0; sDiff[wave][l31] = s0r - s0; }
    }
    __syncthreads();
    float* out = a.s_partial + ((size_t)n * S + slice) * (20 * 65);
    // column 64 carries the exact S0 (dbias, dbeta and k1 are sums that vanish analytically); head_bwd_finalize_k forms
    // S1 = invstd (H1 - mean S0), which must cancel against the ROUNDED d inside H1: fold mean (S0r - S0) into H1 here
    for (int idx = tid; idx < 20 * 65; idx += 256) {
        const int bs = idx / 65, c = idx - bs * 65;
        float v = (sRed[0][idx] + sRed[1][idx]) + (sRed[2][idx] + sRed[3][idx]);
        if (c < 64) v -= a.f.mean[0][c] * ((sDiff[0][bs] + sDiff[1][bs]) + (sDiff[2][bs] + sDiff[3][bs]));
        out[idx] = v;
    }
}

__global__ __launch_bounds__(256, 2) void head_bwd_apply_mfma_k(HeadBwdArgs a, int S)
{
    __shared__ float sWf[4 * 320];
    __shared__ float sBf[4 * 8];
    __shared__ float sRow[20 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int n = blockIdx.x, slice = blockIdx.y;
    const int HW = a.f.OH * a.f.OW;
    for (int b = 0; b < 4; ++b) fold_branch(a.f, b, sWf + b * 320, sBf + b * 8, tid, 256);
    __syncthreads();
    head_row_consts(a, n, sBf, sRow, tid);
    __syncthreads();
    // A operand of logits^T: row bs = l31, 8 consecutive channels
    const bool rowok = l31 < 20;
    const int rb = rowok ? l31 / 5 : 0, rs = rowok ? l31 - 5 * rb : 0;
    bf16x8 wa[4], wal[4];           // high / low parts: the recomputed logits must be the forward's (the saved soft-max statistics)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = rowok ? sWf[rb * 320 + rs * 64 + g * 16 + kh * 8 + j] : 0.f;
            wa[g][j] = (__bf16)v;
            wal[g][j] = (__bf16)(v - (float)wa[g][j]);
        }
    const bool split = a.f.wsplit != 0;
    // per accumulator row e of logits^T (bs = (e & 3) + 8 (e >> 2) + 4 kh; e < 12 covers every bs < 20): row constants
    float ck0[12], cgx[12], cgy[12], ccst[12];
    int ebr[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        const int bs = (e & 3) + 8 * (e >> 2) + 4 * kh;
        const bool ok = bs < 20;
        ebr[e] = ok ? bs / 5 : 0;
        ck0[e] = ok ? sRow[bs * 4 + 0] : -INFINITY;     // exp(-inf) = 0: the padding rows contribute nothing
        cgx[e] = ok ? sRow[bs * 4 + 1] : 0.f;
        cgy[e] = ok ? sRow[bs * 4 + 2] : 0.f;
        ccst[e] = ok ? sRow[bs * 4 + 3] : 0.f;
    }
    // A operands of dh^T, rows = channels c = 32 cblk + l31:
    //   wt[cblk][j]: k-slot i <-> bs(e = 8 j + i): Wf[bs][c]; two spare slots of half kh = 0 (e = 12, 13: bs 24, 25) carry -c1'
    //   dg[cblk][g']: the diagonal -c2'[c] against the h fragment g = 2 cblk + g' (k-slot i <-> channel 16 g + 8 kh + i)
    bf16x8 wt[2][2], dg[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int ch = 32 * c + l31;
        const float iv = a.f.invstd[0][ch], mu = a.f.mean[0][ch];
        const float c2p = a.chan_coef[64 + ch] * iv;
        const float c1p = a.chan_coef[ch] - c2p * mu;
        const __bf16 c1h = (__bf16)(-c1p);
        const __bf16 c1l = (__bf16)(-c1p - (float)c1h);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = 8 * j + i;
                const int bs = (e & 3) + 8 * (e >> 2) + 4 * kh;
                float v = 0.f;
                if (bs < 20) v = sWf[(bs / 5) * 320 + (bs % 5) * 64 + ch];
                __bf16 q = (__bf16)v;
                if (bs == 24) q = c1h;
                if (bs == 25) q = c1l;
                wt[c][j][i] = q;
            }
#pragma unroll
        for (int gp = 0; gp < 2; ++gp)
#pragma unroll
            for (int i = 0; i < 8; ++i) dg[c][gp][i] = (__bf16)((16 * (2 * c + gp) + 8 * kh + i) == ch ? -c2p : 0.f);
    }
    const __bf16 one = (__bf16)1.f, zero = (__bf16)0.f;
    const __bf16* h = static_cast<const __bf16*>(a.f.h) + (size_t)n * HW * 64;
    __bf16* dh = static_cast<__bf16*>(a.dh) + (size_t)n * HW * 64;

    const int ngroup = (HW + 31) / 32;
    for (int grp = slice * 4 + wave; grp < ngroup; grp += 4 * S) {
        const int pbase = grp * 32;
        const bool live = pbase + l31 < HW;
        const int pl = live ? pbase + l31 : HW - 1;
        bf16x8 hb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) hb[g] = *reinterpret_cast<const bf16x8*>(h + (size_t)pl * 64 + (size_t)(g * 16 + kh * 8));
        float px[4], py[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) { px[b] = a.f.pos_x[b][pl]; py[b] = a.f.pos_y[b][pl]; }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[g], hb[g], acc, 0, 0, 0);
        if (split) {
#pragma unroll
            for (int g = 0; g < 4; ++g) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wal[g], hb[g], acc, 0, 0, 0);
        }
        bf16x8 db[2];
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const float d = __expf(acc[e] + ck0[e]) * ((cgx[e] * px[ebr[e]] + cgy[e] * py[ebr[e]]) - ccst[e]);
            db[e >> 3][e & 7] = (__bf16)d;
        }
        db[1][4] = kh == 0 ? one : zero;      // e = 12, 13 of half 0: bs 24, 25 -- the constant-one slots of -c1' (high, low)
        db[1][5] = kh == 0 ? one : zero;
        db[1][6] = zero; db[1][7] = zero;
        f32x16 o[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int e = 0; e < 16; ++e) o[c][e] = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) o[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[c][j], db[j], o[c], 0, 0, 0);
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) o[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dg[c][gp], hb[2 * c + gp], o[c], 0, 0, 0);
        }
        // o[c][e]: channel 32 c + (e & 3) + 8 (e >> 2) + 4 kh of pixel l31: regs 4 q .. 4 q + 3 are 4 consecutive channels
        if (live) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {o[c][4 * q], o[c][4 * q + 1], o[c][4 * q + 2], o[c][4 * q + 3]};
                    *reinterpret_cast<bf16x4*>(dh + (size_t)(pbase + l31) * 64 + (size_t)(32 * c + 8 * q + 4 * kh)) = __builtin_convertvector(v, bf16x4);
                }
        }
    }
}

// pixel-group slices per image of the MFMA backward kernels: about 1024 workgroups, at most one slice per 4 groups
static int head_bwd_slices(int N, int HW)
{
    int S = (1024 + N - 1) / N;
    if (S > 16) S = 16;
    const int cap = ((HW + 31) / 32 + 3) / 4;
    if (S > cap) S = cap;
    return S < 1 ? 1 : S;
}
static bool head_bwd_mfma(const HeadArgs& f) { return f.act_bf16 && !lbc_opt_on(kOptHeadNoMfma); }

}  // namespace

int lbc_head_fwd(const HeadArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.N > 0 && a.OH > 0 && a.OW > 0, "head_fwd: bad shape");
    LbcProfScope prof("head_fwd", 2.0 * a.N * a.OH * a.OW * 64.0 * 20, 4.0 * a.N * (double)a.OH * a.OW * 64, s);
    const bool no_mfma = lbc_opt_on(kOptHeadNoMfma);   // A/B switch
    const int wsplit = a.act_bf16 ? 1 : 0;
    if (a.act_bf16 && !no_mfma) {
        HeadArgs b = a;
        b.nslice = 1; b.wsplit = wsplit;
        if (a.scratch) {
            // about 1024 workgroups whatever the batch: with one workgroup (four waves) per image a 256-image launch left every CU
            // with four waves streaming 0.5 MB each -- latency bound at 1.0 TB/s (122 us for 126 MB, profiles/r04_final_*); the
            // image's 32-pixel groups are split over 2 .. 16 slices, merged in slice order by head_merge_k
            b.nslice = (1024 + a.N - 1) / a.N;
            if (b.nslice > 16) b.nslice = 16;
            const int cap = ((a.OH * a.OW + 31) / 32 + 3) / 4;     // at least one group per wave and slice
            if (b.nslice > cap) b.nslice = cap;
            if (b.nslice < 1) b.nslice = 1;
        }
        hipLaunchKernelGGL(head_fwd_mfma_k, dim3((unsigned)a.N, (unsigned)b.nslice), dim3(256), 0, s, b);
        if (b.nslice > 1) hipLaunchKernelGGL(head_merge_k, dim3((unsigned)lbc_cdiv(a.N * 20, 256)), dim3(256), 0, s, b);
    }
    else if (a.act_bf16) { HeadArgs b = a; b.wsplit = wsplit; hipLaunchKernelGGL((head_fwd_k<__bf16>), dim3((unsigned)a.N, 4), dim3(256), 0, s, b); }
    else hipLaunchKernelGGL((head_fwd_k<float>), dim3((unsigned)a.N, 4), dim3(256), 0, s, a);
    int rc = lbc_check_launch("head_fwd");
    if (rc) return rc;
    if (a.pred_sel) {
        hipLaunchKernelGGL(select_branch_k, dim3((unsigned)lbc_cdiv(a.N * 10, 256)), dim3(256), 0, s, a.pred_all, a.cmd,
                           a.pred_sel, a.N);
        rc = lbc_check_launch("select_branch");
    }
    return rc;
}

int lbc_head_bwd_rows(const HeadArgs& f) { return head_bwd_mfma(f) ? f.N * head_bwd_slices(f.N, f.OH * f.OW) : f.N; }
int lbc_head_bwd_max_rows(int max_batch) { return max_batch + 1024; }

int lbc_head_bwd_reduce(const HeadBwdArgs& a0, hipStream_t s)
{
    HeadBwdArgs a = a0;
    a.f.wsplit = a.f.act_bf16 ? 1 : 0;
    LBC_REQUIRE(a.f.mean[0] == a.f.mean[1] && a.f.mean[0] == a.f.mean[2] && a.f.mean[0] == a.f.mean[3],
                "head backward requires training-mode (shared batch) statistics");
    LbcProfScope prof("head_bwd_reduce", 4.0 * a.f.N * a.f.OH * a.f.OW * 64.0 * 20, 4.0 * a.f.N * (double)a.f.OH * a.f.OW * 64, s);
    if (head_bwd_mfma(a.f)) {
        const int S = head_bwd_slices(a.f.N, a.f.OH * a.f.OW);
        hipLaunchKernelGGL(head_bwd_reduce_mfma_k, dim3((unsigned)a.f.N, (unsigned)S), dim3(256), 0, s, a, S);
        return lbc_check_launch("head_bwd_reduce");
    }
#define LBC_K(T, d) hipLaunchKernelGGL((head_bwd_reduce_k<T>), dim3((unsigned)a.f.N, 4), dim3(256), 0, s, a)
    LBC_DISPATCH_ACT(a.f.act_bf16, LBC_K, 0);
#undef LBC_K
    return lbc_check_launch("head_bwd_reduce");
}

int lbc_head_bwd_finalize(const HeadBwdFinalizeArgs& a, hipStream_t s)
{
    LbcProfScope prof("head_bwd_finalize", 0.0, 4.0 * (double)a.rows * 20 * 65, s);
    hipLaunchKernelGGL(head_bwd_finalize_k, dim3(1), dim3(256), 0, s, a);
    return lbc_check_launch("head_bwd_finalize");
}

int lbc_head_bwd_apply(const HeadBwdArgs& a0, hipStream_t s)
{
    HeadBwdArgs a = a0;
    a.f.wsplit = a.f.act_bf16 ? 1 : 0;
    const int HW = a.f.OH * a.f.OW;
    LbcProfScope prof("head_bwd_apply", 4.0 * a.f.N * (double)HW * 64.0 * 20, 8.0 * a.f.N * (double)HW * 64, s);
    if (head_bwd_mfma(a.f)) {
        const int S = head_bwd_slices(a.f.N, HW);
        hipLaunchKernelGGL(head_bwd_apply_mfma_k, dim3((unsigned)a.f.N, (unsigned)S), dim3(256), 0, s, a, S);
        return lbc_check_launch("head_bwd_apply");
    }
#define LBC_K(T, d) hipLaunchKernelGGL((head_bwd_apply_k<T>), dim3((unsigned)a.f.N, (unsigned)lbc_cdiv(HW, TP)), dim3(256), 0, s, a)
    LBC_DISPATCH_ACT(a.f.act_bf16, LBC_K, 0);
#undef LBC_K
    return lbc_check_launch("head_bwd_apply");
}
