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
        // bf16 activations: the forward multiplies on the bf16 MFMA, so the folded weight is a bf16 value everywhere
        // (forward, and the backward's recomputation of the logits) -- the saved soft-max statistics stay consistent
        sW[idx] = a.act_bf16 ? (float)(__bf16)wf : wf;
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
    bf16x8 wb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) wb[g][j] = (__bf16)(colok ? sWf[cb * 320 + cs * 64 + g * 16 + kh * 8 + j] : 0.f);
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
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int p = pbase + (e & 3) + 8 * (e >> 2) + 4 * kh;
            if (p < HW && colok) {
                SoftAcc o; o.m = acc[e] + bias; o.l = 1.f; o.sx = posx[p]; o.sy = posy[p];
                soft_merge(st, o);
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
        a.dw[b][(bs - b * 5) * 64 + c] = a.gamma[b][c] * sS[bs * 65 + c] + a.beta[b][c] * sS[bs * 65 + 64];
    }
    if (tid < 20) a.dbias[tid / 5][tid % 5] = sS[tid * 65 + 64];
    {
        const int b = tid >> 6, c = tid & 63;   // 256 threads = 4 x 64
        float dg = 0.f, db = 0.f;
        for (int s = 0; s < 5; ++s) {
            const float w = a.w[b][s * 64 + c];
            dg += w * sS[(b * 5 + s) * 65 + c];
            db += w * sS[(b * 5 + s) * 65 + 64];
        }
        a.dgamma[b][c] = dg; a.dbeta[b][c] = db;
        sDG[tid] = dg; sDB[tid] = db;
    }
    __syncthreads();
    if (tid < 64) {
        double k1 = 0.0, k2 = 0.0;
        for (int b = 0; b < 4; ++b) {
            k1 += (double)a.gamma[b][tid] * (double)sDB[b * 64 + tid];
            k2 += (double)a.gamma[b][tid] * (double)sDG[b * 64 + tid];
        }
        k1 /= (double)a.count; k2 /= (double)a.count;
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

}  // namespace

int lbc_head_fwd(const HeadArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.N > 0 && a.OH > 0 && a.OW > 0, "head_fwd: bad shape");
    LbcProfScope prof("head_fwd", 2.0 * a.N * a.OH * a.OW * 64.0 * 20, 4.0 * a.N * (double)a.OH * a.OW * 64, s);
    const bool no_mfma = lbc_opt_on(kOptHeadNoMfma);   // A/B switch
    if (a.act_bf16 && !no_mfma) {
        HeadArgs b = a;
        b.nslice = 1;
        if (a.scratch && a.N < 128) {                    // fill the chip at small batch: 2..16 slices per image
            b.nslice = (256 + a.N - 1) / a.N;
            if (b.nslice > 16) b.nslice = 16;
        }
        hipLaunchKernelGGL(head_fwd_mfma_k, dim3((unsigned)a.N, (unsigned)b.nslice), dim3(256), 0, s, b);
        if (b.nslice > 1) hipLaunchKernelGGL(head_merge_k, dim3((unsigned)lbc_cdiv(a.N * 20, 256)), dim3(256), 0, s, b);
    }
    else if (a.act_bf16) hipLaunchKernelGGL((head_fwd_k<__bf16>), dim3((unsigned)a.N, 4), dim3(256), 0, s, a);
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

int lbc_head_bwd_rows(int N) { return N; }

int lbc_head_bwd_reduce(const HeadBwdArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.f.mean[0] == a.f.mean[1] && a.f.mean[0] == a.f.mean[2] && a.f.mean[0] == a.f.mean[3],
                "head backward requires training-mode (shared batch) statistics");
    LbcProfScope prof("head_bwd_reduce", 4.0 * a.f.N * a.f.OH * a.f.OW * 64.0 * 20, 4.0 * a.f.N * (double)a.f.OH * a.f.OW * 64, s);
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

int lbc_head_bwd_apply(const HeadBwdArgs& a, hipStream_t s)
{
    const int HW = a.f.OH * a.f.OW;
    LbcProfScope prof("head_bwd_apply", 4.0 * a.f.N * (double)HW * 64.0 * 20, 8.0 * a.f.N * (double)HW * 64, s);
#define LBC_K(T, d) hipLaunchKernelGGL((head_bwd_apply_k<T>), dim3((unsigned)a.f.N, (unsigned)lbc_cdiv(HW, TP)), dim3(256), 0, s, a)
    LBC_DISPATCH_ACT(a.f.act_bf16, LBC_K, 0);
#undef LBC_K
    return lbc_check_launch("head_bwd_apply");
}
