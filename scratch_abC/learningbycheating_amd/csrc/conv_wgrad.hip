// Weight-gradient GEMM for gfx950: exact f32 (v_mfma_f32_32x32x2_f32) and bf16 operands (v_mfma_f32_32x32x16_bf16).
//   out[p][tap][q] = sum over pixels m of  P[m][p] * Q[gather(m, tap)][q]
// which is the dW of nn.Conv2d (P = dy, Q = x; reference call sites
// bird_view/models/resnet.py:15-22,102) and, with the roles of input and output
// gradient swapped, of nn.ConvTranspose2d (P = x, Q = dy; image.py:39,42,45).
// The reduction runs over N*OH*OW pixels (up to ~10^6) into a small
// [CP][T][CQ] result, so the pixel range is split across workgroups; each split
// writes its own partial slab and lbc_splitk_reduce adds the slabs in a fixed
// order (deterministic, no atomics).
//
// Both operands are pixel-major in HBM (NHWC), i.e. "depth-outer" for this GEMM.
// f32 kernel: LDS tiles are [pixel][channel] and MFMA fragments are ds_read_b32 reads of
// 32 consecutive channels (conflict free); one MFMA consumes two pixels.
// bf16 kernel: the MFMA wants 8 consecutive pixels of one channel per lane, so micro-tiles of 4 pixels x 4/8
// channels are transposed in registers into [channel][64 pixels] tiles (see the kernel for the lane order that
// avoids LDS write conflicts).  The 3x3 / stride-1 launches on bf16 tensors go to conv_wgrad_tr.hip instead
// (all nine taps per workgroup, transpose reads); this file keeps the stride-2, 1x1, transposed-convolution and
// f32-tensor cases.
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include <type_traits>
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int BR = 32;   // pixels per chunk

template <int BP, int BQ>
__global__ __launch_bounds__(256) void conv_wgrad_f32(WgradArgs a, int rows_per_split)
{
    constexpr int WM = 2, WN = 2;
    constexpr int MT = BP / WM / 32;
    constexpr int NT = BQ / WN / 32;
    constexpr int LP = BP + 4, LQ = BQ + 4;
    constexpr int RP = BP / 32, RQ = BQ / 32;   // float4 loads per thread per chunk

    __shared__ __attribute__((aligned(16))) float sP[2][BR * LP];
    __shared__ __attribute__((aligned(16))) float sQ[2][BR * LQ];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;

    // Work decode: workgroup b runs on XCD b % 8.  The T tap-workgroups of one (split, tile) group read the same P rows
    // and overlapping Q rows, so they are given ids b, b+8, ..., b+8(T-1): same XCD, dispatched back to back, and the
    // operands are fetched from HBM once and re-read from that XCD's L2.  Padding ids (group >= ngroups) exit at once.
    const int T = a.KH * a.KW;
    const int qtiles = a.CQ / BQ;
    const int ntiles = (a.CP / BP) * qtiles;
    const int ngroups = a.nsplit * ntiles;
    const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
    const int tap = qq % T;
    const int group = (qq / T) * 8 + xcd;
    if (group >= ngroups) return;
    const int split = group / ntiles;
    const int tile = group - split * ntiles;
    const int tp = tile / qtiles;
    const int tq = tile - tp * qtiles;
    const int p0 = tp * BP, q0 = tq * BQ;
    const int r = tap / a.KW, s = tap - r * a.KW;

    const int M = a.N * a.OH * a.OW;
    const int mbeg = split * rows_per_split;
    const int mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
    const int nchunk = (mend > mbeg) ? (mend - mbeg + BR - 1) / BR : 0;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 rp[RP], rq[RQ];   // native vector values (see conv_igemm.hip)

    // pixel coordinates of this thread's Q rows, advanced by BR per chunk (no integer divisions in the loop)
    int qn[RQ], qy[RQ], qx[RQ];
#pragma unroll
    for (int j = 0; j < RQ; ++j) {
        const int row = (tid + 256 * j) / (BQ / 4);
        const int m = mbeg + row;
        const int ohw = a.OH * a.OW;
        qn[j] = m / ohw;
        const int rem = m - qn[j] * ohw;
        qy[j] = rem / a.OW;
        qx[j] = rem - qy[j] * a.OW;
    }

    bool pok[RP], qok[RQ];
    const float relu_floor = (a.q_scale && a.q_relu) ? 0.f : -INFINITY;

    // software pipeline written out once (see conv_igemm.hip): loads of chunk ch+1 -> MFMAs of chunk ch -> transform +
    // LDS store of chunk ch+1 -> barrier.  Loads are branch-free; masking / BatchNorm-on-load happen at store time.
    for (int ch = -1; ch < nchunk; ++ch) {
        const bool more = ch + 1 < nchunk;
        if (more) {
            const int mc = mbeg + (ch + 1) * BR;
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const int idx = tid + 256 * j;
                const int row = idx / (BP / 4);
                const int sg = idx - row * (BP / 4);
                const int m = mc + row;
                pok[j] = m < mend;
                const int ms = pok[j] ? m : 0;
                rp[j] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(a.p) + (size_t)ms * (size_t)a.CP + (size_t)(p0 + sg * 4));
            }
#pragma unroll
            for (int j = 0; j < RQ; ++j) {
                const int idx = tid + 256 * j;
                const int row = idx / (BQ / 4);
                const int sg = idx - row * (BQ / 4);
                const int m = mc + row;
                const int iy = qy[j] * a.S + r - a.P;
                const int ix = qx[j] * a.S + s - a.P;
                qok[j] = (m < mend) && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const int pix = qok[j] ? ((qn[j] * a.H + iy) * a.W + ix) : 0;
                rq[j] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(a.q) + (size_t)pix * (size_t)a.CQ + (size_t)(q0 + sg * 4));
                qx[j] += BR;
                while (qx[j] >= a.OW) { qx[j] -= a.OW; ++qy[j]; }
                while (qy[j] >= a.OH) { qy[j] -= a.OH; ++qn[j]; }
            }
        }
        if (ch >= 0) {
            const int buf = ch & 1;
#pragma unroll
            for (int st = 0; st < BR / 2; ++st) {
                const int k = 2 * st + kh;
                float af[MT], bf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = sP[buf][k * LP + (wm * MT + i) * 32 + l31];
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[j] = sQ[buf][k * LQ + (wn * NT + j) * 32 + l31];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (more) {
            const int buf = (ch + 1) & 1;
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const int idx = tid + 256 * j;
                const int row = idx / (BP / 4);
                const int sg = idx - row * (BP / 4);
                f32x4 v = rp[j];
                if (a.p_scale) {
                    const f32x4 ps = *reinterpret_cast<const f32x4*>(a.p_scale + p0 + sg * 4);
                    const f32x4 pt = *reinterpret_cast<const f32x4*>(a.p_shift + p0 + sg * 4);
                    v = v * ps + pt;
                }
                if (!pok[j]) v = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(&sP[buf][row * LP + sg * 4]) = v;
            }
#pragma unroll
            for (int j = 0; j < RQ; ++j) {
                const int idx = tid + 256 * j;
                const int row = idx / (BQ / 4);
                const int sg = idx - row * (BQ / 4);
                f32x4 v = rq[j];
                if (a.q_scale) {
                    const f32x4 ps = *reinterpret_cast<const f32x4*>(a.q_scale + q0 + sg * 4);
                    const f32x4 pt = *reinterpret_cast<const f32x4*>(a.q_shift + q0 + sg * 4);
                    v = v * ps + pt;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], relu_floor);
                }
                if (!qok[j]) v = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(&sQ[buf][row * LQ + sg * 4]) = v;
            }
        }
        __syncthreads();
    }

    float* out = a.partial + (size_t)split * (size_t)a.CP * (size_t)T * (size_t)a.CQ;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int prow = p0 + (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int qcol = q0 + (wn * NT + j) * 32 + l31;
                out[((size_t)prow * (size_t)T + (size_t)tap) * (size_t)a.CQ + (size_t)qcol] = acc[i][j][e];
            }
        }
}

// bf16-MFMA variant (v_mfma_f32_32x32x16_bf16, f32 accumulation; operands stay f32 in HBM and are rounded to bf16 when
// the tile is written to LDS).  The contraction runs over pixels, which are the slow axis of both NHWC operands, while a
// bf16 MFMA fragment wants 8 consecutive depth values per lane.  So every thread stages 4 consecutive pixels x 4 channels
// micro-tiles, transposes them in registers (free) and writes four 8-byte rows [channel][4 pixels]: LDS tiles are
// [channel][64 pixels] and fragments are plain conflict-free ds_read_b128, with no extra LDS traffic.
// AT = element type of P and Q in HBM: float (micro-tiles of 4 pixels x 4 channels) or __bf16 (4 pixels x 8 channels).
//
// Staging: a thread owns micro-tiles of 4 consecutive pixels x CH channels (one 16-byte load per pixel), transposes them
// in registers and writes CH 8-byte columns into the [channel][64 pixels] LDS tiles (rows padded to 144 B, which keeps
// the ds_read_b128 fragment reads conflict-free).  ds_write_b64 is serviced in groups of 16 consecutive lanes over 32
// banks, and a row is 36 dwords, so lanes of a group that differ only in their channel group (8 rows = 288 dwords = 0
// mod 32) collide: the micro-tile index therefore puts KB channel-group bits and 4-KB pixel-group bits into the low 4
// lane bits (KB = 1: f32 input conflict-free, bf16 input 2-way; loads still cover whole 128-byte lines per wave).
template <int BP, int BQ, typename AT, int KB>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_k(WgradArgs a, int rows_per_split)
{
    constexpr bool ABF = Act<AT>::kBf16;
    // channels per load: 16-byte loads, except bf16 input on the 64-wide tiles (8-byte loads keep all 256 threads staging)
    constexpr int CH = (ABF && BP >= 128) ? 8 : 4;
    using areg_t = typename std::conditional<ABF, typename std::conditional<CH == 8, bf16x8, bf16x4>::type, f32x4>::type;
    constexpr int BRH = 64;                 // pixels per chunk
    constexpr int LD = BRH + 8;             // padded LDS row (bf16 elements) = 144 bytes
    constexpr int WM = 2, WN = 2;
    constexpr int MT = BP / WM / 32, NT = BQ / WN / 32;
    constexpr int CGP = BP / CH, CGQ = BQ / CH;          // channel groups
    constexpr int TP_ = 16 * CGP, TQ_ = 16 * CGQ;        // micro-tiles per chunk
    constexpr int NP = (TP_ + 255) / 256, NQ = (TQ_ + 255) / 256;   // micro-tiles per thread
    __shared__ __attribute__((aligned(16))) __bf16 sP[2][BP * LD];
    __shared__ __attribute__((aligned(16))) __bf16 sQ[2][BQ * LD];

    const AT* pin = static_cast<const AT*>(a.p);
    const AT* qin = static_cast<const AT*>(a.q);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;

    const int T = a.KH * a.KW;
    const int qtiles = a.CQ / BQ;
    const int ntiles = (a.CP / BP) * qtiles;
    const int ngroups = a.nsplit * ntiles;
    const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
    const int tap = qq % T;
    const int group = (qq / T) * 8 + xcd;
    if (group >= ngroups) return;
    const int split = group / ntiles;
    const int tile = group - split * ntiles;
    const int tp = tile / qtiles;
    const int tq = tile - tp * qtiles;
    const int p0 = tp * BP, q0 = tq * BQ;
    const int r = tap / a.KW, s = tap - r * a.KW;

    const int M = a.N * a.OH * a.OW;
    const int mbeg = split * rows_per_split;
    const int mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
    const int nchunk = (mend > mbeg) ? (mend - mbeg + BRH - 1) / BRH : 0;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    areg_t rp[NP][4], rq[NQ][4];
    bool pok[NP][4], qok[NQ][4];
    // coordinates of the first pixel of each Q micro-tile, advanced by BRH per chunk
    int qn[NQ], qy[NQ], qx[NQ];
    // per-thread on-load affine of its channels (identity when absent: x*1+0 and max(x,-inf) are exact)
    const float relu_floor = (a.q_scale && a.q_relu) ? 0.f : -INFINITY;
    float psc[NP][CH], psh[NP][CH], qsc[NQ][CH], qsh[NQ][CH];
#pragma unroll
    for (int t = 0; t < NP; ++t) {
        int cg, pg;
        wgrad_tile_coord<CGP, KB>(tid + 256 * t, cg, pg);
        cg = (tid + 256 * t < TP_) ? cg : 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            psc[t][c] = a.p_scale ? a.p_scale[p0 + cg * CH + c] : 1.f;
            psh[t][c] = a.p_scale ? a.p_shift[p0 + cg * CH + c] : 0.f;
        }
    }
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        int cg, pg;
        wgrad_tile_coord<CGQ, KB>(tid + 256 * t, cg, pg);
        cg = (tid + 256 * t < TQ_) ? cg : 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            qsc[t][c] = a.q_scale ? a.q_scale[q0 + cg * CH + c] : 1.f;
            qsh[t][c] = a.q_scale ? a.q_shift[q0 + cg * CH + c] : 0.f;
        }
    }
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        int cg, pg;
        wgrad_tile_coord<CGQ, KB>(tid + 256 * t, cg, pg);
        const int m = mbeg + 4 * pg;
        const int ohw = a.OH * a.OW;
        qn[t] = m / ohw;
        const int rem = m - qn[t] * ohw;
        qy[t] = rem / a.OW;
        qx[t] = rem - qy[t] * a.OW;
    }

    for (int ch = -1; ch < nchunk; ++ch) {
        const bool more = ch + 1 < nchunk;
        if (more) {
            const int mc = mbeg + (ch + 1) * BRH;
#pragma unroll
            for (int t = 0; t < NP; ++t) {
                const int u = tid + 256 * t;
                int cg, pg;
                wgrad_tile_coord<CGP, KB>(u, cg, pg);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = mc + 4 * pg + i;
                    pok[t][i] = (m < mend) && (u < TP_);
                    const int ms = pok[t][i] ? m : 0;
                    rp[t][i] = *reinterpret_cast<const areg_t*>(pin + (size_t)ms * (size_t)a.CP + (size_t)(p0 + cg * CH));
                }
            }
#pragma unroll
            for (int t = 0; t < NQ; ++t) {
                const int u = tid + 256 * t;
                int cg, pg;
                wgrad_tile_coord<CGQ, KB>(u, cg, pg);
                int n = qn[t], y = qy[t], x = qx[t];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = mc + 4 * pg + i;
                    const int iy = y * a.S + r - a.P;
                    const int ix = x * a.S + s - a.P;
                    qok[t][i] = (m < mend) && (u < TQ_) && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    const int pix = qok[t][i] ? ((n * a.H + iy) * a.W + ix) : 0;
                    rq[t][i] = *reinterpret_cast<const areg_t*>(qin + (size_t)pix * (size_t)a.CQ + (size_t)(q0 + cg * CH));
                    if (++x >= a.OW) { x = 0; if (++y >= a.OH) { y = 0; ++n; } }
                }
                qx[t] += BRH;
                while (qx[t] >= a.OW) { qx[t] -= a.OW; ++qy[t]; }
                while (qy[t] >= a.OH) { qy[t] -= a.OH; ++qn[t]; }
            }
        }
        if (ch >= 0) {
            const int buf = ch & 1;
#pragma unroll
            for (int g = 0; g < BRH / 16; ++g) {
                bf16x8 af[MT], bf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[i] = *reinterpret_cast<const bf16x8*>(&sP[buf][((wm * MT + i) * 32 + l31) * LD + g * 16 + kh * 8]);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bf[j] = *reinterpret_cast<const bf16x8*>(&sQ[buf][((wn * NT + j) * 32 + l31) * LD + g * 16 + kh * 8]);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (more) {
            const int buf = (ch + 1) & 1;
#pragma unroll
            for (int t = 0; t < NP; ++t) {
                const int u = tid + 256 * t;
                int cg, pg;
                wgrad_tile_coord<CGP, KB>(u, cg, pg);
                if (u < TP_) {
                    float v[4][CH];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            const float f = (float)rp[t][i][c] * psc[t][c] + psh[t][c];
                            v[i][c] = pok[t][i] ? f : 0.f;
                        }
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const f32x4 col = {v[0][c], v[1][c], v[2][c], v[3][c]};   // channel c of the 4 pixels
                        *reinterpret_cast<bf16x4*>(&sP[buf][(cg * CH + c) * LD + pg * 4]) = __builtin_convertvector(col, bf16x4);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < NQ; ++t) {
                const int u = tid + 256 * t;
                int cg, pg;
                wgrad_tile_coord<CGQ, KB>(u, cg, pg);
                if (u < TQ_) {
                    float v[4][CH];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            const float f = fmaxf((float)rq[t][i][c] * qsc[t][c] + qsh[t][c], relu_floor);
                            v[i][c] = qok[t][i] ? f : 0.f;
                        }
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const f32x4 col = {v[0][c], v[1][c], v[2][c], v[3][c]};
                        *reinterpret_cast<bf16x4*>(&sQ[buf][(cg * CH + c) * LD + pg * 4]) = __builtin_convertvector(col, bf16x4);
                    }
                }
            }
        }
        __syncthreads();
    }

    float* out = a.partial + (size_t)split * (size_t)a.CP * (size_t)T * (size_t)a.CQ;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int prow = p0 + (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int qcol = q0 + (wn * NT + j) * 32 + l31;
                out[((size_t)prow * (size_t)T + (size_t)tap) * (size_t)a.CQ + (size_t)qcol] = acc[i][j][e];
            }
        }
}

// out = beta*out + sum_k partial[k]: G threads share one float4 of the result, thread g adds slabs g, g+G, ... and the G
// partial sums are combined through LDS in a fixed order (deterministic).  G > 1 keeps the chip busy when the result
// is small and the slab count large (layer1 at small batch: 147 KB result, > 100 slabs).
template <int G>
__device__ __forceinline__ void splitk_reduce_body(const float* __restrict__ partial, int nsplit, long long count4, float* __restrict__ out, float beta)
{
    constexpr int E = 256 / G;
    __shared__ __attribute__((aligned(16))) f32x4 red[256];
    const int e = threadIdx.x % E, g = threadIdx.x / E;
    const long long i = (long long)blockIdx.x * E + e;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < count4) {
        const f32x4* src = reinterpret_cast<const f32x4*>(partial) + i;
        int k = g;
        for (; k + 3 * G < nsplit; k += 4 * G) {     // 4 independent loads in flight
            const f32x4 v0 = src[(long long)k * count4], v1 = src[(long long)(k + G) * count4];
            const f32x4 v2 = src[(long long)(k + 2 * G) * count4], v3 = src[(long long)(k + 3 * G) * count4];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < nsplit; k += G) s += src[(long long)k * count4];
    }
    if (G > 1) {
        red[g * E + e] = s;
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int gg = 1; gg < G; ++gg) s += red[gg * E + e];
        }
    }
    if (g == 0 && i < count4) {
        f32x4* o = reinterpret_cast<f32x4*>(out) + i;
        if (beta != 0.f) s += beta * *o;
        *o = s;
    }
}

template <int G>
__global__ __launch_bounds__(256) void splitk_reduce_f32(const float* __restrict__ partial, int nsplit, long long count4,
                                                         float* __restrict__ out, float beta)
{
    splitk_reduce_body<G>(partial, nsplit, count4, out, beta);
}

// blockIdx.y = member of a grouped weight-gradient launch: its slabs lie at partial + member * nsplit * count
struct SplitkOuts { float* out[kLbcWgradGroupMax]; };
template <int G>
__global__ __launch_bounds__(256) void splitk_reduce_group_f32(const float* __restrict__ partial, int nsplit, long long count4, SplitkOuts outs)
{
    splitk_reduce_body<G>(partial + (size_t)blockIdx.y * (size_t)nsplit * (size_t)count4 * 4, nsplit, count4, outs.out[blockIdx.y], 0.f);
}

// 128x128 tiles when the channel counts allow and the reduction is long enough to amortise them; short reductions
// (small batches, late layers) take 64x64 tiles: 4x the tiles, so far fewer split-K slabs to write and re-read.
// Threshold measured on MI355X (scripts/bench_ops.py): 4096 rows beats 16384 at batch 256 (layer4: 0.32 -> 0.23 ms)
// and is neutral at batch 32.
inline bool big_tile(const WgradArgs& a)
{
    const long long min_m = 4096;
    return a.CP % 128 == 0 && a.CQ % 128 == 0 && (long long)a.N * a.OH * a.OW >= min_m;
}

}  // namespace

int lbc_wgrad_pick_split(const WgradArgs& a)
{
    if (lbc_wgrad_tr_eligible(a)) return lbc_wgrad_tr_pick_split(a);
    if (lbc_wgrad_tr2_eligible(a)) return lbc_wgrad_tr2_pick_split(a);
    const int bp = big_tile(a) ? 128 : 64;
    const long long tiles = (long long)(a.CP / bp) * (a.CQ / bp) * a.KH * a.KW;
    const long long M = (long long)a.N * a.OH * a.OW;
    const long long chunks = (M + BR - 1) / BR;
    const long long target = 1024;       // workgroups per launch the split count aims at (r01_run9_wgrad_block_target_*)
    long long ns = (target + tiles - 1) / tiles;
    if (ns < 1) ns = 1;
    if (ns > 256) ns = 256;
    // keep at least 8 chunks (256 pixels) of reduction per split
    const long long maxns = chunks / 8 > 0 ? chunks / 8 : 1;
    if (ns > maxns) ns = maxns;
    return (int)ns;
}

int lbc_wgrad_launch(const WgradArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.CP % 64 == 0 && a.CQ % 64 == 0, "wgrad: channels (%d,%d) must be multiples of 64", a.CP, a.CQ);
    LBC_REQUIRE(a.nsplit >= 1, "wgrad: nsplit %d", a.nsplit);
    LBC_REQUIRE(!a.act_bf16 || a.bf16, "wgrad: bf16 operands need bf16 = 1");
    const long long M = (long long)a.N * a.OH * a.OW;
    LBC_REQUIRE(M > 0 && M * a.CP < (1ll << 31) && (long long)a.N * a.H * a.W * a.CQ < (1ll << 31), "wgrad: bad tensor size");
    if (lbc_wgrad_tr_eligible(a)) return lbc_wgrad_tr_launch(a, s);
    if (lbc_wgrad_tr2_eligible(a)) return lbc_wgrad_tr2_launch(a, s);
    const int br = a.bf16 ? 64 : BR;
    const long long chunks = (M + br - 1) / br;
    const int rows_per_split = (int)((chunks + a.nsplit - 1) / a.nsplit) * br;
    LbcProfScope prof("conv_wgrad", 2.0 * M * a.CP * (double)a.CQ * a.KH * a.KW,
                      (a.act_bf16 ? 2.0 : 4.0) * ((double)M * a.CP + (double)a.N * a.H * a.W * a.CQ) +
                          4.0 * (double)a.nsplit * a.CP * a.KH * a.KW * a.CQ, s);
    const int bt = big_tile(a) ? 128 : 64;
    const long long ngroups = (long long)a.nsplit * (a.CP / bt) * (a.CQ / bt);
    const dim3 grid((unsigned)(((ngroups + 7) / 8) * 8 * a.KH * a.KW));
    // (KB = channel-group bits in the low lane bits of the staging writes, see wgrad_tile_coord: the measured best per element type and tile)
    if (a.act_bf16) {
        if (big_tile(a)) hipLaunchKernelGGL((conv_wgrad_bf16_k<128, 128, __bf16, 1>), grid, dim3(256), 0, s, a, rows_per_split);
        else             hipLaunchKernelGGL((conv_wgrad_bf16_k<64, 64, __bf16, 2>), grid, dim3(256), 0, s, a, rows_per_split);
    } else if (a.bf16) {
        if (big_tile(a)) hipLaunchKernelGGL((conv_wgrad_bf16_k<128, 128, float, 2>), grid, dim3(256), 0, s, a, rows_per_split);
        else             hipLaunchKernelGGL((conv_wgrad_bf16_k<64, 64, float, 2>), grid, dim3(256), 0, s, a, rows_per_split);
    } else if (big_tile(a)) hipLaunchKernelGGL((conv_wgrad_f32<128, 128>), grid, dim3(256), 0, s, a, rows_per_split);
    else                    hipLaunchKernelGGL((conv_wgrad_f32<64, 64>), grid, dim3(256), 0, s, a, rows_per_split);
    return lbc_check_launch("conv_wgrad_f32");
}

int lbc_splitk_reduce(const float* partial, int nsplit, long long count, float* out, float beta, hipStream_t s)
{
    LBC_REQUIRE(count % 4 == 0, "splitk_reduce: count %lld not a multiple of 4", count);
    const long long c4 = count / 4;
    // threads per result element: enough for >= 512 workgroups, at most a quarter of the slab count
    int G = 1;
    while (G < 16 && c4 * G < 512 * 256 && 4 * G <= nsplit) G *= 2;
    const unsigned blocks = (unsigned)((c4 * G + 255) / 256);
    LbcProfScope prof("splitk_reduce", 0.0, 4.0 * (double)count * (nsplit + 1), s);
    switch (G) {
        case 1: hipLaunchKernelGGL(splitk_reduce_f32<1>, dim3(blocks), dim3(256), 0, s, partial, nsplit, c4, out, beta); break;
        case 2: hipLaunchKernelGGL(splitk_reduce_f32<2>, dim3(blocks), dim3(256), 0, s, partial, nsplit, c4, out, beta); break;
        case 4: hipLaunchKernelGGL(splitk_reduce_f32<4>, dim3(blocks), dim3(256), 0, s, partial, nsplit, c4, out, beta); break;
        case 8: hipLaunchKernelGGL(splitk_reduce_f32<8>, dim3(blocks), dim3(256), 0, s, partial, nsplit, c4, out, beta); break;
        default: hipLaunchKernelGGL(splitk_reduce_f32<16>, dim3(blocks), dim3(256), 0, s, partial, nsplit, c4, out, beta); break;
    }
    return lbc_check_launch("splitk_reduce_f32");
}

int lbc_splitk_reduce_group(const float* partial, int nsplit, long long count, int n, float* const* out, hipStream_t s)
{
    LBC_REQUIRE(count % 4 == 0 && n >= 1 && n <= kLbcWgradGroupMax, "splitk_reduce_group: count %lld, %d members", count, n);
    const long long c4 = count / 4;
    SplitkOuts o;
    memset(&o, 0, sizeof(o));
    for (int i = 0; i < n; ++i) o.out[i] = out[i];
    int G = 1;
    while (G < 16 && c4 * G * n < 512 * 256 && 4 * G <= nsplit) G *= 2;
    const dim3 grid((unsigned)((c4 * G + 255) / 256), (unsigned)n);
    LbcProfScope prof("splitk_reduce", 0.0, 4.0 * (double)count * (nsplit + 1) * n, s);
    switch (G) {
        case 1: hipLaunchKernelGGL(splitk_reduce_group_f32<1>, grid, dim3(256), 0, s, partial, nsplit, c4, o); break;
        case 2: hipLaunchKernelGGL(splitk_reduce_group_f32<2>, grid, dim3(256), 0, s, partial, nsplit, c4, o); break;
        case 4: hipLaunchKernelGGL(splitk_reduce_group_f32<4>, grid, dim3(256), 0, s, partial, nsplit, c4, o); break;
        case 8: hipLaunchKernelGGL(splitk_reduce_group_f32<8>, grid, dim3(256), 0, s, partial, nsplit, c4, o); break;
        default: hipLaunchKernelGGL(splitk_reduce_group_f32<16>, grid, dim3(256), 0, s, partial, nsplit, c4, o); break;
    }
    return lbc_check_launch("splitk_reduce_group_f32");
}
