// 3x3 / stride 1 / pad 1 convolution with C = K = 64 (forward, and the input gradient, which is the same operator with
// flipped taps) for bf16 tensors and bf16 weight copies on gfx950: the stem-resolution layer of the ResNets, 12 launches of
// a ResNet-34 step + 4 of the ResNet-18 teacher.  reference: BasicBlock conv1/conv2, bird_view/models/resnet.py:15-22,38-54.
//
// Why a second kernel next to conv_igemm.hip: the generic implicit GEMM re-stages the A operand once per filter tap,
// i.e. nine nearly identical 128 x 64 pixel tiles, and a register-staged tile costs LDS-write issue (~80 B/clk/CU) and
// vector-memory issue (64 B/clk/CU) in proportion to the bytes staged; with only 64 output channels per tile those two
// pipes and the per-tap barrier, not the MFMAs, set the time (282-298 TF/s measured).  Here a workgroup owns 128
// consecutive pixels of the flattened (n, y, x) raster and stages the *halo* [m0 - W - 1, m0 + 127 + W + 1] ONCE
// (128 + 2W + 2 rows instead of 9 x 128); tap (r, s) is the same LDS image read at a row offset r*W + s.  Taps that fall
// outside the image are zeroed per lane when the fragment is read (a 9-bit validity mask per output pixel), so the
// staged halo needs no padding logic, and the producing BatchNorm(+ReLU) is applied once per staged element instead of
// once per tap.  Measured on MI355X at batch 256 (M = 983,040): generic 0.243 ms -> 0.117 ms forward (618 TF/s).
// (The same halo staging with LDS-resident or double-buffered weight tiles for the wider layers was measured slower than
// the generic kernel: its 85-133 KB of LDS leave one workgroup per CU and the load / MFMA / store phases stop
// overlapping; see DESIGN.md.)
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include <stdlib.h>

namespace {

constexpr int kHaloRowsMax = 336;   // 128 + 2*W + 2 with W <= 103

// C = K = 64 (the stem-resolution layer: a quarter of a ResNet step's convolution FLOPs at the lowest arithmetic
// intensity).  Weight-stationary: every wave keeps the MFMA B fragments of its 32 output channels for all nine taps in
// 144 VGPRs for the life of the (persistent) workgroup, so LDS holds nothing but the halo (46 KB -> two workgroups per
// CU, whose load / MFMA / store phases overlap) and the MFMA loop reads one LDS fragment per MFMA with no barrier.
// Measured phase costs of the single-workgroup-per-CU predecessor (weights in LDS, 133 KB) were additive:
// skeleton 0.05 + MFMA 0.06 + output stores 0.05 + halo 0.03 ms per launch = 0.185 ms.
template <int MODE, bool BNB>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_k(IgemmArgs a)
{
    constexpr int BM = 128, BN = 64, BK = 64, LDK = BK + 8;
    constexpr int MT = 2;                                  // 4 waves as 2 x 2: 64 rows x 32 columns per wave
    constexpr int HJ = (kHaloRowsMax * 8 + 255) / 256;     // 11
    __shared__ __attribute__((aligned(16))) __bf16 sH[kHaloRowsMax * LDK];
    __shared__ float sRed[2][4 * BN];   // by tile parity: a fast wave may finish the next tile before a slow one has read this one's

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int W = a.W, H = a.H;
    const int M = a.M;
    const int HR = BM + 2 * W + 2;
    const int ntiles = (M + BM - 1) / BM;
    const int G = (int)gridDim.x;
    const __bf16* xin = static_cast<const __bf16*>(a.x);
    const __bf16* win = static_cast<const __bf16*>(a.w);
    __bf16* yout = static_cast<__bf16*>(a.y);
    const __bf16* resid = static_cast<const __bf16*>(a.resid);
    const int seg = tid & 7, row0 = tid >> 3;
    const float relu_floor = (a.pre_scale && a.pre_relu) ? 0.f : -INFINITY;

    f32x8 lps = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, lpt = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.pre_scale) { lps = ParamVec<8>::ld(a.pre_scale + seg * 8); lpt = ParamVec<8>::ld(a.pre_shift + seg * 8); }

    // stationary weights: B fragment of (tap t, 16-channel group g) for output channel wn*32 + l31: w[k][t][g*16 + kh*8 ..]
    bf16x8 wreg[9][BK / 16];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int g = 0; g < BK / 16; ++g)
            wreg[t][g] = *reinterpret_cast<const bf16x8*>(win + ((unsigned)(wn * 32 + l31) * (unsigned)(9 * BK) + (unsigned)(t * BK + g * 16 + kh * 8)));

    const int xq = ntiles >> 3, xr = ntiles & 7;
    int tpar = 0;
    for (int lin = (int)blockIdx.x; lin < ntiles; lin += G) {
        const int xcd = lin & 7;
        const int mtile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
        const int m0 = mtile * BM;
        const int hbase = m0 - W - 1;
        {
            bf16x8 rh[HJ];
#pragma unroll
            for (int j = 0; j < HJ; ++j) {
                int q = hbase + row0 + 32 * j;                // rows past HR / outside the tensor read a clamped address:
                q = q < 0 ? 0 : (q >= M ? M - 1 : q);         // they are only ever consumed by masked taps
                rh[j] = *reinterpret_cast<const bf16x8*>(xin + ((unsigned)q * (unsigned)BK + (unsigned)(seg * 8)));
            }
            __syncthreads();                                  // every wave is done reading the previous tile's halo
#pragma unroll
            for (int j = 0; j < HJ; ++j) {
                const int hr = row0 + 32 * j;
                if (hr < HR) {
                    bf16x8 h = rh[j];
                    if (a.pre_scale) {                        // BatchNorm(+ReLU) of the producer, once per staged element
                        f32x8 v = __builtin_convertvector(h, f32x8) * lps + lpt;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], relu_floor);
                        h = __builtin_convertvector(v, bf16x8);
                    }
                    *reinterpret_cast<bf16x8*>(&sH[hr * LDK + seg * 8]) = h;
                }
            }
        }
        int vmask[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + (wm * MT + i) * 32 + l31;
            int bits = 0;
            if (m < M) {
                const int x = m % W;
                const int y = (m / W) % H;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int r = t / 3, s = t - 3 * r;
                    const int dy = MODE == 0 ? r - 1 : 1 - r;
                    const int dx = MODE == 0 ? s - 1 : 1 - s;
                    if ((unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W) bits |= 1 << t;
                }
            }
            vmask[i] = bits;
        }
        __syncthreads();
        f32x16 acc[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int r = t / 3, s = t - 3 * r;
            const int shift = MODE == 0 ? r * W + s : (2 - r) * W + (2 - s);
#pragma unroll
            for (int g = 0; g < BK / 16; ++g) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    bf16x8 af = *reinterpret_cast<const bf16x8*>(&sH[((wm * MT + i) * 32 + l31 + shift) * LDK + g * 16 + kh * 8]);
                    if (!((vmask[i] >> t) & 1)) af = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wreg[t][g], acc[i], 0, 0, 0);
                }
            }
        }
        // epilogue.  The residual (identity gradient of the input-gradient launches) is fetched for the whole tile first:
        // read inside the store loop, every 2-byte load is waited for on its own (32 serial L2 round trips per tile).
        float s1 = 0.f, s2 = 0.f;
        const int col = wn * 32 + l31;
        // fused BatchNorm-backward reduce (IgemmArgs::bnb_*): this lane's channel constants
        // (BNB is a template parameter: the register file is full of weights, and the plain instantiation must not pay for it;
        //  the fused instantiation serves conv2's input gradient, which has no residual)
        const __bf16* bnb = BNB ? static_cast<const __bf16*>(a.bnb_y) : nullptr;
        float bsc = 0.f, bsh = 0.f, bmu = 0.f, biv = 0.f;
        if (BNB) { bsc = a.bnb_scale[col]; bsh = a.bnb_shift[col]; bmu = a.bnb_mean[col]; biv = a.bnb_invstd[col]; }
#pragma unroll
        for (int hb = 0; hb < MT * 2; ++hb) {          // 8 accumulator rows at a time (the register file is full of weights)
            const int mi = hb >> 1, e0 = (hb & 1) * 8;
            float rv[8], bv[8];
            if (!BNB && resid) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = e0 + k;
                    const int m = m0 + (wm * MT + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                    rv[k] = (float)resid[(unsigned)(m < M ? m : 0) * (unsigned)BN + (unsigned)col];
                }
            }
            if (BNB) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = e0 + k;
                    const int m = m0 + (wm * MT + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                    bv[k] = (float)bnb[(unsigned)(m < M ? m : 0) * (unsigned)BN + (unsigned)col];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + k;
                const int row = (wm * MT + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                const int m = m0 + row;
                if (m < M) {
                    const unsigned o = (unsigned)m * (unsigned)BN + (unsigned)col;
                    float v = acc[mi][e];
                    if (a.post_scale) v = v * a.post_scale[col] + a.post_shift[col];
                    if (a.bias) v += a.bias[col];
                    if (!BNB && resid) v += rv[k];
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (BNB) {
                        // sums of the STORED (bf16) gradient, as the separate reduce pass sees it
                        const float g = (bv[k] * bsc + bsh > 0.f) ? (float)(__bf16)v : 0.f;
                        yout[o] = (__bf16)g;
                        s1 += g;
                        s2 += g * (bv[k] - bmu) * biv;
                    } else {
                        yout[o] = (__bf16)v;
                        s1 += v;
                        s2 += v * v;
                    }
                }
            }
        }
        if (a.stats) {
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (kh == 0) { sRed[tpar][(wm * 2 + 0) * BN + col] = s1; sRed[tpar][(wm * 2 + 1) * BN + col] = s2; }
            __syncthreads();
            if (tid < BN) {
                float* dst = a.stats + (size_t)(a.stat_row0 + mtile) * 2 * (size_t)BN;
                dst[tid] = sRed[tpar][tid] + sRed[tpar][2 * BN + tid];
                dst[BN + tid] = sRed[tpar][BN + tid] + sRed[tpar][3 * BN + tid];
            }
            tpar ^= 1;
        }
    }
}

}  // namespace

bool lbc_conv3x3_halo_eligible(const IgemmArgs& a, int mode)
{
    return a.w_bf16 && a.act_bf16 && a.KH == 3 && a.KW == 3 && a.S == 1 && a.P == 1 && a.ostep == 1 && a.oy0 == 0 &&
           a.ox0 == 0 && a.C == 64 && a.K == 64 && a.H == a.OH && a.W == a.OW && a.M == a.N * a.H * a.W &&
           128 + 2 * a.W + 2 <= kHaloRowsMax && (mode == 0 || mode == 1);
}

int lbc_conv3x3_halo_launch(const IgemmArgs& a, int mode, hipStream_t s)
{
    LBC_REQUIRE(lbc_conv3x3_halo_eligible(a, mode), "conv3x3_halo: launch not eligible");
    // weights stationary in registers, persistent workgroups, two per CU
    int nb = lbc_cdiv(a.M, 128);
    int cap = 512;
    if (lbc_opt(kOptHaloBlocks) >= 8) cap = (int)lbc_opt(kOptHaloBlocks) & ~7;   // tests: force multi-tile workgroups
    if (nb > cap) nb = cap;
    if (a.bnb_y) {
        LBC_REQUIRE(mode == 1 && !a.resid, "conv3x3_halo: the fused BatchNorm-backward reduce serves input gradients without a residual");
        hipLaunchKernelGGL((conv3x3_c64_k<1, true>), dim3((unsigned)nb), dim3(256), 0, s, a);
    }
    else if (mode == 0) hipLaunchKernelGGL((conv3x3_c64_k<0, false>), dim3((unsigned)nb), dim3(256), 0, s, a);
    else                hipLaunchKernelGGL((conv3x3_c64_k<1, false>), dim3((unsigned)nb), dim3(256), 0, s, a);
    return lbc_check_launch("conv3x3_c64");
}
