// 3x3 / stride 1 / pad 1 convolution (forward, and the input gradient which is the same operator with flipped taps)
// for bf16 tensors and bf16 weight copies on gfx950: the 29 + 29 hottest launches of a ResNet-34 step.
// reference: BasicBlock conv1/conv2, bird_view/models/resnet.py:15-22,38-54 and their autograd.
//
// Why a second kernel next to conv_igemm.hip: the generic implicit GEMM re-stages the A operand once per filter tap,
// i.e. nine nearly identical 128 x 64 pixel tiles per 64-channel slab, and on this chip a register-staged tile costs
// LDS-write issue (~80 B/clk/CU) and vector-memory issue (64 B/clk/CU) in proportion to the bytes staged: at 128 x 64
// output tiles (layer 1, Cout = 64) those two pipes, not the MFMAs, set the time (282 TF/s measured).  Here a workgroup
// owns 128 consecutive pixels of the flattened (n, y, x) raster and stages the *halo* [m0 - W - 1, m0 + 127 + W + 1] of
// a 64-channel slab ONCE (128 + 2W + 2 rows instead of 9 x 128); the tap (r, s) is the same LDS image read at a row
// offset r*W + s.  Taps that fall outside the image are zeroed per lane when the fragment is read (a 9-bit validity
// mask per output pixel), so the staged halo needs no padding logic, and the producing BatchNorm(+ReLU) is applied once
// per staged element instead of once per tap.  The next slab's halo is prefetched into registers under the nine taps of
// the current one; the weight tile of the next tap is double-buffered as in conv_igemm.hip.
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include <stdlib.h>

namespace {

constexpr int kHaloRowsMax = 336;   // 128 + 2*W + 2 with W <= 103

template <int BN, int MODE>
__global__ __launch_bounds__(256) void conv3x3_halo_k(IgemmArgs a)
{
    constexpr int BM = 128, BK = 64, LDK = BK + 8;   // 144-byte LDS rows: conflict-free ds_read_b128
    constexpr int MT = 2, NT = BN / 64;              // 4 waves as 2 x 2: 64 rows x BN/2 columns per wave
    constexpr int RB = BN / 32;                      // 16-byte weight loads per thread per (slab, tap)
    constexpr int HJ = (kHaloRowsMax * 8 + 255) / 256;   // 16-byte halo loads per thread per slab (11)
    __shared__ __attribute__((aligned(16))) __bf16 sH[kHaloRowsMax * LDK];
    __shared__ __attribute__((aligned(16))) __bf16 sB[2][BN * LDK];
    __shared__ float sRed[4 * BN];                   // statistics: [2 wm][2][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int ntn = a.K / BN;
    const int W = a.W, H = a.H, C = a.C;
    const int M = a.M;                       // N * H * W
    const int HR = BM + 2 * W + 2;           // halo rows
    const int ntiles = ((M + BM - 1) / BM) * ntn;
    const int G = (int)gridDim.x;
    const __bf16* xin = static_cast<const __bf16*>(a.x);
    const __bf16* win = static_cast<const __bf16*>(a.w);
    __bf16* yout = static_cast<__bf16*>(a.y);
    const __bf16* resid = static_cast<const __bf16*>(a.resid);

    const int seg = tid & 7;        // 16-byte segment (8 channels) of a 64-channel row
    const int row0 = tid >> 3;      // staging row (+32 per pass)
    const float relu_floor = (a.pre_scale && a.pre_relu) ? 0.f : -INFINITY;
    const int nslab = C / BK;
    const int nit = nslab * 9;

    bf16x8 rh[HJ];
    bf16x8 rb[RB];
    f32x8 lps = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, lpt = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // XCD-aware tile order (see conv_igemm.hip), applied to the persistent sequence lin = blockIdx.x + k * gridDim.x:
    // gridDim.x is a multiple of 8 whenever it is smaller than the tile count, so a workgroup stays on one XCD's range.
    const int xq = ntiles >> 3, xr = ntiles & 7;
    int lin = (int)blockIdx.x;
    int par = 0;                    // LDS weight buffer of the current step
    bool first = true;
    while (lin < ntiles) {
        const int xcd = lin & 7;
        const int tile_id = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
        const int mtile = tile_id / ntn;
        const int m0 = mtile * BM;
        const int n0 = (tile_id - mtile * ntn) * BN;
        const int hbase = m0 - W - 1;            // raster index of halo row 0
        const int nlin = lin + G;
        const bool has_next = nlin < ntiles;
        int nx_hbase = 0, nx_n0 = 0;
        if (has_next) {
            const int xc2 = nlin & 7;
            const int t2 = (xc2 < xr ? xc2 * (xq + 1) : xr * (xq + 1) + (xc2 - xr) * xq) + (nlin >> 3);
            const int mt2 = t2 / ntn;
            nx_hbase = mt2 * BM - W - 1;
            nx_n0 = (t2 - mt2 * ntn) * BN;
        }

        // validity of the 9 taps for this lane's MT output pixels
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

        f32x16 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // it = -1 (first tile of this workgroup only): stage slab 0's halo and tap 0's weights.  Every later tile finds
        // them in place: they were requested under the previous tile's last slab.
        for (int it = first ? -1 : 0; it < nit; ++it) {
            const int ci = it < 0 ? -1 : it / 9;
            const int t = it < 0 ? 8 : it - ci * 9;
            const bool last = it == nit - 1;
            // halo request: the next slab of this tile at its first tap, or slab 0 of the next tile under the last slab
            bool reqH = false;
            int rq_hbase = hbase, rq_c0 = 0;
            if (it < 0) { reqH = true; }
            else if (t == 0) {
                if (ci + 1 < nslab) { reqH = true; rq_c0 = (ci + 1) * BK; }
                else if (has_next) { reqH = true; rq_hbase = nx_hbase; }
            }
            if (reqH) {
                if (a.pre_scale) {
                    lps = ParamVec<8>::ld(a.pre_scale + rq_c0 + seg * 8);
                    lpt = ParamVec<8>::ld(a.pre_shift + rq_c0 + seg * 8);
                }
#pragma unroll
                for (int j = 0; j < HJ; ++j) {
                    int q = rq_hbase + row0 + 32 * j;             // rows past HR / outside the tensor read a clamped address:
                    q = q < 0 ? 0 : (q >= M ? M - 1 : q);         // they are only ever consumed by masked taps
                    rh[j] = *reinterpret_cast<const bf16x8*>(xin + (size_t)q * (size_t)C + (size_t)(rq_c0 + seg * 8));
                }
            }
            // weight request: the next (slab, tap) of this tile, or (0, 0) of the next tile
            const bool reqB = !last || has_next;
            if (reqB) {
                const int nx = last ? 0 : it + 1;
                const int ci2 = nx / 9, t2 = nx - ci2 * 9;
                const int bn0 = last ? nx_n0 : n0;
#pragma unroll
                for (int j = 0; j < RB; ++j)
                    rb[j] = *reinterpret_cast<const bf16x8*>(win + (size_t)(bn0 + row0 + 32 * j) * (size_t)(9 * C) + (size_t)(t2 * C + ci2 * BK + seg * 8));
            }
            if (it >= 0) {
                const int r = t / 3, s = t - 3 * r;
                const int shift = MODE == 0 ? r * W + s : (2 - r) * W + (2 - s);
                bool ok[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) ok[i] = (vmask[i] >> t) & 1;
#pragma unroll
                for (int g = 0; g < BK / 16; ++g) {
                    bf16x8 af[MT], bf[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        af[i] = *reinterpret_cast<const bf16x8*>(&sH[((wm * MT + i) * 32 + l31 + shift) * LDK + g * 16 + kh * 8]);
                        if (!ok[i]) af[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        bf[j] = *reinterpret_cast<const bf16x8*>(&sB[par][((wn * NT + j) * 32 + l31) * LDK + g * 16 + kh * 8]);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
            }
            const int nbuf = it < 0 ? par : par ^ 1;
            if (reqB) {
#pragma unroll
                for (int j = 0; j < RB; ++j) *reinterpret_cast<bf16x8*>(&sB[nbuf][(row0 + 32 * j) * LDK + seg * 8]) = rb[j];
            }
            if (it >= 0) par ^= 1;

            if (last) {
                // ---- epilogue of this tile: dense output rows, optional bias / residual / ReLU, statistics partials ----
                float s1[NT], s2[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (wm * MT + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                        const int m = m0 + row;
                        if (m < M) {
                            const size_t obase = (size_t)m * (size_t)a.K;
#pragma unroll
                            for (int nj = 0; nj < NT; ++nj) {
                                const int col = n0 + (wn * NT + nj) * 32 + l31;
                                float v = acc[mi][nj][e];
                                if (a.bias) v += a.bias[col];
                                if (resid) v += (float)resid[obase + col];
                                if (a.relu) v = fmaxf(v, 0.f);
                                yout[obase + col] = (__bf16)v;
                                s1[nj] += v;
                                s2[nj] += v * v;
                            }
                        }
                    }
                }
                if (a.stats) {
#pragma unroll
                    for (int nj = 0; nj < NT; ++nj) {
                        s1[nj] += __shfl_xor(s1[nj], 32);
                        s2[nj] += __shfl_xor(s2[nj], 32);
                    }
                    if (kh == 0) {
#pragma unroll
                        for (int nj = 0; nj < NT; ++nj) {
                            const int c = (wn * NT + nj) * 32 + l31;
                            sRed[(wm * 2 + 0) * BN + c] = s1[nj];
                            sRed[(wm * 2 + 1) * BN + c] = s2[nj];
                        }
                    }
                }
            }

            const bool storeH = (it < 0) || (t == 8 && (ci + 1 < nslab || has_next));
            if (storeH) {
                if (it >= 0) __syncthreads();      // every wave is done reading the current halo
#pragma unroll
                for (int j = 0; j < HJ; ++j) {
                    const int hr = row0 + 32 * j;
                    if (hr < HR) {
                        bf16x8 h = rh[j];
                        if (a.pre_scale) {         // BatchNorm(+ReLU) of the producer, once per staged element
                            f32x8 v = __builtin_convertvector(h, f32x8) * lps + lpt;
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], relu_floor);
                            h = __builtin_convertvector(v, bf16x8);
                        }
                        *reinterpret_cast<bf16x8*>(&sH[hr * LDK + seg * 8]) = h;
                    }
                }
            }
            __syncthreads();
            if (last && a.stats && tid < BN) {     // sRed is rewritten a whole tile (>= 9 barriers) later
                float* dst = a.stats + (size_t)(a.stat_row0 + mtile) * 2 * (size_t)a.K;
                dst[n0 + tid] = sRed[tid] + sRed[2 * BN + tid];
                dst[a.K + n0 + tid] = sRed[BN + tid] + sRed[3 * BN + tid];
            }
        }
        first = false;
        lin = nlin;
    }
}

}  // namespace

bool lbc_conv3x3_halo_eligible(const IgemmArgs& a, int mode)
{
    static const bool off = getenv("LBC_NO_HALO") && getenv("LBC_NO_HALO")[0] == '1';   // A/B switch
    return !off && a.w_bf16 && a.act_bf16 && a.KH == 3 && a.KW == 3 && a.S == 1 && a.P == 1 && a.ostep == 1 && a.oy0 == 0 &&
           a.ox0 == 0 && a.C % 64 == 0 && a.K % 64 == 0 && a.H == a.OH && a.W == a.OW && a.M == a.N * a.H * a.W &&
           128 + 2 * a.W + 2 <= kHaloRowsMax && (mode == 0 || mode == 1);
}

// bn = 64 or 128 output-channel tile (the caller's tile policy); BM is always 128
int lbc_conv3x3_halo_launch(const IgemmArgs& a, int mode, int bn, hipStream_t s)
{
    LBC_REQUIRE(lbc_conv3x3_halo_eligible(a, mode), "conv3x3_halo: launch not eligible");
    LBC_REQUIRE((bn == 64 || bn == 128) && a.K % bn == 0, "conv3x3_halo: bad column tile %d", bn);
    // persistent workgroups (2 fit a CU): the next tile's halo is prefetched under the current tile's MFMAs
    int nblk = lbc_cdiv(a.M, 128) * (a.K / bn);
    int cap = 512;
    if (const char* e = getenv("LBC_HALO_BLOCKS")) { const int v = atoi(e); if (v >= 8) cap = v & ~7; }   // tests: force multi-tile workgroups
    if (nblk > cap) nblk = cap;
    const dim3 grid((unsigned)nblk);
    if (bn == 128) {
        if (mode == 0) hipLaunchKernelGGL((conv3x3_halo_k<128, 0>), grid, dim3(256), 0, s, a);
        else           hipLaunchKernelGGL((conv3x3_halo_k<128, 1>), grid, dim3(256), 0, s, a);
    } else {
        if (mode == 0) hipLaunchKernelGGL((conv3x3_halo_k<64, 0>), grid, dim3(256), 0, s, a);
        else           hipLaunchKernelGGL((conv3x3_halo_k<64, 1>), grid, dim3(256), 0, s, a);
    }
    return lbc_check_launch("conv3x3_halo");
}
