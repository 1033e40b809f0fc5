// Implicit-GEMM convolution for gfx950 (MI355X): exact f32 on v_mfma_f32_32x32x2_f32, or bf16 operands (f32 / bf16
// tensors, f32 / bf16 weight copies) on v_mfma_f32_32x32x16_bf16.  One kernel template serves
//   * nn.Conv2d forward                (reference bird_view/models/resnet.py:15-22,102)
//   * nn.Conv2d input gradient         (autograd of the same call sites)
//   * nn.ConvTranspose2d forward       (reference bird_view/models/image.py:39,42,45)
//   * nn.ConvTranspose2d input gradient
// No im2col buffer exists anywhere: the A operand (pixels x channels of one
// filter tap) is gathered straight from the NHWC activation into LDS with
// 16-byte loads, zero-filled outside the image, optionally with the producing
// BatchNorm(+ReLU) applied on the fly (pre_scale/pre_shift).
//
// Tiling: a 256-thread workgroup (4 waves, 2x2) owns a BM x BN output tile and
// walks depth in (tap, 32/64/128-channel) chunks, double-buffered through LDS with
// the global loads of the next chunk (the chunk after next in the all-bf16
// kernels) in flight under the current chunk's MFMAs.  LDS rows are padded by
// 16 bytes (144 / 272-byte rows) so ds_read_b128 fragment reads are conflict free.
// f32 path: a lane's f32x4 fragment holds channels {4*(l>>5)+i}; MFMA step i
// contracts the channel pair {i, 4+i} of each 8-channel group -- A and B use the
// same pairing, so the sum over depth is complete.  Stride-2 transposed launches
// split into four output-parity phases (no MFMA on structurally zero taps), all
// four in one grid.  The C = K = 64 3x3 layers of the all-bf16 mode go to
// conv_halo.hip instead.
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include <type_traits>
#include <stdlib.h>

namespace {

// BF16 = false: exact-f32 MFMA (32x32x2), 32-channel chunks, float LDS tiles.
// BF16 = true : operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) when the tile is written to LDS and multiplied on
//               v_mfma_f32_32x32x16_bf16 with f32 accumulation; activations, weights, statistics and everything outside the
//               MFMA stay f32 in HBM.  64-channel chunks, bf16 LDS tiles ([row][k] only: weights must be depth-contiguous).
// AT = element type of the activation tensors x / y / resid in HBM (float, or __bf16 with BF16 = true): bf16 activations
// are loaded 8 channels per 16-byte load and go to LDS without conversion unless a BatchNorm-on-load prologue is set.
// WT = element type of the weight operand in HBM (float, or __bf16 with BF16 = true: a per-step bf16 copy, lbc_weight_prep).
// PF = prefetch distance in depth chunks: the global loads of chunk it+PF are issued while chunk it is multiplied and are
// written to LDS one chunk before their use, from PF register sets.  With two workgroups per CU a chunk lasts about one
// loaded-L2 round trip (measured 1.5 us per chunk = 7x its MFMA time at PF = 1), so the all-bf16 kernels run PF = 2.
// BKV = channels per depth chunk (0: 64 for the bf16 paths, 32 for f32).  The 64 x 64 tiles of small launches take 128: a chunk
// there is 4 MFMAs per wave, so its cost is the barrier and the load round trip, and twice the depth halves their number.
template <int BM, int BN, bool WMAJOR, int MODE, bool BF16, typename AT, typename WT, int PF, int BKV = 0>
__global__ __launch_bounds__(256, PF == 2 ? 2 : 1) void conv_igemm_k(IgemmArgs a)
{
    static_assert(!BF16 || WMAJOR, "the bf16 path needs depth-contiguous weights");
    static_assert(!Act<AT>::kBf16 || BF16, "bf16 activations need the bf16 MFMA path");
    static_assert(!Act<WT>::kBf16 || BF16, "bf16 weights need the bf16 MFMA path");
    constexpr bool ABF = Act<AT>::kBf16;
    constexpr bool WBF = Act<WT>::kBf16;
    using breg_t = typename std::conditional<WBF, bf16x8, f32x4>::type;    // one 16-byte weight load
    constexpr int BEL = WBF ? 8 : 4;
    using areg_t = typename std::conditional<ABF, bf16x8, f32x4>::type;   // one 16-byte activation load
    using lds_t = typename std::conditional<BF16, __bf16, float>::type;
    constexpr int BK = BKV ? BKV : (BF16 ? 64 : 32);      // channels per depth chunk
    constexpr int LDK = BK + (BF16 ? 8 : 4);   // padded LDS row (elements): 144-byte (or 272-byte) rows -> conflict-free b128 reads
    constexpr int SEGS = BK / BEL;          // 16-byte segments per weight-tile row
    constexpr int RPP = 256 / SEGS;         // tile rows staged per pass of the 256 threads
    constexpr int WM = 2, WN = 2;
    constexpr int MT = BM / WM / 32;
    constexpr int NT = BN / WN / 32;
    constexpr int ASEGS = ABF ? BK / 8 : BK / 4;   // 16-byte segments per A tile row
    constexpr int ARPP = 256 / ASEGS;
    constexpr int AEL = ABF ? 8 : 4;        // elements per A load
    constexpr int RA = BM / ARPP;           // A 16-byte loads per thread per chunk
    constexpr int RB = WMAJOR ? BN / RPP : BK * BN / 4 / 256;
    constexpr int LDN = BN + 4;             // padded LDS row for [k][n] tiles (f32 only)
    constexpr int SB = WMAJOR ? BN * LDK : BK * LDN;

    __shared__ __attribute__((aligned(16))) lds_t sA[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) lds_t sB[2][SB];
    __shared__ int sTap[16];
    __shared__ int sNTap;
    __shared__ int sOpix[BM];   // output pixel of every tile row (strided / phase launches; dense launches use m itself)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch rule; speed only, never correctness).  The
    // remap hands every XCD one contiguous range of tiles, column tiles of the same rows first, so neighbouring tiles
    // (shared halo rows, shared A rows across column tiles) hit in that XCD's private L2.  Bijective for any tile count.
    const int ntn = a.K / BN;
    // all four output-parity phases of a stride-2 transposed launch in one grid: phase-major workgroup ranges
    int oy0 = a.oy0, ox0 = a.ox0, stat_row0 = a.stat_row0;
    int nwg = gridDim.x, wgb = blockIdx.x;
    if (a.nphase == 4) {
        nwg = gridDim.x >> 2;
        const int ph = blockIdx.x / nwg;
        wgb = blockIdx.x - ph * nwg;
        oy0 = ph >> 1; ox0 = ph & 1;
        stat_row0 += ph * (nwg / ntn);
    }
    int tile_id;
    {
        const int b = wgb;
        const int xcd = b & 7, q = nwg >> 3, rr = nwg & 7;
        tile_id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
    }
    const int mtile = tile_id / ntn;
    const int m0 = mtile * BM;
    const int n0 = (tile_id - mtile * ntn) * BN;
    const int T = a.KH * a.KW;

    if (tid == 0) {
        int nt = 0;
        for (int t = 0; t < T; ++t) {
            bool ok = true;
            if (MODE == 1 && a.S == 2) {
                const int r = t / a.KW, s = t - r * a.KW;
                ok = (((oy0 + a.P - r) & 1) == 0) && (((ox0 + a.P - s) & 1) == 0);
            }
            if (ok) sTap[nt++] = t;
        }
        sNTap = nt;
    }

    // ---- per-thread A row descriptors -------------------------------------
    const int seg = tid % SEGS;        // weight-tile staging role
    const int arow = tid / SEGS;
    const int aseg = tid % ASEGS;      // activation-tile staging role
    const int aarow = tid / ASEGS;
    const AT* xin = static_cast<const AT*>(a.x);
    int pixbase[RA];
    int ay[RA], ax[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        const int m = m0 + aarow + ARPP * j;
        if (m < a.M) {
            const int lhw = a.LH * a.LW;
            const int n = m / lhw;
            const int rem = m - n * lhw;
            const int ly = rem / a.LW;
            const int lx = rem - ly * a.LW;
            const int oy = ly * a.ostep + oy0;
            const int ox = lx * a.ostep + ox0;
            pixbase[j] = n * a.H * a.W;
            if (MODE == 0) { ay[j] = oy * a.S - a.P; ax[j] = ox * a.S - a.P; }
            else           { ay[j] = oy + a.P;       ax[j] = ox + a.P; }
            if (aseg == 0) sOpix[aarow + ARPP * j] = (n * a.OH + oy) * a.OW + ox;
        } else {
            pixbase[j] = 0; ay[j] = -(1 << 20); ax[j] = -(1 << 20);
        }
    }
    __syncthreads();
    const int ntap = sNTap;
    const int cpt = a.C / BK;
    const int nit = ntap * cpt;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    areg_t ra[PF][RA];      // native vector values (HIP's float4 struct would be copied through a scratch alloca)
    breg_t rb[PF][RB];
    bool aok[PF][RA];
    f32x4 lps[PF][AEL / 4], lpt[PF][AEL / 4];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int q = 0; q < AEL / 4; ++q) { lps[u][q] = f32x4{1.f, 1.f, 1.f, 1.f}; lpt[u][q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float relu_floor = (a.pre_scale && a.pre_relu) ? 0.f : -INFINITY;

    // Software pipeline, written out once (no lambdas: the staging arrays must stay in registers):
    //   iteration `it` issues the global loads of chunk it+1, runs the MFMAs of chunk it from LDS buffer it&1, then
    //   applies the on-load transform to chunk it+1 and writes it to the other LDS buffer; one barrier per chunk.
    // Loads are unconditional and branch-free (out-of-image taps read a valid dummy address and are zeroed when the
    // tile is written to LDS) and every use of a loaded value is deferred to the store phase, so all global loads of
    // chunk it+1 stay in flight underneath the MFMAs of chunk it.
    const int l31 = lane & 31;
    const int kh = lane >> 5;
    // it runs from -PF; the inner loop is unrolled PF times so that the register-set indices are compile-time constants:
    // chunk it+PF is loaded into set u, chunk it+1 is stored from set (u+1) % PF.
    for (int it0 = -PF; it0 < nit; it0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int it = it0 + u;
        if (it >= nit) break;
        const int LS = u;                     // register set receiving the loads of this step
        const int SS = (u + 1) % PF;          // register set written to LDS in this step
        if (it + PF < nit) {
            // depth order: 32-channel slab outer, filter taps inner -- the (up to 9) shifted gathers of one slab re-read
            // the same few KB per workgroup back to back (L1/L2 hits) instead of streaming the whole tile 9 times
            const int nx = it + PF;
            const int ci = nx / ntap;
            const int ti = nx - ci * ntap;
            const int c0 = ci * BK;
            const int tap = sTap[ti];
            const int r = tap / a.KW, s = tap - r * a.KW;
            if (a.pre_scale) {
#pragma unroll
                for (int q = 0; q < AEL / 4; ++q) {
                    lps[LS][q] = *reinterpret_cast<const f32x4*>(a.pre_scale + c0 + aseg * AEL + q * 4);
                    lpt[LS][q] = *reinterpret_cast<const f32x4*>(a.pre_shift + c0 + aseg * AEL + q * 4);
                }
            }
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                int iy, ix;
                if (MODE == 0) { iy = ay[j] + r; ix = ax[j] + s; }
                else if (a.S == 2) { iy = (ay[j] - r) >> 1; ix = (ax[j] - s) >> 1; }
                else { iy = ay[j] - r; ix = ax[j] - s; }
                const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const int pix = ok ? (pixbase[j] + iy * a.W + ix) : 0;
                aok[LS][j] = ok;
                ra[LS][j] = *reinterpret_cast<const areg_t*>(xin + ((size_t)pix * (size_t)a.C + (size_t)(c0 + aseg * AEL)));
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                size_t off;
                if (WMAJOR) {
                    off = (size_t)(n0 + arow + RPP * j) * (size_t)(T * a.C) + (size_t)(tap * a.C + c0 + seg * BEL);
                } else {
                    const int idx = tid + 256 * j;
                    const int krow = idx / (BN / 4);
                    const int s4 = idx - krow * (BN / 4);
                    off = (size_t)(c0 + krow) * (size_t)(T * a.K) + (size_t)(tap * a.K + n0 + s4 * 4);
                }
                rb[LS][j] = *reinterpret_cast<const breg_t*>(static_cast<const WT*>(a.w) + off);
            }
        }
        if (it >= 0) {
            const int buf = it & 1;
            if constexpr (BF16) {
#pragma unroll
                for (int g = 0; g < BK / 16; ++g) {
                    bf16x8 af[MT], bf[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        af[i] = *reinterpret_cast<const bf16x8*>(&sA[buf][((wm * MT + i) * 32 + l31) * LDK + g * 16 + kh * 8]);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        bf[j] = *reinterpret_cast<const bf16x8*>(&sB[buf][((wn * NT + j) * 32 + l31) * LDK + g * 16 + kh * 8]);
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                        for (int nj = 0; nj < NT; ++nj)
                            acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[nj], acc[mi][nj], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < BK / 8; ++g) {
                    f32x4 af[MT], bf[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        af[i] = *reinterpret_cast<const f32x4*>(&sA[buf][((wm * MT + i) * 32 + l31) * LDK + g * 8 + kh * 4]);
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if (WMAJOR) {
                            bf[j] = *reinterpret_cast<const f32x4*>(&sB[buf][((wn * NT + j) * 32 + l31) * LDK + g * 8 + kh * 4]);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                bf[j][i] = sB[buf][(g * 8 + kh * 4 + i) * LDN + (wn * NT + j) * 32 + l31];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                            for (int nj = 0; nj < NT; ++nj)
                                acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][i], bf[nj][i], acc[mi][nj], 0, 0, 0);
                }
            }
        }
        if (it + 1 >= 0 && it + 1 < nit) {
            const int buf = (it + 1) & 1;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                lds_t* dst = &sA[buf][(aarow + ARPP * j) * LDK + aseg * AEL];
                if constexpr (ABF) {
                    bf16x8 h = ra[SS][j];
                    if (a.pre_scale) {      // BatchNorm(+ReLU) on load: unpack, f32 affine, repack
                        f32x8 v = __builtin_convertvector(h, f32x8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] * lps[SS][e >> 2][e & 3] + lpt[SS][e >> 2][e & 3], relu_floor);
                        h = __builtin_convertvector(v, bf16x8);
                    }
                    if (!aok[SS][j]) h = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    *reinterpret_cast<bf16x8*>(dst) = h;
                } else {
                    f32x4 v = ra[SS][j] * lps[SS][0] + lpt[SS][0];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = aok[SS][j] ? fmaxf(v[e], relu_floor) : 0.f;
                    if constexpr (BF16) *reinterpret_cast<bf16x4*>(dst) = __builtin_convertvector(v, bf16x4);
                    else                *reinterpret_cast<f32x4*>(dst) = v;
                }
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                if constexpr (WBF) {
                    *reinterpret_cast<bf16x8*>(&sB[buf][(arow + RPP * j) * LDK + seg * 8]) = rb[SS][j];
                } else if constexpr (BF16) {
                    *reinterpret_cast<bf16x4*>(&sB[buf][(arow + RPP * j) * LDK + seg * 4]) = __builtin_convertvector(rb[SS][j], bf16x4);
                } else if (WMAJOR) {
                    *reinterpret_cast<f32x4*>(&sB[buf][(arow + RPP * j) * LDK + seg * 4]) = rb[SS][j];
                } else {
                    const int idx = tid + 256 * j;
                    const int krow = idx / (BN / 4);
                    const int s4 = idx - krow * (BN / 4);
                    *reinterpret_cast<f32x4*>(&sB[buf][krow * LDN + s4 * 4]) = rb[SS][j];
                }
            }
        }
        __syncthreads();
      }
    }

    // ---- epilogue ------------------------------------------------------------
    float s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

    // The residual is fetched per 32-row block before that block's stores: read inside the store loop, every narrow load
    // is waited for on its own (16 x NT serial L2 round trips per block of rows).
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        float rv[16][NT];
        if (a.resid) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * MT + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int m = m0 + row;
                const size_t ob = (m < a.M) ? (a.ostep == 1 ? (size_t)m * (size_t)a.K : (size_t)sOpix[row] * (size_t)a.K) : 0;
#pragma unroll
                for (int nj = 0; nj < NT; ++nj)
                    rv[r][nj] = Act<AT>::ld1(static_cast<const AT*>(a.resid) + ob + (size_t)(n0 + (wn * NT + nj) * 32 + l31));
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * MT + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const int m = m0 + row;
            if (m < a.M) {
                size_t obase;
                if (a.ostep == 1) {   // dense output (stride-1 either way, or a strided gather): row m is pixel m
                    obase = (size_t)m * (size_t)a.K;
                } else {              // one output-parity phase of a stride-2 transposed launch
                    obase = (size_t)sOpix[row] * (size_t)a.K;
                }
#pragma unroll
                for (int nj = 0; nj < NT; ++nj) {
                    const int col = n0 + (wn * NT + nj) * 32 + l31;
                    float v = acc[mi][nj][r];
                    if (a.post_scale) v = v * a.post_scale[col] + a.post_shift[col];
                    if (a.bias) v += a.bias[col];
                    if (a.resid) v += rv[r][nj];
                    if (a.relu) v = fmaxf(v, 0.f);
                    Act<AT>::st1(static_cast<AT*>(a.y) + obase + col, v);
                    s1[nj] += v;
                    s2[nj] += v * v;
                }
            }
        }
    }
    if (a.stats) {
        // combine the two half-waves (rows 4*kh+...), then the two M-waves through LDS
        float* red = reinterpret_cast<float*>(&sA[0][0]);   // [WM][2][BN]; the main loop's last barrier has passed
#pragma unroll
        for (int nj = 0; nj < NT; ++nj) {
            s1[nj] += __shfl_xor(s1[nj], 32);
            s2[nj] += __shfl_xor(s2[nj], 32);
        }
        if (kh == 0) {
#pragma unroll
            for (int nj = 0; nj < NT; ++nj) {
                const int c = (wn * NT + nj) * 32 + l31;
                red[(wm * 2 + 0) * BN + c] = s1[nj];
                red[(wm * 2 + 1) * BN + c] = s2[nj];
            }
        }
        __syncthreads();
        if (tid < BN) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { t1 += red[(w * 2 + 0) * BN + tid]; t2 += red[(w * 2 + 1) * BN + tid]; }
            float* dst = a.stats + (size_t)(stat_row0 + mtile) * 2 * (size_t)a.K;
            dst[n0 + tid] = t1;
            dst[a.K + n0 + tid] = t2;
        }
    }
}

// w[A][T][B] -> wt[B][T][A] (per tap a 2-D transpose through a padded 32x32 LDS tile).  Used to hand the
// input-gradient GEMM a depth-contiguous weight operand (the fast ds_read_b128 fragment path).
__global__ __launch_bounds__(256) void weight_transpose_k(const float* __restrict__ w, float* __restrict__ wt, int A, int T, int B)
{
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int a = a0 + j, b = b0 + tx;
        tile[j][tx] = (a < A && b < B) ? w[((size_t)a * T + t) * B + b] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int b = b0 + j, a = a0 + tx;
        if (a < A && b < B) wt[((size_t)b * T + t) * A + a] = tile[tx][j];
    }
}

// All convolution weights of a network in one launch: w[A][T][B] (f32) -> bf16 copy in the same layout and bf16
// transposed copy [B][T][A]; a block handles one 64 x 64 (a, b) tile of one tap of one tensor: 16-byte loads of four
// consecutive b, 8-byte stores of four bf16 in both layouts (4-byte loads and 2-byte stores on 32 x 32 tiles ran at 2.5 TB/s:
// 85 us for the 21 M weights of the ResNet-34 student, every step).  A and B are multiples of 4 for every tensor of the
// networks (channel counts); edge tiles are predicated per group of four.
__global__ __launch_bounds__(256) void weight_prep_k(WeightPrepArgs a)
{
    __shared__ float tile[64][65];
    int li = 0;
    for (int i = 1; i < a.count; ++i)
        if ((int)blockIdx.x >= a.item[i].tile_begin) li = i;
    const WeightPrepItem& it = a.item[li];
    const int A = it.A, T = it.T, B = it.B;
    const int nb = (B + 63) / 64, na = (A + 63) / 64;
    int rel = (int)blockIdx.x - it.tile_begin;
    const int bb = rel % nb; rel /= nb;
    const int ab = rel % na;
    const int t = rel / na;
    const int a0 = ab * 64, b0 = bb * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 groups of four x 16 rows
    __bf16* wn = static_cast<__bf16*>(it.wn);
    __bf16* wt = static_cast<__bf16*>(it.wt);
    for (int j = ty; j < 64; j += 16) {
        const int ai = a0 + j, bi = b0 + 4 * tx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ai < A && bi < B) {
            const size_t o = ((size_t)ai * T + t) * B + bi;
            v = *reinterpret_cast<const f32x4*>(it.w + o);
            *reinterpret_cast<bf16x4*>(wn + o) = __builtin_convertvector(v, bf16x4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[j][4 * tx + e] = v[e];
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 16) {
        const int bi = b0 + j, ai = a0 + 4 * tx;
        if (ai < A && bi < B) {
            const f32x4 v = {tile[4 * tx][j], tile[4 * tx + 1][j], tile[4 * tx + 2][j], tile[4 * tx + 3][j]};
            *reinterpret_cast<bf16x4*>(wt + ((size_t)bi * T + t) * A + ai) = __builtin_convertvector(v, bf16x4);
        }
    }
}

template <int BM, int BN>
int launch_cfg(const IgemmArgs& a, int wmajor, int mode, hipStream_t s)
{
    dim3 grid((unsigned)(lbc_cdiv(a.M, BM) * (a.K / BN) * (a.nphase == 4 ? 4 : 1)));
    if (a.w_bf16) {
        if constexpr (BM == 64 && BN == 64) {
            if (a.C % 128 == 0) {   // small launches: 128-channel chunks
                if (mode == 0) hipLaunchKernelGGL((conv_igemm_k<BM, BN, true, 0, true, __bf16, __bf16, 2, 128>), grid, dim3(256), 0, s, a);
                else           hipLaunchKernelGGL((conv_igemm_k<BM, BN, true, 1, true, __bf16, __bf16, 2, 128>), grid, dim3(256), 0, s, a);
                return lbc_check_launch("conv_igemm");
            }
        }
        if (mode == 0) hipLaunchKernelGGL((conv_igemm_k<BM, BN, true, 0, true, __bf16, __bf16, 2>), grid, dim3(256), 0, s, a);
        else           hipLaunchKernelGGL((conv_igemm_k<BM, BN, true, 1, true, __bf16, __bf16, 2>), grid, dim3(256), 0, s, a);
    } else if (a.act_bf16) {
        if (mode == 0) hipLaunchKernelGGL((conv_igemm_k<BM, BN, true, 0, true, __bf16, float, 1>), grid, dim3(256), 0, s, a);
        else           hipLaunchKernelGGL((conv_igemm_k<BM, BN, true, 1, true, __bf16, float, 1>), grid, dim3(256), 0, s, a);
    } else if (a.bf16) {
        if (mode == 0) hipLaunchKernelGGL((conv_igemm_k<BM, BN, true, 0, true, float, float, 1>), grid, dim3(256), 0, s, a);
        else           hipLaunchKernelGGL((conv_igemm_k<BM, BN, true, 1, true, float, float, 1>), grid, dim3(256), 0, s, a);
    } else if (wmajor && mode == 0) hipLaunchKernelGGL((conv_igemm_k<BM, BN, true, 0, false, float, float, 1>), grid, dim3(256), 0, s, a);
    else if (wmajor && mode == 1)   hipLaunchKernelGGL((conv_igemm_k<BM, BN, true, 1, false, float, float, 1>), grid, dim3(256), 0, s, a);
    else if (!wmajor && mode == 0)  hipLaunchKernelGGL((conv_igemm_k<BM, BN, false, 0, false, float, float, 1>), grid, dim3(256), 0, s, a);
    else                            hipLaunchKernelGGL((conv_igemm_k<BM, BN, false, 1, false, float, float, 1>), grid, dim3(256), 0, s, a);
    return lbc_check_launch("conv_igemm");
}

const int kCfgBM[3] = {128, 128, 64};
const int kCfgBN[3] = {64, 128, 64};

}  // namespace

int lbc_weight_transpose(const float* w, float* wt, int A, int T, int B, hipStream_t s)
{
    LbcProfScope prof("weight_transpose", 0.0, 8.0 * A * T * B, s);
    hipLaunchKernelGGL(weight_transpose_k, dim3((unsigned)lbc_cdiv(B, 32), (unsigned)lbc_cdiv(A, 32), (unsigned)T), dim3(256), 0, s, w, wt, A, T, B);
    return lbc_check_launch("weight_transpose");
}

int lbc_weight_prep(const WeightPrepArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.count >= 1 && a.count <= WeightPrepArgs::kMax && a.tiles > 0, "weight_prep: bad table");
    double elems = 0;
    for (int i = 0; i < a.count; ++i) {
        LBC_REQUIRE(a.item[i].A % 4 == 0 && a.item[i].B % 4 == 0, "weight_prep: channel counts must be multiples of 4 (tensor %d: %d x %d)", i, a.item[i].A, a.item[i].B);
        elems += (double)a.item[i].A * a.item[i].T * a.item[i].B;
    }
    LbcProfScope prof("weight_prep", 0.0, 8.0 * elems, s);
    hipLaunchKernelGGL(weight_prep_k, dim3((unsigned)a.tiles), dim3(256), 0, s, a);
    return lbc_check_launch("weight_prep");
}

int lbc_igemm_rows(const IgemmArgs& a, int cfg)
{
    if (cfg >= kLbcCfgHdma) return lbc_conv_hdma_rows(a, cfg);
    if (cfg >= kLbcCfgGlds) return lbc_conv_glds_rows(a, cfg);
    if (lbc_conv3x3_halo_eligible(a, 0)) return lbc_cdiv(a.M, 128);   // that kernel always works on 128-pixel tiles
    return lbc_cdiv(a.M, kCfgBM[cfg]);
}

bool lbc_igemm_fuses_bn_bwd(const IgemmArgs& a, int wmajor, int mode, int cfg)
{
    if (lbc_opt_on(kOptNoBnBwdFuse) || mode != 1 || !a.act_bf16) return false;
    if (cfg >= kLbcCfgGlds) return true;          // conv_glds2_k (lds_dma_epilogue), conv_hdmap_k / conv_c64p_k (EPI 2)
    return wmajor && !a.resid && lbc_conv3x3_halo_eligible(a, mode); // conv3x3_c64_k<1, true>
}

bool lbc_igemm_fuses_bn_bwd_masked(const IgemmArgs& a, int wmajor, int mode, int cfg)
{
    (void)wmajor;
    if (lbc_opt(kOptNoBnBwdFuse) >= 1 || mode != 1 || !a.act_bf16 || !a.resid) return false;     // (LBC_NO_BN_BWD_FUSE=2: only this form off)
    return cfg > kLbcCfgHdma && cfg != kLbcCfgHdma + 3;      // conv_hdmap_k<.., EPI 4> (every shape) and its split-K epilogue; not the 64-channel kernel
}

int lbc_igemm_pick_for(const IgemmArgs& a, int mode)
{
    if (lbc_opt(kOptForceCfg) < 0) {     // a forced tile policy pins conv_igemm.hip
        // 3x3 stride-1 launches: the halo-staged LDS-DMA kernels first (the 64-channel layer has its own persistent variant)
        {
            const int h = lbc_conv_hdma_pick(a, mode);
            if (h >= 0) return h;
        }
        const int g = lbc_conv_glds_pick(a, mode);
        // the 64-channel layers have two candidates: conv_halo.hip, unless the 512 x 64 LDS-DMA shape is selected
        if (g >= 0 && (g == kLbcCfgGlds + 4 || g == kLbcCfgGlds + 5 || !lbc_conv3x3_halo_eligible(a, mode))) return g;   // (+4, +5: the 64-column shapes, picked for such a layer only when pinned)
    }
    return lbc_igemm_pick(a.M, a.K);
}

int lbc_igemm_pick(long long M, int K)
{
    { const long long c = lbc_opt(kOptForceCfg); if (c >= 0 && c < 3 && K % kCfgBN[c] == 0) return (int)c; }   // tests / tuning
    // Prefer the largest tile that still gives the 256 CUs >= 1.5 waves of workgroups.
    const long long want = 384;
    if (K % 128 == 0 && ((M + 127) / 128) * (K / 128) >= want) return 1;
    if (((M + 127) / 128) * (K / 64) >= want) return 0;
    return 2;
}

int lbc_igemm_launch(const IgemmArgs& a, int wmajor, int mode, int cfg, hipStream_t s)
{
    LBC_REQUIRE(cfg >= 0 && cfg < kLbcCfgHdma + kLbcHdmaCfgs, "igemm: bad cfg %d", cfg);
    LBC_REQUIRE(a.C % (a.bf16 ? 64 : 32) == 0, "igemm: gathered channels %d not a multiple of %d", a.C, a.bf16 ? 64 : 32);
    LBC_REQUIRE(!a.bf16 || wmajor, "igemm: the bf16 path needs depth-contiguous weights (transpose first)");
    LBC_REQUIRE(!a.act_bf16 || a.bf16, "igemm: bf16 activations need bf16 = 1");
    LBC_REQUIRE(!a.w_bf16 || a.act_bf16, "igemm: bf16 weight copies are used with bf16 activations only");
    LBC_REQUIRE(cfg >= kLbcCfgGlds || a.K % kCfgBN[cfg] == 0, "igemm: output channels %d not a multiple of the tile", a.K);
    LBC_REQUIRE(a.KH * a.KW <= 16, "igemm: too many taps");
    LBC_REQUIRE(a.S == 1 || a.S == 2, "igemm: stride %d unsupported", a.S);
    LBC_REQUIRE(a.M > 0, "igemm: empty launch");
    LBC_REQUIRE((long long)a.N * a.H * a.W * a.C < (1ll << 31) && (long long)a.N * a.OH * a.OW * a.K < (1ll << 31),
                "igemm: tensor exceeds 2^31 elements");
    // algorithmic work: 2*M*K*C per valid tap; bytes: gathered tensor + output once, weights once
    LBC_REQUIRE(a.nphase == 0 || a.nphase == 1 || (a.nphase == 4 && mode == 1 && a.S == 2 && a.ostep == 2), "igemm: bad nphase %d", a.nphase);
    double taps = 0;     // per output row; with nphase = 4 every tap serves exactly one of the four phases: average = T / 4
    for (int t = 0; t < a.KH * a.KW; ++t) {
        const int r = t / a.KW, q = t - r * a.KW;
        if (a.nphase == 4) { taps += 0.25; continue; }
        if (mode == 1 && a.S == 2 && ((((a.oy0 + a.P - r) & 1) != 0) || (((a.ox0 + a.P - q) & 1) != 0))) continue;
        taps += 1;
    }
    const double nph = a.nphase == 4 ? 4.0 : 1.0;
    // bytes: the gathered tensor (a quarter of it per phase of a stride-2 transposed launch), the output (+ residual), the weights of the taps used
    const double in_elems = (double)a.N * a.H * a.W * a.C * ((mode == 1 && a.S == 2) ? nph / 4.0 : 1.0);
    // profile class = kernel family + GEMM orientation
    const bool halo = cfg < kLbcCfgGlds && wmajor && lbc_conv3x3_halo_eligible(a, mode);
    const bool ksplit = cfg >= kLbcCfgHdma && lbc_conv_hdmap_nsplit(a, mode, cfg) > 1;     // two launches (partial tiles, epilogue) in one bracket
    const char* pname = ksplit ? (mode == 0 ? "conv_hdma_gather_split" : "conv_hdma_transposed_split")
                        : cfg >= kLbcCfgHdma ? (mode == 0 ? "conv_hdma_gather" : "conv_hdma_transposed")
                        : cfg >= kLbcCfgGlds ? (mode == 0 ? "conv_glds_gather" : "conv_glds_transposed")
                        : halo ? (mode == 0 ? "conv_halo_gather" : "conv_halo_transposed")
                               : (mode == 0 ? "conv_igemm_gather" : "conv_igemm_transposed");
    LbcProfScope prof(pname, 2.0 * a.M * nph * a.K * (double)a.C * taps,
                      // (+ the side tensors of the fused BatchNorm-backward reduce: the pre-BatchNorm activation, and in the tensor-masked form the ReLU output)
                      (a.act_bf16 ? 2.0 : 4.0) * (in_elems + nph * (double)a.M * a.K * (1 + (a.resid ? 1 : 0) + (a.bnb_y ? 1 : 0) + (a.bnb_mask ? 1 : 0))) +
                          (a.w_bf16 ? 2.0 : 4.0) * taps * nph * a.C * a.K, s);
    // (round 4's definition of a launch's algorithmic bytes did not count those side tensors: booked separately so that bench.py reports both ratios)
    if (lbc_prof_on() && (a.bnb_y || a.bnb_mask))
        lbc_prof_note("side_tensors_of_fused_reduce", (a.act_bf16 ? 2.0 : 4.0) * nph * (double)a.M * a.K * ((a.bnb_y ? 1 : 0) + (a.bnb_mask ? 1 : 0)));
    if (cfg >= kLbcCfgHdma) {
        LBC_REQUIRE(wmajor && lbc_conv_hdma_pick(a, mode) >= 0, "igemm: launch not eligible for the halo-staged LDS-DMA kernel");
        return lbc_conv_hdma_launch(a, mode, cfg, s);
    }
    if (cfg >= kLbcCfgGlds) {
        LBC_REQUIRE(wmajor && lbc_conv_glds_pick(a, mode) >= 0, "igemm: launch not eligible for the 8-wave LDS-DMA kernel");
        return lbc_conv_glds_launch(a, mode, cfg, s);
    }
    if (wmajor && lbc_conv3x3_halo_eligible(a, mode)) return lbc_conv3x3_halo_launch(a, mode, s);
    switch (cfg) {
        case 0: return launch_cfg<128, 64>(a, wmajor, mode, s);
        case 1: return launch_cfg<128, 128>(a, wmajor, mode, s);
        default: return launch_cfg<64, 64>(a, wmajor, mode, s);
    }
}
