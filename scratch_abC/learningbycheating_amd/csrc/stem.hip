// ResNet stem for gfx950: input preparation, the 7x7/2 convolution (Cin = 3 RGB or
// 7 bird-view channels -> 64) and its weight gradient, on the exact-f32 MFMA.
// reference: bird_view/models/resnet.py:102-103,148 (conv1), common.py:101-109
// (NormalizeV2), image.py:71.
//
// Data layout trick: the image is held NHWC with a 3-pixel zero border
// (xp[N][H+6][W+6][Cin]), so one filter ROW of one output pixel is 7*Cin
// *contiguous* floats starting at xp[n][2oy+r][2ox][0] and needs no bounds checks.
// The implicit GEMM therefore walks depth as 7 chunks (r = 0..6) of L = 7*Cin
// channels (padded to a multiple of 8 with zeros in LDS); the weight tensor in
// channels_last order [64][7][7][Cin] has exactly the same [r][L] structure.
// There is no input gradient (the image is a leaf).
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include "lbc_kernels.hpp"
#include <type_traits>

namespace {

// XT = element type of the padded image: float, or __bf16 in the bf16 modes (the stem rounds it to bf16 anyway)
template <typename XT>
__global__ __launch_bounds__(256) void prep_input_k(const float* __restrict__ img, XT* __restrict__ xp, int N, int C,
                                                    int H, int W, NormConst nc)
{
    // one thread per pixel of the PADDED image: the 3-pixel border is written here as zeros (no separate memset pass; a
    // forward captured into a hipGraph then consists of kernel nodes only)
    const int Hp = H + 6, Wp = W + 6;
    const long long total = (long long)N * Hp * Wp;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int xq = (int)(i % Wp);
        const long long t = i / Wp;
        const int yq = (int)(t % Hp);
        const int n = (int)(t / Hp);
        const int x = xq - 3, y = yq - 3;
        const bool in = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;
        XT* dst = xp + (size_t)i * C;
        for (int c = 0; c < C; ++c) {
            float v = 0.f;
            if (in) {
                v = img[((size_t)(n * C + c) * H + y) * W + x];
                if (nc.enabled) v = (v - nc.mean[c]) / nc.stdv[c];
            }
            dst[c] = (XT)v;
        }
    }
}

// uint8 NHWC frames (what the LMDB dataset stores: reference bird_view/utils/datasets/image_lmdb.py:128-222 decodes them to
// f32 CHW on the host): /255, ImageNet normalisation and the zero border in one pass, 4x fewer input bytes
// CIN = 3 / 7 (the two networks) unrolls the channel loop with every byte load in front of the first use (with the channel count
// a runtime value each 1-byte load was followed by its own s_waitcnt vmcnt(0)); CIN = 0: any C <= 8.  IDX = unsigned when the
// padded pixel index fits 32 bits.
template <typename XT, int CIN, typename IDX>
__global__ __launch_bounds__(256) void prep_input_u8_k(const unsigned char* __restrict__ img, XT* __restrict__ xp, int N, int C,
                                                       int H, int W, NormConst nc)
{
    const int Hp = H + 6, Wp = W + 6;
    const int Cc = CIN ? CIN : C;
    const IDX total = (IDX)((long long)N * Hp * Wp);
    const IDX stride = (IDX)gridDim.x * (IDX)blockDim.x;
    for (IDX i = (IDX)blockIdx.x * (IDX)blockDim.x + (IDX)threadIdx.x; i < total; i += stride) {
        const int xq = (int)(i % (IDX)Wp);
        const IDX t = i / (IDX)Wp;
        const int yq = (int)(t % (IDX)Hp);
        const int n = (int)(t / (IDX)Hp);
        const int x = xq - 3, y = yq - 3;
        const bool in = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;
        XT* dst = xp + (size_t)i * Cc;
        const unsigned char* src = img + ((size_t)(n * H + (in ? y : 0)) * W + (size_t)(in ? x : 0)) * Cc;
        unsigned char b[CIN ? CIN : 8];
#pragma unroll
        for (int c = 0; c < (CIN ? CIN : 8); ++c) b[c] = c < Cc ? src[c] : (unsigned char)0;
#pragma unroll
        for (int c = 0; c < (CIN ? CIN : 8); ++c) {
            if (c >= Cc) break;
            // selects instead of branches: the loads above stay in front, border lanes (clamped address) discard what they read
            const float u = (float)b[c] / 255.0f;                  // torchvision ToTensor
            const float w = (u - nc.mean[c]) / nc.stdv[c];         // (discarded, possibly inf / nan, when normalisation is off)
            const float v = nc.enabled ? w : u;
            dst[c] = (XT)(in ? v : 0.f);
        }
    }
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// two consecutive elements of the padded image (even offsets: 8- or 4-byte aligned)
__device__ __forceinline__ f32x2_t load2(const float* p) { return *reinterpret_cast<const f32x2_t*>(p); }
__device__ __forceinline__ f32x2_t load2(const __bf16* p) { return __builtin_convertvector(*reinterpret_cast<const bf16x2_t*>(p), f32x2_t); }

template <int CIN, typename T>
__global__ __launch_bounds__(256) void stem_fwd_k(StemArgs a)
{
    constexpr int L = 7 * CIN;
    constexpr int L8 = (L + 7) / 8 * 8;
    constexpr int LD = L8 + 4;
    constexpr int BM = 128, BN = 64, MT = 2;
    constexpr int HALF = L8 / 2;          // floats per thread per row (12 or 28): 2 threads per pixel row
    constexpr int NB = BN * L8 / 256;     // weight floats per thread per chunk (6 or 14)
    __shared__ __attribute__((aligned(16))) float sA[2][BM * LD];
    __shared__ __attribute__((aligned(16))) float sB[2][BN * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int OH = a.H / 2, OW = a.W / 2;
    const int Hp = a.H + 6, Wp = a.W + 6;
    const int M = a.N * OH * OW;
    const int m0 = blockIdx.x * BM;

    // this thread stages columns [half*HALF, half*HALF + HALF) of pixel row `arow` (8-byte aligned: every offset is even)
    const int arow = tid >> 1, half = tid & 1;
    long long abase = -1;
    {
        const int m = m0 + arow;
        if (m < M) {
            const int n = m / (OH * OW);
            const int rem = m - n * OH * OW;
            const int oy = rem / OW, ox = rem - oy * OW;
            abase = (long long)((n * Hp + 2 * oy) * Wp + 2 * ox) * CIN + half * HALF;
        }
    }

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    float2 ra[HALF / 2];
    float rb[NB];
    auto load_chunk = [&](int r) {
#pragma unroll
        for (int q = 0; q < HALF / 2; ++q) {
            // unconditional load (rows past M read pixel 0); masking is deferred to store_chunk()
            ra[q] = *reinterpret_cast<const float2*>(static_cast<const float*>(a.xp) + (abase >= 0 ? abase : half * HALF) + (long long)(r * Wp * CIN + 2 * q));
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int idx = tid + 256 * q;
            const int row = idx / L8, j = idx - row * L8;
            rb[q] = j < L ? a.w[(size_t)row * (7 * L) + (size_t)(r * L + j)] : 0.f;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int q = 0; q < HALF / 2; ++q) {
            float2 v = ra[q];
            const int j = half * HALF + 2 * q;     // columns >= L are padding: must be exact zeros
            if (j >= L || abase < 0) v.x = 0.f;
            if (j + 1 >= L || abase < 0) v.y = 0.f;
            *reinterpret_cast<float2*>(&sA[buf][arow * LD + half * HALF + 2 * q]) = v;
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int idx = tid + 256 * q;
            const int row = idx / L8, j = idx - row * L8;
            sB[buf][row * LD + j] = rb[q];
        }
    };

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int r = 0; r < 7; ++r) {
        const int buf = r & 1;
        if (r + 1 < 7) load_chunk(r + 1);
#pragma unroll
        for (int g = 0; g < L8 / 8; ++g) {
            f32x4 af[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(&sA[buf][((wm * MT + i) * 32 + l31) * LD + g * 8 + kh * 4]);
            const f32x4 bf = *reinterpret_cast<const f32x4*>(&sB[buf][(wn * 32 + l31) * LD + g * 8 + kh * 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][i], bf[i], acc[mi], 0, 0, 0);
        }
        if (r + 1 < 7) store_chunk(buf ^ 1);
        __syncthreads();
    }

    float s1 = 0.f, s2 = 0.f;
    const int col = wn * 32 + l31;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (wm * MT + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            const int m = m0 + row;
            if (m < M) {
                const float v = acc[mi][e];
                Act<T>::st1(static_cast<T*>(a.y) + (size_t)m * BN + col, v);
                s1 += v; s2 += v * v;
            }
        }
    if (a.stats) {
        float* red = &sA[0][0];   // [2 wm][2][64]
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (kh == 0) { red[(wm * 2 + 0) * BN + col] = s1; red[(wm * 2 + 1) * BN + col] = s2; }
        __syncthreads();
        if (tid < BN) {
            float* dst = a.stats + (size_t)blockIdx.x * 2 * BN;
            dst[tid] = red[tid] + red[2 * BN + tid];
            dst[BN + tid] = red[BN + tid] + red[3 * BN + tid];
        }
    }
}

// dW[co][r][j] = sum_m dy[m][co] * xp_row(m, r)[j]; grid (split, r)
template <int CIN, typename T>
__global__ __launch_bounds__(256) void stem_wgrad_k(StemWgradArgs a, int rows_per_split)
{
    constexpr int L = 7 * CIN;
    constexpr int LQ = (L + 31) / 32 * 32;   // 32 or 64
    constexpr int QT = LQ / 32;              // q tiles (1 or 2)
    constexpr int KS = 2 / QT;               // with one q tile the two wave pairs split the 32-pixel depth of a chunk
    constexpr int BR = 32;
    constexpr int LDP = 64 + 4, LDQ = LQ + 4;
    constexpr int QPT = LQ / 8;              // floats per thread per row (4 or 8): 8 threads per pixel row
    __shared__ __attribute__((aligned(16))) float sP[2][BR * LDP];
    __shared__ __attribute__((aligned(16))) float sQ[2][BR * LDQ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int pt = wave & 1;                       // which 32 output channels
    const int qt = QT == 2 ? (wave >> 1) : 0;      // which 32 filter-row columns
    const int ks = QT == 2 ? 0 : (wave >> 1);      // which half of the chunk's pixels
    const int OH = a.H / 2, OW = a.W / 2;
    const int Hp = a.H + 6, Wp = a.W + 6;
    const int M = a.N * OH * OW;
    // (split, filter row) from a 1-D grid: the seven filter-row workgroups of one pixel range re-read the same dy rows and image
    // rows; consecutive block ids put them on the chip at the same time and -- with block b on XCD b % 8 (observed, speed only) --
    // in the same XCD's L2, so dy leaves HBM once instead of seven times (measured before: 3.4 GB fetched for 0.6 GB of operands)
    int split, r;
    {
        const int b = blockIdx.x, ns = (int)gridDim.x / 7;
        if ((ns & 7) == 0) { const int t = b >> 3; r = t % 7; split = (t / 7) * 8 + (b & 7); }
        else { split = b / 7; r = b - split * 7; }
    }
    const int mbeg = split * rows_per_split;
    const int mend = mbeg + rows_per_split < M ? mbeg + rows_per_split : M;
    const int nchunk = mend > mbeg ? (mend - mbeg + BR - 1) / BR : 0;

    // staging roles
    const int qrow = tid >> 3, qseg = tid & 7;     // Q: pixel row of the chunk, 8 column segments
    const int prow0 = tid >> 4, pseg = tid & 15;   // P: rows prow0 and prow0 + 16, 16 float4 segments
    // pixel coordinates of Q row (mbeg + qrow), advanced by BR per chunk without divisions
    int qn, qy, qx;
    {
        const int m = mbeg + qrow;
        qn = m / (OH * OW);
        const int rem = m - qn * OH * OW;
        qy = rem / OW; qx = rem - qy * OW;
    }

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    f32x4 rp[2];
    float rq[QPT];
    auto load_chunk = [&](int ch) {
        const int mc = mbeg + ch * BR;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = mc + prow0 + 16 * j;
            rp[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m < mend) rp[j] = Act<T>::ld4(static_cast<const T*>(a.dy) + (size_t)m * 64 + (size_t)(pseg * 4));
        }
        const bool ok = (mc + qrow) < mend;
        const long long base = (long long)((qn * Hp + 2 * qy + r) * Wp + 2 * qx) * CIN + qseg * QPT;
#pragma unroll
        for (int q = 0; q < QPT; q += 2) {
            float2 v = make_float2(0.f, 0.f);
            const int j = qseg * QPT + q;
            if (ok && j < L) v = *reinterpret_cast<const float2*>(static_cast<const float*>(a.xp) + base + q);   // 8-byte aligned (all offsets even)
            if (j + 1 >= L) v.y = 0.f;
            rq[q] = v.x; rq[q + 1] = v.y;
        }
        qx += BR;
        while (qx >= OW) { qx -= OW; ++qy; }
        while (qy >= OH) { qy -= OH; ++qn; }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(&sP[buf][(prow0 + 16 * j) * LDP + pseg * 4]) = rp[j];
#pragma unroll
        for (int q = 0; q < QPT; q += 2)
            *reinterpret_cast<float2*>(&sQ[buf][qrow * LDQ + qseg * QPT + q]) = make_float2(rq[q], rq[q + 1]);
    };

    if (nchunk > 0) { load_chunk(0); store_chunk(0); }
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < nchunk;
        if (more) load_chunk(ch + 1);
#pragma unroll
        for (int st = 0; st < BR / 2 / KS; ++st) {
            const int k = 2 * (st + ks * (BR / 2 / KS)) + kh;
            const float af = sP[buf][k * LDP + pt * 32 + l31];
            const float bf = sQ[buf][k * LDQ + qt * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }
    if (KS == 2) {
        // combine the two depth halves: waves 2,3 hand their accumulators to waves 0,1 through LDS
        float* red = &sP[0][0];   // [2 pt][16][64]
        if (ks == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(pt * 16 + e) * 64 + lane] = acc[e];
        }
        __syncthreads();
        if (ks == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += red[(pt * 16 + e) * 64 + lane];
        }
    }
    if (ks == 0) {
        float* out = a.partial + (size_t)split * 64 * 7 * L;
        const int j = qt * 32 + l31;
        if (j < L) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = pt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                out[(size_t)co * (7 * L) + (size_t)(r * L + j)] = acc[e];
            }
        }
    }
}

// bf16-MFMA weight gradient of the stem (precision >= 1): contraction over pixels on v_mfma_f32_32x32x16_bf16, both
// operands transposed in registers into [channel][64 pixels] LDS tiles exactly like conv_wgrad_bf16_k (4 pixels x 4
// channels per thread, 8-byte column writes, conflict-aware lane order).  grid (split, filter row r).
template <int CIN, typename T, typename XT>
__global__ __launch_bounds__(256) void stem_wgrad_bf16_k(StemWgradArgs a, int rows_per_split)
{
    const XT* xpad = static_cast<const XT*>(a.xp);
    constexpr bool ABF = Act<T>::kBf16;
    using preg_t = typename std::conditional<ABF, bf16x4, f32x4>::type;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int L = 7 * CIN;
    constexpr int LQ = (L + 31) / 32 * 32;   // 32 or 64 filter-row columns (zero padded)
    constexpr int QT = LQ / 32;              // column tiles (1 or 2)
    constexpr int KS = 2 / QT;               // with one column tile the two wave pairs split the 64-pixel depth of a chunk
    constexpr int BRH = 64, LD = BRH + 8;
    constexpr int CGQ = LQ / 4;              // 8 or 16 column groups
    constexpr int TQ_ = 16 * CGQ;            // Q micro-tiles per chunk (128 or 256)
    constexpr int KB = 2;
    __shared__ __attribute__((aligned(16))) __bf16 sP[2][64 * LD];
    __shared__ __attribute__((aligned(16))) __bf16 sQ[2][LQ * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int pt = wave & 1;                       // which 32 output channels
    const int qt = QT == 2 ? (wave >> 1) : 0;      // which 32 filter-row columns
    const int ks = QT == 2 ? 0 : (wave >> 1);      // which half of the chunk's pixels
    const int OH = a.H / 2, OW = a.W / 2;
    const int Hp = a.H + 6, Wp = a.W + 6;
    const int M = a.N * OH * OW;
    // (split, filter row) from a 1-D grid: the seven filter-row workgroups of one pixel range re-read the same dy rows and image
    // rows; consecutive block ids put them on the chip at the same time and -- with block b on XCD b % 8 (observed, speed only) --
    // in the same XCD's L2, so dy leaves HBM once instead of seven times (measured before: 3.4 GB fetched for 0.6 GB of operands)
    int split, r;
    {
        const int b = blockIdx.x, ns = (int)gridDim.x / 7;
        if ((ns & 7) == 0) { const int t = b >> 3; r = t % 7; split = (t / 7) * 8 + (b & 7); }
        else { split = b / 7; r = b - split * 7; }
    }
    const int mbeg = split * rows_per_split;
    const int mend = mbeg + rows_per_split < M ? mbeg + rows_per_split : M;
    const int nchunk = mend > mbeg ? (mend - mbeg + BRH - 1) / BRH : 0;
    const T* dy = static_cast<const T*>(a.dy);

    int pcg, ppg, qcg, qpg;
    wgrad_tile_coord<16, KB>(tid, pcg, ppg);                 // P: 16 channel groups x 16 pixel groups = 256 micro-tiles
    wgrad_tile_coord<CGQ, KB>(tid, qcg, qpg);
    const bool qactive = tid < TQ_;
    // coordinates of the first pixel of this thread's Q micro-tile, advanced by 64 per chunk without divisions
    int qn, qy, qx;
    {
        const int m = mbeg + 4 * qpg;
        qn = m / (OH * OW);
        const int rem = m - qn * OH * OW;
        qy = rem / OW; qx = rem - qy * OW;
    }

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    preg_t rp[4];
    f32x2 rq[4][2];
    bool pok[4], qok[4];
    for (int ch = -1; ch < nchunk; ++ch) {
        const bool more = ch + 1 < nchunk;
        if (more) {
            const int mc = mbeg + (ch + 1) * BRH;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mc + 4 * ppg + i;
                pok[i] = m < mend;
                rp[i] = *reinterpret_cast<const preg_t*>(dy + (size_t)(pok[i] ? m : 0) * 64 + (size_t)(pcg * 4));
            }
            int n = qn, y = qy, x = qx;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mc + 4 * qpg + i;
                qok[i] = (m < mend) && qactive;
                const long long base = qok[i] ? (long long)((n * Hp + 2 * y + r) * Wp + 2 * x) * CIN + qcg * 4 : 0;
                rq[i][0] = load2(xpad + base);        // every offset is even
                rq[i][1] = load2(xpad + base + 2);
                if (++x >= OW) { x = 0; if (++y >= OH) { y = 0; ++n; } }
            }
            qx += BRH;
            while (qx >= OW) { qx -= OW; ++qy; }
            while (qy >= OH) { qy -= OH; ++qn; }
        }
        if (ch >= 0) {
            const int buf = ch & 1;
#pragma unroll
            for (int g = 0; g < BRH / 16 / KS; ++g) {
                const int gg = g + ks * (BRH / 16 / KS);
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(&sP[buf][(pt * 32 + l31) * LD + gg * 16 + kh * 8]);
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(&sQ[buf][(qt * 32 + l31) * LD + gg * 16 + kh * 8]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
            }
        }
        if (more) {
            const int buf = (ch + 1) & 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 col;
#pragma unroll
                for (int i = 0; i < 4; ++i) col[i] = pok[i] ? (float)rp[i][c] : 0.f;
                *reinterpret_cast<bf16x4*>(&sP[buf][(pcg * 4 + c) * LD + ppg * 4]) = __builtin_convertvector(col, bf16x4);
            }
            if (qactive) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 col;
#pragma unroll
                    for (int i = 0; i < 4; ++i) col[i] = (qok[i] && (qcg * 4 + c) < L) ? rq[i][c >> 1][c & 1] : 0.f;   // pad columns: exact zeros
                    *reinterpret_cast<bf16x4*>(&sQ[buf][(qcg * 4 + c) * LD + qpg * 4]) = __builtin_convertvector(col, bf16x4);
                }
            }
        }
        __syncthreads();
    }
    if (KS == 2) {
        // combine the two depth halves: waves 2,3 hand their accumulators to waves 0,1 through LDS
        float* red = reinterpret_cast<float*>(&sP[0][0]);   // [2 pt][16][64] floats = 8 KB (sP holds 18 KB)
        if (ks == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(pt * 16 + e) * 64 + lane] = acc[e];
        }
        __syncthreads();
        if (ks == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += red[(pt * 16 + e) * 64 + lane];
        }
    }
    if (ks == 0) {
        float* out = a.partial + (size_t)split * 64 * 7 * L;
        const int j = qt * 32 + l31;
        if (j < L) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = pt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                out[(size_t)co * (7 * L) + (size_t)(r * L + j)] = acc[e];
            }
        }
    }
}

// bf16 weight gradient of the RGB stem, ALL SEVEN filter rows per workgroup.  The kernel above gives every filter row its own
// workgroup, so the seven workgroups of a pixel range each stage the same 64 x 64 dy tile (7 x 503 MB through L2 at batch 256,
// two MFMAs per staged tile and wave: 664 us for 74 GFLOP).  Here a 64-pixel chunk stages dy once and the seven 21-column image
// tiles next to it (one [24 columns][64 pixels] LDS tile per filter row; columns 21..23 and the fragment rows past them are
// never-used padding: a garbage B column only reaches an output column that is not stored), 14 MFMAs per wave and chunk.  The two
// wave pairs split the 64-pixel depth of a chunk and write separate partial slabs (slab = 2 * workgroup + depth half).
template <typename T>
__global__ __launch_bounds__(256, 2) void stem_wgrad_rows_k(StemWgradArgs a, int rows_per_split)
{
    constexpr int CIN = 3, L = 21, QR = 24, BRH = 64, LD = BRH + 8, KB = 2;
    constexpr bool ABF = Act<T>::kBf16;
    using preg_t = typename std::conditional<ABF, bf16x4, f32x4>::type;
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) __bf16 sP[2][64 * LD];
    __shared__ __attribute__((aligned(16))) __bf16 sQ[2][(7 * QR + 8) * LD];      // + 8 rows: the last tile's fragment reads stay inside

    const __bf16* xpad = static_cast<const __bf16*>(a.xp);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int pt = wave & 1;                       // which 32 output channels
    const int ks = wave >> 1;                      // which half of the chunk's pixels
    const int OH = a.H / 2, OW = a.W / 2;
    const int Hp = a.H + 6, Wp = a.W + 6;
    const int M = a.N * OH * OW;
    const int split = blockIdx.x;
    const int mbeg = split * rows_per_split;
    const int mend = mbeg + rows_per_split < M ? mbeg + rows_per_split : M;
    const int nchunk = mend > mbeg ? (mend - mbeg + BRH - 1) / BRH : 0;
    const T* dy = static_cast<const T*>(a.dy);

    int pcg, ppg;
    wgrad_tile_coord<16, KB>(tid, pcg, ppg);       // P: 16 channel groups x 16 pixel groups = 256 micro-tiles of 4 x 4
    // Q: 7 rows x 6 column groups x 16 pixel groups = 672 micro-tiles; thread t takes t, t + 256, t + 512.  Within 16 lanes the
    // pixel group takes 8 values and the column group 2: the 8-byte column writes of a lane group hit 16 distinct bank pairs.
    int qr[3], qcg[3], qpg[3], qn[3], qy[3], qx[3];
    bool qjob[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int job = tid + 256 * k;
        qjob[k] = job < 7 * 96;
        const int r = job / 96, u = job - r * 96;
        const int lo = u & 15, hi = u >> 4;
        qr[k] = qjob[k] ? r : 0;
        qpg[k] = (lo & 7) + 8 * (hi & 1);
        qcg[k] = (lo >> 3) + 2 * (hi >> 1);
        const int m = mbeg + 4 * qpg[k];
        qn[k] = m / (OH * OW);
        const int rem = m - qn[k] * OH * OW;
        qy[k] = rem / OW; qx[k] = rem - qy[k] * OW;
    }

    f32x16 acc[7];
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    // fused BatchNorm-backward apply (StemWgradArgs::bn_*): this thread's four channels
    const T* bny = static_cast<const T*>(a.bn_y);
    f32x4 cA = {1.f, 1.f, 1.f, 1.f}, cB = {0.f, 0.f, 0.f, 0.f}, cD = cB, cM = cB, cI = cB;
    if (bny) {
        cA = *reinterpret_cast<const f32x4*>(a.bn_coefA + pcg * 4); cB = *reinterpret_cast<const f32x4*>(a.bn_coefB + pcg * 4);
        cD = *reinterpret_cast<const f32x4*>(a.bn_coefD + pcg * 4); cM = *reinterpret_cast<const f32x4*>(a.bn_mean + pcg * 4);
        cI = *reinterpret_cast<const f32x4*>(a.bn_invstd + pcg * 4);
    }
    preg_t rp[4], ry[4];
    unsigned rq[3][4][2];
    bool pok[4], qok[3][4];
    for (int ch = -1; ch < nchunk; ++ch) {
        const bool more = ch + 1 < nchunk;
        if (more) {
            const int mc = mbeg + (ch + 1) * BRH;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mc + 4 * ppg + i;
                pok[i] = m < mend;
                rp[i] = *reinterpret_cast<const preg_t*>(dy + (size_t)(pok[i] ? m : 0) * 64 + (size_t)(pcg * 4));
                if (bny) ry[i] = *reinterpret_cast<const preg_t*>(bny + (size_t)(pok[i] ? m : 0) * 64 + (size_t)(pcg * 4));
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int n = qn[k], y = qy[k], x = qx[k];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = mc + 4 * qpg[k] + i;
                    qok[k][i] = (m < mend) && qjob[k];
                    const long long base = qok[k][i] ? (long long)((n * Hp + 2 * y + qr[k]) * Wp + 2 * x) * CIN + qcg[k] * 4 : 0;
                    rq[k][i][0] = *reinterpret_cast<const unsigned*>(xpad + base);          // every offset is even
                    rq[k][i][1] = *reinterpret_cast<const unsigned*>(xpad + base + 2);
                    if (++x >= OW) { x = 0; if (++y >= OH) { y = 0; ++n; } }
                }
                qx[k] += BRH;
                while (qx[k] >= OW) { qx[k] -= OW; ++qy[k]; }
                while (qy[k] >= OH) { qy[k] -= OH; ++qn[k]; }
            }
        }
        if (ch >= 0) {
            const int buf = ch & 1;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int gg = g + ks * 2;
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(&sP[buf][(pt * 32 + l31) * LD + gg * 16 + kh * 8]);
#pragma unroll
                for (int r = 0; r < 7; ++r) {
                    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(&sQ[buf][(r * QR + l31) * LD + gg * 16 + kh * 8]);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[r], 0, 0, 0);
                }
            }
        }
        if (more) {
            const int buf = (ch + 1) & 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 col;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float g = (float)rp[i][c];
                    if (bny) g = cA[c] * (g - cB[c] - ((float)ry[i][c] - cM[c]) * cI[c] * cD[c]);     // as bn_bwd_apply_k
                    col[i] = pok[i] ? g : 0.f;
                }
                *reinterpret_cast<bf16x4*>(&sP[buf][(pcg * 4 + c) * LD + ppg * 4]) = __builtin_convertvector(col, bf16x4);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (!qjob[k]) continue;
                unsigned v[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i][0] = qok[k][i] ? rq[k][i][0] : 0u; v[i][1] = qok[k][i] ? rq[k][i][1] : 0u; }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // column c of the micro-tile = bf16 half (c & 1) of dword (c >> 1) of each of the four pixels
                    u32x2 w;
                    if (c & 1) {
                        w[0] = (v[0][c >> 1] >> 16) | (v[1][c >> 1] & 0xffff0000u);
                        w[1] = (v[2][c >> 1] >> 16) | (v[3][c >> 1] & 0xffff0000u);
                    } else {
                        w[0] = (v[0][c >> 1] & 0xffffu) | (v[1][c >> 1] << 16);
                        w[1] = (v[2][c >> 1] & 0xffffu) | (v[3][c >> 1] << 16);
                    }
                    *reinterpret_cast<u32x2*>(&sQ[buf][(qr[k] * QR + qcg[k] * 4 + c) * LD + qpg[k] * 4]) = w;
                }
            }
        }
        __syncthreads();
    }
    float* out = a.partial + (size_t)(split * 2 + ks) * 64 * 7 * L;
    if (l31 < L) {
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = pt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                out[(size_t)co * (7 * L) + (size_t)(r * L + l31)] = acc[r][e];
            }
    }
}

// bf16 stem forward, second generation (bf16 padded image).  The kernel above walks the seven filter rows as seven staged
// chunks per 128-pixel tile: per chunk every thread issues 6 - 14 four-byte loads, converts, writes LDS, and the weights
// are re-staged from f32 for every tile -- 343 us (RGB) / 466 us (7-channel bird view) at batch 256 for 0.6 / 0.4 GB of
// traffic.  Here
//   * a tile is 64 output pixels of ONE output row: its input is a band of 7 image rows x (2 * 64 + 5) pixels, i.e. seven
//     contiguous byte ranges, copied raw (bf16, 4-byte loads, coalesced) into LDS -- no conversion, no transposition;
//   * the A fragment of (filter row r, depth step g) for pixel p is the 8 consecutive bf16 at element 2 p Cin + 16 g + 8 kh
//     of band row r: four ds_read_b32 (pixel stride 3 or 7 dwords: conflict-free); the k columns past the 7 Cin real ones read
//     the neighbouring pixels (finite values) against zero weights;
//   * the weights live in registers as bf16 B fragments for the whole (persistent) workgroup: 56 / 112 VGPRs;
//   * the next tile's band is prefetched into registers while the current one is multiplied; the output tile goes through
//     LDS to 16-byte stores; BatchNorm partial sums accumulate in registers across the workgroup's tiles (one row per workgroup).
template <int CIN, typename T>
__global__ __launch_bounds__(256, 2) void stem_fwd_rows_k(StemArgs a, int ntiles, int tiles_x)
{
    constexpr int L = 7 * CIN;
    constexpr int KG = (L + 15) / 16;                          // depth steps per filter row (2 or 4)
    constexpr int BX = 64;                                     // output pixels per tile
    constexpr int BL = (2 * (BX - 1) + 7) * CIN;               // band elements per image row actually needed
    constexpr int BD = (BL + 1) / 2;                           // ... in dwords
    constexpr int RS = ((2 * (BX - 1) * CIN + KG * 16) * 2 + 15) / 16 * 16;   // LDS bytes per band row incl. the over-read of the last pixel
    constexpr int NLD = (7 * BD + 255) / 256;                  // band dwords per thread
    constexpr int OLD = 64 + (sizeof(T) == 2 ? 8 : 4);         // staged output row (elements): 64 channels + pad
    __shared__ __attribute__((aligned(16))) char sBand[2][7 * RS];
    __shared__ __attribute__((aligned(16))) T sOut[BX * OLD];
    __shared__ float sRed[2][2][64];

    const __bf16* xpad = static_cast<const __bf16*>(a.xp);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;                   // pixel half, channel half
    const int OH = a.H / 2, OW = a.W / 2;
    const int Hp = a.H + 6, Wp = a.W + 6;
    const long long total_elems = (long long)a.N * Hp * Wp * CIN;

    // the band rows' tails (elements BL .. RS/2) are only ever read against zero weights: zero them once so they stay finite
    for (int i = tid; i < 2 * 7 * RS / 4; i += 256) reinterpret_cast<unsigned*>(&sBand[0][0])[i] = 0u;

    // B fragments: output channel 32 wn + l31, k-slot i of step (r, g) = filter column 16 g + 8 kh + i (zero past 7 Cin)
    bf16x8 bw[7][KG];
    {
        const float* wrow = a.w + (size_t)(32 * wn + l31) * (49 * CIN);
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int g = 0; g < KG; ++g)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = 16 * g + 8 * kh + i;
                    bw[r][g][i] = (__bf16)(k < L ? wrow[r * L + (k < L ? k : 0)] : 0.f);
                }
    }
    // band copy roles
    int br[NLD], bj[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int idx = tid + 256 * q;
        br[q] = idx / BD;
        bj[q] = idx - br[q] * BD;
    }
    // (round 6) the bands of the NEXT TWO tiles are in flight while one is multiplied: a tile is 14 / 28 MFMAs per wave (~0.3 - 0.5 us), one band
    // ahead left every workgroup waiting a full HBM latency per tile (2.8 / 3.8 us per tile, 3.0 TB/s, profiles/r05_final_*)
    unsigned regs[NLD], regs2[NLD];
    auto band_load = [&](int tile, unsigned (&dst)[NLD]) {
        const int xt = tile % tiles_x;
        const int t2 = tile / tiles_x;
        const int oy = t2 % OH, n = t2 / OH;
        const long long e0 = ((long long)(n * Hp + 2 * oy) * Wp + 2 * xt * BX) * CIN;
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const long long e = e0 + (long long)br[q] * Wp * CIN + 2 * bj[q];
            const bool ok = br[q] < 7 && e + 1 < total_elems;      // past the tensor only behind the last tile's last pixels
            dst[q] = ok ? *reinterpret_cast<const unsigned*>(xpad + e) : 0u;
        }
    };
    auto band_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NLD; ++q)
            if (br[q] < 7) *reinterpret_cast<unsigned*>(&sBand[buf][br[q] * RS + bj[q] * 4]) = regs[q];
    };

    float s1 = 0.f, s2 = 0.f;
    // A workgroup walks a CONTIGUOUS range of tiles (along an output row, then down the rows of an image), and consecutive ranges sit on
    // ONE XCD (workgroup ids go round-robin over the 8 XCDs, each with an L2 of its own): an input row serves 3.5 output rows, and with
    // the grid-stride walk of rounds 2-4 those were tiles of 3-4 different workgroups on different XCDs -- PMC of round 5's evidence
    // call: 387 MB fetched for the 99 MB RGB image, 685 MB for the 140 MB bird view (profiles/r05_final_pmc_summary_bf16.txt)
    int tile, tend, it = 0;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, rr = nwg & 7;
        const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
        const int per = (ntiles + nwg - 1) / nwg;
        tile = logical * per;
        tend = tile + per < ntiles ? tile + per : ntiles;
    }
    if (tile < tend) band_load(tile, regs);
    __syncthreads();                                           // the zero fill above is complete
    if (tile < tend) band_store(0);
    if (tile + 1 < tend) band_load(tile + 1, regs);
    __syncthreads();
    for (; tile < tend; ++tile, ++it) {
        const int buf = it & 1;
        const int next = tile + 1 < tend ? tile + 1 : ntiles;
        if (tile + 2 < tend) band_load(tile + 2, regs2);
        const int xt = tile % tiles_x;
        const int t2 = tile / tiles_x;
        const int oy = t2 % OH, n = t2 / OH;
        const int ox0 = xt * BX;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const char* band = &sBand[buf][0] + ((32 * wm + l31) * 2 * CIN + 8 * kh) * 2;
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const unsigned* src = reinterpret_cast<const unsigned*>(band + r * RS + g * 32);
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 raw = {src[0], src[1], src[2], src[3]};
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, raw), bw[r][g], acc, 0, 0, 0);
            }
        // stage the 64 x 64 output tile and accumulate the statistics of the live pixels
        const int col = 32 * wn + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int prow = 32 * wm + (e & 3) + 8 * (e >> 2) + 4 * kh;
            const float v = acc[e];
            Act<T>::st1(&sOut[prow * OLD + col], v);
            if (ox0 + prow < OW) { s1 += v; s2 += v * v; }
        }
        if (next < ntiles) band_store(buf ^ 1);                // (the band of tile + 1: requested a whole tile ago; tile + 2's stays in flight)
#pragma unroll
        for (int q = 0; q < NLD; ++q) regs[q] = regs2[q];
        __syncthreads();
        {
            constexpr int EPT = 16;                            // elements per thread: 64 x 64 / 256
            const int prow = tid >> 2, c0 = (tid & 3) * EPT;
            if (ox0 + prow < OW) {
                T* dst = static_cast<T*>(a.y) + ((size_t)(n * OH + oy) * OW + (size_t)(ox0 + prow)) * 64 + c0;
                const T* srcp = &sOut[prow * OLD + c0];
#pragma unroll
                for (int v = 0; v < EPT * (int)sizeof(T) / 16; ++v)
                    reinterpret_cast<f32x4*>(dst)[v] = reinterpret_cast<const f32x4*>(srcp)[v];
            }
        }
        __syncthreads();                                       // sOut and the consumed band buffer are free again
    }
    if (a.stats) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (kh == 0) { sRed[wm][0][32 * wn + l31] = s1; sRed[wm][1][32 * wn + l31] = s2; }
        __syncthreads();
        if (tid < 64) {
            float* dst = a.stats + (size_t)blockIdx.x * 2 * 64;
            dst[tid] = sRed[0][0][tid] + sRed[1][0][tid];
            dst[64 + tid] = sRed[0][1][tid] + sRed[1][1][tid];
        }
    }
}

// workgroups of the persistent kernel above (= its BatchNorm partial rows)
static int stem_rows_grid(int N, int H, int W, int Cin)
{
    // as many workgroups as are resident at once (139 VGPRs -> 3 per CU for RGB, 238 -> 2 for the 7-channel bird view): a fourth
    // wave of workgroups would start when the first ones finish and leave most CUs idle at the end
    const long long tiles = (long long)N * (H / 2) * lbc_cdiv(W / 2, 64);
    const long long grid = 256 * (Cin == 3 ? 3 : 2);
    return (int)(tiles < grid ? tiles : grid);
}
static bool stem_fwd_rows(const StemArgs& a) { return a.bf16 != 0; }     // (the bf16 modes always hand over a bf16 padded image)

}  // namespace

int lbc_prep_input_u8(const unsigned char* img_nhwc, void* xp, int xp_bf16, int N, int C, int H, int W, const NormConst& nc, hipStream_t s)
{
    LBC_REQUIRE(C <= 8, "prep_input: at most 8 channels");
    const long long total = (long long)N * (H + 6) * (W + 6);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    LbcProfScope prof("prep_input", 0.0, 1.0 * N * H * (double)W * C + (xp_bf16 ? 2.0 : 4.0) * (double)total * C, s);
    const bool small = total + blocks * 256 < (1ll << 31);
#define LBC_PU(XT, CINv)                                                                                                                 \
    do {                                                                                                                                 \
        if (small) hipLaunchKernelGGL((prep_input_u8_k<XT, CINv, unsigned>), dim3((unsigned)blocks), dim3(256), 0, s, img_nhwc, static_cast<XT*>(xp), N, C, H, W, nc);  \
        else       hipLaunchKernelGGL((prep_input_u8_k<XT, CINv, long long>), dim3((unsigned)blocks), dim3(256), 0, s, img_nhwc, static_cast<XT*>(xp), N, C, H, W, nc); \
    } while (0)
#define LBC_PUC(XT)                                                                                                                      \
    do {                                                                                                                                 \
        if (C == 3) LBC_PU(XT, 3);                                                                                                       \
        else if (C == 7) LBC_PU(XT, 7);                                                                                                  \
        else LBC_PU(XT, 0);                                                                                                              \
    } while (0)
    if (xp_bf16) LBC_PUC(__bf16);
    else         LBC_PUC(float);
#undef LBC_PUC
#undef LBC_PU
    return lbc_check_launch("prep_input_u8");
}

int lbc_prep_input(const float* img_nchw, void* xp, int xp_bf16, int N, int C, int H, int W, const NormConst& nc, hipStream_t s)
{
    LBC_REQUIRE(C <= 8, "prep_input: at most 8 channels");
    const long long total = (long long)N * (H + 6) * (W + 6);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    LbcProfScope prof("prep_input", 0.0, 4.0 * N * H * (double)W * C + (xp_bf16 ? 2.0 : 4.0) * (double)total * C, s);
    if (xp_bf16) hipLaunchKernelGGL(prep_input_k<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, img_nchw, static_cast<__bf16*>(xp), N, C, H, W, nc);
    else         hipLaunchKernelGGL(prep_input_k<float>, dim3((unsigned)blocks), dim3(256), 0, s, img_nchw, static_cast<float*>(xp), N, C, H, W, nc);
    return lbc_check_launch("prep_input");
}

int lbc_stem_rows(const StemArgs& a)
{
    if (stem_fwd_rows(a)) return stem_rows_grid(a.N, a.H, a.W, a.Cin);
    return lbc_cdiv((long long)a.N * (a.H / 2) * (a.W / 2), 128);
}

int lbc_stem_fwd(const StemArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.Cin == 3 || a.Cin == 7, "stem: Cin=%d unsupported (3 or 7)", a.Cin);
    LBC_REQUIRE(a.H % 2 == 0 && a.W % 2 == 0, "stem: odd image size");
    LBC_REQUIRE((long long)a.N * (a.H + 6) * (a.W + 6) * a.Cin < (1ll << 31), "stem: input too large");
    const dim3 grid((unsigned)lbc_stem_rows(a));
    const double Ms = (double)a.N * (a.H / 2) * (a.W / 2);
    LbcProfScope prof("stem_fwd", 2.0 * Ms * 64 * 49 * a.Cin, (a.xp_bf16 ? 2.0 : 4.0) * (double)a.N * (a.H + 6) * (a.W + 6) * a.Cin + (a.act_bf16 ? 2.0 : 4.0) * Ms * 64, s);
    LBC_REQUIRE(!a.act_bf16 || a.bf16, "stem: bf16 output needs bf16 = 1");
    if (stem_fwd_rows(a)) {
        LBC_REQUIRE(a.xp_bf16, "stem: the bf16 kernels read a bf16 padded image");
        const int tx = lbc_cdiv(a.W / 2, 64);
        const int ntiles = a.N * (a.H / 2) * tx;
#define LBC_K(T, CI) hipLaunchKernelGGL((stem_fwd_rows_k<CI, T>), grid, dim3(256), 0, s, a, ntiles, tx)
        if (a.Cin == 3) LBC_DISPATCH_ACT(a.act_bf16, LBC_K, 3);
        else            LBC_DISPATCH_ACT(a.act_bf16, LBC_K, 7);
#undef LBC_K
        return lbc_check_launch("stem_fwd");
    }
    if (a.Cin == 3) hipLaunchKernelGGL((stem_fwd_k<3, float>), grid, dim3(256), 0, s, a);
    else            hipLaunchKernelGGL((stem_fwd_k<7, float>), grid, dim3(256), 0, s, a);
    return lbc_check_launch("stem_fwd");
}

// true when the launch takes stem_wgrad_rows_k (all seven filter rows per workgroup)
static bool stem_wgrad_rows(int Cin, int bf16) { return bf16 && Cin == 3; }
bool lbc_stem_wgrad_fuses_bn_bwd(int Cin, int bf16) { return stem_wgrad_rows(Cin, bf16) && !lbc_opt_on(kOptNoBnBwdFuse); }

int lbc_stem_wgrad_split(int N, int H, int W, int Cin, int bf16)
{
    if (stem_wgrad_rows(Cin, bf16)) {
        // workgroups of 64-pixel chunks, two per CU, at least 8 chunks each; every workgroup writes two slabs (one per depth half)
        const long long chunks = ((long long)N * (H / 2) * (W / 2) + 63) / 64;
        long long ng = chunks / 8;
        if (ng > 512) ng = 512;
        if (ng < 1) ng = 1;
        return (int)(2 * ng);
    }
    const long long chunks = ((long long)N * (H / 2) * (W / 2) + 31) / 32;
    long long ns = chunks / 8;
    if (ns > 256) ns = 256;
    if (ns >= 8) ns &= ~7ll;          // multiples of 8: the filter rows of a split share an XCD (see the kernels)
    if (ns < 1) ns = 1;
    return (int)ns;
}

int lbc_stem_wgrad(const StemWgradArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.Cin == 3 || a.Cin == 7, "stem_wgrad: Cin=%d unsupported", a.Cin);
    const long long M = (long long)a.N * (a.H / 2) * (a.W / 2);
    LBC_REQUIRE(!a.act_bf16 || a.bf16, "stem_wgrad: bf16 gradients need bf16 = 1");
    LbcProfScope prof("stem_wgrad", 2.0 * M * 64 * 49 * a.Cin, 4.0 * ((double)a.N * (a.H + 6) * (a.W + 6) * a.Cin + (double)M * 64), s);
    LBC_REQUIRE(!a.bn_y || lbc_stem_wgrad_fuses_bn_bwd(a.Cin, a.bf16), "stem_wgrad: this kernel has no fused BatchNorm-backward apply");
    if (stem_wgrad_rows(a.Cin, a.bf16)) {
        LBC_REQUIRE(a.xp_bf16, "stem_wgrad: the bf16 kernels read a bf16 padded image");
        LBC_REQUIRE(a.nsplit >= 2 && a.nsplit % 2 == 0, "stem_wgrad: nsplit %d (use lbc_stem_wgrad_split)", a.nsplit);
        const int ng = a.nsplit / 2;
        const long long chunks = (M + 63) / 64;
        const int rows_per_split = (int)((chunks + ng - 1) / ng) * 64;
#define LBC_K(T, d) hipLaunchKernelGGL((stem_wgrad_rows_k<T>), dim3((unsigned)ng), dim3(256), 0, s, a, rows_per_split)
        LBC_DISPATCH_ACT(a.act_bf16, LBC_K, 0);
#undef LBC_K
        return lbc_check_launch("stem_wgrad");
    }
    const int br = a.bf16 ? 64 : 32;                         // pixels per chunk of the kernel
    const long long chunks = (M + br - 1) / br;
    const int rows_per_split = (int)((chunks + a.nsplit - 1) / a.nsplit) * br;
    const dim3 grid((unsigned)a.nsplit * 7);
    if (a.bf16) {
        LBC_REQUIRE(a.xp_bf16, "stem_wgrad: the bf16 kernels read a bf16 padded image");
#define LBC_K(T, CI) hipLaunchKernelGGL((stem_wgrad_bf16_k<CI, T, __bf16>), grid, dim3(256), 0, s, a, rows_per_split)
        if (a.Cin == 3) LBC_DISPATCH_ACT(a.act_bf16, LBC_K, 3);
        else            LBC_DISPATCH_ACT(a.act_bf16, LBC_K, 7);
#undef LBC_K
    } else {
        if (a.Cin == 3) hipLaunchKernelGGL((stem_wgrad_k<3, float>), grid, dim3(256), 0, s, a, rows_per_split);
        else            hipLaunchKernelGGL((stem_wgrad_k<7, float>), grid, dim3(256), 0, s, a, rows_per_split);
    }
    return lbc_check_launch("stem_wgrad");
}
