// Weight gradient of the 3x3 / STRIDE-2 / pad-1 layers for bf16 tensors on gfx950, all nine taps in one workgroup: the first convolution of
// layers 2-4 (bird_view/models/resnet.py:15-22,124-146, `_make_layer` with stride 2) and -- the same index relation with the roles of
// the two tensors swapped -- the three ConvTranspose2d(.,.,3,2,1,1) of the decoder (bird_view/models/image.py:39,42,45); autograd behind
// loss.backward() at training/train_image_phase1.py:204.
//
//   out[p][tap][q] = sum_m  P[m][p] * Q[pix(m, r, s)][q]      m = (n, oy, ox) on the LOW-resolution lattice (OH x OW),
//                                                             pix = (n, 2 oy - 1 + r, 2 ox - 1 + s) on the HIGH-resolution one (H = 2 OH, W = 2 OW)
//
// (Conv2d: P = dY, Q = x;  ConvTranspose2d: P = bn(x), Q = dY.)  The generic kernel (conv_wgrad.hip) runs one workgroup per (tile, tap,
// split), re-loads both operands per tap and transposes them in registers: 220-370 TF/s on these six launches, 0.77 ms of the step at 256
// images.  conv_wgrad_tr.hip's structure carries over once the stride is seen as a RATE: with hi(m) = n H W + 2 oy W + 2 ox the linear
// index of a low-resolution pixel's centre tap, hi(m) = 4 m - 2 ox -- so the high-resolution rows a 32-pixel chunk [m0, m0 + 32) needs
// lie in the fixed window [4 m0 - 2 W, 4 m0 + 128 + W + 2), which advances by exactly 128 rows per chunk:
//   * Q is a ring over high-resolution pixels, 128 + 3 W + 2 live rows + 128 incoming, appended 128 rows per chunk -- every element is
//     loaded once; P is staged per 32-pixel chunk (double-buffered);
//   * a tap is a row offset of the transpose read again: row = window + 64 g - 2 ox(g) + (r + 1) W - 1 (wave-uniform) + 2 d + wraps W
//     + 2 j + s per lane (d = 8 kh + 4 h: the lane's 4-pixel half inside the 16-pixel group, j its pixel, wraps = output-row changes
//     inside the group); the first kMirror ring rows are mirrored behind the ring so that no lane wraps;
//   * only two border classes exist: r = 0 leaves the image at oy = 0, s = 0 at ox = 0 (2 oy + 1 <= H - 1, 2 ox + 1 <= W - 1);
//   * 8 waves (4 x 2 sub-tiles of a 128 x 64 P x Q tile), one workgroup per CU (the ring for W = 96 is 132 KB), 18 MFMAs per wave and
//     barrier.
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include <stdlib.h>
#include <string.h>

namespace {

// ring pitch: 64 channels + 16 = 160 bytes.  The four pixel rows of a transpose read lie TWO ring rows apart here (consecutive low-resolution
// pixels): 2 x 160 = 320 = 64 (mod 256) puts their 64-byte windows on four disjoint bank groups.  (The 192 bytes of the stride-1 kernel
// put rows 0 / 2 and 1 / 3 on the same banks: PMC SQ_LDS_BANK_CONFLICT 37 % of the LDS cycles, profiles/r03_final_pmc_lds_conflicts.txt.)
constexpr int kRsQ = 80;
constexpr int kRsP = 160;        // P pitch: 128 channels + 32 (320 bytes: likewise for the 64-byte window of a wave's 32 channels)

// kRing >= 128 + 3 W + 2 + 128, kMirror >= W + 40; kRing % 8 == 0
template <int kRing, int kMirror, bool PSCALE>
__global__ __launch_bounds__(512, 2) void conv_wgrad_tr2_k(WgradArgs a, int rows_per_split)
{
    constexpr int BRH = 32;
    __shared__ __attribute__((aligned(16))) __bf16 sP[2][BRH * kRsP];
    __shared__ __attribute__((aligned(16))) __bf16 sQ[(kRing + kMirror) * kRsQ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wq = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int G1 = (lane >> 4) & 1, t16 = lane & 15;
    const int OW = a.OW, OH = a.OH, W = a.W;
    const int M = a.N * OH * OW;                      // low-resolution pixels (rows of P)
    const int MQ = a.N * a.H * a.W;                   // high-resolution pixels (rows of Q)
    const int HRW = 128 + 3 * W + 2;                  // live window of a chunk
    const int qtiles = a.CQ / 64;
    const int ntiles = (a.CP / 128) * qtiles;
    int logical;                                      // XCD-major order: the tiles of one split on one L2 (conv_wgrad_tr.hip)
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, rr = nwg & 7;
        logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
    }
    const int tile = logical % ntiles, split = logical / ntiles;
    const int tp = tile / qtiles, tq = tile - tp * qtiles;
    const int p0 = tp * 128, q0 = tq * 64;
    const int mbeg = split * rows_per_split;
    const int mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
    const int nchunk = (mend > mbeg) ? (mend - mbeg + BRH - 1) / BRH : 0;
    const int qorg = 4 * mbeg - 2 * W;                // high-resolution pixel held by ring row 0 (before wrapping)
    const __bf16* pin = static_cast<const __bf16*>(a.p);
    const __bf16* qin = static_cast<const __bf16*>(a.q);

    // staging roles: P row = 128 channels = 16 threads, Q row = 64 channels = 8 threads (two passes of 64 rows)
    const int segP = tid & 15, srowP = tid >> 4;
    const int segQ = tid & 7, srowQ = tid >> 3;
    f32x8 psc = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (PSCALE) { psc = ParamVec<8>::ld(a.p_scale + p0 + segP * 8); psh = ParamVec<8>::ld(a.p_shift + p0 + segP * 8); }
    auto load_p = [&](const int m) {
        bf16x8 h = *reinterpret_cast<const bf16x8*>(pin + ((unsigned)(m < mend ? m : mbeg) * (unsigned)a.CP + (unsigned)(p0 + segP * 8)));
        return h;
    };
    auto store_p = [&](bf16x8 h, const int m, const int buf) {
        if (PSCALE) h = __builtin_convertvector(__builtin_convertvector(h, f32x8) * psc + psh, bf16x8);
        if (m >= mend) h = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};          // (after the transform: rows past the split contribute nothing)
        *reinterpret_cast<bf16x8*>(&sP[buf][srowP * kRsP + segP * 8]) = h;
    };
    auto load_q = [&](const int rel) {
        int q = qorg + rel;
        q = q < 0 ? 0 : (q >= MQ ? MQ - 1 : q);       // rows outside the tensor only ever meet masked taps; keep them finite
        return *reinterpret_cast<const bf16x8*>(qin + ((unsigned)q * (unsigned)a.CQ + (unsigned)(q0 + segQ * 8)));
    };
    auto store_q = [&](const bf16x8 h, int slot) {     // slot in [0, 2 kRing)
        slot = slot >= kRing ? slot - kRing : slot;
        *reinterpret_cast<bf16x8*>(&sQ[slot * kRsQ + segQ * 8]) = h;
        if (slot < kMirror) *reinterpret_cast<bf16x8*>(&sQ[(slot + kRing) * kRsQ + segQ * 8]) = h;
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    if (nchunk > 0) {
        for (int rel = srowQ; rel < HRW; rel += 64) store_q(load_q(rel), rel);
        store_p(load_p(mbeg + srowP), mbeg + srowP, 0);
    }
    __syncthreads();

    int base = 0;                                     // ring row of the current chunk's window start (high-resolution pixel 4 m0 - 2 W)
    // low-resolution coordinates of the chunk's first pixel (wave-uniform)
    int cx = mbeg % OW, cy = (mbeg / OW) % OH;
    const int j = t16 >> 2;                           // this lane's pixel inside a 4-pixel transpose read
    // Loads run TWO chunks ahead of their use: one workgroup per CU and 18 MFMAs per wave and chunk (~1150 matrix-pipe cycles per SIMD) do not
    // cover an HBM round trip.  Chunk k's rows travel in register set k & 1 (set A even, B odd): requested at the top of chunk k - 2, written
    // to LDS at the bottom of chunk k - 1.  The loop is unrolled by two so that the sets stay registers.
    bf16x8 rpA = {}, rqA[2] = {}, rpB = {}, rqB[2] = {};
    auto request = [&](const int k, bf16x8& rp, bf16x8 (&rq)[2]) {        // chunk k >= 1: its P rows and the 128 ring rows its window adds
        if (k < nchunk) {
            rp = load_p(mbeg + k * BRH + srowP);
#pragma unroll
            for (int i = 0; i < 2; ++i) rq[i] = load_q((k - 1) * 128 + HRW + srowQ + 64 * i);
        }
    };
    request(1, rpB, rqB);
    auto chunk_body = [&](const int c, bf16x8& rp_req, bf16x8 (&rq_req)[2], const bf16x8& rp_st, const bf16x8 (&rq_st)[2]) {
        const int buf = c & 1;
        const bool more = c + 1 < nchunk;
        request(c + 2, rp_req, rq_req);

        // ---- 2 groups of 16 pixels x 9 taps
        int gx = cx, gy = cy;
#pragma unroll
        for (int g = 0; g < BRH / 16; ++g) {
            // the lane's two 4-pixel halves: pixels 16 g + 8 kh + 4 h + (0..3) of the chunk; halves never straddle an output row (OW % 4 == 0)
            int rowoff[2];
            bool top[2], left[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int d = 8 * kh + 4 * h;
                int t = gx + d, wr = 0;
                if (t >= OW) { t -= OW; wr = 1; }     // (OW >= 12: a 16-pixel group changes rows at most once before its last half...
                if (t >= OW) { t -= OW; wr = 2; }     //  ...and at most twice at all)
                int y = gy + wr;
                y = y >= OH ? y - OH : y;             // next image
                rowoff[h] = (2 * d + wr * W + 2 * j) * kRsQ;
                top[h] = y == 0;
                left[h] = t == 0;
            }
            const int prow = 16 * g + 8 * kh + j;
            const int pcol = 32 * wp + 16 * G1 + (t16 & 3) * 4;
            const bf16x4 a0 = lds_read_tr16(&sP[buf][prow * kRsP + pcol]);
            const bf16x4 a1 = lds_read_tr16(&sP[buf][(prow + 4) * kRsP + pcol]);
            bf16x4 a0l = a0, a1l = a1;               // halves with their first pixel removed (no left neighbour)
            if (left[0]) a0l[0] = (__bf16)0.f;
            if (left[1]) a1l[0] = (__bf16)0.f;
            const int qcol = 32 * wq + 16 * G1 + (t16 & 3) * 4;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                // wave-uniform ring row of (first pixel of the group, filter row r, column s = 0); >= 0 by construction
                const int lo = base + 64 * g - 2 * gx + (r + 1) * W - 1;
                const __bf16* q = &sQ[(lo >= kRing ? lo - kRing : lo) * kRsQ + qcol];
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const bf16x4 b0 = lds_read_tr16(q + rowoff[0] + s * kRsQ);
                    const bf16x4 b1 = lds_read_tr16(q + rowoff[1] + s * kRsQ);
                    bf16x4 u0 = s == 0 ? a0l : a0, u1 = s == 0 ? a1l : a1;
                    if (r == 0 && top[0]) u0 = bf16x4{0, 0, 0, 0};
                    if (r == 0 && top[1]) u1 = bf16x4{0, 0, 0, 0};
                    const bf16x8 af = __builtin_shufflevector(u0, u1, 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x8 bf = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    // (operands swapped: transposed accumulators, 16-byte slab rows -- conv_wgrad_tr.hip)
                    acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, af, acc[r * 3 + s], 0, 0, 0);
                }
            }
            gx += 16;
            while (gx >= OW) { gx -= OW; if (++gy >= OH) gy = 0; }
        }
        cx += BRH;
        while (cx >= OW) { cx -= OW; if (++cy >= OH) cy = 0; }

        if (more) {
            const int mc = mbeg + (c + 1) * BRH;
            store_p(rp_st, mc + srowP, buf ^ 1);
#pragma unroll
            for (int k = 0; k < 2; ++k) store_q(rq_st[k], base + HRW + srowQ + 64 * k);     // (never overlaps the live window: HRW + 128 <= kRing)
        }
        base += 128;
        base = base >= kRing ? base - kRing : base;
        __syncthreads();
    };
    for (int c = 0; c < nchunk; c += 2) {
        chunk_body(c, rpA, rqA, rpB, rqB);                     // requests chunk c + 2 (even: set A), stores chunk c + 1 (odd: set B)
        if (c + 1 < nchunk) chunk_body(c + 1, rpB, rqB, rpA, rqA);
    }

    float* out = a.partial + (size_t)split * (size_t)a.CP * 9 * (size_t)a.CQ;
    const int prow = p0 + 32 * wp + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int qcol = q0 + 32 * wq + 8 * k + 4 * kh;
            *reinterpret_cast<f32x4*>(out + (((size_t)prow * 9 + (size_t)t) * (size_t)a.CQ + (size_t)qcol)) =
                f32x4{acc[t][4 * k], acc[t][4 * k + 1], acc[t][4 * k + 2], acc[t][4 * k + 3]};
        }
}

}  // namespace

bool lbc_wgrad_tr2_eligible(const WgradArgs& a)
{
    // One workgroup per CU, each paying a window prologue worth ~3.5 chunks of loads: worth it from ~192 workgroups of >= 16 chunks (measured:
    // -0.26 ms per step at 256 images, +0.08 ms at 32 where 60 workgroups remain; LBC_WGRAD_TR2_MIN_WGS overrides, tests use 1)
    const long long min_wgs = lbc_opt(kOptWgradTr2MinWgs) > 0 ? lbc_opt(kOptWgradTr2MinWgs) : 192;
    const long long chunks = ((long long)a.N * a.OH * a.OW + 31) / 32;
    if ((long long)(a.CP / 128) * (a.CQ / 64) * (chunks / 16) < min_wgs) return false;
    return a.act_bf16 && a.KH == 3 && a.KW == 3 && a.S == 2 && a.P == 1 && a.H == 2 * a.OH && a.W == 2 * a.OW && !a.q_scale &&
           (!a.p_scale || a.p_shift) && a.OW % 4 == 0 && a.OW >= 12 && a.W <= 96 && a.CP % 128 == 0 && a.CQ % 64 == 0 &&
           (long long)a.N * a.H * a.W * 4 < (1ll << 31);
}

int lbc_wgrad_tr2_pick_split(const WgradArgs& a)
{
    const long long target = 256;        // one workgroup per CU (r03_run30_wgrad_stride2_blocks_sweep.log)
    const long long tiles = (long long)(a.CP / 128) * (a.CQ / 64);
    const long long M = (long long)a.N * a.OH * a.OW;
    const long long chunks = (M + 31) / 32;
    long long ns = target / tiles;                               // rounded DOWN: one workgroup per CU, a 257th would wait for a whole round (the
                                                                 // decoder's first stage, 20 tiles: 13 splits = 260 workgroups took 145 us, 12 take 75)
    const long long maxns = chunks / 16 > 0 ? chunks / 16 : 1;   // >= 16 chunks per split: the window prologue is ~3-4 chunks of loads
    if (ns > maxns) ns = maxns;
    if (ns > 1024) ns = 1024;
    if (ns < 1) ns = 1;
    return (int)ns;
}

int lbc_wgrad_tr2_launch(const WgradArgs& a, hipStream_t s)
{
    LBC_REQUIRE(lbc_wgrad_tr2_eligible(a), "wgrad_tr2: launch not eligible");
    const long long M = (long long)a.N * a.OH * a.OW;
    const long long chunks = (M + 31) / 32;
    const int rows_per_split = (int)((chunks + a.nsplit - 1) / a.nsplit) * 32;
    const unsigned blocks = (unsigned)((a.CP / 128) * (a.CQ / 64) * a.nsplit);
    LbcProfScope prof("conv_wgrad_tr", 2.0 * (double)M * a.CP * (double)a.CQ * 9,
                      2.0 * ((double)M * a.CP + 4.0 * (double)M * a.CQ) + 4.0 * (double)a.nsplit * a.CP * 9 * a.CQ, s);
#define LBC_W2(RING, MIR)                                                                                                          \
    do {                                                                                                                           \
        if (a.p_scale) hipLaunchKernelGGL((conv_wgrad_tr2_k<RING, MIR, true>), dim3(blocks), dim3(512), 0, s, a, rows_per_split);  \
        else           hipLaunchKernelGGL((conv_wgrad_tr2_k<RING, MIR, false>), dim3(blocks), dim3(512), 0, s, a, rows_per_split); \
    } while (0)
    if (a.W <= 48) LBC_W2(408, 88);       // 128 + 3 * 48 + 2 + 128 = 402
    else           LBC_W2(552, 136);      // 128 + 3 * 96 + 2 + 128 = 546
#undef LBC_W2
    return lbc_check_launch("conv_wgrad_tr2");
}
