// Weight gradient of the 3x3 / stride-1 / pad-1 convolutions for bf16 tensors on gfx950, all nine taps in one workgroup.
// reference: autograd of BasicBlock conv1/conv2 (bird_view/models/resnet.py:15-22,38-54), loss.backward() at
// training/train_image_phase1.py:204.
//
//   out[p][tap][q] = sum_m  P[m][p] * Q[m + (r-1)*W + (s-1)][q]          (taps leaving the image contribute nothing)
//
// The generic kernel (conv_wgrad.hip) launches one workgroup per (tile, tap, split): every tap re-loads the same P and
// an almost identical Q, transposes both in registers on the way into LDS (the contraction runs over pixels, the slow
// axis of NHWC) and gets 4-16 MFMAs per barrier: 184 TF/s on the stem-resolution layer.  Here
//   * a workgroup owns a 64 x 64 (P x Q channel) tile for ALL taps over a contiguous pixel range: nine accumulators per
//     wave (144 registers), P staged once per 64-pixel chunk, Q kept as a ring of 64 + 2W + 2 (+64 incoming) pixel rows
//     to which every chunk appends 64 rows -- each element is loaded from HBM/L2 and written to LDS exactly once;
//   * LDS holds plain [pixel][channel] images (16-byte stores straight from the loads, BatchNorm+ReLU of the producer
//     applied once per element); the "8 consecutive pixels of one channel" MFMA fragments come from
//     ds_read_b64_tr_b16, and the tap shift is just a row offset of the read;
//   * out-of-image taps are removed on the P fragment: with W % 8 == 0 a lane's 8-pixel group lies in one image row, so
//     a tap is either invalid for the whole group (top / bottom row) or for its first / last pixel (left / right column);
//     other widths (the 5 x 12 maps of layer 4) build per-pixel keep-masks for the four border classes;
//   * one barrier per chunk, 36 MFMAs per wave between barriers, two workgroups per CU;
//   * a launch covers n same-shaped convolutions (WgradGroup: a ResNet stage's): the ~512 workgroups a launch needs come from
//     n x tiles x splits, so the pixel range is split n times less -- n times fewer partial tiles to write and to reduce.
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include <stdlib.h>
#include <string.h>

namespace {

// Ring rows: 64 + 2W + 2 live rows + 64 incoming.  LDS row stride: a transpose read serves 32 lanes per cycle = 4 pixel rows x 2
// 16-channel halves x 8 banks; with 192-byte rows the eight 8-bank groups are distinct (rows at 0, 48, 32, 16 of 64 banks, the
// second half + 8), with 160-byte rows row 3 of the second half lands on row 0 of the first (PMC: 45 % of the LDS cycles were
// bank conflicts).  192-byte rows fit two workgroups per CU up to W = 48; the 96-wide layer-1 maps keep 160.
constexpr int kRingWide = 328, kRsWide = 80;      // W <= 96
constexpr int kRingNarrow = 232, kRsNarrow = 96;  // W <= 48

// ALIGN = 8: W % 8 == 0 (a lane's 8-pixel group lies in one image row: whole-group / first-pixel / last-pixel masks).
// ALIGN = 4: W % 4 == 0 (the 5 x 12 maps of layer 4): the same per 4-pixel half of the group (the second half may sit in
// the next row).  ALIGN = 0: any W >= 8, every pixel gets its own flags.
template <int ALIGN, int kRS, int kRing>
__global__ __launch_bounds__(256, 2) void conv_wgrad_tr_k(WgradArgs a, WgradGroup grp, int rows_per_split)
{
    constexpr int BRH = 64;
    __shared__ __attribute__((aligned(16))) __bf16 sP[2][BRH * kRS];
    constexpr int kMirror = 24;                       // >= 18: see the tap loop
    __shared__ __attribute__((aligned(16))) __bf16 sQ[(kRing + kMirror) * kRS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wq = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int G1 = (lane >> 4) & 1, t16 = lane & 15;
    const int W = a.W, H = a.H;
    const int M = a.N * H * W;
    const int HRW = BRH + 2 * W + 2;                  // live halo rows of a chunk
    const int qtiles = a.CQ / 64;
    const int ntiles = (a.CP / 64) * qtiles;
    // workgroup -> (tile, member, split), tile fastest: the tiles of one (member, split) read the same pixel range, P once per Q tile and
    // Q once per P tile.  Workgroup ids go round-robin over the 8 XCDs, each with an L2 of its own: taken as they come, the 16 tiles of a
    // 256-channel (member, split) sit on 8 different L2s and every one of them fetches its share again (PMC: 1.92 GB fetched per
    // layer-3 group launch for 0.68 GB of operands, profiles/r03_final_pmc_summary_bf16.txt).  Logical index: XCD-major, so that
    // consecutive logical ids -- the tiles of one (member, split) -- are consecutive workgroups of ONE XCD.
    int logical;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, rr = nwg & 7;
        logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
        if (grp.linear_order) logical = b;
    }
    const int tile = logical % ntiles;
    const int member = (logical / ntiles) % grp.n, split = logical / (ntiles * grp.n);
    const int tp = tile / qtiles, tq = tile - tp * qtiles;
    const int p0 = tp * 64, q0 = tq * 64;
    const int mbeg = split * rows_per_split;
    const int mend = (mbeg + rows_per_split < M) ? mbeg + rows_per_split : M;
    const int nchunk = (mend > mbeg) ? (mend - mbeg + BRH - 1) / BRH : 0;
    const int qorg = mbeg - W - 1;                    // pixel held by ring row 0 (before wrapping)
    const __bf16* pin = static_cast<const __bf16*>(grp.p[member]);
    const __bf16* qin = static_cast<const __bf16*>(grp.q[member]);

    const int seg = tid & 7, srow = tid >> 3;         // staging: 16-byte segment (8 channels), row (+32 per pass)
    const float relu_floor = (a.q_scale && a.q_relu) ? 0.f : -INFINITY;
    f32x8 qsc = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, qsh = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.q_scale) { qsc = ParamVec<8>::ld(grp.q_scale[member] + q0 + seg * 8); qsh = ParamVec<8>::ld(grp.q_shift[member] + q0 + seg * 8); }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    if (nchunk > 0) {
        // ---- prologue: the halo of chunk 0 and its P rows ---------------------------------------------------------------
        for (int rel = srow; rel < HRW; rel += 32) {
            int q = qorg + rel;
            q = q < 0 ? 0 : (q >= M ? M - 1 : q);     // rows outside the tensor only ever meet masked taps; keep them finite
            bf16x8 h = *reinterpret_cast<const bf16x8*>(qin + ((unsigned)q * (unsigned)a.CQ + (unsigned)(q0 + seg * 8)));
            if (a.q_scale) {
                f32x8 v = __builtin_convertvector(h, f32x8) * qsc + qsh;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], relu_floor);
                h = __builtin_convertvector(v, bf16x8);
            }
            *reinterpret_cast<bf16x8*>(&sQ[rel * kRS + seg * 8]) = h;
            if (rel < kMirror) *reinterpret_cast<bf16x8*>(&sQ[(rel + kRing) * kRS + seg * 8]) = h;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = mbeg + srow + 32 * j;
            bf16x8 h = *reinterpret_cast<const bf16x8*>(pin + ((unsigned)(m < mend ? m : mbeg) * (unsigned)a.CP + (unsigned)(p0 + seg * 8)));
            if (m >= mend) h = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<bf16x8*>(&sP[0][(srow + 32 * j) * kRS + seg * 8]) = h;
        }
    }
    __syncthreads();

    int base = 0;                                     // ring row of the current chunk's first halo pixel
    // coordinates of this lane's first 8-pixel group (chunk 0, g = 0): pixel mbeg + 8*kh
    int gx, gy;
    {
        const int pm = mbeg + 8 * kh;
        gx = pm % W;
        gy = (pm / W) % H;
    }
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < nchunk;
        bf16x8 rp[2], rq[2];
        if (more) {
            const int mc = mbeg + (c + 1) * BRH;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = mc + srow + 32 * j;
                rp[j] = *reinterpret_cast<const bf16x8*>(pin + ((unsigned)(m < mend ? m : mbeg) * (unsigned)a.CP + (unsigned)(p0 + seg * 8)));
                int q = qorg + c * BRH + HRW + srow + 32 * j;
                q = q < 0 ? 0 : (q >= M ? M - 1 : q);
                rq[j] = *reinterpret_cast<const bf16x8*>(qin + ((unsigned)q * (unsigned)a.CQ + (unsigned)(q0 + seg * 8)));
            }
        }

        // ---- 4 groups of 16 pixels x 9 taps ------------------------------------------------------------------------------
        int x0 = gx, y = gy;
#pragma unroll
        for (int g = 0; g < BRH / 16; ++g) {
            // P^T fragment: channel p0 + 32*wp + (lane & 31), pixels 16g + 8kh + 0..7 of the chunk
            const int prow = 16 * g + 8 * kh + (t16 >> 2);
            const int pcol = 32 * wp + 16 * G1 + (t16 & 3) * 4;
            const bf16x4 a0 = lds_read_tr16(&sP[buf][prow * kRS + pcol]);
            const bf16x4 a1 = lds_read_tr16(&sP[buf][(prow + 4) * kRS + pcol]);
            // tap validity of the group's 8 pixels (two halves of 4: the elements of the two transpose reads)
            bool top[2] = {false, false}, bottom[2] = {false, false}, left[2] = {false, false}, right[2] = {false, false};
            unsigned mt[4], mb[4], ml[4], mr[4];      // ALIGN 0: per-dword keep-masks (two bf16 each) of the four border classes
            if constexpr (ALIGN == 8) {
                top[0] = top[1] = y == 0; bottom[0] = bottom[1] = y == H - 1;
                left[0] = x0 == 0; right[1] = x0 == W - 8;
            } else if constexpr (ALIGN == 4) {
                const bool wrap = x0 + 4 >= W;                 // the second half starts the next image row
                const int x1 = wrap ? 0 : x0 + 4;
                const int y1 = wrap ? (y + 1 == H ? 0 : y + 1) : y;
                top[0] = y == 0; bottom[0] = y == H - 1; left[0] = x0 == 0; right[0] = x0 == W - 4;
                top[1] = y1 == 0; bottom[1] = y1 == H - 1; left[1] = x1 == 0; right[1] = x1 == W - 4;
            } else {
#pragma unroll
                for (int d = 0; d < 4; ++d) { mt[d] = 0xffffffffu; mb[d] = 0xffffffffu; ml[d] = 0xffffffffu; mr[d] = 0xffffffffu; }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool wrap = x0 + i >= W;             // W >= 8: at most one row change inside a group
                    const int xi = wrap ? x0 + i - W : x0 + i;
                    const int yi = wrap ? (y + 1 == H ? 0 : y + 1) : y;
                    const unsigned clr = ~(0xffffu << (16 * (i & 1)));
                    if (yi == 0) mt[i >> 1] &= clr;
                    if (yi == H - 1) mb[i >> 1] &= clr;
                    if (xi == 0) ml[i >> 1] &= clr;
                    if (xi == W - 1) mr[i >> 1] &= clr;
                }
            }
            // halves with their first / last pixel removed (no left / right neighbour)
            bf16x4 a0l = a0, a0r = a0, a1l = a1, a1r = a1;
            if (ALIGN != 0) {
                if (left[0]) a0l[0] = (__bf16)0.f;
                if (right[0]) a0r[3] = (__bf16)0.f;
                if (left[1]) a1l[0] = (__bf16)0.f;
                if (right[1]) a1r[3] = (__bf16)0.f;
            }
            // one tap: masked P^T fragment x Q fragment (two transpose reads, 4 + 4 pixels)
            auto tap = [&](const int r, const int s, const bf16x4 b0, const bf16x4 b1) {
                bf16x8 af;
                if constexpr (ALIGN != 0) {
                    bf16x4 u0 = s == 0 ? a0l : (s == 2 ? a0r : a0);
                    bf16x4 u1 = s == 0 ? a1l : (s == 2 ? a1r : a1);
                    if ((r == 0 && top[0]) || (r == 2 && bottom[0])) u0 = bf16x4{0, 0, 0, 0};
                    if ((r == 0 && top[1]) || (r == 2 && bottom[1])) u1 = bf16x4{0, 0, 0, 0};
                    af = __builtin_shufflevector(u0, u1, 0, 1, 2, 3, 4, 5, 6, 7);
                } else {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const bf16x8 full = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                    u32x4 bits = __builtin_bit_cast(u32x4, full);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        unsigned keep = 0xffffffffu;
                        if (r == 0) keep &= mt[d];
                        if (r == 2) keep &= mb[d];
                        if (s == 0) keep &= ml[d];
                        if (s == 2) keep &= mr[d];
                        bits[d] &= keep;
                    }
                    af = __builtin_bit_cast(bf16x8, bits);
                }
                const bf16x8 bf = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                // operands swapped: the accumulator holds the TRANSPOSED tile -- lane = P channel, registers = Q channels, four consecutive
                // ones per register quad -- so that the slab row out[p][tap][q .. q + 3] leaves as one 16-byte store (same products, same
                // summation order: bit-identical to the untransposed form, whose 144 four-byte stores per lane were issue-bound)
                acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, af, acc[r * 3 + s], 0, 0, 0);
            };
            const int qlane = (8 * kh + (t16 >> 2)) * kRS + 32 * wq + 16 * G1 + (t16 & 3) * 4;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                // ring rows of this filter row's three taps: lo + [0, 17] (8 kh + t16 / 4 + s + 4).  lo is wave-uniform and the
                // first kMirror ring rows are mirrored behind the ring, so a span that straddles the wrap point reads on into the
                // mirror: the six reads are ONE address + immediate offsets (the per-lane wrap arithmetic of every read made this
                // kernel VALU-bound -- PMC: VALU busy 36 %, MFMA 28 %).  (Requesting the fragments one filter row ahead of their
                // MFMAs from a second register set was measured 5 % slower.)
                const int lo = base + 16 * g + r * W;
                const __bf16* q = &sQ[(lo >= kRing ? lo - kRing : lo) * kRS + qlane];
#pragma unroll
                for (int s = 0; s < 3; ++s) tap(r, s, lds_read_tr16(q + s * kRS), lds_read_tr16(q + (s + 4) * kRS));
            }
            x0 += 16;
            while (x0 >= W) { x0 -= W; if (++y >= H) y = 0; }
        }
        // next chunk's first group: 64 pixels on
        gx += BRH;
        while (gx >= W) { gx -= W; if (++gy >= H) gy = 0; }

        if (more) {
            const int mc = mbeg + (c + 1) * BRH;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = mc + srow + 32 * j;
                bf16x8 h = rp[j];
                if (m >= mend) h = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<bf16x8*>(&sP[buf ^ 1][(srow + 32 * j) * kRS + seg * 8]) = h;
                bf16x8 hq = rq[j];
                if (a.q_scale) {
                    f32x8 v = __builtin_convertvector(hq, f32x8) * qsc + qsh;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], relu_floor);
                    hq = __builtin_convertvector(v, bf16x8);
                }
                int slot = base + HRW + srow + 32 * j;          // appended rows never overlap the live halo (HRW + 64 <= kRing)
                slot = slot >= kRing ? slot - kRing : slot;
                *reinterpret_cast<bf16x8*>(&sQ[slot * kRS + seg * 8]) = hq;
                if (slot < kMirror) *reinterpret_cast<bf16x8*>(&sQ[(slot + kRing) * kRS + seg * 8]) = hq;
            }
        }
        base += BRH;
        base = base >= kRing ? base - kRing : base;
        __syncthreads();
    }

    float* out = grp.out[member] + (size_t)split * (size_t)a.CP * 9 * (size_t)a.CQ;
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

bool lbc_wgrad_tr_eligible(const WgradArgs& a)
{
    return a.act_bf16 && a.KH == 3 && a.KW == 3 && a.S == 1 && a.P == 1 && a.OH == a.H && a.OW == a.W && !a.p_scale &&
           a.W >= 8 && 64 + 2 * a.W + 2 + 64 <= kRingWide && a.CP % 64 == 0 && a.CQ % 64 == 0;
}

int lbc_wgrad_tr_pick_split(const WgradArgs& a)
{
    const long long target = 512;        // (r03_run4_wgrad_tr_blocks_ab.log: 256 / 384 / 768 / 1024 all slower)
    const long long tiles = (long long)(a.CP / 64) * (a.CQ / 64);
    const long long M = (long long)a.N * a.H * a.W;
    const long long chunks = (M + 63) / 64;
    long long ns = (target + tiles - 1) / tiles;
    const long long maxns = chunks / 8 > 0 ? chunks / 8 : 1;     // >= 8 chunks per split: the halo prologue is ~4 chunks of loads
    if (ns > maxns) ns = maxns;
    if (ns > 1024) ns = 1024;
    if (ns < 1) ns = 1;
    return (int)ns;
}

// Splits of a grouped launch: the slots are 2 workgroups x 256 CUs, workgroups of one launch take the same time, so the launch
// lasts ceil(workgroups / slots) rounds of (chunks per split + ~6 chunks of halo prologue and slab stores).  The smallest split count
// within 2 % of the best: fewer slabs for the same time.
int lbc_wgrad_tr_group_split(const WgradArgs& a, int n)
{
    if (n <= 1) return lbc_wgrad_tr_pick_split(a);
    const long long slots = 512;
    const long long tiles = (long long)(a.CP / 64) * (a.CQ / 64) * n;
    const long long chunks = ((long long)a.N * a.H * a.W + 63) / 64;
    long long maxns = chunks / 8 > 0 ? chunks / 8 : 1;
    if (maxns > 1024) maxns = 1024;
    auto cost = [&](long long ns) { return (double)((tiles * ns + slots - 1) / slots) * ((double)((chunks + ns - 1) / ns) + 6.0); };
    double best = cost(1);
    for (long long ns = 2; ns <= maxns; ++ns) best = cost(ns) < best ? cost(ns) : best;
    for (long long ns = 1; ns <= maxns; ++ns)
        if (cost(ns) <= 1.02 * best) return (int)ns;
    return 1;
}

int lbc_wgrad_tr_launch(const WgradArgs& a, hipStream_t s)
{
    WgradGroup g;
    memset(&g, 0, sizeof(g));
    g.n = 1; g.p[0] = a.p; g.q[0] = a.q; g.q_scale[0] = a.q_scale; g.q_shift[0] = a.q_shift; g.out[0] = a.partial;
    return lbc_wgrad_tr_group_launch(a, g, s);
}

int lbc_wgrad_tr_group_launch(const WgradArgs& a0, const WgradGroup& g_in, hipStream_t s)
{
    LBC_REQUIRE(g_in.n >= 1 && g_in.n <= kLbcWgradGroupMax, "wgrad_tr: group of %d", g_in.n);
    WgradGroup g = g_in;
    g.linear_order = 0;       // XCD-major workgroup order (r03_run31: equal in time, half the fetched bytes)
    WgradArgs a = a0;
    a.p = g.p[0]; a.q = g.q[0]; a.q_scale = g.q_scale[0]; a.q_shift = g.q_shift[0]; a.partial = g.out[0];
    LBC_REQUIRE(lbc_wgrad_tr_eligible(a), "wgrad_tr: launch not eligible");
    for (int i = 0; i < g.n; ++i)
        LBC_REQUIRE(g.p[i] && g.q[i] && g.out[i] && (g.q_scale[i] != nullptr) == (a.q_scale != nullptr) && (g.q_shift[i] != nullptr) == (a.q_scale != nullptr),
                    "wgrad_tr: group member %d incomplete", i);
    const long long M = (long long)a.N * a.H * a.W;
    // (32-bit element offsets in the kernel: the executor's deferred launches come here directly, not through lbc_wgrad_launch)
    LBC_REQUIRE(M * a.CP < (1ll << 31) && M * a.CQ < (1ll << 31) && a.nsplit >= 1, "wgrad_tr: tensors too large for 32-bit indexing");
    const long long chunks = (M + 63) / 64;
    const int rows_per_split = (int)((chunks + a.nsplit - 1) / a.nsplit) * 64;
    const unsigned blocks = (unsigned)((a.CP / 64) * (a.CQ / 64) * a.nsplit * g.n);
    LbcProfScope prof("conv_wgrad_tr", g.n * 2.0 * (double)M * a.CP * (double)a.CQ * 9,
                      g.n * (2.0 * ((double)M * a.CP + (double)M * a.CQ) + 4.0 * (double)a.nsplit * a.CP * 9 * a.CQ), s);
#define LBC_WT(AL)                                                                                                              \
    do {                                                                                                                        \
        if (64 + 2 * a.W + 2 + 64 <= kRingNarrow)                                                                               \
            hipLaunchKernelGGL((conv_wgrad_tr_k<AL, kRsNarrow, kRingNarrow>), dim3(blocks), dim3(256), 0, s, a, g, rows_per_split); \
        else                                                                                                                    \
            hipLaunchKernelGGL((conv_wgrad_tr_k<AL, kRsWide, kRingWide>), dim3(blocks), dim3(256), 0, s, a, g, rows_per_split);     \
    } while (0)
    if (a.W % 8 == 0)      LBC_WT(8);
    else if (a.W % 4 == 0) LBC_WT(4);
    else                   LBC_WT(0);
#undef LBC_WT
    return lbc_check_launch("conv_wgrad_tr");
}
