// EXPERIMENT KIT, not part of the library (round 6; built only by scripts/probe/hdmaw_prof.hip): a WAVE-SPECIALISED form of the persistent
// halo-staged convolution (learningbycheating_amd/csrc/conv_hdmap.hpp explains the staging and the tile stream), 3x3 / stride-1 / pad-1 forward
// and input gradient on bf16 tensors, 256 x 128 output tiles; reference arithmetic bird_view/models/resnet.py:15-22,38-54 and autograd.
// It exists to answer "what is conv_hdmap_k bound by" with stamped cycles instead of guesses; every form of it was also run through the
// library's parity tests on the GPU before it was measured.  Results (profiles/r06_hdmap_cycle_budget.md has the table):
//   * waves 0-3 MULTIPLY (one per SIMD, 128 x 64 wave tiles, no VMEM in the K loop), waves 4-7 LOAD (all LDS-DMA requests);
//   * K loop: 1032 cycles per K-tile for the bare MFMA stream (32 MFMAs: the floor), 1260 with the six fragment reads of a depth step in front
//     of its MFMAs, 1124 with them inside the first three MFMA gaps of the previous step; K-tile barrier 45; the chip runs this at 1.62 - 1.66
//     GHz (bare MFMA stream 1.82, zero operands 1.92: the power cap, not the kernel, sets the clock);
//   * epilogue: 8.6 k cycles per tile wave-private in the one-channel-per-lane accumulator layout, 6.5 k with TRANSPOSED accumulators (the
//     weight fragment as the MFMA's A operand: four consecutive channels of one pixel per register quad, 8-byte LDS writes), and -- this
//     version -- a 2 - 3.7 k DUMP by the multiplying waves with the loaders DRAINING the image under the next tile's K loop ... which made
//     the launch SLOWER (78 vs 66 us): a wave that shares a SIMD with a dense MFMA stream gets about one issue slot per MFMA (a DMA
//     request costs it 137 cycles, a VALU instruction 30 - 45), and raising its priority takes the time out of the MFMA stream instead.
//     Work moved to the partner wave is not free on this chip; only the wave's OWN fillers (<= 5 per MFMA gap) are.
//   * in the training step the best form (transposed accumulators, wave-private epilogue) ran the plain launches 1.5 % faster than
//     conv_hdmap_k and the fused-reduce forms 9 - 34 % slower (one wave per SIMD does the chunk phase two did): the library keeps conv_hdmap_k.
// This file is the last (dump / drain) version; git history of learningbycheating_amd/csrc/conv_hdmaw.hpp has the earlier ones.
#pragma once
#include <type_traits>
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include "conv_lds_dma.hpp"

namespace {

__device__ __forceinline__ const char* hdmaw_uniform_ptr(const char* p)
{
    const unsigned long long v = (unsigned long long)(size_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char*>((size_t)(((unsigned long long)hi << 32) | lo));
}

// LBC_HDMAW_PROF (scripts/probe/hdmaw_prof.hip only): per-wave s_memtime sums -- multiplying waves: [0] K-tile segments between barriers (the dump included),
// [1] inside the K-tile barriers, [2] dump + its barrier, [3] first stamp .. last stamp; loading waves: [0] requests + drain work, [1] counted vmcnt wait,
// [2] barrier, [3] total.  Timing-experiment builds of the same probe: LBC_HDMAW_ABL_NOMFMA / _NOREAD / _NODMA drop the MFMAs / the fragment reads / the
// loaders' requests of the main loop (results are then wrong; only the time is looked at).
#ifdef LBC_HDMAW_PROF
__device__ unsigned long long g_hdmaw_prof[256 * 8 * 4];
#define LBC_PROF(...) __VA_ARGS__
#else
#define LBC_PROF(...)
#endif

// LDS map: two halo buffers, the ring of three weight tiles, [2][2][BN] floats for the statistics rows, and the spare part of the dump image
template <int BN, int HRMAX> struct HdmawLds {
    static constexpr int ABYTES = HRMAX * 128, TILE_B = BN * 128, BRING = 2 * ABYTES, RED = BRING + 3 * TILE_B, SPARE = RED + 2 * 2 * BN * 4;
    static constexpr int SPR = ((160 * 1024 - SPARE) / 256) & ~3;        // rows of the dump image in the spare region
    static constexpr int RX = 256 - SPR;                                 // ... and in the halo buffer (rows [0, RX))
    static constexpr int SMEM = SPARE + SPR * 256;
    static_assert(RX >= 0 && RX * 256 <= ABYTES && RX % 4 == 0 && SMEM <= 160 * 1024, "conv_hdmaw: the dump image does not fit");
};

// MODE 0 forward, 1 input gradient (flipped taps); EPI 0 plain, 1 + residual, 2 fused BatchNorm-backward reduce (IgemmArgs::bnb_*), 4 = 1 + 2
// with the mask from a tensor (IgemmArgs::bnb_mask)
template <int BM, int BN, int HRMAX, int MODE, int EPI>
__global__ __launch_bounds__(512, 2) void conv_hdmaw_k(IgemmArgs a, const void* zero_page, const int ntiles, const int tpw)
{
    constexpr int WM = 2, WN = 2, NWC = WM * WN, NWL = 4;       // multiplying waves (2 x 2), loading waves
    constexpr int WTM = BM / WM, WTN = BN / WN;                 // per-wave output tile
    constexpr int MT = WTM / 32, NT = WTN / 32;
    static_assert(BM == 256 && BN == 128 && MT == 4 && NT == 2 && (MODE == 0 || MODE == 1) && (EPI == 0 || EPI == 1 || EPI == 2 || EPI == 4), "conv_hdmaw: wave tiling / forms");
    using L = HdmawLds<BN, HRMAX>;
    constexpr int KS = 4;                                       // depth steps of 16 channels per K-tile
    constexpr int ABYTES = L::ABYTES, TILE_B = L::TILE_B, BRING = L::BRING, RX = L::RX;
    constexpr int ZROW2 = (HRMAX - 2) * 128;                    // last two rows of either halo buffer: beyond the halo, from the zero page
    __shared__ __attribute__((aligned(16))) char smem[L::SMEM]; // the ONLY LDS object
    constexpr int NP = HRMAX / 8;                               // 1-KiB halo pieces (8 rows) per slab; loader w requests pieces 4 j + w
    constexpr int HPW = (NP + NWL - 1) / NWL;
    constexpr int NBW = BN / (8 * NWL);                         // 1-KiB weight pieces per loading wave per K-tile
    constexpr int HT0 = 4, PPT = (HPW + 2) / 3;                 // the next slab's halo pieces go out at taps 4, 5, 6 (older in the wave's queue than that
                                                                // slab's first weight tile, requested at tap 7: the counted wait for the tile covers them)
    static_assert(HRMAX % 8 == 0 && BN % (8 * NWL) == 0 && PPT <= 4, "conv_hdmaw: staging");
    constexpr bool RES = EPI == 1 || EPI == 4, BNB = EPI == 2 || EPI == 4;
    constexpr int NSIDE = EPI == 4 ? 2 : (EPI == 2 ? 1 : 0);    // side tensors of the drain (16-byte loads per chunk)
    // the drain: 16 passes of 16 rows (a loading wave takes four whole 256-byte rows per pass), passes 3 t .. 3 t + 2 at tap t < 4, 12 .. 15 at tap 4
    constexpr int DTAPS = 5;

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool loader = wave_all >= NWC;
    const int wave = loader ? wave_all - NWC : wave_all;        // role index inside its group
    const int W = a.W, H = a.H, C = a.C;
    const int ntn = a.K / BN;
    const int nslab = C / 64;                                   // (>= 2: the launcher's eligibility)

    // this workgroup's tiles: [first, first + cnt), consecutive ids share the M-tile (XCD-major workgroup order, as conv_hdmap_k)
    int first, cnt;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, rr = nwg & 7;
        const int p = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
        first = p * tpw;
        cnt = ntiles - first < tpw ? ntiles - first : tpw;
    }
    if (cnt <= 0) return;
    const int HR = BM + 2 * W + 2;
    const int hshift = W + 1;

    if (loader) {
        // ================================================= the loading waves ==================================================
        // Halo row hr of a tile with origin m0 holds input pixel m0 - (W + 1) + hr; rows outside the tensor read a clamped pixel (only
        // ever met by taps that the border select sends to the zero rows); pieces entirely past the halo come from the zero page (the
        // launcher guarantees that the last piece, which holds the ZERO ROWS, is one of them)
#ifdef LBC_HDMAW_LOADER_PRIO
        __builtin_amdgcn_s_setprio(LBC_HDMAW_LOADER_PRIO);
#endif
        const int prow = lane >> 3, pseg = lane & 7;
        const int arow0 = wave * 8 + prow;                                         // halo row of piece 4 j + wave: arow0 + 32 j (same swizzle term for every j)
        const unsigned aswz = (unsigned)((pseg ^ ((arow0 >> 1) & 7)) * 16);
        const unsigned zoff = (unsigned)((lane & 7) * 16);
        const char* xbytes = reinterpret_cast<const char*>(a.x);
        const char* zbytes = static_cast<const char*>(zero_page);
        auto issue_a = [&](const int m0x, const int slab, const int buf, const int j) {
            const int piece = 4 * j + wave;
            const bool pad = piece * 8 >= HR;                                      // wave-uniform
            int q = m0x - hshift + arow0 + 32 * j;
            q = q < 0 ? 0 : (q >= a.M ? a.M - 1 : q);
            const unsigned off = (unsigned)q * (unsigned)(2 * C) + aswz;
            const char* sbase = hdmaw_uniform_ptr(pad ? zbytes : xbytes + (size_t)(slab * 128));
            lds_dma16(sbase + (pad ? zoff : off), smem + buf * ABYTES + piece * 1024);
        };
        // halo pieces this wave requests at tap t (HT0 <= t < HT0 + 3)
        auto np_at = [&](const int t) {
            int n = 0;
            if (t >= HT0 && t < HT0 + 3)
                for (int q = 0; q < PPT; ++q) n += ((t - HT0) * PPT + q < HPW && 4 * ((t - HT0) * PPT + q) + wave < NP) ? 1 : 0;
            return n;
        };
        unsigned voffb[NBW];
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int row = (wave * NBW + j) * 8 + prow;
            voffb[j] = (unsigned)row * (unsigned)(18 * C) + (unsigned)((pseg ^ ((row >> 1) & 7)) * 16);
        }
        const char* wbytes = reinterpret_cast<const char*>(a.w);
        auto issue_b = [&](const int n0x, const int slab, const int tap, const int slot) {
            char* base = smem + BRING + slot * TILE_B;
            const char* wsrc = hdmaw_uniform_ptr(wbytes + ((size_t)n0x * (size_t)(18 * C) + (size_t)(2 * (tap * C + slab * 64))));
#pragma unroll
            for (int j = 0; j < NBW; ++j) lds_dma16(wsrc + voffb[j], base + (wave * NBW + j) * 1024);
        };
        auto wait_vm = [&](const int n) {
            switch (n) {
#define LBC_WV(N) case N: LBC_WAIT_VM(N); break;
                LBC_WV(1) LBC_WV(2) LBC_WV(3) LBC_WV(4) LBC_WV(5) LBC_WV(6) LBC_WV(7) LBC_WV(8) LBC_WV(9) LBC_WV(10) LBC_WV(11) LBC_WV(12)
                LBC_WV(13) LBC_WV(14) LBC_WV(15) LBC_WV(16) LBC_WV(17) LBC_WV(18) LBC_WV(19) LBC_WV(20) LBC_WV(21) LBC_WV(22) LBC_WV(23) LBC_WV(24)
                LBC_WV(25) LBC_WV(26) LBC_WV(27) LBC_WV(28) LBC_WV(29) LBC_WV(30) LBC_WV(31) LBC_WV(32)
#undef LBC_WV
                default: LBC_WAIT_VM(0); break;
            }
        };

        // ---- the drain of a dumped tile (origin dm0 / dn0, statistics row dmt, image in halo buffer dbuf + the spare region)
        __bf16* yout = static_cast<__bf16*>(a.y);
        const __bf16* by = BNB ? static_cast<const __bf16*>(a.bnb_y) : nullptr;
        const __bf16* bmask = EPI == 4 ? static_cast<const __bf16*>(a.bnb_mask) : nullptr;
        float* red = reinterpret_cast<float*>(smem + L::RED);                      // [2][2][BN]
        const int drow = wave * 4 + (lane >> 4), dck = lane & 15;                  // pass p: image row 16 p + drow, 16-byte chunk dck (channels 8 dck ..)
        int dm0 = 0, dn0 = 0, dmt = 0, dbuf = 0;
        f32x8 t1 = ParamVec<8>::splat(0.f), t2 = t1;
        f32x8 bsc = t1, bsh = t1, bmu = t1, biv = t1;
        bf16x8 yv[BNB ? 2 : 1][4], mv[EPI == 4 ? 2 : 1][4];
        auto pass_lo = [](const int t) { return 3 * t; };
        auto pass_n = [](const int t) { return t < 4 ? 3 : 4; };
        auto drain_begin = [&](const int m0x, const int n0x, const int mtx, const int bufx) {
            dm0 = m0x; dn0 = n0x; dmt = mtx; dbuf = bufx;
            t1 = ParamVec<8>::splat(0.f); t2 = t1;
            if constexpr (BNB) {
                const int c0 = dn0 + dck * 8;
                if constexpr (EPI == 2) { bsc = ParamVec<8>::ld(a.bnb_scale + c0); bsh = ParamVec<8>::ld(a.bnb_shift + c0); }
                bmu = ParamVec<8>::ld(a.bnb_mean + c0); biv = ParamVec<8>::ld(a.bnb_invstd + c0);
            }
        };
        // side chunks of the passes of drain tap t into register set `set`
        auto drain_side = [&](const int t, const int set) {
            if constexpr (BNB) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u < pass_n(t)) {
                        const int m = dm0 + 16 * (pass_lo(t) + u) + drow;
                        const unsigned o = (unsigned)(m < a.M ? m : 0) * (unsigned)a.K + (unsigned)(dn0 + dck * 8);
                        yv[set][u] = *reinterpret_cast<const bf16x8*>(by + o);
                        if constexpr (EPI == 4) mv[set][u] = *reinterpret_cast<const bf16x8*>(bmask + o);
                    }
            }
        };
        auto drain_tap = [&](const int t, const int set) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < pass_n(t)) {
                    const int row = 16 * (pass_lo(t) + u) + drow;
                    const int m = dm0 + row;
                    // (a wave's four rows of a pass sit on the same side of RX: wave-uniform select)
                    const int rbase = row < RX ? dbuf * ABYTES + row * 256 : L::SPARE + (row - RX) * 256;
                    bf16x8 ch = *reinterpret_cast<const bf16x8*>(smem + rbase + ((dck ^ (row & 7)) << 4));
                    f32x8 g = __builtin_convertvector(ch, f32x8);
                    if constexpr (BNB) {
                        // fused BatchNorm-backward reduce: mask the stored gradient with bn(y) > 0 (form 4: with the given ReLU output > 0), sum (g, g * xhat)
                        const f32x8 yf = __builtin_convertvector(yv[set][u], f32x8);
                        f32x8 z;
                        if constexpr (EPI == 4) z = __builtin_convertvector(mv[set][u], f32x8);
                        else z = yf * bsc + bsh;
#pragma unroll
                        for (int e = 0; e < 8; ++e) g[e] = z[e] > 0.f ? g[e] : 0.f;
                        ch = __builtin_convertvector(g, bf16x8);
                        if (m < a.M) { t1 += g; t2 += g * (yf - bmu) * biv; }
                    } else {
                        if (m < a.M) { t1 += g; t2 += g * g; }
                    }
                    if (m < a.M) *reinterpret_cast<bf16x8*>(yout + ((unsigned)m * (unsigned)a.K + (unsigned)(dn0 + dck * 8))) = ch;
                }
        };
        // The statistics row of the drained tile, three phases with a workgroup barrier between them: (0) lanes with the same chunk combine over
        // lane >> 4, loaders 0 / 1 write their sums; (1) loaders 2 / 3 add theirs; (2) loaders 0 / 1 add the two rows in a fixed order and store
        auto stats_phase = [&](const int ph) {
            if (!a.stats) return;
            if (ph == 0) {
#pragma unroll
                for (int off = 16; off < 64; off <<= 1)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { t1[e] += __shfl_xor(t1[e], off); t2[e] += __shfl_xor(t2[e], off); }
            }
            if (ph < 2) {
                if ((wave >> 1) == ph && lane < 16) {
                    float* r1 = red + ((wave & 1) * 2 + 0) * BN + lane * 8;
                    float* r2 = red + ((wave & 1) * 2 + 1) * BN + lane * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        r1[e] = ph ? r1[e] + t1[e] : t1[e];
                        r2[e] = ph ? r2[e] + t2[e] : t2[e];
                    }
                }
            } else if (wave < 2) {
                float* dst = a.stats + (size_t)(a.stat_row0 + dmt) * 2 * (size_t)a.K + (size_t)wave * (size_t)a.K + dn0;
#pragma unroll
                for (int h = 0; h < BN / 64; ++h) dst[lane + 64 * h] = red[(0 * 2 + wave) * BN + lane + 64 * h] + red[(1 * 2 + wave) * BN + lane + 64 * h];
            }
        };

        int tile = first;
        int mtile = tile / ntn, n0 = (tile - mtile * ntn) * BN, m0 = mtile * BM;
        int sg = 0;
        // prologue: the halo of slab 0 and the first two weight tiles; everything of K-tile 0 landed and visible behind the barrier
#pragma unroll
        for (int j = 0; j < HPW; ++j)
            if (4 * j + wave < NP) issue_a(m0, 0, 0, j);
        issue_b(n0, 0, 0, 0);
        issue_b(n0, 0, 1, 1);
        LBC_WAIT_VM(NBW);
        __builtin_amdgcn_s_barrier();
        LBC_PROF(unsigned long long p_req = 0, p_vm = 0, p_bar = 0; const unsigned long long p_t0 = __builtin_amdgcn_s_memtime(); unsigned long long p_a = p_t0;)
        int h_prev = 0;                                         // halo pieces requested in the previous K-tile (behind its weight tile in the queue)
        for (int it = 0; it < cnt; ++it) {
            const bool more = it + 1 < cnt;
            const int tilen = tile + 1;
            const int mtilen = tilen / ntn, n0n = (tilen - mtilen * ntn) * BN, m0n = mtilen * BM;
            const bool draining = it > 0;                       // the previous tile's image is drained under this tile's first slab
            int s_pre = draining ? NSIDE * pass_n(0) : 0;      // side loads requested in front of the first K-tile (at the dump barrier)
            auto slab_iter = [&](const int c, auto first_tag) {
                constexpr bool FIRST = decltype(first_tag)::value;      // the tile's first slab: the one that carries the drain
                const bool last = c + 1 == nslab;
                const bool follows = !last || more;             // another slab follows this one in the stream
                const int buf = sg & 1;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int islot = (t + 2) % 3;
                    const bool w2 = t + 2 < 9 || follows;       // K-tile k + 2 exists
                    int n_drain = 0;                            // VMEM operations of the drain in this K-tile
                    if constexpr (FIRST) {
                        if (draining) {
                            if (t < DTAPS) {
                                if (t + 1 < DTAPS) { drain_side(t + 1, (t + 1) & 1); n_drain += NSIDE * pass_n(t + 1); }
                                drain_tap(t, t & 1);
                                n_drain += pass_n(t);
                            } else if (t < DTAPS + 3) {
                                stats_phase(t - DTAPS);
                                if (t == DTAPS + 2 && a.stats && wave < 2) n_drain += BN / 64;
                            }
                        }
                    }
                    int h_here = 0;
#ifndef LBC_HDMAW_ABL_NODMA
                    // K-tile k + 2's weight tile -> ring slot (t + 2) % 3 (read last by K-tile k - 1: free since that K-tile's barrier)
                    if (w2) {
                        const int tt = t + 2 < 9 ? t + 2 : t - 7;
                        const int cc = t + 2 < 9 ? c : (last ? 0 : c + 1);
                        const int nn = (t + 2 < 9 || !last) ? n0 : n0n;
                        issue_b(nn, cc, tt, islot);
                    }
                    // the next slab's halo -> the other buffer (read last by the previous slab's last K-tile; the first slab of a tile: the
                    // buffer the previous tile was dumped into, drained by tap 3)
                    if (t >= HT0 && t < HT0 + 3 && follows) {
#pragma unroll
                        for (int q = 0; q < PPT; ++q) {
                            const int j = (t - HT0) * PPT + q;
                            if (j < HPW && 4 * j + wave < NP) {
                                if (!last) issue_a(m0, c + 1, buf ^ 1, j);
                                else issue_a(m0n, 0, buf ^ 1, j);
                            }
                        }
                        h_here = np_at(t);
                    }
#endif
                    LBC_PROF(const unsigned long long p_b = __builtin_amdgcn_s_memtime(); p_req += p_b - p_a;)
                    // The weight tile of K-tile k + 1 (requested during K-tile k - 1) has landed, this wave's pieces; requested after it and
                    // allowed to stay in flight: the halo pieces of K-tile k - 1, the side loads in front of the tile, this K-tile's drain
                    // operations, weight tile and halo pieces
                    wait_vm(w2 ? h_prev + s_pre + n_drain + NBW + h_here : 0);
                    LBC_PROF(const unsigned long long p_c = __builtin_amdgcn_s_memtime(); p_vm += p_c - p_b;)
                    if constexpr (FIRST) LBC_WAIT_LGKM0();      // (the drain's LDS reads / the statistics rows' LDS writes)
                    __builtin_amdgcn_s_barrier();
                    LBC_PROF(p_a = __builtin_amdgcn_s_memtime(); p_bar += p_a - p_c;)
                    h_prev = h_here;
                    s_pre = 0;
                }
                ++sg;
            };
            slab_iter(0, std::true_type{});
            for (int c = 1; c < nslab; ++c) slab_iter(c, std::false_type{});
            // the multiplying waves dump this tile into the halo buffer of its last slab (+ the spare region); behind their barrier: this wave's
            // first side loads of the drain go out
            __builtin_amdgcn_s_barrier();
            drain_begin(m0, n0, mtile, (sg - 1) & 1);
            if (more) drain_side(0, 0);
            tile = tilen; mtile = mtilen; n0 = n0n; m0 = m0n;
        }
        // the last tile: nothing left to hide the drain under
        drain_side(0, 0);
        drain_side(1, 1); drain_tap(0, 0);
        drain_side(2, 0); drain_tap(1, 1);
        drain_side(3, 1); drain_tap(2, 0);
        drain_side(4, 0); drain_tap(3, 1);
        drain_tap(4, 0);
        static_assert(DTAPS == 5, "conv_hdmaw: drain schedule");
        if (a.stats) {
            stats_phase(0);
            LBC_WAIT_LGKM0(); __builtin_amdgcn_s_barrier();
            stats_phase(1);
            LBC_WAIT_LGKM0(); __builtin_amdgcn_s_barrier();
            stats_phase(2);
        }
        LBC_PROF(if (lane == 0) { unsigned long long* o = g_hdmaw_prof + (blockIdx.x * 8 + wave_all) * 4; o[0] = p_req; o[1] = p_vm; o[2] = p_bar; o[3] = __builtin_amdgcn_s_memtime() - p_t0; })
        return;
    }

    // ===================================================== the multiplying waves =====================================================
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;
    // A fragment of depth step g sits in 16-byte slot (2g + kh) ^ f(row) of its 128-byte LDS row, f(row) = (row >> 1) & 7 (the swizzle of
    // the DMA source): address = (base | (kh ^ f) << 4) ^ 32 g -- one v_xor per read
    const int baddr = (BRING + (wn * WTN + l31) * 128) | ((kh ^ ((l31 >> 1) & 7)) << 4);
    int rowc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) rowc[i] = hshift + wm * WTM + i * 32 + l31;
    auto tap_mask = [&](const int m0x, int (&mask)[MT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0x + wm * WTM + i * 32 + l31;
            int bits = 0;
            if (m < a.M) {
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
            mask[i] = bits;
        }
    };
    int amask[MT], amaskn[MT];

    // TRANSPOSED accumulators: acc[i][j] = W_j (32 output channels) x X_i^T (32 pixels): lane (l31, kh) holds pixel i * 32 + l31 of the wave tile and,
    // in registers 4q .. 4q + 3, the channels j * 32 + 8q + 4kh .. + 3
    f32x16 acc[MT][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    // per (tap, 32-row block): LDS address of the lane's depth-step-0 fragment in halo buffer `buf` (its halo row, or -- border lanes -- the
    // zero at its own row's position inside the 256-byte bank period: conflict-free either way, conv_hdmap.hpp)
    int aaddr[MT];
    auto tap_addr = [&](const int tap, const int buf, const int (&mask)[MT]) {
        const int r = tap / 3, s = tap - 3 * r;
        const int off = MODE == 0 ? (r - 1) * W + (s - 1) : (1 - r) * W + (1 - s);
        const int abuf = buf * ABYTES;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int hr = rowc[i] + off;
            const int val = (hr << 7) | ((kh ^ ((hr >> 1) & 7)) << 4), zval = ZROW2 | (val & 255);
            const int m = -((mask[i] >> tap) & 1);
            aaddr[i] = abuf + (((val ^ zval) & m) ^ zval);
        }
    };

    bf16x8 fa[2][MT], fb[2][NT];            // two register sets: depth step g computes from set g & 1 while set (g + 1) & 1 is read
    // the fragment reads are inline asm with a hand-placed lgkmcnt wait (conv_hdmap.hpp: hipcc's own wait insertion does not see them)
    static_assert(3 * TILE_B + (NT - 1) * 4096 < 65536, "conv_hdmaw: the ring slot is an immediate offset of the read");
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
#ifdef LBC_HIP_EMULATED_FOR_TESTS
#define LBC_RD1(DST, ADDR, OFF) DST = *reinterpret_cast<const bf16x8*>(smem + (ADDR) + (OFF))
#define LBC_USE(SET) do { } while (0)
#elif defined(LBC_HDMAW_ABL_NOREAD)
#define LBC_RD1(DST, ADDR, OFF) asm volatile("" : "=v"(DST) : "v"(lds0 + (unsigned)(ADDR)), "n"(OFF))
#define LBC_USE(SET) do { } while (0)
#else
#define LBC_RD1(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(lds0 + (unsigned)(ADDR)), "n"(OFF))
#define LBC_USE(SET)                                                                                                             \
    do {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[SET][i]));                                      \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(fb[SET][j]));                                      \
    } while (0)
#endif
    // pair P of the six fragment reads of (ring slot, depth step G) into register set SET, in the order the MFMAs need them:
    // (weights 0, rows 0) (weights 1, rows 1) (rows 2, rows 3)
#define LBC_RDP(SLOT, G, SET, P)                                                                                                 \
    do {                                                                                                                         \
        if constexpr ((P) == 0) { LBC_RD1(fb[SET][0], baddr ^ (32 * (G)), (SLOT) * TILE_B); LBC_RD1(fa[SET][0], aaddr[0] ^ (32 * (G)), 0); }          \
        else if constexpr ((P) == 1) { LBC_RD1(fb[SET][1], baddr ^ (32 * (G)), (SLOT) * TILE_B + 4096); LBC_RD1(fa[SET][1], aaddr[1] ^ (32 * (G)), 0); } \
        else { LBC_RD1(fa[SET][2], aaddr[2] ^ (32 * (G)), 0); LBC_RD1(fa[SET][3], aaddr[3] ^ (32 * (G)), 0); }                  \
    } while (0)
#ifdef LBC_HDMAW_ABL_NOMFMA
#define LBC_MM1(SET, I, J) do { } while (0)
#else
#define LBC_MM1(SET, I, J) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[SET][J], fa[SET][I], acc[I][J], 0, 0, 0)   /* weights x pixels^T */
#endif
#define LBC_PIN() __builtin_amdgcn_sched_barrier(0)

    // ---- the dump of a finished tile: residual (f32, before rounding; 8-byte pieces one 32-row block ahead), folded-BatchNorm affine / bias (eval
    //      mode), ReLU, rounding, 8-byte LDS writes -- image row = tile row, 16-byte slot = (channel / 8) ^ (row & 7), half = kh
    const __bf16* resid = RES ? static_cast<const __bf16*>(a.resid) : nullptr;
    auto dump = [&](const int m0x, const int n0x, const int bufx) {
        const int colw = n0x + wn * WTN;
        if (a.post_scale || a.bias) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = colw + j * 32 + 8 * q + 4 * kh;
                    f32x4 sc = ParamVec<4>::splat(1.f), sh = ParamVec<4>::splat(0.f);
                    if (a.post_scale) { sc = ParamVec<4>::ld(a.post_scale + c0); sh = ParamVec<4>::ld(a.post_shift + c0); }
                    if (a.bias) sh += ParamVec<4>::ld(a.bias + c0);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = acc[i][j][4 * q + e] * sc[e] + sh[e];
                }
        }
        bf16x4 rv[RES ? 2 : 1][NT * 4];
        auto resid_block = [&](const int i, const int set) {
            const int m = m0x + wm * WTM + i * 32 + l31;
            const unsigned ob = (unsigned)(m < a.M ? m : 0) * (unsigned)a.K + (unsigned)(colw + 4 * kh);
#pragma unroll
            for (int u = 0; u < NT * 4; ++u) rv[set][u] = *reinterpret_cast<const bf16x4*>(resid + ob + (unsigned)(8 * u));
        };
        if constexpr (RES) resid_block(0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if constexpr (RES) { if (i + 1 < MT) resid_block(i + 1, (i + 1) & 1); }
            const int row = wm * WTM + i * 32 + l31;
            const int rbase = (row < RX ? bufx * ABYTES + row * 256 : L::SPARE + (row - RX) * 256) + kh * 8;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                    if constexpr (RES) v += __builtin_convertvector(rv[i & 1][j * 4 + q], f32x4);
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    *reinterpret_cast<bf16x4*>(smem + rbase + (((wn * 8 + j * 4 + q) ^ (l31 & 7)) << 4)) = __builtin_convertvector(v, bf16x4);
                }
        }
    };

    // ---- the tile stream
    int tile = first;
    int mtile = tile / ntn, n0 = (tile - mtile * ntn) * BN, m0 = mtile * BM;
    int sg = 0;                             // slabs consumed so far: halo buffer sg & 1
    tap_mask(m0, amask);
    __builtin_amdgcn_s_barrier();           // the prologue's: K-tile 0 landed and visible
    tap_addr(0, 0, amask);
    LBC_RDP(0, 0, 0, 0); LBC_RDP(0, 0, 0, 1); LBC_RDP(0, 0, 0, 2);
    LBC_PROF(unsigned long long p_seg = 0, p_bar = 0, p_epi = 0; const unsigned long long p_t0 = __builtin_amdgcn_s_memtime(); unsigned long long p_a = p_t0, p_b = p_t0, p_bprev = p_t0;)
    for (int it = 0; it < cnt; ++it) {
        const bool more = it + 1 < cnt;
        const int tilen = tile + 1;
        const int mtilen = tilen / ntn, n0n = (tilen - mtilen * ntn) * BN, m0n = mtilen * BM;
        if (more) tap_mask(m0n, amaskn);

        // One slab = nine K-tiles, taps unrolled.  LAST: the tile's last slab -- what follows in the stream is the next tile (if any).
        auto slab_body = [&](auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            const bool follows = !LAST || more;
            // (row base, XOR term) of a tap do not depend on the slab: left alone, the compiler hoists all 9 x MT pairs out of the loops
#pragma unroll
            for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(rowc[i]), "+v"(amask[i]));
            const int buf = sg & 1;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int slot = t % 3, nslot = (t + 1) % 3;
                const bool has_next = t < 8 || follows;
                // depth steps 0 .. 2: the reads of step g + 1 ride in the first three MFMA gaps of step g
#pragma unroll
                for (int g = 0; g + 1 < KS; ++g) {
                    LBC_WAIT_LGKM0();                                    // set g & 1 is in: its reads went out five to seven MFMAs ago
                    LBC_USE(g & 1);
                    LBC_MM1(g & 1, 0, 0); LBC_PIN(); LBC_RDP(slot, g + 1, (g + 1) & 1, 0); LBC_PIN();
                    LBC_MM1(g & 1, 0, 1); LBC_PIN(); LBC_RDP(slot, g + 1, (g + 1) & 1, 1); LBC_PIN();
                    LBC_MM1(g & 1, 1, 0); LBC_PIN(); LBC_RDP(slot, g + 1, (g + 1) & 1, 2); LBC_PIN();
                    LBC_MM1(g & 1, 1, 1);
                    // the reads of the last depth step are out: the addresses are free for the next K-tile's (tap, slab, tile)
                    if (g == KS - 2 && has_next) {
                        if (t < 8) tap_addr(t + 1, buf, amask);
                        else if (!LAST) tap_addr(0, buf ^ 1, amask);
                        else tap_addr(0, buf ^ 1, amaskn);
                    }
                    LBC_MM1(g & 1, 2, 0); LBC_MM1(g & 1, 2, 1); LBC_MM1(g & 1, 3, 0); LBC_MM1(g & 1, 3, 1);
                    LBC_PIN();
                }
                // every read of this K-tile has returned; behind the barrier the loaders' pieces of K-tile k + 1 are visible and this
                // K-tile's ring slot / (last tap) halo buffer is free
                LBC_WAIT_LGKM0();
                // (stamps a / b of the PREVIOUS K-tile have returned by now: consumed here, behind the wait that is there anyway)
                LBC_PROF(p_bar += p_b - p_a; p_seg += p_a - p_bprev; p_bprev = p_b; p_a = __builtin_amdgcn_s_memtime();)
                __builtin_amdgcn_s_barrier();
                LBC_PROF(p_b = __builtin_amdgcn_s_memtime();)
                LBC_PIN();
                // the last depth step, with depth step 0 of the next K-tile in its gaps
                LBC_USE((KS - 1) & 1);
                LBC_MM1((KS - 1) & 1, 0, 0); LBC_PIN(); if (has_next) LBC_RDP(nslot, 0, 0, 0); LBC_PIN();
                LBC_MM1((KS - 1) & 1, 0, 1); LBC_PIN(); if (has_next) LBC_RDP(nslot, 0, 0, 1); LBC_PIN();
                LBC_MM1((KS - 1) & 1, 1, 0); LBC_PIN(); if (has_next) LBC_RDP(nslot, 0, 0, 2); LBC_PIN();
                LBC_MM1((KS - 1) & 1, 1, 1); LBC_MM1((KS - 1) & 1, 2, 0); LBC_MM1((KS - 1) & 1, 2, 1); LBC_MM1((KS - 1) & 1, 3, 0); LBC_MM1((KS - 1) & 1, 3, 1);
                LBC_PIN();
            }
            ++sg;
        };
        for (int c = 0; c + 1 < nslab; ++c) slab_body(std::false_type{});
        slab_body(std::true_type{});

        // the tile's image -> the halo buffer its last slab has left (every read of it returned in front of the last K-tile barrier) + the spare
        // region (drained since the first slab of this tile); visible to the loaders behind the barrier
        LBC_PROF(const unsigned long long p_e0 = __builtin_amdgcn_s_memtime();)
        dump(m0, n0, (sg - 1) & 1);
        zero_acc();
        LBC_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        LBC_PROF(p_epi += __builtin_amdgcn_s_memtime() - p_e0;)
        tile = tilen; mtile = mtilen; n0 = n0n; m0 = m0n;
#pragma unroll
        for (int i = 0; i < MT; ++i) amask[i] = amaskn[i];
    }
    // the loaders drain the last tile; its statistics row takes two more workgroup barriers
    if (a.stats) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
    LBC_PROF(if (lane == 0) { unsigned long long* o = g_hdmaw_prof + (blockIdx.x * 8 + wave_all) * 4; o[0] = p_seg; o[1] = p_bar; o[2] = p_epi; o[3] = __builtin_amdgcn_s_memtime() - p_t0; })
#undef LBC_RD1
#undef LBC_USE
#undef LBC_RDP
#undef LBC_MM1
#undef LBC_PIN
}

// launches the instantiation for (mode, epilogue form) of one halo size
template <int BM, int BN, int HRMAX>
int conv_hdmaw_launch_shape(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, dim3 grid, hipStream_t s)
{
    const int epi = a.bnb_y ? (a.bnb_mask ? 4 : 2) : (a.resid ? 1 : 0);
    LBC_REQUIRE(!a.pre_scale && (mode == 0 || mode == 1) && a.C >= 128 && BM + 2 * a.W + 2 <= HRMAX - 8, "conv_hdmaw: forward / input gradient of >= 128 channels without BatchNorm-on-load only");
#define LBC_HW(MODEv, EPIv) hipLaunchKernelGGL((conv_hdmaw_k<BM, BN, HRMAX, MODEv, EPIv>), grid, dim3(512), 0, s, a, zero, ntiles, tpw)
    if (mode == 0) {
        LBC_REQUIRE(epi != 2 && epi != 4, "conv_hdmaw: the fused BatchNorm-backward reduce belongs to input-gradient launches");
        if (epi == 1) LBC_HW(0, 1); else LBC_HW(0, 0);
    } else {
        LBC_REQUIRE(epi != 4 || a.resid, "conv_hdmaw: the tensor-masked BatchNorm-backward reduce is the residual form's (IgemmArgs::bnb_mask)");
        if (epi == 4) LBC_HW(1, 4); else if (epi == 2) LBC_HW(1, 2); else if (epi == 1) LBC_HW(1, 1); else LBC_HW(1, 0);
    }
#undef LBC_HW
    return lbc_check_launch("conv_hdmaw");
}

}  // namespace
