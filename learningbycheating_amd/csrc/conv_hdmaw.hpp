// WAVE-SPECIALISED form of the persistent halo-staged convolution (conv_hdmap.hpp explains the staging and the tile stream): 3x3 / stride-1 /
// pad-1 forward and input gradient on bf16 tensors, 256 x 128 output tiles.  Reference arithmetic bird_view/models/resnet.py:15-22,38-54
// (BasicBlock conv1 / conv2) and autograd.
//
// Why (round 6).  conv_hdmap_k's eight waves all do everything: two waves per SIMD, each a 64 x 64 wave tile, each requesting its share of
// the weight tiles and halo pieces by LDS-DMA, one workgroup barrier per K-tile.  Its in-kernel stamps (round 3) put the
// K-tile period at ~1490 cycles for 1024 cycles of MFMA per SIMD: a SIMD's two waves walk the depth steps in lock-step behind the
// barrier, the older one finishes early and waits ~390 cycles for the younger, every DMA request stalls the requesting wave's issue for
// 60 - 180 cycles in the middle of its MFMA stream, and the LDS pipe carries one ds_read_b128 per MFMA.  Removing the LDS bank conflicts
// (13 % -> 1 % of the LDS cycles, profiles/r06_call1_*) did not move the launch: the loss is the structure, not the LDS rate.  Here
//   * waves 0-3 MULTIPLY: one per SIMD (nothing shares its matrix pipe, no intra-SIMD skew at the barrier), a 128 x 64 wave tile each
//     (MT = 4, NT = 2: 6 fragment reads per 8 MFMAs instead of 4 per 4; 128 accumulator registers), no VMEM instruction in the K loop;
//   * waves 4-7 LOAD: one per SIMD next to a multiplying wave, they request every weight tile and halo piece (same ring, same slots,
//     same counted-vmcnt schedule as conv_hdmap_k, four pieces of a weight tile + at most two halo pieces per wave and K-tile) and
//     otherwise sit in the K-tile barrier -- the DMA issue stalls are theirs, the multiplying wave's stream is ds_read / v_xor / MFMA;
//   * the barrier stays one per K-tile (in front of the last depth step): the loaders are always there first, the four multiplying
//     waves run on four different SIMDs at the same pace;
//   * epilogue: the multiplying waves' wave-private copy-out of conv_hdmap_epi.hpp (residual / side chunks one step ahead: a 128 x 64
//     wave tile has no 128 registers to spare).
// 512 threads, <= 256 registers each (the loaders get the same allocation: two waves per SIMD is the register file).
#pragma once
#include <type_traits>
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include "conv_lds_dma.hpp"

namespace {

#define LBC_SG(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)

__device__ __forceinline__ const char* hdmaw_uniform_ptr(const char* p)
{
    const unsigned long long v = (unsigned long long)(size_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char*>((size_t)(((unsigned long long)hi << 32) | lo));
}

// LBC_HDMAW_PROF (scripts/probe/hdmaw_prof.hip only): per-wave s_memtime sums -- multiplying waves: [0] K-tile segments between barriers, [1] inside the
// K-tile barrier, [2] epilogue, [3] first stamp .. last stamp; loading waves: [0] requests, [1] counted vmcnt wait, [2] barrier, [3] total
// Timing-experiment builds of the same probe: LBC_HDMAW_ABL_NOMFMA / _NOREAD / _NODMA / _NOBAR drop the MFMAs / the fragment reads / the loaders' requests
// of the main loop / the K-tile barrier (results are then wrong; only the time is looked at).
#ifdef LBC_HDMAW_PROF
__device__ unsigned long long g_hdmaw_prof[256 * 8 * 4];
#define LBC_PROF(...) __VA_ARGS__
#else
#define LBC_PROF(...)
#endif

template <int BM, int BN, int HRMAX, int SROWS>
constexpr int hdmaw_lds_bytes() { return 2 * HRMAX * 128 + 3 * BN * 128 + 4 * SROWS * 128 + 2 * 2 * BN * 4; }

// Epilogue of the multiplying waves.  The accumulators are TRANSPOSED: acc[i][j] = W_j (32 output channels) x X_i^T (32 pixels) -- the weight
// fragment is the MFMA's A operand -- so lane (l31, kh) holds ONE pixel (row i * 32 + l31 of the wave tile) and, in registers 4q .. 4q + 3,
// the four consecutive channels j * 32 + 8q + 4kh ..: one v_cvt_pk pair and one 8-byte LDS write per register quad (the untransposed
// layout of conv_hdmap_epi.hpp holds one CHANNEL per lane: 128 two-byte LDS writes per lane and tile, and per-element statistics
// arithmetic, measured at 8.6 k cycles per tile with the matrix pipe idle -- profiles/r06_call4_*).  Per (i, j) block: 32 rows x 64 bytes
// staged in the wave's own LDS (chunk slot XOR-ed with (row >> 1) & 3: two-way on the 8-byte writes, conflict-free 16-byte reads), read back
// as 16-byte chunks (lane -> row lane / 4 + 16 p, chunk lane % 4), and in that CHUNK PHASE: the per-channel sums, the store.
//   * statistics (forms 0 / 1): (sum, sum of squares) of the STORED, bf16-rounded value -- the statistics of the tensor the following
//     BatchNorm pass normalises (conv_hdmap_k sums the f32 value before rounding: the two differ by the mean rounding error, ~1e-5 relative);
//   * forms 2 / 4: the fused BatchNorm-backward sums exactly as conv_hdmap_epi.hpp (mask, sum g, sum g * xhat), side chunks one block ahead;
//   * residual (forms 1 / 4): added to the f32 accumulator before rounding, fetched as 8-byte pieces one block ahead;
//   * folded-BatchNorm affine / bias (eval mode): a pass over the accumulators before the blocks.
// SDB: the staging block double-buffered (the writes of block s + 1 need not wait for the read-back of block s).
template <int BN, int MODE, int EPI, int MT, int NT, bool SDB>
__device__ __forceinline__ void hdmaw_tile_epilogue(const IgemmArgs& a, f32x16 (&acc)[1][MT][NT], char* const stg, float* const red, const int wm, const int wn,
                                                    const int lane, const int tid, const int m0, const int n0, const int mtile)
{
    constexpr int WTM = MT * 32, WTN = NT * 32, WMv = 2;
    constexpr bool RES = EPI == 1 || EPI == 4, BNB = EPI == 2 || EPI == 4;
    const int l31 = lane & 31, kh = lane >> 5;
    __bf16* yout = static_cast<__bf16*>(a.y);
    const __bf16* resid = RES ? static_cast<const __bf16*>(a.resid) : nullptr;
    const __bf16* by = BNB ? static_cast<const __bf16*>(a.bnb_y) : nullptr;
    const __bf16* bmask = EPI == 4 ? static_cast<const __bf16*>(a.bnb_mask) : nullptr;
    const int colw = n0 + wn * WTN;                     // first column of this wave
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
                    for (int e = 0; e < 4; ++e) acc[0][i][j][4 * q + e] = acc[0][i][j][4 * q + e] * sc[e] + sh[e];
            }
    }
    // write role: row l31, 16-byte slot (q ^ swizzle), half kh; read-back role: rows crow + 16 p, chunk cck
    const int wbase = l31 * 64 + kh * 8, wswz = (l31 >> 1) & 3;
    const int crow = lane >> 2, cck = lane & 3;
    const int rbase = crow * 64 + ((cck ^ ((crow >> 1) & 3)) << 4);      // (rows crow and crow + 16 share the swizzle term)
    constexpr int NSTEP = MT * NT;                      // blocks, j-major: block s = (i = s % MT, j = s / MT)
    bf16x4 rv[RES ? 2 : 1][4];
    auto resid_block = [&](const int s, const int set) {
        const int i = s % MT, j = s / MT;
        const int m = m0 + wm * WTM + i * 32 + l31;
        const unsigned ob = (unsigned)(m < a.M ? m : 0) * (unsigned)a.K + (unsigned)(colw + j * 32 + 4 * kh);
#pragma unroll
        for (int q = 0; q < 4; ++q) rv[set][q] = *reinterpret_cast<const bf16x4*>(resid + ob + (unsigned)(8 * q));
    };
    bf16x8 yv[BNB ? 2 : 1][2], mv[EPI == 4 ? 2 : 1][2];
    auto side_block = [&](const int s, const int set) {
        const int i = s % MT, j = s / MT;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int m = m0 + wm * WTM + i * 32 + crow + 16 * p;
            const unsigned o = (unsigned)(m < a.M ? m : 0) * (unsigned)a.K + (unsigned)(colw + j * 32 + cck * 8);
            yv[set][p] = *reinterpret_cast<const bf16x8*>(by + o);
            if constexpr (EPI == 4) mv[set][p] = *reinterpret_cast<const bf16x8*>(bmask + o);
        }
    };
    if constexpr (RES) resid_block(0, 0);
    if constexpr (BNB) side_block(0, 0);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        f32x8 t1 = ParamVec<8>::splat(0.f), t2 = t1;
        f32x8 bsc = t1, bsh = t1, bmu = t1, biv = t1;
        if constexpr (BNB) {
            const int c0 = colw + j * 32 + cck * 8;
            if constexpr (EPI == 2) { bsc = ParamVec<8>::ld(a.bnb_scale + c0); bsh = ParamVec<8>::ld(a.bnb_shift + c0); }
            bmu = ParamVec<8>::ld(a.bnb_mean + c0); biv = ParamVec<8>::ld(a.bnb_invstd + c0);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int s = j * MT + i;
            char* const sb = stg + (SDB ? (s & 1) * 2048 : 0);
            // (every load of block s + 1 is requested before the stores of block s: the wait for it covers stores two blocks old)
            if constexpr (RES) { if (s + 1 < NSTEP) resid_block(s + 1, (s + 1) & 1); }
            if constexpr (BNB) { if (s + 1 < NSTEP) side_block(s + 1, (s + 1) & 1); }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[0][i][j][4 * q + e];
                if constexpr (RES) v += __builtin_convertvector(rv[s & 1][q], f32x4);
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                *reinterpret_cast<bf16x4*>(sb + wbase + ((q ^ wswz) << 4)) = __builtin_convertvector(v, bf16x4);
            }
            // (LDS operations of one wave execute in order: its reads below see its writes above; wave_barrier emits nothing, it pins the
            //  order for the compiler -- and for the CPU emulator, whose lanes are fibers)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int m = m0 + wm * WTM + i * 32 + crow + 16 * p;
                bf16x8 ch = *reinterpret_cast<const bf16x8*>(sb + rbase + p * 1024);
                f32x8 g = __builtin_convertvector(ch, f32x8);
                if constexpr (BNB) {
                    const f32x8 yf = __builtin_convertvector(yv[s & 1][p], f32x8);
                    f32x8 z;
                    if constexpr (EPI == 4) z = __builtin_convertvector(mv[s & 1][p], f32x8);
                    else z = yf * bsc + bsh;
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] = z[e] > 0.f ? g[e] : 0.f;
                    ch = __builtin_convertvector(g, bf16x8);
                    if (m < a.M) { t1 += g; t2 += g * (yf - bmu) * biv; }
                } else {
                    if (m < a.M) { t1 += g; t2 += g * g; }
                }
                if (m < a.M) *reinterpret_cast<bf16x8*>(yout + ((unsigned)m * (unsigned)a.K + (unsigned)(colw + j * 32 + cck * 8))) = ch;
            }
            if constexpr (!SDB) __builtin_amdgcn_wave_barrier();
        }
        if (a.stats) {
            // lanes with the same chunk (lane % 4) hold partial sums of the same 8 channels: combine over lane / 4
#pragma unroll
            for (int off = 4; off < 64; off <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) { t1[e] += __shfl_xor(t1[e], off); t2[e] += __shfl_xor(t2[e], off); }
            if (lane < 4) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[(wm * 2 + 0) * BN + wn * WTN + j * 32 + lane * 8 + e] = t1[e];
                    red[(wm * 2 + 1) * BN + wn * WTN + j * 32 + lane * 8 + e] = t2[e];
                }
            }
        }
    }
    if (a.stats) {
        LBC_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        if (tid < BN) {
            float u1 = 0.f, u2 = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WMv; ++w2) { u1 += red[(w2 * 2 + 0) * BN + tid]; u2 += red[(w2 * 2 + 1) * BN + tid]; }
            float* dst = a.stats + (size_t)(a.stat_row0 + mtile) * 2 * (size_t)a.K;
            dst[n0 + tid] = u1;
            dst[a.K + n0 + tid] = u2;
        }
        // (the next write of `red` lies behind at least the nine K-tile barriers of the next tile)
    }
}

// MODE 0 forward, 1 input gradient (flipped taps); EPI 0 / 1 / 2 / 4 as conv_hdmap_epi.hpp
template <int BM, int BN, int HRMAX, int SROWS, int MODE, int EPI>
__global__ __launch_bounds__(512, 2) void conv_hdmaw_k(IgemmArgs a, const void* zero_page, const int ntiles, const int tpw)
{
    constexpr int WM = 2, WN = 2, NWC = WM * WN, NWL = 4;       // multiplying waves (2 x 2), loading waves
    constexpr int WTM = BM / WM, WTN = BN / WN;                 // per-wave output tile
    constexpr int MT = WTM / 32, NT = WTN / 32;
    static_assert(MT == 4 && NT == 2 && (MODE == 0 || MODE == 1) && (EPI == 0 || EPI == 1 || EPI == 2 || EPI == 4), "conv_hdmaw: wave tiling / forms");
    static_assert(HRMAX % (8 * NWL) == 0 && BN % (8 * NWL) == 0 && (SROWS == 16 || SROWS == 32), "conv_hdmaw: staging");
    constexpr int NB = 3;                                       // weight ring depth: slot of K-tile (slab, tap) = tap % 3
    constexpr int KS = 4;                                       // depth steps of 16 channels per K-tile
    constexpr int ABYTES = HRMAX * 128;                         // one halo buffer: HRMAX rows x 64 channels
    constexpr int TILE_B = BN * 128;
    constexpr int BRING = 2 * ABYTES;
    constexpr bool SDB = SROWS == 32;                           // SROWS: staged 64-byte rows per multiplying wave -- one 32 x 32 block (16) or two (32)
    constexpr int STG = BRING + NB * TILE_B;                    // wave-private staging of the multiplying waves: NWC x SROWS x 128 bytes
    constexpr int RED = STG + NWC * SROWS * 128;                // [WM][2][BN] floats
    constexpr int SMEM = RED + WM * 2 * BN * 4;
    static_assert(SMEM == hdmaw_lds_bytes<BM, BN, HRMAX, SROWS>() && SMEM <= 160 * 1024, "conv_hdmaw: LDS");
    constexpr int ZROW2 = (HRMAX - 2) * 128;                    // last two rows of either halo buffer: beyond the halo, from the zero page
    __shared__ __attribute__((aligned(16))) char smem[SMEM];    // the ONLY LDS object
    constexpr int HPW = HRMAX / (8 * NWL);                      // 1-KiB halo pieces (8 rows) per loading wave per slab
    constexpr int NBW = BN / (8 * NWL);                         // 1-KiB weight pieces per loading wave per K-tile
    constexpr int ATAPS = 7;                                    // taps of a slab whose K-tile may carry halo pieces of the next slab (they must be
                                                                // older in the wave's queue than that slab's first weight tile, requested at tap 7)
    constexpr int PPT = (HPW + ATAPS - 1) / ATAPS;
    static_assert(PPT >= 1 && PPT <= 2 && NBW + 2 * PPT <= 24, "conv_hdmaw: halo pieces per tap / counted waits");

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool loader = wave_all >= NWC;
    const int wave = loader ? wave_all - NWC : wave_all;        // role index inside its group
    const int W = a.W, H = a.H, C = a.C;
    const int ntn = a.K / BN;
    const int nslab = C / 64;

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
        const int prow = lane >> 3, pseg = lane & 7;
        const int arow0 = wave * HPW * 8 + prow;                                   // halo row of piece j: arow0 + 8 j
        const unsigned aswz[2] = {(unsigned)((pseg ^ ((arow0 >> 1) & 7)) * 16), (unsigned)((pseg ^ (((arow0 >> 1) + 4) & 7)) * 16)};   // j even / odd
        const unsigned zoff = (unsigned)((lane & 7) * 16);
        const char* xbytes = reinterpret_cast<const char*>(a.x);
        const char* zbytes = static_cast<const char*>(zero_page);
        auto issue_a = [&](const int m0x, const int slab, const int buf, const int j) {
            const bool pad = (wave * HPW + j) * 8 >= HR;                           // wave-uniform
            int q = m0x - hshift + arow0 + 8 * j;
            q = q < 0 ? 0 : (q >= a.M ? a.M - 1 : q);
            const unsigned off = (unsigned)q * (unsigned)(2 * C) + aswz[j & 1];
            const char* sbase = hdmaw_uniform_ptr(pad ? zbytes : xbytes + (size_t)(slab * 128));
            lds_dma16(sbase + (pad ? zoff : off), smem + buf * ABYTES + (wave * HPW + j) * 1024);
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
#undef LBC_WV
                default: LBC_WAIT_VM(0); break;
            }
        };
        int tile = first;
        int mtile = tile / ntn, n0 = (tile - mtile * ntn) * BN, m0 = mtile * BM;
        int sg = 0;
        // prologue: the halo of slab 0 and the first two weight tiles; everything of K-tile 0 landed and visible behind the barrier
#pragma unroll
        for (int j = 0; j < HPW; ++j) issue_a(m0, 0, 0, j);
        issue_b(n0, 0, 0, 0);
        issue_b(n0, 0, 1, 1);
        LBC_WAIT_VM(NBW);
        __builtin_amdgcn_s_barrier();
        LBC_PROF(unsigned long long p_req = 0, p_vm = 0, p_bar = 0; const unsigned long long p_t0 = __builtin_amdgcn_s_memtime(); unsigned long long p_a = p_t0;)
        for (int it = 0; it < cnt; ++it) {
            const bool more = it + 1 < cnt;
            const int tilen = tile + 1;
            const int mtilen = tilen / ntn, n0n = (tilen - mtilen * ntn) * BN, m0n = mtilen * BM;
            for (int c = 0; c < nslab; ++c) {
                const bool last = c + 1 == nslab;
                const bool follows = !last || more;                 // another slab follows this one in the stream
                const int buf = sg & 1;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int islot = (t + 2) % 3;
                    const bool w2 = t + 2 < 9 || follows;           // K-tile k + 2 exists
                    const int np_here = (t * PPT < HPW ? (HPW - t * PPT < PPT ? HPW - t * PPT : PPT) : 0);
                    const int np_prev = t >= 1 ? ((t - 1) * PPT < HPW ? (HPW - (t - 1) * PPT < PPT ? HPW - (t - 1) * PPT : PPT) : 0) : 0;
                    // K-tile k + 2's weight tile -> ring slot (t + 2) % 3 (read last by K-tile k - 1: free since that K-tile's barrier)
#ifndef LBC_HDMAW_ABL_NODMA
                    if (w2) {
                        const int tt = t + 2 < 9 ? t + 2 : t - 7;
                        const int cc = t + 2 < 9 ? c : (last ? 0 : c + 1);
                        const int nn = (t + 2 < 9 || !last) ? n0 : n0n;
                        issue_b(nn, cc, tt, islot);
                    }
                    // the next slab's halo -> the other buffer (read last by the previous slab's last K-tile)
                    if (np_here > 0 && follows) {
#pragma unroll
                        for (int q = 0; q < PPT; ++q)
                            if (t * PPT + q < HPW) {
                                if (!last) issue_a(m0, c + 1, buf ^ 1, t * PPT + q);
                                else issue_a(m0n, 0, buf ^ 1, t * PPT + q);
                            }
                    }
#endif
                    // The weight tile of K-tile k + 1 (requested during K-tile k - 1) has landed, this wave's pieces; requested after it and
                    // allowed to stay in flight: the halo pieces of K-tile k - 1, this K-tile's weight tile and halo pieces
                    LBC_PROF(const unsigned long long p_b = __builtin_amdgcn_s_memtime(); p_req += p_b - p_a;)
                    wait_vm(w2 ? NBW + (follows ? np_prev + np_here : 0) : 0);
                    LBC_PROF(const unsigned long long p_c = __builtin_amdgcn_s_memtime(); p_vm += p_c - p_b;)
#ifndef LBC_HDMAW_ABL_NOBAR
                    __builtin_amdgcn_s_barrier();
#endif
                    LBC_PROF(p_a = __builtin_amdgcn_s_memtime(); p_bar += p_a - p_c;)
                }
                ++sg;
            }
            if (a.stats) __builtin_amdgcn_s_barrier();              // the epilogue's statistics barrier (conv_hdmap_epi.hpp)
            tile = tilen; mtile = mtilen; n0 = n0n; m0 = m0n;
        }
        LBC_PROF(if (lane == 0) { unsigned long long* o = g_hdmaw_prof + (blockIdx.x * 8 + wave_all) * 4; o[0] = p_req; o[1] = p_vm; o[2] = p_bar; o[3] = p_a - p_t0; })
        return;
    }

    // ===================================================== the multiplying waves =====================================================
#ifndef LBC_HDMAW_ABL_NOPRIO
    __builtin_amdgcn_s_setprio(1);                              // (their VALU / LDS issue ahead of the loader that shares the SIMD)
#endif
    const int tid = wave * 64 + lane;
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

    f32x16 acc[1][MT][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][i][j][r] = 0.f;
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
    // the fragment reads are inline asm with hand-counted lgkmcnt (conv_hdmap.hpp: hipcc's own wait would cover the NEXT step's reads too)
    static_assert(NB * TILE_B + (NT - 1) * 4096 < 65536, "conv_hdmaw: the ring slot is an immediate offset of the read");
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
#define LBC_RD(SLOT, G, SET)                                                                                                     \
    do {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) LBC_RD1(fa[SET][i], aaddr[i] ^ (32 * (G)), 0);                            \
        LBC_RD1(fb[SET][0], baddr ^ (32 * (G)), (SLOT) * TILE_B);                                                                \
        LBC_RD1(fb[SET][1], baddr ^ (32 * (G)), (SLOT) * TILE_B + 4096);                                                         \
    } while (0)
#ifdef LBC_HDMAW_ABL_NOMFMA
#define LBC_MM1(SET, I, J) do { } while (0)
#else
#define LBC_MM1(SET, I, J) acc[0][I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[SET][J], fa[SET][I], acc[0][I][J], 0, 0, 0)   /* weights x pixels^T */
#endif
    // pair P of the six fragment reads of (ring slot, depth step G) into register set SET, in the order the MFMAs need them:
    // (weights 0, rows 0) (weights 1, rows 1) (rows 2, rows 3)
#define LBC_RDP(SLOT, G, SET, P)                                                                                                 \
    do {                                                                                                                         \
        if constexpr ((P) == 0) { LBC_RD1(fb[SET][0], baddr ^ (32 * (G)), (SLOT) * TILE_B); LBC_RD1(fa[SET][0], aaddr[0] ^ (32 * (G)), 0); }          \
        else if constexpr ((P) == 1) { LBC_RD1(fb[SET][1], baddr ^ (32 * (G)), (SLOT) * TILE_B + 4096); LBC_RD1(fa[SET][1], aaddr[1] ^ (32 * (G)), 0); } \
        else { LBC_RD1(fa[SET][2], aaddr[2] ^ (32 * (G)), 0); LBC_RD1(fa[SET][3], aaddr[3] ^ (32 * (G)), 0); }                  \
    } while (0)
#define LBC_PIN() __builtin_amdgcn_sched_barrier(0)

    // ---- the tile stream
    int tile = first;
    int mtile = tile / ntn, n0 = (tile - mtile * ntn) * BN, m0 = mtile * BM;
    int sg = 0;                             // slabs consumed so far: halo buffer sg & 1
    tap_mask(m0, amask);
    __builtin_amdgcn_s_barrier();           // the prologue's: K-tile 0 landed and visible
    tap_addr(0, 0, amask);
    LBC_RD(0, 0, 0);
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
#ifndef LBC_HDMAW_ABL_NOBAR
                __builtin_amdgcn_s_barrier();
#endif
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

        LBC_PROF(const unsigned long long p_e0 = __builtin_amdgcn_s_memtime();)
        hdmaw_tile_epilogue<BN, MODE, EPI, MT, NT, SDB>(a, acc, smem + STG + wave * (SROWS * 128), reinterpret_cast<float*>(smem + RED), wm, wn, lane, tid, m0, n0, mtile);
        LBC_PROF(p_epi += __builtin_amdgcn_s_memtime() - p_e0;)
        zero_acc();
        tile = tilen; mtile = mtilen; n0 = n0n; m0 = m0n;
#pragma unroll
        for (int i = 0; i < MT; ++i) amask[i] = amaskn[i];
    }
    LBC_PROF(if (lane == 0) { unsigned long long* o = g_hdmaw_prof + (blockIdx.x * 8 + wave_all) * 4; o[0] = p_seg; o[1] = p_bar; o[2] = p_epi; o[3] = __builtin_amdgcn_s_memtime() - p_t0; })
#undef LBC_RD
#undef LBC_RD1
#undef LBC_USE
#undef LBC_MM1
#undef LBC_RDP
#undef LBC_PIN
}
#undef LBC_SG

// launches the instantiation for (mode, epilogue form) of one halo size
template <int BM, int BN, int HRMAX, int SROWS>
int conv_hdmaw_launch_shape(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, dim3 grid, hipStream_t s)
{
    const int epi = a.bnb_y ? (a.bnb_mask ? 4 : 2) : (a.resid ? 1 : 0);
    LBC_REQUIRE(!a.pre_scale && (mode == 0 || mode == 1), "conv_hdmaw: forward / input gradient without BatchNorm-on-load only");
#define LBC_HW(MODEv, EPIv) hipLaunchKernelGGL((conv_hdmaw_k<BM, BN, HRMAX, SROWS, MODEv, EPIv>), grid, dim3(512), 0, s, a, zero, ntiles, tpw)
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
