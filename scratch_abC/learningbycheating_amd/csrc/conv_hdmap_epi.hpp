// Wave-private epilogue of the persistent halo-staged convolution (conv_hdmap.hpp).  Reference arithmetic: what follows a BasicBlock convolution -- folded BatchNorm affine /
// bias / residual / ReLU (bird_view/models/resnet.py:38-54) and, for input-gradient launches, the BatchNorm-backward sums of autograd.
//   affine / bias / residual / ReLU on the accumulators, SROWS rows at a time through this wave's own staging rows (SROWS x (WTN bf16 +
//   16 bytes) of LDS that belong to nothing else), read back as 16-byte chunks, stored -- no workgroup barrier; statistics rows (or the
//   fused BatchNorm-backward sums, forms 2 / 4) through a small [WM][2][BN] LDS array and ONE workgroup barrier per tile and phase
//   (every wave of the workgroup must take part in it, whatever its role: `epi_on` = this wave has accumulators to store).
// EPI: 0 plain, 1 + residual, 2 fused BatchNorm-backward reduce (IgemmArgs::bnb_*), 4 = 1 + 2 with the mask from a tensor
// (IgemmArgs::bnb_mask); MODE 2 = NPH 4: the output-parity phases of a stride-2 transposed launch, one accumulator set each.
#pragma once
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include "conv_lds_dma.hpp"

namespace {

template <int BN, int WM, int WTM, int WTN, int SROWS, int MODE, int EPI, int NPH, int MT, int NT>
__device__ __forceinline__ void hdmap_tile_epilogue(const IgemmArgs& a, f32x16 (&acc)[NPH][MT][NT], char* const stg, float* const red, const int wm, const int wn,
                                                    const int lane, const int tid, const bool epi_on, const int m0, const int n0, const int mtile, const int mtiles,
                                                    const int W, const int H)
{
    constexpr int SROW_B = WTN * 2 + 16;                        // LDS pitch of a staged row
    constexpr int NSTEP = WTM / SROWS;                          // copy-out steps per wave and tile
    constexpr int SEGS = WTN / 8;                               // 16-byte segments per staged row
    constexpr int CPL = SROWS * SEGS / 64;                      // 16-byte chunks per lane and step
    constexpr int RPP = 64 / SEGS;                              // rows per copy-out pass of the wave
    const int l31 = lane & 31, kh = lane >> 5;
    __bf16* yout = static_cast<__bf16*>(a.y);
    constexpr bool RES = EPI == 1 || EPI == 4, BNB = EPI == 2 || EPI == 4;
    const __bf16* resid = RES ? static_cast<const __bf16*>(a.resid) : nullptr;
    const __bf16* by = BNB ? static_cast<const __bf16*>(a.bnb_y) : nullptr;
    const __bf16* bmask = EPI == 4 ? static_cast<const __bf16*>(a.bnb_mask) : nullptr;
    const int colw = n0 + wn * WTN;                     // first column of this wave
    const int crow = lane / SEGS, cseg = lane % SEGS;   // copy-out role: chunk lane + 64 q = (row crow + RPP q, segment cseg)
    float psc[NT], psh[NT], bia[NT];
#pragma unroll
    for (int nj = 0; nj < NT; ++nj) {
        const int col = colw + nj * 32 + l31;
        psc[nj] = a.post_scale ? a.post_scale[col] : 1.f;
        psh[nj] = a.post_scale ? a.post_shift[col] : 0.f;
        bia[nj] = a.bias ? a.bias[col] : 0.f;
    }
    // MODE 2: lattice position m = (n, ly, lx) -> element index of output pixel (n, 2 ly, 2 lx) of its copy-out chunks; phase
    // (oy0, ox0) adds oy0 * OW + ox0 pixels
    unsigned obase[MODE == 2 ? NSTEP : 1][CPL];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int s = 0; s < NSTEP; ++s)
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int m = m0 + wm * WTM + s * SROWS + crow + RPP * q;
                const int mm = m < a.M ? m : 0;
                const int x = mm % W, t = mm / W;
                const int y = t % H, n = t / H;
                obase[s][q] = (unsigned)((n * 2 * H + 2 * y) * (2 * W) + 2 * x);
            }
    }
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
    f32x8 t1 = ParamVec<8>::splat(0.f), t2 = t1;
    float s1[NT], s2[NT];
#pragma unroll
    for (int nj = 0; nj < NT; ++nj) { s1[nj] = 0.f; s2[nj] = 0.f; }
    if (epi_on) {
    // every load of the epilogue is requested before its first store (a load behind a store would wait for the store)
    constexpr int SPB = 32 / SROWS, RPS = SROWS / 2;                          // steps per 32-row block, accumulator registers per step
    float rv[RES ? MT : 1][16][NT];
    if constexpr (RES) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // (32-bit element offsets from a uniform base -- M * K < 2^31 is part of the launch's eligibility: one address
                //  register per gather instead of a 64-bit pair)
                const int m = m0 + wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const unsigned ob = (unsigned)(m < a.M ? m : 0) * (unsigned)a.K + (unsigned)(colw + l31);
#pragma unroll
                for (int nj = 0; nj < NT; ++nj) rv[mi][r][nj] = (float)resid[ob + (unsigned)(nj * 32)];
            }
    }
    // Form 2 requests all of its side chunks up front (64 / 128 bytes per lane).  Form 4 has the residual's 64 registers as well: its two
    // side tensors come one copy-out step ahead instead (two register sets; the loads of step s + 1 are requested before the stores
    // of step s, so the wait for them covers stores that are two steps old) -- all up front the eight-wave shapes spilled 26-38 registers
    constexpr bool AHEAD = EPI == 4;
    constexpr int YSETS = AHEAD ? 2 : (BNB ? NSTEP : 1);
    bf16x8 yv[YSETS][CPL], mv[EPI == 4 ? 2 : 1][CPL];
    f32x8 bsc, bsh, bmu, biv;
    auto side_chunks = [&](const int s, const int set) {
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int m = m0 + wm * WTM + s * SROWS + crow + RPP * q;
            const unsigned o = (unsigned)(m < a.M ? m : 0) * (unsigned)a.K + (unsigned)(colw + cseg * 8);
            yv[set][q] = *reinterpret_cast<const bf16x8*>(by + o);
            if constexpr (EPI == 4) mv[set][q] = *reinterpret_cast<const bf16x8*>(bmask + o);
        }
    };
    if constexpr (BNB) {
        const int c0 = colw + cseg * 8;
        if constexpr (EPI == 2) { bsc = ParamVec<8>::ld(a.bnb_scale + c0); bsh = ParamVec<8>::ld(a.bnb_shift + c0); }
        bmu = ParamVec<8>::ld(a.bnb_mean + c0); biv = ParamVec<8>::ld(a.bnb_invstd + c0);
        if constexpr (AHEAD) side_chunks(0, 0);
        else {
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) side_chunks(s, s);
        }
    }
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int mi = s / SPB;
        const int yset = AHEAD ? (s & 1) : s;
        if constexpr (BNB && AHEAD) { if (s + 1 < NSTEP) side_chunks(s + 1, (s + 1) & 1); }
#pragma unroll
        for (int r8 = 0; r8 < RPS; ++r8) {
            const int r = (s % SPB) * RPS + r8;
            const int lr = (r & 3) + 4 * kh + 8 * ((r >> 2) % (SROWS / 8));   // row inside the step
            const bool live = m0 + wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh < a.M;
#pragma unroll
            for (int nj = 0; nj < NT; ++nj) {
                float v = acc[ph][mi][nj][r];
                if (a.post_scale) v = v * psc[nj] + psh[nj];
                if (a.bias) v += bia[nj];
                if constexpr (RES) v += rv[mi][r][nj];
                if (a.relu) v = fmaxf(v, 0.f);
                *reinterpret_cast<__bf16*>(stg + lr * SROW_B + (nj * 32 + l31) * 2) = (__bf16)v;
                if (!BNB && live) { s1[nj] += v; s2[nj] += v * v; }
            }
        }
        // (LDS operations of one wave execute in order: its reads below see its writes above, and the next step's writes
        //  cannot overtake these reads.  wave_barrier emits nothing; it pins the order for the compiler -- and for the
        //  CPU emulator, whose lanes are fibers)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int row = crow + RPP * q;
            const int m = m0 + wm * WTM + s * SROWS + row;
            bf16x8 ch = *reinterpret_cast<const bf16x8*>(stg + row * SROW_B + cseg * 16);
            if constexpr (BNB) {
                // fused BatchNorm-backward reduce (IgemmArgs::bnb_*): mask the stored gradient with bn(y) > 0 (form 4: with the given
                // ReLU output > 0), sum (g, g * xhat)
                const f32x8 yf = __builtin_convertvector(yv[yset][q], f32x8);
                f32x8 g = __builtin_convertvector(ch, f32x8);
                f32x8 z;
                if constexpr (EPI == 4) z = __builtin_convertvector(mv[yset][q], f32x8);
                else z = yf * bsc + bsh;
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = z[e] > 0.f ? g[e] : 0.f;
                ch = __builtin_convertvector(g, bf16x8);
                if (m < a.M) { t1 += g; t2 += g * (yf - bmu) * biv; }
            }
            const unsigned opix = MODE == 2 ? obase[MODE == 2 ? s : 0][q] + (unsigned)((ph >> 1) * 2 * W + (ph & 1)) : (unsigned)m;
            if (m < a.M) *reinterpret_cast<bf16x8*>(yout + (opix * (unsigned)a.K + (unsigned)(colw + cseg * 8))) = ch;
        }
        __builtin_amdgcn_wave_barrier();
    }
    }       // epi_on
    if (a.stats) {
        if (!epi_on) {
            // (instance 1 of a K-split workgroup: nothing to contribute)
        } else if constexpr (BNB) {
            // lanes with the same segment (lane % SEGS) hold partial sums of the same 8 channels: combine over lane / SEGS
#pragma unroll
            for (int off = SEGS; off < 64; off <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) { t1[e] += __shfl_xor(t1[e], off); t2[e] += __shfl_xor(t2[e], off); }
            if (lane < SEGS) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[(wm * 2 + 0) * BN + wn * WTN + lane * 8 + e] = t1[e];
                    red[(wm * 2 + 1) * BN + wn * WTN + lane * 8 + e] = t2[e];
                }
            }
        } else {
#pragma unroll
            for (int nj = 0; nj < NT; ++nj) {
                s1[nj] += __shfl_xor(s1[nj], 32);
                s2[nj] += __shfl_xor(s2[nj], 32);
            }
            if (kh == 0) {
#pragma unroll
                for (int nj = 0; nj < NT; ++nj) {
                    red[(wm * 2 + 0) * BN + wn * WTN + nj * 32 + l31] = s1[nj];
                    red[(wm * 2 + 1) * BN + wn * WTN + nj * 32 + l31] = s2[nj];
                }
            }
        }
        LBC_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        if (epi_on && tid < BN) {
            float u1 = 0.f, u2 = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WM; ++w2) { u1 += red[(w2 * 2 + 0) * BN + tid]; u2 += red[(w2 * 2 + 1) * BN + tid]; }
            // (MODE 2: statistics rows phase-major, as the per-tap kernel writes them: row ph * M-tiles + M-tile)
            float* dst = a.stats + (size_t)(a.stat_row0 + ph * mtiles + mtile) * 2 * (size_t)a.K;
            dst[n0 + tid] = u1;
            dst[a.K + n0 + tid] = u2;
        }
        // (the next write of `red` lies behind at least the nine K-tile barriers of the next tile -- or, between the phases of
        //  a MODE 2 tile, behind this barrier)
        if constexpr (NPH > 1) { LBC_WAIT_LGKM0(); __builtin_amdgcn_s_barrier(); }
    }
    }       // ph
}

}  // namespace
