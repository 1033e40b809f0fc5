// Shared pieces of the LDS-DMA convolutions (conv_glds.hip, conv_hdma.hip): the DMA primitive, counted waits, and the epilogue
// (fused affine / bias / residual / ReLU, bf16 tile staged in LDS -> 16-byte stores, BatchNorm statistics or the fused
// BatchNorm-backward reduce of IgemmArgs::bnb_*).
#pragma once
#include "lbc_common.hpp"
#include "lbc_act.hpp"

namespace {

// s_waitcnt immediate, gfx9 layout: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] at [15:14]
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 15) << 8); }
#define LBC_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(waitcnt_imm((n), 15))
#define LBC_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0))

typedef __attribute__((address_space(1))) const void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;
// 16 bytes per lane from a per-lane global address to (wave-uniform LDS base) + 16 * lane
__device__ __forceinline__ void lds_dma16(const void* g, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((gas_ptr)g, (las_ptr)lds_wave_base, 16, 0, 0);
}

// LDS the epilogue below needs: the staged output tile + the statistics rows
template <int BM, int BN, int WM> constexpr int lds_dma_epilogue_bytes() { return BM * (BN * 2 + 16) + WM * 2 * BN * 4; }

// Epilogue of a workgroup whose WM x WN waves (8, or 4 for the two-workgroups-per-CU shapes) hold a BM x BN output tile in 32 x 32 MFMA accumulators.
// Every wave must have left its main loop reads before this is entered (it starts with a barrier); smem is reused from byte 0.
template <int BM, int BN, int WM, int WN, int MT, int NT>
// ostep = 2 (phased stride-2 transposed launches): row m is lattice point (n, ly, lx) of a.LH x a.LW and lands on output pixel
// (2 ly + oy0, 2 lx + ox0); such launches carry neither a residual nor the fused BatchNorm-backward reduce.
__device__ __forceinline__ void lds_dma_epilogue(const IgemmArgs& a, f32x16 (&acc)[MT][NT], char* smem, const int m0, const int n0, const int mtile,
                                                 const int ostep = 1, const int oy0 = 0, const int ox0 = 0)
{
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int NTH = WM * WN * 64;                           // threads of the workgroup
    constexpr int OROW = BN * 2 + 16;                           // staged output row: BN bf16 + 16 bytes (rows 4 apart on distinct banks)
    constexpr int STAGE = BM * OROW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;
    // affine / bias / residual / ReLU on the accumulators, per-channel (sum, sum^2), bf16 tile staged in LDS
    LBC_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();                      // every wave has left the main loop: the ring is free
    float s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const __bf16* resid = static_cast<const __bf16*>(a.resid);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        float rv[16][NT];
        if (resid) {        // fetched per 32-row block before its use: inside the loop every 2-byte load is waited for alone
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const size_t ob = (size_t)(m < a.M ? m : 0) * (size_t)a.K;
#pragma unroll
                for (int nj = 0; nj < NT; ++nj) rv[r][nj] = (float)resid[ob + (size_t)(n0 + wn * WTN + nj * 32 + l31)];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const bool live = m0 + row < a.M;
#pragma unroll
            for (int nj = 0; nj < NT; ++nj) {
                const int cl = wn * WTN + nj * 32 + l31;
                const int col = n0 + cl;
                float v = acc[mi][nj][r];
                if (a.post_scale) v = v * a.post_scale[col] + a.post_shift[col];
                if (a.bias) v += a.bias[col];
                if (resid) v += rv[r][nj];
                if (a.relu) v = fmaxf(v, 0.f);
                *reinterpret_cast<__bf16*>(smem + row * OROW + cl * 2) = (__bf16)v;
                if (live) { s1[nj] += v; s2[nj] += v * v; }
            }
        }
    }
    float* red = reinterpret_cast<float*>(smem + STAGE);   // [WM][2][BN]
    if (a.stats) {
#pragma unroll
        for (int nj = 0; nj < NT; ++nj) {
            s1[nj] += __shfl_xor(s1[nj], 32);
            s2[nj] += __shfl_xor(s2[nj], 32);
        }
        if (kh == 0) {
#pragma unroll
            for (int nj = 0; nj < NT; ++nj) {
                const int c = wn * WTN + nj * 32 + l31;
                red[(wm * 2 + 0) * BN + c] = s1[nj];
                red[(wm * 2 + 1) * BN + c] = s2[nj];
            }
        }
    }
    __syncthreads();
    __bf16* yout = static_cast<__bf16*>(a.y);
    constexpr int SEG = BN / 8;                         // 16-byte segments per output row
    if (a.bnb_y == nullptr) {
#pragma unroll 4
        for (int idx = tid; idx < BM * SEG; idx += NTH) {
            const int row = idx / SEG, sg = idx - row * SEG;
            const int m = m0 + row;
            if (m < a.M) {
                size_t pix = (size_t)m;
                if (ostep == 2) {
                    const int lx = m % a.LW, t2 = m / a.LW;
                    const int ly = t2 % a.LH, n = t2 / a.LH;
                    pix = ((size_t)n * a.OH + (size_t)(2 * ly + oy0)) * a.OW + (size_t)(2 * lx + ox0);
                }
                *reinterpret_cast<bf16x8*>(yout + pix * (size_t)a.K + (size_t)(n0 + sg * 8)) =
                    *reinterpret_cast<const bf16x8*>(smem + row * OROW + sg * 16);
            }
        }
        if (a.stats && tid < BN) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { t1 += red[(w * 2 + 0) * BN + tid]; t2 += red[(w * 2 + 1) * BN + tid]; }
            float* dst = a.stats + (size_t)(a.stat_row0 + mtile) * 2 * (size_t)a.K;
            dst[n0 + tid] = t1;
            dst[a.K + n0 + tid] = t2;
        }
    } else {
        // Fused BatchNorm-backward reduce (IgemmArgs::bnb_*): the copy-out pass reads the pre-BN activation next to the staged
        // gradient (16 bytes each), masks, stores, and sums (g, g * xhat) for the thread's fixed 8-channel segment
        // (NTH % SEG == 0); the NTH / SEG threads of a segment are combined through LDS in thread order (deterministic).
        static_assert(NTH % SEG == 0 && NTH * 64 <= STAGE, "conv_glds2: segment ownership / scratch");
        const __bf16* by = static_cast<const __bf16*>(a.bnb_y);
        const int sg = tid % SEG;
        const int c0 = n0 + sg * 8;
        const f32x8 bsc = ParamVec<8>::ld(a.bnb_scale + c0), bsh = ParamVec<8>::ld(a.bnb_shift + c0);
        const f32x8 bmu = ParamVec<8>::ld(a.bnb_mean + c0), biv = ParamVec<8>::ld(a.bnb_invstd + c0);
        f32x8 t1 = ParamVec<8>::splat(0.f), t2 = t1;
#pragma unroll 4
        for (int row = tid / SEG; row < BM; row += NTH / SEG) {
            const int m = m0 + row;
            if (m < a.M) {
                const size_t o = (size_t)m * (size_t)a.K + (size_t)c0;
                const f32x8 yv = __builtin_convertvector(*reinterpret_cast<const bf16x8*>(by + o), f32x8);
                f32x8 g = __builtin_convertvector(*reinterpret_cast<const bf16x8*>(smem + row * OROW + sg * 16), f32x8);
                const f32x8 z = yv * bsc + bsh;
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = z[e] > 0.f ? g[e] : 0.f;
                *reinterpret_cast<bf16x8*>(yout + o) = __builtin_convertvector(g, bf16x8);
                t1 += g;
                t2 += g * (yv - bmu) * biv;
            }
        }
        __syncthreads();                                // the staged tile has been consumed: its LDS holds the partial sums now
        float* ps = reinterpret_cast<float*>(smem);     // [NTH][16]
        ParamVec<8>::st(ps + tid * 16, t1);
        ParamVec<8>::st(ps + tid * 16 + 8, t2);
        __syncthreads();
        if (a.stats && tid < BN) {
            const int seg = tid >> 3, e = tid & 7;
            float u1 = 0.f, u2 = 0.f;
            for (int k = 0; k < NTH / SEG; ++k) {
                u1 += ps[(k * SEG + seg) * 16 + e];
                u2 += ps[(k * SEG + seg) * 16 + 8 + e];
            }
            float* dst = a.stats + (size_t)(a.stat_row0 + mtile) * 2 * (size_t)a.K;
            dst[n0 + tid] = u1;
            dst[a.K + n0 + tid] = u2;
        }
    }
}

}  // namespace
