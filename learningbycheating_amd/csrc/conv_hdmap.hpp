// PERSISTENT form of the halo-staged LDS-DMA convolution (conv_hdma.hip explains the staging): 3x3 / stride-1 / pad-1 forward and
// input gradient on bf16 tensors, reference arithmetic bird_view/models/resnet.py:15-22,38-54 (BasicBlock conv1 / conv2) and autograd.
//
// Why (round-3 ablations of conv_hdma_k on the layer-3 shape at batch 256, profiles/r03_run2_hdma_ablation.log, r03_run3_hdma_pmc_ablation.log):
// MFMA-only 45 us, everything-but-MFMA 45 us, together 72 us.  The workgroups of a launch run in lock-step rounds (one per CU, 160 KB of
// LDS): every round opens with all 256 CUs fetching their first halo + weight tiles at once (14 MB, no MFMA can run) and closes with all
// of them storing their output tiles at once (16 MB, the workgroup cannot retire before its stores have), plus two workgroup-wide
// barriers and 64 two-byte LDS writes per lane in the copy-out.  That is ~7 us of HBM bursts per round in a 36 us round.  Here
//   * a workgroup walks `tpw` consecutive output tiles (same M-tile first: the second tile's halo comes from L2) as ONE continuous
//     stream of K-tiles: the halo of the next tile's first slab and its first two weight tiles are requested during the last slab
//     of the current tile (DMA roles are recomputed per piece from the tile origin: no per-tile address arrays);
//   * the epilogue is WAVE-PRIVATE: each wave stages 16 rows x 64 columns of its own accumulators in 2.3 KB of LDS that belongs to
//     nothing else (not the halo buffers, not the weight ring: both already hold the next tile), reads them back as 16-byte chunks
//     and stores them -- no barrier, and the stores stay in flight under the next tile's K-tiles (vmcnt retires in order on gfx950:
//     the first K-tile of the next tile waits with vmcnt(#stores), i.e. for the DMA pieces issued BEFORE the stores only);
//   * statistics rows / the fused BatchNorm-backward sums go through a small [WM][2][BN] LDS array and one barrier per tile.
// Weight ring: 3 tiles (slot = tap % 3: a compile-time constant, since 9 taps per slab).  K-tile k requests the weight tile of K-tile
// k + 2 and one halo piece of the next slab, spread over its first three depth steps (a burst of 24 DMA instructions from 8 waves
// right after the barrier stalled every wave's issue: profiles/r03_run6_hdmap_prof.log); every K-tile waits with a counted vmcnt for its
// successor's weight tile: what this wave requested after that tile stays in flight, the previous tile's stores included.
#pragma once
#include <type_traits>
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include "conv_lds_dma.hpp"

namespace {

#define LBC_SG(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)

// A wave-uniform pointer the compiler must keep in scalar registers: the DMA source is then (SGPR base, 32-bit VGPR offset) instead
// of a per-thread 64-bit pointer that is re-formed with two-instruction 64-bit adds per piece
__device__ __forceinline__ const char* uniform_ptr(const char* p)
{
    const unsigned long long v = (unsigned long long)(size_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char*>((size_t)(((unsigned long long)hi << 32) | lo));
}

// EPI: 0 = plain epilogue (affine / bias / ReLU / statistics), 1 = + residual, 2 = fused BatchNorm-backward reduce (IgemmArgs::bnb_*).
// One instantiation per form: the residual prefetch (64 registers) and the BatchNorm-backward operands (64) never coexist.
// EPI 4 (round 5) = 1 + 2 with the mask from a tensor: the input gradient of a block's conv1 (+ the identity-path gradient as residual) IS
// the gradient wrt the previous block's output relu(bn2(y2) + identity); masking it with that output (IgemmArgs::bnb_mask > 0) and summing
// (g, g * xhat(y2)) here removes the previous block's whole channel_reduce pass (three tensors read, one written).  The residual is added
// to the f32 accumulator as in form 1 (same rounding as the unfused path), mask and sums happen in the chunk phase as in form 2.
// MODE 2 (round 5; four-wave shape, plain epilogue): the stride-2 TRANSPOSED launches -- input gradient of the stride-2 3x3 convolutions
// (resnet.py:132-138), ConvTranspose2d(.,.,3,2,1,1) forward (image.py:37-47) -- with all four output-parity phases of a lattice position
// in ONE tile.  Output pixel (2 ly + oy0, 2 lx + ox0) gathers x at (ly + dy, lx + dx) through the taps (r, s) with dy = [r == 0],
// dx = [s == 0] and oy0 = [r != 1], ox0 = [s != 1]: a 2 x 2 neighbourhood, nine taps, each feeding exactly one of four accumulator sets
// (1 / 2 / 2 / 4 taps per phase).  So the halo is BM + W + 2 rows, staged once per 64-channel slab like the stride-1 forms, a tap is a
// row offset in {0, 1, W, W + 1}, and the K loop is the stride-1 loop with the accumulator set chosen by the (compile-time) tap.  The
// per-tap LDS-DMA kernel (conv_glds2_k<.., PH>) these launches used stages one shifted activation tile per (tap, slab) for K loops of
// 2 - 8 K-tiles per (phase, tile) workgroup: 320 - 600 TF/s and 226 MB fetched for a 63 MB operand (profiles/r05_final_*).
// KG = 2 (round 5; four-wave shape, launches of at most one tile per CU): an IN-WORKGROUP split of the channel contraction.  A launch with
// <= 256 tiles of 128 x 64 puts one four-wave workgroup on a CU -- one wave per SIMD, nothing hides that wave's LDS / DMA-issue / barrier
// latencies, and the serial K loop (36 - 72 K-tiles) IS the launch (layers 3 / 4 at 32 images per GPU: 15 / 22 us for 5.4 GFLOP).  With
// KG = 2 the workgroup has eight waves = two independent instances of the four-wave pipeline (own halo buffers, own weight ring, own
// staging: 2 x 78 KB of LDS), instance g contracting the slabs [g * nslab, (g + 1) * nslab); they share nothing but the workgroup
// barriers (same K-tile count, same control flow).  After the K loop instance 1 hands its accumulators over through LDS (its halo
// buffers are free by then), instance 0 adds them in a fixed order and does the epilogue: half the K loop, no partial tiles in HBM and
// no second launch (what form 3 pays).  One tile per workgroup (the launcher guarantees it).
// EPI 3 = split-K (round 4; launches with few tiles, i.e. the deep layers at the per-GPU batches of the 8-GPU run): workgroup p serves
// (tile p / nsplit, slab range p % nsplit), contracts C / nsplit of the gathered channels and leaves its accumulators as an f32 partial
// tile in IgemmArgs::split_ws [range][M][K]; conv_split_epilogue_k (conv_hdmap.hip) sums the ranges in a fixed order and does the
// epilogue of form 0 / 1 / 2.  One tile per workgroup (the launcher guarantees ntiles * nsplit workgroups).
template <int BM, int BN, int WM, int WN, int HRMAX, int SROWS>
constexpr int hdmap_lds_bytes() { return 2 * HRMAX * 128 + 3 * BN * 128 + WM * WN * SROWS * ((BN / WN) * 2 + 16) + WM * 2 * BN * 4; }
template <int BM, int BN, int WM, int WN, int HRMAX, int SROWS, int MODE, int EPI, int KG = 1>
__global__ __launch_bounds__(WM * WN * 64 * KG, (BM / WM == 128 && WM * WN == 4) ? 1 : 2) void conv_hdmap_k(IgemmArgs a, const void* zero_page, const int ntiles, const int tpw, const int nsplit)
{
    static_assert(KG == 1 || (KG == 2 && WM * WN == 4 && EPI != 3), "conv_hdmap: the in-workgroup K split doubles the four-wave shape");
    static_assert(MODE != 2 || (WM * WN == 4 && EPI == 0 && KG == 1), "conv_hdmap: the phased transposed form is a plain-epilogue form of the four-wave shape");
    constexpr int NPH = MODE == 2 ? 4 : 1;                      // accumulator sets (output-parity phases)
    constexpr int WTM = BM / WM, WTN = BN / WN;                 // per-wave output tile
    constexpr int MT = WTM / 32, NT = WTN / 32;
    constexpr int NW = WM * WN;                                 // waves per workgroup: 8 (256 x 128 / 128 x 256 tiles, one workgroup per CU) or
                                                                // 4 (128 x 64 tiles for launches with few rows: two workgroups per CU)
    // (MT = 4 with four waves: ONE wave per SIMD with a 128 x 64 tile and the whole register file -- 32 MFMAs per K-tile barrier, 24
    //  fragment reads, no second wave competing for the matrix pipe)
    static_assert((NW == 8 || NW == 4) && (NT == 2 || NT == 1) && (MT == 2 || MT == 4), "conv_hdmap: wave tiling");
    static_assert(HRMAX % (8 * NW) == 0 && BN % (8 * NW) == 0 && (SROWS == 8 || SROWS == 16), "conv_hdmap: staging");
    constexpr int NB = 3;                                       // weight ring depth: slot of K-tile (slab, tap) = tap % 3 (9 taps per slab)
    constexpr int KS = 4;                                       // depth steps of 16 channels per K-tile
    constexpr int ABYTES = HRMAX * 128;                         // one halo buffer: HRMAX rows x 64 channels
    constexpr int TILE_B = BN * 128;
    constexpr int BRING = 2 * ABYTES;                           // ring of NB weight tiles behind the two halo buffers
    constexpr int SROW_B = WTN * 2 + 16;                        // LDS pitch of a staged row (64 bf16 + 16 bytes); SROWS rows per copy-out step
    constexpr int STG = BRING + NB * TILE_B;                    // wave-private staging: NW x SROWS x SROW_B
    constexpr int RED = STG + NW * SROWS * SROW_B;              // [WM][2][BN] floats
    constexpr int SMEM = RED + WM * 2 * BN * 4;
    static_assert(SMEM == hdmap_lds_bytes<BM, BN, WM, WN, HRMAX, SROWS>(), "conv_hdmap: LDS layout");
    constexpr int ZROW = (HRMAX - 1) * 128;                     // last row of either halo buffer: beyond the halo, filled from the zero page
    static_assert(SMEM * KG <= 160 * 1024, "conv_hdmap: LDS");
    __shared__ __attribute__((aligned(16))) char smem_all[SMEM * KG];    // the ONLY LDS object (KG = 2: one SMEM-sized region per instance)
    constexpr int HPW = HRMAX / (8 * NW);                       // 1-KiB halo pieces (8 rows) per wave per slab
    constexpr int NBW = BN / (8 * NW);                          // 1-KiB weight pieces per wave per K-tile
    constexpr int ATAPS = 7;                                    // taps of a slab whose issue slot may carry halo pieces of the next slab
    constexpr int PPT = (HPW + ATAPS - 1) / ATAPS;              // halo pieces of the next slab requested per tap (taps 0 .. ATAPS - 1)
    static_assert(PPT >= 1 && PPT <= 2, "conv_hdmap: halo pieces per tap");
    constexpr int NSTEP = WTM / SROWS;                          // copy-out steps per wave and tile
    constexpr int SEGS = WTN / 8;                               // 16-byte segments per staged row
    constexpr int CPL = SROWS * SEGS / 64;                      // 16-byte chunks per lane and step
    constexpr int RPP = 64 / SEGS;                              // rows per copy-out pass of the wave
    constexpr int NST = NSTEP * CPL;                            // 16-byte store instructions per wave and tile

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int grp = KG == 2 ? wave_all / NW : 0;                 // K-split instance of this wave
    const int wave = KG == 2 ? wave_all % NW : wave_all;         // its role inside the instance
    const int tid = wave * 64 + lane;                           // thread id inside the instance
    char* const smem = smem_all + grp * SMEM;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;
    const int W = a.W, H = a.H, C = a.C;
    const int ntn = a.K / BN;
    const int nslab = EPI == 3 ? C / 64 / nsplit : C / 64 / KG;     // slabs (64 gathered channels) this workgroup (KG = 2: this instance) contracts

    // this workgroup's tiles: [first, first + cnt), consecutive ids share the M-tile
    int first, cnt, split = 0;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, rr = nwg & 7;
        const int p = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
        if constexpr (EPI == 3) {
            first = p / nsplit;
            split = p - first * nsplit;
            cnt = first < ntiles ? 1 : 0;
        } else {
            first = p * tpw;
            cnt = ntiles - first < tpw ? ntiles - first : tpw;
        }
    }
    if (cnt <= 0) return;

    // (split-K: the range's first slab is folded into the operand bases -- 128 bytes per slab in a pixel's row and in a weight row)
    const __bf16* xin = static_cast<const __bf16*>(a.x) + (EPI == 3 ? split * nslab * 64 : (KG == 2 ? grp * nslab * 64 : 0));
    const __bf16* win = static_cast<const __bf16*>(a.w) + (EPI == 3 ? split * nslab * 64 : (KG == 2 ? grp * nslab * 64 : 0));
    const int prow = lane >> 3, pseg = lane & 7;

    // ---- DMA roles: per-thread constants + a wave-uniform origin per piece (the issue slots sit next to MFMAs that leave room
    //      for ~5 other instructions each: every VALU counts).  Halo row hr of a tile with origin m0 holds input pixel
    //      m0 - (W + 1) + hr.  Rows outside the tensor read a clamped pixel (they are only ever met by taps that the border select
    //      sends to the zero row); pieces that lie entirely past the halo (8 p >= BM + 2W + 2: the launcher guarantees that the
    //      last piece, which holds the ZERO ROW, is one of them) come from the zero page
    const int HR = MODE == 2 ? BM + W + 2 : BM + 2 * W + 2;     // (MODE 2: halo row hr holds input pixel m0 + hr -- no taps above / left of the position)
    const int hshift = MODE == 2 ? 0 : W + 1;
    const int arow0 = wave * HPW * 8 + prow;                                   // halo row of piece j: arow0 + 8 j
    const unsigned aswz[2] = {(unsigned)((pseg ^ ((arow0 >> 1) & 7)) * 16), (unsigned)((pseg ^ (((arow0 >> 1) + 4) & 7)) * 16)};   // j even / odd
    const unsigned zoff = (unsigned)((lane & 7) * 16);
    const char* xbytes = reinterpret_cast<const char*>(xin);
    const char* zbytes = static_cast<const char*>(zero_page);
    auto issue_a = [&](const int m0x, const int slab, const int buf, const int j) {
        const bool pad = (wave * HPW + j) * 8 >= HR;                           // wave-uniform
        int q = m0x - hshift + arow0 + 8 * j;
        q = q < 0 ? 0 : (q >= a.M ? a.M - 1 : q);
        const unsigned off = (unsigned)q * (unsigned)(2 * C) + aswz[j & 1];
        const char* sbase = uniform_ptr(pad ? zbytes : xbytes + (size_t)(slab * 128));
        lds_dma16(sbase + (pad ? zoff : off), smem + buf * ABYTES + (wave * HPW + j) * 1024);
    };
    // weight tile (slab, tap) of the tile with column origin n0x into ring slot `slot`: uniform base + per-thread byte offset
    unsigned voffb[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int row = (wave * NBW + j) * 8 + prow;
        voffb[j] = (unsigned)row * (unsigned)(18 * C) + (unsigned)((pseg ^ ((row >> 1) & 7)) * 16);
    }
    const char* wbytes = reinterpret_cast<const char*>(win);
    auto issue_b = [&](const int n0x, const int slab, const int tap, const int slot, const int j0, const int j1) {
        char* base = smem + BRING + slot * TILE_B;
        const char* wsrc = uniform_ptr(wbytes + ((size_t)n0x * (size_t)(18 * C) + (size_t)(2 * (tap * C + slab * 64))));
#pragma unroll
        for (int j = 0; j < NBW; ++j)
            if (j >= j0 && j < j1) lds_dma16(wsrc + voffb[j], base + (wave * NBW + j) * 1024);
    };
    // ---- fragment roles.  A fragment of depth step g sits in 16-byte slot (2g + kh) ^ f(row) of its 128-byte LDS row, f(row) =
    //      (row >> 1) & 7 (the swizzle of the DMA source).  With the row base a multiple of 128:  address = (base | (kh ^ f) << 4) ^ 32 g
    //      -- one v_xor per read, everything else is formed once per tap (activations) or once per kernel (weights)
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
                    const int dy = MODE == 0 ? r - 1 : (MODE == 2 ? (r == 0 ? 1 : 0) : 1 - r);
                    const int dx = MODE == 0 ? s - 1 : (MODE == 2 ? (s == 0 ? 1 : 0) : 1 - s);
                    if ((unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W) bits |= 1 << t;
                }
            }
            mask[i] = bits;
        }
    };
    int amask[MT], amaskn[MT];

    f32x16 acc[NPH][MT][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int p = 0; p < NPH; ++p)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[p][i][j][r] = 0.f;
    };
    zero_acc();

    // per (tap, 32-row block): LDS address of the lane's depth-step-0 fragment in halo buffer `buf` (its halo row, or the zero row)
    int aaddr[MT];
    auto tap_addr = [&](const int tap, const int buf, const int (&mask)[MT]) {
        const int r = tap / 3, s = tap - 3 * r;
        const int off = MODE == 0 ? (r - 1) * W + (s - 1) : (MODE == 2 ? (r == 0 ? W : 0) + (s == 0 ? 1 : 0) : (1 - r) * W + (1 - s));
        const int abuf = buf * ABYTES;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int hr = rowc[i] + off;
            // (every border lane reads ONE shared zero slot: a broadcast.  It sits on 4 banks that one of the 15 other lanes of its ds_read_b128
            //  group also needs -- 13 % of the LDS cycles at W = 24, 30 % at W = 12 are bank conflicts from exactly this, predicted to the
            //  per cent by scripts/probe/lds_conflict_model.py.  Round 6 measured the conflict-free alternative -- two zero rows, the zero read
            //  at the lane's own position inside the 256-byte bank period: PMC conflicts 13.5 -> 1.1 % / 29.5 -> 0.9 % and the step 0.5 %
            //  SLOWER (profiles/r06_call1_*, r06_call15_*): these launches run at the chip's power cap, LDS cycles are not their limiter,
            //  and sixteen distinct zero reads cost more than one broadcast.  The broadcast stays.)
            const int val = (hr << 7) | ((kh ^ ((hr >> 1) & 7)) << 4), zval = ZROW | (kh << 4);
            const int m = -((mask[i] >> tap) & 1);               // all ones when the tap is inside the image (written as a bit
            aaddr[i] = abuf + (((val ^ zval) & m) ^ zval);       // select: as `ok ? val : zval` the compiler branches over exec)
        }
    };

    bf16x8 fa[2][MT], fb[2][NT];            // two register sets: depth step g computes from set g & 1 while set (g + 1) & 1 is read
    // ASMRD: the fragment reads are inline asm the compiler does not track, waited for with hand-counted lgkmcnt.  With LDS-DMA
    // in flight hipcc's own wait insertion puts `s_waitcnt lgkmcnt(0)` in front of every depth step's MFMAs -- which also waits for the
    // reads of the NEXT step issued just before it: a full LDS latency per depth step that only the SIMD's other wave can cover.  Here
    // the wait in front of step g leaves the MT + NT youngest reads (step g + 1) in flight; LBC_USE (an empty asm the MFMAs depend on)
    // keeps the MFMAs behind that wait.
    constexpr bool ASMRD = NB * TILE_B + (NT - 1) * 4096 < 65536;     // (the ring slot is an immediate offset of the read)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_all + (unsigned)(grp * SMEM);
#ifdef LBC_HIP_EMULATED_FOR_TESTS
#define LBC_RD1(DST, ADDR, OFF) DST = *reinterpret_cast<const bf16x8*>(smem + (ADDR) + (OFF))
#define LBC_USE(SET) do { } while (0)
#else
#define LBC_RD1(DST, ADDR, OFF)                                                                                                  \
    do {                                                                                                                         \
        if constexpr (ASMRD) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(lds0 + (unsigned)(ADDR)), "n"(OFF)); \
        else DST = *reinterpret_cast<const bf16x8*>(smem + (ADDR) + (OFF));                                                      \
    } while (0)
#define LBC_USE(SET)                                                                                                             \
    do {                                                                                                                         \
        if constexpr (ASMRD) {                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[SET][i]));                                  \
            _Pragma("unroll") for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(fb[SET][j]));                                  \
        }                                                                                                                        \
    } while (0)
#endif
    // all but the MT + NT youngest LDS reads of this wave have returned (ASMRD: fragment reads are the only LGKM traffic of the K loop)
#define LBC_WAIT_OLDER_READS() do { if constexpr (ASMRD) __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, MT + NT)); } while (0)
#define LBC_RD(SLOT, G, SET)                                                                                                     \
    do {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) LBC_RD1(fa[SET][i], aaddr[i] ^ (32 * (G)), 0);                            \
        if constexpr (NT == 2) { LBC_RD1(fb[SET][0], baddr ^ (32 * (G)), (SLOT) * TILE_B); LBC_RD1(fb[SET][NT - 1], baddr ^ (32 * (G)), (SLOT) * TILE_B + 4096); } \
        else LBC_RD1(fb[SET][0], baddr ^ (32 * (G)), (SLOT) * TILE_B);                                                           \
    } while (0)
    // (PHX: the accumulator set of the K-tile's tap -- a constant once the tap loop is unrolled; 0 outside MODE 2)
#define LBC_MM(SET, PHX)                                                                                                         \
    do {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                           \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                       \
                acc[PHX][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][i], fb[SET][j], acc[PHX][i][j], 0, 0, 0);       \
    } while (0)

    // s_waitcnt vmcnt(n) for the handful of counts the stream produces (the immediate must be a constant)
    auto wait_vm = [&](const int n) {
        switch (n) {
#define LBC_WV(N) case N: LBC_WAIT_VM(N); break;
            LBC_WV(1) LBC_WV(2) LBC_WV(3) LBC_WV(4) LBC_WV(5) LBC_WV(6) LBC_WV(7) LBC_WV(8) LBC_WV(9) LBC_WV(10) LBC_WV(11) LBC_WV(12)
            LBC_WV(13) LBC_WV(14) LBC_WV(15) LBC_WV(16) LBC_WV(17) LBC_WV(18) LBC_WV(19) LBC_WV(20) LBC_WV(21) LBC_WV(22) LBC_WV(23) LBC_WV(24)
#undef LBC_WV
            default: LBC_WAIT_VM(0); break;      // (0, and anything unforeseen: wait for everything)
        }
    };
    static_assert(NBW + 2 * PPT + NPH * NST <= 24, "conv_hdmap: counted waits");

    // ---- the tile stream
    int tile = first;
    int mtile = tile / ntn, n0 = (tile - mtile * ntn) * BN, m0 = mtile * BM;
    int sg = 0;                             // slabs consumed so far: halo buffer sg & 1
    tap_mask(m0, amask);

    // prologue: the halo of slab 0 and the first two weight tiles in flight; everything of K-tile 0 landed and visible
#pragma unroll
    for (int j = 0; j < HPW; ++j) issue_a(m0, 0, 0, j);
    issue_b(n0, 0, 0, 0, 0, NBW);
    issue_b(n0, 0, 1, 1, 0, NBW);
    LBC_WAIT_VM(NBW);
    __builtin_amdgcn_s_barrier();
    tap_addr(0, 0, amask);
    LBC_RD(0, 0, 0);
    bool stores_pending = false;            // the previous tile's output stores may still be in this wave's VMEM queue
    for (int it = 0; it < cnt; ++it) {
        const bool more = it + 1 < cnt;
        const int tilen = tile + 1;
        const int mtilen = tilen / ntn, n0n = (tilen - mtilen * ntn) * BN, m0n = mtilen * BM;
        if (more) tap_mask(m0n, amaskn);

        // One slab = nine K-tiles, taps unrolled.  LAST: the tile's last slab -- what follows in the stream is the next tile (if any).
        auto slab_body = [&](const int c, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            const bool follows = !LAST || more;                 // another slab follows this one in the stream
            // (row base, XOR term) of a tap do not depend on the slab: left alone, the compiler hoists all 9 x MT pairs out of the loops
#pragma unroll
            for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(rowc[i]), "+v"(amask[i]));
            const int buf = sg & 1;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int slot = t % 3, nslot = (t + 1) % 3, islot = (t + 2) % 3;
                const int tph = MODE == 2 ? (t / 3 == 1 ? 0 : 2) + (t % 3 == 1 ? 0 : 1) : 0;      // phase 2 oy0 + ox0 this tap feeds
                const bool has_next = t < 8 || follows;
                const bool w2 = t + 2 < 9 || follows;                          // K-tile k + 2 exists
                const int np_here = (t * PPT < HPW ? (HPW - t * PPT < PPT ? HPW - t * PPT : PPT) : 0);            // halo pieces of the next slab requested here
                const int np_prev = t >= 1 ? ((t - 1) * PPT < HPW ? (HPW - (t - 1) * PPT < PPT ? HPW - (t - 1) * PPT : PPT) : 0) : 0;
                const bool hp = np_here > 0 && follows;
                // DMA requests of this K-tile: K-tile k + 2's weight tile -> ring slot (t + 2) % 3 (read last by K-tile k - 1: free since
                // that K-tile's barrier), piece by piece; then the halo piece
                auto issue_w = [&](const int j0, const int j1) {
                    if (!w2) return;
                    const int tt = t + 2 < 9 ? t + 2 : t - 7;
                    const int cc = t + 2 < 9 ? c : (LAST ? 0 : c + 1);
                    const int nn = (t + 2 < 9 || !LAST) ? n0 : n0n;
                    issue_b(nn, cc, tt, islot, j0, j1);
                };
                auto issue_h = [&]() {
                    if (!hp) return;
#pragma unroll
                    for (int q = 0; q < PPT; ++q)
                        if (t * PPT + q < HPW) {
                            if (!LAST) issue_a(m0, c + 1, buf ^ 1, t * PPT + q);
                            else issue_a(m0n, 0, buf ^ 1, t * PPT + q);
                        }
                };
#pragma unroll
                for (int g = 0; g + 1 < KS; ++g) {
                    LBC_RD(slot, g + 1, (g + 1) & 1);
                    // the reads of the last depth step are out: the addresses are free for the next K-tile's (tap, slab, tile)
                    if (g == KS - 2 && has_next) {
                        if (t < 8) tap_addr(t + 1, buf, amask);
                        else if (!LAST) tap_addr(0, buf ^ 1, amask);
                        else tap_addr(0, buf ^ 1, amaskn);
                    }
                    if (g == 0) issue_w(0, NBW / 2);
                    else if (g == 1) issue_w(NBW / 2, NBW);
                    else issue_h();
                    LBC_WAIT_OLDER_READS();                              // set g & 1 is in (its reads were issued a full step ago)
                    LBC_USE(g & 1);
                    LBC_MM(g & 1, tph);
                    // the step's fragment reads first: a full step of MFMAs (128 cycles of this wave's own, 256 with its SIMD
                    // partner) between a read and the wait that needs it -- a wave that runs alone no longer stalls on LDS latency
                    LBC_SG(0x100, MT + NT);
#pragma unroll
                    for (int q = 0; q < MT * NT; ++q) { LBC_SG(0x008, 1); LBC_SG(0x036, 5); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // The weight tile of K-tile k + 1 (requested during K-tile k - 1) has landed, this wave's pieces.  Requested after it and
                // allowed to stay in flight: the halo piece of K-tile k - 1, this K-tile's weight tile and halo piece -- and, in the first
                // K-tile behind a tile boundary, the previous tile's stores (they sit between K-tile k - 1's requests and this one's)
                {
                    int n = 0;
                    if (w2) {
                        n = NBW + (follows ? np_prev + np_here : 0);
                        if (c == 0 && t == 0 && stores_pending) n += NPH * NST;
                    }
                    wait_vm(n);
                }
                LBC_WAIT_LGKM0();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (has_next) LBC_RD(nslot, 0, 0);
                LBC_USE((KS - 1) & 1);                                   // in since the lgkmcnt(0) in front of the barrier
                LBC_MM((KS - 1) & 1, tph);
                LBC_SG(0x100, MT + NT);
#pragma unroll
                for (int q = 0; q < MT * NT; ++q) { LBC_SG(0x008, 1); LBC_SG(0x036, 6); }
                __builtin_amdgcn_sched_barrier(0);
            }
            ++sg;
        };
        for (int c = 0; c + 1 < nslab; ++c) slab_body(c, std::false_type{});
        slab_body(nslab - 1, std::true_type{});

        // ---- KG = 2: instance 1's accumulators -> instance 0, through instance 1's halo buffers (idle: one tile per workgroup, nothing is
        //      prefetched behind the last slab), [wave][register][lane] floats -- lane-contiguous, conflict-free both ways
        if constexpr (KG == 2) {
            static_assert(NW * MT * NT * 16 * 64 * 4 <= 2 * ABYTES, "conv_hdmap: accumulator hand-over buffer");
            float* xch = reinterpret_cast<float*>(smem_all + SMEM) + (size_t)wave * (MT * NT * 16 * 64) + lane;
            LBC_WAIT_LGKM0();
            __builtin_amdgcn_s_barrier();           // every wave of instance 1 has read its last fragments: its halo buffers may be overwritten
            if (grp == 1) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) xch[((i * NT + j) * 16 + r) * 64] = acc[0][i][j][r];
            }
            LBC_WAIT_LGKM0();
            __builtin_amdgcn_s_barrier();
            if (grp == 0) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[0][i][j][r] += xch[((i * NT + j) * 16 + r) * 64];
            }
        }
        const bool epi_on = KG == 1 || grp == 0;   // (instance 1 only keeps the epilogue's workgroup barrier company)
        // ---- wave-private epilogue: affine / bias / residual / ReLU on the accumulators, 16 rows at a time through this wave's own
        //      staging rows, 16-byte stores; statistics (or the fused BatchNorm-backward sums) per tile
        if constexpr (EPI == 3) {
            // split-K: the accumulators as they are, f32, into this range's partial tile (a 32-lane half-wave writes 128 contiguous bytes)
            float* part = a.split_ws + (size_t)split * (size_t)a.M * (size_t)a.K;
            const int colw = n0 + wn * WTN;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (m < a.M) {
#pragma unroll
                        for (int nj = 0; nj < NT; ++nj) part[(unsigned)m * (unsigned)a.K + (unsigned)(colw + nj * 32 + l31)] = acc[0][mi][nj][r];
                    }
                }
        } else {
            char* stg = smem + STG + wave * (SROWS * SROW_B);
            float* red = reinterpret_cast<float*>(smem + RED);
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
            constexpr int YSETS = EPI == 4 ? 2 : (BNB ? NSTEP : 1);
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
                if constexpr (EPI == 4) side_chunks(0, 0);
                else {
#pragma unroll
                    for (int s = 0; s < NSTEP; ++s) side_chunks(s, s);
                }
            }
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                constexpr int SPB = 32 / SROWS, RPS = SROWS / 2;                      // steps per 32-row block, accumulator registers per step
                const int mi = s / SPB;
                const int yset = EPI == 4 ? (s & 1) : s;
                if constexpr (EPI == 4) { if (s + 1 < NSTEP) side_chunks(s + 1, (s + 1) & 1); }
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
                    float* dst = a.stats + (size_t)(a.stat_row0 + ph * (ntiles / ntn) + mtile) * 2 * (size_t)a.K;
                    dst[n0 + tid] = u1;
                    dst[a.K + n0 + tid] = u2;
                }
                // (the next write of `red` lies behind at least the nine K-tile barriers of the next tile -- or, between the phases of
                //  a MODE 2 tile, behind this barrier)
                if constexpr (NPH > 1) { LBC_WAIT_LGKM0(); __builtin_amdgcn_s_barrier(); }
            }
            }       // ph
        }
        zero_acc();
        stores_pending = true;
        tile = tilen; mtile = mtilen; n0 = n0n; m0 = m0n;
#pragma unroll
        for (int i = 0; i < MT; ++i) amask[i] = amaskn[i];
    }
#undef LBC_RD
#undef LBC_RD1
#undef LBC_USE
#undef LBC_WAIT_OLDER_READS
#undef LBC_MM
}
#undef LBC_SG

// launches the instantiation for (mode, epilogue form) of one tile shape
template <int BM, int BN, int WM, int WN, int HRMAX, int SROWS>
int conv_hdmap_launch_shape(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, dim3 grid, hipStream_t s, int nsplit = 1, int kgroups = 1)
{
    if (nsplit > 1) {
        // split-K (EPI 3): one tile per workgroup, grid = ntiles * nsplit; the epilogue is the caller's second launch
        if constexpr (BM == 128 && BN == 64) {
            LBC_REQUIRE(a.split_ws && tpw == 1 && grid.x == (unsigned)(ntiles * nsplit) && (a.C / 64) % nsplit == 0, "conv_hdmap: bad split-K launch");
            if (mode == 0) hipLaunchKernelGGL((conv_hdmap_k<BM, BN, WM, WN, HRMAX, SROWS, 0, 3>), grid, dim3(WM * WN * 64), 0, s, a, zero, ntiles, tpw, nsplit);
            else hipLaunchKernelGGL((conv_hdmap_k<BM, BN, WM, WN, HRMAX, SROWS, 1, 3>), grid, dim3(WM * WN * 64), 0, s, a, zero, ntiles, tpw, nsplit);
            return lbc_check_launch("conv_hdmap(split)");
        } else {
            LBC_REQUIRE(false, "conv_hdmap: split-K exists for the 128 x 64 shape only");
        }
    }
    const int epi = a.bnb_y ? (a.bnb_mask ? 4 : 2) : (a.resid ? 1 : 0);
    LBC_REQUIRE(!a.pre_scale, "conv_hdmap: no BatchNorm-on-load form (measured slower on the MI355X, profiles/r05_call1_hdmap_pre_land_or_kill.txt)");
    if (mode == 2) {
        // the phased stride-2 transposed form: four-wave shape, plain epilogue
        if constexpr (BM == 128 && BN == 64) {
            LBC_REQUIRE(!a.resid && !a.bnb_y && a.nphase == 4 && kgroups == 1, "conv_hdmap: the phased transposed form has the plain epilogue only");
            hipLaunchKernelGGL((conv_hdmap_k<BM, BN, WM, WN, HRMAX, SROWS, 2, 0>), grid, dim3(WM * WN * 64), 0, s, a, zero, ntiles, tpw, 1);
            return lbc_check_launch("conv_hdmap(phased)");
        } else {
            LBC_REQUIRE(false, "conv_hdmap: the phased transposed form exists for the 128 x 64 shape only");
        }
    }
    if (kgroups == 2) {
        // in-workgroup K split (KG = 2): eight waves, one tile per workgroup
        if constexpr (BM == 128 && BN == 64) {
            LBC_REQUIRE(tpw == 1 && grid.x == (unsigned)ntiles && (a.C / 64) % 2 == 0, "conv_hdmap: bad K-split launch");
#define LBC_HK(MODEv, EPIv) hipLaunchKernelGGL((conv_hdmap_k<BM, BN, WM, WN, HRMAX, SROWS, MODEv, EPIv, 2>), grid, dim3(WM * WN * 128), 0, s, a, zero, ntiles, tpw, 1)
            if (mode == 0) {
                LBC_REQUIRE(epi != 2 && epi != 4, "conv_hdmap: the fused BatchNorm-backward reduce belongs to input-gradient launches");
                if (epi == 1) LBC_HK(0, 1); else LBC_HK(0, 0);
            } else {
                LBC_REQUIRE(epi != 4 || a.resid, "conv_hdmap: the tensor-masked BatchNorm-backward reduce is the residual form's (IgemmArgs::bnb_mask)");
                if (epi == 4) LBC_HK(1, 4); else if (epi == 2) LBC_HK(1, 2); else if (epi == 1) LBC_HK(1, 1); else LBC_HK(1, 0);
            }
#undef LBC_HK
            return lbc_check_launch("conv_hdmap(kg2)");
        } else {
            LBC_REQUIRE(false, "conv_hdmap: the in-workgroup K split exists for the 128 x 64 shape only");
        }
    }
#define LBC_HP(MODEv, EPIv) hipLaunchKernelGGL((conv_hdmap_k<BM, BN, WM, WN, HRMAX, SROWS, MODEv, EPIv>), grid, dim3(WM * WN * 64), 0, s, a, zero, ntiles, tpw, 1)
    if (mode == 0) {
        LBC_REQUIRE(epi != 2 && epi != 4, "conv_hdmap: the fused BatchNorm-backward reduce belongs to input-gradient launches");
        if (epi == 1) LBC_HP(0, 1); else LBC_HP(0, 0);
    } else {
        LBC_REQUIRE(epi != 4 || a.resid, "conv_hdmap: the tensor-masked BatchNorm-backward reduce is the residual form's (IgemmArgs::bnb_mask)");
        if (epi == 4) LBC_HP(1, 4); else if (epi == 2) LBC_HP(1, 2); else if (epi == 1) LBC_HP(1, 1); else LBC_HP(1, 0);
    }
#undef LBC_HP
    return lbc_check_launch("conv_hdmap");
}

}  // namespace
