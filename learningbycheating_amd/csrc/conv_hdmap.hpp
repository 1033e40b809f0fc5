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
// EPI 3 = split-K (round 4; launches with few tiles, i.e. the deep layers at the per-GPU batches of the 8-GPU run): workgroup p serves
// (tile p / nsplit, slab range p % nsplit), contracts C / nsplit of the gathered channels and leaves its accumulators as an f32 partial
// tile in IgemmArgs::split_ws [range][M][K]; conv_split_epilogue_k (conv_hdmap.hip) sums the ranges in a fixed order and does the
// epilogue of form 0 / 1 / 2.  One tile per workgroup (the launcher guarantees ntiles * nsplit workgroups).
// PRE (forward, plain epilogue; behind LBC_HDMAP_PRE=1 until it is measured): the producer's BatchNorm + ReLU applied to the input
// (IgemmArgs::pre_*; a block's conv2 reads the raw conv1 output, resnet.py:38-54) as an in-place transform of the staged halo.  The
// lane that requested a 16-byte piece of the NEXT slab's halo (8 channels of one row) transforms exactly those 16 bytes once they have
// landed -- no other lane is involved, so the slab's closing barrier is the only synchronization it needs: the piece requested in
// K-tile s is read back in the tail of K-tile s + 2 (the counted vmcnt wait in front of that K-tile's barrier covers it), transformed
// and written under the MFMAs of K-tile s + 3.  The halo pieces are therefore requested in the first three K-tiles of a slab (ATAPS 4),
// and the per-channel scale / shift sit in an LDS table (filled once; 2 x 8 floats per lane and slab re-read into registers).  The
// extra LDS operations sit between the hand-counted fragment reads: a counted lgkmcnt wait can only become stricter by them.
// PROF (diagnostic builds, LBC_HDMAP_PROF = device address of 8 x u64 per wave): s_memtime stamps around the waits of every K-tile;
// per wave: [0] K-tiles, [1] cycles in the three leading depth steps, [2] in the vmcnt wait, [3] in lgkmcnt(0) + barrier, [4] in the
// tail (step-0 reads of the next K-tile, last depth step, DMA issue), [5] in epilogues, [6] whole stream, [7] tiles
template <int BM, int BN, int WM, int WN, int HRMAX, int SROWS>
constexpr int hdmap_lds_bytes() { return 2 * HRMAX * 128 + 3 * BN * 128 + WM * WN * SROWS * ((BN / WN) * 2 + 16) + WM * 2 * BN * 4; }
// gathered channels the PRE form's LDS table has room for (the 384-row halo of layer 2 leaves 2 KB: 256 channels; so does the four-wave shape)
template <int BM, int BN, int WM, int WN, int HRMAX, int SROWS>
constexpr int hdmap_pre_channels()
{
    constexpr int used = hdmap_lds_bytes<BM, BN, WM, WN, HRMAX, SROWS>();
    constexpr int room = WM * WN == 4 ? 80 * 1024 : 160 * 1024;          // (the four-wave shape must stay at two workgroups per CU)
    return used + 4096 <= room ? 512 : (used + 2048 <= room ? 256 : (used + 1024 <= room ? 128 : 0));
}

template <int BM, int BN, int WM, int WN, int HRMAX, int SROWS, int MODE, int EPI, bool PROF = false, int VAR = 0, bool PRE = false>
__global__ __launch_bounds__(WM * WN * 64, ((BM / WM == 128 && WM * WN == 4) || (PRE && WM * WN == 8)) ? 1 : 2) void conv_hdmap_k(IgemmArgs a, const void* zero_page, const int ntiles, const int tpw, unsigned long long* prof, const int nsplit)
{
    constexpr int WTM = BM / WM, WTN = BN / WN;                 // per-wave output tile
    constexpr int MT = WTM / 32, NT = WTN / 32;
    constexpr int NW = WM * WN;                                 // waves per workgroup: 8 (256 x 128 / 128 x 256 tiles, one workgroup per CU) or
                                                                // 4 (128 x 64 tiles for launches with few rows: two workgroups per CU)
    // (MT = 4 with four waves: ONE wave per SIMD with a 128 x 64 tile and the whole register file -- 32 MFMAs per K-tile barrier, 24
    //  fragment reads, no second wave competing for the matrix pipe)
    static_assert((NW == 8 || NW == 4) && (NT == 2 || NT == 1) && (MT == 2 || MT == 4), "conv_hdmap: wave tiling");
    static_assert(HRMAX % (8 * NW) == 0 && BN % (8 * NW) == 0 && (SROWS == 8 || SROWS == 16), "conv_hdmap: staging");
    constexpr int NB = 3;                                       // weight ring depth: slot of K-tile (slab, tap) = tap % 3 (9 taps per slab)
    constexpr bool PRIO = (VAR & 1) != 0;                       // A/B variants: 1 = priority alternation, 2 = DMA pieces in the tail (one burst), 4 = reads interleaved with the MFMAs
    constexpr int KS = 4;                                       // depth steps of 16 channels per K-tile
    constexpr int ABYTES = HRMAX * 128;                         // one halo buffer: HRMAX rows x 64 channels
    constexpr int TILE_B = BN * 128;
    constexpr int BRING = 2 * ABYTES;                           // ring of NB weight tiles behind the two halo buffers
    constexpr int SROW_B = WTN * 2 + 16;                        // LDS pitch of a staged row (64 bf16 + 16 bytes); SROWS rows per copy-out step
    constexpr int STG = BRING + NB * TILE_B;                    // wave-private staging: NW x SROWS x SROW_B
    constexpr int RED = STG + NW * SROWS * SROW_B;              // [WM][2][BN] floats
    constexpr int PTAB = RED + WM * 2 * BN * 4;                 // PRE: [2][PRE_C] floats, scale then shift of the gathered channels
    constexpr int PRE_C = PRE ? hdmap_pre_channels<BM, BN, WM, WN, HRMAX, SROWS>() : 0;
    static_assert(!PRE || (MODE == 0 && EPI == 0 && PRE_C > 0 && (VAR & 8) != 0), "conv_hdmap: BatchNorm-on-load is a form of the plain forward");
    static_assert(PTAB == hdmap_lds_bytes<BM, BN, WM, WN, HRMAX, SROWS>(), "conv_hdmap: LDS layout");
    constexpr int SMEM = PTAB + 2 * PRE_C * 4;
    constexpr int ZROW = (HRMAX - 1) * 128;                     // last row of either halo buffer: beyond the halo, filled from the zero page
    static_assert(SMEM <= 160 * 1024, "conv_hdmap: LDS");
    __shared__ __attribute__((aligned(16))) char smem[SMEM];    // the ONLY LDS object
    constexpr int HPW = HRMAX / (8 * NW);                       // 1-KiB halo pieces (8 rows) per wave per slab
    constexpr int NBW = BN / (8 * NW);                          // 1-KiB weight pieces per wave per K-tile
    constexpr int ATAPS = PRE ? 4 : 7;                          // taps of a slab whose issue slot may carry halo pieces of the next slab (PRE: early enough to be transformed in place before the slab ends)
    constexpr int PPT = (HPW + ATAPS - 1) / ATAPS;              // halo pieces of the next slab requested per tap (taps 0 .. ATAPS - 1)
    static_assert(PPT >= 1 && PPT <= 2, "conv_hdmap: halo pieces per tap");
    constexpr int NSTEP = WTM / SROWS;                          // copy-out steps per wave and tile
    constexpr int SEGS = WTN / 8;                               // 16-byte segments per staged row
    constexpr int CPL = SROWS * SEGS / 64;                      // 16-byte chunks per lane and step
    constexpr int RPP = 64 / SEGS;                              // rows per copy-out pass of the wave
    constexpr int NST = NSTEP * CPL;                            // 16-byte store instructions per wave and tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int upper = wave / (NW / 2);      // the younger half of the workgroup's waves
    const int l31 = lane & 31, kh = lane >> 5;
    const int W = a.W, H = a.H, C = a.C;
    const int ntn = a.K / BN;
    const int nslab = EPI == 3 ? C / 64 / nsplit : C / 64;     // slabs (64 gathered channels) this workgroup contracts

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
    const __bf16* xin = static_cast<const __bf16*>(a.x) + (EPI == 3 ? split * nslab * 64 : 0);
    const __bf16* win = static_cast<const __bf16*>(a.w) + (EPI == 3 ? split * nslab * 64 : 0);
    const int prow = lane >> 3, pseg = lane & 7;

    // ---- DMA roles: per-thread constants + a wave-uniform origin per piece (the issue slots sit next to MFMAs that leave room
    //      for ~5 other instructions each: every VALU counts).  Halo row hr of a tile with origin m0 holds input pixel
    //      m0 - (W + 1) + hr.  Rows outside the tensor read a clamped pixel (they are only ever met by taps that the border select
    //      sends to the zero row); pieces that lie entirely past the halo (8 p >= BM + 2W + 2: the launcher guarantees that the
    //      last piece, which holds the ZERO ROW, is one of them) come from the zero page
    const int HR = BM + 2 * W + 2;
    const int arow0 = wave * HPW * 8 + prow;                                   // halo row of piece j: arow0 + 8 j
    const unsigned aswz[2] = {(unsigned)((pseg ^ ((arow0 >> 1) & 7)) * 16), (unsigned)((pseg ^ (((arow0 >> 1) + 4) & 7)) * 16)};   // j even / odd
    const unsigned zoff = (unsigned)((lane & 7) * 16);
    const char* xbytes = reinterpret_cast<const char*>(xin);
    const char* zbytes = static_cast<const char*>(zero_page);
    auto issue_a = [&](const int m0x, const int slab, const int buf, const int j) {
        const bool pad = (wave * HPW + j) * 8 >= HR;                           // wave-uniform
        int q = m0x - (W + 1) + arow0 + 8 * j;
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
    for (int i = 0; i < MT; ++i) rowc[i] = W + 1 + wm * WTM + i * 32 + l31;
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

    // per (tap, 32-row block): LDS address of the lane's depth-step-0 fragment in halo buffer `buf` (its halo row, or the zero row)
    int aaddr[MT];
    auto tap_addr = [&](const int tap, const int buf, const int (&mask)[MT]) {
        const int r = tap / 3, s = tap - 3 * r;
        const int off = MODE == 0 ? (r - 1) * W + (s - 1) : (1 - r) * W + (1 - s);
        const int abuf = buf * ABYTES;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int hr = rowc[i] + off;
            const int val = (hr << 7) | ((kh ^ ((hr >> 1) & 7)) << 4), zval = ZROW | (kh << 4);
            const int m = -((mask[i] >> tap) & 1);               // all ones when the tap is inside the image (written as a bit
            aaddr[i] = abuf + (((val ^ zval) & m) ^ zval);       // select: as `ok ? val : zval` the compiler branches over exec)
        }
    };

    bf16x8 fa[2][MT], fb[2][NT];            // two register sets: depth step g computes from set g & 1 while set (g + 1) & 1 is read
    // ASMRD (VAR & 8): the fragment reads are inline asm the compiler does not track, waited for with hand-counted lgkmcnt.  With LDS-DMA
    // in flight hipcc's own wait insertion puts `s_waitcnt lgkmcnt(0)` in front of every depth step's MFMAs -- which also waits for the
    // reads of the NEXT step issued just before it: a full LDS latency per depth step that only the SIMD's other wave can cover.  Here
    // the wait in front of step g leaves the MT + NT youngest reads (step g + 1) in flight; LBC_USE (an empty asm the MFMAs depend on)
    // keeps the MFMAs behind that wait.
    constexpr bool ASMRD = (VAR & 8) != 0 && NB * TILE_B + (NT - 1) * 4096 < 65536;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
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
#define LBC_MM(SET)                                                                                                              \
    do {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                           \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                       \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);                 \
    } while (0)

    // ---- PRE: in-place BatchNorm (+ ReLU) of the halo pieces this lane requested.  16-byte LDS accesses of the same kind as the fragment
    //      reads (inline asm with ASMRD: ordered among themselves and with the fragment reads, not tracked by the compiler)
#ifdef LBC_HIP_EMULATED_FOR_TESTS
#define LBC_LDS_RD16(DST, ADDR) DST = *reinterpret_cast<const std::remove_reference_t<decltype(DST)>*>(smem + (ADDR))
#define LBC_LDS_WR16(ADDR, SRC) *reinterpret_cast<std::remove_cv_t<std::remove_reference_t<decltype(SRC)>>*>(smem + (ADDR)) = (SRC)
#define LBC_FENCE(V) do { } while (0)
#else
#define LBC_LDS_RD16(DST, ADDR)                                                                                                  \
    do {                                                                                                                         \
        if constexpr (ASMRD) asm volatile("ds_read_b128 %0, %1" : "=v"(DST) : "v"(lds0 + (unsigned)(ADDR)));                     \
        else DST = *reinterpret_cast<const std::remove_reference_t<decltype(DST)>*>(smem + (ADDR));                                                       \
    } while (0)
#define LBC_LDS_WR16(ADDR, SRC)                                                                                                  \
    do {                                                                                                                         \
        if constexpr (ASMRD) asm volatile("ds_write_b128 %0, %1" : : "v"(lds0 + (unsigned)(ADDR)), "v"(SRC) : "memory");         \
        else *reinterpret_cast<std::remove_cv_t<std::remove_reference_t<decltype(SRC)>>*>(smem + (ADDR)) = (SRC);                                                           \
    } while (0)
#define LBC_FENCE(V) do { if constexpr (ASMRD) asm volatile("" : "+v"(V)); } while (0)
#endif
    // channel chunk (8 channels) of the lane's 16 bytes in pieces with even / odd index (the swizzle of the DMA source, aswz above)
    const int pcs[2] = {pseg ^ ((arow0 >> 1) & 7), pseg ^ (((arow0 >> 1) + 4) & 7)};
    f32x4 pco[PRE ? 2 : 1][4];              // [piece parity][scale lo, scale hi, shift lo, shift hi] of the slab being transformed
    const float pfloor = a.pre_relu ? 0.f : -INFINITY;
    auto pre_coef = [&](const int slab) {   // (consumed two K-tiles later at the earliest: behind at least one lgkmcnt(0) + barrier)
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int o = PTAB + (slab * 64 + pcs[par] * 8) * 4;
            LBC_LDS_RD16(pco[par][0], o); LBC_LDS_RD16(pco[par][1], o + 16);
            LBC_LDS_RD16(pco[par][2], o + PRE_C * 4); LBC_LDS_RD16(pco[par][3], o + PRE_C * 4 + 16);
        }
    };
    auto pre_piece_addr = [&](const int buf, const int j) { return buf * ABYTES + (wave * HPW + j) * 1024 + lane * 16; };
    auto pre_pad = [&](const int j) { return (wave * HPW + j) * 8 >= HR; };            // wave-uniform: the piece came from the zero page
    auto pre_xform = [&](const bf16x8 raw, const int par) {
        const f32x8 v = __builtin_convertvector(raw, f32x8);
        f32x4 lo = __builtin_shufflevector(v, v, 0, 1, 2, 3) * pco[par][0] + pco[par][2];
        f32x4 hi = __builtin_shufflevector(v, v, 4, 5, 6, 7) * pco[par][1] + pco[par][3];
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] = fmaxf(lo[e], pfloor); hi[e] = fmaxf(hi[e], pfloor); }
        return __builtin_convertvector(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7), bf16x8);
    };
    bf16x8 pxr[PRE ? PPT : 1];              // pieces read back in a K-tile's tail, transformed in the next K-tile

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
    static_assert(NBW + 2 * PPT + NST <= 24, "conv_hdmap: counted waits");

    // ---- the tile stream
    int tile = first;
    int mtile = tile / ntn, n0 = (tile - mtile * ntn) * BN, m0 = mtile * BM;
    int sg = 0;                             // slabs consumed so far: halo buffer sg & 1
    tap_mask(m0, amask);

    if constexpr (PRE) {                    // the coefficient table, visible to every wave before the first piece is transformed
        float* tab = reinterpret_cast<float*>(smem + PTAB);
        for (int i = tid; i < C; i += NW * 64) { tab[i] = a.pre_scale[i]; tab[PRE_C + i] = a.pre_shift[i]; }
        LBC_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
    }
    // prologue: the halo of slab 0 and the first two weight tiles in flight; everything of K-tile 0 landed and visible
#pragma unroll
    for (int j = 0; j < HPW; ++j) issue_a(m0, 0, 0, j);
    issue_b(n0, 0, 0, 0, 0, NBW);
    issue_b(n0, 0, 1, 1, 0, NBW);
    LBC_WAIT_VM(NBW);
    if constexpr (PRE) {                    // the first halo: all of this wave's pieces at once
        bf16x8 x0[HPW];
        pre_coef(0);
#pragma unroll
        for (int j = 0; j < HPW; ++j)
            if (!pre_pad(j)) LBC_LDS_RD16(x0[j], pre_piece_addr(0, j));
        LBC_WAIT_LGKM0();
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int q = 0; q < 4; ++q) LBC_FENCE(pco[par][q]);
#pragma unroll
        for (int j = 0; j < HPW; ++j)
            if (!pre_pad(j)) { LBC_FENCE(x0[j]); const bf16x8 t = pre_xform(x0[j], j & 1); LBC_LDS_WR16(pre_piece_addr(0, j), t); }
        LBC_WAIT_LGKM0();
    }
    __builtin_amdgcn_s_barrier();
    tap_addr(0, 0, amask);
    LBC_RD(0, 0, 0);
    bool stores_pending = false;            // the previous tile's output stores may still be in this wave's VMEM queue
    unsigned long long pf_steps = 0, pf_vm = 0, pf_bar = 0, pf_tail = 0, pf_epi = 0, pf_kt = 0, pf_t0 = 0, pf_prev = 0;
#ifdef LBC_HIP_EMULATED_FOR_TESTS
#define LBC_NOW() 0ull
#else
#define LBC_NOW() __builtin_amdgcn_s_memtime()
#endif
    if (PROF) { pf_t0 = LBC_NOW(); pf_prev = pf_t0; }

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
                    if (PRIO) {     // the two waves of a SIMD take turns at the matrix pipe (equal priority = the older wave always wins)
                        if ((g & 1) == upper) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                    }
                    LBC_RD(slot, g + 1, (g + 1) & 1);
                    // the reads of the last depth step are out: the addresses are free for the next K-tile's (tap, slab, tile)
                    if (g == KS - 2 && has_next) {
                        if (t < 8) tap_addr(t + 1, buf, amask);
                        else if (!LAST) tap_addr(0, buf ^ 1, amask);
                        else tap_addr(0, buf ^ 1, amaskn);
                    }
                    if (!(VAR & 2)) {
                        if (g == 0) issue_w(0, NBW / 2);
                        else if (g == 1) issue_w(NBW / 2, NBW);
                        else issue_h();
                    }
                    LBC_WAIT_OLDER_READS();                              // set g & 1 is in (its reads were issued a full step ago)
                    LBC_USE(g & 1);
                    if constexpr (PRE) {
                        // the pieces requested in K-tile t - 3, read back in the tail of K-tile t - 1 (older than this step's fragment reads:
                        // the wait above covers them): transform, write back -- complete by this K-tile's lgkmcnt(0) + barrier
                        if (g == 0 && t >= 3 && follows) {
#pragma unroll
                            for (int q = 0; q < PPT; ++q)
                                if ((t - 3) * PPT + q < HPW && !pre_pad((t - 3) * PPT + q)) {
                                    LBC_FENCE(pxr[q]);
                                    const bf16x8 tq = pre_xform(pxr[q], ((t - 3) * PPT + q) & 1);
                                    LBC_LDS_WR16(pre_piece_addr(buf ^ 1, (t - 3) * PPT + q), tq);
                                }
                        }
                    }
                    LBC_MM(g & 1);
                    if (VAR & 4) {
#pragma unroll
                        for (int q = 0; q < MT + NT; ++q) { LBC_SG(0x008, 1); LBC_SG(0x100, 1); LBC_SG(0x036, 4); }
                    } else {
                        // the step's fragment reads first: a full step of MFMAs (128 cycles of this wave's own, 256 with its SIMD
                        // partner) between a read and the wait that needs it -- a wave that runs alone no longer stalls on LDS latency
                        LBC_SG(0x100, MT + NT);
#pragma unroll
                        for (int q = 0; q < MT * NT; ++q) { LBC_SG(0x008, 1); LBC_SG(0x036, 5); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                unsigned long long pf_a = 0, pf_b = 0, pf_c = 0;
                if (PROF) { LBC_WAIT_LGKM0(); pf_a = LBC_NOW(); }
                // The weight tile of K-tile k + 1 (requested during K-tile k - 1) has landed, this wave's pieces.  Requested after it and
                // allowed to stay in flight: the halo piece of K-tile k - 1, this K-tile's weight tile and halo piece -- and, in the first
                // K-tile behind a tile boundary, the previous tile's stores (they sit between K-tile k - 1's requests and this one's)
                {
                    int n = 0;
                    if (VAR & 2) {      // burst variant: the requests of K-tile k - 1 came after its wait: [halo piece][weight tile k + 1] -> all landed
                        n = (c == 0 && t == 0 && stores_pending) ? NST : 0;
                    } else if (w2) {
                        n = NBW + (follows ? np_prev + np_here : 0);
                        if (c == 0 && t == 0 && stores_pending) n += NST;
                    }
                    wait_vm(n);
                }
                if (PROF) pf_b = LBC_NOW();
                LBC_WAIT_LGKM0();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (PROF) pf_c = LBC_NOW();
                if (PRIO) {
                    if (((KS - 1) & 1) == upper) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                }
                if (has_next) LBC_RD(nslot, 0, 0);
                if constexpr (PRE) {
                    if (follows) {
                        if (t == 0) pre_coef(LAST ? 0 : c + 1);          // (the previous slab's last piece was transformed in its K-tile 5)
                        if (t >= 2) {
                            // the pieces requested in K-tile t - 2 have landed (the vmcnt wait in front of this K-tile's barrier left only
                            // the requests of K-tiles t - 1 and t in flight): read them back, behind the next K-tile's step-0 fragments
#pragma unroll
                            for (int q = 0; q < PPT; ++q)
                                if ((t - 2) * PPT + q < HPW && !pre_pad((t - 2) * PPT + q)) LBC_LDS_RD16(pxr[q], pre_piece_addr(buf ^ 1, (t - 2) * PPT + q));
                        }
                    }
                }
                LBC_USE((KS - 1) & 1);                                   // in since the lgkmcnt(0) in front of the barrier
                LBC_MM((KS - 1) & 1);
                if (VAR & 2) {          // everything in one burst behind the barrier: K-tile k + 1's successor is K-tile k + 2 -> slot (t + 2) % 3
                    issue_h();
                    issue_w(0, NBW);
                }
                LBC_SG(0x100, MT + NT);
#pragma unroll
                for (int q = 0; q < MT * NT; ++q) { LBC_SG(0x008, 1); LBC_SG(0x036, 6); }
                __builtin_amdgcn_sched_barrier(0);
                if (PROF) {
                    const unsigned long long pf_d = LBC_NOW();
                    pf_steps += pf_a - pf_prev; pf_vm += pf_b - pf_a; pf_bar += pf_c - pf_b; pf_tail += pf_d - pf_c; pf_prev = pf_d; ++pf_kt;
                }
            }
            ++sg;
        };
        for (int c = 0; c + 1 < nslab; ++c) slab_body(c, std::false_type{});
        slab_body(nslab - 1, std::true_type{});

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
                        for (int nj = 0; nj < NT; ++nj) part[(unsigned)m * (unsigned)a.K + (unsigned)(colw + nj * 32 + l31)] = acc[mi][nj][r];
                    }
                }
        } else {
            char* stg = smem + STG + wave * (SROWS * SROW_B);
            float* red = reinterpret_cast<float*>(smem + RED);
            __bf16* yout = static_cast<__bf16*>(a.y);
            const __bf16* resid = EPI == 1 ? static_cast<const __bf16*>(a.resid) : nullptr;
            const __bf16* by = EPI == 2 ? static_cast<const __bf16*>(a.bnb_y) : nullptr;
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
            // every load of the epilogue is requested before its first store (a load behind a store would wait for the store)
            float rv[EPI == 1 ? MT : 1][16][NT];
            if constexpr (EPI == 1) {
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
            bf16x8 yv[EPI == 2 ? NSTEP : 1][CPL];
            f32x8 bsc, bsh, bmu, biv;
            f32x8 t1 = ParamVec<8>::splat(0.f), t2 = t1;
            if constexpr (EPI == 2) {
                const int c0 = colw + cseg * 8;
                bsc = ParamVec<8>::ld(a.bnb_scale + c0); bsh = ParamVec<8>::ld(a.bnb_shift + c0);
                bmu = ParamVec<8>::ld(a.bnb_mean + c0); biv = ParamVec<8>::ld(a.bnb_invstd + c0);
#pragma unroll
                for (int s = 0; s < NSTEP; ++s)
#pragma unroll
                    for (int q = 0; q < CPL; ++q) {
                        const int m = m0 + wm * WTM + s * SROWS + crow + RPP * q;
                        yv[s][q] = *reinterpret_cast<const bf16x8*>(by + ((unsigned)(m < a.M ? m : 0) * (unsigned)a.K + (unsigned)(colw + cseg * 8)));
                    }
            }
            float s1[NT], s2[NT];
#pragma unroll
            for (int nj = 0; nj < NT; ++nj) { s1[nj] = 0.f; s2[nj] = 0.f; }
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                constexpr int SPB = 32 / SROWS, RPS = SROWS / 2;                      // steps per 32-row block, accumulator registers per step
                const int mi = s / SPB;
#pragma unroll
                for (int r8 = 0; r8 < RPS; ++r8) {
                    const int r = (s % SPB) * RPS + r8;
                    const int lr = (r & 3) + 4 * kh + 8 * ((r >> 2) % (SROWS / 8));   // row inside the step
                    const bool live = m0 + wm * WTM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh < a.M;
#pragma unroll
                    for (int nj = 0; nj < NT; ++nj) {
                        float v = acc[mi][nj][r];
                        if (a.post_scale) v = v * psc[nj] + psh[nj];
                        if (a.bias) v += bia[nj];
                        if constexpr (EPI == 1) v += rv[mi][r][nj];
                        if (a.relu) v = fmaxf(v, 0.f);
                        *reinterpret_cast<__bf16*>(stg + lr * SROW_B + (nj * 32 + l31) * 2) = (__bf16)v;
                        if (EPI != 2 && live) { s1[nj] += v; s2[nj] += v * v; }
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
                    if constexpr (EPI == 2) {
                        // fused BatchNorm-backward reduce (IgemmArgs::bnb_*): mask the stored gradient with bn(y) > 0, sum (g, g * xhat)
                        const f32x8 yf = __builtin_convertvector(yv[s][q], f32x8);
                        f32x8 g = __builtin_convertvector(ch, f32x8);
                        const f32x8 z = yf * bsc + bsh;
#pragma unroll
                        for (int e = 0; e < 8; ++e) g[e] = z[e] > 0.f ? g[e] : 0.f;
                        ch = __builtin_convertvector(g, bf16x8);
                        if (m < a.M) { t1 += g; t2 += g * (yf - bmu) * biv; }
                    }
                    if (m < a.M) *reinterpret_cast<bf16x8*>(yout + ((unsigned)m * (unsigned)a.K + (unsigned)(colw + cseg * 8))) = ch;
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (a.stats) {
                if constexpr (EPI == 2) {
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
                if (tid < BN) {
                    float u1 = 0.f, u2 = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < WM; ++w2) { u1 += red[(w2 * 2 + 0) * BN + tid]; u2 += red[(w2 * 2 + 1) * BN + tid]; }
                    float* dst = a.stats + (size_t)(a.stat_row0 + mtile) * 2 * (size_t)a.K;
                    dst[n0 + tid] = u1;
                    dst[a.K + n0 + tid] = u2;
                }
                // (the next write of `red` lies behind at least the nine K-tile barriers of the next tile)
            }
        }
        zero_acc();
        if (PROF) { const unsigned long long t = LBC_NOW(); pf_epi += t - pf_prev; pf_prev = t; }
        stores_pending = true;
        tile = tilen; mtile = mtilen; n0 = n0n; m0 = m0n;
#pragma unroll
        for (int i = 0; i < MT; ++i) amask[i] = amaskn[i];
    }
    if (PROF && prof && lane == 0) {
        unsigned long long* o = prof + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = pf_kt; o[1] = pf_steps; o[2] = pf_vm; o[3] = pf_bar; o[4] = pf_tail; o[5] = pf_epi; o[6] = LBC_NOW() - pf_t0; o[7] = (unsigned long long)cnt;
    }
#undef LBC_NOW
#undef LBC_LDS_RD16
#undef LBC_LDS_WR16
#undef LBC_FENCE
#undef LBC_RD
#undef LBC_RD1
#undef LBC_USE
#undef LBC_WAIT_OLDER_READS
#undef LBC_MM
}
#undef LBC_SG

// launches the instantiation for (mode, epilogue form) of one tile shape
template <int BM, int BN, int WM, int WN, int HRMAX, int SROWS>
int conv_hdmap_launch_shape(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, dim3 grid, hipStream_t s, int nsplit = 1)
{
    if (nsplit > 1) {
        // split-K (EPI 3): one tile per workgroup, grid = ntiles * nsplit; the epilogue is the caller's second launch
        if constexpr (BM == 128 && BN == 64) {
            LBC_REQUIRE(a.split_ws && tpw == 1 && grid.x == (unsigned)(ntiles * nsplit) && (a.C / 64) % nsplit == 0, "conv_hdmap: bad split-K launch");
            if (mode == 0) hipLaunchKernelGGL((conv_hdmap_k<BM, BN, WM, WN, HRMAX, SROWS, 0, 3, false, 8>), grid, dim3(WM * WN * 64), 0, s, a, zero, ntiles, tpw, (unsigned long long*)nullptr, nsplit);
            else hipLaunchKernelGGL((conv_hdmap_k<BM, BN, WM, WN, HRMAX, SROWS, 1, 3, false, 8>), grid, dim3(WM * WN * 64), 0, s, a, zero, ntiles, tpw, (unsigned long long*)nullptr, nsplit);
            return lbc_check_launch("conv_hdmap(split)");
        } else {
            LBC_REQUIRE(false, "conv_hdmap: split-K exists for the 128 x 64 shape only");
        }
    }
    const int epi = a.bnb_y ? 2 : (a.resid ? 1 : 0);
    if (a.pre_scale) {
        constexpr int PC = hdmap_pre_channels<BM, BN, WM, WN, HRMAX, SROWS>();
        if constexpr (PC > 0) {
            LBC_REQUIRE(mode == 0 && epi == 0 && a.C <= PC, "conv_hdmap: BatchNorm-on-load is a form of the plain forward with at most %d gathered channels", PC);
            hipLaunchKernelGGL((conv_hdmap_k<BM, BN, WM, WN, HRMAX, SROWS, 0, 0, false, 8, true>), grid, dim3(WM * WN * 64), 0, s, a, zero, ntiles, tpw, (unsigned long long*)nullptr, 1);
            return lbc_check_launch("conv_hdmap(pre)");
        } else {
            LBC_REQUIRE(false, "conv_hdmap: this tile shape has no LDS left for the BatchNorm-on-load table");
        }
    }
    // A/B variants (plain forward only); 16 = the shipped schedule with compiler-managed fragment reads instead of the asm ones
    const long long var = lbc_opt(kOptHdmapVar) > 0 ? lbc_opt(kOptHdmapVar) : 0;
    unsigned long long* prof = lbc_opt(kOptHdmapProf) > 0 ? reinterpret_cast<unsigned long long*>((uintptr_t)lbc_opt(kOptHdmapProf)) : nullptr;
    if ((prof || var) && mode == 0 && epi == 0) {
#define LBC_HV(PROFv, VARv) hipLaunchKernelGGL((conv_hdmap_k<BM, BN, WM, WN, HRMAX, SROWS, 0, 0, PROFv, VARv>), grid, dim3(WM * WN * 64), 0, s, a, zero, ntiles, tpw, prof, 1)
        if (prof) { if (var == 16) LBC_HV(true, 0); else if (var == 2) LBC_HV(true, 2); else if (var == 4) LBC_HV(true, 4); else LBC_HV(true, 8); }
        else      { if (var == 16) LBC_HV(false, 0); else if (var == 1) LBC_HV(false, 1); else if (var == 2) LBC_HV(false, 2); else if (var == 4) LBC_HV(false, 4); else LBC_HV(false, 8); }
#undef LBC_HV
        return lbc_check_launch("conv_hdmap");
    }
#define LBC_HP(MODEv, EPIv) hipLaunchKernelGGL((conv_hdmap_k<BM, BN, WM, WN, HRMAX, SROWS, MODEv, EPIv, false, 8>), grid, dim3(WM * WN * 64), 0, s, a, zero, ntiles, tpw, (unsigned long long*)nullptr, 1)
    if (mode == 0) {
        LBC_REQUIRE(epi != 2, "conv_hdmap: the fused BatchNorm-backward reduce belongs to input-gradient launches");
        if (epi == 1) LBC_HP(0, 1); else LBC_HP(0, 0);
    } else {
        if (epi == 2) LBC_HP(1, 2); else if (epi == 1) LBC_HP(1, 1); else LBC_HP(1, 0);
    }
#undef LBC_HP
    return lbc_check_launch("conv_hdmap");
}

}  // namespace
