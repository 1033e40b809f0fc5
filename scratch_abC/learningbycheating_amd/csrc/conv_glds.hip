// 3x3 / 1x1 stride-1 convolution (forward and input gradient) for bf16 tensors + bf16 weight copies on gfx950, as an
// implicit GEMM on 256 x 256 / 512 x 128 / 128 x 256 / 256 x 128 output tiles with 8 waves per workgroup and LDS-DMA staging
// (global_load_lds_dwordx4): the wide residual layers (layer 2-4) of the ResNets at training batch sizes.
// reference arithmetic: BasicBlock conv1 / conv2, bird_view/models/resnet.py:15-22,38-54, and their autograd.
//
// Why a third kernel next to conv_igemm.hip (128 x 128, 4 waves, register staging) -- measured on that kernel at batch 256:
// MFMA busy 0.21-0.24; a 64-channel depth chunk of a 128 x 128 tile stages 32 KB through VGPRs and ds_write_b128
// (~80 B/clk/CU) for 16 MFMAs per wave, every chunk ends in a barrier that waits for the chunk's global loads, and the
// four waves of a workgroup run in lockstep, so the matrix pipe idles while fragments are read.  Here
//   * a tile is 4x (2x) larger: half the staged bytes per MFMA, and 8 waves = two per SIMD;
//   * operands go HBM/L2 -> LDS by DMA: no staging registers, no ds_write issue, and the loads stay in flight across
//     barriers (counted s_waitcnt vmcnt, raw s_barrier);
//   * the two waves of a SIMD run half a phase apart (the second group of four waves passes one extra barrier up
//     front): while one wave issues its 8 MFMAs (256 cycles) the other reads its next fragments and issues DMA pieces;
//   * LDS rows are 128 bytes (64 bf16) with the 16-byte segment index XOR-ed with (row >> 1) & 7 -- applied to the
//     SOURCE address of the DMA (the LDS image of a DMA is lane-linear) and to the fragment read -- which makes every
//     ds_read_b128 lane group hit 16 distinct 16-byte slots.
// Zero padding: taps that fall outside the image DMA from a 256-byte zero page instead (an LDS-DMA cannot be masked).
// A depth step (K-tile) = one filter tap x 64 channels, channel slab outer / taps inner so that the nine shifted reads of
// a slab hit in L2.  No BatchNorm-on-load prologue (a DMA cannot transform): launches that need one keep conv_igemm.hip.
//
// (The first-generation kernel of this file -- two barriers per 8 MFMAs, conv_glds_k -- lost to the one below in round 2 and was
// removed in round 5 with its LBC_GLDS_V1 switch, like the timing-experiment builds LBC_GLDS_DIAG and the early-read variant.)
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include "conv_lds_dma.hpp"

namespace {

// Measured on the first generation at batch 256 (round 2): with the DMA stream removed a 256 x 256 launch still needed 68 of 77 us,
// the 2-byte output stores cost 12 us and the barrier-separated fragment reads 14 us -- the matrix pipe waited on the phase
// structure (two barriers per 8 MFMAs), not on memory.  Here
//   * ONE barrier per K-tile (one filter tap x 32 or 64 channels: 16 / 32 MFMAs per wave); a DMA piece has ~2000 MFMA cycles to
//     land (four 32-channel tiles or two 64-channel tiles in the ring).  A wave's fragment reads for the next depth step are
//     issued between the MFMAs of the current one (two register sets), so the barrier finds every wave with its next
//     fragments in registers;
//   * the output tile goes through LDS: bf16 rows, then 16-byte coalesced stores (was 128 two-byte stores per lane).
#define LBC_SG(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
// PH = 1 (MODE 1 only): the four output-parity phases of a stride-2 transposed launch in one grid (input gradient of the stride-2
// 3x3 convolutions, ConvTranspose2d forward): workgroups [ph * n, (ph + 1) * n) serve phase ph = 2 oy0 + ox0, whose output pixels
// (2 ly + oy0, 2 lx + ox0) gather x at (ly + dy, lx + dx) through the 1 / 2 / 2 / 4 taps with (oy0 + 1 - r, ox0 + 1 - s) even.
// WM x WN = 8 waves (one workgroup per CU), or 4 (256 x 64 / 128 x 128 tiles in half the LDS: TWO workgroups per CU, for the launches
// whose K loop is a handful of K-tiles -- the phased stride-2 transposed ones, the 1x1 downsamples -- where a workgroup is mostly
// prologue and epilogue and a second one has something to overlap them with)
template <int BM, int BN, int WM, int WN, int MODE, int KT, int PH = 0>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_glds2_k(IgemmArgs a, const void* zero_page)
{
    constexpr int WTM = BM / WM, WTN = BN / WN;                 // per-wave output tile
    constexpr int MT = WTM / 32, NT = WTN / 32;
    constexpr int NW = WM * WN;
    static_assert((NW == 8 || NW == 4) && NT == 2 && (MT == 2 || MT == 4), "conv_glds2: wave tiling");
    // K-tile = one filter tap x KT channels.  KT = 32: 64-byte LDS rows, four tiles in the ring.  KT = 64: 128-byte rows = whole
    // cache lines per DMA row (a 64-byte row leaves half of every 128-byte line it pulls through L2 -> L1 unused; the same
    // line comes again nine K-tiles later), two tiles in the ring, half as many barriers; same prefetch distance in cycles.
    static_assert(KT == 32 || KT == 64, "conv_glds2: K-tile depth");
    constexpr int ROWB = KT * 2;                                // bytes per LDS row
    constexpr int NBUF = KT == 32 ? 4 : 2;
    constexpr int KS = KT / 16;                                 // depth steps (one MFMA k) per K-tile
    constexpr int SEGS = ROWB / 16;                             // 16-byte segments per row
    constexpr int PROWS = 1024 / ROWB;                          // rows per 1-KiB DMA piece
    constexpr int TILE_A = BM * ROWB, TILE_B = BN * ROWB, BUF = TILE_A + TILE_B;
    // 1-KiB DMA pieces per wave per K-tile.  When the weight tile has fewer than eight pieces (BN = 64) the upper waves re-issue
    // the pieces of the lower ones (same bytes to the same LDS rows), which keeps every wave's vmcnt arithmetic equal
    constexpr int NA = BM / (PROWS * NW), NB = BN >= PROWS * NW ? BN / (PROWS * NW) : 1, NL = NA + NB;
    constexpr int BWAVES = BN >= PROWS * NW ? NW : BN / PROWS;
    static_assert(BM % (PROWS * NW) == 0 && (BN % (PROWS * NW) == 0 || BN == 64), "conv_glds2: tile extents");
    constexpr int OROW = BN * 2 + 16;                           // staged output row: BN bf16 + 16 bytes (rows 4 apart on distinct banks)
    constexpr int STAGE = BM * OROW;
    constexpr int RED = WM * 2 * BN * 4;
    constexpr int SMEM = NBUF * BUF > STAGE + RED ? NBUF * BUF : STAGE + RED;
    static_assert(SMEM <= (NW == 4 ? 80 : 160) * 1024, "conv_glds2: LDS");
    __shared__ __attribute__((aligned(16))) char smem[SMEM];    // the ONLY LDS object

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;
    const int W = a.W, H = a.H, C = a.C, T = a.KH * a.KW, KW = a.KW, PAD = a.P;

    const int ntn = a.K / BN;
    int tile_id, oy0 = 0, ox0 = 0, stat_tile0 = 0;
    {
        int nwg = gridDim.x, b = blockIdx.x;
        if (PH) {
            nwg = gridDim.x >> 2;
            const int ph = blockIdx.x / nwg;
            b = blockIdx.x - ph * nwg;
            oy0 = ph >> 1; ox0 = ph & 1;
            stat_tile0 = ph * (nwg / ntn);                   // statistics rows: phase-major
        }
        const int xcd = b & 7, q = nwg >> 3, rr = nwg & 7;
        tile_id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
    }
    const int mtile = tile_id / ntn;
    const int m0 = mtile * BM;
    const int n0 = (tile_id - mtile * ntn) * BN;

    const __bf16* xin = static_cast<const __bf16*>(a.x);
    const __bf16* win = static_cast<const __bf16*>(a.w);
    const __bf16* zero = static_cast<const __bf16*>(zero_page) + (lane & 7) * 8;

    // 16-byte slot XOR of a row: rows 4 (64-byte rows) / 2 (128-byte rows) apart differ, so that every ds_read_b128 lane group
    // (16 consecutive rows, one segment) hits 16 distinct slots
    auto rowswz = [](int row) { return KT == 32 ? (row >> 2) & 3 : (row >> 1) & 7; };
    // PH: the phase's taps: element shift of the gathered pixel and weight offset per tap (wave-uniform)
    int ph_ntap = 0, ph_shift[4] = {0, 0, 0, 0}, ph_koff[4] = {0, 0, 0, 0}, ph_dy[4] = {0, 0, 0, 0}, ph_dx[4] = {0, 0, 0, 0};
    if (PH) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int r = t / 3, sx = t - 3 * r;
            if ((((oy0 + 1 - r) & 1) == 0) && (((ox0 + 1 - sx) & 1) == 0)) {
                const int dy = (oy0 + 1 - r) >> 1, dx = (ox0 + 1 - sx) >> 1;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q == ph_ntap) { ph_shift[q] = (dy * W + dx) * C; ph_koff[q] = t * C; ph_dy[q] = dy; ph_dx[q] = dx; }
                ++ph_ntap;
            }
        }
    }
    // ---- DMA roles: piece (wave * NA + j) of the A tile = PROWS rows, lane -> (row = lane / SEGS, segment = lane % SEGS)
    const int prow = lane / SEGS, pseg = lane % SEGS;
    int aoff[NA], amask[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int row = (wave * NA + j) * PROWS + prow;
        const int m = m0 + row;
        int bits = 0;
        // output pixel m = (n, oy, ox) reads the gathered tensor around (oy * S, ox * S): S = 2 (forward of the stride-2 convolutions,
        // input gradient of the transposed convolutions) only in the gather mode
        const int mm = m < a.M ? m : 0;
        const int ox = PH ? mm % a.LW : mm % a.OW;              // PH: lattice coordinates (ly, lx)
        const int oy = PH ? (mm / a.LW) % a.LH : (mm / a.OW) % a.OH;
        const int n = PH ? mm / (a.LW * a.LH) : mm / (a.OW * a.OH);
        const int x = PH ? ox : ox * a.S, y = PH ? oy : oy * a.S;
        if (PH) {
            if (m < a.M) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < ph_ntap && y + ph_dy[q] < H && x + ph_dx[q] < W) bits |= 1 << q;
            }
        } else if (m < a.M) {
            for (int t = 0; t < T; ++t) {
                const int r = t / KW, s = t - r * KW;
                const int dy = MODE == 0 ? r - PAD : PAD - r;
                const int dx = MODE == 0 ? s - PAD : PAD - s;
                if ((unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W) bits |= 1 << t;
            }
        }
        amask[j] = bits;
        // swizzle on the SOURCE: LDS slot (row, s) holds segment s ^ rowswz(row)
        aoff[j] = ((n * H + y) * W + x) * C + (pseg ^ rowswz(row)) * 8;
    }
    int boff[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = ((wave % BWAVES) * NB + j) * PROWS + prow;
        boff[j] = (n0 + row) * (T * C) + (pseg ^ rowswz(row)) * 8;
    }
    // ---- fragment roles: row l31 of a 32-row block, depth step g (16 channels), half kh: slot (2g + kh) ^ rowswz(l31)
    const int swz = rowswz(l31);
    int koff[KS];
#pragma unroll
    for (int g = 0; g < KS; ++g) koff[g] = ((2 * g + kh) ^ swz) << 4;
    const int aBase = (wm * WTM + l31) * ROWB;
    const int bBase = TILE_A + (wn * WTN + l31) * ROWB;

    const int cpt = C / KT;
    const int ntaps = PH ? ph_ntap : T;
    const int nit = ntaps * cpt;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // the DMA stream walks K-tiles in order: (KT-channel slab, tap), taps inner (the nine shifted reads of a slab hit in L2)
    int is_ti = 0, is_s = 0, is_buf = 0;
    int is_shift = PH ? ph_shift[0] : (MODE == 0 ? -(PAD * W + PAD) : PAD * W + PAD) * C;     // element offset of the first tap, slab 0
    int is_koffs = PH ? ph_koff[0] : 0;
    int is_slab = 0;
    auto issue = [&]() {
        char* base = smem + is_buf * BUF;
#pragma unroll
        for (int j = 0; j < NB; ++j) lds_dma16(win + (boff[j] + is_koffs), base + TILE_A + ((wave % BWAVES) * NB + j) * 1024);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const bool ok = ((amask[j] >> is_ti) & 1) != 0;
            const __bf16* src = ok ? xin + (aoff[j] + is_shift) : zero;
            lds_dma16(src, base + (wave * NA + j) * 1024);
        }
        // next K-tile: tap (r, s) -> (r, s + 1) -> (r + 1, 0) -> next slab, tap (0, 0).  (With KT = 32, walking the two halves of
        // a 128-byte line back to back instead was measured 4-6 % slower at batch 256.)
        if (PH) {
            // next tap of the phase's list, then the next slab
            ++is_ti;
            if (is_ti == ph_ntap) { is_ti = 0; is_slab += KT; }
            const int sh = is_ti == 0 ? ph_shift[0] : is_ti == 1 ? ph_shift[1] : is_ti == 2 ? ph_shift[2] : ph_shift[3];
            const int ko = is_ti == 0 ? ph_koff[0] : is_ti == 1 ? ph_koff[1] : is_ti == 2 ? ph_koff[2] : ph_koff[3];
            is_shift = sh + is_slab; is_koffs = ko + is_slab;
        } else {
            const int step = MODE == 0 ? C : -C;
            ++is_ti; ++is_s;
            is_shift += step; is_koffs += C;
            if (is_s == KW) { is_s = 0; is_shift += step * (W - KW); }
            if (is_ti == T) {
                is_ti = 0;
                is_shift += KT - step * (W * a.KH);
                is_koffs += KT - T * C;
            }
        }
        is_buf = (is_buf + 1) & (NBUF - 1);
    };

    bf16x8 fa[2][MT], fb[2][NT];        // two register sets: depth step g computes from set g & 1 while set (g + 1) & 1 is read
#define LBC_RD(bufp, G, SET)                                                                                         \
    do {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) fa[SET][i] = *reinterpret_cast<const bf16x8*>((bufp) + aBase + i * 32 * ROWB + koff[G]); \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) fb[SET][j] = *reinterpret_cast<const bf16x8*>((bufp) + bBase + j * 32 * ROWB + koff[G]); \
    } while (0)
#define LBC_MM(SET)                                                                                                  \
    do {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                               \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                           \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);     \
    } while (0)
    // one MFMA, one fragment read, ...: the reads of the next depth step between the MFMAs of the current one
#define LBC_MIX()                                                                                                    \
    do {                                                                                                             \
        _Pragma("unroll") for (int k = 0; k < MT + NT; ++k) { LBC_SG(0x008, 1); LBC_SG(0x100, 1); }                  \
        if (MT * NT > MT + NT) LBC_SG(0x008, MT * NT - (MT + NT));                                                   \
    } while (0)
    // wait until at most `tiles` K-tiles of this wave's DMA pieces are outstanding (s_waitcnt takes an immediate)
    auto wait_tiles = [&](int tiles) {
        if (tiles >= 3) LBC_WAIT_VM(3 * NL);
        else if (tiles == 2) LBC_WAIT_VM(2 * NL);
        else if (tiles == 1) LBC_WAIT_VM(NL);
        else LBC_WAIT_VM(0);
    };

    // ---- prologue: the ring full (up to NBUF K-tiles in flight), tile 0 landed and visible, its first fragments in registers
#pragma unroll
    for (int i = 0; i < NBUF; ++i)
        if (i < nit) issue();
    wait_tiles((nit < NBUF ? nit : NBUF) - 1);
    __builtin_amdgcn_s_barrier();
    LBC_RD(smem, 0, 0);

    // Synchronisation of K-tile t (buffer t % NBUF), once per tile, in front of its LAST depth step:
    //   s_waitcnt vmcnt: own pieces of tile t + 1 landed (NBUF - 2 younger tiles may stay in flight);  lgkmcnt(0);  s_barrier
    //   then: DMA pieces of tile t + NBUF -> the buffer of tile t;  reads (t + 1, step 0) | MFMAs (t, last step)
    //   RAW: tile t + 1 is read only after this barrier, which every wave enters after its pieces of t + 1 have landed.
    //   WAR: the buffer of tile t is refilled after this barrier; its last reads (step KS - 1, issued during step KS - 2) were
    //        retired by the lgkmcnt(0) in front of it.
    // ---- steady state: tiles that still have a tile t + NBUF to issue: straight-line body, no conditionals, the DMA address
    //      arithmetic and the fragment reads spread between the MFMAs
    int t = 0;
    for (; t + NBUF < nit; ++t) {
        const int ob = (t & (NBUF - 1)) * BUF, on = ((t + 1) & (NBUF - 1)) * BUF;
        const char* bb = smem + ob;
        const char* bn = smem + on;
#pragma unroll
        for (int g = 0; g + 1 < KS; ++g) {
            LBC_RD(bb, g + 1, (g + 1) & 1);
            LBC_MM(g & 1);
            LBC_MIX();
            __builtin_amdgcn_sched_barrier(0);    // the MFMAs of a step stay inside it: nothing is scheduled across the barrier below
        }
        LBC_WAIT_VM((NBUF - 2) * NL);
        LBC_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        LBC_RD(bn, 0, 0);
        LBC_MM((KS - 1) & 1);
        issue();
#pragma unroll
        for (int k = 0; k < MT * NT; ++k) {
            LBC_SG(0x008, 1);
            if (k < MT + NT) LBC_SG(0x100, 1);
            LBC_SG(0x036, 10);                    // VALU | SALU | VMEM: the address arithmetic and DMA pieces of tile t + NBUF
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- the last (up to) NBUF tiles: nothing left to issue
    for (; t < nit; ++t) {
        const int ob = (t & (NBUF - 1)) * BUF, on = ((t + 1) & (NBUF - 1)) * BUF;
        const char* bb = smem + ob;
        const char* bn = smem + on;
#pragma unroll
        for (int g = 0; g + 1 < KS; ++g) {
            LBC_RD(bb, g + 1, (g + 1) & 1);
            LBC_MM(g & 1);
            LBC_MIX();
            __builtin_amdgcn_sched_barrier(0);
        }
        const int left = nit - t - 2;             // tiles younger than t + 1 that were issued
        wait_tiles(left < 0 ? 0 : (left > NBUF - 2 ? NBUF - 2 : left));
        LBC_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        if (t + 1 < nit) LBC_RD(bn, 0, 0);
        LBC_MM((KS - 1) & 1);
    }
#undef LBC_RD
#undef LBC_MM
#undef LBC_MIX

    // ---- epilogue (conv_lds_dma.hpp): affine / bias / residual / ReLU, LDS-staged 16-byte stores, statistics / fused BN-backward reduce
    lds_dma_epilogue<BM, BN, WM, WN, MT, NT>(a, acc, smem, m0, n0, stat_tile0 + mtile, PH ? 2 : 1, oy0, ox0);
}
#undef LBC_SG

struct GldsCfg { int bm, bn; double eff; };
// cfg ids kLbcCfgGlds + 0 .. 3; eff = measured relative MFMA efficiency of a full round of tiles (MI355X, batch 256)
// 5, 6: the four-wave shapes (two workgroups per CU); chosen by lbc_conv_glds_pick for short K loops only
const GldsCfg kGldsCfg[kLbcGldsCfgs] = {{256, 256, 1.0}, {256, 128, 0.62}, {128, 256, 0.95}, {512, 128, 1.0}, {512, 64, 1.0}, {256, 64, 1.02}, {128, 128, 1.02}};

}  // namespace

// the four-phase stride-2 transposed launches (input gradient of the stride-2 3x3 convolutions, ConvTranspose2d forward) that
// conv_glds2_k<.., PH = 1> serves: all four parity phases in one grid, no residual, no BatchNorm-on-load
static bool lbc_glds_phased(const IgemmArgs& a, int mode)
{
    return mode == 1 && a.nphase == 4 && a.S == 2 && a.ostep == 2 && a.KH == 3 && a.KW == 3 && a.P == 1 && a.H == a.LH && a.W == a.LW &&
           a.OH == 2 * a.LH && a.OW == 2 * a.LW && a.M == a.N * a.LH * a.LW && !a.resid && !a.bnb_y && !lbc_opt_on(kOptNoGldsPhased);
}

// The four-wave shapes (two workgroups per CU) take a launch only where measured better (256 images, profiles/r03_run34_glds_four_wave_*):
// the phased stride-2 transposed launches gain a little (layer 2's first input gradient 133 -> 115 us on 256 x 64, layers 3 / 4
// 76 / 65 -> 70 / 63 on 128 x 128; step -0.07 ms); the stride-2 forwards lose 10-30 %, the 1x1 downsamples are level -- a second
// workgroup per CU is not what those short-K launches lack
static bool lbc_glds_wants_four_waves(bool phased) { return phased; }

// Tile configuration for a launch, or -1 when the launch keeps conv_igemm.hip / conv_halo.hip.
int lbc_conv_glds_pick(const IgemmArgs& a, int mode)
{
    if (lbc_opt_on(kOptNoGemm256)) return -1;
    if (!(a.w_bf16 && a.act_bf16) || a.pre_scale || (mode != 0 && mode != 1)) return -1;
    if ((long long)a.N * a.H * a.W * a.C >= (1ll << 31) || a.C % 64) return -1;
    const bool phased = lbc_glds_phased(a, mode);
    if (!phased) {
        if (a.ostep != 1 || a.nphase > 1 || a.oy0 || a.ox0) return -1;
        if (a.KH != a.KW || (a.KH != 3 && a.KH != 1) || a.P != (a.KH - 1) / 2) return -1;
        if (a.M != a.N * a.OH * a.OW) return -1;
        if (a.S == 1) { if (a.H != a.OH || a.W != a.OW) return -1; }
        else {
            // stride 2: gather mode of the second-generation kernel only (forward of the stride-2 convolutions and downsamples,
            // input gradient of the transposed convolutions)
            if (a.S != 2 || mode != 0 || a.OH != (a.H + 2 * a.P - a.KH) / 2 + 1 || a.OW != (a.W + 2 * a.P - a.KW) / 2 + 1) return -1;
        }
    }
    // One workgroup per CU: a tile shape qualifies when it fills at least three quarters of the 256 CUs; among the shapes
    // that do, the one with the best (round quantisation x per-shape efficiency) wins.
    const long long fill = lbc_opt(kOptGemm256MinTiles) > 0 ? lbc_opt(kOptGemm256MinTiles) : 192;
    const long long forced = lbc_opt(kOptGemm256Cfg);       // tests / tuning: pin one shape
    int best = -1;
    double best_score = 0.0;
    for (int i = 0; i < kLbcGldsCfgs; ++i) {
        const GldsCfg& c = kGldsCfg[i];
        if (a.K % c.bn) continue;
        if (forced >= 0 && forced != i) continue;
        if (i >= 5 && forced != i && !lbc_glds_wants_four_waves(phased)) continue;
        // 64 output channels: the stride-1 3x3 layer is better off in conv_halo.hip (0.187 vs 0.126 ms), so this shape is chosen only when
        // pinned -- or for the phased stride-2 transposed launches (layer 2's first input gradient: 0.173 -> 0.135 ms)
        if (c.bn == 64 && (a.K != 64 || (forced != i && !phased))) continue;
        const long long tiles = (long long)lbc_cdiv(a.M, c.bm) * (a.K / c.bn) * (phased ? 4 : 1);
        if (tiles < fill) continue;
        const double score = c.eff * (double)tiles / (double)(((tiles + 255) / 256) * 256);
        if (score > best_score) { best_score = score; best = i; }
    }
    return best < 0 ? -1 : kLbcCfgGlds + best;
}

int lbc_conv_glds_rows(const IgemmArgs& a, int cfg) { return lbc_cdiv(a.M, kGldsCfg[cfg - kLbcCfgGlds].bm); }

int lbc_conv_glds_launch(const IgemmArgs& a, int mode, int cfg, hipStream_t s)
{
    LBC_REQUIRE(cfg >= kLbcCfgGlds && cfg < kLbcCfgGlds + kLbcGldsCfgs, "conv_glds: bad cfg %d", cfg);
    const GldsCfg c = kGldsCfg[cfg - kLbcCfgGlds];
    LBC_REQUIRE(a.K % c.bn == 0 && a.C % 64 == 0 && a.KH * a.KW <= 9, "conv_glds: shape not tileable");
    LBC_REQUIRE((long long)a.K * a.KH * a.KW * a.C < (1ll << 31), "conv_glds: weight tensor too large");
    const void* zero = nullptr;
    int rc = lbc_zero_page(&zero);
    if (rc) return rc;
    const bool phased = lbc_glds_phased(a, mode);
    const dim3 grid((unsigned)(lbc_cdiv(a.M, c.bm) * (a.K / c.bn) * (phased ? 4 : 1)));
    LBC_REQUIRE(a.C % 32 == 0, "conv_glds: channel count");
    // K-tile depth: 64 channels (whole cache lines per DMA row, half the barriers), except the 512 x 128 shape on >= 128
    // channels (measured at batch 256: layer 2 0.098 ms with 32-channel tiles, 0.104 with 64; everything else equal or better with 64)
    const bool kt64 = !(cfg == kLbcCfgGlds + 3 && a.C >= 128);
#define LBC_GL2(BMv, BNv, WMv, WNv)                                                                                          \
    do {                                                                                                                     \
        if (kt64) {                                                                                                          \
            if (mode == 0) hipLaunchKernelGGL((conv_glds2_k<BMv, BNv, WMv, WNv, 0, 64>), grid, dim3(WMv * WNv * 64), 0, s, a, zero);    \
            else           hipLaunchKernelGGL((conv_glds2_k<BMv, BNv, WMv, WNv, 1, 64>), grid, dim3(WMv * WNv * 64), 0, s, a, zero);    \
        } else {                                                                                                             \
            if (mode == 0) hipLaunchKernelGGL((conv_glds2_k<BMv, BNv, WMv, WNv, 0, 32>), grid, dim3(WMv * WNv * 64), 0, s, a, zero);    \
            else           hipLaunchKernelGGL((conv_glds2_k<BMv, BNv, WMv, WNv, 1, 32>), grid, dim3(WMv * WNv * 64), 0, s, a, zero);    \
        }                                                                                                                    \
    } while (0)
    if (phased) {
#define LBC_GLP(BMv, BNv, WMv, WNv) hipLaunchKernelGGL((conv_glds2_k<BMv, BNv, WMv, WNv, 1, 64, 1>), grid, dim3(WMv * WNv * 64), 0, s, a, zero)
        if (cfg == kLbcCfgGlds + 0) LBC_GLP(256, 256, 2, 4);
        else if (cfg == kLbcCfgGlds + 1) LBC_GLP(256, 128, 4, 2);
        else if (cfg == kLbcCfgGlds + 2) LBC_GLP(128, 256, 2, 4);
        else if (cfg == kLbcCfgGlds + 3) LBC_GLP(512, 128, 4, 2);
        else if (cfg == kLbcCfgGlds + 4) LBC_GLP(512, 64, 8, 1);
        else if (cfg == kLbcCfgGlds + 5) LBC_GLP(256, 64, 4, 1);
        else LBC_GLP(128, 128, 2, 2);
#undef LBC_GLP
        return lbc_check_launch("conv_glds2");
    }
    if (cfg == kLbcCfgGlds + 0) LBC_GL2(256, 256, 2, 4);
    else if (cfg == kLbcCfgGlds + 1) LBC_GL2(256, 128, 4, 2);
    else if (cfg == kLbcCfgGlds + 2) LBC_GL2(128, 256, 2, 4);
    else if (cfg == kLbcCfgGlds + 3) LBC_GL2(512, 128, 4, 2);
    else if (cfg == kLbcCfgGlds + 4) LBC_GL2(512, 64, 8, 1);
    else if (cfg == kLbcCfgGlds + 5) LBC_GL2(256, 64, 4, 1);
    else LBC_GL2(128, 128, 2, 2);
#undef LBC_GL2
    return lbc_check_launch("conv_glds2");
}
