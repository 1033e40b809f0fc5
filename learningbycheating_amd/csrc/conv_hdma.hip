// 3x3 / stride-1 / pad-1 convolution (forward and input gradient) on bf16 tensors for gfx950 with the ACTIVATION HALO staged
// once per 64-channel slab: the third member of the LDS-DMA family (conv_glds.hip explains the family).
// reference arithmetic: BasicBlock conv1 / conv2, bird_view/models/resnet.py:15-22,38-54, and their autograd.
//
// Why: conv_glds2_k streams the activation tile of every (tap, slab) K-tile separately -- nine shifted copies of the same
// rows -- and its timing experiments put ~20 us of a 72 us layer-3 launch on the latency of that stream.
// Here a workgroup stages, per 64-channel slab, the BM + 2W + 2 input rows its BM output pixels can touch ONCE (LDS-DMA, two
// buffers: slab c + 1 lands while slab c is multiplied); the nine taps are row offsets of the fragment reads, exactly as in
// conv_halo.hip.  Only the weights still stream per (tap, slab): a ring of NBUFB tiles of BN x 64, as in conv_glds2_k.
//   * LDS rows are 128 bytes, 16-byte slot XOR-ed with (row >> 1) & 7 (on the DMA source and on the read) -> conflict-free
//     ds_read_b128; a tap changes the row, hence the XOR term: 2 VALU per fragment read (one v_xor, one v_lshl_add).
//   * Image borders: a lane whose tap leaves the image reads a 128-byte ZERO ROW instead (the last row of the halo buffer, which
//     lies past the halo and is filled from the zero page; one select on the row base per (tap, 32-row block)) -- no masking
//     of the fragments themselves.
//   * Synchronisation as conv_glds2_k with 64-channel K-tiles: one barrier per K-tile in front of its last depth step; the
//     wave's own weight pieces of the next K-tile are waited for with a counted vmcnt (halo pieces are always OLDER in the
//     wave's DMA queue than the first weight tile of their slab: they are issued in the first 9 - NBUFB taps of the
//     previous slab, that weight tile after them -- so the same wait covers them).
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include "conv_lds_dma.hpp"

namespace {

#define LBC_SG(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)

// PRE = 1 (forward only): BatchNorm(+ReLU) of the producer applied to the input (IgemmArgs::pre_*), as an IN-PLACE transform of the
// staged halo -- once per element and slab instead of once per tap: every thread rewrites its 16-byte slots of slab c + 1 during
// taps 6-7 of slab c (its channel segment is the same for all of its rows, so the 8 + 8 coefficients sit in registers; they are
// fetched one tap earlier -- extra VMEM loads in the wave's queue can only make the counted DMA waits stricter, never laxer).
// Rows from the zero page are rewritten too; they are met only by taps that the border select sends to the zero row, which is excluded.
// EARLY = 1 (LBC_HDMA_EARLY=1; not yet measured): the fragment reads of depth step g + 1 are the FIRST instructions of step g's
// scheduling region instead of trailing its MFMAs -- their byte offsets are formed one region ahead (ra / rb), so nothing a read
// needs is computed in its own region.  Static picture of the default code (scripts/isa_mix.py): every step is
// `s_waitcnt lgkmcnt(0), MFMA x3, read x2, MFMA, read x2`, i.e. the reads a step waits for were issued 0-1 MFMAs (<= 32 cycles)
// earlier against ~64+ cycles of LDS latency, in all 8 waves at once (they leave each tap's barrier together).
template <int BM, int BN, int WM, int WN, int HRMAX, int NBUFB, int MODE, int PRE = 0, int EARLY = 0, int DIAG = 0>
__global__ __launch_bounds__(512, 2) void conv_hdma_k(IgemmArgs a, const void* zero_page)
{
    constexpr int WTM = BM / WM, WTN = BN / WN;                 // per-wave output tile
    constexpr int MT = WTM / 32, NT = WTN / 32;
    static_assert(WM * WN == 8 && NT == 2 && (MT == 2 || MT == 4), "conv_hdma: wave tiling");
    static_assert(HRMAX % 64 == 0 && BN % 64 == 0 && (NBUFB == 2 || NBUFB == 4), "conv_hdma: staging");
    constexpr int KS = 4;                                       // depth steps of 16 channels per K-tile
    constexpr int ABYTES = HRMAX * 128;                         // one halo buffer: HRMAX rows x 64 channels
    constexpr int TILE_B = BN * 128;
    constexpr int BRING = 2 * ABYTES;                           // weight ring behind the two halo buffers
    constexpr int MAIN = BRING + NBUFB * TILE_B;
    constexpr int ZROW = (HRMAX - 1) * 128;                     // last row of either halo buffer: always beyond the halo, filled from the zero page
    constexpr int EPI = lds_dma_epilogue_bytes<BM, BN, WM>();
    constexpr int SMEM = MAIN > EPI ? MAIN : EPI;
    static_assert(SMEM <= 160 * 1024, "conv_hdma: LDS");
    __shared__ __attribute__((aligned(16))) char smem[SMEM];    // the ONLY LDS object
    constexpr int HPW = HRMAX / 64;                             // 1-KiB halo pieces (8 rows) per wave per slab
    constexpr int NBW = BN / 64;                                // 1-KiB weight pieces per wave per K-tile
    constexpr int ATAPS = 9 - NBUFB;                            // taps of a slab whose issue slot may carry halo pieces (see above)
    constexpr int AP = (HPW + ATAPS - 1) / ATAPS;               // halo pieces per such tap
    static_assert(!PRE || MODE == 0, "conv_hdma: the input transform belongs to the forward");
    static_assert(!PRE || (HPW + AP - 1) / AP <= 8 - NBUFB, "conv_hdma: the next halo must have landed by the barrier of tap 6");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;
    const int W = a.W, H = a.H, C = a.C;

    const int ntn = a.K / BN;
    int tile_id;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, rr = nwg & 7;
        tile_id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
    }
    const int mtile = tile_id / ntn;
    const int m0 = mtile * BM;
    const int n0 = (tile_id - mtile * ntn) * BN;

    const __bf16* xin = static_cast<const __bf16*>(a.x);
    const __bf16* win = static_cast<const __bf16*>(a.w);
    const __bf16* zero = static_cast<const __bf16*>(zero_page) + (lane & 7) * 8;

    // ---- DMA roles.  Halo row hr holds input pixel m0 - (W + 1) + hr; rows outside the tensor come from the zero page (they are
    //      only ever met by taps that the border select below redirects, but they must not be read from unmapped memory), and so
    //      do the buffer rows past the halo (BM + 2W + 2 < HRMAX): the last of them is the ZERO ROW of the border select
    const int prow = lane >> 3, pseg = lane & 7;
    int aoff[HPW];
    bool aval[HPW];
#pragma unroll
    for (int j = 0; j < HPW; ++j) {
        const int row = (wave * HPW + j) * 8 + prow;
        const int q = m0 - (W + 1) + row;
        aval[j] = q >= 0 && q < a.M && row < BM + 2 * W + 2;
        aoff[j] = (aval[j] ? q : 0) * C + (pseg ^ ((row >> 1) & 7)) * 8;
    }
    int boff[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int row = (wave * NBW + j) * 8 + prow;
        boff[j] = (n0 + row) * (9 * C) + (pseg ^ ((row >> 1) & 7)) * 8;
    }
    // ---- fragment roles.  Weights: row l31 of a 32-row block, slot (2g + kh) ^ ((l31 >> 1) & 7).  Activations: halo row of the
    //      centre tap per 32-row block + per-lane tap validity
    const int bxor = kh ^ ((l31 >> 1) & 7);                    // slot (2g + kh) ^ f(row) = 2g ^ (kh ^ f(row))
    const int bBase = BRING + (wn * WTN + l31) * 128;
    int rowc[MT], amask[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = wm * WTM + i * 32 + l31;
        rowc[i] = W + 1 + row;
        const int m = m0 + row;
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
        amask[i] = bits;
    }

    const int nslab = C / 64;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // halo pieces [p0, p1) of slab `slab` into halo buffer slab & 1
    auto issue_a = [&](const int slab, const int p0, const int p1) {
        char* base = smem + (slab & 1) * ABYTES;
#pragma unroll
        for (int j = 0; j < HPW; ++j) {
            if (j < p0 || j >= p1) continue;
            const __bf16* src = aval[j] ? xin + (aoff[j] + slab * 64) : zero;
            lds_dma16(src, base + (wave * HPW + j) * 1024);
        }
    };
    // weight tile of K-tile k = (slab, tap) into ring slot k % NBUFB
    auto issue_b = [&](const int slab, const int tap, const int slot) {
        char* base = smem + BRING + slot * TILE_B;
        const int koffs = tap * C + slab * 64;
#pragma unroll
        for (int j = 0; j < NBW; ++j) lds_dma16(win + (boff[j] + koffs), base + (wave * NBW + j) * 1024);
    };

    // per (tap, 32-row block): byte offset of the lane's halo row in its buffer (or of the zero row) and the XOR term of its slot
    int abase[MT], axor[MT];                // of the K-tile whose fragments are being read; recomputed for the next one after its last read
    auto tap_addr = [&](const int tap, const int slab, int (&base)[MT], int (&xr)[MT]) {
        const int r = tap / 3, s = tap - 3 * r;
        const int off = MODE == 0 ? (r - 1) * W + (s - 1) : (1 - r) * W + (1 - s);
        const int abuf = (slab & 1) * ABYTES;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int hr = rowc[i] + off;
            const bool ok = (amask[i] >> tap) & 1;
            base[i] = abuf + (ok ? (hr << 7) : ZROW);
            xr[i] = ok ? (kh ^ ((hr >> 1) & 7)) : kh;            // slot (2g + kh) ^ f(hr) = 2g ^ (kh ^ f(hr))
        }
    };

    bf16x8 fa[2][MT], fb[2][NT];            // two register sets: depth step g computes from set g & 1 while set (g + 1) & 1 is read
#define LBC_RD(SLOT, G, SET)                                                                                                     \
    do {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                           \
            fa[SET][i] = *reinterpret_cast<const bf16x8*>(smem + abase[i] + (((2 * (G)) ^ axor[i]) << 4));           \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                           \
            fb[SET][j] = *reinterpret_cast<const bf16x8*>(smem + (SLOT) * TILE_B + bBase + j * 32 * 128 + (((2 * (G)) ^ bxor) << 4));             \
    } while (0)
#ifdef LBC_HIP_EMULATED_FOR_TESTS
#define LBC_KEEP(SET) do { } while (0)
#else
#define LBC_KEEP(SET)                                                                                                            \
    do {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) asm volatile("" :: "v"(fa[SET][i]));                                     \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) asm volatile("" :: "v"(fb[SET][j]));                                     \
    } while (0)
#endif
#define LBC_MM(SET)                                                                                                              \
    do {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                           \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                       \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);                 \
    } while (0)
    int ra[MT], rb[NT];                     // EARLY: byte offsets of the NEXT fragment reads (formed one scheduling region ahead)
#define LBC_AD(SLOT, G)                                                                                                          \
    do {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) ra[i] = abase[i] + (((2 * (G)) ^ axor[i]) << 4);                          \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) rb[j] = (SLOT) * TILE_B + bBase + j * 32 * 128 + (((2 * (G)) ^ bxor) << 4); \
    } while (0)
// EARLY reads are inline asm with hand-counted lgkmcnt waits: with LDS-DMA in flight the compiler's own wait insertion falls back
// to lgkmcnt(0) in front of every depth step (it no longer trusts the return order of the LGKM queue), i.e. it would also wait
// for the reads just issued for the NEXT step.  The compiler does not see these loads as pending, so each consumer set goes
// through LBC_USE (an empty asm the MFMAs depend on) placed after the wait that covers it.
#ifdef LBC_HIP_EMULATED_FOR_TESTS
#define LBC_RDA(SET)                                                                                                             \
    do {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) fa[SET][i] = *reinterpret_cast<const bf16x8*>(smem + ra[i]);              \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) fb[SET][j] = *reinterpret_cast<const bf16x8*>(smem + rb[j]);              \
    } while (0)
#define LBC_USE(SET) do { } while (0)
#else
#define LBC_RDA(SET)                                                                                                             \
    do {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[SET][i]) : "v"(lds0 + ra[i])); \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(fb[SET][j]) : "v"(lds0 + rb[j])); \
    } while (0)
#define LBC_USE(SET)                                                                                                             \
    do {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[SET][i]));                                      \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(fb[SET][j]));                                      \
    } while (0)
#endif
    // all but the MT + NT youngest LDS reads of this wave have returned (in EARLY code LDS reads are the only LGKM traffic of the loop)
#define LBC_WAIT_OLDER_READS() __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, MT + NT))
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;      // LDS byte offset of smem (its only object: 0)

    // ---- PRE: in-place BatchNorm(+ReLU) of halo buffer slab & 1 (rows [0, BM + 2W + 2): the zero row and the padding rows stay)
    const int tseg = (tid & 7) ^ (((tid >> 3) >> 1) & 7);       // this thread's channel segment: rows tid / 8 + 64 i share (row >> 1) & 7
    f32x8 psc = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float relu_floor = (PRE && a.pre_relu) ? 0.f : -INFINITY;
    auto pre_coef = [&](const int slab) {
        psc = ParamVec<8>::ld(a.pre_scale + slab * 64 + tseg * 8);
        psh = ParamVec<8>::ld(a.pre_shift + slab * 64 + tseg * 8);
    };
    auto pre_apply = [&](const int slab) {
        char* base = smem + (slab & 1) * ABYTES + (tid & 7) * 16;
        const int hr_end = BM + 2 * W + 2;
        for (int row = tid >> 3; row < hr_end; row += 64) {
            bf16x8* p = reinterpret_cast<bf16x8*>(base + row * 128);
            f32x8 v = __builtin_convertvector(*p, f32x8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] * psc[e] + psh[e], relu_floor);        // as conv_igemm.hip's on-load path
            *p = __builtin_convertvector(v, bf16x8);
        }
    };

    // ---- prologue: the halo of slab 0 and up to NBUFB weight tiles in flight; everything of K-tile 0 landed and visible
    if (PRE) pre_coef(0);
    issue_a(0, 0, HPW);
    const int nk = 9 * nslab;
#pragma unroll
    for (int k = 0; k < NBUFB; ++k)
        if (k < nk) issue_b(0, k, k);        // (nk >= 9 > NBUFB: the first NBUFB K-tiles are taps of slab 0)
    LBC_WAIT_VM((NBUFB - 1) * NBW);
    __builtin_amdgcn_s_barrier();
    if (PRE) {
        pre_apply(0);
        LBC_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
    }
    tap_addr(0, 0, abase, axor);
    if constexpr (EARLY) { LBC_AD(0, 0); LBC_RDA(0); LBC_AD(0, 1); }
    else LBC_RD(0, 0, 0);
    (void)lds0;

    // One slab = nine K-tiles, taps unrolled.  LAST: no slab c + 1 to prefetch, and the weight ring drains.
    auto slab_body = [&](const int c, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        // The (row base, XOR term) of a tap do not depend on the slab: left alone, the compiler hoists all 9 x MT pairs out of
        // the slab loop (72 registers at MT = 4 -> 800 bytes of scratch per lane).  Make the inputs opaque per slab instead:
        // recomputing them is 5 VALU per (tap, 32-row block) next to 4 MT MFMAs.
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(rowc[i]), "+v"(amask[i]));
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int k = 9 * c + t;
            const int slot = k & (NBUFB - 1), nslot = (k + 1) & (NBUFB - 1);
            const bool has_next = !LAST || t < 8;
#pragma unroll
            for (int g = 0; g + 1 < KS; ++g) {
                if constexpr (EARLY) {
                    LBC_RDA((g + 1) & 1);                            // offsets formed in the previous region
                    if (g < KS - 2) LBC_AD(slot, g + 2);             // ... and here those of the next region's reads
                    else if (has_next) {                             // the next K-tile's (tap, slab) and its first step
                        tap_addr(t < 8 ? t + 1 : 0, t < 8 ? c : c + 1, abase, axor);
                        LBC_AD(nslot, 0);
                    }
                    LBC_WAIT_OLDER_READS();                          // set g & 1 is in: issued a full step (or the tap boundary) ago
                    LBC_USE(g & 1);
                    LBC_MM(g & 1);
                    LBC_SG(0x100, MT + NT);
#pragma unroll
                    for (int q = 0; q < MT * NT; ++q) {
                        LBC_SG(0x008, 1);
                        if (g < KS - 2) LBC_SG(0x002, 3);
                        else LBC_SG(0x002, 6);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
                if (!(DIAG & 2)) LBC_RD(slot, g + 1, (g + 1) & 1);
                // the reads of the last depth step are out: the addresses are free for the next K-tile's (tap, slab)
                if (g == KS - 2 && has_next) tap_addr(t < 8 ? t + 1 : 0, t < 8 ? c : c + 1, abase, axor);
                if (!(DIAG & 16)) LBC_MM(g & 1); else LBC_KEEP(g & 1);
#pragma unroll
                for (int q = 0; q < MT + NT; ++q) { LBC_SG(0x008, 1); LBC_SG(0x100, 1); LBC_SG(0x002, 3); }
                if (MT * NT > MT + NT) LBC_SG(0x008, MT * NT - (MT + NT));
                __builtin_amdgcn_sched_barrier(0);
            }
            // own weight pieces of K-tile k + 1 landed; up to NBUFB - 2 younger tiles may stay in flight (fewer while the ring drains)
            {
                constexpr int younger_max = NBUFB - 2;
                const int left = LAST ? (7 - t > 0 ? 7 - t : 0) : younger_max;
                const int allow = left < younger_max ? left : younger_max;
                if (allow >= 2) LBC_WAIT_VM(2 * NBW);
                else if (allow == 1) LBC_WAIT_VM(NBW);
                else LBC_WAIT_VM(0);
            }
            LBC_WAIT_LGKM0();
            if (!(DIAG & 4)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (PRE && !LAST && t == 5) pre_coef(c + 1);                   // (before this tap's DMA issue: older in the queue)
            if constexpr (EARLY) {
                if (has_next) { LBC_RDA(0); LBC_AD(nslot, 1); }
                LBC_USE((KS - 1) & 1);                                     // in since the lgkmcnt(0) in front of the barrier
            } else {
                if (has_next && !(DIAG & 2)) LBC_RD(nslot, 0, 0);
            }
            if (!(DIAG & 16)) LBC_MM((KS - 1) & 1); else LBC_KEEP((KS - 1) & 1);
            if (PRE && !LAST && t == 6) pre_apply(c + 1);                  // landed and visible since the barrier above; the barriers of
                                                                           // taps 7 and 8 (after lgkmcnt(0)) publish the rewrite
            if (!(DIAG & 1) && (!LAST || t + NBUFB < 9)) {
                const int kn = t + NBUFB;                                  // K-tile k + NBUFB -> the ring slot of K-tile k
                issue_b(kn < 9 ? c : c + 1, kn < 9 ? kn : kn - 9, slot);
            }
            if (!(DIAG & 1) && !LAST && t < ATAPS && t * AP < HPW) issue_a(c + 1, t * AP, (t + 1) * AP < HPW ? (t + 1) * AP : HPW);
            if (EARLY && has_next) LBC_SG(0x100, MT + NT);
#pragma unroll
            for (int q = 0; q < MT * NT; ++q) {
                LBC_SG(0x008, 1);
                if (!EARLY && q < MT + NT) LBC_SG(0x100, 1);
                LBC_SG(0x036, 8);                                          // VALU | SALU | VMEM: address arithmetic and DMA pieces
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int c = 0; c + 1 < nslab; ++c) slab_body(c, std::false_type{});
    slab_body(nslab - 1, std::true_type{});
#undef LBC_RD
#undef LBC_MM
#undef LBC_KEEP
#undef LBC_AD
#undef LBC_RDA
#undef LBC_USE
#undef LBC_WAIT_OLDER_READS

    // ---- epilogue (conv_lds_dma.hpp)
    if constexpr (DIAG & 8) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 123.456f) static_cast<__bf16*>(a.y)[tid] = (__bf16)t;
    } else
    lds_dma_epilogue<BM, BN, WM, WN, MT, NT>(a, acc, smem, m0, n0, mtile);
}
#undef LBC_SG

struct HdmaCfg { int bm, bn, hrmax; };
// cfg ids kLbcCfgHdma + 0 .. 2
const HdmaCfg kHdmaCfg[kLbcHdmaCfgs] = {{256, 256, 320}, {256, 128, 384}, {128, 256, 192}, {256, 64, 456}, {128, 64, 192}};   // 3: conv_c64p.hip, 4: conv_hdmap.hpp with four waves

}  // namespace

// Tile configuration for a launch, or -1 when the launch keeps conv_glds.hip / conv_igemm.hip.
int lbc_conv_hdma_pick(const IgemmArgs& a, int mode)
{
    if (lbc_opt_on(kOptNoHdma) || lbc_opt_on(kOptNoGemm256) || lbc_opt_on(kOptGldsV1)) return -1;
    if (!(a.w_bf16 && a.act_bf16) || a.ostep != 1 || a.nphase > 1 || a.oy0 || a.ox0) return -1;
    // BatchNorm-on-load as an in-LDS transform of the halo: forward only, behind LBC_HDMA_PROLOGUE=1 until it is measured
    // (the 64-channel layer's persistent kernel has it always: conv_c64p_k<0, 0, true>; LBC_NO_C64P_PRE=1 sends those launches back to conv_halo.hip)
    const bool c64 = a.C == 64 && a.K == 64;
    const bool hdmap_pre = lbc_opt_on(kOptHdmapPre) && !a.resid;      // conv_hdmap_k<.., PRE> where the launch is eligible for it, conv_igemm.hip otherwise
    if (a.pre_scale && (mode != 0 || (c64 ? (lbc_opt_on(kOptNoC64pPre) || a.resid != nullptr) : !(lbc_opt_on(kOptHdmaPrologue) || hdmap_pre)))) return -1;
    if (a.KH != 3 || a.KW != 3 || a.P != 1 || a.S != 1 || a.C % 64 || (mode != 0 && mode != 1)) return -1;
    if (a.H != a.OH || a.W != a.OW || a.M != a.N * a.H * a.W || (long long)a.N * a.H * a.W * a.C >= (1ll << 31)) return -1;
    if ((long long)a.K * 9 * a.C >= (1ll << 31)) return -1;
    // one workgroup per CU; worth it from about 96 tiles (measured at 120 tiles = layer 2 at batch 32: 0.020 ms against 0.027 ms for
    // the 64 x 64 register-staged tiles; at 60 tiles = layer 3 at batch 32 it loses, 0.029 vs 0.027)
    const long long fill = lbc_opt(kOptGemm256MinTiles) > 0 ? lbc_opt(kOptGemm256MinTiles) : 96;
    const long long forced = lbc_opt(kOptHdmaCfg);          // tests / tuning: pin one shape
    if (a.C == 64 && a.K == 64) {                           // the 64-channel layer: conv_c64p.hip (persistent, weights in registers)
        // (its round-2 predecessor conv_hdma64_k lost to conv_halo.hip, 0.224 vs 0.127 ms at batch 256: 62 spilled registers in a
        //  workgroup-wide LDS-staged epilogue; the wave-private epilogue of round 3 needs none)
        if (lbc_opt_on(kOptNoHdma64) || (forced >= 0 && forced != 3)) return -1;
        if (256 + 2 * a.W + 2 >= kHdmaCfg[3].hrmax || lbc_cdiv(a.M, 256) < fill) return -1;
        return kLbcCfgHdma + 3;
    }
    int best = -1;
    long long best_tiles = 0;
    double best_score = 0.0;
    for (int i = 0; i < 3; ++i) {
        const HdmaCfg& c = kHdmaCfg[i];
        if (a.K % c.bn) continue;
        if (forced >= 0 && forced != i) continue;
        if (i == 0 && forced != 0) continue;                // 256 x 256: 128 accumulator registers per wave leave too few for the rest (108 bytes of scratch): tests only
        if (c.bm + 2 * a.W + 2 >= c.hrmax) continue;        // the halo of a tile + one zero row must fit its LDS buffer
        const long long tiles = (long long)lbc_cdiv(a.M, c.bm) * (a.K / c.bn);
        if (tiles < fill) continue;
        const double score = (double)tiles / (double)(((tiles + 255) / 256) * 256) * (c.bm * c.bn >= 256 * 256 ? 1.0 : 0.9);
        if (a.pre_scale && !lbc_opt_on(kOptHdmaPrologue) && !lbc_conv_hdmap_eligible(a, mode, kLbcCfgHdma + i)) continue;      // (LBC_HDMAP_PRE alone: only shapes whose persistent form has the transform)
        if (score > best_score) { best_score = score; best = i; best_tiles = tiles; }
    }
    // An eight-wave launch that leaves half the CUs idle runs faster as four-wave 128 x 64 tiles, two workgroups per CU and four times the
    // workgroups, where that shape's 184-row halo holds the image rows (layers 3 / 4): 120 tiles = layer 3 at 64 images 29 -> 22 us per
    // launch (the step 7.20 -> 6.89 ms on that box), layer 4 at 128 images 50 -> 38 us (10.34 -> 10.14 ms); at 240 tiles (layer 3 at 128
    // images) it loses, 35 -> 41 us (profiles/r04_run16_small_tiles_at_120.log).  Nothing in between was measured: the threshold sits at
    // 160 tiles (62 % of the CUs).  LBC_HDMA_SMALL_BELOW=0: never.
    const long long below = lbc_opt(kOptHdmaSmallBelow) >= 0 ? lbc_opt(kOptHdmaSmallBelow) : 160;
    // (not under LBC_GEMM256_MIN_TILES: the tests' switch that sends small launches to the eight-wave shapes keeps its meaning)
    const bool prefer_small = best >= 0 && forced < 0 && best_tiles < below && a.K % 64 == 0 && lbc_opt(kOptGemm256MinTiles) <= 0 &&
                              lbc_conv_hdmap_eligible(a, mode, kLbcCfgHdma + 4);
    if (best >= 0 && !prefer_small) return kLbcCfgHdma + best;
    // Few rows (the per-GPU load of the 8-GPU run: layer 3 / 4 at 32 images have 7680 / 1920 output pixels): 128 x 64 tiles, four waves,
    // two workgroups per CU (conv_hdmap.hpp) instead of the 64 x 64 register-staged tiles of conv_igemm.hip (31 us per 9-GFLOP launch)
    if ((forced < 0 || forced == 4) && a.K % 64 == 0 && lbc_conv_hdmap_eligible(a, mode, kLbcCfgHdma + 4)) {
        const long long tiles = (long long)lbc_cdiv(a.M, 128) * (a.K / 64);
        // (its own knob; LBC_GEMM256_MIN_TILES -- the per-tap kernel's threshold, which tests set to 1 -- still applies when this one is unset)
        const long long small_fill = lbc_opt(kOptHdmaSmallMinTiles) > 0 ? lbc_opt(kOptHdmaSmallMinTiles)
                                     : (lbc_opt(kOptGemm256MinTiles) > 0 ? lbc_opt(kOptGemm256MinTiles) : 48);
        if (tiles >= small_fill) return kLbcCfgHdma + 4;
    }
    return -1;
}

int lbc_conv_hdma_rows(const IgemmArgs& a, int cfg)
{
    if (cfg == kLbcCfgHdma + 3) return lbc_conv_c64p_rows(a);         // one row per persistent workgroup
    return lbc_cdiv(a.M, kHdmaCfg[cfg - kLbcCfgHdma].bm);
}

int lbc_conv_hdma_launch(const IgemmArgs& a, int mode, int cfg, hipStream_t s)
{
    LBC_REQUIRE(cfg >= kLbcCfgHdma && cfg < kLbcCfgHdma + kLbcHdmaCfgs, "conv_hdma: bad cfg %d", cfg);
    const HdmaCfg c = kHdmaCfg[cfg - kLbcCfgHdma];
    LBC_REQUIRE(a.K % c.bn == 0 && a.C % 64 == 0 && c.bm + 2 * a.W + 2 < c.hrmax, "conv_hdma: shape not tileable");
    if (cfg == kLbcCfgHdma + 4) return lbc_conv_hdmap_launch(a, mode, cfg, s);
    const void* zero = nullptr;
    int rc = lbc_zero_page(&zero);
    if (rc) return rc;
    if (cfg == kLbcCfgHdma + 3) return lbc_conv_c64p_launch(a, mode, s);
    const dim3 grid((unsigned)(lbc_cdiv(a.M, c.bm) * (a.K / c.bn)));
    const bool early = lbc_opt_on(kOptHdmaEarly);
    if (!early && lbc_opt(kOptHdmaDiag) <= 0 && lbc_conv_hdmap_eligible(a, mode, cfg)) return lbc_conv_hdmap_launch(a, mode, cfg, s);
    const long long diag = lbc_opt(kOptHdmaDiag);
    if (diag > 0 && mode == 0 && !a.pre_scale && cfg == kLbcCfgHdma + 1) {     // timing experiments (wrong results)
#define LBC_HDD(D) case D: hipLaunchKernelGGL((conv_hdma_k<256, 128, 4, 2, 384, 4, 0, 0, 0, D>), grid, dim3(512), 0, s, a, zero); return lbc_check_launch("conv_hdma")
        switch (diag) {
            LBC_HDD(1); LBC_HDD(2); LBC_HDD(8); LBC_HDD(16); LBC_HDD(15);
            default: break;
        }
#undef LBC_HDD
    }
    if (diag > 0 && mode == 0 && !a.pre_scale && cfg == kLbcCfgHdma + 0) {
#define LBC_HDD(D) case D: hipLaunchKernelGGL((conv_hdma_k<256, 256, 2, 4, 320, 2, 0, 0, 0, D>), grid, dim3(512), 0, s, a, zero); return lbc_check_launch("conv_hdma")
        switch (diag) {
            LBC_HDD(1); LBC_HDD(2); LBC_HDD(8); LBC_HDD(16); LBC_HDD(15);
            default: break;
        }
#undef LBC_HDD
    }
#define LBC_HD(BMv, BNv, WMv, WNv, HRv, NBv)                                                                                 \
    do {                                                                                                                     \
        if (mode == 0 && a.pre_scale) hipLaunchKernelGGL((conv_hdma_k<BMv, BNv, WMv, WNv, HRv, NBv, 0, 1>), grid, dim3(512), 0, s, a, zero); \
        else if (early && mode == 0) hipLaunchKernelGGL((conv_hdma_k<BMv, BNv, WMv, WNv, HRv, NBv, 0, 0, 1>), grid, dim3(512), 0, s, a, zero); \
        else if (early)     hipLaunchKernelGGL((conv_hdma_k<BMv, BNv, WMv, WNv, HRv, NBv, 1, 0, 1>), grid, dim3(512), 0, s, a, zero); \
        else if (mode == 0) hipLaunchKernelGGL((conv_hdma_k<BMv, BNv, WMv, WNv, HRv, NBv, 0>), grid, dim3(512), 0, s, a, zero);   \
        else           hipLaunchKernelGGL((conv_hdma_k<BMv, BNv, WMv, WNv, HRv, NBv, 1>), grid, dim3(512), 0, s, a, zero);   \
    } while (0)
    if (cfg == kLbcCfgHdma + 0) LBC_HD(256, 256, 2, 4, 320, 2);
    else if (cfg == kLbcCfgHdma + 1) LBC_HD(256, 128, 4, 2, 384, 4);
    else LBC_HD(128, 256, 2, 4, 192, 2);
#undef LBC_HD
    return lbc_check_launch("conv_hdma");
}
