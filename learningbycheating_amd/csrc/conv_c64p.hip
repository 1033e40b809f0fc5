// 3x3 / stride-1 / pad-1 convolution with C = K = 64 (forward and input gradient) on bf16 tensors: the stem-resolution layer of the
// ResNets -- 16 launches of a training step, 72.5 GFLOP and 252 MB of activations each at batch 256, i.e. as close to the HBM
// roofline (50 us) as to the MFMA one (30 us).  reference: BasicBlock conv1 / conv2, bird_view/models/resnet.py:15-22,38-54.
//
// Persistent successor of conv_halo.hip (register-staged halo, 2-byte stores, 0.124 ms) built from the pieces that worked elsewhere:
//   * weights of all nine taps stationary in registers (144 VGPRs per wave: its 32 output channels), as conv_halo.hip: no weight
//     stream, no barrier inside a tile -- 72 MFMAs per wave between barriers, one LDS fragment read per MFMA;
//   * the ACTIVATION HALO of a 256-pixel tile (256 + 2W + 2 rows of 128 bytes) arrives by LDS-DMA into one of two buffers; a
//     workgroup walks a contiguous range of tiles and requests tile i + 1's halo piece by piece under tile i's MFMAs;
//   * image borders: a lane whose tap leaves the image reads the zero row of the buffer instead (conv_hdma.hip), one select per
//     (tap, 32-row block) -- no per-fragment masking;
//   * wave-private epilogue (conv_hdmap.hpp): 16 rows x 32 columns staged in the wave's own 1.25 KB of LDS, read back as 16-byte
//     chunks and stored; the stores stay in flight under the next tile (the next tile's halo was requested BEFORE them, so the
//     counted vmcnt at the tile boundary leaves them alone); the residual / pre-BatchNorm activation of the fused forms reach the
//     accumulator layout through LDS: each wave DMAs its own 64 x 32 sub-tile (four 1-KiB pieces under the MFMA loop) and reads
//     it back with 2-byte LDS reads (2-byte GLOBAL gathers, 32 per lane and tile, tripled the launch time);
//   * statistics rows through a [2 parities][4][2][64] LDS array, combined after the next tile's opening barrier.
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include "conv_lds_dma.hpp"

namespace {

// EPI: 0 = plain (affine / bias / ReLU / statistics), 1 = + residual, 2 = fused BatchNorm-backward reduce (IgemmArgs::bnb_*)
// PRE (forward only): BatchNorm + ReLU of the producer applied to the input (IgemmArgs::pre_*) -- a DMA cannot transform what it stages, so
// the landed halo is transformed IN PLACE once per tile (450 rows x 128 bytes: seven 16-byte chunks per thread, one extra barrier) before the
// nine taps read it; rows that came from the zero page stay zero.  conv2 of the 64-channel layer reads y1 this way instead of going to
// conv_halo.hip's register-staged kernel (147 us per launch at 256 images).
// BMv (round 4): 256 = the shape above (8 waves, halo double-buffered across tiles, one workgroup per CU); 128 = four waves on a 128-pixel
// tile with ONE halo buffer (328 rows, 41 KB) and TWO workgroups per CU.  In-kernel stamps of the 256 shape (scripts/c64p_prof.py,
// profiles/r04_run5_c64p_prof.txt): a wave spends 47 % of a tile in its K loop, 21 % in the epilogue, 29 % in the opening wait + barrier
// -- the eight waves of the workgroup walk the phases in lock-step (one barrier per tile), so the matrix pipe idles through every epilogue
// and every barrier skew: MFMA busy 0.3.  With two independent four-wave workgroups per CU a SIMD holds one wave of each, and one runs its
// K loop while the other stores, waits for its halo or sits in a barrier -- latency hidden by occupancy instead of by a software pipeline
// that a workgroup-wide barrier per tile keeps breaking.  The single buffer costs a second barrier per tile (everybody has left the K loop
// before the next halo is requested); its landing is covered by the epilogue and by the other workgroup.
// ... and then the halo of the 128 shape became a RING (this version): consecutive tiles of a workgroup overlap in 2W + 2 of their
// BM + 2W + 2 halo rows (194 of 322 at W = 96), and the single buffer re-requested them from L2 for every tile -- 2.5x the input in
// L2 -> LDS traffic, part of it from HBM in the launches that also stream a side tensor (PMC: 445 MB fetched per fused-BatchNorm
// input-gradient launch against 285 MB for the 256 shape).  The halo now lives in a ring of R = 456 rows addressed by the running row
// number g = 128 * tile + halo row (ring row = g mod R): a tile requests only its BM NEW rows, during the PREVIOUS tile's K loop (they
// land in ring rows whose last reader was the tile before that: R >= BM + halo rows, no second barrier), lanes of a piece whose row is
// not new are masked off, and the on-load BatchNorm transform (PRE) touches each row once instead of 2.5 times.
template <int MODE, int EPI, bool PRE = false, int BMv = 256>
__global__ __launch_bounds__(BMv * 2, 2) void conv_c64p_k(IgemmArgs a, const void* zero_page, const int ntiles, const int tpw)
{
    constexpr int BM = BMv, BN = 64, WM = BMv / 64, WN = 2, MT = 2;
    constexpr bool RING = BMv == 128;
    constexpr int NWAVES = WM * WN, NBUF = RING ? 1 : 2;
    constexpr int R = 456;                                      // ring rows (RING): >= BM + halo rows = 128 + 322, a multiple of 8
    constexpr int HRMAX = RING ? R + 1 : 456;                   // rows per buffer; RING: the ring + the zero row.  BM + 2 W + 2 < 456 - 128 resp. 456 <=> W <= 98
    constexpr int NP = (RING ? 328 : HRMAX) / 8;                // 1-KiB halo pieces of a WHOLE halo (57 / 41: RING requests them for its first tile only)
    constexpr int PPW = (NP + NWAVES - 1) / NWAVES;             // ... per wave (8 / 11; the last wave has fewer)
    constexpr int NPN = BM / 8 + 1, PPN = (NPN + NWAVES - 1) / NWAVES;   // RING: pieces that hold a tile's BM new rows (17: the range is not piece-aligned), per wave (5)
    constexpr int ABYTES = HRMAX * 128;
    constexpr int ZROW = RING ? R * 128 : (HRMAX - 1) * 128;
    constexpr int SROWS = (RING && EPI != 0) ? 8 : 16, SROW_B = 32 * 2 + 16;   // staged rows per copy-out step (8 where the side tile leaves no room for 16), their LDS pitch (32 bf16 + 16 bytes)
    constexpr int STG = NBUF * ABYTES;                          // wave-private staging: NWAVES x SROWS x SROW_B
    constexpr int RED = STG + NWAVES * SROWS * SROW_B;          // [2][WM][2][BN] floats
    constexpr int GT = RED + 2 * WM * 2 * BN * 4;               // EPI 1 / 2: per wave [64 rows][32 columns] of the residual / pre-BatchNorm activation
    constexpr int SMEM = GT + (EPI != 0 ? NWAVES * 64 * 64 : 0);
    constexpr int NSTEP = 64 / SROWS;                           // copy-out steps per wave and tile = 16-byte stores per lane
    static_assert(SMEM <= (BMv == 256 ? 160 : 80) * 1024 && PPW <= 18 && PPN <= 18 && (BMv == 256 || BMv == 128) && R % 8 == 0 && R >= 128 + 322,
                  "conv_c64p: LDS / piece arithmetic");
    __shared__ __attribute__((aligned(16))) char smem[SMEM];    // the ONLY LDS object

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;
    const int W = a.W, H = a.H;
    const __bf16* xin = static_cast<const __bf16*>(a.x);
    const __bf16* win = static_cast<const __bf16*>(a.w);
    const __bf16* zero = static_cast<const __bf16*>(zero_page) + (lane & 7) * 8;

    // this workgroup's tiles: a contiguous range (neighbouring tiles share 2W + 2 halo rows: the second read comes from L2)
    int first, cnt;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, q = nwg >> 3, rr = nwg & 7;
        const int p = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
        first = p * tpw;
        cnt = ntiles - first < tpw ? ntiles - first : tpw;
    }
    if (cnt <= 0) return;
    float wg_u1 = 0.f, wg_u2 = 0.f;         // threads < BN: this workgroup's statistics row, accumulated over its tiles

    // stationary weights: output channel 32 wn + l31, k-slots = channels 16 g + 8 kh .. + 7 of tap t
    bf16x8 wreg[9][4];
    {
        const __bf16* wrow = win + (size_t)(32 * wn + l31) * (9 * 64) + 8 * kh;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) wreg[t][g] = *reinterpret_cast<const bf16x8*>(wrow + t * 64 + g * 16);
    }
    // DMA roles: piece wave * PPW + j, lane -> (row, segment); halo row hr <-> input pixel m0 - (W + 1) + hr.  Rows outside the tensor
    // or past the halo come from the zero page (the last buffer row is the ZERO ROW of the border select)
    const int prow = lane >> 3, pseg = lane & 7;
    const int prel0 = wave * PPW * 8 + prow - (W + 1);
    const int pswz_even = (pseg ^ (prow >> 1)) * 8, pswz_odd = (pseg ^ (4 + (prow >> 1))) * 8;     // swizzle of an even / odd PIECE (8 rows each)
    auto issue_piece = [&](const int m0x, const int buf, const int j) {
        const int piece = wave * PPW + j;
        if (piece < NP) {                                       // wave-uniform
            const int rel = prel0 + 8 * j;
            const int q = m0x + rel;
            const bool ok = q >= 0 && q < a.M && rel + (W + 1) < BM + 2 * W + 2;
            const __bf16* src = ok ? xin + ((size_t)q * 64 + (size_t)((piece & 1) ? pswz_odd : pswz_even)) : zero;
            lds_dma16(src, smem + buf * ABYTES + piece * 1024);
        }
    };
    // RING: piece pg = rows g = 8 pg .. 8 pg + 7 of the workgroup's running row numbering (row g <-> input pixel pix0 + g), into ring
    // slot pg mod (R / 8); only lanes whose row lies in [g_lo, g_hi) -- the rows that are NEW -- take part (LDS-DMA honours EXEC)
    const int pix0 = first * BM - (W + 1);
    auto issue_ring = [&](const int pg, const int g_lo, const int g_hi) {
        const int g = 8 * pg + prow;
        if (g >= g_lo && g < g_hi) {
            const int q = pix0 + g;
            const int rr = g % R;
            const bool ok = q >= 0 && q < a.M;
            const __bf16* src = ok ? xin + ((size_t)q * 64 + (size_t)((pseg ^ ((rr >> 1) & 7)) * 8)) : zero;
            lds_dma16(src, smem + (pg % (R / 8)) * 1024);
        }
    };
    const int npw = wave * PPW + PPW <= NP ? PPW : (NP - wave * PPW > 0 ? NP - wave * PPW : 0);   // pieces this wave requests per tile
    int rowc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) rowc[i] = W + 1 + wm * 64 + i * 32 + l31;

    char* stg = smem + STG + wave * (SROWS * SROW_B);
    float* red = reinterpret_cast<float*>(smem + RED);
    __bf16* yout = static_cast<__bf16*>(a.y);
    const int col = 32 * wn + l31;                              // this lane's output channel (accumulator layout)
    const int crow = lane >> 2, cseg = lane & 3;                // copy-out role: 16-byte chunk (row crow, segment cseg) of a 16 x 32 step
    float psc = 1.f, psh = 0.f, bia = 0.f;
    if (a.post_scale) { psc = a.post_scale[col]; psh = a.post_shift[col]; }
    if (a.bias) bia = a.bias[col];
    const __bf16* gsrc = EPI == 1 ? static_cast<const __bf16*>(a.resid) : (EPI == 2 ? static_cast<const __bf16*>(a.bnb_y) : nullptr);

    // the first tile's halo
    if constexpr (RING) {
        if (tid < 8) *reinterpret_cast<f32x4*>(smem + ZROW + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};     // the zero row of the border select (visible behind the first barrier)
#pragma unroll
        for (int j = 0; j < PPW; ++j)
            if (wave * PPW + j < NP) issue_ring(wave * PPW + j, 0, BM + 2 * W + 2);
    } else {
#pragma unroll
        for (int j = 0; j < PPW; ++j) issue_piece(first * BM, 0, j);
    }

    bool stores_pending = false;
    for (int it = 0; it < cnt; ++it) {
        const int tile = first + it;
        const int m0 = tile * BM;
        const int buf = NBUF == 2 ? (it & 1) : 0;
        const bool more = it + 1 < cnt;
        // tap validity of this lane's rows in this tile
        int amask[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * 64 + i * 32 + l31;
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
        // own halo pieces of this tile landed (they are OLDER in this wave's queue than the previous tile's stores, which stay in
        // flight); then everybody's are visible -- and nobody reads the other buffer any more
        if (stores_pending) LBC_WAIT_VM(NSTEP); else LBC_WAIT_VM(0);
        __builtin_amdgcn_s_barrier();
        if constexpr (PRE) {
            // thread -> channel group tid & 7 (its scale / shift: loaded here, through an address the compiler cannot hoist out of the tile
            // loop -- 16 more registers across the K loop would spill), halo rows tid >> 3, + 64, ...; LDS slot of (row, group) = group ^ swizzle(row)
            int cg = tid & 7;
#ifndef LBC_HIP_EMULATED_FOR_TESTS
            asm volatile("" : "+v"(cg));
#endif
            const f32x8 ps8 = ParamVec<8>::ld(a.pre_scale + cg * 8), pt8 = ParamVec<8>::ld(a.pre_shift + cg * 8);
            const float floor8 = a.pre_relu ? 0.f : -INFINITY;
            const int HR = BM + 2 * W + 2;
            // (RING: only the rows this tile brought in -- the others were transformed when they arrived)
            const int hr0 = (RING && it > 0) ? 2 * W + 2 : 0;
            for (int hr = hr0 + (tid >> 3); hr < HR; hr += NWAVES * 8) {
                const int q = m0 - (W + 1) + hr;
                if (q >= 0 && q < a.M) {
                    const int rr = RING ? (it * BM + hr) % R : hr;
                    bf16x8* p = reinterpret_cast<bf16x8*>(smem + buf * ABYTES + rr * 128 + ((cg ^ ((rr >> 1) & 7)) << 4));
                    f32x8 v = __builtin_convertvector(*p, f32x8) * ps8 + pt8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], floor8);
                    *p = __builtin_convertvector(v, bf16x8);
                }
            }
            LBC_WAIT_LGKM0();
            __builtin_amdgcn_s_barrier();
        }
        if (a.stats && it > 0 && tid < BN) {                    // the previous tile's statistics: into this workgroup's running row
            const float* rp = red + ((it - 1) & 1) * (WM * 2 * BN);
            float u1 = 0.f, u2 = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WM; ++w2) { u1 += rp[(w2 * 2 + 0) * BN + tid]; u2 += rp[(w2 * 2 + 1) * BN + tid]; }
            wg_u1 += u1; wg_u2 += u2;
        }

        f32x16 acc[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        const int abuf = buf * ABYTES;
        int aaddr[MT];
        // (the row base and XOR term of a tap depend on the lane's row only: left alone, the compiler keeps all 9 x MT of them in registers
        //  across the tile loop -- next to 144 weight registers that is what spilled into the K loop of the fused-epilogue forms)
#ifndef LBC_HIP_EMULATED_FOR_TESTS
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(rowc[i]));
#endif
        const int rbase = RING ? (it * BM) % R : 0;          // ring row of this tile's halo row 0
        auto tap_addr = [&](const int tap) {
            const int r = tap / 3, s = tap - 3 * r;
            const int off = MODE == 0 ? (r - 1) * W + (s - 1) : (1 - r) * W + (1 - s);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                int hr = rowc[i] + off;
                if constexpr (RING) { hr += rbase; hr = hr >= R ? hr - R : hr; }
                const int val = (hr << 7) | ((kh ^ ((hr >> 1) & 7)) << 4), zval = ZROW | (kh << 4);
                const int m = -((amask[i] >> tap) & 1);
                aaddr[i] = abuf + (((val ^ zval) & m) ^ zval);
            }
        };
        bf16x8 fa[2][MT];
        tap_addr(0);
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(smem + aaddr[i]);
#pragma unroll
        for (int st = 0; st < 36; ++st) {                       // (tap, depth step) = (st / 4, st % 4)
            const int t = st >> 2, g = st & 3;
            if (st + 1 < 36) {
                if (g == 3) tap_addr(t + 1);
                const int g1 = (g + 1) & 3;
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[(st + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(smem + (aaddr[i] ^ (32 * g1)));
            }
            // the next tile's halo, one piece every fourth step (its last readers passed this tile's opening barrier)
            // the next tile's halo, one piece every second step (all requested by step 15 of 36: in-kernel stamps, scripts/c64p_prof.py,
            // show a wave 28 % of its time in the tile's opening wait + barrier -- with the pieces requested as late as step 29 just the
            // same (profiles/r04_run6_*): that time is barrier skew between the SIMD's older and younger wave, not halo latency)
            if constexpr (RING) {
                // the BM new rows of the next tile: g in [it BM + HR, it BM + HR + BM), the pieces that hold them
                if (more && (st & 1) == 1 && (st >> 1) < PPN) {
                    const int g_lo = it * BM + BM + 2 * W + 2;
                    const int pg = (g_lo >> 3) + wave * PPN + (st >> 1);
                    if (pg <= ((g_lo + BM - 1) >> 3)) issue_ring(pg, g_lo, g_lo + BM);
                }
            } else {
                if (more && (st & 1) == 1 && (st >> 1) < PPW) issue_piece(m0 + BM, buf ^ 1, st >> 1);
            }
            if (EPI != 0 && (st & 7) == 3 && st < 32) {
                // this wave's 64 x 32 sub-tile of the residual / pre-BatchNorm activation, 16 rows (one DMA piece: 4 lanes per 64-byte
                // row) at a time -> wave-private: its own vmcnt in front of the epilogue is all the synchronisation it needs
                const int j = st >> 3;
                int m = m0 + wm * 64 + j * 16 + (lane >> 2);
                m = m < a.M ? m : a.M - 1;
                lds_dma16(gsrc + ((unsigned)m * 64u + (unsigned)(32 * wn + (lane & 3) * 8)), smem + GT + wave * 4096 + j * 1024);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st & 1][i], wreg[t][g], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                  // keeps the address arithmetic of later taps out of this step (registers)
        }

        // ---- wave-private epilogue
        if (EPI != 0) {     // own pieces of the side tile landed
            // (requested at steps 3 .. 27: the youngest requests of the tile, the halo pieces are all older)
            LBC_WAIT_VM(0);
        }
        const char* gt = smem + GT + wave * 4096 + l31 * 2;
        float s1 = 0.f, s2 = 0.f;
        // EPI 2 (fused BatchNorm-backward reduce): mask and sums in the CHUNK phase -- the staged gradient and the pre-BatchNorm activation
        // as 16-byte LDS reads, eight channels per lane in vector arithmetic -- instead of per accumulator element (32 two-byte LDS reads
        // and ~15 VALU each: 163 us per launch against 91 plain).  The per-channel operands are loaded HERE, through an address the
        // compiler cannot hoist: live across the K loop they would spill next to the 144 weight registers.
        f32x8 t1 = ParamVec<8>::splat(0.f), t2 = t1, bsc8 = t1, bsh8 = t1;
        int c0 = 32 * wn + cseg * 8;
        if (EPI == 2) {
#ifndef LBC_HIP_EMULATED_FOR_TESTS
            asm volatile("" : "+v"(c0));
#endif
            bsc8 = ParamVec<8>::ld(a.bnb_scale + c0); bsh8 = ParamVec<8>::ld(a.bnb_shift + c0);
        }
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            constexpr int SPB = 32 / SROWS, RPS = SROWS / 2;                         // copy-out steps per 32-row block, accumulator registers per step
            const int mi = s / SPB;
#pragma unroll
            for (int r8 = 0; r8 < RPS; ++r8) {
                const int e = (s % SPB) * RPS + r8;
                const int lr = (e & 3) + 4 * kh + 8 * ((e >> 2) % (SROWS / 8));      // row inside the step
                const bool live = m0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh < a.M;
                float v = acc[mi][e];
                if (a.post_scale) v = v * psc + psh;
                if (a.bias) v += bia;
                const int grow = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;        // row of this element inside the wave's 64 x 32 sub-tile
                if (EPI == 1) v += (float)*reinterpret_cast<const __bf16*>(gt + grow * 64);
                if (a.relu) v = fmaxf(v, 0.f);
                if (EPI == 2) {
                    *reinterpret_cast<__bf16*>(stg + lr * SROW_B + l31 * 2) = (__bf16)v;
                } else {
                    *reinterpret_cast<__bf16*>(stg + lr * SROW_B + l31 * 2) = (__bf16)v;
                    if (live) { s1 += v; s2 += v * v; }
                }
            }
            __builtin_amdgcn_wave_barrier();                    // (one wave's LDS operations execute in order; this pins the compiler -- and the emulator's fibers)
            // (8-row steps: the upper half of the lanes has no chunk; their row index is kept inside the staging rows)
            const bool cact = crow < SROWS;
            const int crw = cact ? crow : 0;
            const int m = cact ? m0 + wm * 64 + s * SROWS + crow : a.M;
            bf16x8 ch = *reinterpret_cast<const bf16x8*>(stg + crw * SROW_B + cseg * 16);
            if (EPI == 2) {
                // sums of the STORED (bf16) gradient, as the separate reduce pass sees it; second sum as sum g * y, centred per tile below
                const f32x8 yf = __builtin_convertvector(*reinterpret_cast<const bf16x8*>(smem + GT + wave * 4096 + (s * SROWS + crw) * 64 + cseg * 16), f32x8);
                f32x8 g = __builtin_convertvector(ch, f32x8);
                const f32x8 z = yf * bsc8 + bsh8;
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = z[e] > 0.f ? g[e] : 0.f;
                ch = __builtin_convertvector(g, bf16x8);
                if (m < a.M) { t1 += g; t2 += g * yf; }
            }
            if (m < a.M) *reinterpret_cast<bf16x8*>(yout + ((unsigned)m * 64u + (unsigned)(32 * wn + cseg * 8))) = ch;
            __builtin_amdgcn_wave_barrier();
        }
        stores_pending = true;
        if (a.stats && EPI == 2) {
            // lanes with the same segment (lane & 3) hold partial sums of the same 8 channels: combine over lane >> 2, then centre:
            // sum g * xhat = (sum g * y - mean * sum g) * invstd
#pragma unroll
            for (int off = 4; off < 64; off <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) { t1[e] += __shfl_xor(t1[e], off); t2[e] += __shfl_xor(t2[e], off); }
            float* rp = red + (it & 1) * (WM * 2 * BN);
            if (lane < 4) {
                const f32x8 bmu8 = ParamVec<8>::ld(a.bnb_mean + c0), biv8 = ParamVec<8>::ld(a.bnb_invstd + c0);
                t2 = (t2 - bmu8 * t1) * biv8;
#pragma unroll
                for (int e = 0; e < 8; ++e) { rp[(wm * 2 + 0) * BN + c0 + e] = t1[e]; rp[(wm * 2 + 1) * BN + c0 + e] = t2[e]; }
            }
        } else if (a.stats) {
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            float* rp = red + (it & 1) * (WM * 2 * BN);
            if (kh == 0) { rp[(wm * 2 + 0) * BN + col] = s1; rp[(wm * 2 + 1) * BN + col] = s2; }
        }
    }
    if (a.stats) {                                              // the last tile's statistics row
        LBC_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        if (tid < BN) {
            const float* rp = red + ((cnt - 1) & 1) * (WM * 2 * BN);
            float u1 = 0.f, u2 = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WM; ++w2) { u1 += rp[(w2 * 2 + 0) * BN + tid]; u2 += rp[(w2 * 2 + 1) * BN + tid]; }
            // ONE row per persistent workgroup (its tiles summed in tile order): <= 256 rows per launch whatever the batch -- the
            // finalize needs no pre-reduction pass (3840 per-tile rows at batch 256 did), and its consumer may fold it at small batch
            float* dst = a.stats + (size_t)(a.stat_row0 + first / tpw) * 2 * BN;
            dst[tid] = wg_u1 + u1;
            dst[BN + tid] = wg_u2 + u2;
        }
    }
    (void)npw;
}

}  // namespace

// statistics rows of a launch: one per persistent workgroup
// tile rows of the launch: LBC_C64P_BM = 256 / 128 pins a shape; default: the four-wave 128-pixel ring shape (two workgroups per CU),
// except for the fused BatchNorm-backward form: next to its side tile the ring leaves LDS for 8-row copy-out steps only, and its chunk
// phase -- mask, sums, eight channels per lane -- then runs with half the lanes: 195 us per launch against 148 us on the 256 shape
// (profiles/r04_final_* of the run before this policy)
static int c64p_bm(const IgemmArgs& a)
{
    const long long v = lbc_opt(kOptC64pBm);
    if (v == 256 || v == 128) return (int)v;
    return a.bnb_y ? 256 : 128;
}
static int c64p_cap(int bm) { return lbc_opt(kOptHaloBlocks) > 0 ? (int)lbc_opt(kOptHaloBlocks) : (bm == 256 ? 256 : 512); }   // persistent workgroups (tests: fewer)

int lbc_conv_c64p_rows(const IgemmArgs& a)
{
    const int bm = c64p_bm(a);
    const int ntiles = lbc_cdiv(a.M, bm);
    return lbc_cdiv(ntiles, lbc_cdiv(ntiles, c64p_cap(bm)));
}

// C = K = 64, 3x3 / stride 1 on bf16 tensors with bf16 weight copies (lbc_conv_hdma_pick: cfg kLbcCfgHdma + 3); statistics rows
// are per persistent workgroup (lbc_conv_c64p_rows)
int lbc_conv_c64p_launch(const IgemmArgs& a, int mode, hipStream_t s)
{
    LBC_REQUIRE(a.C == 64 && a.K == 64 && !a.post_scale == !a.post_shift && (mode == 0 || mode == 1), "conv_c64p: shape");
    LBC_REQUIRE(!a.pre_scale || (mode == 0 && a.pre_shift && !a.resid && !a.bnb_y), "conv_c64p: BatchNorm-on-load serves plain forward launches");
    LBC_REQUIRE(2 * a.W + 2 < 200, "conv_c64p: image too wide for the halo buffer");
    LBC_REQUIRE(!a.bnb_y || (mode == 1 && !a.resid), "conv_c64p: the fused BatchNorm-backward reduce serves input gradients without a residual");
    const void* zero = nullptr;
    int rc = lbc_zero_page(&zero);
    if (rc) return rc;
    const int bm = c64p_bm(a);
    const int ntiles = lbc_cdiv(a.M, bm);
    const int tpw = lbc_cdiv(ntiles, c64p_cap(bm));
    const dim3 grid((unsigned)lbc_cdiv(ntiles, tpw));
    const int epi = a.bnb_y ? 2 : (a.resid ? 1 : 0);
#define LBC_C6(MODEv, EPIv)                                                                                                          \
    do {                                                                                                                             \
        if (bm == 256) hipLaunchKernelGGL((conv_c64p_k<MODEv, EPIv, false, 256>), grid, dim3(512), 0, s, a, zero, ntiles, tpw); \
        else           hipLaunchKernelGGL((conv_c64p_k<MODEv, EPIv, false, 128>), grid, dim3(256), 0, s, a, zero, ntiles, tpw); \
    } while (0)
    if (mode == 0 && a.pre_scale) {
        if (bm == 256) hipLaunchKernelGGL((conv_c64p_k<0, 0, true, 256>), grid, dim3(512), 0, s, a, zero, ntiles, tpw);
        else           hipLaunchKernelGGL((conv_c64p_k<0, 0, true, 128>), grid, dim3(256), 0, s, a, zero, ntiles, tpw);
        return lbc_check_launch("conv_c64p");
    }
    if (mode == 0) { if (epi == 1) LBC_C6(0, 1); else LBC_C6(0, 0); }
    else { if (epi == 2) LBC_C6(1, 2); else if (epi == 1) LBC_C6(1, 1); else LBC_C6(1, 0); }
#undef LBC_C6
    return lbc_check_launch("conv_c64p");
}
