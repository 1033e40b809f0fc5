// Stem tail for gfx950: BatchNorm + ReLU + MaxPool2d(3,2,1) fused in one pass over
// the largest activation of the network (N x 80 x 192 x 64 for the RGB model), and
// the matching backward (max-pool scatter as a gather + ReLU mask + BatchNorm
// gradient reductions) in one pass.  reference: bird_view/models/resnet.py:104-106,
// 149-152.  The forward stores the arg-max tap (0..8, first maximum in row-major
// order, as torch does) per pooled element so the backward never re-scans windows.
#include "lbc_common.hpp"
#include "lbc_act.hpp"
#include "lbc_kernels.hpp"

namespace {

// Workgroup ids go round-robin over the 8 XCDs, each with an L2 of its own.  Both passes below re-read rows across workgroup boundaries
// (a 3 x 3 / 2 window shares an input row with the window below it; a pixel range shares pooled rows with the next range): with
// consecutive ranges on consecutive ids every shared row was fetched by two XCDs (PMC: 629 MB for the 503 MB forward input, 1041 MB for
// the backward's 693, profiles/r05_final_pmc_summary_bf16.txt).  Logical id = XCD-major: consecutive ranges sit on ONE XCD.
__device__ __forceinline__ unsigned xcd_major_block()
{
    const unsigned nwg = gridDim.x, b = blockIdx.x;
    const unsigned xcd = b & 7u, q = nwg >> 3, rr = nwg & 7u;
    return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (b >> 3);
}

template <typename T, typename IDX>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_k(PoolFwdArgs a)
{
    constexpr int V = Act<T>::kVec;          // 16-byte accesses: 4 f32 or 8 bf16 channels per thread
    using vec = typename Act<T>::vec;
    using PV = ParamVec<V>;
    const T* y = static_cast<const T*>(a.y);
    T* pout = static_cast<T*>(a.p);
    const int OH = a.H / 2, OW = a.W / 2;
    const int cvn = a.C / V;
    // IDX = unsigned when every index of the launch fits 32 bits (lbc_bn_relu_maxpool_fwd): four 64-bit divisions per element otherwise
    const IDX total = (IDX)((long long)a.N * OH * OW * cvn);
    const IDX stride = (IDX)gridDim.x * (IDX)blockDim.x;
    // (each workgroup owns a contiguous span of ceil(total / grid) elements, spans in XCD-major order: vertical neighbours share an L2)
    const IDX span = (total + (IDX)gridDim.x - 1) / (IDX)gridDim.x;
    const IDX i0 = (IDX)xcd_major_block() * span;
    const IDX i1 = i0 + span < total ? i0 + span : total;
    (void)stride;
    for (IDX i = i0 + (IDX)threadIdx.x; i < i1; i += (IDX)blockDim.x) {
        IDX t = i;
        const int cg = (int)(t % (IDX)cvn); t /= (IDX)cvn;
        const int ox = (int)(t % (IDX)OW); t /= (IDX)OW;
        const int oy = (int)(t % (IDX)OH);
        const int n = (int)(t / (IDX)OH);
        const int c = cg * V;
        const vec sc = PV::ld(a.scale + c), sh = PV::ld(a.shift + c);
        vec best = PV::splat(-INFINITY);
        int bi[V];
#pragma unroll
        for (int e = 0; e < V; ++e) bi[e] = 0;
        // all nine taps are requested before the first is used (border taps from a clamped, always valid address; they are
        // dropped below): with `continue` on the border tests every load sat in a block of its own behind an s_waitcnt vmcnt(0)
        // -- nine memory latencies in a row per output element
        typename Act<T>::raw raw[9];
        bool ok[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = 2 * oy - 1 + r;
            const bool oky = (unsigned)iy < (unsigned)a.H;
            const int cy = oky ? iy : oy * 2;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ix = 2 * ox - 1 + s;
                const bool okx = (unsigned)ix < (unsigned)a.W;
                const int cx = okx ? ix : ox * 2;
                ok[r * 3 + s] = oky && okx;
                raw[r * 3 + s] = Act<T>::ldr(y + ((size_t)(n * a.H + cy) * a.W + (size_t)cx) * a.C + c);
            }
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            vec v = Act<T>::cvt(raw[tap]);
            v = v * sc + sh;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float z = ok[tap] ? fmaxf(v[e], 0.f) : -INFINITY;      // (-inf never beats the running maximum)
                if (z > best[e]) { best[e] = z; bi[e] = tap; }
            }
        }
        Act<T>::stv(pout + i * V, best);
        if (a.idx) {
#pragma unroll
            for (int q = 0; q < V / 4; ++q) {
                uchar4 u;
                u.x = (unsigned char)bi[4 * q]; u.y = (unsigned char)bi[4 * q + 1]; u.z = (unsigned char)bi[4 * q + 2]; u.w = (unsigned char)bi[4 * q + 3];
                reinterpret_cast<uchar4*>(a.idx)[i * (V / 4) + q] = u;
            }
        }
    }
}

// Backward: for every stem-output element gather the pooled gradients whose arg-max
// is this element, apply the ReLU mask, store g and reduce (sum g, sum g*xhat).
template <typename T, typename IDX>      // IDX = unsigned when the pixel index fits 32 bits (three 64-bit divisions per pixel otherwise)
__global__ __launch_bounds__(256) void maxpool_relu_bwd_reduce_k(PoolBwdArgs a)
{
    constexpr int V = Act<T>::kVec;
    using vec = typename Act<T>::vec;
    using PV = ParamVec<V>;
    __shared__ __attribute__((aligned(16))) float red[2 * 256 * V];
    const T* dp = static_cast<const T*>(a.dp);
    const T* y = static_cast<const T*>(a.y);
    T* gout = static_cast<T*>(a.g);
    const int OH = a.H / 2, OW = a.W / 2;
    const int cvn = a.C / V;
    const int rl = 256 / cvn;
    const int cg = threadIdx.x % cvn;
    const int pl = threadIdx.x / cvn;
    const int c = cg * V;
    vec s1 = PV::splat(0.f), s2 = s1;
    if (pl < rl) {
        const vec sc = PV::ld(a.scale + c), sh = PV::ld(a.shift + c);
        const vec mean = PV::ld(a.mean + c), inv = PV::ld(a.invstd + c);
        const long long pixels = (long long)a.N * a.H * a.W;
        const long long p0 = (long long)xcd_major_block() * a.pix_per_block;
        long long p1 = p0 + a.pix_per_block;
        if (p1 > pixels) p1 = pixels;
        for (long long p = p0 + pl; p < p1; p += rl) {
            const IDX pi = (IDX)p;
            const int x = (int)(pi % (IDX)a.W);
            const IDX t = pi / (IDX)a.W;
            const int yy = (int)(t % (IDX)a.H);
            const int n = (int)(t / (IDX)a.H);
            vec g = PV::splat(0.f);
            const int oy0 = yy >> 1, oy1 = (yy + 1) >> 1;   // windows covering row y (equal when y is even)
            const int ox0 = x >> 1, ox1 = (x + 1) >> 1;
            // the (up to) 2 x 2 windows as a fixed set with validity flags, every load requested before the first use (row / column 0
            // always exist: oy0 <= OH - 1; the second exists when it differs and lies inside) -- the data-dependent loops with
            // `continue` put each window's two loads behind the previous window's wait
            const bool rok = oy1 != oy0 && oy1 < OH, cok = ox1 != ox0 && ox1 < OW;
            const int oyw[2] = {oy0, rok ? oy1 : oy0}, oxw[2] = {ox0, cok ? ox1 : ox0};
            typename Act<T>::raw draw[4];
            uchar4 uraw[4][V / 4];
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) {
                const size_t o = ((size_t)(n * OH + oyw[wv >> 1]) * OW + (size_t)oxw[wv & 1]) * cvn + cg;
                draw[wv] = Act<T>::ldr(dp + o * V);
#pragma unroll
                for (int q = 0; q < V / 4; ++q) uraw[wv][q] = reinterpret_cast<const uchar4*>(a.idx)[o * (V / 4) + q];
            }
            const typename Act<T>::raw yraw = Act<T>::ldr(y + (p * cvn + cg) * V);
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) {                // same order as the loops it replaces: (oy0, ox0), (oy0, ox1), (oy1, ox0), (oy1, ox1)
                const bool okw = ((wv >> 1) == 0 || rok) && ((wv & 1) == 0 || cok);
                const int tap = okw ? (yy - (2 * oyw[wv >> 1] - 1)) * 3 + (x - (2 * oxw[wv & 1] - 1)) : 255;     // 255: no arg-max index matches
                const vec d = Act<T>::cvt(draw[wv]);
#pragma unroll
                for (int q = 0; q < V / 4; ++q) {
                    const uchar4 u = uraw[wv][q];
                    if (u.x == tap) g[4 * q] += d[4 * q];
                    if (u.y == tap) g[4 * q + 1] += d[4 * q + 1];
                    if (u.z == tap) g[4 * q + 2] += d[4 * q + 2];
                    if (u.w == tap) g[4 * q + 3] += d[4 * q + 3];
                }
            }
            const vec v = Act<T>::cvt(yraw);
            const vec z = v * sc + sh;
#pragma unroll
            for (int e = 0; e < V; ++e) g[e] = z[e] > 0.f ? g[e] : 0.f;
            Act<T>::stv(gout + (p * cvn + cg) * V, g);
            s1 += g;
            s2 += g * (v - mean) * inv;
        }
    }
    PV::st(red + threadIdx.x * V, s1);
    PV::st(red + (256 + threadIdx.x) * V, s2);
    __syncthreads();
    if (threadIdx.x < cvn) {
        vec t1 = PV::splat(0.f), t2 = t1;
        for (int k = 0; k < rl; ++k) {
            t1 += PV::ld(red + (k * cvn + threadIdx.x) * V);
            t2 += PV::ld(red + (256 + k * cvn + threadIdx.x) * V);
        }
        float* dst = a.partial + (size_t)blockIdx.x * 2 * a.C;
        PV::st(dst + c, t1);
        PV::st(dst + a.C + c, t2);
    }
}

}  // namespace

int lbc_bn_relu_maxpool_fwd(const PoolFwdArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.C % 8 == 0 && a.H % 2 == 0 && a.W % 2 == 0, "maxpool: bad shape");
    const long long total = (long long)a.N * (a.H / 2) * (a.W / 2) * (a.C / 4);
    long long blocks = (total / (a.act_bf16 ? 2 : 1) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    LbcProfScope prof("bn_relu_maxpool_fwd", 0.0, (a.act_bf16 ? 2.0 : 4.0) * total * 4 * (4.0 + 1.0) + total * 4.0, s);
    const bool small = total + blocks * 256 < (1ll << 31);       // (the loop index passes `total` by less than one grid stride)
#define LBC_K(T, g)                                                                                                          \
    do {                                                                                                                     \
        if (small) hipLaunchKernelGGL((bn_relu_maxpool_fwd_k<T, unsigned>), dim3((unsigned)(g)), dim3(256), 0, s, a);        \
        else       hipLaunchKernelGGL((bn_relu_maxpool_fwd_k<T, long long>), dim3((unsigned)(g)), dim3(256), 0, s, a);       \
    } while (0)
    LBC_DISPATCH_ACT(a.act_bf16, LBC_K, blocks);
#undef LBC_K
    return lbc_check_launch("bn_relu_maxpool_fwd");
}

int lbc_pool_bwd_rows(int N, int H, int W, int C) { return lbc_chan_reduce_rows((long long)N * H * W, C); }

int lbc_maxpool_relu_bwd_reduce(PoolBwdArgs a, hipStream_t s)
{
    LBC_REQUIRE(a.C % 8 == 0 && a.C / 4 <= 256, "maxpool_bwd: bad C");
    const long long pixels = (long long)a.N * a.H * a.W;
    const int rows = lbc_pool_bwd_rows(a.N, a.H, a.W, a.C);
    a.pix_per_block = (pixels + rows - 1) / rows;
    LbcProfScope prof("maxpool_relu_bwd_reduce", 0.0, (a.act_bf16 ? 2.0 : 4.0) * (double)pixels * a.C * (2.0 + 0.25) + (double)pixels * a.C * 0.25, s);
    const bool small = pixels < (1ll << 31);
#define LBC_K(T, g)                                                                                                              \
    do {                                                                                                                         \
        if (small) hipLaunchKernelGGL((maxpool_relu_bwd_reduce_k<T, unsigned>), dim3((unsigned)(g)), dim3(256), 0, s, a);        \
        else       hipLaunchKernelGGL((maxpool_relu_bwd_reduce_k<T, long long>), dim3((unsigned)(g)), dim3(256), 0, s, a);       \
    } while (0)
    LBC_DISPATCH_ACT(a.act_bf16, LBC_K, rows);
#undef LBC_K
    return lbc_check_launch("maxpool_relu_bwd_reduce");
}
