// Stem tail for gfx950: BatchNorm + ReLU + MaxPool2d(3,2,1) fused in one pass over
// the largest activation of the network (N x 80 x 192 x 64 for the RGB model), and
// the matching backward (max-pool scatter as a gather + ReLU mask + BatchNorm
// gradient reductions) in one pass.  reference: bird_view/models/resnet.py:104-106,
// 149-152.  The forward stores the arg-max tap (0..8, first maximum in row-major
// order, as torch does) per pooled element so the backward never re-scans windows.
#include "lbc_common.hpp"
#include "lbc_kernels.hpp"

namespace {

__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_k(PoolFwdArgs a)
{
    const int OH = a.H / 2, OW = a.W / 2;
    const int c4n = a.C / 4;
    const long long total = (long long)a.N * OH * OW * c4n;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int cg = (int)(i % c4n);
        long long t = i / c4n;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        const int c = cg * 4;
        const float4 sc = *reinterpret_cast<const float4*>(a.scale + c);
        const float4 sh = *reinterpret_cast<const float4*>(a.shift + c);
        const float ninf = -INFINITY;
        float4 best = make_float4(ninf, ninf, ninf, ninf);
        int bx = 0, by = 0, bz = 0, bw = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = 2 * oy - 1 + r;
            if ((unsigned)iy >= (unsigned)a.H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ix = 2 * ox - 1 + s;
                if ((unsigned)ix >= (unsigned)a.W) continue;
                float4 v = *reinterpret_cast<const float4*>(a.y + ((size_t)(n * a.H + iy) * a.W + (size_t)ix) * a.C + c);
                v.x = fmaxf(v.x * sc.x + sh.x, 0.f); v.y = fmaxf(v.y * sc.y + sh.y, 0.f);
                v.z = fmaxf(v.z * sc.z + sh.z, 0.f); v.w = fmaxf(v.w * sc.w + sh.w, 0.f);
                const int tap = r * 3 + s;
                if (v.x > best.x) { best.x = v.x; bx = tap; }
                if (v.y > best.y) { best.y = v.y; by = tap; }
                if (v.z > best.z) { best.z = v.z; bz = tap; }
                if (v.w > best.w) { best.w = v.w; bw = tap; }
            }
        }
        reinterpret_cast<float4*>(a.p)[i] = best;
        if (a.idx) {
            uchar4 u;
            u.x = (unsigned char)bx; u.y = (unsigned char)by; u.z = (unsigned char)bz; u.w = (unsigned char)bw;
            reinterpret_cast<uchar4*>(a.idx)[i] = u;
        }
    }
}

// Backward: for every stem-output element gather the pooled gradients whose arg-max
// is this element, apply the ReLU mask, store g and reduce (sum g, sum g*xhat).
__global__ __launch_bounds__(256) void maxpool_relu_bwd_reduce_k(PoolBwdArgs a)
{
    __shared__ __attribute__((aligned(16))) float red[2 * 256 * 4];
    const int OH = a.H / 2, OW = a.W / 2;
    const int c4n = a.C / 4;
    const int rl = 256 / c4n;
    const int cg = threadIdx.x % c4n;
    const int pl = threadIdx.x / c4n;
    const int c = cg * 4;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (pl < rl) {
        const float4 sc = *reinterpret_cast<const float4*>(a.scale + c);
        const float4 sh = *reinterpret_cast<const float4*>(a.shift + c);
        const float4 mean = *reinterpret_cast<const float4*>(a.mean + c);
        const float4 inv = *reinterpret_cast<const float4*>(a.invstd + c);
        const long long pixels = (long long)a.N * a.H * a.W;
        const long long p0 = (long long)blockIdx.x * a.pix_per_block;
        long long p1 = p0 + a.pix_per_block;
        if (p1 > pixels) p1 = pixels;
        for (long long p = p0 + pl; p < p1; p += rl) {
            const int x = (int)(p % a.W);
            const long long t = p / a.W;
            const int y = (int)(t % a.H);
            const int n = (int)(t / a.H);
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            const int oy0 = y >> 1, oy1 = (y + 1) >> 1;   // windows covering row y (equal when y is even)
            const int ox0 = x >> 1, ox1 = (x + 1) >> 1;
            for (int oy = oy0; oy <= oy1; ++oy) {
                if (oy >= OH) continue;
                const int r = y - (2 * oy - 1);
                for (int ox = ox0; ox <= ox1; ++ox) {
                    if (ox >= OW) continue;
                    const int s = x - (2 * ox - 1);
                    const int tap = r * 3 + s;
                    const size_t o = ((size_t)(n * OH + oy) * OW + (size_t)ox) * c4n + cg;
                    const uchar4 u = reinterpret_cast<const uchar4*>(a.idx)[o];
                    const float4 d = reinterpret_cast<const float4*>(a.dp)[o];
                    if (u.x == tap) g.x += d.x;
                    if (u.y == tap) g.y += d.y;
                    if (u.z == tap) g.z += d.z;
                    if (u.w == tap) g.w += d.w;
                }
            }
            const float4 v = reinterpret_cast<const float4*>(a.y)[p * c4n + cg];
            g.x = (v.x * sc.x + sh.x) > 0.f ? g.x : 0.f; g.y = (v.y * sc.y + sh.y) > 0.f ? g.y : 0.f;
            g.z = (v.z * sc.z + sh.z) > 0.f ? g.z : 0.f; g.w = (v.w * sc.w + sh.w) > 0.f ? g.w : 0.f;
            reinterpret_cast<float4*>(a.g)[p * c4n + cg] = g;
            s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
            s2.x += g.x * (v.x - mean.x) * inv.x; s2.y += g.y * (v.y - mean.y) * inv.y;
            s2.z += g.z * (v.z - mean.z) * inv.z; s2.w += g.w * (v.w - mean.w) * inv.w;
        }
    }
    reinterpret_cast<float4*>(red)[threadIdx.x] = s1;
    reinterpret_cast<float4*>(red)[256 + threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < c4n) {
        float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
        for (int k = 0; k < rl; ++k) {
            const float4 u = reinterpret_cast<const float4*>(red)[k * c4n + threadIdx.x];
            const float4 w = reinterpret_cast<const float4*>(red)[256 + k * c4n + threadIdx.x];
            t1.x += u.x; t1.y += u.y; t1.z += u.z; t1.w += u.w;
            t2.x += w.x; t2.y += w.y; t2.z += w.z; t2.w += w.w;
        }
        float* dst = a.partial + (size_t)blockIdx.x * 2 * a.C;
        *reinterpret_cast<float4*>(dst + c) = t1;
        *reinterpret_cast<float4*>(dst + a.C + c) = t2;
    }
}

}  // namespace

int lbc_bn_relu_maxpool_fwd(const PoolFwdArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.C % 4 == 0 && a.H % 2 == 0 && a.W % 2 == 0, "maxpool: bad shape");
    const long long total = (long long)a.N * (a.H / 2) * (a.W / 2) * (a.C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    LbcProfScope prof("bn_relu_maxpool_fwd", 0.0, 4.0 * total * 4 * (4.0 + 1.0 + 0.25), s);
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_k, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return lbc_check_launch("bn_relu_maxpool_fwd");
}

int lbc_pool_bwd_rows(int N, int H, int W, int C) { return lbc_chan_reduce_rows((long long)N * H * W, C); }

int lbc_maxpool_relu_bwd_reduce(PoolBwdArgs a, hipStream_t s)
{
    LBC_REQUIRE(a.C % 4 == 0 && a.C / 4 <= 256, "maxpool_bwd: bad C");
    const long long pixels = (long long)a.N * a.H * a.W;
    const int rows = lbc_pool_bwd_rows(a.N, a.H, a.W, a.C);
    a.pix_per_block = (pixels + rows - 1) / rows;
    LbcProfScope prof("maxpool_relu_bwd_reduce", 0.0, 4.0 * (double)pixels * a.C * (2.0 + 0.25 + 0.0625), s);
    hipLaunchKernelGGL(maxpool_relu_bwd_reduce_k, dim3((unsigned)rows), dim3(256), 0, s, a);
    return lbc_check_launch("maxpool_relu_bwd_reduce");
}
