// Device-side input pipeline for the dataset's uint8 frames: the steps the reference runs in 8
// CPU dataloader workers per sample -- the fixed bird-view crop and the imgaug "super_hard" colour augmentation of the RGB
// frame -- as HBM-bound kernels over whole batches.
//   reference: bird_view/utils/datasets/image_lmdb.py:150-163 (crop rows/cols of the 320 x 320 x 7 map),
//              bird_view/augmenter.py:227-279 (super_hard: GaussianBlur, AdditiveGaussianNoise, CoarseDropout, Dropout, Add,
//              Multiply, ContrastNormalization, each with probability `frequency`, in random order, per-channel with
//              probability `color`), applied at image_lmdb.py:137-140.
// The augmentation is a RESTATEMENT of the recipe, not of imgaug's random stream (imgaug==0.2.8 is not installed and its
// numpy Mersenne-Twister draws cannot be reproduced on a GPU): per-image parameters are drawn on the host
// (bird_view/augmenter.py of this package) and per-pixel randomness comes from a counter-based hash of
// (image seed, operator, pixel | cell, channel), so results are reproducible and testable against a numpy restatement.
// Between operators a pixel is a uint8: every operator rounds to nearest and clips to [0, 255] like imgaug's uint8 paths.
#include "lbc_common.hpp"
#include "lbc_kernels.hpp"

namespace {

__device__ __forceinline__ unsigned hash_u32(unsigned x)
{
    // "lowbias32" integer finaliser (public domain, Chris Wellons): full avalanche in 2 multiplies
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned hash3(unsigned seed, unsigned a, unsigned b) { return hash_u32(seed ^ hash_u32(a * 0x9E3779B9U + hash_u32(b + 0x85EBCA6BU))); }
__device__ __forceinline__ float u01(unsigned h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ float clip_u8(float v) { return fminf(fmaxf(rintf(v), 0.f), 255.f); }

// bird-view crop: src [N][SH][SW][C] -> dst [N][H][W][C], window origin (y0, x0); 16 bytes per thread where rows allow
__global__ __launch_bounds__(256) void crop_u8_k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int N, int SH, int SW,
                                                 int C, int y0, int x0, int H, int W)
{
    const long long row_bytes = (long long)W * C;
    const long long total = (long long)N * H * row_bytes;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long r = i / row_bytes;
        const int b = (int)(i - r * row_bytes);
        const int n = (int)(r / H), y = (int)(r - (long long)n * H);
        dst[i] = src[(((long long)n * SH + (y0 + y)) * SW + x0) * C + b];
    }
}

// Bird-view rotation + crop, one pass: dst[n] = crop(warpAffine(src[n], M_n)) as the reference's privileged-agent loader does per sample
// on the CPU (bird_view/utils/datasets/birdview_lmdb.py:103-125: cv2.warpAffine(bird_view, cv2.getRotationMatrix2D((160, 260), delta_angle,
// 1.0), (320, 320), flags=cv2.INTER_LINEAR), then the jittered 192 x 192 window).  The arithmetic restates OpenCV's 8-bit bilinear
// warpAffine (imgwarp.cpp, remap with INTER_BITS = 5): source coordinates in 1/1024 fixed point from the INVERTED matrix (rounded half to
// even like cvRound, + 16, >> 5 -> 1/32 pixel), a 32 x 32 table of 15-bit weights whose four entries sum to 32768,
// (sum + 16384) >> 15, zero outside the image (BORDER_CONSTANT).
// p.im = the inverted matrix (iM0, iM1, b1, iM3, iM4, b2) in double, p.y0 / p.x0 the window origin in the warped image.
__global__ __launch_bounds__(256) void warp_crop_u8_k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, const WarpParams* __restrict__ params,
                                                      int N, int SH, int SW, int C, int H, int W)
{
    const long long total = (long long)N * H * W;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int n = (int)(i / ((long long)H * W));
        const int pix = (int)(i - (long long)n * H * W);
        const WarpParams p = params[n];
        const int y = p.y0 + pix / W, x = p.x0 + pix % W;
        const int adelta = (int)rint(p.im[0] * x * 1024.0), bdelta = (int)rint(p.im[3] * x * 1024.0);
        const int X0 = (int)rint((p.im[1] * y + p.im[2]) * 1024.0) + 16, Y0 = (int)rint((p.im[4] * y + p.im[5]) * 1024.0) + 16;
        const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
        const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
        // the weights of table entry (fy, fx)
        const float ax = (float)fx * (1.f / 32.f), ay = (float)fy * (1.f / 32.f);
        const float wf[4] = {(1.f - ay) * (1.f - ax), (1.f - ay) * ax, ay * (1.f - ax), ay * ax};
        // (bilinear weights are products of two multiples of 1/32: every w[k] is an exact multiple of 32 and the four sum to 32768
        //  exactly -- OpenCV's fix-up of the table sum can never fire for INTER_LINEAR, it is not restated)
        int w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (int)rintf(wf[k] * 32768.f);
        const bool in00 = (unsigned)sy < (unsigned)SH && (unsigned)sx < (unsigned)SW, in01 = (unsigned)sy < (unsigned)SH && (unsigned)(sx + 1) < (unsigned)SW;
        const bool in10 = (unsigned)(sy + 1) < (unsigned)SH && (unsigned)sx < (unsigned)SW, in11 = (unsigned)(sy + 1) < (unsigned)SH && (unsigned)(sx + 1) < (unsigned)SW;
        const unsigned char* s0 = src + (((long long)n * SH + sy) * SW + sx) * C;
        unsigned char* d = dst + i * C;
        for (int c = 0; c < C; ++c) {
            const int v00 = in00 ? s0[c] : 0, v01 = in01 ? s0[C + c] : 0, v10 = in10 ? s0[(long long)SW * C + c] : 0, v11 = in11 ? s0[(long long)SW * C + C + c] : 0;
            const int v = (v00 * w[0] + v01 * w[1] + v10 * w[2] + v11 * w[3] + (1 << 14)) >> 15;
            d[c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

// pointwise operators of the per-image sequence order[first[n] .. last[n])
__global__ __launch_bounds__(256) void aug_pointwise_k(unsigned char* __restrict__ img, const AugParams* __restrict__ params, int N, int H, int W,
                                                       int stage)
{
    const long long total = (long long)N * H * W;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int n = (int)(i / ((long long)H * W));
        const int pix = (int)(i - (long long)n * H * W);
        const AugParams& p = params[n];
        const int k0 = stage == 0 ? 0 : p.blur_pos + 1;
        const int k1 = stage == 0 ? (p.blur_pos < p.n_ops ? p.blur_pos : p.n_ops) : p.n_ops;
        if (k0 >= k1) continue;
        const int y = pix / W, x = pix - y * W;
        unsigned char* q = img + i * 3;
        float v[3] = {(float)q[0], (float)q[1], (float)q[2]};
        for (int k = k0; k < k1; ++k) {
            const int op = p.order[k];
            if (op == kAugNoise) {            // AdditiveGaussianNoise(loc 0, scale): one draw per pixel, or per channel
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const unsigned cc = p.noise_pc ? (unsigned)c : 0u;
                    const float u1 = u01(hash3(p.seed, 0x100u + cc, (unsigned)pix)) + (0.5f / 16777216.0f);
                    const float u2 = u01(hash3(p.seed, 0x110u + cc, (unsigned)pix));
                    const float z = sqrtf(-2.f * logf(u1)) * cosf(6.28318530718f * u2);
                    v[c] = clip_u8(v[c] + p.noise_scale * z);
                }
            } else if (op == kAugCoarseDropout) {   // mask drawn on a (cd_h x cd_w) grid, nearest-neighbour upsampled
                const unsigned cell = (unsigned)(((long long)y * p.cd_h / H) * p.cd_w + (long long)x * p.cd_w / W);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const unsigned cc = p.cd_pc ? (unsigned)c : 0u;
                    if (u01(hash3(p.seed, 0x200u + cc, cell)) < p.cd_p) v[c] = 0.f;
                }
            } else if (op == kAugDropout) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const unsigned cc = p.do_pc ? (unsigned)c : 0u;
                    if (u01(hash3(p.seed, 0x300u + cc, (unsigned)pix)) < p.do_p) v[c] = 0.f;
                }
            } else if (op == kAugAdd) {
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = clip_u8(v[c] + p.add_v[c]);
            } else if (op == kAugMultiply) {
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = clip_u8(v[c] * p.mul_v[c]);
            } else if (op == kAugContrast) {  // ContrastNormalization: 128 + alpha * (v - 128)
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = clip_u8(128.f + p.con_a[c] * (v[c] - 128.f));
            }
        }
        q[0] = (unsigned char)v[0]; q[1] = (unsigned char)v[1]; q[2] = (unsigned char)v[2];
    }
}

// separable Gaussian blur, kernel truncated at 4 sigma (scipy.ndimage.gaussian_filter's default), 'reflect' boundary
// (d c b a | a b c d | d c b a); pass 0: rows (src -> tmp as float), pass 1: columns (tmp -> dst, rounded).  Images whose
// sequence has no blur are copied through.
constexpr int kBlurRadiusMax = 16;
__device__ __forceinline__ int reflect(int i, int n)
{
    if (i < 0) i = -i - 1;
    if (i >= n) i = 2 * n - 1 - i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
__global__ __launch_bounds__(256) void aug_blur_k(const unsigned char* __restrict__ src, float* __restrict__ tmp, unsigned char* __restrict__ dst,
                                                  const AugParams* __restrict__ params, int N, int H, int W, int pass)
{
    const long long total = (long long)N * H * W;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int n = (int)(i / ((long long)H * W));
        const int pix = (int)(i - (long long)n * H * W);
        const AugParams& p = params[n];
        const bool on = p.blur_pos < p.n_ops && p.blur_sigma > 1e-3f;
        if (!on) continue;                                    // the image stays as it is (in place: src == dst)
        const int y = pix / W, x = pix - y * W;
        int rad = (int)(4.f * p.blur_sigma + 0.5f);
        if (rad > kBlurRadiusMax) rad = kBlurRadiusMax;
        const float inv2s2 = 0.5f / (p.blur_sigma * p.blur_sigma);
        float wsum = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int d = -rad; d <= rad; ++d) {
            const float w = expf(-(float)(d * d) * inv2s2);
            wsum += w;
            if (pass == 0) {
                const unsigned char* s = src + ((long long)n * H * W + (long long)y * W + reflect(x + d, W)) * 3;
                a0 += w * (float)s[0]; a1 += w * (float)s[1]; a2 += w * (float)s[2];
            } else {
                const float* s = tmp + ((long long)n * H * W + (long long)reflect(y + d, H) * W + x) * 3;
                a0 += w * s[0]; a1 += w * s[1]; a2 += w * s[2];
            }
        }
        const float inv = 1.f / wsum;
        if (pass == 0) {
            float* o = tmp + i * 3;
            o[0] = a0 * inv; o[1] = a1 * inv; o[2] = a2 * inv;
        } else {
            unsigned char* o = dst + i * 3;
            o[0] = (unsigned char)clip_u8(a0 * inv); o[1] = (unsigned char)clip_u8(a1 * inv); o[2] = (unsigned char)clip_u8(a2 * inv);
        }
    }
}

int grid_1d(long long total)
{
    long long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

int lbc_crop_u8(const unsigned char* src, unsigned char* dst, int N, int SH, int SW, int C, int y0, int x0, int H, int W, hipStream_t s)
{
    LBC_REQUIRE(src && dst && N > 0 && y0 >= 0 && x0 >= 0 && y0 + H <= SH && x0 + W <= SW && C > 0, "crop_u8: window outside the source");
    LbcProfScope prof("crop_u8", 0.0, 2.0 * N * H * (double)W * C, s);
    hipLaunchKernelGGL(crop_u8_k, dim3((unsigned)grid_1d((long long)N * H * W * C)), dim3(256), 0, s, src, dst, N, SH, SW, C, y0, x0, H, W);
    return lbc_check_launch("crop_u8");
}

int lbc_warp_crop_u8(const unsigned char* src, unsigned char* dst, const WarpParams* params_dev, int N, int SH, int SW, int C, int H, int W, hipStream_t s)
{
    LBC_REQUIRE(src && dst && params_dev && N > 0 && SH > 1 && SW > 1 && C > 0 && H > 0 && W > 0, "warp_crop_u8: bad arguments");
    LbcProfScope prof("warp_crop_u8", 0.0, 5.0 * N * H * (double)W * C, s);
    hipLaunchKernelGGL(warp_crop_u8_k, dim3((unsigned)grid_1d((long long)N * H * W)), dim3(256), 0, s, src, dst, params_dev, N, SH, SW, C, H, W);
    return lbc_check_launch("warp_crop_u8");
}

int lbc_augment_u8(unsigned char* img, const AugParams* params_dev, float* tmp, int N, int H, int W, int any_blur, hipStream_t s)
{
    LBC_REQUIRE(img && params_dev && N > 0 && H > 0 && W > 0, "augment_u8: bad arguments");
    LBC_REQUIRE(!any_blur || tmp, "augment_u8: the blur needs a float scratch image");
    const long long total = (long long)N * H * W;
    LbcProfScope prof("augment_u8", 0.0, 6.0 * total * (any_blur ? 4.0 : 1.0), s);
    hipLaunchKernelGGL(aug_pointwise_k, dim3((unsigned)grid_1d(total)), dim3(256), 0, s, img, params_dev, N, H, W, 0);
    if (any_blur) {
        hipLaunchKernelGGL(aug_blur_k, dim3((unsigned)grid_1d(total)), dim3(256), 0, s, img, tmp, img, params_dev, N, H, W, 0);
        hipLaunchKernelGGL(aug_blur_k, dim3((unsigned)grid_1d(total)), dim3(256), 0, s, img, tmp, img, params_dev, N, H, W, 1);
        hipLaunchKernelGGL(aug_pointwise_k, dim3((unsigned)grid_1d(total)), dim3(256), 0, s, img, params_dev, N, H, W, 1);
    }
    return lbc_check_launch("augment_u8");
}
