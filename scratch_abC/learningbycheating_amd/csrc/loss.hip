// Waypoint losses of the LbC training phases, forward + analytic gradient in one
// tiny kernel each (one thread per (sample, row); a row is one (x, y) waypoint).
//   phase 1: reference training/train_image_phase1.py:35-70 -- student camera-space
//            waypoints -> differentiable unprojection to the teacher's map frame ->
//            L1 over all rows -> per-sample mean.
//   phase 0: reference training/train_image_phase0.py:36-89 -- teacher map waypoints
//            -> metres -> pinhole projection (cv2.projectPoints with zero rvec/tvec,
//            done there on the CPU in float64) -> clip to the image -> L1 against the
//            student's normalised image-space prediction.
//   l1:      reference training/train_birdview.py:33-54 (choice='l1').
// loss_per_sample[n] = mean over the sample's R rows x 2 coordinates of |.|;
// dpred = d(grad_scale * sum_n loss_per_sample[n]) / dpred.
#include "lbc_common.hpp"
#include "lbc_kernels.hpp"

namespace {

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// block = one sample; thread r < R handles one row; reduction over rows in LDS
template <int KIND>
__global__ __launch_bounds__(64) void loss_k(LossArgs a, float t_scale, float t_shift)
{
    __shared__ float red[64];
    const int n = blockIdx.x, r = threadIdx.x;
    float part = 0.f;
    if (r < a.R) {
        const size_t o = ((size_t)n * a.R + r) * 2;
        const float x = a.pred[o], y = a.pred[o + 1];
        const float tx = a.target[o], ty = a.target[o + 1];
        const float gs = a.grad_scale / (float)(a.R * 2);
        float gx = 0.f, gy = 0.f;
        if (KIND == 1) {
            // camera (normalised) -> pixels -> rays -> ground plane -> map pixels -> normalised map
            const float f = (float)((double)a.w / (2.0 * tan((double)a.fov * 3.14159265358979323846 / 360.0)));
            const float cx = (x + 1.f) * a.w / 2.f, cy = (y + 1.f) * a.h / 2.f;
            const float xt = (cx - a.w / 2.f) / f;
            const float yt = (cy - a.h / 2.f) / f;
            const float wz = a.world_y / yt;
            const float wx = wz * xt;
            float mx = wx * a.pixels_per_meter;
            float my = wz * a.pixels_per_meter;
            my = a.crop_size - my;
            mx += a.crop_size / 2.f;
            my += a.fixed_offset * a.pixels_per_meter;
            const float half = 0.5f * a.crop_size;
            const float px = mx / half - 1.f, py = my / half - 1.f;
            const float ex = px - tx, ey = py - ty;
            part = fabsf(ex) + fabsf(ey);
            const float gmx = sgn(ex) * gs, gmy = sgn(ey) * gs;
            const float k = a.pixels_per_meter / half;
            // d px/d xt = wz*k ; d px/d wz = xt*k ; d py/d wz = -k ; d wz/d yt = -world_y/yt^2
            const float dxt = (a.w / 2.f) / f, dyt = (a.h / 2.f) / f;
            gx = gmx * wz * k * dxt;
            gy = (gmx * xt * k - gmy * k) * (-a.world_y / (yt * yt)) * dyt;
        } else if (KIND == 0) {
            // teacher map (normalised) -> image pixels, in double like the numpy/cv2 path
            double mxp = ((double)(float)((tx + 1.f) * a.crop_size / 2.f));
            double myp = ((double)(float)((ty + 1.f) * a.crop_size / 2.f));
            float fy = a.crop_size - (float)myp;
            float fx = (float)mxp - a.crop_size / 2.f;
            fx = fx / a.pixels_per_meter;
            fy = fy / a.pixels_per_meter;
            fy += a.fixed_offset;
            const double f = (double)a.w / (2.0 * tan((double)a.fov * 3.14159265358979323846 / 360.0));
            const double X = (double)fx, Y = (double)a.world_y, Z = (double)fy;
            double u = f * X / Z + (double)a.w / 2.0;
            double v = f * Y / Z + (double)a.h / 2.0;
            u = u < 0.0 ? 0.0 : (u > (double)a.w ? (double)a.w : u);
            v = v < 0.0 ? 0.0 : (v > (double)a.h ? (double)a.h : v);
            const float tu = (float)u / (0.5f * a.w) - 1.f;
            const float tv = (float)v / (0.5f * a.h) - 1.f;
            const float ex = x - tu, ey = y - tv;
            part = fabsf(ex) + fabsf(ey);
            gx = sgn(ex) * gs; gy = sgn(ey) * gs;
        } else {
            const float ex = x - (tx * t_scale + t_shift), ey = y - (ty * t_scale + t_shift);
            part = fabsf(ex) + fabsf(ey);
            gx = sgn(ex) * gs; gy = sgn(ey) * gs;
        }
        if (a.dpred) { a.dpred[o] = gx; a.dpred[o + 1] = gy; }
    }
    red[r] = part;
    __syncthreads();
    if (r == 0) {
        float t = 0.f;
        for (int i = 0; i < a.R; ++i) t += red[i];
        a.loss_per_sample[n] = t / (float)(a.R * 2);
    }
}

// Phase-2 (DAgger) resampling weight of every sample, reference training/phase2_utils.py:50-59 (get_weight) applied as in
// train_image_phase2.py:203-206: the student's SELECTED-branch camera-space prediction is unprojected to the map frame and
// normalised (train_image_phase1-style CoordConverter + /(0.5*CROP_SIZE) - 1), then
//   w[n] = mean_t( (0.7*|dx| + 0.3*|dy|) * 0.7^t )   over the 5 waypoints.
__global__ __launch_bounds__(64) void phase2_weight_k(LossArgs a)
{
    __shared__ float red[8];
    const int n = blockIdx.x, r = threadIdx.x;
    float part = 0.f;
    if (r < 5) {
        const size_t o = ((size_t)n * 5 + r) * 2;
        const float x = a.pred[o], y = a.pred[o + 1];
        const float f = (float)((double)a.w / (2.0 * tan((double)a.fov * 3.14159265358979323846 / 360.0)));
        const float cx = (x + 1.f) * a.w / 2.f, cy = (y + 1.f) * a.h / 2.f;
        const float xt = (cx - a.w / 2.f) / f;
        const float yt = (cy - a.h / 2.f) / f;
        const float wz = a.world_y / yt;
        const float wx = wz * xt;
        float mx = wx * a.pixels_per_meter;
        float my = wz * a.pixels_per_meter;
        my = a.crop_size - my;
        mx += a.crop_size / 2.f;
        my += a.fixed_offset * a.pixels_per_meter;
        const float half = 0.5f * a.crop_size;
        const float px = mx / half - 1.f, py = my / half - 1.f;
        float decay = 1.f;
        for (int i = 0; i < r; ++i) decay *= 0.7f;
        part = (fabsf(px - a.target[o]) * 0.7f + fabsf(py - a.target[o + 1]) * 0.3f) * decay;
    }
    if (r < 8) red[r] = part;
    __syncthreads();
    if (r == 0) a.loss_per_sample[n] = (red[0] + red[1] + red[2] + red[3] + red[4]) / 5.f;
}

}  // namespace

int lbc_phase2_weight_launch(const LossArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.N > 0, "phase2_weight: bad shape");
    hipLaunchKernelGGL(phase2_weight_k, dim3((unsigned)a.N), dim3(64), 0, s, a);
    return lbc_check_launch("phase2_weight");
}

int lbc_loss_phase1(const LossArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.R > 0 && a.R <= 64 && a.N > 0, "loss: bad shape");
    hipLaunchKernelGGL((loss_k<1>), dim3((unsigned)a.N), dim3(64), 0, s, a, 0.f, 0.f);
    return lbc_check_launch("loss_phase1");
}
int lbc_loss_phase0(const LossArgs& a, hipStream_t s)
{
    LBC_REQUIRE(a.R > 0 && a.R <= 64 && a.N > 0, "loss: bad shape");
    hipLaunchKernelGGL((loss_k<0>), dim3((unsigned)a.N), dim3(64), 0, s, a, 0.f, 0.f);
    return lbc_check_launch("loss_phase0");
}
int lbc_loss_l1(const LossArgs& a, float target_scale, float target_shift, hipStream_t s)
{
    LBC_REQUIRE(a.R > 0 && a.R <= 64 && a.N > 0, "loss: bad shape");
    hipLaunchKernelGGL((loss_k<2>), dim3((unsigned)a.N), dim3(64), 0, s, a, target_scale, target_shift);
    return lbc_check_launch("loss_l1");
}
