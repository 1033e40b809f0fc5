// Multi-tensor Adam for gfx950: ONE launch updates every parameter tensor of the
// model (136 tensors / 23.1 M elements for the ResNet-34 student), 16 bytes per lane,
// reading p, g, m, v and writing p, m, v exactly once (28 B/element: HBM roofline).
// Semantics = torch.optim.Adam(lr, betas, eps, weight_decay, amsgrad=False) as called
// at reference training/train_image_phase1.py:252 (container torch 2.10 formulation):
//   m = m + (g - m)*(1-b1);  v = b2*v + (1-b2)*g*g
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
#include "lbc_common.hpp"
#include "lbc_kernels.hpp"

namespace {

__global__ __launch_bounds__(256) void adam_k(const AdamChunk* __restrict__ chunks, float lr_over_bc1, float inv_bc2_sqrt,
                                              float beta1, float beta2, float omb1, float omb2, float eps, float wd)
{
    const AdamChunk ch = chunks[blockIdx.x];
    const int n4 = ch.n >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
        float4 p = reinterpret_cast<float4*>(ch.p)[i];
        float4 g = reinterpret_cast<const float4*>(ch.g)[i];
        float4 m = reinterpret_cast<float4*>(ch.m)[i];
        float4 v = reinterpret_cast<float4*>(ch.v)[i];
#define LBC_ADAM1(c)                                                   \
        {                                                              \
            float gg = g.c + wd * p.c;                                 \
            m.c = m.c + (gg - m.c) * omb1;                    \
            v.c = beta2 * v.c + omb2 * gg * gg;               \
            const float denom = sqrtf(v.c) * inv_bc2_sqrt + eps;       \
            p.c = p.c - lr_over_bc1 * (m.c / denom);                   \
        }
        LBC_ADAM1(x) LBC_ADAM1(y) LBC_ADAM1(z) LBC_ADAM1(w)
        reinterpret_cast<float4*>(ch.p)[i] = p;
        reinterpret_cast<float4*>(ch.m)[i] = m;
        reinterpret_cast<float4*>(ch.v)[i] = v;
    }
    for (int i = (n4 << 2) + threadIdx.x; i < ch.n; i += 256) {
        float p = ch.p[i], m = ch.m[i], v = ch.v[i];
        const float gg = ch.g[i] + wd * p;
        m = m + (gg - m) * omb1;
        v = beta2 * v + omb2 * gg * gg;
        const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
        p = p - lr_over_bc1 * (m / denom);
        ch.p[i] = p; ch.m[i] = m; ch.v[i] = v;
    }
}

static long long g_adam_prof_elems = 0;      // lbc_adam_profile_elems: what the launch profiler books for an optimizer launch
}  // namespace

extern "C" void lbc_adam_profile_elems(long long n) { g_adam_prof_elems = n > 0 ? n : 0; }

int lbc_adam_launch(const AdamChunk* chunks_dev, int nchunks, double lr, double beta1, double beta2, double eps,
                    double weight_decay, int step, hipStream_t s)
{
    LBC_REQUIRE(nchunks > 0 && step >= 1, "adam: bad args");
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float lr_over_bc1 = (float)(lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    // algorithmic bytes: read p, g, m, v and write p, m, v = 7 x 4 bytes per element (648 MB for the 23.13 M parameters of the student)
    LbcProfScope prof("adam", 0.0, 28.0 * (double)g_adam_prof_elems, s);
    hipLaunchKernelGGL(adam_k, dim3((unsigned)nchunks), dim3(256), 0, s, chunks_dev, lr_over_bc1, inv_bc2_sqrt, (float)beta1,
                       (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay);
    return lbc_check_launch("adam");
}
