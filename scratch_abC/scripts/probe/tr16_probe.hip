// Build: hipcc --offload-arch=gfx950 -O2 scripts/probe/tr16_probe.hip -o scripts/probe/tr16_probe ; run on the GPU box.
// Probe of ds_read_b64_tr_b16 (gfx950) semantics with per-lane addresses: each lane supplies the address of 4 contiguous
// bf16; within a 16-lane group lane t's chunk is taken as B[t>>2][(t&3)*4 .. +3] of a 4x16 matrix and lane t receives column t.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_v4;

__global__ void k(const int* offs, unsigned short* out)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;   // bit pattern = element index
    __syncthreads();
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(&lds[offs[threadIdx.x]]));
    unsigned short* p = reinterpret_cast<unsigned short*>(&v);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = p[j];
}

int main()
{
    int h_offs[64];
    unsigned short h_out[256];
    int *d_offs; unsigned short* d_out;
    hipMalloc(&d_offs, sizeof(h_offs)); hipMalloc(&d_out, sizeof(h_out));
    for (int variant = 0; variant < 2; ++variant) {
        // variant 0: linear (lane * 4 elements); variant 1: rows of 96 elements (192 B), row = px, col = channel:
        //   lane l: group G = l>>4, t = l&15 -> element offset ((t>>2) + 4*(G>>1)) * 96 + 16*(G&1) + (t&3)*4
        for (int l = 0; l < 64; ++l) {
            const int G = l >> 4, t = l & 15;
            h_offs[l] = variant == 0 ? l * 4 : ((t >> 2) + 4 * (G >> 1)) * 96 + 16 * (G & 1) + (t & 3) * 4;
        }
        hipMemcpy(d_offs, h_offs, sizeof(h_offs), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_offs, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                // model: element j of lane l = chunk of lane (l & ~15) + j*4 + ((l&15)>>2), sub-element (l&15)&3
                const int src = (l & ~15) + j * 4 + ((l & 15) >> 2);
                const int expect = h_offs[src] + ((l & 15) & 3);
                if (h_out[l * 4 + j] != expect) { if (bad < 8) printf("variant %d lane %d elem %d: got %d expected %d\n", variant, l, j, h_out[l * 4 + j], expect); ++bad; }
            }
        printf("variant %d: %s (%d mismatches)\n", variant, bad ? "MODEL WRONG" : "model ok", bad);
        if (variant == 0) { printf("lane0: %d %d %d %d  lane1: %d %d %d %d  lane17: %d %d %d %d\n", h_out[0], h_out[1], h_out[2], h_out[3], h_out[4], h_out[5], h_out[6], h_out[7], h_out[68], h_out[69], h_out[70], h_out[71]); }
    }
    return 0;
}
