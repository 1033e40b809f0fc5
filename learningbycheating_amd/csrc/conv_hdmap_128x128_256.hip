// conv_hdmap.hpp instantiated for one tile shape (see conv_hdmap.hip): 128 x 128 tiles, eight waves (2 x 4, 64 x 32 each), one workgroup
// per CU (124 KB of LDS: two 256-row halo buffers hold image rows up to W = 59) -- round 5, for launches of fewer than 160 tiles of 256 x 128
#include "conv_hdmap.hpp"

int lbc_conv_hdmap_launch_128x128_256(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, unsigned grid, hipStream_t s)
{
    return conv_hdmap_launch_shape<128, 128, 2, 4, 256, 16>(a, mode, zero, ntiles, tpw, dim3(grid), s);
}
