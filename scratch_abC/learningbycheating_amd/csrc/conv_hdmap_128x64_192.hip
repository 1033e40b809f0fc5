// conv_hdmap.hpp instantiated for one tile shape (see conv_hdmap.hip): 128 x 64 tiles, four waves (2 x 2, 64 x 32 each), 78 KB of LDS
#include "conv_hdmap.hpp"

int lbc_conv_hdmap_launch_128x64_192(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, unsigned grid, hipStream_t s, int nsplit, int kgroups)
{
    return conv_hdmap_launch_shape<128, 64, 2, 2, 192, 16>(a, mode, zero, ntiles, tpw, dim3(grid), s, nsplit, kgroups);
}
