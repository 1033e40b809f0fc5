// conv_hdmap.hpp instantiated for one tile shape (see conv_hdmap.hip)
#include "conv_hdmap.hpp"

int lbc_conv_hdmap_launch_256x128_320(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, unsigned grid, hipStream_t s)
{
    return conv_hdmap_launch_shape<256, 128, 4, 2, 320, 16>(a, mode, zero, ntiles, tpw, dim3(grid), s);
}
