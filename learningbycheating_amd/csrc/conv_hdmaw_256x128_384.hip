// conv_hdmaw.hpp (wave-specialised persistent halo-staged convolution) instantiated for image rows of up to 59 pixels (layer 2)
#include "conv_hdmaw.hpp"

int lbc_conv_hdmaw_launch_256x128_384(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, unsigned grid, hipStream_t s)
{
    return conv_hdmaw_launch_shape<256, 128, 384, 16>(a, mode, zero, ntiles, tpw, dim3(grid), s);
}
