// conv_hdmaw.hpp (wave-specialised persistent halo-staged convolution) instantiated for image rows of at most 27 pixels (layers 3 / 4)
#include "conv_hdmaw.hpp"

int lbc_conv_hdmaw_launch_256x128_320(const IgemmArgs& a, int mode, const void* zero, int ntiles, int tpw, unsigned grid, hipStream_t s)
{
    return conv_hdmaw_launch_shape<256, 128, 320, 32>(a, mode, zero, ntiles, tpw, dim3(grid), s);
}
