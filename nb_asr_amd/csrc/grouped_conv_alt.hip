// fp32 A/B variants of the fused grouped-convolution node kernel (grouped_conv_impl.h): 8 frames per lane and / or the
// [group][ci][tap][co] weight layout.  The default (4 frames per lane, torch weight layout) lives in grouped_conv.hip.
#include "grouped_conv_impl.h"

namespace nbasr {

int grouped_conv_f32_alt(int variant, const GroupedArgs<float>& a, int kernel, int dilation, hipStream_t stream)
{
    switch (variant) {
        case NBASR_GC_FPL8:                  return grouped_conv_variant<float, 8, false>(a, kernel, dilation, stream);
        case NBASR_GC_WPERM:                 return grouped_conv_variant<float, 4, true>(a, kernel, dilation, stream);
        case NBASR_GC_FPL8 | NBASR_GC_WPERM: return grouped_conv_variant<float, 8, true>(a, kernel, dilation, stream);
        default:
            set_error("nbasr_grouped_conv1d_node: fp32 variant %d does not exist", variant);
            return NBASR_EINVAL;
    }
}

}  // namespace nbasr
