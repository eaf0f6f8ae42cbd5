// bf16-storage variants of the fused grouped-convolution node kernel (grouped_conv_impl.h): x / skips / y are bfloat16
// rows (8 frames per 16-byte lane access, or 4 per 8-byte access), every product and sum is fp32, the result is rounded
// once.  BASELINE config 4 (bf16): half the HBM bytes of the fp32 node op.
#include "grouped_conv_impl.h"

namespace nbasr {

int grouped_conv_bf16(int variant, const GroupedArgs<bf16_t>& a, int kernel, int dilation, hipStream_t stream)
{
    switch (variant) {
        case 0:                              return grouped_conv_variant<bf16_t, 4, false>(a, kernel, dilation, stream);
        case NBASR_GC_FPL8:                  return grouped_conv_variant<bf16_t, 8, false>(a, kernel, dilation, stream);
        case NBASR_GC_WPERM:                 return grouped_conv_variant<bf16_t, 4, true>(a, kernel, dilation, stream);
        case NBASR_GC_FPL8 | NBASR_GC_WPERM: return grouped_conv_variant<bf16_t, 8, true>(a, kernel, dilation, stream);
        default:
            set_error("nbasr_grouped_conv1d_node: bf16 variant %d does not exist", variant);
            return NBASR_EINVAL;
    }
}

}  // namespace nbasr
