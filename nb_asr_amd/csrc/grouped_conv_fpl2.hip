// fp32 variant of the fused grouped-convolution node kernel with TWO frames per lane (8-byte accesses): twice as many, half as
// heavy waves.  For the narrow, wide-channel blocks (C = 1200 at 250 frames) where the 4-frame kernel's 6.2 waves per SIMD leave
// a quarter of the SIMDs running a 7th wave alone at the end of the launch (profiles/r02_pmc_valu_issue.csv).
#include "grouped_conv_impl.h"

namespace nbasr {

int grouped_conv_f32_fpl2(const GroupedArgs<float>& a, int kernel, int dilation, hipStream_t stream)
{
    return grouped_conv_variant<float, 2, false>(a, kernel, dilation, stream);
}

}  // namespace nbasr
