// Instantiates the cosine-attention kernels for rows read 1 float(s) per lane.
#include "attn_impl.h"

namespace sngnn {

int launch_attn_fwd_v1(const RowCfg &cfg, const AttnArgs &a, hipStream_t st)
{
    SNGNN_DISPATCH_GR(launch_attn_fwd, 1, cfg, a, st)
}

int launch_attn_bwd_v1(const RowCfg &cfg, const BwdArgs &a, hipStream_t st)
{
    SNGNN_DISPATCH_GR(launch_attn_bwd, 1, cfg, a, st)
}

}  // namespace sngnn
