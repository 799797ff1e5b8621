// Instantiates the signed cosine-attention kernels for rows read 1 float(s) per lane.
#include "signed_impl.h"

namespace sngnn {

int launch_signed_fwd_v1(const RowCfg &cfg, const SignedArgs &a, hipStream_t st)
{
    SNGNN_DISPATCH_GR(launch_signed_fwd, 1, cfg, a, st)
}

int launch_signed_bwd_v1(const RowCfg &cfg, const BwdArgs &a, const SignedBwdExtra &x, hipStream_t st)
{
    SNGNN_DISPATCH_GR(launch_signed_bwd, 1, cfg, a, x, st)
}

}  // namespace sngnn
