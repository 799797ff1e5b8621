// Instantiates the fused aggregation forward WITH the hidden-layer store epilogue (rows read 4
// floats per lane: the widths hidden layers have).
#include "agg_fwd_impl.h"

namespace sngnn {

int launch_agg_fwd_epi_v4(const RowCfg &cfg, const FwdArgs &a, int max_split_deg, hipEvent_t *ev,
                          hipStream_t st)
{
    SNGNN_DISPATCH_GR(launch_agg_fwd_epi, 4, cfg, a, max_split_deg, ev, st)
}

}  // namespace sngnn
