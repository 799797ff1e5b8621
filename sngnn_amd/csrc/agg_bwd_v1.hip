// Instantiates the aggregation backward for rows read 1 float(s) per lane.
#include "agg_bwd_impl.h"

namespace sngnn {

int launch_agg_bwd_v1(const RowCfg &cfg, const BwdArgs &a, hipStream_t st)
{
    SNGNN_DISPATCH_GR(launch_agg_bwd, 1, cfg, a, st)
}

}  // namespace sngnn
