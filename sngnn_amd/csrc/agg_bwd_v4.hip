// Instantiates the aggregation backward for rows read 4 float(s) per lane.
#include "agg_bwd_impl.h"

namespace sngnn {

int launch_agg_bwd_v4(const RowCfg &cfg, const BwdArgs &a, hipStream_t st)
{
    SNGNN_DISPATCH_GR(launch_agg_bwd, 4, cfg, a, st)
}

}  // namespace sngnn
