// Instantiates the fused aggregation forward for rows read 2 float(s) per lane.
#include "agg_fwd_impl.h"

namespace sngnn {

int launch_agg_fwd_v2(const RowCfg &cfg, const FwdArgs &a, int max_split_deg, hipEvent_t *ev,
                      hipStream_t st)
{
    SNGNN_DISPATCH_GR(launch_agg_fwd, 2, cfg, a, max_split_deg, ev, st)
}

int launch_normalize_v2(const RowCfg &cfg, const float *h, int64_t rows, int C, float *n, float *nrm,
                        void *filt, hipStream_t st)
{
    SNGNN_DISPATCH_GR(launch_normalize_rows, 2, cfg, h, rows, C, n, nrm, filt, st)
}

}  // namespace sngnn
