// TEMPORARY: entry points not implemented yet (fail loudly).
#include "common.h"
using namespace sngnn;
extern "C" {
int sngnn_agg_backward(const sngnn_graph_t *, const float *, int, const float *, const float *,
                       const float *, float *, void *, void *)
{ set_error("sngnn_agg_backward: not implemented"); return SNGNN_EINVAL; }
int sngnn_adj_linear_forward(const sngnn_graph_t *, const float *, const float *, int, float *, void *)
{ set_error("not implemented"); return SNGNN_EINVAL; }
int sngnn_adj_linear_backward(const sngnn_graph_t *, const float *, int, float *, void *)
{ set_error("not implemented"); return SNGNN_EINVAL; }
int sngnn_cosine_dense(const float *, int64_t, int64_t, float *, void *)
{ set_error("not implemented"); return SNGNN_EINVAL; }
int sngnn_cosine_class_sums(const float *, int64_t, int64_t, const int32_t *, int, double *, double *, void *)
{ set_error("not implemented"); return SNGNN_EINVAL; }
int sngnn_edge_cosine(const float *, int64_t, int64_t, const int64_t *, int64_t, float *, void *)
{ set_error("not implemented"); return SNGNN_EINVAL; }
}
