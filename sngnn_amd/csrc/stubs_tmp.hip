// TEMPORARY: entry points not implemented yet (fail loudly).
#include "common.h"
using namespace sngnn;
extern "C" {
int sngnn_cosine_dense(const float *, int64_t, int64_t, float *, void *)
{ set_error("not implemented"); return SNGNN_EINVAL; }
int sngnn_cosine_class_sums(const float *, int64_t, int64_t, const int32_t *, int, double *, double *, void *)
{ set_error("not implemented"); return SNGNN_EINVAL; }
int sngnn_edge_cosine(const float *, int64_t, int64_t, const int64_t *, int64_t, float *, void *)
{ set_error("not implemented"); return SNGNN_EINVAL; }
}
