// placeholder until the kd-tree FMM lands
#include "nbco_internal.hpp"
int fmm_kdtree_eval(nbco_ctx *c, float *, float *, long long, const float *) { return c->fail(NBCO_ERR_UNSUPPORTED, "kd-tree FMM not built yet"); }
int kd_copy_out(nbco_ctx *c, int, void *, long long) { return c->fail(NBCO_ERR_UNSUPPORTED, "kd-tree FMM not built yet"); }
