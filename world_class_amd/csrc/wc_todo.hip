// Entry points of include/world_class_c.h whose device implementation has not landed yet.
// They fail loudly (no CPU fallback); each block is deleted when the stage's .hip file arrives.
#include "wc_internal.hpp"
using namespace wc;
extern "C" {
wc_harvest *wc_harvest_create(int, double, double, double, double, double, int) { set_error("harvest: not implemented yet"); return nullptr; }
void wc_harvest_destroy(wc_harvest *) {}
int wc_harvest_compute(wc_harvest *, const double *, int, double *, double *) { return fail(WC_ERR_UNSUPPORTED, "harvest: not implemented yet"); }
int wc_harvest_compute_device(wc_harvest *, int, const double *, const int *, double *, double *) { return fail(WC_ERR_UNSUPPORTED, "harvest: not implemented yet"); }
}
