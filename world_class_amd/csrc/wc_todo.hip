// Entry points of include/world_class_c.h whose device implementation has not landed yet.
// They fail loudly (no CPU fallback); each block is deleted when the stage's .hip file arrives.
#include "wc_internal.hpp"
using namespace wc;
extern "C" {
wc_harvest *wc_harvest_create(int, double, double, double, double, double, int) { set_error("harvest: not implemented yet"); return nullptr; }
void wc_harvest_destroy(wc_harvest *) {}
int wc_harvest_compute(wc_harvest *, const double *, int, double *, double *) { return fail(WC_ERR_UNSUPPORTED, "harvest: not implemented yet"); }
int wc_harvest_compute_device(wc_harvest *, int, const double *, const int *, double *, double *) { return fail(WC_ERR_UNSUPPORTED, "harvest: not implemented yet"); }
wc_synthesis *wc_synthesis_create(int, int, double) { set_error("synthesis: not implemented yet"); return nullptr; }
void wc_synthesis_destroy(wc_synthesis *) {}
int wc_synthesis_compute(wc_synthesis *, const double *, int, const double *const *, const double *const *, int, double *) { return fail(WC_ERR_UNSUPPORTED, "synthesis: not implemented yet"); }
int wc_synthesis_compute_device(wc_synthesis *, int, const double *, const int *, const double *, const double *, const int *, double *, uint64_t *) { return fail(WC_ERR_UNSUPPORTED, "synthesis: not implemented yet"); }
}
