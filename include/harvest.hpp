// Drop-in for reference include/harvest.hpp:16-44 (HarvestOption, Harvest) backed by the HIP library.
// Source compatible: same names, argument order and meaning; callers recompile against this header and
// link libworldclass_hip.so.  Objects are reusable across calls and utterance lengths; they are not
// copyable (the reference's implicit copy would double-free).
#ifndef WORLD_CLASS_HARVEST_HPP
#define WORLD_CLASS_HARVEST_HPP

#include "world_class_common.hpp"

namespace world_class {

typedef struct HarvestOption {
	double f0_floor;
	double f0_ceil;
	double frame_period;
	double target_fs;
	double channels_in_octave;
	bool use_cos_table;  // the refinement window from the reference's 8001-entry cosine table (src/harvest.cpp:152-170)

	// defaults of reference src/harvest.cpp:52-56
	HarvestOption() : f0_floor(71.0), f0_ceil(800.0), frame_period(5), target_fs(8000.), channels_in_octave(40.), use_cos_table(false) {}
	void copy(const HarvestOption &option) { *this = option; }
} HarvestOption;

class Harvest {
public:
	Harvest(const int fs, const HarvestOption &option)
		: option_(option), fs_(fs),
		  h_(detail::checked(wc_harvest_create(fs, option.f0_floor, option.f0_ceil, option.frame_period, option.target_fs,
											   option.channels_in_octave, option.use_cos_table ? 1 : 0),
							 "Harvest")) {}
	~Harvest() { wc_harvest_destroy(h_); }
	// not copyable (the object owns device workspaces); movable, so that the reference's own idiom
	// `Harvest harvest = Harvest(fs, option);` (test/test.cpp:97) compiles under C++11
	Harvest(const Harvest &) = delete;
	Harvest &operator=(const Harvest &) = delete;
	Harvest(Harvest &&o) noexcept : option_(o.option_), fs_(o.fs_), h_(o.h_) { o.h_ = nullptr; }

	// reference src/harvest.cpp:183-208
	void compute(const double *x, int x_length, double *temporal_positions, double *f0) {
		detail::check(wc_harvest_compute(h_, x, x_length, temporal_positions, f0), "Harvest::compute");
	}
	// reference src/harvest.cpp:173-181
	int getSamples(int fs, int x_length, double frame_period) { return wc_get_samples(fs, x_length, frame_period); }
	int getSamples(int fs, int x_length) { return wc_get_samples(fs, x_length, option_.frame_period); }

	// extension: packed batch resident in device memory (see world_class_c.h)
	void computeDevice(int n_utt, const double *d_x, const int *x_length, double *d_tpos, double *d_f0) {
		detail::check(wc_harvest_compute_device(h_, n_utt, d_x, x_length, d_tpos, d_f0), "Harvest::computeDevice");
	}

private:
	HarvestOption option_;
	int fs_;
	wc_harvest *h_;
};

}  // namespace world_class

#endif
