// Drop-in for reference include/d4c.hpp:16-36 (D4COption, D4C) backed by the HIP library.
#ifndef WORLD_CLASS_D4C_HPP
#define WORLD_CLASS_D4C_HPP

#include "world_class_common.hpp"

namespace world_class {

typedef struct D4COption {
	double threshold;
	D4COption() : threshold(0.85) {}  // reference src/d4c.cpp:31-33
} D4COption;

class D4C {
public:
	explicit D4C(int fs) : D4C(fs, D4COption()) {}
	D4C(int fs, const D4COption &option) : d_(detail::checked(wc_d4c_create(fs, option.threshold), "D4C")) {}
	~D4C() { wc_d4c_destroy(d_); }
	D4C(const D4C &) = delete;
	D4C &operator=(const D4C &) = delete;
	D4C(D4C &&o) noexcept : d_(o.d_) { o.d_ = nullptr; }  // movable: `D4C x = D4C(...)` of the reference's demo

	// reference src/d4c.cpp:113-173; aperiodicity[i] points at fft_size / 2 + 1 doubles
	void compute(const double *x, int x_length, const double *temporal_positions, const double *f0, int f0_length, int fft_size,
				 double **aperiodicity) {
		detail::check(wc_d4c_compute(d_, x, x_length, temporal_positions, f0, f0_length, fft_size, aperiodicity), "D4C::compute");
	}
	void computeDevice(int n_utt, const double *d_x, const int *x_length, const double *d_tpos, const double *d_f0,
					   const int *f0_length, int fft_size, double *d_ap, uint64_t *rng_pos = nullptr) {
		detail::check(wc_d4c_compute_device(d_, n_utt, d_x, x_length, d_tpos, d_f0, f0_length, fft_size, d_ap, rng_pos),
					  "D4C::computeDevice");
	}

private:
	wc_d4c *d_;
};

}  // namespace world_class

#endif
