// Drop-in for reference include/synthesis.hpp:29-51 (Synthesis) backed by the HIP library.
#ifndef WORLD_CLASS_SYNTHESIS_HPP
#define WORLD_CLASS_SYNTHESIS_HPP

#include "world_class_common.hpp"

namespace world_class {

class Synthesis {
public:
	// fs: sampling frequency, fft_size: FFT size of the spectrogram, frame_period: analysis hop in ms
	Synthesis(int fs, int fft_size, double frame_period)
		: s_(detail::checked(wc_synthesis_create(fs, fft_size, frame_period), "Synthesis")) {}
	~Synthesis() { wc_synthesis_destroy(s_); }
	Synthesis(const Synthesis &) = delete;
	Synthesis &operator=(const Synthesis &) = delete;
	Synthesis(Synthesis &&o) noexcept : s_(o.s_) { o.s_ = nullptr; }  // movable: `Synthesis x = Synthesis(...)` of the reference's demo

	// reference src/synthesis.cpp:77-177.  Defined (unlike the reference) for an all-unvoiced contour: noise only.
	// f0_length must be at least 2 (the reference reads f0[f0_length - 2]).
	void compute(const double *f0, int f0_length, const double *const *spectrogram, const double *const *aperiodicity, int out_length,
				 double *out) {
		detail::check(wc_synthesis_compute(s_, f0, f0_length, spectrogram, aperiodicity, out_length, out), "Synthesis::compute");
	}
	void computeDevice(int n_utt, const double *d_f0, const int *f0_length, const double *d_sp, const double *d_ap,
					   const int *out_length, double *d_out, uint64_t *rng_pos = nullptr) {
		detail::check(wc_synthesis_compute_device(s_, n_utt, d_f0, f0_length, d_sp, d_ap, out_length, d_out, rng_pos),
					  "Synthesis::computeDevice");
	}

private:
	wc_synthesis *s_;
};

}  // namespace world_class

#endif
