// Drop-in for reference include/cheaptrick.hpp:14-38 (CheapTrickOption, CheapTrick) backed by the HIP library.
#ifndef WORLD_CLASS_CHEAPTRICK_HPP
#define WORLD_CLASS_CHEAPTRICK_HPP

#include "world_class_common.hpp"

namespace world_class {

typedef struct CheapTrickOption {
	double q1;
	double f0_floor;
	int fft_size;
	CheapTrickOption() : q1(-0.15), f0_floor(71.0), fft_size(0) {}  // reference src/cheaptrick.cpp:22-24
} CheapTrickOption;

class CheapTrick {
public:
	explicit CheapTrick(int fs) : CheapTrick(fs, CheapTrickOption()) {}
	CheapTrick(int fs, const CheapTrickOption &option)
		: c_(detail::checked(wc_cheaptrick_create(fs, option.q1, option.f0_floor, option.fft_size), "CheapTrick")) {}
	~CheapTrick() { wc_cheaptrick_destroy(c_); }
	CheapTrick(const CheapTrick &) = delete;
	CheapTrick &operator=(const CheapTrick &) = delete;
	CheapTrick(CheapTrick &&o) noexcept : c_(o.c_) { o.c_ = nullptr; }  // movable: `CheapTrick x = CheapTrick(...)` of the reference's demo

	// reference src/cheaptrick.cpp:48-95; spectrogram[i] points at fft_size / 2 + 1 doubles (rows need not be contiguous).
	// Noise draws use and advance the process-wide stream position like the reference's global randn().
	void compute(const double *x, int x_length, const double *temporal_positions, const double *f0, int f0_length,
				 double **spectrogram) {
		detail::check(wc_cheaptrick_compute(c_, x, x_length, temporal_positions, f0, f0_length, spectrogram), "CheapTrick::compute");
	}
	int getFFTSizeForCheapTrick(int fs, double f0_floor) { return wc_cheaptrick_fft_size(fs, f0_floor); }
	double getF0FloorForCheapTrick(int fs, int fft_size) { return wc_cheaptrick_f0_floor(fs, fft_size); }

	void computeDevice(int n_utt, const double *d_x, const int *x_length, const double *d_tpos, const double *d_f0,
					   const int *f0_length, double *d_sp, uint64_t *rng_pos = nullptr) {
		detail::check(wc_cheaptrick_compute_device(c_, n_utt, d_x, x_length, d_tpos, d_f0, f0_length, d_sp, rng_pos),
					  "CheapTrick::computeDevice");
	}

private:
	wc_cheaptrick *c_;
};

}  // namespace world_class

#endif
