// Drop-in for the reference's include/world_common.hpp (:17-129): the FFT buffer bundles (ForwardRealFFT, InverseRealFFT,
// InverseComplexFFT, MinimumPhaseAnalysis -- same fields, methods restating reference src/world_common.cpp:131-233) on top of
// world_fft.hpp, and the free helpers (declared in world_matlabfunctions.hpp of this repository together with the MATLAB ones).
#ifndef WORLD_COMMON_HPP
#define WORLD_COMMON_HPP

#include <math.h>

#include "world_fft.hpp"
#include "world_matlabfunctions.hpp"

#ifdef __cplusplus
#include <algorithm>
#include <cmath>
#include <complex>

typedef struct ForwardRealFFT {
	int fft_size;
	double *waveform;
	fft_complex *spectrum;
	fft_plan forward_fft;
	void initialize(int n) {
		fft_size = n;
		waveform = new double[n];
		spectrum = new fft_complex[n / 2 + 1];
		forward_fft = fft_plan_dft_r2c_1d(n, waveform, spectrum, FFT_ESTIMATE);
	}
	void destroy() {
		fft_destroy_plan(forward_fft);
		delete[] spectrum;
		delete[] waveform;
	}
} ForwardRealFFT;

typedef struct InverseRealFFT {
	int fft_size;
	double *waveform;
	fft_complex *spectrum;
	fft_plan inverse_fft;
	void initialize(int n) {
		fft_size = n;
		waveform = new double[n];
		spectrum = new fft_complex[n / 2 + 1];
		inverse_fft = fft_plan_dft_c2r_1d(n, spectrum, waveform, FFT_ESTIMATE);
	}
	void destroy() {
		fft_destroy_plan(inverse_fft);
		delete[] spectrum;
		delete[] waveform;
	}
} InverseRealFFT;

typedef struct InverseComplexFFT {
	int fft_size;
	fft_complex *input;
	fft_complex *output;
	fft_plan inverse_fft;
	void initialize(int n) {
		fft_size = n;
		input = new fft_complex[n];
		output = new fft_complex[n];
		inverse_fft = fft_plan_dft_1d(n, input, output, FFT_BACKWARD, FFT_ESTIMATE);
	}
	void destroy() {
		fft_destroy_plan(inverse_fft);
		delete[] input;
		delete[] output;
	}
} InverseComplexFFT;

typedef struct MinimumPhaseAnalysis {
	int fft_size;
	double *log_spectrum;
	fft_complex *minimum_phase_spectrum;
	fft_complex *cepstrum;
	fft_plan inverse_fft;
	fft_plan forward_fft;
	void initialize(int n) {
		fft_size = n;
		log_spectrum = new double[n];
		minimum_phase_spectrum = new fft_complex[n];
		cepstrum = new fft_complex[n];
		inverse_fft = fft_plan_dft_r2c_1d(n, log_spectrum, cepstrum, FFT_ESTIMATE);
		forward_fft = fft_plan_dft_1d(n, cepstrum, minimum_phase_spectrum, FFT_FORWARD, FFT_ESTIMATE);
	}
	void destroy() {
		fft_destroy_plan(forward_fft);
		fft_destroy_plan(inverse_fft);
		delete[] cepstrum;
		delete[] log_spectrum;
		delete[] minimum_phase_spectrum;
	}
	// log_spectrum[0 .. fft_size/2] in, minimum_phase_spectrum[0 .. fft_size/2] out: the causal part of the cepstrum of an even log
	// spectrum, transformed back and exponentiated (what reference src/world_common.cpp:196-233 computes, in this header's own terms)
	void compute() {
		const int half = fft_size / 2;
		std::reverse_copy(log_spectrum + 1, log_spectrum + half, log_spectrum + half + 1);  // the even extension: bin N - i = bin i
		fft_execute(inverse_fft);
		// fold the two-sided cepstrum onto its causal half: weights 1, 2, .., 2, 1, 0, .., 0; the plan's e^{+i} convention leaves the
		// conjugate of what the forward transform below expects
		for (int i = 0; i < fft_size; ++i) {
			const double w = (i == 0 || i == half) ? 1.0 : (i < half ? 2.0 : 0.0);
			cepstrum[i][0] = i <= half ? w * cepstrum[i][0] : 0.0;
			cepstrum[i][1] = i <= half ? -w * cepstrum[i][1] : 0.0;
		}
		fft_execute(forward_fft);
		for (int i = 0; i <= half; ++i) {
			const std::complex<double> z = std::polar(std::exp(minimum_phase_spectrum[i][0] / fft_size), minimum_phase_spectrum[i][1] / fft_size);
			minimum_phase_spectrum[i][0] = z.real();
			minimum_phase_spectrum[i][1] = z.imag();
		}
	}
} MinimumPhaseAnalysis;

// (fast_fftfilt of reference src/world_matlabfunctions.cpp:266-301 is not provided: nothing in the reference calls it)
#endif  // __cplusplus
#endif
