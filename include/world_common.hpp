// Drop-in for the reference's include/world_common.hpp (:17-129): the FFT buffer bundles (ForwardRealFFT, InverseRealFFT,
// InverseComplexFFT, MinimumPhaseAnalysis -- same fields, methods restating reference src/world_common.cpp:131-233) on top of
// world_fft.hpp, and the free helpers (declared in world_matlabfunctions.hpp of this repository together with the MATLAB ones).
#ifndef WORLD_COMMON_HPP
#define WORLD_COMMON_HPP

#include <math.h>

#include "world_fft.hpp"
#include "world_matlabfunctions.hpp"

#ifdef __cplusplus
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
	// log_spectrum[0 .. fft_size/2] in, minimum_phase_spectrum[0 .. fft_size/2] out (reference src/world_common.cpp:196-233)
	void compute() {
		for (int i = fft_size / 2 + 1; i < fft_size; ++i) log_spectrum[i] = log_spectrum[fft_size - i];
		fft_execute(inverse_fft);
		cepstrum[0][1] *= -1.0;
		for (int i = 1; i < fft_size / 2; ++i) {
			cepstrum[i][0] *= 2.0;
			cepstrum[i][1] *= -2.0;
		}
		cepstrum[fft_size / 2][1] *= -1.0;
		for (int i = fft_size / 2 + 1; i < fft_size; ++i) cepstrum[i][0] = cepstrum[i][1] = 0.0;
		fft_execute(forward_fft);
		for (int i = 0; i <= fft_size / 2; ++i) {
			const double mag = exp(minimum_phase_spectrum[i][0] / fft_size);
			const double arg = minimum_phase_spectrum[i][1] / fft_size;
			minimum_phase_spectrum[i][0] = mag * cos(arg);
			minimum_phase_spectrum[i][1] = mag * sin(arg);
		}
	}
} MinimumPhaseAnalysis;

// reference src/world_matlabfunctions.cpp:266-301: y = ifft(fft(x / N) .* fft(h / N)), fft_size points
inline void fast_fftfilt(const double *x, int x_length, const double *h, int h_length, int fft_size,
						 const ForwardRealFFT *forward_real_fft, const InverseRealFFT *inverse_real_fft, double *y) {
	fft_complex *x_spectrum = new fft_complex[fft_size];
	for (int i = 0; i < fft_size; ++i) forward_real_fft->waveform[i] = i < x_length ? x[i] / fft_size : 0.0;
	fft_execute(forward_real_fft->forward_fft);
	for (int i = 0; i <= fft_size / 2; ++i) {
		x_spectrum[i][0] = forward_real_fft->spectrum[i][0];
		x_spectrum[i][1] = forward_real_fft->spectrum[i][1];
	}
	for (int i = 0; i < fft_size; ++i) forward_real_fft->waveform[i] = i < h_length ? h[i] / fft_size : 0.0;
	fft_execute(forward_real_fft->forward_fft);
	for (int i = 0; i <= fft_size / 2; ++i) {
		const double hr = forward_real_fft->spectrum[i][0], hi = forward_real_fft->spectrum[i][1];
		inverse_real_fft->spectrum[i][0] = x_spectrum[i][0] * hr - x_spectrum[i][1] * hi;
		inverse_real_fft->spectrum[i][1] = x_spectrum[i][0] * hi + x_spectrum[i][1] * hr;
	}
	fft_execute(inverse_real_fft->inverse_fft);
	for (int i = 0; i < fft_size; ++i) y[i] = inverse_real_fft->waveform[i];
	delete[] x_spectrum;
}
#endif  // __cplusplus
#endif
