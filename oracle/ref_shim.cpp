// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or called from the product path.
//
// C-callable shim over the *real* reference (yukara-ikemiya/world-class), compiled from the
// sources where they lie under /root/reference by oracle/Makefile into oracle/_ref/ (git-ignored).
// No reference source is copied into this repository: this file only #includes the reference's
// public headers through -I/root/reference/include and calls its public classes/functions.
//
// Determinism note: Harvest::fixStep1 (reference src/harvest.cpp:277-291) reads f0_step1[i]
// without ever writing it when f0_base[i]==0 (buffer comes from `new double[]`,
// src/harvest.cpp:621-622).  That is undefined behaviour in the reference; upstream WORLD
// semantics are "0".  To make the oracle deterministic this shim supplies zero-filling
// operator new[] / delete[] for the reference objects linked into this .so (-Wl,-Bsymbolic).
// Nothing else about the reference is altered.
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "harvest.hpp"
#include "cheaptrick.hpp"
#include "d4c.hpp"
#include "synthesis.hpp"
#include "world_common.hpp"
#include "world_matlabfunctions.hpp"
#include "world_fft.hpp"

// (REF_PLAIN_NEW: the build that is only TIMED -- oracle/_ref/libworld_ref_omp.so, bench.py's cpu_baseline -- keeps the
// toolchain's own operator new[]: the reference allocates and plans inside its OpenMP loops, and a calloc per allocation would
// make the baseline slower than the reference really is)
#ifndef REF_PLAIN_NEW
void *operator new[](std::size_t n) {
	void *p = std::calloc(1, n ? n : 1);
	if (!p) throw std::bad_alloc();
	return p;
}
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }
#endif

using namespace world_class;

extern "C" {

int ref_get_samples(int fs, int x_length, double frame_period) {
	return static_cast<int>(1000.0 * x_length / fs / frame_period) + 1;
}

// Harvest (reference include/harvest.hpp:16-44)
void ref_harvest(const double *x, int x_length, int fs, double f0_floor, double f0_ceil,
				 double frame_period, double *tpos, double *f0) {
	HarvestOption opt;
	opt.f0_floor = f0_floor;
	opt.f0_ceil = f0_ceil;
	opt.frame_period = frame_period;
	Harvest h(fs, opt);
	h.compute(x, x_length, tpos, f0);
}

// all six fields of HarvestOption (reference include/harvest.hpp:16-24)
void ref_harvest_opt(const double *x, int x_length, int fs, double f0_floor, double f0_ceil, double frame_period,
					 double target_fs, double channels_in_octave, int use_cos_table, double *tpos, double *f0) {
	HarvestOption opt;
	opt.f0_floor = f0_floor;
	opt.f0_ceil = f0_ceil;
	opt.frame_period = frame_period;
	opt.target_fs = target_fs;
	opt.channels_in_octave = channels_in_octave;
	opt.use_cos_table = use_cos_table != 0;
	Harvest h(fs, opt);
	h.compute(x, x_length, tpos, f0);
}

int ref_cheaptrick_fft_size(int fs, double f0_floor) {
	CheapTrick c(fs);
	return c.getFFTSizeForCheapTrick(fs, f0_floor);
}

double ref_cheaptrick_f0_floor(int fs, int fft_size) {
	CheapTrick c(fs);
	return c.getF0FloorForCheapTrick(fs, fft_size);
}

// CheapTrick (reference include/cheaptrick.hpp:14-38); sp is row-major [f0_length][fft_size/2+1]
void ref_cheaptrick(const double *x, int x_length, int fs, const double *tpos, const double *f0,
					int f0_length, double q1, double f0_floor, int fft_size, double *sp) {
	CheapTrickOption opt;
	opt.q1 = q1;
	opt.f0_floor = f0_floor;
	opt.fft_size = fft_size;
	CheapTrick c(fs, opt);
	int nfft = fft_size ? fft_size : c.getFFTSizeForCheapTrick(fs, f0_floor);
	int bins = nfft / 2 + 1;
	std::vector<double *> rows(f0_length);
	for (int i = 0; i < f0_length; ++i) rows[i] = sp + static_cast<size_t>(i) * bins;
	c.compute(x, x_length, tpos, f0, f0_length, rows.data());
}

// D4C (reference include/d4c.hpp:16-36)
void ref_d4c(const double *x, int x_length, int fs, const double *tpos, const double *f0,
			 int f0_length, int fft_size, double threshold, double *ap) {
	D4COption opt;
	opt.threshold = threshold;
	D4C d(fs, opt);
	int bins = fft_size / 2 + 1;
	std::vector<double *> rows(f0_length);
	for (int i = 0; i < f0_length; ++i) rows[i] = ap + static_cast<size_t>(i) * bins;
	d.compute(x, x_length, tpos, f0, f0_length, fft_size, rows.data());
}

// Synthesis (reference include/synthesis.hpp:29-51)
void ref_synthesis(const double *f0, int f0_length, const double *sp, const double *ap,
				   int fft_size, int fs, double frame_period_ms, int out_length, double *out) {
	Synthesis s(fs, fft_size, frame_period_ms);
	int bins = fft_size / 2 + 1;
	std::vector<const double *> sr(f0_length), ar(f0_length);
	for (int i = 0; i < f0_length; ++i) {
		sr[i] = sp + static_cast<size_t>(i) * bins;
		ar[i] = ap + static_cast<size_t>(i) * bins;
	}
	s.compute(f0, f0_length, sr.data(), ar.data(), out_length, out);
}

// ---- helper-level entry points (reference include/world_matlabfunctions.hpp, world_common.hpp) ----
void ref_randn(int n, double *out) { for (int i = 0; i < n; ++i) out[i] = randn(); }
int ref_matlab_round(double x) { return matlab_round(x); }
int ref_suitable_fft_size(int n) { return GetSuitableFFTSize(n); }
void ref_interp1(const double *x, const double *y, int n, const double *xi, int m, double *yi) {
	interp1(x, y, n, xi, m, yi);
}
void ref_interp1Q(double x0, double dx, const double *y, int n, const double *xi, int m, double *yi) {
	interp1Q(x0, dx, y, n, xi, m, yi);
}
void ref_histc(const double *x, int n, const double *edges, int m, int *index) {
	histc(x, n, edges, m, index);
}
void ref_decimate(const double *x, int n, int r, double *y) { decimate(x, n, r, y); }
void ref_dc_correction(const double *in, double f0, int fs, int fft_size, double *out) {
	DCCorrection(in, f0, fs, fft_size, out);
}
void ref_linear_smoothing(const double *in, double width, int fs, int fft_size, double *out) {
	LinearSmoothing(in, width, fs, fft_size, out);
}
void ref_nuttall(int n, double *y) { NuttallWindow(n, y); }

// r2c: out is [n/2+1][2]
void ref_fft_r2c(int n, const double *in, double *out) {
	ForwardRealFFT f;
	f.initialize(n);
	std::memcpy(f.waveform, in, sizeof(double) * n);
	fft_execute(f.forward_fft);
	std::memcpy(out, f.spectrum, sizeof(double) * 2 * (n / 2 + 1));
	f.destroy();
}
void ref_fft_c2r(int n, const double *in, double *out) {
	InverseRealFFT f;
	f.initialize(n);
	std::memcpy(f.spectrum, in, sizeof(double) * 2 * (n / 2 + 1));
	fft_execute(f.inverse_fft);
	std::memcpy(out, f.waveform, sizeof(double) * n);
	f.destroy();
}
// c2c, sign = FFT_FORWARD(1) / FFT_BACKWARD(2); in/out are [n][2]
void ref_fft_c2c(int n, int sign, const double *in, double *out) {
	fft_complex *a = new fft_complex[n];
	fft_complex *b = new fft_complex[n];
	std::memcpy(a, in, sizeof(double) * 2 * n);
	fft_plan p = fft_plan_dft_1d(n, a, b, sign, FFT_ESTIMATE);
	fft_execute(p);
	std::memcpy(out, b, sizeof(double) * 2 * n);
	fft_destroy_plan(p);
	delete[] a;
	delete[] b;
}
// log_spectrum[0..n/2] in -> minimum-phase spectrum [n/2+1][2] out
void ref_minimum_phase(int n, const double *log_spectrum, double *out) {
	MinimumPhaseAnalysis m;
	m.initialize(n);
	std::memcpy(m.log_spectrum, log_spectrum, sizeof(double) * (n / 2 + 1));
	m.compute();
	std::memcpy(out, m.minimum_phase_spectrum, sizeof(double) * 2 * (n / 2 + 1));
	m.destroy();
}

}  // extern "C"
