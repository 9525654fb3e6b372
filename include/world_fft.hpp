// Drop-in for the reference's include/world_fft.hpp (:20-70): the FFTW-like plan API of its bundled FFT, as host functions
// of libworldclass_hip.so.  Same struct fields, same conventions (reference src/world_fft.cpp:31-167): r2c computes
// X[k] = sum x[n] e^{+2 pi i k n / N} for k = 0..N/2 (imaginary parts of bins 0 and N/2 set to 0); c2r is its unnormalised
// inverse (r2c -> c2r = N x; it ignores the imaginary parts of bins 0 and N/2); c2c FFT_FORWARD is e^{+i}, FFT_BACKWARD e^{-i},
// both unnormalised.  Power-of-two sizes, like the reference.  (The kernels do not use this; they have their own in-LDS FFT.)
#ifndef WORLD_FFT_HPP
#define WORLD_FFT_HPP

#ifdef __cplusplus
extern "C" {
#endif

#define FFT_FORWARD 1
#define FFT_BACKWARD 2
#define FFT_ESTIMATE 3

typedef double fft_complex[2];

typedef struct {
	int n;
	int sign;
	unsigned int flags;
	fft_complex *c_in;
	double *in;
	fft_complex *c_out;
	double *out;
	double *input;  /* scratch owned by the plan */
	int *ip;        /* owned by the plan (kept for layout compatibility) */
	double *w;      /* twiddle table owned by the plan */
} fft_plan;

fft_plan fft_plan_dft_1d(int n, fft_complex *in, fft_complex *out, int sign, unsigned int flags);
fft_plan fft_plan_dft_c2r_1d(int n, fft_complex *in, double *out, unsigned int flags);
fft_plan fft_plan_dft_r2c_1d(int n, double *in, fft_complex *out, unsigned int flags);
void fft_execute(fft_plan p);
void fft_destroy_plan(fft_plan p);

#ifdef __cplusplus
}
#endif
#endif
