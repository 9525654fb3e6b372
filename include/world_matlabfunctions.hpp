// Drop-in for the reference's include/world_matlabfunctions.hpp (:16-147) and the free functions of
// include/world_common.hpp (:74-125): the MATLAB-compatible helpers callers of the reference may use next to the four
// classes (the demo's ParameterModification calls interp1, test/test.cpp:231).  Host functions of libworldclass_hip.so with the
// reference's names, argument meaning and arithmetic; randn() draws from the same process-wide noise stream the stages
// consume (wc_rng_get_position / wc_rng_set_position), as in the reference.
// The FFT buffer structs are in world_common.hpp, the plan API in world_fft.hpp (fast_fftfilt, which nothing in the reference calls, is not provided).
#ifndef WORLD_MATLABFUNCTIONS_HPP
#define WORLD_MATLABFUNCTIONS_HPP

#ifdef __cplusplus
extern "C" {
#endif

/* src/world_matlabfunctions.cpp:129-134 */
void fftshift(const double *x, int x_length, double *y);
/* :136-155; index[i] is 1-based */
void histc(const double *x, int x_length, const double *edges, int edges_length, int *index);
/* :157-182 linear interpolation with histc's segment choice (linear extrapolation outside) */
void interp1(const double *x, const double *y, int x_length, const double *xi, int xi_length, double *yi);
/* :184-210 zero-phase order-3 IIR decimation by r = 2..12; writes x_length / r + 1 ... values (see the reference) */
void decimate(const double *x, int x_length, int r, double *y);
/* :212-214 half away from zero */
int matlab_round(double x);
/* :216-218 */
void diff(const double *x, int x_length, double *y);
/* :220-241 equally spaced abscissa, index truncated toward zero */
void interp1Q(double x, double shift, const double *y, int x_length, const double *xi, int xi_length, double *yi);
/* :243-264 */
double randn(void);
/* :303-313 */
double matlab_std(const double *x, int x_length);

/* include/world_common.hpp:74-125, src/world_common.cpp:56-126 */
int GetSuitableFFTSize(int sample);
void DCCorrection(const double *input, double current_f0, int fs, int fft_size, double *output);
void LinearSmoothing(const double *input, double width, int fs, int fft_size, double *output);
void NuttallWindow(int y_length, double *y);

#ifdef __cplusplus
}
static inline int MyMaxInt(int x, int y) { return x > y ? x : y; }
static inline double MyMaxDouble(double x, double y) { return x > y ? x : y; }
static inline int MyMinInt(int x, int y) { return x < y ? x : y; }
static inline double MyMinDouble(double x, double y) { return x < y ? x : y; }
static inline double GetSafeAperiodicity(double x) { return MyMaxDouble(0.001, MyMinDouble(0.999999999999, x)); }
#endif
#endif
