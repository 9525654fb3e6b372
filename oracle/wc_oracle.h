/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference WORLD hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 * It is the checker, never the product: the product path (world_class_amd/csrc) does not link,
 * include or call anything here.
 *
 * Parity status: PINNED.  Every function here is checked in tests/test_oracle_*.py against
 * (a) the real reference compiled into oracle/_ref (when present) and (b) the golden vectors in
 * tests/golden/ generated from that reference by oracle/gen_golden.py.
 */
#ifndef WC_ORACLE_H
#define WC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* RNG (reference src/world_matlabfunctions.cpp:243-264): explicit stream position */
void wco_rng_reset(void);
void wco_rng_seek(uint64_t position);
uint64_t wco_rng_position(void);
double wco_randn(void);
void wco_randn_fill(int n, double *out);

/* helpers */
int wco_matlab_round(double x);
int wco_suitable_fft_size(int sample);
void wco_interp1(const double *x, const double *y, int n, const double *xi, int m, double *yi);
void wco_interp1Q(double x0, double dx, const double *y, int n, const double *xi, int m, double *yi);
void wco_histc(const double *x, int n, const double *edges, int m, int *index);
void wco_decimate(const double *x, int n, int r, double *y);
void wco_dc_correction(const double *in, double f0, int fs, int fft_size, double *out);
void wco_linear_smoothing(const double *in, double width, int fs, int fft_size, double *out);
void wco_nuttall(int n, double *y);
void wco_fft_r2c(int n, const double *in, double *out /* [n/2+1][2] */);
void wco_fft_c2r(int n, const double *in /* [n/2+1][2] */, double *out);
void wco_fft_c2c(int n, int sign /* 1 fwd e^{+i}, 2 bwd e^{-i} */, const double *in, double *out);
void wco_minimum_phase(int n, const double *log_spectrum /* n/2+1 */, double *out /* [n/2+1][2] */);

/* stages; sp / ap are row-major [f0_length][fft_size/2+1].
 * threads: 0 = serial in reference call order (RNG consumed sequentially);
 *          >0 = OpenMP over the reference's own parallel loops, each frame/pulse seeking to the
 *          stream position the serial order would have given it (deterministic, same numbers). */
void wco_set_threads(int threads);
int wco_get_samples(int fs, int x_length, double frame_period);
/* the other three HarvestOption fields (process-wide; defaults 8000, 40, 0) */
void wco_set_harvest_options(double target_fs, double channels_in_octave, int use_cos_table);
void wco_harvest(const double *x, int x_length, int fs, double f0_floor, double f0_ceil,
                 double frame_period, double *tpos, double *f0);
int wco_cheaptrick_fft_size(int fs, double f0_floor);
double wco_cheaptrick_f0_floor(int fs, int fft_size);
void wco_cheaptrick(const double *x, int x_length, int fs, const double *tpos, const double *f0,
                    int f0_length, double q1, double f0_floor, int fft_size, double *sp);
void wco_d4c(const double *x, int x_length, int fs, const double *tpos, const double *f0,
             int f0_length, int fft_size, double threshold, double *ap);
void wco_synthesis(const double *f0, int f0_length, const double *sp, const double *ap,
                   int fft_size, int fs, double frame_period_ms, int out_length, double *out);

/* pulses produced by the Synthesis time base; *reference_capacity = slots the reference allocates */
int wco_synthesis_pulses(const double *f0, int f0_length, int fft_size, int fs, double frame_period_ms,
                         int out_length, int *reference_capacity);

/* draw-count contract ("RNG draw-count contract") */
uint64_t wco_cheaptrick_draws(int fs, const double *f0, int f0_length, double f0_floor, int fft_size);

/* Harvest intermediates for debugging device kernels (1 ms grid):
 * y[y_length] decimated signal; raw[n_bands][L1] raw candidates; cand/score [L1][max_cand] after
 * refinement+removal; f0_1ms[L1] final 1 ms contour.  Any pointer may be NULL.  Returns L1.
 * dims: {y_length, n_bands, max_candidates, number_of_candidates} */
int wco_harvest_debug(const double *x, int x_length, int fs, double f0_floor, double f0_ceil,
                      int *dims, double *y, double *raw, double *cand, double *score,
                      double *f0_base, double *f0_fixed, double *f0_1ms);
#ifdef __cplusplus
}
#endif
#endif
