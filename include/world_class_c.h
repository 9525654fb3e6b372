/* world_class_c.h -- C-ABI of the MI355X-native WORLD hot path (libworldclass_hip.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  The header-only C++
 * classes in include/{harvest,cheaptrick,d4c,synthesis}.hpp keep the reference's class signatures and
 * forward to these entry points; a ctypes/cgo/JNI binding would bind the same symbols.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference repo).
 * All numeric data are IEEE double; indices are int.  Every function returning int returns 0 on
 * success and a negative code on failure; wc_last_error() then describes the failure (thread-local).
 * There is NO CPU fallback: if no HIP device is usable, creation fails with an error.
 *
 * Two families of compute entry points:
 *   *_compute        host pointers, one utterance, exactly the reference's argument meaning
 *                    (row-pointer tables for spectrogram / aperiodicity); H2D/D2H inside the call.
 *   *_compute_device device pointers, a batch of n_utt utterances in packed layout (extension: the
 *                    only way to fill an MI355X).  Packed layout: utterance u's samples start at
 *                    sum(x_length[0..u)) in d_x; its frames start at row sum(f0_length[0..u)) in
 *                    d_tpos / d_f0 / d_sp / d_ap (rows of fft_size/2+1 doubles); its output samples
 *                    start at sum(out_length[0..u)) in d_out.  Length arrays are HOST arrays.
 *
 * RNG: the reference draws its noise from one process-global xorshift128 stream
 * (src/world_matlabfunctions.cpp:243-264) shared by CheapTrick, D4C and Synthesis in call order.
 * Here the stream position is explicit.  Host-pointer calls use and advance the process-global
 * position (wc_rng_get_position / wc_rng_set_position), reproducing the reference's serial order.
 * Device batch calls take an optional host array rng_pos[n_utt] (in: start position of each
 * utterance, out: position after the stage); NULL means "every utterance starts at 0", i.e. each
 * utterance is processed as if by a fresh reference process.
 */
#ifndef WORLD_CLASS_C_H
#define WORLD_CLASS_C_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WC_OK 0
#define WC_ERR_INVALID (-1)   /* bad argument */
#define WC_ERR_DEVICE (-2)    /* HIP runtime / no device */
#define WC_ERR_UNSUPPORTED (-3)

typedef struct wc_harvest wc_harvest;
typedef struct wc_cheaptrick wc_cheaptrick;
typedef struct wc_d4c wc_d4c;
typedef struct wc_synthesis wc_synthesis;

/* ---- library-wide ---------------------------------------------------------------------------- */
const char *wc_last_error(void);
const char *wc_version(void);
int wc_device_count(void);            /* number of HIP devices visible, <0 on error */
int wc_set_device(int device);        /* device used by handles created afterwards (default 0) */
int wc_get_device(void);
/* Stream (a hipStream_t passed as void*) on which the CALLING THREAD's subsequent calls enqueue their
 * work, whichever handle they go through; NULL = back to the library's own non-blocking stream.  Per host
 * thread: other threads, existing handles' buffers and the library's stream are unaffected.  Lets a caller
 * order our kernels with its own work: everything a call launches -- also what the fused pipeline puts on
 * its internal side streams -- is ordered after the work already on that stream; the pcm / codec /
 * modify device entry points only enqueue (stream-ordered), the stage and pipeline calls also wait for their
 * own work before returning (they hand back noise-stream positions and overflow flags).  The stream must
 * outlive the calls made on it. */
int wc_set_stream(void *hip_stream);
/* Blocks until all work the calling thread submitted on its current stream is done. */
int wc_synchronize(void);
/* Several host threads may drive handles on the same device: public calls on one device are serialised
 * (they share the noise draw table and the process-wide noise-stream position). */
/* SHA-256 (hex) of the sources and compiler flags the library was built from; build.py compares it with the tree. */
const char *wc_build_hash(void);
uint64_t wc_rng_get_position(void);
void wc_rng_set_position(uint64_t position);
/* The host-pointer batch calls (wc_*_compute_batch) keep their page-locked staging and its device twin on the device between
 * calls (grown on demand; ~1 GB each per row matrix of a 64 x 10 s batch at 48 kHz).  They are released when the last stage
 * handle on the device is destroyed, and by this call (on the calling thread's device); the next batch call allocates again. */
int wc_release_scratch(void);

/* ---- size helpers (pure host arithmetic) ------------------------------------------------------- */
/* Harvest::getSamples, include/harvest.hpp:41-43, src/harvest.cpp:173-181 */
int wc_get_samples(int fs, int x_length, double frame_period);
/* CheapTrick::getFFTSizeForCheapTrick / getF0FloorForCheapTrick, include/cheaptrick.hpp:35-37,
 * src/cheaptrick.cpp:97-105 */
int wc_cheaptrick_fft_size(int fs, double f0_floor);
double wc_cheaptrick_f0_floor(int fs, int fft_size);
/* y_length of the demo driver, test/test.cpp:362-363 */
int wc_synthesis_out_length(int f0_length, double frame_period, int fs);

/* ---- Harvest: include/harvest.hpp:16-44 -------------------------------------------------------- */
/* HarvestOption fields (include/harvest.hpp:16-28, defaults src/harvest.cpp:52-56: 71, 800, 5, 8000,
 * 40, false).  use_cos_table selects the reference's tabulated refinement window (src/harvest.cpp:152-170, :779-787). */
wc_harvest *wc_harvest_create(int fs, double f0_floor, double f0_ceil, double frame_period,
                              double target_fs, double channels_in_octave, int use_cos_table);
void wc_harvest_destroy(wc_harvest *h);
/* Harvest::compute, include/harvest.hpp:37-39, src/harvest.cpp:183-208 */
int wc_harvest_compute(wc_harvest *h, const double *x, int x_length, double *temporal_positions,
                       double *f0);
int wc_harvest_compute_device(wc_harvest *h, int n_utt, const double *d_x, const int *x_length,
                              double *d_tpos, double *d_f0);
/* frames this handle returns for x_length samples (Harvest::getSamples with the handle's fs and frame period) */
int wc_harvest_get_samples(const wc_harvest *h, int x_length);
/* n_utt utterances held as separate HOST arrays in one call (the reference's calling convention, one compute() per utterance,
 * include/harvest.hpp:37-39, batched: packed through page-locked staging, one trip over PCIe each way, one batch on the GPU).
 * temporal_positions[u], f0[u]: wc_harvest_get_samples(h, x_length[u]) doubles each, the caller's. */
int wc_harvest_compute_batch(wc_harvest *h, int n_utt, const double *const *x, const int *x_length,
                             double *const *temporal_positions, double *const *f0);

/* ---- CheapTrick: include/cheaptrick.hpp:14-38 --------------------------------------------------- */
/* CheapTrickOption{q1,f0_floor,fft_size} (src/cheaptrick.cpp:22-45); fft_size 0 = automatic */
wc_cheaptrick *wc_cheaptrick_create(int fs, double q1, double f0_floor, int fft_size);
void wc_cheaptrick_destroy(wc_cheaptrick *c);
int wc_cheaptrick_get_fft_size(const wc_cheaptrick *c);
/* CheapTrick::compute, include/cheaptrick.hpp:30-33, src/cheaptrick.cpp:48-95 */
int wc_cheaptrick_compute(wc_cheaptrick *c, const double *x, int x_length,
                          const double *temporal_positions, const double *f0, int f0_length,
                          double **spectrogram);
int wc_cheaptrick_compute_device(wc_cheaptrick *c, int n_utt, const double *d_x, const int *x_length,
                                 const double *d_tpos, const double *d_f0, const int *f0_length,
                                 double *d_sp, uint64_t *rng_pos);
/* host arrays of n_utt utterances in one call; spectrogram[u][i]: row of frame i of utterance u (fft_size / 2 + 1 doubles, the
 * caller's, as in include/cheaptrick.hpp:30-33).  rng_pos: per-utterance noise-stream positions in / out as in the device call,
 * NULL = every utterance as in a fresh process. */
int wc_cheaptrick_compute_batch(wc_cheaptrick *c, int n_utt, const double *const *x, const int *x_length,
                                const double *const *temporal_positions, const double *const *f0, const int *f0_length,
                                double *const *const *spectrogram, uint64_t *rng_pos);

/* ---- D4C: include/d4c.hpp:16-36 ---------------------------------------------------------------- */
wc_d4c *wc_d4c_create(int fs, double threshold);
void wc_d4c_destroy(wc_d4c *d);
/* D4C::compute, include/d4c.hpp:30-34, src/d4c.cpp:113-173 */
int wc_d4c_compute(wc_d4c *d, const double *x, int x_length, const double *temporal_positions,
                   const double *f0, int f0_length, int fft_size, double **aperiodicity);
int wc_d4c_compute_device(wc_d4c *d, int n_utt, const double *d_x, const int *x_length,
                          const double *d_tpos, const double *d_f0, const int *f0_length,
                          int fft_size, double *d_ap, uint64_t *rng_pos);
int wc_d4c_compute_batch(wc_d4c *d, int n_utt, const double *const *x, const int *x_length, const double *const *temporal_positions,
                         const double *const *f0, const int *f0_length, int fft_size, double *const *const *aperiodicity,
                         uint64_t *rng_pos);

/* ---- Synthesis: include/synthesis.hpp:29-51 ------------------------------------------------------ */
wc_synthesis *wc_synthesis_create(int fs, int fft_size, double frame_period_ms);
void wc_synthesis_destroy(wc_synthesis *s);
int wc_synthesis_get_fft_size(const wc_synthesis *s);
/* Synthesis::compute, include/synthesis.hpp:45-49, src/synthesis.cpp:77-177 */
int wc_synthesis_compute(wc_synthesis *s, const double *f0, int f0_length,
                         const double *const *spectrogram, const double *const *aperiodicity,
                         int out_length, double *out);
int wc_synthesis_compute_device(wc_synthesis *s, int n_utt, const double *d_f0, const int *f0_length,
                                const double *d_sp, const double *d_ap, const int *out_length,
                                double *d_out, uint64_t *rng_pos);
/* host arrays of n_utt utterances in one call; fft_size: that of the handle (the row length is fft_size / 2 + 1; another value
 * is refused with WC_ERR_INVALID -- the reference's Synthesis holds one fft_size per object too, include/synthesis.hpp:31-33) */
int wc_synthesis_compute_batch(wc_synthesis *s, int n_utt, const double *const *f0, const int *f0_length, int fft_size,
                               const double *const *const *spectrogram, const double *const *const *aperiodicity,
                               const int *out_length, double *const *out, uint64_t *rng_pos);

/* ---- fused pipeline (extension): Harvest -> CheapTrick -> D4C -> Synthesis in the demo's order (reference
 * test/test.cpp:288-384) for a packed batch, everything device resident, the stages overlapped on several HIP
 * streams and the noise-stream positions chained on the device (CheapTrick -> D4C -> Synthesis).  Frames per
 * utterance = wc_get_samples(fs, x_length[u], frame_period), output samples = wc_synthesis_out_length(...).
 * rng_pos: optional in/out host array as for the stage calls. -------------------------------------------- */
typedef struct wc_pipeline wc_pipeline;
wc_pipeline *wc_pipeline_create(int fs, double frame_period, double harvest_f0_floor, double harvest_f0_ceil, double q1,
                                double cheaptrick_f0_floor, int fft_size, double d4c_threshold);
void wc_pipeline_destroy(wc_pipeline *p);
int wc_pipeline_get_fft_size(const wc_pipeline *p);
/* A schedule knob of this handle (development / A-B switches; value NULL or "" = the default).  The WC_PIPELINE_* environment
 * variables of the same names are read once, when the handle is created; a run reads none.  Names: "unchain_below" (seconds),
 * "schedule" ("chains"), "side" ("h" / "c"), "syn_streams", "chain_min", "groups", "tail_after_bp", "chain", "direct", "eager",
 * "host_splits" ("5,8,12"), "force_tie" (test hook).  Not for use beside a run of the same handle. */
int wc_pipeline_set_option(wc_pipeline *p, const char *name, const char *value);
int wc_pipeline_run_device(wc_pipeline *p, int n_utt, const double *d_x, const int *x_length, double *d_tpos, double *d_f0,
                           double *d_sp, double *d_ap, double *d_y, uint64_t *rng_pos);

/* Host batch front-end: n_utt ragged utterances given by host pointers: doubles (x_is_pcm16 = 0), the 16-bit PCM samples of a
 * WAV file (x_is_pcm16 = 1: expanded on the device as sample / 32768, what the reference's wavread returns) or 32-bit floats
 * (x_is_pcm16 = 2: widened on the device, exactly).
 * Inputs are gathered into pinned memory and cross PCIe in one copy; the fused pipeline runs; the requested outputs come back
 * in one packed pinned region and are scattered to the caller's per-utterance buffers: tpos[u], f0[u] (frames doubles),
 * sp[u], ap[u] (frames x (fft_size/2+1) doubles, contiguous), y[u] (out_length doubles, or int16 quantised like the
 * reference's wavwrite when y_is_pcm16 = 1).  Any of the five output tables may be NULL (not downloaded). */
int wc_pipeline_run_batch_host(wc_pipeline *p, int n_utt, const void *const *x, int x_is_pcm16, const int *x_length, double *const *tpos,
                               double *const *f0, double *const *sp, double *const *ap, void *const *y, int y_is_pcm16,
                               uint64_t *rng_pos);
/* The same with the reference's feature codec (include/codec.hpp, src/codec.cpp:211-325; world_class_codec.h) as the epilogue of
 * CheapTrick / D4C: coded_sp[u] receives frames x number_of_dimensions mel-cepstral coefficients, coded_ap[u] frames x
 * GetNumberOfAperiodicities(fs) band aperiodicities -- 65 instead of 2050 doubles per 48 kHz frame cross PCIe.  Any table may
 * be NULL.  (wc_pipeline_run_batch_host: destination rows in page-locked memory are written by the copy engine directly.) */
int wc_pipeline_run_batch_host_coded(wc_pipeline *p, int n_utt, const void *const *x, int x_is_pcm16, const int *x_length, double *const *tpos,
                                     double *const *f0, double *const *coded_sp, int number_of_dimensions, double *const *coded_ap,
                                     void *const *y, int y_is_pcm16, uint64_t *rng_pos);

/* ---- device memory plumbing for callers without their own HIP allocator (tests, C++ demo) ------ */
void *wc_device_malloc(uint64_t bytes);
void wc_device_free(void *p);
int wc_memcpy_h2d(void *dst, const void *src, uint64_t bytes);
int wc_memcpy_d2h(void *dst, const void *src, uint64_t bytes);

/* ---- profiling aid: time of the most recent *_compute_device call's dominant kernel, measured
 * with HIP events on the library's stream (ms); <0 if not available ------------------------------ */
int wc_set_kernel_timing(int enable);
float wc_last_kernel_ms(const char *kernel_name);

#ifdef __cplusplus
}
#endif
#endif /* WORLD_CLASS_C_H */
