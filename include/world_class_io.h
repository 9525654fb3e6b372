/* world_class_io.h -- the data formats on either side of the hot path,
 * exported by the same libworldclass_hip.so as world_class_c.h.
 *
 *   - the reference's WAV reader / writer (reference tools/audioio.hpp:22-47, tools/audioio.cpp:116-253) and its
 *     F0 / spectral-envelope / aperiodicity parameter files (reference tools/parameterio.hpp:24-121,
 *     tools/parameterio.cpp:60-244): SAME function names, argument meaning, file bytes and return values, so a caller
 *     of the reference's tools links against this library unchanged.  Host code only (no GPU needed).
 *     Where the reference prints a message and returns, these do the same (message on stderr, also kept for
 *     wc_last_error()).
 *   - device-side sample conversion, so that PCM travels over PCIe as int16 (4x fewer bytes than double) and is
 *     expanded / quantised on the GPU with exactly wavread's / wavwrite's arithmetic;
 *   - the demo's parameter modification (reference test/test.cpp:201-243: F0 scaling, spectral stretching) as a
 *     device kernel between analysis and synthesis, no host round trip.
 */
#ifndef WORLD_CLASS_IO_H
#define WORLD_CLASS_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference tools/audioio.hpp --------------------------------------------------------------------------- */
/* 16-bit mono PCM; nbit is ignored exactly as in the reference (tools/audioio.cpp:116-165).
 * sample = clamp(int(x * 32767), -32768, 32767), the conversion truncating toward zero. */
void wavwrite(const double *x, int x_length, int fs, int nbit, const char *filename);
/* number of samples; 0 if the file cannot be opened, -1 on a header the reference rejects
 * (tools/audioio.cpp:167-207) */
int GetAudioLength(const char *filename);
/* x[i] = signed little-endian sample / 2^(nbit-1) for nbit = 8, 16, 24 or 32 (tools/audioio.cpp:209-253); outputs
 * are left untouched when the header is rejected */
void wavread(const char *filename, int *fs, int *nbit, double *x);

/* ---- reference tools/parameterio.hpp ------------------------------------------------------------------------ */
void WriteF0(const char *filename, int f0_length, double frame_period, const double *temporal_positions, const double *f0,
			 int text_flag);
int ReadF0(const char *filename, double *temporal_positions, double *f0);
double GetHeaderInformation(const char *filename, const char *parameter);
void WriteSpectralEnvelope(const char *filename, int fs, int f0_length, double frame_period, int fft_size,
						   int number_of_dimensions, const double *const *spectrogram);
int ReadSpectralEnvelope(const char *filename, double **spectrogram);
void WriteAperiodicity(const char *filename, int fs, int f0_length, double frame_period, int fft_size, int number_of_dimensions,
					   const double *const *aperiodicity);
int ReadAperiodicity(const char *filename, double **aperiodicity);

/* ---- extensions (wc_ prefix): raw PCM access and device-side conversion -------------------------------------- */
/* The 16-bit samples of a WAV file as stored (no scaling); returns the number of samples read (<= capacity),
 * 0 / -1 like GetAudioLength, -2 if the file is not 16-bit. */
int wc_wavread_pcm16(const char *filename, int *fs, int16_t *pcm, int capacity);
/* d_x[i] = d_pcm[i] / 32768.0 (wavread's scaling) on the current device / stream; pointers are device pointers */
int wc_pcm16_to_double_device(const int16_t *d_pcm, long long n, double *d_x);
/* d_x[i] = (double)d_f[i]: 32-bit float samples (the other common in-memory format) widened on the device, exactly */
int wc_float_to_double_device(const float *d_f, long long n, double *d_x);
/* d_pcm[i] = wavwrite's quantisation of d_y[i] */
int wc_double_to_pcm16_device(const double *d_y, long long n, int16_t *d_pcm);

/* ---- parameter modification (reference test/test.cpp:201-243) on device-resident parameters ------------------ */
/* f0[i] *= f0_scale for n_frames frames (pass 1.0 to leave it), then, if spectral_ratio != 0, every row of d_sp
 * (n_frames rows of fft_size/2+1 doubles, packed like the batch layout of world_class_c.h) is stretched:
 * log -> interp1 from the axis ratio * k * fs / fft_size onto k * fs / fft_size -> exp, and for ratio < 1 the
 * bins from int(fft_size / 2.0 * ratio) upward repeat the bin just below. */
int wc_modify_parameters_device(int fs, int fft_size, long long n_frames, double *d_f0, double *d_sp, double f0_scale,
								double spectral_ratio);

#ifdef __cplusplus
}
#endif
#endif /* WORLD_CLASS_IO_H */
