/* world_class_codec.h -- the reference's feature codec on the MI355X, exported by
 * libworldclass_hip.so: mel-cepstral coding of the spectral envelope by DCT-through-FFT and 3 kHz-band coding of the
 * aperiodicity (reference include/codec.hpp:23-90, src/codec.cpp:211-325).
 *
 * The five functions of the reference keep their names, argument meaning and row-pointer tables (include/codec.hpp is the
 * drop-in header); they copy the rows to the device, run the kernels below and copy back.  The *_device variants work on
 * device-resident parameters in the packed layout of world_class_c.h (rows of fft_size/2+1 doubles, coded rows of
 * number_of_dimensions / GetNumberOfAperiodicities(fs) doubles), which is the point of the codec on a GPU: it is the
 * natural epilogue of CheapTrick / D4C and shrinks what has to cross PCIe by 10-40x.
 * No CPU fallback: without a HIP device the host-pointer functions print the error and leave their outputs untouched
 * (they are void in the reference), the device variants return WC_ERR_DEVICE.
 */
#ifndef WORLD_CLASS_CODEC_H
#define WORLD_CLASS_CODEC_H

#ifdef __cplusplus
extern "C" {
#endif

/* reference include/codec.hpp:23, src/codec.cpp:211-214: int(min(15000, fs / 2 - 3000) / 3000) */
int GetNumberOfAperiodicities(int fs);
/* reference include/codec.hpp:38-39, src/codec.cpp:216-236 */
void CodeAperiodicity(const double *const *aperiodicity, int f0_length, int fs, int fft_size, double **coded_aperiodicity);
/* reference include/codec.hpp:53-54, src/codec.cpp:238-267 */
void DecodeAperiodicity(const double *const *coded_aperiodicity, int f0_length, int fs, int fft_size, double **aperiodicity);
/* reference include/codec.hpp:69-71, src/codec.cpp:269-296 */
void CodeSpectralEnvelope(const double *const *spectrogram, int f0_length, int fs, int fft_size, int number_of_dimensions,
						  double **coded_spectral_envelope);
/* reference include/codec.hpp:86-88, src/codec.cpp:298-325 */
void DecodeSpectralEnvelope(const double *const *coded_spectral_envelope, int f0_length, int fs, int fft_size,
							int number_of_dimensions, double **spectrogram);

/* device-resident batches; all return 0 or a negative WC_ERR_* code (wc_last_error() has the message) */
int wc_code_spectral_envelope_device(int fs, int fft_size, long long n_frames, int number_of_dimensions, const double *d_sp,
									 double *d_coded);
int wc_decode_spectral_envelope_device(int fs, int fft_size, long long n_frames, int number_of_dimensions, const double *d_coded,
									   double *d_sp);
int wc_code_aperiodicity_device(int fs, int fft_size, long long n_frames, const double *d_ap, double *d_coded);
int wc_decode_aperiodicity_device(int fs, int fft_size, long long n_frames, const double *d_coded, double *d_ap);

#ifdef __cplusplus
}
#endif
#endif /* WORLD_CLASS_CODEC_H */
