// Data formats either side of the hot path (include/world_class_io.h): the reference's WAV and parameter files with
// the reference's function names and file bytes, device-side PCM conversion, and the demo's parameter modification
// as a kernel.  Restates the behaviour of reference tools/audioio.cpp:116-253, tools/parameterio.cpp:60-244 and
// test/test.cpp:201-243; the parsers below work on the whole file image instead of a FILE* cursor.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/world_class_c.h"
#include "../../include/world_class_io.h"
#include "wc_device.hpp"
#include "wc_internal.hpp"

using namespace wc;

namespace {

// ---- small file helpers ---------------------------------------------------------------------------------------
bool slurp(const char *filename, std::vector<unsigned char> &out) {
	FILE *fp = std::fopen(filename, "rb");
	if (!fp) return false;
	std::fseek(fp, 0, SEEK_END);
	long n = std::ftell(fp);
	std::fseek(fp, 0, SEEK_SET);
	out.resize(n > 0 ? static_cast<size_t>(n) : 0);
	size_t got = out.empty() ? 0 : std::fread(out.data(), 1, out.size(), fp);
	out.resize(got);
	std::fclose(fp);
	return true;
}

void complain(const char *msg) {
	std::fprintf(stderr, "%s\n", msg);
	set_error(msg);
}

struct Cursor {
	const std::vector<unsigned char> &b;
	size_t pos;
	explicit Cursor(const std::vector<unsigned char> &buf) : b(buf), pos(0) {}
	bool has(size_t n) const { return pos + n <= b.size(); }
	// reads past the end yield zeros, like fread leaving a zero-initialised buffer untouched
	unsigned char u8() { return pos < b.size() ? b[pos++] : (++pos, 0); }
	bool tag(const char *t) {
		bool ok = true;
		for (int i = 0; i < 4; ++i) ok = (u8() == static_cast<unsigned char>(t[i])) && ok;
		return ok;
	}
	int le(int bytes) {  // the reference's accumulation v = v * 256 + byte from the top byte down, in int
		unsigned char v[4] = {0, 0, 0, 0};
		for (int i = 0; i < bytes; ++i) v[i] = u8();
		unsigned int r = 0;
		for (int i = bytes - 1; i >= 0; --i) r = r * 256u + v[i];
		return static_cast<int>(r);
	}
	void skip(size_t n) { pos += n; }
	template <class T>
	T raw() {
		T v;
		std::memset(&v, 0, sizeof v);
		if (has(sizeof v)) std::memcpy(&v, b.data() + pos, sizeof v);
		pos += sizeof v;
		return v;
	}
};

// ---- WAV ------------------------------------------------------------------------------------------------------
struct WavInfo {
	int fs = 0, nbit = 0, length = 0;
	size_t data_pos = 0;
};

// reference CheckHeader (tools/audioio.cpp:27-63): RIFF, 4 bytes skipped, WAVE, "fmt " right behind it, chunk size 16,
// PCM, mono.  Returns 0 with the reference's message on a mismatch.
int wav_check_header(Cursor &c) {
	if (!c.tag("RIFF")) { complain("RIFF error."); return 0; }
	c.skip(4);
	if (!c.tag("WAVE")) { complain("WAVE error."); return 0; }
	if (!c.tag("fmt ")) { complain("fmt error."); return 0; }
	const unsigned char s0 = c.u8(), s1 = c.u8(), s2 = c.u8(), s3 = c.u8();
	if (!(s0 == 16 && s1 == 0 && s2 == 0 && s3 == 0)) { complain("fmt (2) error."); return 0; }
	const unsigned char f0 = c.u8(), f1 = c.u8();
	if (!(f0 == 1 && f1 == 0)) { complain("Format ID error."); return 0; }
	const unsigned char c0 = c.u8(), c1 = c.u8();
	if (!(c0 == 1 && c1 == 0)) { complain("This function cannot support stereo file"); return 0; }
	return 1;
}

// first occurrence of "data" at or after the cursor (the reference scans byte by byte, tools/audioio.cpp:77-86)
bool wav_find_data(Cursor &c) {
	while (c.pos < c.b.size()) {
		if (c.b[c.pos] == 'd' && c.pos + 4 <= c.b.size() && std::memcmp(c.b.data() + c.pos, "data", 4) == 0) {
			c.pos += 4;
			return true;
		}
		++c.pos;
	}
	return false;
}

// 1 ok, 0 cannot open, -1 rejected
int wav_parse(const char *filename, std::vector<unsigned char> &buf, WavInfo &w, bool quiet_open) {
	if (!slurp(filename, buf)) {
		if (!quiet_open) complain("File not found.");
		return 0;
	}
	Cursor c(buf);
	if (!wav_check_header(c)) return -1;
	w.fs = c.le(4);
	c.skip(6);  // byte rate, block align
	w.nbit = c.u8();
	c.skip(1);
	if (!wav_find_data(c)) { complain("data error."); return -1; }
	const int bytes = c.le(4);
	// whole-byte PCM of 1 to 4 bytes only: the reference divides by zero below 8 bits and, like any reader that sizes its
	// sample scratch for 32 bits, runs off it above (the header field is one byte: up to 255)
	if (w.nbit != 8 && w.nbit != 16 && w.nbit != 24 && w.nbit != 32) { complain("data error."); return -1; }
	w.length = bytes / (w.nbit / 8);
	w.data_pos = c.pos;
	return 1;
}

// ---- parameter files ------------------------------------------------------------------------------------------
void put_tagged(FILE *fp, const char *tag, double value, int size) {  // reference WriteOneParameter :12-21
	std::fwrite(tag, 1, 4, fp);
	if (size == 4) {
		const int v = static_cast<int>(value);
		std::fwrite(&v, 4, 1, fp);
	} else {
		std::fwrite(&value, 8, 1, fp);
	}
}

void write_matrix(const char *filename, const char *magic, int fs, int f0_length, double frame_period, int fft_size,
				  int number_of_dimensions, const double *const *rows) {
	FILE *fp = std::fopen(filename, "wb");
	if (!fp) { complain("File cannot be opened."); return; }
	std::fwrite(magic, 1, 4, fp);
	put_tagged(fp, "NOF ", f0_length, 4);
	put_tagged(fp, "FP  ", frame_period, 8);
	put_tagged(fp, "FFT ", fft_size, 4);
	put_tagged(fp, "NOD ", number_of_dimensions, 4);
	put_tagged(fp, "FS  ", fs, 4);
	const int nd = number_of_dimensions == 0 ? fft_size / 2 + 1 : number_of_dimensions;
	for (int i = 0; i < f0_length; ++i) std::fwrite(rows[i], 8, nd, fp);
	std::fclose(fp);
}

int read_matrix(const char *filename, const char *magic, double **rows) {
	std::vector<unsigned char> buf;
	if (!slurp(filename, buf)) { complain("File cannot be opened."); return 0; }
	Cursor c(buf);
	if (!c.tag(magic)) { complain("Header error."); return 0; }
	// reference LoadParameters :23-41: fixed field order, tags not checked
	c.skip(4);
	const int frames = c.raw<int>();
	c.skip(12);  // "FP  " + double
	c.skip(4);
	const int fft_size = c.raw<int>();
	c.skip(4);
	int nd = c.raw<int>();
	nd = nd == 0 ? fft_size / 2 + 1 : nd;
	c.skip(8);  // "FS  " + int
	for (int i = 0; i < frames; ++i) {
		const size_t want = static_cast<size_t>(nd) * 8;
		const size_t have = c.pos < buf.size() ? std::min(want, buf.size() - c.pos) : 0;
		if (have) std::memcpy(rows[i], buf.data() + c.pos, have);  // a short file leaves the rest untouched, like fread
		c.pos += want;
	}
	return 1;
}

// wavwrite's sample conversion: static_cast<int>(x * 32767) on the reference's platform (x86 cvttsd2si: NaN and
// out-of-range values become INT_MIN), then clamp to int16
__host__ __device__ inline int pcm16_of(double x) {
	const double v = x * 32767;
	int iv;
	if (!(v > -2147483649.0 && v < 2147483648.0)) iv = INT_MIN;
	else iv = static_cast<int>(v);
	return iv < -32768 ? -32768 : (iv > 32767 ? 32767 : iv);
}

// ---- device kernels -------------------------------------------------------------------------------------------
__global__ void pcm16_to_double_kernel(const int16_t *__restrict__ p, long long n, double *__restrict__ x) {
	for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
		x[i] = static_cast<double>(p[i]) / 32768.0;
}
__global__ void float_to_double_kernel(const float *__restrict__ p, long long n, double *__restrict__ x) {
	for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
		x[i] = static_cast<double>(p[i]);
}
__global__ void double_to_pcm16_kernel(const double *__restrict__ y, long long n, int16_t *__restrict__ p) {
	for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
		p[i] = static_cast<int16_t>(pcm16_of(y[i]));
}
__global__ void scale_f0_kernel(double *__restrict__ f0, long long n, double s) {
	for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) f0[i] *= s;
}

// One workgroup per frame: the row's logarithm goes to LDS, every thread interpolates its bins (reference interp1,
// src/world_matlabfunctions.cpp:157-182, with histc's clamp(#{x[j] <= xi}, 1, n - 1) and linear extrapolation).
constexpr int MOD_MAX_BINS = 4096 / 2 + 1;
__global__ __launch_bounds__(256) void stretch_kernel(double *__restrict__ sp, int fs, int fft_size, double ratio) {
	__shared__ double lg[MOD_MAX_BINS];
	const int bins = fft_size / 2 + 1;
	double *__restrict__ row = sp + (long long)blockIdx.x * bins;
	for (int j = threadIdx.x; j < bins; j += 256) lg[j] = log(row[j]);
	__syncthreads();
	auto axis1 = [&](int j) { return ratio * j / fft_size * fs; };  // reference test/test.cpp:222
	const int cut = static_cast<int>(fft_size / 2.0 * ratio);
	for (int i = threadIdx.x; i < bins; i += 256) {
		const double xi = static_cast<double>(i) / fft_size * fs;
		// c = #{j : axis1(j) <= xi}: start from the real-number estimate and settle with the reference's expressions
		int c = static_cast<int>(i / ratio) + 1;
		c = c < 0 ? 0 : (c > bins ? bins : c);
		while (c < bins && axis1(c) <= xi) ++c;
		while (c > 0 && !(axis1(c - 1) <= xi)) --c;
		const int k = c < 1 ? 1 : (c > bins - 1 ? bins - 1 : c);
		const double x0 = axis1(k - 1), x1 = axis1(k);
		const double s = (xi - x0) / (x1 - x0);
		double v = exp(lg[k - 1] + s * (lg[k] - lg[k - 1]));
		row[i] = v;
	}
	if (ratio < 1.0 && cut >= 1) {  // bins from `cut` upward repeat bin cut - 1 (reference :236-240)
		__syncthreads();
		const double fill = row[cut - 1];
		__syncthreads();
		for (int j = cut + threadIdx.x; j < bins; j += 256) row[j] = fill;
	}
}

}  // namespace

extern "C" {

void wavwrite(const double *x, int x_length, int fs, int nbit, const char *filename) {
	(void)nbit;
	FILE *fp = std::fopen(filename, "wb");
	if (!fp) { complain("File cannot be opened."); return; }
	std::vector<unsigned char> img(44 + 2 * static_cast<size_t>(x_length > 0 ? x_length : 0));
	auto put32 = [&](size_t at, uint32_t v) { for (int i = 0; i < 4; ++i) img[at + i] = (v >> (8 * i)) & 0xff; };
	auto put16 = [&](size_t at, uint32_t v) { img[at] = v & 0xff; img[at + 1] = (v >> 8) & 0xff; };
	std::memcpy(&img[0], "RIFF", 4);
	put32(4, 36u + static_cast<uint32_t>(x_length) * 2u);
	std::memcpy(&img[8], "WAVEfmt ", 8);
	put32(16, 16);
	put16(20, 1);                                   // PCM
	put16(22, 1);                                   // mono
	put32(24, static_cast<uint32_t>(fs));
	put32(28, static_cast<uint32_t>(fs) * 2u);      // bytes per second
	put16(32, 2);                                   // block align
	put16(34, 16);                                  // bits per sample
	std::memcpy(&img[36], "data", 4);
	put32(40, static_cast<uint32_t>(x_length) * 2u);
	for (int i = 0; i < x_length; ++i) put16(44 + 2 * static_cast<size_t>(i), static_cast<uint32_t>(pcm16_of(x[i])) & 0xffffu);
	std::fwrite(img.data(), 1, img.size(), fp);
	std::fclose(fp);
}

int GetAudioLength(const char *filename) {
	std::vector<unsigned char> buf;
	WavInfo w;
	const int rc = wav_parse(filename, buf, w, true);
	return rc == 1 ? w.length : rc;
}

void wavread(const char *filename, int *fs, int *nbit, double *x) {
	std::vector<unsigned char> buf;
	WavInfo w;
	if (wav_parse(filename, buf, w, false) != 1) return;
	*fs = w.fs;
	*nbit = w.nbit;
	const int qb = w.nbit / 8;
	const double zero_line = std::pow(2.0, w.nbit - 1);
	Cursor c(buf);
	c.pos = w.data_pos;
	for (int i = 0; i < w.length; ++i) {
		unsigned char v[4] = {0, 0, 0, 0};
		for (int j = 0; j < qb && j < 4; ++j) v[j] = c.u8();
		double bias = 0.0, tmp = 0.0;
		if (v[qb - 1] >= 128) {
			bias = zero_line;
			v[qb - 1] &= 0x7f;
		}
		for (int j = qb - 1; j >= 0; --j) tmp = tmp * 256.0 + v[j];
		x[i] = (tmp - bias) / zero_line;
	}
}

int wc_wavread_pcm16(const char *filename, int *fs, int16_t *pcm, int capacity) {
	std::vector<unsigned char> buf;
	WavInfo w;
	const int rc = wav_parse(filename, buf, w, true);
	if (rc != 1) return rc;
	if (w.nbit != 16) return -2;
	if (fs) *fs = w.fs;
	const int n = w.length < capacity ? w.length : capacity;
	Cursor c(buf);
	c.pos = w.data_pos;
	for (int i = 0; i < n; ++i) {
		const unsigned lo = c.u8(), hi = c.u8();
		pcm[i] = static_cast<int16_t>(static_cast<uint16_t>(lo | (hi << 8)));
	}
	return n;
}

void WriteF0(const char *filename, int f0_length, double frame_period, const double *temporal_positions, const double *f0,
			 int text_flag) {
	if (text_flag == 1) {
		FILE *fp = std::fopen(filename, "w");
		if (!fp) { complain("File cannot be opened."); return; }
		for (int i = 0; i < f0_length; ++i) std::fprintf(fp, "%.5f %.5f\r\n", temporal_positions[i], f0[i]);
		std::fclose(fp);
		return;
	}
	FILE *fp = std::fopen(filename, "wb");
	if (!fp) { complain("File cannot be opened."); return; }
	std::fwrite("F0  ", 1, 4, fp);
	put_tagged(fp, "NOF ", f0_length, 4);
	put_tagged(fp, "FP  ", frame_period, 8);
	std::fwrite(f0, 8, f0_length, fp);
	std::fclose(fp);
}

int ReadF0(const char *filename, double *temporal_positions, double *f0) {
	std::vector<unsigned char> buf;
	if (!slurp(filename, buf)) { complain("File cannot be opened."); return 0; }
	Cursor c(buf);
	if (!c.tag("F0  ")) { complain("Header error."); return 0; }
	c.skip(4);
	const int frames = c.raw<int>();
	c.skip(4);
	const double frame_period = c.raw<double>();
	for (int i = 0; i < frames; ++i) {
		if (c.has(8)) f0[i] = c.raw<double>();
		else c.skip(8);
	}
	for (int i = 0; i < frames; ++i) temporal_positions[i] = i / 1000.0 * frame_period;
	return 1;
}

double GetHeaderInformation(const char *filename, const char *parameter) {
	std::vector<unsigned char> buf;
	if (!slurp(filename, buf)) { complain("File cannot be opened."); return 0; }
	// the reference walks the first 13 four-byte words and reads the value behind the first word equal to the tag
	Cursor c(buf);
	for (int i = 0; i < 13; ++i) {
		char word[5] = {0, 0, 0, 0, 0};
		for (int j = 0; j < 4; ++j) word[j] = static_cast<char>(c.u8());
		if (std::strcmp(word, parameter) != 0) continue;
		if (std::strcmp(parameter, "FP  ") == 0) return c.raw<double>();
		return static_cast<double>(c.raw<int>());
	}
	return 0;
}

void WriteSpectralEnvelope(const char *filename, int fs, int f0_length, double frame_period, int fft_size,
						   int number_of_dimensions, const double *const *spectrogram) {
	write_matrix(filename, "SPEC", fs, f0_length, frame_period, fft_size, number_of_dimensions, spectrogram);
}
int ReadSpectralEnvelope(const char *filename, double **spectrogram) { return read_matrix(filename, "SPEC", spectrogram); }
void WriteAperiodicity(const char *filename, int fs, int f0_length, double frame_period, int fft_size, int number_of_dimensions,
					   const double *const *aperiodicity) {
	write_matrix(filename, "AP  ", fs, f0_length, frame_period, fft_size, number_of_dimensions, aperiodicity);
}
int ReadAperiodicity(const char *filename, double **aperiodicity) { return read_matrix(filename, "AP  ", aperiodicity); }

int wc_pcm16_to_double_device(const int16_t *d_pcm, long long n, double *d_x) {
	if (n < 0 || (n > 0 && (!d_pcm || !d_x))) return fail(WC_ERR_INVALID, "pcm16_to_double: bad argument");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	if (n == 0) return WC_OK;
	const unsigned blocks = static_cast<unsigned>(std::min<long long>((n + 255) / 256, 65536));
	hipLaunchKernelGGL(pcm16_to_double_kernel, dim3(blocks), dim3(256), 0, dev->active(), d_pcm, n, d_x);
	WC_HIP(hipGetLastError());
	return WC_OK;
}

int wc_float_to_double_device(const float *d_f, long long n, double *d_x) {
	if (n < 0 || (n > 0 && (!d_f || !d_x))) return fail(WC_ERR_INVALID, "float_to_double: bad argument");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	if (n == 0) return WC_OK;
	const unsigned blocks = static_cast<unsigned>(std::min<long long>((n + 255) / 256, 65536));
	hipLaunchKernelGGL(float_to_double_kernel, dim3(blocks), dim3(256), 0, dev->active(), d_f, n, d_x);
	WC_HIP(hipGetLastError());
	return WC_OK;
}

int wc_double_to_pcm16_device(const double *d_y, long long n, int16_t *d_pcm) {
	if (n < 0 || (n > 0 && (!d_pcm || !d_y))) return fail(WC_ERR_INVALID, "double_to_pcm16: bad argument");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	if (n == 0) return WC_OK;
	const unsigned blocks = static_cast<unsigned>(std::min<long long>((n + 255) / 256, 65536));
	hipLaunchKernelGGL(double_to_pcm16_kernel, dim3(blocks), dim3(256), 0, dev->active(), d_y, n, d_pcm);
	WC_HIP(hipGetLastError());
	return WC_OK;
}

int wc_modify_parameters_device(int fs, int fft_size, long long n_frames, double *d_f0, double *d_sp, double f0_scale,
								double spectral_ratio) {
	if (fs <= 0 || fft_size < 2 || fft_size / 2 + 1 > MOD_MAX_BINS || n_frames < 0 || spectral_ratio < 0.0)
		return fail(WC_ERR_INVALID, "modify_parameters: bad argument (fft_size <= 4096, ratio >= 0)");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	if (n_frames == 0) return WC_OK;
	if (d_f0 && f0_scale != 1.0) {
		const unsigned blocks = static_cast<unsigned>(std::min<long long>((n_frames + 255) / 256, 65536));
		hipLaunchKernelGGL(scale_f0_kernel, dim3(blocks), dim3(256), 0, dev->active(), d_f0, n_frames, f0_scale);
	}
	if (d_sp && spectral_ratio != 0.0)
		hipLaunchKernelGGL(stretch_kernel, dim3(static_cast<unsigned>(n_frames)), dim3(256), 0, dev->active(), d_sp, fs, fft_size, spectral_ratio);
	WC_HIP(hipGetLastError());
	return WC_OK;
}

}  // extern "C"
