// Feature codec (include/world_class_codec.h): restates reference src/codec.cpp:12-325 as four kernels.
//
//   code_sp_kernel<MD>     one workgroup per frame: log -> interp1 onto the mel axis (segment indices and fractions of
//                          the two fixed axes are computed once on the host with the reference's histc / interp1
//                          arithmetic) -> the even/odd reordering of DCTForCodec (:47-61) written straight into the
//                          real-FFT buffer -> r2c of fft_size/2 points in LDS -> weights
//   decode_sp_kernel<MD>   weights -> c2c BACKWARD of fft_size/2 points in LDS (IDCTForCodec :63-85) -> interp1 from the
//                          mel axis in Hz onto the linear axis -> exp
//   code_ap_kernel         thread per (frame, band): 20 log10 and interp1Q at 3 kHz multiples (:216-236)
//   decode_ap_kernel       workgroup per frame: voiced/unvoiced test on the mean (:19-30), interp1 + 10^(v/20) (:32-40)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/world_class_c.h"
#include "../../include/world_class_codec.h"
#include "wc_device.hpp"
#include "wc_internal.hpp"

using namespace wc;

namespace {

constexpr double kM0 = 1127.01048, kF0c = 700.0, kFloorFrequency = 40.0, kCeilFrequency = 20000.0;  // world_constantnumbers.hpp
constexpr double kUpperLimit = 15000.0, kFrequencyInterval = 3000.0, kSafeGuard = 0.000000000001;
constexpr double kPiH = 3.1415926535897932384;

double frequency_to_mel(double f) { return kM0 * std::log(f / kF0c + 1.0); }          // reference :42-44
double mel_to_frequency(double mel) { return kF0c * (std::exp(mel / kM0) - 1.0); }    // reference :46-48

// reference histc (src/world_matlabfunctions.cpp:136-155), 1-based segment index per edge
void histc(const std::vector<double> &x, const std::vector<double> &edges, std::vector<int> &index) {
	const int x_length = static_cast<int>(x.size()), edges_length = static_cast<int>(edges.size());
	index.assign(edges_length, 0);
	int count = 1, i = 0;
	for (; i < edges_length; ++i) {
		index[i] = 1;
		if (edges[i] >= x[0]) break;
	}
	for (; i < edges_length; ++i) {
		if (edges[i] < x[count]) index[i] = count;
		else index[i--] = count++;
		if (count == x_length) break;
	}
	count--;
	for (i++; i < edges_length; ++i) index[i] = count;
}

// k and s of reference interp1 (:157-182) for fixed axes: yi = y[k-1] + s (y[k] - y[k-1])
void interp1_plan(const std::vector<double> &x, const std::vector<double> &xi, std::vector<int> &k, std::vector<double> &s) {
	histc(x, xi, k);
	s.resize(xi.size());
	for (size_t i = 0; i < xi.size(); ++i) s[i] = (xi[i] - x[k[i] - 1]) / (x[k[i]] - x[k[i] - 1]);
}

struct SpPlan {
	int *k;
	double *s;
	double2 *w;
};

template <int MD>
__global__ __launch_bounds__(256) void code_sp_kernel(const double *__restrict__ sp, double *__restrict__ coded, int nd, SpPlan p,
													  const double2 *__restrict__ tw) {
	constexpr int M = MD / 2, T = 256;
	__shared__ double lg[MD + 1];
	__shared__ double2 A[fft_lds_size(M)];
	double *Ar = reinterpret_cast<double *>(A);
	int tid = threadIdx.x;
	const double *__restrict__ row = sp + (long long)blockIdx.x * (MD + 1);
	for (int j = tid; j <= MD; j += T) lg[j] = log(row[j]);
	__syncthreads();
	for (int m = tid; m < MD; m += T) {
		const int k = p.k[m];
		const double v = lg[k - 1] + p.s[m] * (lg[k] - lg[k - 1]);
		const int pos = (m & 1) ? M + (MD - 1 - m) / 2 : m / 2;  // waveform[i] = mel[2i], waveform[i + M] = mel[MD - 2i - 1]
		Ar[pos] = v;
	}
	__syncthreads();
	WC_FRESH(tid);
	fft_lds<M, T, +1>(A, tw, tid);
	r2c_post<M, T>(A, tw, tid);
	const double normalization = sqrt((double)MD);
	double *__restrict__ out = coded + (long long)blockIdx.x * nd;
	for (int i = tid; i < nd; i += T) {
		const double re = (i == M) ? A[0].y : A[i].x;
		const double im = (i == 0 || i == M) ? 0.0 : A[i].y;
		const double2 w = p.w[i];
		out[i] = (re * w.x - im * w.y) / normalization;
	}
}

template <int MD>
__global__ __launch_bounds__(256) void decode_sp_kernel(const double *__restrict__ coded, double *__restrict__ sp, int nd, SpPlan p,
														const double2 *__restrict__ tw) {
	constexpr int T = 256;
	__shared__ double2 A[fft_lds_size(MD)];
	__shared__ double mel[MD + 2];
	int tid = threadIdx.x;
	const double *__restrict__ c = coded + (long long)blockIdx.x * nd;
	const double normalization = sqrt((double)MD);
	for (int i = tid; i < MD; i += T) {
		double2 v = make_double2(0.0, 0.0);
		if (i < nd) {
			const double2 w = p.w[i];
			v = make_double2(c[i] * w.x * normalization, -c[i] * w.y * normalization);
		}
		A[i] = v;
	}
	__syncthreads();
	WC_FRESH(tid);
	fft_lds<MD, T, -1>(A, tw, tid);
	for (int i = tid; i < MD / 2; i += T) {
		mel[1 + 2 * i] = A[i].x;
		mel[2 + 2 * i] = A[MD - i - 1].x;
	}
	__syncthreads();
	if (tid == 0) { mel[0] = mel[1]; mel[MD + 1] = mel[MD]; }
	__syncthreads();
	double *__restrict__ row = sp + (long long)blockIdx.x * (MD + 1);
	for (int j = tid; j <= MD; j += T) {
		const int k = p.k[j];
		const double v = mel[k - 1] + p.s[j] * (mel[k] - mel[k - 1]);
		row[j] = exp(v / MD);
	}
}

__global__ void code_ap_kernel(const double *__restrict__ ap, double *__restrict__ coded, long long n_frames, int n_ap, int fs,
							   int fft_size) {
	const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_frames * n_ap) return;
	const long long f = g / n_ap;
	const int b = (int)(g - f * n_ap);
	const int bins = fft_size / 2 + 1;
	const double *__restrict__ row = ap + f * bins;
	const double delta_x = static_cast<double>(fs) / fft_size;
	const double xi = kFrequencyInterval * (b + 1.0);
	const int base = static_cast<int>((xi - 0) / delta_x);       // interp1Q, reference src/world_matlabfunctions.cpp:220-241
	const double frac = (xi - 0) / delta_x - base;
	const double y0 = 20 * log10(row[base]);
	const double dy = (base == bins - 1) ? 0.0 : 20 * log10(row[base + 1]) - y0;
	coded[g] = y0 + dy * frac;
}

__global__ __launch_bounds__(256) void decode_ap_kernel(const double *__restrict__ coded, double *__restrict__ ap, int n_ap, int fs,
														int fft_size) {
	const int bins = fft_size / 2 + 1;
	const double *__restrict__ c = coded + (long long)blockIdx.x * n_ap;
	double *__restrict__ row = ap + (long long)blockIdx.x * bins;
	double tmp = 0.0;
	for (int i = 0; i < n_ap; ++i) tmp += c[i];
	tmp /= n_ap;
	if (tmp > -0.5) {  // CheckVUV: treated as unvoiced, the initial value stays
		for (int j = threadIdx.x; j < bins; j += 256) row[j] = 1.0 - kSafeGuard;
		return;
	}
	const int na = n_ap + 2;
	auto axis = [&](int q) { return q == na - 1 ? fs / 2.0 : q * kFrequencyInterval; };
	auto val = [&](int q) { return q == 0 ? -60.0 : (q == na - 1 ? -kSafeGuard : c[q - 1]); };
	for (int j = threadIdx.x; j < bins; j += 256) {
		const double f = static_cast<double>(fs) / fft_size * j;
		int k = 1;  // histc: clamp(#{q : axis(q) <= f}, 1, na - 1)
		while (k < na && f >= axis(k)) ++k;
		k = k < na - 1 ? k : na - 1;
		const double x0 = axis(k - 1), x1 = axis(k);
		const double s = (f - x0) / (x1 - x0);
		const double v = val(k - 1) + s * (val(k) - val(k - 1));
		row[j] = pow(10.0, v / 20.0);
	}
}

bool sp_sizes_ok(int fs, int fft_size, int nd, bool coding) {
	if (fs <= 0 || !(fft_size == 512 || fft_size == 1024 || fft_size == 2048 || fft_size == 4096)) return false;
	// the reference reads spectrum[i] for i < number_of_dimensions out of fft_size/4+1 bins when coding and fills
	// fft_size/2 inputs when decoding
	return nd >= 1 && nd <= (coding ? fft_size / 4 + 1 : fft_size / 2);
}

struct ScopedBuf : DevBuf {  // per-call scratch: released on scope exit
	~ScopedBuf() { release(); }
};
struct PlanBufs {
	ScopedBuf k, s, w;
};

int upload_plan(Device *dev, PlanBufs &b, const std::vector<int> &k, const std::vector<double> &s, const std::vector<double2> &w,
				SpPlan &out) {
	int rc;
	if ((rc = b.k.reserve(sizeof(int) * k.size()))) return rc;
	if ((rc = b.s.reserve(sizeof(double) * s.size()))) return rc;
	if ((rc = b.w.reserve(sizeof(double2) * w.size()))) return rc;
	WC_HIP(hipMemcpyAsync(b.k.p, k.data(), sizeof(int) * k.size(), hipMemcpyHostToDevice, dev->active()));
	WC_HIP(hipMemcpyAsync(b.s.p, s.data(), sizeof(double) * s.size(), hipMemcpyHostToDevice, dev->active()));
	WC_HIP(hipMemcpyAsync(b.w.p, w.data(), sizeof(double2) * w.size(), hipMemcpyHostToDevice, dev->active()));
	WC_HIP(hipStreamSynchronize(dev->active()));  // the host vectors go out of scope with the caller
	out.k = b.k.as<int>();
	out.s = b.s.as<double>();
	out.w = b.w.as<double2>();
	return WC_OK;
}

// host-pointer wrappers: rows <-> packed device buffers
int rows_to_device(const double *const *rows, int n, int width, DevBuf &buf, hipStream_t s) {
	int rc;
	if ((rc = buf.reserve(sizeof(double) * (size_t)n * width))) return rc;
	std::vector<double> flat((size_t)n * width);
	for (int i = 0; i < n; ++i) std::memcpy(&flat[(size_t)i * width], rows[i], sizeof(double) * width);
	WC_HIP(hipMemcpyAsync(buf.p, flat.data(), sizeof(double) * flat.size(), hipMemcpyHostToDevice, s));
	WC_HIP(hipStreamSynchronize(s));
	return WC_OK;
}
int device_to_rows(const DevBuf &buf, int n, int width, double **rows, hipStream_t s) {
	std::vector<double> flat((size_t)n * width);
	WC_HIP(hipMemcpyAsync(flat.data(), buf.p, sizeof(double) * flat.size(), hipMemcpyDeviceToHost, s));
	WC_HIP(hipStreamSynchronize(s));
	for (int i = 0; i < n; ++i) std::memcpy(rows[i], &flat[(size_t)i * width], sizeof(double) * width);
	return WC_OK;
}

void report(int rc) {
	if (rc != WC_OK) std::fprintf(stderr, "world_class codec: %s\n", wc_last_error());
}

}  // namespace

extern "C" {

int GetNumberOfAperiodicities(int fs) {
	const double lim = fs / 2.0 - kFrequencyInterval;
	return static_cast<int>((kUpperLimit < lim ? kUpperLimit : lim) / kFrequencyInterval);
}

int wc_code_spectral_envelope_device(int fs, int fft_size, long long n_frames, int nd, const double *d_sp, double *d_coded) {
	if (!sp_sizes_ok(fs, fft_size, nd, true) || n_frames < 0)
		return fail(WC_ERR_INVALID, "code_spectral_envelope: fft_size must be 512..4096 and 1 <= number_of_dimensions <= fft_size/4+1");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	if (n_frames == 0) return WC_OK;
	const int md = fft_size / 2;
	// GetParametersForCoding, reference :125-142
	const double floor_mel = frequency_to_mel(kFloorFrequency);
	const double ceil_mel = frequency_to_mel(fs / 2.0 < kCeilFrequency ? fs / 2.0 : kCeilFrequency);
	std::vector<double> mel_axis(md), freq_axis(md + 1);
	std::vector<double2> w(md);
	for (int i = 0; i < md; ++i) {
		mel_axis[i] = (ceil_mel - floor_mel) * i / md + floor_mel;
		w[i] = make_double2(2.0 * std::cos(i * kPiH / fft_size) / std::sqrt((double)fft_size),
							2.0 * std::sin(i * kPiH / fft_size) / std::sqrt((double)fft_size));
		freq_axis[i] = frequency_to_mel(static_cast<double>(i) * fs / fft_size);
	}
	w[0].x /= std::sqrt(2.0);
	// the reference leaves frequency_axis[fft_size/2] unset (:140-141) and never reaches it: every mel point lies below
	// frequency_axis[fft_size/2 - 1]; the natural value keeps the axis monotone
	freq_axis[md] = frequency_to_mel(static_cast<double>(md) * fs / fft_size);
	std::vector<int> k;
	std::vector<double> s;
	interp1_plan(freq_axis, mel_axis, k, s);
	PlanBufs bufs;
	SpPlan plan;
	int rc;
	if ((rc = upload_plan(dev, bufs, k, s, w, plan))) return rc;
	const dim3 grid((unsigned)n_frames), block(256);
	switch (md) {
		case 256: hipLaunchKernelGGL(code_sp_kernel<256>, grid, block, 0, dev->active(), d_sp, d_coded, nd, plan, dev->twiddle); break;
		case 512: hipLaunchKernelGGL(code_sp_kernel<512>, grid, block, 0, dev->active(), d_sp, d_coded, nd, plan, dev->twiddle); break;
		case 1024: hipLaunchKernelGGL(code_sp_kernel<1024>, grid, block, 0, dev->active(), d_sp, d_coded, nd, plan, dev->twiddle); break;
		default: hipLaunchKernelGGL(code_sp_kernel<2048>, grid, block, 0, dev->active(), d_sp, d_coded, nd, plan, dev->twiddle); break;
	}
	WC_HIP(hipGetLastError());
	WC_HIP(hipStreamSynchronize(dev->active()));  // the plan buffers are freed on return
	return WC_OK;
}

int wc_decode_spectral_envelope_device(int fs, int fft_size, long long n_frames, int nd, const double *d_coded, double *d_sp) {
	if (!sp_sizes_ok(fs, fft_size, nd, false) || n_frames < 0)
		return fail(WC_ERR_INVALID, "decode_spectral_envelope: fft_size must be 512..4096 and 1 <= number_of_dimensions <= fft_size/2");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	if (n_frames == 0) return WC_OK;
	const int md = fft_size / 2;
	// GetParametersForDecoding, reference :144-166
	const double floor_mel = frequency_to_mel(kFloorFrequency);
	const double ceil_mel = frequency_to_mel(fs / 2.0 < kCeilFrequency ? fs / 2.0 : kCeilFrequency);
	std::vector<double2> w(md, make_double2(0.0, 0.0));
	for (int i = 0; i < nd; ++i)
		w[i] = make_double2(std::cos(i * kPiH / fft_size) * std::sqrt((double)fft_size), std::sin(i * kPiH / fft_size) * std::sqrt((double)fft_size));
	w[0].x /= std::sqrt(2.0);
	std::vector<double> mel_axis(md + 2), freq_axis(md + 1);
	for (int i = 0; i < md; ++i) mel_axis[i + 1] = mel_to_frequency((ceil_mel - floor_mel) * i / md + floor_mel);
	mel_axis[0] = 0;
	mel_axis[md + 1] = fs / 2.0;
	for (int i = 0; i < md + 1; ++i) freq_axis[i] = static_cast<double>(i) * fs / fft_size;
	std::vector<int> k;
	std::vector<double> s;
	interp1_plan(mel_axis, freq_axis, k, s);
	PlanBufs bufs;
	SpPlan plan;
	int rc;
	if ((rc = upload_plan(dev, bufs, k, s, w, plan))) return rc;
	const dim3 grid((unsigned)n_frames), block(256);
	switch (md) {
		case 256: hipLaunchKernelGGL(decode_sp_kernel<256>, grid, block, 0, dev->active(), d_coded, d_sp, nd, plan, dev->twiddle); break;
		case 512: hipLaunchKernelGGL(decode_sp_kernel<512>, grid, block, 0, dev->active(), d_coded, d_sp, nd, plan, dev->twiddle); break;
		case 1024: hipLaunchKernelGGL(decode_sp_kernel<1024>, grid, block, 0, dev->active(), d_coded, d_sp, nd, plan, dev->twiddle); break;
		default: hipLaunchKernelGGL(decode_sp_kernel<2048>, grid, block, 0, dev->active(), d_coded, d_sp, nd, plan, dev->twiddle); break;
	}
	WC_HIP(hipGetLastError());
	WC_HIP(hipStreamSynchronize(dev->active()));
	return WC_OK;
}

int wc_code_aperiodicity_device(int fs, int fft_size, long long n_frames, const double *d_ap, double *d_coded) {
	const int n_ap = GetNumberOfAperiodicities(fs);
	if (fs <= 0 || fft_size < 2 || n_frames < 0 || n_ap < 1) return fail(WC_ERR_INVALID, "code_aperiodicity: bad argument (fs must exceed 12 kHz)");
	if (kFrequencyInterval * n_ap / (static_cast<double>(fs) / fft_size) >= fft_size / 2 + 1) return fail(WC_ERR_INVALID, "code_aperiodicity: fft_size too small");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	if (n_frames == 0) return WC_OK;
	const long long total = n_frames * n_ap;
	hipLaunchKernelGGL(code_ap_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, dev->active(), d_ap, d_coded, n_frames, n_ap, fs, fft_size);
	WC_HIP(hipGetLastError());
	return WC_OK;
}

int wc_decode_aperiodicity_device(int fs, int fft_size, long long n_frames, const double *d_coded, double *d_ap) {
	const int n_ap = GetNumberOfAperiodicities(fs);
	if (fs <= 0 || fft_size < 2 || n_frames < 0 || n_ap < 1) return fail(WC_ERR_INVALID, "decode_aperiodicity: bad argument (fs must exceed 12 kHz)");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	if (n_frames == 0) return WC_OK;
	hipLaunchKernelGGL(decode_ap_kernel, dim3((unsigned)n_frames), dim3(256), 0, dev->active(), d_coded, d_ap, n_ap, fs, fft_size);
	WC_HIP(hipGetLastError());
	return WC_OK;
}

void CodeSpectralEnvelope(const double *const *spectrogram, int f0_length, int fs, int fft_size, int number_of_dimensions,
						  double **coded_spectral_envelope) {
	Device *dev = current_device();
	if (!dev) { report(WC_ERR_DEVICE); return; }
	if (f0_length <= 0) return;
	ScopedBuf in, out;
	int rc = rows_to_device(spectrogram, f0_length, fft_size / 2 + 1, in, dev->active());
	if (!rc) rc = out.reserve(sizeof(double) * (size_t)f0_length * number_of_dimensions);
	if (!rc) rc = wc_code_spectral_envelope_device(fs, fft_size, f0_length, number_of_dimensions, in.as<double>(), out.as<double>());
	if (!rc) rc = device_to_rows(out, f0_length, number_of_dimensions, coded_spectral_envelope, dev->active());
	report(rc);
}

void DecodeSpectralEnvelope(const double *const *coded_spectral_envelope, int f0_length, int fs, int fft_size,
							int number_of_dimensions, double **spectrogram) {
	Device *dev = current_device();
	if (!dev) { report(WC_ERR_DEVICE); return; }
	if (f0_length <= 0) return;
	ScopedBuf in, out;
	int rc = rows_to_device(coded_spectral_envelope, f0_length, number_of_dimensions, in, dev->active());
	if (!rc) rc = out.reserve(sizeof(double) * (size_t)f0_length * (fft_size / 2 + 1));
	if (!rc) rc = wc_decode_spectral_envelope_device(fs, fft_size, f0_length, number_of_dimensions, in.as<double>(), out.as<double>());
	if (!rc) rc = device_to_rows(out, f0_length, fft_size / 2 + 1, spectrogram, dev->active());
	report(rc);
}

void CodeAperiodicity(const double *const *aperiodicity, int f0_length, int fs, int fft_size, double **coded_aperiodicity) {
	Device *dev = current_device();
	if (!dev) { report(WC_ERR_DEVICE); return; }
	if (f0_length <= 0) return;
	const int n_ap = GetNumberOfAperiodicities(fs);
	ScopedBuf in, out;
	int rc = rows_to_device(aperiodicity, f0_length, fft_size / 2 + 1, in, dev->active());
	if (!rc) rc = out.reserve(sizeof(double) * (size_t)f0_length * (n_ap > 0 ? n_ap : 1));
	if (!rc) rc = wc_code_aperiodicity_device(fs, fft_size, f0_length, in.as<double>(), out.as<double>());
	if (!rc) rc = device_to_rows(out, f0_length, n_ap, coded_aperiodicity, dev->active());
	report(rc);
}

void DecodeAperiodicity(const double *const *coded_aperiodicity, int f0_length, int fs, int fft_size, double **aperiodicity) {
	Device *dev = current_device();
	if (!dev) { report(WC_ERR_DEVICE); return; }
	if (f0_length <= 0) return;
	const int n_ap = GetNumberOfAperiodicities(fs);
	ScopedBuf in, out;
	int rc = n_ap >= 1 ? rows_to_device(coded_aperiodicity, f0_length, n_ap, in, dev->active()) : fail(WC_ERR_INVALID, "decode_aperiodicity: fs must exceed 12 kHz");
	if (!rc) rc = out.reserve(sizeof(double) * (size_t)f0_length * (fft_size / 2 + 1));
	if (!rc) rc = wc_decode_aperiodicity_device(fs, fft_size, f0_length, in.as<double>(), out.as<double>());
	if (!rc) rc = device_to_rows(out, f0_length, fft_size / 2 + 1, aperiodicity, dev->active());
	report(rc);
}

}  // extern "C"
