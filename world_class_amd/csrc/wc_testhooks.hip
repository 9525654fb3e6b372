// Development / test hooks (not part of include/world_class_c.h): the device building blocks of wc_device.hpp run on
// their own, so that -m gpu tests can pin them directly -- the in-LDS FFT against the reference's transform conventions
// (reference src/world_fft.cpp:31-167; goldens fft/* made by the real reference), the order-faithful cumulative sum
// against a sequential loop.  One workgroup per transform / sequence; nothing here is on the product path.
#include <vector>

#include "wc_device.hpp"
#include "wc_internal.hpp"
#include "wc_wavefft.hpp"

namespace wc {

// kind 0: r2c, in N doubles, out (N/2+1) complex.  kind 1: c2r, in (N/2+1) complex, out N doubles (imaginary parts of
// bins 0 and N/2 ignored, unnormalised).  One transform per block.
// (the direction is a template parameter: the FFT launders its table pointer through a scalar register per call, which the
// compiler cannot do under a run-time branch)
template <int N, int T, int KIND>
__global__ __launch_bounds__(T) void hook_real_fft_kernel(const double *__restrict__ in, double *__restrict__ out,
														  const double2 *__restrict__ tw) {
	constexpr int M = N / 2;
	__shared__ double2 A[fft_lds_size(M)];
	double *Ar = reinterpret_cast<double *>(A);
	const int tid = threadIdx.x;
	if constexpr (KIND == 0) {
		const double *x = in + (size_t)blockIdx.x * N;
		double2 *X = reinterpret_cast<double2 *>(out) + (size_t)blockIdx.x * (M + 1);
		for (int i = tid; i < N; i += T) Ar[i] = x[i];
		__syncthreads();
		fft_lds<M, T, +1>(A, tw, tid);
		r2c_post<M, T>(A, tw, tid);
		for (int k = tid; k <= M; k += T) X[k] = k == 0 ? make_double2(A[0].x, 0.0) : k == M ? make_double2(A[0].y, 0.0) : A[k];
	} else {
		const double2 *Y = reinterpret_cast<const double2 *>(in) + (size_t)blockIdx.x * (M + 1);
		double *y = out + (size_t)blockIdx.x * N;
		for (int k = tid; k < M; k += T) A[k] = k == 0 ? make_double2(Y[0].x, Y[M].x) : Y[k];
		__syncthreads();
		c2r_pre<M, T>(A, tw, tid);
		fft_lds<M, T, -1>(A, tw, tid);
		for (int i = tid; i < N; i += T) y[i] = Ar[i];
	}
}

// complex transform of N points, sign +1 (the reference's FFT_FORWARD, e^{+i}) or -1 (FFT_BACKWARD)
template <int N, int T, int SIGN>
__global__ __launch_bounds__(T) void hook_complex_fft_kernel(const double2 *__restrict__ in, double2 *__restrict__ out,
															 const double2 *__restrict__ tw) {
	__shared__ double2 A[fft_lds_size(N)];
	const int tid = threadIdx.x;
	const double2 *x = in + (size_t)blockIdx.x * N;
	double2 *X = out + (size_t)blockIdx.x * N;
	for (int i = tid; i < N; i += T) A[i] = x[i];
	__syncthreads();
	fft_lds<N, T, SIGN>(A, tw, tid);
	for (int i = tid; i < N; i += T) X[i] = A[i];
}

template <int T>
__global__ __launch_bounds__(T) void hook_cumsum_kernel(const double *__restrict__ v, int n, double *__restrict__ out) {
	__shared__ double S[4096];
	__shared__ double scr[T + 2 * (T / 64)];
	__shared__ double red[T / 64 + 2];
	const int tid = threadIdx.x;
	const double *src = v + (size_t)blockIdx.x * n;
	for (int i = tid; i < n; i += T) S[i] = src[i];
	__syncthreads();
	seq_cumsum_nonneg<T>(S, n, scr, red, tid);
	for (int i = tid; i < n; i += T) out[(size_t)blockIdx.x * n + i] = S[i];
}

// seq_cumsum_nonneg_wave: one wavefront per sequence of n <= 2304 terms
template <bool SIGNED>
__global__ __launch_bounds__(64) void hook_cumsum_wave_kernel(const double *__restrict__ v, int n, double *__restrict__ out) {
	__shared__ double S[2304];
	const int lane = threadIdx.x;
	const double *src = v + (size_t)blockIdx.x * n;
	for (int i = lane; i < n; i += 64) S[i] = src[i];
	__syncthreads();
	if (SIGNED) seq_cumsum_signed_wave<36>(S, n, lane);
	else if (n <= 1152) seq_cumsum_nonneg_wave<18>(S, n, lane);
	else seq_cumsum_nonneg_wave<36>(S, n, lane);
	for (int i = lane; i < n; i += 64) out[(size_t)blockIdx.x * n + i] = S[i];
}

// The one-wavefront transforms of wc_wavefft.hpp: a 2048-point real transform per 64-thread workgroup.
// KIND 0: r2c (in 2048 doubles, out 1025 complex); 1: c2r (in 1025 complex, out 2048 doubles, unnormalised);
// 2: r2c with the input zero beyond its first quarter, through the pruned leading stage
template <int KIND>
__global__ __launch_bounds__(64) void hook_wave_fft_kernel(const double *__restrict__ in, double *__restrict__ out,
															const double2 *__restrict__ tw) {
	__shared__ __attribute__((aligned(16))) double L[kWfLds];
	const int lane = threadIdx.x;
	double re[16], im[16];
	if constexpr (KIND != 1) {
		const double *x = in + (size_t)blockIdx.x * 2048;
		double2 *X = reinterpret_cast<double2 *>(out) + (size_t)blockIdx.x * 1025;
#pragma unroll
		for (int q = 0; q < 16; ++q) {
			re[q] = x[2 * (lane + 64 * q)];
			im[q] = x[2 * (lane + 64 * q) + 1];
		}
		if constexpr (KIND == 2) wdft16<+1, 1>(re, im);
		else wdft16<+1>(re, im);
		wf_fft1024_dit_rest<+1>(re, im, L, tw, lane);
		double nyq;
		wf_r2c_unpack(re, im, nyq, tw, lane);
#pragma unroll
		for (int g = 0; g < 4; ++g)
#pragma unroll
			for (int q = 0; q < 4; ++q) X[wf_bin(lane, g, q)] = make_double2(0.5 * re[4 * g + q], 0.5 * im[4 * g + q]);
		if (lane == 0) X[1024] = make_double2(0.5 * nyq, 0.0);
	} else {
		const double2 *Y = reinterpret_cast<const double2 *>(in) + (size_t)blockIdx.x * 1025;
		double *y = out + (size_t)blockIdx.x * 2048;
#pragma unroll
		for (int g = 0; g < 4; ++g)
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const double2 v = Y[wf_bin(lane, g, q)];
				re[4 * g + q] = v.x;
				im[4 * g + q] = v.y;
			}
		const double nyq = Y[1024].x;
		wf_c2r_pack(re, im, nyq, tw, lane);
		wf_fft1024_dif<-1>(re, im, L, tw, lane);
#pragma unroll
		for (int q = 0; q < 16; ++q) {
			y[2 * (lane + 64 * q)] = re[q];
			y[2 * (lane + 64 * q) + 1] = im[q];
		}
	}
}
// The same at eight points per lane: a 1024-point real transform per 64-thread workgroup (wf8_*).
// KIND 0: r2c (in 1024 doubles, out 513 complex); 1: c2r (in 513 complex, out 1024 doubles, unnormalised); 2 / 3: r2c with
// the input zero beyond its first quarter / half, through the pruned leading stage
template <int KIND>
__global__ __launch_bounds__(64) void hook_wave8_fft_kernel(const double *__restrict__ in, double *__restrict__ out,
															 const double2 *__restrict__ tw) {
	__shared__ __attribute__((aligned(16))) double L[kWf8Lds];
	const int lane = threadIdx.x;
	double re[8], im[8];
	if constexpr (KIND != 1) {
		const double *x = in + (size_t)blockIdx.x * 1024;
		double2 *X = reinterpret_cast<double2 *>(out) + (size_t)blockIdx.x * 513;
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			re[q] = x[2 * (lane + 64 * q)];
			im[q] = x[2 * (lane + 64 * q) + 1];
		}
		if constexpr (KIND == 2) wdft8p<+1, 1>(re, im);
		else if constexpr (KIND == 3) wdft8p<+1, 2>(re, im);
		else wdft8p<+1, 4>(re, im);
		wf8_fft512_dit_rest<+1>(re, im, L, tw, lane);
		double nyq;
		wf8_r2c_unpack(re, im, nyq, tw, lane);
#pragma unroll
		for (int g = 0; g < 2; ++g)
#pragma unroll
			for (int q = 0; q < 4; ++q) X[wf8_bin(lane, g, q)] = make_double2(0.5 * re[4 * g + q], 0.5 * im[4 * g + q]);
		if (lane == 0) X[512] = make_double2(0.5 * nyq, 0.0);
	} else {
		const double2 *Y = reinterpret_cast<const double2 *>(in) + (size_t)blockIdx.x * 513;
		double *y = out + (size_t)blockIdx.x * 1024;
#pragma unroll
		for (int g = 0; g < 2; ++g)
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const double2 v = Y[wf8_bin(lane, g, q)];
				re[4 * g + q] = v.x;
				im[4 * g + q] = v.y;
			}
		const double nyq = Y[512].x;
		wf8_c2r_pack(re, im, nyq, tw, lane);
		wf8_fft512_dif<-1>(re, im, L, tw, lane);
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			y[2 * (lane + 64 * q)] = re[q];
			y[2 * (lane + 64 * q) + 1] = im[q];
		}
	}
}
// the real even transform of 2048 points (wf_even2048): in x[0 .. 1024], out F[0 .. 1024]
__global__ __launch_bounds__(64) void hook_even2048_kernel(const double *__restrict__ in, double *__restrict__ out, const double2 *__restrict__ tw) {
	__shared__ __attribute__((aligned(16))) double L[kWfLds];
	__shared__ __attribute__((aligned(16))) double X[1026];
	const int lane = threadIdx.x;
	const double *x = in + (size_t)blockIdx.x * 1025;
	double *F = out + (size_t)blockIdx.x * 1025;
	for (int i = lane; i <= 1024; i += 64) X[i] = x[i];
	wf_fence();
	double lo[8], hi[8], mid;
	wf_even2048(X, L, tw, lane, lo, hi, mid);
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		F[lane + 64 * j] = lo[j];
		F[1024 - lane - 64 * j] = hi[j];
	}
	if (lane == 0) F[512] = mid;
}
// 4096-point real transform by two wavefronts (one 128-thread workgroup): in 4096 doubles, out 2049 complex
__global__ __launch_bounds__(128) void hook_wave2_r2c_kernel(const double *__restrict__ in, double *__restrict__ out,
															 const double2 *__restrict__ tw) {
	__shared__ double L[kWf2Lds];
	const int t = threadIdx.x;
	const double *x = in + (size_t)blockIdx.x * 4096;
	double2 *X = reinterpret_cast<double2 *>(out) + (size_t)blockIdx.x * 2049;
	double re[16], im[16], nyq;
#pragma unroll
	for (int q = 0; q < 16; ++q) {
		re[q] = x[2 * (t + 128 * q)];
		im[q] = x[2 * (t + 128 * q) + 1];
	}
	wf2_fft2048_dit<+1>(re, im, L, tw, t);
	wf2_r2c_unpack(re, im, nyq, tw, t);
#pragma unroll
	for (int g = 0; g < 2; ++g)
#pragma unroll
		for (int q = 0; q < 8; ++q) X[wf2_bin(t, g, q)] = make_double2(0.5 * re[8 * g + q], 0.5 * im[8 * g + q]);
	if (t == 0) X[2048] = make_double2(0.5 * nyq, 0.0);
}
// streaming read of n doubles, 8 bytes per lane and load (WIDE = false) or 16 (WIDE = true): the known byte count the
// FETCH_SIZE counter is calibrated against (tools/fetch_calibrate.py, MI355X_MICROARCH.md HBM section)
template <bool WIDE>
__global__ __launch_bounds__(256) void hook_stream_read_kernel(const double *__restrict__ in, long long n, double *__restrict__ out) {
	double acc = 0.0;
	const long long stride = (long long)gridDim.x * blockDim.x;
	if (WIDE) {
		const double2 *p = reinterpret_cast<const double2 *>(in);
		for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n / 2; i += stride) { const double2 v = p[i]; acc += v.x + v.y; }
	} else {
		for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += in[i];
	}
	if (acc == 123.456) out[0] = acc;  // (keeps the loads)
}
// kind 0: wf_log, 1: wf_exp
__global__ void hook_logexp_kernel(int kind, const double *__restrict__ in, double *__restrict__ out, long long n,
								   const double2 *__restrict__ tw) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = kind == 0 ? wf_log(in[i], tw) : wf_exp(in[i], tw);
}

}  // namespace wc

using namespace wc;

namespace {
struct Scoped {
	void *p = nullptr;
	~Scoped() { if (p) (void)hipFree(p); }
};
template <int N>
void launch_real(int kind, int batch, const double *in, double *out, const double2 *tw, hipStream_t s) {
	if (kind == 0) hipLaunchKernelGGL((hook_real_fft_kernel<N, 256, 0>), dim3(batch), dim3(256), 0, s, in, out, tw);
	else hipLaunchKernelGGL((hook_real_fft_kernel<N, 256, 1>), dim3(batch), dim3(256), 0, s, in, out, tw);
}
template <int N>
void launch_complex(int sign, int batch, const double2 *in, double2 *out, const double2 *tw, hipStream_t s) {
	if (sign > 0) hipLaunchKernelGGL((hook_complex_fft_kernel<N, 256, +1>), dim3(batch), dim3(256), 0, s, in, out, tw);
	else hipLaunchKernelGGL((hook_complex_fft_kernel<N, 256, -1>), dim3(batch), dim3(256), 0, s, in, out, tw);
}
}  // namespace

__global__ void hook_div_const_kernel(const double *__restrict__ in, double *__restrict__ out, long long n, double c, double rc) {
	const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
	if (i < n) out[i] = div_const(in[i], c, rc);
}

extern "C" {

// kind 0 r2c, 1 c2r, 2 c2c forward (e^{+i}), 3 c2c backward; host pointers; `batch` transforms back to back.
// doubles in: N / N+2 / 2N / 2N per transform, doubles out: N+2 / N / 2N / 2N.
int wc_debug_fft(int kind, int n, int batch, const double *in, double *out) {
	if (kind < 0 || kind > 3 || batch <= 0 || !in || !out) return fail(WC_ERR_INVALID, "debug fft: bad argument");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	hipStream_t s = dev->active();
	const size_t n_in = (kind == 0 ? n : kind == 1 ? n + 2 : 2 * n) * (size_t)batch;
	const size_t n_out = (kind == 0 ? n + 2 : kind == 1 ? n : 2 * n) * (size_t)batch;
	Scoped d_in, d_out;
	WC_HIP(hipMalloc(&d_in.p, sizeof(double) * n_in));
	WC_HIP(hipMalloc(&d_out.p, sizeof(double) * n_out));
	WC_HIP(hipMemcpyAsync(d_in.p, in, sizeof(double) * n_in, hipMemcpyHostToDevice, s));
	const double *di = static_cast<const double *>(d_in.p);
	double *dout = static_cast<double *>(d_out.p);
	if (kind < 2) {
		switch (n) {
			case 128: launch_real<128>(kind, batch, di, dout, dev->twiddle, s); break;
			case 256: launch_real<256>(kind, batch, di, dout, dev->twiddle, s); break;
			case 512: launch_real<512>(kind, batch, di, dout, dev->twiddle, s); break;
			case 1024: launch_real<1024>(kind, batch, di, dout, dev->twiddle, s); break;
			case 2048: launch_real<2048>(kind, batch, di, dout, dev->twiddle, s); break;
			case 4096: launch_real<4096>(kind, batch, di, dout, dev->twiddle, s); break;
			case 8192: launch_real<8192>(kind, batch, di, dout, dev->twiddle, s); break;
			default: return fail(WC_ERR_UNSUPPORTED, "debug fft: real sizes 128 .. 8192");
		}
	} else {
		const int sign = kind == 2 ? +1 : -1;
		const double2 *ci = reinterpret_cast<const double2 *>(di);
		double2 *co = reinterpret_cast<double2 *>(dout);
		switch (n) {
			case 64: launch_complex<64>(sign, batch, ci, co, dev->twiddle, s); break;
			case 128: launch_complex<128>(sign, batch, ci, co, dev->twiddle, s); break;
			case 256: launch_complex<256>(sign, batch, ci, co, dev->twiddle, s); break;
			case 512: launch_complex<512>(sign, batch, ci, co, dev->twiddle, s); break;
			case 1024: launch_complex<1024>(sign, batch, ci, co, dev->twiddle, s); break;
			case 2048: launch_complex<2048>(sign, batch, ci, co, dev->twiddle, s); break;
			case 4096: launch_complex<4096>(sign, batch, ci, co, dev->twiddle, s); break;
			default: return fail(WC_ERR_UNSUPPORTED, "debug fft: complex sizes 64 .. 4096");
		}
	}
	WC_HIP(hipGetLastError());
	WC_HIP(hipMemcpyAsync(out, d_out.p, sizeof(double) * n_out, hipMemcpyDeviceToHost, s));
	WC_HIP(hipStreamSynchronize(s));
	return WC_OK;
}

// `batch` sequences of n (<= 4096) non-negative terms each: out = their cumulative sums as seq_cumsum_nonneg forms them
// (threads: 256 or 512, the two block sizes the stages use; 64: the one-wavefront form; -64: its form for signed terms)
int wc_debug_seq_cumsum(const double *v, int n, int batch, int threads, double *out) {
	if (!v || !out || n <= 0 || n > 4096 || batch <= 0 || (threads != 64 && threads != -64 && threads != 256 && threads != 512) || ((threads == 64 || threads == -64) && n > 2304)) return fail(WC_ERR_INVALID, "debug cumsum: bad argument");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	hipStream_t s = dev->active();
	Scoped d_in, d_out;
	const size_t total = (size_t)n * batch;
	WC_HIP(hipMalloc(&d_in.p, sizeof(double) * total));
	WC_HIP(hipMalloc(&d_out.p, sizeof(double) * total));
	WC_HIP(hipMemcpyAsync(d_in.p, v, sizeof(double) * total, hipMemcpyHostToDevice, s));
	if (threads == -64) hipLaunchKernelGGL(hook_cumsum_wave_kernel<true>, dim3(batch), dim3(64), 0, s, static_cast<const double *>(d_in.p), n, static_cast<double *>(d_out.p));
	else if (threads == 64) hipLaunchKernelGGL(hook_cumsum_wave_kernel<false>, dim3(batch), dim3(64), 0, s, static_cast<const double *>(d_in.p), n, static_cast<double *>(d_out.p));
	else if (threads == 256) hipLaunchKernelGGL(hook_cumsum_kernel<256>, dim3(batch), dim3(256), 0, s, static_cast<const double *>(d_in.p), n, static_cast<double *>(d_out.p));
	else hipLaunchKernelGGL(hook_cumsum_kernel<512>, dim3(batch), dim3(512), 0, s, static_cast<const double *>(d_in.p), n, static_cast<double *>(d_out.p));
	WC_HIP(hipGetLastError());
	WC_HIP(hipMemcpyAsync(out, d_out.p, sizeof(double) * total, hipMemcpyDeviceToHost, s));
	WC_HIP(hipStreamSynchronize(s));
	return WC_OK;
}

// 2048-point real transforms by one wavefront each (wc_wavefft.hpp).  kind 0 r2c, 1 c2r, 2 r2c of an input whose last
// three quarters are zero (pruned leading stage); host pointers; doubles in 2048 / 2050 / 2048, out 2050 / 2048 / 2050.
int wc_debug_wave_fft(int kind, int batch, const double *in, double *out) {
	if (kind < 0 || kind > 8 || batch <= 0 || !in || !out) return fail(WC_ERR_INVALID, "debug wave fft: bad argument");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	hipStream_t s = dev->active();
	// (kind 3: the 4096-point r2c by two wavefronts, 4096 doubles in, 4098 out; kinds 4 .. 7: the 1024-point transforms at eight
	// points per lane -- r2c, c2r, r2c pruned to a quarter / a half: 1024 / 1026 / 1024 / 1024 doubles in, 1026 / 1024 / 1026 / 1026 out)
	// (kind 8: the real even transform of 2048 points, 1025 doubles in and out)
	const size_t n_in = kind == 8 ? 1025 * (size_t)batch : (kind >= 4 ? (kind == 5 ? 1026 : 1024) : kind == 3 ? 4096 : kind == 1 ? 2050 : 2048) * (size_t)batch;
	const size_t n_out = kind == 8 ? 1025 * (size_t)batch : (kind >= 4 ? (kind == 5 ? 1024 : 1026) : kind == 3 ? 4098 : kind == 1 ? 2048 : 2050) * (size_t)batch;
	Scoped d_in, d_out;
	WC_HIP(hipMalloc(&d_in.p, sizeof(double) * n_in));
	WC_HIP(hipMalloc(&d_out.p, sizeof(double) * n_out));
	WC_HIP(hipMemcpyAsync(d_in.p, in, sizeof(double) * n_in, hipMemcpyHostToDevice, s));
	const double *di = static_cast<const double *>(d_in.p);
	double *dout = static_cast<double *>(d_out.p);
	if (kind == 0) hipLaunchKernelGGL(hook_wave_fft_kernel<0>, dim3(batch), dim3(64), 0, s, di, dout, dev->twiddle);
	else if (kind == 1) hipLaunchKernelGGL(hook_wave_fft_kernel<1>, dim3(batch), dim3(64), 0, s, di, dout, dev->twiddle);
	else if (kind == 2) hipLaunchKernelGGL(hook_wave_fft_kernel<2>, dim3(batch), dim3(64), 0, s, di, dout, dev->twiddle);
	else if (kind == 3) hipLaunchKernelGGL(hook_wave2_r2c_kernel, dim3(batch), dim3(128), 0, s, di, dout, dev->twiddle);
	else if (kind == 4) hipLaunchKernelGGL(hook_wave8_fft_kernel<0>, dim3(batch), dim3(64), 0, s, di, dout, dev->twiddle);
	else if (kind == 5) hipLaunchKernelGGL(hook_wave8_fft_kernel<1>, dim3(batch), dim3(64), 0, s, di, dout, dev->twiddle);
	else if (kind == 6) hipLaunchKernelGGL(hook_wave8_fft_kernel<2>, dim3(batch), dim3(64), 0, s, di, dout, dev->twiddle);
	else if (kind == 7) hipLaunchKernelGGL(hook_wave8_fft_kernel<3>, dim3(batch), dim3(64), 0, s, di, dout, dev->twiddle);
	else hipLaunchKernelGGL(hook_even2048_kernel, dim3(batch), dim3(64), 0, s, di, dout, dev->twiddle);
	WC_HIP(hipGetLastError());
	WC_HIP(hipMemcpyAsync(out, d_out.p, sizeof(double) * n_out, hipMemcpyDeviceToHost, s));
	WC_HIP(hipStreamSynchronize(s));
	return WC_OK;
}
// the lean logarithm / exponential of wc_wavefft.hpp on n host values (kind 0 log, 1 exp)
int wc_debug_logexp(int kind, long long n, const double *in, double *out) {
	if (kind < 0 || kind > 1 || n <= 0 || !in || !out) return fail(WC_ERR_INVALID, "debug logexp: bad argument");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	hipStream_t s = dev->active();
	Scoped d_in, d_out;
	WC_HIP(hipMalloc(&d_in.p, sizeof(double) * n));
	WC_HIP(hipMalloc(&d_out.p, sizeof(double) * n));
	WC_HIP(hipMemcpyAsync(d_in.p, in, sizeof(double) * n, hipMemcpyHostToDevice, s));
	hipLaunchKernelGGL(hook_logexp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, kind, static_cast<const double *>(d_in.p),
					   static_cast<double *>(d_out.p), n, dev->twiddle);
	WC_HIP(hipGetLastError());
	WC_HIP(hipMemcpyAsync(out, d_out.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
	WC_HIP(hipStreamSynchronize(s));
	return WC_OK;
}

// div_const of wc_device.hpp on n host values: out[i] = in[i] / c by reciprocal, remainder and correction
int wc_debug_div_const(long long n, const double *in, double c, double *out) {
	if (n <= 0 || !in || !out || c == 0.0) return fail(WC_ERR_INVALID, "debug div_const: bad argument");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	hipStream_t s = dev->active();
	Scoped d_in, d_out;
	WC_HIP(hipMalloc(&d_in.p, sizeof(double) * n));
	WC_HIP(hipMalloc(&d_out.p, sizeof(double) * n));
	WC_HIP(hipMemcpyAsync(d_in.p, in, sizeof(double) * n, hipMemcpyHostToDevice, s));
	hipLaunchKernelGGL(hook_div_const_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, static_cast<const double *>(d_in.p),
					   static_cast<double *>(d_out.p), n, c, 1.0 / c);
	WC_HIP(hipGetLastError());
	WC_HIP(hipMemcpyAsync(out, d_out.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
	WC_HIP(hipStreamSynchronize(s));
	return WC_OK;
}

// reads `n` doubles of a device buffer it allocates (filled with ones) with 8-byte (wide = 0) or 16-byte (wide = 1) loads per
// lane, `reps` times; the byte count is n * 8 * reps (calibration of the HBM read counter)
int wc_debug_stream_read(long long n, int wide, int reps) {
	if (n <= 0 || reps <= 0) return fail(WC_ERR_INVALID, "debug stream read: bad argument");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	hipStream_t s = dev->active();
	Scoped d_in, d_out;
	WC_HIP(hipMalloc(&d_in.p, sizeof(double) * n));
	WC_HIP(hipMalloc(&d_out.p, sizeof(double)));
	WC_HIP(hipMemsetAsync(d_in.p, 0, sizeof(double) * n, s));
	for (int r = 0; r < reps; ++r) {
		if (wide) hipLaunchKernelGGL(hook_stream_read_kernel<true>, dim3(256 * 16), dim3(256), 0, s, static_cast<const double *>(d_in.p), n, static_cast<double *>(d_out.p));
		else hipLaunchKernelGGL(hook_stream_read_kernel<false>, dim3(256 * 16), dim3(256), 0, s, static_cast<const double *>(d_in.p), n, static_cast<double *>(d_out.p));
	}
	WC_HIP(hipGetLastError());
	WC_HIP(hipStreamSynchronize(s));
	return WC_OK;
}

}  // extern "C"
