// D4C band-aperiodicity estimation on gfx950.
//
// Restates reference src/d4c.cpp:113-503:
//   d4c_lovetrain_kernel  the "Love Train" voiced/unvoiced gate (reference :181-240), one workgroup per frame: Blackman
//                         3 T0 window, one real FFT, ratio of cumulative powers; also writes the 1 - 1e-12 rows of
//                         frames that fail the gate (reference :127-132)
//   d4c_frames_kernel     gated frames (reference :308-460), one workgroup per frame: two energy centroids (2 real
//                         FFTs each), smoothed power spectrum, static group delay (three prefix-sum smoothings)
//   d4c_band_kernel       (reference :466-503) one workgroup per (gated frame, 3 kHz band): Nuttall-windowed FFT whose
//                         power spectrum is ranked by an in-LDS radix select instead of std::sort (the reference only
//                         needs the sum of the bins-boundary-1 smallest values)
//   d4c_rows_kernel       coarse dB values -> interp1 -> linear rows (reference :162-168)
//   (WC_D4C_SPLIT=0 runs the last three in one fused d4c_frames_kernel.)
// The noise draws come from the exact stream positions of the reference's serial order: all
// LoveTrain frames first, then the gated frames (RNG draw-count contract).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "wc_device.hpp"
#include "wc_internal.hpp"
#include "wc_frames.hpp"
#include "wc_wavefft.hpp"
#include "wc_hostcopy.hpp"

namespace wc {

constexpr double kSafe = 0.000000000001;
constexpr int kMaxBands = 5;  // min(15000, fs/2 - 3000) / 3000

struct D4cArgs {
	const double *x;
	const UttDesc *utts;
	int n_utt;
	const double *tpos, *f0;
	const unsigned long long *rng_off;
	const uint32_t *rng_table;
	unsigned long long rng_base;
	const double2 *tw;
	double *ap;        // [total_frames][bins_out]
	double *ap0;       // [total_frames] LoveTrain result
	double *sgd;       // [total_frames][N/2+1] static group delay of gated frames (split schedule)
	double *coarse;    // [total_frames][kMaxBands] coarse aperiodicity (split schedule)
	uint32_t *cnt;     // [total_frames] draws of the main pass (written by LoveTrain)
	const double *nuttall;  // window_length_ entries
	long long total_frames;
	int fs, fft_size_out, n_ap, window_length;
	int select64;  // the band kernels' selection with 64-bit compares throughout (WC_D4C_SELECT=64: rounds 3-5; A/B and the bit-identity test)
	double threshold;
	long long sgd_stride;  // doubles per frame in sgd
	int *long_list, *long_cnt;  // gated frames whose windows exceed 2048 samples: left by d4c2_frames_kernel<false>, done by <true>
	const int *uidx;  // [total_frames] utterance of every frame (d4c_lt_count_kernel): the one-wavefront kernels read it instead of bisecting
	int rare_only;  // (unused)
	int *rare_list;  // [0]: count, then the gated frames the one-wavefront kernels leave to the block kernel
};

// F0-adaptive window of reference src/d4c.cpp:246-303 for the calling block; each thread keeps its
// N/T samples in registers.  type 1 = Hanning, 2 = Blackman.  Returns the window length.
// The window phase kappa * (i - hw) of a thread's samples i = tid + e T advances by a rotation recurrence
// from one exact sincos (16 steps at most: a few 1e-16 of drift), and the window is re-generated for the
// mean-removal pass instead of being kept, which is what keeps the register count down.
template <int N, int T>
__device__ __forceinline__ int d4c_windowed(const double *__restrict__ x, int x_len, int fs, double f0, double pos,
											int type, double ratio, const uint32_t *__restrict__ rng,
											unsigned long long roff, double (&wave)[N / T], double *red, int tid) {
	constexpr int EPT = N / T;
	// (the three windows of a frame share their cosines; an opaque thread index keeps the compiler from holding
	// them in registers -- and spilling them -- across the FFTs in between)
	asm volatile("" : "+v"(tid));
	const int hw = mround(ratio * fs / f0 / 2.0);
	const int wl = 2 * hw + 1;
	const int origin = mround(pos * fs + 0.001);
	const double c1 = 2.0 / ratio / fs;
	double cs0, sn0, csd, snd;  // angles in units of pi: no large-argument reduction code (and its registers)
	sincospi(f0 * (c1 * (tid - hw)), &sn0, &cs0);
	sincospi(f0 * (c1 * T), &snd, &csd);
	snd = uniform_d(snd);
	csd = uniform_d(csd);
	// Hanning 0.5 c + 0.5; Blackman 0.42 + 0.5 c + 0.08 cos 2theta = 0.34 + c (0.5 + 0.16 c): two fused multiply-adds
	auto win = [&](double c) { return type == 1 ? fma(0.5, c, 0.5) : fma(c, fma(0.16, c, 0.5), 0.34); };
	// (select forms instead of conditional updates: predicated register updates inside the unrolled loops cost a
	// register copy per value and iteration)
	double s1 = 0.0, s2 = 0.0;
	{
		double c = cs0, sn = sn0;
#pragma unroll
		for (int e = 0; e < EPT; ++e) {
			const int i = tid + e * T;
			const bool in = i < wl;
			const double w = in ? win(c) : 0.0;
			const int si = clampi(origin + i - hw, 0, x_len - 1);
			const double noise = randn_at(rng, roff + (in ? i : 0)) * kSafe;
			wave[e] = in ? fma(x[si], w, noise) : 0.0;
			s1 += wave[e];
			s2 += w;
			const double cn = fma(c, csd, -(sn * snd));
			sn = fma(sn, csd, c * snd);
			c = cn;
		}
	}
	block_sum2<T>(s1, s2, red, tid);
	const double wc = uniform_d(s1 / s2);
	{
		double c = cs0, sn = sn0;
#pragma unroll
		for (int e = 0; e < EPT; ++e) {
			const int i = tid + e * T;
			wave[e] = fma(-(i < wl ? win(c) : 0.0), wc, wave[e]);
			const double cn = fma(c, csd, -(sn * snd));
			sn = fma(sn, csd, c * snd);
			c = cn;
		}
	}
	return wl;
}

// DCCorrection of reference src/world_common.cpp:61-80, in place on P[0..M] in LDS.
template <int M, int T>
__device__ __forceinline__ void dc_correction_lds(double *P, double f0, int fs, int tid) {
	constexpr int N = 2 * M;
	const int upper = 2 + (int)(f0 * N / fs);
	const double dx = -(double)fs / N, rdx = 1.0 / dx;
	double rep[2];
#pragma unroll
	for (int e = 0; e < 2; ++e) {
		int i = tid + e * T;
		rep[e] = 0.0;
		if (i < upper - 1 && i <= M) {
			double axis = (double)i * fs / N;
			rep[e] = interp1q_rcp(f0, dx, rdx, [&](int b) { return P[min(max(b, 0), M)]; }, upper + 1, axis);
		}
	}
	__syncthreads();
#pragma unroll
	for (int e = 0; e < 2; ++e) {
		int i = tid + e * T;
		if (i < upper - 1 && i <= M) P[i] += rep[e];
	}
	__syncthreads();
}

// LinearSmoothing of reference src/world_common.cpp:27-52, :82-116.  P[0..M] in LDS is the input,
// S (>= N doubles of LDS) receives the mirrored cumulative segment, out(k, value) is called for
// every bin once the segment is complete (it may overwrite P[k]).  Ends with a __syncthreads().
template <int M, int T, bool NONNEG, class Out>
__device__ __forceinline__ void linear_smoothing_lds(const double *P, double *S, double width, int fs, double *red,
													 int tid, Out out) {
#pragma clang fp contract(fast)  // values only (the segment indices come from interp1q_rcp's own arithmetic)
	constexpr int N = 2 * M;
	int b = (int)(width * N / fs) + 1;
	if (M + 2 * b + 1 > N) b = (N - M - 1) / 2;
	const int len = M + 2 * b + 1;
	auto mir = [&](int i) -> double {
		if (i < b) return P[b - i];
		if (i < M + b) return P[i - b];
		return P[M - (i - (M + b))];
	};
	if (NONNEG) {
		// a power spectrum (the smoothed spectrum becomes a divisor): the cumulative sum in the reference's own sequential
		// rounding, seq_cumsum_nonneg of wc_device.hpp.  Once the terms stand in S, P is free until `out` rewrites it and
		// serves as the scratch of the sum.
		static_assert(M + 2 >= T + 2 * (T / 64), "P doubles as the scratch of the cumulative sum");
		for (int i = tid; i < len; i += T) S[i] = mir(i) * fs / N;
		__syncthreads();
		seq_cumsum_nonneg<T>(S, len, const_cast<double *>(P), red, tid);
	} else {
		// signed input (the group-delay numerator): block-scanned, re-associated sum
		const int ch = (len + T - 1) / T;
		const int lo = tid * ch, hi = min(len, lo + ch);
		double loc = 0.0;
		for (int i = lo; i < hi; ++i) loc += mir(i) * fs / N;
		double run = block_excl_scan<T>(loc, red, tid);
		for (int i = lo; i < hi; ++i) {
			run = mir(i) * fs / N + run;
			S[i] = run;
		}
		__syncthreads();
	}
	const double origin_axis = -(b - 0.5) * fs / N;
	const double step = (double)fs / N, rstep = 1.0 / step;
	auto seg = [&](int i) -> double { return S[min(max(i, 0), len - 1)]; };
	for (int k = tid; k <= M; k += T) {
		double lo_axis = (double)k / N * fs - width / 2.0;
		double hi_axis = lo_axis + width;
		double lo_v = interp1q_rcp(origin_axis, step, rstep, seg, len, lo_axis);
		double hi_v = interp1q_rcp(origin_axis, step, rstep, seg, len, hi_axis);
		out(k, (hi_v - lo_v) / width);
	}
	__syncthreads();
}

// Which gated frames the two-wavefront kernels (d4c2_*, N = 4096) take: their 18 KB of LDS hold the mirrored segment of the
// widest smoothing (2049 + 2 b + 1 terms, b = f0 N / fs + 1) and the low bins of the DC correction for F0 below ~1.4 kHz at
// 48 kHz (Harvest's ceiling is 800 Hz).  Frames above that go through d4c_frames_kernel, launched behind with rare_only.
__host__ __device__ __forceinline__ bool d4c2_can(double f0, int fs) {
	const int v = (int)(f0 * 4096 / fs);
	return v + 1 <= 120 && v + 2 <= 122;
}

// number of draws of one frame's LoveTrain window / of its three D4C windows
__global__ void d4c_lt_count_kernel(const double *__restrict__ f0, long long total, int fs, uint32_t *__restrict__ cnt,
									const UttDesc *__restrict__ utts, int n_utt, int *__restrict__ uidx, int *__restrict__ long_cnt,
									int *__restrict__ rare_cnt) {
	long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= total) return;
	if (g == 0) { *long_cnt = 0; *rare_cnt = 0; }
	uidx[g] = find_utt(utts, n_utt, g);  // (looked up once here rather than by every frame's wavefront, a chain of dependent loads each)
	double f = f0[g];
	cnt[g] = (f == 0.0) ? 0u : (uint32_t)(2 * mround(3.0 * fs / fmax(f, 40.0) / 2.0) + 1);
}

template <int N, int T>
__global__ __launch_bounds__(T, T == 512 ? 6 : 1) void d4c_lovetrain_kernel(D4cArgs a) {
	constexpr int M = N / 2;
	constexpr int EPT = N / T;
	__shared__ double2 A[fft_lds_size(M)];
	__shared__ double red[2 * (T / 64) + 2];
	double *Ar = reinterpret_cast<double *>(A);
	int tid = threadIdx.x;
	long long g = xcd_frame(blockIdx.x, a.total_frames);
	if (g >= a.total_frames) return;
	const int bins_out = a.fft_size_out / 2 + 1;
	double *__restrict__ row = a.ap + g * (long long)bins_out;
	const double f0v = a.f0[g];
	double ap0 = 0.0;
	if (f0v != 0.0) {
		const int u = find_utt(a.utts, a.n_utt, g);
		const UttDesc ud = a.utts[u];
		const int fs = a.fs;
		const double f0c = fmax(f0v, 40.0);
		double wave[EPT];
		const int wl = d4c_windowed<N, T>(a.x + ud.x_off, ud.x_len, fs, f0c, a.tpos[g], 2, 3.0, a.rng_table,
										  a.rng_off[g] - a.rng_base, wave, red, tid);
#pragma unroll
		for (int e = 0; e < EPT; ++e) {
			int i = tid + e * T;
			Ar[i] = (i < wl) ? wave[e] : 0.0;
		}
		__syncthreads();
		WC_FRESH(tid);
		fft_lds<M, T, +1>(A, a.tw, tid);
		WC_FRESH(tid);
		r2c_post<M, T>(A, a.tw, tid);
		// cumulative powers above 100 Hz up to 4000 Hz and 7900 Hz (reference :184-186, :226-235)
		const int b0 = (int)ceil(100.0 * N / fs);
		const int b1 = (int)ceil(4000.0 * N / fs);
		const int b2 = (int)ceil(7900.0 * N / fs);
		double p1 = 0.0, p2 = 0.0;
		for (int k = b0 + 1 + tid; k <= min(b2, M); k += T) {
			double2 v = A[k == M ? 0 : k];
			double p = (k == M) ? v.y * v.y : fma(v.x, v.x, v.y * v.y);
			p2 += p;
			if (k <= b1) p1 += p;
		}
		block_sum2<T>(p1, p2, red, tid);
		ap0 = p1 / p2;
	}
	const bool gate = !(f0v == 0.0 || ap0 <= a.threshold);  // reference :147
	if (tid == 0) {
		a.ap0[g] = ap0;
		a.cnt[g] = gate ? (uint32_t)(3 * (2 * mround(4.0 * a.fs / fmax(47.0, f0v) / 2.0) + 1)) : 0u;
	}
	if (!gate) {
		const double init_val = 1.0 - kSafe;
		for (int k = tid; k < bins_out; k += T) row[k] = init_val;
	}
}

#ifndef WC_D4C_FFTSYNC
#define WC_D4C_FFTSYNC 1  // FFT flags; 0 (no barriers) and 3 (no twiddle loads) are timing ablations with wrong results
#endif
template <int N, int T, bool SPLIT>
__device__ __forceinline__ void d4c_frames_body(const D4cArgs &a, const long long g) {
	constexpr int M = N / 2;
	constexpr int EPT = N / T;
	constexpr int KPT = (M + 1 + T - 1) / T;  // power-spectrum keys per thread
	// LDS: 32 KB + 2 x 16 KB at N = 4096 -> two workgroups per CU
	__shared__ double2 A[fft_lds_size(M)];  // FFT workspace / cumulative segment
	__shared__ double Br[M + 2];  // smoothed power spectrum, scratch of the last smoothing, radix-select histograms
	__shared__ double Cc[M + 2];  // centroid -> static group delay
	__shared__ double red[2 * (T / 64) + 2];
	__shared__ double red3[3 * (T / 64)];
	__shared__ double coarse[kMaxBands + 2];
	double *Ar = reinterpret_cast<double *>(A);

	int tid = threadIdx.x;
#define WC_FRESH_TID() WC_FRESH(tid)  // see wc_device.hpp
	const double f0v = a.f0[g];
	if (f0v == 0.0 || a.ap0[g] <= a.threshold) return;  // reference :147
	const int u = find_utt(a.utts, a.n_utt, g);
	const UttDesc ud = a.utts[u];
	const double *__restrict__ x = a.x + ud.x_off;
	const int fs = a.fs;
	const double f0 = uniform_d(fmax(47.0, f0v));
	const double pos = uniform_d(a.tpos[g]);
	const unsigned long long roff = a.rng_off[g] - a.rng_base;

	// ---- static centroid (reference :339-405) ----
	int wl = 0;
#pragma unroll 1
	for (int c = 0; c < 2; ++c) {
		WC_FRESH_TID();
		double wave[EPT];
		const double p = (c == 0) ? pos - 0.25 / f0 : pos + 0.25 / f0;
		wl = d4c_windowed<N, T>(x, ud.x_len, fs, f0, p, 2, 4.0, a.rng_table, roff + (unsigned long long)c * wl, wave, red, tid);
		double pw = 0.0;
#pragma unroll
		for (int e = 0; e < EPT; ++e) pw = fma(wave[e], wave[e], pw);
		pw = 1.0 / sqrt(block_sum<T>(pw, red, tid));  // (the reference divides every sample: an ulp apart)
#pragma unroll
		for (int e = 0; e < EPT; ++e) {
			int i = tid + e * T;
			wave[e] = (i < wl) ? wave[e] * pw : 0.0;
			Ar[i] = wave[e];
		}
		__syncthreads();
		fft_lds<M, T, +1, WC_D4C_FFTSYNC>(A, a.tw, tid);
		r2c_post<M, T>(A, a.tw, tid);
		double2 s1[KPT];  // spectrum of the plain windowed signal, kept in registers
#pragma unroll
		for (int e = 0; e < KPT; ++e) {
			int k = tid + e * T;
			s1[e] = make_double2(0.0, 0.0);
			if (k < M) s1[e] = A[k];
			else if (k == M) s1[e] = make_double2(A[0].y, 0.0);
		}
		__syncthreads();
		WC_FRESH_TID();
#pragma unroll
		for (int e = 0; e < EPT; ++e) {
			int i = tid + e * T;
			Ar[i] = wave[e] * (i + 1.0);
		}
		__syncthreads();
		fft_lds<M, T, +1, WC_D4C_FFTSYNC>(A, a.tw, tid);
		r2c_post<M, T>(A, a.tw, tid);
#pragma unroll
		for (int e = 0; e < KPT; ++e) {
			int k = tid + e * T;
			if (k <= M) {
				double v;
				if (k == 0) v = A[0].x * s1[e].x;
				else if (k == M) v = A[0].y * s1[e].x;
				else v = fma(A[k].x, s1[e].x, s1[e].y * A[k].y);
				Cc[k] = (c == 0) ? v : Cc[k] + v;
			}
		}
		__syncthreads();
	}
	dc_correction_lds<M, T>(Cc, f0, fs, tid);
#if defined(WC_D4C_STOP) && WC_D4C_STOP == 1
	if (tid == 0) a.ap[g * (long long)(a.fft_size_out / 2 + 1)] = Cc[1] + Br[1];
	return;
#endif

	// ---- smoothed power spectrum (reference :411-434) ----
	{
		WC_FRESH_TID();
		double wave[EPT];
		wl = d4c_windowed<N, T>(x, ud.x_len, fs, f0, pos, 1, 4.0, a.rng_table, roff + 2ull * wl, wave, red, tid);
#pragma unroll
		for (int e = 0; e < EPT; ++e) {
			int i = tid + e * T;
			Ar[i] = (i < wl) ? wave[e] : 0.0;
		}
		__syncthreads();
		fft_lds<M, T, +1, WC_D4C_FFTSYNC>(A, a.tw, tid);
		r2c_post<M, T>(A, a.tw, tid);
		for (int k = tid; k <= M; k += T) {
			double2 v = A[k == M ? 0 : k];
			Br[k] = (k == 0) ? v.x * v.x : (k == M) ? v.y * v.y : fma(v.x, v.x, v.y * v.y);
		}
		__syncthreads();
		dc_correction_lds<M, T>(Br, f0, fs, tid);
		linear_smoothing_lds<M, T, true>(Br, Ar, f0, fs, red, tid, [&](int k, double v) { Br[k] = v; });
	}
#if defined(WC_D4C_STOP) && WC_D4C_STOP == 2
	if (tid == 0) a.ap[g * (long long)(a.fft_size_out / 2 + 1)] = Cc[1] + Br[1];
	return;
#endif
	// ---- static group delay (reference :440-460) ----
	WC_FRESH_TID();
	for (int k = tid; k <= M; k += T) Cc[k] = Cc[k] / Br[k];
	__syncthreads();
	linear_smoothing_lds<M, T, false>(Cc, Ar, f0 / 2.0, fs, red, tid, [&](int k, double v) { Cc[k] = v; });
	linear_smoothing_lds<M, T, false>(Cc, Ar, f0, fs, red, tid, [&](int k, double v) { Br[k] = v; });
	for (int k = tid; k <= M; k += T) Cc[k] -= Br[k];
	__syncthreads();

#if defined(WC_D4C_STOP) && WC_D4C_STOP == 3
	if (tid == 0) a.ap[g * (long long)(a.fft_size_out / 2 + 1)] = Cc[1] + Br[1];
	return;
#endif
	if (SPLIT) {  // the band loop runs as its own kernel (one band per workgroup, leaner and with more waves in flight)
		double *__restrict__ dst = a.sgd + g * a.sgd_stride;
		for (int k = tid; k <= M; k += T) dst[k] = Cc[k];
		return;
	}
	// ---- coarse aperiodicity (reference :466-503) ----
	const int n_ap = a.n_ap;
	const int wln = a.window_length;
	const int hwl = wln / 2;
	const int boundary = mround(N * 8.0 / wln);
	const int bins = M + 1;
	const unsigned int K = (unsigned int)(bins - boundary - 1);  // sum of the K smallest powers
	// Per band: FFT of the Nuttall-windowed group delay, the power spectrum straight out of the real-FFT unpacking
	// into registers (KEYS per thread), then a radix select (8 bits per pass, most significant first;
	// non-negative doubles order like their bit patterns) of the K-th smallest power.  Every wave scans the
	// pass's 256-bin histogram itself, so the prefix / rank live in wave-uniform registers and a pass costs a
	// single barrier; a band stops as soon as its chosen bucket holds one element (typically 2-3 passes).
	// Histograms: one 1 KB row per pass in Br (no longer needed).
	constexpr int PAIRS = (M / 2) / T;
	constexpr int KEYS = 2 * PAIRS + 1;  // last slot: bin M/2, thread 0 only
	static_assert((M / 2) % T == 0, "pairs per thread");
	unsigned int(*hist)[256] = reinterpret_cast<unsigned int(*)[256]>(Br);
	for (int bnd = 0; bnd < n_ap; ++bnd) {
		WC_FRESH_TID();
		const int lane = tid & 63;
		const int center = (int)(3000.0 * (bnd + 1) * N / fs);
#pragma unroll
		for (int e = 0; e < EPT; ++e) {
			int i = tid + e * T;
			Ar[i] = (i < wln) ? Cc[center - hwl + i] * a.nuttall[i] : 0.0;
		}
		for (int i = tid; i < 8 * 256; i += T) (&hist[0][0])[i] = 0u;
		__syncthreads();
#ifndef WC_D4C_NOBANDFFT
		fft_lds<M, T, +1, WC_D4C_FFTSYNC>(A, a.tw, tid);
#endif
		double key[KEYS];
		r2c_power<M, T>(A, a.tw, tid, key);
		const int nkeys = (tid == 0) ? KEYS : KEYS - 1;
		unsigned long long pre = 0ull;
		unsigned int need = K, in_bucket = 0u;
		int shift = 56;
#ifndef WC_D4C_NPASS
#define WC_D4C_NPASS 8
#endif
		for (int pass = 0; pass < WC_D4C_NPASS; ++pass) {
			shift = 56 - 8 * pass;
#pragma unroll
			for (int e = 0; e < KEYS; ++e) {
				const unsigned long long bits = (unsigned long long)__double_as_longlong(key[e]);
				bool match = (e < nkeys) && ((pass == 0) || ((bits >> (shift + 8)) == (pre >> (shift + 8))));
				const unsigned int bucket = (unsigned int)((bits >> shift) & 255ull);
				// The leading digits are shared by nearly all keys (same exponent range): 64 lanes adding to one LDS
				// word serialise.  The two most common buckets of the wave are counted by ballot, one add each.
				unsigned long long act = __ballot(match);
#pragma unroll
				for (int it = 0; it < 2; ++it) {
					if (act != 0ull) {  // wave-uniform
						const int leader = __ffsll((long long)act) - 1;
						const unsigned int bl = __shfl(bucket, leader, 64);
						const unsigned long long same = __ballot(match && bucket == bl);
						if (lane == leader) atomicAdd(&hist[pass][bl], (unsigned int)__popcll(same));
						act &= ~same;
						match = match && bucket != bl;
					}
				}
				if (match) atomicAdd(&hist[pass][bucket], 1u);
			}
			__syncthreads();
			const uint4 h4 = reinterpret_cast<const uint4 *>(&hist[pass][0])[lane];
			const unsigned int own = h4.x + h4.y + h4.z + h4.w;
			unsigned int inc = own;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				unsigned int t = __shfl_up(inc, o, 64);
				if (lane >= o) inc += t;
			}
			const unsigned long long reach = __ballot(inc >= need);  // never empty: the total count >= need
			const int first = __ffsll((long long)reach) - 1;
			unsigned int acc = inc - own;
			unsigned int d = 4u * lane, cnt;
			if (acc + h4.x >= need) { cnt = h4.x; }
			else if (acc + h4.x + h4.y >= need) { acc += h4.x; d += 1; cnt = h4.y; }
			else if (acc + h4.x + h4.y + h4.z >= need) { acc += h4.x + h4.y; d += 2; cnt = h4.z; }
			else { acc += h4.x + h4.y + h4.z; d += 3; cnt = h4.w; }
			acc = __shfl(acc, first, 64);
			d = __shfl(d, first, 64);
			in_bucket = __shfl(cnt, first, 64);
			need -= acc;  // rank inside the chosen digit bucket
			pre |= ((unsigned long long)d) << shift;
			if (in_bucket == 1u) break;  // block-uniform: every wave derives the same values
		}
		// sum of the K smallest = sum(values below the bucket) + rank * v*, and the total.  v* is the one element of
		// the bucket (its value = the sum over the matching elements), or the full bit pattern after 8 passes.
		double low = 0.0, eq = 0.0, tot = 0.0;
#pragma unroll
		for (int e = 0; e < KEYS; ++e) {
			if (e < nkeys) {
				const double pw = key[e];
				const unsigned long long hb = ((unsigned long long)__double_as_longlong(pw)) >> shift;
				tot += pw;
				low += (hb < (pre >> shift)) ? pw : 0.0;
				eq += (hb == (pre >> shift)) ? pw : 0.0;
			}
		}
		block_sum3<T>(low, eq, tot, red3, tid);
		if (tid == 0) {
			const double thr = (in_bucket == 1u) ? eq : __longlong_as_double((long long)pre);
			const double part = low + (double)need * thr;
			const double cv = 10 * log10(part / tot);
			coarse[bnd + 1] = fmin(0.0, cv + (f0 - 100) / 50.0);  // reference :326-328
		}
	}
	if (tid == 0) { coarse[0] = -60.0; coarse[n_ap + 1] = -kSafe; }
	__syncthreads();
#if defined(WC_D4C_STOP) && WC_D4C_STOP == 4
	if (tid == 0) a.ap[g * (long long)(a.fft_size_out / 2 + 1)] = Cc[1] + Br[1];
	return;
#endif
	// ---- interp1 onto the output grid + dB -> linear (reference :162-168) ----
	WC_FRESH_TID();
	const int bins_out = a.fft_size_out / 2 + 1;
	double *__restrict__ row = a.ap + g * (long long)bins_out;
	const int na = n_ap + 2;
	for (int k = tid; k < bins_out; k += T) {
		double f = (double)k * fs / a.fft_size_out;
		int c = 1;  // histc semantics: clamp(#{j : axis[j] <= f}, 1, n-1)
		while (c < na && f >= ((c == na - 1) ? fs / 2.0 : c * 3000.0)) ++c;
		c = min(c, na - 1);
		double x0 = (c - 1) * 3000.0;
		double x1 = (c == na - 1) ? fs / 2.0 : c * 3000.0;
		double s = (f - x0) / (x1 - x0);
		double v = coarse[c - 1] + s * (coarse[c] - coarse[c - 1]);
		row[k] = exp(v * 0.11512925464970228);  // 10^(v/20) (reference :166) as e^(v ln10/20): an ulp-level difference, a fraction of pow's cost
	}
}

#undef WC_FRESH_TID
// every frame (XCD-contiguous order), or -- rare_only, behind the one-wavefront kernels -- the frames those have listed
// (two kernels: the loop over a list keeps what is invariant from frame to frame in registers, 147 - 190 of them instead of 124)
template <int N, int T, bool SPLIT, bool RARE = false>
__global__ __launch_bounds__(T, T >= 1024 ? 4 : (2 * T) / 256) void d4c_frames_kernel(D4cArgs a) {
	if constexpr (RARE) {
		const int n = a.rare_list[0];
#pragma unroll 1
		for (int i = blockIdx.x; i < n; i += gridDim.x) {
			d4c_frames_body<N, T, SPLIT>(a, a.rare_list[1 + i]);
			__syncthreads();
		}
	} else {
		const long long g = xcd_frame(blockIdx.x, a.total_frames);
		if (g >= a.total_frames) return;
		d4c_frames_body<N, T, SPLIT>(a, g);
	}
}

#ifndef WC_D4C_PRUNE
#define WC_D4C_PRUNE 1
#endif
// Split schedule, second kernel: one workgroup per (gated frame, band).  Same arithmetic as the band loop of
// d4c_frames_kernel (reference :466-503), with the group delay read back from global memory: 36 KB of LDS and 54
// registers instead of 64 KB / 128 VGPRs, i.e. four instead of two workgroups per CU.
template <int N, int T>
__global__ __launch_bounds__(T, T >= 1024 ? 8 : (4 * T) / 256) void d4c_band_kernel(D4cArgs a) {
	constexpr int M = N / 2;
	constexpr int EPT = N / T;
	constexpr int PAIRS = (M / 2) / T;
	constexpr int KEYS = 2 * PAIRS + 1;
	__shared__ double2 A[fft_lds_size(M)];
	__shared__ unsigned int hist[4][256];  // ring of pass histograms: pass p uses row p & 3 (36 KB of LDS: four workgroups per CU)
	__shared__ double red3[3 * (T / 64)];
	double *Ar = reinterpret_cast<double *>(A);
	int tid = threadIdx.x;
	const int n_ap = a.n_ap;
	// (frame, band) pairs XCD by XCD -- the XCD of a block is its index mod 8, so the pair is chosen from the block index itself:
	// the bands of a frame, whose slices of the group delay overlap by half, then meet in one L2 (no measurable difference to
	// dealing the frames alone, 6.9 ms either way)
	const long long blk = xcd_frame(blockIdx.x, (long long)gridDim.x);
	const long long g = blk / n_ap;
	const int bnd = (int)(blk % n_ap);
	if (g >= a.total_frames) return;
	const double f0v = a.f0[g];
	if (f0v == 0.0 || a.ap0[g] <= a.threshold) return;
	const double f0 = fmax(47.0, f0v);
	const int fs = a.fs;
	const int wln = a.window_length, hwl = wln / 2;
	const int boundary = mround(N * 8.0 / wln);
	const int bins = M + 1;
	const unsigned int K = (unsigned int)(bins - boundary - 1);
	const double *__restrict__ sgd = a.sgd + g * a.sgd_stride;
	const int center = (int)(3000.0 * (bnd + 1) * N / fs);
	for (int i = tid; i < 4 * 256; i += T) (&hist[0][0])[i] = 0u;
	if constexpr (M == 2048 && WC_D4C_PRUNE) {
		// The windowed group delay has 513 (<= 2 * 257) real samples of 4096: as interleaved complex z[0..256] it is zero
		// beyond entry 256, so the radix-2 pass and the first radix-4 pass of the transform only replicate -- the array
		// after them is A[p] = z[p >> 3], plus the terms of z[256] in the two butterflies that see it (p < 8).  Written
		// directly (bit-identical to running the passes), then the remaining four passes.
		if (wln <= 513) {
			auto xs = [&](int i) { return sgd[center - hwl + i] * a.nuttall[i]; };
#pragma unroll
			for (int e = 0; e < M / T; ++e) {
				const int p = tid + e * T;
				const int i0 = 2 * (p >> 3);
				double2 v = make_double2(xs(i0), xs(i0 + 1));
				if (p < 8 && wln == 513) {
					double2 x1 = make_double2(xs(512), 0.0);                      // z[256]
					if (p & 1) x1 = cmul(x1, tw_load(a.tw, kTwiddleN / 8));         // W_8^1 of the first radix-4 pass
					const int q = p >> 1;                                           // dft4 output q with x2 = x3 = 0
					const double2 t = (q & 1) ? make_double2(-x1.y, x1.x) : x1;     // i x1 for q = 1, 3
					v = (q & 2) ? csub(v, t) : cadd(v, t);
				}
				A[p] = v;
			}
			__syncthreads();
			WC_FRESH(tid);
			fft_lds_tail<M, T, +1, 8>(A, a.tw, tid);
		} else {
#pragma unroll
			for (int e = 0; e < EPT; ++e) {
				int i = tid + e * T;
				Ar[i] = (i < wln) ? sgd[center - hwl + i] * a.nuttall[i] : 0.0;
			}
			__syncthreads();
			WC_FRESH(tid);
			fft_lds<M, T, +1>(A, a.tw, tid);
		}
	} else {
#pragma unroll
		for (int e = 0; e < EPT; ++e) {
			int i = tid + e * T;
			Ar[i] = (i < wln) ? sgd[center - hwl + i] * a.nuttall[i] : 0.0;
		}
		__syncthreads();
		WC_FRESH(tid);
		fft_lds<M, T, +1>(A, a.tw, tid);
	}
	double key[KEYS];
	r2c_power<M, T>(A, a.tw, tid, key);
	WC_FRESH(tid);
	const int lane = tid & 63;
	const int nkeys = (tid == 0) ? KEYS : KEYS - 1;
	unsigned long long pre = 0ull;
	unsigned int need = K, in_bucket = 0u;
	int shift = 56;
	for (int pass = 0; pass < 8; ++pass) {
		shift = 56 - 8 * pass;
		unsigned int *__restrict__ hrow = hist[pass & 3];
		// row (pass - 2) & 3 was last read before the previous barrier; clear it for pass + 2
		if (pass >= 2) for (int i = tid; i < 256; i += T) hist[(pass - 2) & 3][i] = 0u;
#pragma unroll
		for (int e = 0; e < KEYS; ++e) {
			const unsigned long long bits = (unsigned long long)__double_as_longlong(key[e]);
			bool match = (e < nkeys) && ((pass == 0) || ((bits >> (shift + 8)) == (pre >> (shift + 8))));
			const unsigned int bucket = (unsigned int)((bits >> shift) & 255ull);
			unsigned long long act = __ballot(match);
#pragma unroll
			for (int it = 0; it < 2; ++it) {
				if (act != 0ull) {
					const int leader = __ffsll((long long)act) - 1;
					const unsigned int bl = __shfl(bucket, leader, 64);
					const unsigned long long same = __ballot(match && bucket == bl);
					if (lane == leader) atomicAdd(&hrow[bl], (unsigned int)__popcll(same));
					act &= ~same;
					match = match && bucket != bl;
				}
			}
			if (match) atomicAdd(&hrow[bucket], 1u);
		}
		__syncthreads();
		const uint4 h4 = reinterpret_cast<const uint4 *>(hrow)[lane];
		const unsigned int own = h4.x + h4.y + h4.z + h4.w;
		unsigned int inc = own;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			unsigned int t = __shfl_up(inc, o, 64);
			if (lane >= o) inc += t;
		}
		const unsigned long long reach = __ballot(inc >= need);
		const int first = __ffsll((long long)reach) - 1;
		unsigned int acc = inc - own;
		unsigned int d = 4u * lane, cnt;
		if (acc + h4.x >= need) { cnt = h4.x; }
		else if (acc + h4.x + h4.y >= need) { acc += h4.x; d += 1; cnt = h4.y; }
		else if (acc + h4.x + h4.y + h4.z >= need) { acc += h4.x + h4.y; d += 2; cnt = h4.z; }
		else { acc += h4.x + h4.y + h4.z; d += 3; cnt = h4.w; }
		acc = __shfl(acc, first, 64);
		d = __shfl(d, first, 64);
		in_bucket = __shfl(cnt, first, 64);
		need -= acc;
		pre |= ((unsigned long long)d) << shift;
		if (in_bucket == 1u) break;
	}
	double low = 0.0, eq = 0.0, tot = 0.0;
#pragma unroll
	for (int e = 0; e < KEYS; ++e) {
		if (e < nkeys) {
			const double pw = key[e];
			const unsigned long long hb = ((unsigned long long)__double_as_longlong(pw)) >> shift;
			tot += pw;
			low += (hb < (pre >> shift)) ? pw : 0.0;
			eq += (hb == (pre >> shift)) ? pw : 0.0;
		}
	}
	block_sum3<T>(low, eq, tot, red3, tid);
	if (tid == 0) {
		const double thr = (in_bucket == 1u) ? eq : __longlong_as_double((long long)pre);
		const double part = low + (double)need * thr;
		const double cv = 10 * log10(part / tot);
		a.coarse[g * kMaxBands + bnd] = fmin(0.0, cv + (f0 - 100) / 50.0);
	}
}

// Split schedule, third kernel: interp1 of the coarse aperiodicity onto the output grid + dB -> linear (reference :162-168)
__global__ __launch_bounds__(256) void d4c_rows_kernel(D4cArgs a) {
	const long long g = blockIdx.x;
	const double f0v = a.f0[g];
	if (f0v == 0.0 || a.ap0[g] <= a.threshold) return;
	const int n_ap = a.n_ap, fs = a.fs;
	const int bins_out = a.fft_size_out / 2 + 1;
	const double *__restrict__ co = a.coarse + g * kMaxBands;
	double *__restrict__ row = a.ap + g * (long long)bins_out;
	const int na = n_ap + 2;
	auto val = [&](int q) { return q == 0 ? -60.0 : (q == na - 1 ? -kSafe : co[q - 1]); };
	for (int k = threadIdx.x; k < bins_out; k += 256) {
		double f = (double)k * fs / a.fft_size_out;
		int c = 1;
		while (c < na && f >= ((c == na - 1) ? fs / 2.0 : c * 3000.0)) ++c;
		c = min(c, na - 1);
		double x0 = (c - 1) * 3000.0;
		double x1 = (c == na - 1) ? fs / 2.0 : c * 3000.0;
		double s = (f - x0) / (x1 - x0);
		double v = val(c - 1) + s * (val(c) - val(c - 1));
		row[k] = exp(v * 0.11512925464970228);
	}
}


// ==== N = 4096 (sampling rates above 24 kHz): one wavefront per frame, no barrier ==========================================
// The same arithmetic as the kernels above on the register-resident transforms of wc_wavefft.hpp.  A 4096-point real
// transform is two independent 1024-point complex ones (even and odd bins, wc_wavefft.hpp "even / odd split"), which the
// wavefront runs one after the other: 16 complex points per lane, two LDS exchanges per half, nothing shared with another
// wavefront, so no workgroup barrier anywhere (a first version with two cooperating wavefronts per frame spent more than
// half its time in the barriers around the exchanges: 7.9 ms against 3.7 ms without them for d4c_frames per 64
// utterances).  Bins stay in the lane that computed them: slot 4 g + q of parity p holds bin 2 (j_g + 256 q) + p.
// LDS: 18.5 KB per frame wavefront (exchange buffer = mirrored segment of the smoothings), 9.8 KB per band wavefront,
// instead of 64 KB / 36 KB per 8-wavefront workgroup.  Deviations from the kernels above, all at the 1e-15 level: the
// halving of the real-transform unpacking is folded into scale factors; the smoothings' two interpolation abscissae are
// k + c_lo, k + c_hi with one pair per frame; (hi - lo) * (1 / width).  The window is generated once per half (the
// wavefront cannot hold a frame's 4096 samples next to a transform in flight): 6 instead of 3 window generations per frame.
constexpr long long kD4Row = 4160;  // doubles per frame of the group-delay array when d4c2_frames_kernel parks its halves in it
constexpr int kD4Lds = 2304;  // doubles: the smoothings' 2049 + 2 b + 1 terms, b <= 120 (d4c2_can); the exchange buffer is its head

// F0-adaptive window of reference src/d4c.cpp:246-303 as the strided packed input of one half of the transform: slot q
// holds z[n] +- z[n + 1024] for n = lane + 64 q, z[m] = (sample 2 m, sample 2 m + 1) of the mean-removed windowed signal
// (minus for the odd half; windows longer than 2048 samples -- F0 below 94 Hz at 48 kHz -- reach the second term).
// type 1 = Hanning, 2 = Blackman.  rng: the frame's draws for this window.  weighted: sample i times (i + 1) (the second
// transform of the centroid).  Returns the window length; sumsq = sum of squares of the (unweighted) samples.
// SLOTS = 16: the window is known to be at most 2048 samples long (every lane's 16 slots hold all of it, z[n + 1024] = 0, `odd`
// plays no part: the caller forms both halves' inputs from the one result); SLOTS = 32: any length.
template <int SLOTS>
__device__ __forceinline__ int d4c2_windowed(int type, const double *__restrict__ x, int x_last, int fs, double f0, double pos, double ratio,
											 const uint32_t *__restrict__ rng, int odd, bool weighted, double (&re)[16],
											 double (&im)[16], double &sumsq, int lane) {
	WC_FRESH(lane);
	const int hw = __builtin_amdgcn_readfirstlane(mround(ratio * fs / f0 / 2.0));
	const int wl = 2 * hw + 1;
	const int base = __builtin_amdgcn_readfirstlane(mround(pos * fs + 0.001)) - hw;
	const double c1 = 2.0 / ratio / fs;
	double ce0, se0, co0, so0, cd, sd;  // angles in units of pi
	wf_sincospi(f0 * (c1 * (2 * lane - hw)), se0, ce0);
	wf_sincospi(f0 * (c1 * (2 * lane + 1 - hw)), so0, co0);
	wf_sincospi(f0 * (c1 * 128.0), sd, cd);
	sd = uniform_d(sd);
	cd = uniform_d(cd);
	// Hanning 0.5 c + 0.5; Blackman 0.42 + 0.5 c + 0.08 cos 2theta = 0.34 + c (0.5 + 0.16 c): a0 + c (0.5 + a2 c) for both
	const double a0 = type == 1 ? 0.5 : 0.34, a2 = type == 1 ? 0.0 : 0.16;
	auto win = [&](double c) { return fma(c, fma(a2, c, 0.5), a0); };
	// Walks window samples 2 lane + 128 q (+ 1), q = 0 .. 31 at most, in whole groups of four slots while they are live.  A
	// group's sixteen loads (signal and draws) are issued together and fenced off from their uses, so that they are in flight
	// at once (left alone, the scheduler waits for every one of them in turn).  LOADS: 0 none, 1 all groups, 2 second half only.
	auto walk = [&](auto loads_c, auto &&body) {
		constexpr int LOADS = decltype(loads_c)::value;
		double ce = ce0, se = se0, co = co0, so = so0;
#pragma unroll
		for (int qg = 0; qg < SLOTS; qg += 4) {
			if (qg * 128 >= wl) break;
			double xs[8];
			uint32_t ns[8];
			const bool ld = LOADS == 1 || (LOADS == 2 && qg >= 16);
			if (ld) {
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					const int i = 2 * lane + 128 * (qg + (k >> 1)) + (k & 1);
					xs[k] = x[clampi(base + i, 0, x_last)];
					ns[k] = rng[i < wl ? i : 0];
				}
				WF_SCHED_FENCE();
			}
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int q = qg + k, i0 = 2 * lane + 128 * q;
				const double We = (i0 < wl) ? win(ce) : 0.0, Wo = (i0 + 1 < wl) ? win(co) : 0.0;
				// the windowed samples with their noise floor (reference :287-291)
				const double ve = (ld && i0 < wl) ? fma(xs[2 * k], We, (ns[2 * k] / 268435456.0 - 6.0) * kSafe) : 0.0;
				const double vo = (ld && i0 + 1 < wl) ? fma(xs[2 * k + 1], Wo, (ns[2 * k + 1] / 268435456.0 - 6.0) * kSafe) : 0.0;
				body(q, i0, We, Wo, ve, vo);
				const double cen = fma(ce, cd, -(se * sd)), con = fma(co, cd, -(so * sd));
				se = fma(se, cd, ce * sd);
				so = fma(so, cd, co * sd);
				ce = cen;
				co = con;
			}
		}
	};
#pragma unroll
	for (int q = 0; q < 16; ++q) re[q] = im[q] = 0.0;
	double s1 = 0.0, s2 = 0.0;
	walk(std::integral_constant<int, 1>(), [&](int q, int, double We, double Wo, double ve, double vo) {
		if (q < 16) { re[q] = ve; im[q] = vo; }  // (the second half is formed again below: no room to keep it)
		s1 += ve + vo;
		s2 += We + Wo;
	});
	s1 = wave_sum_all(s1);
	s2 = wave_sum_all(s2);
	const double wc = s1 / s2;
	sumsq = 0.0;
	walk(std::integral_constant<int, 2>(), [&](int q, int i0, double We, double Wo, double ve, double vo) {
		if (q < 16) { ve = re[q]; vo = im[q]; }
		ve = fma(-We, wc, ve);
		vo = fma(-Wo, wc, vo);
		sumsq = fma(vo, vo, fma(ve, ve, sumsq));
		if (weighted) { ve *= i0 + 1.0; vo *= i0 + 2.0; }
		if (q < 16) { re[q] = ve; im[q] = vo; }
		else if (odd) { re[q & 15] -= ve; im[q & 15] -= vo; }
		else { re[q & 15] += ve; im[q & 15] += vo; }
	});
	sumsq = wave_sum_all(sumsq);
	return wl;
}
// number of leading four-slot groups of a half's input that are not all zero
__device__ __forceinline__ int d4c2_groups(int wl) { return wl > 2048 ? 4 : (wl + 511) >> 9; }

#ifndef WC_D4C2_LT_OCC
#define WC_D4C2_LT_OCC 2
#endif
// LONG = false: frames whose window fits 2048 samples (and the unvoiced ones): the window is formed once and serves both
// halves of the transform; LONG = true: the others (F0 below 70 Hz at 48 kHz), window formed per half.  Every frame is done by
// exactly one of the two launches.
template <bool LONG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WC_D4C2_LT_OCC, WC_D4C2_LT_OCC))) void d4c2_lovetrain_kernel(D4cArgs a) {
	constexpr int N = 4096, M = 2048;
	__shared__ __attribute__((aligned(16))) double L[kWfLds];
	const int lane = threadIdx.x;
	const long long g = xcd_frame(blockIdx.x, a.total_frames);
	if (g >= a.total_frames) return;
	const int bins_out = a.fft_size_out / 2 + 1;
	double *__restrict__ row = a.ap + g * (long long)bins_out;
	const double f0v = a.f0[g];
	const int fs = a.fs;
	const double f0c = uniform_d(fmax(f0v, 40.0));
	const bool is_long = f0v != 0.0 && 2 * mround(3.0 * fs / f0c / 2.0) + 1 > 2048;
	if (is_long != LONG) return;
	double ap0 = 0.0;
	if (f0v != 0.0) {
		const int u = a.uidx[g];
		const UttDesc ud = a.utts[u];
		// cumulative powers above 100 Hz up to 4000 Hz and 7900 Hz (reference :184-186, :226-235); a common factor (the
		// unpacking's 2) does not matter to their ratio
		const int b0 = (int)ceil(100.0 * N / fs);
		const int b1 = (int)ceil(4000.0 * N / fs);
		const int b2 = min((int)ceil(7900.0 * N / fs), M);
		double p1 = 0.0, p2 = 0.0;
		double mr[16], mi[16], unused;
		int wl = 0;
		if (!LONG) wl = d4c2_windowed<16>(2, a.x + ud.x_off, ud.x_len - 1, fs, f0c, a.tpos[g], 3.0, a.rng_table + (a.rng_off[g] - a.rng_base), 0,
										  false, mr, mi, unused, lane);
#pragma unroll 1
		for (int odd = 0; odd < 2; ++odd) {
			double re[16], im[16], nyq;
			if (LONG) {
				wl = d4c2_windowed<32>(2, a.x + ud.x_off, ud.x_len - 1, fs, f0c, a.tpos[g], 3.0, a.rng_table + (a.rng_off[g] - a.rng_base), odd,
									   false, re, im, unused, lane);
			} else {
#pragma unroll
				for (int q = 0; q < 16; ++q) { re[q] = mr[q]; im[q] = mi[q]; }
			}
			wf_r2c4096_half(re, im, nyq, d4c2_groups(wl), L, a.tw, lane, odd);
#pragma unroll
			for (int gq = 0; gq < 4; ++gq) {
				const int j = wf_j(lane, odd, gq);
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const int k = 2 * (j + 256 * q) + odd;
					const double p = fma(re[4 * gq + q], re[4 * gq + q], im[4 * gq + q] * im[4 * gq + q]);
					p2 += (k > b0 && k <= b2) ? p : 0.0;
					p1 += (k > b0 && k <= min(b1, b2)) ? p : 0.0;
				}
			}
			if (odd == 0 && lane == 0 && b2 == M) {
				p2 += nyq * nyq;
				if (b1 >= M) p1 += nyq * nyq;
			}
		}
		p1 = wave_sum_all(p1);
		p2 = wave_sum_all(p2);
		ap0 = p1 / p2;
	}
	const bool gate = !(f0v == 0.0 || ap0 <= a.threshold);  // reference :147
	if (lane == 0) {
		a.ap0[g] = ap0;
		a.cnt[g] = gate ? (uint32_t)(3 * (2 * mround(4.0 * a.fs / fmax(47.0, f0v) / 2.0) + 1)) : 0u;
	}
	if (!gate) {
		const double init_val = 1.0 - kSafe;
		for (int k = lane; k < bins_out; k += 64) row[k] = init_val;
	}
}

// A frame's per-bin values: v[p][4 g + q] = bin 2 (j_g + 256 q) + p, vM = bin 2048 (lane 0).
struct D4Bins {
	double v[2][16];
	double vM;
};
// DCCorrection (reference src/world_common.cpp:61-80): only bins below upper - 1 <= 121 change, all in slot A_0 of either
// parity (bins 2 lane, 2 lane + 1), from bins <= upper + 1 <= 123
__device__ __forceinline__ void d4c2_dc_correction(D4Bins &s, double f0, int fs, double *L, int lane) {
	constexpr int N = 4096;
	WC_FRESH(lane);
	const int upper = __builtin_amdgcn_readfirstlane(2 + (int)(f0 * N / fs));
	const double dx = -(double)fs / N, rdx = 1.0 / dx;
	L[2 * lane] = s.v[0][0];
	L[2 * lane + 1] = s.v[1][0];
	wf_fence();
	auto rep = [&](int i) {
		const double axis = (double)i * fs / N;
		return interp1q_rcp(f0, dx, rdx, [&](int b) { return L[min(max(b, 0), 127)]; }, upper + 1, axis);
	};
	if (2 * lane < upper - 1) s.v[0][0] += rep(2 * lane);
	if (2 * lane + 1 < upper - 1) s.v[1][0] += rep(2 * lane + 1);
	wf_fence();
}
// LinearSmoothing (reference src/world_common.cpp:27-52, :82-116), in place.  NONNEG: the cumulative sum in the reference's
// own sequential rounding (seq_cumsum_nonneg); otherwise scanned over the lanes.
template <bool NONNEG>
__device__ __forceinline__ void d4c2_smooth(D4Bins &s, double width, int fs, double *L, int lane) {
	constexpr int N = 4096, M = 2048;
	WC_FRESH(lane);
	const int b = __builtin_amdgcn_readfirstlane((int)(width * N / fs) + 1);  // <= 120 (d4c2_can)
	const int len = M + 2 * b + 1;
	int jg[2][4];
#pragma unroll
	for (int p = 0; p < 2; ++p)
#pragma unroll
		for (int gq = 0; gq < 4; ++gq) jg[p][gq] = 2 * wf_j(lane, p, gq) + p;
	// mirrored segment: position i holds bin b - i (i < b), bin i - b (b <= i < M + b), bin 2 M + b - i (M + b <= i <= M + 2 b).
	// The low mirror comes from slot A_0 (bins 2 lane, 2 lane + 1), the high one from slot B_3 (bins 2048 - 2 lane, 2047 - 2 lane).
#pragma unroll
	for (int p = 0; p < 2; ++p)
#pragma unroll
		for (int gq = 0; gq < 4; ++gq)
#pragma unroll
			for (int q = 0; q < 4; ++q) L[jg[p][gq] + 512 * q + b] = s.v[p][4 * gq + q] * fs / N;
	if (lane == 0) L[M + b] = s.vM * fs / N;
	{
		const int ke = 2 * lane, ko = 2 * lane + 1;
		if (ke >= 1 && ke <= b) {
			L[b - ke] = s.v[0][0] * fs / N;
			L[M + b + ke] = s.v[0][7] * fs / N;  // bin 2048 - 2 lane
		}
		if (ko <= b) {
			L[b - ko] = s.v[1][0] * fs / N;
			L[M + b + ko] = s.v[1][7] * fs / N;  // bin 2047 - 2 lane
		}
	}
	wf_fence();
	if (NONNEG) {
		seq_cumsum_nonneg_wave<36>(L, len, lane);
	} else {
		// the reference's sequential sum bit for bit here too (seq_cumsum_signed_wave): the group-delay numerator of a noise-free
		// band is a difference of neighbourhoods of this sum, decided by how every single addition rounded
		seq_cumsum_signed_wave<36>(L, len, lane);
	}
	const double step = (double)fs / N;
	const double origin_axis = -(b - 0.5) * fs / N;
	const double rstep = 1.0 / step;
	const double rwidth = uniform_d(1.0 / width);
	// (the abscissae in the reference's own per-bin arithmetic, see ct_wave_kernel)
	auto at = [&](int k) {
		const double lo_axis = (double)k / N * fs - width / 2.0, hi_axis = lo_axis + width;
		return (wf_interp1q(origin_axis, step, rstep, L, len, hi_axis) - wf_interp1q(origin_axis, step, rstep, L, len, lo_axis)) * rwidth;
	};
#pragma unroll
	for (int p = 0; p < 2; ++p)
#pragma unroll
		for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
			for (int q = 0; q < 4; ++q) s.v[p][4 * gq + q] = at(jg[p][gq] + 512 * q);
			WF_SCHED_FENCE();  // (four bins at a time: interleaving all of them overflows the registers)
		}
	s.vM = at(M);
	wf_fence();
}

#ifndef WC_D4C2_OCC
#define WC_D4C2_OCC 2
#endif
#ifndef WC_D4C2_MASTER_LDS
#define WC_D4C2_MASTER_LDS 1
#endif
// WC_D4C2_TRACE (development builds only): lane 0 stamps the shader clock at the phase boundaries of every gated frame into
// the tail of the frame's row (doubles 4128 .. 4143), which WC_D4C_TRACE=<file> dumps after the call (tools/d4c_trace.py)
#ifndef WC_D4C2_TRACE
#define WC_D4C2_TRACE 0
#endif
#if WC_D4C2_TRACE
#define D4_STAMP(i) do { if (lane == 0) reinterpret_cast<unsigned long long *>(park + 4128)[i] = __builtin_readcyclecounter(); } while (0)
#else
#define D4_STAMP(i) do { } while (0)
#endif
// gated frames up to the static group delay (reference :308-460), which the band kernel reads back
// LONG = false: frames whose windows fit 2048 samples (F0 of 94 Hz and above at 48 kHz): a window is formed once per position,
// parked in the frame's row and read back by both halves and both transforms; LONG = true: the others, window formed per half
// and transform.  Every gated frame is done by exactly one of the two launches.
// The frame's row of the group-delay array through a buffer resource: an access is one 32-bit offset register (shared by all
// the accesses of a lane), an immediate and a scalar offset -- as plain pointers the parked layouts' offsets (up to 33 KB, beyond
// the 4 KB immediate of a global access) made the compiler form a 64-bit vector address per access and keep dozens of them, half
// of the kernel's scratch.  Intrinsic accesses are not forwarded through registers either: what is parked really leaves them.
#ifndef WC_D4C2_BUFFER
#define WC_D4C2_BUFFER 1
#endif
#if WC_D4C2_BUFFER
typedef unsigned int d4_u2 __attribute__((ext_vector_type(2)));
struct D4Row { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ D4Row d4_row(double *p) {
	D4Row w;
	w.r = __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)(kD4Row * 8), 0x00020000);
	return w;
}
__device__ __forceinline__ double d4_ld(const D4Row &w, int idx) {
	const d4_u2 v = __builtin_amdgcn_raw_buffer_load_b64(w.r, idx * 8, 0, 0);
	return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ void d4_st(const D4Row &w, int idx, double d) {
	d4_u2 v;
	v.x = (unsigned)__double2loint(d);
	v.y = (unsigned)__double2hiint(d);
	__builtin_amdgcn_raw_buffer_store_b64(v, w.r, idx * 8, 0, 0);
}
#else
struct D4Row { double *p; };
__device__ __forceinline__ D4Row d4_row(double *p) { D4Row w; w.p = p; asm volatile("" : "+s"(w.p)); return w; }
__device__ __forceinline__ double d4_ld(const D4Row &w, int idx) { return w.p[idx]; }
__device__ __forceinline__ void d4_st(const D4Row &w, int idx, double d) { w.p[idx] = d; }
#endif
template <bool LONG>
__device__ __forceinline__ void d4c2_frame(const D4cArgs &a, const long long g, double *L, const int lane) {
	constexpr int M = 2048;
	const double f0v = a.f0[g];
	if (f0v == 0.0 || a.ap0[g] <= a.threshold) return;  // reference :147
	const int fs = a.fs;
	const double f0 = uniform_d(fmax(47.0, f0v));
	if (!d4c2_can(f0, fs)) {  // left for the block kernel behind this one
		if (!LONG && lane == 0) a.rare_list[1 + atomicAdd(a.rare_list, 1)] = (int)g;
		return;
	}
	const int wl = __builtin_amdgcn_readfirstlane(2 * mround(4.0 * fs / f0 / 2.0) + 1);
	if constexpr (!LONG) {
		if (wl > 2048) {  // left for the second launch (before anything else is fetched)
			if (lane == 0) a.long_list[1 + atomicAdd(a.long_cnt, 1)] = (int)g;
			return;
		}
	}
	const int u = a.uidx[g];
	const UttDesc ud = a.utts[u];
	const double *__restrict__ x = a.x + ud.x_off;
	const int x_last = ud.x_len - 1;
	const double pos = uniform_d(a.tpos[g]);
	const uint32_t *__restrict__ rng = a.rng_table + (a.rng_off[g] - a.rng_base);
	// Each half's results (centroid, power spectrum) wait in the frame's own row of the group-delay array (kD4Row doubles: four
	// blocks of 1024) until both halves are through: nothing but the running products stays in registers across the
	// transforms.  Every lane reads back what it wrote itself; the row's head is written for good at the end.
#ifdef WC_D4C2_PARK_ALIAS  // timing experiment only (results are garbage): all frames park in 2048 rows, which stay in the L2
	double *park = a.sgd + (g & 2047) * a.sgd_stride;
#else
	double *park = a.sgd + g * a.sgd_stride;
#endif
	const D4Row row = d4_row(park);
	D4_STAMP(0);

	// ---- static centroid (reference :339-405): at t -+ T0/4, Re S1 Re S2 + Im S1 Im S2 of the unit-energy windowed signal and
	// of the same signal times (n + 1), and the power spectrum of the Hanning-windowed frame (reference :411-434); bin by
	// bin, so half by half.  Jobs 0, 1: the centroid's two positions, job 2: the power spectrum, through one copy of the code.
	const int ng = d4c2_groups(wl);
	double cenM = 0.0, spsM = 0.0;  // bin 2048 (lane 0, even half)
	if constexpr (!LONG) {
#pragma unroll 1
		for (int job = 0; job < 3; ++job) {
			int ln = lane;
			WC_FRESH(ln);
			const double p = (job == 0) ? pos - 0.25 / f0 : (job == 1) ? pos + 0.25 / f0 : pos;
			double pw;
			{
				// the mean-removed window, once; parked in the row's second half (free until job 2 puts the power spectrum there)
				double mr[16], mi[16], sumsq;
				d4c2_windowed<16>(job == 2 ? 1 : 2, x, x_last, fs, f0, p, 4.0, rng + (long long)job * wl, 0, false, mr, mi, sumsq, ln);
				// (the reference divides every sample by the norm: an ulp apart); half of it: the transforms below then yield X, not
				// 2 X.  The power spectrum's window is not normalised: its 2 X is put right by the 0.25 below.
				pw = (job == 2) ? 1.0 : 0.5 * (1.0 / sqrt(sumsq));
#pragma unroll
				for (int q = 0; q < 16; ++q) {
					if (q < 4 * ng) {
						// (jobs 0, 1 read their window four times: its even samples wait in the LDS behind the exchange buffer,
						// free until job 2 puts the even half's power there)
						if (WC_D4C2_MASTER_LDS && job != 2) L[kWfLds + 64 * q + ln] = mr[q] * pw;
						else d4_st(row, 2048 + 64 * q + ln, mr[q] * pw);
						d4_st(row, 3072 + 64 * q + ln, mi[q] * pw);
					}
				}
				if (job == 0) D4_STAMP(1);
			}
			// (what a lane parks it reads back itself: program order is all the ordering the round trip needs)
			auto master = [&](double (&re)[16], double (&im)[16], bool weighted) {
				int ln = lane;
				WC_FRESH(ln);  // (the weights below must not be hoisted out of the loop over the halves and spilled)
#pragma unroll
				for (int q = 0; q < 16; ++q) {
					re[q] = im[q] = 0.0;
					if (q < 4 * ng) {
						if (WC_D4C2_MASTER_LDS && job != 2) re[q] = L[kWfLds + 64 * q + ln];
						else re[q] = d4_ld(row, 2048 + 64 * q + ln);
						im[q] = d4_ld(row, 3072 + 64 * q + ln);
					}
				}
				WF_SCHED_FENCE();
				if (weighted) {
#pragma unroll
					for (int q = 0; q < 16; ++q) {
						const int i0 = 2 * ln + 128 * q;
						re[q] *= i0 + 1.0;
						im[q] *= i0 + 2.0;
					}
				}
			};
#pragma unroll 1
			for (int odd = 0; odd < 2; ++odd) {
				double re[16], im[16], nyq1;
				master(re, im, false);
				wf_r2c4096_half(re, im, nyq1, ng, L, a.tw, ln, odd);
				if (job == 0 && odd == 0) D4_STAMP(2);
				if (job == 2) {
					// (the master copy of job 2 has been read by both halves only after the odd half's load above: the even
					// half's power goes to the row's FIRST quarter pair for now and is moved below)
					double pwr[16];
#pragma unroll
					for (int s = 0; s < 16; ++s) pwr[s] = 0.25 * fma(re[s], re[s], im[s] * im[s]);
					if (odd == 0) {
						spsM = 0.25 * (nyq1 * nyq1);
#pragma unroll
						for (int s = 0; s < 16; ++s) L[kWfLds + 64 * s + ln] = pwr[s];  // (LDS behind the exchange buffer: 1024 doubles)
					} else {
#pragma unroll
						for (int s = 0; s < 16; ++s) {
							d4_st(row, 3072 + 64 * s + ln, pwr[s]);
							d4_st(row, 2048 + 64 * s + ln, L[kWfLds + 64 * s + ln]);
						}
					}
				} else {
					double ar[16], ai[16], nyq2;
					master(ar, ai, true);
					wf_r2c4096_half(ar, ai, nyq2, ng, L, a.tw, ln, odd);
					double acc[16];
#pragma unroll
					for (int s = 0; s < 16; ++s) acc[s] = fma(re[s], ar[s], im[s] * ai[s]);
					if (job == 1) {
						double prev[16];
#pragma unroll
						for (int s = 0; s < 16; ++s) prev[s] = d4_ld(row, 1024 * odd + 64 * s + ln);
						WF_SCHED_FENCE();
#pragma unroll
						for (int s = 0; s < 16; ++s) acc[s] += prev[s];
					}
#pragma unroll
					for (int s = 0; s < 16; ++s) d4_st(row, 1024 * odd + 64 * s + ln, acc[s]);
					if (odd == 0) cenM += nyq1 * nyq2;
					if (job == 0 && odd == 0) D4_STAMP(3);
				}
			}
			if (job == 0) D4_STAMP(4);
		}
	} else
#pragma unroll 1
	for (int odd = 0; odd < 2; ++odd) {
		double acc[16], accM = 0.0;
#pragma unroll
		for (int s = 0; s < 16; ++s) acc[s] = 0.0;
#pragma unroll 1
		for (int job = 0; job < 3; ++job) {
			int ln = lane;
			WC_FRESH(ln);  // (what derives from the lane index must not be hoisted out of the loops and spilled)
			double ar[16], ai[16], re[16], im[16], sumsq, nyq1;
			const double p = (job == 0) ? pos - 0.25 / f0 : (job == 1) ? pos + 0.25 / f0 : pos;
			d4c2_windowed<32>(job == 2 ? 1 : 2, x, x_last, fs, f0, p, 4.0, rng + (long long)job * wl, odd, false, ar, ai, sumsq, ln);
			if (odd == 0 && job == 0) D4_STAMP(1);
			// (the reference divides every sample by the norm: an ulp apart); half of it: the transforms below then yield X, not 2 X.
			// The power spectrum's window is not normalised: its 2 X is put right by the 0.25 below.
			const double pw = (job == 2) ? 1.0 : 0.5 * (1.0 / sqrt(sumsq));
#pragma unroll
			for (int q = 0; q < 16; ++q) {
				re[q] = ar[q] * pw;
				im[q] = ai[q] * pw;
			}
			wf_r2c4096_half(re, im, nyq1, ng, L, a.tw, lane, odd);
			if (odd == 0 && job == 0) D4_STAMP(2);
			if (job == 2) {
#pragma unroll
				for (int s = 0; s < 16; ++s) d4_st(row, 2048 + 1024 * odd + 64 * s + ln, 0.25 * fma(re[s], re[s], im[s] * im[s]));
				if (odd == 0) spsM = 0.25 * (nyq1 * nyq1);
			} else {
				double nyq2;
				if (wl > 2048) {  // the weights (i + 1) differ between the two samples folded into a slot: form them again
					d4c2_windowed<32>(2, x, x_last, fs, f0, p, 4.0, rng + (long long)job * wl, odd, true, ar, ai, sumsq, ln);
#pragma unroll
					for (int q = 0; q < 16; ++q) { ar[q] *= pw; ai[q] *= pw; }
				} else {
#pragma unroll
					for (int q = 0; q < 16; ++q) {
						const int i0 = 2 * ln + 128 * q;
						ar[q] *= pw * (i0 + 1.0);
						ai[q] *= pw * (i0 + 2.0);
					}
				}
				wf_r2c4096_half(ar, ai, nyq2, ng, L, a.tw, lane, odd);
#pragma unroll
				for (int s = 0; s < 16; ++s) acc[s] += fma(re[s], ar[s], im[s] * ai[s]);
				accM += nyq1 * nyq2;
				if (odd == 0 && job == 0) D4_STAMP(3);
			}
		}
#pragma unroll
		for (int s = 0; s < 16; ++s) d4_st(row, 1024 * odd + 64 * s + lane, acc[s]);
		if (odd == 0) cenM = accM;
		if (odd == 0) D4_STAMP(4);
	}
	D4_STAMP(5);
	D4Bins cen, sps;
#pragma unroll
	for (int p = 0; p < 2; ++p)
#pragma unroll
		for (int s = 0; s < 16; ++s) {
			cen.v[p][s] = d4_ld(row, 1024 * p + 64 * s + lane);
			sps.v[p][s] = d4_ld(row, 2048 + 1024 * p + 64 * s + lane);
		}
	cen.vM = cenM;
	sps.vM = spsM;
	D4_STAMP(6);
	d4c2_dc_correction(cen, f0, fs, L, lane);
	d4c2_dc_correction(sps, f0, fs, L, lane);
	D4_STAMP(7);
	d4c2_smooth<true>(sps, f0, fs, L, lane);
	D4_STAMP(8);
	// ---- static group delay (reference :440-460) ----
#pragma unroll
	for (int p = 0; p < 2; ++p)
#pragma unroll
		for (int s = 0; s < 16; ++s) cen.v[p][s] = cen.v[p][s] / sps.v[p][s];
	cen.vM = cen.vM / sps.vM;
	// smoothed over f0 / 2, minus that smoothed once more over f0 (one copy of the code, run twice)
#pragma unroll 1
	for (int it = 0; it < 2; ++it) {
		d4c2_smooth<false>(cen, it ? f0 : f0 / 2.0, fs, L, lane);
		if (it == 0) sps = cen;
	}
	// (sps: once smoothed; cen: twice)
	D4_STAMP(9);
#pragma unroll
	for (int p = 0; p < 2; ++p)
#pragma unroll
		for (int gq = 0; gq < 4; ++gq) {
			const int j = 2 * wf_j(lane, p, gq) + p;
#pragma unroll
			for (int q = 0; q < 4; ++q) d4_st(row, j + 512 * q, sps.v[p][4 * gq + q] - cen.v[p][4 * gq + q]);
		}
	if (lane == 0) d4_st(row, M, sps.vM - cen.vM);
	D4_STAMP(10);
}

// LONG = false: a wavefront per frame.  LONG = true: the frames with longer windows (F0 below 94 Hz) are few or none, and a grid
// of one workgroup per frame that leaves at once still costs a launch of 64 k workgroups with 256 registers and scratch each
// (0.29 ms per half batch): the first launch lists them, a grid sized for the chip walks the list.
template <bool LONG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WC_D4C2_OCC, WC_D4C2_OCC))) void d4c2_frames_kernel(D4cArgs a) {
	__shared__ __attribute__((aligned(16))) double L[kD4Lds];
	const int lane = threadIdx.x;
	if constexpr (!LONG) {
		const long long g = xcd_frame(blockIdx.x, a.total_frames);
		if (g >= a.total_frames) return;
		d4c2_frame<false>(a, g, L, lane);
	} else {
		const int n_long = *a.long_cnt;
#pragma unroll 1
		for (int i = blockIdx.x; i < n_long; i += gridDim.x) {
			d4c2_frame<true>(a, a.long_list[1 + i], L, lane);
			wf_fence();
		}
	}
}

#ifndef WC_D4C2_BAND_OCC
#define WC_D4C2_BAND_OCC 2
#endif
#ifndef WC_D4C2_BRACKET
#define WC_D4C2_BRACKET 1
#endif
#ifndef WC_D4C2_BRACKET_FIRST
#define WC_D4C2_BRACKET_FIRST 0x1p-14
#endif
// one wavefront per (gated frame, band) (reference :466-503): Nuttall-windowed group delay (<= 1023 samples) -> the two
// halves of the 4096-point transform with pruned leading stages -> power spectrum in registers (33 keys per lane) -> the sum
// of the K = bins - boundary - 1 smallest powers by bisecting the bit patterns (non-negative doubles order like integers):
// per step one 64-bit compare per key and the count from ballots in scalar registers; it stops as soon as a threshold has
// exactly K keys below it (after ~log2(2^62 / gap between the K-th and the next key) steps), or with the K-th key itself.
// WC_D4C2_BAND_PF (round 5): a wavefront's first touch of its frame's group delay is an HBM round trip in front of everything it
// does, and two wavefronts per SIMD do not hide it (the rows were written by another launch, a gigabyte ago).  Every wavefront
// therefore also asks for the segment of the block that will run on ITS XCD about WC_D4C2_BAND_PF blocks later (xcd_frame:
// block b runs on XCD b % 8 with local index b >> 3): one load per lane, a cache line each, whose result nothing waits for until
// the wavefront ends -- by then that block finds its lines in the XCD's L2.  0: off.
// Measured (profiles/r05_b_prefetch_ab.txt, 64 x 10 s): 3.52 ms without, 3.35 / 3.57 / 3.40 ms at distances 192 / 384 / 768 -- inside
// the run-to-run spread of +-0.1 ms; the first touch is not what the kernel waits for.  Off by default.
#ifndef WC_D4C2_BAND_PF
#define WC_D4C2_BAND_PF 0
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WC_D4C2_BAND_OCC, WC_D4C2_BAND_OCC))) void d4c2_band_kernel(D4cArgs a) {
	constexpr int N = 4096, M = 2048;
	__shared__ __attribute__((aligned(16))) double L[kWfLds];
	const int lane = threadIdx.x;
	const int n_ap = a.n_ap;
	const long long blk = xcd_frame(blockIdx.x, (long long)gridDim.x);
	const long long g = blk / n_ap;
	const int bnd = (int)(blk % n_ap);
	if (g >= a.total_frames) return;
	const double f0v = a.f0[g];
	if (f0v == 0.0 || a.ap0[g] <= a.threshold) return;
	const double f0 = fmax(47.0, f0v);
	const int fs = a.fs;
	const int wln = a.window_length, hwl = wln / 2;  // <= 1023 samples = 512 packed points: slots 0 .. 7
	const int boundary = mround(N * 8.0 / wln);
	const unsigned int K = (unsigned int)(M + 1 - boundary - 1);
	const int center = (int)(3000.0 * (bnd + 1) * N / fs);
	const double *__restrict__ src = a.sgd + g * a.sgd_stride + (center - hwl);
	const int ng = wln > 512 ? 2 : 1;
	double key[2][16], keyM = 0.0;
	[[maybe_unused]] double pf = 0.0;
#pragma unroll
	for (int odd = 0; odd < 2; ++odd) {
		// the packed windowed group delay, elements lane + 64 q (read again for the second half rather than held across the first)
		double re[16], im[16], nyq;
		{
			double sv[16], nv[16];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				const int i = min(2 * lane + 128 * (k >> 1) + (k & 1), wln - 1);
				sv[k] = src[i];
				nv[k] = a.nuttall[i];
			}
#if WC_D4C2_BAND_PF > 0
			if (odd == 0) {
				// (no branch around the load -- the tail of the grid asks for the last block's lines again --: behind a branch the
				// compiler no longer knows how many loads are in flight and waits for all of them, this one included)
				const long long b2 = min((long long)blockIdx.x + 8ll * WC_D4C2_BAND_PF, (long long)gridDim.x - 1);
				const long long blk2 = xcd_frame(b2, (long long)gridDim.x);
				const long long g2 = min(blk2 / n_ap, a.total_frames - 1);
				const int center2 = (int)(3000.0 * ((int)(blk2 % n_ap) + 1) * N / fs);
				pf = (a.sgd + g2 * a.sgd_stride + (center2 - hwl))[min(16 * lane, wln - 1)];  // (a line per lane)
			}
#endif
			WF_SCHED_FENCE();
#pragma unroll
			for (int q = 0; q < 16; ++q) {
				re[q] = im[q] = 0.0;
				if (q < 8) {
					const int i0 = 2 * lane + 128 * q;
					re[q] = (i0 < wln) ? sv[2 * q] * nv[2 * q] : 0.0;
					im[q] = (i0 + 1 < wln) ? sv[2 * q + 1] * nv[2 * q + 1] : 0.0;
				}
			}
		}
		if (odd) wf_odd_twist(re, im, a.tw, lane, 8);
		if (ng == 1) wdft16<+1, 1>(re, im);
		else wdft16<+1, 2>(re, im);
		wf_fft1024_dit_rest_p<+1>(re, im, L, a.tw, lane, odd);
		if (odd) wf_r2c_unpack_odd(re, im, a.tw, lane);
		else wf_r2c_unpack(re, im, nyq, a.tw, lane);  // 2 X: a common factor of all powers
#pragma unroll
		for (int s = 0; s < 16; ++s) key[odd][s] = fma(re[s], re[s], im[s] * im[s]);
		if (!odd) keyM = nyq * nyq;  // lane 0
	}
	// largest key
	double mx = (lane == 0) ? keyM : 0.0;
#pragma unroll
	for (int s = 0; s < 16; ++s) mx = fmax(mx, fmax(key[0][s], key[1][s]));
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
	mx = uniform_d(mx);
	long long lo = -1, hi = __double_as_longlong(mx);
	unsigned int c_lo = 0;
	if (!a.select64) {
		// The same search on the keys' HIGH words first (round 6): a threshold (h, 0xFFFFFFFF) has a key at or below it exactly when the
		// key's high word is at most h -- a 32-bit compare per key and step instead of a 64-bit one, on registers that are there anyway.
		// It ends with exactly K keys below a threshold (nearly always: two neighbours of the ranking share a high word only when they
		// are within 1e-6 of each other) or with the one high word the K-th key has, where the 64-bit search below takes over.
		auto count_hi = [&](int h) {
			unsigned int c = (unsigned int)__popcll(__ballot(lane == 0 && __double2hiint(keyM) <= h));
#pragma unroll
			for (int s = 0; s < 16; ++s) {
				c += (unsigned int)__popcll(__ballot(__double2hiint(key[0][s]) <= h));
				c += (unsigned int)__popcll(__ballot(__double2hiint(key[1][s]) <= h));
			}
			return c;
		};
		int lo_h = -1, hi_h = __double2hiint(mx);
#pragma unroll 1
		for (int tr = 0; tr < 2; ++tr) {  // (the bracket of the 64-bit search: 2^-14, else 2^-40 below the largest key)
			const int cand = __double2hiint(mx * (tr == 0 ? WC_D4C2_BRACKET_FIRST : 0x1p-40)) - 1;
			if (cand < 0) continue;
			const unsigned int c = count_hi(cand);
			if (c <= K) { lo_h = cand; c_lo = c; break; }
			hi_h = cand;
		}
		while (hi_h - lo_h > 1 && c_lo != K) {
			const int mid = lo_h + ((hi_h - lo_h) >> 1);
			const unsigned int c = count_hi(mid);
			if (c >= K) hi_h = mid;
			if (c <= K) { lo_h = mid; c_lo = c; }
		}
		lo = lo_h < 0 ? -1ll : (((long long)lo_h << 32) | 0xFFFFFFFFll);
		hi = min(hi, ((long long)hi_h << 32) | 0xFFFFFFFFll);
	} else {
		// The keys left out of the sum are the boundary + 1 largest of a smooth spectrum (the main lobe of its strongest line,
		// mostly): they lie within a few binades of the largest one, so the search starts from a threshold 2^-14 below it where
		// that holds (at most K keys at or below it), else from 2^-40 -- seven steps fewer than from the whole range of bit patterns.
		auto count_le = [&](long long t) {
			unsigned int c = (unsigned int)__popcll(__ballot(lane == 0 && __double_as_longlong(keyM) <= t));
#pragma unroll
			for (int s = 0; s < 16; ++s) {
				c += (unsigned int)__popcll(__ballot(__double_as_longlong(key[0][s]) <= t));
				c += (unsigned int)__popcll(__ballot(__double_as_longlong(key[1][s]) <= t));
			}
			return c;
		};
#pragma unroll 1
		for (int tr = 0; tr < 2; ++tr) {
			const long long cand = __double_as_longlong(mx * (tr == 0 ? WC_D4C2_BRACKET_FIRST : 0x1p-40));
			const unsigned int c = count_le(cand);
			if (c <= K && cand > 0) { lo = cand; c_lo = c; break; }
			hi = cand > 0 && c >= K ? cand : hi;  // (more than K below: the K-th smallest is at or below cand)
		}
	}
	for (int it = 0; it < 64 && hi - lo > 1 && c_lo != K; ++it) {
		const long long mid = lo + ((hi - lo) >> 1);
		unsigned int c = (unsigned int)__popcll(__ballot(lane == 0 && __double_as_longlong(keyM) <= mid));
#pragma unroll
		for (int s = 0; s < 16; ++s) {
			c += (unsigned int)__popcll(__ballot(__double_as_longlong(key[0][s]) <= mid));
			c += (unsigned int)__popcll(__ballot(__double_as_longlong(key[1][s]) <= mid));
		}
		if (c >= K) hi = mid;
		if (c <= K) { lo = mid; c_lo = c; }
		if (c == K) break;
	}
	// sum of the K smallest = sum(keys <= lo) + (K - #{keys <= lo}) * (the key at hi), and the total
	double low = 0.0, tot = 0.0;
#pragma unroll
	for (int p = 0; p < 2; ++p)
#pragma unroll
		for (int s = 0; s < 16; ++s) {
			tot += key[p][s];
			low += (__double_as_longlong(key[p][s]) <= lo) ? key[p][s] : 0.0;
		}
	if (lane == 0) {
		tot += keyM;
		low += (__double_as_longlong(keyM) <= lo) ? keyM : 0.0;
	}
	low = wave_sum_all(low);
	tot = wave_sum_all(tot);
	if (lane == 0) {
		const double part = low + (double)(K - c_lo) * __longlong_as_double(hi);
		const double cv = 10 * log10(part / tot);
		a.coarse[g * kMaxBands + bnd] = fmin(0.0, cv + (f0 - 100) / 50.0);  // reference :326-328
	}
#if WC_D4C2_BAND_PF > 0
	// (the prefetched value is "used" only here, behind the wavefront's last store, so that the load is neither dropped nor waited
	// for earlier: an empty statement that takes it in a register)
	asm volatile("" ::"v"(pf));
#endif
}


// ==== N = 2048 (16 / 22.05 / 24 kHz): the same kernels where a real transform is ONE 1024-point complex one ===================
// d4c2_* without the even / odd split: every window fits the 2048 samples of the sixteen slots (4 fs / 47 <= 2043 at 24 kHz), a
// transform is wf_r2c4096_half's even form on the unfolded input (the 2048-point real transform itself), slot 4 g + q holds bin
// j_g + 256 q, bin 1024 travels beside them in lane 0.  No second launch for long windows, half the parked traffic.
constexpr long long kD4Row1 = 3136;  // doubles per frame of the group-delay array: centroid | power | parked odd samples (+ pad)
constexpr int kD4Lds1 = 2176;  // doubles: exchange buffer (1152) + the parked even samples (1024); the smoothings' 1025 + 2 b + 1 terms, b <= 120, fit its head
__host__ __device__ __forceinline__ bool d4c1_can(double f0, int fs) {
	const int v = (int)(f0 * 2048 / fs);
	return v + 1 <= 120 && v + 2 <= 122;
}
struct D4Bins1 {
	double v[16];
	double vM;
};
// DCCorrection (reference src/world_common.cpp:61-80): only bins below upper - 1 <= 121 change: slot A_0 (bin lane) and C_0
// (bin 64 + lane), from bins <= upper + 1 <= 123
__device__ __forceinline__ void d4c1_dc_correction(D4Bins1 &s, double f0, int fs, double *L, int lane) {
	constexpr int N = 2048;
	WC_FRESH(lane);
	const int upper = __builtin_amdgcn_readfirstlane(2 + (int)(f0 * N / fs));
	const double dx = -(double)fs / N, rdx = 1.0 / dx;
	L[lane] = s.v[0];
	L[64 + lane] = s.v[8];
	wf_fence();
	auto rep = [&](int i) {
		const double axis = (double)i * fs / N;
		return interp1q_rcp(f0, dx, rdx, [&](int b) { return L[min(max(b, 0), 127)]; }, upper + 1, axis);
	};
	if (lane < upper - 1) s.v[0] += rep(lane);
	if (upper - 1 > 64) {
		if (64 + lane < upper - 1) s.v[8] += rep(64 + lane);
	}
	wf_fence();
}
// LinearSmoothing (reference src/world_common.cpp:27-52, :82-116), in place; the cumulative sum in the reference's own
// sequential rounding, for non-negative terms (NONNEG) or terms of either sign (see d4c2_smooth)
template <bool NONNEG>
__device__ __forceinline__ void d4c1_smooth(D4Bins1 &s, double width, int fs, double *L, int lane) {
	constexpr int N = 2048, M = 1024;
	WC_FRESH(lane);
	const int b = __builtin_amdgcn_readfirstlane((int)(width * N / fs) + 1);  // <= 120 (d4c1_can)
	const int len = M + 2 * b + 1;
	int jg[4];
#pragma unroll
	for (int gq = 0; gq < 4; ++gq) jg[gq] = wf_bin(lane, gq, 0);
	// mirrored segment: position i holds bin b - i (i < b), bin i - b (b <= i < M + b), bin 2 M + b - i (M + b <= i <= M + 2 b).
	// The low mirror comes from slots A_0 (bins 1 .. 63) and C_0 (bins 64 .. 127), the high one from B_3 (bins 1024 - lane) and
	// D_3 (bins 960 - lane).
#pragma unroll
	for (int gq = 0; gq < 4; ++gq)
#pragma unroll
		for (int q = 0; q < 4; ++q) L[jg[gq] + 256 * q + b] = s.v[4 * gq + q] * fs / N;
	if (lane == 0) L[M + b] = s.vM * fs / N;
	if (lane >= 1 && lane <= b) {
		L[b - lane] = s.v[0] * fs / N;
		L[M + b + lane] = s.v[7] * fs / N;
	}
	if (64 + lane <= b) {
		L[b - 64 - lane] = s.v[8] * fs / N;
		L[M + b + 64 + lane] = s.v[15] * fs / N;
	}
	wf_fence();
	if (NONNEG) seq_cumsum_nonneg_wave<20>(L, len, lane);
	else seq_cumsum_signed_wave<20>(L, len, lane);
	const double step = (double)fs / N;
	const double origin_axis = -(b - 0.5) * fs / N;
	const double rstep = 1.0 / step;
	const double rwidth = uniform_d(1.0 / width);
	auto at = [&](int k) {
		const double lo_axis = (double)k / N * fs - width / 2.0, hi_axis = lo_axis + width;
		return (wf_interp1q(origin_axis, step, rstep, L, len, hi_axis) - wf_interp1q(origin_axis, step, rstep, L, len, lo_axis)) * rwidth;
	};
#pragma unroll
	for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
		for (int q = 0; q < 4; ++q) s.v[4 * gq + q] = at(jg[gq] + 256 * q);
		WF_SCHED_FENCE();  // (four bins at a time: interleaving all of them overflows the registers)
	}
	s.vM = at(M);
	wf_fence();
}

#ifndef WC_D4C1_LT_OCC
#define WC_D4C1_LT_OCC 3
#endif
#ifndef WC_D4C1_OCC
#define WC_D4C1_OCC 2
#endif
#ifndef WC_D4C1_BAND_OCC
#define WC_D4C1_BAND_OCC 3
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WC_D4C1_LT_OCC, WC_D4C1_LT_OCC))) void d4c1_lovetrain_kernel(D4cArgs a) {
	constexpr int N = 2048, M = 1024;
	__shared__ __attribute__((aligned(16))) double L[kWfLds];
	const int lane = threadIdx.x;
	const long long g = xcd_frame(blockIdx.x, a.total_frames);
	if (g >= a.total_frames) return;
	const int bins_out = a.fft_size_out / 2 + 1;
	double *__restrict__ row = a.ap + g * (long long)bins_out;
	const double f0v = a.f0[g];
	const int fs = a.fs;
	const double f0c = uniform_d(fmax(f0v, 40.0));
	double ap0 = 0.0;
	if (f0v != 0.0) {
		const int u = a.uidx[g];
		const UttDesc ud = a.utts[u];
		// cumulative powers above 100 Hz up to 4000 Hz and 7900 Hz (reference :184-186, :226-235); a common factor (the
		// unpacking's 2) does not matter to their ratio
		const int b0 = (int)ceil(100.0 * N / fs);
		const int b1 = (int)ceil(4000.0 * N / fs);
		const int b2 = min((int)ceil(7900.0 * N / fs), M);
		double p1 = 0.0, p2 = 0.0;
		double re[16], im[16], unused, nyq;
		const int wl = d4c2_windowed<16>(2, a.x + ud.x_off, ud.x_len - 1, fs, f0c, a.tpos[g], 3.0, a.rng_table + (a.rng_off[g] - a.rng_base), 0,
										 false, re, im, unused, lane);
		wf_r2c4096_half(re, im, nyq, d4c2_groups(wl), L, a.tw, lane, 0);
#pragma unroll
		for (int gq = 0; gq < 4; ++gq) {
			const int j = wf_bin(lane, gq, 0);
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int k = j + 256 * q;
				const double p = fma(re[4 * gq + q], re[4 * gq + q], im[4 * gq + q] * im[4 * gq + q]);
				p2 += (k > b0 && k <= b2) ? p : 0.0;
				p1 += (k > b0 && k <= min(b1, b2)) ? p : 0.0;
			}
		}
		if (lane == 0 && b2 == M) {
			p2 += nyq * nyq;
			if (b1 >= M) p1 += nyq * nyq;
		}
		p1 = wave_sum_all(p1);
		p2 = wave_sum_all(p2);
		ap0 = p1 / p2;
	}
	const bool gate = !(f0v == 0.0 || ap0 <= a.threshold);  // reference :147
	if (lane == 0) {
		a.ap0[g] = ap0;
		a.cnt[g] = gate ? (uint32_t)(3 * (2 * mround(4.0 * a.fs / fmax(47.0, f0v) / 2.0) + 1)) : 0u;
	}
	if (!gate) {
		const double init_val = 1.0 - kSafe;
		for (int k = lane; k < bins_out; k += 64) row[k] = init_val;
	}
}

// gated frames up to the static group delay (reference :308-460), which the band kernel reads back (see d4c2_frame)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WC_D4C1_OCC, WC_D4C1_OCC))) void d4c1_frames_kernel(D4cArgs a) {
	constexpr int M = 1024;
	__shared__ __attribute__((aligned(16))) double L[kD4Lds1];
	const int lane = threadIdx.x;
	const long long g = xcd_frame(blockIdx.x, a.total_frames);
	if (g >= a.total_frames) return;
	const double f0v = a.f0[g];
	if (f0v == 0.0 || a.ap0[g] <= a.threshold) return;  // reference :147
	const int fs = a.fs;
	const double f0 = uniform_d(fmax(47.0, f0v));
	if (!d4c1_can(f0, fs)) {  // left for the block kernel behind this one
		if (lane == 0) a.rare_list[1 + atomicAdd(a.rare_list, 1)] = (int)g;
		return;
	}
	const int wl = __builtin_amdgcn_readfirstlane(2 * mround(4.0 * fs / f0 / 2.0) + 1);
	const int u = a.uidx[g];
	const UttDesc ud = a.utts[u];
	const double *__restrict__ x = a.x + ud.x_off;
	const int x_last = ud.x_len - 1;
	const double pos = uniform_d(a.tpos[g]);
	const uint32_t *__restrict__ rng = a.rng_table + (a.rng_off[g] - a.rng_base);
	// the frame's own row of the group-delay array: [0, 1024) the centroid, [1024, 2048) the power spectrum, [2048, 3072) the
	// parked odd samples of the current window; the row's head is written for good at the end
	double *park = a.sgd + g * a.sgd_stride;
	const D4Row row = d4_row(park);

	// ---- static centroid (reference :339-405) at t -+ T0/4 and the power spectrum of the Hanning-windowed frame (:411-434):
	// jobs 0, 1, 2 through one copy of the code ----
	const int ng = d4c2_groups(wl);
	double cenM = 0.0, spsM = 0.0;  // bin 1024 (lane 0)
#pragma unroll 1
	for (int job = 0; job < 3; ++job) {
		int ln = lane;
		WC_FRESH(ln);
		const double p = (job == 0) ? pos - 0.25 / f0 : (job == 1) ? pos + 0.25 / f0 : pos;
		double re[16], im[16], nyq1;
		{
			double sumsq;
			d4c2_windowed<16>(job == 2 ? 1 : 2, x, x_last, fs, f0, p, 4.0, rng + (long long)job * wl, 0, false, re, im, sumsq, ln);
			// (the reference divides every sample by the norm: an ulp apart); half of it: the transforms below then yield X, not
			// 2 X.  The power spectrum's window is not normalised: its 2 X is put right by the 0.25 below.
			const double pw = (job == 2) ? 1.0 : 0.5 * (1.0 / sqrt(sumsq));
#pragma unroll
			for (int q = 0; q < 16; ++q) {
				re[q] *= pw;
				im[q] *= pw;
				if (job != 2 && q < 4 * ng) {  // the centroid's second transform reads the window again
					L[kWfLds + 64 * q + ln] = re[q];
					d4_st(row, 2048 + 64 * q + ln, im[q]);
				}
			}
		}
		wf_r2c4096_half(re, im, nyq1, ng, L, a.tw, ln, 0);
		if (job == 2) {
#pragma unroll
			for (int s = 0; s < 16; ++s) d4_st(row, 1024 + 64 * s + ln, 0.25 * fma(re[s], re[s], im[s] * im[s]));
			spsM = 0.25 * (nyq1 * nyq1);
		} else {
			double ar[16], ai[16], nyq2;
			{
				int l2 = lane;
				WC_FRESH(l2);  // (the weights below must not be hoisted out of the loop over the jobs and spilled)
#pragma unroll
				for (int q = 0; q < 16; ++q) {
					ar[q] = ai[q] = 0.0;
					if (q < 4 * ng) {
						ar[q] = L[kWfLds + 64 * q + l2];
						ai[q] = d4_ld(row, 2048 + 64 * q + l2);
					}
				}
				WF_SCHED_FENCE();
#pragma unroll
				for (int q = 0; q < 16; ++q) {
					const int i0 = 2 * l2 + 128 * q;
					ar[q] *= i0 + 1.0;
					ai[q] *= i0 + 2.0;
				}
			}
			wf_r2c4096_half(ar, ai, nyq2, ng, L, a.tw, ln, 0);
			double acc[16];
#pragma unroll
			for (int s = 0; s < 16; ++s) acc[s] = fma(re[s], ar[s], im[s] * ai[s]);
			if (job == 1) {
				double prev[16];
#pragma unroll
				for (int s = 0; s < 16; ++s) prev[s] = d4_ld(row, 64 * s + ln);
				WF_SCHED_FENCE();
#pragma unroll
				for (int s = 0; s < 16; ++s) acc[s] += prev[s];
			}
#pragma unroll
			for (int s = 0; s < 16; ++s) d4_st(row, 64 * s + ln, acc[s]);
			cenM += nyq1 * nyq2;
		}
	}
	D4Bins1 cen, sps;
#pragma unroll
	for (int s = 0; s < 16; ++s) {
		cen.v[s] = d4_ld(row, 64 * s + lane);
		sps.v[s] = d4_ld(row, 1024 + 64 * s + lane);
	}
	cen.vM = cenM;
	sps.vM = spsM;
	d4c1_dc_correction(cen, f0, fs, L, lane);
	d4c1_dc_correction(sps, f0, fs, L, lane);
	d4c1_smooth<true>(sps, f0, fs, L, lane);
	// ---- static group delay (reference :440-460) ----
#pragma unroll
	for (int s = 0; s < 16; ++s) cen.v[s] = cen.v[s] / sps.v[s];
	cen.vM = cen.vM / sps.vM;
	// smoothed over f0 / 2, minus that smoothed once more over f0 (one copy of the code, run twice)
#pragma unroll 1
	for (int it = 0; it < 2; ++it) {
		d4c1_smooth<false>(cen, it ? f0 : f0 / 2.0, fs, L, lane);
		if (it == 0) sps = cen;
	}
	// (sps: once smoothed; cen: twice)
#pragma unroll
	for (int gq = 0; gq < 4; ++gq) {
		const int j = wf_bin(lane, gq, 0);
#pragma unroll
		for (int q = 0; q < 4; ++q) d4_st(row, j + 256 * q, sps.v[4 * gq + q] - cen.v[4 * gq + q]);
	}
	if (lane == 0) d4_st(row, M, sps.vM - cen.vM);
}

// one wavefront per (gated frame, band) (reference :466-503), see d4c2_band_kernel: one transform, 17 keys per lane
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WC_D4C1_BAND_OCC, WC_D4C1_BAND_OCC))) void d4c1_band_kernel(D4cArgs a) {
	constexpr int N = 2048, M = 1024;
	__shared__ __attribute__((aligned(16))) double L[kWfLds];
	const int lane = threadIdx.x;
	const int n_ap = a.n_ap;
	const long long blk = xcd_frame(blockIdx.x, (long long)gridDim.x);
	const long long g = blk / n_ap;
	const int bnd = (int)(blk % n_ap);
	if (g >= a.total_frames) return;
	const double f0v = a.f0[g];
	if (f0v == 0.0 || a.ap0[g] <= a.threshold) return;
	const double f0 = fmax(47.0, f0v);
	const int fs = a.fs;
	const int wln = a.window_length, hwl = wln / 2;  // <= 1023 samples = 512 packed points: slots 0 .. 7
	const int boundary = mround(N * 8.0 / wln);
	const unsigned int K = (unsigned int)(M + 1 - boundary - 1);
	const int center = (int)(3000.0 * (bnd + 1) * N / fs);
	const double *__restrict__ src = a.sgd + g * a.sgd_stride + (center - hwl);
	const int ng = wln > 512 ? 2 : 1;
	double key[16], keyM = 0.0;
	{
		double re[16], im[16], nyq;
		{
			double sv[16], nv[16];
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				const int i = min(2 * lane + 128 * (k >> 1) + (k & 1), wln - 1);
				sv[k] = src[i];
				nv[k] = a.nuttall[i];
			}
			WF_SCHED_FENCE();
#pragma unroll
			for (int q = 0; q < 16; ++q) {
				re[q] = im[q] = 0.0;
				if (q < 8) {
					const int i0 = 2 * lane + 128 * q;
					re[q] = (i0 < wln) ? sv[2 * q] * nv[2 * q] : 0.0;
					im[q] = (i0 + 1 < wln) ? sv[2 * q + 1] * nv[2 * q + 1] : 0.0;
				}
			}
		}
		wf_r2c4096_half(re, im, nyq, ng, L, a.tw, lane, 0);  // 2 X: a common factor of all powers
#pragma unroll
		for (int s = 0; s < 16; ++s) key[s] = fma(re[s], re[s], im[s] * im[s]);
		keyM = nyq * nyq;  // lane 0
	}
	// largest key
	double mx = (lane == 0) ? keyM : 0.0;
#pragma unroll
	for (int s = 0; s < 16; ++s) mx = fmax(mx, key[s]);
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
	mx = uniform_d(mx);
	long long lo = -1, hi = __double_as_longlong(mx);
	unsigned int c_lo = 0;
	auto count_le = [&](long long t) {
		unsigned int c = (unsigned int)__popcll(__ballot(lane == 0 && __double_as_longlong(keyM) <= t));
#pragma unroll
		for (int s = 0; s < 16; ++s) c += (unsigned int)__popcll(__ballot(__double_as_longlong(key[s]) <= t));
		return c;
	};
	if (!a.select64) {
		// (the search on the keys' high words first: see d4c2_band_kernel)
		auto count_hi = [&](int h) {
			unsigned int c = (unsigned int)__popcll(__ballot(lane == 0 && __double2hiint(keyM) <= h));
#pragma unroll
			for (int s = 0; s < 16; ++s) c += (unsigned int)__popcll(__ballot(__double2hiint(key[s]) <= h));
			return c;
		};
		int lo_h = -1, hi_h = __double2hiint(mx);
#pragma unroll 1
		for (int tr = 0; tr < 2; ++tr) {
			const int cand = __double2hiint(mx * (tr == 0 ? WC_D4C2_BRACKET_FIRST : 0x1p-40)) - 1;
			if (cand < 0) continue;
			const unsigned int c = count_hi(cand);
			if (c <= K) { lo_h = cand; c_lo = c; break; }
			hi_h = cand;
		}
		while (hi_h - lo_h > 1 && c_lo != K) {
			const int mid = lo_h + ((hi_h - lo_h) >> 1);
			const unsigned int c = count_hi(mid);
			if (c >= K) hi_h = mid;
			if (c <= K) { lo_h = mid; c_lo = c; }
		}
		lo = lo_h < 0 ? -1ll : (((long long)lo_h << 32) | 0xFFFFFFFFll);
		hi = min(hi, ((long long)hi_h << 32) | 0xFFFFFFFFll);
	} else {
#pragma unroll 1
		for (int tr = 0; tr < 2; ++tr) {
			const long long cand = __double_as_longlong(mx * (tr == 0 ? WC_D4C2_BRACKET_FIRST : 0x1p-40));
			const unsigned int c = count_le(cand);
			if (c <= K && cand > 0) { lo = cand; c_lo = c; break; }
			hi = cand > 0 && c >= K ? cand : hi;  // (more than K below: the K-th smallest is at or below cand)
		}
	}
	for (int it = 0; it < 64 && hi - lo > 1 && c_lo != K; ++it) {
		const long long mid = lo + ((hi - lo) >> 1);
		const unsigned int c = count_le(mid);
		if (c >= K) hi = mid;
		if (c <= K) { lo = mid; c_lo = c; }
		if (c == K) break;
	}
	// sum of the K smallest = sum(keys <= lo) + (K - #{keys <= lo}) * (the key at hi), and the total
	double low = 0.0, tot = 0.0;
#pragma unroll
	for (int s = 0; s < 16; ++s) {
		tot += key[s];
		low += (__double_as_longlong(key[s]) <= lo) ? key[s] : 0.0;
	}
	if (lane == 0) {
		tot += keyM;
		low += (__double_as_longlong(keyM) <= lo) ? keyM : 0.0;
	}
	low = wave_sum_all(low);
	tot = wave_sum_all(tot);
	if (lane == 0) {
		const double part = low + (double)(K - c_lo) * __longlong_as_double(hi);
		const double cv = 10 * log10(part / tot);
		a.coarse[g * kMaxBands + bnd] = fmin(0.0, cv + (f0 - 100) / 50.0);  // reference :326-328
	}
}

}  // namespace wc

using namespace wc;

struct wc_d4c {
	int fs, fft_size_d4c, fft_size_lt, n_ap, window_length;
	double threshold;
	bool select64 = false;  // WC_D4C_SELECT=64 (D4cArgs::select64)
	bool split;  // band loop and row output as separate kernels (default; WC_D4C_SPLIT=0: one fused kernel)
	bool wave2;  // 4096-point transforms by two wavefronts per frame (d4c2_*; default where they apply, WC_D4C_IMPL=block: never)
	Device *dev;
	double f0_bound = 0.0;  // (a caller's promise about the contour's highest F0: no longer relied on -- the frames the wavefront kernels leave out are listed on the device)
	DevBuf nuttall, utts, cnt, uidx, long_list, rare_list, off, endpos, endpos2, ap0, sgd, coarse, d_x, d_tpos, d_f0, d_ap;
	HostBuf h_stage, h_rows, h_x;
};

template <int N>
static void launch_lt(const D4cArgs &a, hipStream_t s) {
	long long blocks = ((a.total_frames + 7) / 8) * 8;
	constexpr int TL = (N >= 8192) ? 1024 : (N / 8 < 64 ? 64 : N / 8);  // eight samples per thread (128 threads at N = 1024, the 8 kHz case: 0.48 -> 0.34 ms)
	hipLaunchKernelGGL((d4c_lovetrain_kernel<N, TL>), dim3((unsigned)blocks), dim3(TL), 0, s, a);
}
// part 0: frames kernel (fused, or up to the group delay when split); part 1: bands + rows of the split schedule
template <int N>
static void launch_main(const D4cArgs &a, hipStream_t s, bool split, int part) {
	long long blocks = ((a.total_frames + 7) / 8) * 8;
	// eight samples per thread: 8 waves per frame at N = 4096 (half the registers per thread, 2 WG/CU), 128 threads at N = 1024 (8 kHz: 3.27 -> 2.16 ms)
	constexpr int TF = (N >= 8192) ? 1024 : (N / 8 < 64 ? 64 : N / 8);
	if (part == 0) {
		if (!split) hipLaunchKernelGGL((d4c_frames_kernel<N, TF, false>), dim3((unsigned)blocks), dim3(TF), 0, s, a);
		else hipLaunchKernelGGL((d4c_frames_kernel<N, TF, true>), dim3((unsigned)blocks), dim3(TF), 0, s, a);
		return;
	}
	if (a.n_ap > 0) hipLaunchKernelGGL((d4c_band_kernel<N, TF>), dim3((unsigned)(blocks * a.n_ap)), dim3(TF), 0, s, a);
	hipLaunchKernelGGL(d4c_rows_kernel, dim3((unsigned)a.total_frames), dim3(256), 0, s, a);
}

// Enqueue-only (no host synchronisation), shared with the fused pipeline: everything on stream s; the stream
// positions start at rng_pos[u] (host, may be NULL = 0) or, when d_start is given, at the device array
// d_start[u] (e.g. CheapTrick's end positions); the end positions are left in d->endpos2 (device).
int d4c_enqueue(wc_d4c *d, hipStream_t s, int n_utt, const double *d_x, const int *x_length, const double *d_tpos,
				const double *d_f0, const int *f0_length, int fft_size, double *d_ap, const uint64_t *rng_pos,
				const unsigned long long *d_start) {
	Device *dev = d->dev;
	if (fft_size < 2 || (fft_size & 1)) return fail(WC_ERR_INVALID, "d4c: fft_size must be even and positive");
	std::vector<UttDesc> utts(n_utt);
	long long xo = 0, fo = 0;
	uint64_t min_pos = ~0ull, max_end = 0;
	const uint64_t lt_max = (uint64_t)(2 * (int)(3.0 * d->fs / 40.0 / 2.0 + 1.0) + 1);
	const uint64_t main_max = 3ull * (uint64_t)(2 * (int)(4.0 * d->fs / 47.0 / 2.0 + 1.0) + 1);
	for (int u = 0; u < n_utt; ++u) {
		if (x_length[u] <= 0 || f0_length[u] < 0) return fail(WC_ERR_INVALID, "d4c: non-positive length");
		UttDesc &t = utts[u];
		t.x_off = xo; t.f_off = fo; t.y_off = 0;
		t.x_len = x_length[u]; t.f_len = f0_length[u]; t.y_len = 0; t.pad = 0;
		t.rng_pos = rng_pos ? rng_pos[u] : 0ull;
		xo += x_length[u];
		fo += f0_length[u];
		if (t.rng_pos < min_pos) min_pos = t.rng_pos;
		uint64_t e = t.rng_pos + (lt_max + main_max) * (uint64_t)t.f_len;
		if (e > max_end) max_end = e;
	}
	const long long total = fo;
	(void)min_pos; (void)max_end;
	if (total == 0) return WC_OK;
	int rc;
	if ((rc = d->utts.reserve(sizeof(UttDesc) * n_utt))) return rc;
	if ((rc = d->cnt.reserve(sizeof(uint32_t) * total))) return rc;
	if ((rc = d->uidx.reserve(sizeof(int) * total))) return rc;
	if ((rc = d->long_list.reserve(sizeof(int) * (total + 1)))) return rc;  // [0]: the count
	if ((rc = d->rare_list.reserve(sizeof(int) * (total + 1)))) return rc;  // [0]: the count
	if ((rc = d->off.reserve(sizeof(uint64_t) * total))) return rc;
	if ((rc = d->ap0.reserve(sizeof(double) * total))) return rc;
	const bool split = d->split;
	// the one-wavefront kernels (d4c2_* at N = 4096, d4c1_* at N = 2048) park their intermediate results in the frame's row of the group-delay array
	const bool main2 = d->wave2 && split && (d->fft_size_d4c == 4096 || d->fft_size_d4c == 2048) && d->window_length <= 1023;
	const long long row2 = d->fft_size_d4c == 4096 ? kD4Row : kD4Row1;
	if (split) {
		if ((rc = d->sgd.reserve(sizeof(double) * (size_t)total * (main2 ? (size_t)row2 : (size_t)(d->fft_size_d4c / 2 + 1))))) return rc;
		if ((rc = d->coarse.reserve(sizeof(double) * (size_t)total * kMaxBands))) return rc;
	}
	if ((rc = d->endpos.reserve(sizeof(uint64_t) * n_utt))) return rc;
	if ((rc = d->endpos2.reserve(sizeof(uint64_t) * n_utt))) return rc;
	if ((rc = d->h_stage.reserve(sizeof(UttDesc) * n_utt + sizeof(uint64_t) * n_utt))) return rc;
	std::memcpy(d->h_stage.p, utts.data(), sizeof(UttDesc) * n_utt);
	WC_HIP(hipMemcpyAsync(d->utts.p, d->h_stage.p, sizeof(UttDesc) * n_utt, hipMemcpyHostToDevice, s));
	if ((rc = d->h_stage.mark(s))) return rc;
	const unsigned grid1 = (unsigned)((total + 255) / 256);
	hipLaunchKernelGGL(d4c_lt_count_kernel, dim3(grid1), dim3(256), 0, s, d_f0, total, d->fs, d->cnt.as<uint32_t>(), d->utts.as<UttDesc>(), n_utt,
					   d->uidx.as<int>(), d->long_list.as<int>(), d->rare_list.as<int>());
	hipLaunchKernelGGL(utt_scan_kernel, dim3(n_utt), dim3(256), 0, s, d->cnt.as<uint32_t>(), d->utts.as<UttDesc>(),
					   d_start, d->off.as<unsigned long long>(), d->endpos.as<unsigned long long>());
	D4cArgs a;
	a.x = d_x; a.utts = d->utts.as<UttDesc>(); a.n_utt = n_utt; a.tpos = d_tpos; a.f0 = d_f0;
	a.rng_off = d->off.as<unsigned long long>(); a.rng_table = dev->rng_table.as<uint32_t>(); a.rng_base = dev->rng_base;
	a.tw = dev->twiddle; a.ap = d_ap; a.ap0 = d->ap0.as<double>(); a.cnt = d->cnt.as<uint32_t>(); a.uidx = d->uidx.as<int>(); a.long_cnt = d->long_list.as<int>(); a.long_list = d->long_list.as<int>();
	a.sgd = d->sgd.as<double>(); a.coarse = d->coarse.as<double>();
	a.nuttall = d->nuttall.as<double>(); a.total_frames = total; a.fs = d->fs; a.fft_size_out = fft_size;
	a.n_ap = d->n_ap; a.window_length = d->window_length; a.threshold = d->threshold; a.select64 = d->select64 ? 1 : 0;
	a.rare_only = 0;
	a.rare_list = d->rare_list.as<int>();
	a.sgd_stride = main2 ? row2 : (long long)(d->fft_size_d4c / 2 + 1);
	const long long blocks8 = ((total + 7) / 8) * 8;
	if ((rc = dev->time_begin("d4c_lovetrain", s))) return rc;
	if (d->wave2 && d->fft_size_lt == 4096) {
		hipLaunchKernelGGL(d4c2_lovetrain_kernel<false>, dim3((unsigned)blocks8), dim3(64), 0, s, a);
		hipLaunchKernelGGL(d4c2_lovetrain_kernel<true>, dim3((unsigned)blocks8), dim3(64), 0, s, a);
	} else if (d->wave2 && d->fft_size_lt == 2048) {
		hipLaunchKernelGGL(d4c1_lovetrain_kernel, dim3((unsigned)blocks8), dim3(64), 0, s, a);
	}
	else switch (d->fft_size_lt) {
		case 1024: launch_lt<1024>(a, s); break;
		case 2048: launch_lt<2048>(a, s); break;
		case 4096: launch_lt<4096>(a, s); break;
		case 8192: launch_lt<8192>(a, s); break;
		default: return fail(WC_ERR_UNSUPPORTED, "d4c: unsupported LoveTrain FFT size (fs must be 8..96 kHz)");
	}
	WC_HIP(hipGetLastError());
	if ((rc = dev->time_end("d4c_lovetrain", s))) return rc;
	// offsets of the main pass start where the LoveTrain draws of the utterance end
	hipLaunchKernelGGL(utt_scan_kernel, dim3(n_utt), dim3(256), 0, s, d->cnt.as<uint32_t>(), d->utts.as<UttDesc>(),
					   d->endpos.as<unsigned long long>(), d->off.as<unsigned long long>(), d->endpos2.as<unsigned long long>());
	for (int part = 0; part < (split ? 2 : 1); ++part) {
		const char *name = part == 0 ? "d4c_frames" : "d4c_bands";
		if ((rc = dev->time_begin(name, s))) return rc;
		if (main2 && part == 0) {
			// one wavefront per frame; the frames it leaves out (F0 above ~1.4 kHz at 48 kHz, ~930 Hz at 16 kHz: d4c2_can / d4c1_can) are
			// listed and done by a small grid of the block kernel behind it (a contour out of Harvest lists none: the grid leaves at once)
			a.rare_only = 1;
			if (d->fft_size_d4c == 4096) {
				hipLaunchKernelGGL(d4c2_frames_kernel<false>, dim3((unsigned)blocks8), dim3(64), 0, s, a);
				hipLaunchKernelGGL(d4c2_frames_kernel<true>, dim3((unsigned)std::min<long long>(blocks8, 4096)), dim3(64), 0, s, a);
				hipLaunchKernelGGL((d4c_frames_kernel<4096, 512, true, true>), dim3(256), dim3(512), 0, s, a);
			} else {
				hipLaunchKernelGGL(d4c1_frames_kernel, dim3((unsigned)blocks8), dim3(64), 0, s, a);
				hipLaunchKernelGGL((d4c_frames_kernel<2048, 256, true, true>), dim3(256), dim3(256), 0, s, a);
			}
			a.rare_only = 0;
		} else if (main2 && part == 1) {
			if (a.n_ap > 0) {
				if (d->fft_size_d4c == 4096) hipLaunchKernelGGL(d4c2_band_kernel, dim3((unsigned)(blocks8 * a.n_ap)), dim3(64), 0, s, a);
				else hipLaunchKernelGGL(d4c1_band_kernel, dim3((unsigned)(blocks8 * a.n_ap)), dim3(64), 0, s, a);
			}
			hipLaunchKernelGGL(d4c_rows_kernel, dim3((unsigned)a.total_frames), dim3(256), 0, s, a);
		} else
		switch (d->fft_size_d4c) {
			case 1024: launch_main<1024>(a, s, split, part); break;
			case 2048: launch_main<2048>(a, s, split, part); break;
			case 4096: launch_main<4096>(a, s, split, part); break;
			case 8192: launch_main<8192>(a, s, split, part); break;
			default: return fail(WC_ERR_UNSUPPORTED, "d4c: unsupported FFT size (fs must be 8..96 kHz)");
		}
		WC_HIP(hipGetLastError());
		if ((rc = dev->time_end(name, s))) return rc;
	}
#if WC_D4C2_TRACE
	if (const char *path = getenv("WC_D4C_TRACE")) {
		std::vector<double> row((size_t)kD4Row);
		std::vector<unsigned long long> out;
		WC_HIP(hipStreamSynchronize(s));
		for (long long g = 0; g < total; g += 7) {
			WC_HIP(hipMemcpy(row.data(), d->sgd.as<double>() + g * kD4Row, sizeof(double) * kD4Row, hipMemcpyDeviceToHost));
			const unsigned long long *st = reinterpret_cast<const unsigned long long *>(row.data() + 4128);
			out.insert(out.end(), st, st + 16);
		}
		if (FILE *f = fopen(path, "wb")) { fwrite(out.data(), 8, out.size(), f); fclose(f); }
	}
#endif
	return WC_OK;
}

void d4c_set_f0_bound(wc_d4c *d, double f0_bound) { d->f0_bound = f0_bound; }
const unsigned long long *d4c_end_positions(const wc_d4c *d) { return d->endpos2.as<unsigned long long>(); }

// upper bound of the stream positions one utterance can consume (LoveTrain + three windows per frame)
uint64_t d4c_draw_bound(const wc_d4c *d, int f0_length) {
	const uint64_t lt_max = (uint64_t)(2 * (int)(3.0 * d->fs / 40.0 / 2.0 + 1.0) + 1);
	const uint64_t main_max = 3ull * (uint64_t)(2 * (int)(4.0 * d->fs / 47.0 / 2.0 + 1.0) + 1);
	return (lt_max + main_max) * (uint64_t)(f0_length > 0 ? f0_length : 0);
}

static int d4c_run_device(wc_d4c *d, int n_utt, const double *d_x, const int *x_length, const double *d_tpos,
						  const double *d_f0, const int *f0_length, int fft_size, double *d_ap, uint64_t *rng_pos) {
	Device *dev = d->dev;
	hipStream_t s = dev->active();
	int rc;
	uint64_t lo = ~0ull, hi = 0;
	long long total = 0;
	for (int u = 0; u < n_utt; ++u) {
		uint64_t p0 = rng_pos ? rng_pos[u] : 0ull;
		lo = p0 < lo ? p0 : lo;
		uint64_t e = p0 + d4c_draw_bound(d, f0_length[u]);
		hi = e > hi ? e : hi;
		total += f0_length[u] > 0 ? f0_length[u] : 0;
	}
	if ((rc = dev->ensure_rng(lo, hi))) return rc;
	if ((rc = d4c_enqueue(d, s, n_utt, d_x, x_length, d_tpos, d_f0, f0_length, fft_size, d_ap, rng_pos, nullptr))) return rc;
	if (rng_pos && total > 0) {
		std::vector<uint64_t> h_end(n_utt);
		WC_HIP(hipMemcpyAsync(h_end.data(), d->endpos2.p, sizeof(uint64_t) * n_utt, hipMemcpyDeviceToHost, s));
		WC_HIP(hipStreamSynchronize(s));
		for (int u = 0; u < n_utt; ++u) rng_pos[u] = h_end[u];
	}
	return WC_OK;
}

wc::Device *d4c_device(const wc_d4c *d) { return d->dev; }

extern "C" {

wc_d4c *wc_d4c_create(int fs, double threshold) {
	if (fs <= 0) { set_error("d4c: fs must be positive"); return nullptr; }
	Device *dev = current_device();
	if (!dev) return nullptr;
	wc_d4c *d = new wc_d4c();
	d->fs = fs;
	d->threshold = threshold;
	d->dev = dev;
	{
		const char *sp = getenv("WC_D4C_SPLIT");  // default: split schedule; WC_D4C_SPLIT=0 runs the single fused kernel
		d->split = !(sp && sp[0] == '0');
		const char *sel = getenv("WC_D4C_SELECT");
		d->select64 = sel && std::strcmp(sel, "64") == 0;
		const char *impl = getenv("WC_D4C_IMPL");
		d->wave2 = !(impl && std::strcmp(impl, "block") == 0);
	}
	// reference src/d4c.cpp:60-111
	d->fft_size_d4c = static_cast<int>(std::pow(2.0, 1.0 + static_cast<int>(std::log(4.0 * fs / 47.0 + 1) / 0.69314718055994529)));
	d->n_ap = static_cast<int>(std::fmin(15000.0, fs / 2.0 - 3000.0) / 3000.0);
	d->window_length = static_cast<int>(3000.0 * d->fft_size_d4c / fs) * 2 + 1;
	d->fft_size_lt = static_cast<int>(std::pow(2.0, 1.0 + static_cast<int>(std::log(3.0 * fs / 40.0 + 1) / 0.69314718055994529)));
	if (d->n_ap < 0) d->n_ap = 0;
	if (d->n_ap > kMaxBands || (d->fft_size_d4c != 1024 && d->fft_size_d4c != 2048 && d->fft_size_d4c != 4096 && d->fft_size_d4c != 8192) ||
		(d->fft_size_lt != 1024 && d->fft_size_lt != 2048 && d->fft_size_lt != 4096 && d->fft_size_lt != 8192)) {
		set_error("d4c: unsupported sampling rate (supported: 8 kHz .. 96 kHz)");
		delete d;
		return nullptr;
	}
	std::vector<double> win(d->window_length);
	for (int i = 0; i < d->window_length; ++i) {  // NuttallWindow, reference src/world_common.cpp:118-126
		double t = i / (d->window_length - 1.0);
		win[i] = 0.355768 - 0.487396 * std::cos(2.0 * 3.1415926535897932384 * t) +
				 0.144232 * std::cos(4.0 * 3.1415926535897932384 * t) - 0.012604 * std::cos(6.0 * 3.1415926535897932384 * t);
	}
	if (d->nuttall.reserve(sizeof(double) * win.size()) ||
		hipMemcpy(d->nuttall.p, win.data(), sizeof(double) * win.size(), hipMemcpyHostToDevice) != hipSuccess) {
		set_error("d4c: window upload failed");
		delete d;
		return nullptr;
	}
	dev->handle_born();
	return d;
}
void wc_d4c_destroy(wc_d4c *d) {
	if (!d) return;
	d->dev->quiesce();
	d->dev->handle_gone();
	d->nuttall.release(); d->utts.release(); d->cnt.release(); d->uidx.release(); d->long_list.release(); d->rare_list.release(); d->off.release(); d->endpos.release(); d->endpos2.release();
	d->ap0.release(); d->sgd.release(); d->coarse.release(); d->d_x.release(); d->d_tpos.release(); d->d_f0.release(); d->d_ap.release(); d->h_stage.release(); d->h_rows.release(); d->h_x.release();
	delete d;
}

int wc_d4c_compute_device(wc_d4c *d, int n_utt, const double *d_x, const int *x_length, const double *d_tpos,
						  const double *d_f0, const int *f0_length, int fft_size, double *d_ap, uint64_t *rng_pos) {
	if (!d || n_utt <= 0 || !d_x || !x_length || !d_tpos || !d_f0 || !f0_length || !d_ap)
		return fail(WC_ERR_INVALID, "d4c: null argument");
	WC_HIP(hipSetDevice(d->dev->id));
	DeviceLock lock(d->dev);
	return d4c_run_device(d, n_utt, d_x, x_length, d_tpos, d_f0, f0_length, fft_size, d_ap, rng_pos);
}

int wc_d4c_compute(wc_d4c *d, const double *x, int x_length, const double *temporal_positions, const double *f0,
				   int f0_length, int fft_size, double **aperiodicity) {
	if (!d || !x || !temporal_positions || !f0 || !aperiodicity) return fail(WC_ERR_INVALID, "d4c: null argument");
	if (x_length <= 0 || f0_length < 0) return fail(WC_ERR_INVALID, "d4c: bad length");
	if (f0_length == 0) return WC_OK;
	WC_HIP(hipSetDevice(d->dev->id));
	DeviceLock lock(d->dev);
	hipStream_t s = d->dev->active();
	const int bins = fft_size / 2 + 1;
	int rc;
	if ((rc = d->d_x.reserve(sizeof(double) * x_length))) return rc;
	if ((rc = d->d_tpos.reserve(sizeof(double) * f0_length))) return rc;
	if ((rc = d->d_f0.reserve(sizeof(double) * f0_length))) return rc;
	if ((rc = d->d_ap.reserve(sizeof(double) * (size_t)f0_length * bins))) return rc;
	if ((rc = array_up(s, x, (size_t)x_length, d->h_x, d->d_x.as<double>()))) return rc;
	WC_HIP(hipMemcpyAsync(d->d_tpos.p, temporal_positions, sizeof(double) * f0_length, hipMemcpyHostToDevice, s));
	WC_HIP(hipMemcpyAsync(d->d_f0.p, f0, sizeof(double) * f0_length, hipMemcpyHostToDevice, s));
	uint64_t pos = global_rng_position();
	rc = d4c_run_device(d, 1, d->d_x.as<double>(), &x_length, d->d_tpos.as<double>(), d->d_f0.as<double>(), &f0_length,
						fft_size, d->d_ap.as<double>(), &pos);
	if (rc) return rc;
	set_global_rng_position(pos);
	const size_t n_ap = (size_t)f0_length * bins;  // (page-locked staging, rows handed over by a few threads: see wc_cheaptrick_compute)
	if ((rc = d->h_rows.reserve(sizeof(double) * n_ap))) return rc;
	return rows_down(s, aperiodicity, f0_length, bins, d->d_ap.as<double>(), d->h_rows.as<double>());
}

}  // extern "C"
