// CheapTrick spectral-envelope estimation on gfx950: one workgroup per analysis frame, the whole
// frame resident in LDS from the windowed gather to the final exp().
//
// Restates reference src/cheaptrick.cpp:48-276 (compute, generalBody, getWindowedWaveform,
// getPowerSpectrum, addInfinitesimalNoise, smoothingWithRecovery) with DCCorrection and
// LinearSmoothing of reference src/world_common.cpp:27-116.  Differences from the reference are
// confined to summation order (block reductions / block scan instead of sequential loops), our own
// FFT, and device libm; the noise draws come from the exact stream positions the reference's serial
// order would use (see wc_core.hip).
#include <cmath>
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

struct CtArgs {
	const double *x;
	const UttDesc *utts;
	int n_utt;
	const double *tpos, *f0;
	const unsigned long long *rng_off;  // absolute stream position of each frame's first draw
	const uint32_t *rng_table;
	unsigned long long rng_base;
	const double2 *tw;
	double *sp;
	long long total_frames;
	int fs;
	double q1, f0_floor;  // f0_floor = 3 fs / (N - 3)
	const int *uidx;      // utterance of every frame (ct_count_kernel)
	int *rare_list;       // [0]: count, then the frames the one-wavefront kernels leave to the block kernel behind them (ct_wave_can)
};

// per-frame number of draws: window (2 hw + 1) then one per bin (reference :153, :227)
__global__ void ct_count_kernel(const double *__restrict__ f0, long long total, int fs, double f0_floor,
								int bins, uint32_t *__restrict__ cnt, const UttDesc *__restrict__ utts, int n_utt, int *__restrict__ uidx,
								int *__restrict__ rare_list) {
	long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= total) return;
	if (g == 0) rare_list[0] = 0;
	uidx[g] = find_utt(utts, n_utt, g);  // (looked up once here rather than by every frame's wavefront, eight dependent loads each)
	double f = f0[g];
	double f0c = (f <= f0_floor) ? 500.0 : f;
	cnt[g] = (uint32_t)(2 * mround(1.5 * fs / f0c) + 1 + bins);
}

// Which frames the one-wavefront kernel (ct_wave_kernel, N = 2048) takes: its 9 KB of LDS hold the mirrored segment of the
// smoothing (1025 + 2 b + 1 terms, b = half width in bins) and the low bins of the DC correction only for F0 below ~2 kHz at
// 48 kHz -- every contour Harvest can produce (ceiling 800 Hz) and anything a caller could mean by a pitch.  Frames above
// that are listed (rare_list) and done by a small grid of the block kernel launched behind it.
template <int N>
__host__ __device__ __forceinline__ bool ct_wave_can(double f0c, int fs) {
	const int b = (int)(f0c * 2.0 / 3.0 * N / fs) + 1;
	const int upper = 2 + (int)(f0c * N / fs);
	return b <= 60 && upper <= 120;
}

template <int N, int T>
__device__ __forceinline__ void ct_frame_block(const CtArgs &a, long long g, double2 *A, double *scr, double *red) {
	constexpr int M = N / 2;
	constexpr int EPT = (N + T - 1) / T;       // window elements per thread
	constexpr int BPT = (M + 1 + T - 1) / T;   // bins per thread
	double *Ar = reinterpret_cast<double *>(A);
	int tid = threadIdx.x;
	const int u = find_utt(a.utts, a.n_utt, g);
	const UttDesc ud = a.utts[u];
	const double *__restrict__ x = a.x + ud.x_off;
	const int fs = a.fs;
	const double f0v = a.f0[g];
	const double f0c = (f0v <= a.f0_floor) ? 500.0 : f0v;  // reference :77
	const double pos = a.tpos[g];
	const unsigned long long roff = a.rng_off[g] - a.rng_base;

	// ---- F0-adaptive window (reference :137-196) ----
	const int hw = mround(1.5 * fs / f0c);
	const int wl = 2 * hw + 1;
	const int origin = mround(pos * fs + 0.001);
	// The window phase kappa * (i - hw) of this thread's samples i = tid + e T advances by a rotation recurrence
	// from one exact sincos (EPT steps: a few 1e-16 of drift) instead of EPT libm cosines.
	double w[EPT];
	double ssq = 0.0;
	{
		const double kappa = f0c / 1.5 / fs;  // angle per sample in units of pi: pi * ((i - hw) / 1.5 / fs) * f0c
		double c, sn, cd, sd;
		sincospi(kappa * (tid - hw), &sn, &c);
		sincospi(kappa * T, &sd, &cd);
#pragma unroll
		for (int e = 0; e < EPT; ++e) {
			int i = tid + e * T;
			w[e] = 0.0;
			if (i < wl) {
				w[e] = fma(0.5, c, 0.5);
				ssq = fma(w[e], w[e], ssq);
			}
			const double cn = fma(c, cd, -(sn * sd));
			sn = fma(sn, cd, c * sd);
			c = cn;
		}
	}
	ssq = block_sum<T>(ssq, red, tid);
	WC_FRESH(tid);
	const double rnorm = 1.0 / sqrt(ssq);  // (the reference divides each window sample by the norm: an ulp apart)
	double s1 = 0.0, s2 = 0.0;
	double wv[EPT];
#pragma unroll
	for (int e = 0; e < EPT; ++e) {
		int i = tid + e * T;
		wv[e] = 0.0;
		if (i < wl) {
			w[e] = w[e] * rnorm;
			int si = clampi(origin + i - hw, 0, ud.x_len - 1);
			wv[e] = fma(x[si], w[e], randn_at(a.rng_table, roff + i) * 0.000000000000001);
			s1 += wv[e];
			s2 += w[e];
		}
	}
	block_sum2<T>(s1, s2, red, tid);
	const double wc = s1 / s2;
#pragma unroll
	for (int e = 0; e < EPT; ++e) {
		int i = tid + e * T;
		if (i < N) Ar[i] = (i < wl) ? fma(-w[e], wc, wv[e]) : 0.0;
	}
	__syncthreads();

	// ---- power spectrum (reference :198-218) ----
	WC_FRESH(tid);
	fft_lds<M, T, +1>(A, a.tw, tid);
	r2c_post<M, T>(A, a.tw, tid);
	double pw[BPT];
#pragma unroll
	for (int e = 0; e < BPT; ++e) {
		const int k = tid + e * T;
		pw[e] = 0.0;
		if (k <= M) {
			const double2 v = A[k == M ? 0 : k];
			pw[e] = (k == 0) ? v.x * v.x : (k == M) ? v.y * v.y : fma(v.x, v.x, v.y * v.y);
		}
	}
	__syncthreads();  // the spectrum has been read: the workspace becomes P[0 .. M]
#pragma unroll
	for (int e = 0; e < BPT; ++e) {
		const int k = tid + e * T;
		if (k <= M) Ar[k] = pw[e];
	}
	__syncthreads();
	// DC correction (reference src/world_common.cpp:61-80)
	WC_FRESH(tid);
	{
		double *P = Ar;
		const int upper = 2 + (int)(f0c * N / fs);
		const double dx = -(double)fs / N, rdx = 1.0 / dx;
		double rep[2];
#pragma unroll
		for (int e = 0; e < 2; ++e) {
			int i = tid + e * T;
			rep[e] = 0.0;
			if (i < upper - 1 && i <= M) {
				double axis = (double)i * fs / N;
				rep[e] = interp1q_rcp(f0c, dx, rdx, [&](int b) { return P[min(max(b, 0), M)]; }, upper + 1, axis);
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
#pragma unroll
	for (int e = 0; e < BPT; ++e) {
		const int k = tid + e * T;
		if (k <= M) pw[e] = Ar[k];
	}
	__syncthreads();  // the corrected spectrum is in registers: the workspace becomes the mirrored segment

	// ---- linear smoothing, width 2 f0 / 3 (reference src/world_common.cpp:27-52, :82-116) ----
	double lp[BPT];
	WC_FRESH(tid);
	{
		const double width = f0c * 2.0 / 3.0;
		int b = (int)(width * N / fs) + 1;
		if (M + 2 * b + 1 > N) b = (N - M - 1) / 2;  // cannot happen for f0 < 3 fs / 8
		const int len = M + 2 * b + 1;
		// mirrored segment (reference src/world_common.cpp:33-44): position i holds bin b - i (i < b), bin i - b (b <= i < M + b),
		// bin 2 M + b - i (M + b <= i <= M + 2 b); every thread scatters the terms of its own bins.  The cumulative sum is then
		// formed in the reference's own sequential rounding (seq_cumsum_nonneg, wc_device.hpp) -- non-decreasing like the
		// reference's, which matters because the smoothed value below is a difference of two neighbourhoods of it and goes into
		// a logarithm
#pragma unroll
		for (int e = 0; e < BPT; ++e) {
			const int k = tid + e * T;
			if (k <= M) {
				const double v = pw[e] * fs / N;
				if (k < M) Ar[k + b] = v;
				if (k >= 1 && k <= b) Ar[b - k] = v;
				if (k >= M - b) Ar[2 * M + b - k] = v;
			}
		}
		__syncthreads();
		seq_cumsum_nonneg<T>(Ar, len, scr, red, tid);
		const double origin_axis = -(b - 0.5) * fs / N;
		const double step = (double)fs / N, rstep = 1.0 / step;
		auto seg = [&](int i) -> double { return Ar[min(max(i, 0), len - 1)]; };
#pragma unroll
		for (int e = 0; e < BPT; ++e) {
			int k = tid + e * T;
			lp[e] = 0.0;
			if (k <= M) {
				double lo_axis = (double)k / N * fs - width / 2.0;
				double hi_axis = lo_axis + width;
				double lo_v = interp1q_rcp(origin_axis, step, rstep, seg, len, lo_axis);
				double hi_v = interp1q_rcp(origin_axis, step, rstep, seg, len, hi_axis);
				double sm = (hi_v - lo_v) / width;
				// infinitesimal noise (reference :220-228) then log (reference :251-252)
				sm = fma(fabs(randn_at(a.rng_table, roff + wl + k)), 0.00000000000000022204460492503131, sm);
				lp[e] = log(sm);
			}
		}
		__syncthreads();  // all reads of the segment are done; Ar becomes the mirrored log spectrum
	}
#pragma unroll
	for (int e = 0; e < BPT; ++e) {
		int k = tid + e * T;
		if (k <= M) {
			Ar[k] = lp[e];
			if (k > 0 && k < M) Ar[N - k] = lp[e];
		}
	}
	__syncthreads();

	// ---- smoothing + recovery lifters in the cepstral domain (reference :230-276) ----
	WC_FRESH(tid);
	fft_lds<M, T, +1>(A, a.tw, tid);
	r2c_post<M, T>(A, a.tw, tid);
	{
		const double q1 = a.q1;
		// lifters at quefrency k / fs: sinc(f0 q) and 1 - 2 q1 + 2 q1 cos(2 pi q f0); with a = pi f0 k / fs the
		// cosine is 1 - 2 sin^2 a, and (cos a, sin a) advances over this thread's bins k = tid + e T by rotation
		const double alpha = kPi * f0c / fs;
		double c, sn, cd, sd;
		sincospi(f0c / fs * tid, &sn, &c);
		sincospi(f0c / fs * T, &sd, &cd);
		// every thread reads and rewrites its own bins only; bin M travels in the imaginary slot of bin 0, and both are thread 0's
		double bin0 = 0.0, binM = 0.0;
		for (int k = tid; k <= M; k += T) {
			double sl, cl;
			if (k == 0) {
				sl = 1.0;
				cl = (1.0 - 2.0 * q1) + 2.0 * q1;
			} else {
				sl = sn / (alpha * k);
				cl = fma(2.0 * q1, fma(-2.0 * sn, sn, 1.0), 1.0 - 2.0 * q1);
			}
			if (k == M) {
				binM = A[0].y * sl * cl / N;
			} else {
				const double re = (k == 0) ? A[0].x : A[k].x;
				const double v = re * sl * cl / N;
				if (k == 0) bin0 = v;
				else A[k] = make_double2(v, 0.0);
			}
			const double cn = fma(c, cd, -(sn * sd));
			sn = fma(sn, cd, c * sd);
			c = cn;
		}
		if (tid == 0) A[0] = make_double2(bin0, binM);
		__syncthreads();
	}
	WC_FRESH(tid);
	c2r_pre<M, T>(A, a.tw, tid);
	fft_lds<M, T, -1>(A, a.tw, tid);
	WC_FRESH(tid);
	double *__restrict__ out = a.sp + g * (long long)(M + 1);
	for (int k = tid; k <= M; k += T) out[k] = exp(Ar[k]);
}

template <int N, int T, bool RARE = false>
__global__ __launch_bounds__(T) void ct_frames_kernel(CtArgs a) {
	constexpr int M = N / 2;
	// LDS: the FFT workspace and the scratch of the cumulative sum only (18.3 KB at N = 2048: eight workgroups per CU).  The power
	// spectrum has no array of its own: it lives in the workspace until it has been DC-corrected, goes through registers and
	// comes back as the mirrored terms of the cumulative sum.
	__shared__ double2 A[fft_lds_size(M)];
	__shared__ double scr[T + 2 * (T / 64)];
	__shared__ double red[2 * (T / 64) + 2];
	if constexpr (RARE) {
		// the frames the one-wavefront kernel in front of this one has listed (none for a contour out of Harvest)
		const int n = a.rare_list[0];
#pragma unroll 1
		for (int i = blockIdx.x; i < n; i += gridDim.x) {
			ct_frame_block<N, T>(a, a.rare_list[1 + i], A, scr, red);
			__syncthreads();
		}
	} else {
		const long long g = xcd_frame(blockIdx.x, a.total_frames);
		if (g >= a.total_frames) return;
		ct_frame_block<N, T>(a, g, A, scr, red);
	}
}


// ---- one wavefront per frame (N = 2048: 48 kHz) ------------------------------------------------------------------------
// The same arithmetic as ct_frame_block with the frame in the registers of ONE wavefront from the windowed gather to the
// final exp(): 16 complex points per lane, the three transforms by wc_wavefft.hpp (two LDS exchanges each, no barrier), the
// spectrum in the "paired" layout so that the real-transform unpacking, the power spectrum, the lifters and everything else
// per bin stay in the lane.  LDS: 9.6 KB (exchange buffer = mirrored segment of the smoothing, scratch of the cumulative
// sum), 16 wavefronts per CU.  What is done differently from the block kernel, none of it above 1e-15 relative:
//   * the halving of the real-transform unpacking is folded into the window norm / the lifter scale (exact);
//   * (hi - lo) * (1 / width), sin / (alpha k) as a product with tabulated 1 / k, the lean log / exp of wc_wavefft.hpp.
// The FFT of the log spectrum is real, so only real parts are unpacked (wf_r2c_unpack_re) and the liftered spectrum goes
// back through the real-spectrum packing (wf_c2r_pack_re).
#ifndef WC_CT_WAVE_OCC
#define WC_CT_WAVE_OCC 2
#endif
#ifndef WC_CT_PRELOAD
#define WC_CT_PRELOAD 1
#endif
#ifndef WC_CT_EVEN
#define WC_CT_EVEN 1  // the two cepstral transforms as real even transforms (wf_even2048); 0: 1024-point complex transforms (A/B)
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WC_CT_WAVE_OCC, WC_CT_WAVE_OCC))) void ct_wave_kernel(CtArgs a) {
	constexpr int N = 2048, M = 1024;
	__shared__ __attribute__((aligned(16))) double L[kWfLds];
	__shared__ __attribute__((aligned(16))) double T[kWfTabLds];  // the tables of the lean log / exp
	__shared__ __attribute__((aligned(16))) double P[1026];       // the log spectrum, bins 0 .. 1024
	const int lane = threadIdx.x;
	const long long g = xcd_frame(blockIdx.x, a.total_frames);
	if (g >= a.total_frames) return;
	const int fs = a.fs;
	const double f0v = a.f0[g];
	const double f0c = uniform_d((f0v <= a.f0_floor) ? 500.0 : f0v);  // reference :77
	if (!ct_wave_can<N>(f0c, fs)) {  // left for the block kernel behind this one
		if (lane == 0) a.rare_list[1 + atomicAdd(a.rare_list, 1)] = (int)g;
		return;
	}
	const UttDesc ud = a.utts[a.uidx[g]];
	const double *__restrict__ x = a.x + ud.x_off;
	const int x_last = ud.x_len - 1;
	const double pos = a.tpos[g];
	const uint32_t *__restrict__ rng = a.rng_table + (a.rng_off[g] - a.rng_base);
	wf_tables_to_lds(T, a.tw, lane);

	// ---- F0-adaptive window (reference :137-196): window sample i of slot q is 2 lane + 128 q (+ 1) ----
	const int hw = __builtin_amdgcn_readfirstlane(mround(1.5 * fs / f0c));
	const int wl = 2 * hw + 1;
	const int base = __builtin_amdgcn_readfirstlane(mround(pos * fs + 0.001)) - hw;  // signal index of window sample 0
	double ce0, se0, co0, so0, cd, sd;
	{
		const double kappa = f0c / 1.5 / fs;  // angle per sample in units of pi
		wf_sincospi(kappa * (2 * lane - hw), se0, ce0);
		wf_sincospi(kappa * (2 * lane + 1 - hw), so0, co0);
		wf_sincospi(kappa * 128.0, sd, cd);
		sd = uniform_d(sd);
		cd = uniform_d(cd);
	}
	// walks the live slots (whole groups of four slots beyond the window are skipped: uniform branches) with the raw window
	// values of the slot's two samples, advanced by a rotation recurrence from the exact start phases.  LOADS = 1: the group's
	// sixteen loads (signal and draws) are issued together and fenced off from their uses, so that they are in flight at once
	// (left alone, the scheduler waits for every one of them in turn).
#if WC_CT_PRELOAD
	// every live sample and draw of the window requested now: the requests are in flight while the first walk below (the window's
	// norm: arithmetic only) runs, instead of four dependent round trips in front of the second walk's four groups
	double xs_all[32];
	uint32_t ns_all[32];
#pragma unroll
	for (int qg = 0; qg < 16; qg += 4) {
		if (qg * 128 < wl) {
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				const int i = 2 * lane + 128 * (qg + (k >> 1)) + (k & 1);
				xs_all[2 * qg + k] = x[clampi(base + i, 0, x_last)];
				ns_all[2 * qg + k] = rng[i < wl ? i : 0];
			}
		}
	}
	WF_SCHED_FENCE();
#endif
	auto walk = [&](auto loads_c, auto &&body) {
		constexpr int LOADS = decltype(loads_c)::value;
		double ce = ce0, se = se0, co = co0, so = so0;
#pragma unroll
		for (int qg = 0; qg < 16; qg += 4) {
			if (qg * 128 >= wl) break;
			double xs[8];
			uint32_t ns[8];
			if (LOADS) {
#if WC_CT_PRELOAD
#pragma unroll
				for (int k = 0; k < 8; ++k) { xs[k] = xs_all[2 * qg + k]; ns[k] = ns_all[2 * qg + k]; }
#else
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					const int i = 2 * lane + 128 * (qg + (k >> 1)) + (k & 1);
					xs[k] = x[clampi(base + i, 0, x_last)];
					ns[k] = rng[i < wl ? i : 0];
				}
				WF_SCHED_FENCE();
#endif
			}
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int q = qg + k, i0 = 2 * lane + 128 * q;
				const double We = (i0 < wl) ? fma(0.5, ce, 0.5) : 0.0;
				const double Wo = (i0 + 1 < wl) ? fma(0.5, co, 0.5) : 0.0;
				body(q, i0, We, Wo, LOADS ? xs[2 * k] : 0.0, LOADS ? xs[2 * k + 1] : 0.0, LOADS ? ns[2 * k] : 0u, LOADS ? ns[2 * k + 1] : 0u);
				const double cen = fma(ce, cd, -(se * sd)), con = fma(co, cd, -(so * sd));
				se = fma(se, cd, ce * sd);
				so = fma(so, cd, co * sd);
				ce = cen;
				co = con;
			}
		}
	};
	double re[16], im[16];
#pragma unroll
	for (int q = 0; q < 16; ++q) re[q] = im[q] = 0.0;
	double ssq = 0.0;
	walk(std::integral_constant<int, 0>(), [&](int, int, double We, double Wo, double, double, uint32_t, uint32_t) { ssq = fma(Wo, Wo, fma(We, We, ssq)); });
	ssq = wave_sum_all(ssq);
	const double hr = 0.5 * (1.0 / sqrt(ssq));  // half the window norm: the transform below then yields X, not 2 X
	double s1 = 0.0, s2 = 0.0;
	walk(std::integral_constant<int, 1>(), [&](int q, int i0, double We, double Wo, double xe, double xo, uint32_t re_, uint32_t ro_) {
		const bool le = i0 < wl, lo = i0 + 1 < wl;
		const double we = We * hr, wo = Wo * hr;
		const double ne = (re_ / 268435456.0 - 6.0) * (0.5 * 0.000000000000001);
		const double no = (ro_ / 268435456.0 - 6.0) * (0.5 * 0.000000000000001);
		re[q] = le ? fma(xe, we, ne) : 0.0;
		im[q] = lo ? fma(xo, wo, no) : 0.0;
		s1 += re[q] + im[q];
		s2 += we + wo;
	});
	s1 = wave_sum_all(s1);
	s2 = wave_sum_all(s2);
	const double wc = s1 / s2;
	walk(std::integral_constant<int, 0>(), [&](int q, int, double We, double Wo, double, double, uint32_t, uint32_t) {
		re[q] = fma(-(We * hr), wc, re[q]);
		im[q] = fma(-(Wo * hr), wc, im[q]);
	});

	// ---- power spectrum (reference :198-218) ----
	if (wl <= 512) wdft16<+1, 1>(re, im);
	else if (wl <= 1024) wdft16<+1, 2>(re, im);
	else wdft16<+1, 4>(re, im);
	wf_fft1024_dit_rest<+1>(re, im, L, a.tw, lane);
	double pw[16], pwM;
	{
		double nyq;
		wf_r2c_unpack(re, im, nyq, a.tw, lane);
#pragma unroll
		for (int s = 0; s < 16; ++s) pw[s] = fma(re[s], re[s], im[s] * im[s]);
		pwM = nyq * nyq;
	}
	// DC correction (reference src/world_common.cpp:61-80): bins below upper - 1 <= 119, i.e. slots A_0 (bin lane) and C_0
	// (bin 64 + lane), from bins <= upper + 1
	{
		const int upper = __builtin_amdgcn_readfirstlane(2 + (int)(f0c * N / fs));
		const double dx = -(double)fs / N, rdx = 1.0 / dx;
		L[lane] = pw[0];
		L[64 + lane] = pw[8];
		wf_fence();
		auto rep = [&](int i) {
			const double axis = (double)i * fs / N;
			return interp1q_rcp(f0c, dx, rdx, [&](int b) { return L[min(max(b, 0), 127)]; }, upper + 1, axis);
		};
		if (lane < upper - 1) pw[0] += rep(lane);
		if (upper - 1 > 64) {
			if (64 + lane < upper - 1) pw[8] += rep(64 + lane);
		}
		wf_fence();
	}

	// ---- linear smoothing, width 2 f0 / 3 (reference src/world_common.cpp:27-52, :82-116), infinitesimal noise, log ----
	int jg[4];
#pragma unroll
	for (int gq = 0; gq < 4; ++gq) jg[gq] = wf_bin(lane, gq, 0);
	{
		const double width = f0c * 2.0 / 3.0;
		const int b = __builtin_amdgcn_readfirstlane((int)(width * N / fs) + 1);  // <= 60 (ct_wave_can)
		const int len = M + 2 * b + 1;
		// mirrored segment (reference src/world_common.cpp:33-44): position i holds bin b - i (i < b), bin i - b (b <= i < M + b),
		// bin 2 M + b - i (M + b <= i <= M + 2 b).  The low mirror comes from slot A_0 (bins 1 .. b of lanes 1 .. b), the high one
		// from slot B_3 (bins 1024 - lane); bin 1024 itself sits at M + b.
#pragma unroll
		for (int gq = 0; gq < 4; ++gq)
#pragma unroll
			for (int q = 0; q < 4; ++q) L[jg[gq] + 256 * q + b] = pw[4 * gq + q] * fs * (1.0 / N);
		if (lane == 0) L[M + b] = pwM * fs * (1.0 / N);
		if (lane >= 1 && lane <= b) {
			L[b - lane] = pw[0] * fs * (1.0 / N);
			L[M + b + lane] = pw[7] * fs * (1.0 / N);
		}
		wf_fence();
		seq_cumsum_nonneg_wave<18>(L, len, lane);
		const double step = (double)fs / N;
		const double origin_axis = -(b - 0.5) * fs / N;
		const double rstep = 1.0 / step;
		const double rwidth = 1.0 / width;
		const uint32_t *__restrict__ rngb = rng + wl;
		bool odd = false;  // a smoothed value that is not a positive finite number (the reference then takes log of it all the same)
		auto smooth = [&](int k, bool slow) {
			// the two abscissae in the reference's own per-bin arithmetic (interp1Q, reference src/world_matlabfunctions.cpp:220-241):
			// where the terms are small against the running sum the result hangs on the last bits of these fractions (a chirp
			// without dither: 5e-3 on the envelope with one fraction per frame instead)
			const double lo_axis = (double)k / N * fs - width / 2.0, hi_axis = lo_axis + width;
			const double lo_v = wf_interp1q(origin_axis, step, rstep, L, len, lo_axis);
			const double hi_v = wf_interp1q(origin_axis, step, rstep, L, len, hi_axis);
			double sm = (hi_v - lo_v) * rwidth;
			// infinitesimal noise (reference :220-228) then log (reference :251-252)
			sm = fma(fabs(randn_at(rngb, k)), 0.00000000000000022204460492503131, sm);
			if (slow) return wf_log_libm(sm);
			odd = odd || !wf_log_ok(sm);
			return wf_log_fast_l(sm, T);
		};
		// (every log value goes straight to its place in a second LDS array, the packed input of the second transform: held in
		// registers until the segment L may be overwritten, the sixteen values per lane were spilled to scratch memory)
#pragma unroll
		for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
			for (int q = 0; q < 4; ++q) P[jg[gq] + 256 * q] = smooth(jg[gq] + 256 * q, false);
			WF_SCHED_FENCE();  // (four bins at a time: interleaving all seventeen overflows the registers)
		}
		{
			const double lpM = smooth(M, false);
			if (lane == 0) P[M] = lpM;
		}
		if (__any(odd)) {  // (never on signals with a noise floor)
#pragma unroll
			for (int gq = 0; gq < 4; ++gq)
#pragma unroll
				for (int q = 0; q < 4; ++q) P[jg[gq] + 256 * q] = smooth(jg[gq] + 256 * q, true);
			const double lpM = smooth(M, true);
			if (lane == 0) P[M] = lpM;
		}
		wf_fence();
	}
#if WC_CT_EVEN
	// ---- smoothing + recovery lifters in the cepstral domain (reference :230-276) ----
	// Both transforms act on real even sequences -- the log spectrum, the liftered cepstrum -- and their results are real and
	// even: wf_even2048 (a 512-point complex transform + O(N) passes) instead of a 1024-point complex transform with
	// unpacking / packing each.  F[k] sits in the lane as k = t + 64 j and its partner 1024 - k.
	double lo[8], hi[8], mid;
	wf_even2048(P, L, a.tw, lane, lo, hi, mid);
	{
		// lifters at quefrency k / fs: sinc(f0 q) and 1 - 2 q1 + 2 q1 cos(2 pi q f0); with theta = pi f0 k / fs the cosine is
		// 1 - 2 sin^2 theta.  (cos, sin) of theta_k by rotations of E_t by E_64; the partner's sine from E_1024 conj(E_k)
		const double q1 = a.q1;
		const double ralpha = 1.0 / (kPi * f0c / fs);
		const double scale = 1.0 / N;  // the reference's / fft_size
		double ec, es, c64, s64, c1024, s1024, c512, s512;
		wf_sincospi(f0c / fs * lane, es, ec);
		wf_sincospi(f0c / fs * 64.0, s64, c64);
		wf_sincospi(f0c / fs * 1024.0, s1024, c1024);
		wf_sincospi(f0c / fs * 512.0, s512, c512);
		s64 = uniform_d(s64); c64 = uniform_d(c64);
		s1024 = uniform_d(s1024); c1024 = uniform_d(c1024);
		s512 = uniform_d(s512);
		const double cl0 = 1.0 - 2.0 * q1, cl1 = 2.0 * q1;
		auto lift = [&](double v, double sn, int k) {
			const double sl = sn * (ralpha * tw_load_d(a.tw + kTwInvK, k));
			const double cl = fma(cl1, fma(-2.0 * sn, sn, 1.0), cl0);
			return v * sl * cl * scale;
		};
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const int k = lane + 64 * j;
			const double sp = fma(s1024, ec, -(c1024 * es));
			double vlo = lift(lo[j], es, k);
			const double vhi = lift(hi[j], sp, 1024 - k);
			if (j == 0) vlo = (lane == 0) ? lo[0] * (cl0 + cl1) * scale : vlo;  // k = 0: sinc = 1
			P[k] = vlo;
			P[1024 - k] = vhi;
			const double cn = fma(ec, c64, -(es * s64));
			es = fma(es, c64, ec * s64);
			ec = cn;
		}
		const double vmid = lift(mid, s512, 512);
		if (lane == 0) P[512] = vmid;
	}
	wf_fence();
	wf_even2048(P, L, a.tw, lane, lo, hi, mid);
	double *__restrict__ out = a.sp + g * (long long)(M + 1);
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		out[lane + 64 * j] = wf_exp_l(lo[j], T);
		out[1024 - lane - 64 * j] = wf_exp_l(hi[j], T);
	}
	const double last = wf_exp_l(mid, T);
	if (lane == 0) out[512] = last;
}
#else
	// the mirrored log spectrum as the packed input of the second transform: sample n of slot q is 2 lane + 128 q (+ 1),
	// samples beyond 1024 are the mirror images 2048 - n
#pragma unroll
	for (int q = 0; q < 8; ++q) {
		const double2 v = *reinterpret_cast<const double2 *>(&P[2 * lane + 128 * q]);
		re[q] = v.x;
		im[q] = v.y;
	}
#pragma unroll
	for (int q = 8; q < 16; ++q) {
		re[q] = P[2048 - 2 * lane - 128 * q];
		im[q] = P[2047 - 2 * lane - 128 * q];
	}
	wf_fence();

	// ---- smoothing + recovery lifters in the cepstral domain (reference :230-276) ----
	wf_fft1024_dit<+1>(re, im, L, a.tw, lane);
	double yM;
	{
		double nyq;
		wf_r2c_unpack_re(re, im, nyq, a.tw, lane);  // 2 X, real
		// lifters at quefrency k / fs: sinc(f0 q) and 1 - 2 q1 + 2 q1 cos(2 pi q f0); with theta = pi f0 k / fs the cosine is
		// 1 - 2 sin^2 theta.  (cos, sin) of every slot's bin by rotations from three exact sincos: E_t (the lane), E_64, and
		// E_256 = E_64^4; slot A_q = E_t E_256^q, C_q = E_64 A_q, B_q = E_256^{q+1} conj(E_t) (lane 0: E_128 E_256^q),
		// D_q = E_256^{q+1} conj(E_64 E_t)
		const double q1 = a.q1;
		const double ralpha = 1.0 / (kPi * f0c / fs);
		const double scale = 0.5 / N;  // the halving left over from the unpacking and the reference's / fft_size
		double ct, st, c64, s64;
		wf_sincospi(f0c / fs * lane, st, ct);
		wf_sincospi(f0c / fs * 64.0, s64, c64);
		const double c128 = fma(-2.0 * s64, s64, 1.0), s128 = 2.0 * s64 * c64;
		const double c256 = uniform_d(fma(-2.0 * s128, s128, 1.0)), s256 = uniform_d(2.0 * s128 * c128);
		double ec[4], es[4];  // the four butterflies' running (cos, sin)
		ec[0] = ct; es[0] = st;
		ec[2] = fma(ct, c64, -(st * s64)); es[2] = fma(st, c64, ct * s64);
		ec[1] = lane ? fma(c256, ct, s256 * st) : c128; es[1] = lane ? fma(s256, ct, -(c256 * st)) : s128;
		ec[3] = fma(c256, ec[2], s256 * es[2]); es[3] = fma(s256, ec[2], -(c256 * es[2]));
		const double cl0 = 1.0 - 2.0 * q1, cl1 = 2.0 * q1;
		auto lift = [&](double v, double sn, int k) {
			const double sl = sn * (ralpha * tw_load_d(a.tw + kTwInvK, k));
			const double cl = fma(cl1, fma(-2.0 * sn, sn, 1.0), cl0);
			return v * sl * cl * scale;
		};
#pragma unroll
		for (int q = 0; q < 4; ++q) {
#pragma unroll
			for (int gq = 0; gq < 4; ++gq) {
				const int k = jg[gq] + 256 * q;
				double v = lift(re[4 * gq + q], es[gq], k);
				if (gq == 0 && q == 0) v = (lane == 0) ? re[0] * (cl0 + cl1) * scale : v;  // k = 0: sinc = 1
				re[4 * gq + q] = v;
				const double cn = fma(ec[gq], c256, -(es[gq] * s256));
				es[gq] = fma(es[gq], c256, ec[gq] * s256);
				ec[gq] = cn;
			}
		}
		yM = lift(nyq, es[0], M);  // lane 0: A_3 advanced once more is bin 1024
	}
	wf_c2r_pack_re(re, im, yM, a.tw, lane);
	wf_fft1024_dif<-1>(re, im, L, a.tw, lane);
	double *__restrict__ out = a.sp + g * (long long)(M + 1);
#pragma unroll
	for (int q = 0; q < 8; ++q) {
		out[2 * lane + 128 * q] = wf_exp_l(re[q], T);
		out[2 * lane + 128 * q + 1] = wf_exp_l(im[q], T);
	}
	const double last = wf_exp_l(re[8], T);
	if (lane == 0) out[M] = last;
}
#endif


// ---- one wavefront per frame at N = 2048 on EIGHT complex points per lane (round 5) ---------------------------------------------
// ct_wave_kernel's frame with every transform on the 512-point transforms wf8_*: the two cepstral ones already are
// (wf_even2048); the forward one -- a 1024-point complex transform of z[m] = x[2 m] + i x[2 m + 1] -- is split in time,
//     Z[k] = E[k] + W_1024^k O[k],   Z[k + 512] = E[k] - W_1024^k O[k],   E = DFT_512(z[2 n]),   O = DFT_512(z[2 n + 1]),
// the two halves one after the other by the same wavefront (E waits in sixteen registers while O is transformed).  In the
// paired8 layout a lane holds bins k and 512 - k of both, which is all the real-transform unpacking of X[k], X[1024 - k],
// X[512 - k], X[512 + k] needs: sixteen bins of the power spectrum per lane, nothing leaves the lane.  About 150 registers
// instead of 233 and ONE 9 KB buffer of LDS for everything (exchanges, DC-correction staging, the smoothing's mirrored segment,
// then the log spectrum and the cepstral transforms' exchanges in the same place; the log spectrum waits in registers until the
// segment has been read): three wavefronts per SIMD instead of two -- by tools/issue_rate.hip's figures a 32-bit instruction
// costs 1.06 instead of 1.56 ns there and an FP64 one 2.58 instead of 2.82.
// Bins of a lane's sixteen values (s = 4 c + r, k = lane + 128 c): r = 0: k, 1: 1024 - k, 2: 512 - k, 3: 512 + k; lane 0 holds
// k = 128 / 64 / 192 in rows 1 / 2 / 3 and bins 0, 512, 256, 768 in row 0; bin 1024 travels beside them (pwM, lane 0).
#ifndef WC_CT_SPLIT_OCC
#define WC_CT_SPLIT_OCC 3
#endif
__device__ __forceinline__ int ct_split_bin(int lane, int s) {
	const int c = s >> 2, r = s & 3;
	const int k = lane ? lane + 128 * c : (c == 1 ? 128 : c == 2 ? 64 : 192);
	const int gen = r == 0 ? k : r == 1 ? 1024 - k : r == 2 ? 512 - k : 512 + k;
	if (c == 0) return lane ? gen : (r == 0 ? 0 : r == 1 ? 512 : r == 2 ? 256 : 768);
	return gen;
}
// E, O: paired8 outputs of wf8_fft512_dit<+1> on z[2 n] / z[2 n + 1].  Out: pw[s] = |2 X[bin s]|^2, pwM = |2 X[1024]|^2 (lane 0).
__device__ __forceinline__ void ct_split_power(const double (&er)[8], const double (&ei)[8], const double (&orr)[8], const double (&oi)[8],
												double (&pw)[16], double &pwM, const double2 *__restrict__ tw_, int lane) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	// one quartet: E[k], E[512 - k], O[k], O[512 - k], w1 = W_1024^k, w2 = W_2048^k  ->  powers of bins k, 1024 - k, 512 - k, 512 + k
	auto quartet = [&](double ear, double eai, double ebr, double ebi, double oar, double oai, double obr, double obi, double w1r, double w1i,
					   double w2r, double w2i, double &p0, double &p1, double &p2, double &p3) {
		const double tr = fma(w1r, oar, -(w1i * oai)), ti = fma(w1r, oai, w1i * oar);       // W_1024^k O[k]
		const double ur = -fma(w1r, obr, w1i * obi), ui = -fma(w1r, obi, -(w1i * obr));     // W_1024^(512 - k) O[512 - k] = -conj(w1) O[512 - k]
		double zkr = ear + tr, zki = eai + ti, zk5r = ear - tr, zk5i = eai - ti;             // Z[k], Z[k + 512]
		double zmr = ebr + ur, zmi = ebi + ui, zm5r = ebr - ur, zm5i = ebi - ui;             // Z[512 - k], Z[1024 - k]
		wf_r2c_pair(zkr, zki, zm5r, zm5i, w2r, w2i);                                        // 2 X[k], 2 X[1024 - k]
		wf_r2c_pair(zmr, zmi, zk5r, zk5i, w2i, w2r);                                        // 2 X[512 - k], 2 X[512 + k]: W_2048^(512 - k) = i conj(w2)
		p0 = fma(zkr, zkr, zki * zki);
		p1 = fma(zm5r, zm5r, zm5i * zm5i);
		p2 = fma(zmr, zmr, zmi * zmi);
		p3 = fma(zk5r, zk5r, zk5i * zk5i);
	};
	double w1r[4], w1i[4], w2r[4], w2i[4];
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const double2 a = tw_load(tw + kTw8U + 128 * c, lane), b = tw_load(tw + kTwU + 128 * c, lane);
		w1r[c] = a.x; w1i[c] = a.y;
		w2r[c] = b.x; w2i[c] = b.y;
	}
	WF_SCHED_FENCE();
	pwM = 0.0;
	if (lane == 0) {
		// slots: 0: k = 0, 1: 128, 2: 256, 3: 384, 4: 64, 5: 192, 6: 320, 7: 448
		{
			const double ar = er[0] + orr[0], ai = ei[0] + oi[0];     // Z[0]
			const double br = er[0] - orr[0], bi = ei[0] - oi[0];     // Z[512] = X[512]
			const double x0 = 2.0 * (ar + ai), xm = 2.0 * (ar - ai);
			pw[0] = x0 * x0;
			pwM = xm * xm;
			pw[1] = 4.0 * fma(br, br, bi * bi);
		}
		{
			// k = 256: W_1024^256 = i
			double zr = er[2] - oi[2], zi = ei[2] + orr[2], yr = er[2] + oi[2], yi = ei[2] - orr[2];  // Z[256], Z[768]
			wf_r2c_pair(zr, zi, yr, yi, kH, kH);
			pw[2] = fma(zr, zr, zi * zi);
			pw[3] = fma(yr, yr, yi * yi);
		}
		quartet(er[1], ei[1], er[3], ei[3], orr[1], oi[1], orr[3], oi[3], kH, kH, kC8, kS8, pw[4], pw[5], pw[6], pw[7]);
		quartet(er[4], ei[4], er[7], ei[7], orr[4], oi[4], orr[7], oi[7], kC8, kS8, 0.98078528040323044913, 0.19509032201612826785,
				pw[8], pw[9], pw[10], pw[11]);
		quartet(er[5], ei[5], er[6], ei[6], orr[5], oi[5], orr[6], oi[6], kS8, kC8, 0.83146961230254523708, 0.55557023301960222474,
				pw[12], pw[13], pw[14], pw[15]);
	} else {
#pragma unroll
		for (int c = 0; c < 4; ++c)
			quartet(er[c], ei[c], er[7 - c], ei[7 - c], orr[c], oi[c], orr[7 - c], oi[7 - c], w1r[c], w1i[c], w2r[c], w2i[c], pw[4 * c], pw[4 * c + 1],
					pw[4 * c + 2], pw[4 * c + 3]);
	}
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WC_CT_SPLIT_OCC, WC_CT_SPLIT_OCC))) void ct_wave_split_kernel(CtArgs a) {
	constexpr int N = 2048, M = 1024;
	__shared__ __attribute__((aligned(16))) double L[kWfLds];
	__shared__ __attribute__((aligned(16))) double T[kWfTabLds];  // the tables of the lean log / exp
	const int lane = threadIdx.x;
	const long long g = xcd_frame(blockIdx.x, a.total_frames);
	if (g >= a.total_frames) return;
	const int fs = a.fs;
	const double f0v = a.f0[g];
	const double f0c = uniform_d((f0v <= a.f0_floor) ? 500.0 : f0v);  // reference :77
	if (!ct_wave_can<N>(f0c, fs)) {  // left for the block kernel behind this one
		if (lane == 0) a.rare_list[1 + atomicAdd(a.rare_list, 1)] = (int)g;
		return;
	}
	const UttDesc ud = a.utts[a.uidx[g]];
	const double *__restrict__ x = a.x + ud.x_off;
	const int x_last = ud.x_len - 1;
	const double pos = a.tpos[g];
	const uint32_t *__restrict__ rng = a.rng_table + (a.rng_off[g] - a.rng_base);
	wf_tables_to_lds(T, a.tw, lane);

	// ---- F0-adaptive window (reference :137-196): window sample i of slot q is 4 lane + 256 q + r, r < 4; the packed halves'
	// element lane + 64 q is (samples r = 0, 1) for E and (r = 2, 3) for O ----
	const int hw = __builtin_amdgcn_readfirstlane(mround(1.5 * fs / f0c));
	const int wl = 2 * hw + 1;
	const int base = __builtin_amdgcn_readfirstlane(mround(pos * fs + 0.001)) - hw;  // signal index of window sample 0
	double c0[4], s0[4], cd, sd;
	{
		const double kappa = f0c / 1.5 / fs;  // angle per sample in units of pi
#pragma unroll
		for (int r = 0; r < 4; ++r) wf_sincospi(kappa * (4 * lane + r - hw), s0[r], c0[r]);
		wf_sincospi(kappa * 256.0, sd, cd);
		sd = uniform_d(sd);
		cd = uniform_d(cd);
	}
	// walks the live slots (two at a time: a uniform branch skips what lies beyond the window) with the raw window values of the
	// slot's four samples, advanced by a rotation recurrence from the exact start phases; LOADS = 1: the pair's sixteen loads
	// (signal and draws) are issued together and fenced off from their uses
	auto walk = [&](auto loads_c, auto &&body) {
		constexpr int LOADS = decltype(loads_c)::value;
		double cc[4], ss[4];
#pragma unroll
		for (int r = 0; r < 4; ++r) { cc[r] = c0[r]; ss[r] = s0[r]; }
#pragma unroll
		for (int qg = 0; qg < 8; qg += 2) {
			if (qg * 256 >= wl) break;
			double xs[8];
			uint32_t ns[8];
			if (LOADS) {
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					const int i = 4 * lane + 256 * (qg + (k >> 2)) + (k & 3);
					xs[k] = x[clampi(base + i, 0, x_last)];
					ns[k] = rng[i < wl ? i : 0];
				}
				WF_SCHED_FENCE();
			}
#pragma unroll
			for (int k = 0; k < 2; ++k) {
				const int q = qg + k, i0 = 4 * lane + 256 * q;
				double W[4];
#pragma unroll
				for (int r = 0; r < 4; ++r) W[r] = (i0 + r < wl) ? fma(0.5, cc[r], 0.5) : 0.0;
				body(q, i0, W, LOADS ? &xs[4 * k] : nullptr, LOADS ? &ns[4 * k] : nullptr);
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const double cn = fma(cc[r], cd, -(ss[r] * sd));
					ss[r] = fma(ss[r], cd, cc[r] * sd);
					cc[r] = cn;
				}
			}
		}
	};
	double er[8], ei[8], orr[8], oi[8];
#pragma unroll
	for (int q = 0; q < 8; ++q) er[q] = ei[q] = orr[q] = oi[q] = 0.0;
	double ssq = 0.0;
	walk(std::integral_constant<int, 0>(), [&](int, int, const double (&W)[4], const double *, const uint32_t *) {
#pragma unroll
		for (int r = 0; r < 4; ++r) ssq = fma(W[r], W[r], ssq);
	});
	ssq = wave_sum_all(ssq);
	const double hr = 0.5 * (1.0 / sqrt(ssq));  // half the window norm: the unpacking below then yields X, not 2 X
	double s1 = 0.0, s2 = 0.0;
	walk(std::integral_constant<int, 1>(), [&](int q, int i0, const double (&W)[4], const double *xs, const uint32_t *ns) {
		double v[4];
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const double w = W[r] * hr;
			const double nz = (ns[r] / 268435456.0 - 6.0) * (0.5 * 0.000000000000001);
			v[r] = (i0 + r < wl) ? fma(xs[r], w, nz) : 0.0;
			s1 += v[r];
			s2 += w;
		}
		er[q] = v[0]; ei[q] = v[1]; orr[q] = v[2]; oi[q] = v[3];
	});
	s1 = wave_sum_all(s1);
	s2 = wave_sum_all(s2);
	const double wc = s1 / s2;
	walk(std::integral_constant<int, 0>(), [&](int q, int, const double (&W)[4], const double *, const uint32_t *) {
		er[q] = fma(-(W[0] * hr), wc, er[q]);
		ei[q] = fma(-(W[1] * hr), wc, ei[q]);
		orr[q] = fma(-(W[2] * hr), wc, orr[q]);
		oi[q] = fma(-(W[3] * hr), wc, oi[q]);
	});

	// ---- power spectrum (reference :198-218): the two half transforms, E first ----
	if (wl <= 512) { wdft8p<+1, 1>(er, ei); wdft8p<+1, 1>(orr, oi); }
	else if (wl <= 1024) { wdft8p<+1, 2>(er, ei); wdft8p<+1, 2>(orr, oi); }
	else { wdft8p<+1, 4>(er, ei); wdft8p<+1, 4>(orr, oi); }
	wf8_fft512_dit_rest<+1>(er, ei, L, a.tw, lane);
	wf8_fft512_dit_rest<+1>(orr, oi, L, a.tw, lane);
	double pw[16], pwM;
	ct_split_power(er, ei, orr, oi, pw, pwM, a.tw, lane);
	int bin[16];
#pragma unroll
	for (int s = 0; s < 16; ++s) bin[s] = ct_split_bin(lane, s);
	// DC correction (reference src/world_common.cpp:61-80): bins below upper - 1 <= 119 change, from bins <= upper + 1: bin lane is
	// value 0 of every lane, bin 128 - lane value 14 (lane 0: bin 64 is value 8)
	{
		const int upper = __builtin_amdgcn_readfirstlane(2 + (int)(f0c * N / fs));
		const double dx = -(double)fs / N, rdx = 1.0 / dx;
		L[lane] = pw[0];
		if (lane) L[128 - lane] = pw[14];
		else L[64] = pw[8];
		wf_fence();
		auto rep = [&](int i) {
			const double axis = (double)i * fs / N;
			return interp1q_rcp(f0c, dx, rdx, [&](int b) { return L[min(max(b, 0), 127)]; }, upper + 1, axis);
		};
		if (lane < upper - 1) pw[0] += rep(lane);
		if (upper - 1 > 64) {
			if (lane && 128 - lane < upper - 1) pw[14] += rep(128 - lane);
			if (!lane && 64 < upper - 1) pw[8] += rep(64);
		}
		wf_fence();
	}

	// ---- linear smoothing, width 2 f0 / 3 (reference src/world_common.cpp:27-52, :82-116), infinitesimal noise, log ----
	double lp[16], lpM;
	{
		const double width = f0c * 2.0 / 3.0;
		const int b = __builtin_amdgcn_readfirstlane((int)(width * N / fs) + 1);  // <= 60 (ct_wave_can)
		const int len = M + 2 * b + 1;
		// mirrored segment (reference src/world_common.cpp:33-44): position i holds bin b - i (i < b), bin i - b (b <= i < M + b),
		// bin 2 M + b - i (M + b <= i <= M + 2 b).  The low mirror comes from value 0 (bin lane of lanes 1 .. b), the high one from
		// value 1 (bin 1024 - lane); bin 1024 itself sits at M + b.
		const double sc = fs * (1.0 / N);
#pragma unroll
		for (int s = 0; s < 16; ++s) L[bin[s] + b] = pw[s] * sc;
		if (lane == 0) L[M + b] = pwM * sc;
		if (lane >= 1 && lane <= b) {
			L[b - lane] = pw[0] * sc;
			L[M + b + lane] = pw[1] * sc;
		}
		wf_fence();
		seq_cumsum_nonneg_wave<18>(L, len, lane);
		const double step = (double)fs / N;
		const double origin_axis = -(b - 0.5) * fs / N;
		const double rstep = 1.0 / step;
		const double rwidth = 1.0 / width;
		const uint32_t *__restrict__ rngb = rng + wl;
		bool odd = false;  // a smoothed value that is not a positive finite number (the reference then takes log of it all the same)
		auto smooth = [&](int k, bool slow) {
			// (the two abscissae in the reference's own per-bin arithmetic: see ct_wave_kernel)
			const double lo_axis = (double)k / N * fs - width / 2.0, hi_axis = lo_axis + width;
			const double lo_v = wf_interp1q(origin_axis, step, rstep, L, len, lo_axis);
			const double hi_v = wf_interp1q(origin_axis, step, rstep, L, len, hi_axis);
			double sm = (hi_v - lo_v) * rwidth;
			// infinitesimal noise (reference :220-228) then log (reference :251-252)
			sm = fma(fabs(randn_at(rngb, k)), 0.00000000000000022204460492503131, sm);
			if (slow) return wf_log_libm(sm);
			odd = odd || !wf_log_ok(sm);
			return wf_log_fast_l(sm, T);
		};
#pragma unroll
		for (int c = 0; c < 4; ++c) {
#pragma unroll
			for (int r = 0; r < 4; ++r) lp[4 * c + r] = smooth(bin[4 * c + r], false);
			WF_SCHED_FENCE();  // (four bins at a time)
		}
		lpM = smooth(M, false);
		if (__any(odd)) {  // (never on signals with a noise floor)
			int ln = lane, km = M;
			WC_FRESH(ln);
			WC_FRESH(km);
#pragma unroll
			for (int s = 0; s < 16; ++s) lp[s] = smooth(ct_split_bin(ln, s), true);
			lpM = smooth(km, true);
		}
		wf_fence();
	}
	// the log spectrum, bins 0 .. 1024 in natural order, where the segment was (it has been read)
#pragma unroll
	for (int s = 0; s < 16; ++s) L[bin[s]] = lp[s];
	if (lane == 0) L[M] = lpM;
	wf_fence();

	// ---- smoothing + recovery lifters in the cepstral domain (reference :230-276): see ct_wave_kernel; the real even transforms
	// read their input into registers before their first exchange, so input, exchanges and output share the one buffer ----
	double lo[8], hi[8], mid;
	wf_even2048(L, L, a.tw, lane, lo, hi, mid);
	{
		const double q1 = a.q1;
		const double ralpha = 1.0 / (kPi * f0c / fs);
		const double scale = 1.0 / N;  // the reference's / fft_size
		double ec, es, c64, s64, c1024, s1024, c512, s512;
		wf_sincospi(f0c / fs * lane, es, ec);
		wf_sincospi(f0c / fs * 64.0, s64, c64);
		wf_sincospi(f0c / fs * 1024.0, s1024, c1024);
		wf_sincospi(f0c / fs * 512.0, s512, c512);
		s64 = uniform_d(s64); c64 = uniform_d(c64);
		s1024 = uniform_d(s1024); c1024 = uniform_d(c1024);
		s512 = uniform_d(s512);
		const double cl0 = 1.0 - 2.0 * q1, cl1 = 2.0 * q1;
		auto lift = [&](double v, double sn, int k) {
			const double sl = sn * (ralpha * tw_load_d(a.tw + kTwInvK, k));
			const double cl = fma(cl1, fma(-2.0 * sn, sn, 1.0), cl0);
			return v * sl * cl * scale;
		};
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const int k = lane + 64 * j;
			const double sp = fma(s1024, ec, -(c1024 * es));
			double vlo = lift(lo[j], es, k);
			const double vhi = lift(hi[j], sp, 1024 - k);
			if (j == 0) vlo = (lane == 0) ? lo[0] * (cl0 + cl1) * scale : vlo;  // k = 0: sinc = 1
			L[k] = vlo;
			L[1024 - k] = vhi;
			const double cn = fma(ec, c64, -(es * s64));
			es = fma(es, c64, ec * s64);
			ec = cn;
		}
		const double vmid = lift(mid, s512, 512);
		if (lane == 0) L[512] = vmid;
	}
	wf_fence();
	wf_even2048(L, L, a.tw, lane, lo, hi, mid);
	double *__restrict__ out = a.sp + g * (long long)(M + 1);
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		out[lane + 64 * j] = wf_exp_l(lo[j], T);
		out[1024 - lane - 64 * j] = wf_exp_l(hi[j], T);
	}
	const double last = wf_exp_l(mid, T);
	if (lane == 0) out[512] = last;
}


// ---- one wavefront per frame at N = 1024 (16 / 22.05 / 24 kHz) ------------------------------------------------------------------
// ct_wave_kernel at EIGHT complex points per lane on the 512-point transforms wf8_* (wc_wavefft.hpp): half the registers (four
// wavefronts per SIMD instead of two), 7.7 KB of LDS.  The arithmetic is the wavefront kernel's, statement for statement; the
// log spectrum waits in registers (eight values per lane fit) until the smoothing segment may be overwritten.
#ifndef WC_CT_WAVE8_OCC
#define WC_CT_WAVE8_OCC 4
#endif
constexpr int kCt8Lds = 640;  // doubles: the mirrored segment's 513 + 2 b + 1 terms, b <= 60 (ct_wave_can<1024>); the exchange buffer is its head
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WC_CT_WAVE8_OCC, WC_CT_WAVE8_OCC))) void ct_wave8_kernel(CtArgs a) {
	constexpr int N = 1024, M = 512;
	__shared__ __attribute__((aligned(16))) double L[kCt8Lds];
	__shared__ __attribute__((aligned(16))) double T[kWfTabLds];  // the tables of the lean log / exp
	const int lane = threadIdx.x;
	const long long g = xcd_frame(blockIdx.x, a.total_frames);
	if (g >= a.total_frames) return;
	const int fs = a.fs;
	const double f0v = a.f0[g];
	const double f0c = uniform_d((f0v <= a.f0_floor) ? 500.0 : f0v);  // reference :77
	if (!ct_wave_can<N>(f0c, fs)) {  // left for the block kernel behind this one
		if (lane == 0) a.rare_list[1 + atomicAdd(a.rare_list, 1)] = (int)g;
		return;
	}
	const UttDesc ud = a.utts[a.uidx[g]];
	const double *__restrict__ x = a.x + ud.x_off;
	const int x_last = ud.x_len - 1;
	const double pos = a.tpos[g];
	const uint32_t *__restrict__ rng = a.rng_table + (a.rng_off[g] - a.rng_base);
	wf_tables_to_lds(T, a.tw, lane);

	// ---- F0-adaptive window (reference :137-196): window sample i of slot q is 2 lane + 128 q (+ 1) ----
	const int hw = __builtin_amdgcn_readfirstlane(mround(1.5 * fs / f0c));
	const int wl = 2 * hw + 1;
	const int base = __builtin_amdgcn_readfirstlane(mround(pos * fs + 0.001)) - hw;  // signal index of window sample 0
	double ce0, se0, co0, so0, cd, sd;
	{
		const double kappa = f0c / 1.5 / fs;  // angle per sample in units of pi
		wf_sincospi(kappa * (2 * lane - hw), se0, ce0);
		wf_sincospi(kappa * (2 * lane + 1 - hw), so0, co0);
		wf_sincospi(kappa * 128.0, sd, cd);
		sd = uniform_d(sd);
		cd = uniform_d(cd);
	}
	// every live sample and draw of the window requested now (see ct_wave_kernel)
	double xs_all[16];
	uint32_t ns_all[16];
#pragma unroll
	for (int qg = 0; qg < 8; qg += 4) {
		if (qg * 128 < wl) {
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				const int i = 2 * lane + 128 * (qg + (k >> 1)) + (k & 1);
				xs_all[2 * qg + k] = x[clampi(base + i, 0, x_last)];
				ns_all[2 * qg + k] = rng[i < wl ? i : 0];
			}
		}
	}
	WF_SCHED_FENCE();
	auto walk = [&](auto loads_c, auto &&body) {
		constexpr int LOADS = decltype(loads_c)::value;
		double ce = ce0, se = se0, co = co0, so = so0;
#pragma unroll
		for (int qg = 0; qg < 8; qg += 4) {
			if (qg * 128 >= wl) break;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int q = qg + k, i0 = 2 * lane + 128 * q;
				const double We = (i0 < wl) ? fma(0.5, ce, 0.5) : 0.0;
				const double Wo = (i0 + 1 < wl) ? fma(0.5, co, 0.5) : 0.0;
				body(q, i0, We, Wo, LOADS ? xs_all[2 * q] : 0.0, LOADS ? xs_all[2 * q + 1] : 0.0, LOADS ? ns_all[2 * q] : 0u, LOADS ? ns_all[2 * q + 1] : 0u);
				const double cen = fma(ce, cd, -(se * sd)), con = fma(co, cd, -(so * sd));
				se = fma(se, cd, ce * sd);
				so = fma(so, cd, co * sd);
				ce = cen;
				co = con;
			}
		}
	};
	double re[8], im[8];
#pragma unroll
	for (int q = 0; q < 8; ++q) re[q] = im[q] = 0.0;
	double ssq = 0.0;
	walk(std::integral_constant<int, 0>(), [&](int, int, double We, double Wo, double, double, uint32_t, uint32_t) { ssq = fma(Wo, Wo, fma(We, We, ssq)); });
	ssq = wave_sum_all(ssq);
	const double hr = 0.5 * (1.0 / sqrt(ssq));  // half the window norm: the transform below then yields X, not 2 X
	double s1 = 0.0, s2 = 0.0;
	walk(std::integral_constant<int, 1>(), [&](int q, int i0, double We, double Wo, double xe, double xo, uint32_t re_, uint32_t ro_) {
		const bool le = i0 < wl, lo = i0 + 1 < wl;
		const double we = We * hr, wo = Wo * hr;
		const double ne = (re_ / 268435456.0 - 6.0) * (0.5 * 0.000000000000001);
		const double no = (ro_ / 268435456.0 - 6.0) * (0.5 * 0.000000000000001);
		re[q] = le ? fma(xe, we, ne) : 0.0;
		im[q] = lo ? fma(xo, wo, no) : 0.0;
		s1 += re[q] + im[q];
		s2 += we + wo;
	});
	s1 = wave_sum_all(s1);
	s2 = wave_sum_all(s2);
	const double wc = s1 / s2;
	walk(std::integral_constant<int, 0>(), [&](int q, int, double We, double Wo, double, double, uint32_t, uint32_t) {
		re[q] = fma(-(We * hr), wc, re[q]);
		im[q] = fma(-(Wo * hr), wc, im[q]);
	});

	// ---- power spectrum (reference :198-218) ----
	if (wl <= 256) wdft8p<+1, 1>(re, im);
	else if (wl <= 512) wdft8p<+1, 2>(re, im);
	else wdft8p<+1, 4>(re, im);
	wf8_fft512_dit_rest<+1>(re, im, L, a.tw, lane);
	double pw[8], pwM;
	{
		double nyq;
		wf8_r2c_unpack(re, im, nyq, a.tw, lane);
#pragma unroll
		for (int s = 0; s < 8; ++s) pw[s] = fma(re[s], re[s], im[s] * im[s]);
		pwM = nyq * nyq;
	}
	int jg[2];
#pragma unroll
	for (int gq = 0; gq < 2; ++gq) jg[gq] = wf8_bin(lane, gq, 0);
	// DC correction (reference src/world_common.cpp:61-80): bins below upper - 1 <= 119, i.e. slot 0 (bin lane) and slot 4 (bins
	// 64 .. 127), from bins <= upper + 1
	{
		const int upper = __builtin_amdgcn_readfirstlane(2 + (int)(f0c * N / fs));
		const double dx = -(double)fs / N, rdx = 1.0 / dx;
		L[lane] = pw[0];
		L[jg[1]] = pw[4];
		wf_fence();
		auto rep = [&](int i) {
			const double axis = (double)i * fs / N;
			return interp1q_rcp(f0c, dx, rdx, [&](int b) { return L[min(max(b, 0), 127)]; }, upper + 1, axis);
		};
		if (lane < upper - 1) pw[0] += rep(lane);
		if (upper - 1 > 64) {
			if (jg[1] < upper - 1) pw[4] += rep(jg[1]);
		}
		wf_fence();
	}

	// ---- linear smoothing, width 2 f0 / 3 (reference src/world_common.cpp:27-52, :82-116), infinitesimal noise, log ----
	double lp[8], lpM;
	{
		const double width = f0c * 2.0 / 3.0;
		const int b = __builtin_amdgcn_readfirstlane((int)(width * N / fs) + 1);  // <= 60 (ct_wave_can)
		const int len = M + 2 * b + 1;
		// mirrored segment (reference src/world_common.cpp:33-44): position i holds bin b - i (i < b), bin i - b (b <= i < M + b),
		// bin 2 M + b - i (M + b <= i <= M + 2 b).  The low mirror comes from slot 0 (bins 1 .. b of lanes 1 .. b), the high one
		// from slot 7 (bins 512 - lane); bin 512 itself sits at M + b.
#pragma unroll
		for (int gq = 0; gq < 2; ++gq)
#pragma unroll
			for (int q = 0; q < 4; ++q) L[jg[gq] + 128 * q + b] = pw[4 * gq + q] * fs * (1.0 / N);
		if (lane == 0) L[M + b] = pwM * fs * (1.0 / N);
		if (lane >= 1 && lane <= b) {
			L[b - lane] = pw[0] * fs * (1.0 / N);
			L[M + b + lane] = pw[7] * fs * (1.0 / N);
		}
		wf_fence();
		seq_cumsum_nonneg_wave<10>(L, len, lane);
		const double step = (double)fs / N;
		const double origin_axis = -(b - 0.5) * fs / N;
		const double rstep = 1.0 / step;
		const double rwidth = 1.0 / width;
		const uint32_t *__restrict__ rngb = rng + wl;
		bool odd = false;  // a smoothed value that is not a positive finite number (the reference then takes log of it all the same)
		auto smooth = [&](int k, bool slow) {
			// (the two abscissae in the reference's own per-bin arithmetic: see ct_wave_kernel)
			const double lo_axis = (double)k / N * fs - width / 2.0, hi_axis = lo_axis + width;
			const double lo_v = wf_interp1q(origin_axis, step, rstep, L, len, lo_axis);
			const double hi_v = wf_interp1q(origin_axis, step, rstep, L, len, hi_axis);
			double sm = (hi_v - lo_v) * rwidth;
			// infinitesimal noise (reference :220-228) then log (reference :251-252)
			sm = fma(fabs(randn_at(rngb, k)), 0.00000000000000022204460492503131, sm);
			if (slow) return wf_log_libm(sm);
			odd = odd || !wf_log_ok(sm);
			return wf_log_fast_l(sm, T);
		};
#pragma unroll
		for (int gq = 0; gq < 2; ++gq) {
#pragma unroll
			for (int q = 0; q < 4; ++q) lp[4 * gq + q] = smooth(jg[gq] + 128 * q, false);
			WF_SCHED_FENCE();  // (four bins at a time)
		}
		lpM = smooth(M, false);
		if (__any(odd)) {  // (never on signals with a noise floor)
			// (bins from an opaque copy of the lane index: otherwise every address and fraction of the pass above is kept -- spilled --
			// for this one to reuse)
			int ln = lane, km = M;
			WC_FRESH(ln);
			WC_FRESH(km);
#pragma unroll
			for (int gq = 0; gq < 2; ++gq)
#pragma unroll
				for (int q = 0; q < 4; ++q) lp[4 * gq + q] = smooth(wf8_bin(ln, gq, q), true);
			lpM = smooth(km, true);
		}
		wf_fence();
	}
	// the mirrored log spectrum as the packed input of the second transform: sample n of slot q is 2 lane + 128 q (+ 1),
	// samples beyond 512 are the mirror images 1024 - n; through L (the segment has been read)
#pragma unroll
	for (int gq = 0; gq < 2; ++gq)
#pragma unroll
		for (int q = 0; q < 4; ++q) L[jg[gq] + 128 * q] = lp[4 * gq + q];
	if (lane == 0) L[M] = lpM;
	wf_fence();
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const double2 v = *reinterpret_cast<const double2 *>(&L[2 * lane + 128 * q]);
		re[q] = v.x;
		im[q] = v.y;
	}
#pragma unroll
	for (int q = 4; q < 8; ++q) {
		re[q] = L[1024 - 2 * lane - 128 * q];
		im[q] = L[1023 - 2 * lane - 128 * q];
	}
	wf_fence();

	// ---- smoothing + recovery lifters in the cepstral domain (reference :230-276) ----
	wf8_fft512_dit<+1>(re, im, L, a.tw, lane);
	double yM;
	{
		double nyq;
		wf8_r2c_unpack_re(re, im, nyq, a.tw, lane);  // 2 X, real
		// lifters (see ct_wave_kernel); (cos, sin) of theta = pi f0 k / fs for every slot's bin by rotations from two exact sincos:
		// E_t (the lane) and E_64, E_128 = E_64^2; slot c = E_t E_128^c, slot 4 + c = E_128^{c+1} conj(E_t) (lane 0: E_64 E_128^c)
		const double q1 = a.q1;
		const double ralpha = 1.0 / (kPi * f0c / fs);
		const double scale = 0.5 / N;  // the halving left over from the unpacking and the reference's / fft_size
		double ct, st, c64, s64;
		wf_sincospi(f0c / fs * lane, st, ct);
		wf_sincospi(f0c / fs * 64.0, s64, c64);
		const double c128 = uniform_d(fma(-2.0 * s64, s64, 1.0)), s128 = uniform_d(2.0 * s64 * c64);
		double ec[2], es[2];
		ec[0] = ct; es[0] = st;
		ec[1] = lane ? fma(c128, ct, s128 * st) : c64; es[1] = lane ? fma(s128, ct, -(c128 * st)) : s64;
		const double cl0 = 1.0 - 2.0 * q1, cl1 = 2.0 * q1;
		auto lift = [&](double v, double sn, int k) {
			const double sl = sn * (ralpha * tw_load_d(a.tw + kTwInvK, k));
			const double cl = fma(cl1, fma(-2.0 * sn, sn, 1.0), cl0);
			return v * sl * cl * scale;
		};
#pragma unroll
		for (int q = 0; q < 4; ++q) {
#pragma unroll
			for (int gq = 0; gq < 2; ++gq) {
				const int k = jg[gq] + 128 * q;
				double v = lift(re[4 * gq + q], es[gq], k);
				if (gq == 0 && q == 0) v = (lane == 0) ? re[0] * (cl0 + cl1) * scale : v;  // k = 0: sinc = 1
				re[4 * gq + q] = v;
				const double cn = fma(ec[gq], c128, -(es[gq] * s128));
				es[gq] = fma(es[gq], c128, ec[gq] * s128);
				ec[gq] = cn;
			}
		}
		yM = lift(nyq, es[0], M);  // lane 0: slot 3 advanced once more is bin 512
	}
	wf8_c2r_pack_re(re, im, yM, a.tw, lane);
	wf8_fft512_dif<-1>(re, im, L, a.tw, lane);
	double *__restrict__ out = a.sp + g * (long long)(M + 1);
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		out[2 * lane + 128 * q] = wf_exp_l(re[q], T);
		out[2 * lane + 128 * q + 1] = wf_exp_l(im[q], T);
	}
	const double last = wf_exp_l(re[4], T);
	if (lane == 0) out[M] = last;
}

}  // namespace wc

using namespace wc;

struct wc_cheaptrick {
	int fs, fft_size;
	bool wave;  // N = 2048 / 1024: one wavefront per frame (default; WC_CT_IMPL=block: the workgroup-per-frame kernel for every frame)
	bool split = false;  // N = 2048: the wavefront kernel on eight points per lane (WC_CT_IMPL=split)
	double q1, f0_floor_opt, f0_floor;
	Device *dev;
	DevBuf utts, cnt, uidx, rare, off, endpos, d_x, d_tpos, d_f0, d_sp;
	HostBuf h_stage, h_rows, h_x;
	double f0_bound = 0.0;  // (a caller's promise about the contour's highest F0: no longer relied on -- the frames the wavefront kernels leave out are listed on the device)
};

// Threads per frame: eight samples per thread up to N = 2048 (one radix-4 butterfly per thread and pass, nobody idle): 256
// threads at N = 1024 / 512 left half / three quarters of them idle in every FFT pass while barriers, reductions and the
// cumulative sum's walk cost the same -- 2.20 -> 1.85 ms per 128 k frames at N = 1024, 1.54 -> 0.86 ms at N = 512.  Above that
// the thread count stays at 256: 128 threads per 2048-point frame (two butterflies each) was slower, 14.2 against 12.9 ms per
// 512 k frames, as halving D4C's threads was in round 1.
#ifndef WC_CT_THREADS
#define WC_CT_THREADS 256
#endif
template <int N>
static void launch_ct(const CtArgs &a, hipStream_t s) {
	long long blocks = ((a.total_frames + 7) / 8) * 8;
	constexpr int T = (N >= 2048) ? WC_CT_THREADS : (N / 8 < 64 ? 64 : N / 8);
	hipLaunchKernelGGL((ct_frames_kernel<N, T>), dim3((unsigned)blocks), dim3(T), 0, s, a);
}

// Enqueue-only building blocks (no host synchronisation), shared with the fused pipeline (wc_pipeline.hip):
//   ct_prepare  descriptors upload, per-frame draw counts, per-utterance scan -> c->off, c->endpos (device)
//   ct_frames   the per-frame kernel (may run on another stream once ct_prepare's work is done)
int ct_prepare(wc_cheaptrick *c, hipStream_t s, int n_utt, const int *x_length, const double *d_f0, const int *f0_length,
			   const uint64_t *rng_pos, long long *total_out, uint64_t *min_pos_out, uint64_t *max_end_out) {
	const int bins = c->fft_size / 2 + 1;
	std::vector<UttDesc> utts(n_utt);
	long long xo = 0, fo = 0;
	uint64_t min_pos = ~0ull, max_end = 0;
	const uint64_t per_frame_max = (uint64_t)(2 * (c->fft_size / 2) + 1 + bins);
	for (int u = 0; u < n_utt; ++u) {
		if (x_length[u] <= 0 || f0_length[u] < 0) return fail(WC_ERR_INVALID, "cheaptrick: non-positive length");
		UttDesc &d = utts[u];
		d.x_off = xo; d.f_off = fo; d.y_off = 0;
		d.x_len = x_length[u]; d.f_len = f0_length[u]; d.y_len = 0; d.pad = 0;
		d.rng_pos = rng_pos ? rng_pos[u] : 0ull;
		xo += x_length[u];
		fo += f0_length[u];
		if (d.rng_pos < min_pos) min_pos = d.rng_pos;
		uint64_t e = d.rng_pos + per_frame_max * (uint64_t)d.f_len;
		if (e > max_end) max_end = e;
	}
	const long long total = fo;
	*total_out = total;
	*min_pos_out = min_pos;
	*max_end_out = max_end;
	if (total == 0) return WC_OK;
	int rc;
	if ((rc = c->utts.reserve(sizeof(UttDesc) * n_utt))) return rc;
	if ((rc = c->cnt.reserve(sizeof(uint32_t) * total))) return rc;
	if ((rc = c->uidx.reserve(sizeof(int) * total))) return rc;
	if ((rc = c->rare.reserve(sizeof(int) * (total + 1)))) return rc;
	if ((rc = c->off.reserve(sizeof(uint64_t) * total))) return rc;
	if ((rc = c->endpos.reserve(sizeof(uint64_t) * n_utt))) return rc;
	if ((rc = c->h_stage.reserve(sizeof(UttDesc) * n_utt + sizeof(uint64_t) * n_utt))) return rc;
	std::memcpy(c->h_stage.p, utts.data(), sizeof(UttDesc) * n_utt);
	WC_HIP(hipMemcpyAsync(c->utts.p, c->h_stage.p, sizeof(UttDesc) * n_utt, hipMemcpyHostToDevice, s));
	if ((rc = c->h_stage.mark(s))) return rc;
	hipLaunchKernelGGL(ct_count_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_f0, total, c->fs,
					   c->f0_floor, bins, c->cnt.as<uint32_t>(), c->utts.as<UttDesc>(), n_utt, c->uidx.as<int>(), c->rare.as<int>());
	hipLaunchKernelGGL(utt_scan_kernel, dim3(n_utt), dim3(256), 0, s, c->cnt.as<uint32_t>(), c->utts.as<UttDesc>(),
					   (const unsigned long long *)nullptr, c->off.as<unsigned long long>(), c->endpos.as<unsigned long long>());
	WC_HIP(hipGetLastError());
	return WC_OK;
}

int ct_frames(wc_cheaptrick *c, hipStream_t s, int n_utt, const double *d_x, const double *d_tpos, const double *d_f0,
			  double *d_sp, long long total, hipEvent_t *rows_done) {
	if (rows_done) *rows_done = nullptr;
	Device *dev = c->dev;
	if (total == 0) return WC_OK;
	int rc;
	CtArgs a;
	a.x = d_x; a.utts = c->utts.as<UttDesc>(); a.n_utt = n_utt; a.tpos = d_tpos; a.f0 = d_f0;
	a.rng_off = c->off.as<unsigned long long>(); a.rng_table = dev->rng_table.as<uint32_t>();
	a.rng_base = dev->rng_base; a.tw = dev->twiddle; a.sp = d_sp; a.total_frames = total; a.fs = c->fs;
	a.q1 = c->q1; a.f0_floor = c->f0_floor; a.uidx = c->uidx.as<int>();
	if ((rc = dev->time_begin("cheaptrick_frames", s))) return rc;
	a.rare_list = c->rare.as<int>();
	const unsigned grid8 = (unsigned)(((total + 7) / 8) * 8);
	switch (c->fft_size) {
		case 512: launch_ct<512>(a, s); break;
		case 1024:
			if (c->wave) {
				// one wavefront per frame at eight points per lane; the frames it leaves out (ct_wave_can: none for a contour out of
				// Harvest) are listed and done by a small grid of the block kernel behind it
				hipLaunchKernelGGL(ct_wave8_kernel, dim3(grid8), dim3(64), 0, s, a);
				hipLaunchKernelGGL((ct_frames_kernel<1024, 128, true>), dim3(64), dim3(128), 0, s, a);
			} else {
				launch_ct<1024>(a, s);
			}
			break;
		case 2048:
			if (c->wave) {
				if (c->split) hipLaunchKernelGGL(ct_wave_split_kernel, dim3(grid8), dim3(64), 0, s, a);
				else hipLaunchKernelGGL(ct_wave_kernel, dim3(grid8), dim3(64), 0, s, a);
				hipLaunchKernelGGL((ct_frames_kernel<2048, WC_CT_THREADS, true>), dim3(64), dim3(WC_CT_THREADS), 0, s, a);
			} else {
				launch_ct<2048>(a, s);
			}
			break;
		case 4096: launch_ct<4096>(a, s); break;
		default: return fail(WC_ERR_UNSUPPORTED, "cheaptrick: fft_size must be 512, 1024, 2048 or 4096");
	}
	WC_HIP(hipGetLastError());
	return dev->time_end("cheaptrick_frames", s);
}

void ct_set_f0_bound(wc_cheaptrick *c, double f0_bound) { c->f0_bound = f0_bound; }
const unsigned long long *ct_end_positions(const wc_cheaptrick *c) { return c->endpos.as<unsigned long long>(); }

static int ct_run_device(wc_cheaptrick *c, int n_utt, const double *d_x, const int *x_length, const double *d_tpos,
						 const double *d_f0, const int *f0_length, double *d_sp, uint64_t *rng_pos) {
	Device *dev = c->dev;
	hipStream_t s = dev->active();
	long long total = 0;
	uint64_t min_pos = 0, max_end = 0;
	int rc;
	// sizes first (the RNG table must exist before anything is enqueued that may outlive a reallocation)
	{
		const int bins = c->fft_size / 2 + 1;
		const uint64_t per_frame_max = (uint64_t)(2 * (c->fft_size / 2) + 1 + bins);
		uint64_t lo = ~0ull, hi = 0;
		for (int u = 0; u < n_utt; ++u) {
			if (f0_length[u] <= 0) continue;  // (no frames, no draws: its position -- a stream reset hours after the others -- must not stretch the table)
			uint64_t p0 = rng_pos ? rng_pos[u] : 0ull;
			lo = p0 < lo ? p0 : lo;
			uint64_t e = p0 + per_frame_max * (uint64_t)f0_length[u];
			hi = e > hi ? e : hi;
		}
		if (hi > lo && (rc = dev->ensure_rng(lo, hi))) return rc;
	}
	if ((rc = ct_prepare(c, s, n_utt, x_length, d_f0, f0_length, rng_pos, &total, &min_pos, &max_end))) return rc;
	if (total == 0) return WC_OK;
	if ((rc = ct_frames(c, s, n_utt, d_x, d_tpos, d_f0, d_sp, total, nullptr))) return rc;
	if (rng_pos) {
		std::vector<uint64_t> h_end(n_utt);
		WC_HIP(hipMemcpyAsync(h_end.data(), c->endpos.p, sizeof(uint64_t) * n_utt, hipMemcpyDeviceToHost, s));
		WC_HIP(hipStreamSynchronize(s));
		for (int u = 0; u < n_utt; ++u) rng_pos[u] = h_end[u];
	}
	return WC_OK;
}

wc::Device *ct_device(const wc_cheaptrick *c) { return c->dev; }

extern "C" {

wc_cheaptrick *wc_cheaptrick_create(int fs, double q1, double f0_floor, int fft_size) {
	if (fs <= 0 || f0_floor <= 0) {
		set_error("cheaptrick: fs and f0_floor must be positive");
		return nullptr;
	}
	Device *dev = current_device();
	if (!dev) return nullptr;
	wc_cheaptrick *c = new wc_cheaptrick();
	c->fs = fs;
	c->q1 = q1;
	c->f0_floor_opt = f0_floor;
	c->fft_size = fft_size ? fft_size : wc_cheaptrick_fft_size(fs, f0_floor);  // reference :36-41
	c->f0_floor = wc_cheaptrick_f0_floor(fs, c->fft_size);                      // reference :44
	c->dev = dev;
	{
		const char *impl = getenv("WC_CT_IMPL");
		c->wave = !(impl && std::strcmp(impl, "block") == 0);
		c->split = impl && std::strcmp(impl, "split") == 0;  // N = 2048 on eight points per lane (ct_wave_split_kernel; A/B)
	}
	if (c->fft_size != 512 && c->fft_size != 1024 && c->fft_size != 2048 && c->fft_size != 4096) {
		set_error("cheaptrick: fft_size must be 512, 1024, 2048 or 4096 (fs between 8 kHz and 96 kHz)");
		delete c;
		return nullptr;
	}
	dev->handle_born();
	return c;
}
void wc_cheaptrick_destroy(wc_cheaptrick *c) {
	if (!c) return;
	c->dev->quiesce();
	c->dev->handle_gone();
	c->utts.release(); c->cnt.release(); c->uidx.release(); c->rare.release(); c->off.release(); c->endpos.release();
	c->d_x.release(); c->d_tpos.release(); c->d_f0.release(); c->d_sp.release();
	c->h_stage.release();
	c->h_x.release();
	c->h_rows.release();
	delete c;
}
int wc_cheaptrick_get_fft_size(const wc_cheaptrick *c) { return c ? c->fft_size : WC_ERR_INVALID; }

int wc_cheaptrick_compute_device(wc_cheaptrick *c, int n_utt, const double *d_x, const int *x_length,
								 const double *d_tpos, const double *d_f0, const int *f0_length, double *d_sp,
								 uint64_t *rng_pos) {
	if (!c || n_utt <= 0 || !d_x || !x_length || !d_tpos || !d_f0 || !f0_length || !d_sp)
		return fail(WC_ERR_INVALID, "cheaptrick: null argument");
	WC_HIP(hipSetDevice(c->dev->id));
	DeviceLock lock(c->dev);
	return ct_run_device(c, n_utt, d_x, x_length, d_tpos, d_f0, f0_length, d_sp, rng_pos);
}

// host-pointer, single utterance, reference argument meaning (rows of `spectrogram` need not be contiguous)
int wc_cheaptrick_compute(wc_cheaptrick *c, const double *x, int x_length, const double *temporal_positions,
						  const double *f0, int f0_length, double **spectrogram) {
	if (!c || !x || !temporal_positions || !f0 || !spectrogram) return fail(WC_ERR_INVALID, "cheaptrick: null argument");
	if (x_length <= 0 || f0_length < 0) return fail(WC_ERR_INVALID, "cheaptrick: bad length");
	if (f0_length == 0) return WC_OK;
	WC_HIP(hipSetDevice(c->dev->id));
	DeviceLock lock(c->dev);
	hipStream_t s = c->dev->active();
	const int bins = c->fft_size / 2 + 1;
	int rc;
	if ((rc = c->d_x.reserve(sizeof(double) * x_length))) return rc;
	if ((rc = c->d_tpos.reserve(sizeof(double) * f0_length))) return rc;
	if ((rc = c->d_f0.reserve(sizeof(double) * f0_length))) return rc;
	if ((rc = c->d_sp.reserve(sizeof(double) * (size_t)f0_length * bins))) return rc;
	if ((rc = array_up(s, x, (size_t)x_length, c->h_x, c->d_x.as<double>()))) return rc;
	WC_HIP(hipMemcpyAsync(c->d_tpos.p, temporal_positions, sizeof(double) * f0_length, hipMemcpyHostToDevice, s));
	WC_HIP(hipMemcpyAsync(c->d_f0.p, f0, sizeof(double) * f0_length, hipMemcpyHostToDevice, s));
	uint64_t pos = global_rng_position();
	rc = ct_run_device(c, 1, c->d_x.as<double>(), &x_length, c->d_tpos.as<double>(), c->d_f0.as<double>(), &f0_length,
					   c->d_sp.as<double>(), &pos);
	if (rc) return rc;
	set_global_rng_position(pos);
	// the rows come down into page-locked staging (a pageable destination takes the copy engine's slow path, and a fresh 16 MB
	// vector per call is 4096 page faults) and go to the caller's rows by a few threads
	const size_t n_sp = (size_t)f0_length * bins;
	if ((rc = c->h_rows.reserve(sizeof(double) * n_sp))) return rc;
	return rows_down(s, spectrogram, f0_length, bins, c->d_sp.as<double>(), c->h_rows.as<double>());
}

}  // extern "C"
