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
#include <cstring>
#include <vector>

#include "wc_device.hpp"
#include "wc_internal.hpp"
#include "wc_frames.hpp"

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
	double threshold;
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

// number of draws of one frame's LoveTrain window / of its three D4C windows
__global__ void d4c_lt_count_kernel(const double *__restrict__ f0, long long total, int fs, uint32_t *__restrict__ cnt) {
	long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= total) return;
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
__global__ __launch_bounds__(T, T >= 1024 ? 4 : (2 * T) / 256) void d4c_frames_kernel(D4cArgs a) {
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
	long long g = xcd_frame(blockIdx.x, a.total_frames);
	if (g >= a.total_frames) return;
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
		double *__restrict__ dst = a.sgd + g * (long long)(M + 1);
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
	const double *__restrict__ sgd = a.sgd + g * (long long)(M + 1);
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

}  // namespace wc

using namespace wc;

struct wc_d4c {
	int fs, fft_size_d4c, fft_size_lt, n_ap, window_length;
	double threshold;
	bool split;  // band loop and row output as separate kernels (default; WC_D4C_SPLIT=0: one fused kernel)
	Device *dev;
	DevBuf nuttall, utts, cnt, off, endpos, endpos2, ap0, sgd, coarse, d_x, d_tpos, d_f0, d_ap;
	HostBuf h_stage;
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
	if ((rc = d->off.reserve(sizeof(uint64_t) * total))) return rc;
	if ((rc = d->ap0.reserve(sizeof(double) * total))) return rc;
	const bool split = d->split;
	if (split) {
		if ((rc = d->sgd.reserve(sizeof(double) * (size_t)total * (d->fft_size_d4c / 2 + 1)))) return rc;
		if ((rc = d->coarse.reserve(sizeof(double) * (size_t)total * kMaxBands))) return rc;
	}
	if ((rc = d->endpos.reserve(sizeof(uint64_t) * n_utt))) return rc;
	if ((rc = d->endpos2.reserve(sizeof(uint64_t) * n_utt))) return rc;
	if ((rc = d->h_stage.reserve(sizeof(UttDesc) * n_utt + sizeof(uint64_t) * n_utt))) return rc;
	std::memcpy(d->h_stage.p, utts.data(), sizeof(UttDesc) * n_utt);
	WC_HIP(hipMemcpyAsync(d->utts.p, d->h_stage.p, sizeof(UttDesc) * n_utt, hipMemcpyHostToDevice, s));
	if ((rc = d->h_stage.mark(s))) return rc;
	const unsigned grid1 = (unsigned)((total + 255) / 256);
	hipLaunchKernelGGL(d4c_lt_count_kernel, dim3(grid1), dim3(256), 0, s, d_f0, total, d->fs, d->cnt.as<uint32_t>());
	hipLaunchKernelGGL(utt_scan_kernel, dim3(n_utt), dim3(256), 0, s, d->cnt.as<uint32_t>(), d->utts.as<UttDesc>(),
					   d_start, d->off.as<unsigned long long>(), d->endpos.as<unsigned long long>());
	D4cArgs a;
	a.x = d_x; a.utts = d->utts.as<UttDesc>(); a.n_utt = n_utt; a.tpos = d_tpos; a.f0 = d_f0;
	a.rng_off = d->off.as<unsigned long long>(); a.rng_table = dev->rng_table.as<uint32_t>(); a.rng_base = dev->rng_base;
	a.tw = dev->twiddle; a.ap = d_ap; a.ap0 = d->ap0.as<double>(); a.cnt = d->cnt.as<uint32_t>();
	a.sgd = d->sgd.as<double>(); a.coarse = d->coarse.as<double>();
	a.nuttall = d->nuttall.as<double>(); a.total_frames = total; a.fs = d->fs; a.fft_size_out = fft_size;
	a.n_ap = d->n_ap; a.window_length = d->window_length; a.threshold = d->threshold;
	if ((rc = dev->time_begin("d4c_lovetrain", s))) return rc;
	switch (d->fft_size_lt) {
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
	return WC_OK;
}

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
	return d;
}
void wc_d4c_destroy(wc_d4c *d) {
	if (!d) return;
	d->dev->quiesce();
	d->nuttall.release(); d->utts.release(); d->cnt.release(); d->off.release(); d->endpos.release(); d->endpos2.release();
	d->ap0.release(); d->sgd.release(); d->coarse.release(); d->d_x.release(); d->d_tpos.release(); d->d_f0.release(); d->d_ap.release(); d->h_stage.release();
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
	WC_HIP(hipMemcpyAsync(d->d_x.p, x, sizeof(double) * x_length, hipMemcpyHostToDevice, s));
	WC_HIP(hipMemcpyAsync(d->d_tpos.p, temporal_positions, sizeof(double) * f0_length, hipMemcpyHostToDevice, s));
	WC_HIP(hipMemcpyAsync(d->d_f0.p, f0, sizeof(double) * f0_length, hipMemcpyHostToDevice, s));
	uint64_t pos = global_rng_position();
	rc = d4c_run_device(d, 1, d->d_x.as<double>(), &x_length, d->d_tpos.as<double>(), d->d_f0.as<double>(), &f0_length,
						fft_size, d->d_ap.as<double>(), &pos);
	if (rc) return rc;
	set_global_rng_position(pos);
	std::vector<double> host((size_t)f0_length * bins);
	WC_HIP(hipMemcpyAsync(host.data(), d->d_ap.p, sizeof(double) * host.size(), hipMemcpyDeviceToHost, s));
	WC_HIP(hipStreamSynchronize(s));
	for (int i = 0; i < f0_length; ++i) std::memcpy(aperiodicity[i], &host[(size_t)i * bins], sizeof(double) * bins);
	return WC_OK;
}

}  // extern "C"
