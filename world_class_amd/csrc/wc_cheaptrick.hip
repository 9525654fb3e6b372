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
#include <cstring>
#include <vector>

#include "wc_device.hpp"
#include "wc_internal.hpp"
#include "wc_frames.hpp"

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
};

// per-frame number of draws: window (2 hw + 1) then one per bin (reference :153, :227)
__global__ void ct_count_kernel(const double *__restrict__ f0, long long total, int fs, double f0_floor,
								int bins, uint32_t *__restrict__ cnt) {
	long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= total) return;
	double f = f0[g];
	double f0c = (f <= f0_floor) ? 500.0 : f;
	cnt[g] = (uint32_t)(2 * mround(1.5 * fs / f0c) + 1 + bins);
}

template <int N, int T>
__global__ __launch_bounds__(T) void ct_frames_kernel(CtArgs a) {
	constexpr int M = N / 2;
	constexpr int EPT = (N + T - 1) / T;       // window elements per thread
	constexpr int BPT = (M + 1 + T - 1) / T;   // bins per thread
	__shared__ double2 A[fft_lds_size(M)];
	// LDS: the FFT workspace and the scratch of the cumulative sum only (18.3 KB at N = 2048: eight workgroups per CU).  The power
	// spectrum has no array of its own: it lives in the workspace until it has been DC-corrected, goes through registers and
	// comes back as the mirrored terms of the cumulative sum.
	__shared__ double scr[T + 2 * (T / 64)];
	__shared__ double red[2 * (T / 64) + 2];
	double *Ar = reinterpret_cast<double *>(A);

	int tid = threadIdx.x;
	long long g = xcd_frame(blockIdx.x, a.total_frames);
	if (g >= a.total_frames) return;
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

}  // namespace wc

using namespace wc;

struct wc_cheaptrick {
	int fs, fft_size;
	double q1, f0_floor_opt, f0_floor;
	Device *dev;
	DevBuf utts, cnt, off, endpos, d_x, d_tpos, d_f0, d_sp;
	HostBuf h_stage;
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
	if ((rc = c->off.reserve(sizeof(uint64_t) * total))) return rc;
	if ((rc = c->endpos.reserve(sizeof(uint64_t) * n_utt))) return rc;
	if ((rc = c->h_stage.reserve(sizeof(UttDesc) * n_utt + sizeof(uint64_t) * n_utt))) return rc;
	std::memcpy(c->h_stage.p, utts.data(), sizeof(UttDesc) * n_utt);
	WC_HIP(hipMemcpyAsync(c->utts.p, c->h_stage.p, sizeof(UttDesc) * n_utt, hipMemcpyHostToDevice, s));
	if ((rc = c->h_stage.mark(s))) return rc;
	hipLaunchKernelGGL(ct_count_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_f0, total, c->fs,
					   c->f0_floor, bins, c->cnt.as<uint32_t>());
	hipLaunchKernelGGL(utt_scan_kernel, dim3(n_utt), dim3(256), 0, s, c->cnt.as<uint32_t>(), c->utts.as<UttDesc>(),
					   (const unsigned long long *)nullptr, c->off.as<unsigned long long>(), c->endpos.as<unsigned long long>());
	WC_HIP(hipGetLastError());
	return WC_OK;
}

int ct_frames(wc_cheaptrick *c, hipStream_t s, int n_utt, const double *d_x, const double *d_tpos, const double *d_f0,
			  double *d_sp, long long total) {
	Device *dev = c->dev;
	if (total == 0) return WC_OK;
	int rc;
	CtArgs a;
	a.x = d_x; a.utts = c->utts.as<UttDesc>(); a.n_utt = n_utt; a.tpos = d_tpos; a.f0 = d_f0;
	a.rng_off = c->off.as<unsigned long long>(); a.rng_table = dev->rng_table.as<uint32_t>();
	a.rng_base = dev->rng_base; a.tw = dev->twiddle; a.sp = d_sp; a.total_frames = total; a.fs = c->fs;
	a.q1 = c->q1; a.f0_floor = c->f0_floor;
	if ((rc = dev->time_begin("cheaptrick_frames", s))) return rc;
	switch (c->fft_size) {
		case 512: launch_ct<512>(a, s); break;
		case 1024: launch_ct<1024>(a, s); break;
		case 2048: launch_ct<2048>(a, s); break;
		case 4096: launch_ct<4096>(a, s); break;
		default: return fail(WC_ERR_UNSUPPORTED, "cheaptrick: fft_size must be 512, 1024, 2048 or 4096");
	}
	WC_HIP(hipGetLastError());
	return dev->time_end("cheaptrick_frames", s);
}

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
			uint64_t p0 = rng_pos ? rng_pos[u] : 0ull;
			lo = p0 < lo ? p0 : lo;
			uint64_t e = p0 + per_frame_max * (uint64_t)(f0_length[u] > 0 ? f0_length[u] : 0);
			hi = e > hi ? e : hi;
		}
		if ((rc = dev->ensure_rng(lo, hi))) return rc;
	}
	if ((rc = ct_prepare(c, s, n_utt, x_length, d_f0, f0_length, rng_pos, &total, &min_pos, &max_end))) return rc;
	if (total == 0) return WC_OK;
	if ((rc = ct_frames(c, s, n_utt, d_x, d_tpos, d_f0, d_sp, total))) return rc;
	if (rng_pos) {
		std::vector<uint64_t> h_end(n_utt);
		WC_HIP(hipMemcpyAsync(h_end.data(), c->endpos.p, sizeof(uint64_t) * n_utt, hipMemcpyDeviceToHost, s));
		WC_HIP(hipStreamSynchronize(s));
		for (int u = 0; u < n_utt; ++u) rng_pos[u] = h_end[u];
	}
	return WC_OK;
}

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
	if (c->fft_size != 512 && c->fft_size != 1024 && c->fft_size != 2048 && c->fft_size != 4096) {
		set_error("cheaptrick: fft_size must be 512, 1024, 2048 or 4096 (fs between 8 kHz and 96 kHz)");
		delete c;
		return nullptr;
	}
	return c;
}
void wc_cheaptrick_destroy(wc_cheaptrick *c) {
	if (!c) return;
	c->dev->quiesce();
	c->utts.release(); c->cnt.release(); c->off.release(); c->endpos.release();
	c->d_x.release(); c->d_tpos.release(); c->d_f0.release(); c->d_sp.release();
	c->h_stage.release();
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
	WC_HIP(hipMemcpyAsync(c->d_x.p, x, sizeof(double) * x_length, hipMemcpyHostToDevice, s));
	WC_HIP(hipMemcpyAsync(c->d_tpos.p, temporal_positions, sizeof(double) * f0_length, hipMemcpyHostToDevice, s));
	WC_HIP(hipMemcpyAsync(c->d_f0.p, f0, sizeof(double) * f0_length, hipMemcpyHostToDevice, s));
	uint64_t pos = global_rng_position();
	rc = ct_run_device(c, 1, c->d_x.as<double>(), &x_length, c->d_tpos.as<double>(), c->d_f0.as<double>(), &f0_length,
					   c->d_sp.as<double>(), &pos);
	if (rc) return rc;
	set_global_rng_position(pos);
	std::vector<double> host((size_t)f0_length * bins);
	WC_HIP(hipMemcpyAsync(host.data(), c->d_sp.p, sizeof(double) * host.size(), hipMemcpyDeviceToHost, s));
	WC_HIP(hipStreamSynchronize(s));
	for (int i = 0; i < f0_length; ++i) std::memcpy(spectrogram[i], &host[(size_t)i * bins], sizeof(double) * bins);
	return WC_OK;
}

}  // extern "C"
