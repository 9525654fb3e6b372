// Pulse-by-pulse overlap-add synthesis on gfx950.
//
// Restates reference src/synthesis.cpp:77-530 and MinimumPhaseAnalysis of
// src/world_common.cpp:196-233.
//   syn_increment_kernel sample-rate F0 / VUV interpolation (reference :180-243) -> phase increments
//   syn_phase_kernel     the reference's running phase sum (:245-262) reproduced bit for bit by integer prefix sums
//                        per binade, exceptional samples (ties, binade crossings) added in floating point
//   syn_pulses_from_phase_kernel   wrap detection (:262-288) and ordered compaction of the pulses
//   syn_timebase_kernel  the same sum as a sequential one-wavefront chain (WC_SYN_TIMEBASE=serial; cross-check)
//   syn_pulse_kernel     one workgroup per pulse (reference :308-530): sp/ap row blend, two
//                        minimum-phase analyses (r2c + c2c in LDS), fractional delay, DC removal,
//                        noise excitation from the exact stream position, overlap-add with FP64
//                        atomics (reference :118-139)
// The sum order of the overlap-add (FP64 atomics) differs from the reference's sequential loop;
// the phase accumulation is exact.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "wc_device.hpp"
#include "wc_internal.hpp"
#include "wc_frames.hpp"
#include "wc_wavefft.hpp"
#include "wc_hostcopy.hpp"

namespace wc {

constexpr double kSafe = 0.000000000001;

struct PulseBuf {
	int *index;       // sample index of the pulse
	double *shift;    // fractional time shift (s)
	int *noise_size;  // samples to the next pulse (0 for the last pulse)
	int *vuv;         // interpolated VUV at the pulse
};

struct TbArgs {
	const UttDesc *utts;
	const double *f0;          // packed frames
	const long long *cap_off;  // per-utterance first slot in the pulse arrays
	const int *cap;            // per-utterance capacity
	PulseBuf p;
	int *count;                // per-utterance number of pulses (may exceed cap: overflow)
	int *first_index;          // per-utterance index of the first pulse (for RNG offsets)
	const long long *inc_off;  // per-utterance first slot in the padded increment scratch
	int fs, fft_size;
	double frame_period;       // seconds
};

// interp1 of a coarse contour given on the uniform axis j * fp (j = 0 .. L) at time t, with the
// reference's histc semantics (reference src/world_matlabfunctions.cpp:136-182): k = clamp(#{j : j fp <= t}, 1, L)
struct Coarse {
	const double *f0;
	int L;
	double lowest_f0, fp;
	__device__ __forceinline__ double cf_in(int j) const {  // reference :232-236
		double v = f0[j];
		return (v < lowest_f0) ? 0.0 : v;
	}
	__device__ __forceinline__ double cv_in(int j) const { return (cf_in(j) == 0.0) ? 0.0 : 1.0; }
	// one extrapolated point at j == L (reference :239-242)
	__device__ __forceinline__ double cf(int j) const { return j < L ? cf_in(j) : cf_in(L - 1) * 2 - cf_in(L - 2); }
	__device__ __forceinline__ double cv(int j) const { return j < L ? cv_in(j) : cv_in(L - 1) * 2 - cv_in(L - 2); }
	__device__ __forceinline__ void at(double t, double &f, double &v) const {
		int j = (int)(t / fp);
		j = max(0, min(j, L));
		while (j < L && t >= (j + 1) * fp) ++j;
		while (j > 0 && t < j * fp) --j;
		int k = min(max(j + 1, 1), L);
		double x0 = (k - 1) * fp, x1 = k * fp;
		double s = (t - x0) / (x1 - x0);
		double f_a = cf(k - 1), f_b = cf(k), v_a = cv(k - 1), v_b = cv(k);
		f = f_a + s * (f_b - f_a);
		v = v_a + s * (v_b - v_a);
	}
};

// Phase increment of every output sample (reference :211-216, :255-262): 2 pi f0_i / fs with the
// sample-rate F0 (500 Hz where the interpolated VUV is <= 0.5), into a scratch where every utterance is
// zero-padded to a multiple of 64 samples; the sign carries the VUV: negative = unvoiced.
__global__ void syn_increment_kernel(TbArgs a, int n_utt, long long total_out, double *__restrict__ inc) {
	long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= total_out) return;
	// (the scratch is zero-filled beforehand: every utterance is padded to a multiple of 64 samples)
	int lo = 0, hi = n_utt - 1;
	while (lo < hi) {
		int mid = (lo + hi + 1) >> 1;
		if (a.utts[mid].y_off <= g) lo = mid; else hi = mid - 1;
	}
	const UttDesc ud = a.utts[lo];
	const int i = (int)(g - ud.y_off);
	Coarse co{a.f0 + ud.f_off, ud.f_len, a.fs / a.fft_size + 1.0, a.frame_period};  // integer division (reference :97)
	double f, v;
	co.at(i / (double)a.fs, f, v);
	const bool voiced = v > 0.5;
	f = voiced ? f : 500.0;
	const double cval = 2.0 * kPi / a.fs;
	const double d = f * cval;
	inc[a.inc_off[lo] + i] = voiced ? d : -d;
}

// 64 steps of the sequential phase sum, entirely in one asm block: lane L ends with run + |p[0]| + ... + |p[L]|
// added in exactly that order.  The increments are fetched with scalar loads (8 doubles per s_load_dwordx16)
// into two register tuples that are refilled while the other one is being consumed (SMEM returns out of
// order, so the only legal wait is lgkmcnt(0): wait, issue the next load, then run the 8 dependent adds).
// The set of participating lanes shrinks by shifting EXEC, so each step is one dependent v_add_f64.
// (the two tuples are the fixed registers s[40:55] and s[56:71], declared as clobbers)
__device__ __forceinline__ void chain64(double &mine, const double *__restrict__ p) {
	unsigned long long save;
	asm volatile(
		"s_mov_b64 %[sv], exec\n\t"
		"s_load_dwordx16 s[40:55], %[p], 0x0\n\t"
		"s_waitcnt lgkmcnt(0)\n\t"
		"s_load_dwordx16 s[56:71], %[p], 0x40\n\t"
		"v_add_f64 %[m], %[m], |s[40:41]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[42:43]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[44:45]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[46:47]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[48:49]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[50:51]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[52:53]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[54:55]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"s_waitcnt lgkmcnt(0)\n\t"
		"s_load_dwordx16 s[40:55], %[p], 0x80\n\t"
		"v_add_f64 %[m], %[m], |s[56:57]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[58:59]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[60:61]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[62:63]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[64:65]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[66:67]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[68:69]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[70:71]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"s_waitcnt lgkmcnt(0)\n\t"
		"s_load_dwordx16 s[56:71], %[p], 0xc0\n\t"
		"v_add_f64 %[m], %[m], |s[40:41]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[42:43]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[44:45]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[46:47]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[48:49]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[50:51]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[52:53]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[54:55]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"s_waitcnt lgkmcnt(0)\n\t"
		"s_load_dwordx16 s[40:55], %[p], 0x100\n\t"
		"v_add_f64 %[m], %[m], |s[56:57]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[58:59]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[60:61]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[62:63]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[64:65]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[66:67]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[68:69]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[70:71]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"s_waitcnt lgkmcnt(0)\n\t"
		"s_load_dwordx16 s[56:71], %[p], 0x140\n\t"
		"v_add_f64 %[m], %[m], |s[40:41]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[42:43]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[44:45]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[46:47]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[48:49]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[50:51]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[52:53]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[54:55]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"s_waitcnt lgkmcnt(0)\n\t"
		"s_load_dwordx16 s[40:55], %[p], 0x180\n\t"
		"v_add_f64 %[m], %[m], |s[56:57]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[58:59]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[60:61]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[62:63]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[64:65]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[66:67]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[68:69]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[70:71]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"s_waitcnt lgkmcnt(0)\n\t"
		"s_load_dwordx16 s[56:71], %[p], 0x1c0\n\t"
		"v_add_f64 %[m], %[m], |s[40:41]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[42:43]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[44:45]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[46:47]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[48:49]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[50:51]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[52:53]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[54:55]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"s_waitcnt lgkmcnt(0)\n\t"
		"v_add_f64 %[m], %[m], |s[56:57]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[58:59]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[60:61]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[62:63]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[64:65]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[66:67]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[68:69]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"v_add_f64 %[m], %[m], |s[70:71]|\n\t s_lshl_b64 exec, exec, 1\n\t"
		"s_mov_b64 exec, %[sv]\n\t"
		: [m] "+v"(mine), [sv] "=&s"(save)
		: [p] "s"(p)
		: "scc", "memory", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53",
		  "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70",
		  "s71");
}

// One wavefront per utterance.  The reference accumulates the phase with a sequential running sum
// (reference :255-262) and places a pulse wherever the wrapped phase jumps by more than pi.  During
// unvoiced stretches (500 Hz) at the usual sampling rates the phase hits multiples of 2 pi exactly at
// sample instants, so the pulse positions depend on the rounding of that very sum: a re-associated
// (parallel) prefix sum moves pulses by one sample.  The sum is therefore evaluated in the
// reference's order -- every lane runs the same 64-step dependent chain on values broadcast with
// v_readlane -- and only the wrap / compare / compaction part is lane-parallel.
__global__ __launch_bounds__(64) void syn_timebase_kernel(TbArgs a, const double *__restrict__ inc_all) {
	const UttDesc ud = a.utts[blockIdx.x];
	const int lane = threadIdx.x;
	const int n = ud.y_len;
	const double *__restrict__ inc_g = inc_all + a.inc_off[blockIdx.x];
	const double two_pi = 2.0 * kPi;
	const long long slot0 = a.cap_off[blockIdx.x];
	const int cap = a.cap[blockIdx.x];
	double run = 0.0;        // total phase so far (identical in every lane)
	double prev_wrap = 0.0;  // wrapped phase / VUV of the previous sample
	double prev_vu = 0.0;
	int n_pulses = 0;
	for (int base = 0; base < n; base += 64) {
		const int i = base + lane;
		const double sv = inc_g[i];  // padded with zeros past n
		const double vu = sv > 0.0 ? 1.0 : 0.0;
		// lane L accumulates run + v_0 + ... + v_L in exactly that order: the increments are wave-uniform
		// (scalar) loads and lanes drop out of the chain one by one
		double mine = run;
		const double *__restrict__ pu = inc_g + base;  // wave-uniform address: scalar loads
		chain64(mine, pu);
		run = __shfl(mine, 63, 64);
		const double wrap = fmod(mine, two_pi);
		double w_prev = __shfl_up(wrap, 1, 64);
		double v_prev = __shfl_up(vu, 1, 64);
		if (lane == 0) { w_prev = prev_wrap; v_prev = prev_vu; }
		// pulse between samples i-1 and i  <=>  |wrap[i] - wrap[i-1]| > pi ; the pulse sits at i-1
		const bool is_pulse = (i < n) && (i >= 1) && (fabs(wrap - w_prev) > kPi);
		const unsigned long long mask = __ballot(is_pulse);
		if (is_pulse) {
			const int slot = n_pulses + __popcll(mask & ((1ull << lane) - 1ull));
			if (slot < cap) {
				const double y1 = w_prev - two_pi, y2 = wrap;
				const double xx = -y1 / (y2 - y1);
				a.p.index[slot0 + slot] = i - 1;
				a.p.shift[slot0 + slot] = xx / a.fs;
				a.p.vuv[slot0 + slot] = v_prev > 0.5 ? 1 : 0;
			}
		}
		n_pulses += __popcll(mask);
		prev_wrap = __shfl(wrap, 63, 64);
		prev_vu = __shfl(vu, 63, 64);
	}
	if (lane == 0) a.count[blockIdx.x] = n_pulses;
}

// ------------------------------------------------------------------------------------------------
// The same running sum, exactly, in parallel.  While the sum stays inside one binade [2^e, 2^(e+1)) every partial sum is
// an integer multiple m u of u = 2^(e-52), and  fl(S + v) = S + (floor(v / u) + [fraction of v / u > 1/2]) u  unless the
// fraction is exactly 1/2 (round-to-even then depends on the parity of the running integer) or the sum leaves the
// binade (the rounding unit doubles).  So a window of samples is an integer prefix sum -- exact and associative -- up to
// the first such exceptional sample; that one sample is added in floating point (which *is* the reference's
// operation) and the scan restarts behind it.  Exceptions are the ~20 binade crossings of an utterance plus rare ties;
// a run of consecutive exceptions (the first few dozen samples, where the sum doubles every few steps, or a constant
// increment that happens to tie throughout a binade) is walked serially.  One workgroup per utterance.
// ------------------------------------------------------------------------------------------------
#ifndef WC_TB_K
#define WC_TB_K 16
#endif
#ifndef WC_TB_T
#define WC_TB_T 512
#endif
constexpr int TB_T = WC_TB_T, TB_K = WC_TB_K, TB_W = TB_T * TB_K;  // threads, samples per thread, samples per window
constexpr int TB_SERIAL = 64;                             // serial stretch at the start and after back-to-back exceptions

// Windows of TB_W samples are staged in LDS with coalesced loads (the next window's are in flight while this one is summed),
// summed in place -- a stretch up to the first exceptional sample per pass over the window, the exceptional sample by one exact
// addition, the rest in the next pass without touching memory again -- and leave with coalesced stores.
__global__ __launch_bounds__(TB_T) void syn_phase_kernel(TbArgs a, const double *__restrict__ inc_all, double *__restrict__ phase_all) {
	__shared__ double W[TB_W + TB_W / 16];  // |increment| of the window, replaced sample by sample with the running sum
	__shared__ unsigned long long wsum[TB_T / 64];
	__shared__ int wmin[TB_T / 64];
	__shared__ double s_state;  // exact sum up to the sample in front of s_off
	__shared__ int s_off, s_serial;
	const UttDesc ud = a.utts[blockIdx.x];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int n = ud.y_len;
	const double *__restrict__ inc = inc_all + a.inc_off[blockIdx.x];
	double *__restrict__ phase = phase_all + a.inc_off[blockIdx.x];
	auto pad = [](int i) { return i + (i >> 4); };  // a thread's TB_K consecutive samples against the lanes' stride of TB_K
	double nxt[TB_K];
#pragma unroll
	for (int k = 0; k < TB_K; ++k) {
		const int i = k * TB_T + tid;
		nxt[k] = i < n ? inc[i] : 0.0;
	}
	if (tid == 0) { s_state = 0.0; s_serial = 1; }
	for (int base = 0; base < n; base += TB_W) {
#pragma unroll
		for (int k = 0; k < TB_K; ++k) W[pad(k * TB_T + tid)] = fabs(nxt[k]);
#pragma unroll
		for (int k = 0; k < TB_K; ++k) {
			const int i = base + TB_W + k * TB_T + tid;
			nxt[k] = i < n ? inc[i] : 0.0;
		}
		if (tid == 0) s_off = 0;
		__syncthreads();
		const int wn = min(TB_W, n - base);
		while (true) {
			const int off = s_off;
			if (off >= wn) break;
			if (s_serial) {  // serial stretch by one thread: plain floating-point adds, the reference's own operation
				__syncthreads();  // (everyone has read s_off / s_serial)
				if (tid == 0) {
					double S = s_state;
					const int end = min(wn, off + TB_SERIAL);
					for (int i = off; i < end; ++i) {
						S = S + W[pad(i)];
						W[pad(i)] = S;
					}
					s_state = S;
					s_off = end;
					s_serial = 0;
				}
				__syncthreads();
				continue;
			}
			const double S = s_state;
			const long long sb = __double_as_longlong(S);
			const int e = (int)((sb >> 52) & 0x7ff);               // biased exponent of the running sum (S > 0, normal)
			const unsigned long long m0 = (unsigned long long)((sb & 0xfffffffffffffll) | (1ll << 52));  // S = m0 * 2^(e - 1075)
			// this thread's samples of the window that are still to do
			const int r0 = tid * TB_K;
			unsigned long long d[TB_K];  // (unsigned: partial sums past the first crossing may wrap, harmlessly)
			int exc = TB_W;  // first exceptional sample of the pass (window-relative), TB_W = none
			unsigned long long loc = 0;
#pragma unroll
			for (int k = 0; k < TB_K; ++k) {
				const int r = r0 + k;
				unsigned long long dk = 0;
				if (r >= off && r < wn) {
					const long long vb = __double_as_longlong(W[pad(r)]);
					const int ev = (int)((vb >> 52) & 0x7ff);
					const long long mant = (vb & 0xfffffffffffffll) | (1ll << 52);
					const int sh = e - ev;  // v = mant * 2^(ev - 1075) = (mant >> sh) u + remainder
					if (ev == 0 || sh < 0) {  // zero / subnormal increment or one larger than the sum: leave it to the exact add
						exc = min(exc, r);
					} else if (sh == 0) {
						dk = (unsigned long long)mant;
					} else if (sh <= 53) {
						const long long rem = mant & ((1ll << sh) - 1ll), half = 1ll << (sh - 1);
						dk = (unsigned long long)(mant >> sh);
						if (rem > half) dk += 1;
						else if (rem == half) exc = min(exc, r);  // tie: parity decides
					}  // sh > 53: less than half a unit, the sum does not move
				}
				d[k] = dk;
				loc += dk;
			}
			// block exclusive scan of the per-thread sums
			unsigned long long incl = loc;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const unsigned long long t = __shfl_up(incl, o, 64);
				if (lane >= o) incl += t;
			}
			if (lane == 63) wsum[wv] = incl;
			__syncthreads();
			unsigned long long bsum = m0;
#pragma unroll
			for (int w = 0; w < TB_T / 64; ++w) if (w < wv) bsum += wsum[w];
			unsigned long long run = bsum + incl - loc;
			// partial sums of this thread's samples; the first one that reaches 2^53 has left the binade
			unsigned long long mk[TB_K];
#pragma unroll
			for (int k = 0; k < TB_K; ++k) {
				run += d[k];
				mk[k] = run;
				if (run >= (1ull << 53) && r0 + k >= off && r0 + k < wn) exc = min(exc, r0 + k);
			}
			// first exception of the pass over the block
			int mn = exc;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o, 64));
			if (lane == 0) wmin[wv] = mn;
			__syncthreads();
			int x = TB_W;
#pragma unroll
			for (int w = 0; w < TB_T / 64; ++w) x = min(x, wmin[w]);
			const int vend = min(x, wn);  // samples off .. vend - 1 are exact integer sums
			const double unit = __longlong_as_double((long long)(e - 52) << 52);  // 2^(e - 1023 - 52), e >= 53 here
#pragma unroll
			for (int k = 0; k < TB_K; ++k) {
				const int r = r0 + k;
				if (r >= off && r < vend) {
					const double v = (double)mk[k] * unit;
					W[pad(r)] = v;
					if (r == vend - 1) s_state = v;  // (no exact sample leaves the state alone)
				}
			}
			__syncthreads();
			if (tid == 0) {
				int np = vend;
				int serial = 0;
				if (x < TB_W && np < wn) {  // the exceptional sample: one exact floating-point add
					const double Sx = s_state + W[pad(np)];
					W[pad(np)] = Sx;
					s_state = Sx;
					++np;
					serial = (vend == off);  // two exceptions in a row: walk a stretch serially
				}
				s_off = np;
				s_serial = serial;
			}
			__syncthreads();
		}
#pragma unroll
		for (int k = 0; k < TB_K; ++k) {
			const int i = k * TB_T + tid;
			if (i < wn) phase[base + i] = W[pad(i)];
		}
		__syncthreads();
	}
}

// ------------------------------------------------------------------------------------------------
// The same sum over many workgroups (default; WC_SYN_PHASE=single: the kernel above).  A segment of PH_W samples whose
// running sum provably stays inside one binade and meets no tie is summed as integers without knowing where it starts:
//   syn_phase_est_kernel    tree-ordered sum of every segment (an estimate, good to ~1e-13 relative);
//   syn_phase_local_kernel  from the estimated start and end (margin 2^-30 relative, against the 2^-34 a sequential sum of
//                           half a million terms can be off by) the binade; the integer increments in its unit, their prefix
//                           sums within the segment (left in the phase array as integers) and their total; "dirty" when the
//                           estimates straddle a binade edge, an increment ties, is zero or larger than the sum;
//   syn_phase_walk_kernel   one workgroup per utterance walks the segments in order: a clean one adds its total to the exact
//                           running sum (an integer add), a dirty one -- the ~15 binade crossings of an utterance, the first
//                           segment, the odd tie -- goes through the in-place passes of the kernel above with the exact start;
//   syn_phase_apply_kernel  clean segments: (mantissa at the segment's start + local prefix) x unit.
// Every value written is the reference's own floating-point sum: same argument as above, the segments only decide who adds.
// ------------------------------------------------------------------------------------------------
constexpr int PH_T = 256, PH_K = 8, PH_W = PH_T * PH_K;
constexpr int PH_MAXSEG = 2048;  // segments per utterance the walk keeps in LDS (87 s at 48 kHz); longer ones take the single-workgroup kernel

// integer increment of |v| (bits vb) in units of 2^(e - 1075); false: exceptional (zero / subnormal, larger than the sum, a tie)
__device__ __forceinline__ bool phase_increment(long long vb, int e, unsigned long long &dk) {
	const int ev = (int)((vb >> 52) & 0x7ff);
	const long long mant = (vb & 0xfffffffffffffll) | (1ll << 52);
	const int sh = e - ev;  // v = mant * 2^(ev - 1075) = (mant >> sh) u + remainder
	dk = 0;
	if (ev == 0 || sh < 0) return false;
	if (sh == 0) { dk = (unsigned long long)mant; return true; }
	if (sh <= 53) {
		const long long rem = mant & ((1ll << sh) - 1ll), half = 1ll << (sh - 1);
		dk = (unsigned long long)(mant >> sh);
		if (rem > half) dk += 1;
		else if (rem == half) return false;  // tie: parity decides
	}  // sh > 53: less than half a unit, the sum does not move
	return true;
}

struct PhArgs {
	TbArgs t;
	const double *inc;
	double *phase;
	double *segsum;             // [utt][nseg_max]
	unsigned long long *segD;   // [utt][nseg_max] integer total of a clean segment
	unsigned long long *segM0;  // [utt][nseg_max] mantissa of the exact sum in front of a clean segment
	int *segflag;               // [utt][nseg_max] biased exponent of a clean segment's binade, -1: dirty
	int nseg_max;
};

__global__ __launch_bounds__(PH_T) void syn_phase_est_kernel(PhArgs a) {
	__shared__ double red[PH_T / 64];
	const UttDesc ud = a.t.utts[blockIdx.y];
	const int n = ud.y_len, base = blockIdx.x * PH_W;
	if (base >= n) return;
	const double *__restrict__ inc = a.inc + a.t.inc_off[blockIdx.y] + base;
	const int wn = min(PH_W, n - base);
	double s = 0.0;
#pragma unroll
	for (int k = 0; k < PH_K; ++k) {
		const int i = k * PH_T + threadIdx.x;
		s += i < wn ? fabs(inc[i]) : 0.0;
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) {
		double t = 0.0;
		for (int w = 0; w < PH_T / 64; ++w) t += red[w];
		a.segsum[(long long)blockIdx.y * a.nseg_max + blockIdx.x] = t;
	}
}

__global__ __launch_bounds__(PH_T) void syn_phase_local_kernel(PhArgs a) {
	__shared__ double W[PH_W + PH_W / PH_K];
	__shared__ double red[PH_T / 64];
	__shared__ unsigned long long wsum[PH_T / 64];
	__shared__ int s_bad;
	const UttDesc ud = a.t.utts[blockIdx.y];
	const int n = ud.y_len, seg = blockIdx.x, base = seg * PH_W;
	if (base >= n) return;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const long long so = (long long)blockIdx.y * a.nseg_max;
	auto pad = [](int i) { return i + (i >> 3); };
	const double *__restrict__ inc = a.inc + a.t.inc_off[blockIdx.y] + base;
	const int wn = min(PH_W, n - base);
	// stage the segment (coalesced), estimate its start meanwhile
	double v[PH_K];
#pragma unroll
	for (int k = 0; k < PH_K; ++k) {
		const int i = k * PH_T + tid;
		v[k] = i < wn ? fabs(inc[i]) : 0.0;
	}
	double es = 0.0;
	for (int j = tid; j < seg; j += PH_T) es += a.segsum[so + j];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) es += __shfl_xor(es, o, 64);
	if (lane == 0) red[wv] = es;
	if (tid == 0) s_bad = 0;
#pragma unroll
	for (int k = 0; k < PH_K; ++k) W[pad(k * PH_T + tid)] = v[k];
	__syncthreads();
	double est0 = 0.0;
	for (int w = 0; w < PH_T / 64; ++w) est0 += red[w];
	const double est1 = est0 + a.segsum[so + seg];
	const double lo = est0 * (1.0 - 0x1p-30), hi = est1 * (1.0 + 0x1p-30);
	const int e = (int)((__double_as_longlong(lo) >> 52) & 0x7ff);
	const bool one_binade = seg > 0 && lo > 0.0 && e == (int)((__double_as_longlong(hi) >> 52) & 0x7ff) && e >= 64 && e < 2046;
	if (!one_binade) {
		if (tid == 0) a.segflag[so + seg] = -1;
		return;
	}
	unsigned long long d[PH_K], loc = 0;
	bool bad = false;
#pragma unroll
	for (int k = 0; k < PH_K; ++k) {
		const int r = tid * PH_K + k;
		unsigned long long dk = 0;
		if (r < wn) bad = !phase_increment(__double_as_longlong(W[pad(r)]), e, dk) || bad;
		d[k] = dk;
		loc += dk;
	}
	if (bad) s_bad = 1;
	unsigned long long incl = loc;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const unsigned long long t = __shfl_up(incl, o, 64);
		if (lane >= o) incl += t;
	}
	if (lane == 63) wsum[wv] = incl;
	__syncthreads();
	if (s_bad) {
		if (tid == 0) a.segflag[so + seg] = -1;
		return;
	}
	unsigned long long run = incl - loc, tot = 0;
#pragma unroll
	for (int w = 0; w < PH_T / 64; ++w) {
		if (w < wv) run += wsum[w];
		tot += wsum[w];
	}
	// the local prefix sums leave through LDS, coalesced, as integers in the phase array
	unsigned long long *Wi = reinterpret_cast<unsigned long long *>(W);
#pragma unroll
	for (int k = 0; k < PH_K; ++k) {
		run += d[k];
		Wi[pad(tid * PH_K + k)] = run;
	}
	__syncthreads();
	unsigned long long *__restrict__ out = reinterpret_cast<unsigned long long *>(a.phase + a.t.inc_off[blockIdx.y] + base);
#pragma unroll
	for (int k = 0; k < PH_K; ++k) {
		const int i = k * PH_T + tid;
		if (i < wn) out[i] = Wi[pad(i)];
	}
	if (tid == 0) {
		a.segD[so + seg] = tot;
		a.segflag[so + seg] = e;
	}
}

__global__ __launch_bounds__(PH_T) void syn_phase_walk_kernel(PhArgs a) {
	__shared__ double W[PH_W + PH_W / PH_K];
	__shared__ unsigned long long s_D[PH_MAXSEG];
	__shared__ short s_flag[PH_MAXSEG];
	__shared__ unsigned long long wsum[PH_T / 64];
	__shared__ int wmin[PH_T / 64];
	__shared__ double s_state;
	__shared__ int s_off, s_serial, s_seg;
	const UttDesc ud = a.t.utts[blockIdx.x];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int n = ud.y_len;
	const int nseg = (n + PH_W - 1) / PH_W;
	const long long so = (long long)blockIdx.x * a.nseg_max;
	const double *__restrict__ inc = a.inc + a.t.inc_off[blockIdx.x];
	double *__restrict__ phase = a.phase + a.t.inc_off[blockIdx.x];
	auto pad = [](int i) { return i + (i >> 3); };
	for (int j = tid; j < nseg; j += PH_T) {
		s_flag[j] = (short)a.segflag[so + j];
		s_D[j] = a.segD[so + j];
	}
	if (tid == 0) { s_state = 0.0; s_serial = 1; s_seg = 0; }
	__syncthreads();
	while (true) {
		// a run of clean segments: integer adds by one thread
		if (tid == 0) {
			double S = s_state;
			int j = s_seg;
			while (j < nseg && s_flag[j] >= 0) {
				const int e = s_flag[j];
				const long long sb = __double_as_longlong(S);
				const unsigned long long m0 = (unsigned long long)((sb & 0xfffffffffffffll) | (1ll << 52));
				const unsigned long long m1 = m0 + s_D[j];
				if ((int)((sb >> 52) & 0x7ff) != e || m1 >= (1ull << 53)) {  // (cannot happen within the margins; the segment is simply walked)
					s_flag[j] = -1;
					a.segflag[so + j] = -1;
					break;
				}
				a.segM0[so + j] = m0;
				S = (double)m1 * __longlong_as_double((long long)(e - 52) << 52);
				++j;
			}
			s_state = S;
			s_seg = j;
			s_off = 0;
		}
		__syncthreads();
		const int seg = s_seg;
		if (seg >= nseg) break;
		// a dirty segment: the in-place passes of syn_phase_kernel from the exact running sum
		const int base = seg * PH_W;
		const int wn = min(PH_W, n - base);
#pragma unroll
		for (int k = 0; k < PH_K; ++k) {
			const int i = k * PH_T + tid;
			W[pad(i)] = i < wn ? fabs(inc[base + i]) : 0.0;
		}
		__syncthreads();
		while (true) {
			const int off = s_off;
			if (off >= wn) break;
			if (s_serial) {
				__syncthreads();
				if (tid == 0) {
					double S = s_state;
					const int end = min(wn, off + TB_SERIAL);
					for (int i = off; i < end; ++i) {
						S = S + W[pad(i)];
						W[pad(i)] = S;
					}
					s_state = S;
					s_off = end;
					s_serial = 0;
				}
				__syncthreads();
				continue;
			}
			const double S = s_state;
			const long long sb = __double_as_longlong(S);
			const int e = (int)((sb >> 52) & 0x7ff);
			const unsigned long long m0 = (unsigned long long)((sb & 0xfffffffffffffll) | (1ll << 52));
			const int r0 = tid * PH_K;
			unsigned long long d[PH_K], loc = 0;
			int exc = PH_W;
#pragma unroll
			for (int k = 0; k < PH_K; ++k) {
				const int r = r0 + k;
				unsigned long long dk = 0;
				if (r >= off && r < wn && !phase_increment(__double_as_longlong(W[pad(r)]), e, dk)) exc = min(exc, r);
				d[k] = dk;
				loc += dk;
			}
			unsigned long long incl = loc;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const unsigned long long t = __shfl_up(incl, o, 64);
				if (lane >= o) incl += t;
			}
			if (lane == 63) wsum[wv] = incl;
			__syncthreads();
			unsigned long long run = m0 + incl - loc;
#pragma unroll
			for (int w = 0; w < PH_T / 64; ++w) if (w < wv) run += wsum[w];
			unsigned long long mk[PH_K];
#pragma unroll
			for (int k = 0; k < PH_K; ++k) {
				run += d[k];
				mk[k] = run;
				if (run >= (1ull << 53) && r0 + k >= off && r0 + k < wn) exc = min(exc, r0 + k);
			}
			int mn = exc;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o, 64));
			if (lane == 0) wmin[wv] = mn;
			__syncthreads();
			int x = PH_W;
#pragma unroll
			for (int w = 0; w < PH_T / 64; ++w) x = min(x, wmin[w]);
			const int vend = min(x, wn);
			const double unit = __longlong_as_double((long long)(e - 52) << 52);
#pragma unroll
			for (int k = 0; k < PH_K; ++k) {
				const int r = r0 + k;
				if (r >= off && r < vend) {
					const double v = (double)mk[k] * unit;
					W[pad(r)] = v;
					if (r == vend - 1) s_state = v;
				}
			}
			__syncthreads();
			if (tid == 0) {
				int np = vend;
				int serial = 0;
				if (x < PH_W && np < wn) {
					const double Sx = s_state + W[pad(np)];
					W[pad(np)] = Sx;
					s_state = Sx;
					++np;
					serial = (vend == off);
				}
				s_off = np;
				s_serial = serial;
			}
			__syncthreads();
		}
#pragma unroll
		for (int k = 0; k < PH_K; ++k) {
			const int i = k * PH_T + tid;
			if (i < wn) phase[base + i] = W[pad(i)];
		}
		if (tid == 0) s_seg = seg + 1;
		__syncthreads();
	}
}

__global__ __launch_bounds__(PH_T) void syn_phase_apply_kernel(PhArgs a) {
	const UttDesc ud = a.t.utts[blockIdx.y];
	const int n = ud.y_len, seg = blockIdx.x, base = seg * PH_W;
	if (base >= n) return;
	const long long so = (long long)blockIdx.y * a.nseg_max;
	const int e = a.segflag[so + seg];
	if (e < 0) return;  // walked: its phases are in place
	const unsigned long long m0 = a.segM0[so + seg];
	const double unit = __longlong_as_double((long long)(e - 52) << 52);
	double *__restrict__ ph = a.phase + a.t.inc_off[blockIdx.y] + base;
	const int wn = min(PH_W, n - base);
#pragma unroll
	for (int k = 0; k < PH_K; ++k) {
		const int i = k * PH_T + threadIdx.x;
		if (i < wn) ph[i] = (double)(m0 + (unsigned long long)__double_as_longlong(ph[i])) * unit;
	}
}

// Wrap, pulse detection (reference :262-288) and ordered compaction from the exact phases; one workgroup per utterance.
__global__ __launch_bounds__(TB_T) void syn_pulses_from_phase_kernel(TbArgs a, const double *__restrict__ inc_all,
																	 const double *__restrict__ phase_all) {
	__shared__ int wcnt[TB_T / 64];
	const UttDesc ud = a.utts[blockIdx.x];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int n = ud.y_len;
	const double *__restrict__ inc = inc_all + a.inc_off[blockIdx.x];
	const double *__restrict__ phase = phase_all + a.inc_off[blockIdx.x];
	const double two_pi = 2.0 * kPi;
	const long long slot0 = a.cap_off[blockIdx.x];
	const int cap = a.cap[blockIdx.x];
	constexpr int K = 8;
	int n_pulses = 0;
	for (int base = 0; base < n; base += TB_T * K) {
		const int i0 = base + tid * K;
		double w[K + 1];
		w[0] = (i0 >= 1 && i0 - 1 < n) ? fmod(phase[i0 - 1], two_pi) : 0.0;
		unsigned int flags = 0;
#pragma unroll
		for (int k = 0; k < K; ++k) {
			const int i = i0 + k;
			w[k + 1] = (i < n) ? fmod(phase[i], two_pi) : 0.0;
			// pulse between samples i-1 and i  <=>  |wrap[i] - wrap[i-1]| > pi ; the pulse sits at i-1
			if (i < n && i >= 1 && fabs(w[k + 1] - w[k]) > kPi) flags |= 1u << k;
		}
		const int mine = __popc(flags);
		int incl = mine;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const int t = __shfl_up(incl, o, 64);
			if (lane >= o) incl += t;
		}
		__syncthreads();  // (wcnt of the previous tile has been read by everyone)
		if (lane == 63) wcnt[wv] = incl;
		__syncthreads();
		int before = 0, total = 0;
#pragma unroll
		for (int q = 0; q < TB_T / 64; ++q) {
			if (q < wv) before += wcnt[q];
			total += wcnt[q];
		}
		int slot = n_pulses + before + incl - mine;
#pragma unroll
		for (int k = 0; k < K; ++k) {
			if (flags & (1u << k)) {
				const int i = i0 + k;
				if (slot < cap) {
					const double y1 = w[k] - two_pi, y2 = w[k + 1];
					const double xx = -y1 / (y2 - y1);
					a.p.index[slot0 + slot] = i - 1;
					a.p.shift[slot0 + slot] = xx / a.fs;
					a.p.vuv[slot0 + slot] = inc[i - 1] > 0.0 ? 1 : 0;
				}
				++slot;
			}
		}
		n_pulses += total;
	}
	if (tid == 0) a.count[blockIdx.x] = n_pulses;
}

// The same pulse extraction with a workgroup per tile of TB_T x PT_K samples instead of one per utterance: where the pulses are is
// decided sample by sample from the finished phase, only their slot -- the number of pulses before them -- spans the utterance.
// PASS 0 counts the pulses of every tile; PASS 1 finds them again, adds up the counts of the tiles before its own and writes them.
// (One workgroup walked the 235 tiles of a 10 s utterance in 1.1 ms, which nobody waits for in a batch of 64 but is a fifth of the
// latency of a single utterance.)
constexpr int PT_K = 8;
template <int PASS>
__global__ __launch_bounds__(TB_T) void syn_pulse_tiles_kernel(TbArgs a, const double *__restrict__ inc_all, const double *__restrict__ phase_all,
															   int *__restrict__ tile_cnt, int max_tiles) {
	__shared__ int wcnt[TB_T / 64];
	__shared__ int s_before;
	const UttDesc ud = a.utts[blockIdx.y];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int n = ud.y_len;
	const int tile = blockIdx.x;
	const int base = tile * TB_T * PT_K;
	int *__restrict__ cnt_u = tile_cnt + (long long)blockIdx.y * max_tiles;
	if (base >= n) {
		if (PASS == 0 && tid == 0) cnt_u[tile] = 0;
		return;
	}
	const double *__restrict__ inc = inc_all + a.inc_off[blockIdx.y];
	const double *__restrict__ phase = phase_all + a.inc_off[blockIdx.y];
	const double two_pi = 2.0 * kPi;
	const int i0 = base + tid * PT_K;
	double w[PT_K + 1];
	w[0] = (i0 >= 1 && i0 - 1 < n) ? fmod(phase[i0 - 1], two_pi) : 0.0;
	unsigned int flags = 0;
#pragma unroll
	for (int k = 0; k < PT_K; ++k) {
		const int i = i0 + k;
		w[k + 1] = (i < n) ? fmod(phase[i], two_pi) : 0.0;
		// pulse between samples i-1 and i  <=>  |wrap[i] - wrap[i-1]| > pi ; the pulse sits at i-1
		if (i < n && i >= 1 && fabs(w[k + 1] - w[k]) > kPi) flags |= 1u << k;
	}
	const int mine = __popc(flags);
	int incl = mine;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int t = __shfl_up(incl, o, 64);
		if (lane >= o) incl += t;
	}
	if (lane == 63) wcnt[wv] = incl;
	if (PASS == 1) {
		// pulses of the tiles before this one (and, for the workgroup of tile 0, of the whole utterance)
		const int n_tiles = (n + TB_T * PT_K - 1) / (TB_T * PT_K);
		int part = 0, all = 0;
		for (int t = tid; t < n_tiles; t += TB_T) {
			const int c = cnt_u[t];
			all += c;
			if (t < tile) part += c;
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { part += __shfl_xor(part, o, 64); all += __shfl_xor(all, o, 64); }
		if (tid == 0) s_before = 0;
		__syncthreads();
		if (lane == 0) atomicAdd(&s_before, part);
		if (tile == 0) {
			__shared__ int s_all;
			if (tid == 0) s_all = 0;
			__syncthreads();
			if (lane == 0) atomicAdd(&s_all, all);
			__syncthreads();
			if (tid == 0) a.count[blockIdx.y] = s_all;
		}
	}
	__syncthreads();
	int before = 0, total = 0;
#pragma unroll
	for (int q = 0; q < TB_T / 64; ++q) {
		if (q < wv) before += wcnt[q];
		total += wcnt[q];
	}
	if (PASS == 0) {
		if (tid == 0) cnt_u[tile] = total;
		return;
	}
	const long long slot0 = a.cap_off[blockIdx.y];
	const int cap = a.cap[blockIdx.y];
	int slot = s_before + before + incl - mine;
#pragma unroll
	for (int k = 0; k < PT_K; ++k) {
		if (flags & (1u << k)) {
			const int i = i0 + k;
			if (slot < cap) {
				const double y1 = w[k] - two_pi, y2 = w[k + 1];
				const double xx = -y1 / (y2 - y1);
				a.p.index[slot0 + slot] = i - 1;
				a.p.shift[slot0 + slot] = xx / a.fs;
				a.p.vuv[slot0 + slot] = inc[i - 1] > 0.0 ? 1 : 0;
			}
			++slot;
		}
	}
}

// noise_size of every pulse and the first pulse index of the utterance
__global__ void syn_noise_size_kernel(const long long *__restrict__ cap_off, const int *__restrict__ count,
									  const int *__restrict__ cap, PulseBuf p, int *__restrict__ first_index,
									  int *__restrict__ last_index) {
	const int u = blockIdx.y;
	const int np = min(count[u], cap[u]);
	const long long s0 = cap_off[u];
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		first_index[u] = np > 0 ? p.index[s0] : 0;
		last_index[u] = np > 0 ? p.index[s0 + np - 1] : 0;
	}
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < np; i += gridDim.x * blockDim.x) {
		int nxt = p.index[s0 + min(np - 1, i + 1)];
		p.noise_size[s0 + i] = nxt - p.index[s0 + i];
	}
}

struct SynArgs {
	const UttDesc *utts;
	int n_utt;
	const long long *pulse_prefix;  // exclusive prefix of the per-utterance pulse counts (n_utt + 1)
	const long long *cap_off;
	const int *first_index;
	PulseBuf p;
	const double *f0, *sp, *ap;
	const uint32_t *rng_table;
	unsigned long long rng_base;
	const double2 *tw;
	const double *dc_remover;
	double *out;
	const int *pulse_utt;  // the one-wavefront kernel: utterance of every pulse of the compact numbering (syn_pulse_utt_kernel)
	double *resp;  // the one-wavefront kernel: [pulse][N] responses in output order, summed by syn_overlap_add_kernel (NULL: atomics into out)
	long long total_pulses;  // launch size (capacity); the real count is pulse_prefix[n_utt]
	const unsigned long long *rng_start;  // per-utterance stream position (device), NULL = utts[u].rng_pos
	unsigned long long *trace;  // WC_SYN_TRACE builds: 16 shader-clock stamps per pulse
	long long only_pulse;  // debugging aid (builds with -DWC_DEBUG_HOOKS, env WC_DEBUG_ONLY_PULSE): synthesise only this pulse, -1 = all
	int fs;
	double frame_period;
};

// MinimumPhaseAnalysis::compute (reference src/world_common.cpp:196-233) for the block.
// ls[BPT]: log spectrum of this thread's bins k = tid + e T (k <= M).  On return A[0..M] holds the
// minimum-phase spectrum (full complex, M+1 entries).  Ends with a __syncthreads().
#ifndef WC_SYN_MINPHASE_REAL
#define WC_SYN_MINPHASE_REAL 1
#endif
template <int N, int T>
__device__ __forceinline__ void minimum_phase_lds(double2 *A, const double (&ls)[(N / 2 + T) / T],
												  const double2 *__restrict__ tw, int tid) {
	constexpr int M = N / 2;
	constexpr int BPT = (M + T) / T;
	double *Ar = reinterpret_cast<double *>(A);
	WC_FRESH(tid);
#pragma unroll
	for (int e = 0; e < BPT; ++e) {
		int k = tid + e * T;
		if (k <= M) {
			Ar[k] = ls[e];
			if (k > 0 && k < M) Ar[N - k] = ls[e];  // mirroring (reference :199-200)
		}
	}
	__syncthreads();
	fft_lds<M, T, +1>(A, tw, tid);
	r2c_post<M, T>(A, tw, tid);
	WC_FRESH(tid);
#if WC_SYN_MINPHASE_REAL
	// cepstrum folding (reference :207-217): bins 1..M-1 doubled (and conjugated), upper half zeroed.  The cepstrum of a
	// real even log spectrum is real -- the imaginary parts the reference carries along are its FFT's rounding noise,
	// 1e-16 of the real parts -- so the second transform is taken as a real one too (an M-point complex FFT and the
	// unpacking pass instead of an N-point complex FFT; SYN_MINPHASE_REAL=0 keeps the complex transform for comparison).
	double cr[2 * BPT];  // this thread's real samples n = tid + e T, n < N
#pragma unroll
	for (int e = 0; e < 2 * BPT; ++e) {
		int n = tid + e * T;
		cr[e] = 0.0;
		if (n == 0) cr[e] = A[0].x;
		else if (n == M) cr[e] = A[0].y;
		else if (n < M) cr[e] = A[n].x * 2.0;
	}
	__syncthreads();
#pragma unroll
	for (int e = 0; e < 2 * BPT; ++e) {
		int n = tid + e * T;
		if (n < N) Ar[n] = cr[e];
	}
	__syncthreads();
	fft_lds<M, T, +1>(A, tw, tid);
	r2c_post<M, T>(A, tw, tid);
	WC_FRESH(tid);
	double2 c[BPT];
#pragma unroll
	for (int e = 0; e < BPT; ++e) {
		int k = tid + e * T;
		if (k <= M) {
			double2 m = (k == 0) ? make_double2(A[0].x, 0.0) : (k == M) ? make_double2(A[0].y, 0.0) : A[k];
			double t = exp(m.x / N), sn, cs;
			sincos(m.y / N, &sn, &cs);
			c[e] = make_double2(t * cs, t * sn);
		}
	}
#else
	// cepstrum folding (reference :207-217): bins 1..M-1 doubled and conjugated, upper half zeroed
	double2 c[BPT];
#pragma unroll
	for (int e = 0; e < BPT; ++e) {
		int k = tid + e * T;
		c[e] = make_double2(0.0, 0.0);
		if (k == 0) c[e] = make_double2(A[0].x, 0.0);
		else if (k == M) c[e] = make_double2(A[0].y, 0.0);
		else if (k < M) c[e] = make_double2(A[k].x * 2.0, A[k].y * -2.0);
	}
	__syncthreads();
#pragma unroll
	for (int e = 0; e < BPT; ++e) {
		int k = tid + e * T;
		if (k <= M) A[k] = c[e];
		if (k >= 1 && k < M) A[M + k] = make_double2(0.0, 0.0);
	}
	__syncthreads();
	fft_lds<N, T, +1>(A, tw, tid);
	WC_FRESH(tid);
#pragma unroll
	for (int e = 0; e < BPT; ++e) {
		int k = tid + e * T;
		if (k <= M) {
			double2 m = A[k];
			double t = exp(m.x / N), sn, cs;
			sincos(m.y / N, &sn, &cs);
			c[e] = make_double2(t * cs, t * sn);
		}
	}
#endif
	__syncthreads();
#pragma unroll
	for (int e = 0; e < BPT; ++e) {
		int k = tid + e * T;
		if (k <= M) A[k] = c[e];
	}
	__syncthreads();
}

__device__ __forceinline__ double safe_ap(double v) { return fmax(0.001, fmin(0.999999999999, v)); }

#ifndef WC_SYN_XCD
#define WC_SYN_XCD 1
#endif
template <int N, int T>
__global__ __launch_bounds__(T) void syn_pulse_kernel(SynArgs a) {
	constexpr int M = N / 2;
	constexpr int BPT = (M + T) / T;  // bins per thread (k <= M)
	constexpr int EPT = N / T;        // time samples per thread
	__shared__ double2 A[WC_SYN_MINPHASE_REAL ? fft_lds_size(N / 2) + 1 : fft_lds_size(N)];  // M + 1 complex bins are the largest array once no N-point transform is left
	__shared__ double red[2 * (T / 64) + 2];
	double *Ar = reinterpret_cast<double *>(A);
	int tid = threadIdx.x;
#if WC_SYN_XCD
	// consecutive pulses read the same two spectrogram / aperiodicity rows: blocks are dealt round-robin to the eight XCDs, so
	// block b takes pulse (b mod 8) * ceil(total / 8) + b / 8 and an XCD's L2 sees a contiguous eighth of the pulses
	const long long total_p = a.pulse_prefix[a.n_utt];
	if ((long long)blockIdx.x >= 8 * ((total_p + 7) / 8)) return;
	const long long gp = xcd_frame(blockIdx.x, total_p);
	if (gp >= total_p) return;
#else
	const long long gp = blockIdx.x;
	if (gp >= a.pulse_prefix[a.n_utt]) return;
#endif
	if (a.only_pulse >= 0 && gp != a.only_pulse) return;
	// utterance of this pulse
	int lo = 0, hi = a.n_utt - 1;
	while (lo < hi) {
		int mid = (lo + hi + 1) >> 1;
		if (a.pulse_prefix[mid] <= gp) lo = mid; else hi = mid - 1;
	}
	const int u = lo;
	const UttDesc ud = a.utts[u];
	const long long slot = a.cap_off[u] + (gp - a.pulse_prefix[u]);
	const int pidx = a.p.index[slot];
	const double shift = a.p.shift[slot];
	const int noise_size = a.p.noise_size[slot];
	const double vuv = (double)a.p.vuv[slot];
	const int fs = a.fs, L = ud.f_len;
	const double fp = a.frame_period;
	const double t = pidx / (double)fs;  // time_axis[ii] (reference :227)

	// ---- spectral envelope / aperiodic ratio at the pulse (reference :346-393) ----
	const int fl = min(L - 1, (int)floor(t / fp));
	const int ce = min(L - 1, (int)ceil(t / fp));
	const double ipol = t / fp - fl;
	const double *__restrict__ sf = a.sp + (ud.f_off + fl) * (long long)(M + 1);
	const double *__restrict__ sc = a.sp + (ud.f_off + ce) * (long long)(M + 1);
	const double *__restrict__ af = a.ap + (ud.f_off + fl) * (long long)(M + 1);
	const double *__restrict__ ac = a.ap + (ud.f_off + ce) * (long long)(M + 1);
	double env[BPT], ar[BPT], ls[BPT];
#pragma unroll
	for (int e = 0; e < BPT; ++e) {
		int k = tid + e * T;
		env[e] = 1.0;
		ar[e] = 0.5;
		if (k <= M) {
			if (fl == ce) {
				env[e] = fabs(sf[k]);
				double s = safe_ap(af[k]);
				ar[e] = s * s;
			} else {
				env[e] = fma(1.0 - ipol, fabs(sf[k]), ipol * fabs(sc[k]));
				// both products rounded as in the reference (:388-390), no fused multiply-add: near the clamp at
				// 1 - 1e-12 the periodic weight 1 - s^2 is 2e-12, one ulp of s moves it by 1e-4 of itself and the
				// minimum-phase transform spreads that notch over the whole response (3e-8 on the waveform)
				double s = (1.0 - ipol) * safe_ap(af[k]) + ipol * safe_ap(ac[k]);
				ar[e] = s * s;
			}
		}
	}
	// aperiodic_ratio[0] decides whether there is a periodic response (reference :410)
	if (tid == 0) red[2 * (T / 64)] = ar[0];
	__syncthreads();
	const double ar0 = red[2 * (T / 64)];
	__syncthreads();

	// ---- periodic response (reference :403-474) ----
	WC_FRESH(tid);
	double periodic[EPT];
#pragma unroll
	for (int e = 0; e < EPT; ++e) periodic[e] = 0.0;
	if (!(vuv <= 0.5 || ar0 > 0.999)) {
#pragma unroll
		for (int e = 0; e < BPT; ++e) ls[e] = log(env[e] * (1.0 - ar[e]) + kSafe) / 2.0;
		minimum_phase_lds<N, T>(A, ls, a.tw, tid);
		// fractional time shift (reference :443-457), then pack for c2r
		WC_FRESH(tid);
		const double coef = 2.0 * kPi * shift * fs / N;
		double2 sp_[BPT];
#pragma unroll
		for (int e = 0; e < BPT; ++e) {
			int k = tid + e * T;
			sp_[e] = make_double2(0.0, 0.0);
			if (k <= M) {
				double2 m = A[k];
				double re2 = cos(coef * k);
				double im2 = sqrt(1.0 - re2 * re2);
				sp_[e] = make_double2(fma(m.x, re2, -(m.y * im2)), fma(m.x, im2, m.y * re2));
			}
		}
		__syncthreads();
#pragma unroll
		for (int e = 0; e < BPT; ++e) {
			int k = tid + e * T;
			if (k > 0 && k < M) A[k] = sp_[e];
			else if (k == 0) Ar[0] = sp_[e].x;   // A[0] = (Y[0].re, Y[M].re)
			else if (k == M) Ar[1] = sp_[e].x;
		}
		__syncthreads();
		c2r_pre<M, T>(A, a.tw, tid);
		fft_lds<M, T, -1>(A, a.tw, tid);
		WC_FRESH(tid);
		// fftshift + DC removal (reference :459-474): dc = sum of the shifted second half = sum wave[0..M)
		double part = 0.0;
		for (int i = tid; i < M; i += T) part += Ar[i];
		const double dc = block_sum<T>(part, red, tid);
#pragma unroll
		for (int e = 0; e < EPT; ++e) {
			int j = tid + e * T;  // index in the shifted response
			if (j < M) periodic[e] = -dc * a.dc_remover[j];
			else periodic[e] = fma(-dc, a.dc_remover[j - M], Ar[j - M]);
		}
		__syncthreads();
	}

	// ---- aperiodic response (reference :479-530) ----
	{
		WC_FRESH(tid);
		const unsigned long long rstart = a.rng_start ? a.rng_start[u] : ud.rng_pos;
		const unsigned long long roff = rstart + (unsigned long long)(pidx - a.first_index[u]) - a.rng_base;
		double nz[EPT];
		double s = 0.0;
#pragma unroll
		for (int e = 0; e < EPT; ++e) {
			int i = tid + e * T;
			nz[e] = 0.0;
			if (i < noise_size) { nz[e] = randn_at(a.rng_table, roff + i); s += nz[e]; }
		}
		s = block_sum<T>(s, red, tid);
		const double avg = s / noise_size;
#pragma unroll
		for (int e = 0; e < EPT; ++e) {
			int i = tid + e * T;
			Ar[i] = (i < noise_size) ? nz[e] - avg : 0.0;
		}
		__syncthreads();
		fft_lds<M, T, +1>(A, a.tw, tid);
		r2c_post<M, T>(A, a.tw, tid);
		WC_FRESH(tid);
		double2 ns[BPT];
#pragma unroll
		for (int e = 0; e < BPT; ++e) {
			int k = tid + e * T;
			ns[e] = make_double2(0.0, 0.0);
			if (k == 0) ns[e] = make_double2(A[0].x, 0.0);
			else if (k == M) ns[e] = make_double2(A[0].y, 0.0);
			else if (k < M) ns[e] = A[k];
		}
		__syncthreads();
		if (vuv != 0.0) {
#pragma unroll
			for (int e = 0; e < BPT; ++e) ls[e] = log(env[e] * ar[e]) / 2.0;
		} else {
#pragma unroll
			for (int e = 0; e < BPT; ++e) ls[e] = log(env[e]) / 2.0;
		}
		minimum_phase_lds<N, T>(A, ls, a.tw, tid);
		WC_FRESH(tid);
		double2 pr[BPT];
#pragma unroll
		for (int e = 0; e < BPT; ++e) {
			int k = tid + e * T;
			pr[e] = make_double2(0.0, 0.0);
			if (k <= M) {
				double2 m = A[k];
				pr[e] = make_double2(fma(m.x, ns[e].x, -(m.y * ns[e].y)), fma(m.x, ns[e].y, m.y * ns[e].x));
			}
		}
		__syncthreads();
#pragma unroll
		for (int e = 0; e < BPT; ++e) {
			int k = tid + e * T;
			if (k > 0 && k < M) A[k] = pr[e];
			else if (k == 0) Ar[0] = pr[e].x;
			else if (k == M) Ar[1] = pr[e].x;
		}
		__syncthreads();
		c2r_pre<M, T>(A, a.tw, tid);
		fft_lds<M, T, -1>(A, a.tw, tid);
	}
	// ---- mix + overlap-add (reference :339-343, :118-139) ----
	WC_FRESH(tid);
	const double sq = sqrt((double)noise_size);
	double *__restrict__ out = a.out + ud.y_off;
	const int index = pidx - M;
#pragma unroll
	for (int e = 0; e < EPT; ++e) {
		int j = tid + e * T;
		double aper = (j < M) ? Ar[j + M] : Ar[j - M];  // fftshift
		double r = fma(periodic[e], sq, aper) / N;
		int o = index + 1 + j;
		if (o >= 0 && o < ud.y_len) atomicAdd(&out[o], r);
	}
}


// ==== N = 2048 (48 kHz): one wavefront per pulse ==========================================================================
// The same arithmetic as syn_pulse_kernel on the register-resident transforms of wc_wavefft.hpp: the pulse's spectra live in
// the registers of ONE wavefront (16 complex points per lane, bins in the "paired" layout), seven transforms with two LDS
// exchanges each and no workgroup barrier; lean log / exp (wf_log, wf_exp).  Only the first half of the periodic response
// survives the reference's DC removal (:459-474: the shifted first half is overwritten with -dc * remover), so 16 values per
// lane are all that is kept of it while the aperiodic response is formed.  LDS: 9.2 KB per pulse.
//
// MinimumPhaseAnalysis::compute (reference src/world_common.cpp:196-233): in: the log spectrum ls[4 g + q] of bin
// j_g + 256 q and lsM of bin 1024 (lane 0); out: the minimum-phase spectrum (mr, mi) in the same layout, (mMr, 0) for bin 1024.
// (the caller has put the log spectrum into L itself, value by value as it formed them: L[bin], bins 0 .. 1024)
__device__ __forceinline__ void minimum_phase_wave(double (&mr)[16], double (&mi)[16], double &mMr,
												   double *L, const double *T, const double2 *__restrict__ tw, int lane) {
	constexpr int N = 2048, M = 1024;
	WC_FRESH(lane);
	int jg[4];
#pragma unroll
	for (int g = 0; g < 4; ++g) jg[g] = wf_bin(lane, g, 0);
	// the mirrored log spectrum (reference :199-200) as the packed input of the first transform
	wf_fence();
#pragma unroll
	for (int q = 0; q < 8; ++q) {
		const double2 v = *reinterpret_cast<const double2 *>(&L[2 * lane + 128 * q]);
		mr[q] = v.x;
		mi[q] = v.y;
	}
#pragma unroll
	for (int q = 8; q < 16; ++q) {
		mr[q] = L[2048 - 2 * lane - 128 * q];
		mi[q] = L[2047 - 2 * lane - 128 * q];
	}
	wf_fence();
	wf_fft1024_dit<+1>(mr, mi, L, tw, lane);
	double nyq;
	wf_r2c_unpack_re(mr, mi, nyq, tw, lane);  // twice the (real) cepstrum
	// folding (reference :207-217): bins 1 .. M-1 doubled, 0 and M kept, the upper half zero; the cepstrum of a real even log
	// spectrum is real, so the second transform is a real one too (see minimum_phase_lds)
#pragma unroll
	for (int g = 0; g < 4; ++g)
#pragma unroll
		for (int q = 0; q < 4; ++q) L[jg[g] + 256 * q] = (g == 0 && q == 0 && lane == 0) ? 0.5 * mr[0] : mr[4 * g + q];
	if (lane == 0) L[M] = 0.5 * nyq;
	wf_fence();
#pragma unroll
	for (int q = 0; q < 8; ++q) {
		const double2 v = *reinterpret_cast<const double2 *>(&L[2 * lane + 128 * q]);
		mr[q] = v.x;
		mi[q] = v.y;
	}
	mr[8] = lane == 0 ? L[M] : 0.0;
	mi[8] = 0.0;
#pragma unroll
	for (int q = 9; q < 16; ++q) mr[q] = mi[q] = 0.0;
	wf_fence();
	wf_fft1024_dit<+1>(mr, mi, L, tw, lane);
	wf_r2c_unpack(mr, mi, nyq, tw, lane);  // twice the spectrum of the folded cepstrum
#pragma unroll
	for (int s = 0; s < 16; ++s) {
		const double t = wf_exp_l(mr[s] * (0.5 / N), T);
		double sn, cs;
		wf_sincos(mi[s] * (0.5 / N), sn, cs);
		mr[s] = t * cs;
		mi[s] = t * sn;
	}
	mMr = wf_exp_l(nyq * (0.5 / N), T);
}

#ifndef WC_SYN_WAVE_OCC
#define WC_SYN_WAVE_OCC 2
#endif
#ifndef WC_SYN_ROWS_G
#define WC_SYN_ROWS_G 2
#endif
#ifndef WC_SYN_PARK_NOISE
#define WC_SYN_PARK_NOISE 1
#endif
#ifndef WC_SYN_ROW_PF
#define WC_SYN_ROW_PF 0  // frames ahead whose spectrogram / aperiodicity rows a pulse asks for on behalf of later pulses (0: off; measured at 60 / 200 frames: 4.59 / 4.66 ms against 4.54 - 4.72 without, profiles/r05_b_prefetch_ab.txt: nothing)
#endif
// WC_SYN_TRACE (development builds only): lane 0 stamps the shader clock at the phase boundaries of every pulse;
// WC_SYN_TRACE_FILE=<file> dumps them after the call (tools/syn_trace.py)
#ifndef WC_SYN_TRACE
#define WC_SYN_TRACE 0
#endif
#if WC_SYN_TRACE
#define SYN_STAMP(i) do { if (lane == 0) a.trace[gp * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define SYN_STAMP(i) do { } while (0)
#endif
// ATOMIC (WC_SYN_OLA=atomic, an A/B variant): the response goes straight into y with FP64 atomics -- the periodic half as soon as it
// exists, the aperiodic response and the DC term at the end -- instead of into a row that syn_overlap_add_kernel sums in pulse
// order: no row written and read (32 KB per pulse), no parked periodic half (16 KB), no second kernel; the sums then carry the
// order in which the atomics land (1e-16 of a sample, not the same bits on every run).  Only the noise spectrum's imaginary parts
// still wait in global memory (a row of 1024 doubles per pulse).
template <bool ATOMIC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WC_SYN_WAVE_OCC, WC_SYN_WAVE_OCC))) void syn_pulse_wave_kernel(SynArgs a) {
	constexpr int N = 2048, M = 1024;
	__shared__ __attribute__((aligned(16))) double L[kWfLds];
	__shared__ __attribute__((aligned(16))) double T[kWfTabLds];
#if WC_SYN_PARK_NOISE
	__shared__ __attribute__((aligned(16))) double P[1024];
#endif
	const int lane = threadIdx.x;
#ifdef WC_SYN_LDS_PAD  // (development builds: fewer wavefronts per SIMD, to see what one of them does alone)
	__shared__ double PAD[WC_SYN_LDS_PAD];
	if (a.fs < 0) PAD[lane] = 0.0;
#endif
	const long long total_p = a.pulse_prefix[a.n_utt];
	if ((long long)blockIdx.x >= 8 * ((total_p + 7) / 8)) return;
	const long long gp = xcd_frame(blockIdx.x, total_p);
	if (gp >= total_p) return;
	if (a.only_pulse >= 0 && gp != a.only_pulse) return;
	const int u = a.pulse_utt[gp];  // (one load instead of a bisection of the prefix: a chain of dependent loads in front of everything)
	wf_tables_to_lds(T, a.tw, lane);  // (requested first: in flight while the pulse's own data are looked up)
	const UttDesc ud = a.utts[u];
	const long long slot = a.cap_off[u] + (gp - a.pulse_prefix[u]);
	const int pidx = a.p.index[slot];
	const double shift = a.p.shift[slot];
	const int noise_size = a.p.noise_size[slot];
	const double vuv = (double)a.p.vuv[slot];
	const int fs = a.fs, Lf = ud.f_len;
	const double fp = a.frame_period;
	const double t = pidx / (double)fs;  // time_axis[ii] (reference :227)

	// ---- spectral envelope / aperiodic ratio at the pulse (reference :346-393), then the two log spectra ----
	const int fl = min(Lf - 1, (int)floor(t / fp));
	const int ce = min(Lf - 1, (int)ceil(t / fp));
	const double ipol = uniform_d(t / fp - fl);
	const double *__restrict__ sf = a.sp + (ud.f_off + fl) * (long long)(M + 1);
	const double *__restrict__ sc = a.sp + (ud.f_off + ce) * (long long)(M + 1);
	const double *__restrict__ af = a.ap + (ud.f_off + fl) * (long long)(M + 1);
	const double *__restrict__ ac = a.ap + (ud.f_off + ce) * (long long)(M + 1);
	const bool same = fl == ce;
	auto blend = [&](double s0, double s1, double a0, double a1, double &env, double &ar) {
		if (same) {
			env = fabs(s0);
			const double s = safe_ap(a0);
			ar = s * s;
		} else {
			env = fma(1.0 - ipol, fabs(s0), ipol * fabs(s1));
			// both products rounded as in the reference (:388-390), no fused multiply-add (see syn_pulse_kernel)
			const double s = (1.0 - ipol) * safe_ap(a0) + ipol * safe_ap(a1);
			ar = s * s;
		}
	};
	double ar0;  // aperiodic_ratio[0] decides whether there is a periodic response (reference :410)
	{
		double env;
		blend(sf[0], sc[0], af[0], ac[0], env, ar0);
		ar0 = uniform_d(ar0);
	}
#if WC_SYN_ROW_PF > 0
	// (round 5) the rows of the frame WC_SYN_ROW_PF frames further on in this utterance -- pulses that will run on this XCD a few hundred
	// pulses from now -- are asked for here, a cache line per lane, and looked at only when the wavefront ends (see d4c2_band_kernel)
	double pf_s, pf_a;
	{
		const long long r2 = (ud.f_off + min(ce + WC_SYN_ROW_PF, Lf - 1)) * (long long)(M + 1) + 16 * lane;
		pf_s = a.sp[r2];
		pf_a = a.ap[r2];
	}
#endif
	SYN_STAMP(0);
	SYN_STAMP(1);

	// ---- periodic response (reference :403-474; what survives of it is wave[n] - dc remover[n], n < M), then the aperiodic
	// response (reference :479-530), through one copy of the code: minimum phase of the part's log spectrum, times the
	// fractional delay / the noise spectrum, back to the time domain ----
	double dc = 0.0;
	const double sq = sqrt((double)noise_size);
#ifdef WC_SYN_RESP_ALIAS  // timing experiment only (results are garbage): all pulses share 2048 rows, which stay in the L2
	double *__restrict__ resp = a.resp + (gp & 2047) * (ATOMIC ? M : N);
#else
	double *__restrict__ resp = a.resp + gp * (ATOMIC ? M : N);
#endif
	double *__restrict__ yout = a.out + ud.y_off;
	const int ylen = ud.y_len;
	auto add2 = [&](int o, double v0, double v1) {  // y[o] += v0, y[o + 1] += v1 where they exist (reference :118-139: y[index + 1 + j] += response[j])
		if (o >= 0 && o < ylen) atomicAdd(yout + o, v0);
		if (o + 1 >= 0 && o + 1 < ylen) atomicAdd(yout + o + 1, v1);
	};
	const bool has_periodic = !(vuv <= 0.5 || ar0 > 0.999);
#pragma unroll 1
	for (int part = has_periodic ? 0 : 1; part < 2; ++part) {
		int ln = lane;
		WC_FRESH(ln);
		double wr[16], wi[16];
		double nr[16], ni[16], nsM = 0.0;
		if (part == 1) {
			// the noise (reference :514-530): noise_size draws from the pulse's place in the stream, mean removed
			const unsigned long long rstart = a.rng_start ? a.rng_start[u] : ud.rng_pos;
			const uint32_t *__restrict__ rng = a.rng_table + (rstart + (unsigned long long)(pidx - a.first_index[u]) - a.rng_base);
			double s = 0.0;
			{
				uint32_t raw[32];
#pragma unroll
				for (int q = 0; q < 16; ++q) {
					if ((q & 3) == 0 && q * 128 >= noise_size) break;
					const int i0 = 2 * ln + 128 * q;
					raw[2 * q] = rng[i0 < noise_size ? i0 : 0];
					raw[2 * q + 1] = rng[i0 + 1 < noise_size ? i0 + 1 : 0];
				}
				WF_SCHED_FENCE();
#pragma unroll
				for (int q = 0; q < 16; ++q) {
					nr[q] = ni[q] = 0.0;
					if (q * 128 < ((noise_size + 511) & ~511)) {
						const int i0 = 2 * ln + 128 * q;
						if (i0 < noise_size) nr[q] = raw[2 * q] / 268435456.0 - 6.0;
						if (i0 + 1 < noise_size) ni[q] = raw[2 * q + 1] / 268435456.0 - 6.0;
						s += nr[q] + ni[q];
					}
				}
			}
			s = wave_sum_all(s);
			const double avg = s / noise_size;
#pragma unroll
			for (int q = 0; q < 16; ++q) {
				const int i0 = 2 * ln + 128 * q;
				nr[q] = (i0 < noise_size) ? nr[q] - avg : 0.0;
				ni[q] = (i0 + 1 < noise_size) ? ni[q] - avg : 0.0;
			}
			if (noise_size <= 512) wdft16<+1, 1>(nr, ni);
			else if (noise_size <= 1024) wdft16<+1, 2>(nr, ni);
			else wdft16<+1, 4>(nr, ni);
			wf_fft1024_dit_rest<+1>(nr, ni, L, a.tw, ln);
			wf_r2c_unpack(nr, ni, nsM, a.tw, ln);  // twice the noise spectrum
			SYN_STAMP(6);
#if WC_SYN_PARK_NOISE
			// The noise spectrum waits outside the registers while the minimum phase is worked out (held, it pushes the two
			// transforms in between over 256 registers: 368 bytes of scratch per lane, half the kernel's memory traffic): its real
			// parts in LDS, its imaginary parts in the first half of the pulse's own response row (free until the mix below).
#pragma unroll
			for (int sI = 0; sI < 16; ++sI) {
				P[64 * sI + ln] = nr[sI];
				resp[64 * sI + ln] = ni[sI];
			}
			wf_fence();
#endif
		}
		// the part's log spectrum from the two rows around the pulse: log(env (1 - ar) + safeguard) / 2 for the periodic part
		// (reference :416-417), log(env ar) / 2 or, unvoiced, log(env) / 2 for the aperiodic one (:490-497)
		double mM;
		{
			auto logspec = [&](double env, double ar) {
				return wf_log_l(part == 0 ? env * (1.0 - ar) + kSafe : (vuv != 0.0 ? env * ar : env), T) / 2.0;
			};
#pragma unroll
			for (int g0 = 0; g0 < 4; g0 += WC_SYN_ROWS_G) {
				// (the rows of WC_SYN_ROWS_G groups of four bins requested together: the first touch of a pulse's rows is an HBM round
				// trip, the longest single wait of the kernel)
				double v[4 * WC_SYN_ROWS_G][4];
#pragma unroll
				for (int q = 0; q < 4 * WC_SYN_ROWS_G; ++q) {
					const int k = wf_bin(ln, g0 + (q >> 2), q & 3);
					v[q][0] = sf[k]; v[q][1] = sc[k]; v[q][2] = af[k]; v[q][3] = ac[k];
				}
				WF_SCHED_FENCE();
#pragma unroll
				for (int q = 0; q < 4 * WC_SYN_ROWS_G; ++q) {
					double env, ar;
					blend(v[q][0], v[q][1], v[q][2], v[q][3], env, ar);
					L[wf_bin(ln, g0 + (q >> 2), q & 3)] = logspec(env, ar);  // (straight to its place in the transform's input)
				}
			}
			double env, ar;
			blend(sf[M], sc[M], af[M], ac[M], env, ar);
			const double lsM = logspec(env, ar);
			if (ln == 0) L[M] = lsM;
		}
		SYN_STAMP(part ? 7 : 2);
		minimum_phase_wave(wr, wi, mM, L, T, a.tw, ln);
		SYN_STAMP(part ? 8 : 3);
		double yM;
		if (part == 0) {
			// fractional time shift (reference :443-457)
			const double coef = 2.0 * kPi * shift * fs / N;
#pragma unroll
			for (int g = 0; g < 4; ++g)
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const int k = wf_bin(ln, g, q);
					double sn_, re2;
					wf_sincos(coef * k, sn_, re2);
					const double im2 = sqrt(1.0 - re2 * re2);
					const double x = wr[4 * g + q], y = wi[4 * g + q];
					wr[4 * g + q] = fma(x, re2, -(y * im2));
					wi[4 * g + q] = fma(x, im2, y * re2);
				}
			double sn_, reM;
			wf_sincos(coef * M, sn_, reM);
			yM = mM * reM;
		} else {
#if WC_SYN_PARK_NOISE
			{
				const double *rp = resp;
				asm volatile("" : "+s"(rp));  // (an opaque pointer: real loads, not the stored values kept in registers)
#pragma unroll
				for (int sI = 0; sI < 16; ++sI) {
					ni[sI] = rp[64 * sI + ln];
					nr[sI] = P[64 * sI + ln];
				}
				WF_SCHED_FENCE();
			}
#endif
#pragma unroll
			for (int sI = 0; sI < 16; ++sI) {
				const double x = wr[sI], y = wi[sI], nx = 0.5 * nr[sI], ny = 0.5 * ni[sI];
				wr[sI] = fma(x, nx, -(y * ny));
				wi[sI] = fma(x, ny, y * nx);
			}
			yM = mM * (0.5 * nsM);
		}
		wf_c2r_pack(wr, wi, yM, a.tw, ln);
		wf_fft1024_dif<-1>(wr, wi, L, a.tw, ln);
		SYN_STAMP(part ? 9 : 4);
		if (part == 0) {
			// DC removal (reference :459-474): dc = sum of the shifted second half = sum wave[0 .. M); the shifted first half is
			// overwritten with -dc * remover, so wave[0 .. M) (shifted samples M + n) is all that is left of the response itself.
			// It goes to the output now (times sqrt(noise_size) / fft_size, reference :339-343) rather than being held across the
			// aperiodic part; the -dc * remover term joins the aperiodic response below.
#pragma unroll
			for (int q = 0; q < 8; ++q) {
				dc += wr[q] + wi[q];
				if (ATOMIC) {
					// row place M + n is output sample index - M + 1 + (M + n)
					add2(pidx + 1 + 2 * ln + 128 * q, wr[q] * sq / N, wi[q] * sq / N);
				} else {
					// (parked in the pulse's own row of the response buffer, unscaled; the mix below picks it up)
					*reinterpret_cast<double2 *>(resp + M + 2 * ln + 128 * q) = make_double2(wr[q] * sq, wi[q] * sq);
				}
			}
			dc = wave_sum_all(dc);
			SYN_STAMP(5);
		} else {
			// ---- mix + overlap-add (reference :339-343, :118-139): shifted sample j is unshifted sample j - M (j >= M) / j + M ----
			// The response (periodic sqrt(noise_size) + aperiodic) / fft_size goes to the pulse's row of the response buffer in output
			// order; syn_overlap_add_kernel sums the rows into y pulse after pulse, the reference's order (:118-139) -- no atomics,
			// the same bits on every run.
			const double dcs = has_periodic ? -dc * sq : 0.0;
			if (ATOMIC) {
#pragma unroll
				for (int q0 = 0; q0 < 8; q0 += 4) {
					double2 dr[4];
#pragma unroll
					for (int q = 0; q < 4; ++q) dr[q] = *reinterpret_cast<const double2 *>(a.dc_remover + 2 * ln + 128 * (q0 + q));
					WF_SCHED_FENCE();
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						const int n = 2 * ln + 128 * (q0 + q);
						add2(pidx + 1 + n, fma(dcs, dr[q].x, wr[q0 + q]) / N, fma(dcs, dr[q].y, wi[q0 + q]) / N);
						add2(pidx + 1 - M + n, fma(dcs, dr[q].x, wr[q0 + q + 8]) / N, fma(dcs, dr[q].y, wi[q0 + q + 8]) / N);
					}
				}
			} else
#pragma unroll
			for (int q0 = 0; q0 < 8; q0 += 4) {
				double2 dr[4], pp[4];
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					dr[q] = *reinterpret_cast<const double2 *>(a.dc_remover + 2 * ln + 128 * (q0 + q));
					pp[q] = has_periodic ? *reinterpret_cast<const double2 *>(resp + M + 2 * ln + 128 * (q0 + q)) : make_double2(0.0, 0.0);
				}
				WF_SCHED_FENCE();
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const int n = 2 * ln + 128 * (q0 + q);  // unshifted samples n, n + 1 (output place n + M) and n + M, n + M + 1 (place n)
					const double r0 = fma(dcs, dr[q].x, wr[q0 + q]) + pp[q].x, r1 = fma(dcs, dr[q].y, wi[q0 + q]) + pp[q].y;
					*reinterpret_cast<double2 *>(resp + n + M) = make_double2(r0 / N, r1 / N);
					const double r2 = fma(dcs, dr[q].x, wr[q0 + q + 8]), r3 = fma(dcs, dr[q].y, wi[q0 + q + 8]);
					*reinterpret_cast<double2 *>(resp + n) = make_double2(r2 / N, r3 / N);
				}
			}
			SYN_STAMP(10);
		}
	}
#if WC_SYN_ROW_PF > 0
	asm volatile("" ::"v"(pf_s), "v"(pf_a));
#endif
}

// ==== N = 1024 (16 / 22.05 / 24 kHz): one wavefront per pulse at eight points per lane ========================================
// syn_pulse_wave_kernel on the 512-point transforms wf8_* (wc_wavefft.hpp), statement for statement: half the registers, four
// wavefronts per SIMD.  A workgroup is FOUR such wavefronts, each on a pulse of its own with its own exchange buffer and parking
// array; all they share is the table of the lean log / exp (2.5 KB a wavefront would otherwise copy for itself: with it the
// sixteen wavefronts of a CU would not fit its LDS), loaded in front of the only barrier of the kernel.
__device__ __forceinline__ void minimum_phase_wave8(double (&mr)[8], double (&mi)[8], double &mMr,
													double *L, const double *T, const double2 *__restrict__ tw, int lane) {
	constexpr int N = 1024, M = 512;
	WC_FRESH(lane);
	int jg[2];
#pragma unroll
	for (int g = 0; g < 2; ++g) jg[g] = wf8_bin(lane, g, 0);
	// the mirrored log spectrum (reference :199-200) as the packed input of the first transform
	wf_fence();
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const double2 v = *reinterpret_cast<const double2 *>(&L[2 * lane + 128 * q]);
		mr[q] = v.x;
		mi[q] = v.y;
	}
#pragma unroll
	for (int q = 4; q < 8; ++q) {
		mr[q] = L[1024 - 2 * lane - 128 * q];
		mi[q] = L[1023 - 2 * lane - 128 * q];
	}
	wf_fence();
	wf8_fft512_dit<+1>(mr, mi, L, tw, lane);
	double nyq;
	wf8_r2c_unpack_re(mr, mi, nyq, tw, lane);  // twice the (real) cepstrum
	// folding (reference :207-217): bins 1 .. M-1 doubled, 0 and M kept, the upper half zero (see minimum_phase_wave)
#pragma unroll
	for (int g = 0; g < 2; ++g)
#pragma unroll
		for (int q = 0; q < 4; ++q) L[jg[g] + 128 * q] = (g == 0 && q == 0 && lane == 0) ? 0.5 * mr[0] : mr[4 * g + q];
	if (lane == 0) L[M] = 0.5 * nyq;
	wf_fence();
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const double2 v = *reinterpret_cast<const double2 *>(&L[2 * lane + 128 * q]);
		mr[q] = v.x;
		mi[q] = v.y;
	}
	mr[4] = lane == 0 ? L[M] : 0.0;
	mi[4] = 0.0;
#pragma unroll
	for (int q = 5; q < 8; ++q) mr[q] = mi[q] = 0.0;
	wf_fence();
	wf8_fft512_dit<+1>(mr, mi, L, tw, lane);
	wf8_r2c_unpack(mr, mi, nyq, tw, lane);  // twice the spectrum of the folded cepstrum
#pragma unroll
	for (int s = 0; s < 8; ++s) {
		const double t = wf_exp_l(mr[s] * (0.5 / N), T);
		double sn, cs;
		wf_sincos(mi[s] * (0.5 / N), sn, cs);
		mr[s] = t * cs;
		mi[s] = t * sn;
	}
	mMr = wf_exp_l(nyq * (0.5 / N), T);
}

#ifndef WC_SYN_WAVE8_OCC
#define WC_SYN_WAVE8_OCC 4
#endif
constexpr int kSyn8Waves = 4;  // wavefronts (pulses) per workgroup
__global__ __launch_bounds__(64 * kSyn8Waves) __attribute__((amdgpu_waves_per_eu(WC_SYN_WAVE8_OCC, WC_SYN_WAVE8_OCC))) void syn_pulse_wave8_kernel(SynArgs a) {
	constexpr int N = 1024, M = 512;
	__shared__ __attribute__((aligned(16))) double Ls[kSyn8Waves][kWf8Lds];
	__shared__ __attribute__((aligned(16))) double T[kWfTabLds];
	__shared__ __attribute__((aligned(16))) double Ps[kSyn8Waves][512];
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	double *L = Ls[wv], *P = Ps[wv];
	if (threadIdx.x < 160) reinterpret_cast<double2 *>(T)[threadIdx.x] = tw_load(a.tw + kTwLog, threadIdx.x);  // 128 + 32 entries, contiguous
	__syncthreads();
	// (the pulses are dealt to the XCDs in contiguous eighths -- workgroup b runs on XCD b mod 8 -- and within an eighth four
	// consecutive pulses to a workgroup: neighbours read the same spectrogram / aperiodicity rows)
	const long long total_p = a.pulse_prefix[a.n_utt];
	const long long per = (total_p + 7) / 8;
	const long long idx = (long long)(blockIdx.x >> 3) * kSyn8Waves + wv;
	if (idx >= per) return;
	const long long gp = (blockIdx.x & 7) * per + idx;
	if (gp >= total_p) return;
	if (a.only_pulse >= 0 && gp != a.only_pulse) return;
	const int u = a.pulse_utt[gp];
	const UttDesc ud = a.utts[u];
	const long long slot = a.cap_off[u] + (gp - a.pulse_prefix[u]);
	const int pidx = a.p.index[slot];
	const double shift = a.p.shift[slot];
	const int noise_size = a.p.noise_size[slot];
	const double vuv = (double)a.p.vuv[slot];
	const int fs = a.fs, Lf = ud.f_len;
	const double fp = a.frame_period;
	const double t = pidx / (double)fs;  // time_axis[ii] (reference :227)

	// ---- spectral envelope / aperiodic ratio at the pulse (reference :346-393), then the two log spectra ----
	const int fl = min(Lf - 1, (int)floor(t / fp));
	const int ce = min(Lf - 1, (int)ceil(t / fp));
	const double ipol = uniform_d(t / fp - fl);
	const double *__restrict__ sf = a.sp + (ud.f_off + fl) * (long long)(M + 1);
	const double *__restrict__ sc = a.sp + (ud.f_off + ce) * (long long)(M + 1);
	const double *__restrict__ af = a.ap + (ud.f_off + fl) * (long long)(M + 1);
	const double *__restrict__ ac = a.ap + (ud.f_off + ce) * (long long)(M + 1);
	const bool same = fl == ce;
	auto blend = [&](double s0, double s1, double a0, double a1, double &env, double &ar) {
		if (same) {
			env = fabs(s0);
			const double s = safe_ap(a0);
			ar = s * s;
		} else {
			env = fma(1.0 - ipol, fabs(s0), ipol * fabs(s1));
			// both products rounded as in the reference (:388-390), no fused multiply-add (see syn_pulse_kernel)
			const double s = (1.0 - ipol) * safe_ap(a0) + ipol * safe_ap(a1);
			ar = s * s;
		}
	};
	double ar0;  // aperiodic_ratio[0] decides whether there is a periodic response (reference :410)
	{
		double env;
		blend(sf[0], sc[0], af[0], ac[0], env, ar0);
		ar0 = uniform_d(ar0);
	}

	// ---- periodic response, then the aperiodic response, through one copy of the code (see syn_pulse_wave_kernel) ----
	double dc = 0.0;
	const double sq = sqrt((double)noise_size);
	double *__restrict__ resp = a.resp + gp * N;
	const bool has_periodic = !(vuv <= 0.5 || ar0 > 0.999);
#pragma unroll 1
	for (int part = has_periodic ? 0 : 1; part < 2; ++part) {
		int ln = lane;
		WC_FRESH(ln);
		double wr[8], wi[8];
		double nr[8], ni[8], nsM = 0.0;
		if (part == 1) {
			// the noise (reference :514-530): noise_size draws from the pulse's place in the stream, mean removed
			const unsigned long long rstart = a.rng_start ? a.rng_start[u] : ud.rng_pos;
			const uint32_t *__restrict__ rng = a.rng_table + (rstart + (unsigned long long)(pidx - a.first_index[u]) - a.rng_base);
			double s = 0.0;
			{
				uint32_t raw[16];
#pragma unroll
				for (int q = 0; q < 8; ++q) {
					if ((q & 3) == 0 && q * 128 >= noise_size) break;
					const int i0 = 2 * ln + 128 * q;
					raw[2 * q] = rng[i0 < noise_size ? i0 : 0];
					raw[2 * q + 1] = rng[i0 + 1 < noise_size ? i0 + 1 : 0];
				}
				WF_SCHED_FENCE();
#pragma unroll
				for (int q = 0; q < 8; ++q) {
					nr[q] = ni[q] = 0.0;
					if (q * 128 < ((noise_size + 511) & ~511)) {
						const int i0 = 2 * ln + 128 * q;
						if (i0 < noise_size) nr[q] = raw[2 * q] / 268435456.0 - 6.0;
						if (i0 + 1 < noise_size) ni[q] = raw[2 * q + 1] / 268435456.0 - 6.0;
						s += nr[q] + ni[q];
					}
				}
			}
			s = wave_sum_all(s);
			const double avg = s / noise_size;
#pragma unroll
			for (int q = 0; q < 8; ++q) {
				const int i0 = 2 * ln + 128 * q;
				nr[q] = (i0 < noise_size) ? nr[q] - avg : 0.0;
				ni[q] = (i0 + 1 < noise_size) ? ni[q] - avg : 0.0;
			}
			if (noise_size <= 256) wdft8p<+1, 1>(nr, ni);
			else if (noise_size <= 512) wdft8p<+1, 2>(nr, ni);
			else wdft8p<+1, 4>(nr, ni);
			wf8_fft512_dit_rest<+1>(nr, ni, L, a.tw, ln);
			wf8_r2c_unpack(nr, ni, nsM, a.tw, ln);  // twice the noise spectrum
			// the noise spectrum waits outside the registers while the minimum phase is worked out: its real parts in LDS, its
			// imaginary parts in the first half of the pulse's own response row (free until the mix below)
#pragma unroll
			for (int sI = 0; sI < 8; ++sI) {
				P[64 * sI + ln] = nr[sI];
				resp[64 * sI + ln] = ni[sI];
			}
			wf_fence();
		}
		// the part's log spectrum from the two rows around the pulse: log(env (1 - ar) + safeguard) / 2 for the periodic part
		// (reference :416-417), log(env ar) / 2 or, unvoiced, log(env) / 2 for the aperiodic one (:490-497)
		double mM;
		{
			auto logspec = [&](double env, double ar) {
				return wf_log_l(part == 0 ? env * (1.0 - ar) + kSafe : (vuv != 0.0 ? env * ar : env), T) / 2.0;
			};
#pragma unroll
			for (int g0 = 0; g0 < 2; ++g0) {
				double v[4][4];
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const int k = wf8_bin(ln, g0, q);
					v[q][0] = sf[k]; v[q][1] = sc[k]; v[q][2] = af[k]; v[q][3] = ac[k];
				}
				WF_SCHED_FENCE();
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					double env, ar;
					blend(v[q][0], v[q][1], v[q][2], v[q][3], env, ar);
					L[wf8_bin(ln, g0, q)] = logspec(env, ar);  // (straight to its place in the transform's input)
				}
			}
			double env, ar;
			blend(sf[M], sc[M], af[M], ac[M], env, ar);
			const double lsM = logspec(env, ar);
			if (ln == 0) L[M] = lsM;
		}
		minimum_phase_wave8(wr, wi, mM, L, T, a.tw, ln);
		double yM;
		if (part == 0) {
			// fractional time shift (reference :443-457)
			const double coef = 2.0 * kPi * shift * fs / N;
#pragma unroll
			for (int g = 0; g < 2; ++g)
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const int k = wf8_bin(ln, g, q);
					double sn_, re2;
					wf_sincos(coef * k, sn_, re2);
					const double im2 = sqrt(1.0 - re2 * re2);
					const double x = wr[4 * g + q], y = wi[4 * g + q];
					wr[4 * g + q] = fma(x, re2, -(y * im2));
					wi[4 * g + q] = fma(x, im2, y * re2);
				}
			double sn_, reM;
			wf_sincos(coef * M, sn_, reM);
			yM = mM * reM;
		} else {
			{
				const double *rp = resp;
				asm volatile("" : "+s"(rp));  // (an opaque pointer: real loads, not the stored values kept in registers)
#pragma unroll
				for (int sI = 0; sI < 8; ++sI) {
					ni[sI] = rp[64 * sI + ln];
					nr[sI] = P[64 * sI + ln];
				}
				WF_SCHED_FENCE();
			}
#pragma unroll
			for (int sI = 0; sI < 8; ++sI) {
				const double x = wr[sI], y = wi[sI], nx = 0.5 * nr[sI], ny = 0.5 * ni[sI];
				wr[sI] = fma(x, nx, -(y * ny));
				wi[sI] = fma(x, ny, y * nx);
			}
			yM = mM * (0.5 * nsM);
		}
		wf8_c2r_pack(wr, wi, yM, a.tw, ln);
		wf8_fft512_dif<-1>(wr, wi, L, a.tw, ln);
		if (part == 0) {
			// DC removal (reference :459-474), see syn_pulse_wave_kernel: wave[0 .. M) is all that is left of the response itself
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				dc += wr[q] + wi[q];
				*reinterpret_cast<double2 *>(resp + M + 2 * ln + 128 * q) = make_double2(wr[q] * sq, wi[q] * sq);
			}
			dc = wave_sum_all(dc);
		} else {
			// ---- mix (reference :339-343): shifted sample j is unshifted sample j - M (j >= M) / j + M; the row in output order ----
			const double dcs = has_periodic ? -dc * sq : 0.0;
			double2 dr[4], pp[4];
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				dr[q] = *reinterpret_cast<const double2 *>(a.dc_remover + 2 * ln + 128 * q);
				pp[q] = has_periodic ? *reinterpret_cast<const double2 *>(resp + M + 2 * ln + 128 * q) : make_double2(0.0, 0.0);
			}
			WF_SCHED_FENCE();
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int n = 2 * ln + 128 * q;  // unshifted samples n, n + 1 (output place n + M) and n + M, n + M + 1 (place n)
				const double r0 = fma(dcs, dr[q].x, wr[q]) + pp[q].x, r1 = fma(dcs, dr[q].y, wi[q]) + pp[q].y;
				*reinterpret_cast<double2 *>(resp + n + M) = make_double2(r0 / N, r1 / N);
				const double r2 = fma(dcs, dr[q].x, wr[q + 4]), r3 = fma(dcs, dr[q].y, wi[q + 4]);
				*reinterpret_cast<double2 *>(resp + n) = make_double2(r2 / N, r3 / N);
			}
		}
	}
}

// utterance of every pulse of the compact numbering (prefix[u] <= gp < prefix[u + 1]); one workgroup per utterance
__global__ void syn_pulse_utt_kernel(const long long *__restrict__ prefix, int *__restrict__ pulse_utt) {
	const int u = blockIdx.x;
	const long long lo = prefix[u], hi = prefix[u + 1];
	for (long long g = lo + threadIdx.x; g < hi; g += blockDim.x) pulse_utt[g] = u;
}

// Overlap-add of the response rows (reference :118-139: y[index + 1 + j] += response[j], pulse after pulse).  One workgroup per
// tile of OA_TILE output samples: the pulses that reach into the tile are a contiguous run of the utterance's (sorted) pulse
// list, found by bisection; every thread adds the rows' samples to its four outputs in pulse order, so y carries the
// reference's own summation order.  Writes every sample of y (zeros where no pulse reaches): no clearing pass.
constexpr int OA_T = 256, OA_K = 4, OA_TILE = OA_T * OA_K;
template <int N>
__global__ __launch_bounds__(OA_T) void syn_overlap_add_kernel(SynArgs a) {
	constexpr int M = N / 2;
	const int u = blockIdx.y;
	const UttDesc ud = a.utts[u];
	const int t0 = blockIdx.x * OA_TILE;
	if (t0 >= ud.y_len) return;
	const long long pre = a.pulse_prefix[u];
	const int n_p = (int)(a.pulse_prefix[u + 1] - pre);
	const int *__restrict__ pidx = a.p.index + a.cap_off[u];
	// pulse i covers outputs pidx[i] - M + 1 .. pidx[i] + M: those with pidx in [t0 - M, t0 + OA_TILE - 2 + M] touch the tile
	int lo = 0, hi = n_p;
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if (pidx[mid] < t0 - M) lo = mid + 1; else hi = mid;
	}
	const int last = t0 + OA_TILE - 2 + M;
	double acc[OA_K];
#pragma unroll
	for (int m = 0; m < OA_K; ++m) acc[m] = 0.0;
	const int o0 = t0 + threadIdx.x;
	const double *__restrict__ rows = a.resp + pre * N;
	constexpr int PB = 4;  // pulses per trip: their loads are requested together, the additions stay in pulse order
	for (int i = lo; i < n_p; i += PB) {
		if (pidx[i] > last) break;
		double v[PB][OA_K];
#pragma unroll
		for (int b = 0; b < PB; ++b) {
			const int ii = min(i + b, n_p - 1);
			const int start = pidx[ii] - M + 1;
			const bool on = i + b < n_p && pidx[ii] <= last;
#pragma unroll
			for (int m = 0; m < OA_K; ++m) {
				const int j = o0 + m * OA_T - start;
				v[b][m] = (on && j >= 0 && j < N) ? rows[(long long)ii * N + j] : 0.0;
			}
		}
#pragma unroll
		for (int b = 0; b < PB; ++b)
#pragma unroll
			for (int m = 0; m < OA_K; ++m) acc[m] += v[b][m];
	}
	double *__restrict__ out = a.out + ud.y_off;
#pragma unroll
	for (int m = 0; m < OA_K; ++m) {
		const int o = o0 + m * OA_T;
		if (o < ud.y_len) out[o] = acc[m];
	}
}

}  // namespace wc

using namespace wc;

struct wc_synthesis {
	int fs, fft_size;
	double frame_period;  // seconds
	Device *dev;
	wc_synthesis *twin = nullptr;  // second half of a large batch (syn_run_device), created on first use
	hipStream_t s_twin = nullptr;
	hipEvent_t e_twin = nullptr;
	bool is_twin = false;
	DevBuf dc_remover, utts, meta, pulses, incs, phase, phase_seg, tile_cnt, resp, pulse_utt, d_f0, d_sp, d_ap, d_out;
	bool pulses_by_utterance;  // WC_SYN_PULSES=utterance: one workgroup walks an utterance's tiles (A/B and the bit-identity test)
	bool wave;  // N = 2048 / 1024: one wavefront per pulse (default; WC_SYN_IMPL=block: the workgroup-per-pulse kernel)
	bool rows = false;  // of the most recent syn_prepare: pulses through response rows + syn_overlap_add_kernel (else atomics into the output)
	bool wave_atomic = false;  // WC_SYN_OLA=atomic: the one-wavefront kernel (N = 2048) adds into the output itself (A/B variant)
	size_t rows_budget = 0;  // bytes the response rows may take (an eighth of the device's memory; WC_SYN_ROWS_BUDGET_MB)
	bool phase_single;  // WC_SYN_PHASE=single: the phase sum by one workgroup per utterance (A/B and the bit-identity test)
	bool serial_timebase;  // WC_SYN_TIMEBASE=serial: the one-wavefront sequential accumulation instead of the exact parallel one
	HostBuf h_stage, h_rows;
	long long total_out = 0, cap_total = 0;  // of the most recent syn_prepare
	int n_utt = 0, max_out = 0;
};

template <int N>
static void launch_pulses(const SynArgs &a, hipStream_t s) {
	// eight samples per thread up to N = 2048 (at N = 1024 / 512 a 256-thread block idles half / three quarters of its threads in
	// every FFT pass: 3.53 -> 3.35 ms per 64 x 10 s at 16 kHz, 2.08 -> 1.53 ms at 8 kHz); 512 threads per 2048-point pulse measured
	// slower (13.2 vs 10.0 ms per 64 x 10 s batch)
	constexpr int TP = (N >= 2048) ? 256 : (N / 8 < 64 ? 64 : N / 8);
	// (the pulses are dealt to the XCDs in eighths: the grid is a multiple of eight, the kernel drops what lies beyond the count)
	hipLaunchKernelGGL((syn_pulse_kernel<N, TP>), dim3((unsigned)(8 * ((a.total_pulses + 7) / 8))), dim3(TP), 0, s, a);
}

// per-utterance pulse prefix, overflow flag and end-of-stage stream positions (one small workgroup)
__global__ void syn_prefix_kernel(int n_utt, const int *__restrict__ count, const int *__restrict__ cap,
								  const int *__restrict__ first_index, const int *__restrict__ last_index,
								  const UttDesc *__restrict__ utts, const unsigned long long *__restrict__ d_start,
								  long long *__restrict__ prefix, unsigned long long *__restrict__ end_pos, int *__restrict__ overflow) {
	if (threadIdx.x != 0) return;
	long long run = 0;
	int ovf = 0;
	for (int u = 0; u < n_utt; ++u) {
		prefix[u] = run;
		if (count[u] > cap[u]) ovf = 1;
		run += min(count[u], cap[u]);
		const unsigned long long st = d_start ? d_start[u] : utts[u].rng_pos;
		end_pos[u] = st + (unsigned long long)(last_index[u] - first_index[u]);  // reference :106-107, :519-521
	}
	prefix[n_utt] = run;
	*overflow = ovf;
}

// Enqueue-only building blocks (no host synchronisation), shared with the fused pipeline:
//   syn_prepare  time base: increments, sequential phase sum, pulse compaction, noise sizes, pulse prefix
//   syn_pulses   the per-pulse kernel over the pulse *capacity* (workgroups beyond the real count exit)
// Overflow of the rate-bounded pulse capacity is reported through sy->overflow (device) and handled by the
// caller by re-running with full == true.
int syn_prepare(wc_synthesis *sy, hipStream_t s, int n_utt, const double *d_f0, const int *f0_length, const int *out_length,
				double *d_out, const uint64_t *rng_pos, bool full) {
	Device *dev = sy->dev;
	std::vector<UttDesc> utts(n_utt);
	long long fo = 0, yo = 0;
	int max_out = 0;
	for (int u = 0; u < n_utt; ++u) {
		if (f0_length[u] < 2) return fail(WC_ERR_INVALID, "synthesis: f0_length must be at least 2 (reference src/synthesis.cpp:241-242)");
		if (out_length[u] < 0) return fail(WC_ERR_INVALID, "synthesis: negative out_length");
		UttDesc &t = utts[u];
		t.x_off = 0; t.f_off = fo; t.y_off = yo; t.x_len = 0; t.f_len = f0_length[u]; t.y_len = out_length[u]; t.pad = 0;
		t.rng_pos = rng_pos ? rng_pos[u] : 0ull;
		fo += f0_length[u];
		yo += out_length[u];
		max_out = std::max(max_out, out_length[u]);
	}
	const long long total_out = yo;
	sy->total_out = total_out;
	sy->n_utt = n_utt;
	sy->max_out = max_out;
	sy->cap_total = 0;
	if (total_out == 0) return WC_OK;
	int rc;
	// meta layout (device): cap_off[n] (i64) | pulse_prefix[n+1] (i64) | inc_off[n] (i64) | end_pos[n] (u64) |
	//                       cap[n] | count[n] | first_index[n] | last_index[n] | overflow
	const size_t meta_bytes = sizeof(long long) * (4 * (size_t)n_utt + 1) + sizeof(int) * (4 * (size_t)n_utt + 1);
	if ((rc = sy->utts.reserve(sizeof(UttDesc) * n_utt))) return rc;
	if ((rc = sy->meta.reserve(meta_bytes))) return rc;
	if ((rc = sy->h_stage.reserve(sizeof(UttDesc) * n_utt + meta_bytes))) return rc;
	char *hs = static_cast<char *>(sy->h_stage.p);
	std::memcpy(hs, utts.data(), sizeof(UttDesc) * n_utt);
	long long *h_cap_off = reinterpret_cast<long long *>(hs + sizeof(UttDesc) * n_utt);
	long long *h_inc_off = h_cap_off + 2 * n_utt + 1;
	int *h_cap = reinterpret_cast<int *>(h_cap_off + 4 * n_utt + 1);
	char *dm = static_cast<char *>(sy->meta.p);
	long long *d_cap_off = reinterpret_cast<long long *>(dm);
	long long *d_prefix = d_cap_off + n_utt;
	long long *d_inc_off = d_prefix + n_utt + 1;
	unsigned long long *d_end = reinterpret_cast<unsigned long long *>(d_inc_off + n_utt);
	int *d_cap = reinterpret_cast<int *>(d_end + n_utt);
	int *d_count = d_cap + n_utt;
	int *d_first = d_count + n_utt;
	int *d_last = d_first + n_utt;
	long long inc_total = 0, co = 0;
	for (int u = 0; u < n_utt; ++u) {
		h_inc_off[u] = inc_total;
		inc_total += ((long long)out_length[u] + 63) / 64 * 64 + 64;
		// pulses <= samples; the rate bound covers F0 up to 960 Hz and the 500 Hz unvoiced pulses down to 8 kHz
		const long long soft = (long long)out_length[u] * 960 / sy->fs + 16;
		const int cap = (full || soft > out_length[u]) ? out_length[u] + 1 : (int)soft;
		h_cap_off[u] = co;
		h_cap[u] = cap;
		co += cap;
	}
	sy->cap_total = co;
	// The one-wavefront pulse kernels go through a response row per pulse slot and syn_overlap_add_kernel, which writes every
	// sample.  The rows are reserved for the pulse CAPACITY (the real count is known on the device only): 157 MB per 10 s utterance
	// at 48 kHz under the rate bound, but out_length + 1 rows per utterance on the overflow retry -- beyond the budget the
	// atomics kernel takes over (into a cleared output) instead of an allocation that cannot succeed.
	sy->rows = sy->wave && (sy->fft_size == 2048 || sy->fft_size == 1024) &&
			   (double)co * sy->fft_size * sizeof(double) <= (double)sy->rows_budget;
	if (!sy->rows || (sy->wave_atomic && sy->fft_size == 2048)) WC_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * total_out, s));
	if ((rc = sy->incs.reserve(sizeof(double) * inc_total))) return rc;
	if (!sy->serial_timebase && (rc = sy->phase.reserve(sizeof(double) * inc_total))) return rc;
	if ((rc = sy->pulses.reserve((size_t)co * (sizeof(int) * 3 + sizeof(double))))) return rc;
	PulseBuf pb;
	pb.shift = sy->pulses.as<double>();
	pb.index = reinterpret_cast<int *>(pb.shift + co);
	pb.noise_size = pb.index + co;
	pb.vuv = pb.noise_size + co;
	WC_HIP(hipMemcpyAsync(sy->utts.p, hs, sizeof(UttDesc) * n_utt, hipMemcpyHostToDevice, s));
	WC_HIP(hipMemcpyAsync(d_cap_off, h_cap_off, sizeof(long long) * n_utt, hipMemcpyHostToDevice, s));
	WC_HIP(hipMemcpyAsync(d_inc_off, h_inc_off, sizeof(long long) * n_utt, hipMemcpyHostToDevice, s));
	WC_HIP(hipMemcpyAsync(d_cap, h_cap, sizeof(int) * n_utt, hipMemcpyHostToDevice, s));
	if ((rc = sy->h_stage.mark(s))) return rc;
	TbArgs ta;
	ta.inc_off = d_inc_off;
	ta.utts = sy->utts.as<UttDesc>(); ta.f0 = d_f0; ta.cap_off = d_cap_off; ta.cap = d_cap; ta.p = pb;
	ta.count = d_count; ta.first_index = d_first; ta.fs = sy->fs; ta.fft_size = sy->fft_size;
	ta.frame_period = sy->frame_period;
	if ((rc = dev->time_begin("synthesis_timebase", s))) return rc;
	WC_HIP(hipMemsetAsync(sy->incs.p, 0, sizeof(double) * inc_total, s));
	hipLaunchKernelGGL(syn_increment_kernel, dim3((unsigned)((total_out + 255) / 256)), dim3(256), 0, s, ta, n_utt,
					   total_out, sy->incs.as<double>());
	if (sy->serial_timebase) {
		hipLaunchKernelGGL(syn_timebase_kernel, dim3(n_utt), dim3(64), 0, s, ta, (const double *)sy->incs.as<double>());
	} else {
		const int nseg_max = (max_out + PH_W - 1) / PH_W;
		if (sy->phase_single || nseg_max > PH_MAXSEG) {
			hipLaunchKernelGGL(syn_phase_kernel, dim3(n_utt), dim3(TB_T), 0, s, ta, (const double *)sy->incs.as<double>(), sy->phase.as<double>());
		} else {
			const size_t per = (size_t)nseg_max * n_utt;
			if ((rc = sy->phase_seg.reserve(per * (3 * sizeof(double) + sizeof(int))))) return rc;
			PhArgs pa;
			pa.t = ta; pa.inc = sy->incs.as<double>(); pa.phase = sy->phase.as<double>();
			pa.segsum = sy->phase_seg.as<double>();
			pa.segD = reinterpret_cast<unsigned long long *>(pa.segsum + per);
			pa.segM0 = pa.segD + per;
			pa.segflag = reinterpret_cast<int *>(pa.segM0 + per);
			pa.nseg_max = nseg_max;
			hipLaunchKernelGGL(syn_phase_est_kernel, dim3(nseg_max, n_utt), dim3(PH_T), 0, s, pa);
			hipLaunchKernelGGL(syn_phase_local_kernel, dim3(nseg_max, n_utt), dim3(PH_T), 0, s, pa);
			hipLaunchKernelGGL(syn_phase_walk_kernel, dim3(n_utt), dim3(PH_T), 0, s, pa);
			hipLaunchKernelGGL(syn_phase_apply_kernel, dim3(nseg_max, n_utt), dim3(PH_T), 0, s, pa);
		}
		if (sy->pulses_by_utterance) {
			hipLaunchKernelGGL(syn_pulses_from_phase_kernel, dim3(n_utt), dim3(TB_T), 0, s, ta, (const double *)sy->incs.as<double>(),
							   (const double *)sy->phase.as<double>());
		} else {
			const int max_tiles = (max_out + TB_T * PT_K - 1) / (TB_T * PT_K);
			if ((rc = sy->tile_cnt.reserve(sizeof(int) * (size_t)max_tiles * n_utt))) return rc;
			// (an utterance without output samples has no tile that writes its count: it must not keep a stale one)
			WC_HIP(hipMemsetAsync(d_count, 0, sizeof(int) * n_utt, s));
			hipLaunchKernelGGL(syn_pulse_tiles_kernel<0>, dim3(max_tiles, n_utt), dim3(TB_T), 0, s, ta, (const double *)sy->incs.as<double>(),
							   (const double *)sy->phase.as<double>(), sy->tile_cnt.as<int>(), max_tiles);
			hipLaunchKernelGGL(syn_pulse_tiles_kernel<1>, dim3(max_tiles, n_utt), dim3(TB_T), 0, s, ta, (const double *)sy->incs.as<double>(),
							   (const double *)sy->phase.as<double>(), sy->tile_cnt.as<int>(), max_tiles);
		}
	}
	hipLaunchKernelGGL(syn_noise_size_kernel, dim3(8, n_utt), dim3(256), 0, s, d_cap_off, d_count, d_cap, pb, d_first, d_last);
	WC_HIP(hipGetLastError());
	return dev->time_end("synthesis_timebase", s);
}

int syn_pulses(wc_synthesis *sy, hipStream_t s, const double *d_f0, const double *d_sp, const double *d_ap, double *d_out,
			   const unsigned long long *d_start) {
	Device *dev = sy->dev;
	if (sy->total_out == 0 || sy->cap_total == 0) return WC_OK;
	const int n_utt = sy->n_utt;
	int rc;
	char *dm = static_cast<char *>(sy->meta.p);
	long long *d_cap_off = reinterpret_cast<long long *>(dm);
	long long *d_prefix = d_cap_off + n_utt;
	long long *d_inc_off = d_prefix + n_utt + 1;
	unsigned long long *d_end = reinterpret_cast<unsigned long long *>(d_inc_off + n_utt);
	int *d_cap = reinterpret_cast<int *>(d_end + n_utt);
	int *d_count = d_cap + n_utt;
	int *d_first = d_count + n_utt;
	int *d_last = d_first + n_utt;
	int *d_ovf = d_last + n_utt;
	const long long co = sy->cap_total;
	PulseBuf pb;
	pb.shift = sy->pulses.as<double>();
	pb.index = reinterpret_cast<int *>(pb.shift + co);
	pb.noise_size = pb.index + co;
	pb.vuv = pb.noise_size + co;
	hipLaunchKernelGGL(syn_prefix_kernel, dim3(1), dim3(64), 0, s, n_utt, d_count, d_cap, d_first, d_last, sy->utts.as<UttDesc>(),
					   d_start, d_prefix, d_end, d_ovf);
	SynArgs a;
	a.utts = sy->utts.as<UttDesc>(); a.n_utt = n_utt; a.pulse_prefix = d_prefix; a.cap_off = d_cap_off;
	a.first_index = d_first; a.p = pb; a.f0 = d_f0; a.sp = d_sp; a.ap = d_ap;
	a.rng_table = dev->rng_table.as<uint32_t>(); a.rng_base = dev->rng_base; a.tw = dev->twiddle;
	a.dc_remover = sy->dc_remover.as<double>(); a.out = d_out; a.total_pulses = co; a.fs = sy->fs;
	a.frame_period = sy->frame_period; a.rng_start = d_start;
#ifdef WC_DEBUG_HOOKS  // output-altering debugging aid: compiled in only on request (WC_EXTRA_FLAGS=-DWC_DEBUG_HOOKS)
	{
		const char *dbg = getenv("WC_DEBUG_ONLY_PULSE");
		a.only_pulse = dbg ? atoll(dbg) : -1;
	}
#else
	a.only_pulse = -1;
#endif
	a.trace = nullptr;
	a.resp = nullptr;
	a.pulse_utt = nullptr;
	if (sy->rows) {
		// a response row per pulse slot of the rate bound (only the rows of real pulses are ever touched)
		const bool atomic_rows = sy->wave_atomic && sy->fft_size == 2048;  // (only the parked noise half then: 1024 doubles per pulse)
		if ((rc = sy->resp.reserve(sizeof(double) * (size_t)(atomic_rows ? sy->fft_size / 2 : sy->fft_size) * (size_t)co))) return rc;
		a.resp = sy->resp.as<double>();
		if ((rc = sy->pulse_utt.reserve(sizeof(int) * (size_t)co))) return rc;
		a.pulse_utt = sy->pulse_utt.as<int>();
		hipLaunchKernelGGL(syn_pulse_utt_kernel, dim3(n_utt), dim3(256), 0, s, (const long long *)d_prefix, sy->pulse_utt.as<int>());
	}
#if WC_SYN_TRACE
	static DevBuf tracebuf;
	if (tracebuf.reserve(sizeof(unsigned long long) * 16 * (size_t)co)) return WC_ERR_DEVICE;
	(void)hipMemsetAsync(tracebuf.p, 0, sizeof(unsigned long long) * 16 * (size_t)co, s);
	a.trace = tracebuf.as<unsigned long long>();
#endif
	if ((rc = dev->time_begin("synthesis_pulses", s))) return rc;
	switch (sy->fft_size) {
		case 512: launch_pulses<512>(a, s); break;
		case 1024:
			if (sy->rows) {
				const long long per = (a.total_pulses + 7) / 8;
				hipLaunchKernelGGL(syn_pulse_wave8_kernel, dim3((unsigned)(8 * ((per + kSyn8Waves - 1) / kSyn8Waves))), dim3(64 * kSyn8Waves), 0, s, a);
				hipLaunchKernelGGL(syn_overlap_add_kernel<1024>, dim3((unsigned)((sy->max_out + OA_TILE - 1) / OA_TILE), n_utt), dim3(OA_T), 0, s, a);
			} else {
				launch_pulses<1024>(a, s);
			}
			break;
		case 2048:
			if (sy->rows) {
				if (sy->wave_atomic) {
					hipLaunchKernelGGL(syn_pulse_wave_kernel<true>, dim3((unsigned)(8 * ((a.total_pulses + 7) / 8))), dim3(64), 0, s, a);
				} else {
					hipLaunchKernelGGL(syn_pulse_wave_kernel<false>, dim3((unsigned)(8 * ((a.total_pulses + 7) / 8))), dim3(64), 0, s, a);
					hipLaunchKernelGGL(syn_overlap_add_kernel<2048>, dim3((unsigned)((sy->max_out + OA_TILE - 1) / OA_TILE), n_utt), dim3(OA_T), 0, s, a);
				}
			} else {
				launch_pulses<2048>(a, s);
			}
			break;
		case 4096: launch_pulses<4096>(a, s); break;
		default: return fail(WC_ERR_UNSUPPORTED, "synthesis: fft_size must be 512, 1024, 2048 or 4096");
	}
	WC_HIP(hipGetLastError());
#if WC_SYN_TRACE
	if (const char *path = getenv("WC_SYN_TRACE_FILE")) {
		std::vector<unsigned long long> h((size_t)16 * co);
		WC_HIP(hipStreamSynchronize(s));
		WC_HIP(hipMemcpy(h.data(), a.trace, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
		if (FILE *f = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
	}
#endif
	return dev->time_end("synthesis_pulses", s);
}

// after the stream has been synchronised: overflow flag and (optionally) the end positions
int syn_finish(wc_synthesis *sy, hipStream_t s, uint64_t *rng_pos_out, bool *overflow) {
	*overflow = false;
	if (sy->total_out == 0 || sy->cap_total == 0) return WC_OK;
	const int n_utt = sy->n_utt;
	char *dm = static_cast<char *>(sy->meta.p);
	long long *d_cap_off = reinterpret_cast<long long *>(dm);
	unsigned long long *d_end = reinterpret_cast<unsigned long long *>(d_cap_off + 3 * n_utt + 1);
	int *d_ovf = reinterpret_cast<int *>(d_end + n_utt) + 4 * n_utt;
	int ovf = 0;
	std::vector<uint64_t> ends(n_utt);
	WC_HIP(hipMemcpyAsync(&ovf, d_ovf, sizeof(int), hipMemcpyDeviceToHost, s));
	if (rng_pos_out) WC_HIP(hipMemcpyAsync(ends.data(), d_end, sizeof(uint64_t) * n_utt, hipMemcpyDeviceToHost, s));
	WC_HIP(hipStreamSynchronize(s));
	*overflow = ovf != 0;
	if (rng_pos_out && !ovf) for (int u = 0; u < n_utt; ++u) rng_pos_out[u] = ends[u];
	return WC_OK;
}

static int syn_run_device(wc_synthesis *sy, int n_utt, const double *d_f0, const int *f0_length, const double *d_sp,
						  const double *d_ap, const int *out_length, double *d_out, uint64_t *rng_pos) {
	Device *dev = sy->dev;
	hipStream_t s = dev->active();
	int rc;
	uint64_t lo = ~0ull, hi = 0;
	for (int u = 0; u < n_utt; ++u) {
		uint64_t p0 = rng_pos ? rng_pos[u] : 0ull;
		lo = p0 < lo ? p0 : lo;
		uint64_t e = p0 + (uint64_t)(out_length[u] > 0 ? out_length[u] : 0);
		hi = e > hi ? e : hi;
	}
	if ((rc = dev->ensure_rng(lo, hi))) return rc;
	// A large batch runs as two halves with a twin handle on a second stream (round 4): the second half's time base -- one
	// wavefront or workgroup per utterance, ~1 ms of latency whatever the batch -- runs beside the first half's pulses instead of
	// in front of everything (BASELINE config 4, 128 utterances: 10.4 -> 9.7 ms).  WC_SYN_HALVES=0: one piece.
	const bool halves_on = !(getenv("WC_SYN_HALVES") && getenv("WC_SYN_HALVES")[0] == '0');  // (read per call: the tests switch it)
	if (halves_on && n_utt >= 16 && !sy->is_twin) {
		if (!sy->twin) {
			OnDeviceOf here(sy->dev);
			sy->twin = wc_synthesis_create(sy->fs, sy->fft_size, sy->frame_period * 1000.0);
			if (!sy->twin) return WC_ERR_DEVICE;
			sy->twin->is_twin = true;
			WC_HIP(hipStreamCreateWithFlags(&sy->s_twin, hipStreamNonBlocking));
			WC_HIP(hipEventCreateWithFlags(&sy->e_twin, hipEventDisableTiming));
		}
		const int nA = n_utt / 2, nB = n_utt - nA;
		const int bins = sy->fft_size / 2 + 1;
		long long foA = 0, yoA = 0;
		for (int u = 0; u < nA; ++u) { foA += f0_length[u]; yoA += out_length[u] > 0 ? out_length[u] : 0; }
		wc_synthesis *syB = sy->twin;
		hipStream_t sB = sy->s_twin;
		// The start positions are read from a private copy and the end positions collected in one: rng_pos is also the output, a
		// half that overflowed is run again from the SAME start positions, and the caller's array is written once, when both halves
		// are through (a half's end positions written early would be the other attempt's start positions: ADVICE round 4).
		std::vector<uint64_t> rng_in, rng_end;
		if (rng_pos) { rng_in.assign(rng_pos, rng_pos + n_utt); rng_end = rng_in; }
		const uint64_t *startA = rng_pos ? rng_in.data() : nullptr, *startB = rng_pos ? rng_in.data() + nA : nullptr;
		uint64_t *endA = rng_pos ? rng_end.data() : nullptr, *endB = rng_pos ? rng_end.data() + nA : nullptr;
		// (an error between the enqueues leaves work on the twin's stream that still writes into the caller's d_out: nothing of
		// this call may be moving when the caller gets its buffers back)
		auto bail = [&](int code) { (void)hipStreamSynchronize(sB); (void)hipStreamSynchronize(s); dev->time_tag = -1; return code; };
		bool fullA = false, fullB = false, runA = true, runB = true;
		for (int attempt = 0; attempt < 2; ++attempt) {
			WC_HIP(hipEventRecord(sy->e_twin, s));  // the twin's stream starts behind whatever precedes this call on the caller's
			WC_HIP(hipStreamWaitEvent(sB, sy->e_twin, 0));
			dev->time_tag = 0;  // (wc_last_kernel_ms sums the two halves' launches, as with the pipeline's groups)
			if (runA && (rc = syn_prepare(sy, s, nA, d_f0, f0_length, out_length, d_out, startA, fullA))) return bail(rc);
			dev->time_tag = 1;
			if (runB && (rc = syn_prepare(syB, sB, nB, d_f0 + foA, f0_length + nA, out_length + nA, d_out + yoA, startB, fullB))) return bail(rc);
			dev->time_tag = 0;
			if (runA && (rc = syn_pulses(sy, s, d_f0, d_sp, d_ap, d_out, nullptr))) return bail(rc);
			dev->time_tag = 1;
			WC_HIP(hipEventRecord(sy->e_twin, s));  // the two halves' pulses one after the other: full-grid kernels gain nothing side by side
			WC_HIP(hipStreamWaitEvent(sB, sy->e_twin, 0));
			if (runB && (rc = syn_pulses(syB, sB, d_f0 + foA, d_sp + foA * bins, d_ap + foA * bins, d_out + yoA, nullptr))) return bail(rc);
			dev->time_tag = -1;
			bool oA = false, oB = false;
			if (runA && (rc = syn_finish(sy, s, endA, &oA))) return bail(rc);
			if (runB && (rc = syn_finish(syB, sB, endB, &oB))) return bail(rc);  // (synchronises the twin's stream)
			if (!runA) WC_HIP(hipStreamSynchronize(s));
			if (!runB) WC_HIP(hipStreamSynchronize(sB));
			if (!oA && !oB) {
				if (rng_pos) std::copy(rng_end.begin(), rng_end.end(), rng_pos);
				return WC_OK;
			}
			// only the half that overflowed runs again (with the hard bound); the other half's samples and end positions stand
			runA = oA; runB = oB;
			fullA = fullA || oA;
			fullB = fullB || oB;
		}
		return fail(WC_ERR_DEVICE, "synthesis: pulse buffer overflow");
	}
	for (int attempt = 0; attempt < 2; ++attempt) {
		if ((rc = syn_prepare(sy, s, n_utt, d_f0, f0_length, out_length, d_out, rng_pos, attempt == 1))) return rc;
		if ((rc = syn_pulses(sy, s, d_f0, d_sp, d_ap, d_out, nullptr))) return rc;
		bool overflow = false;
		if ((rc = syn_finish(sy, s, rng_pos, &overflow))) return rc;
		if (!overflow) return WC_OK;
	}
	return fail(WC_ERR_DEVICE, "synthesis: pulse buffer overflow");
}

wc::Device *syn_device(const wc_synthesis *sy) { return sy->dev; }

extern "C" {

wc_synthesis *wc_synthesis_create(int fs, int fft_size, double frame_period_ms) {
	if (fs <= 0 || frame_period_ms <= 0) { set_error("synthesis: fs and frame_period must be positive"); return nullptr; }
	if (fft_size != 512 && fft_size != 1024 && fft_size != 2048 && fft_size != 4096) {
		set_error("synthesis: fft_size must be 512, 1024, 2048 or 4096");
		return nullptr;
	}
	Device *dev = current_device();
	if (!dev) return nullptr;
	wc_synthesis *s = new wc_synthesis();
	s->fs = fs;
	s->fft_size = fft_size;
	s->frame_period = frame_period_ms / 1000.;  // reference :31
	{
		const char *tb = getenv("WC_SYN_TIMEBASE");
		s->serial_timebase = tb && std::string(tb) == "serial";
		const char *ph = getenv("WC_SYN_PHASE");
		s->phase_single = ph && std::string(ph) == "single";
		const char *pu = getenv("WC_SYN_PULSES");
		s->pulses_by_utterance = pu && std::string(pu) == "utterance";
		const char *impl = getenv("WC_SYN_IMPL");
		s->wave = !(impl && std::string(impl) == "block");
		const char *ola = getenv("WC_SYN_OLA");
		s->wave_atomic = ola && std::string(ola) == "atomic";
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)64 << 30;
		s->rows_budget = total_b / 8;
		if (const char *mb = getenv("WC_SYN_ROWS_BUDGET_MB")) s->rows_budget = (size_t)atoll(mb) << 20;
	}
	s->dev = dev;
	// getDCRemover, reference :290-303
	std::vector<double> d(fft_size);
	const double cv = 2.0 * 3.1415926535897932384 / (1.0 + fft_size);
	for (int i = 0; i < fft_size / 2; ++i) d[i] = 0.5 - 0.5 * std::cos(cv * (i + 1.0));
	double dc = 0.0;
	for (int i = 0; i < fft_size / 2; ++i) dc += d[i];
	dc *= 2;
	for (int i = 0; i < fft_size / 2; ++i) { d[i] /= dc; d[fft_size - i - 1] = d[i]; }
	if (s->dc_remover.reserve(sizeof(double) * fft_size) ||
		hipMemcpy(s->dc_remover.p, d.data(), sizeof(double) * fft_size, hipMemcpyHostToDevice) != hipSuccess) {
		set_error("synthesis: table upload failed");
		delete s;
		return nullptr;
	}
	dev->handle_born();
	return s;
}
void wc_synthesis_destroy(wc_synthesis *s) {
	if (!s) return;
	s->dev->quiesce();
	s->dev->handle_gone();
	if (s->twin) wc_synthesis_destroy(s->twin);
	if (s->s_twin) (void)hipStreamDestroy(s->s_twin);
	if (s->e_twin) (void)hipEventDestroy(s->e_twin);
	s->dc_remover.release(); s->utts.release(); s->meta.release(); s->pulses.release(); s->incs.release(); s->phase.release(); s->tile_cnt.release(); s->phase_seg.release(); s->resp.release(); s->pulse_utt.release();
	s->d_f0.release(); s->d_sp.release(); s->d_ap.release(); s->d_out.release(); s->h_stage.release(); s->h_rows.release();
	delete s;
}

int wc_synthesis_get_fft_size(const wc_synthesis *s) { return s ? s->fft_size : WC_ERR_INVALID; }

int wc_synthesis_compute_device(wc_synthesis *s, int n_utt, const double *d_f0, const int *f0_length, const double *d_sp,
								const double *d_ap, const int *out_length, double *d_out, uint64_t *rng_pos) {
	if (!s || n_utt <= 0 || !d_f0 || !f0_length || !d_sp || !d_ap || !out_length || !d_out)
		return fail(WC_ERR_INVALID, "synthesis: null argument");
	WC_HIP(hipSetDevice(s->dev->id));
	DeviceLock lock(s->dev);
	return syn_run_device(s, n_utt, d_f0, f0_length, d_sp, d_ap, out_length, d_out, rng_pos);
}

int wc_synthesis_compute(wc_synthesis *s, const double *f0, int f0_length, const double *const *spectrogram,
						 const double *const *aperiodicity, int out_length, double *out) {
	if (!s || !f0 || !spectrogram || !aperiodicity || !out) return fail(WC_ERR_INVALID, "synthesis: null argument");
	if (f0_length < 2) return fail(WC_ERR_INVALID, "synthesis: f0_length must be at least 2");
	if (out_length <= 0) return WC_OK;
	WC_HIP(hipSetDevice(s->dev->id));
	DeviceLock lock(s->dev);
	hipStream_t st = s->dev->active();
	const int bins = s->fft_size / 2 + 1;
	int rc;
	if ((rc = s->d_f0.reserve(sizeof(double) * f0_length))) return rc;
	if ((rc = s->d_sp.reserve(sizeof(double) * (size_t)f0_length * bins))) return rc;
	if ((rc = s->d_ap.reserve(sizeof(double) * (size_t)f0_length * bins))) return rc;
	if ((rc = s->d_out.reserve(sizeof(double) * out_length))) return rc;
	// the rows are gathered into page-locked staging by a few threads (runs of rows that lie one behind the other as one piece)
	// and go up from there; the waveform comes down into staging as well (a pageable destination takes the slow path)
	const size_t n_rows = (size_t)f0_length * bins;
	if ((rc = s->h_rows.reserve(sizeof(double) * (2 * n_rows + (size_t)out_length)))) return rc;
	double *hsp = s->h_rows.as<double>(), *hap = hsp + n_rows, *hy = hap + n_rows;
	WC_HIP(hipMemcpyAsync(s->d_f0.p, f0, sizeof(double) * f0_length, hipMemcpyHostToDevice, st));
	if ((rc = rows_up(st, spectrogram, f0_length, bins, hsp, s->d_sp.as<double>()))) return rc;
	if ((rc = rows_up(st, aperiodicity, f0_length, bins, hap, s->d_ap.as<double>()))) return rc;
	uint64_t pos = global_rng_position();
	rc = syn_run_device(s, 1, s->d_f0.as<double>(), &f0_length, s->d_sp.as<double>(), s->d_ap.as<double>(), &out_length,
						s->d_out.as<double>(), &pos);
	if (rc) return rc;
	set_global_rng_position(pos);
	WC_HIP(hipMemcpyAsync(hy, s->d_out.p, sizeof(double) * out_length, hipMemcpyDeviceToHost, st));
	WC_HIP(hipStreamSynchronize(st));
	parallel_copy({{out, hy, sizeof(double) * (size_t)out_length}});
	return WC_OK;
}

}  // extern "C"
