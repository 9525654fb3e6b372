// Register-resident FFT for ONE wavefront (gfx950, wave64, FP64): a 1024-point complex transform whose 16 points
// per lane live in VGPRs from the first butterfly to the last.  The data cross the lanes twice through a 9 KB
// exchange buffer in LDS (real and imaginary parts one after the other, so the buffer is half the transform) and
// never meet a workgroup barrier: a wavefront's LDS instructions execute in issue order, so a read that follows
// the writes of all 64 lanes sees them.  Three radix stages (16 x 16 x 4) instead of five radix-4 round trips
// through LDS; all index arithmetic is folded into immediate offsets.
//
// Conventions are the reference's (reference src/world_fft.cpp:31-77): sign S = +1 is its "forward" e^{+i}.
//
// Layouts (t = lane):
//   strided   slot q (0..15) holds element t + 64 q
//   paired    slot 4 g + q (g = 0..3: A, B, C, D; q = 0..3) holds element j_g + 256 q with
//                 j_A = t,  j_B = 256 - t (lane 0: 128),  j_C = 64 + t,  j_D = 192 - t
//             so that element k and element 1024 - k always sit in the same lane (A_q with B_{3-q}, C_q with D_{3-q};
//             lane 0 pairs A_1 with A_3, B_0 with B_3, B_1 with B_2 and keeps the self-paired 0 and 512 in A_0, A_2):
//             the unpacking passes of the real transforms need no further exchange.
//   wf_fft1024_dit  strided -> paired  (decimation in time)
//   wf_fft1024_dif  paired -> strided  (the transposed factorisation)
#pragma once
#include "wc_device.hpp"

namespace wc {

constexpr int kWfLds = 1152;  // doubles of LDS per wavefront: 1024 + one pad per 16, rounded up (also holds 1025 + 2 * 63 terms)
constexpr double kC8 = 0.92387953251128675613;  // cos(pi/8)
constexpr double kS8 = 0.38268343236508977173;  // sin(pi/8)
constexpr double kH = 0.70710678118654752440;   // sqrt(1/2)

// Orders this lane's LDS accesses for the compiler.  The hardware needs nothing: the wavefront's DS instructions
// execute in order, and a workgroup is one wavefront.
#ifndef WC_WF_SYNC
#define WC_WF_SYNC 0  // 1: __syncthreads() instead (a wait for all outstanding LDS traffic; debugging aid)
#endif
__device__ __forceinline__ void wf_fence() {
#if WC_WF_SYNC
	__syncthreads();
#else
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// ---- 4- and 16-point DFTs on split real / imaginary registers ------------------------------------------------------
// y[q] = sum_r x[r] e^{S 2 pi i r q / 4}, in place
template <int S>
__device__ __forceinline__ void wdft4(double &ar, double &ai, double &br, double &bi, double &cr, double &ci, double &dr,
									  double &di) {
	const double s02r = ar + cr, s02i = ai + ci, d02r = ar - cr, d02i = ai - ci;
	const double s13r = br + dr, s13i = bi + di, e13r = br - dr, e13i = bi - di;
	ar = s02r + s13r; ai = s02i + s13i;
	cr = s02r - s13r; ci = s02i - s13i;
	if (S > 0) { br = d02r - e13i; bi = d02i + e13r; dr = d02r + e13i; di = d02i - e13r; }
	else       { br = d02r + e13i; bi = d02i - e13r; dr = d02r - e13i; di = d02i + e13r; }
}
// the same with x[2] = x[3] = 0 / x[1] = x[2] = x[3] = 0 (leading stage of a transform whose tail is zero padding)
template <int S>
__device__ __forceinline__ void wdft4_2(double &ar, double &ai, double &br, double &bi, double &cr, double &ci, double &dr,
										double &di) {
	const double xr = ar, xi = ai, yr = br, yi = bi;
	ar = xr + yr; ai = xi + yi;
	cr = xr - yr; ci = xi - yi;
	if (S > 0) { br = xr - yi; bi = xi + yr; dr = xr + yi; di = xi - yr; }
	else       { br = xr + yi; bi = xi - yr; dr = xr - yi; di = xi + yr; }
}
// (x + i y) (cr + i ci)
__device__ __forceinline__ void wrot(double &x, double &y, double cr, double ci) {
	const double nx = fma(x, cr, -(y * ci));
	y = fma(x, ci, y * cr);
	x = nx;
}
// 16-point DFT, sign S, natural order in and out.  NG: only the first 4 NG inputs are non-zero (NG = 4: all).
//   r = 4 a + b, q = c + 4 d:  y[c + 4 d] = sum_b (S i)^{b d} w16^{b c} sum_a (S i)^{a c} x[4 a + b],  w16 = e^{S 2 pi i / 16}
template <int S, int NG = 4>
__device__ __forceinline__ void wdft16(double (&xr)[16], double (&xi)[16]) {
	// over a, for each b: slot 4 c + b <- u_b[c]
#pragma unroll
	for (int b = 0; b < 4; ++b) {
		if constexpr (NG >= 3) {
			wdft4<S>(xr[b], xi[b], xr[4 + b], xi[4 + b], xr[8 + b], xi[8 + b], xr[12 + b], xi[12 + b]);
		} else if constexpr (NG == 2) {
			wdft4_2<S>(xr[b], xi[b], xr[4 + b], xi[4 + b], xr[8 + b], xi[8 + b], xr[12 + b], xi[12 + b]);
		} else {
			xr[4 + b] = xr[8 + b] = xr[12 + b] = xr[b];
			xi[4 + b] = xi[8 + b] = xi[12 + b] = xi[b];
		}
	}
	// slot 4 c + b *= w16^{b c}
	constexpr double s = S > 0 ? 1.0 : -1.0;
	// c = 1: w^1, w^2, w^3
	wrot(xr[5], xi[5], kC8, s * kS8);
	{ const double x = xr[6], y = xi[6]; xr[6] = kH * (S > 0 ? x - y : x + y); xi[6] = kH * (S > 0 ? x + y : y - x); }
	wrot(xr[7], xi[7], kS8, s * kC8);
	// c = 2: w^2, w^4, w^6
	{ const double x = xr[9], y = xi[9]; xr[9] = kH * (S > 0 ? x - y : x + y); xi[9] = kH * (S > 0 ? x + y : y - x); }
	{ const double x = xr[10], y = xi[10]; xr[10] = S > 0 ? -y : y; xi[10] = S > 0 ? x : -x; }
	{ const double x = xr[11], y = xi[11]; xr[11] = S > 0 ? -kH * (x + y) : kH * (y - x); xi[11] = S > 0 ? kH * (x - y) : -kH * (x + y); }
	// c = 3: w^3, w^6, w^9
	wrot(xr[13], xi[13], kS8, s * kC8);
	{ const double x = xr[14], y = xi[14]; xr[14] = S > 0 ? -kH * (x + y) : kH * (y - x); xi[14] = S > 0 ? kH * (x - y) : -kH * (x + y); }
	wrot(xr[15], xi[15], -kC8, -s * kS8);
	// over b, for each c: slot 4 c + d <- y[c + 4 d]
#pragma unroll
	for (int c = 0; c < 4; ++c)
		wdft4<S>(xr[4 * c], xi[4 * c], xr[4 * c + 1], xi[4 * c + 1], xr[4 * c + 2], xi[4 * c + 2], xr[4 * c + 3], xi[4 * c + 3]);
	// natural order (register renaming only)
	double tr[16], ti[16];
#pragma unroll
	for (int c = 0; c < 4; ++c)
#pragma unroll
		for (int d = 0; d < 4; ++d) { tr[c + 4 * d] = xr[4 * c + d]; ti[c + 4 * d] = xi[4 * c + d]; }
#pragma unroll
	for (int q = 0; q < 16; ++q) { xr[q] = tr[q]; xi[q] = ti[q]; }
}

// ---- the exchanges ---------------------------------------------------------------------------------------------------
// 16 values per lane out to LDS at wbase + q * WS, back from rbase + r * RS (element indices carry one pad per 16)
template <int WS, int RS>
__device__ __forceinline__ void wf_xchg(double (&v)[16], double *lds, int wbase, int rbase) {
#pragma unroll
	for (int q = 0; q < 16; ++q) lds[wbase + q * WS] = v[q];
	wf_fence();
#pragma unroll
	for (int r = 0; r < 16; ++r) v[r] = lds[rbase + r * RS];
	wf_fence();
}
// pass 3 side: four butterflies of four values each, slot 4 g + r at jbase[g] + 272 r
__device__ __forceinline__ void wf_xchg_in3(double (&v)[16], double *lds, int wbase, const int (&jb)[4]) {
#pragma unroll
	for (int q = 0; q < 16; ++q) lds[wbase + q * 17] = v[q];
	wf_fence();
#pragma unroll
	for (int g = 0; g < 4; ++g)
#pragma unroll
		for (int r = 0; r < 4; ++r) v[4 * g + r] = lds[jb[g] + 272 * r];
	wf_fence();
}
__device__ __forceinline__ void wf_xchg_out3(double (&v)[16], double *lds, const int (&jb)[4], int rbase) {
#pragma unroll
	for (int g = 0; g < 4; ++g)
#pragma unroll
		for (int r = 0; r < 4; ++r) lds[jb[g] + 272 * r] = v[4 * g + r];
	wf_fence();
#pragma unroll
	for (int q = 0; q < 16; ++q) v[q] = lds[rbase + q * 17];
	wf_fence();
}

// Tables of the wavefront transforms, behind the main twiddle table.  Every one is laid out so that the 64 lanes of a load
// read CONSECUTIVE entries: the main table indexed with a stride (W_1024^{r j} = tw[4 r j], ...) makes one load instruction
// touch up to 96 cache lines for 1 KB of twiddles, and the L1 of a CU (32 KB) then turns over with every transform.
constexpr int kTwT2 = kTwiddleN;          // 256 entries: W_256^{r k} at [16 r + k], r, k < 16 (second stage)
constexpr int kTwLog = kTwT2 + 256;       // 128 entries (1 / c_i, log c_i), c_i = 1/2 + (i + 1/2) / 256
constexpr int kTwExp = kTwLog + 128;      // 32 entries = 64 doubles 2^{j/64}
constexpr int kTwInvK = kTwExp + 32;      // 1040 doubles 1 / k (k = 0: 0)
constexpr int kTwP3 = kTwInvK + 520;      // 3 x 256 entries: W_1024^{r j} at [256 (r - 1) + j], j < 256 (radix-4 stage)
constexpr int kTwU = kTwP3 + 768;         // 1040 entries: W_2048^n, n <= 1024 (real-transform unpacking; the odd half's twist)
constexpr int kTwUo = kTwU + 1040;        // 1024 entries: W_4096^{2 j + 1} (unpacking of the odd half of a 4096-point transform)
// tables of the 512-point transform at eight points per lane (wf8_*, below)
constexpr int kTw8A = kTwUo + 1024;       // 64 entries: W_64^{n1 ka} at [8 n1 + ka] (second stage)
constexpr int kTw8B = kTw8A + 64;         // 512 entries: W_512^{(t & 7) ((t >> 3) + 8 kb)} at [64 kb + t] (third stage)
constexpr int kTw8U = kTw8B + 512;        // 528 entries: W_1024^n, n <= 512 (real-transform unpacking)
constexpr int kTwIS = kTw8U + 528;        // 256 entries = 512 doubles 1 / (4 sin(2 pi k / 2048)), k < 512 (k = 0: 0): wf_even2048
constexpr int kTwTotal = kTwIS + 256;     // double2 entries of the whole table

// Left alone, the scheduler puts every twiddle load right in front of its use (load, wait ~500 cycles, use, 15 times per
// stage: measured, the transforms ran five times slower than their arithmetic).  The loads of a stage are therefore issued
// together, ahead of the LDS exchange in front of the stage, and fenced off: nothing is scheduled across the fence, so they
// are all in flight while the exchange runs and the wait stands once, in front of the first use.
#define WF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void wf_t2_load(double2 (&w)[16], const double2 *__restrict__ tw, int lane) {
#pragma unroll
	for (int r = 1; r < 16; ++r) w[r] = tw_load(tw + kTwT2 + 16 * r, lane & 15);
	WF_SCHED_FENCE();
}
template <int S>
__device__ __forceinline__ void wf_t2_apply(double (&re)[16], double (&im)[16], const double2 (&w)[16]) {
#pragma unroll
	for (int r = 1; r < 16; ++r) wrot(re[r], im[r], w[r].x, S > 0 ? w[r].y : -w[r].y);
}

// The second stage's 256 twiddles in LDS (kWfT2Lds doubles, copied once per wavefront by wf_t2_to_lds): held in registers
// they are 60 VGPRs across the exchange in front of the stage -- what tips a kernel that keeps another 32 complex values
// alive across a transform (a noise spectrum, a first transform's result) over its 256 registers and into scratch memory.
// From LDS they are fetched five at a time right where they are used, the next five requested while these are applied.
constexpr int kWfT2Lds = 512;
__device__ __forceinline__ void wf_t2_to_lds(double *T2, const double2 *__restrict__ tw, int lane) {
	double2 v[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) v[i] = tw_load(tw + kTwT2, lane + 64 * i);
	WF_SCHED_FENCE();
#pragma unroll
	for (int i = 0; i < 4; ++i) reinterpret_cast<double2 *>(T2)[lane + 64 * i] = v[i];
	// (the caller's wf_fence() / the first exchange orders these writes before the first look-up)
}
template <int S>
__device__ __forceinline__ void wf_t2_apply_lds(double (&re)[16], double (&im)[16], const double *T2, int lane) {
	const double2 *t = reinterpret_cast<const double2 *>(T2) + (lane & 15);
	double2 w[2][5];
#pragma unroll
	for (int k = 0; k < 5; ++k) w[0][k] = t[16 * (1 + k)];
#pragma unroll
	for (int b = 0; b < 3; ++b) {
		if (b < 2) {
#pragma unroll
			for (int k = 0; k < 5; ++k) w[(b + 1) & 1][k] = t[16 * (1 + 5 * (b + 1) + k)];
		}
#pragma unroll
		for (int k = 0; k < 5; ++k) {
			const int r = 1 + 5 * b + k;
			wrot(re[r], im[r], w[b & 1][k].x, S > 0 ? w[b & 1][k].y : -w[b & 1][k].y);
		}
	}
}

struct WfIdx {
	int jb[4];       // padded element index of butterfly g's first input: j + (j >> 4)
	int x1w, x1r;    // exchange 1: write 17 t (+ q), read t + (t >> 4) (+ 68 r)
	int x2;          // exchange 2, 16-point side: 272 (t >> 4) + (t & 15) (+ 17 q)
	int jB;          // j_B
};
__device__ __forceinline__ WfIdx wf_idx(int lane) {
	WfIdx x;
	const int jA = lane, jB = lane ? 256 - lane : 128, jC = 64 + lane, jD = 192 - lane;
	x.jb[0] = jA + (jA >> 4); x.jb[1] = jB + (jB >> 4); x.jb[2] = jC + (jC >> 4); x.jb[3] = jD + (jD >> 4);
	x.x1w = 17 * lane;
	x.x1r = lane + (lane >> 4);
	x.x2 = 272 * (lane >> 4) + (lane & 15);
	x.jB = jB;
	return x;
}

// the twiddles W_1024^{r j}, r = 1..3, of the four butterflies of the radix-4 stage (sign S), from two table rows
template <int S>
__device__ __forceinline__ void wf_tw3(const double2 *__restrict__ tw, int lane, double (&wr)[4][3], double (&wi)[4][3]) {
	constexpr double s = S > 0 ? 1.0 : -1.0;
	double2 a[3], c[3];
#pragma unroll
	for (int r = 1; r <= 3; ++r) {
		a[r - 1] = tw_load(tw + kTwP3 + 256 * (r - 1), lane);       // j = t
		c[r - 1] = tw_load(tw + kTwP3 + 256 * (r - 1) + 64, lane);  // j = 64 + t
	}
	WF_SCHED_FENCE();
#pragma unroll
	for (int r = 0; r < 3; ++r) {
		wr[0][r] = a[r].x; wi[0][r] = s * a[r].y;
		wr[2][r] = c[r].x; wi[2][r] = s * c[r].y;
	}
	// j' = 256 - j:  W^{r j'} = (S i)^r conj(W^{r j})
#pragma unroll
	for (int g = 0; g < 4; g += 2) {
		wr[g + 1][0] = s * wi[g][0];  wi[g + 1][0] = s * wr[g][0];
		wr[g + 1][1] = -wr[g][1];     wi[g + 1][1] = wi[g][1];
		wr[g + 1][2] = -s * wi[g][2]; wi[g + 1][2] = -s * wr[g][2];
	}
	if (lane == 0) {  // j_B = 128: e^{S i pi r / 4}
		wr[1][0] = kH;  wi[1][0] = s * kH;
		wr[1][1] = 0.0; wi[1][1] = s;
		wr[1][2] = -kH; wi[1][2] = s * kH;
	}
}

// ---- the transforms ----------------------------------------------------------------------------------------------------
// strided -> paired; the caller has already run the leading 16-point stage (wdft16<S, NG>: pruned where the input's tail is
// zero padding) on the strided data
template <int S>
__device__ __forceinline__ void wf_fft1024_dit_rest(double (&re)[16], double (&im)[16], double *lds, const double2 *__restrict__ tw_,
													 int lane, const double *T2 = nullptr) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	const WfIdx ix = wf_idx(lane);
	if (T2) {
		wf_xchg<1, 68>(re, lds, ix.x1w, ix.x1r);
		wf_xchg<1, 68>(im, lds, ix.x1w, ix.x1r);
		wf_t2_apply_lds<S>(re, im, T2, lane);
	} else {
		double2 w2[16];
		wf_t2_load(w2, tw, lane);
		wf_xchg<1, 68>(re, lds, ix.x1w, ix.x1r);
		wf_xchg<1, 68>(im, lds, ix.x1w, ix.x1r);
		wf_t2_apply<S>(re, im, w2);
	}
	wdft16<S>(re, im);
	double wr[4][3], wi[4][3];
	wf_tw3<S>(tw, lane, wr, wi);
	wf_xchg_in3(re, lds, ix.x2, ix.jb);
	wf_xchg_in3(im, lds, ix.x2, ix.jb);
#pragma unroll
	for (int g = 0; g < 4; ++g) {
#pragma unroll
		for (int r = 1; r <= 3; ++r) wrot(re[4 * g + r], im[4 * g + r], wr[g][r - 1], wi[g][r - 1]);
		wdft4<S>(re[4 * g], im[4 * g], re[4 * g + 1], im[4 * g + 1], re[4 * g + 2], im[4 * g + 2], re[4 * g + 3], im[4 * g + 3]);
	}
}
// strided -> paired
template <int S>
__device__ __forceinline__ void wf_fft1024_dit(double (&re)[16], double (&im)[16], double *lds, const double2 *__restrict__ tw,
												int lane, const double *T2 = nullptr) {
	wdft16<S>(re, im);
	wf_fft1024_dit_rest<S>(re, im, lds, tw, lane, T2);
}
// paired -> strided
template <int S>
__device__ __forceinline__ void wf_fft1024_dif(double (&re)[16], double (&im)[16], double *lds, const double2 *__restrict__ tw_,
												int lane, const double *T2 = nullptr) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	const WfIdx ix = wf_idx(lane);
	{
		double wr[4][3], wi[4][3];
		wf_tw3<S>(tw, lane, wr, wi);
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			wdft4<S>(re[4 * g], im[4 * g], re[4 * g + 1], im[4 * g + 1], re[4 * g + 2], im[4 * g + 2], re[4 * g + 3], im[4 * g + 3]);
#pragma unroll
			for (int r = 1; r <= 3; ++r) wrot(re[4 * g + r], im[4 * g + r], wr[g][r - 1], wi[g][r - 1]);
		}
	}
	if (T2) {
		wf_xchg_out3(re, lds, ix.jb, ix.x2);
		wf_xchg_out3(im, lds, ix.jb, ix.x2);
		wdft16<S>(re, im);
		wf_t2_apply_lds<S>(re, im, T2, lane);
	} else {
		double2 w2[16];
		wf_t2_load(w2, tw, lane);
		wf_xchg_out3(re, lds, ix.jb, ix.x2);
		wf_xchg_out3(im, lds, ix.jb, ix.x2);
		wdft16<S>(re, im);
		wf_t2_apply<S>(re, im, w2);
	}
	wf_xchg<68, 1>(re, lds, ix.x1r, ix.x1w);
	wf_xchg<68, 1>(im, lds, ix.x1r, ix.x1w);
	wdft16<S>(re, im);
}

// ---- real transforms of 2048 points on top of them -------------------------------------------------------------------------
// Unpacking of one pair of bins (k, 1024 - k) of the real transform from the half-size complex one, reference convention
// (r2c: X[k] = sum x[n] e^{+2 pi i k n / N}); w = e^{+2 pi i k / 2048}.  Yields TWICE the spectrum (the halving of the
// even / odd split is left to the caller, who folds it into a scale factor he applies anyway).
__device__ __forceinline__ void wf_r2c_pair(double &kr, double &ki, double &mr, double &mi, double wr, double wi) {
	const double sx = kr + mr, sy = ki - mi, dx = kr - mr, dy = ki + mi;
	const double ox = fma(wr, dy, wi * dx), oy = fma(wi, dy, -(wr * dx));
	kr = sx + ox; ki = sy + oy;
	mr = sx - ox; mi = oy - sy;
}
// the twiddles e^{+2 pi i k / 2048} of the bins of slots A_q and C_q
__device__ __forceinline__ void wf_tw_real(const double2 *__restrict__ tw, int lane, double (&wr)[8], double (&wi)[8]) {
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const double2 a = tw_load(tw + kTwU + 256 * q, lane), c = tw_load(tw + kTwU + 256 * q + 64, lane);
		wr[q] = a.x; wi[q] = a.y;
		wr[4 + q] = c.x; wi[4 + q] = c.y;
	}
	WF_SCHED_FENCE();
}
// In: the paired output of wf_fft1024_dit<+1> on the packed signal z[m] = x[2 m] + i x[2 m + 1].  Out: slot (g, q) holds
// 2 X[j_g + 256 q]; lane 0's A_0 holds (2 X[0], 0) and nyq = 2 X[1024] (valid on lane 0).
__device__ __forceinline__ void wf_r2c_unpack(double (&re)[16], double (&im)[16], double &nyq, const double2 *__restrict__ tw_,
											   int lane) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	double wr[8], wi[8];
	wf_tw_real(tw, lane, wr, wi);
	nyq = 0.0;
	if (lane == 0) {
		const double a = re[0], b = im[0];
		re[0] = 2.0 * (a + b); im[0] = 0.0;
		nyq = 2.0 * (a - b);
		re[2] = 2.0 * re[2]; im[2] = 2.0 * im[2];                     // bin 512: X = Z
		wf_r2c_pair(re[1], im[1], re[3], im[3], kH, kH);              // 256 | 768
		wf_r2c_pair(re[4], im[4], re[7], im[7], kC8, kS8);            // 128 | 896
		wf_r2c_pair(re[5], im[5], re[6], im[6], kS8, kC8);            // 384 | 640
	} else {
#pragma unroll
		for (int q = 0; q < 4; ++q) wf_r2c_pair(re[q], im[q], re[7 - q], im[7 - q], wr[q], wi[q]);
	}
#pragma unroll
	for (int q = 0; q < 4; ++q) wf_r2c_pair(re[8 + q], im[8 + q], re[15 - q], im[15 - q], wr[4 + q], wi[4 + q]);
}
// Inverse: in: the Hermitian spectrum Y in the paired layout (lane 0: A_0 = (Y[0], .), nyq = Y[1024], imaginary parts of
// both ignored).  Out: Z such that wf_fft1024_dif<-1> yields the real signal y[n] = sum_k Yh[k] e^{-2 pi i k n / N}
// (reference c2r, unnormalised) as strided packed pairs (y[2 m], y[2 m + 1]).
__device__ __forceinline__ void wf_c2r_pair(double &kr, double &ki, double &mr, double &mi, double wr, double wi) {
	const double ex = kr + mr, ey = ki - mi, dx = kr - mr, dy = ki + mi;
	const double ox = fma(dx, wr, dy * wi), oy = fma(dy, wr, -(dx * wi));  // (dx + i dy) conj(w)
	kr = ex - oy; ki = ey + ox;
	mr = ex + oy; mi = ox - ey;
}
__device__ __forceinline__ void wf_c2r_pack(double (&re)[16], double (&im)[16], double nyq, const double2 *__restrict__ tw_,
											 int lane) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	double wr[8], wi[8];
	wf_tw_real(tw, lane, wr, wi);
	if (lane == 0) {
		const double y0 = re[0];
		re[0] = y0 + nyq; im[0] = y0 - nyq;
		re[2] = 2.0 * re[2]; im[2] = 2.0 * im[2];
		wf_c2r_pair(re[1], im[1], re[3], im[3], kH, kH);
		wf_c2r_pair(re[4], im[4], re[7], im[7], kC8, kS8);
		wf_c2r_pair(re[5], im[5], re[6], im[6], kS8, kC8);
	} else {
#pragma unroll
		for (int q = 0; q < 4; ++q) wf_c2r_pair(re[q], im[q], re[7 - q], im[7 - q], wr[q], wi[q]);
	}
#pragma unroll
	for (int q = 0; q < 4; ++q) wf_c2r_pair(re[8 + q], im[8 + q], re[15 - q], im[15 - q], wr[4 + q], wi[4 + q]);
}
// The same for a transform whose result is known to be real (the input is real and even): real parts only, 7 instead
// of 12 instructions per pair.  The imaginary parts are left unspecified.
__device__ __forceinline__ void wf_r2c_pair_re(double &kr, double ki, double &mr, double mi, double wr, double wi) {
	const double sx = kr + mr, dx = kr - mr, dy = ki + mi;
	const double ox = fma(wr, dy, wi * dx);
	kr = sx + ox;
	mr = sx - ox;
}
__device__ __forceinline__ void wf_r2c_unpack_re(double (&re)[16], const double (&im)[16], double &nyq,
												  const double2 *__restrict__ tw_, int lane) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	double wr[8], wi[8];
	wf_tw_real(tw, lane, wr, wi);
	nyq = 0.0;
	if (lane == 0) {
		const double a = re[0], b = im[0];
		re[0] = 2.0 * (a + b);
		nyq = 2.0 * (a - b);
		re[2] = 2.0 * re[2];
		wf_r2c_pair_re(re[1], im[1], re[3], im[3], kH, kH);
		wf_r2c_pair_re(re[4], im[4], re[7], im[7], kC8, kS8);
		wf_r2c_pair_re(re[5], im[5], re[6], im[6], kS8, kC8);
	} else {
#pragma unroll
		for (int q = 0; q < 4; ++q) wf_r2c_pair_re(re[q], im[q], re[7 - q], im[7 - q], wr[q], wi[q]);
	}
#pragma unroll
	for (int q = 0; q < 4; ++q) wf_r2c_pair_re(re[8 + q], im[8 + q], re[15 - q], im[15 - q], wr[4 + q], wi[4 + q]);
}
// wf_c2r_pack for a REAL spectrum (re[] in, nyq = Y[1024]; im[] is written)
__device__ __forceinline__ void wf_c2r_pair_re(double &kr, double &ki, double &mr, double &mi, double wr, double wi) {
	const double ex = kr + mr, dx = kr - mr;
	const double o = dx * wr;
	kr = fma(dx, wi, ex); ki = o;
	mr = fma(-dx, wi, ex); mi = o;
}
__device__ __forceinline__ void wf_c2r_pack_re(double (&re)[16], double (&im)[16], double nyq, const double2 *__restrict__ tw_,
												int lane) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	double wr[8], wi[8];
	wf_tw_real(tw, lane, wr, wi);
	if (lane == 0) {
		const double y0 = re[0];
		re[0] = y0 + nyq; im[0] = y0 - nyq;
		re[2] = 2.0 * re[2]; im[2] = 0.0;
		wf_c2r_pair_re(re[1], im[1], re[3], im[3], kH, kH);
		wf_c2r_pair_re(re[4], im[4], re[7], im[7], kC8, kS8);
		wf_c2r_pair_re(re[5], im[5], re[6], im[6], kS8, kC8);
	} else {
#pragma unroll
		for (int q = 0; q < 4; ++q) wf_c2r_pair_re(re[q], im[q], re[7 - q], im[7 - q], wr[q], wi[q]);
	}
#pragma unroll
	for (int q = 0; q < 4; ++q) wf_c2r_pair_re(re[8 + q], im[8 + q], re[15 - q], im[15 - q], wr[4 + q], wi[4 + q]);
}
// sum over the wavefront, in every lane (and in scalar registers)
__device__ __forceinline__ double wave_sum_all(double v) { return uniform_d(wave_sum(v)); }

// bin held by slot 4 g + q of the paired layout
__device__ __forceinline__ int wf_bin(int lane, int g, int q) {
	const int j = g == 0 ? lane : g == 1 ? (lane ? 256 - lane : 128) : g == 2 ? 64 + lane : 192 - lane;
	return j + 256 * q;
}

// ---- either half of a 2048-point transform by the same wavefront ("even / odd split") --------------------------------------
// A 4096-point real transform (D4C) packs into 2048 complex points z[m]; one decimation-in-frequency step splits their
// transform Z into two independent 1024-point ones:  Z[2 j] = FFT(z[n] + z[n + 1024]),  Z[2 j + 1] = FFT((z[n] - z[n + 1024]) W_2048^n).
// The real-transform unpacking pairs Z[k] with Z[2048 - k] -- both even or both odd -- so each half is unpacked on its own:
//   even  pairs (E[j], E[1024 - j]), twiddle W_4096^{2 j}: exactly the 2048-point real unpacking (wf_r2c_unpack)
//   odd   pairs (O[j], O[1023 - j]), twiddle W_4096^{2 j + 1}: the paired layout with j_B = 255 - t, j_D = 191 - t
//         (no self-paired element, no special lane)
// One wavefront runs the two halves one after the other with the same code: the parity is a run-time value that only moves
// a few indices.  Slot 4 g + q of parity p then holds bin 2 (j_g + 256 q) + p of the 4096-point real transform.
__device__ __forceinline__ int wf_j(int lane, int odd, int g) {
	return g == 0 ? lane : g == 1 ? (odd ? 255 - lane : (lane ? 256 - lane : 128)) : g == 2 ? 64 + lane : (odd ? 191 - lane : 192 - lane);
}
__device__ __forceinline__ WfIdx wf_idx_p(int lane, int odd) {
	WfIdx x;
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		const int j = wf_j(lane, odd, g);
		x.jb[g] = j + (j >> 4);
	}
	x.x1w = 17 * lane;
	x.x1r = lane + (lane >> 4);
	x.x2 = 272 * (lane >> 4) + (lane & 15);
	x.jB = wf_j(lane, odd, 1);
	return x;
}
// strided -> paired layout of parity `odd`; the caller has run the leading wdft16<S, NG>
template <int S>
__device__ __forceinline__ void wf_fft1024_dit_rest_p(double (&re)[16], double (&im)[16], double *lds, const double2 *__restrict__ tw_,
													   int lane, int odd, const double *T2 = nullptr) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	const WfIdx ix = wf_idx_p(lane, odd);
	if (T2) {
		wf_xchg<1, 68>(re, lds, ix.x1w, ix.x1r);
		wf_xchg<1, 68>(im, lds, ix.x1w, ix.x1r);
		wf_t2_apply_lds<S>(re, im, T2, lane);
	} else {
		double2 w2[16];
		wf_t2_load(w2, tw, lane);
		wf_xchg<1, 68>(re, lds, ix.x1w, ix.x1r);
		wf_xchg<1, 68>(im, lds, ix.x1w, ix.x1r);
		wf_t2_apply<S>(re, im, w2);
	}
	wdft16<S>(re, im);
	double2 w3[4][3];
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		const int j = wf_j(lane, odd, g);
#pragma unroll
		for (int r = 1; r <= 3; ++r) w3[g][r - 1] = tw_load(tw + kTwP3 + 256 * (r - 1), j);
	}
	WF_SCHED_FENCE();
	wf_xchg_in3(re, lds, ix.x2, ix.jb);
	wf_xchg_in3(im, lds, ix.x2, ix.jb);
#pragma unroll
	for (int g = 0; g < 4; ++g) {
#pragma unroll
		for (int r = 1; r <= 3; ++r) wrot(re[4 * g + r], im[4 * g + r], w3[g][r - 1].x, S > 0 ? w3[g][r - 1].y : -w3[g][r - 1].y);
		wdft4<S>(re[4 * g], im[4 * g], re[4 * g + 1], im[4 * g + 1], re[4 * g + 2], im[4 * g + 2], re[4 * g + 3], im[4 * g + 3]);
	}
}
// the odd half's input: element n = lane + 64 q times W_2048^n
__device__ __forceinline__ void wf_odd_twist(double (&re)[16], double (&im)[16], const double2 *__restrict__ tw_, int lane, int nslots) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	double2 w[16];
#pragma unroll
	for (int q = 0; q < 16; ++q)
		if (q < nslots) w[q] = tw_load(tw + kTwU + 64 * q, lane);
	WF_SCHED_FENCE();
#pragma unroll
	for (int q = 0; q < 16; ++q)
		if (q < nslots) wrot(re[q], im[q], w[q].x, w[q].y);
}
// unpacking of the odd half: slot (g, q) <- 2 X[2 (j_g + 256 q) + 1]
__device__ __forceinline__ void wf_r2c_unpack_odd(double (&re)[16], double (&im)[16], const double2 *__restrict__ tw_, int lane) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	double2 a[4], c[4];
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		a[q] = tw_load(tw + kTwUo + 256 * q, lane);
		c[q] = tw_load(tw + kTwUo + 256 * q + 64, lane);
	}
	WF_SCHED_FENCE();
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		wf_r2c_pair(re[q], im[q], re[7 - q], im[7 - q], a[q].x, a[q].y);
		wf_r2c_pair(re[8 + q], im[8 + q], re[15 - q], im[15 - q], c[q].x, c[q].y);
	}
}
// the 4096-point real transform's half of parity `odd`, from the strided packed input of that half (the even half's input is
// z[n] + z[n + 1024], the odd half's (z[n] - z[n + 1024]) before the twist, which is applied here).  ng: the input's slots
// 4 ng .. 15 are zero.  Out: slot (g, q) = 2 X[2 (j_g + 256 q) + odd]; even half: lane 0's A_0 = (2 X[0], 0), nyq = 2 X[2048].
__device__ __forceinline__ void wf_r2c4096_half(double (&re)[16], double (&im)[16], double &nyq, int ng, double *lds,
												const double2 *__restrict__ tw, int lane, int odd, const double *T2 = nullptr) {
	WC_FRESH(lane);  // (lane-derived addresses must not be hoisted out of the caller's loops and spilled)
	if (odd) {
		if (ng <= 1) wf_odd_twist(re, im, tw, lane, 4);
		else if (ng == 2) wf_odd_twist(re, im, tw, lane, 8);
		else wf_odd_twist(re, im, tw, lane, 16);
	}
	if (ng <= 1) wdft16<+1, 1>(re, im);
	else if (ng == 2) wdft16<+1, 2>(re, im);
	else wdft16<+1, 4>(re, im);
	wf_fft1024_dit_rest_p<+1>(re, im, lds, tw, lane, odd, T2);
	nyq = 0.0;
	if (odd) wf_r2c_unpack_odd(re, im, tw, lane);
	else wf_r2c_unpack(re, im, nyq, tw, lane);
}

// ======== 2048-point complex transform by TWO wavefronts (one 128-thread workgroup), 16 points per lane ================
// The same construction one size up (16 x 16 x 8), for the 4096-point real transforms of D4C: the lanes of both
// wavefronts meet in one 17 KB exchange buffer, so every exchange is bracketed by workgroup barriers (two wavefronts:
// cheap).  t = thread 0..127.
//   strided   slot q (0..15) holds element t + 128 q
//   paired    slot 8 g + q (g = 0: A, 1: B; q = 0..7) holds element j_g + 256 q, j_A = t, j_B = 256 - t (thread 0: 128):
//             element k and element 2048 - k share a lane (A_q with B_{7-q}; thread 0 pairs A_q with A_{8-q}, B_q with
//             B_{7-q} and keeps the self-paired 0 and 1024 in A_0, A_4)
constexpr int kWf2Lds = 2304;  // doubles: 2048 + one pad per 16, and room for D4C's 2049 + 2 * 127 smoothing terms

// 8-point DFT, sign S, natural order in and out
template <int S>
__device__ __forceinline__ void wdft8(double (&xr)[8], double (&xi)[8]) {
	// r = 2 a + b (a = 0..3, b = 0..1), q = c + 4 d:  y[c + 4 d] = sum_b (-1)^{b d} w8^{b c} sum_a (S i)^{a c} x[2 a + b]
	wdft4<S>(xr[0], xi[0], xr[2], xi[2], xr[4], xi[4], xr[6], xi[6]);  // slot 2 c     <- u_0[c]
	wdft4<S>(xr[1], xi[1], xr[3], xi[3], xr[5], xi[5], xr[7], xi[7]);  // slot 2 c + 1 <- u_1[c]
	// u_1[c] *= w8^c
	{ const double x = xr[3], y = xi[3]; xr[3] = kH * (S > 0 ? x - y : x + y); xi[3] = kH * (S > 0 ? x + y : y - x); }
	{ const double x = xr[5], y = xi[5]; xr[5] = S > 0 ? -y : y; xi[5] = S > 0 ? x : -x; }
	{ const double x = xr[7], y = xi[7]; xr[7] = S > 0 ? -kH * (x + y) : kH * (y - x); xi[7] = S > 0 ? kH * (x - y) : -kH * (x + y); }
	double tr[8], ti[8];
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		tr[c] = xr[2 * c] + xr[2 * c + 1]; ti[c] = xi[2 * c] + xi[2 * c + 1];
		tr[c + 4] = xr[2 * c] - xr[2 * c + 1]; ti[c + 4] = xi[2 * c] - xi[2 * c + 1];
	}
#pragma unroll
	for (int q = 0; q < 8; ++q) { xr[q] = tr[q]; xi[q] = ti[q]; }
}

struct Wf2Idx {
	int jb[2];     // padded element index of butterfly g's first input
	int x1w, x1r;  // exchange 1: write 17 t (+ q), read t + (t >> 4) (+ 136 r)
	int x2;        // exchange 2, 16-point side: 272 (t >> 4) + (t & 15) (+ 17 q)
	int jB;
};
__device__ __forceinline__ Wf2Idx wf2_idx(int t) {
	Wf2Idx x;
	const int jB = t ? 256 - t : 128;
	x.jb[0] = t + (t >> 4);
	x.jb[1] = jB + (jB >> 4);
	x.x1w = 17 * t;
	x.x1r = t + (t >> 4);
	x.x2 = 272 * (t >> 4) + (t & 15);
	x.jB = jB;
	return x;
}
#ifndef WC_WF2_NOBAR
#define WC_WF2_NOBAR 0  // 1: timing ablation (wrong results): the exchanges of the two-wavefront transform without barriers
#endif
#if WC_WF2_NOBAR
#define WF2_SYNC() wf_fence()
#else
#define WF2_SYNC() __syncthreads()
#endif
template <int WS, int RS>
__device__ __forceinline__ void wf2_xchg(double (&v)[16], double *lds, int wbase, int rbase) {
#pragma unroll
	for (int q = 0; q < 16; ++q) lds[wbase + q * WS] = v[q];
	WF2_SYNC();
#pragma unroll
	for (int r = 0; r < 16; ++r) v[r] = lds[rbase + r * RS];
	WF2_SYNC();
}
__device__ __forceinline__ void wf2_xchg_in3(double (&v)[16], double *lds, int wbase, const int (&jb)[2]) {
#pragma unroll
	for (int q = 0; q < 16; ++q) lds[wbase + q * 17] = v[q];
	WF2_SYNC();
#pragma unroll
	for (int g = 0; g < 2; ++g)
#pragma unroll
		for (int r = 0; r < 8; ++r) v[8 * g + r] = lds[jb[g] + 272 * r];
	WF2_SYNC();
}
__device__ __forceinline__ void wf2_xchg_out3(double (&v)[16], double *lds, const int (&jb)[2], int rbase) {
#pragma unroll
	for (int g = 0; g < 2; ++g)
#pragma unroll
		for (int r = 0; r < 8; ++r) lds[jb[g] + 272 * r] = v[8 * g + r];
	WF2_SYNC();
#pragma unroll
	for (int q = 0; q < 16; ++q) v[q] = lds[rbase + q * 17];
	WF2_SYNC();
}
// radix-8 stage of one butterfly group: twiddles W_2048^{r j} (sign S) before (DIT) or after (DIF) the 8-point DFT
template <int S, bool DIT>
__device__ __forceinline__ void wf2_stage8(double (&re)[16], double (&im)[16], const double2 *__restrict__ tw, int g, int j) {
	double xr[8], xi[8];
#pragma unroll
	for (int r = 0; r < 8; ++r) { xr[r] = re[8 * g + r]; xi[r] = im[8 * g + r]; }
	if (!DIT) wdft8<S>(xr, xi);
#pragma unroll
	for (int r = 1; r < 8; ++r) {
		const double2 w = tw_load(tw, 2 * r * j);
		wrot(xr[r], xi[r], w.x, S > 0 ? w.y : -w.y);
	}
	if (DIT) wdft8<S>(xr, xi);
#pragma unroll
	for (int r = 0; r < 8; ++r) { re[8 * g + r] = xr[r]; im[8 * g + r] = xi[r]; }
}
// strided -> paired; the caller has run the leading wdft16<S, NG> on the strided data
template <int S>
__device__ __forceinline__ void wf2_fft2048_dit_rest(double (&re)[16], double (&im)[16], double *lds, const double2 *__restrict__ tw_,
													  int t) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	const Wf2Idx ix = wf2_idx(t);
	wf2_xchg<1, 136>(re, lds, ix.x1w, ix.x1r);
	wf2_xchg<1, 136>(im, lds, ix.x1w, ix.x1r);
	{
		const double2 *__restrict__ t2 = tw + kTwT2;
#pragma unroll
		for (int r = 1; r < 16; ++r) {
			const double2 w = tw_load(t2 + 16 * r, t & 15);
			wrot(re[r], im[r], w.x, S > 0 ? w.y : -w.y);
		}
	}
	wdft16<S>(re, im);
	wf2_xchg_in3(re, lds, ix.x2, ix.jb);
	wf2_xchg_in3(im, lds, ix.x2, ix.jb);
	wf2_stage8<S, true>(re, im, tw, 0, t);
	wf2_stage8<S, true>(re, im, tw, 1, ix.jB);
}
template <int S>
__device__ __forceinline__ void wf2_fft2048_dit(double (&re)[16], double (&im)[16], double *lds, const double2 *__restrict__ tw,
												 int t) {
	wdft16<S>(re, im);
	wf2_fft2048_dit_rest<S>(re, im, lds, tw, t);
}
// paired -> strided
template <int S>
__device__ __forceinline__ void wf2_fft2048_dif(double (&re)[16], double (&im)[16], double *lds, const double2 *__restrict__ tw_,
												 int t) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	const Wf2Idx ix = wf2_idx(t);
	wf2_stage8<S, false>(re, im, tw, 0, t);
	wf2_stage8<S, false>(re, im, tw, 1, ix.jB);
	wf2_xchg_out3(re, lds, ix.jb, ix.x2);
	wf2_xchg_out3(im, lds, ix.jb, ix.x2);
	wdft16<S>(re, im);
	{
		const double2 *__restrict__ t2 = tw + kTwT2;
#pragma unroll
		for (int r = 1; r < 16; ++r) {
			const double2 w = tw_load(t2 + 16 * r, t & 15);
			wrot(re[r], im[r], w.x, S > 0 ? w.y : -w.y);
		}
	}
	wf2_xchg<136, 1>(re, lds, ix.x1r, ix.x1w);
	wf2_xchg<136, 1>(im, lds, ix.x1r, ix.x1w);
	wdft16<S>(re, im);
}
// bin held by slot 8 g + q
__device__ __forceinline__ int wf2_bin(int t, int g, int q) { return (g == 0 ? t : (t ? 256 - t : 128)) + 256 * q; }
// 4096-point real transform, unpacking (see wf_r2c_unpack): slot (g, q) <- 2 X[j_g + 256 q]; thread 0: A_0 = (2 X[0], 0),
// nyq = 2 X[2048]
__device__ __forceinline__ void wf2_r2c_unpack(double (&re)[16], double (&im)[16], double &nyq, const double2 *__restrict__ tw_,
												int t) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	nyq = 0.0;
	if (t == 0) {
		const double a = re[0], b = im[0];
		re[0] = 2.0 * (a + b); im[0] = 0.0;
		nyq = 2.0 * (a - b);
		re[4] = 2.0 * re[4]; im[4] = 2.0 * im[4];  // bin 1024: X = Z
#pragma unroll
		for (int q = 1; q < 4; ++q) {  // 256 q | 2048 - 256 q
			const double2 w = tw_load(tw, 256 * q);
			wf_r2c_pair(re[q], im[q], re[8 - q], im[8 - q], w.x, w.y);
		}
#pragma unroll
		for (int q = 0; q < 4; ++q) {  // 128 + 256 q | 1920 - 256 q
			const double2 w = tw_load(tw, 128 + 256 * q);
			wf_r2c_pair(re[8 + q], im[8 + q], re[15 - q], im[15 - q], w.x, w.y);
		}
	} else {
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			const double2 w = tw_load(tw + 256 * q, t);
			wf_r2c_pair(re[q], im[q], re[15 - q], im[15 - q], w.x, w.y);
		}
	}
}

// interp1Q (reference src/world_matlabfunctions.cpp:220-241) on a table in LDS, in interp1q_rcp's arithmetic (wc_device.hpp)
// without its branch: the difference beyond the last entry is zero because the index is clamped, not because it is tested.
__device__ __forceinline__ double wf_interp1q(double x0, double dx, double rdx, const double *S, int n, double xi) {
	const double t = xi - x0;
	double q = t * rdx;
	q = fma(fma(-dx, q, t), rdx, q);
	const int b = (int)q;
	const double frac = q - b;
	const double y0 = S[b], y1 = S[min(b + 1, n - 1)];
	return fma(y1 - y0, frac, y0);
}

// ---- the reference's sequential cumulative sum, bit for bit, by one wavefront ---------------------------------------------
// seq_cumsum_nonneg (wc_device.hpp) for a single wavefront with every lane's chunk (<= CHMAX consecutive terms) held in
// registers: the block version walks its chunks through LDS one dependent read at a time, which a lone wavefront cannot hide
// (measured: 30 k cycles per 2291-term sum).  Same construction, same results: clean lanes (no binade crossing, no tie) get
// their exact increment by adding their terms to 2^e and a segmented scan; the other ("dirty") lanes are visited in order,
// each adding its own terms to the running sum in the reference's order while the rest of the wavefront waits; every clean
// lane then re-adds its terms from its exact start value.  In: S[0 .. len) terms (non-negative), out: their running sums.
template <int CHMAX>
__device__ __forceinline__ void seq_cumsum_nonneg_wave(double *S, int len, int lane) {
	const int ch = (len + 63) / 64;  // <= CHMAX
	const int lo = min(lane * ch, len), hi = min(len, lo + ch), n = hi - lo;
	double v[CHMAX];
#pragma unroll
	for (int k = 0; k < CHMAX; ++k) v[k] = S[min(lo + k, len - 1)];
	wf_fence();
	double loc = 0.0;
	bool bad = false;
#pragma unroll
	for (int k = 0; k < CHMAX; ++k) {
		v[k] = (k < n) ? v[k] : 0.0;
		loc += v[k];
		bad = bad || !(v[k] >= 0.0);
	}
	const double base_a = wave_incl_scan(loc, lane) - loc;
	const double lower = base_a * (1.0 - 0x1p-30), upper = (base_a + loc) * (1.0 + 0x1p-30);
	const int E = (__double2hiint(lower) >> 20) & 0x7ff;
	bool dirty = lane == 0 || bad || !(lower > 0.0) || E != ((__double2hiint(upper) >> 20) & 0x7ff) || E < 64 || E > 1984;
	double d = 0.0;
	if (n <= 0) {
		dirty = lane == 0;
	} else if (!dirty) {
		const double C = __hiloint2double(E << 20, 0);               // 2^e, the start of the binade
		const double rulp = __hiloint2double((2098 - E) << 20, 0);   // 2^(52 - e) = 1 / ulp
		double r = C;
#pragma unroll
		for (int k = 0; k < CHMAX; ++k) {
			const double t = v[k] * rulp;
			dirty = dirty || (t - floor(t)) == 0.5;
			r = v[k] + r;
		}
		d = dirty ? 0.0 : r - C;
	}
	// inclusive segmented scan of d over the lanes, restarting behind every dirty lane
	double x = d;
	int f = dirty ? 1 : 0;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const double xo = __shfl_up(x, o, 64);
		const int fo = __shfl_up(f, o, 64);
		if (lane >= o) {
			if (!f) x += xo;
			f |= fo;
		}
	}
	unsigned long long mk = __ballot(dirty);
	const unsigned long long m = mk;
	// the walk over the dirty lanes, in order; endv: the running sum behind a dirty lane's terms (kept in that lane)
	double so = 0.0, endv = 0.0;
	int prev = -1;
	while (mk) {
		const int t = __ffsll((long long)mk) - 1;
		mk &= mk - 1;
		const double xp = __shfl(x, max(t - 1, 0), 64);
		const double si = prev < 0 ? 0.0 : (t == prev + 1 ? so : so + xp);
		if (lane == t) {
			double run = si;
#pragma unroll
			for (int k = 0; k < CHMAX; ++k) {
				run = v[k] + run;  // (terms beyond the chunk are zero: the sum does not move)
				v[k] = run;
			}
			endv = run;
		}
		so = __shfl(endv, t, 64);
		prev = t;
	}
	{
		// (the cross-lane read stands outside the branch: a lane switched off there would hand over nothing)
		const unsigned long long below = m & ((1ull << lane) - 1ull);  // lane 0 is always dirty, so never empty for a clean lane
		const int j = below ? 63 - __clzll((long long)below) : 0;
		const double start = __shfl(endv, j, 64) + (x - d);
		if (!dirty) {
			double run = start;
#pragma unroll
			for (int k = 0; k < CHMAX; ++k) {
				run = v[k] + run;
				v[k] = run;
			}
		}
	}
#pragma unroll
	for (int k = 0; k < CHMAX; ++k)
		if (k < n) S[lo + k] = v[k];
	wf_fence();
}

// The same for terms of either sign (D4C's two smoothings of the group-delay numerator, reference src/d4c.cpp:440-460 through
// src/world_common.cpp:82-116).  While the running sum c stays inside one binade and keeps its sign, fl(c + v) = c + u rn(v / u)
// (u = the binade's ulp) for v of either sign, so a lane whose chunk provably keeps c inside one binade and meets no tie is
// "clean" exactly as above; its increment is formed on a stand-in C for c (the estimate snapped to 2^-20 of the binade, so
// that C + partial sums stay inside the binade with c).  "Provably": from a tree-ordered estimate of c whose error is bounded
// against the sum of the ABSOLUTE values so far (cancellation makes c small against its own rounding history), with that
// bound and the snapping as margins.  Everything else -- zero crossings, binade changes, ties -- is walked in order.
template <int CHMAX>
__device__ __forceinline__ void seq_cumsum_signed_wave(double *S, int len, int lane) {
	const int ch = (len + 63) / 64;
	const int lo = min(lane * ch, len), hi = min(len, lo + ch), n = hi - lo;
	double v[CHMAX];
#pragma unroll
	for (int k = 0; k < CHMAX; ++k) v[k] = S[min(lo + k, len - 1)];
	wf_fence();
	double loc = 0.0, aloc = 0.0;
#pragma unroll
	for (int k = 0; k < CHMAX; ++k) {
		v[k] = (k < n) ? v[k] : 0.0;
		loc += v[k];
		aloc += fabs(v[k]);
	}
	const double e0 = wave_incl_scan(loc, lane) - loc;           // estimate of c in front of this lane's terms
	const double a1 = wave_incl_scan(aloc, lane);                // sum of |v| up to and including them
	double emin = e0, emax = e0;
	{
		double e = e0;
#pragma unroll
		for (int k = 0; k < CHMAX; ++k) {
			e += v[k];
			emin = fmin(emin, e);
			emax = fmax(emax, e);
		}
	}
	// margins: the estimate against the sequential sum (both within ~2304 half-ulps of sum |v| of the exact sum), the snapping
	const double m = a1 * 0x1p-40 + fmax(fabs(emin), fabs(emax)) * 0x1p-19;
	const double lower = emin - m, upper = emax + m;
	const int E = (__double2hiint(lower) >> 20) & 0x7ff;
	bool dirty = lane == 0 || !(a1 < __builtin_huge_val()) || !(lower * upper > 0.0) || E != ((__double2hiint(upper) >> 20) & 0x7ff) ||
				 E < 64 || E > 1984;
	double d = 0.0;
	if (n <= 0) {
		dirty = lane == 0;
	} else if (!dirty) {
		const double rulp = __hiloint2double((2098 - E) << 20, 0);  // 2^(52 - e) = 1 / ulp
		const double grid = __hiloint2double((E - 20) << 20, 0);     // 2^(e - 20)
		const double C = rint(e0 / grid) * grid;                     // same binade as c (the margins above), a multiple of 2^32 ulps
		double r = C;
#pragma unroll
		for (int k = 0; k < CHMAX; ++k) {
			const double t = v[k] * rulp;
			dirty = dirty || (t - floor(t)) == 0.5;
			r = v[k] + r;
		}
		d = dirty ? 0.0 : r - C;
	}
	double x = d;
	int f = dirty ? 1 : 0;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const double xo = __shfl_up(x, o, 64);
		const int fo = __shfl_up(f, o, 64);
		if (lane >= o) {
			if (!f) x += xo;
			f |= fo;
		}
	}
	unsigned long long mk = __ballot(dirty);
	const unsigned long long msk = mk;
	double so = 0.0, endv = 0.0;
	int prev = -1;
	while (mk) {
		const int t = __ffsll((long long)mk) - 1;
		mk &= mk - 1;
		const double xp = __shfl(x, max(t - 1, 0), 64);
		const double si = prev < 0 ? -0.0 : (t == prev + 1 ? so : so + xp);  // (-0 + v = v for every v: the first sum is the first term)
		if (lane == t) {
			double run = si;
#pragma unroll
			for (int k = 0; k < CHMAX; ++k) {
				run = (k < n) ? v[k] + run : run;  // (a signed sum may be -0: adding the padding's +0 would flip it)
				v[k] = run;
			}
			endv = run;
		}
		so = __shfl(endv, t, 64);
		prev = t;
	}
	{
		const unsigned long long below = msk & ((1ull << lane) - 1ull);
		const int j = below ? 63 - __clzll((long long)below) : 0;
		const double start = __shfl(endv, j, 64) + (x - d);
		if (!dirty) {
			double run = start;
#pragma unroll
			for (int k = 0; k < CHMAX; ++k) {
				run = (k < n) ? v[k] + run : run;
				v[k] = run;
			}
		}
	}
#pragma unroll
	for (int k = 0; k < CHMAX; ++k)
		if (k < n) S[lo + k] = v[k];
	wf_fence();
}

// ---- lean log / exp ------------------------------------------------------------------------------------------------------
// log(x) for finite x > 0 to ~3e-16 absolute (relative for |log x| > 1): x = 2^e m, m in [1/2, 1); the top seven mantissa
// bits pick c_i with (1 / c_i, log c_i) tabulated; log m = log c_i + log1p(r), r = m / c_i - 1, |r| < 2^-8, degree-6 series.
// 14 vector instructions and one 16-byte load against ocml's 98 (double-double arithmetic for the last bit, which nothing
// here needs: the values are envelope logarithms compared at 1e-7).  Anything else (0, negative, inf, NaN) goes to libm.
// wf_log_fast: the arithmetic alone (garbage outside finite x > 0); wf_log_ok: whether x is inside.
__device__ __forceinline__ double wf_log_fast(double x, const double2 *__restrict__ tw) {
	const int e = __builtin_amdgcn_frexp_exp(x);
	const double m = __builtin_amdgcn_frexp_mant(x);
	const int i = (__double2hiint(m) >> 13) & 127;
	const double2 c = tw_load(tw + kTwLog, i);
	const double r = fma(m, c.x, -1.0);
	double q = fma(r, -1.0 / 6.0, 0.2);
	q = fma(r, q, -0.25);
	q = fma(r, q, 1.0 / 3.0);
	q = fma(r, q, -0.5);
	const double p = fma(r * r, q, r);
	return fma((double)e, 0.69314718055994530942, c.y) + p;
}
__device__ __forceinline__ bool wf_log_ok(double x) { return x > 0.0 && x < __builtin_huge_val(); }
__device__ __noinline__ double wf_log_libm(double x) { return log(x); }
__device__ __forceinline__ double wf_log(double x, const double2 *__restrict__ tw) {
	double res = wf_log_fast(x, tw);
	if (!wf_log_ok(x)) res = wf_log_libm(x);
	return res;
}
// exp(x) to ~2e-16 relative: x = (64 k + j) ln2 / 64 + r, |r| <= ln2 / 128; 2^{j/64} tabulated, degree-5 series.
__device__ __forceinline__ double tw_load_d(const double2 *tw, int idx) {
	typedef const double __attribute__((address_space(1))) *gptr;
	return ((gptr)tw)[idx];
}
__device__ __forceinline__ double wf_exp(double x, const double2 *__restrict__ tw) {
	const double n = rint(x * 0x1.71547652b82fep+6);  // 64 / ln 2
	double r = fma(n, -0x1.62e42feep-7, x);            // ln2 / 64: leading 32 bits (n times it is exact)
	r = fma(n, -0x1.a39ef35793c76p-39, r);             // the rest
	const int ni = (int)n;
	const double t = tw_load_d(tw + kTwExp, ni & 63);
	double q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
	q = fma(r, q, 1.0 / 6.0);
	q = fma(r, q, 0.5);
	const double p = fma(r * r, q, r);
	return ldexp(fma(t, p, t), ni >> 6);
}

// The tables of wf_log / wf_exp in LDS (kWfTabLds doubles: 128 x (1 / c, log c), then 64 x 2^{j/64}), copied once per
// wavefront by wf_tables_to_lds: a look-up is then an LDS read (~100 cycles) instead of a gather from L2 (~500), which a
// lone wavefront meets 34 times per frame or pulse.
constexpr int kWfTabLds = 320;
__device__ __forceinline__ void wf_tables_to_lds(double *T, const double2 *__restrict__ tw, int lane) {
	double2 v[3];
#pragma unroll
	for (int i = 0; i < 3; ++i) v[i] = tw_load(tw + kTwLog, min(lane + 64 * i, 159));  // 128 + 32 entries, contiguous
	WF_SCHED_FENCE();
#pragma unroll
	for (int i = 0; i < 3; ++i)
		if (lane + 64 * i < 160) reinterpret_cast<double2 *>(T)[lane + 64 * i] = v[i];
	wf_fence();
}
__device__ __forceinline__ double wf_log_fast_l(double x, const double *T) {
	const int e = __builtin_amdgcn_frexp_exp(x);
	const double m = __builtin_amdgcn_frexp_mant(x);
	const int i = (__double2hiint(m) >> 13) & 127;
	const double2 c = reinterpret_cast<const double2 *>(T)[i];
	const double r = fma(m, c.x, -1.0);
	double q = fma(r, -1.0 / 6.0, 0.2);
	q = fma(r, q, -0.25);
	q = fma(r, q, 1.0 / 3.0);
	q = fma(r, q, -0.5);
	const double p = fma(r * r, q, r);
	return fma((double)e, 0.69314718055994530942, c.y) + p;
}
__device__ __forceinline__ double wf_log_l(double x, const double *T) {
	double res = wf_log_fast_l(x, T);
	if (!wf_log_ok(x)) res = wf_log_libm(x);
	return res;
}
__device__ __forceinline__ double wf_exp_l(double x, const double *T) {
	const double n = rint(x * 0x1.71547652b82fep+6);  // 64 / ln 2
	double r = fma(n, -0x1.62e42feep-7, x);
	r = fma(n, -0x1.a39ef35793c76p-39, r);
	const int ni = (int)n;
	const double t = T[256 + (ni & 63)];
	double q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
	q = fma(r, q, 1.0 / 6.0);
	q = fma(r, q, 0.5);
	const double p = fma(r * r, q, r);
	return ldexp(fma(t, p, t), ni >> 6);
}

// sin and cos of x to ~1 ulp for |x| up to ~1e5 (the phases here are a few radians): x = n pi/2 + r, |r| <= pi/4, by
// two fused steps; the kernels are fdlibm's polynomials.  ~35 vector instructions against ~120 of ocml's sincos with
// its large-argument path, 17 times per minimum-phase spectrum.
__device__ __forceinline__ void wf_sincos(double x, double &sn, double &cs) {
	const double n = rint(x * 0.63661977236758134308);
	double r = fma(n, -1.57079632679489655800e+00, x);
	r = fma(n, -6.12323399573676603587e-17, r);
	const int q = (int)n;
	const double z = r * r;
	double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
	ps = fma(z, ps, 2.75573137070700676789e-06);
	ps = fma(z, ps, -1.98412698298579493134e-04);
	ps = fma(z, ps, 8.33333333332248946124e-03);
	ps = fma(z, ps, -1.66666666666666324348e-01);
	const double s = fma(z * r, ps, r);
	double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
	pc = fma(z, pc, -2.75573143513906633035e-07);
	pc = fma(z, pc, 2.48015872894767294178e-05);
	pc = fma(z, pc, -1.38888888888741095749e-03);
	pc = fma(z, pc, 4.16666666666666019037e-02);
	const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
	const double a = (q & 1) ? c : s, b = (q & 1) ? s : c;
	sn = (q & 2) ? -a : a;
	cs = ((q + 1) & 2) ? -b : b;
}

// sin(pi x), cos(pi x) for the window phases (|x| of a few units): the reduction x = n / 2 + r, |r| <= 1/4, is exact, so
// the results keep sincospi's symmetries (sin(pi) = 0 exactly, ...); then wf_sincos' kernels on pi r.  ~35 vector
// instructions against ocml's 70.
__device__ __forceinline__ void wf_sincospi(double x, double &sn, double &cs) {
	const double n = rint(x + x);
	const double r = fma(n, -0.5, x) * 3.14159265358979311600e+00;  // (n / 2 and the difference are exact)
	const int q = (int)n;
	const double z = r * r;
	double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
	ps = fma(z, ps, 2.75573137070700676789e-06);
	ps = fma(z, ps, -1.98412698298579493134e-04);
	ps = fma(z, ps, 8.33333333332248946124e-03);
	ps = fma(z, ps, -1.66666666666666324348e-01);
	const double s = fma(z * r, ps, r);
	double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
	pc = fma(z, pc, -2.75573143513906633035e-07);
	pc = fma(z, pc, 2.48015872894767294178e-05);
	pc = fma(z, pc, -1.38888888888741095749e-03);
	pc = fma(z, pc, 4.16666666666666019037e-02);
	const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
	const double a = (q & 1) ? c : s, b = (q & 1) ? s : c;
	sn = (q & 2) ? -a : a;
	cs = ((q + 1) & 2) ? -b : b;
}

// ======== 512-point complex transform at EIGHT points per lane (CheapTrick / Synthesis at N = 1024: 16 and 24 kHz) ==========
// The same construction one size down: 8 x 8 x 8 with two LDS exchanges and no barrier, half the registers of the 1024-point
// transform (four wavefronts per SIMD instead of two).  A radix-8 last stage would leave bin k and bin 512 - k in different
// lanes (k = j + 64 c is closed under negation only for j = 0, 32), so the last stage is split: a lane takes the EVEN outputs
// of butterfly j = t and the ODD outputs of butterfly j' = 64 - t (lane 0: j' = 0), each a 4-point transform of sums /
// differences of the butterfly's eight inputs -- the arithmetic of one radix-8 butterfly, twice its inputs read.  All twiddles
// are applied on the strided side of the second exchange, so the paired side only meets eighth roots of unity and no lane
// is special.  Layouts (t = lane, n0 = t & 7, n1 = t >> 3):
//   strided   slot q (0..7) holds element t + 64 q
//   paired8   slot c (0..3) holds element t + 128 c, slot 4 + c holds element 128 - t + 128 c (lane 0: 64 + 128 c): element k
//             and element 512 - k share a lane (slot c with slot 7 - c; lane 0 pairs slot 1 with 3, 4 with 7, 5 with 6 and
//             keeps the self-paired 0 and 256 in slots 0 and 2) -- the first two groups of the 1024-point "paired" layout
// LDS: 576 doubles.  Exchange 1: strided side t + 72 ka, middle side n0 + 72 ka + 8 n1; exchange 2: middle side P n0 + j
// (j = ka + 8 kb), paired side P m + t and P m + j'; P = 66 going to the paired layout, 68 coming from it (where rows
// 0..3 hold the even-half sums and rows 4..7 the odd-half ones): every access is conflict-free for its instruction's lane
// groups (ds_read_b64 32 lanes over 32 double-banks, ds_write_b64 16 lanes over 16).
constexpr int kWf8Lds = 576;

// 8-point DFT with only the first 2 NG inputs non-zero (NG = 4: all)
template <int S, int NG>
__device__ __forceinline__ void wdft8p(double (&xr)[8], double (&xi)[8]) {
	if constexpr (NG >= 3) {
		wdft8<S>(xr, xi);
	} else {
		if constexpr (NG == 2) {
			wdft4_2<S>(xr[0], xi[0], xr[2], xi[2], xr[4], xi[4], xr[6], xi[6]);
			wdft4_2<S>(xr[1], xi[1], xr[3], xi[3], xr[5], xi[5], xr[7], xi[7]);
		} else {
			xr[2] = xr[4] = xr[6] = xr[0]; xi[2] = xi[4] = xi[6] = xi[0];
			xr[3] = xr[5] = xr[7] = xr[1]; xi[3] = xi[5] = xi[7] = xi[1];
		}
		{ const double x = xr[3], y = xi[3]; xr[3] = kH * (S > 0 ? x - y : x + y); xi[3] = kH * (S > 0 ? x + y : y - x); }
		{ const double x = xr[5], y = xi[5]; xr[5] = S > 0 ? -y : y; xi[5] = S > 0 ? x : -x; }
		{ const double x = xr[7], y = xi[7]; xr[7] = S > 0 ? -kH * (x + y) : kH * (y - x); xi[7] = S > 0 ? kH * (x - y) : -kH * (x + y); }
		double tr[8], ti[8];
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			tr[c] = xr[2 * c] + xr[2 * c + 1]; ti[c] = xi[2 * c] + xi[2 * c + 1];
			tr[c + 4] = xr[2 * c] - xr[2 * c + 1]; ti[c + 4] = xi[2 * c] - xi[2 * c + 1];
		}
#pragma unroll
		for (int q = 0; q < 8; ++q) { xr[q] = tr[q]; xi[q] = ti[q]; }
	}
}
template <int WS, int RS>
__device__ __forceinline__ void wf8_xchg(double (&v)[8], double *lds, int wbase, int rbase) {
#pragma unroll
	for (int q = 0; q < 8; ++q) lds[wbase + q * WS] = v[q];
	wf_fence();
#pragma unroll
	for (int r = 0; r < 8; ++r) v[r] = lds[rbase + r * RS];
	wf_fence();
}
// slots 5, 6, 7 times W_8^{S}, W_8^{2 S}, W_8^{3 S}
template <int S>
__device__ __forceinline__ void wf8_rot_odd(double (&re)[8], double (&im)[8]) {
	{ const double x = re[5], y = im[5]; re[5] = kH * (S > 0 ? x - y : x + y); im[5] = kH * (S > 0 ? x + y : y - x); }
	{ const double x = re[6], y = im[6]; re[6] = S > 0 ? -y : y; im[6] = S > 0 ? x : -x; }
	{ const double x = re[7], y = im[7]; re[7] = S > 0 ? -kH * (x + y) : kH * (y - x); im[7] = S > 0 ? kH * (x - y) : -kH * (x + y); }
}
// strided -> paired8; the caller has run the leading wdft8p<S, NG> on the strided data
template <int S>
__device__ __forceinline__ void wf8_fft512_dit_rest(double (&re)[8], double (&im)[8], double *lds, const double2 *__restrict__ tw_, int lane) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	const int mid = (lane & 7) + 72 * (lane >> 3);
	{
		double2 wa[8];
#pragma unroll
		for (int r = 1; r < 8; ++r) wa[r] = tw_load(tw + kTw8A + 8 * r, lane >> 3);
		WF_SCHED_FENCE();
		wf8_xchg<72, 8>(re, lds, lane, mid);
		wf8_xchg<72, 8>(im, lds, lane, mid);
#pragma unroll
		for (int r = 1; r < 8; ++r) wrot(re[r], im[r], wa[r].x, S > 0 ? wa[r].y : -wa[r].y);
	}
	wdft8<S>(re, im);
	{
		double2 wb[8];
#pragma unroll
		for (int r = 0; r < 8; ++r) wb[r] = tw_load(tw + kTw8B + 64 * r, lane);
		WF_SCHED_FENCE();
#pragma unroll
		for (int r = 0; r < 8; ++r) wrot(re[r], im[r], wb[r].x, S > 0 ? wb[r].y : -wb[r].y);
	}
	const int w2 = 66 * (lane & 7) + (lane >> 3), jp = (64 - lane) & 63;
	auto half = [&](double (&v)[8]) {
#pragma unroll
		for (int q = 0; q < 8; ++q) lds[w2 + 8 * q] = v[q];
		wf_fence();
		double a[8], b[8];
#pragma unroll
		for (int m = 0; m < 8; ++m) { a[m] = lds[66 * m + lane]; b[m] = lds[66 * m + jp]; }
		wf_fence();
#pragma unroll
		for (int m = 0; m < 4; ++m) { v[m] = a[m] + a[m + 4]; v[4 + m] = b[m] - b[m + 4]; }
	};
	half(re);
	half(im);
	wf8_rot_odd<S>(re, im);
	wdft4<S>(re[0], im[0], re[1], im[1], re[2], im[2], re[3], im[3]);
	wdft4<S>(re[4], im[4], re[5], im[5], re[6], im[6], re[7], im[7]);
}
template <int S>
__device__ __forceinline__ void wf8_fft512_dit(double (&re)[8], double (&im)[8], double *lds, const double2 *__restrict__ tw, int lane) {
	wdft8<S>(re, im);
	wf8_fft512_dit_rest<S>(re, im, lds, tw, lane);
}
// paired8 -> strided
template <int S>
__device__ __forceinline__ void wf8_fft512_dif(double (&re)[8], double (&im)[8], double *lds, const double2 *__restrict__ tw_, int lane) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	wdft4<S>(re[0], im[0], re[1], im[1], re[2], im[2], re[3], im[3]);
	wdft4<S>(re[4], im[4], re[5], im[5], re[6], im[6], re[7], im[7]);
	wf8_rot_odd<S>(re, im);
	const int jp = (64 - lane) & 63, r2 = 68 * (lane & 3) + (lane >> 3);
	const double sg = (lane & 4) ? -1.0 : 1.0;
	{
		double2 wb[8];
#pragma unroll
		for (int r = 0; r < 8; ++r) wb[r] = tw_load(tw + kTw8B + 64 * r, lane);
		WF_SCHED_FENCE();
		auto half = [&](double (&v)[8]) {
#pragma unroll
			for (int m = 0; m < 4; ++m) { lds[68 * m + lane] = v[m]; lds[68 * (4 + m) + jp] = v[4 + m]; }
			wf_fence();
			double e[8], g[8];
#pragma unroll
			for (int q = 0; q < 8; ++q) { e[q] = lds[r2 + 8 * q]; g[q] = lds[r2 + 272 + 8 * q]; }
			wf_fence();
#pragma unroll
			for (int q = 0; q < 8; ++q) v[q] = fma(sg, g[q], e[q]);  // (exact: sg = +-1)
		};
		half(re);
		half(im);
#pragma unroll
		for (int r = 0; r < 8; ++r) wrot(re[r], im[r], wb[r].x, S > 0 ? wb[r].y : -wb[r].y);
	}
	wdft8<S>(re, im);
	const int mid = (lane & 7) + 72 * (lane >> 3);
	{
		double2 wa[8];
#pragma unroll
		for (int r = 1; r < 8; ++r) wa[r] = tw_load(tw + kTw8A + 8 * r, lane >> 3);
		WF_SCHED_FENCE();
#pragma unroll
		for (int r = 1; r < 8; ++r) wrot(re[r], im[r], wa[r].x, S > 0 ? wa[r].y : -wa[r].y);
	}
	wf8_xchg<8, 72>(re, lds, mid, lane);
	wf8_xchg<8, 72>(im, lds, mid, lane);
	wdft8<S>(re, im);
}
// bin held by slot 4 g + c of the paired8 layout
__device__ __forceinline__ int wf8_bin(int lane, int g, int c) { return (g == 0 ? lane : (lane ? 128 - lane : 64)) + 128 * c; }

// ---- real transforms of 1024 points on top of it: wf_r2c_unpack and friends on the first two groups --------------------------
__device__ __forceinline__ void wf8_tw_real(const double2 *__restrict__ tw, int lane, double (&wr)[4], double (&wi)[4]) {
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const double2 a = tw_load(tw + kTw8U + 128 * q, lane);
		wr[q] = a.x; wi[q] = a.y;
	}
	WF_SCHED_FENCE();
}
// In: the paired8 output of wf8_fft512_dit<+1> on the packed signal z[m] = x[2 m] + i x[2 m + 1].  Out: slot (g, c) holds
// 2 X[bin]; lane 0's slot 0 holds (2 X[0], 0) and nyq = 2 X[512] (valid on lane 0).
__device__ __forceinline__ void wf8_r2c_unpack(double (&re)[8], double (&im)[8], double &nyq, const double2 *__restrict__ tw_, int lane) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	double wr[4], wi[4];
	wf8_tw_real(tw, lane, wr, wi);
	nyq = 0.0;
	if (lane == 0) {
		const double a = re[0], b = im[0];
		re[0] = 2.0 * (a + b); im[0] = 0.0;
		nyq = 2.0 * (a - b);
		re[2] = 2.0 * re[2]; im[2] = 2.0 * im[2];                     // bin 256: X = Z
		wf_r2c_pair(re[1], im[1], re[3], im[3], kH, kH);              // 128 | 384
		wf_r2c_pair(re[4], im[4], re[7], im[7], kC8, kS8);            // 64 | 448
		wf_r2c_pair(re[5], im[5], re[6], im[6], kS8, kC8);            // 192 | 320
	} else {
#pragma unroll
		for (int q = 0; q < 4; ++q) wf_r2c_pair(re[q], im[q], re[7 - q], im[7 - q], wr[q], wi[q]);
	}
}
__device__ __forceinline__ void wf8_c2r_pack(double (&re)[8], double (&im)[8], double nyq, const double2 *__restrict__ tw_, int lane) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	double wr[4], wi[4];
	wf8_tw_real(tw, lane, wr, wi);
	if (lane == 0) {
		const double y0 = re[0];
		re[0] = y0 + nyq; im[0] = y0 - nyq;
		re[2] = 2.0 * re[2]; im[2] = 2.0 * im[2];
		wf_c2r_pair(re[1], im[1], re[3], im[3], kH, kH);
		wf_c2r_pair(re[4], im[4], re[7], im[7], kC8, kS8);
		wf_c2r_pair(re[5], im[5], re[6], im[6], kS8, kC8);
	} else {
#pragma unroll
		for (int q = 0; q < 4; ++q) wf_c2r_pair(re[q], im[q], re[7 - q], im[7 - q], wr[q], wi[q]);
	}
}
// real parts only (the input is real and even) / a REAL spectrum in: see wf_r2c_unpack_re, wf_c2r_pack_re
__device__ __forceinline__ void wf8_r2c_unpack_re(double (&re)[8], const double (&im)[8], double &nyq, const double2 *__restrict__ tw_, int lane) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	double wr[4], wi[4];
	wf8_tw_real(tw, lane, wr, wi);
	nyq = 0.0;
	if (lane == 0) {
		const double a = re[0], b = im[0];
		re[0] = 2.0 * (a + b);
		nyq = 2.0 * (a - b);
		re[2] = 2.0 * re[2];
		wf_r2c_pair_re(re[1], im[1], re[3], im[3], kH, kH);
		wf_r2c_pair_re(re[4], im[4], re[7], im[7], kC8, kS8);
		wf_r2c_pair_re(re[5], im[5], re[6], im[6], kS8, kC8);
	} else {
#pragma unroll
		for (int q = 0; q < 4; ++q) wf_r2c_pair_re(re[q], im[q], re[7 - q], im[7 - q], wr[q], wi[q]);
	}
}
__device__ __forceinline__ void wf8_c2r_pack_re(double (&re)[8], double (&im)[8], double nyq, const double2 *__restrict__ tw_, int lane) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	double wr[4], wi[4];
	wf8_tw_real(tw, lane, wr, wi);
	if (lane == 0) {
		const double y0 = re[0];
		re[0] = y0 + nyq; im[0] = y0 - nyq;
		re[2] = 2.0 * re[2]; im[2] = 0.0;
		wf_c2r_pair_re(re[1], im[1], re[3], im[3], kH, kH);
		wf_c2r_pair_re(re[4], im[4], re[7], im[7], kC8, kS8);
		wf_c2r_pair_re(re[5], im[5], re[6], im[6], kS8, kC8);
	} else {
#pragma unroll
		for (int q = 0; q < 4; ++q) wf_c2r_pair_re(re[q], im[q], re[7 - q], im[7 - q], wr[q], wi[q]);
	}
}

// ======== real EVEN transform of 2048 points at half a real transform's cost ================================================
// F[k] = sum_{n < 2048} x[n] cos(2 pi n k / 2048) for x[n] = x[2048 - n] -- the DFT of a real even sequence (CheapTrick's two
// cepstral transforms, the first transform of a minimum-phase analysis), itself real and even.  With
//     c[n] = x[2 n] + i (x[2 n + 1] - x[2 n - 1]),   n < 1024   (Hermitian: c[1024 - n] = conj c[n]),
// C = DFT_1024(c) is real: C[k] = E[k] - 2 sin(theta_k) T[k], theta_k = 2 pi k / 2048, where E is the transform of the even
// samples and T[k] = w^k O[k] the twiddled transform of the odd ones, F[k] = E[k] + T[k].  E[1024 - k] = E[k] and
// T[1024 - k] = -T[k], so the pair (C[k], C[1024 - k]) gives E[k] = (C[k] + C[1024 - k]) / 2, T[k] = (C[1024 - k] - C[k]) / (4 sin
// theta_k), hence F[k] and F[1024 - k]; T[0] is the plain sum of the odd samples.  C itself is a 1024-point real-output
// transform of a Hermitian sequence: wf8_c2r_pack + the 512-point complex transform at eight points per lane.  About 700
// instructions where the 1024-point complex transform + unpacking take 1250; the division by 4 sin theta amplifies the
// rounding of C by at most 163 (k = 1): 1e-13 of the largest |C|.
// In: x[0 .. 1024] in LDS (X, 16-byte aligned), natural order.  L: the exchange buffer (>= 1026 doubles, not X).
// Out: lo[j] = F[t + 64 j], hi[j] = F[1024 - t - 64 j], j < 8 (lane 0, j = 0: F[0] and F[1024]); mid = F[512] (every lane).
__device__ __forceinline__ void wf_even2048(const double *X, double *L, const double2 *__restrict__ tw_, int lane, double (&lo)[8],
											 double (&hi)[8], double &mid) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	double cr[8], ci[8], so = 0.0;
#pragma unroll
	for (int g = 0; g < 2; ++g)
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			const int n = wf8_bin(lane, g, c);  // 0 .. 511
			const double2 v = *reinterpret_cast<const double2 *>(&X[2 * n]);
			const double m = X[n ? 2 * n - 1 : 1];
			cr[4 * g + c] = v.x;
			ci[4 * g + c] = v.y - m;
			so += v.y;
		}
	const double nyq = X[1024];
	wf_fence();
	wf8_c2r_pack(cr, ci, nyq, tw, lane);
	wf8_fft512_dif<-1>(cr, ci, L, tw, lane);
	double is[8];
#pragma unroll
	for (int j = 0; j < 8; ++j) is[j] = tw_load_d(tw + kTwIS, lane + 64 * j);
	WF_SCHED_FENCE();
#pragma unroll
	for (int q = 0; q < 8; ++q) *reinterpret_cast<double2 *>(&L[2 * (lane + 64 * q)]) = make_double2(cr[q], ci[q]);
	wf_fence();
	so = 2.0 * wave_sum_all(so);
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const int k = lane + 64 * j;
		const double a = L[k], b = L[(1024 - k) & 1023];
		const double e = 0.5 * (a + b), t = (b - a) * is[j];
		lo[j] = e + t;
		hi[j] = e - t;
	}
	if (lane == 0) {
		const double e0 = L[0];
		lo[0] = e0 + so;
		hi[0] = e0 - so;
	}
	mid = L[512];
	wf_fence();
}

}  // namespace wc
