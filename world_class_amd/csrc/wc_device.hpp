// Device-side building blocks shared by the stage kernels (gfx950, wave64, FP64).
//   * block reductions / scans built on 64-lane wave shuffles
//   * LDS-resident Stockham FFT (radix-4 passes, optional leading radix-2) on interleaved complex
//     doubles, plus the real<->half-size-complex pre/post passes
//   * the reference's small arithmetic helpers (matlab_round, interp1Q) as __device__ inlines
//
// FFT conventions are the reference's (reference src/world_fft.cpp:31-77 over Ooura):
//   r2c  X[k] = sum x[n] e^{+2 pi i k n / N}            (SIGN = +1, "forward")
//   c2r  y[n] = sum_k Yh[k] e^{-2 pi i k n / N}, Yh = Hermitian extension, imag of bins 0, N/2 ignored
//   c2c  forward e^{+i}, backward e^{-i}; nothing is normalised.
// Compiled with -ffp-contract=off: every fused multiply-add below is an explicit fma().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wc {

constexpr double kPi = 3.1415926535897932384;
constexpr int kTwiddleLog2 = 12;
constexpr int kTwiddleN = 4096;  // table W[k] = e^{+2 pi i k / 4096}, k < 4096

// reference src/world_matlabfunctions.cpp:212-214
__device__ __forceinline__ int mround(double x) { return x > 0 ? (int)(x + 0.5) : (int)(x - 0.5); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return max(lo, min(hi, v)); }
// x / c for a divisor known in advance, with rc = 1 / c correctly rounded (an IEEE division on the host): a product, the exact
// remainder, one correction -- three instructions where the division's scale / reciprocal / refine / fix-up sequence takes eleven.
// The quotient is the correctly rounded one (Markstein's theorem for a correctly rounded reciprocal; held to the hardware's
// division bit for bit on 2^24 values per divisor in tests/test_gpu_blocks.py), for finite x away from the over- / underflow range.
__device__ __forceinline__ double div_const(double x, double c, double rc) {
	const double q0 = x * rc;
	const double r = fma(-q0, c, x);
	return fma(r, rc, q0);
}

// reference src/world_matlabfunctions.cpp:220-241 for one abscissa (y indexed by a functor so the
// table may live in LDS or be a mirrored view)
template <class F>
__device__ __forceinline__ double interp1q(double x0, double dx, F y, int n, double xi) {
	double q = (xi - x0) / dx;
	int b = (int)q;
	double frac = q - b;
	double y0 = y(b);
	double dy = (b == n - 1) ? 0.0 : y(b + 1) - y0;
	return y0 + dy * frac;
}

// The same with the quotient (xi - x0) / dx formed from a precomputed reciprocal and one residual correction
// (within an ulp of the division at a fifth of its cost).  Where the last bit moves q across an integer the
// interpolant is continuous, so the result moves by rounding noise only.
template <class F>
__device__ __forceinline__ double interp1q_rcp(double x0, double dx, double rdx, F y, int n, double xi) {
	const double t = xi - x0;
	double q = t * rdx;
	q = fma(fma(-dx, q, t), rdx, q);
	int b = (int)q;
	double frac = q - b;
	double y0 = y(b);
	double dy = (b == n - 1) ? 0.0 : y(b + 1) - y0;
	return fma(dy, frac, y0);
}

// ---- wave / block collectives ------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	return v;  // valid in lane 0
}
__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		double t = __shfl_up(v, o, 64);
		if (lane >= o) v += t;
	}
	return v;
}
// Sum over the block; result broadcast to every thread.  scratch: >= T/64 doubles of LDS.
template <int T>
__device__ __forceinline__ double block_sum(double v, double *scratch, int tid) {
	v = wave_sum(v);
	__syncthreads();
	if ((tid & 63) == 0) scratch[tid >> 6] = v;
	__syncthreads();
	double s = 0.0;
#pragma unroll
	for (int w = 0; w < T / 64; ++w) s += scratch[w];
	return s;
}
// Two sums at once.
template <int T>
__device__ __forceinline__ void block_sum2(double &a, double &b, double *scratch, int tid) {
	a = wave_sum(a);
	b = wave_sum(b);
	__syncthreads();
	if ((tid & 63) == 0) { scratch[tid >> 6] = a; scratch[T / 64 + (tid >> 6)] = b; }
	__syncthreads();
	double sa = 0.0, sb = 0.0;
#pragma unroll
	for (int w = 0; w < T / 64; ++w) { sa += scratch[w]; sb += scratch[T / 64 + w]; }
	a = sa;
	b = sb;
}
// Thread-index arithmetic (LDS addresses of every FFT pass, index -> double conversions ...) is loop/phase invariant, so
// the compiler computes all of it up front and keeps -- or spills -- it for the whole kernel.  An opaque copy of the
// index per phase keeps those values short-lived; they cost a few integer instructions to recompute.
#define WC_FRESH(v) asm volatile("" : "+v"(v))
// A wave-uniform double moved to scalar registers: loop-invariant uniform values otherwise sit in (and spill from)
// vector registers, one copy per lane, because f64 arithmetic is vector-only.
__device__ __forceinline__ double uniform_d(double v) {
	const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
	const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
	return __hiloint2double(hi, lo);
}
// Three sums at once.  scratch: >= 3 * T/64 doubles.
template <int T>
__device__ __forceinline__ void block_sum3(double &a, double &b, double &c, double *scratch, int tid) {
	a = wave_sum(a);
	b = wave_sum(b);
	c = wave_sum(c);
	__syncthreads();
	if ((tid & 63) == 0) {
		scratch[tid >> 6] = a;
		scratch[T / 64 + (tid >> 6)] = b;
		scratch[2 * (T / 64) + (tid >> 6)] = c;
	}
	__syncthreads();
	double sa = 0.0, sb = 0.0, sc = 0.0;
#pragma unroll
	for (int w = 0; w < T / 64; ++w) { sa += scratch[w]; sb += scratch[T / 64 + w]; sc += scratch[2 * (T / 64) + w]; }
	a = sa;
	b = sb;
	c = sc;
}
// Exclusive scan of one value per thread across the block.  scratch: >= T/64 doubles.
template <int T>
__device__ __forceinline__ double block_excl_scan(double v, double *scratch, int tid) {
	int lane = tid & 63, w = tid >> 6;
	double inc = wave_incl_scan(v, lane);
	__syncthreads();
	if (lane == 63) scratch[w] = inc;
	__syncthreads();
	double base = 0.0;
#pragma unroll
	for (int k = 0; k < T / 64; ++k) if (k < w) base += scratch[k];
	return base + inc - v;
}
// Exclusive running maximum of one value per thread across the block (values >= 0).  Used to take the ulp-sized
// inversions out of tree-scanned prefix sums of non-negative terms: a tree scan gives every position its own association
// order, so such sums can DEcrease by an ulp from one thread's segment to the next, which the reference's sequential
// cumulative sum (src/world_common.cpp:95-101) never does -- and LinearSmoothing's differences of them feed a logarithm
// (CheapTrick) and a divisor (D4C).  max is exact in any order, so the clamped sequence is non-decreasing.
// scratch: >= T/64 doubles.
template <int T>
__device__ __forceinline__ double block_excl_max_scan(double v, double *scratch, int tid) {
	const int lane = tid & 63, w = tid >> 6;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		double t = __shfl_up(v, o, 64);
		if (lane >= o) v = fmax(v, t);
	}
	__syncthreads();
	if (lane == 63) scratch[w] = v;
	__syncthreads();
	double m = __shfl_up(v, 1, 64);
	if (lane == 0) m = 0.0;
#pragma unroll
	for (int k = 0; k < T / 64; ++k) if (k < w) m = fmax(m, scratch[k]);
	return m;
}

// ---- the reference's sequential cumulative sum, bit for bit, in parallel ------------------------------------------
// c[i] = fl(v[i] + c[i-1]), strictly left to right, is what reference src/world_common.cpp:47-51 computes for
// LinearSmoothing, and the smoothed value is a DIFFERENCE of two neighbourhoods of c: where the terms are small against
// the running sum (noise-free bands of a clean chirp, the gaps between the harmonics of a synthetic voice) the result is
// decided by how each single addition rounded -- a tree-ordered prefix sum is more accurate and exactly therefore
// 1e-6 away from the reference.  The sequential rounding is reproduced instead:
//   * while the running sum c stays inside one binade [2^e, 2^(e+1)), every addition of a term v >= 0 adds the integer
//     rn(v / ulp) ulps, whatever c is -- unless v / ulp lies exactly half way (round-to-even then looks at c's parity).
//     A thread whose CH consecutive terms provably stay inside one binade (decided from a tree-ordered estimate of c,
//     good to 1e-13, with a 2^-30 margin either side) and contain no tie is "clean": its exact increment d is obtained
//     by adding its terms to 2^e; sums of such increments of one binade are exact in any order (segmented scan).
//   * the remaining threads (binade crossings, ties, the very first terms; a handful per spectrum) are walked in order
//     by one wavefront with the reference's own floating-point additions, hopping over the clean runs in between.
//     (it writes their partial sums itself)
//   * every clean thread then re-adds its terms from its exact start value.
// In: S[0 .. len) = the terms (non-negative).  Out: S[i] = c[i].  scr: >= T + 2 (T / 64) doubles of LDS, red: >= T / 64.
// Ends with a __syncthreads().
template <int T>
__device__ __forceinline__ void seq_cumsum_nonneg(double *S, int len, double *scr, double *red, int tid) {
	constexpr int W = T / 64;
	double *pre = scr, *wt = scr + T + W;  // pre[t]: inclusive run prefix of a clean thread / sum behind a dirty thread's terms
	unsigned long long *msk = reinterpret_cast<unsigned long long *>(scr + T);
	const int lane = tid & 63, w = tid >> 6;
	const int ch = (len + T - 1) / T;
	const int lo = min(tid * ch, len), hi = min(len, lo + ch);
	double loc = 0.0;
	bool bad = false;
	for (int i = lo; i < hi; ++i) {
		const double v = S[i];
		loc += v;
		bad = bad || !(v >= 0.0);
	}
	const double base_a = block_excl_scan<T>(loc, red, tid);
	const double lower = base_a * (1.0 - 0x1p-30), upper = (base_a + loc) * (1.0 + 0x1p-30);
	const int E = (__double2hiint(lower) >> 20) & 0x7ff;
	bool dirty = tid == 0 || bad || !(lower > 0.0) || E != ((__double2hiint(upper) >> 20) & 0x7ff) || E < 64 || E > 1984;
	double d = 0.0;
	if (lo >= hi) {
		dirty = tid == 0;
	} else if (!dirty) {
		const double C = __hiloint2double(E << 20, 0);               // 2^e, the start of the binade
		const double rulp = __hiloint2double((2098 - E) << 20, 0);   // 2^(52 - e) = 1 / ulp
		double r = C;
		for (int i = lo; i < hi; ++i) {
			const double v = S[i], t = v * rulp;
			dirty = dirty || (t - floor(t)) == 0.5;
			r = v + r;
		}
		d = dirty ? 0.0 : r - C;
	}
	// inclusive segmented scan of d over the threads, restarting behind every dirty thread
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
	const unsigned long long m = __ballot(dirty);
	if (lane == 63) wt[w] = x;
	if (lane == 0) msk[w] = m;
	__syncthreads();
	if (!f)
		for (int k = w - 1; k >= 0; --k) {
			x += wt[k];
			if (msk[k]) break;
		}
	pre[tid] = x;
	__syncthreads();
	if (w == 0) {  // the walk: wave-uniform, every lane computes the same values
		double so = 0.0;
		int prev = -1;
		for (int k = 0; k < W; ++k) {
			unsigned long long mk = msk[k];
			mk = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(mk >> 32)) << 32) |
				 (unsigned)__builtin_amdgcn_readfirstlane((int)(mk & 0xffffffffull));
			while (mk) {
				const int t = k * 64 + __ffsll((long long)mk) - 1;
				mk &= mk - 1;
				const double si = prev < 0 ? 0.0 : (t == prev + 1 ? so : so + pre[t - 1]);
				const int l2 = min(t * ch, len), h2 = min(len, l2 + ch);
				double run = si;
				for (int i = l2; i < h2; ++i) {
					run = S[i] + run;
					if (lane == 0) S[i] = run;
				}
				if (lane == 0) pre[t] = run;
				so = run;
				prev = t;
			}
		}
	}
	__syncthreads();
	if (!dirty) {
		const unsigned long long below = m & ((1ull << lane) - 1ull);
		int j;
		if (below) {
			j = w * 64 + 63 - __clzll((long long)below);
		} else {  // (thread 0 is always dirty: the search ends in wave 0 at the latest)
			int k = w - 1;
			while (k > 0 && msk[k] == 0ull) --k;
			j = k * 64 + 63 - __clzll((long long)msk[k]);
		}
		double run = pre[j] + (x - d);
		for (int i = lo; i < hi; ++i) {
			run = S[i] + run;
			S[i] = run;
		}
	}
	__syncthreads();
}

// ---- complex helpers ----------------------------------------------------------------------------
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
	return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// multiply by +i (S=+1) or -i (S=-1)
template <int S>
__device__ __forceinline__ double2 cmul_i(double2 a) {
	return S > 0 ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
}
// Load from the (global) twiddle table through an explicit global-address-space pointer: a pointer that went
// through tw_fresh() is opaque to the compiler, which would otherwise emit FLAT loads -- those count against
// lgkmcnt as well, so every LDS wait would also wait for the outstanding twiddle loads.
#ifndef WC_TW_ABLATE
#define WC_TW_ABLATE 0  // 1: timing ablation (wrong results): twiddles made up from the index instead of loaded
#endif
__device__ __forceinline__ double2 tw_load(const double2 *tw, int idx) {
#if WC_TW_ABLATE
	return make_double2(1.0 - idx * 1e-9, idx * 1e-9);
#endif
	typedef double v2d __attribute__((ext_vector_type(2)));
	typedef const v2d __attribute__((address_space(1))) *gptr;
	const v2d v = ((gptr)tw)[idx];
	return make_double2(v.x, v.y);
}
template <int S>
__device__ __forceinline__ double2 twiddle(const double2 *__restrict__ tw, int idx) {
	double2 w = tw_load(tw, idx);
	return S > 0 ? w : cconj(w);
}

// ---- in-LDS complex FFT of M points by T threads, sign S (+1: e^{+i}) -------------------------------
// a: M interleaved complex doubles in LDS (fft_lds_size(M) entries), natural order in and out.
// Stockham autosort, radix-4 passes (a leading radix-2 pass when log2 M is odd), two barriers per pass.  tw: global twiddle table of kTwiddleN entries, loaded before the LDS
// reads of each pass so that the L2 latency overlaps them.  Ends with a __syncthreads().
#ifndef WC_FFT_PAD
#define WC_FFT_PAD 0  // measured: the padded layout does not pay for its index arithmetic with the unrolled passes
#endif
// intermediate passes keep the data in a padded layout (one spare slot after every 4 entries): the stride-4 /
// stride-16 scatter of the early passes then hits distinct LDS banks
__host__ __device__ constexpr int fft_lds_size(int M) { return WC_FFT_PAD ? M + M / 4 : M; }
__device__ __forceinline__ int fft_pad(int e, bool pad) { return (WC_FFT_PAD && pad) ? e + (e >> 2) : e; }
// j < NB, decided at compile time when the butterflies divide evenly among the threads (no predication then)
template <int NB, int T>
__device__ __forceinline__ bool fft_in_range(int j) { return (NB % T == 0) ? true : (j < NB); }

// R-point DFT in registers, sign S: y[q] = sum_r x[r] e^{S 2 pi i r q / R}
template <int S>
__device__ __forceinline__ void dft2(double2 &a, double2 &b) {
	const double2 t = csub(a, b);
	a = cadd(a, b);
	b = t;
}
template <int S>
__device__ __forceinline__ void dft4(double2 (&x)[4]) {
	const double2 s02 = cadd(x[0], x[2]), d02 = csub(x[0], x[2]);
	const double2 s13 = cadd(x[1], x[3]), d13 = cmul_i<S>(csub(x[1], x[3]));
	x[0] = cadd(s02, s13);
	x[1] = cadd(d02, d13);
	x[2] = csub(s02, s13);
	x[3] = csub(d02, d13);
}
template <int S>
__device__ __forceinline__ void dft8(double2 (&x)[8]) {
	constexpr double h = 0.70710678118654752440;
	// even / odd halves
	double2 e[4] = {x[0], x[2], x[4], x[6]};
	double2 o[4] = {x[1], x[3], x[5], x[7]};
	dft4<S>(e);
	dft4<S>(o);
	// o[q] *= e^{S i pi q / 4}
	const double2 o1 = S > 0 ? make_double2(h * (o[1].x - o[1].y), h * (o[1].x + o[1].y))
							 : make_double2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));
	const double2 o2 = cmul_i<S>(o[2]);
	const double2 o3 = S > 0 ? make_double2(-h * (o[3].x + o[3].y), h * (o[3].x - o[3].y))
							 : make_double2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y));
	x[0] = cadd(e[0], o[0]); x[4] = csub(e[0], o[0]);
	x[1] = cadd(e[1], o1);   x[5] = csub(e[1], o1);
	x[2] = cadd(e[2], o2);   x[6] = csub(e[2], o2);
	x[3] = cadd(e[3], o3);   x[7] = csub(e[3], o3);
}

// The twiddles a thread needs are the same for every transform of one size, so the compiler would load them once
// per kernel and pin them in registers (about 60 VGPRs for M = 2048) -- which costs more in spills / occupancy
// than re-reading them from the (L1/L2 resident) table.  Laundering the table pointer per call prevents that.
// WC_FFT_TW: 0 = launder per pass (twiddles loaded pass by pass), 1 = per transform (the compiler may prefetch a
// whole transform's twiddles), 2 = never (pinned for the kernel).
#ifndef WC_FFT_RADIX8
#define WC_FFT_RADIX8 0
#endif
#ifndef WC_FFT_TW
#define WC_FFT_TW 1
#endif
__device__ __forceinline__ const double2 *tw_fresh(const double2 *tw) {
#if WC_FFT_TW < 2
	asm volatile("" : "+s"(tw));
#endif
	return tw;
}

// radix of the pass that starts at Ns (product of the earlier radices); 1 when the transform is complete
template <int M, int Ns>
__host__ __device__ constexpr int fft_radix() {
	if (Ns >= M) return 1;
	constexpr int rem = M / (Ns < M ? Ns : M);
#if WC_FFT_RADIX8
	// radix 8 with a leading radix-2 / radix-4 pass when log2(M) is not a multiple of 3
	constexpr int lead = 1 << (__builtin_ctz(M) % 3);
	return (Ns == 1 && lead > 1) ? lead : (rem >= 8 ? 8 : rem);
#else
	// radix 4 throughout, with a leading twiddle-free radix-2 pass when log2(M) is odd
	constexpr bool odd = (__builtin_ctz(M) & 1) != 0;
	return (Ns == 1 && odd) ? 2 : (rem >= 4 ? 4 : rem);
#endif
}

// the twiddles W_{R Ns}^{r k}, r = 1 .. R-1, of a thread's butterflies in the pass (R, Ns)
template <int M, int T, int R, int Ns>
struct FftTw {
	static constexpr int NB = M / (R > 0 ? R : 1);
	static constexpr int BPT = (NB + T - 1) / T;
	double2 w[BPT][R > 1 ? R - 1 : 1];
};
template <int M, int T, int S, int R, int Ns, int FL>
__device__ __forceinline__ void fft_load_tw(FftTw<M, T, R, Ns> &tw_regs, const double2 *__restrict__ tw, int tid) {
	if constexpr (R > 1 && Ns > 1 && Ns < M) {
		constexpr int NB = M / R;
		constexpr int BPT = (NB + T - 1) / T;
		constexpr int tstride = kTwiddleN / (R * Ns);  // W_{R Ns}^k = tw[k * tstride]
#pragma unroll
		for (int b = 0; b < BPT; ++b) {
			const int j = tid + b * T;
			if (fft_in_range<NB, T>(j)) {
				const int idx = (j & (Ns - 1)) * tstride;
#pragma unroll
				for (int r = 1; r < R; ++r)
					tw_regs.w[b][r - 1] = (FL & 2) ? make_double2(1.0 - r * idx * 1e-9, r * idx * 1e-9) /* timing ablation */ : twiddle<S>(tw, r * idx);
			}
		}
	}
}

// One Stockham pass of radix R with Ns = product of the earlier radices.  The pass's own twiddles arrive in
// registers; the NEXT pass's twiddles are requested from the global table right after the LDS reads, so their
// latency hides behind this pass's barrier and butterflies instead of standing at the head of the next pass.
template <int M, int T, int S, int R, int Ns, int FL = 1>
__device__ __forceinline__ void fft_pass(double2 *a, const double2 *__restrict__ tw, int tid, const FftTw<M, T, R, Ns> &cur,
										 FftTw<M, T, fft_radix<M, Ns * R>(), Ns * R> &next) {
	constexpr int NB = M / R;
	constexpr int BPT = (NB + T - 1) / T;
	double2 v[BPT][R];
#pragma unroll
	for (int b = 0; b < BPT; ++b) {
		const int j = tid + b * T;
		if (fft_in_range<NB, T>(j)) {
#pragma unroll
			for (int r = 0; r < R; ++r) v[b][r] = a[fft_pad(j + r * NB, Ns > 1)];
		}
	}
	// (with several butterflies per thread the extra registers cost occupancy: request late, overlapping only the
	// closing barrier)
	constexpr bool early = BPT == 1;
	if (early) fft_load_tw<M, T, S, fft_radix<M, Ns * R>(), Ns * R, FL>(next, tw, tid);
	if (FL & 1) __syncthreads();
#pragma unroll
	for (int b = 0; b < BPT; ++b) {
		const int j = tid + b * T;
		if (fft_in_range<NB, T>(j)) {
			if (Ns > 1) {
#pragma unroll
				for (int r = 1; r < R; ++r) v[b][r] = cmul(v[b][r], cur.w[b][r - 1]);
			}
			if (R == 8) dft8<S>(reinterpret_cast<double2(&)[8]>(v[b]));
			else if (R == 4) dft4<S>(reinterpret_cast<double2(&)[4]>(v[b]));
			else dft2<S>(v[b][0], v[b][1]);
			const int k = j & (Ns - 1);
			const int j0 = (j - k) * R + k;
#pragma unroll
			for (int r = 0; r < R; ++r) a[fft_pad(j0 + r * Ns, Ns * R < M)] = v[b][r];
		}
	}
	if (!early) fft_load_tw<M, T, S, fft_radix<M, Ns * R>(), Ns * R, FL>(next, tw, tid);
	if (FL & 1) __syncthreads();
}

template <int M, int T, int S, int Ns, int FL = 1>
__device__ __forceinline__ void fft_chain(double2 *a, const double2 *__restrict__ tw, int tid,
										  const FftTw<M, T, fft_radix<M, Ns>(), Ns> &cur) {
	if constexpr (Ns < M) {
		constexpr int R = fft_radix<M, Ns>();
		FftTw<M, T, fft_radix<M, Ns * R>(), Ns * R> next;
		fft_pass<M, T, S, R, Ns, FL>(a, tw, tid, cur, next);
		fft_chain<M, T, S, Ns * R, FL>(a, tw, tid, next);
	}
}

template <int M, int T, int S, int FL = 1>
__device__ __forceinline__ void fft_lds(double2 *a, const double2 *__restrict__ tw_, int tid) {
	static_assert((M & (M - 1)) == 0 && M >= 16 && M <= kTwiddleN, "M must be a power of two in [16, 4096]");
	const double2 *__restrict__ tw = tw_fresh(tw_);
	FftTw<M, T, fft_radix<M, 1>(), 1> first;  // the first pass has no twiddles (Ns = 1)
	fft_chain<M, T, S, 1, FL>(a, tw, tid, first);
}

// The passes from Ns = NS0 on, for callers that can write the state after the first passes directly (an input that is
// zero beyond its first M / NS0 entries makes those passes pure replication).
template <int M, int T, int S, int NS0>
__device__ __forceinline__ void fft_lds_tail(double2 *a, const double2 *__restrict__ tw_, int tid) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	FftTw<M, T, fft_radix<M, NS0>(), NS0> first;
	fft_load_tw<M, T, S, fft_radix<M, NS0>(), NS0, 1>(first, tw, tid);
	fft_chain<M, T, S, NS0, 1>(a, tw, tid, first);
}

// W_N^k = e^{+2 pi i k / N}, N = 2 M, for the unpacking passes of the real transforms: from the table while it is fine
// enough, computed where N = 8192 needs odd multiples of half a table step (the 96 kHz D4C transforms; a few per thread)
template <int M>
__device__ __forceinline__ double2 tw_real(const double2 *__restrict__ tw, int k) {
	if constexpr (2 * M <= kTwiddleN) {
		return tw_load(tw, k * (kTwiddleN / (2 * M)));
	} else {
		double sn, cs;
		sincospi((double)k / M, &sn, &cs);
		return make_double2(cs, sn);
	}
}

// ---- real FFT of N = 2M points held as M interleaved complex (x[2k], x[2k+1]) -----------------------
// After fft_lds<M,T,+1> on that array, unpack to the spectrum X[0..M] (reference r2c convention).
// Packed in place: a[0] = (X[0].re, X[M].re); a[k] = X[k] for 0 < k < M.  Ends with a __syncthreads().
template <int M, int T>
__device__ __forceinline__ void r2c_post(double2 *a, const double2 *__restrict__ tw_, int tid) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	// pairs (k, M-k), k = 1 .. M/2-1 ; k = 0 and k = M/2 handled apart
	for (int k = tid; k <= M / 2; k += T) {
		if (k == 0) {
			double2 z = a[0];
			a[0] = make_double2(z.x + z.y, z.x - z.y);
		} else if (k == M / 2) {
			// X[M/2] = Z[M/2] for the e^{+i} convention
		} else {
			double2 zk = a[k], zm = a[M - k];
			double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));   // (Zk + conj Zm)/2
			double2 o = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));  // (Zk - conj Zm)/(2i)
			double2 w = tw_real<M>(tw, k);
			double2 wo = cmul(w, o);
			a[k] = cadd(e, wo);
			// X[M-k] = conj(E) + W^{M-k} conj(O),  W^{M-k} = -conj(W^k)  =>  X[M-k] = conj(E - W O)
			a[M - k] = cconj(csub(e, wo));
		}
	}
	__syncthreads();
}
// Power spectrum |X[k]|^2, k = 0..M, straight from the half-size FFT output into registers, with the arithmetic of
// r2c_post followed by re^2 + im^2.  Thread t gets the bin pairs (k, M-k) for k = t + e T (k = 0: bins 0 and M) in
// key[2e], key[2e+1]; thread 0 also gets bin M/2 in the last slot (unused on the other threads).  Read-only on a.
template <int M, int T>
__device__ __forceinline__ void r2c_power(const double2 *a, const double2 *__restrict__ tw_, int tid,
										  double (&key)[2 * ((M / 2) / T) + 1]) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	constexpr int PAIRS = (M / 2) / T;
#pragma unroll
	for (int e = 0; e < PAIRS; ++e) {
		const int k = tid + e * T;
		const double2 zk = a[k], zm = a[(M - k) & (M - 1)];
		const double2 ev = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
		const double2 od = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
		const double2 wo = cmul(tw_real<M>(tw, k), od);
		const double2 xk = cadd(ev, wo), xm = csub(ev, wo);
		const double r0 = zk.x + zk.y, rm = zk.x - zk.y;  // k = 0: X[0], X[M] (both real)
		key[2 * e] = (k == 0) ? r0 * r0 : fma(xk.x, xk.x, xk.y * xk.y);
		key[2 * e + 1] = (k == 0) ? rm * rm : fma(xm.x, xm.x, xm.y * xm.y);
	}
	const double2 zh = a[M / 2];
	key[2 * PAIRS] = fma(zh.x, zh.x, zh.y * zh.y);
}
// Inverse of the above: a holds the packed spectrum Y (a[0] = (Y[0].re, Y[M].re)); produce Z so that
// fft_lds<M,T,-1> yields the real signal y[n] interleaved (reference c2r convention, unnormalised).
template <int M, int T>
__device__ __forceinline__ void c2r_pre(double2 *a, const double2 *__restrict__ tw_, int tid) {
	const double2 *__restrict__ tw = tw_fresh(tw_);
	for (int k = tid; k <= M / 2; k += T) {
		if (k == 0) {
			double2 y = a[0];
			// E0 = Y0 + YM, O0 = Y0 - YM (both real) ; Z0 = E0 + i O0
			a[0] = make_double2(y.x + y.y, y.x - y.y);
		} else if (k == M / 2) {
			// E = Y + conj Y = 2 Re, O = (Y - conj Y) * conj(W^{M/2}) = 2i Im * (-i) = 2 Im ; Z = E + iO
			double2 y = a[k];
			a[k] = make_double2(2.0 * y.x, 2.0 * y.y);
		} else {
			double2 yk = a[k], ym = a[M - k];
			double2 e = make_double2(yk.x + ym.x, yk.y - ym.y);  // Yk + conj Ym
			double2 d = make_double2(yk.x - ym.x, yk.y + ym.y);  // Yk - conj Ym
			double2 w = cconj(tw_real<M>(tw, k));
			double2 o = cmul(d, w);
			a[k] = make_double2(e.x - o.y, e.y + o.x);  // E + i O
			// index M-k: E' = conj(E), D' = -conj(D), conj(W^{M-k}) = -W^k ... O' = conj(D) W^k = conj(D conj W) = conj(O)
			a[M - k] = make_double2(e.x + o.y, -e.y + o.x);  // conj(E) + i conj(O)
		}
	}
	__syncthreads();
}

}  // namespace wc
