// TEST INFRASTRUCTURE ONLY -- see wc_oracle.h.  CPU restatement (own code, FP64) of the reference
// hot path Harvest -> CheapTrick -> D4C -> Synthesis of yukara-ikemiya/world-class.  Each function
// cites the reference file:line whose arithmetic it restates.  The FFT is our own radix-2 code
// with the reference's conventions (forward = e^{+i w n}, unnormalised); rounding therefore differs
// from the reference's Ooura FFT at the 1e-16 relative level and nothing here is bit-exact.
#include "wc_oracle.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <map>
#include <numeric>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

typedef std::complex<double> cd;
typedef std::vector<double> vd;

// reference include/world_constantnumbers.hpp:12-41
const double kPi = 3.1415926535897932384;
const double kSafe = 0.000000000001;
const double kEps = 0.00000000000000022204460492503131;
const double kDefaultF0 = 500.0;
const double kLog2 = 0.69314718055994529;
const double kFrequencyInterval = 3000.0;
const double kUpperLimit = 15000.0;
const double kFloorF0D4C = 47.0;

int g_threads = 0;
int g_only_pulse = -1;  // debugging aid: synthesise only this pulse (-1 = all)

// ------------------------------------------------------------------------------------------
// RNG: xorshift128, one shift-only step then 12 full steps per draw
// (reference src/world_matlabfunctions.cpp:243-264)
// ------------------------------------------------------------------------------------------
struct Rng {
	uint32_t x, y, z, w;
	uint64_t pos;
	void reset() { x = 123456789u; y = 362436069u; z = 521288629u; w = 88675123u; pos = 0; }
	uint32_t raw() {
		uint32_t t;
		t = x ^ (x << 11); x = y; y = z; z = w;  // first step never updates w
		(void)t;
		uint32_t tmp = 0;
		for (int i = 0; i < 12; ++i) {
			t = x ^ (x << 11); x = y; y = z; z = w;
			w = (w ^ (w >> 19)) ^ (t ^ (t >> 8));
			tmp += w >> 4;
		}
		++pos;
		return tmp;
	}
	double randn() { return raw() / 268435456.0 - 6.0; }
};

// GF(2) jump-ahead: the per-draw state transition is linear in the 128 state bits.
struct BitMat { uint32_t col[128][4]; };  // col[j] = image of basis vector e_j
static void state_to_words(const Rng &r, uint32_t s[4]) { s[0] = r.x; s[1] = r.y; s[2] = r.z; s[3] = r.w; }
static void matvec(const BitMat &m, const uint32_t s[4], uint32_t o[4]) {
	o[0] = o[1] = o[2] = o[3] = 0;
	for (int j = 0; j < 128; ++j)
		if ((s[j >> 5] >> (j & 31)) & 1u)
			for (int k = 0; k < 4; ++k) o[k] ^= m.col[j][k];
}
static void matmul(const BitMat &a, const BitMat &b, BitMat &c) {  // c = a*b
	for (int j = 0; j < 128; ++j) matvec(a, b.col[j], c.col[j]);
}
static std::vector<BitMat> build_jump_table() {
	std::vector<BitMat> tab(48);
	for (int j = 0; j < 128; ++j) {
		Rng r; r.pos = 0;
		uint32_t s[4] = {0, 0, 0, 0};
		s[j >> 5] = 1u << (j & 31);
		r.x = s[0]; r.y = s[1]; r.z = s[2]; r.w = s[3];
		r.raw();
		state_to_words(r, tab[0].col[j]);
	}
	for (int k = 1; k < 48; ++k) matmul(tab[k - 1], tab[k - 1], tab[k]);
	return tab;
}
static std::vector<BitMat> &jump_table() {
	static std::vector<BitMat> tab = build_jump_table();  // thread-safe magic static
	return tab;
}
static void rng_seek(Rng &r, uint64_t position) {
	std::vector<BitMat> &tab = jump_table();
	r.reset();
	uint32_t s[4], o[4];
	state_to_words(r, s);
	for (int k = 0; k < 48; ++k)
		if ((position >> k) & 1ull) { matvec(tab[k], s, o); std::memcpy(s, o, sizeof(s)); }
	r.x = s[0]; r.y = s[1]; r.z = s[2]; r.w = s[3];
	r.pos = position;
}

Rng g_rng = {123456789u, 362436069u, 521288629u, 88675123u, 0};

// ------------------------------------------------------------------------------------------
// FFT with the reference's conventions (reference src/world_fft.cpp:31-77 wrappers over Ooura):
//   r2c : X[k] = sum x[n] e^{+2 pi i k n / N}, k = 0..N/2
//   c2r : y[n] = sum over the Hermitian extension of Y[k] e^{-2 pi i k n / N} (imag of bins 0, N/2 ignored)
//   c2c : forward e^{+i}, backward e^{-i}, all unnormalised
// ------------------------------------------------------------------------------------------
struct Plan {
	int n;
	std::vector<cd> w;    // w[k] = e^{+2 pi i k / n}, k < n/2 (n >= 2)
	std::vector<int> rev;
};
static const Plan &plan(int n) {
	static thread_local std::map<int, Plan> cache;
	std::map<int, Plan>::iterator it = cache.find(n);
	if (it != cache.end()) return it->second;
	Plan &p = cache[n];
	p.n = n;
	p.w.resize(std::max(1, n / 2));
	for (int k = 0; k < n / 2; ++k) p.w[k] = cd(std::cos(2.0 * kPi * k / n), std::sin(2.0 * kPi * k / n));
	p.rev.assign(n, 0);
	int bits = 0;
	while ((1 << bits) < n) ++bits;
	for (int i = 0; i < n; ++i) {
		int r = 0;
		for (int b = 0; b < bits; ++b) if (i & (1 << b)) r |= 1 << (bits - 1 - b);
		p.rev[i] = r;
	}
	return p;
}
static void cfft(cd *a, int n, int sign) {  // in place, sign=+1: e^{+i}
	if (n == 1) return;
	const Plan &p = plan(n);
	for (int i = 0; i < n; ++i) if (i < p.rev[i]) std::swap(a[i], a[p.rev[i]]);
	for (int len = 2; len <= n; len <<= 1) {
		int half = len >> 1, step = n / len;
		for (int i = 0; i < n; i += len)
			for (int k = 0; k < half; ++k) {
				cd w = p.w[k * step];
				if (sign < 0) w = std::conj(w);
				cd u = a[i + k], v = a[i + k + half] * w;
				a[i + k] = u + v;
				a[i + k + half] = u - v;
			}
	}
}
static void r2c(const double *x, int n, cd *X) {
	int m = n / 2;
	std::vector<cd> z(m);
	for (int k = 0; k < m; ++k) z[k] = cd(x[2 * k], x[2 * k + 1]);
	cfft(z.data(), m, +1);
	const Plan &p = plan(n);
	for (int k = 0; k <= m; ++k) {
		cd zk = z[k % m], zc = std::conj(z[(m - k) % m]);
		cd e = 0.5 * (zk + zc);
		cd o = (zk - zc) * cd(0.0, -0.5);
		cd w = (k < m) ? p.w[k] : cd(-1.0, 0.0);
		X[k] = e + w * o;
	}
	X[0] = cd(X[0].real(), 0.0);
	X[m] = cd(X[m].real(), 0.0);
}
static void c2r(const cd *Y, int n, double *out) {
	int m = n / 2;
	std::vector<cd> z(m);
	const Plan &p = plan(n);
	for (int k = 0; k < m; ++k) {
		cd yk = (k == 0) ? cd(Y[0].real(), 0.0) : Y[k];
		cd ym = (k == 0) ? cd(Y[m].real(), 0.0) : Y[m - k];
		cd e = yk + std::conj(ym);
		cd o = (yk - std::conj(ym)) * std::conj(p.w[k]);
		z[k] = e + cd(0.0, 1.0) * o;
	}
	cfft(z.data(), m, -1);
	for (int k = 0; k < m; ++k) { out[2 * k] = z[k].real(); out[2 * k + 1] = z[k].imag(); }
}

// ------------------------------------------------------------------------------------------
// MATLAB-compatible helpers (reference src/world_matlabfunctions.cpp)
// ------------------------------------------------------------------------------------------
inline int mround(double x) { return x > 0 ? static_cast<int>(x + 0.5) : static_cast<int>(x - 0.5); }  // :212-214

// :136-155 -- for non-decreasing edges: index = clamp(#{j : x[j] <= edge}, 1, n-1)
static void histc(const double *x, int n, const double *edges, int m, int *index) {
	int c = 1;
	for (int i = 0; i < m; ++i) {
		while (c < n && edges[i] >= x[c]) ++c;
		index[i] = std::min(c, n - 1);
	}
}
// :157-182 linear interpolation with linear extrapolation outside
static void interp1(const double *x, const double *y, int n, const double *xi, int m, double *yi) {
	std::vector<int> k(m);
	histc(x, n, xi, m, k.data());
	for (int i = 0; i < m; ++i) {
		int j = k[i];
		double h = x[j] - x[j - 1];
		double s = (xi[i] - x[j - 1]) / h;
		yi[i] = y[j - 1] + s * (y[j] - y[j - 1]);
	}
}
// :220-241 equally spaced abscissa, truncation toward zero, last difference = 0
static inline double interp1Q_one(double x0, double dx, const double *y, int n, double xi) {
	int b = static_cast<int>((xi - x0) / dx);
	double frac = (xi - x0) / dx - b;
	double dy = (b == n - 1) ? 0.0 : y[b + 1] - y[b];
	return y[b] + dy * frac;
}
// :27-125 order-3 IIR used by decimate
static void filter_for_decimate(const double *x, int n, int r, double *y) {
	static const double A[13][3] = {
		{0, 0, 0}, {0, 0, 0},
		{0.041156734567757189, -0.42599112459189636, 0.041037215479961225},
		{0.95039378983237421, -0.67429146741526791, 0.15412211621346475},
		{1.4499664446880227, -0.98943497080950582, 0.24578252340690215},
		{1.7610939654280557, -1.2554914843859768, 0.3237186507788215},
		{1.9715352749512141, -1.4686795689225347, 0.3893908434965701},
		{2.1225239019534703, -1.6395144861046302, 0.44469707800587366},
		{2.2357462340187593, -1.7780899984041358, 0.49152555365968692},
		{2.3236003491759578, -1.8921545617463598, 0.53148928133729068},
		{2.3936475118069387, -1.9873904075111861, 0.5658879979027055},
		{2.450743295230728, -2.06794904601978, 0.59574774438332101},
		{2.4981398605924205, -2.1368928194784025, 0.62187513816221485}};
	static const double B[13][2] = {
		{0, 0}, {0, 0},
		{0.16797464681802227, 0.50392394045406674},
		{0.071221945171178636, 0.21366583551353591},
		{0.036710750339322612, 0.11013225101796784},
		{0.021334858522387423, 0.06400457556716227},
		{0.013469181309343825, 0.040407543928031475},
		{0.0090366882681608418, 0.027110064804482525},
		{0.0063522763407111993, 0.019056829022133598},
		{0.0046331164041389372, 0.013899349212416812},
		{0.0034818622251927556, 0.010445586675578267},
		{0.0026822508007163792, 0.0080467524021491377},
		{0.0021097275904709001, 0.0063291827714127002}};
	int idx = (r >= 2 && r <= 12) ? r : 0;
	const double *a = A[idx], *b = B[idx];
	double w0 = 0, w1 = 0, w2 = 0;
	for (int i = 0; i < n; ++i) {
		double wt = x[i] + a[0] * w0 + a[1] * w1 + a[2] * w2;
		y[i] = b[0] * wt + b[1] * w0 + b[1] * w1 + b[0] * w2;
		w2 = w1; w1 = w0; w0 = wt;
	}
}
// :184-210
static void decimate(const double *x, int n, int r, double *y) {
	const int kNFact = 9;
	int len = n + 2 * kNFact;
	vd t1(len), t2(len);
	for (int i = 0; i < kNFact; ++i) t1[i] = 2 * x[0] - x[kNFact - i];
	for (int i = 0; i < n; ++i) t1[kNFact + i] = x[i];
	for (int i = 0; i < kNFact; ++i) t1[kNFact + n + i] = 2 * x[n - 1] - x[n - 2 - i];
	filter_for_decimate(t1.data(), len, r, t2.data());
	for (int i = 0; i < len; ++i) t1[i] = t2[len - i - 1];
	filter_for_decimate(t1.data(), len, r, t2.data());
	for (int i = 0; i < len; ++i) t1[i] = t2[len - i - 1];
	int nout = n / r + 1;
	int nbeg = r - r * nout + n;
	int count = 0;
	for (int i = nbeg; i < n + kNFact; i += r) y[count++] = t1[i + kNFact - 1];
}

// ------------------------------------------------------------------------------------------
// shared DSP helpers (reference src/world_common.cpp)
// ------------------------------------------------------------------------------------------
static int suitable_fft_size(int sample) {  // :56-59
	return static_cast<int>(std::pow(2.0, static_cast<int>(std::log(static_cast<double>(sample)) / kLog2) + 1.0));
}
// :61-80 (in place: every reference caller passes output == input)
static void dc_correction(double *p, double f0, int fs, int fft_size) {
	int upper = 2 + static_cast<int>(f0 * fft_size / fs);
	vd rep(upper - 1);
	for (int i = 0; i < upper - 1; ++i) {
		double axis = static_cast<double>(i) * fs / fft_size;
		rep[i] = interp1Q_one(f0, -static_cast<double>(fs) / fft_size, p, upper + 1, axis);
	}
	for (int i = 0; i < upper - 1; ++i) p[i] += rep[i];
}
// :27-52, :82-116
static void linear_smoothing(const double *in, double width, int fs, int fft_size, double *out) {
	int b = static_cast<int>(width * fft_size / fs) + 1;
	int half = fft_size / 2;
	int len = half + 2 * b + 1;
	vd mir(len), seg(len);
	for (int i = 0; i < b; ++i) mir[i] = in[b - i];
	for (int i = b; i < half + b; ++i) mir[i] = in[i - b];
	for (int i = half + b; i <= half + 2 * b; ++i) mir[i] = in[half - (i - (half + b))];
	seg[0] = mir[0] * fs / fft_size;
	for (int i = 1; i < len; ++i) seg[i] = mir[i] * fs / fft_size + seg[i - 1];
	double origin = -(b - 0.5) * fs / fft_size;
	double step = static_cast<double>(fs) / fft_size;
	vd res(half + 1);
	for (int i = 0; i <= half; ++i) {
		double lo_axis = static_cast<double>(i) / fft_size * fs - width / 2.0;
		double hi_axis = lo_axis + width;
		double lo = interp1Q_one(origin, step, seg.data(), len, lo_axis);
		double hi = interp1Q_one(origin, step, seg.data(), len, hi_axis);
		res[i] = (hi - lo) / width;
	}
	std::copy(res.begin(), res.end(), out);
}
static void nuttall(int n, double *y) {  // :118-126
	for (int i = 0; i < n; ++i) {
		double t = i / (n - 1.0);
		y[i] = 0.355768 - 0.487396 * std::cos(2.0 * kPi * t) + 0.144232 * std::cos(4.0 * kPi * t) -
			   0.012604 * std::cos(6.0 * kPi * t);
	}
}
// :196-233; log_spectrum holds bins 0..n/2 on entry, result = bins 0..n/2
static void minimum_phase(int n, const double *log_spectrum, cd *out) {
	vd ls(n);
	for (int i = 0; i <= n / 2; ++i) ls[i] = log_spectrum[i];
	for (int i = n / 2 + 1; i < n; ++i) ls[i] = ls[n - i];
	std::vector<cd> cep(n);
	r2c(ls.data(), n, cep.data());
	cep[0] = cd(cep[0].real(), -cep[0].imag());
	for (int i = 1; i < n / 2; ++i) cep[i] = cd(cep[i].real() * 2.0, cep[i].imag() * -2.0);
	cep[n / 2] = cd(cep[n / 2].real(), -cep[n / 2].imag());
	for (int i = n / 2 + 1; i < n; ++i) cep[i] = cd(0.0, 0.0);
	cfft(cep.data(), n, +1);
	for (int i = 0; i <= n / 2; ++i) {
		double t = std::exp(cep[i].real() / n);
		out[i] = cd(t * std::cos(cep[i].imag() / n), t * std::sin(cep[i].imag() / n));
	}
}

// F0-adaptive window gather shared by CheapTrick and D4C:
// origin = matlab_round(t*fs + 0.001), samples clamped into [0, x_length-1]
// (reference src/cheaptrick.cpp:169-196, src/d4c.cpp:246-303)
static inline int clampi(int v, int lo, int hi) { return std::max(lo, std::min(hi, v)); }

// ------------------------------------------------------------------------------------------
// CheapTrick (reference src/cheaptrick.cpp)
// ------------------------------------------------------------------------------------------
static int ct_fft_size(int fs, double f0_floor) {  // :97-100
	return static_cast<int>(std::pow(2.0, 1.0 + static_cast<int>(std::log(3.0 * fs / f0_floor + 1) / kLog2)));
}
static double ct_f0_floor(int fs, int fft_size) { return 3 * fs / (fft_size - 3.0); }  // :102-105

static void ct_frame(const double *x, int x_length, int fs, double f0, double pos, int N, double q1,
					 Rng &rng, double *envelope) {
	int hw = mround(1.5 * fs / f0);
	int wl = 2 * hw + 1;
	vd wave(N, 0.0), win(wl);
	// window (:169-196)
	int origin = mround(pos * fs + 0.001);
	double avg = 0.0;
	for (int i = 0; i < wl; ++i) {
		double position = (i - hw) / 1.5 / fs;
		win[i] = 0.5 * std::cos(kPi * position * f0) + 0.5;
		avg += win[i] * win[i];
	}
	avg = std::sqrt(avg);
	for (int i = 0; i < wl; ++i) win[i] /= avg;
	// windowing + infinitesimal noise + weighted-mean removal (:137-167)
	for (int i = 0; i < wl; ++i) {
		int si = clampi(origin + i - hw, 0, x_length - 1);
		wave[i] = x[si] * win[i] + rng.randn() * 0.000000000000001;
	}
	double tw1 = 0, tw2 = 0;
	for (int i = 0; i < wl; ++i) { tw1 += wave[i]; tw2 += win[i]; }
	double wc = tw1 / tw2;
	for (int i = 0; i < wl; ++i) wave[i] -= win[i] * wc;
	// power spectrum + DC correction (:198-218)
	std::vector<cd> spec(N / 2 + 1);
	r2c(wave.data(), N, spec.data());
	vd p(N, 0.0);
	for (int i = 0; i <= N / 2; ++i) p[i] = spec[i].real() * spec[i].real() + spec[i].imag() * spec[i].imag();
	dc_correction(p.data(), f0, fs, N);
	// linear smoothing, width 2 f0 / 3 (:122-123)
	linear_smoothing(p.data(), f0 * 2.0 / 3.0, fs, N, p.data());
	// infinitesimal noise (:220-228)
	for (int i = 0; i <= N / 2; ++i) p[i] += std::fabs(rng.randn()) * kEps;
	// smoothing with recovery (:230-276)
	for (int i = 0; i <= N / 2; ++i) p[i] = std::log(p[i]);
	for (int i = 1; i < N / 2; ++i) p[N - i] = p[i];
	r2c(p.data(), N, spec.data());
	for (int i = 0; i <= N / 2; ++i) {
		double sl, cl;
		if (i == 0) {
			sl = 1.0;
			cl = (1.0 - 2.0 * q1) + 2.0 * q1;
		} else {
			double q = static_cast<double>(i) / fs;
			sl = std::sin(kPi * f0 * q) / (kPi * f0 * q);
			cl = (1.0 - 2.0 * q1) + 2.0 * q1 * std::cos(2.0 * kPi * q * f0);
		}
		spec[i] = cd(spec[i].real() * sl * cl / N, 0.0);
	}
	vd w2(N);
	c2r(spec.data(), N, w2.data());
	for (int i = 0; i <= N / 2; ++i) envelope[i] = std::exp(w2[i]);
}

static inline uint64_t ct_frame_draws(int fs, double f0c, int N) {
	return static_cast<uint64_t>(2 * mround(1.5 * fs / f0c) + 1) + static_cast<uint64_t>(N / 2 + 1);
}

static void cheaptrick(const double *x, int x_length, int fs, const double *tpos, const double *f0,
					   int L, double q1, double f0_floor_opt, int fft_size, double *sp) {
	int N = fft_size ? fft_size : ct_fft_size(fs, f0_floor_opt);
	double floor_ = ct_f0_floor(fs, N);
	int bins = N / 2 + 1;
	std::vector<uint64_t> start(L + 1);
	start[0] = g_rng.pos;
	std::vector<double> f0c(L);
	for (int i = 0; i < L; ++i) {
		f0c[i] = (f0[i] <= floor_) ? kDefaultF0 : f0[i];  // :77
		start[i + 1] = start[i] + ct_frame_draws(fs, f0c[i], N);
	}
	if (g_threads <= 0) {
		for (int i = 0; i < L; ++i)
			ct_frame(x, x_length, fs, f0c[i], tpos[i], N, q1, g_rng, sp + static_cast<size_t>(i) * bins);
	} else {
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 8)
		for (int i = 0; i < L; ++i) {
			Rng r;
			rng_seek(r, start[i]);
			ct_frame(x, x_length, fs, f0c[i], tpos[i], N, q1, r, sp + static_cast<size_t>(i) * bins);
		}
		rng_seek(g_rng, start[L]);
	}
}

// ------------------------------------------------------------------------------------------
// D4C (reference src/d4c.cpp)
// ------------------------------------------------------------------------------------------
struct D4CSetup {
	int fs, N, n_ap, wl, N_lt;
	vd window, coarse_axis;
};
static D4CSetup d4c_setup(int fs) {  // :60-111
	D4CSetup s;
	s.fs = fs;
	s.N = static_cast<int>(std::pow(2.0, 1.0 + static_cast<int>(std::log(4.0 * fs / kFloorF0D4C + 1) / kLog2)));
	s.n_ap = static_cast<int>(std::min(kUpperLimit, fs / 2.0 - kFrequencyInterval) / kFrequencyInterval);
	s.wl = static_cast<int>(kFrequencyInterval * s.N / fs) * 2 + 1;
	s.window.resize(s.wl);
	nuttall(s.wl, s.window.data());
	s.coarse_axis.resize(s.n_ap + 2);
	for (int i = 0; i <= s.n_ap; ++i) s.coarse_axis[i] = i * kFrequencyInterval;
	s.coarse_axis[s.n_ap + 1] = fs / 2.0;
	s.N_lt = static_cast<int>(std::pow(2.0, 1.0 + static_cast<int>(std::log(3.0 * fs / 40.0 + 1) / kLog2)));
	return s;
}
// :246-303; window_type 1 = Hanning, 2 = Blackman.  Returns the window length.
static int d4c_windowed(const double *x, int x_length, int fs, double f0, double pos, int type,
						double ratio, Rng &rng, double *wave) {
	int hw = mround(ratio * fs / f0 / 2.0);
	int wl = 2 * hw + 1;
	vd win(wl);
	int origin = mround(pos * fs + 0.001);
	double c1 = 2.0 / ratio / fs;
	double c2 = kPi * f0;
	for (int i = 0; i < wl; ++i) {
		double position = c1 * (i - hw);
		if (type == 1) win[i] = 0.5 * std::cos(c2 * position) + 0.5;
		else win[i] = 0.42 + 0.5 * std::cos(c2 * position) + 0.08 * std::cos(c2 * position * 2);
	}
	for (int i = 0; i < wl; ++i) {
		int si = clampi(origin + i - hw, 0, x_length - 1);
		wave[i] = x[si] * win[i] + rng.randn() * kSafe;
	}
	double tw1 = 0, tw2 = 0;
	for (int i = 0; i < wl; ++i) tw1 += wave[i];
	for (int i = 0; i < wl; ++i) tw2 += win[i];
	double wc = tw1 / tw2;
	for (int i = 0; i < wl; ++i) wave[i] -= win[i] * wc;
	return wl;
}
// :209-240
static double love_train_one(const double *x, int x_length, const D4CSetup &s, double f0, double pos,
							 Rng &rng) {
	int N = s.N_lt, fs = s.fs;
	int b0 = static_cast<int>(std::ceil(100.0 * N / fs));
	int b1 = static_cast<int>(std::ceil(4000.0 * N / fs));
	int b2 = static_cast<int>(std::ceil(7900.0 * N / fs));
	vd wave(N, 0.0);
	d4c_windowed(x, x_length, fs, f0, pos, 2, 3.0, rng, wave.data());
	std::vector<cd> spec(N / 2 + 1);
	r2c(wave.data(), N, spec.data());
	vd p(N, 0.0);
	for (int i = b0 + 1; i < N / 2 + 1; ++i) p[i] = std::norm(spec[i]);
	for (int i = b0; i <= b2; ++i) p[i] += p[i - 1];
	return p[b1] / p[b2];
}
static inline uint64_t lt_draws(int fs, double f0) { return 2 * mround(3.0 * fs / std::max(f0, 40.0) / 2.0) + 1; }
static inline uint64_t d4c_frame_draws(int fs, double f0) {
	return 3ull * (2 * mround(4.0 * fs / std::max(kFloorF0D4C, f0) / 2.0) + 1);
}
// :366-405
static void d4c_centroid(const double *x, int x_length, const D4CSetup &s, double f0, double pos,
						 Rng &rng, double *centroid) {
	int N = s.N;
	vd wave(N, 0.0);
	int wl = d4c_windowed(x, x_length, s.fs, f0, pos, 2, 4.0, rng, wave.data());
	double power = 0.0;
	for (int i = 0; i < wl; ++i) power += wave[i] * wave[i];
	power = std::sqrt(power);
	for (int i = 0; i < wl; ++i) wave[i] /= power;
	std::vector<cd> s1(N / 2 + 1), s2(N / 2 + 1);
	r2c(wave.data(), N, s1.data());
	for (int i = 0; i < N; ++i) wave[i] *= i + 1.0;
	r2c(wave.data(), N, s2.data());
	for (int i = 0; i <= N / 2; ++i) centroid[i] = s2[i].real() * s1[i].real() + s1[i].imag() * s2[i].imag();
}
// :308-333 with :339-360, :411-434, :440-460, :466-503
static void d4c_frame(const double *x, int x_length, const D4CSetup &s, double f0, double pos, Rng &rng,
					  double *coarse /* n_ap */) {
	int N = s.N, fs = s.fs, bins = N / 2 + 1;
	vd c1(bins), c2(bins), sc(bins), sps(N, 0.0), sgd(bins), sm(bins);
	d4c_centroid(x, x_length, s, f0, pos - 0.25 / f0, rng, c1.data());
	d4c_centroid(x, x_length, s, f0, pos + 0.25 / f0, rng, c2.data());
	for (int i = 0; i < bins; ++i) sc[i] = c1[i] + c2[i];
	dc_correction(sc.data(), f0, fs, N);
	// smoothed power spectrum
	{
		vd wave(N, 0.0);
		d4c_windowed(x, x_length, fs, f0, pos, 1, 4.0, rng, wave.data());
		std::vector<cd> sp(bins);
		r2c(wave.data(), N, sp.data());
		for (int i = 0; i < bins; ++i) sps[i] = sp[i].real() * sp[i].real() + sp[i].imag() * sp[i].imag();
		dc_correction(sps.data(), f0, fs, N);
		linear_smoothing(sps.data(), f0, fs, N, sps.data());
	}
	// static group delay
	for (int i = 0; i < bins; ++i) sgd[i] = sc[i] / sps[i];
	linear_smoothing(sgd.data(), f0 / 2.0, fs, N, sgd.data());
	linear_smoothing(sgd.data(), f0, fs, N, sm.data());
	for (int i = 0; i < bins; ++i) sgd[i] -= sm[i];
	// coarse aperiodicity
	int boundary = mround(N * 8.0 / s.wl);
	int hwl = s.wl / 2;
	vd wave(N, 0.0), ps(bins);
	std::vector<cd> sp(bins);
	for (int b = 0; b < s.n_ap; ++b) {
		int center = static_cast<int>(kFrequencyInterval * (b + 1) * N / fs);
		for (int j = 0; j <= hwl * 2; ++j) wave[j] = sgd[center - hwl + j] * s.window[j];
		r2c(wave.data(), N, sp.data());
		for (int j = 0; j < bins; ++j) ps[j] = sp[j].real() * sp[j].real() + sp[j].imag() * sp[j].imag();
		std::sort(ps.begin(), ps.end());
		for (int j = 1; j < bins; ++j) ps[j] += ps[j - 1];
		coarse[b] = 10 * std::log10(ps[bins - boundary - 2] / ps[bins - 1]);
	}
	double cv = (f0 - 100) / 50.0;
	for (int b = 0; b < s.n_ap; ++b) coarse[b] = std::min(0.0, coarse[b] + cv);
}
// :113-173
static void d4c(const double *x, int x_length, int fs, const double *tpos, const double *f0, int L,
				int fft_size, double threshold, double *ap) {
	D4CSetup s = d4c_setup(fs);
	int bins = fft_size / 2 + 1;
	const double init_val = 1.0 - kSafe;
	for (size_t i = 0; i < static_cast<size_t>(L) * bins; ++i) ap[i] = init_val;
	// LoveTrain over all frames first (:181-207)
	vd ap0(L, 0.0);
	std::vector<uint64_t> st(L + 1);
	st[0] = g_rng.pos;
	for (int i = 0; i < L; ++i) st[i + 1] = st[i] + (f0[i] == 0.0 ? 0 : lt_draws(fs, f0[i]));
	if (g_threads <= 0) {
		for (int i = 0; i < L; ++i)
			if (f0[i] != 0.0) ap0[i] = love_train_one(x, x_length, s, std::max(f0[i], 40.0), tpos[i], g_rng);
	} else {
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 8)
		for (int i = 0; i < L; ++i) {
			if (f0[i] == 0.0) continue;
			Rng r;
			rng_seek(r, st[i]);
			ap0[i] = love_train_one(x, x_length, s, std::max(f0[i], 40.0), tpos[i], r);
		}
		rng_seek(g_rng, st[L]);
	}
	vd axis(bins);
	for (int i = 0; i < bins; ++i) axis[i] = static_cast<double>(i) * fs / fft_size;
	std::vector<uint64_t> st2(L + 1);
	st2[0] = g_rng.pos;
	for (int i = 0; i < L; ++i) {
		bool on = !(f0[i] == 0 || ap0[i] <= threshold);
		st2[i + 1] = st2[i] + (on ? d4c_frame_draws(fs, f0[i]) : 0);
	}
#pragma omp parallel for num_threads(g_threads > 0 ? g_threads : 1) schedule(dynamic, 4) if (g_threads > 0)
	for (int i = 0; i < L; ++i) {
		if (f0[i] == 0 || ap0[i] <= threshold) continue;
		Rng local;
		Rng *r = &g_rng;
		if (g_threads > 0) { rng_seek(local, st2[i]); r = &local; }
		vd coarse(s.n_ap + 2);
		coarse[0] = -60.0;
		coarse[s.n_ap + 1] = -kSafe;
		d4c_frame(x, x_length, s, std::max(kFloorF0D4C, f0[i]), tpos[i], *r, coarse.data() + 1);
		double *row = ap + static_cast<size_t>(i) * bins;
		interp1(s.coarse_axis.data(), coarse.data(), s.n_ap + 2, axis.data(), bins, row);
		for (int j = 0; j < bins; ++j) row[j] = std::pow(10.0, row[j] / 20.0);
	}
	if (g_threads > 0) rng_seek(g_rng, st2[L]);
}

// ------------------------------------------------------------------------------------------
// Synthesis (reference src/synthesis.cpp)
// ------------------------------------------------------------------------------------------
struct Pulses {
	std::vector<double> time, shift;
	std::vector<int> index;
	vd vuv;
};
// :180-288
static void synth_time_base(const double *f0, int L, int fs, double fp, int out_len, double lowest_f0,
							Pulses &P) {
	vd time_axis(out_len), ct(L + 1), cf(L + 1), cv(L + 1), if0(out_len);
	P.vuv.assign(out_len, 0.0);
	for (int i = 0; i < out_len; ++i) time_axis[i] = i / static_cast<double>(fs);
	for (int i = 0; i < L; ++i) {
		ct[i] = i * fp;
		cf[i] = (f0[i] < lowest_f0) ? 0.0 : f0[i];
		cv[i] = (cf[i] == 0.0) ? 0.0 : 1.0;
	}
	ct[L] = L * fp;
	cf[L] = cf[L - 1] * 2 - cf[L - 2];
	cv[L] = cv[L - 1] * 2 - cv[L - 2];
	interp1(ct.data(), cf.data(), L + 1, time_axis.data(), out_len, if0.data());
	interp1(ct.data(), cv.data(), L + 1, time_axis.data(), out_len, P.vuv.data());
	for (int i = 0; i < out_len; ++i) {
		P.vuv[i] = P.vuv[i] > 0.5 ? 1.0 : 0.0;
		if0[i] = P.vuv[i] == 0.0 ? kDefaultF0 : if0[i];
	}
	double two_pi = 2.0 * kPi, cval = two_pi / fs;
	vd wrap(out_len);
	double total = if0[0] * cval;
	wrap[0] = std::fmod(total, two_pi);
	for (int i = 1; i < out_len; ++i) {
		total = total + if0[i] * cval;
		wrap[i] = std::fmod(total, two_pi);
	}
	for (int i = 0; i < out_len - 1; ++i) {
		if (std::fabs(wrap[i + 1] - wrap[i]) > kPi) {
			P.time.push_back(time_axis[i]);
			P.index.push_back(i);
			double y1 = wrap[i] - two_pi, y2 = wrap[i + 1];
			double xx = -y1 / (y2 - y1);
			P.shift.push_back(xx / fs);
		}
	}
}
static void dc_remover(int N, double *d) {  // :290-303
	double cval = 2.0 * kPi / (1.0 + N);
	for (int i = 0; i < N / 2; ++i) d[i] = 0.5 - 0.5 * std::cos(cval * (i + 1.0));
	double dc = 0.0;
	for (int i = 0; i < N / 2; ++i) dc += d[i];
	dc *= 2;
	for (int i = 0; i < N / 2; ++i) { d[i] /= dc; d[N - i - 1] = d[i]; }
}
static inline double safe_ap(double v) { return std::max(0.001, std::min(0.999999999999, v)); }
// :308-344 with :346-393, :403-474, :479-530
static void synth_pulse(const double *sp, const double *ap, int L, int N, int fs, double fp, double vuv,
						int noise_size, double t, double shift, const double *dcr, Rng &rng,
						double *response) {
	int bins = N / 2 + 1;
	int fl = std::min(L - 1, static_cast<int>(std::floor(t / fp)));
	int ce = std::min(L - 1, static_cast<int>(std::ceil(t / fp)));
	double a = t / fp - fl;
	vd env(bins), ar(bins);
	const double *sf = sp + static_cast<size_t>(fl) * bins, *sc = sp + static_cast<size_t>(ce) * bins;
	const double *af = ap + static_cast<size_t>(fl) * bins, *ac = ap + static_cast<size_t>(ce) * bins;
	if (fl == ce) {
		for (int i = 0; i < bins; ++i) env[i] = std::fabs(sf[i]);
		for (int i = 0; i < bins; ++i) ar[i] = std::pow(safe_ap(af[i]), 2.0);
	} else {
		for (int i = 0; i < bins; ++i) env[i] = (1.0 - a) * std::fabs(sf[i]) + a * std::fabs(sc[i]);
		for (int i = 0; i < bins; ++i) ar[i] = std::pow((1.0 - a) * safe_ap(af[i]) + a * safe_ap(ac[i]), 2.0);
	}
	vd periodic(N, 0.0), aperiodic(N), ls(bins), tmp(N);
	std::vector<cd> mp(bins), spec(bins);
	// periodic response (:403-437)
	if (!(vuv <= 0.5 || ar[0] > 0.999)) {
		for (int i = 0; i < bins; ++i) ls[i] = std::log(env[i] * (1.0 - ar[i]) + kSafe) / 2.0;
		minimum_phase(N, ls.data(), mp.data());
		double coef = 2.0 * kPi * shift * fs / N;
		for (int i = 0; i < bins; ++i) {  // :443-457
			double re = mp[i].real(), im = mp[i].imag();
			double re2 = std::cos(coef * i);
			double im2 = std::sqrt(1.0 - re2 * re2);
			spec[i] = cd(re * re2 - im * im2, re * im2 + im * re2);
		}
		c2r(spec.data(), N, tmp.data());
		for (int i = 0; i < N / 2; ++i) { periodic[i] = tmp[i + N / 2]; periodic[i + N / 2] = tmp[i]; }
		double dc = 0.0;  // :459-474 (first half overwritten, dc_remover[ii] reused for the second half)
		for (int i = N / 2; i < N; ++i) dc += periodic[i];
		for (int i = 0; i < N / 2; ++i) {
			double v = -dc * dcr[i];
			periodic[i] = v;
			periodic[i + N / 2] += v;
		}
	}
	// aperiodic response (:479-530)
	vd noise(N, 0.0);
	for (int i = 0; i < noise_size; ++i) noise[i] = rng.randn();
	double avg = 0.0;
	for (int i = 0; i < noise_size; ++i) avg += noise[i];
	avg /= noise_size;
	for (int i = 0; i < noise_size; ++i) noise[i] -= avg;
	std::vector<cd> nspec(bins);
	r2c(noise.data(), N, nspec.data());
	if (vuv != 0.0) for (int i = 0; i < bins; ++i) ls[i] = std::log(env[i] * ar[i]) / 2.0;
	else for (int i = 0; i < bins; ++i) ls[i] = std::log(env[i]) / 2.0;
	minimum_phase(N, ls.data(), mp.data());
	for (int i = 0; i < bins; ++i)
		spec[i] = cd(mp[i].real() * nspec[i].real() - mp[i].imag() * nspec[i].imag(),
					 mp[i].real() * nspec[i].imag() + mp[i].imag() * nspec[i].real());
	c2r(spec.data(), N, tmp.data());
	for (int i = 0; i < N / 2; ++i) { aperiodic[i] = tmp[i + N / 2]; aperiodic[i + N / 2] = tmp[i]; }
	double sq = std::sqrt(static_cast<double>(noise_size));
	for (int i = 0; i < N; ++i) response[i] = (periodic[i] * sq + aperiodic[i]) / N;
}
// :77-177
static void synthesis(const double *f0, int L, const double *sp, const double *ap, int N, int fs,
					  double fp_ms, int out_len, double *out) {
	double fp = fp_ms / 1000.;
	for (int i = 0; i < out_len; ++i) out[i] = 0;
	Pulses P;
	synth_time_base(f0, L, fs, fp, out_len, fs / N + 1.0, P);  // integer division fs/N (:97)
	int np = static_cast<int>(P.index.size());
	vd dcr(N);
	dc_remover(N, dcr.data());
	std::vector<uint64_t> st(np + 1);
	st[0] = g_rng.pos;
	std::vector<int> ns(np);
	for (int i = 0; i < np; ++i) {
		ns[i] = P.index[std::min(np - 1, i + 1)] - P.index[i];
		st[i + 1] = st[i] + ns[i];
	}
	std::vector<double> resp;
	if (g_threads > 0) resp.resize(static_cast<size_t>(N) * np);
	else resp.resize(N);
#pragma omp parallel for num_threads(g_threads > 0 ? g_threads : 1) schedule(dynamic, 8) if (g_threads > 0)
	for (int i = 0; i < np; ++i) {
		Rng local;
		Rng *r = &g_rng;
		if (g_threads > 0) { rng_seek(local, st[i]); r = &local; }
		double *rp = g_threads > 0 ? resp.data() + static_cast<size_t>(N) * i : resp.data();
		synth_pulse(sp, ap, L, N, fs, fp, P.vuv[P.index[i]], ns[i], P.time[i], P.shift[i], dcr.data(), *r, rp);
		if (g_only_pulse >= 0 && i != g_only_pulse) continue;
		if (g_threads > 0) continue;
		int index = P.index[i] - N / 2;  // overlap-add (:156-168)
		if (index + N < 0 || index + 1 >= out_len) continue;
		int b = (index + 1 < 0) ? std::abs(index + 1) : 0;
		int e = (index + N >= out_len) ? out_len - index - 1 : N;
		for (int j = b; j < e; ++j) out[index + 1 + j] += rp[j];
	}
	if (g_threads > 0) {
		for (int i = 0; i < np; ++i) {  // serial overlap-add (:118-139)
			const double *rp = resp.data() + static_cast<size_t>(N) * i;
			int index = P.index[i] - N / 2;
			if (index + N < 0 || index + 1 >= out_len) continue;
			int b = (index + 1 < 0) ? std::abs(index + 1) : 0;
			int e = (index + N >= out_len) ? out_len - index - 1 : N;
			for (int j = b; j < e; ++j) out[index + 1 + j] += rp[j];
		}
		rng_seek(g_rng, st[np]);
	}
}

// ------------------------------------------------------------------------------------------
// Harvest (reference src/harvest.cpp)
// ------------------------------------------------------------------------------------------
// HarvestOption fields beyond floor / ceil / frame period (reference include/harvest.hpp:16-24, defaults src/harvest.cpp:52-56)
static double g_hv_target_fs = 8000.0, g_hv_channels_in_octave = 40.0;
static int g_hv_use_cos_table = 0;
// get_cos_table (:152-170): 8001 entries over one period built from the first quarter
static const vd &hv_cos_table() {
	static vd t;
	if (t.empty()) {
		const int n = 2000;
		t.assign(n * 4 + 1, 0.0);
		double interval = kPi / 2. / n;
		for (int i = 0; i < n + 1; ++i) t[i] = std::cos(interval * i);
		for (int i = 0; i < n; ++i) t[i + n + 1] = -t[n - 1 - i];
		for (int i = 0; i < n; ++i) t[i + n * 2 + 1] = -t[i + 1];
		for (int i = 0; i < n; ++i) t[i + n * 3 + 1] = t[n - 1 - i];
	}
	return t;
}
struct Harvest {
	int fs, decim, y_length, L, n_bands, max_cand, n_cand, fft_size;
	double fs_d, floor_, ceil_;
	vd y, tpos, bands;
	std::vector<vd> raw;         // [band][L]
	std::vector<vd> cand, score;  // [L][max_cand]
};
static int get_samples(int fs, int x_length, double fp) {  // :173-181
	return static_cast<int>(1000.0 * x_length / fs / fp) + 1;
}
// :213-248
static void hv_waveform(Harvest &H, const double *x, int x_length, std::vector<cd> &Y) {
	H.y.assign(H.fft_size, 0.0);
	if (H.decim == 1) {
		for (int i = 0; i < x_length; ++i) H.y[i] = x[i];
	} else {
		int lag = static_cast<int>(std::ceil(140.0 / H.decim) * H.decim);
		int nn = x_length + lag * 2;
		vd nx(nn), ny(nn, 0.0);
		for (int i = 0; i < lag; ++i) nx[i] = x[0];
		for (int i = 0; i < x_length; ++i) nx[lag + i] = x[i];
		for (int i = lag + x_length; i < nn; ++i) nx[i] = x[x_length - 1];
		decimate(nx.data(), nn, H.decim, ny.data());
		for (int i = 0; i < H.y_length; ++i) H.y[i] = ny[lag / H.decim + i];
	}
	// "DC removal" with an int-typed accumulator (:239): acc = int(acc + y[i]) at every step
	int acc = 0;
	for (int i = 0; i < H.y_length; ++i) acc = static_cast<int>(acc + H.y[i]);
	double mean_y = acc;
	mean_y /= H.y_length;
	for (int i = 0; i < H.y_length; ++i) H.y[i] -= mean_y;
	Y.resize(H.fft_size / 2 + 1);
	r2c(H.y.data(), H.fft_size, Y.data());
}
// :1179-1219; returns number of intervals
static int zero_crossing_engine(const double *s, int n, double fs, double *loc, double *itv) {
	std::vector<int> edges;
	for (int i = 0; i < n - 1; ++i)
		if (0.0 < s[i] && s[i + 1] <= 0.0) edges.push_back(i + 1);
	int count = static_cast<int>(edges.size());
	if (count < 2) return 0;
	vd fine(count);
	for (int i = 0; i < count; ++i) fine[i] = edges[i] - s[edges[i] - 1] / (s[edges[i]] - s[edges[i] - 1]);
	for (int i = 0; i < count - 1; ++i) {
		itv[i] = fs / (fine[i + 1] - fine[i]);
		loc[i] = (fine[i] + fine[i + 1]) / 2.0 / fs;
	}
	return count - 1;
}
// one band: :1261-1305 (filter), :1228-1255 (4 zero-crossing sets), :1098-1143 (contour)
static void hv_band(const Harvest &H, const std::vector<cd> &Y, double fb, double *out) {
	int N = H.fft_size;
	int hl = mround(H.fs_d / fb * 2.0);
	vd bp(N, 0.0);
	nuttall(hl * 2 + 1, bp.data());
	for (int i = -hl; i <= hl; ++i) bp[i + hl] *= std::cos(2 * kPi * fb * i / H.fs_d);
	std::vector<cd> B(N / 2 + 1);
	r2c(bp.data(), N, B.data());
	for (int i = 0; i <= N / 2; ++i) B[i] = cd(Y[i].real() * B[i].real() - Y[i].imag() * B[i].imag(),
											   Y[i].real() * B[i].imag() + Y[i].imag() * B[i].real());
	vd fsig(N);
	c2r(B.data(), N, fsig.data());
	std::rotate(fsig.begin(), fsig.begin() + hl + 1, fsig.end());
	int yl = H.y_length;
	vd loc[4], itv[4];
	int cnt[4];
	for (int k = 0; k < 4; ++k) { loc[k].resize(yl); itv[k].resize(yl); }
	cnt[0] = zero_crossing_engine(fsig.data(), yl, H.fs_d, loc[0].data(), itv[0].data());
	for (int i = 0; i < yl; ++i) fsig[i] *= -1;
	cnt[1] = zero_crossing_engine(fsig.data(), yl, H.fs_d, loc[1].data(), itv[1].data());
	for (int i = 0; i < yl - 1; ++i) fsig[i] -= fsig[i + 1];
	cnt[2] = zero_crossing_engine(fsig.data(), yl - 1, H.fs_d, loc[2].data(), itv[2].data());
	for (int i = 0; i < yl - 1; ++i) fsig[i] *= -1;
	cnt[3] = zero_crossing_engine(fsig.data(), yl - 1, H.fs_d, loc[3].data(), itv[3].data());
	int L = H.L;
	if (!(cnt[0] > 2 && cnt[1] > 2 && cnt[2] > 2 && cnt[3] > 2)) {
		for (int i = 0; i < L; ++i) out[i] = 0.0;
		return;
	}
	vd ip[4];
	for (int k = 0; k < 4; ++k) {
		ip[k].resize(L);
		interp1(loc[k].data(), itv[k].data(), cnt[k], H.tpos.data(), L, ip[k].data());
	}
	double upper = fb * 1.1, lower = fb * 0.9;
	for (int i = 0; i < L; ++i) {
		double v = (ip[0][i] + ip[1][i] + ip[2][i] + ip[3][i]) / 4.0;
		if (v > upper || v < lower || v > H.ceil_ || v < H.floor_) v = 0.0;
		out[i] = v;
	}
}
// :1052-1083 with Sub1 :1032-1046 and Sub2 :1005-1027; returns the max count over frames
static int hv_detect(Harvest &H) {
	int nb = H.n_bands, best = 0;
	std::vector<int> vuv(nb), st(nb), ed(nb);
	for (int i = 0; i < H.L; ++i) {
		for (int j = 0; j < nb; ++j) vuv[j] = H.raw[j][i] > 0 ? 1 : 0;
		vuv[0] = vuv[nb - 1] = 0;
		int ns = 0;
		for (int j = 1; j < nb; ++j) {
			int d = vuv[j] - vuv[j - 1];
			if (d == 1) st[ns] = j;
			if (d == -1) ed[ns++] = j;
		}
		int nc = 0;
		for (int s = 0; s < ns; ++s) {
			if (ed[s] - st[s] < 10) continue;
			double t = 0.0;
			for (int j = st[s]; j < ed[s]; ++j) t += H.raw[j][i];
			t /= (ed[s] - st[s]);
			H.cand[i][nc++] = t;
		}
		for (int j = nc; j < H.max_cand; ++j) H.cand[i][j] = 0.0;
		best = std::max(best, nc);
	}
	return best;
}
// :987-1000
static void hv_overlap(Harvest &H, int nc) {
	int n = 3, L = H.L;
	for (int i = 1; i <= n; ++i)
		for (int j = 0; j < nc; ++j) {
			for (int k = i; k < L; ++k) H.cand[k][j + nc * i] = H.cand[k - i][j];
			for (int k = 0; k < L - i; ++k) H.cand[k][j + nc * (i + n)] = H.cand[k + i][j];
		}
}
// :883-927 with :750-878
static void hv_refine_one(const Harvest &H, double pos, double f, double *rf, double *rs) {
	double fs = H.fs_d;
	int hw = static_cast<int>(1.5 * fs / f + 1.0);
	double wlt = (2.0 * hw + 1.0) / fs;
	int bt = hw * 2 + 1;
	int fft_index = 2 + static_cast<int>(std::log(hw * 2 + 1.0) / kLog2);
	int N = static_cast<int>(std::pow(2.0, fft_index));
	double bt0 = (-hw) / fs;
	int basic = mround((pos + bt0) * fs + 0.001);
	vd mw(bt), dw(bt), wave(N, 0.0);
	if (!g_hv_use_cos_table) {
		for (int i = 0; i < bt; ++i) {
			double tmp = (basic + i - 1.0) / fs - pos;
			double tmp2 = 2.0 * kPi * tmp / wlt;
			mw[i] = 0.42 + 0.5 * std::cos(tmp2) + 0.08 * std::cos(2 * tmp2);
		}
	} else {  // :779-787
		const vd &tab = hv_cos_table();
		const double two_pi = 2.0 * kPi;
		const int num_div = 2000 * 4;
		for (int i = 0; i < bt; ++i) {
			double tmp = (basic + i - 1.0) / fs - pos;
			double tmp2 = two_pi * (tmp / wlt + 1);
			double dindex = std::fmod(tmp2, two_pi) / two_pi * num_div;
			double dindex2 = std::fmod(dindex * 2, num_div);
			mw[i] = 0.42 + 0.5 * tab[static_cast<int>(std::round(dindex))] + 0.08 * tab[static_cast<int>(std::round(dindex2))];
		}
	}
	dw[0] = -mw[1] / 2.0;
	dw[bt - 1] = mw[bt - 2] / 2.0;
	for (int i = 1; i < bt - 1; ++i) dw[i] = -(mw[i + 1] - mw[i - 1]) / 2.0;
	std::vector<cd> ms(N / 2 + 1), ds(N / 2 + 1);
	for (int i = 0; i < bt; ++i) wave[i] = mw[i] * H.y[clampi(basic + i - 1, 0, H.y_length - 1)];
	r2c(wave.data(), N, ms.data());
	for (int i = 0; i < bt; ++i) wave[i] = dw[i] * H.y[clampi(basic + i - 1, 0, H.y_length - 1)];
	r2c(wave.data(), N, ds.data());
	int nh = std::min(static_cast<int>(fs / 2.0 / f), 6);
	double num = 0, den = 0, sc = 0;
	for (int i = 0; i < nh; ++i) {
		int idx = mround(f * N / fs * (i + 1));
		cd m = std::conj(ms[idx]), d = std::conj(ds[idx]);  // explicit sign flip (:831,:840)
		double pw = m.real() * m.real() + m.imag() * m.imag();
		double ni = m.real() * d.imag() - m.imag() * d.real();
		double inst = (pw == 0.0) ? 0.0 : static_cast<double>(idx) * fs / N + ni / pw * fs / 2.0 / kPi;
		double amp = std::sqrt(pw);
		num += amp * inst;
		den += amp * (i + 1.0);
		sc += std::fabs((inst / (i + 1.0) - f) / f);
	}
	*rf = num / (den + kSafe);
	*rs = 1.0 / (sc / nh + kSafe);
}
// :932-982
static void hv_refine(Harvest &H) {
#pragma omp parallel for num_threads(g_threads > 0 ? g_threads : 1) schedule(dynamic, 16) if (g_threads > 0)
	for (int i = 0; i < H.L; ++i)
		for (int j = 0; j < H.n_cand; ++j) {
			double f = H.cand[i][j];
			if (f <= 0.0) { H.cand[i][j] = 0.0; H.score[i][j] = 0.0; continue; }
			double rf, rs;
			hv_refine_one(H, H.tpos[i], f, &rf, &rs);
			if (rf < H.floor_ || rf > H.ceil_ || rs < 2.5) { rf = 0.0; rs = 0.0; }
			H.cand[i][j] = rf;
			H.score[i][j] = rs;
		}
}
// :347-365
static double select_best_f0(double ref, const double *c, int n, double allowed, double &best_error) {
	double best = 0.0;
	best_error = allowed;
	for (int i = 0; i < n; ++i) {
		double t = std::fabs(ref - c[i]) / ref;
		if (t > best_error) continue;
		best = c[i];
		best_error = t;
	}
	return best;
}
// :708-744
static void hv_remove_unreliable(Harvest &H) {
	// the reference copies frames 1 .. L-2 only (:714-715); rows 0 and L-1 of its copy are never written
	// (uninitialised read; zero with the zero-filling operator new[] of ref_shim.cpp, restated as zero)
	std::vector<vd> tmp(H.cand);
	if (H.L > 0) {
		std::fill(tmp[0].begin(), tmp[0].end(), 0.0);
		std::fill(tmp[H.L - 1].begin(), tmp[H.L - 1].end(), 0.0);
	}
	for (int i = 1; i < H.L - 1; ++i)
		for (int j = 0; j < H.n_cand; ++j) {
			double ref = H.cand[i][j];
			if (ref == 0) continue;
			double e1, e2;
			select_best_f0(ref, tmp[i + 1].data(), H.n_cand, 1.0, e1);
			select_best_f0(ref, tmp[i - 1].data(), H.n_cand, 1.0, e2);
			if (std::min(e1, e2) <= 0.05) continue;
			H.cand[i][j] = 0;
			H.score[i][j] = 0;
		}
}
// :296-314
static int boundary_list(const double *f0, int n, std::vector<int> &bl) {
	bl.assign(n, 0);
	std::vector<int> vuv(n);
	vuv[0] = vuv[n - 1] = 0;
	for (int i = 1; i < n - 1; ++i) vuv[i] = f0[i] > 0 ? 1 : 0;
	int nb = 0;
	for (int i = 1; i < n; ++i)
		if (vuv[i] - vuv[i - 1] != 0) { bl[nb] = i - nb % 2; nb++; }
	return nb;
}
// :371-403
static int extend_f0(const Harvest &H, double *ext, int origin, int last_point, int shift, double allowed) {
	int threshold = 4;
	double tmp_f0 = ext[origin];
	int shifted_origin = origin;
	int distance = std::abs(last_point - origin);
	int count = 0;
	double dummy;
	for (int i = 0; i <= distance; ++i) {
		int idx = origin + shift * i + shift;
		ext[idx] = select_best_f0(tmp_f0, H.cand[idx].data(), H.n_cand, allowed, dummy);
		if (ext[idx] == 0.0) {
			count++;
		} else {
			tmp_f0 = ext[idx];
			count = 0;
			shifted_origin = idx;
		}
		if (count == threshold) break;
	}
	return shifted_origin;
}
static double search_score(double f0, const double *c, const double *s, int n) {  // :463-470
	double score = 0.0;
	for (int i = 0; i < n; ++i)
		if (f0 == c[i] && score < s[i]) score = s[i];
	return score;
}
// :475-497
static int merge_f0_sub(const Harvest &H, double *merged, int st1, int ed1, const double *f2, int st2, int ed2) {
	if (st1 <= st2 && ed1 >= ed2) return ed1;
	double s1 = 0.0, s2 = 0.0;
	for (int i = st2; i <= ed1; ++i) {
		s1 += search_score(merged[i], H.cand[i].data(), H.score[i].data(), H.n_cand);
		s2 += search_score(f2[i], H.cand[i].data(), H.score[i].data(), H.n_cand);
	}
	if (s1 > s2) std::copy(f2 + ed1, f2 + ed2 + 1, merged + ed1);
	else std::copy(f2 + st2, f2 + ed2 + 1, merged + st2);
	return ed2;
}
// fixStep1..4 (:277-291, :319-334, :560-585 incl. :427-458 and :502-536, :590-614)
static void hv_fix_contour(Harvest &H, vd &base_out, vd &best) {
	int L = H.L;
	vd base(L, 0.0), s1(L, 0.0), s2, s3, s4;
	for (int i = 0; i < L; ++i) {  // searchF0Base :254-272
		double bs = 0.0;
		for (int j = 0; j < H.n_cand; ++j)
			if (H.score[i][j] > bs) { base[i] = H.cand[i][j]; bs = H.score[i][j]; }
	}
	base_out = base;
	// step 1: entries with f0_base == 0 are never written by the reference (uninitialised read);
	// restated as 0 (upstream WORLD semantics; ref_shim.cpp zero-fills new[] for the same reason)
	for (int i = 2; i < L; ++i) {
		if (base[i] == 0.0) continue;
		double ref = base[i - 1] * 2 - base[i - 2];
		s1[i] = (std::fabs((base[i] - ref) / ref) > 0.008 &&
				 std::fabs((base[i] - base[i - 1])) / base[i - 1] > 0.008) ? 0.0 : base[i];
	}
	// step 2
	s2 = s1;
	std::vector<int> bl;
	int nb = boundary_list(s1.data(), L, bl);
	for (int i = 0; i < nb / 2; ++i) {
		if (bl[i * 2 + 1] - bl[i * 2] >= 6) continue;
		for (int j = bl[i * 2]; j <= bl[i * 2 + 1]; ++j) s2[j] = 0.0;
	}
	// step 3
	s3 = s2;
	nb = boundary_list(s2.data(), L, bl);
	int ns = nb / 2;
	std::vector<vd> mc(ns, vd(L, 0.0));
	std::vector<int> chan(ns);  // channel permutation (the reference swaps row pointers)
	for (int i = 0; i < ns; ++i) {
		chan[i] = i;
		for (int j = bl[i * 2]; j <= bl[i * 2 + 1]; ++j) mc[i][j] = s2[j];
	}
	// extend (:427-458)
	for (int i = 0; i < ns; ++i) {
		bl[i * 2 + 1] = extend_f0(H, mc[i].data(), bl[i * 2 + 1], std::min(L - 2, bl[i * 2 + 1] + 100), 1, 0.18);
		bl[i * 2] = extend_f0(H, mc[i].data(), bl[i * 2], std::max(1, bl[i * 2] - 100), -1, 0.18);
	}
	int count = 0;
	double mean_f0 = 0.0;  // not reset between sections (:446-452) -- kept
	for (int i = 0; i < ns; ++i) {
		int st = bl[i * 2], ed = bl[i * 2 + 1];
		for (int j = st; j < ed; ++j) mean_f0 += mc[chan[i]][j];
		mean_f0 /= ed - st;
		if (2200.0 / mean_f0 < ed - st) {
			std::swap(chan[count], chan[i]);
			std::swap(bl[count * 2], bl[i * 2]);
			std::swap(bl[count * 2 + 1], bl[i * 2 + 1]);
			count++;
		}
	}
	// merge (:502-536)
	// (with zero selected channels the reference still copies the row at position 0)
	if (ns > 0) {
		std::vector<int> order(count);
		for (int i = 0; i < count; ++i) order[i] = i;
		std::sort(order.begin(), order.end(), [&](int a, int b) { return bl[a * 2] < bl[b * 2]; });
		s3 = mc[chan[0]];
		for (int i = 1; i < count; ++i) {
			int i1 = bl[order[i] * 2], i2 = bl[order[i] * 2 + 1];
			const vd &src = mc[chan[order[i]]];
			if (i1 - bl[1] > 0) {
				std::copy(src.begin() + i1, src.begin() + i2 + 1, s3.begin() + i1);
				bl[0] = i1;
				bl[1] = i2;
			} else {
				bl[1] = merge_f0_sub(H, s3.data(), bl[0], bl[1], src.data(), i1, i2);
			}
		}
	}
	// step 4
	s4 = s3;
	nb = boundary_list(s3.data(), L, bl);
	for (int i = 0; i < nb / 2 - 1; ++i) {
		int distance = bl[(i + 1) * 2] - bl[i * 2 + 1] - 1;
		if (distance >= 9) continue;
		double t0 = s3[bl[i * 2 + 1]] + 1;
		double t1 = s3[bl[(i + 1) * 2]] - 1;
		double coef = (t1 - t0) / (distance + 1.0);
		int c = 1;
		for (int j = bl[i * 2 + 1] + 1; j <= bl[(i + 1) * 2] - 1; ++j) s4[j] = t0 + coef * c++;
	}
	best = s4;
}
// :670-703 with :639-665
static void hv_smooth(const Harvest &H, const vd &f0, double *smoothed) {
	const double b[2] = {0.0078202080334971724, 0.015640416066994345};
	const double a[2] = {1.7347257688092754, -0.76600660094326412};
	int lag = 300, L = H.L, n = L + lag * 2;
	vd contour(n, 0.0);
	std::copy(f0.begin(), f0.end(), contour.begin() + lag);
	std::vector<int> bl;
	int nb = boundary_list(contour.data(), n, bl);
	std::vector<vd> mc(nb / 2, vd(n, 0.0));
	for (int i = 0; i < nb / 2; ++i)
		for (int j = bl[i * 2]; j <= bl[i * 2 + 1]; ++j) mc[i][j] = contour[j];
	vd tmp(n), yv(n);
	for (int i = 0; i < nb / 2; ++i) {
		vd &xx = mc[i];
		int st = bl[i * 2], ed = bl[i * 2 + 1];
		for (int k = 0; k < st; ++k) xx[k] = xx[st];
		for (int k = ed + 1; k < n; ++k) xx[k] = xx[ed];
		double w0 = 0, w1 = 0;
		for (int k = 0; k < n; ++k) {
			double wt = xx[k] + a[0] * w0 + a[1] * w1;
			tmp[n - k - 1] = b[0] * wt + b[1] * w0 + b[0] * w1;
			w1 = w0; w0 = wt;
		}
		w0 = w1 = 0;
		for (int k = 0; k < n; ++k) {
			double wt = tmp[k] + a[0] * w0 + a[1] * w1;
			yv[n - k - 1] = b[0] * wt + b[1] * w0 + b[0] * w1;
			w1 = w0; w0 = wt;
		}
		for (int j = st; j <= ed; ++j) smoothed[j - lag] = yv[j];
	}
}
// :1380-1453 at frame_period 1 ms
static void hv_general_body(Harvest &H, const double *x, int x_length, int fs, double floor_, double ceil_,
							vd &f0_base, vd &f0_fixed, vd &f0_1ms) {
	const double target_fs = g_hv_target_fs, cio = g_hv_channels_in_octave;
	H.fs = fs;
	H.floor_ = floor_;
	H.ceil_ = ceil_;
	H.decim = std::max(std::min(mround(fs / target_fs), 12), 1);  // :82-84
	H.fs_d = static_cast<double>(fs) / H.decim;
	double adj_floor = floor_ * 0.9, adj_ceil = ceil_ * 1.1;
	H.n_bands = 1 + static_cast<int>(std::log(adj_ceil / adj_floor) / kLog2 * cio);
	H.bands.resize(H.n_bands);
	for (int i = 0; i < H.n_bands; ++i) H.bands[i] = adj_floor * std::pow(2.0, static_cast<double>(i + 1) / cio);
	H.y_length = 1 + static_cast<int>(x_length / H.decim);
	H.fft_size = suitable_fft_size(H.y_length + (4 * static_cast<int>(1.0 + H.fs_d / H.bands[0] / 2.0)));
	std::vector<cd> Y;
	hv_waveform(H, x, x_length, Y);
	H.L = get_samples(fs, x_length, 1);
	H.tpos.resize(H.L);
	for (int i = 0; i < H.L; ++i) H.tpos[i] = i * 1 / 1000.0;
	H.max_cand = mround(H.n_bands / 10) * 7;
	H.cand.assign(H.L, vd(H.max_cand, 0.0));
	H.score.assign(H.L, vd(H.max_cand, 0.0));
	H.raw.assign(H.n_bands, vd(H.L));
#pragma omp parallel for num_threads(g_threads > 0 ? g_threads : 1) schedule(dynamic, 1) if (g_threads > 0)
	for (int b = 0; b < H.n_bands; ++b) hv_band(H, Y, H.bands[b], H.raw[b].data());
	int nc = hv_detect(H);
	hv_overlap(H, nc);
	H.n_cand = nc * 7;
	hv_refine(H);
	hv_remove_unreliable(H);
	hv_fix_contour(H, f0_base, f0_fixed);
	f0_1ms.assign(H.L, 0.0);
	hv_smooth(H, f0_fixed, f0_1ms.data());
}
// :183-208
static void harvest(const double *x, int x_length, int fs, double floor_, double ceil_, double fp,
					double *tpos, double *f0) {
	Harvest H;
	vd base, fixed, f1;
	hv_general_body(H, x, x_length, fs, floor_, ceil_, base, fixed, f1);
	if (fp == 1.0) {
		for (int i = 0; i < H.L; ++i) { tpos[i] = H.tpos[i]; f0[i] = f1[i]; }
		return;
	}
	int L = get_samples(fs, x_length, fp);
	for (int i = 0; i < L; ++i) {
		tpos[i] = i * fp / 1000.0;
		f0[i] = f1[std::min(H.L - 1, mround(tpos[i] * 1000.0))];
	}
}

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" {
void wco_rng_reset(void) { g_rng.reset(); }
void wco_rng_seek(uint64_t position) { rng_seek(g_rng, position); }
uint64_t wco_rng_position(void) { return g_rng.pos; }
double wco_randn(void) { return g_rng.randn(); }
void wco_randn_fill(int n, double *out) { for (int i = 0; i < n; ++i) out[i] = g_rng.randn(); }
int wco_matlab_round(double x) { return mround(x); }
int wco_suitable_fft_size(int sample) { return suitable_fft_size(sample); }
void wco_interp1(const double *x, const double *y, int n, const double *xi, int m, double *yi) { interp1(x, y, n, xi, m, yi); }
void wco_interp1Q(double x0, double dx, const double *y, int n, const double *xi, int m, double *yi) {
	for (int i = 0; i < m; ++i) yi[i] = interp1Q_one(x0, dx, y, n, xi[i]);
}
void wco_histc(const double *x, int n, const double *edges, int m, int *index) { histc(x, n, edges, m, index); }
void wco_decimate(const double *x, int n, int r, double *y) { decimate(x, n, r, y); }
void wco_dc_correction(const double *in, double f0, int fs, int fft_size, double *out) {
	int upper = 2 + static_cast<int>(f0 * fft_size / fs);
	vd tmp(in, in + std::max(fft_size / 2 + 1, upper + 1));
	dc_correction(tmp.data(), f0, fs, fft_size);
	std::copy(tmp.begin(), tmp.begin() + fft_size / 2 + 1, out);
}
void wco_linear_smoothing(const double *in, double width, int fs, int fft_size, double *out) { linear_smoothing(in, width, fs, fft_size, out); }
void wco_nuttall(int n, double *y) { nuttall(n, y); }
void wco_fft_r2c(int n, const double *in, double *out) {
	std::vector<cd> X(n / 2 + 1);
	r2c(in, n, X.data());
	for (int i = 0; i <= n / 2; ++i) { out[2 * i] = X[i].real(); out[2 * i + 1] = X[i].imag(); }
}
void wco_fft_c2r(int n, const double *in, double *out) {
	std::vector<cd> Y(n / 2 + 1);
	for (int i = 0; i <= n / 2; ++i) Y[i] = cd(in[2 * i], in[2 * i + 1]);
	c2r(Y.data(), n, out);
}
void wco_fft_c2c(int n, int sign, const double *in, double *out) {
	std::vector<cd> a(n);
	for (int i = 0; i < n; ++i) a[i] = cd(in[2 * i], in[2 * i + 1]);
	cfft(a.data(), n, sign == 1 ? +1 : -1);
	for (int i = 0; i < n; ++i) { out[2 * i] = a[i].real(); out[2 * i + 1] = a[i].imag(); }
}
void wco_minimum_phase(int n, const double *log_spectrum, double *out) {
	std::vector<cd> m(n / 2 + 1);
	minimum_phase(n, log_spectrum, m.data());
	for (int i = 0; i <= n / 2; ++i) { out[2 * i] = m[i].real(); out[2 * i + 1] = m[i].imag(); }
}
void wco_set_threads(int threads) { g_threads = threads; }
void wco_debug_only_pulse(int p) { g_only_pulse = p; }
int wco_get_samples(int fs, int x_length, double frame_period) { return get_samples(fs, x_length, frame_period); }
void wco_harvest(const double *x, int x_length, int fs, double f0_floor, double f0_ceil, double frame_period,
				 double *tpos, double *f0) { harvest(x, x_length, fs, f0_floor, f0_ceil, frame_period, tpos, f0); }
int wco_cheaptrick_fft_size(int fs, double f0_floor) { return ct_fft_size(fs, f0_floor); }
double wco_cheaptrick_f0_floor(int fs, int fft_size) { return ct_f0_floor(fs, fft_size); }
void wco_cheaptrick(const double *x, int x_length, int fs, const double *tpos, const double *f0, int f0_length,
					double q1, double f0_floor, int fft_size, double *sp) {
	cheaptrick(x, x_length, fs, tpos, f0, f0_length, q1, f0_floor, fft_size, sp);
}
void wco_d4c(const double *x, int x_length, int fs, const double *tpos, const double *f0, int f0_length,
			 int fft_size, double threshold, double *ap) {
	d4c(x, x_length, fs, tpos, f0, f0_length, fft_size, threshold, ap);
}
void wco_synthesis(const double *f0, int f0_length, const double *sp, const double *ap, int fft_size, int fs,
				   double frame_period_ms, int out_length, double *out) {
	synthesis(f0, f0_length, sp, ap, fft_size, fs, frame_period_ms, out_length, out);
}
// number of pulses the time base produces, and the capacity the reference allocates for them
// (reference src/synthesis.cpp:85-93: out_length / int(fs / max_f0)); count > capacity is a heap
// overflow in the reference.
int wco_synthesis_pulses(const double *f0, int f0_length, int fft_size, int fs, double frame_period_ms,
						 int out_length, int *reference_capacity) {
	Pulses P;
	synth_time_base(f0, f0_length, fs, frame_period_ms / 1000., out_length, fs / fft_size + 1.0, P);
	double max_f0 = *std::max_element(f0, f0 + f0_length);
	if (reference_capacity) *reference_capacity = max_f0 > 0 ? out_length / static_cast<int>(fs / max_f0) : 0;
	return static_cast<int>(P.index.size());
}
uint64_t wco_cheaptrick_draws(int fs, const double *f0, int f0_length, double f0_floor, int fft_size) {
	int N = fft_size ? fft_size : ct_fft_size(fs, f0_floor);
	double floor_ = ct_f0_floor(fs, N);
	uint64_t t = 0;
	for (int i = 0; i < f0_length; ++i) t += ct_frame_draws(fs, (f0[i] <= floor_) ? kDefaultF0 : f0[i], N);
	return t;
}
void wco_set_harvest_options(double target_fs, double channels_in_octave, int use_cos_table) {
	g_hv_target_fs = target_fs;
	g_hv_channels_in_octave = channels_in_octave;
	g_hv_use_cos_table = use_cos_table;
}
int wco_harvest_debug(const double *x, int x_length, int fs, double f0_floor, double f0_ceil, int *dims,
					  double *y, double *raw, double *cand, double *score, double *f0_base, double *f0_fixed,
					  double *f0_1ms) {
	Harvest H;
	vd base, fixed, f1;
	hv_general_body(H, x, x_length, fs, f0_floor, f0_ceil, base, fixed, f1);
	if (dims) { dims[0] = H.y_length; dims[1] = H.n_bands; dims[2] = H.max_cand; dims[3] = H.n_cand; }
	if (y) std::copy(H.y.begin(), H.y.begin() + H.y_length, y);
	if (raw) for (int b = 0; b < H.n_bands; ++b) std::copy(H.raw[b].begin(), H.raw[b].end(), raw + static_cast<size_t>(b) * H.L);
	if (cand) for (int i = 0; i < H.L; ++i) std::copy(H.cand[i].begin(), H.cand[i].end(), cand + static_cast<size_t>(i) * H.max_cand);
	if (score) for (int i = 0; i < H.L; ++i) std::copy(H.score[i].begin(), H.score[i].end(), score + static_cast<size_t>(i) * H.max_cand);
	if (f0_base) std::copy(base.begin(), base.end(), f0_base);
	if (f0_fixed) std::copy(fixed.begin(), fixed.end(), f0_fixed);
	if (f0_1ms) std::copy(f1.begin(), f1.end(), f0_1ms);
	return H.L;
}
}  // extern "C"
