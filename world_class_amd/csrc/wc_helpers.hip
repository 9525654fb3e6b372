// Host implementations of the reference's free helper functions (include/world_matlabfunctions.hpp) and of its FFT plan API
// (include/world_fft.hpp): restates reference src/world_matlabfunctions.cpp:27-241, :303-313, src/world_common.cpp:27-126 and the
// conventions of src/world_fft.cpp:31-167 (with a plain radix-2 transform instead of the bundled split-radix one).  They exist so that callers
// of those helpers link against this library unchanged; the kernels do not use them.
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/world_fft.hpp"
#include "../../include/world_matlabfunctions.hpp"

namespace {

// FilterForDecimate, reference src/world_matlabfunctions.cpp:27-125: order-3 IIR, coefficient sets for r = 2..12
void iir3(const double *x, int n, int r, double *y) {
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
	const int idx = (r >= 2 && r <= 12) ? r : 0;  // (the reference leaves the coefficients at zero for other ratios)
	const double *a = A[idx], *b = B[idx];
	double w0 = 0, w1 = 0, w2 = 0;
	for (int i = 0; i < n; ++i) {
		const double wt = x[i] + a[0] * w0 + a[1] * w1 + a[2] * w2;
		y[i] = b[0] * wt + b[1] * w0 + b[1] * w1 + b[0] * w2;
		w2 = w1;
		w1 = w0;
		w0 = wt;
	}
}

inline double interp1q_at(double x0, double dx, const double *y, int n, double xi) {
	const int base = static_cast<int>((xi - x0) / dx);
	const double frac = (xi - x0) / dx - base;
	const double dy = (base == n - 1) ? 0.0 : y[base + 1] - y[base];
	return y[base] + dy * frac;
}

// In-place radix-2 transform of n complex numbers (n a power of two) with e^{sign * 2 pi i k m / n}; tw holds
// cos / sin of 2 pi j / n for j < n / 2.
void cfft_pow2(double *a, int n, int sign, const double *tw) {
	for (int i = 1, j = 0; i < n; ++i) {  // bit reversal
		int bit = n >> 1;
		for (; j & bit; bit >>= 1) j ^= bit;
		j ^= bit;
		if (i < j) {
			std::swap(a[2 * i], a[2 * j]);
			std::swap(a[2 * i + 1], a[2 * j + 1]);
		}
	}
	for (int len = 2; len <= n; len <<= 1) {
		const int half = len >> 1, step = n / len;
		for (int s0 = 0; s0 < n; s0 += len) {
			for (int k = 0; k < half; ++k) {
				const double wr = tw[2 * (k * step)], wi = sign * tw[2 * (k * step) + 1];
				double *u = a + 2 * (s0 + k), *v = a + 2 * (s0 + k + half);
				const double tr = v[0] * wr - v[1] * wi, ti = v[0] * wi + v[1] * wr;
				v[0] = u[0] - tr;
				v[1] = u[1] - ti;
				u[0] += tr;
				u[1] += ti;
			}
		}
	}
}

fft_plan make_plan(int n, int sign, unsigned int flags) {
	fft_plan p;
	p.n = n;
	p.sign = sign;
	p.flags = flags;
	p.c_in = nullptr; p.in = nullptr; p.c_out = nullptr; p.out = nullptr;
	p.input = new double[2 * static_cast<size_t>(n > 0 ? n : 1)];
	p.ip = new int[1];
	p.ip[0] = 0;
	p.w = new double[static_cast<size_t>(n > 1 ? n : 2)];
	const double pi = 3.1415926535897932384;
	for (int j = 0; j < n / 2; ++j) {
		p.w[2 * j] = std::cos(2.0 * pi * j / n);
		p.w[2 * j + 1] = std::sin(2.0 * pi * j / n);
	}
	return p;
}

}  // namespace

extern "C" {

fft_plan fft_plan_dft_1d(int n, fft_complex *in, fft_complex *out, int sign, unsigned int flags) {
	fft_plan p = make_plan(n, sign, flags);
	p.c_in = in;
	p.c_out = out;
	return p;
}
fft_plan fft_plan_dft_c2r_1d(int n, fft_complex *in, double *out, unsigned int flags) {
	fft_plan p = make_plan(n, FFT_BACKWARD, flags);
	p.c_in = in;
	p.out = out;
	return p;
}
fft_plan fft_plan_dft_r2c_1d(int n, double *in, fft_complex *out, unsigned int flags) {
	fft_plan p = make_plan(n, FFT_FORWARD, flags);
	p.in = in;
	p.c_out = out;
	return p;
}

// reference src/world_fft.cpp:31-77, :151-157
void fft_execute(fft_plan p) {
	const int n = p.n;
	double *a = p.input;
	if (p.sign == FFT_FORWARD && p.c_in == nullptr) {  // r2c
		for (int i = 0; i < n; ++i) { a[2 * i] = p.in[i]; a[2 * i + 1] = 0.0; }
		cfft_pow2(a, n, +1, p.w);
		for (int i = 0; i <= n / 2; ++i) { p.c_out[i][0] = a[2 * i]; p.c_out[i][1] = a[2 * i + 1]; }
		p.c_out[0][1] = 0.0;
		p.c_out[n / 2][1] = 0.0;
	} else if (p.sign != FFT_FORWARD && p.c_out == nullptr) {  // c2r: Hermitian extension, bins 0 and n/2 real
		a[0] = p.c_in[0][0]; a[1] = 0.0;
		a[2 * (n / 2)] = p.c_in[n / 2][0]; a[2 * (n / 2) + 1] = 0.0;
		for (int i = 1; i < n / 2; ++i) {
			a[2 * i] = p.c_in[i][0]; a[2 * i + 1] = p.c_in[i][1];
			a[2 * (n - i)] = p.c_in[i][0]; a[2 * (n - i) + 1] = -p.c_in[i][1];
		}
		cfft_pow2(a, n, -1, p.w);
		for (int i = 0; i < n; ++i) p.out[i] = a[2 * i];
	} else {  // c2c
		for (int i = 0; i < n; ++i) { a[2 * i] = p.c_in[i][0]; a[2 * i + 1] = p.c_in[i][1]; }
		cfft_pow2(a, n, p.sign == FFT_FORWARD ? +1 : -1, p.w);
		for (int i = 0; i < n; ++i) { p.c_out[i][0] = a[2 * i]; p.c_out[i][1] = a[2 * i + 1]; }
	}
}

void fft_destroy_plan(fft_plan p) {
	delete[] p.input;
	delete[] p.ip;
	delete[] p.w;
}

void fftshift(const double *x, int x_length, double *y) {
	const int h = x_length / 2;
	for (int i = 0; i < h; ++i) {
		y[i] = x[i + h];
		y[i + h] = x[i];
	}
}

// MATLAB's histc as the reference uses it (interp1's segment search, reference src/world_matlabfunctions.cpp:136-155), from its
// closed form for non-decreasing edges: index[i] = #{j : x[j] <= edges[i]}, kept inside [1, x_length - 1] (SURVEY.md,
// appendix A) -- one merge walk over the two sorted sequences.
void histc(const double *x, int x_length, const double *edges, int edges_length, int *index) {
	int below = 0;  // how many grid points lie at or below the current edge
	for (int e = 0; e < edges_length; ++e) {
		while (below < x_length && x[below] <= edges[e]) ++below;
		index[e] = std::min(std::max(below, 1), x_length - 1);
	}
}

void interp1(const double *x, const double *y, int x_length, const double *xi, int xi_length, double *yi) {
	std::vector<int> k(xi_length > 0 ? xi_length : 0, 0);
	histc(x, x_length, xi, xi_length, k.data());
	for (int i = 0; i < xi_length; ++i) {
		const int j = k[i];
		const double s = (xi[i] - x[j - 1]) / (x[j] - x[j - 1]);
		yi[i] = y[j - 1] + s * (y[j] - y[j - 1]);
	}
}

// MATLAB's decimate as the reference uses it (reference src/world_matlabfunctions.cpp:184-210): the signal, extended by nine
// samples mirrored through each end point, goes through the order-3 low-pass forwards and then backwards (zero phase); every
// r-th sample of the result is kept, the grid being anchored so that it ends on the last input sample's side -- output m is
// the filtered sample at signal index (x_length mod r) - 1 + m r, for as long as that stays inside the extension.
void decimate(const double *x, int x_length, int r, double *y) {
	constexpr int kPad = 9;
	const int total = x_length + 2 * kPad;
	std::vector<double> fwd(total), tmp(total);
	for (int k = 1; k <= kPad; ++k) {
		fwd[kPad - k] = 2 * x[0] - x[k];                                          // left mirror through x[0]
		fwd[kPad + x_length - 1 + k] = 2 * x[x_length - 1] - x[x_length - 1 - k];  // right mirror through x[n - 1]
	}
	std::copy(x, x + x_length, fwd.begin() + kPad);
	iir3(fwd.data(), total, r, tmp.data());
	std::reverse(tmp.begin(), tmp.end());
	iir3(tmp.data(), total, r, fwd.data());
	std::reverse(fwd.begin(), fwd.end());
	int m = 0;
	for (int pos = kPad - 1 + x_length % r; pos < total - 1; pos += r) y[m++] = fwd[pos];
}

int matlab_round(double x) { return x > 0 ? static_cast<int>(x + 0.5) : static_cast<int>(x - 0.5); }

void diff(const double *x, int x_length, double *y) {
	for (int i = 0; i < x_length - 1; ++i) y[i] = x[i + 1] - x[i];
}

void interp1Q(double x, double shift, const double *y, int x_length, const double *xi, int xi_length, double *yi) {
	for (int i = 0; i < xi_length; ++i) yi[i] = interp1q_at(x, shift, y, x_length, xi[i]);
}

double matlab_std(const double *x, int x_length) {
	double average = 0.0;
	for (int i = 0; i < x_length; ++i) average += x[i];
	average /= x_length;
	double s = 0.0;
	for (int i = 0; i < x_length; ++i) s += std::pow(x[i] - average, 2.0);
	return std::sqrt(s / (x_length - 1));
}

int GetSuitableFFTSize(int sample) {
	return static_cast<int>(std::pow(2.0, static_cast<int>(std::log(static_cast<double>(sample)) / 0.69314718055994529) + 1.0));
}

void DCCorrection(const double *input, double current_f0, int fs, int fft_size, double *output) {
	const int upper = 2 + static_cast<int>(current_f0 * fft_size / fs);
	std::vector<double> rep(upper > 1 ? upper - 1 : 0);
	for (int i = 0; i < upper - 1; ++i)
		rep[i] = interp1q_at(current_f0, -static_cast<double>(fs) / fft_size, input, upper + 1, static_cast<double>(i) * fs / fft_size);
	for (int i = 0; i < upper - 1; ++i) output[i] = input[i] + rep[i];
}

void LinearSmoothing(const double *input, double width, int fs, int fft_size, double *output) {
	const int b = static_cast<int>(width * fft_size / fs) + 1;
	const int half = fft_size / 2, len = half + 2 * b + 1;
	std::vector<double> seg(len);
	auto mirrored = [&](int i) { return i < b ? input[b - i] : (i < half + b ? input[i - b] : input[half - (i - (half + b))]); };
	seg[0] = mirrored(0) * fs / fft_size;
	for (int i = 1; i < len; ++i) seg[i] = mirrored(i) * fs / fft_size + seg[i - 1];
	const double origin = -(b - 0.5) * fs / fft_size, step = static_cast<double>(fs) / fft_size;
	std::vector<double> res(half + 1);
	for (int i = 0; i <= half; ++i) {
		const double lo = static_cast<double>(i) / fft_size * fs - width / 2.0;
		res[i] = (interp1q_at(origin, step, seg.data(), len, lo + width) - interp1q_at(origin, step, seg.data(), len, lo)) / width;
	}
	std::copy(res.begin(), res.end(), output);
}

void NuttallWindow(int y_length, double *y) {
	const double pi = 3.1415926535897932384;
	for (int i = 0; i < y_length; ++i) {
		const double t = i / (y_length - 1.0);
		y[i] = 0.355768 - 0.487396 * std::cos(2.0 * pi * t) + 0.144232 * std::cos(4.0 * pi * t) - 0.012604 * std::cos(6.0 * pi * t);
	}
}

}  // extern "C"
