// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or called from the product path.
//
// Stage taps on the *real* reference's Harvest, compiled from the sources where they lie under /root/reference by
// oracle/Makefile into oracle/_ref/libworld_ref_taps.so (git-ignored).  The reference keeps its stages private
// (include/harvest.hpp:46-); this translation unit opens them with a macro and calls the reference's OWN member
// functions in the order its generalBody does (src/harvest.cpp:1380-1453, :1352-1374, :619-634), copying what lies
// between them out: the decimated signal, the raw per-band candidates, the candidates and scores after refinement and
// after removeUnreliableCandidates, the base contour, the fixed contour and the smoothed 1 ms contour.  This is what
// lets tests/ pin the restatement (oracle/wc_oracle.cpp) stage by stage rather than end to end only.
// No reference source is copied; nothing about the reference is altered (the zero-filling operator new[] is the one of
// ref_shim.cpp, for the same reason).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <numeric>
#include <vector>

#define private public
#include "harvest.hpp"
#undef private
#include "world_common.hpp"
#include "world_constantnumbers.hpp"
#include "world_fft.hpp"
#include "world_matlabfunctions.hpp"

void *operator new[](std::size_t n) {
	void *p = std::calloc(1, n ? n : 1);
	if (!p) throw std::bad_alloc();
	return p;
}
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }

using namespace world_class;

extern "C" {

// dims = {y_length, number of bands, max_candidates, number_of_candidates (after the overlap)}; every output pointer may
// be NULL; returns the number of 1 ms frames.  raw is [band][frame], cand* / score* are [frame][max_candidates].
int ref_harvest_taps(const double *x, int x_length, int fs, double f0_floor, double f0_ceil, int *dims, double *y, double *raw,
					 double *cand_refined, double *score_refined, double *cand, double *score, double *f0_base, double *f0_fixed,
					 double *f0_1ms) {
	HarvestOption opt;
	opt.f0_floor = f0_floor;
	opt.f0_ceil = f0_ceil;
	opt.frame_period = 1.0;
	Harvest h(fs, opt);
	const int L = h.getSamples(fs, x_length, 1.0);
	std::vector<double> tpos(L), out(L);

	h.x_ = x;
	h.x_length_ = x_length;
	h.temporal_positions_ = tpos.data();
	const double lo = h.option_.f0_floor * 0.9, hi = h.option_.f0_ceil * 1.1;
	const int bands = 1 + static_cast<int>(std::log(hi / lo) / world::kLog2 * h.option_.channels_in_octave);
	std::vector<double> edges(bands);
	for (int i = 0; i < bands; ++i) edges[i] = lo * std::pow(2.0, static_cast<double>(i + 1) / h.option_.channels_in_octave);
	h.y_length_ = 1 + static_cast<int>(x_length / h.decimation_ratio_);
	const int fft_size = GetSuitableFFTSize(h.y_length_ + 4 * static_cast<int>(1.0 + h.actual_fs_ / edges[0] / 2.0));
	h.y_ = new double[fft_size]();
	fft_complex *spec = new fft_complex[fft_size / 2 + 1];
	h.getWaveformAndSpectrum(fft_size, h.decimation_ratio_, spec);
	h.f0_length_ = L;
	for (int i = 0; i < L; ++i) tpos[i] = i * 1 / 1000.0;
	const int max_cand = matlab_round(bands / 10) * 7;
	h.f0_candidates_ = new double *[L];
	h.f0_candidates_score_ = new double *[L];
	for (int i = 0; i < L; ++i) {
		h.f0_candidates_[i] = new double[max_cand]();
		h.f0_candidates_score_[i] = new double[max_cand]();
	}
	if (y) std::copy(h.y_, h.y_ + h.y_length_, y);

	std::vector<double *> rows(bands);
	std::vector<double> raw_store(static_cast<size_t>(bands) * L);
	for (int b = 0; b < bands; ++b) rows[b] = raw_store.data() + static_cast<size_t>(b) * L;
	h.getRawF0Candidates(edges.data(), bands, spec, fft_size, rows.data());
	if (raw) std::copy(raw_store.begin(), raw_store.end(), raw);
	const int n0 = h.detectOfficialF0Candidates(rows.data(), bands, L, max_cand, h.f0_candidates_);
	h.overlapF0Candidates(L, n0, h.f0_candidates_);
	h.number_of_candidates_ = n0 * 7;
	if (dims) { dims[0] = h.y_length_; dims[1] = bands; dims[2] = max_cand; dims[3] = h.number_of_candidates_; }

	auto tap = [&](double **src, double *dst) {
		if (dst) for (int i = 0; i < L; ++i) std::copy(src[i], src[i] + max_cand, dst + static_cast<size_t>(i) * max_cand);
	};
	h.refineF0Candidates();
	tap(h.f0_candidates_, cand_refined);
	tap(h.f0_candidates_score_, score_refined);
	h.removeUnreliableCandidates();
	tap(h.f0_candidates_, cand);
	tap(h.f0_candidates_score_, score);
	if (f0_base) h.searchF0Base(h.f0_candidates_, h.f0_candidates_score_, L, h.number_of_candidates_, f0_base);
	std::vector<double> best(L);
	h.fixF0Contour(best.data());
	if (f0_fixed) std::copy(best.begin(), best.end(), f0_fixed);
	h.smoothF0Contour(best.data(), out.data());
	if (f0_1ms) std::copy(out.begin(), out.end(), f0_1ms);

	delete[] h.y_;
	delete[] spec;
	for (int i = 0; i < L; ++i) { delete[] h.f0_candidates_[i]; delete[] h.f0_candidates_score_[i]; }
	delete[] h.f0_candidates_;
	delete[] h.f0_candidates_score_;
	return L;
}

}  // extern "C"
