// The reference demo's call sequence (reference test/test.cpp:76-264, :288-384) against the drop-in headers:
// Harvest (f0_floor 40 like the demo) -> CheapTrick -> D4C -> Synthesis on a raw little-endian float64 file.
//   demo <in.f64> <fs> <out_prefix>   writes <out_prefix>.{f0,sp,ap,y}.f64
// Compiled by tests/test_cpp_dropin.py with plain g++ (no HIP headers needed on the caller's side).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cheaptrick.hpp"
#include "d4c.hpp"
#include "harvest.hpp"
#include "synthesis.hpp"

using namespace world_class;

static void dump(const std::string &path, const double *p, size_t n) {
	FILE *f = std::fopen(path.c_str(), "wb");
	if (!f || std::fwrite(p, sizeof(double), n, f) != n) { std::perror(path.c_str()); std::exit(2); }
	std::fclose(f);
}

int main(int argc, char **argv) {
	if (argc != 4) { std::fprintf(stderr, "usage: demo in.f64 fs out_prefix\n"); return 2; }
	FILE *f = std::fopen(argv[1], "rb");
	if (!f) { std::perror(argv[1]); return 2; }
	std::fseek(f, 0, SEEK_END);
	const int x_length = static_cast<int>(std::ftell(f) / sizeof(double));
	std::fseek(f, 0, SEEK_SET);
	std::vector<double> x(x_length);
	if (std::fread(x.data(), sizeof(double), x_length, f) != static_cast<size_t>(x_length)) return 2;
	std::fclose(f);
	const int fs = std::atoi(argv[2]);
	const std::string prefix = argv[3];
	try {
		HarvestOption hopt;
		hopt.frame_period = 5.0;
		hopt.f0_floor = 40.0;  // reference test/test.cpp:87
		Harvest harvest(fs, hopt);
		const int f0_length = harvest.getSamples(fs, x_length);
		std::vector<double> f0(f0_length), time_axis(f0_length);
		harvest.compute(x.data(), x_length, time_axis.data(), f0.data());

		CheapTrickOption copt;
		copt.f0_floor = 71.0;
		CheapTrick cheaptrick(fs, copt);
		const int fft_size = cheaptrick.getFFTSizeForCheapTrick(fs, copt.f0_floor);
		const int bins = fft_size / 2 + 1;
		std::vector<double> sp(static_cast<size_t>(f0_length) * bins), ap(sp.size());
		std::vector<double *> sp_rows(f0_length), ap_rows(f0_length);
		for (int i = 0; i < f0_length; ++i) { sp_rows[i] = &sp[static_cast<size_t>(i) * bins]; ap_rows[i] = &ap[static_cast<size_t>(i) * bins]; }
		cheaptrick.compute(x.data(), x_length, time_axis.data(), f0.data(), f0_length, sp_rows.data());

		D4COption dopt;
		dopt.threshold = 0.85;
		D4C d4c(fs, dopt);
		d4c.compute(x.data(), x_length, time_axis.data(), f0.data(), f0_length, fft_size, ap_rows.data());

		const int y_length = static_cast<int>((f0_length - 1) * 5.0 / 1000.0 * fs) + 1;  // reference test/test.cpp:362-363
		std::vector<double> y(y_length);
		Synthesis synthesis(fs, fft_size, 5.0);
		synthesis.compute(f0.data(), f0_length, sp_rows.data(), ap_rows.data(), y_length, y.data());

		dump(prefix + ".f0.f64", f0.data(), f0.size());
		dump(prefix + ".sp.f64", sp.data(), sp.size());
		dump(prefix + ".ap.f64", ap.data(), ap.size());
		dump(prefix + ".y.f64", y.data(), y.size());
		std::printf("frames %d fft_size %d y_length %d\n", f0_length, fft_size, y_length);
	} catch (const std::exception &e) {
		std::fprintf(stderr, "error: %s\n", e.what());
		return 1;
	}
	return 0;
}
