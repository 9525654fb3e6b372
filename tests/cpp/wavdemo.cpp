// File-level drop-in check: the reference's tool and codec headers (tools/audioio.hpp, tools/parameterio.hpp,
// include/codec.hpp) replaced by include/{audioio,parameterio,codec}.hpp of this repository.
//   wavdemo in.wav out_prefix   -> out_prefix.wav (resynthesis), .f0 / .sp / .ap parameter files, .mcep (coded envelope)
// Compiled by tests/test_gpu_cpp_dropin.py with plain g++.
#include <cstdio>
#include <string>
#include <vector>

#include "audioio.hpp"
#include "cheaptrick.hpp"
#include "codec.hpp"
#include "d4c.hpp"
#include "harvest.hpp"
#include "parameterio.hpp"
#include "synthesis.hpp"

using namespace world_class;

int main(int argc, char **argv) {
	if (argc != 3) { std::fprintf(stderr, "usage: wavdemo in.wav out_prefix\n"); return 2; }
	const int x_length = GetAudioLength(argv[1]);
	if (x_length <= 0) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
	std::vector<double> x(x_length);
	int fs = 0, nbit = 0;
	wavread(argv[1], &fs, &nbit, x.data());
	const std::string prefix = argv[2];
	const double frame_period = 5.0;
	try {
		HarvestOption hopt;
		hopt.frame_period = frame_period;
		Harvest harvest(fs, hopt);
		const int n = harvest.getSamples(fs, x_length);
		std::vector<double> f0(n), tpos(n);
		harvest.compute(x.data(), x_length, tpos.data(), f0.data());
		CheapTrick cheaptrick(fs);
		const int fft_size = cheaptrick.getFFTSizeForCheapTrick(fs, 71.0);
		const int bins = fft_size / 2 + 1;
		std::vector<double> sp(static_cast<size_t>(n) * bins), ap(sp.size());
		std::vector<double *> sp_rows(n), ap_rows(n);
		for (int i = 0; i < n; ++i) { sp_rows[i] = &sp[static_cast<size_t>(i) * bins]; ap_rows[i] = &ap[static_cast<size_t>(i) * bins]; }
		cheaptrick.compute(x.data(), x_length, tpos.data(), f0.data(), n, sp_rows.data());
		D4C d4c(fs);
		d4c.compute(x.data(), x_length, tpos.data(), f0.data(), n, fft_size, ap_rows.data());
		WriteF0((prefix + ".f0").c_str(), n, frame_period, tpos.data(), f0.data(), 0);
		WriteSpectralEnvelope((prefix + ".sp").c_str(), fs, n, frame_period, fft_size, 0, sp_rows.data());
		WriteAperiodicity((prefix + ".ap").c_str(), fs, n, frame_period, fft_size, 0, ap_rows.data());
		const int nd = 40;
		std::vector<double> mcep(static_cast<size_t>(n) * nd);
		std::vector<double *> mcep_rows(n);
		for (int i = 0; i < n; ++i) mcep_rows[i] = &mcep[static_cast<size_t>(i) * nd];
		CodeSpectralEnvelope(sp_rows.data(), n, fs, fft_size, nd, mcep_rows.data());
		WriteSpectralEnvelope((prefix + ".mcep").c_str(), fs, n, frame_period, fft_size, nd, mcep_rows.data());
		const int y_length = static_cast<int>((n - 1) * frame_period / 1000.0 * fs) + 1;
		std::vector<double> y(y_length);
		Synthesis synthesis(fs, fft_size, frame_period);
		synthesis.compute(f0.data(), n, sp_rows.data(), ap_rows.data(), y_length, y.data());
		wavwrite(y.data(), y_length, fs, 16, (prefix + ".wav").c_str());
	} catch (const std::exception &e) {
		std::fprintf(stderr, "wavdemo: %s\n", e.what());
		return 1;
	}
	return 0;
}
