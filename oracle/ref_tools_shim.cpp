// TEST INFRASTRUCTURE ONLY.  Gives the golden generator access to the reference's demo-level code: the tools
// (tools/audioio.cpp, tools/parameterio.cpp are extern "C" already and are compiled alongside) and
// ParameterModification, which lives in the unnamed namespace of test/test.cpp -- so that translation unit is
// compiled here from where it lies, with its main() renamed.
#include <cstdio>

#define main ref_demo_main
#include "test/test.cpp"  // found through -I$(REF)
#undef main

extern "C" void ref_parameter_modification(int fs, int f0_length, int fft_size, double *f0, double **spectrogram, int n_args,
											  double shift, double ratio) {
	// the reference parses its factors with atof from argv[3] / argv[4]; %.17g round-trips a double exactly
	char a3[64], a4[64];
	std::snprintf(a3, sizeof a3, "%.17g", shift);
	std::snprintf(a4, sizeof a4, "%.17g", ratio);
	char a0[] = "demo", a1[] = "in.wav", a2[] = "out.wav";
	char *argv[5] = {a0, a1, a2, a3, a4};
	ParameterModification(3 + n_args, argv, fs, f0_length, fft_size, f0, spectrogram);
}
