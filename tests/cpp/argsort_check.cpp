// Compares wc_argsort::sort_like_libstdcxx with std::sort of the C++ library this is compiled with (GNU libstdc++ here,
// the one the reference is built with) on tie-heavy keys, the way Harvest's mergeF0 uses it (reference
// src/harvest.cpp:508-513): order[] = 0..n-1 sorted by key[2 * i].  Exit code 0 = identical on every case.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../world_class_amd/csrc/wc_argsort.hpp"

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned rnd() {
	rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
	return (unsigned)(rng_state >> 32);
}

int main() {
	long cases = 0;
	for (int n = 0; n <= 400; ++n) {
		for (int rep = 0; rep < 60; ++rep) {
			std::vector<int> key(2 * n + 2);
			const int kind = rep % 6;
			int run = 0;
			for (int i = 0; i < n; ++i) {
				int k;
				switch (kind) {
				case 0: k = (int)(rnd() % (unsigned)(n / 3 + 1)); break;           // many ties, random
				case 1: run += (int)(rnd() % 3); k = run; break;                    // ascending with ties (the usual case)
				case 2: k = n - i / 2; break;                                       // descending pairs
				case 3: k = (int)(rnd() % 2u); break;                               // two values
				case 4: k = i < 3 ? 0 : i * 7; break;                               // several sections reaching frame 0
				default: k = (int)(rnd() % 1000000u); break;                        // mostly distinct
				}
				key[2 * i] = k;
				key[2 * i + 1] = -1;
			}
			std::vector<int> a(n), b(n);
			for (int i = 0; i < n; ++i) a[i] = b[i] = i;
			std::sort(a.begin(), a.end(), [&](int i1, int i2) { return key[i1 * 2] < key[i2 * 2]; });
			wc_argsort::sort_like_libstdcxx(b.data(), n, wc_argsort::ByKey{key.data(), 2});
			if (a != b) {
				std::printf("mismatch at n=%d kind=%d\n", n, kind);
				return 1;
			}
			++cases;
		}
	}
	// adversarial for the depth limit: a median-of-three killer sequence drives introsort into its heap sort
	for (int n = 64; n <= 4096; n *= 2) {
		std::vector<int> key(2 * n);
		// classic killer: first half odd positions ascending, evens in the second half
		std::vector<int> v(n);
		for (int i = 0; i < n; ++i) v[i] = i;
		const int k = n / 2;
		for (int i = 1; i <= k; ++i) {
			if (i % 2 == 1) { v[i - 1] = i; v[i] = k + i; }
			v[k + i - 1] = 2 * i;
		}
		for (int i = 0; i < n; ++i) key[2 * i] = v[i] / 2;  // and ties on top
		std::vector<int> a(n), b(n);
		for (int i = 0; i < n; ++i) a[i] = b[i] = i;
		std::sort(a.begin(), a.end(), [&](int i1, int i2) { return key[i1 * 2] < key[i2 * 2]; });
		wc_argsort::sort_like_libstdcxx(b.data(), n, wc_argsort::ByKey{key.data(), 2});
		if (a != b) {
			std::printf("mismatch on the killer sequence, n=%d\n", n);
			return 1;
		}
		++cases;
	}
	std::printf("ok %ld cases\n", cases);
	return 0;
}
