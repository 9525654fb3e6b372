// Host side of the host-pointer entry points (world_class_amd/csrc/wc_hostcopy.hpp): the copy threads are started once and woken
// per copy.  No GPU needed: rows are scattered from one host array into another -- many times over (the threads' hand-over), in
// a forked child (which has none of the parent's threads and starts its own) and from two callers at once.
//   hipcc -x hip --offload-arch=gfx950 -O2 -std=c++17 -pthread -I world_class_amd/csrc -I include tests/cpp/hostcopy_pool.cpp \
//         -L world_class_amd -lworldclass_hip -Wl,-rpath,$PWD/world_class_amd -o hostcopy_pool && ./hostcopy_pool
#include <sys/wait.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "wc_hostcopy.hpp"

int main() {
	const int n_rows = 2001, bins = 1025;
	const size_t n = (size_t)n_rows * bins;
	double *a = static_cast<double *>(aligned_alloc(4096, n * 8)), *b = static_cast<double *>(aligned_alloc(4096, n * 8));
	for (size_t i = 0; i < n; ++i) a[i] = (double)i;
	std::vector<double *> rows(n_rows);
	for (int i = 0; i < n_rows; ++i) rows[i] = b + (size_t)i * bins;
	for (int rep = 0; rep < 100; ++rep) {
		std::memset(b, 0, n * 8);
		wc::rows_copy(rows.data(), n_rows, bins, a, true);
		if (std::memcmp(a, b, n * 8)) { std::printf("MISMATCH in repetition %d\n", rep); return 1; }
	}
	// rows that do NOT lie one behind the other (every second row of a wider matrix)
	{
		std::vector<double> wide(2 * n);
		std::vector<double *> apart(n_rows);
		for (int i = 0; i < n_rows; ++i) apart[i] = wide.data() + (size_t)2 * i * bins;
		wc::rows_copy(apart.data(), n_rows, bins, a, true);
		for (int i = 0; i < n_rows; ++i)
			if (std::memcmp(apart[i], a + (size_t)i * bins, bins * 8)) { std::printf("MISMATCH in separate row %d\n", i); return 1; }
		std::memset(b, 0, n * 8);
		wc::rows_copy(apart.data(), n_rows, bins, b, false);
		if (std::memcmp(a, b, n * 8)) { std::printf("MISMATCH gathering separate rows\n"); return 1; }
	}
	const pid_t child = fork();
	if (child == 0) {
		std::memset(b, 0, n * 8);
		wc::rows_copy(rows.data(), n_rows, bins, a, true);
		_exit(std::memcmp(a, b, n * 8) ? 3 : 0);
	}
	int status = 0;
	waitpid(child, &status, 0);
	if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) { std::printf("forked child failed (%d)\n", status); return 1; }
	std::memset(b, 0, n * 8);
	std::thread t1([&] { for (int r = 0; r < 40; ++r) wc::rows_copy(rows.data(), 1000, bins, a, true); });
	std::thread t2([&] { for (int r = 0; r < 40; ++r) wc::rows_copy(rows.data() + 1000, n_rows - 1000, bins, a + (size_t)1000 * bins, true); });
	t1.join();
	t2.join();
	if (std::memcmp(a, b, n * 8)) { std::printf("MISMATCH with two callers\n"); return 1; }
	std::printf("ok\n");
	return 0;
}
