// Host-side copies of the host-pointer entry points, spread over a few threads: one thread moves 5-10 GB/s, and the
// reference's interface hands spectrogram / aperiodicity over as one pointer per frame (reference include/cheaptrick.hpp:24,
// d4c.hpp:27, synthesis.hpp:36): 16 MB per 10 s utterance at 48 kHz in 2001 rows.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

namespace wc {

struct CopyJob { void *dst; const void *src; size_t bytes; };

inline void parallel_copy(const std::vector<CopyJob> &jobs) {
	constexpr size_t kPiece = 1u << 20;
	struct Piece { char *dst; const char *src; size_t bytes; };
	std::vector<Piece> pieces;
	size_t total = 0;
	for (const CopyJob &j : jobs)
		for (size_t o = 0; o < j.bytes; o += kPiece) {
			pieces.push_back({static_cast<char *>(j.dst) + o, static_cast<const char *>(j.src) + o, std::min(kPiece, j.bytes - o)});
			total += pieces.back().bytes;
		}
	unsigned hw = std::thread::hardware_concurrency();
	// (a thread moves 5 - 10 GB/s and costs ~50 us to start: one per 2 MB up to 16 -- with one per 8 MB the 16 MB of rows of a
	// 10 s utterance went over three threads, 1 ms of the 2 ms of a host-pointer CheapTrick call)
	size_t nt = std::min<size_t>({hw ? hw : 4u, 16u, total / (2u << 20) + 1, pieces.size()});
	std::atomic<size_t> next{0};
	auto work = [&]() {
		for (size_t i = next.fetch_add(1); i < pieces.size(); i = next.fetch_add(1)) std::memcpy(pieces[i].dst, pieces[i].src, pieces[i].bytes);
	};
	std::vector<std::thread> th;
	for (size_t t = 1; t < nt; ++t) th.emplace_back(work);
	work();
	for (std::thread &t : th) t.join();
}

// rows[i] (n_rows pointers to `bins` doubles each) <-> one packed array; runs of rows that lie one behind the other in the
// caller's memory (a numpy matrix, one big allocation cut into rows) are moved as one piece
inline void rows_copy(double *const *rows, int n_rows, int bins, double *packed, bool to_rows) {
	std::vector<CopyJob> jobs;
	int i = 0;
	while (i < n_rows) {
		int j = i + 1;
		while (j < n_rows && rows[j] == rows[j - 1] + bins) ++j;
		const size_t bytes = sizeof(double) * (size_t)(j - i) * bins;
		if (to_rows) jobs.push_back({rows[i], packed + (size_t)i * bins, bytes});
		else jobs.push_back({packed + (size_t)i * bins, rows[i], bytes});
		i = j;
	}
	parallel_copy(jobs);
}

}  // namespace wc
