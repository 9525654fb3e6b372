// Host-side copies of the host-pointer entry points, spread over a few threads: one thread moves 5-10 GB/s, and the
// reference's interface hands spectrogram / aperiodicity over as one pointer per frame (reference include/cheaptrick.hpp:24,
// d4c.hpp:27, synthesis.hpp:36): 16 MB per 10 s utterance at 48 kHz in 2001 rows.
// Round 6: the threads are started once and wait for work (starting eight of them cost 0.1-0.3 ms of every call of a drop-in
// caller), and rows cross PCIe in pieces, each piece's host copy beside the next one's transfer (rows_down / rows_up).
#pragma once
#include <pthread.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "wc_internal.hpp"

namespace wc {

struct CopyJob { void *dst; const void *src; size_t bytes; };

namespace hostcopy {
struct Piece { char *dst; const char *src; size_t bytes; };

// A process-wide set of waiting threads.  Never destroyed (the threads are detached and sleep on a condition variable; the
// process's exit ends them); a forked child starts its own on first use.
class Pool {
 public:
	static Pool &get() {
		std::lock_guard<std::mutex> g(*slot_mutex());
		Pool *&p = slot();
		if (!p) {
			p = new Pool;
			static std::once_flag once;
			// (a forked child has none of the threads, and the mutex may have been held by a thread it does not have either)
			std::call_once(once, [] { pthread_atfork(nullptr, nullptr, [] { slot() = nullptr; slot_mutex() = new std::mutex; }); });
		}
		return *p;
	}
	// the pieces are copied by the caller and up to `threads - 1` of the waiting threads; returns when all are done
	void run(const std::vector<Piece> &pieces, size_t threads) {
		std::lock_guard<std::mutex> one(run_m_);  // one set of pieces at a time
		std::atomic<size_t> next{0};
		const std::function<void()> work = [&]() {
			for (size_t i = next.fetch_add(1); i < pieces.size(); i = next.fetch_add(1)) std::memcpy(pieces[i].dst, pieces[i].src, pieces[i].bytes);
		};
		const size_t helpers = std::min(threads > 0 ? threads - 1 : 0, n_workers_);
		if (helpers) {
			std::lock_guard<std::mutex> lk(m_);
			work_ = &work;
			tickets_ = helpers;
			cv_work_.notify_all();
		}
		work();
		if (helpers) {
			std::unique_lock<std::mutex> lk(m_);
			tickets_ = 0;  // (a thread that wakes up late finds nothing to join)
			cv_done_.wait(lk, [&] { return active_ == 0; });
			work_ = nullptr;
		}
	}

 private:
	Pool() {
		const unsigned hw = std::thread::hardware_concurrency();
		n_workers_ = std::min<size_t>(hw ? hw : 4u, 16u) - 1;
		for (size_t t = 0; t < n_workers_; ++t) std::thread([this] { loop(); }).detach();
	}
	void loop() {
		std::unique_lock<std::mutex> lk(m_);
		for (;;) {
			cv_work_.wait(lk, [&] { return tickets_ > 0; });
			--tickets_;
			++active_;
			const std::function<void()> *w = work_;
			lk.unlock();
			(*w)();
			lk.lock();
			if (--active_ == 0) cv_done_.notify_all();
		}
	}
	static Pool *&slot() { static Pool *p = nullptr; return p; }
	static std::mutex *&slot_mutex() { static std::mutex *m = new std::mutex; return m; }
	std::mutex m_, run_m_;
	std::condition_variable cv_work_, cv_done_;
	const std::function<void()> *work_ = nullptr;
	size_t tickets_ = 0, active_ = 0, n_workers_ = 0;
};
}  // namespace hostcopy

inline void parallel_copy(const std::vector<CopyJob> &jobs) {
	constexpr size_t kPiece = 256u << 10;
	std::vector<hostcopy::Piece> pieces;
	size_t total = 0;
	for (const CopyJob &j : jobs)
		for (size_t o = 0; o < j.bytes; o += kPiece) {
			pieces.push_back({static_cast<char *>(j.dst) + o, static_cast<const char *>(j.src) + o, std::min(kPiece, j.bytes - o)});
			total += pieces.back().bytes;
		}
	if (pieces.empty()) return;
	// a thread moves 5 - 10 GB/s and is woken in ~10 us: one per 512 KB up to 16
	const size_t nt = std::min<size_t>({16u, total / (512u << 10) + 1, pieces.size()});
	if (nt <= 1) {
		for (const hostcopy::Piece &p : pieces) std::memcpy(p.dst, p.src, p.bytes);
		return;
	}
	hostcopy::Pool::get().run(pieces, nt);
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

// one host array -> device through page-locked staging (a pageable source takes the runtime's slow path: 3.8 MB of samples in
// 0.2 ms and four blit kernels where the link needs 0.07): gathered by the waiting threads in two pieces, the second beside the
// first one's transfer.  Asynchronous; `stage` is marked busy until the stream has passed.
inline int array_up(hipStream_t s, const double *src, size_t n, HostBuf &stage, double *d_dst) {
	if (n == 0) return WC_OK;
	int rc;
	if ((rc = stage.reserve(sizeof(double) * n))) return rc;
	double *h = stage.as<double>();
	const size_t cut = n > ((size_t)1 << 17) ? n / 2 : n;
	for (size_t o = 0; o < n; o += cut) {
		const size_t m = std::min(cut, n - o);
		parallel_copy({{h + o, src + o, sizeof(double) * m}});
		WC_HIP(hipMemcpyAsync(d_dst + o, h + o, sizeof(double) * m, hipMemcpyHostToDevice, s));
	}
	return stage.mark(s);
}

// pieces of ~4 MB of rows (at most eight): small matrices go as one
inline int rows_pieces(int n_rows, int bins, int *first) {
	const size_t bytes = sizeof(double) * (size_t)n_rows * bins;
	const int k = (int)std::min<size_t>({(size_t)8, bytes / ((size_t)4 << 20) + 1, (size_t)std::max(n_rows, 1)});
	for (int p = 0; p <= k; ++p) first[p] = (int)((long long)n_rows * p / k);
	return k;
}

// device matrices (one per utterance, packed one behind the other in `d_src`) -> the callers' rows through page-locked staging
// `h_stage`: the transfers are all enqueued, in pieces, and every piece goes to its rows while the next ones are still on the
// link.  Returns with everything in place (the stream is drained).
inline int rows_down_many(hipStream_t s, int n, double *const *const *rows, const int *n_rows, int bins, const double *d_src, double *h_stage) {
	struct Part { int u, r0, r1; size_t off; hipEvent_t ev; };
	std::vector<Part> parts;
	size_t base = 0;
	for (int u = 0; u < n; ++u) {
		if (n_rows[u] <= 0) continue;
		int first[9];
		const int k = rows_pieces(n_rows[u], bins, first);
		for (int p = 0; p < k; ++p) parts.push_back({u, first[p], first[p + 1], base + (size_t)first[p] * bins, nullptr});
		base += (size_t)n_rows[u] * bins;
	}
	for (Part &q : parts) {
		WC_HIP(hipMemcpyAsync(h_stage + q.off, d_src + q.off, sizeof(double) * (size_t)(q.r1 - q.r0) * bins, hipMemcpyDeviceToHost, s));
		if (parts.size() > 1) {
			WC_HIP(hipEventCreateWithFlags(&q.ev, hipEventDisableTiming));
			WC_HIP(hipEventRecord(q.ev, s));
		}
	}
	for (Part &q : parts) {
		if (q.ev) {
			WC_HIP(hipEventSynchronize(q.ev));
			WC_HIP(hipEventDestroy(q.ev));
		} else {
			WC_HIP(hipStreamSynchronize(s));
		}
		rows_copy(rows[q.u] + q.r0, q.r1 - q.r0, bins, h_stage + q.off, true);
	}
	return WC_OK;
}
inline int rows_down(hipStream_t s, double *const *rows, int n_rows, int bins, const double *d_src, double *h_stage) {
	return rows_down_many(s, 1, &rows, &n_rows, bins, d_src, h_stage);
}

// the caller's rows -> device matrix through page-locked staging: a piece is gathered while the one before it is on the link
// (asynchronous: `h_stage` has to stay untouched until the stream has passed)
inline int rows_up(hipStream_t s, const double *const *rows, int n_rows, int bins, double *h_stage, double *d_dst) {
	int first[9];
	const int k = rows_pieces(n_rows, bins, first);
	for (int p = 0; p < k; ++p) {
		const size_t o = (size_t)first[p] * bins, n = (size_t)(first[p + 1] - first[p]) * bins;
		rows_copy(const_cast<double *const *>(rows) + first[p], first[p + 1] - first[p], bins, h_stage + o, false);
		WC_HIP(hipMemcpyAsync(d_dst + o, h_stage + o, sizeof(double) * n, hipMemcpyHostToDevice, s));
	}
	return WC_OK;
}

}  // namespace wc
