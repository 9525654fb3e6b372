// A C++ host that shards a ragged batch over two host threads (SURVEY.md section 8(e), 8(b) "Threading"): each thread selects
// the device (wc_set_device), takes the utterances wc_shard_partition deals to its rank, runs the fused pipeline on them through
// a pipeline handle and a HIP stream of its own (wc_set_stream), and brings its results to the host; put back into utterance
// order they must equal ONE wc_pipeline_run_device over the whole batch, bit for bit.  (On a node with several GPUs thread r
// would select device r and the final gather would be wc_gather_to_root_device on the caller's RCCL communicator; RCCL does not
// form a group of two ranks on one device, so on a one-GPU box the gather is the host copy below.)
//   threads <x.f64> <fs> <len_0> <len_1> ...      exit code 0 and "threads ok" on success
// Compiled by tests/test_gpu_multirank.py with g++ against include/ and libamdhip64 (for the streams).
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "world_class_c.h"
#include "world_class_shard.h"

struct Batch {
	std::vector<int> x_len, f_len, y_len;
	std::vector<double> x, tpos, f0, sp, ap, y;
};

static int run(int fs, int bins, Batch &b, void *stream, char *err, size_t errn) {
	if (wc_set_device(0)) return 1;
	if (wc_set_stream(stream)) return 1;
	wc_pipeline *p = wc_pipeline_create(fs, 5.0, 71.0, 800.0, -0.15, 71.0, 0, 0.85);
	if (!p) { std::snprintf(err, errn, "create: %s", wc_last_error()); return 1; }
	const int n = (int)b.x_len.size();
	long long nx = 0, nf = 0, ny = 0;
	b.f_len.resize(n); b.y_len.resize(n);
	for (int u = 0; u < n; ++u) {
		b.f_len[u] = wc_get_samples(fs, b.x_len[u], 5.0);
		b.y_len[u] = wc_synthesis_out_length(b.f_len[u], 5.0, fs);
		nx += b.x_len[u]; nf += b.f_len[u]; ny += b.y_len[u];
	}
	double *d_x = (double *)wc_device_malloc(8 * nx), *d_t = (double *)wc_device_malloc(8 * nf), *d_f = (double *)wc_device_malloc(8 * nf);
	double *d_sp = (double *)wc_device_malloc(8 * nf * bins), *d_ap = (double *)wc_device_malloc(8 * nf * bins), *d_y = (double *)wc_device_malloc(8 * ny);
	int rc = (d_x && d_t && d_f && d_sp && d_ap && d_y) ? 0 : 1;
	if (!rc) rc = wc_memcpy_h2d(d_x, b.x.data(), 8 * nx);
	if (!rc) rc = wc_pipeline_run_device(p, n, d_x, b.x_len.data(), d_t, d_f, d_sp, d_ap, d_y, nullptr);
	b.tpos.resize(nf); b.f0.resize(nf); b.sp.resize(nf * bins); b.ap.resize(nf * bins); b.y.resize(ny);
	if (!rc) rc = wc_memcpy_d2h(b.tpos.data(), d_t, 8 * nf) || wc_memcpy_d2h(b.f0.data(), d_f, 8 * nf) || wc_memcpy_d2h(b.sp.data(), d_sp, 8 * nf * bins) ||
				  wc_memcpy_d2h(b.ap.data(), d_ap, 8 * nf * bins) || wc_memcpy_d2h(b.y.data(), d_y, 8 * ny);
	if (rc) std::snprintf(err, errn, "run: %s", wc_last_error());
	wc_device_free(d_x); wc_device_free(d_t); wc_device_free(d_f); wc_device_free(d_sp); wc_device_free(d_ap); wc_device_free(d_y);
	wc_pipeline_destroy(p);
	wc_set_stream(nullptr);
	return rc;
}

int main(int argc, char **argv) {
	if (argc < 5) { std::fprintf(stderr, "usage: threads x.f64 fs len...\n"); return 2; }
	const int fs = std::atoi(argv[2]);
	const int n = argc - 3, world = 2;
	std::vector<int> len(n);
	long long total = 0;
	for (int u = 0; u < n; ++u) { len[u] = std::atoi(argv[3 + u]); total += len[u]; }
	std::vector<double> x(total);
	FILE *f = std::fopen(argv[1], "rb");
	if (!f || std::fread(x.data(), 8, total, f) != (size_t)total) { std::perror(argv[1]); return 2; }
	std::fclose(f);
	const int bins = wc_cheaptrick_fft_size(fs, 71.0) / 2 + 1;
	std::vector<long long> off(n + 1, 0);
	for (int u = 0; u < n; ++u) off[u + 1] = off[u] + len[u];

	// the whole batch on one thread: the result the shards have to reproduce
	Batch whole;
	whole.x_len = len; whole.x = x;
	char err[512] = "";
	if (run(fs, bins, whole, nullptr, err, sizeof err)) { std::fprintf(stderr, "whole batch: %s\n", err); return 1; }

	std::vector<int> rank_of(n);
	if (wc_shard_partition(len.data(), n, world, rank_of.data())) { std::fprintf(stderr, "partition: %s\n", wc_last_error()); return 1; }
	Batch shard[2];
	std::vector<int> mine[2];
	for (int u = 0; u < n; ++u) {
		const int r = rank_of[u];
		if (r < 0 || r >= world) { std::fprintf(stderr, "utterance %d dealt to rank %d\n", u, r); return 1; }
		mine[r].push_back(u);
		shard[r].x_len.push_back(len[u]);
		shard[r].x.insert(shard[r].x.end(), x.begin() + off[u], x.begin() + off[u + 1]);
	}
	if (mine[0].empty() || mine[1].empty()) { std::fprintf(stderr, "a rank got nothing\n"); return 1; }
	int rcs[2] = {0, 0};
	char errs[2][512] = {"", ""};
	hipStream_t st[2];
	for (int r = 0; r < world; ++r)
		if (hipStreamCreateWithFlags(&st[r], hipStreamNonBlocking) != hipSuccess) { std::fprintf(stderr, "hipStreamCreate failed\n"); return 1; }
	std::thread th[2];
	for (int r = 0; r < world; ++r) th[r] = std::thread([&, r] { rcs[r] = run(fs, bins, shard[r], st[r], errs[r], sizeof errs[r]); });
	for (int r = 0; r < world; ++r) th[r].join();
	for (int r = 0; r < world; ++r) {
		if (rcs[r]) { std::fprintf(stderr, "rank %d: %s\n", r, errs[r]); return 1; }
		(void)hipStreamDestroy(st[r]);
	}
	// the "gather": the shards back into utterance order, against the one-thread run
	std::vector<long long> wf(n + 1, 0), wy(n + 1, 0);
	for (int u = 0; u < n; ++u) { wf[u + 1] = wf[u] + whole.f_len[u]; wy[u + 1] = wy[u] + whole.y_len[u]; }
	for (int r = 0; r < world; ++r) {
		long long fo = 0, yo = 0;
		for (size_t k = 0; k < mine[r].size(); ++k) {
			const int u = mine[r][k];
			const long long nf = shard[r].f_len[k], ny = shard[r].y_len[k];
			if (nf != whole.f_len[u] || ny != whole.y_len[u]) { std::fprintf(stderr, "utterance %d: sizes differ\n", u); return 1; }
			const bool same = !std::memcmp(&shard[r].f0[fo], &whole.f0[wf[u]], 8 * nf) && !std::memcmp(&shard[r].tpos[fo], &whole.tpos[wf[u]], 8 * nf) &&
							  !std::memcmp(&shard[r].sp[fo * bins], &whole.sp[wf[u] * bins], 8 * nf * bins) &&
							  !std::memcmp(&shard[r].ap[fo * bins], &whole.ap[wf[u] * bins], 8 * nf * bins) &&
							  !std::memcmp(&shard[r].y[yo], &whole.y[wy[u]], 8 * ny);
			if (!same) { std::fprintf(stderr, "utterance %d (rank %d): the shard's result differs from the whole batch's\n", u, r); return 1; }
			fo += nf; yo += ny;
		}
	}
	std::printf("threads ok: %d utterances, %zu + %zu over two threads\n", n, mine[0].size(), mine[1].size());
	return 0;
}
