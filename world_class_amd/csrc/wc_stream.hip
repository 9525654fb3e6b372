// Chunked Harvest + CheapTrick for many concurrent streams: include/world_class_stream.h states the semantics (the
// reference has none -- Harvest is non-causal, reference src/harvest.cpp:431-440, :676-703).  This file is the host logic
// around the batched stages: per-stream histories in HBM (ping-pong rows), the window batch handed to Harvest, the
// bookkeeping of which absolute frames a push commits, and the carried noise-stream positions of CheapTrick.
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "wc_internal.hpp"
#include "../../include/world_class_stream.h"

namespace wc {

struct StreamDesc {
	long long src_off;    // first kept sample of the old history row (row start + dropped samples)
	long long dst_off;    // start of the new history row
	long long chunk_off;  // first new sample in the packed chunk array
	long long batch_off;  // start of this stream's window in the packed batch handed to CheapTrick
	long long hbatch_off; // ... and in the packed batch handed to Harvest (windows cut to a multiple of the decimation ratio)
	int h_len, pad_;      // samples of the window Harvest sees
	int keep, n_new;      // old samples kept, new samples appended (window length = keep + n_new)
	// commit: `count` frames from window row `row0` (packed Harvest output offset hf_off) to packed output offset out_off
	long long hf_off, out_off, first_frame;
	int row0, count;
};

// new history row = kept tail of the old one followed by the new chunk; the same samples also go to the packed batch
__global__ void stream_update_kernel(const StreamDesc *__restrict__ desc, const double *__restrict__ old_hist,
									 const double *__restrict__ chunk, double *__restrict__ new_hist, double *__restrict__ batch,
									 double *__restrict__ hbatch) {
	const StreamDesc d = desc[blockIdx.y];
	const int n = d.keep + d.n_new;
	for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
		const double v = j < d.keep ? old_hist[d.src_off + j] : chunk[d.chunk_off + (j - d.keep)];
		new_hist[d.dst_off + j] = v;
		batch[d.batch_off + j] = v;
		if (hbatch != batch && j < d.h_len) hbatch[d.hbatch_off + j] = v;
	}
}

// committed frames of every stream: F0 from the window's contour, absolute time as the reference forms it
// (i * frame_period / 1000, reference src/harvest.cpp:189), window-relative time for CheapTrick's sample origin
__global__ void stream_commit_kernel(const StreamDesc *__restrict__ desc, const double *__restrict__ win_tpos,
									 const double *__restrict__ win_f0, double frame_period, double *__restrict__ tpos_abs,
									 double *__restrict__ tpos_rel, double *__restrict__ f0) {
	const StreamDesc d = desc[blockIdx.y];
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= d.count) return;
	const long long src = d.hf_off + d.row0 + i;
	f0[d.out_off + i] = win_f0[src];
	tpos_rel[d.out_off + i] = win_tpos[src];
	tpos_abs[d.out_off + i] = (double)(d.first_frame + i) * frame_period / 1000.0;
}

}  // namespace wc

using namespace wc;

struct wc_stream {
	int fs, n_streams, fp_ms, chunk_ms, back_ms, ahead_ms, align_ms, decim;
	int chunk_s, win_s;  // samples of a chunk / of the longest history
	double frame_period;
	Device *dev;
	wc_harvest *hv;
	wc_cheaptrick *ct;
	int fft_size;
	// per stream, host side
	std::vector<long long> n_recv, hist_start, next_frame;  // samples received, absolute sample index of the history start, next frame to commit
	std::vector<int> hist_len, parity;                         // samples in the history, which of the two history buffers holds it
	std::vector<char> closed;
	std::vector<uint64_t> rng_pos;
	DevBuf hist[2], batch, hbatch, win_tpos, win_f0, tpos_rel, desc;
	HostBuf h_desc;
};

static long long gcd_ll(long long a, long long b) { return b ? gcd_ll(b, a % b) : a; }

extern "C" {

wc_stream *wc_stream_create(int fs, int n_streams, double frame_period_ms, int chunk_ms, int lookback_ms, int lookahead_ms,
							double harvest_f0_floor, double harvest_f0_ceil, double q1, double cheaptrick_f0_floor, int fft_size) {
	const int fp = (int)frame_period_ms;
	if (fs <= 0 || n_streams <= 0 || fp < 1 || (double)fp != frame_period_ms || chunk_ms <= 0 || lookback_ms < 0 || lookahead_ms < 0) {
		set_error("stream: bad argument (frame period must be a whole number of ms)");
		return nullptr;
	}
	const int decim = std::max(1, std::min(12, (int)(fs / 8000.0 + 0.5)));  // Harvest's ratio at target_fs 8 kHz, reference src/harvest.cpp:78
	if (fs % 1000 != 0 || (fs / 1000) % decim != 0) {
		set_error("stream: fs must be a multiple of 1000 Hz whose samples per ms are a multiple of Harvest's decimation ratio");
		return nullptr;
	}
	const int align = (int)(8ll * fp / gcd_ll(8, fp));
	if (chunk_ms % align || lookback_ms % align || lookahead_ms % align) {
		set_error("stream: chunk, lookback and lookahead must be multiples of lcm(8 ms, frame period) = " + std::to_string(align) + " ms");
		return nullptr;
	}
	Device *dev = current_device();
	if (!dev) return nullptr;
	wc_stream *s = new wc_stream();
	s->fs = fs; s->n_streams = n_streams; s->fp_ms = fp; s->frame_period = frame_period_ms;
	s->decim = decim;
	s->chunk_ms = chunk_ms; s->back_ms = lookback_ms; s->ahead_ms = lookahead_ms; s->align_ms = align;
	s->chunk_s = fs / 1000 * chunk_ms;
	s->win_s = fs / 1000 * (lookback_ms + chunk_ms + lookahead_ms);
	s->dev = dev;
	s->hv = wc_harvest_create(fs, harvest_f0_floor, harvest_f0_ceil, frame_period_ms, 8000.0, 40.0, 0);
	s->ct = s->hv ? wc_cheaptrick_create(fs, q1, cheaptrick_f0_floor, fft_size) : nullptr;
	if (!s->ct) {
		std::string err = wc_last_error();
		wc_harvest_destroy(s->hv);
		delete s;
		set_error(err);
		return nullptr;
	}
	s->fft_size = wc_cheaptrick_get_fft_size(s->ct);
	s->n_recv.assign(n_streams, 0); s->hist_start.assign(n_streams, 0); s->next_frame.assign(n_streams, 0);
	s->hist_len.assign(n_streams, 0); s->parity.assign(n_streams, 0); s->closed.assign(n_streams, 0);
	s->rng_pos.assign(n_streams, 0);
	return s;
}

void wc_stream_destroy(wc_stream *s) {
	if (!s) return;
	s->dev->quiesce();
	wc_cheaptrick_destroy(s->ct);
	wc_harvest_destroy(s->hv);
	for (DevBuf *b : {&s->hist[0], &s->hist[1], &s->batch, &s->hbatch, &s->win_tpos, &s->win_f0, &s->tpos_rel, &s->desc}) b->release();
	s->h_desc.release();
	delete s;
}

int wc_stream_get_fft_size(const wc_stream *s) { return s ? s->fft_size : WC_ERR_INVALID; }
int wc_stream_chunk_samples(const wc_stream *s) { return s ? s->chunk_s : WC_ERR_INVALID; }
int wc_stream_max_frames_per_push(const wc_stream *s) {
	return s ? (s->ahead_ms + s->chunk_ms) / s->fp_ms + 2 : WC_ERR_INVALID;
}
long long wc_stream_frames_committed(const wc_stream *s, int u) { return (s && u >= 0 && u < s->n_streams) ? s->next_frame[u] : -1; }
long long wc_stream_samples_received(const wc_stream *s, int u) { return (s && u >= 0 && u < s->n_streams) ? s->n_recv[u] : -1; }

unsigned long long wc_stream_rng_position(const wc_stream *s, int u) { return (s && u >= 0 && u < s->n_streams) ? s->rng_pos[u] : 0ull; }
int wc_stream_set_rng_position(wc_stream *s, int u, unsigned long long position) {
	if (!s || u < 0 || u >= s->n_streams) return fail(WC_ERR_INVALID, "stream: bad stream index");
	DeviceLock lock(s->dev);
	s->rng_pos[u] = position;
	return WC_OK;
}

int wc_stream_reset(wc_stream *s, int u) {
	if (!s || u < 0 || u >= s->n_streams) return fail(WC_ERR_INVALID, "stream reset: bad stream index");
	DeviceLock lock(s->dev);
	s->n_recv[u] = 0; s->hist_start[u] = 0; s->next_frame[u] = 0; s->hist_len[u] = 0; s->closed[u] = 0; s->rng_pos[u] = 0;
	return WC_OK;
}

int wc_stream_push_device(wc_stream *s, const double *d_chunk, const int *n_new, const int *flush, double *d_tpos, double *d_f0,
						  double *d_sp, int *frames_out) {
	if (!s || !d_chunk || !d_tpos || !d_f0 || !d_sp || !frames_out) return fail(WC_ERR_INVALID, "stream push: null argument");
	WC_HIP(hipSetDevice(s->dev->id));
	DeviceLock lock(s->dev);
	hipStream_t st = s->dev->active();
	const int n = s->n_streams, spm = s->fs / 1000, align_s = spm * s->align_ms;
	// ---- bookkeeping on the host: which streams take part, what their windows are, which frames they commit ----
	std::vector<int> act;       // streams that take part in this push
	std::vector<StreamDesc> desc;
	std::vector<int> win_len, h_len, count;
	long long chunk_off = 0, batch_off = 0, hbatch_off = 0, hf_off = 0, out_off = 0;
	for (int u = 0; u < n; ++u) {
		frames_out[u] = 0;
		const int nn = n_new ? n_new[u] : s->chunk_s;
		const bool fl = flush && flush[u];
		if (nn < 0 || nn > s->chunk_s) return fail(WC_ERR_INVALID, "stream push: n_new out of range");
		if (nn != s->chunk_s && nn != 0 && !fl) return fail(WC_ERR_INVALID, "stream push: a short chunk is only allowed together with flush");
		if (nn == 0 && !fl) continue;
		if (s->closed[u]) return fail(WC_ERR_INVALID, "stream push: stream was flushed; wc_stream_reset it first");
		if (s->hist_len[u] + nn <= 0) { chunk_off += nn; continue; }  // flush of a stream that never got a sample
		StreamDesc d;
		std::memset(&d, 0, sizeof(d));
		int drop = 0;
		if (s->hist_len[u] + nn > s->win_s) drop = ((s->hist_len[u] + nn - s->win_s + align_s - 1) / align_s) * align_s;
		if (drop > s->hist_len[u]) return fail(WC_ERR_INVALID, "stream push: internal window arithmetic");
		d.keep = s->hist_len[u] - drop;
		d.n_new = nn;
		d.src_off = (long long)u * s->win_s + drop;
		d.dst_off = (long long)u * s->win_s;
		d.chunk_off = chunk_off;
		d.batch_off = batch_off;
		const int wl = d.keep + nn;
		const long long start = s->hist_start[u] + drop, recv = s->n_recv[u] + nn;
		// Harvest's decimator aligns its sampling phase to the END of what it is given (reference src/world_matlabfunctions.cpp:201-206:
		// nbeg = length mod ratio), so the contour of a whole-utterance call depends on (total length mod ratio) -- which a stream
		// cannot know in advance.  Every window Harvest sees is a multiple of the ratio long: full chunks are, and of a final
		// short chunk the last (length mod ratio) samples are left to CheapTrick alone.
		d.h_len = wl - wl % s->decim;
		d.hbatch_off = hbatch_off;
		if (d.h_len < 3 * spm) return fail(WC_ERR_INVALID, "stream push: a stream needs at least 3 ms of signal");
		const int L = wc_get_samples(s->fs, d.h_len, s->frame_period);
		// frames committed: all whose time lies more than `lookahead` before the newest sample; everything on a flush
		long long c1;
		// (Harvest::getSamples, reference src/harvest.cpp:173-181, in integer arithmetic: samples per ms and frame period are whole
		// numbers here, and a stream may have received more samples than an int holds)
		if (fl) c1 = (recv - recv % s->decim) / ((long long)spm * s->fp_ms) + 1;
		else c1 = (recv / spm - s->ahead_ms) / s->fp_ms;  // frames k with k * fp < T - lookahead (T a whole number of ms here)
		if (!fl && recv / spm < s->ahead_ms) c1 = 0;
		if (c1 < s->next_frame[u]) c1 = s->next_frame[u];
		d.first_frame = s->next_frame[u];
		d.count = (int)(c1 - s->next_frame[u]);
		d.row0 = (int)(s->next_frame[u] - start / spm / s->fp_ms);
		if (d.count > 0 && (d.row0 < 0 || d.row0 + d.count > L)) return fail(WC_ERR_INVALID, "stream push: committed frames outside the window");
		if (d.count > wc_stream_max_frames_per_push(s)) return fail(WC_ERR_INVALID, "stream push: more frames than a push may commit");
		d.hf_off = hf_off;
		d.out_off = out_off;
		desc.push_back(d);
		act.push_back(u);
		win_len.push_back(wl);
		h_len.push_back(d.h_len);
		count.push_back(d.count);
		chunk_off += nn; batch_off += wl; hbatch_off += d.h_len; hf_off += L; out_off += d.count;
		// state after this push
		s->hist_start[u] = start; s->hist_len[u] = wl; s->n_recv[u] = recv; s->next_frame[u] = c1;
		if (fl) s->closed[u] = 1;
	}
	const int na = (int)act.size();
	if (na == 0) return WC_OK;
	int rc;
	const size_t hist_bytes = sizeof(double) * (size_t)n * s->win_s;
	if ((rc = s->hist[0].reserve(hist_bytes)) || (rc = s->hist[1].reserve(hist_bytes))) return rc;
	if ((rc = s->batch.reserve(sizeof(double) * (size_t)batch_off))) return rc;
	const bool cut = hbatch_off != batch_off;  // some window ends on a short final chunk
	if (cut && (rc = s->hbatch.reserve(sizeof(double) * (size_t)hbatch_off))) return rc;
	double *d_hbatch = cut ? s->hbatch.as<double>() : s->batch.as<double>();
	if ((rc = s->win_tpos.reserve(sizeof(double) * (size_t)hf_off)) || (rc = s->win_f0.reserve(sizeof(double) * (size_t)hf_off))) return rc;
	if ((rc = s->tpos_rel.reserve(sizeof(double) * (size_t)std::max<long long>(out_off, 1)))) return rc;
	if ((rc = s->desc.reserve(sizeof(StreamDesc) * na)) || (rc = s->h_desc.reserve(sizeof(StreamDesc) * na))) return rc;
	// ---- histories: every participating stream moves to its other buffer; the two parities are handled by two launches ----
	std::memcpy(s->h_desc.p, desc.data(), sizeof(StreamDesc) * na);
	WC_HIP(hipMemcpyAsync(s->desc.p, s->h_desc.p, sizeof(StreamDesc) * na, hipMemcpyHostToDevice, st));
	if ((rc = s->h_desc.mark(st))) return rc;
	{
		// streams are grouped by parity so that each launch reads one buffer and writes the other
		std::vector<int> order(na);
		std::iota(order.begin(), order.end(), 0);
		int n0 = 0;
		for (int a = 0; a < na; ++a) n0 += s->parity[act[a]] == 0;
		if (n0 != 0 && n0 != na) {
			// mixed parities (after a reset of single streams): bring the odd ones over first with a plain row copy
			for (int a = 0; a < na; ++a) {
				const int u = act[a];
				if (s->parity[u] == 1) {
					WC_HIP(hipMemcpyAsync(s->hist[0].as<double>() + (size_t)u * s->win_s, s->hist[1].as<double>() + (size_t)u * s->win_s,
										  sizeof(double) * s->win_s, hipMemcpyDeviceToDevice, st));
					s->parity[u] = 0;
				}
			}
		}
		const int par = s->parity[act[0]];
		int max_len = 0;
		for (int a = 0; a < na; ++a) max_len = std::max(max_len, win_len[a]);
		dim3 grid((unsigned)std::min(64, (max_len + 255) / 256), (unsigned)na);
		hipLaunchKernelGGL(stream_update_kernel, grid, dim3(256), 0, st, s->desc.as<StreamDesc>(), s->hist[par].as<double>(), d_chunk,
						   s->hist[1 - par].as<double>(), s->batch.as<double>(), d_hbatch);
		WC_HIP(hipGetLastError());
		for (int a = 0; a < na; ++a) s->parity[act[a]] = 1 - par;
	}
	// ---- Harvest on every window (the whole-utterance kernels; host-synchronous) ----
	if ((rc = wc_harvest_compute_device(s->hv, na, d_hbatch, h_len.data(), s->win_tpos.as<double>(), s->win_f0.as<double>())))
		return rc;
	// ---- commit ----
	for (int a = 0; a < na; ++a) frames_out[act[a]] = count[a];
	if (out_off == 0) return WC_OK;
	int max_count = 0;
	for (int a = 0; a < na; ++a) max_count = std::max(max_count, count[a]);
	hipLaunchKernelGGL(stream_commit_kernel, dim3((unsigned)((max_count + 127) / 128), (unsigned)na), dim3(128), 0, st, s->desc.as<StreamDesc>(),
					   s->win_tpos.as<double>(), s->win_f0.as<double>(), s->frame_period, d_tpos, s->tpos_rel.as<double>(), d_f0);
	WC_HIP(hipGetLastError());
	// ---- CheapTrick on the committed frames, noise positions carried per stream ----
	std::vector<uint64_t> pos(na);
	uint64_t lo = ~0ull, hi = 0;
	for (int a = 0; a < na; ++a) {
		pos[a] = s->rng_pos[act[a]];
		if (count[a] > 0) { lo = std::min(lo, pos[a]); hi = std::max(hi, pos[a]); }
	}
	const int bins = s->fft_size / 2 + 1;
	if (hi - lo <= (1ull << 28)) {
		if ((rc = wc_cheaptrick_compute_device(s->ct, na, s->batch.as<double>(), win_len.data(), s->tpos_rel.as<double>(), d_f0, count.data(),
											   d_sp, pos.data())))
			return rc;
	} else {
		// streams whose noise positions lie further apart than one draw table covers (one of them was reset hours after the
		// others started): one call per stream, each with its own stretch of the table
		for (int a = 0; a < na; ++a) {
			if (count[a] == 0) continue;
			if ((rc = wc_cheaptrick_compute_device(s->ct, 1, s->batch.as<double>() + desc[a].batch_off, &win_len[a],
												   s->tpos_rel.as<double>() + desc[a].out_off, d_f0 + desc[a].out_off, &count[a],
												   d_sp + desc[a].out_off * bins, &pos[a])))
				return rc;
		}
	}
	for (int a = 0; a < na; ++a) s->rng_pos[act[a]] = pos[a];
	return WC_OK;
}

}  // extern "C"
