// Chunked Harvest + CheapTrick for many concurrent streams: include/world_class_stream.h states the semantics (the
// reference has none -- Harvest is non-causal, reference src/harvest.cpp:431-440, :676-703).  This file is the host logic
// around the batched stages: per-stream histories in HBM (ping-pong rows), the window batches handed to Harvest, the
// bookkeeping of which absolute frames a push commits, and the carried noise-stream positions of CheapTrick.
//
// Two ways of getting a window's contour:
//   whole windows (default)   every push runs all of Harvest on every stream's whole history window;
//   incremental               (wc_stream_set_incremental) Harvest's front -- decimation, band-pass, zero crossings, raw
//                             candidates, refinement: local operations, +-`context` ms of signal per 1 ms frame, and 85 % of its
//                             work -- runs on the newest `chunk + 2 context` ms only; the refined candidate and score rows it
//                             yields for the frames that have their full context are appended to a per-stream ring of rows,
//                             and only Harvest's tail (unreliable-candidate test, contour logic, smoothing) runs over the
//                             window, on rows from the ring.
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "wc_stages.hpp"
#include "../../include/world_class_io.h"
#include "../../include/world_class_stream.h"

namespace wc {

struct StreamDesc {
	long long src_off;    // first kept sample of the old history row (row start + dropped samples)
	long long dst_off;    // start of the new history row
	long long chunk_off;  // first new sample in the packed chunk array
	long long batch_off;  // start of this stream's window in the packed batch handed to CheapTrick
	long long hbatch_off; // start of the stretch handed to Harvest (its front) in that packed batch
	int h_skip, h_len;    // the stretch: window samples [h_skip, h_skip + h_len), a multiple of the decimation ratio long
	int keep, n_new;      // old samples kept, new samples appended (window length = keep + n_new)
	// commit: `count` frames from contour row `row0` (packed Harvest output offset hf_off) to packed output offset out_off
	long long hf_off, out_off, first_frame;
	int row0, count;
	long long hist_ms;    // absolute time of the history's first sample, ms (CheapTrick's window-relative frame times)
	// incremental mode, candidate rows ([1 ms frame][7 S]): kept rows of the old ring, new rows from the front's batch
	long long r_old, r_new, r_front, r_tail;  // row offsets: old ring (after dropping), new ring, front batch, tail batch (-1: none)
	int r_keep, r_add;
};

// new history row = kept tail of the old one followed by the new chunk; the same samples also go to the packed batches
__global__ void stream_update_kernel(const StreamDesc *__restrict__ desc, const double *__restrict__ old_hist,
									 const double *__restrict__ chunk, double *__restrict__ new_hist, double *__restrict__ batch,
									 double *__restrict__ hbatch) {
	const StreamDesc d = desc[blockIdx.y];
	const int n = d.keep + d.n_new;
	for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
		const double v = j < d.keep ? old_hist[d.src_off + j] : chunk[d.chunk_off + (j - d.keep)];
		new_hist[d.dst_off + j] = v;
		batch[d.batch_off + j] = v;
		if (hbatch != batch && j >= d.h_skip && j < d.h_skip + d.h_len) hbatch[d.hbatch_off + (j - d.h_skip)] = v;
	}
}

// candidate / score rows of every stream: the kept part of the old ring and the front's new rows go to the new ring and, for
// the streams whose tail runs in this push, to the tail's packed batch
__global__ void stream_rows_kernel(const StreamDesc *__restrict__ desc, int nc, const double *__restrict__ old_c,
								   const double *__restrict__ old_s, const double *__restrict__ front_c, const double *__restrict__ front_s,
								   double *__restrict__ new_c, double *__restrict__ new_s, double *__restrict__ tail_c, double *__restrict__ tail_s) {
	const StreamDesc d = desc[blockIdx.y];
	const long long total = (long long)(d.r_keep + d.r_add) * nc;
	for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long long)gridDim.x * blockDim.x) {
		const long long row = k / nc;
		const int c = (int)(k - row * nc);
		double vc, vs;
		if (row < d.r_keep) {
			vc = old_c[(d.r_old + row) * nc + c];
			vs = old_s[(d.r_old + row) * nc + c];
		} else {
			vc = front_c[(d.r_front + (row - d.r_keep)) * nc + c];
			vs = front_s[(d.r_front + (row - d.r_keep)) * nc + c];
		}
		new_c[(d.r_new + row) * nc + c] = vc;
		new_s[(d.r_new + row) * nc + c] = vs;
		if (d.r_tail >= 0) {
			tail_c[(d.r_tail + row) * nc + c] = vc;
			tail_s[(d.r_tail + row) * nc + c] = vs;
		}
	}
}

// committed frames of every stream: F0 from the window's contour, absolute time as the reference forms it
// (i * frame_period / 1000, reference src/harvest.cpp:189), history-relative time for CheapTrick's sample origin
__global__ void stream_commit_kernel(const StreamDesc *__restrict__ desc, const double *__restrict__ win_f0, double frame_period,
									 double *__restrict__ tpos_abs, double *__restrict__ tpos_rel, double *__restrict__ f0) {
	const StreamDesc d = desc[blockIdx.y];
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= d.count) return;
	f0[d.out_off + i] = win_f0[d.hf_off + d.row0 + i];
	const long long k = d.first_frame + i;
	tpos_abs[d.out_off + i] = (double)k * frame_period / 1000.0;
	// whole milliseconds from the history's first sample: CheapTrick only derives a sample index from it (reference
	// src/cheaptrick.cpp:150, matlab_round(t fs + 0.001))
	tpos_rel[d.out_off + i] = (double)(k * (long long)frame_period - d.hist_ms) / 1000.0;
}

}  // namespace wc

using namespace wc;

struct wc_stream {
	int fs, n_streams, fp_ms, chunk_ms, back_ms, ahead_ms, align_ms, decim;
	int ctx_ms;          // > 0: incremental mode
	int chunk_s, win_s;  // samples of a chunk / of the longest history
	int row_cap;         // rows of one stream's ring
	bool started;
	double frame_period;
	Device *dev;
	wc_harvest *hv, *hv_front, *hv_tail;
	wc_cheaptrick *ct;
	int fft_size, nc;
	// per stream, host side
	std::vector<long long> n_recv, hist_start, next_frame;  // samples received, absolute sample index of the history start, next frame to commit
	std::vector<long long> rows_start;                         // absolute 1 ms frame of the ring's first row
	std::vector<int> hist_len, parity, rows_len, rparity;      // samples in the history / rows in the ring, which ping-pong buffer holds them
	std::vector<char> closed;
	std::vector<uint64_t> rng_pos;
	DevBuf hist[2], rows_c[2], rows_s[2], batch, hbatch, win_tpos, win_f0, tpos_rel, desc, chunk_f64;
	HostBuf h_desc;
	// harvest option copies for the handles created on demand
	double hv_floor, hv_ceil;
};

// A push that fails half way (bad argument for a later stream, a device error) leaves every stream where it was.
struct StreamStateGuard {
	wc_stream *s;
	std::vector<long long> n_recv, hist_start, next_frame, rows_start;
	std::vector<int> hist_len, parity, rows_len, rparity;
	std::vector<char> closed;
	std::vector<uint64_t> rng_pos;
	bool keep = false;
	explicit StreamStateGuard(wc_stream *st)
		: s(st), n_recv(st->n_recv), hist_start(st->hist_start), next_frame(st->next_frame), rows_start(st->rows_start), hist_len(st->hist_len),
		  parity(st->parity), rows_len(st->rows_len), rparity(st->rparity), closed(st->closed), rng_pos(st->rng_pos) {}
	~StreamStateGuard() {
		if (keep) return;
		s->n_recv = n_recv; s->hist_start = hist_start; s->next_frame = next_frame; s->rows_start = rows_start; s->hist_len = hist_len;
		s->parity = parity; s->rows_len = rows_len; s->rparity = rparity; s->closed = closed; s->rng_pos = rng_pos;
	}
};

static long long gcd_ll(long long a, long long b) { return b ? gcd_ll(b, a % b) : a; }
static long long floor_to(long long v, long long unit) { return v <= 0 ? 0 : v / unit * unit; }

// brings the active streams of a ping-pong pair to one parity (single streams change sides after a reset) and returns it
template <class Copy>
static int common_parity(std::vector<int> &parity, const std::vector<int> &act, Copy copy_row) {
	int n0 = 0;
	for (int u : act) n0 += parity[u] == 0;
	if (n0 != 0 && n0 != (int)act.size())
		for (int u : act)
			if (parity[u] == 1) {
				if (copy_row(u)) return -1;
				parity[u] = 0;
			}
	return parity[act[0]];
}

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
	s->decim = decim; s->ctx_ms = 0; s->started = false;
	s->chunk_ms = chunk_ms; s->back_ms = lookback_ms; s->ahead_ms = lookahead_ms; s->align_ms = align;
	s->chunk_s = fs / 1000 * chunk_ms;
	s->win_s = fs / 1000 * (lookback_ms + chunk_ms + lookahead_ms);
	s->row_cap = 0;
	s->dev = dev;
	s->hv_floor = harvest_f0_floor; s->hv_ceil = harvest_f0_ceil;
	s->hv_front = s->hv_tail = nullptr;
	s->hv = wc_harvest_create(fs, harvest_f0_floor, harvest_f0_ceil, frame_period_ms, 8000.0, 40.0, 0);
	if (s->hv && !hv_exact_twin(s->hv)) { wc_harvest_destroy(s->hv); s->hv = nullptr; }  // (built now rather than inside the first push that meets a tie)
	s->ct = s->hv ? wc_cheaptrick_create(fs, q1, cheaptrick_f0_floor, fft_size) : nullptr;
	if (!s->ct) {
		std::string err = wc_last_error();
		wc_harvest_destroy(s->hv);
		delete s;
		set_error(err);
		return nullptr;
	}
	s->fft_size = wc_cheaptrick_get_fft_size(s->ct);
	s->nc = hv_row_width(s->hv);
	s->n_recv.assign(n_streams, 0); s->hist_start.assign(n_streams, 0); s->next_frame.assign(n_streams, 0);
	s->rows_start.assign(n_streams, 0); s->rows_len.assign(n_streams, 0); s->rparity.assign(n_streams, 0);
	s->hist_len.assign(n_streams, 0); s->parity.assign(n_streams, 0); s->closed.assign(n_streams, 0);
	s->rng_pos.assign(n_streams, 0);
	return s;
}

void wc_stream_destroy(wc_stream *s) {
	if (!s) return;
	s->dev->quiesce();
	wc_cheaptrick_destroy(s->ct);
	wc_harvest_destroy(s->hv);
	wc_harvest_destroy(s->hv_front);
	wc_harvest_destroy(s->hv_tail);
	for (DevBuf *b : {&s->hist[0], &s->hist[1], &s->rows_c[0], &s->rows_c[1], &s->rows_s[0], &s->rows_s[1], &s->batch, &s->hbatch, &s->win_tpos,
					  &s->win_f0, &s->tpos_rel, &s->desc, &s->chunk_f64})
		b->release();
	s->h_desc.release();
	delete s;
}

int wc_stream_set_incremental(wc_stream *s, int context_ms) {
	if (!s) return fail(WC_ERR_INVALID, "stream: null handle");
	DeviceLock lock(s->dev);
	if (s->started) return fail(WC_ERR_INVALID, "stream: the mode can only be chosen before the first push");
	const int spm = s->fs / 1000;
	if (context_ms == 0) {
		s->ctx_ms = 0;
		s->win_s = spm * (s->back_ms + s->chunk_ms + s->ahead_ms);
		return WC_OK;
	}
	if (context_ms < 0 || context_ms % s->align_ms) return fail(WC_ERR_INVALID, "stream: context must be a multiple of lcm(8 ms, frame period)");
	if (context_ms > s->ahead_ms) return fail(WC_ERR_INVALID, "stream: the context is part of the lookahead and cannot exceed it");
	if (!s->hv_front) {
		OnDeviceOf here(s->dev);
		s->hv_front = wc_harvest_create(s->fs, s->hv_floor, s->hv_ceil, s->frame_period, 8000.0, 40.0, 0);
		s->hv_tail = s->hv_front ? wc_harvest_create(s->fs, s->hv_floor, s->hv_ceil, s->frame_period, 8000.0, 40.0, 0) : nullptr;
		if (!s->hv_tail) {  // (both or neither: a later call must not find a front without its tail)
			if (s->hv_front) wc_harvest_destroy(s->hv_front);
			s->hv_front = nullptr;
			return fail(WC_ERR_DEVICE, "stream: the incremental mode's Harvest handles could not be created");
		}
	}
	{
		hv_set_phases(s->hv_front, 1);
		hv_set_phases(s->hv_tail, 2);
		// (the handle that re-runs windows on a tie with the band-pass as direct FIR sums: built here, not inside the first push that
		// meets a tie -- its tables are milliseconds of host work and synchronous uploads, a latency spike in a real-time stream)
		if (!hv_exact_twin(s->hv_front)) return fail(WC_ERR_DEVICE, "stream: the incremental mode's Harvest handles could not be created");
	}
	s->ctx_ms = context_ms;
	// samples kept: what the front needs (chunk + 2 context) and what CheapTrick needs behind the oldest uncommitted frame
	const int need = std::max(s->chunk_ms + 2 * context_ms, s->ahead_ms + s->chunk_ms + 40) + s->align_ms;
	s->win_s = spm * ((need + s->align_ms - 1) / s->align_ms * s->align_ms);
	s->row_cap = s->back_ms + s->chunk_ms + s->ahead_ms + 2 * s->align_ms + 8;
	return WC_OK;
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
	s->rows_start[u] = 0; s->rows_len[u] = 0;
	return WC_OK;
}

int wc_stream_push_device(wc_stream *s, const double *d_chunk, const int *n_new, const int *flush, double *d_tpos, double *d_f0,
						  double *d_sp, int *frames_out) {
	if (!s || !d_chunk || !d_tpos || !d_f0 || !d_sp || !frames_out) return fail(WC_ERR_INVALID, "stream push: null argument");
	WC_HIP(hipSetDevice(s->dev->id));
	DeviceLock lock(s->dev);
	hipStream_t st = s->dev->active();
	s->started = true;
	StreamStateGuard guard(s);
	const bool inc = s->ctx_ms > 0;
	const int n = s->n_streams, spm = s->fs / 1000, align_s = spm * s->align_ms, nc = s->nc;
	// ---- bookkeeping on the host: which streams take part, what their windows are, which frames they commit ----
	std::vector<int> act;         // streams that take part in this push
	std::vector<StreamDesc> desc;
	std::vector<int> win_len, count;
	std::vector<int> front_len;   // samples per entry of Harvest's (front) batch
	std::vector<int> tail_len;    // incremental: pseudo sample counts of the tail batch's entries ((rows - 1) x samples per ms)
	long long chunk_off = 0, batch_off = 0, hbatch_off = 0, hf_off = 0, out_off = 0, front_rows = 0, tail_rows = 0;
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
		d.r_tail = -1;
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
		d.hist_ms = start / spm;
		// Harvest's decimator aligns its sampling phase to the END of what it is given (reference src/world_matlabfunctions.cpp:201-206:
		// nbeg = length mod ratio), so the contour of a whole-utterance call depends on (total length mod ratio) -- which a stream
		// cannot know in advance.  Every stretch Harvest sees ends on a multiple of the ratio: full chunks do, and of a final
		// short chunk the last (length mod ratio) samples are left to CheapTrick alone.
		const long long recv_h = recv - recv % s->decim;
		// frames committed: all whose time lies more than `lookahead` before the newest sample; everything on a flush
		// (Harvest::getSamples, reference src/harvest.cpp:173-181, in integer arithmetic: samples per ms and frame period are whole
		// numbers here, and a stream may have received more samples than an int holds)
		long long c1;
		if (fl) c1 = recv_h / ((long long)spm * s->fp_ms) + 1;
		else c1 = (recv / spm - s->ahead_ms) / s->fp_ms;  // frames k with k * fp < T - lookahead (T a whole number of ms here)
		if (!fl && recv / spm < s->ahead_ms) c1 = 0;
		if (c1 < s->next_frame[u]) c1 = s->next_frame[u];
		d.first_frame = s->next_frame[u];
		d.count = (int)(c1 - s->next_frame[u]);
		if (d.count > wc_stream_max_frames_per_push(s)) return fail(WC_ERR_INVALID, "stream push: more frames than a push may commit");
		d.out_off = out_off;
		if (!inc) {
			// ---- whole windows: Harvest sees the history from its first sample ----
			d.h_skip = 0;
			d.h_len = (int)(recv_h - start);
			d.hbatch_off = hbatch_off;
			if (d.h_len < 3 * spm) return fail(WC_ERR_INVALID, "stream push: a stream needs at least 3 ms of signal");
			const int L = wc_get_samples(s->fs, d.h_len, s->frame_period);
			d.row0 = (int)(s->next_frame[u] - start / spm / s->fp_ms);
			if (d.count > 0 && (d.row0 < 0 || d.row0 + d.count > L)) return fail(WC_ERR_INVALID, "stream push: committed frames outside the window");
			d.hf_off = hf_off;
			hf_off += L;
			front_len.push_back(d.h_len);
			hbatch_off += d.h_len;
		} else {
			// ---- incremental: new candidate rows from a short front window, the contour from the ring of rows ----
			const long long r_prev = s->rows_start[u] + s->rows_len[u];  // first 1 ms frame without rows
			long long r_new = fl ? recv_h / spm + 1 : recv / spm - s->ctx_ms;  // rows exist for the frames that have their full right context
			if (r_new < r_prev) r_new = r_prev;
			d.r_add = (int)(r_new - r_prev);
			if (d.r_add > 0) {
				const long long fw0_ms = floor_to(r_prev - s->ctx_ms, s->align_ms);  // the front window starts `context` before the first new row
				if (fw0_ms * spm < start) return fail(WC_ERR_INVALID, "stream push: the history no longer holds the front's context");
				d.h_skip = (int)(fw0_ms * spm - start);
				d.h_len = (int)(recv_h - fw0_ms * spm);
				if (d.h_len < 3 * spm) return fail(WC_ERR_INVALID, "stream push: a stream needs at least 3 ms of signal");
				d.hbatch_off = hbatch_off;
				const int L1f = wc_get_samples(s->fs, d.h_len, 1.0);
				const long long r0 = r_prev - fw0_ms;
				if (r0 < 0 || r0 + d.r_add > L1f) return fail(WC_ERR_INVALID, "stream push: new rows outside the front window");
				d.r_front = front_rows + r0;
				front_rows += L1f;
				front_len.push_back(d.h_len);
				hbatch_off += d.h_len;
			}
			// the tail's window of rows: from `lookback` before the oldest uncommitted frame (on the 8 ms / frame-period grid, so
			// that the smoothing filter's start phase falls where it falls in the whole contour) to the newest row
			long long tw0 = floor_to(s->next_frame[u] * s->fp_ms - s->back_ms, s->align_ms);
			if (tw0 < s->rows_start[u]) tw0 = s->rows_start[u];
			const long long r_drop = tw0 - s->rows_start[u];
			d.r_keep = (int)(s->rows_len[u] - r_drop);
			if (d.r_keep < 0) return fail(WC_ERR_INVALID, "stream push: internal row arithmetic");
			const int lw = d.r_keep + d.r_add;
			if (lw > s->row_cap) return fail(WC_ERR_INVALID, "stream push: row ring too small");
			d.r_old = (long long)u * s->row_cap + r_drop;
			d.r_new = (long long)u * s->row_cap;
			if (d.count > 0) {
				if (lw < 3) return fail(WC_ERR_INVALID, "stream push: a stream needs at least 3 ms of signal");
				d.r_tail = tail_rows;
				tail_rows += lw;
				tail_len.push_back((lw - 1) * spm);  // Harvest::getSamples of this length is lw rows
				const int L = wc_get_samples(s->fs, (lw - 1) * spm, s->frame_period);
				d.row0 = (int)(s->next_frame[u] - tw0 / s->fp_ms);
				if (d.row0 < 0 || d.row0 + d.count > L) return fail(WC_ERR_INVALID, "stream push: committed frames outside the window");
				d.hf_off = hf_off;
				hf_off += L;
			}
			s->rows_start[u] = tw0;
			s->rows_len[u] = lw;
		}
		desc.push_back(d);
		act.push_back(u);
		win_len.push_back(wl);
		count.push_back(d.count);
		chunk_off += nn; batch_off += wl; out_off += d.count;
		// state after this push
		s->hist_start[u] = start; s->hist_len[u] = wl; s->n_recv[u] = recv; s->next_frame[u] = c1;
		if (fl) s->closed[u] = 1;
	}
	const int na = (int)act.size();
	if (na == 0) { guard.keep = true; return WC_OK; }
	int rc;
	const size_t hist_bytes = sizeof(double) * (size_t)n * s->win_s;
	if ((rc = s->hist[0].reserve(hist_bytes)) || (rc = s->hist[1].reserve(hist_bytes))) return rc;
	if ((rc = s->batch.reserve(sizeof(double) * (size_t)batch_off))) return rc;
	const bool own_hbatch = inc || hbatch_off != batch_off;  // whole windows: only when some window ends on a short final chunk
	if (own_hbatch && (rc = s->hbatch.reserve(sizeof(double) * (size_t)std::max<long long>(hbatch_off, 1)))) return rc;
	double *d_hbatch = own_hbatch ? s->hbatch.as<double>() : s->batch.as<double>();
	if ((rc = s->win_tpos.reserve(sizeof(double) * (size_t)std::max<long long>(std::max(hf_off, front_rows), 1))) ||
		(rc = s->win_f0.reserve(sizeof(double) * (size_t)std::max<long long>(std::max(hf_off, front_rows), 1))))
		return rc;
	if ((rc = s->tpos_rel.reserve(sizeof(double) * (size_t)std::max<long long>(out_off, 1)))) return rc;
	if ((rc = s->desc.reserve(sizeof(StreamDesc) * na)) || (rc = s->h_desc.reserve(sizeof(StreamDesc) * na))) return rc;
	std::memcpy(s->h_desc.p, desc.data(), sizeof(StreamDesc) * na);
	WC_HIP(hipMemcpyAsync(s->desc.p, s->h_desc.p, sizeof(StreamDesc) * na, hipMemcpyHostToDevice, st));
	if ((rc = s->h_desc.mark(st))) return rc;
	// ---- histories: every participating stream moves to its other buffer ----
	{
		const int par = common_parity(s->parity, act, [&](int u) -> int {
			WC_HIP(hipMemcpyAsync(s->hist[0].as<double>() + (size_t)u * s->win_s, s->hist[1].as<double>() + (size_t)u * s->win_s,
								  sizeof(double) * s->win_s, hipMemcpyDeviceToDevice, st));
			return WC_OK;
		});
		if (par < 0) return WC_ERR_DEVICE;
		int max_len = 0;
		for (int a = 0; a < na; ++a) max_len = std::max(max_len, win_len[a]);
		dim3 grid((unsigned)std::min(64, (max_len + 255) / 256), (unsigned)na);
		hipLaunchKernelGGL(stream_update_kernel, grid, dim3(256), 0, st, s->desc.as<StreamDesc>(), s->hist[par].as<double>(), d_chunk,
						   s->hist[1 - par].as<double>(), s->batch.as<double>(), d_hbatch);
		WC_HIP(hipGetLastError());
		for (int a = 0; a < na; ++a) s->parity[act[a]] = 1 - par;
	}
	if (!inc) {
		// ---- Harvest on every window (the whole-utterance kernels; host-synchronous) ----
		if ((rc = wc_harvest_compute_device(s->hv, na, d_hbatch, front_len.data(), s->win_tpos.as<double>(), s->win_f0.as<double>()))) return rc;
	} else {
		// ---- front on the short windows, rows into the rings and the tail's batch, tail on the rows ----
		if (!front_len.empty() &&
			(rc = wc_harvest_compute_device(s->hv_front, (int)front_len.size(), d_hbatch, front_len.data(), s->win_tpos.as<double>(), s->win_f0.as<double>())))
			return rc;
		const size_t ring_bytes = sizeof(double) * (size_t)n * s->row_cap * nc;
		for (int k = 0; k < 2; ++k)
			if ((rc = s->rows_c[k].reserve(ring_bytes)) || (rc = s->rows_s[k].reserve(ring_bytes))) return rc;
		if ((rc = hv_reserve_rows(s->hv_tail, std::max<long long>(tail_rows, 1)))) return rc;
		if ((rc = hv_reserve_rows(s->hv_front, 1))) return rc;  // (valid pointers even before the front has run once)
		const int par = common_parity(s->rparity, act, [&](int u) -> int {
			const size_t off = (size_t)u * s->row_cap * nc, bytes = sizeof(double) * (size_t)s->row_cap * nc;
			WC_HIP(hipMemcpyAsync(s->rows_c[0].as<double>() + off, s->rows_c[1].as<double>() + off, bytes, hipMemcpyDeviceToDevice, st));
			WC_HIP(hipMemcpyAsync(s->rows_s[0].as<double>() + off, s->rows_s[1].as<double>() + off, bytes, hipMemcpyDeviceToDevice, st));
			return WC_OK;
		});
		if (par < 0) return WC_ERR_DEVICE;
		hipLaunchKernelGGL(stream_rows_kernel, dim3(64, (unsigned)na), dim3(256), 0, st, s->desc.as<StreamDesc>(), nc, s->rows_c[par].as<double>(),
						   s->rows_s[par].as<double>(), hv_candidate_rows(s->hv_front), hv_score_rows(s->hv_front), s->rows_c[1 - par].as<double>(),
						   s->rows_s[1 - par].as<double>(), hv_candidate_rows(s->hv_tail), hv_score_rows(s->hv_tail));
		WC_HIP(hipGetLastError());
		for (int a = 0; a < na; ++a) s->rparity[act[a]] = 1 - par;
		if (!tail_len.empty() &&
			(rc = wc_harvest_compute_device(s->hv_tail, (int)tail_len.size(), s->batch.as<double>(), tail_len.data(), s->win_tpos.as<double>(), s->win_f0.as<double>())))
			return rc;
	}
	// ---- commit ----
	for (int a = 0; a < na; ++a) frames_out[act[a]] = count[a];
	if (out_off == 0) { guard.keep = true; return WC_OK; }
	int max_count = 0;
	for (int a = 0; a < na; ++a) max_count = std::max(max_count, count[a]);
	hipLaunchKernelGGL(stream_commit_kernel, dim3((unsigned)((max_count + 127) / 128), (unsigned)na), dim3(128), 0, st, s->desc.as<StreamDesc>(),
					   s->win_f0.as<double>(), s->frame_period, d_tpos, s->tpos_rel.as<double>(), d_f0);
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
	guard.keep = true;
	return WC_OK;
}

// The same with the new samples as they come off a capture device or a WAV file: chunk_format 0 = float64, 1 = int16 PCM
// (sample / 32768, the reference's wavread scaling), 2 = float32; widened on the device into a staging buffer.
int wc_stream_push_device_fmt(wc_stream *s, const void *d_chunk, int chunk_format, const int *n_new, const int *flush, double *d_tpos,
							  double *d_f0, double *d_sp, int *frames_out) {
	if (!s || !d_chunk) return fail(WC_ERR_INVALID, "stream push: null argument");
	if (chunk_format == 0) return wc_stream_push_device(s, static_cast<const double *>(d_chunk), n_new, flush, d_tpos, d_f0, d_sp, frames_out);
	if (chunk_format != 1 && chunk_format != 2) return fail(WC_ERR_INVALID, "stream push: chunk format must be 0 (float64), 1 (int16 PCM) or 2 (float32)");
	WC_HIP(hipSetDevice(s->dev->id));
	DeviceLock lock(s->dev);
	long long total = 0;
	for (int u = 0; u < s->n_streams; ++u) {
		const int nn = n_new ? n_new[u] : s->chunk_s;
		if (nn < 0 || nn > s->chunk_s) return fail(WC_ERR_INVALID, "stream push: n_new out of range");
		total += nn;
	}
	int rc;
	if ((rc = s->chunk_f64.reserve(sizeof(double) * (size_t)std::max<long long>(total, 1)))) return rc;
	if (chunk_format == 1) rc = wc_pcm16_to_double_device(static_cast<const int16_t *>(d_chunk), total, s->chunk_f64.as<double>());
	else rc = wc_float_to_double_device(static_cast<const float *>(d_chunk), total, s->chunk_f64.as<double>());
	if (rc) return rc;
	return wc_stream_push_device(s, s->chunk_f64.as<double>(), n_new, flush, d_tpos, d_f0, d_sp, frames_out);
}

}  // extern "C"
