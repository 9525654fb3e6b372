// Fused analysis + synthesis pipeline (extension; the reference has no such entry point -- its demo calls the
// four stages one after the other, reference test/test.cpp:288-384).  One call enqueues the whole hot path
// for a packed batch without any host synchronisation in between:
//
//   main stream   Harvest -> CheapTrick counts/scan --E0--> Synthesis time base ............ --(E1,E2)--> pulses
//   stream 1                                        E0 -> CheapTrick frames --E1
//   stream 2                                        E0 -> D4C (LoveTrain, scans, frames) --E2
//
// CheapTrick, D4C and the Synthesis time base only depend on the F0 contour, so they overlap; the noise-stream
// positions are chained on the device (CheapTrick end -> D4C start -> Synthesis start), exactly the order a
// single reference process would consume its global randn() stream.  Capacity overflows of the rate-bounded
// buffers (Harvest zero crossings, Synthesis pulses) are checked once at the end and re-run with hard bounds.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "wc_stages.hpp"

using namespace wc;

struct wc_pipeline {
	int fs, fft_size;
	double frame_period;
	Device *dev;
	wc_harvest *hv[4];  // the batch is split into up to 4 contiguous groups of utterances, one Harvest chain per stream:
	hipStream_t hs[4];  // the latency-bound Harvest kernels of one group overlap the ALU-bound ones of the others
	hipEvent_t he[4];
	int n_split;
	wc_cheaptrick *ct;
	wc_d4c *d4;
	wc_synthesis *sy;
	hipStream_t s1, s2;
	hipEvent_t e0, e1, e2;
};

extern "C" {

wc_pipeline *wc_pipeline_create(int fs, double frame_period, double harvest_f0_floor, double harvest_f0_ceil, double q1,
								double cheaptrick_f0_floor, int fft_size, double d4c_threshold) {
	Device *dev = current_device();
	if (!dev) return nullptr;
	wc_pipeline *p = new wc_pipeline();
	std::memset(p, 0, sizeof(*p));
	p->fs = fs;
	p->frame_period = frame_period;
	p->dev = dev;
	{
		const char *env = getenv("WC_PIPELINE_SPLIT");
		p->n_split = env ? atoi(env) : 2;  // measured on MI355X: 2 groups beat 1 and 4
		if (p->n_split < 1) p->n_split = 1;
		if (p->n_split > 4) p->n_split = 4;
	}
	bool hv_ok = true;
	for (int k = 0; k < p->n_split; ++k) {
		p->hv[k] = wc_harvest_create(fs, harvest_f0_floor, harvest_f0_ceil, frame_period, 8000.0, 40.0, 0);
		hv_ok = hv_ok && p->hv[k] != nullptr;
		if (k > 0 && hv_ok) {
			hv_ok = hipStreamCreateWithFlags(&p->hs[k], hipStreamNonBlocking) == hipSuccess &&
					hipEventCreateWithFlags(&p->he[k], hipEventDisableTiming) == hipSuccess;
		}
	}
	p->ct = hv_ok ? wc_cheaptrick_create(fs, q1, cheaptrick_f0_floor, fft_size) : nullptr;
	p->fft_size = p->ct ? wc_cheaptrick_get_fft_size(p->ct) : 0;
	p->d4 = p->ct ? wc_d4c_create(fs, d4c_threshold) : nullptr;
	p->sy = p->d4 ? wc_synthesis_create(fs, p->fft_size, frame_period) : nullptr;
	bool ok = p->sy != nullptr;
	ok = ok && hipStreamCreateWithFlags(&p->s1, hipStreamNonBlocking) == hipSuccess;
	ok = ok && hipStreamCreateWithFlags(&p->s2, hipStreamNonBlocking) == hipSuccess;
	ok = ok && hipEventCreateWithFlags(&p->e0, hipEventDisableTiming) == hipSuccess;
	ok = ok && hipEventCreateWithFlags(&p->e1, hipEventDisableTiming) == hipSuccess;
	ok = ok && hipEventCreateWithFlags(&p->e2, hipEventDisableTiming) == hipSuccess;
	if (!ok) {
		std::string err = wc_last_error();
		void wc_pipeline_destroy(wc_pipeline *);
		wc_pipeline_destroy(p);
		set_error(err.empty() ? "pipeline: stream/event creation failed" : err);
		return nullptr;
	}
	return p;
}

void wc_pipeline_destroy(wc_pipeline *p) {
	if (!p) return;
	if (p->dev) (void)hipStreamSynchronize(p->dev->stream);
	if (p->s1) { (void)hipStreamSynchronize(p->s1); (void)hipStreamDestroy(p->s1); }
	if (p->s2) { (void)hipStreamSynchronize(p->s2); (void)hipStreamDestroy(p->s2); }
	if (p->e0) (void)hipEventDestroy(p->e0);
	if (p->e1) (void)hipEventDestroy(p->e1);
	if (p->e2) (void)hipEventDestroy(p->e2);
	wc_synthesis_destroy(p->sy);
	wc_d4c_destroy(p->d4);
	wc_cheaptrick_destroy(p->ct);
	for (int k = 0; k < 4; ++k) {
		if (p->hs[k]) { (void)hipStreamSynchronize(p->hs[k]); (void)hipStreamDestroy(p->hs[k]); }
		if (p->he[k]) (void)hipEventDestroy(p->he[k]);
		wc_harvest_destroy(p->hv[k]);
	}
	delete p;
}

int wc_pipeline_get_fft_size(const wc_pipeline *p) { return p ? p->fft_size : WC_ERR_INVALID; }

int wc_pipeline_run_device(wc_pipeline *p, int n_utt, const double *d_x, const int *x_length, double *d_tpos, double *d_f0,
						   double *d_sp, double *d_ap, double *d_y, uint64_t *rng_pos) {
	if (!p || n_utt <= 0 || !d_x || !x_length || !d_tpos || !d_f0 || !d_sp || !d_ap || !d_y)
		return fail(WC_ERR_INVALID, "pipeline: null argument");
	WC_HIP(hipSetDevice(p->dev->id));
	Device *dev = p->dev;
	hipStream_t s0 = dev->stream;
	std::vector<int> f_len(n_utt), y_len(n_utt);
	uint64_t lo = ~0ull, hi = 0;
	const int bins = p->fft_size / 2 + 1;
	for (int u = 0; u < n_utt; ++u) {
		if (x_length[u] <= 0) return fail(WC_ERR_INVALID, "pipeline: non-positive x_length");
		f_len[u] = wc_get_samples(p->fs, x_length[u], p->frame_period);
		y_len[u] = wc_synthesis_out_length(f_len[u], p->frame_period, p->fs);
		if (f_len[u] < 2) return fail(WC_ERR_INVALID, "pipeline: utterance shorter than two frames");
		const uint64_t p0 = rng_pos ? rng_pos[u] : 0ull;
		const uint64_t bound = (uint64_t)(2 * (p->fft_size / 2) + 1 + bins) * (uint64_t)f_len[u] + d4c_draw_bound(p->d4, f_len[u]) +
							   (uint64_t)y_len[u];
		lo = p0 < lo ? p0 : lo;
		hi = p0 + bound > hi ? p0 + bound : hi;
	}
	int rc;
	if ((rc = dev->ensure_rng(lo, hi))) return rc;
	bool hv_full = false, syn_full = false;
	const int ns = p->n_split < n_utt ? p->n_split : n_utt;
	for (int attempt = 0; attempt < 3; ++attempt) {
		{
			// group k = utterances [u0, u1); the packed layout makes every group a contiguous slice
			long long xo = 0, fo = 0;
			if (ns > 1) WC_HIP(hipEventRecord(p->e0, s0));  // later groups start after whatever precedes on s0
			for (int k = 0; k < ns; ++k) {
				const int u0 = (int)((long long)n_utt * k / ns), u1 = (int)((long long)n_utt * (k + 1) / ns);
				hipStream_t sk = k == 0 ? s0 : p->hs[k];
				if (k > 0) WC_HIP(hipStreamWaitEvent(sk, p->e0, 0));
				if ((rc = hv_enqueue(p->hv[k], sk, u1 - u0, d_x + xo, x_length + u0, d_tpos + fo, d_f0 + fo, hv_full))) return rc;
				if (k > 0) WC_HIP(hipEventRecord(p->he[k], sk));
				for (int u = u0; u < u1; ++u) { xo += x_length[u]; fo += f_len[u]; }
			}
			for (int k = 1; k < ns; ++k) WC_HIP(hipStreamWaitEvent(s0, p->he[k], 0));
		}
		long long total = 0;
		uint64_t a0 = 0, a1 = 0;
		if ((rc = ct_prepare(p->ct, s0, n_utt, x_length, d_f0, f_len.data(), rng_pos, &total, &a0, &a1))) return rc;
		WC_HIP(hipEventRecord(p->e0, s0));
		WC_HIP(hipStreamWaitEvent(ns > 1 ? p->hs[1] : p->s1, p->e0, 0));
		WC_HIP(hipStreamWaitEvent(p->s2, p->e0, 0));
		// CheapTrick's frames go to the second Harvest stream when there is one: HIP multiplexes streams onto a few
		// hardware queues, and two streams that land on the same queue would serialise CheapTrick and D4C
		hipStream_t sct = ns > 1 ? p->hs[1] : p->s1;
		if ((rc = ct_frames(p->ct, sct, n_utt, d_x, d_tpos, d_f0, d_sp, total))) return rc;
		WC_HIP(hipEventRecord(p->e1, sct));
		if ((rc = d4c_enqueue(p->d4, p->s2, n_utt, d_x, x_length, d_tpos, d_f0, f_len.data(), p->fft_size, d_ap, nullptr,
							  ct_end_positions(p->ct))))
			return rc;
		WC_HIP(hipEventRecord(p->e2, p->s2));
		if ((rc = syn_prepare(p->sy, s0, n_utt, d_f0, f_len.data(), y_len.data(), d_y, nullptr, syn_full))) return rc;
		WC_HIP(hipStreamWaitEvent(s0, p->e1, 0));
		WC_HIP(hipStreamWaitEvent(s0, p->e2, 0));
		if ((rc = syn_pulses(p->sy, s0, d_f0, d_sp, d_ap, d_y, d4c_end_positions(p->d4)))) return rc;
		bool o1 = false, o2 = false;
		if ((rc = syn_finish(p->sy, s0, rng_pos, &o2))) return rc;  // synchronises s0 (and, through E1/E2, s1 and s2)
		for (int k = 0; k < ns; ++k) {
			bool ok = false;
			if ((rc = hv_overflowed(p->hv[k], s0, &ok))) return rc;  // s0 is already idle; the flag copies are tiny
			o1 = o1 || ok;
		}
		if (!o1 && !o2) return WC_OK;
		hv_full = hv_full || o1;
		syn_full = syn_full || o2;
	}
	return fail(WC_ERR_DEVICE, "pipeline: buffer overflow");
}

}  // extern "C"
